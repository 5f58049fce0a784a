"""The ONE result line of bench.py: nothing else may ever reach the process's stdout, and nothing may follow the line.

Round 5 lost its driver-run record to five lines of librccl banner: the library `printf`s into the C stdio buffer of fd 1, which
is flushed at process exit -- after Python's `print(json.dumps(result))`.  The reference's harness prints one result per run and
nothing after it (benchmark/matmul_benchmark.py:99-125); this module makes that a property of the process instead of a habit:

  * `StdoutGuard.install()` (first thing in main, every rank) duplicates fd 1 to a private descriptor and points fd 1 at
    stderr.  From then on everything any library, C or Python, writes to "stdout" lands on stderr;
  * `emit_final(line)` flushes the C and Python buffers (they drain to stderr), writes the line to the private descriptor with
    `os.write` -- exactly once per process -- and ends the process with `os._exit(0)`: no teardown of a library (RCCL communicators,
    captured graphs) can hang or crash the run after its result is out.  Under a profiler (rocprofv3 writes its kernel stats and
    counter files in exit handlers: `AQLM_BENCH_SOFT_EXIT=1`, `--soft-exit`, or a ROCPROF* variable in the environment) the exit is
    a normal `sys.exit(0)` with a 30 s hard stop behind it; whatever the handlers print goes to stderr.
    Ranks other than 0 never touch the private descriptor.

`ExtrasWatchdog` guards the untimed sections after the timed region the same way: if they hang, rank 0 emits what it has.
"""
import ctypes
import json
import os
import sys
import threading
import time


class StdoutGuard:
    """fd 1 -> stderr for the life of the process; the real stdout is kept on a private descriptor for the one result line."""

    _real_fd = None
    _lock = threading.Lock()
    _emitted = False

    @classmethod
    def install(cls):
        with cls._lock:
            if cls._real_fd is not None:
                return
            sys.stdout.flush()
            cls._real_fd = os.dup(1)
            os.set_inheritable(cls._real_fd, False)
            os.dup2(2, 1)  # every later write to fd 1 (C stdio of librccl / hip, Python's sys.stdout) goes to stderr

    @classmethod
    def installed(cls):
        return cls._real_fd is not None

    @classmethod
    def write_line(cls, line):
        """Write `line` (one JSON document, no newlines inside) to the real stdout, once per process."""
        assert "\n" not in line
        with cls._lock:
            if cls._emitted:
                return False
            cls._emitted = True
            fd = cls._real_fd if cls._real_fd is not None else 1
        data = (line + "\n").encode()
        while data:
            n = os.write(fd, data)
            data = data[n:]
        return True


def flush_all():
    """Drain Python's and the C library's stdio buffers (with the guard installed they drain to stderr)."""
    try:
        sys.stdout.flush()
        sys.stderr.flush()
    except Exception:  # noqa: BLE001
        pass
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass


def dumps(result):
    try:
        return json.dumps(result, default=lambda o: None)
    except Exception:  # noqa: BLE001 - a dict mutated mid-dump (watchdog path): fall back to the headline fields
        return json.dumps({k: v for k, v in result.items() if k not in ("detail", "sharded_70b")}, default=lambda o: None)


def soft_exit_wanted():
    """A profiler around the process needs the exit handlers to run."""
    if os.environ.get("AQLM_BENCH_SOFT_EXIT") == "1":
        return True
    return any(k.startswith(("ROCPROF", "ROCP_TOOL")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def emit_final(result, rank, side_file=None, hard=False):
    """Rank 0: write the full result to `side_file` (best effort), then the line as the only bytes of stdout.  Every rank then
    exits: `os._exit(0)` (also always from the watchdog's thread: `hard`), or normally when a profiler needs the exit handlers."""
    if rank == 0:
        line = dumps(result)
        if side_file:
            try:
                with open(side_file, "w") as f:
                    f.write(line + "\n")
            except OSError:
                pass
        flush_all()
        StdoutGuard.write_line(line)
    flush_all()
    if hard or threading.current_thread() is not threading.main_thread() or not soft_exit_wanted():
        os._exit(0)

    def _hard_stop():
        time.sleep(30.0)
        os._exit(0)

    threading.Thread(target=_hard_stop, daemon=True).start()
    sys.exit(0)


class ExtrasWatchdog:
    """The one JSON line must come out whatever happens after the timed region.  A daemon thread waits `budget_s`; if the main
    thread has not called finish() by then, rank 0 emits the result as it stands (with `extras_timed_out` naming the section that
    was running) and every rank leaves through os._exit -- a rank stuck in a collective cannot be joined."""

    def __init__(self, result, rank, budget_s, side_file=None):
        self.result, self.rank, self.budget_s, self.side_file = result, rank, budget_s, side_file
        self.section = "detail"
        self.lock = threading.Lock()
        self.done = False
        self.fired = False
        if budget_s > 0:
            threading.Thread(target=self._run, daemon=True).start()

    def _run(self):
        time.sleep(self.budget_s)
        with self.lock:
            if self.done:
                return
            self.fired = True
        if self.rank == 0:
            self.result["extras_timed_out"] = {"after_s": self.budget_s, "section": self.section}
        else:
            time.sleep(5.0)  # rank 0 writes first
        emit_final(self.result, self.rank, self.side_file, hard=True)

    def finish(self):
        with self.lock:
            if self.fired:
                time.sleep(3600)  # the watchdog thread is emitting / exiting
                return False
            self.done = True
        return True
