"""One-shot all-reduce over xGMI for row-parallel ("in"-split) layers: the device-side state, its exchange between the
ranks of a node (IPC), and a host-memory twin of the protocol for tests.

The kernels (aqlm_amd/csrc/xgmi_reduce.hip) replace ``dist.all_reduce`` behind the shard's matvec: every rank publishes
its slice-summed fp32 vector in a buffer its peers have mapped, raises a flag, and the finalize of every rank reads all
vectors over the point-to-point xGMI links and adds them in rank order (bit-identical replicas of y).  No reference
counterpart (the reference has no tensor parallelism).

State layout per rank (int32 words of one zero-filled device allocation; mirrored by aqlm_hip_xgmi_state_bytes):
    [0, 2 n)            pub[2][n] fp32          n = max_elems
    [2 n, 2 n + 2)      flag[2]
    [2 n + 16, +3)      epoch (starts at 1), publish ticket, reduce ticket
    [2 n + 20, +9)      arrival counters of the publishing matvec (8 workgroup shards + 1)
    [2 n + 32]          status (1 = a peer timed out)
"""
from __future__ import annotations

import ctypes
import time
from typing import List, Optional

import torch
import torch.distributed as dist

EPOCH_OFF, STATUS_OFF = 16, 32


def _exchange(state: torch.Tensor, group) -> List[torch.Tensor]:
    """Every rank's state tensor as seen from this process (own tensor for own rank).  The tensors travel through
    torch.multiprocessing's pickler: CUDA tensors as IPC handles (``hipIpcGetMemHandle``), CPU tensors as shared-memory
    handles -- the receiver's tensor aliases the sender's memory."""
    import pickle
    from multiprocessing.reduction import ForkingPickler

    import torch.multiprocessing  # noqa: F401  (registers the tensor reductions with ForkingPickler)

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        return [state]
    handles: List[Optional[bytes]] = [None] * world
    try:  # a rank that cannot export its state still takes part in the collective (the others must not wait for it)
        mine: Optional[bytes] = bytes(ForkingPickler.dumps(state))
    except Exception:  # noqa: BLE001
        mine = None
    dist.all_gather_object(handles, mine, group=group)
    missing = [r for r, h in enumerate(handles) if h is None]
    if missing:
        raise RuntimeError(f"one-shot all-reduce: ranks {missing} could not export their state over IPC")
    return [state if r == rank else pickle.loads(h) for r, h in enumerate(handles)]


class OneShotAllReduce:
    """Device-side state of the fused finalize + all-reduce for up to ``max_elems`` (= batch x out_features) values.
    Construction is collective (every rank of ``group`` calls it).  All ranks must sit on one node with peer access."""

    def __init__(self, max_elems: int, device: torch.device, group=None, spin_limit: int = 0):
        from . import _native

        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.max_elems = int(max_elems)
        nbytes = _native.lib.aqlm_hip_xgmi_state_bytes(self.max_elems)
        self.state = torch.zeros((nbytes // 4,), dtype=torch.int32, device=device)
        self.state[2 * self.max_elems + EPOCH_OFF] = 1
        torch.cuda.synchronize(device)
        err = None
        try:
            self.peers = _exchange(self.state, group)      # keeps the mappings alive
            for p in self.peers:                            # touch every mapping once: enables peer access in torch
                if p is not self.state:
                    _ = p[2 * self.max_elems + EPOCH_OFF:2 * self.max_elems + EPOCH_OFF + 1].to(device)
        except Exception as e:  # noqa: BLE001 - a rank that cannot map a peer must not leave the others waiting
            err = e
        if self.world > 1:  # collective verdict: either every rank has every mapping, or all of them give up together
            ok = torch.tensor([0 if err else 1], device=device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if not int(ok):
                raise RuntimeError(f"one-shot all-reduce: IPC mapping of the peers' state failed on some rank ({err})")
        elif err:
            raise err
        base = [p.data_ptr() for p in self.peers]
        self._pub = torch.tensor(base, dtype=torch.int64, device=device)
        self._flag = torch.tensor([b + 8 * self.max_elems for b in base], dtype=torch.int64, device=device)
        own = self.state.data_ptr()
        self.xg = _native.Xgmi(self._pub.data_ptr(), self._flag.data_ptr(), own + (2 * self.max_elems + EPOCH_OFF) * 4,
                               own + (2 * self.max_elems + STATUS_OFF) * 4, self.rank, self.world, self.max_elems,
                               int(spin_limit))
        torch.cuda.synchronize(device)
        if self.world > 1:
            dist.barrier(group=group)  # nobody publishes before everybody has mapped everybody

    def finalize(self, partials: torch.Tensor, scales: torch.Tensor, bias: Optional[torch.Tensor], y: torch.Tensor,
                 out_features: int, batch: int, dtype_id: int, stream: int) -> None:
        from . import _native

        rc = _native.lib.aqlm_hip_xgmi_finalize(ctypes.byref(self.xg), partials.data_ptr(), scales.data_ptr(),
                                                None if bias is None else bias.data_ptr(), y.data_ptr(), out_features, batch,
                                                out_features, dtype_id, stream)
        if rc:
            _native.check(rc, "aqlm xgmi finalize")

    def own_pub_flag(self):
        """Addresses of this rank's own pub buffer and flag words (what the publishing matvec writes)."""
        base = self.state.data_ptr()
        return base, base + 8 * self.max_elems

    def reduce(self, scales: torch.Tensor, bias: Optional[torch.Tensor], y: torch.Tensor, out_features: int, batch: int,
               dtype_id: int, stream: int) -> None:
        """The reduce half alone: the shard's matvec has published its totals itself
        (aqlm_hip_gemv_1x16_packed_publish), so the call is one launch."""
        from . import _native

        rc = _native.lib.aqlm_hip_xgmi_finalize(ctypes.byref(self.xg), None, scales.data_ptr(),
                                                None if bias is None else bias.data_ptr(), y.data_ptr(), out_features, batch,
                                                out_features, dtype_id, stream)
        if rc:
            _native.check(rc, "aqlm xgmi reduce")

    def timed_out(self) -> bool:
        """Synchronising read of the status word (diagnostics / tests)."""
        return bool(int(self.state[2 * self.max_elems + STATUS_OFF]))


class HostOneShotAllReduce:
    """The same protocol on host shared memory, in Python: publish (pub[e & 1], then flag[e & 1] = e), poll the peers'
    flags, sum in rank order, bump the epoch.  Test infrastructure for the hand-shake logic (double buffering by epoch
    parity, ranks racing ahead, bit-identical replicas) with the gloo backend -- the device kernels follow it line by line."""

    def __init__(self, max_elems: int, group=None, timeout_s: float = 30.0):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n = int(max_elems)
        self.state = torch.zeros((2 * self.n + 64,), dtype=torch.float32).share_memory_()
        self.peers = _exchange(self.state, group)
        self.epoch = 1
        self.timeout_s = timeout_s
        if self.world > 1:
            dist.barrier(group=group)

    def all_reduce(self, v: torch.Tensor) -> torch.Tensor:
        e, n = self.epoch, self.n
        half = (e & 1) * n
        mine = self.peers[self.rank]
        mine[half:half + v.numel()] = v.to(torch.float32).reshape(-1)
        mine[2 * n + (e & 1)] = float(e)                      # the flag goes up after the payload
        out = torch.zeros(v.numel(), dtype=torch.float32)
        t0 = time.time()
        for r in range(self.world):                           # rank order: every replica adds in the same order
            while int(self.peers[r][2 * n + (e & 1)]) != e:
                if time.time() - t0 > self.timeout_s:
                    raise TimeoutError(f"rank {self.rank}: peer {r} never reached epoch {e}")
                time.sleep(0.0002)
            out += self.peers[r][half:half + v.numel()]
        self.epoch = e + 1
        return out.reshape(v.shape)
