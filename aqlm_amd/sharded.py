"""Row-/column-sharded QuantizedLinear over several MI355X (one process per GPU, torch.distributed over RCCL/xGMI).

New design -- the reference has no tensor parallelism anywhere; BASELINE.json's north star
asks for the 70B layer shapes (8192 -> 28672) to be "sharded row-wise across 8 GPUs with an RCCL all-reduce over xGMI".

Two partitions of one layer  y = (W x) * scales + bias :

  * ``"in"``  (row-parallel, the north star's variant): rank r owns the input-group slice
    ``codes[:, j0:j1, :]`` and consumes ``x[..., j0*g : j1*g]``; codebooks and scales are replicated.  Each rank
    runs the ordinary gemv on its ``[out, in/R]`` slice with the scales applied (the layer is linear in the slice
    contributions) and the bias added on rank 0 only; partial outputs are summed with ONE all-reduce of
    ``batch x out`` elements (56 KiB in fp16 for out = 28672 -- latency-bound on xGMI, not bandwidth-bound).
  * ``"out"`` (column-parallel): rank r owns ``codes[i0:i1]``, ``scales[i0:i1]``, ``bias[i0:i1]``; x and the
    codebooks are replicated; outputs are concatenated with an all-gather (or left sharded when the next layer is
    in-split -- the Megatron pairing).  Bit-identical to the single-GPU result.

The per-shard compute is a ``QuantizedLinear``-equivalent call on the shard: the prepacked 1x16 kernel when the shard is
large enough (a 1024-wide shard of the 70B layer is 3.7 M codes: 10.8 us prepacked vs 24 us on the direct kernel), else
whatever ``get_forward_pass_kernel`` returns for the shard's codebooks; tests inject a different ``kernel`` to exercise
the sharding + collective logic on CPU with the gloo backend.

``collective="xgmi"`` (in-split, prepacked shards, one node) replaces the library all-reduce by the fused finalize of
``aqlm_amd/csrc/xgmi_reduce.hip``: the shard's main kernel leaves fp32 slice partials, every rank publishes their sum in
IPC-mapped memory and the finalize of every rank adds the R vectors in rank order straight over xGMI -- no extra launch,
fp32 on the wire, one rounding, replicas of y bit-identical.  The ranks agree collectively at first use whether every
shard can take that path; otherwise all of them fall back to ``dist.all_reduce``.

Reduction precision of the in-split: partial outputs are fp16 / bf16 tensors; summing 8 of them in the storage dtype
costs up to ~3 roundings of 2^-11 relative each on top of the per-shard rounding.  ``reduce_dtype=torch.float32`` (default
for fp16 / bf16 layers) all-reduces fp32 partials (112 KiB instead of 56 KiB for the 70B layer: still latency-bound) and
rounds once; ``reduce_dtype=None`` keeps the storage dtype on the wire.
"""
from __future__ import annotations

import ctypes
from typing import Callable, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from .inference_kernels import get_forward_pass_kernel


XGMI_FUSED_PUBLISH = True  # collective="xgmi": the shard's matvec publishes its totals itself (2 launches instead of 3)


def shard_bounds(n: int, world: int, rank: int, multiple: int = 1):
    """Contiguous, nearly equal split of ``n`` units in chunks of ``multiple`` units: returns [lo, hi)."""
    blocks = (n + multiple - 1) // multiple
    base, rem = divmod(blocks, world)
    lo_b = rank * base + min(rank, rem)
    hi_b = lo_b + base + (1 if rank < rem else 0)
    return min(lo_b * multiple, n), min(hi_b * multiple, n)


class ShardedQuantizedLinear(nn.Module):
    """One rank's shard of an AQLM layer.  Build it with :meth:`from_full` (every rank passes the full tensors, or
    loads only its slice using :func:`shard_bounds`)."""

    def __init__(self, codes, codebooks, scales, bias, *, mode: str, in_group_size: int, in_lo: int, in_hi: int,
                 out_lo: int, out_hi: int, out_features: int, group=None, gather_output: bool = True,
                 kernel: Optional[Callable] = None, reduce_dtype: Optional[torch.dtype] = torch.float32,
                 collective: str = "rccl", bias_all: Optional[torch.Tensor] = None):
        super().__init__()
        assert collective in ("rccl", "xgmi")
        self.collective = collective
        self._bias_all = bias_all      # the full bias on every rank (the fused finalize writes the full y everywhere)
        self._xgmi = None              # OneShotAllReduce, built collectively at first use
        self._xgmi_ok = None
        assert mode in ("in", "out")
        self.mode = mode
        self.group = group
        self.gather_output = gather_output
        self.in_group_size = in_group_size
        self.in_lo, self.in_hi, self.out_lo, self.out_hi = in_lo, in_hi, out_lo, out_hi
        self.out_features = out_features
        self.reduce_dtype = reduce_dtype
        self.codes = nn.Parameter(codes, requires_grad=False)
        self.codebooks = nn.Parameter(codebooks, requires_grad=False)
        self.scales = nn.Parameter(scales, requires_grad=False)
        self.bias = nn.Parameter(bias, requires_grad=False) if bias is not None else None
        self._kernel = kernel
        self._packed = None      # prepacked codes of this shard (derived; built at first use on the GPU)
        self._packed_tried = False
        self._packed_fingerprint = None  # identity / storage / version of `codes` at pack time (as QuantizedLinear does)
        self._selector_kernel = None

    def _codes_fingerprint(self):
        c = self.codes
        try:
            v = c._version
        except RuntimeError:  # inference tensors carry no version counter
            v = 0
        return (id(c), c.data_ptr() if c.numel() else 0, tuple(c.shape), v)

    def _drop_derived_if_stale(self) -> None:
        """The prepacked buffer (and the one-shot all-reduce state sized for it) is derived from ``codes``: after
        ``load_state_dict`` / an in-place write / a rebind it is rebuilt at this call instead of being multiplied with."""
        if self._packed_tried and self._packed_fingerprint != self._codes_fingerprint():
            self._packed, self._packed_tried, self._packed_fingerprint = None, False, None
            self._xgmi_ok = None  # the collective decision depends on every shard being packed: agree again

    def _apply(self, fn, *args, **kwargs):
        """``.to()`` / ``.cuda()`` / ``.half()`` replace the parameters: everything derived from them goes."""
        out = super()._apply(fn, *args, **kwargs)
        self._packed, self._packed_tried, self._packed_fingerprint = None, False, None
        self._xgmi, self._xgmi_ok = None, None
        self._selector_kernel = None
        return out

    @classmethod
    def from_full(cls, codes, codebooks, scales, bias, *, mode: str = "in", group=None, gather_output: bool = True,
                  kernel: Optional[Callable] = None, reduce_dtype: Optional[torch.dtype] = torch.float32,
                  collective: str = "rccl"):
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        out_groups, in_groups, _ = codes.shape
        g = codebooks.shape[3]
        if mode == "in":
            # multiples of 8 groups keep each shard on the tuned (16-B code word) kernels
            j0, j1 = shard_bounds(in_groups, world, rank, multiple=8)
            c = codes[:, j0:j1, :].contiguous()
            b = bias if (bias is not None and rank == 0) else None
            return cls(c, codebooks, scales, b, mode=mode, in_group_size=g, in_lo=j0 * g, in_hi=j1 * g, out_lo=0,
                       out_hi=out_groups, out_features=out_groups, group=group, gather_output=gather_output,
                       kernel=kernel, reduce_dtype=reduce_dtype, collective=collective,
                       bias_all=bias if collective == "xgmi" else None)
        i0, i1 = shard_bounds(out_groups, world, rank)
        return cls(codes[i0:i1].contiguous(), codebooks, scales[i0:i1].contiguous(),
                   None if bias is None else bias[i0:i1].contiguous(), mode=mode, in_group_size=g, in_lo=0,
                   in_hi=in_groups * g, out_lo=i0, out_hi=i1, out_features=out_groups, group=group,
                   gather_output=gather_output, kernel=kernel, reduce_dtype=reduce_dtype)

    def _k(self):
        if self._kernel is not None:  # injected (tests): used as is
            return self._kernel
        if self._selector_kernel is None:  # depends on the codebooks' device / dtype: dropped by _apply
            self._selector_kernel = get_forward_pass_kernel(self.codebooks, False)
        return self._selector_kernel

    def _shard_matvec(self, x, bias):
        """The shard's ``(W_shard x) * scales (+ bias)``: prepacked kernel for big 1x16 g8 shards on the GPU (same rule as
        ``QuantizedLinear``), else the selector's kernel / the injected one."""
        from . import inference

        self._drop_derived_if_stale()
        if (self._kernel is None and not self._packed_tried and self.codes.is_cuda and inference.PREPACK_MIN_CODES
                and tuple(self.codebooks.shape[:3]) == (1, 65536, 1) and self.codebooks.shape[3] == 8
                and self.codes.shape[0] * self.codes.shape[1] >= inference.PREPACK_MIN_CODES
                and not torch.cuda.is_current_stream_capturing()):
            from .inference_kernels import hip_kernel

            self._packed_tried = True
            self._packed = hip_kernel.prepack_1x16(self.codes, 8, codebooks=self.codebooks)
            self._packed_fingerprint = self._codes_fingerprint()
        if (self._packed is not None and x.dtype == self.codebooks.dtype
                and x.numel() // x.shape[-1] <= inference.GEMV_MAX_ROWS):
            from .inference_kernels import hip_kernel

            return hip_kernel.code1x16_matmat_packed(x, self._packed, self.codebooks, self.scales, bias)
        return self._k()(x, self.codes, self.codebooks, self.scales, bias)

    def _xgmi_forward(self, xs: torch.Tensor, rows: int) -> Optional[torch.Tensor]:
        """Shard matvec + fused finalize / all-reduce; None when the ranks agreed to use the library collective."""
        from . import _native, inference
        from .inference_kernels import hip_kernel
        from .xgmi import OneShotAllReduce

        self._drop_derived_if_stale()
        if self._xgmi_ok is None:  # first use: collective decision + state exchange
            if not self._packed_tried and self.codes.is_cuda and self.codes.shape[1] > 0:
                self._packed_tried = True
                self._packed_fingerprint = self._codes_fingerprint()
                if (tuple(self.codebooks.shape[:3]) == (1, 65536, 1) and self.codebooks.shape[3] == 8
                        and self.codes.shape[0] * self.codes.shape[1] >= inference.PREPACK_MIN_CODES):
                    self._packed = hip_kernel.prepack_1x16(self.codes, 8, codebooks=self.codebooks)
            ok = torch.tensor([1 if self._packed is not None else 0], device=xs.device)
            if dist.is_initialized() and dist.get_world_size(self.group) > 1:
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
            self._xgmi_ok = bool(int(ok))
            if self._xgmi_ok:
                self._xgmi = OneShotAllReduce(_native.MAX_GEMV_BATCH * self.out_features, xs.device, self.group)
        if not self._xgmi_ok or rows > _native.MAX_GEMV_BATCH or xs.dtype != self.codebooks.dtype:
            return None
        x2 = hip_kernel._flat_rows(xs)
        dt = hip_kernel._dtype_id(xs)
        y = torch.empty((rows, self.out_features), dtype=xs.dtype, device=xs.device)
        stream = hip_kernel._stream_ptr(xs.device)
        hip_kernel._refresh_range(self._packed, self.codebooks)
        if XGMI_FUSED_PUBLISH and hip_kernel.FUSED_FINALIZE and self._packed.desc.codebook_absmax > 0.0:
            # two launches: the matvec publishes the shard's totals itself (last-arrival branch of its finalize), then the
            # reduce.  Falls back to partials + publish + reduce when the rows do not fit one launch.
            pub, flag = self._xgmi.own_pub_flag()
            with torch.cuda.device(xs.device):
                rc = _native.lib.aqlm_hip_gemv_1x16_packed_publish(ctypes.byref(self._packed.desc), self._packed.data_ptr(),
                                                                   self.codebooks.data_ptr(), x2.data_ptr(), rows, x2.stride(0),
                                                                   dt, ctypes.byref(self._xgmi.xg), pub, flag, stream)
                if rc == 0:
                    self._xgmi.reduce(self.scales, self._bias_all, y, self.out_features, rows, dt, stream)
                    return y.reshape(xs.shape[:-1] + (self.out_features,))
                if rc != _native.E_UNSUPPORTED:
                    _native.check(rc, "aqlm gemv_1x16_packed_publish")
        ws = hip_kernel._workspace(xs.device, 16 * rows * self.out_features * 4)
        with torch.cuda.device(xs.device):
            rc = _native.lib.aqlm_hip_gemv_1x16_packed_partials(ctypes.byref(self._packed.desc), self._packed.data_ptr(),
                                                                self.codebooks.data_ptr(), x2.data_ptr(), rows, x2.stride(0),
                                                                dt, ws.data_ptr(), ws.numel() * 4, stream)
            if rc:
                _native.check(rc, "aqlm gemv_1x16_packed_partials")
            self._xgmi.finalize(ws, self.scales, self._bias_all, y, self.out_features, rows, dt, stream)
        return y.reshape(xs.shape[:-1] + (self.out_features,))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if self.mode == "in" and self.collective == "xgmi" and x.is_cuda and self._kernel is None:
            y = self._xgmi_forward(x[..., self.in_lo:self.in_hi], x.numel() // x.shape[-1])
            if y is not None:
                return y
        if self.mode == "in":
            xs = x[..., self.in_lo:self.in_hi]
            if self.codes.shape[1] == 0:  # more ranks than 8-group blocks: this rank contributes nothing
                y = torch.zeros(x.shape[:-1] + (self.out_features,), dtype=x.dtype, device=x.device)
            else:
                y = self._shard_matvec(xs, self.bias)
            if world > 1:
                if self.reduce_dtype is not None and self.reduce_dtype != y.dtype:
                    acc = y.to(self.reduce_dtype)
                    dist.all_reduce(acc, group=self.group)
                    y = acc.to(y.dtype)
                else:
                    dist.all_reduce(y, group=self.group)
            return y
        y = self._shard_matvec(x, self.bias)
        if world == 1 or not self.gather_output:
            return y
        sizes = [shard_bounds(self.out_features, world, r)[1] - shard_bounds(self.out_features, world, r)[0]
                 for r in range(world)]
        width = max(sizes)  # all_gather needs equal shapes: pad ragged shards, trim after
        mine = y if y.shape[-1] == width else torch.nn.functional.pad(y, (0, width - y.shape[-1]))
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine.contiguous(), group=self.group)
        return torch.cat([p[..., :s] for p, s in zip(parts, sizes)], dim=-1)
