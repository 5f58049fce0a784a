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


def packed_matvec_cost_us(out_features: int, in_features: int, in_group_size: int = 8) -> float:
    """Empirical cost line of the prepacked 1x16 matvec on one MI355X (bs = 1, cold), fitted to bench.py's five shapes: ~3.3 us fixed
    (launch boundary, slice fill, hand-in) + ~1.1 us per million entry slots, where a (row, slice) bucket counts as its lane-steps of
    4 entries.  What the slot count stands in for: short rows cost more per code than long ones -- a 1024-wide shard has 8 codes per
    bucket, i.e. a row end (flush of the lane's sums, and one returning atomic per row and slice in the epilogue: 459 k of them for
    28672 rows) every 2-3 lane-steps --: 10.1 us for the 3.7 M codes of 1024 -> 28672 where the 8192 -> 3584 shard of the same layer
    takes 8.0 us (4096 x 4096 6.0, 4096 -> 11008 8.8, 8192 -> 28672 26.9).  Good to ~15 % on those; not a model of anything else."""
    slices = 16 if in_group_size == 8 else 32
    per_bucket = (in_features // in_group_size) / slices
    lane_steps = max(1, -(-int(per_bucket + 1.5) // 4))       # expected codes + the Poisson tail, in steps of 4 entries
    return 3.3 + 1.1e-6 * out_features * slices * 4 * lane_steps


def preferred_partition(out_features: int, in_features: int, world: int, in_group_size: int = 8, allreduce_us: float = 12.0,
                        allgather_us: float = 10.0, gather_output: bool = True) -> str:
    """"in" (north star: row-sharded codes + all-reduce of out values) or "out" (column shards + all-gather of out / world values per
    rank, or nothing when the output stays sharded) for ONE layer: shard kernel by `packed_matvec_cost_us` + the collective.  The
    collective terms are latency-bound on xGMI at these sizes (56-112 KiB); the defaults are placeholders until the 8-GPU tier has
    measured them -- the decision for the 70B layer (8192 -> 28672 over 8) does not hinge on them: 10.1 + reduce vs 8.0 + gather."""
    t_in = packed_matvec_cost_us(out_features, max(in_group_size, in_features // world), in_group_size) + allreduce_us
    t_out = packed_matvec_cost_us(max(1, out_features // world), in_features, in_group_size) + (allgather_us if gather_output else 0.0)
    return "in" if t_in < t_out else "out"


class ShardedQuantizedLinear(nn.Module):
    """One rank's shard of an AQLM layer.  Build it with :meth:`from_full` (every rank passes the full tensors, or
    loads only its slice using :func:`shard_bounds`)."""

    def __init__(self, codes, codebooks, scales, bias, *, mode: str, in_group_size: int, in_lo: int, in_hi: int,
                 out_lo: int, out_hi: int, out_features: int, group=None, gather_output: bool = True,
                 kernel: Optional[Callable] = None, reduce_dtype: Optional[torch.dtype] = torch.float32,
                 collective: str = "rccl", bias_all: Optional[torch.Tensor] = None, input_is_sharded: bool = False):
        super().__init__()
        self.input_is_sharded = input_is_sharded  # in-split only: x arrives as this rank's slice (the paired out-split layer's output)
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
        self._gather_sizes = None
        self._cpu_codes_alt = None  # (fingerprint, codes permuted for the host LUT kernel); derived

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
        self._cpu_codes_alt = None
        return out

    @classmethod
    def from_full(cls, codes, codebooks, scales, bias, *, mode: str = "in", group=None, gather_output: bool = True,
                  kernel: Optional[Callable] = None, reduce_dtype: Optional[torch.dtype] = torch.float32,
                  collective: str = "rccl", bounds=None, input_is_sharded: bool = False):
        """``bounds`` = this rank's [lo, hi) -- input groups for ``mode="in"``, output rows for ``mode="out"`` -- when the
        caller fixes the partition itself (paired layers must cut at the same places: :func:`shard_mlp`)."""
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        out_groups, in_groups, _ = codes.shape
        g = codebooks.shape[3]
        if mode == "auto":  # one layer on its own: the cheaper of the two partitions by the cost line above
            mode = preferred_partition(out_groups, in_groups * g, world, g, gather_output=gather_output) if bounds is None else "in"
        if mode == "in":
            # multiples of 8 groups keep each shard on the tuned (16-B code word) kernels
            j0, j1 = bounds if bounds is not None else shard_bounds(in_groups, world, rank, multiple=8)
            c = codes[:, j0:j1, :].contiguous()
            b = bias if (bias is not None and rank == 0) else None
            return cls(c, codebooks, scales, b, mode=mode, in_group_size=g, in_lo=j0 * g, in_hi=j1 * g, out_lo=0,
                       out_hi=out_groups, out_features=out_groups, group=group, gather_output=gather_output,
                       kernel=kernel, reduce_dtype=reduce_dtype, collective=collective,
                       bias_all=bias if collective == "xgmi" else None, input_is_sharded=input_is_sharded)
        i0, i1 = bounds if bounds is not None else shard_bounds(out_groups, world, rank)
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
        if self._kernel is None and not self.codes.is_cuda:
            from .inference_kernels.kernel_selector import cpu_kernel_takes_permuted_codes

            if cpu_kernel_takes_permuted_codes(self.codebooks):
                # host shards with 8-bit codebooks: the native LUT kernel reads codes permuted to [in_groups, out, K]; a derived
                # copy, as in QuantizedLinear.prepare_matmul_op (the reference permutes the parameter in place, inference.py:78-83)
                fp = self._codes_fingerprint()
                if self._cpu_codes_alt is None or self._cpu_codes_alt[0] != fp:
                    from .inference_kernels.cpu_kernel import permute_codes_for_lut

                    self._cpu_codes_alt = (fp, permute_codes_for_lut(self.codes))
                return self._k()(x, self._cpu_codes_alt[1], self.codebooks, self.scales, bias)
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
                    # (the publish form of the matvec exists for the 16 x 16 geometry only)
                    self._packed = hip_kernel.prepack_1x16(self.codes, 8, codebooks=self.codebooks, uniform_only=True)
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
        if self.mode == "in" and self.input_is_sharded and x.shape[-1] != self.in_hi - self.in_lo:
            raise ValueError(f"this in-split shard expects its own slice of the input ({self.in_hi - self.in_lo} features), got {x.shape[-1]}")
        if self.mode == "in" and self.collective == "xgmi" and x.is_cuda and self._kernel is None:
            y = self._xgmi_forward(x if self.input_is_sharded else x[..., self.in_lo:self.in_hi], x.numel() // x.shape[-1])
            if y is not None:
                return y
        if self.mode == "in":
            xs = x if self.input_is_sharded else x[..., self.in_lo:self.in_hi]
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
        if self._gather_sizes is None:  # the ranks' row counts (explicit bounds need not be the default split): agreed once
            mine_n = torch.tensor([self.out_hi - self.out_lo], dtype=torch.int64, device=y.device)
            alln = [torch.empty_like(mine_n) for _ in range(world)]
            dist.all_gather(alln, mine_n, group=self.group)
            self._gather_sizes = [int(t) for t in alln]
        sizes = self._gather_sizes
        width = max(sizes)  # all_gather needs equal shapes: pad ragged shards, trim after
        mine = y if y.shape[-1] == width else torch.nn.functional.pad(y, (0, width - y.shape[-1]))
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine.contiguous(), group=self.group)
        return torch.cat([p[..., :s] for p, s in zip(parts, sizes)], dim=-1)


# ---------------------------------------------------------------------------------------------------------------------
# Megatron pairing: an out-split layer whose output stays sharded feeds an in-split layer that
# takes its own slice -- gate/up -> act * mul -> down needs ONE collective per MLP (the in-split's all-reduce), q/k/v ->
# attention over the rank's heads -> o_proj one per attention block; an "in-split everywhere" plan pays one per layer
# (7 per decoder block instead of 2).  The modules replace the QuantizedLinear children in place, so Hugging Face's modeling
# code (`down_proj(act_fn(gate_proj(x)) * up_proj(x))`, `o_proj(attn(q_proj(x), k_proj(x), v_proj(x)))`) runs unchanged on
# per-rank slices.
# ---------------------------------------------------------------------------------------------------------------------
def _layer_tensors(m):
    codes = m._canonical_codes() if hasattr(m, "_canonical_codes") else m.codes
    return codes.detach(), m.codebooks.detach(), m.scales.detach(), (None if m.bias is None else m.bias.detach())


def _world_rank(group):
    if dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def shard_pair(first_layers, second, *, unit: int = 1, group=None, kernel: Optional[Callable] = None,
               reduce_dtype: Optional[torch.dtype] = torch.float32, collective: str = "rccl"):
    """Column-parallel ``first_layers`` (their outputs stay sharded) feeding row-parallel ``second``: the shared dimension
    (rows of the first layers = input features of the second) is cut ONCE, in blocks of ``unit`` features (rounded up to whole
    8-group code words of ``second``; ``unit`` = head_dim x heads-per-kv-head for attention so that heads stay whole).
    Returns ``([sharded first layers], sharded second)``."""
    world, rank = _world_rank(group)
    codes2, cb2, sc2, b2 = _layer_tensors(second)
    g2 = int(cb2.shape[3])
    inner = codes2.shape[1] * g2
    for m in first_layers:
        if _layer_tensors(m)[0].shape[0] * int(m.codebooks.shape[2]) != inner:
            raise ValueError(f"paired layers must share the inner dimension ({inner} input features of the second layer)")
    block = unit * 8 * g2 // _gcd(unit, 8 * g2)  # lcm: whole units AND whole 8-group code words of the second layer
    if inner % block:
        block = unit * g2 // _gcd(unit, g2)      # ragged inner dimension: whole groups at least
    lo, hi = shard_bounds(inner, world, rank, multiple=block)
    firsts = []
    for m in first_layers:
        c, cb, sc, b = _layer_tensors(m)
        firsts.append(ShardedQuantizedLinear.from_full(c, cb, sc, b, mode="out", group=group, gather_output=False, kernel=kernel,
                                                       bounds=(lo, hi)))
    sec = ShardedQuantizedLinear.from_full(codes2, cb2, sc2, b2, mode="in", group=group, kernel=kernel, reduce_dtype=reduce_dtype,
                                           collective=collective, bounds=(lo // g2, hi // g2), input_is_sharded=True)
    return firsts, sec


def _gcd(a: int, b: int) -> int:
    while b:
        a, b = b, a % b
    return a


def shard_mlp(mlp: nn.Module, *, names=("gate_proj", "up_proj", "down_proj"), group=None, kernel: Optional[Callable] = None,
              reduce_dtype: Optional[torch.dtype] = torch.float32, collective: str = "rccl") -> nn.Module:
    """Replace ``mlp.gate_proj`` / ``up_proj`` (out-split, outputs left sharded) and ``down_proj`` (in-split on the same cut,
    one all-reduce) in place.  No collective for gate / up; the activation and the product act on the rank's slice."""
    *first_names, second_name = names
    firsts, sec = shard_pair([getattr(mlp, n) for n in first_names], getattr(mlp, second_name), group=group, kernel=kernel,
                             reduce_dtype=reduce_dtype, collective=collective)
    for n, m in zip(first_names, firsts):
        setattr(mlp, n, m)
    setattr(mlp, second_name, sec)
    return mlp


def shard_attention(attn: nn.Module, *, head_dim: int, num_heads: int, num_kv_heads: int,
                    names=("q_proj", "k_proj", "v_proj", "o_proj"), group=None, kernel: Optional[Callable] = None,
                    reduce_dtype: Optional[torch.dtype] = torch.float32, collective: str = "rccl") -> nn.Module:
    """q / k / v out-split by whole heads (a rank keeps the query heads of its key / value heads), o_proj in-split on the
    same cut.  Needs ``num_kv_heads`` divisible by the world size (Llama-3-70B: 8 kv heads, 8 GPUs)."""
    world, rank = _world_rank(group)
    if num_kv_heads % world or num_heads % num_kv_heads:
        raise ValueError(f"{num_kv_heads} key/value heads do not split over {world} ranks")
    qn, kn, vn, on = names
    rep = num_heads // num_kv_heads
    (q,), o = shard_pair([getattr(attn, qn)], getattr(attn, on), unit=head_dim * rep * (num_kv_heads // world), group=group,
                         kernel=kernel, reduce_dtype=reduce_dtype, collective=collective)
    kv_lo, kv_hi = shard_bounds(num_kv_heads * head_dim, world, rank, multiple=head_dim * (num_kv_heads // world))
    for n in (kn, vn):
        c, cb, sc, b = _layer_tensors(getattr(attn, n))
        setattr(attn, n, ShardedQuantizedLinear.from_full(c, cb, sc, b, mode="out", group=group, gather_output=False, kernel=kernel,
                                                          bounds=(kv_lo, kv_hi)))
    setattr(attn, qn, q)
    setattr(attn, on, o)
    for attr, val in (("num_heads", num_heads // world), ("num_key_value_heads", num_kv_heads // world)):
        if hasattr(attn, attr):  # older Hugging Face attention modules reshape with these
            setattr(attn, attr, val)
    return attn


def shard_model(model: nn.Module, *, group=None, attention: bool = True, kernel: Optional[Callable] = None,
                reduce_dtype: Optional[torch.dtype] = torch.float32, collective: str = "rccl") -> dict:
    """Apply the pairing to every decoder block of a Hugging Face style model whose linears are ``QuantizedLinear``: MLPs
    always; attention blocks when the head counts allow it (``model.config`` gives them).  Returns a summary
    ``{"mlp": n, "attention": n, "collectives_per_block": ...}``.  Everything else (norms, embeddings, lm_head) stays replicated."""
    from .inference import QuantizedLinear

    cfg = getattr(model, "config", None)
    done = {"mlp": 0, "attention": 0}
    for mod in list(model.modules()):
        kids = dict(mod.named_children())
        if all(isinstance(kids.get(n), QuantizedLinear) for n in ("gate_proj", "up_proj", "down_proj")):
            shard_mlp(mod, group=group, kernel=kernel, reduce_dtype=reduce_dtype, collective=collective)
            done["mlp"] += 1
        if attention and cfg is not None and all(isinstance(kids.get(n), QuantizedLinear) for n in ("q_proj", "k_proj", "v_proj", "o_proj")):
            heads = int(cfg.num_attention_heads)
            kv = int(getattr(cfg, "num_key_value_heads", heads) or heads)
            hd = int(getattr(cfg, "head_dim", None) or cfg.hidden_size // heads)
            world, _ = _world_rank(group)
            if kv % world == 0:
                shard_attention(mod, head_dim=hd, num_heads=heads, num_kv_heads=kv, group=group, kernel=kernel,
                                reduce_dtype=reduce_dtype, collective=collective)
                done["attention"] += 1
    done["collectives_per_block"] = (1 if done["mlp"] else 0) + (1 if done["attention"] else 0)
    return done
