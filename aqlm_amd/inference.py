"""``QuantizedLinear``: the module Hugging Face instantiates for every AQLM-quantized ``nn.Linear``.

Host-side mirror of the reference module (inference_lib/src/aqlm/inference.py:11-142): identical constructor
signature, parameter names / shapes / dtypes (so checkpoints written by convert_to_hf.py load unchanged,
including construction on the ``meta`` device by transformers/integrations/aqlm.py:47-57), identical
gemv-vs-gemm rule, autograd support -- with the kernels behind it replaced by the MI355X ops.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn

from .inference_kernels import get_backward_pass_kernel, get_forward_pass_kernel
from .utils import get_int_dtype

# rows (batch * seq) at or below which the gemv op is used; the reference's value (inference.py:95-96)
GEMV_MAX_ROWS = 6

# 1x16 (g8 / g16) matvecs (<= GEMV_MAX_ROWS rows) of layers with at least this many codes (out_features * in_features / g) run on
# slice-bucketed ("prepacked") codes (aqlm_hip_gemv_1x16_packed): 1.2-5x faster than the direct L2-gather kernel on
# MI355X, at the price of a one-off repack at first use and ~2.3x the code bytes (kept next to the original codes unless
# `drop_canonical_codes()` is called).  0 disables.  Measured cross-over (cold, single launch each): 4096->1024 (0.5 M
# codes) 4.6 us packed vs 5.3 us direct; below that the fixed cost of the 64 KiB LDS fill per CU dominates.
PREPACK_MIN_CODES = 500_000

# Everything a module derives from its parameters (prepacked / planar / permuted codes, the codebook image and range, a dense W)
# is keyed on the parameters' identity and version counter -- and a write through `.data` (`m.codes.data.copy_(...)`, some
# loaders, hand-written surgery) changes neither.  The reference has no such state (its launcher reads the live tensors on every
# call, cuda_kernel.cpp:148-182).  So every DERIVED_CHECK_EVERY-th forward of a module (never while a hipGraph is being captured
# or traced) re-takes a 128-bit checksum of the parameters the derived state came from (aqlm_hip_checksum: one small kernel + a
# 16-byte read-back per parameter, ~40 us) and rebuilds what no longer matches; `invalidate_derived_state()` does it on demand.
# 0 = never check.  The check is EAGER-ONLY by construction: it never runs under hipGraph replay (a captured forward does not pass
# through this Python at all) nor in a torch.compile'd graph -- there, call `invalidate_derived_state()` after such a write and
# re-capture.  The modules of a model start their counters at different offsets, so the checks do not land on the same step.
DERIVED_CHECK_EVERY = 256


class QuantizedLinear(nn.Module):
    def __init__(
        self,
        in_features: int,
        out_features: int,
        in_group_size: int,
        out_group_size: int,
        num_codebooks: int,
        nbits_per_codebook: int,
        bias=True,
        device=None,
        dtype=None,
    ):
        super().__init__()
        if in_features % in_group_size or out_features % out_group_size:
            raise AssertionError("feature counts must be divisible by the group sizes")
        self.in_features, self.out_features = in_features, out_features
        self.in_group_size, self.out_group_size = in_group_size, out_group_size
        self.num_codebooks = num_codebooks
        self.nbits_per_codebook = nbits_per_codebook
        self.codebook_size = 2**nbits_per_codebook
        n_out, n_in = out_features // out_group_size, in_features // in_group_size
        fkw = {"device": device, "dtype": dtype}

        def frozen(t):
            return nn.Parameter(t, requires_grad=False)

        # same registration order as the reference (inference.py:39-61): codebooks, codes, scales, bias
        self.codebooks = frozen(torch.empty((num_codebooks, self.codebook_size, out_group_size, in_group_size), **fkw))
        self.codes = frozen(torch.empty((n_out, n_in, num_codebooks), device=device, dtype=get_int_dtype(nbits_per_codebook)))
        self.scales = frozen(torch.empty((n_out, 1, 1, 1), **fkw))
        if bias:
            self.bias = frozen(torch.empty(out_features, **fkw))
        else:
            self.register_parameter("bias", None)

        self.gemv_op = None
        self.gemm_op = None
        self.use_gemv_rule = None
        self._packed_codes = None  # derived, never saved: rebuilt from `codes` at first use
        self._packed_fingerprint = None
        self._codes_dropped = False
        self._codes_drop_strict = True  # False: the first call that needs the canonical codes restores them for good (prepack_model's default)
        self._codes_shape = None
        self._cpu_codes_alt = None  # host modules with 8-bit codebooks: codes permuted for the LUT kernel (derived)
        self._prepack_deferred = False
        self._shared_input_group = None  # set by aqlm_amd.fusion.fuse_shared_input_linears
        self._fast = None  # compiled fast lane of the decode call (aqlm_amd/_front.py); derived, rebuilt with the kernel choice
        # Escape hatch for mid-size batches (7 .. N - 1 rows: prefill chunks, speculative verification): run them as a dense GEMM
        # on a cached fp16 / bf16 copy of W instead of the fused dequant -> MFMA op.  Measured on MI355X at 4096 x 4096
        # (BENCH detail `bs128_1x16g8_4096x4096`): dense 12.1 / 12.6 / 14.5 us at 16 / 32 / 64 rows against 15.6 / 16.7 / 18.1 us
        # fused -- the fused op's 2.1 M codebook gathers cannot stream like a dense weight does.  The price is the memory the
        # format exists to save (2 bytes per weight for every layer that uses it), so it is off (0) unless asked for:
        # `module.prefer_dense_below_rows = 129`, or `aqlm.checkpoint.enable_dense_below_rows(model, 129)`.
        self.prefer_dense_below_rows = 0
        self._dense = None  # (fingerprint, W) -- derived, never saved
        self._derived_checks = None  # checksums of the parameters the derived state was built from (DERIVED_CHECK_EVERY)
        self._calls_since_check = 0

    # Everything the module derives from its parameters (kernel choice, autograd ops, prepacked / permuted codes, the compiled
    # fast lane -- a pybind11 object that cannot be pickled) is left out of copies and pickles: `copy.deepcopy(model)`,
    # `torch.save(model)` and `pickle` work at any time (EMA copies, PEFT `modules_to_save`, draft-model clones), and the copy
    # rebuilds its derived state at its first forward.  A module whose canonical codes were dropped hands them back to the copy.
    _DERIVED_DEFAULTS = {"gemv_op": None, "gemm_op": None, "use_gemv_rule": None, "_fast": None, "_packed_codes": None,
                         "_packed_fingerprint": None, "_cpu_codes_alt": None, "_prepack_deferred": False, "_dense": None,
                         "_derived_checks": None, "_calls_since_check": 0}

    def __getstate__(self):
        state = dict(self.__dict__)
        if self._codes_dropped:
            params = state["_parameters"].copy()
            params["codes"] = nn.Parameter(self._canonical_codes(), requires_grad=False)
            state["_parameters"] = params
            state["_codes_dropped"] = False
        state.update(self._DERIVED_DEFAULTS)
        return state

    def extra_repr(self) -> str:
        return (f"in_features={self.in_features}, out_features={self.out_features}, scheme="
                f"{self.num_codebooks}x{self.nbits_per_codebook}g{self.in_group_size}, bias={self.bias is not None}")

    def invalidate_derived_state(self) -> None:
        """Forget everything derived from the parameters (kernel choice, prepacked / planar / permuted codes, the codebook image
        and range, the dense W, the compiled fast lane); the next forward rebuilds it.  Call it after writing a parameter through
        a path PyTorch does not version (``m.codes.data.copy_(...)``) when the periodic check (DERIVED_CHECK_EVERY) is too late
        or switched off.  A module whose canonical codes were dropped keeps its packed buffer (it IS the weights) and forgets only
        what came from the codebooks."""
        self.gemv_op = self.gemm_op = self.use_gemv_rule = None
        self._fast = None
        self._dense = None
        self._cpu_codes_alt = None
        self._prepack_deferred = False
        self._derived_checks = None
        self._calls_since_check = 0
        if self._codes_dropped:
            if self._packed_codes is not None:
                self._packed_codes._range_of = None
        else:
            self._packed_codes = None
        group = self._shared_input_group
        if group is not None and hasattr(group, "invalidate"):
            group.invalidate()

    def _record_derived_checks(self) -> None:
        """Checksums of the parameters the derived state was just built from (see DERIVED_CHECK_EVERY)."""
        self._derived_checks = None
        # staggered start: the modules of a model were all built at the same forward; without an offset every 256th decode step would
        # verify all of them at once (one multi-millisecond spike per 256 tokens instead of one module's ~40 us now and then)
        self._calls_since_check = (id(self) >> 4) % DERIVED_CHECK_EVERY if DERIVED_CHECK_EVERY else 0
        if not DERIVED_CHECK_EVERY or (self._packed_codes is None and self._cpu_codes_alt is None and self._dense is None):
            return
        if torch.cuda.is_available() and self.codes.is_cuda and torch.cuda.is_current_stream_capturing():
            return
        from .inference_kernels.hip_kernel import tensor_checksum

        checks = {}
        if not self._codes_dropped:
            checks["codes"] = tensor_checksum(self.codes)
        if self._packed_codes is not None or self._dense is not None:
            checks["codebooks"] = tensor_checksum(self.codebooks)
        if self._dense is not None:
            checks["scales"] = tensor_checksum(self.scales)
        self._derived_checks = checks

    def verify_derived_state(self) -> bool:
        """Re-take the checksums of `_record_derived_checks` and compare; a mismatch (a parameter was overwritten behind the
        version counter's back) invalidates the derived state.  Returns whether everything still matched.  Synchronises."""
        checks = self._derived_checks
        if not checks:
            return True
        from .inference_kernels.hip_kernel import tensor_checksum

        ok = all(tensor_checksum(getattr(self, name)) == value for name, value in checks.items())
        if not ok:
            self.invalidate_derived_state()
        return ok

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        # (is_compiling() FIRST: Dynamo specialises int attributes of a module, so a counter touched while tracing installs a guard
        # `_calls_since_check == k` that fails on the next call and recompiles until the limit -- ADVICE r05)
        if not torch.compiler.is_compiling() and self._derived_checks is not None:
            n = self._calls_since_check + 1
            if n >= DERIVED_CHECK_EVERY > 0 and not (input.is_cuda and torch.cuda.is_current_stream_capturing()):
                n = 0
                self.verify_derived_state()
            self._calls_since_check = n
        group = self._shared_input_group
        if group is not None and group.applicable(input):
            return group.forward(self, input)  # one launch for all projections of this input (fusion.py)
        fast = self._fast
        if fast is not None and not torch.compiler.is_compiling():
            out = fast(input)  # None: not a call for the fast lane (rows, dtype, grad, a parameter changed) -> below
            if out is not None:
                return out
            if not fast.is_current():
                self.gemv_op = None  # a parameter was rebound / written in place: resolve everything again
        if self.gemv_op is None or (not torch.compiler.is_compiling() and self._derived_state_is_stale()):
            self.prepare_matmul_op(input)
        packed = self._packed_codes
        if (packed is not None and input.dtype == self.codebooks.dtype and input.is_cuda
                and math.prod(input.shape[:-1]) <= GEMV_MAX_ROWS and not (torch.is_grad_enabled() and input.requires_grad)):
            from .inference_kernels import hip_kernel

            if isinstance(packed, hip_kernel.PlanarCodes):
                # 8x8 on planar codes: the look-up-table matvec takes the gemv rule's 1..6 rows (2+: one launch of rows x the
                # single-row workgroups); anything else goes through the ordinary ops below.  8x8 g32 calls of
                # fused_8x8_min_rows(layer)+ rows (3 at 4096 x 4096) go to the ops too while the checkpoint-layout codes are there:
                # the fused MFMA kernel (aqlm_hip_gemm_8x8_mfma, round 5) costs the same for 3 rows as for 16.
                if input.numel() >= input.shape[-1] > 0 and not self._rows_take_the_fused_8x8_op(input):
                    if torch.compiler.is_compiling():  # traced: the dispatcher op (it has a fake implementation)
                        return torch.ops.aqlm.code8x8_matmat_planar(input, packed.buf, self.codebooks, self.scales, self.bias,
                                                                    [packed.out_features, packed.in_features, packed.in_group_size],
                                                                    packed.codebook_absmax)
                    return hip_kernel.code8x8_matmat_planar(input, packed, self.codebooks, self.scales, self.bias)
            elif (not torch.compiler.is_compiling() and torch.cuda.is_current_stream_capturing() and packed.desc.relabelled
                  and not self._codes_dropped and not packed.range_is_current(self.codebooks)):
                # a relabelled buffer's kernels read a derived codebook IMAGE; the codebook changed and the image cannot be rewritten
                # inside a capture (it reads a bound back): this call runs the direct kernel on the canonical codes and the live
                # codebook instead of raising; the first eager forward rewrites the image
                return self.gemv_op.apply(input, self.codes, self.codebooks, self.scales, self.bias)
            elif torch.compiler.is_compiling():  # traced: go through the dispatcher op (it has a fake implementation)
                return torch.ops.aqlm.code1x16_matmat_packed(input, packed.buf, self.codebooks, self.scales, self.bias,
                                                             packed.op_ints())
            else:
                return hip_kernel.code1x16_matmat_packed(input, packed, self.codebooks, self.scales, self.bias)
        if (self.prefer_dense_below_rows and input.is_cuda and GEMV_MAX_ROWS < math.prod(input.shape[:-1]) < self.prefer_dense_below_rows
                and input.dtype == self.codebooks.dtype and not (torch.is_grad_enabled() and input.requires_grad)
                and not torch.compiler.is_compiling()):
            return torch.nn.functional.linear(input, self._dense_weight(), self.bias)
        op = self.gemv_op if self.use_gemv_rule(input) else self.gemm_op
        return op.apply(input, self._codes_for_ops(), self.codebooks, self.scales, self.bias)

    def _rows_take_the_fused_8x8_op(self, input: torch.Tensor) -> bool:
        from .inference_kernels import hip_kernel

        return (hip_kernel.USE_FUSED_8X8_MFMA and not self._codes_dropped and self.in_group_size == 32 and self.in_features % 256 == 0
                and self.in_features >= 2048
                and math.prod(input.shape[:-1]) >= hip_kernel.fused_8x8_min_rows(self.out_features, self.in_features)
                and not torch.compiler.is_compiling())

    def _dense_weight(self) -> torch.Tensor:
        """W in the storage dtype, dequantised once and kept while codes / codebooks / scales are what they were (identity +
        version, and the periodic checksum of DERIVED_CHECK_EVERY for writes through ``.data``).  The scales are folded into W
        BEFORE it is rounded to fp16 / bf16 -- one more rounding per weight than the reference's large-batch path, which
        dequantises unscaled and scales y after the GEMM (cuda_kernel.cpp:249-301, 294-300), and than the fused ops here
        (fp32 accumulate, scale, one rounding); the price of running the opt-in dense route as ONE library GEMM with the bias
        fused."""
        def ver(t):
            try:
                return t._version
            except RuntimeError:
                return 0

        fp = (self._codes_fingerprint(), self.codebooks.data_ptr(), ver(self.codebooks), self.scales.data_ptr(), ver(self.scales))
        if self._dense is None or self._dense[0] != fp:
            from .utils import _dequantize_weight, unpack_int_data

            with torch.no_grad():
                w = _dequantize_weight(unpack_int_data(self._canonical_codes(), self.nbits_per_codebook), self.codebooks, self.scales)
            self._dense = (fp, w.to(self.codebooks.dtype).contiguous())
            self._record_derived_checks()
        return self._dense[1]

    def _codes_fingerprint(self):
        c = self.codes
        try:
            v = c._version
        except RuntimeError:  # inference tensors carry no version counter
            v = 0
        return (id(c), c.data_ptr() if c.numel() else 0, tuple(c.shape), v)

    def _derived_state_is_stale(self) -> bool:
        """The prepacked buffer is derived from ``codes``: rebuild it when ``codes`` was written in place
        (``load_state_dict`` / ``copy_`` after the first forward), rebound (``module.codes = ...``, accelerate's
        ``set_module_tensor_to_device``) or when a repack was postponed during graph capture."""
        if self._prepack_deferred:
            return not torch.cuda.is_current_stream_capturing()
        if (self._packed_codes is None and self._cpu_codes_alt is None) or self._codes_dropped:
            return False
        return self._codes_fingerprint() != self._packed_fingerprint

    def _canonical_codes(self) -> torch.Tensor:
        """``codes`` in the checkpoint layout; rebuilt from the prepacked buffer (lossless) when they were dropped."""
        if not self._codes_dropped:
            return self.codes
        return self._packed_codes.unpack()  # lossless inverse of the load-time re-layout (1x16 slices / 8x8 planes)

    def _codes_for_ops(self) -> torch.Tensor:
        """The canonical codes for an op that reads them (> 6 rows, backward, a call the packed kernel does not take).  Dropped and not
        `strict`: this module turns out to be used that way -- the codes come back for good (one unpack, 32 us for a 4096 x 4096
        layer, instead of one per call: `detail.unpack_1x16_us` puts a transient unpack at 2.4-2.9 x the 16-row op it would precede).
        Never inside a hipGraph capture or a trace (the restored parameter would belong to the capture): those calls unpack
        transiently."""
        if (self._codes_dropped and not self._codes_drop_strict and not torch.compiler.is_compiling()
                and not (self.codebooks.is_cuda and torch.cuda.is_current_stream_capturing())):
            self.restore_canonical_codes()
        return self._canonical_codes()

    def drop_canonical_codes(self, strict: bool = True) -> bool:
        """Inference-only memory saver: free ``codes`` of a prepacked layer (the packed buffer holds the same
        information; ``state_dict()`` and the large-batch / backward ops rebuild them on demand).  ``strict=False``: the first
        forward that needs them restores them for good (decode-only deployments keep one copy, a model that is also called with 7+
        rows pays one unpack per layer and ends up with both).  Returns whether anything was freed."""
        if self._packed_codes is None or self._codes_dropped:
            return False
        self._codes_drop_strict = bool(strict)
        self._codes_shape = tuple(self.codes.shape)
        self.codes = nn.Parameter(torch.empty((0,), dtype=self.codes.dtype, device=self.codes.device), requires_grad=False)
        self._codes_dropped = True
        self._build_fast_lane()
        return True

    def restore_canonical_codes(self) -> None:
        if self._codes_dropped:
            codes = self._canonical_codes()
            self._codes_dropped = False
            self.codes = nn.Parameter(codes, requires_grad=False)
            self._packed_fingerprint = self._codes_fingerprint()
            self._build_fast_lane()

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        if self._codes_dropped:  # checkpoints always carry the reference layout
            destination[prefix + "codes"] = self._canonical_codes()

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        if self._codes_dropped and prefix + "codes" in state_dict:  # new codes arrive: give them a parameter to land in
            self._codes_dropped = False
            self._packed_codes = None
            self._fast = None
            self.codes = nn.Parameter(torch.empty(self._codes_shape, dtype=self.codes.dtype, device=self.codes.device),
                                      requires_grad=False)
            self.gemv_op = None
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        """``.to()`` / ``.half()`` / ``.cuda()`` replace the parameters: drop everything derived from them (kernel
        choice, prepacked codes); it is rebuilt at the next forward."""
        self.restore_canonical_codes()  # the derived buffer does not survive a conversion; the codes must
        out = super()._apply(fn, *args, **kwargs)
        self.gemv_op = self.gemm_op = self.use_gemv_rule = None
        self._fast = None
        self._packed_codes = None
        self._cpu_codes_alt = None
        self._prepack_deferred = False
        self._dense = None
        return out

    def prepare_matmul_op(self, input: torch.Tensor):
        self._prepare_matmul_op(input)
        self._record_derived_checks()

    def _prepare_matmul_op(self, input: torch.Tensor):
        """Resolve the decode (gemv) and batch (gemm) operators once (reference inference.py:77-96).  For host modules
        with 8-bit codebooks the reference permutes ``codes`` IN PLACE to the LUT kernel's layout (inference.py:78-83,
        "TODO: fix this thing"); here the permuted copy is a derived buffer and ``codes`` keeps the checkpoint layout."""
        from .inference_kernels.kernel_selector import cpu_kernel_takes_permuted_codes

        self._fast = None
        self._cpu_codes_alt = None
        if cpu_kernel_takes_permuted_codes(self.codebooks):
            from .inference_kernels.cpu_kernel import permute_codes_for_lut

            self._cpu_codes_alt = permute_codes_for_lut(self.codes)
            self._packed_fingerprint = self._codes_fingerprint()
        fwd_gemv, fwd_gemm = get_forward_pass_kernel(self.codebooks, False), get_forward_pass_kernel(self.codebooks, True)
        if self._cpu_codes_alt is not None:
            # the LUT kernel reads the permuted copy; the op still receives (and saves for backward) the canonical codes
            alt, lut_gemv, lut_gemm = self._cpu_codes_alt, fwd_gemv, fwd_gemm
            fwd_gemv = lambda input, codes, codebooks, scales, bias: lut_gemv(input, alt, codebooks, scales, bias)  # noqa: E731
            fwd_gemm = lambda input, codes, codebooks, scales, bias: lut_gemm(input, alt, codebooks, scales, bias)  # noqa: E731
        self.gemv_op = _get_autograd_matmul_op(fwd_gemv, get_backward_pass_kernel(self.codebooks, False))
        self.gemm_op = _get_autograd_matmul_op(fwd_gemm, get_backward_pass_kernel(self.codebooks, True))
        self.use_gemv_rule = lambda x: math.prod(x.shape[:-1]) <= GEMV_MAX_ROWS
        # load-time re-layout of the codes for the decode kernel (the reference does the analogous thing for its CPU
        # kernel here, inference.py:78-83 -- but in place; we keep `codes` untouched and add a derived buffer)
        if self._codes_dropped:
            self._build_fast_lane()
            return  # the packed buffer IS the weights now
        self._packed_codes = None
        self._prepack_deferred = False
        if (PREPACK_MIN_CODES and self.out_features * (self.in_features // self.in_group_size) >= PREPACK_MIN_CODES
                and self.num_codebooks == 1 and self.nbits_per_codebook == 16 and self.in_group_size in (8, 16) and self.out_group_size == 1
                and self.codes.is_cuda and self.codebooks.dtype in (torch.float16, torch.bfloat16)):
            if torch.cuda.is_current_stream_capturing():
                # the repack synchronises its stream: not allowed inside a hipGraph capture.  This call runs on the
                # direct kernel; the repack happens at the first forward outside a capture.
                self._prepack_deferred = True
                return
            from .inference_kernels import hip_kernel

            self._packed_codes = hip_kernel.prepack_1x16(self.codes, self.in_group_size, codebooks=self.codebooks)
            self._packed_fingerprint = self._codes_fingerprint()
        elif (PREPACK_MIN_CODES and self.out_features * (self.in_features // self.in_group_size) * 8 >= PREPACK_MIN_CODES
                and self.num_codebooks == 8 and self.nbits_per_codebook == 8 and self.in_group_size in (8, 16, 32) and self.out_group_size == 1
                and self.codes.is_cuda and self.codebooks.dtype in (torch.float16, torch.bfloat16)):
            # 8 x 8-bit schemes: planar copy of the codes for the look-up-table matvec (same size as `codes`, lossless)
            if torch.cuda.is_current_stream_capturing():
                self._prepack_deferred = True  # the codebook bound is a host read-back: first forward outside a capture
                return
            from .inference_kernels import hip_kernel

            self._packed_codes = hip_kernel.planar_8x8_pack(self.codes, self.in_group_size, codebooks=self.codebooks)
            self._packed_fingerprint = self._codes_fingerprint()
        self._build_fast_lane()

    def _build_fast_lane(self) -> None:
        """The compiled fast lane of decode calls (<= GEMV_MAX_ROWS rows, no grad): flatten / allocate / launch without the
        interpreter.  Same kernels and same results as the ops the selector returned; it watches the parameters and hands
        any call it does not recognise back to the Python path (aqlm_amd/csrc_front/front.cpp)."""
        self._fast = None
        from . import _front

        if (not _front.available() or not self.codebooks.is_cuda or self.out_group_size != 1
                or self.codebooks.dtype not in (torch.float16, torch.bfloat16) or self.scales.dtype != self.codebooks.dtype
                or (self.bias is not None and self.bias.dtype != self.codebooks.dtype)
                or not (self.codebooks.is_contiguous() and self.scales.is_contiguous())):
            return
        from .inference_kernels import hip_kernel

        packed = self._packed_codes
        scheme = (self.num_codebooks, self.nbits_per_codebook, self.in_group_size)
        if isinstance(packed, hip_kernel.PlanarCodes):
            if not (hip_kernel.USE_8X8_LUT and hip_kernel.USE_8X8_LUT_FUSED and packed.codebook_absmax > 0.0 and packed.range_is_current(self.codebooks)):
                return  # two-kernel form (needs a workspace): Python path
            import struct

            kind, buf, desc = _front.KIND_LUT_PLANAR_8X8, packed.buf, struct.pack("<f", packed.codebook_absmax)
        elif packed is not None:
            if not (hip_kernel.FUSED_FINALIZE and packed.desc.codebook_absmax > 0.0 and packed.range_is_current(self.codebooks)):
                return  # two-kernel finalize (needs a workspace): Python path
            kind, buf, desc = _front.KIND_PACKED_1X16, packed.buf, bytes(packed.desc)
        elif self._codes_dropped or not (self.codes.is_cuda and self.codes.is_contiguous()):
            return
        elif scheme in ((1, 16, 8), (1, 16, 16)):
            if hip_kernel.RAW_OP_PREPACK and self.out_features * (self.in_features // self.in_group_size) >= hip_kernel.RAW_OP_PREPACK_MIN_CODES:
                return  # the raw op would pack this layer on its own (PREPACK_MIN_CODES was raised): leave it to the op
            kind, buf, desc = _front.KIND_GEMV_1X16, None, ""
        elif scheme in ((2, 8, 8), (1, 8, 8)):
            kind, buf, desc = _front.KIND_GEMV_KX8, None, ""
        else:
            return
        fused_from = 0
        if (kind == _front.KIND_LUT_PLANAR_8X8 and hip_kernel.USE_FUSED_8X8_MFMA and not self._codes_dropped and self.in_group_size == 32
                and self.in_features % 256 == 0 and self.in_features >= 2048 and self.codes.is_cuda and self.codes.is_contiguous()):
            # more rows: the lane launches the fused MFMA op on the checkpoint-layout codes (what forward would do through the ops)
            fused_from = hip_kernel.fused_8x8_min_rows(self.out_features, self.in_features)
        try:
            self._fast = _front.ext.FastLinear(self._parameters, kind, buf, desc, self.in_features, self.out_features,
                                               self.num_codebooks, self.in_group_size, not self._codes_dropped, GEMV_MAX_ROWS, fused_from)
        except RuntimeError:
            self._fast = None


def _get_autograd_matmul_op(forward_pass_kernel, backward_pass_kernel):
    """autograd.Function around a (forward, backward) kernel pair (reference inference.py:99-142): gradient flows
    to ``input`` only; codes / codebooks / scales / bias are frozen."""

    class _QuantizedMatmul(torch.autograd.Function):
        @staticmethod
        def forward(ctx, input, codes, codebooks, scales, bias: Optional[torch.Tensor]):
            ctx.save_for_backward(codes, codebooks, scales)
            return forward_pass_kernel(input, codes, codebooks, scales, bias)

        @staticmethod
        def backward(ctx, grad_output):
            codes, codebooks, scales = ctx.saved_tensors
            grad_input = backward_pass_kernel(grad_output.contiguous(), codes, codebooks, scales, None)
            return grad_input, None, None, None, None

    return _QuantizedMatmul
