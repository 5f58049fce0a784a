"""Shared-input launches for decoder layers.

The q/k/v projections (and gate/up) of a decoder layer multiply the same hidden state.  The reference issues one
``code1x16_matmat`` per projection (inference.py:68-76 -> cuda_kernel.cpp:148-182); on MI355X a single-row matvec of a
<= 4096-row layer is bounded by launch + memory-latency floors, so running the 2-3 projections in ONE launch
(``aqlm_hip_gemv_1x16_multi`` / ``aqlm_hip_gemv_1x16_packed_multi``) removes most of that cost.

``fuse_shared_input_linears(model)`` groups sibling ``QuantizedLinear`` modules by name; the modules stay in place
(same parameters, same state_dict), Hugging Face's modeling code keeps calling ``q_proj(x)``, ``k_proj(x)``,
``v_proj(x)`` one after the other: the first call launches the whole group and parks the siblings' outputs, the
following calls on the *same tensor object* pick theirs up.  Every member runs on the kernel it would use alone
(prepacked members share one packed launch, the rest one direct launch), so 1x16 / 8x8 outputs are bit-identical to the
unfused modules.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from .inference import GEMV_MAX_ROWS, QuantizedLinear

DEFAULT_PATTERNS: Tuple[Tuple[str, ...], ...] = (("q_proj", "k_proj", "v_proj"), ("gate_proj", "up_proj"))
MAX_MEMBERS = 4  # AQLM_HIP_MAX_SEGMENTS


def _version(t: torch.Tensor) -> int:
    try:
        return t._version
    except RuntimeError:  # inference tensors carry no version counter (and cannot be modified in place outside
        return 0          # inference mode anyway)


class SharedInputGroup:
    """2..4 ``QuantizedLinear`` modules of one scheme -- 1x16 (g 8 / 16), 1x8 / 2x8 (g 8) or 8x8 (g 8 / 16 / 32) -- with
    equal in_features / dtype / device, that are always applied to the same input."""

    def __init__(self, members: Sequence[QuantizedLinear]):
        members = list(members)
        if not (2 <= len(members) <= MAX_MEMBERS):
            raise ValueError(f"a shared-input group has 2..{MAX_MEMBERS} members, got {len(members)}")
        first = members[0]
        for m in members:
            if not isinstance(m, QuantizedLinear):
                raise TypeError(f"shared-input groups hold QuantizedLinear modules, got {type(m).__name__}")
            scheme = (m.num_codebooks, m.nbits_per_codebook, m.in_group_size)
            ok = ((scheme[:2] == (1, 16) and scheme[2] in (8, 16)) or (scheme[1] == 8 and scheme[0] in (1, 2) and scheme[2] == 8)
                  or (scheme[:2] == (8, 8) and scheme[2] in (8, 16, 32)))
            if not ok or m.out_group_size != 1:
                raise NotImplementedError("shared-input launches cover 1x16 (g 8 / 16), 1x8 / 2x8 (g 8) and 8x8 (g 8 / 16 / 32)")
            if (m.in_features,) + scheme != (first.in_features, first.num_codebooks, first.nbits_per_codebook, first.in_group_size):
                raise ValueError("members of a shared-input group must agree on in_features and scheme")
            if m.codebooks.dtype != first.codebooks.dtype or m.codebooks.device != first.codebooks.device:
                raise ValueError("members of a shared-input group must share dtype and device")
        self.members: List[QuantizedLinear] = members
        self._input: Optional[torch.Tensor] = None
        self._version = -1
        self._pending: Dict[int, torch.Tensor] = {}
        self.launches = 0  # statistics: fused launches issued / outputs served from a previous launch
        self.served = 0
        self._fast_group = None      # compiled launch of the group (csrc_front FastGroup), valid for exactly these lanes:
        self._fast_group_of = None   # ids of the members' FastLinear objects it was built from

    def __getstate__(self):
        """Copies and pickles carry the membership only: parked outputs and the compiled group launch (a pybind11 object) are
        rebuilt by the copy at its first call."""
        state = dict(self.__dict__)
        state.update({"_input": None, "_version": -1, "_pending": {}, "_fast_group": None, "_fast_group_of": None})
        return state

    def applicable(self, input: torch.Tensor) -> bool:
        if not input.is_cuda or math.prod(input.shape[:-1]) > GEMV_MAX_ROWS:
            return False
        if torch.is_grad_enabled() and input.requires_grad:
            return False
        if torch.compiler.is_compiling():
            return False
        first = self.members[0]
        if first.num_codebooks == 8 and first.in_group_size == 32 and math.prod(input.shape[:-1]) > 2:
            # 8x8 g32 from a few rows on: every member is better off alone on the fused MFMA kernel (its cost does not grow with the
            # rows; the shared-input table kernel is launched once per row) -- measured in the Hugging Face decode loop at 4 rows:
            # 389 tokens/s member by member, 314 through the group
            if any(m._rows_take_the_fused_8x8_op(input) for m in self.members):
                return False
        return True

    def forward(self, member: QuantizedLinear, input: torch.Tensor) -> torch.Tensor:
        idx = next(i for i, m in enumerate(self.members) if m is member)
        if self._input is input and self._version == _version(input) and idx in self._pending:
            out = self._pending.pop(idx)
            if not self._pending:
                self._input = None
            self.served += 1
            return out
        outs = self._launch(input)
        self._input, self._version = input, _version(input)  # the reference keeps `input` alive while outputs are parked
        self._pending = {i: o for i, o in enumerate(outs) if i != idx}
        self.launches += 1
        return outs[idx]

    def _compiled_group(self):
        from . import _front

        lanes = [m._fast for m in self.members]
        if any(f is None for f in lanes) or not _front.available():
            return None
        key = tuple(id(f) for f in lanes)
        if self._fast_group_of != key:
            self._fast_group, self._fast_group_of = None, key
            if all(f.kind == lanes[0].kind for f in lanes) and hasattr(_front.ext, "FastGroup"):
                try:
                    self._fast_group = _front.ext.FastGroup(lanes)
                except RuntimeError:
                    self._fast_group = None
        return self._fast_group

    def _launch(self, input: torch.Tensor) -> List[torch.Tensor]:
        from .inference_kernels import hip_kernel

        ms = self.members
        for m in ms:
            if m.gemv_op is None or m._derived_state_is_stale():
                m.prepare_matmul_op(input)
        # all members served by compiled lanes of one kind (prepacked / direct 1x16, K x 8): check x once, allocate, ONE launch,
        # without the interpreter
        # (an eager q/k/v call through the Python path below costs ~35 us of host time, as much as three separate calls)
        fg = self._compiled_group()
        if fg is not None:
            res = fg(input)  # None: not a call for the lane (a parameter changed, rows, dtype, ...) -> Python path
            if res is not None:
                return list(res)
        # every member runs on the kernel it would use alone (so outputs stay bit-identical to the unfused modules):
        # prepacked members share one packed launch, the others one direct launch
        packed_idx = [i for i, m in enumerate(ms) if m._packed_codes is not None and input.dtype == m.codebooks.dtype]
        direct_idx = [i for i in range(len(ms)) if i not in packed_idx]
        outs: List[Optional[torch.Tensor]] = [None] * len(ms)
        if packed_idx:
            sub = [ms[i] for i in packed_idx]
            planar = isinstance(sub[0]._packed_codes, hip_kernel.PlanarCodes)  # a group has one scheme: all 1x16 or all 8x8
            op = hip_kernel.code8x8_matmat_planar_multi if planar else hip_kernel.code1x16_matmat_packed_multi
            res = op(input, [m._packed_codes for m in sub], [m.codebooks for m in sub], [m.scales for m in sub], [m.bias for m in sub])
            for i, o in zip(packed_idx, res):
                outs[i] = o
        if direct_idx:
            # direct call of the op implementations: the group never runs under torch.compile tracing (applicable()), and
            # the dispatcher costs ~15 us per call for Tensor[] arguments -- as much as the launch itself in eager decode
            op = hip_kernel.code1x16_matmat_multi if ms[0].nbits_per_codebook == 16 else hip_kernel.codekx8_matmat_multi
            sub = [ms[i] for i in direct_idx]
            res = op(input, [m._canonical_codes() for m in sub], [m.codebooks for m in sub], [m.scales for m in sub],
                     [m.bias for m in sub])
            for i, o in zip(direct_idx, res):
                outs[i] = o
        return outs


def fuse_shared_input_linears(model: nn.Module,
                              patterns: Iterable[Sequence[str]] = DEFAULT_PATTERNS) -> List[SharedInputGroup]:
    """Group sibling ``QuantizedLinear`` children named like one of ``patterns`` (default: q/k/v and gate/up) under
    every sub-module of ``model``.  Siblings that cannot share a launch (other scheme, different in_features) are left
    alone.  Returns the groups created.  Idempotent; ``unfuse_shared_input_linears`` undoes it."""
    groups = []
    for parent in model.modules():
        for names in patterns:
            members = [getattr(parent, n, None) for n in names]
            if not all(isinstance(m, QuantizedLinear) for m in members):
                continue
            if any(getattr(m, "_shared_input_group", None) is not None for m in members):
                continue
            try:
                group = SharedInputGroup(members)
            except (NotImplementedError, ValueError):
                continue
            for m in members:
                m._shared_input_group = group
            groups.append(group)
    return groups


def unfuse_shared_input_linears(model: nn.Module) -> None:
    for m in model.modules():
        if isinstance(m, QuantizedLinear):
            g = m._shared_input_group
            if g is not None:  # drop the parked outputs and the input they keep alive
                g._pending, g._input = {}, None
            m._shared_input_group = None
