"""Loader-side tooling for AQLM checkpoints.

The on-disk format is the reference's (convert_to_hf.py:57-100): per quantized Linear ``<name>.codes`` (int8 / int16
two's-complement containers of unsigned indices, ``[out/og, in/ig, K]``), ``<name>.codebooks`` (``[K, 2**nbits, og, ig]``),
``<name>.scales`` (``[out/og, 1, 1, 1]``), optional ``<name>.bias``; the scheme lives in ``config.json`` under
``quantization_config`` (``quant_method == "aqlm"``) or, in checkpoints that predate the transformers integration, under
a top-level ``aqlm`` key (benchmark/benchmark_generate_cpu.py:68-73).  Nothing here changes the format: these helpers
validate a checkpoint before it reaches the kernels, translate the legacy config, and run the MI355X load-time repack
(``prepack_model``) eagerly instead of at the first forward.
"""
from __future__ import annotations

import copy
from typing import Dict, List, Mapping, Optional

import torch

from .utils import get_int_dtype

_SCHEME_KEYS = ("nbits_per_codebook", "num_codebooks", "out_group_size", "in_group_size")


def quantization_config_from(config: Mapping) -> Dict:
    """The AQLM scheme of a ``config.json`` dict, whichever of the two layouts it uses."""
    q = config.get("quantization_config")
    if q is not None:
        if q.get("quant_method", "aqlm") != "aqlm":
            raise ValueError(f"quantization_config.quant_method is {q.get('quant_method')!r}, not 'aqlm'")
        src = q
    elif config.get("aqlm") is not None:
        src = config["aqlm"]
    else:
        raise ValueError("config has neither `quantization_config` nor the legacy `aqlm` entry")
    missing = [k for k in _SCHEME_KEYS if k not in src]
    if missing:
        raise ValueError(f"AQLM config lacks {missing}")
    out = {k: int(src[k]) for k in _SCHEME_KEYS}
    if out["nbits_per_codebook"] < 1 or out["nbits_per_codebook"] > 16 or min(out.values()) < 1:
        raise ValueError(f"implausible AQLM scheme {out}")
    out["quant_method"] = "aqlm"
    out["linear_weights_not_to_quantize"] = list(src.get("linear_weights_not_to_quantize", ["lm_head.weight"]))
    return out


def upgrade_legacy_config(config: Mapping) -> Dict:
    """Copy of ``config`` with a transformers-style ``quantization_config`` (what update_config writes,
    convert_to_hf.py:85-95) built from the legacy top-level ``aqlm`` entry; configs that already have one are returned
    unchanged (copied)."""
    out = copy.deepcopy(dict(config))
    if out.get("quantization_config") is None:
        out["quantization_config"] = quantization_config_from(config)
        out.pop("aqlm", None)
        out.setdefault("torch_dtype", "float16")
    return out


def validate_quantized_state_dict(state_dict: Mapping[str, torch.Tensor], config: Mapping,
                                  check_values: bool = True) -> List[str]:
    """Check every quantized Linear of ``state_dict`` against the scheme in ``config`` (a config.json dict or a
    quantization_config dict).  Returns the list of problems found (empty = loadable by ``QuantizedLinear``)."""
    q = quantization_config_from(config if ("quantization_config" in config or "aqlm" in config)
                                 else {"quantization_config": dict(config)})
    K, nbits, og, ig = q["num_codebooks"], q["nbits_per_codebook"], q["out_group_size"], q["in_group_size"]
    problems: List[str] = []
    bases = sorted(k[: -len(".codes")] for k in state_dict if k.endswith(".codes"))
    if not bases:
        problems.append("no `<name>.codes` tensors: not an AQLM checkpoint")
    for b in bases:
        codes = state_dict[f"{b}.codes"]
        cb, sc = state_dict.get(f"{b}.codebooks"), state_dict.get(f"{b}.scales")
        if cb is None or sc is None:
            problems.append(f"{b}: codes without {'codebooks' if cb is None else 'scales'}")
            continue
        if f"{b}.weight" in state_dict:
            problems.append(f"{b}: both a dense `weight` and AQLM tensors")
        if codes.dtype != get_int_dtype(nbits):
            problems.append(f"{b}.codes: dtype {codes.dtype}, expected {get_int_dtype(nbits)} for {nbits}-bit codes")
        if codes.dim() != 3 or codes.shape[2] != K:
            problems.append(f"{b}.codes: shape {tuple(codes.shape)}, expected [out/{og}, in/{ig}, {K}]")
            continue
        n_out, n_in = codes.shape[0], codes.shape[1]
        if tuple(cb.shape) != (K, 2**nbits, og, ig):
            problems.append(f"{b}.codebooks: shape {tuple(cb.shape)}, expected {(K, 2**nbits, og, ig)}")
        if not cb.is_floating_point() or not sc.is_floating_point():
            problems.append(f"{b}: codebooks / scales must be floating point")
        if tuple(sc.shape) != (n_out, 1, 1, 1):
            problems.append(f"{b}.scales: shape {tuple(sc.shape)}, expected {(n_out, 1, 1, 1)}")
        bias = state_dict.get(f"{b}.bias")
        if bias is not None and tuple(bias.shape) != (n_out * og,):
            problems.append(f"{b}.bias: shape {tuple(bias.shape)}, expected {(n_out * og,)}")
        if check_values and codes.numel() and not codes.is_meta and nbits not in (8, 16):
            # containers wider than the code: pack_int_data (utils.py:18-26) leaves values in [-2^(n-1), 2^(n-1))
            lo, hi = int(codes.min()), int(codes.max())
            if lo < -(2 ** (nbits - 1)) or hi >= 2 ** (nbits - 1):
                problems.append(f"{b}.codes: values [{lo}, {hi}] outside the {nbits}-bit container range")
        if check_values and not cb.is_meta and cb.is_floating_point() and not torch.isfinite(cb).all():
            problems.append(f"{b}.codebooks: non-finite entries")
    return problems


def memory_report(model: torch.nn.Module) -> Dict[str, float]:
    """Bytes held by the QuantizedLinear modules of ``model``: checkpoint tensors and the derived prepacked buffers,
    plus the resulting bits per weight of the code storage (canonical 1x16 g8 = 2.0) and of everything."""
    from .inference import QuantizedLinear

    rep = {"quantized_linears": 0, "codes": 0, "codebooks": 0, "scales_bias": 0, "prepacked_layers": 0, "prepacked": 0,
           "codes_dropped_layers": 0, "weights": 0}
    for m in model.modules():
        if isinstance(m, QuantizedLinear):
            rep["quantized_linears"] += 1
            rep["weights"] += m.in_features * m.out_features
            rep["codes"] += m.codes.numel() * m.codes.element_size()
            rep["codebooks"] += m.codebooks.numel() * m.codebooks.element_size()
            rep["scales_bias"] += m.scales.numel() * m.scales.element_size()
            if m.bias is not None:
                rep["scales_bias"] += m.bias.numel() * m.bias.element_size()
            if m._packed_codes is not None:
                rep["prepacked_layers"] += 1
                rep["prepacked"] += m._packed_codes.numel()
            if m._codes_dropped:
                rep["codes_dropped_layers"] += 1
    if rep["weights"]:
        rep["code_bits_per_weight"] = 8.0 * (rep["codes"] + rep["prepacked"]) / rep["weights"]
        rep["total_bits_per_weight"] = 8.0 * (rep["codes"] + rep["prepacked"] + rep["codebooks"] + rep["scales_bias"]) / rep["weights"]
    return rep


_DEFAULT = object()


def prepack_model(model: torch.nn.Module, min_codes: Optional[int] = None, drop_canonical=_DEFAULT, compact: bool = False,
                  thorough: bool = False) -> Dict[str, float]:
    """Resolve the kernels and run the load-time repack of every eligible QuantizedLinear now (GPU-resident modules
    only) instead of at its first forward; ``min_codes`` overrides ``inference.PREPACK_MIN_CODES`` for this call.
    ``drop_canonical``: free the checkpoint-layout ``codes`` of repacked layers -- the packed buffer is lossless; ``state_dict()``,
    the > 6-row ops and backward rebuild the codes through ``aqlm_hip_unpack_1x16``.  That unpack is not cheap (bench.py
    ``detail.unpack_1x16_us``: 32 us for a 4096 x 4096 layer, 94 us for 4096 -> 11008 -- 2.4-2.9 x the 16-row op it would precede), so:
      * default (round 6: this is the deployment call, and a second copy of the codes is what the format exists to avoid): the
        1x16 layers -- whose packed buffer is a second, larger copy -- drop their codes NOW (4.8 instead of 6.8 resident bits per
        weight) and a layer that is later called with 7+ rows / backward takes them back for good at that call (one unpack, not one
        per call): decode-only deployments stay at one copy, everything else converges to round 5's behaviour.  Planar 8x8 layers
        keep both (same size as the canonical codes, and the fused 8x8 g32 MFMA kernel reads the canonical layout);
      * ``True``: drop everywhere, planar 8x8 included, and never restore (every call that needs the codes unpacks transiently):
        the smallest footprint, 2.0 / 4.8 bits per weight;
      * ``False``: keep everything (training, speculative verification at 7+ rows on every step).
    A module that was only packed lazily at its first forward keeps both copies.
    ``compact=True`` packs 1x16 g8 layers with 24-bit entries (3.5 instead of 4.5 bytes per code: a 70B model holds 30 GB of
    packed codes instead of 39; measured 1-5 % slower matvecs).  ``thorough=True`` runs the local search of the entry order on
    every layer, not only on those of <= 8 Mi codes (1-3 % faster matvecs on the big layers for ~5x their prepack time).
    Returns ``memory_report(model)`` plus ``prepack_seconds``."""
    import time

    from . import _native, inference
    from .inference import QuantizedLinear

    explicit_drop = drop_canonical is not _DEFAULT
    if not explicit_drop:
        drop_canonical = True
    old = inference.PREPACK_MIN_CODES
    if min_codes is not None:
        inference.PREPACK_MIN_CODES = int(min_codes)
    old_eb, old_arr = _native.get_tuning("packed_entry_bytes"), _native.get_tuning("packed_arrange")
    if compact:
        _native.set_tuning("packed_entry_bytes", 3)
    if thorough:
        _native.set_tuning("packed_arrange", 3)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    try:
        for m in model.modules():
            if isinstance(m, QuantizedLinear):
                if not m.codes.is_cuda:
                    raise NotImplementedError("prepack_model needs the model on an MI355X (`model.to('cuda')` first)")
                m.prepare_matmul_op(m.codebooks)
                # explicit True: every repacked layer (planar 8x8 included); the default drops where a SECOND copy exists -- the
                # slice-bucketed 1x16 buffers (4.8 bits per weight next to the 2.0 of the codes).  Planar 8x8 codes are the same
                # size as the canonical ones and the fused 8x8 g32 MFMA kernel (3+ rows) reads the canonical layout: those stay
                # unless asked for.
                if drop_canonical is True and (explicit_drop or m.nbits_per_codebook == 16):
                    m.drop_canonical_codes(strict=explicit_drop)
    finally:
        inference.PREPACK_MIN_CODES = old
        _native.set_tuning("packed_entry_bytes", old_eb)
        _native.set_tuning("packed_arrange", old_arr)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    rep = memory_report(model)
    rep["prepack_seconds"] = time.perf_counter() - t0
    return rep


def enable_dense_below_rows(model: torch.nn.Module, rows: int) -> int:
    """Opt-in escape hatch: calls with 7 .. rows - 1 input rows run as a dense GEMM on a cached fp16 / bf16 copy
    of W (built at the first such call: +2 bytes per weight resident) instead of the fused dequant -> MFMA op, which is slower
    than a dense GEMM below ~128 rows on MI355X.  ``rows = 0`` switches it off and frees the copies.  Returns the number of layers
    touched."""
    from .inference import QuantizedLinear

    n = 0
    for m in model.modules():
        if isinstance(m, QuantizedLinear):
            m.prefer_dense_below_rows = int(rows)
            if not rows:
                m._dense = None
            n += 1
    return n
