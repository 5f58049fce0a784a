"""CPU-only checks of the host-side mirror of the reference interface (module, selector, op schemas, utils)."""
import importlib.metadata

import numpy as np
import pytest
import torch

import aqlm
import aqlm_amd
from aqlm_amd import utils
from oracle import aqlm_oracle as orc


def test_dropin_import_paths():
    from aqlm import QuantizedLinear, optimize_for_training  # noqa: F401
    from aqlm.inference import QuantizedLinear as Q2
    from aqlm.inference_kernels import get_backward_pass_kernel, get_forward_pass_kernel  # noqa: F401
    from aqlm.inference_kernels.kernel_selector import get_forward_pass_kernel as g2  # noqa: F401
    from aqlm.utils import _dequantize_weight, get_int_dtype, pack_int_data, unpack_int_data  # noqa: F401

    assert Q2 is aqlm_amd.QuantizedLinear
    # what quantizer_aqlm.py:63-72 probes
    assert importlib.metadata.version("aqlm") >= "1.0.2"


@pytest.mark.parametrize("K,nbits,g,bias", [(1, 16, 8, True), (2, 8, 8, False), (8, 8, 32, True), (1, 16, 16, False)])
def test_quantized_linear_parameters_match_reference_contract(K, nbits, g, bias):
    m = aqlm.QuantizedLinear(1024, 256, g, 1, K, nbits, bias=bias, device="meta", dtype=torch.float16)
    sd = {k: (tuple(v.shape), v.dtype) for k, v in m.state_dict().items()}
    want = {
        "codebooks": ((K, 2**nbits, 1, g), torch.float16),
        "codes": ((256, 1024 // g, K), torch.int8 if nbits <= 8 else torch.int16),
        "scales": ((256, 1, 1, 1), torch.float16),
    }
    if bias:
        want["bias"] = ((256,), torch.float16)
    assert sd == want
    assert list(m.state_dict()) == list(want)  # registration order (reference inference.py:39-61)
    assert all(not p.requires_grad for p in m.parameters())
    for attr in ("in_features", "out_features", "in_group_size", "out_group_size", "num_codebooks",
                 "nbits_per_codebook", "codebook_size"):
        assert hasattr(m, attr)
    assert m.codebook_size == 2**nbits and m.gemv_op is None and m.gemm_op is None


def test_state_dict_roundtrip_cpu():
    m = aqlm.QuantizedLinear(256, 64, 8, 1, 2, 8, bias=True, dtype=torch.float16)
    L = orc.make_layer(3, 256, 64, 2, 8, 8)
    sd = {"codes": torch.from_numpy(L["codes"]), "codebooks": torch.from_numpy(L["codebooks"]),
          "scales": torch.from_numpy(L["scales"]), "bias": torch.from_numpy(L["bias"])}
    m.load_state_dict(sd)
    assert torch.equal(m.codes, sd["codes"]) and m.codes.dtype == torch.int8


def test_host_tensors_take_the_cpu_branch_and_other_devices_raise():
    """Host tensors are served by the native CPU kernels (reference kernel_selector.py:95-102); a device the package
    has no kernels for raises instead of silently dequantising."""
    from aqlm_amd.inference_kernels import cpu_kernel, kernel_selector

    assert aqlm.get_forward_pass_kernel(torch.zeros(2, 256, 1, 8), False) is cpu_kernel.cpu_gemm_lut
    assert aqlm.get_forward_pass_kernel(torch.zeros(1, 65536, 1, 8), False) is cpu_kernel.cpu_gemv_1xn
    assert aqlm.get_forward_pass_kernel(torch.zeros(2, 256, 2, 8), False) is kernel_selector._torch_forward   # out_group_size 2
    assert aqlm.get_backward_pass_kernel(torch.zeros(2, 256, 1, 8), True) is kernel_selector._torch_backward


def test_selector_table_on_meta_codebooks_raises():
    cb = torch.empty(1, 65536, 1, 8, device="meta", dtype=torch.float16)
    with pytest.raises(NotImplementedError):
        aqlm.get_forward_pass_kernel(cb, False)


def test_op_schemas_and_fake_impls():
    from aqlm_amd.inference_kernels import hip_kernel  # noqa: F401  (registers the ops)

    x = torch.empty(2, 3, 512, device="meta", dtype=torch.bfloat16)
    codes = torch.empty(96, 64, 1, device="meta", dtype=torch.int16)
    cb = torch.empty(1, 65536, 1, 8, device="meta", dtype=torch.bfloat16)
    sc = torch.empty(96, 1, 1, 1, device="meta", dtype=torch.bfloat16)
    for name in ("code1x16_matmat", "code1x16_matmat_dequant", "code2x8_matmat", "code2x8_matmat_dequant",
                 "code1x8_matmat", "code1x8_matmat_dequant", "codekx8_matmat", "generic_matmat"):
        y = getattr(torch.ops.aqlm, name)(x, codes, cb, sc, None)
        assert y.shape == (2, 3, 96) and y.dtype == torch.bfloat16 and y.device.type == "meta"
    g = torch.empty(2, 3, 96, device="meta", dtype=torch.bfloat16)
    for name in ("code1x16_matmat_dequant_transposed", "code2x8_matmat_dequant_transposed",
                 "code1x8_matmat_dequant_transposed"):
        assert getattr(torch.ops.aqlm, name)(g, codes, cb, sc, None).shape == (2, 3, 512)
    schema = str(torch.ops.aqlm.code1x16_matmat.default._schema)
    assert "Tensor input, Tensor codes, Tensor codebooks, Tensor scales, Tensor? bias" in schema
    from aqlm.inference_kernels.cuda_kernel import CUDA_KERNEL

    for fn in ("code1x16_matmat", "code2x8_matmat", "code1x16_dequant", "code2x8_dequant", "code1x8_dequant"):
        assert callable(getattr(CUDA_KERNEL, fn))


def test_utils_against_reference_golden(golden):
    for nbits in (8, 12, 16):
        vals = torch.from_numpy(golden[f"kat/pack{nbits}_in"])
        keep = vals.clone()
        packed = utils.pack_int_data(vals, nbits)
        assert torch.equal(vals, keep), "pack_int_data must not mutate its argument"
        np.testing.assert_array_equal(packed.numpy(), golden[f"kat/pack{nbits}_out"])
        np.testing.assert_array_equal(utils.unpack_int_data(packed, nbits).numpy(), golden[f"kat/unpack{nbits}_out"])
    for name in ("c2x8g8_f16", "c2x8g8_og2_f32", "c8x8g32_f16"):
        seed, fin, fout, K, nbits, g, batch, bias, ogs = [int(v) for v in golden[f"{name}/cfg"]]
        L = orc.make_layer(seed, fin, fout, K, nbits, g, batch=batch, bias=bool(bias), out_group_size=ogs,
                           float_dtype=np.float32 if "f32" in name else np.float16)
        W = utils._dequantize_weight(torch.from_numpy(L["codes_unsigned"]), torch.from_numpy(L["codebooks"]).float(),
                                     torch.from_numpy(L["scales"]).float())
        np.testing.assert_allclose(W.numpy(), golden[f"{name}/W_ref32"], rtol=1e-5, atol=1e-5)


def test_hf_integration_replaces_linears_with_our_module():
    """transformers/integrations/aqlm.py:27-70 builds our QuantizedLinear on the meta device."""
    transformers = pytest.importorskip("transformers")
    try:
        from transformers import AqlmConfig, LlamaConfig, LlamaForCausalLM
        from transformers.integrations.aqlm import replace_with_aqlm_linear
    except Exception as e:  # pragma: no cover
        pytest.skip(f"transformers AQLM integration not importable: {e}")
    cfg = LlamaConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=4,
                      num_key_value_heads=4, vocab_size=64)
    with torch.device("meta"):
        model = LlamaForCausalLM(cfg)
    qcfg = AqlmConfig(in_group_size=8, out_group_size=1, num_codebooks=1, nbits_per_codebook=16,
                      linear_weights_not_to_quantize=["lm_head"])
    try:
        out = replace_with_aqlm_linear(model, quantization_config=qcfg, modules_to_not_convert=["lm_head"])
    except TypeError:
        out = replace_with_aqlm_linear(model, quantization_config=qcfg, linear_weights_not_to_quantize=["lm_head"])
    model = out[0] if isinstance(out, tuple) else out
    q = model.model.layers[0].self_attn.q_proj
    assert type(q) is aqlm_amd.QuantizedLinear
    assert q.codes.shape == (128, 16, 1) and q.codes.dtype == torch.int16
    assert not isinstance(model.lm_head, aqlm_amd.QuantizedLinear)


def test_shared_input_grouping_rules_on_meta_modules():
    """fuse_shared_input_linears groups q/k/v and gate/up siblings that can share a launch and nothing else."""
    import torch.nn as nn

    def ql(fin, fout, K=1, nbits=16, g=8):
        return aqlm.QuantizedLinear(fin, fout, g, 1, K, nbits, bias=False, device="meta", dtype=torch.float16)

    class Attn(nn.Module):
        def __init__(self):
            super().__init__()
            self.q_proj, self.k_proj, self.v_proj, self.o_proj = ql(512, 512), ql(512, 128), ql(512, 128), ql(512, 512)

    class Mlp(nn.Module):
        def __init__(self, K=1, nbits=16):
            super().__init__()
            self.gate_proj, self.up_proj = ql(512, 1024, K, nbits), ql(512, 1024, K, nbits)
            self.down_proj = ql(1024, 512, K, nbits)

    class Odd(nn.Module):  # siblings with different in_features, or a dense member, stay unfused
        def __init__(self):
            super().__init__()
            self.gate_proj, self.up_proj = ql(512, 1024), ql(256, 1024)
            self.q_proj, self.k_proj, self.v_proj = ql(512, 512), nn.Linear(8, 8), ql(512, 512)

    class Mlp8x8(nn.Module):  # 4x8 g16: no shared-input kernel
        def __init__(self):
            super().__init__()
            self.gate_proj, self.up_proj = ql(512, 1024, 4, 8, 16), ql(512, 1024, 4, 8, 16)

    model = nn.ModuleList([Attn(), Mlp(), Mlp(2, 8), Odd(), Mlp8x8()])
    keys_before = list(model.state_dict().keys()) if False else [n for n, _ in model.named_parameters()]
    groups = aqlm.fuse_shared_input_linears(model)
    assert [[m.out_features for m in g.members] for g in groups] == [[512, 128, 128], [1024, 1024], [1024, 1024]]
    assert groups[2].members[0].nbits_per_codebook == 8      # 2x8 g8 siblings share a launch too
    assert model[0].o_proj._shared_input_group is None and model[1].down_proj._shared_input_group is None
    assert model[4].gate_proj._shared_input_group is None      # 4x8 g16: not covered by the shared-input kernels
    assert model[3].gate_proj._shared_input_group is None and model[3].q_proj._shared_input_group is None
    assert [n for n, _ in model.named_parameters()] == keys_before   # no parameters added or renamed
    assert aqlm.fuse_shared_input_linears(model) == []
    # CPU / meta inputs never take the group path (host tensors go to the CPU kernels one module at a time)
    assert not groups[0].applicable(torch.zeros(1, 512))
    with pytest.raises(ValueError):
        aqlm.SharedInputGroup([model[0].q_proj])
    with pytest.raises(ValueError):
        aqlm.SharedInputGroup([model[0].q_proj, model[1].down_proj])
    with pytest.raises(NotImplementedError):
        aqlm.SharedInputGroup([model[4].gate_proj, model[4].up_proj])
    with pytest.raises(ValueError):
        aqlm.SharedInputGroup([model[1].gate_proj, model[2].up_proj])   # 1x16 next to 2x8
    aqlm.unfuse_shared_input_linears(model)
    assert model[0].q_proj._shared_input_group is None


def test_checkpoint_tooling_validates_and_upgrades_configs():
    from aqlm.checkpoint import quantization_config_from, upgrade_legacy_config, validate_quantized_state_dict

    scheme = dict(nbits_per_codebook=16, num_codebooks=1, out_group_size=1, in_group_size=8)
    hf_cfg = {"model_type": "llama", "quantization_config": dict(scheme, quant_method="aqlm",
                                                                linear_weights_not_to_quantize=["lm_head.weight"])}
    legacy = {"model_type": "llama", "aqlm": dict(scheme)}      # benchmark_generate_cpu.py:68-73 style
    assert quantization_config_from(hf_cfg) == quantization_config_from(legacy)
    up = upgrade_legacy_config(legacy)
    assert "aqlm" not in up and up["quantization_config"]["quant_method"] == "aqlm" and up["torch_dtype"] == "float16"
    assert "aqlm" in legacy                                       # input untouched
    assert upgrade_legacy_config(hf_cfg) == hf_cfg
    with pytest.raises(ValueError):
        quantization_config_from({"model_type": "llama"})
    with pytest.raises(ValueError):
        quantization_config_from({"quantization_config": {"quant_method": "gptq"}})
    with pytest.raises(ValueError):
        quantization_config_from({"aqlm": {"nbits_per_codebook": 16}})

    m = aqlm.QuantizedLinear(64, 32, 8, 1, 1, 16, bias=True, dtype=torch.float16)
    with torch.no_grad():
        m.codes.zero_(); m.codebooks.normal_(); m.scales.fill_(1); m.bias.zero_()
    sd = {f"model.layers.0.q.{k}": v.detach().clone() for k, v in m.state_dict().items()}
    sd["lm_head.weight"] = torch.zeros(4, 4)
    assert validate_quantized_state_dict(sd, hf_cfg) == []
    assert validate_quantized_state_dict(sd, scheme) == []      # a bare scheme dict works too
    bad = dict(sd)
    bad["model.layers.0.q.codes"] = sd["model.layers.0.q.codes"].to(torch.int8)
    bad["model.layers.0.q.scales"] = sd["model.layers.0.q.scales"].reshape(-1)
    bad["model.layers.0.q.codebooks"] = sd["model.layers.0.q.codebooks"][:, :256]
    bad["model.layers.0.q.weight"] = torch.zeros(32, 64)
    probs = validate_quantized_state_dict(bad, hf_cfg)
    assert len(probs) == 4 and any("dtype" in p for p in probs) and any("dense" in p for p in probs)
    del bad["model.layers.0.q.scales"]
    assert any("without scales" in p for p in validate_quantized_state_dict(bad, hf_cfg))
    assert validate_quantized_state_dict({"w": torch.zeros(1)}, hf_cfg) == ["no `<name>.codes` tensors: not an AQLM checkpoint"]
    # 12-bit codes live in int16 containers: out-of-range container values are caught
    s12 = dict(scheme, nbits_per_codebook=12)
    m12 = aqlm.QuantizedLinear(64, 16, 8, 1, 1, 12, bias=False, dtype=torch.float16)
    with torch.no_grad():
        m12.codes.fill_(2047); m12.codebooks.normal_(); m12.scales.fill_(1)
    sd12 = {f"l.{k}": v.detach().clone() for k, v in m12.state_dict().items()}
    assert validate_quantized_state_dict(sd12, s12) == []
    sd12["l.codes"][0, 0, 0] = 2048
    assert any("container range" in p for p in validate_quantized_state_dict(sd12, s12))
    sd12["l.codebooks"][0, 0, 0, 0] = float("nan")
    assert any("non-finite" in p for p in validate_quantized_state_dict(sd12, s12))
    # memory report / eager repack: the MI355X repack refuses host models loudly
    from aqlm.checkpoint import memory_report, prepack_model

    rep = memory_report(torch.nn.Sequential(m))
    assert rep["quantized_linears"] == 1 and rep["codes"] == 32 * 8 * 2 and rep["prepacked"] == 0
    with pytest.raises(NotImplementedError):
        prepack_model(torch.nn.Sequential(m))


@pytest.mark.parametrize("K,nbits,g", [(2, 8, 8), (1, 16, 8)])
def test_module_copies_and_pickles_after_a_forward_cpu(K, nbits, g):
    """after the first forward the module holds derived state (autograd ops, a lambda, permuted codes and, on
    the GPU, a pybind11 fast lane); `copy.deepcopy`, `pickle` and `torch.save(module)` must still work and the copy must compute
    the same thing."""
    import copy
    import io
    import pickle

    m = aqlm.QuantizedLinear(256, 64, g, 1, K, nbits, bias=True, dtype=torch.float32)
    L = orc.make_layer(5, 256, 64, K, nbits, g, float_dtype=np.float32)
    m.load_state_dict({"codes": torch.from_numpy(L["codes"]), "codebooks": torch.from_numpy(L["codebooks"]),
                       "scales": torch.from_numpy(L["scales"]), "bias": torch.from_numpy(L["bias"])})
    x = torch.from_numpy(L["x"])
    y = m(x)
    assert m.gemv_op is not None
    for clone in (copy.deepcopy(m), pickle.loads(pickle.dumps(m))):
        assert clone.gemv_op is None and clone._fast is None and clone._cpu_codes_alt is None
        assert torch.equal(clone.codes, m.codes)
        assert torch.equal(clone(x), y)
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    assert torch.equal(torch.load(buf, weights_only=False)(x), y)
    assert m.gemv_op is not None  # the original keeps its derived state


def test_fused_8x8_switch_rows_follow_the_measured_cost_model():
    """hip_kernel.fused_8x8_min_rows: the row count from which an 8x8 g32 layer leaves the look-up-table matvec for the fused MFMA
    kernel -- the values the round-5 measurements gave (profiles/r05_gemm_8x8_mfma.log), monotone in the layer's work per round,
    never below 2, pinned by FUSED_8X8_MFMA_MIN_ROWS; and the shared-input group steps aside exactly when a member would switch."""
    from aqlm_amd.inference_kernels import hip_kernel as hk

    assert hk.fused_8x8_min_rows(4096, 4096) == 3 and hk.fused_8x8_min_rows(11008, 4096) == 5 and hk.fused_8x8_min_rows(4096, 11008) == 5
    assert hk.fused_8x8_min_rows(1024, 4096) == 3 and hk.fused_8x8_min_rows(16, 2048) >= 2
    rows = [hk.fused_8x8_min_rows(m, 4096) for m in (4096, 8192, 16384, 32768)]
    assert rows == sorted(rows)                      # more rounds of tiles -> the per-row table kernel stays ahead for longer
    old = hk.FUSED_8X8_MFMA_MIN_ROWS
    try:
        hk.FUSED_8X8_MFMA_MIN_ROWS = 4
        assert hk.fused_8x8_min_rows(4096, 4096) == 4 and hk.fused_8x8_min_rows(28672, 8192) == 4
    finally:
        hk.FUSED_8X8_MFMA_MIN_ROWS = old
