"""End-to-end drop-in check: a Hugging Face checkpoint whose config carries an AQLM `quantization_config` loads through
`AutoModelForCausalLM.from_pretrained` into OUR QuantizedLinear modules (transformers/integrations/aqlm.py), and -- on
the GPU -- produces the logits / greedy tokens of the equivalent dense model."""
import json
import os

import numpy as np
import pytest
import torch

transformers = pytest.importorskip("transformers")
safetensors_torch = pytest.importorskip("safetensors.torch")

from oracle import aqlm_oracle as orc  # noqa: E402  (checker only)

HID, INTER, LAYERS, HEADS, VOCAB = 256, 512, 2, 4, 128
SCHEME = dict(in_group_size=8, out_group_size=1, num_codebooks=1, nbits_per_codebook=16)


def _build_checkpoint(tmp_path):
    """Random dense Llama whose every decoder Linear equals the dequantised weight of random AQLM tensors; writes the
    QUANTIZED checkpoint (codes / codebooks / scales, keys as convert_to_hf.py:57-68) and returns the dense model."""
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=HID, intermediate_size=INTER, num_hidden_layers=LAYERS, num_attention_heads=HEADS,
                      num_key_value_heads=HEADS, vocab_size=VOCAB, max_position_embeddings=64, tie_word_embeddings=False)
    dense = LlamaForCausalLM(cfg).to(torch.float16)
    qstate = {}
    seed = 100
    for name, mod in dense.named_modules():
        if isinstance(mod, torch.nn.Linear) and "lm_head" not in name:
            seed += 1
            L = orc.make_layer(seed, mod.in_features, mod.out_features, 1, 16, 8, batch=1, bias=False, edge_codes=False)
            cb = (L["codebooks"].astype(np.float32) * 0.05).astype(np.float16)   # keep activations tame
            sc = (np.abs(L["scales"].astype(np.float32)) * 0.2 + 0.05).astype(np.float16)
            W = orc.dequantize_weight(L["codes_unsigned"], cb, sc).astype(np.float16)
            with torch.no_grad():
                mod.weight.copy_(torch.from_numpy(W))
            qstate[f"{name}.codes"] = torch.from_numpy(L["codes"])
            qstate[f"{name}.codebooks"] = torch.from_numpy(cb)
            qstate[f"{name}.scales"] = torch.from_numpy(sc)
    for k, v in dense.state_dict().items():
        base = k.rsplit(".", 1)[0]
        if f"{base}.codes" not in qstate:
            qstate[k] = v.contiguous()
    out = tmp_path / "aqlm_tiny_llama"
    out.mkdir()
    safetensors_torch.save_file(qstate, str(out / "model.safetensors"), metadata={"format": "pt"})
    c = cfg.to_dict()
    c["architectures"] = ["LlamaForCausalLM"]
    c["torch_dtype"] = "float16"
    c["quantization_config"] = dict(quant_method="aqlm", linear_weights_not_to_quantize=["lm_head"], **SCHEME)
    (out / "config.json").write_text(json.dumps(c))
    return dense, str(out)


def _load(path, device):
    from transformers import AutoModelForCausalLM

    kw = dict(torch_dtype=torch.float16, low_cpu_mem_usage=True)
    try:
        return AutoModelForCausalLM.from_pretrained(path, device_map=device, **kw)
    except TypeError:
        return AutoModelForCausalLM.from_pretrained(path, **kw).to(device)


def test_hf_checkpoint_loads_into_our_modules_cpu(tmp_path):
    import aqlm_amd

    dense, path = _build_checkpoint(tmp_path)
    try:
        model = _load(path, "cpu")
    except Exception as e:  # pragma: no cover - depends on the installed transformers' CPU policy for AQLM
        pytest.skip(f"transformers refuses to load AQLM checkpoints on CPU here: {type(e).__name__}: {e}")
    q = model.model.layers[0].self_attn.q_proj
    assert type(q) is aqlm_amd.QuantizedLinear
    assert q.codes.dtype == torch.int16 and tuple(q.codes.shape) == (HID, HID // 8, 1)
    assert not q.codes.is_meta and not q.codebooks.is_meta
    assert isinstance(model.lm_head, torch.nn.Linear)


@pytest.mark.gpu
def test_hf_generate_matches_dense_model_gpu(tmp_path):
    import aqlm_amd

    assert torch.cuda.is_available()
    dense, path = _build_checkpoint(tmp_path)
    model = _load(path, "cuda:0")
    n_q = sum(isinstance(m, aqlm_amd.QuantizedLinear) for m in model.modules())
    assert n_q == LAYERS * 7
    dense = dense.to("cuda:0")
    ids = torch.randint(0, VOCAB, (1, 9), generator=torch.Generator().manual_seed(5)).to("cuda:0")
    with torch.no_grad():
        lq = model(ids).logits.float()       # 9 rows -> gemm ops
        ld = dense(ids).logits.float()
        lq1 = model(ids[:, :1]).logits.float()  # 1 row -> gemv ops
        ld1 = dense(ids[:, :1]).logits.float()
    for a, b, what in ((lq, ld, "prefill"), (lq1, ld1, "decode")):
        rel = (a - b).abs().mean() / b.abs().mean()
        assert rel < 2e-2, f"{what}: logits differ from the dense model by {rel:.3e}"
    with torch.no_grad():
        tq = model.generate(ids, max_new_tokens=6, do_sample=False)
        td = dense.generate(ids, max_new_tokens=6, do_sample=False)
    assert tq.shape == td.shape
    agree = (tq == td).float().mean().item()
    assert agree >= 0.8, f"greedy tokens agree only {agree:.2f}"
