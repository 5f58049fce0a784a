#!/usr/bin/env python
"""Decode tokens/s of a Hugging Face Llama whose decoder Linears are AQLM ``QuantizedLinear`` modules (random weights of
the real shapes), the way the reference measures it in benchmark/generate_benchmark.py:67-79 and in
notebooks/aqlm_cuda_graph.ipynb: greedy decode, one token at a time, static KV cache, optionally captured into a
hipGraph, optionally with shared-input launches (aqlm.fuse_shared_input_linears: q/k/v and gate/up in one launch).

    python tools/decode_benchmark.py --model llama3-8b --scheme 1x16g8 --tokens 64

Prints one JSON object.  GPU only (the AQLM modules have no CPU path); --dense-only --device cpu exercises the loop on CPU.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

MODELS = {
    # hidden, intermediate, layers, heads, kv_heads, vocab
    "llama3-8b": (4096, 14336, 32, 32, 8, 128256),
    "llama2-7b": (4096, 11008, 32, 32, 32, 32000),
    "tiny": (256, 512, 2, 4, 2, 512),
}
SCHEMES = {"1x16g8": (1, 16, 8), "1x16g16": (1, 16, 16), "2x8g8": (2, 8, 8), "1x8g8": (1, 8, 8), "8x8g32": (8, 8, 32)}


def build_dense(name, device, dtype, max_len):
    from transformers import LlamaConfig, LlamaForCausalLM

    hid, inter, layers, heads, kv, vocab = MODELS[name]
    cfg = LlamaConfig(hidden_size=hid, intermediate_size=inter, num_hidden_layers=layers, num_attention_heads=heads,
                      num_key_value_heads=kv, vocab_size=vocab, max_position_embeddings=max(max_len, 64),
                      tie_word_embeddings=False)
    torch.manual_seed(0)
    with torch.device(device):
        old = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        try:
            model = LlamaForCausalLM(cfg)
        finally:
            torch.set_default_dtype(old)
    return model.eval()


def quantize_in_place(model, scheme, device, dtype):
    """Swap every decoder nn.Linear for a QuantizedLinear of the same shape with random codes / codebooks (the
    reference's fake-model trick, generate_benchmark.py:67-79).  lm_head and embeddings stay dense, as in AQLM models."""
    from aqlm import QuantizedLinear

    K, nbits, g = SCHEMES[scheme]
    gen = torch.Generator(device=device).manual_seed(1)
    n = 0
    for layer in model.model.layers:
        for parent in (layer.self_attn, layer.mlp):
            for cname, child in list(parent.named_children()):
                if not isinstance(child, torch.nn.Linear):
                    continue
                q = QuantizedLinear(child.in_features, child.out_features, g, 1, K, nbits, bias=child.bias is not None,
                                    device=device, dtype=dtype)
                lo, hi = -(2 ** (nbits - 1)), 2 ** (nbits - 1)
                with torch.no_grad():
                    q.codes.copy_(torch.randint(lo, hi, q.codes.shape, generator=gen, device=device, dtype=torch.int32))
                    q.codebooks.copy_(torch.randn(q.codebooks.shape, generator=gen, device=device) * (0.02 / K**0.5))
                    q.scales.fill_(1.0)
                    if q.bias is not None:
                        q.bias.zero_()
                setattr(parent, cname, q)
                n += 1
    return n


class Decoder:
    """Greedy single-token decode with a static KV cache; every tensor the step touches is allocated once, so the step
    can be captured into a hipGraph (what notebooks/aqlm_cuda_graph.ipynb does with torch.compile / CUDA graphs)."""

    def __init__(self, model, max_len, device, batch=1):
        from transformers import StaticCache

        self.model, self.device = model, device
        self.cache = StaticCache(config=model.config, max_cache_len=max_len)
        self.tok = torch.zeros((batch, 1), dtype=torch.long, device=device)
        self.pos = torch.zeros((1,), dtype=torch.long, device=device)
        self.graph = None

    @torch.no_grad()
    def prefill(self, ids):
        n = ids.shape[1]
        out = self.model(ids, past_key_values=self.cache, cache_position=torch.arange(n, device=self.device), use_cache=True)
        self.tok.copy_(out.logits[:, -1:].argmax(-1))
        self.pos.fill_(n)

    @torch.no_grad()
    def step(self):
        out = self.model(self.tok, past_key_values=self.cache, cache_position=self.pos, use_cache=True)
        self.tok.copy_(out.logits[:, -1:].argmax(-1))
        self.pos.add_(1)

    def capture(self):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self.step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.step()

    def run(self, n):
        out = []
        for _ in range(n):
            if self.graph is not None:
                self.graph.replay()
            else:
                self.step()
            out.append(self.tok.clone())
        return out


def measure(model, args, device, use_graph, prompt):
    dec = Decoder(model, args.prompt + 3 * args.tokens + 16, device, batch=prompt.shape[0])
    dec.prefill(prompt)
    if use_graph:
        dec.capture()
    dec.run(min(8, args.tokens))  # warm-up
    sync = torch.cuda.synchronize if device.type == "cuda" else (lambda: None)
    sync()
    t0 = time.perf_counter()
    toks = dec.run(args.tokens)
    sync()
    dt = time.perf_counter() - t0
    return {"tokens_per_s": args.tokens * prompt.shape[0] / dt, "ms_per_token": dt / args.tokens * 1e3,   # ms per decode step (all sequences)
            "first_tokens": [int(t[0]) for t in toks[:8]]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama3-8b", choices=sorted(MODELS))
    ap.add_argument("--scheme", default="1x16g8", choices=sorted(SCHEMES))
    ap.add_argument("--tokens", type=int, default=64)
    ap.add_argument("--prompt", type=int, default=16)
    ap.add_argument("--batch", type=int, default=1, help="sequences decoded together (rows of every linear's input); tokens_per_s counts all of them")
    ap.add_argument("--tune", default="", help="library tuning keys for A/B runs, e.g. kx8_xres=0,kx8_multi_xres_min_rows=0")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--dense-only", action="store_true", help="skip the AQLM model (CPU smoke test of the loop)")
    ap.add_argument("--no-dense", action="store_true", help="skip the dense fp16 baseline")
    args = ap.parse_args()
    device = torch.device(args.device)
    dtype = torch.float16 if device.type == "cuda" else torch.float32
    if device.type == "cuda":
        torch.cuda.set_device(device)
    res = {"model": args.model, "scheme": args.scheme, "new_tokens": args.tokens, "prompt_tokens": args.prompt,
           "dtype": str(dtype), "batch": args.batch,
           "data": f"random weights of the real shapes, greedy decode, static KV cache, batch {args.batch}"}
    model = build_dense(args.model, device, dtype, args.prompt + 3 * args.tokens + 16)
    prompt = torch.randint(0, model.config.vocab_size, (args.batch, args.prompt), generator=torch.Generator().manual_seed(3)).to(device)
    graphs = [False, True] if device.type == "cuda" else [False]
    if not args.no_dense:
        for g in graphs:
            res[f"dense_fp16_{'hipgraph' if g else 'eager'}"] = measure(model, args, device, g, prompt)
    if not args.dense_only:
        import aqlm

        if args.tune:
            from aqlm_amd import _native

            for kv in args.tune.split(","):
                k, v = kv.split("=")
                _native.set_tuning(k.strip(), int(v))
            res["tune"] = args.tune
        res["quantized_linears"] = quantize_in_place(model, args.scheme, device, dtype)
        torch.cuda.empty_cache()
        for fused in (False, True):
            groups = aqlm.fuse_shared_input_linears(model) if fused else []
            for g in graphs:
                key = f"aqlm_{'shared_input_' if fused else ''}{'hipgraph' if g else 'eager'}"
                res[key] = measure(model, args, device, g, prompt)
            if fused:
                res["shared_input_groups"] = len(groups)
                res["shared_input_launches"] = sum(gr.launches for gr in groups)
                aqlm.unfuse_shared_input_linears(model)
        a, b = res["aqlm_hipgraph"]["first_tokens"], res["aqlm_shared_input_hipgraph"]["first_tokens"]
        res["shared_input_tokens_identical"] = a == b
    print(json.dumps(res))


if __name__ == "__main__":
    main()
