"""One-off timing of a shared-input group of 1x16g16 layers: pipelined kernel vs workgroup-per-segment vs separate launches."""
import sys, torch
sys.path.insert(0, ".")
from aqlm_amd import _native
from aqlm_amd.inference_kernels import hip_kernel as hk

def layer(fi, fo, g, seed):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    codes = torch.randint(-32768, 32768, (fo, fi // g, 1), generator=gen, device="cuda", dtype=torch.int32).to(torch.int16)
    cb = torch.randn((1, 65536, 1, g), generator=gen, device="cuda").half()
    sc = torch.ones((fo, 1, 1, 1), device="cuda", dtype=torch.float16)
    return codes, cb, sc

def time_graph(fn, n=20):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record();
    for _ in range(n): g.replay()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

for g in (8, 16):
    for name, fin, outs in (("q/k/v", 4096, (4096, 1024, 1024)), ("gate/up", 4096, (11008, 11008))):
        NL = 24  # rotate over distinct layer sets (cold)
        sets = []
        for i in range(NL):
            ls = [layer(fin, o, g, 100 * i + k) for k, o in enumerate(outs)]
            pk = [hk.prepack_1x16(c, g, codebooks=cb) for c, cb, sc in ls]
            sets.append((ls, pk))
        x = torch.randn((1, fin), device="cuda").half()
        def multi():
            for ls, pk in sets:
                hk.code1x16_matmat_packed_multi(x, pk, [l[1] for l in ls], [l[2] for l in ls], [None] * len(ls))
        def sep():
            for ls, pk in sets:
                for (c, cb, sc), p in zip(ls, pk):
                    hk.code1x16_matmat_packed(x, p, cb, sc, None)
        t_sep = time_graph(sep) / NL
        _native.set_tuning("packed_pipe", 0); t_plain = time_graph(multi) / NL
        _native.set_tuning("packed_pipe", 1); t_pipe = time_graph(multi) / NL
        print(f"g{g} {name}: separate {t_sep:.2f} us, one launch (workgroup per segment) {t_plain:.2f}, pipelined {t_pipe:.2f}")
        del sets
