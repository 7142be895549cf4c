"""The per-block grouped weight-gradient launch at MMDiT-L widths (8 problems: image + text rows), rows scaled down to see whether the
per-K-tile cost depends on the length of the reduction (TLB reach / L2 footprint of the k-major walks)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sd3_amd  # noqa
from sd3_amd import ops

def bench(fn, reps=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(s, generator=g, device="cuda").to(torch.bfloat16)
for d, label in ((1024, "L"), (768, "B")):
    h = 4 * d
    shapes = [(3 * d, d), (d, d), (2 * h, d), (d, h)]
    for scale in (1, 2, 4):
        Mx, Mc = (64 * 1024 if d == 1024 else 16384 * 4) // scale, (9856) // scale // 8 * 8
        probs, fl = [], 0.0
        for N, K in shapes:
            for Mr in (Mx, Mc):
                probs.append(dict(A=rnd(Mr, N), B=rnd(Mr, K), a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, stream_k=True))
                fl += 2.0 * Mr * N * K
        t = bench(lambda: ops.gemm_grouped(probs))
        print(f"{label} widths, rows {Mx}+{Mc}: {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF   per 64-row K-tile of the image problems: {t/(Mx/64)*1e6:6.3f} us")
        del probs
