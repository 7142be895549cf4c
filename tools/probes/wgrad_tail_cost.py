"""What do the atomic partial tiles of the weight-gradient launch cost?  (a) a block's image problems + three of the four text problems:
252 tiles of 256x256 = one round, no tail, no atomics; (b) the full block (288 tiles: 256 + a 32-tile tail cut 3 ways = 96 atomic partial
tiles, balanced onto the text workgroups); (c) the image problems alone (144 tiles, whole K) and (d) the text problems alone."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sd3_amd  # noqa
from sd3_amd import ops

g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(s, generator=g, device="cuda").to(torch.bfloat16)
d, h, Mx, Mc = 768, 3072, 16384, 9856
shapes = [("qkv", 3 * d, d), ("out", d, d), ("w12", 2 * h, d), ("w3", d, h)]


def bench(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def probs(sel):
    out, fl, tiles, units = [], 0.0, 0, 0
    for (name, N, K), Mr in sel:
        out.append(dict(A=rnd(Mr, N), B=rnd(Mr, K), a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, stream_k=True))
        fl += 2.0 * Mr * N * K
        t = ((N + 255) // 256) * ((K + 255) // 256)
        tiles += t
        units += t * (Mr // 64)
    return out, fl, tiles, units


cases = {
    "image x4 + text qkv/out/w12 (252 tiles, no tail)": [(s, Mx) for s in shapes] + [(s, Mc) for s in shapes[:3]],
    "full block (288 tiles, 96 atomic partials)": [(s, M) for s in shapes for M in (Mx, Mc)],
    "image x4 (144 tiles)": [(s, Mx) for s in shapes],
    "text x4 (144 tiles)": [(s, Mc) for s in shapes],
}
for name, sel in cases.items():
    p, fl, tiles, units = probs(sel)
    t = bench(lambda: ops.gemm_grouped(p))
    print(f"{name:<52} {t:8.1f} us  {fl / t / 1e6:7.1f} TF   tiles {tiles}  K-tile units {units}  ({units / 256:.1f} per CU if perfectly spread)")
