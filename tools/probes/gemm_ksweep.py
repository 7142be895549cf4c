"""Where does a K = 768 GEMM spend its time?  Time the lean / wide kernel at fixed (M, N) over a sweep of K: t(K) = fixed + slope * K.
slope -> cycles per 64-deep K step of a tile (the operand stream), intercept -> launch + pipeline fill + the unhidden epilogues.
python tools/probes/gemm_ksweep.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sd3_amd  # noqa: E402,F401
from sd3_amd import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(s, generator=g, device="cuda").to(torch.bfloat16)
M = 26240


def timed(fn):
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


for kind in ("fwd", "dgrad"):
    for N in (768, 2304, 6144):
        pts = []
        for K in (128, 256, 512, 768, 1536, 3072, 6144):
            A = rnd(M, K)
            B = rnd(N, K) if kind == "fwd" else rnd(K, N)
            kw = {} if kind == "fwd" else dict(b_kmajor=True)
            out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
            t = timed(lambda: ops.gemm(A, B, out=out, **kw))
            pts.append((K, t))
        # least squares on the K >= 512 points
        xs = [k for k, _ in pts if k >= 512]
        ys = [t for k, t in pts if k >= 512]
        n = len(xs)
        mx, my = sum(xs) / n, sum(ys) / n
        slope = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs)
        icpt = my - slope * mx
        line = "  ".join(f"K={k}: {t:7.1f}us {2.0 * M * N * k / t / 1e6:6.0f}TF" for k, t in pts)
        print(f"{kind} M={M} N={N}: {line}")
        print(f"    fit: {icpt:6.1f} us fixed + {slope * 64:6.3f} us per 64-deep K step  (asymptote {2.0 * M * N * 64 / (slope * 64) / 1e6:6.0f} TF)")
