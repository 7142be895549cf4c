"""Does the leading dimension of the OUTPUT matter for the epilogue's store burst?  Same GEMM (M = 26240, K = 768) into a contiguous C and
into a view with padded rows (row stride N + pad elements).  python tools/probes/ldc_pad.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sd3_amd  # noqa: E402,F401
from sd3_amd import ops  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(s, generator=g, device="cuda").to(torch.bfloat16)
M, K = 26240, 768


def timed(fn, reps=30):
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


for N in (768, 2304, 6144):
    A, B = rnd(M, K), rnd(N, K)
    row = []
    for pad in (0, 8, 64, 72, 192):
        full = torch.empty((M, N + pad), dtype=torch.bfloat16, device="cuda")
        out = full[:, :N]
        t = timed(lambda: ops.gemm(A, B, out=out))
        row.append(f"pad {pad:3d}: {t:6.1f} us {2.0 * M * N * K / t / 1e6:5.0f} TF")
    print(f"N = {N}: " + "   ".join(row))
