"""Run ONE GEMM shape a few times (for rocprofv3 --pmc passes).  python tools/gemm_one.py fwd|dgrad|wgrad M N K [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sd3_amd  # noqa: E402,F401
from sd3_amd import ops  # noqa: E402

kind, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(s, generator=g, device="cuda").to(torch.bfloat16)
if kind == "fwd":
    A, B, kw = rnd(M, K), rnd(N, K), {}
elif kind == "dgrad":
    A, B, kw = rnd(M, K), rnd(K, N), dict(b_kmajor=True)
else:
    A, B, kw = rnd(K, M), rnd(K, N), dict(a_kmajor=True, b_kmajor=True)
out = torch.empty((M, N), dtype=torch.bfloat16 if kind != "wgrad" else torch.float32, device="cuda")
for _ in range(reps):
    ops.gemm(A, B, out=out, **kw)
torch.cuda.synchronize()
print("done", kind, M, N, K)
