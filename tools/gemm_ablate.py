"""Time ONE GEMM shape with HIP events (ablation runs: set MMDIT_GEMM_DEBUG / MMDIT_GEMM_CFG per process).
python tools/gemm_ablate.py fwd|dgrad|wgrad M N K [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sd3_amd  # noqa: E402,F401
from sd3_amd import ops  # noqa: E402

kind, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 30
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(s, generator=g, device="cuda").to(torch.bfloat16)
if kind == "fwd":
    A, B, kw = rnd(M, K), rnd(N, K), {}
elif kind == "dgrad":
    A, B, kw = rnd(M, K), rnd(K, N), dict(b_kmajor=True)
else:
    A, B, kw = rnd(K, M), rnd(K, N), dict(a_kmajor=True, b_kmajor=True, stream_k=True)
out = torch.zeros((M, N), dtype=torch.bfloat16 if kind != "wgrad" else torch.float32, device="cuda")
for _ in range(5):
    ops.gemm(A, B, out=out, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.gemm(A, B, out=out, **kw)
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / reps * 1e-3
print(f"{kind} {M}x{N}x{K} dbg={os.environ.get('MMDIT_GEMM_DEBUG', '0')} cfg={os.environ.get('MMDIT_GEMM_CFG', '-')}: {t * 1e6:.1f} us  {2.0 * M * N * K / t / 1e12:.1f} TF")
