"""Same-box A/B of the 8-phase GEMM's epilogue-under-the-next-tile ("drain") schedule: the K = 768 launches of MMDiT-B on 256 x 256 tiles.
One configuration per process (the probes library caches its environment switches):
    MMDIT_LIB=tools/scratch/probes/libmmdit_hip.so MMDIT_GEMM_CFG=2 [MMDIT_GEMM_DEBUG=128] python tools/probes/drain_ab.py [reps]
MMDIT_GEMM_DEBUG=128 = the staged deferred epilogue everywhere (round 4's kernel)."""
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


tag = f"cfg={os.environ.get('MMDIT_GEMM_CFG', '-')} debug={os.environ.get('MMDIT_GEMM_DEBUG', '0')}"
for kind, N, K in (("fwd", 6144, 768), ("fwd", 2304, 768), ("fwd", 3072, 768), ("dgrad", 3072, 768), ("dgrad", 6144, 768), ("fwd", 768, 3072), ("fwd", 6144, 256), ("fwd", 6144, 3072)):
    A = rnd(M, K)
    B = rnd(N, K) if kind == "fwd" else rnd(K, N)
    kw = {} if kind == "fwd" else dict(b_kmajor=True)
    out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    t = timed(lambda: ops.gemm(A, B, out=out, **kw))
    ref = (A.float() @ (B.float().t() if kind == "fwd" else B.float()))
    err = float((out.float() - ref).abs().max() / ref.abs().max())
    print(f"[{tag}] plain {kind} N={N} K={K}: {t:7.1f} us {2.0 * M * N * K / t / 1e6:6.0f} TF  max err {err:.1e}")
# the SwiGLU-fused up-projection (with the pre-activation output) and the fused SwiGLU backward launch at the MMDiT-B shapes
d, h = 768, 3072
X, W12, b12 = rnd(M, d), rnd(2 * h, d), torch.randn(2 * h, generator=g, device="cuda")
aux = torch.empty((M, 2 * h), dtype=torch.bfloat16, device="cuda")
hout = torch.empty((M, h), dtype=torch.bfloat16, device="cuda")
t = timed(lambda: ops.gemm(X, W12, bias=b12, act=ops.ACT_SWIGLU, aux=aux, out=hout))
print(f"[{tag}] SwiGLU up-projection (aux): {t:7.1f} us {2.0 * M * 2 * h * d / t / 1e6:6.0f} TF")
dY, W3 = rnd(M, d) * 0.01, rnd(d, h)
dgu = torch.empty((M, 2 * h), dtype=torch.bfloat16, device="cuda")
dbias = torch.zeros(2 * h, device="cuda")
r = ops.gemm_swiglu_bwd([dict(A=dY, B=W3, aux=aux, dbias=dbias, out=dgu)])
if r is not None:
    t = timed(lambda: ops.gemm_swiglu_bwd([dict(A=dY, B=W3, aux=aux, dbias=dbias, out=dgu)]))
    print(f"[{tag}] SwiGLU backward launch: {t:7.1f} us {2.0 * M * h * d / t / 1e6:6.0f} TF")
