"""(round 4) The QKV projection of one MMDiT-B block at batch 64 (image 16384 + text 9856 rows, d = 768, 12 heads) with the QK-norm + RoPE +
joint-layout store in its epilogue (mmdit_gemm_qkv_norm_rope), timed; against GEMM + mmdit_qk_norm_rope_fwd_pair.  With a probes build,
MMDIT_GEMM_CFG=2 forces 256 x 256 tiles (the 8-phase kernel's QK epilogue) instead of the planner's 320 x 256 (the wide kernel).
python tools/probes/qkv_fused_bench.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sd3_amd  # noqa: E402,F401
from sd3_amd import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B, N, Mt, H, d = 64, 256, 154, 12, 768
S = N + Mt
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(s, generator=g, device="cuda") * sc).to(torch.bfloat16)
ax, ac = rnd(B * N, d), rnd(B * Mt, d)
wx, wc = rnd(3 * d, d, sc=0.04), rnd(3 * d, d, sc=0.04)
nw = [1 + 0.1 * torch.randn(64, generator=g, device="cuda") for _ in range(4)]
ang = torch.rand(N, 64, generator=g, device="cuda") * 6.28
cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
Q, K, V = (torch.empty((B, H, S, 64), dtype=torch.bfloat16, device="cuda") for _ in range(3))


def fused():
    return ops.gemm_qkv_norm_rope([dict(A=ax, B=wx, out_dtype=torch.bfloat16), dict(A=ac, B=wc, out_dtype=torch.bfloat16)],
                                  [(nw[0], nw[1], cos, sin, N, 0), (nw[2], nw[3], None, None, Mt, N)], H, S, Q, K, V)


def two_pass():
    qx, qc = ops.gemm_grouped([dict(A=ax, B=wx, out_dtype=torch.bfloat16), dict(A=ac, B=wc, out_dtype=torch.bfloat16)])
    ops.qk_norm_rope_fwd_pair((qx, nw[0], nw[1], cos, sin, N, 0), (qc, nw[2], nw[3], None, None, Mt, N), B, H, S, Q, K, V)
    return qx, qc


def timed(fn):
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return out, e0.elapsed_time(e1) / reps * 1e3


r = fused()
if r is None:
    print("the planner does not give these problems to a kernel with the QKV epilogue")
    sys.exit(0)
_, tf = timed(fused)
qf, kf = Q.clone(), K.clone()
_, tt = timed(two_pass)
fl = 2.0 * (B * N + B * Mt) * 3 * d * d
print(f"fused {tf:7.1f} us ({fl / tf / 1e6:6.1f} TF)   two passes {tt:7.1f} us   Q / K identical to the two-pass form: {torch.equal(qf, Q)} / {torch.equal(kf, K)}   "
      f"(MMDIT_GEMM_CFG={os.environ.get('MMDIT_GEMM_CFG', '-')})")
