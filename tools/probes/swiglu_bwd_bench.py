"""(round 4) The MLP backward of one MMDiT-B block at batch 64 (image 16384 + text 9856 rows, d = 768, h = 3072): data gradient of the
down-projection + SwiGLU backward as two passes (grouped GEMM, mlp_act_bwd_pair) and as ONE launch (MMDIT_ACT_SWIGLU_BWD epilogue).
python tools/probes/swiglu_bwd_bench.py [reps]      (MMDIT_LIB=tools/scratch/<variant>/libmmdit_hip.so for A/B builds)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sd3_amd  # noqa: E402,F401
from sd3_amd import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
d, h = 768, 3072
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(s, generator=g, device="cuda") * sc).to(torch.bfloat16)
P = [(rnd(M, d), rnd(d, h, sc=0.05), rnd(M, 2 * h), torch.zeros(2 * h, device="cuda")) for M in (16384, 9856)]


def two_pass():
    dh = ops.gemm_grouped([dict(A=a, B=w, b_kmajor=True, out_dtype=torch.bfloat16) for a, w, _, _ in P])
    return ops.mlp_act_bwd_pair((dh[0], P[0][2], P[0][3]), (dh[1], P[1][2], P[1][3]), h, False)


def fused():
    return ops.gemm_swiglu_bwd([dict(A=a, B=w, aux=gu, dbias=db) for a, w, gu, db in P])


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


a, ta = timed(two_pass)
b, tb = timed(fused)
fl = sum(2.0 * p[0].shape[0] * d * h for p in P)
nbytes = sum(p[0].shape[0] * (2 * h * 2 * 2 + d * 2) for p in P)
same = all(torch.equal(x, y) for x, y in zip(a, b))
print(f"two passes {ta:7.1f} us   fused {tb:7.1f} us  ({fl / tb / 1e6:6.1f} TF, {nbytes / tb / 1e6:5.2f} TB/s of [g|u] + d[g|u] + dY traffic)   bit-identical: {same}")
if not same:
    for x, y in zip(a, b):
        ne = (x != y)
        print("  mismatches", int(ne.sum()), "of", x.numel(), " max |diff|", float((x.float() - y.float()).abs().max()))
