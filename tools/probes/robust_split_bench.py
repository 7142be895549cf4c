"""(round 6) The one-round data gradients of an MMDiT-B block (image + text grouped, N = 768) under the whole-chip plan and under the data-parallel backward's plan
(mmdit_set_cu_budget(224): 256 x 256 tiles claimed dynamically, split tail through the workspace), alone and beside an occupant kernel that holds C compute units.
python tools/probes/robust_split_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sd3_amd  # noqa: E402,F401
from sd3_amd import _lib, ops  # noqa: E402

L = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: (torch.randn(s, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
side = torch.cuda.Stream()


def timed(fn, C, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if C:
        L.mmdit_debug_occupy(C, int(60e-3 * 2.0e9), side.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for K in (768, 2304, 6144):
    probs = [dict(A=rnd(M, K), B=rnd(K, 768), b_kmajor=True, out_dtype=torch.bfloat16) for M in (16384, 9856)]
    fn = lambda: ops.gemm_grouped(probs)
    row = f"dgrad 26240 x 768 x {K:4d}:"
    for budget in (256, 224):
        L.mmdit_set_cu_budget(budget)
        arr = (_lib.GemmArgs * 2)()
        for i, p in enumerate(probs):
            ops._fill_gemm(arr[i], **p)
        plan = L.mmdit_gemm_plan(arr, 2)
        row += f"   plan({budget}) = {plan:3d}:"
        for C in (0, 8, 32):
            row += f" C={C}: {timed(fn, C):6.1f} us"
            torch.cuda.synchronize()
    L.mmdit_set_cu_budget(256)
    print(row, flush=True)
