"""fp8 (e4m3: per-tensor scales / MX block scales) vs bf16 operand GEMM on the MMDiT forward shapes (NT, bf16 out), and the quantisers."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sd3_amd
from sd3_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(s, generator=g, device="cuda").to(torch.bfloat16)
def bench(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
for name, M, N, K in (("qkv", 26240, 2304, 768), ("out", 26240, 768, 768), ("w12", 26240, 6144, 768), ("w3", 26240, 768, 3072), ("L w12", 75392, 8192, 1024), ("sq8k", 8192, 8192, 8192)):
    A, W = rnd(M, K), rnd(N, K)
    qa, sa = ops.quant_fp8(A); qw, sw = ops.quant_fp8(W)
    tb = bench(lambda: ops.gemm(A, W, out_dtype=torch.bfloat16))
    tf = bench(lambda: ops.gemm(qa, qw, out_dtype=torch.bfloat16, scale_a=sa, scale_b=sw))
    tq = bench(lambda: ops.quant_fp8(A))
    ma, msa = ops.quant_mxfp8(A); mw, msw = ops.quant_mxfp8(W)
    tm = bench(lambda: ops.gemm(ma, mw, out_dtype=torch.bfloat16, scale_a=msa, scale_b=msw, scale_mode=1))
    tmq = bench(lambda: ops.quant_mxfp8(A))
    site = ops.Fp8Site(); site.quantise(A)
    tdq = bench(lambda: site.quantise(A))
    fl = 2.0 * M * N * K
    print(f"{name:<6} {M}x{N}x{K}: bf16 {tb*1e6:8.1f} us {fl/tb/1e12:7.1f} TF | fp8 {tf*1e6:8.1f} us {fl/tf/1e12:7.1f} TF ({tb/tf:.2f}x) | mxfp8 {tm*1e6:8.1f} us {fl/tm/1e12:7.1f} TF ({tb/tm:.2f}x)"
          f" | quantise A: two-pass {tq*1e6:5.1f}, delayed {tdq*1e6:5.1f}, MX {tmq*1e6:5.1f} us")
