"""Do power-of-two leading dimensions slow the weight-gradient GEMM (k-major operands walked down the rows)?  MMDiT-L shapes,
operands as views of wider buffers (ld = width + pad)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sd3_amd  # noqa
from sd3_amd import ops

def bench(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

g = torch.Generator(device="cuda").manual_seed(0)
M = 65536
for name, N, K in (("qkv", 3072, 1024), ("out", 1024, 1024), ("w12", 8192, 1024), ("w3", 1024, 4096), ("B w12", 6144, 768)):
    rows = M if name != "B w12" else 16384
    for pad in (0, 64, 192):
        dYb = torch.randn((rows, N + pad), generator=g, device="cuda").to(torch.bfloat16)
        Xb = torch.randn((rows, K + pad), generator=g, device="cuda").to(torch.bfloat16)
        dY, X = dYb[:, :N], Xb[:, :K]
        t = bench(lambda: ops.gemm(dY, X, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, stream_k=True))
        fwd_w = torch.randn((N, K), generator=g, device="cuda").to(torch.bfloat16)
        tf = bench(lambda: ops.gemm(X, fwd_w, out_dtype=torch.bfloat16))
        print(f"{name:<6} rows {rows} N {N} K {K} pad {pad:3d}: wgrad {t*1e6:8.1f} us {2.0*rows*N*K/t/1e12:7.1f} TF | fwd {tf*1e6:8.1f} us {2.0*rows*N*K/tf/1e12:7.1f} TF")
