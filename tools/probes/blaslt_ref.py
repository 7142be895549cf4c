"""Library reference for the GEMM shapes of the MMDiT-B step: torch.matmul (hipBLASLt / rocBLAS) on the same shapes tools/gemm_bench.py
times through the C ABI.  Not a product path: a yardstick for the hand-written kernels."""
import torch


def bench(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(s, generator=g, device="cuda").to(torch.bfloat16)
M, d, h = 26240, 768, 3072
print(f"{'case':<30}{'M':>7}{'N':>7}{'K':>7}{'us':>10}{'TFLOP/s':>10}")
for name, N, K in (("qkv", 3 * d, d), ("out", d, d), ("w12", 2 * h, d), ("w3", d, h)):
    A, W, dY = rnd(M, K), rnd(N, K), rnd(M, N)
    out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    dA = torch.empty((M, K), dtype=torch.bfloat16, device="cuda")
    dW = torch.empty((N, K), dtype=torch.bfloat16, device="cuda")
    dW32 = torch.empty((N, K), dtype=torch.float32, device="cuda")
    for cname, fn in ((f"{name} fwd  A W^T", lambda: torch.matmul(A, W.t(), out=out)), (f"{name} dgrad dY W", lambda: torch.matmul(dY, W, out=dA)),
                      (f"{name} wgrad dY^T A (bf16 out)", lambda: torch.matmul(dY.t(), A, out=dW))):
        t = bench(fn)
        print(f"{cname:<30}{M:>7}{N:>7}{K:>7}{t * 1e6:>10.1f}{2.0 * M * N * K / t / 1e12:>10.1f}")
for n in (4096, 8192):
    A, B = rnd(n, n), rnd(n, n)
    t = bench(lambda: torch.matmul(A, B.t()), 5)
    print(f"{'square NT':<30}{n:>7}{n:>7}{n:>7}{t * 1e6:>10.1f}{2.0 * n ** 3 / t / 1e12:>10.1f}")
