"""GEMM micro-benchmark on the MMDiT-B shapes (random bf16 data), timed with HIP events through the C ABI.
Usage (GPU box): python tools/gemm_bench.py [--reps 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sd3_amd  # noqa: E402,F401
from sd3_amd import ops  # noqa: E402


def bench(fn, reps):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)

    def rnd(*s, dt=torch.bfloat16):
        return torch.randn(s, generator=g, device=dev).to(dt)

    Mx, Mc, d, h = 16384, 9856, 768, 3072
    shapes = [("qkv", 3 * d, d), ("out", d, d), ("w12", 2 * h, d), ("w3", d, h)]
    print(f"{'case':<34}{'M':>7}{'N':>7}{'K':>7}{'us':>10}{'TFLOP/s':>10}")
    for name, N, K in shapes:
        Ax, Ac, W = rnd(Mx, K), rnd(Mc, K), rnd(N, K)
        dYx, dYc = rnd(Mx, N), rnd(Mc, N)
        res_x, gate = rnd(Mx, N, dt=torch.float32), rnd(64, N, dt=torch.float32)
        cases = {
            f"{name} fwd bf16out (img)": (lambda: ops.gemm(Ax, W, out_dtype=torch.bfloat16), 2.0 * Mx * N * K),
            f"{name} fwd grouped img+txt": (lambda: ops.gemm_grouped([dict(A=Ax, B=W, out_dtype=torch.bfloat16), dict(A=Ac, B=W, out_dtype=torch.bfloat16)]), 2.0 * (Mx + Mc) * N * K),
            f"{name} dgrad grouped img+txt": (lambda: ops.gemm_grouped([dict(A=dYx, B=W, b_kmajor=True, out_dtype=torch.bfloat16), dict(A=dYc, B=W, b_kmajor=True, out_dtype=torch.bfloat16)]), 2.0 * (Mx + Mc) * N * K),
            f"{name} wgrad (img)": (lambda: ops.gemm(dYx, Ax, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, stream_k=True), 2.0 * Mx * N * K),
            f"{name} wgrad grouped img+txt": (lambda: ops.gemm_grouped([dict(A=dYx, B=Ax, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, stream_k=True), dict(A=dYc, B=Ac, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, stream_k=True)]), 2.0 * (Mx + Mc) * N * K),
        }
        if name in ("out", "w3"):
            aux = torch.empty((Mx, N), dtype=torch.bfloat16, device=dev)
            cases[f"{name} fwd gate+res+aux f32 (img)"] = (lambda: ops.gemm(Ax, W, out_dtype=torch.float32, gate=gate, rows_per_batch=256, residual=res_x, aux=aux), 2.0 * Mx * N * K)
        if name == "w12":
            auxx, auxc = torch.empty((Mx, N), dtype=torch.bfloat16, device=dev), torch.empty((Mc, N), dtype=torch.bfloat16, device=dev)
            bias = rnd(N, dt=torch.float32)
            cases["w12 fwd SwiGLU epilogue grouped"] = (lambda: ops.gemm_grouped([dict(A=Ax, B=W, bias=bias, act=ops.ACT_SWIGLU, aux=auxx), dict(A=Ac, B=W, bias=bias, act=ops.ACT_SWIGLU, aux=auxc)]), 2.0 * (Mx + Mc) * N * K)
            cases["w12 fwd SwiGLU, no aux (sampler)"] = (lambda: ops.gemm_grouped([dict(A=Ax, B=W, bias=bias, act=ops.ACT_SWIGLU), dict(A=Ac, B=W, bias=bias, act=ops.ACT_SWIGLU)]), 2.0 * (Mx + Mc) * N * K)
        for cname, (fn, fl) in cases.items():
            t = bench(fn, args.reps)
            print(f"{cname:<34}{Mx:>7}{N:>7}{K:>7}{t * 1e6:>10.1f}{fl / t / 1e12:>10.1f}")
    # all weight gradients of one block in one launch
    probs, fl = [], 0.0
    for name, N, K in shapes:
        for Mr in (Mx, Mc):
            probs.append(dict(A=rnd(Mr, N), B=rnd(Mr, K), a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, stream_k=True))
            fl += 2.0 * Mr * N * K
    t = bench(lambda: ops.gemm_grouped(probs), args.reps)
    print(f"{'block wgrads, 8 problems grouped':<34}{'':>21}{t * 1e6:>10.1f}{fl / t / 1e12:>10.1f}")
    # the weight gradients of TWO blocks, grouped by stream (equal K inside a launch) instead of by block (image + text mixed)
    for label, Mr in (("image", Mx), ("text", Mc)):
        probs, fl = [], 0.0
        for _ in range(2):
            for name, N, K in shapes:
                probs.append(dict(A=rnd(Mr, N), B=rnd(Mr, K), a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, stream_k=True))
                fl += 2.0 * Mr * N * K
        t = bench(lambda: ops.gemm_grouped(probs), args.reps)
        print(f"{'two blocks, ' + label + ' wgrads (8 problems)':<34}{'':>21}{t * 1e6:>10.1f}{fl / t / 1e12:>10.1f}")
    # square reference shapes
    for n in (4096, 8192):
        A, Bm = rnd(n, n), rnd(n, n)
        t = bench(lambda: ops.gemm(A, Bm, out_dtype=torch.bfloat16), max(3, args.reps // 4))
        print(f"{'square NT bf16':<34}{n:>7}{n:>7}{n:>7}{t * 1e6:>10.1f}{2.0 * n ** 3 / t / 1e12:>10.1f}")
    # the same square problem in the data-gradient (B k-major) and weight-gradient (both k-major, fp32 out) layouts
    n = 8192
    A, Bm = rnd(n, n), rnd(n, n)
    t = bench(lambda: ops.gemm(A, Bm, b_kmajor=True, out_dtype=torch.bfloat16), max(3, args.reps // 4))
    print(f"{'square NN (dgrad layout) bf16':<34}{n:>7}{n:>7}{n:>7}{t * 1e6:>10.1f}{2.0 * n ** 3 / t / 1e12:>10.1f}")
    out32 = torch.empty((n, n), dtype=torch.float32, device=dev)
    t = bench(lambda: ops.gemm(A, Bm, a_kmajor=True, b_kmajor=True, out=out32), max(3, args.reps // 4))
    print(f"{'square TN (wgrad layout) f32':<34}{n:>7}{n:>7}{n:>7}{t * 1e6:>10.1f}{2.0 * n ** 3 / t / 1e12:>10.1f}")


if __name__ == "__main__":
    main()
