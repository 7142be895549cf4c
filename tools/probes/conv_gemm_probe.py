"""Implicit-GEMM 3x3 convolution vs a plain GEMM of the same (M, N, K): where does the VAE's conv time go?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sd3_amd  # noqa: E402,F401
from sd3_amd import ops  # noqa: E402


def bench(fn, reps=5):
    for _ in range(2):
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
for B, H, W, C, Co in ((16, 512, 512, 128, 128), (16, 256, 256, 256, 256), (16, 128, 128, 512, 512), (16, 64, 64, 512, 512)):
    xin = torch.randn((B, H + 2, W + 2, C), generator=g, device="cuda").to(torch.bfloat16)
    w = torch.randn((Co, 9 * C), generator=g, device="cuda").to(torch.bfloat16)
    bias = torch.randn((Co,), generator=g, device="cuda")
    M, K = B * H * W, 9 * C
    fl = 2.0 * M * Co * K
    for od in (torch.float32, torch.bfloat16):
        t = bench(lambda: ops.gemm(xin, w, bias=bias, out_dtype=od, conv=(1, H, W, C)))
        print(f"conv3x3 implicit  B{B} {H}x{W} C{C}->{Co} out {str(od)[6:]:<9}: {t * 1e3:7.2f} ms {fl / t / 1e12:7.1f} TF")
    A = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16) if M * K * 2 < 20e9 else None
    if A is not None:
        for od in (torch.float32, torch.bfloat16):
            t = bench(lambda: ops.gemm(A, w, bias=bias, out_dtype=od))
            print(f"plain GEMM  M{M} N{Co} K{K}            out {str(od)[6:]:<9}: {t * 1e3:7.2f} ms {fl / t / 1e12:7.1f} TF")
    del A
