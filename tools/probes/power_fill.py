"""Premise check for running an HBM-bound kernel BESIDE a narrowed GEMM (two streams): sequential (GEMM on 256 CUs, then the row kernel) against concurrent
(GEMM grid limited to n CUs by mmdit_set_cu_budget on stream 1, the row kernel on stream 2).  The GEMMs are power-limited and the row kernels are not, so the
concurrent form should finish the same work sooner if the hardware really runs both.
    python tools/probes/power_fill.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sd3_amd  # noqa: E402,F401
from sd3_amd import _lib, ops  # noqa: E402
from tools.gpu_sensors import GpuSensors  # noqa: E402

L = _lib.lib()
sens = GpuSensors(torch.device("cuda", 0))
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(s, generator=g, device="cuda").to(torch.bfloat16)
M = 26240
A, B = rnd(M, 768), rnd(6144, 768)
o = torch.empty((M, 6144), dtype=torch.bfloat16, device="cuda")
x = torch.randn(1 << 27, device="cuda")          # 512 MiB fp32: a copy moves 1 GiB
y = torch.empty_like(x)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
NG, NR = 40, 20                                   # ~9 ms of GEMMs, ~4.5 ms of copies when run alone


def run(mode, cus):
    L.mmdit_set_cu_budget(cus)
    torch.cuda.synchronize()
    sens.start()
    t0 = time.perf_counter()
    reps = 60
    for _ in range(reps):
        if mode == "sequential":
            with torch.cuda.stream(s1):
                for _ in range(NG):
                    ops.gemm(A, B, out=o)
                for _ in range(NR):
                    y.copy_(x)
        else:
            with torch.cuda.stream(s1):
                for _ in range(NG):
                    ops.gemm(A, B, out=o)
            with torch.cuda.stream(s2):
                for _ in range(NR):
                    y.copy_(x)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps * 1e3
    r = sens.stop(skip=0.2)
    print(f"{mode:11s} GEMM grid <= {cus:3d} CUs: {dt:7.2f} ms per ({NG} GEMMs + {NR} x 1 GiB copies)   clock {r.get('clock_mhz', 0):5.0f} MHz  power {r.get('power_w', 0):5.0f} W", flush=True)


try:
    run("sequential", 256)
    for cus in (256, 224, 192, 160, 128, 96):
        run("concurrent", cus)
    run("sequential", 256)
finally:
    L.mmdit_set_cu_budget(256)
