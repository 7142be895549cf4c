"""GEMM throughput against the number of compute units its persistent grid may use (mmdit_set_cu_budget), with the clock / power the launch gets:
under the board's power limit a GEMM on fewer CUs clocks higher -- how much of the lost width comes back?
    python tools/probes/cu_budget_scaling.py [seconds per point]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sd3_amd  # noqa: E402,F401
from sd3_amd import _lib, ops  # noqa: E402
from tools.gpu_sensors import GpuSensors  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
sens = GpuSensors(torch.device("cuda", 0))
L = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(s, generator=g, device="cuda").to(torch.bfloat16)


def point(fn):
    torch.cuda.synchronize()
    sens.start()
    t0 = time.time(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    r = sens.stop(skip=0.25)
    return e0.elapsed_time(e1) / n * 1e3, r.get("clock_mhz", float("nan")), r.get("power_w", float("nan"))


M = 26240
A, B = rnd(M, 768), rnd(6144, 768)
o = torch.empty((M, 6144), dtype=torch.bfloat16, device="cuda")
A8, B8 = rnd(8192, 8192), rnd(8192, 8192)
o8 = torch.empty((8192, 8192), dtype=torch.bfloat16, device="cuda")
Mx, Mc, d = 16384, 9856, 768
probs = []
for N, K in ((3 * d, d), (d, d), (8 * d, d), (d, 4 * d)):
    for Mr in (Mx, Mc):
        probs.append(dict(A=rnd(Mr, N), B=rnd(Mr, K), a_kmajor=True, b_kmajor=True, out=torch.empty((N, K), dtype=torch.float32, device="cuda"), stream_k=True))
cases = (("forward Linear 26240 x 6144 x 768", lambda: ops.gemm(A, B, out=o)), ("gemm 8192^3", lambda: ops.gemm(A8, B8, out=o8)),
         ("weight gradients of a block (grouped)", lambda: ops.gemm_grouped(probs)))
try:
    for name, fn in cases:
        base = None
        for cus in (256, 224, 192, 160, 128, 96, 64):
            assert L.mmdit_set_cu_budget(cus) == 0
            us, mhz, w = point(fn)
            base = base or us
            print(f"{name:40s} {cus:4d} CUs  {us:8.1f} us  x{us / base:5.2f} (width alone: x{256 / cus:4.2f})   clock {mhz:5.0f} MHz  power {w:5.0f} W", flush=True)
finally:
    L.mmdit_set_cu_budget(256)
