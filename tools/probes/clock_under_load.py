"""Shader clock and board power while each class of kernel runs (sysfs hwmon of THIS process's GPU, polled from a thread every 20 ms): is the MFMA
roofline's 2.4 GHz the clock the launches actually get?
    python tools/probes/clock_under_load.py [seconds per phase]
Measured (round 5, profiles/r05_clock_under_load.txt): no -- the GEMMs sit at the board's power limit (~1.35-1.39 kW) and clock 1.86-1.94 GHz."""
import glob
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sd3_amd  # noqa: E402,F401
from sd3_amd import ops  # noqa: E402
from tools.gpu_sensors import GpuSensors  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
sens = GpuSensors(torch.device("cuda", 0))
print("sensors of", sens.card, "(selected by:", sens.how + ")")


def phase(name, fn, flop=0.0, peak=2500.0):      # peak: dense MFMA TFLOP/s of the operand type at 2.4 GHz
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
    us = e0.elapsed_time(e1) / max(n, 1) * 1e3
    tf = f"{flop / us / 1e6:6.0f} TF" if flop else "         "
    at = f" = {flop / us / 1e6 / (peak * r['clock_mhz'] / 2400.0):.3f} of the MFMA peak at that clock" if flop and r.get("clock_mhz") else ""
    print(f"{name:44s} {us:9.1f} us/launch {tf} | clock {r.get('clock_mhz', float('nan')):5.0f} MHz [{r.get('clock_min', 0):.0f}, {r.get('clock_max', 0):.0f}]"
          f"  power {r.get('power_w', float('nan')):5.0f} W  busy {r.get('busy', float('nan')):3.0f} %  ({r.get('samples', 0)} samples){at}", flush=True)


g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(s, generator=g, device="cuda").to(torch.bfloat16)
phase("idle (host sleeps)", lambda: time.sleep(0.005))
A, B = rnd(8192, 8192), rnd(8192, 8192)
o = torch.empty((8192, 8192), dtype=torch.bfloat16, device="cuda")
phase("gemm 8192^3 bf16 (NT)", lambda: ops.gemm(A, B, out=o), 2.0 * 8192 ** 3)
Z = torch.zeros_like(A)
phase("  the same launch on ZERO operands", lambda: ops.gemm(Z, Z, out=o), 2.0 * 8192 ** 3)
ma, msa = ops.quant_mxfp8(A); mb, msb = ops.quant_mxfp8(B)
phase("gemm 8192^3 MX e4m3 operands", lambda: ops.gemm(ma, mb, out=o, scale_a=msa, scale_b=msb, scale_mode=1), 2.0 * 8192 ** 3, 5000.0)
del A, B, Z, o, ma, mb
M = 26240
A2, B2 = rnd(M, 768), rnd(6144, 768)
o2 = torch.empty((M, 6144), dtype=torch.bfloat16, device="cuda")
phase("forward Linear 26240 x 6144 x 768", lambda: ops.gemm(A2, B2, out=o2), 2.0 * M * 6144 * 768)
Z2 = torch.zeros_like(A2)
phase("  the same launch, zero activations", lambda: ops.gemm(Z2, B2, out=o2), 2.0 * M * 6144 * 768)
AL, BL = rnd(75392, 1024), rnd(8192, 1024)
mal, msal = ops.quant_mxfp8(AL); mbl, msbl = ops.quant_mxfp8(BL)
oL = torch.empty((75392, 8192), dtype=torch.bfloat16, device="cuda")
phase("MMDiT-L w12 75392 x 8192 x 1024, MX operands", lambda: ops.gemm(mal, mbl, out=oL, scale_a=msal, scale_b=msbl, scale_mode=1), 2.0 * 75392 * 8192 * 1024, 5000.0)
del AL, BL, mal, mbl, oL, Z2
# a block's weight gradients as the step launches them: one grouped stream-K launch (both token streams, 4 Linears each)
Mx, Mc, d = 16384, 9856, 768
probs = []
for N, K in ((3 * d, d), (d, d), (8 * d, d), (d, 4 * d)):
    for Mr in (Mx, Mc):
        probs.append(dict(A=rnd(Mr, N), B=rnd(Mr, K), a_kmajor=True, b_kmajor=True, out=torch.empty((N, K), dtype=torch.float32, device="cuda"), stream_k=True))
phase("weight gradients of a block (grouped launch)", lambda: ops.gemm_grouped(probs), sum(2.0 * p["A"].shape[0] * p["A"].shape[1] * p["B"].shape[1] for p in probs))
del probs
Bn, H, S, hd = 64, 12, 410, 64
Q, K, V = rnd(Bn, H, S, hd), rnd(Bn, H, S, hd), rnd(Bn, H, S, hd)
Ox, Oc, lse = ops.attn_fwd(Q, K, V, 256, 0.125, 0)
phase("attention forward (64 x 12 heads, S = 410)", lambda: ops.attn_fwd(Q, K, V, 256, 0.125, 0), 4.0 * Bn * H * S * S * hd)
dOx, dOc = torch.randn_like(Ox), torch.randn_like(Oc)
phase("attention backward (dkv + dq)", lambda: ops.attn_bwd(Q, K, V, Ox, Oc, dOx, dOc, lse, 256, 0.125, torch.bfloat16), 14.0 * Bn * H * S * S * hd)
x = torch.randn(1 << 28, device="cuda")
y = torch.empty_like(x)
phase("copy 1 GiB fp32 (HBM-bound)", lambda: y.copy_(x))
phase("idle again", lambda: time.sleep(0.005))
