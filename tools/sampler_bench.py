"""BASELINE.json config 5 on one GPU: 28-step rectified-flow sampler (Euler, CFG: a batch of 2B forwards per step) at MMDiT-L / 512^2
(or --B: MMDiT-B / 256^2) through diff_model.sample_imgs, bf16, per-tensor fp8 and MX fp8 operands; identity stand-ins for the text encoders / VAE
decode (out of this build's scope).  Prints images/s = B / wall.  python tools/sampler_bench.py [--B] [--batch 32] [--steps 28]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sd3_amd  # noqa: E402,F401
from sd3_amd.models.diff_model import diff_model  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", action="store_true")
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--steps", type=int, default=28)
ap.add_argument("--modes", default="fast,fp8,mxfp8", help="precision modes to time (comma separated)")
args = ap.parse_args()
cfg = dict(dim=768, num_heads=12, num_blocks=12) if args.B else dict(dim=1024, num_heads=16, num_blocks=24)
res = 256 if args.B else 512
dev = torch.device("cuda:0")
net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=dev, positional_encoding="RoPE2d", **cfg)


class _Dec:
    def __init__(self, s):
        self.sample = s


class _Cfg:
    latent_channels, shift_factor, scaling_factor = 16, 0.0, 8.0


class _VAE:
    config, dtype = _Cfg(), torch.float32

    def decode(self, z):
        return _Dec(z)


class _Enc:
    VAE = _VAE()

    def text_to_embedding(self, text):
        g = torch.Generator().manual_seed(1)
        return torch.randn(1, 154, 2304, generator=g) * 3, torch.randn(1, 768, generator=g)


def fwd_flops(d, blocks, N, M=154):
    """algorithmic FLOPs of one forward per image (GEMM + attention matmuls, multiply-add = 2)."""
    S, h = N + M, 4 * d
    f = 2 * d * d + 2 * 768 * d + 2 * M * 2304 * d + 2 * N * 64 * d + 2 * N * d * d + 4 * d * d + 2 * N * d * 64
    for i in range(blocks):
        last = i == blocks - 1
        f += 2 * d * d + 4 * d * d * (3 if last else 4) + 2 * d * d * (2 if last else 4) + 8 * N * d * d + 2 * M * d * d * (3 if last else 4) + 4 * S * S * d
        f += 6 * d * h * N + (0 if last else 6 * d * h * M)
    return f


import json  # noqa: E402
F_FWD = fwd_flops(cfg["dim"], cfg["num_blocks"], (res // 16) ** 2)
net.text_encoders = _Enc()
for prec in args.modes.split(","):
    net.set_precision(prec)
    net.sample_imgs(args.batch, 2, ["x"], cfg_scale=3.0, width=res, height=res, generator=torch.Generator().manual_seed(0))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = net.sample_imgs(args.batch, args.steps, ["x"], cfg_scale=3.0, width=res, height=res, generator=torch.Generator().manual_seed(0))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{'MMDiT-B 256^2' if args.B else 'MMDiT-L 512^2'} sampler, {args.steps} Euler steps, CFG, batch {args.batch} [{'bf16' if prec == 'fast' else prec}]: "
          f"{dt * 1e3:.1f} ms, {args.batch / dt:.2f} images/s, finite={bool(torch.isfinite(out).all())}")
    peak = 2.5e15 if prec == "fast" else 5.0e15        # dense bf16 / fp8 MFMA peak (MI355X_MICROARCH.md)
    ach = args.batch / dt * args.steps * 2 * F_FWD     # CFG: two forwards per step and image
    print(json.dumps({"metric": "sampler images/sec", "value": round(args.batch / dt, 2), "config": {"workload": f"{'MMDiT-B 256^2' if args.B else 'MMDiT-L 512^2'} {args.steps}-step Euler CFG sampler", "batch": args.batch},
                      "dtype": "bf16" if prec == "fast" else prec, "roofline": {"bound": "mfma", "achieved": round(ach / 1e12, 1), "peak": peak / 1e12, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                                                                               "tflop_per_image": round(args.steps * 2 * F_FWD / 1e12, 2)}}))
