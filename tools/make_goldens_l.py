"""MMDiT-L goldens (BASELINE.json configs 4 and 5: 24 blocks, d = 1024, 16 heads, 64x64x16 latents = 512^2 images) from the REAL
reference, imported in the build container through tools/ref_import.py exactly like tools/make_goldens.py (same seeded weights
and inputs of oracle/weights.py).  Writes only data:

  tests/golden/forward_l_plain.npz   one forward, batch 1, Gemma-like text (x30), t = 0.35            (~7 s of CPU)
  tests/golden/sampler_l.npz         the reference's own sample_imgs loop: 28 Euler steps, CFG 3.0, batch 1, 512^2,
                                     stand-in text / VAE objects (identity decode)                         (~7 min of CPU)
  tests/golden/generation_report_l.json   oracle-vs-reference distances measured at generation time

Usage:  python tools/make_goldens_l.py [--no-sampler]
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from oracle import mmdit_oracle as O  # noqa: E402
from oracle.weights import make_inputs  # noqa: E402
from make_goldens import GOLD, build_ref, checksum, rel_l2  # noqa: E402
from ref_import import import_reference  # noqa: E402

L_CFG = dict(dim=1024, num_heads=16, num_blocks=24)


def main():
    torch.set_num_threads(8)
    refmod = import_reference()
    net, sd = build_ref(refmod, L_CFG)
    report = {}

    x, c, cp = make_inputs(50, 1, 64, 64, text_scale=30.0)
    t = torch.tensor([0.35])
    t0 = time.time()
    with torch.no_grad():
        v = net(x.clone(), t, c.clone(), cp.clone())
    print(f"reference L forward: {time.time() - t0:.1f} s", flush=True)
    with torch.no_grad():
        vo = O.forward(sd, O.OracleConfig(**L_CFG), x.clone(), t, c.clone(), cp.clone())
        vf = O.forward(sd, O.OracleConfig(**L_CFG, attn_core="flash_bf16", gemm="bf16"), x.clone(), t, c.clone(), cp.clone())
    report["l_plain"] = {"oracle_vs_ref": rel_l2(vo, v), "fast_rounding_vs_ref": rel_l2(vf, v), "v_std": float(v.std())}
    print(report, flush=True)
    np.savez_compressed(os.path.join(GOLD, "forward_l_plain.npz"), v=v.numpy(), inputs_checksum=np.array(checksum(x, c, cp)))

    if "--no-sampler" not in sys.argv:
        class _Cfg:
            latent_channels, shift_factor, scaling_factor = 16, 0.0, 8.0     # (identity decode: keeps the latents inside the final clamp(-1, 1))

        class _Dec:
            def __init__(self, s):
                self.sample = s

        class _VAE:
            config, dtype = _Cfg(), torch.float32

            def decode(self, z):
                return _Dec(z)

        class _Enc:
            VAE = _VAE()

            def __init__(self, th, tp):
                self.th, self.tp = th, tp

            def text_to_embedding(self, text):
                return self.th.clone(), self.tp.clone()

        _, th, tp = make_inputs(51, 1, 64, 64, text_scale=30.0)
        net.text_encoders = _Enc(th, tp)
        gen = torch.Generator().manual_seed(123)
        t0 = time.time()
        img = net.sample_imgs(1, 28, ["x"], cfg_scale=3.0, width=512, height=512, sampler="euler", generator=gen)
        print(f"reference L sampler (28 steps, CFG): {time.time() - t0:.1f} s", flush=True)
        noise = torch.randn((1, 16, 64, 64), generator=torch.Generator().manual_seed(123))
        np.savez_compressed(os.path.join(GOLD, "sampler_l.npz"), out=img.numpy(), noise_checksum=np.array(checksum(noise)))
        report["sampler_l"] = {"steps": 28, "cfg_scale": 3.0, "out_std": float(img.std()), "clamped_frac": float((img.abs() >= 1).float().mean())}

    with open(os.path.join(GOLD, "generation_report_l.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
