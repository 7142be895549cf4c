"""Goldens for the two samplers of the reference that had none: "heun" and "euler_stochastic" (src/models/diff_model.py:434-462),
from the REAL reference's own sample_imgs loop (imported in the build container through tools/ref_import.py, seeded weights and
inputs of oracle/weights.py, stand-in text / VAE objects with an identity decode, shift 0 / scale 8 -- as section 7 of tools/make_goldens.py does for
"euler").  Writes only data:

  tests/golden/sampler_micro_variants.npz   out_heun, out_euler_stochastic (micro config, batch 2, 4 steps, CFG 3.0, 128^2),
                                            the initial noise and the per-step noises the stochastic sampler drew (CPU generator 99)
  tests/golden/generation_report_samplers.json   oracle-vs-reference distances at generation time

Usage:  python tools/make_goldens_samplers.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from oracle import mmdit_oracle as O  # noqa: E402
from oracle.weights import make_inputs  # noqa: E402
from make_goldens import CONFIGS, GOLD, build_ref, rel_l2  # noqa: E402
from ref_import import import_reference  # noqa: E402


class _Cfg:
    latent_channels, shift_factor, scaling_factor = 16, 0.0, 8.0     # (keeps the latents inside the final clamp(-1, 1): every element is informative)


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


def main():
    torch.set_num_threads(8)
    refmod = import_reference()
    cfg = CONFIGS["micro"][0]
    net, sd = build_ref(refmod, cfg)
    _, th, tp = make_inputs(40, 1, 16, 16, text_scale=30.0)
    net.text_encoders = _Enc(th, tp)
    out, report = {}, {}
    for sampler in ("heun", "euler_stochastic"):
        img = net.sample_imgs(2, 4, ["x"], cfg_scale=3.0, width=128, height=128, sampler=sampler, generator=torch.Generator().manual_seed(99))
        gen = torch.Generator().manual_seed(99)
        noise = torch.randn((2, 16, 16, 16), generator=gen)
        lat = O.cfg_sample(sd, O.OracleConfig(**cfg), noise, th, tp, 4, 3.0, sampler, gen)
        dec = (lat / 8.0).clamp(-1, 1)
        report[sampler] = {"oracle_vs_ref": rel_l2(dec, img), "out_std": float(img.std()), "clamped_frac": float((img.abs() >= 1).float().mean())}
        out["out_" + sampler] = img.numpy()
    gen = torch.Generator().manual_seed(99)
    out["noise"] = torch.randn((2, 16, 16, 16), generator=gen).numpy()
    out["step_noise"] = torch.stack([torch.randn((2, 16, 16, 16), generator=gen) for _ in range(4)]).numpy()
    # distance between the samplers: a test that confuses them must fail
    report["heun_vs_euler_stochastic"] = rel_l2(torch.from_numpy(out["out_heun"]), torch.from_numpy(out["out_euler_stochastic"]))
    np.savez_compressed(os.path.join(GOLD, "sampler_micro_variants.npz"), **out)
    with open(os.path.join(GOLD, "generation_report_samplers.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
