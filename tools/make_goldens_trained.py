"""Goldens at the reference's OWN trained shape (src/train.py:35-41: num_blocks = 19, dim = 64 * 19 = 1216, num_heads = 19;
README.md:251-262: trained 256^2 -> 512^2 -> 1024^2) from the REAL reference, imported in the build container through
tools/ref_import.py exactly like tools/make_goldens.py (seeded weights and inputs of oracle/weights.py).  d = 1216 is not a
multiple of 256, N = 1216 is 4.75 tiles of 256, H = 19 is odd: shapes none of the micro / XS / B / L fixtures take.  The width
and head count are the trained ones; the depth is 3 blocks (first / middle / last-block asymmetry) so that the fixture stays small
and the CPU oracle finishes in seconds -- the 19-block model is exercised by size-independent properties on the GPU
(tests/test_trained_shape_gpu.py).  Writes only data:

  tests/golden/forward_trained_sq.npz      32x32 latents (256^2 stage), batch 2, nulls, t = [0.25, 0.8]; v + 8 token rows of every
                                           block's outputs (forward hooks on the reference's blocks) + their checksums
  tests/golden/forward_trained_nonsq.npz   24x40 latents (an aspect-ratio bucket), batch 2
  tests/golden/grads_trained.npz           loss = v.pow(2).mean(): grad norms + 8 samples per parameter
  tests/golden/state_dict_spec_trained_swiglu.json   key order / shapes at 19 blocks (the real checkpoint's layout)
  tests/golden/generation_report_trained.json        oracle-vs-reference distances at generation time

Usage:  python tools/make_goldens_trained.py
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
from make_goldens import GOLD, build_ref, checksum, rel_l2  # noqa: E402
from ref_import import import_reference  # noqa: E402

T_CFG = dict(dim=1216, num_heads=19, num_blocks=3)
CASES = [
    # (case, h, w, seed, t, nulls)
    ("trained_sq", 32, 32, 70, [0.25, 0.8], ([0, 1], [0, 0], [1, 0])),
    ("trained_nonsq", 24, 40, 71, [0.6, 0.05], None),
]


def main():
    torch.set_num_threads(8)
    refmod = import_reference()
    report = {}

    # the real checkpoint's layout: 19 blocks (state-dict spec only; no weights stored)
    net19, _ = build_ref(refmod, dict(dim=1216, num_heads=19, num_blocks=19))
    spec = [[k, list(v.shape), str(v.dtype)] for k, v in net19.state_dict().items()]
    with open(os.path.join(GOLD, "state_dict_spec_trained_swiglu.json"), "w") as f:
        json.dump({"state_dict": spec, "named_parameters": [n for n, _ in net19.named_parameters()],
                   "no_grad": [n for n, p in net19.named_parameters() if not p.requires_grad],
                   "num_params": sum(p.numel() for p in net19.parameters())}, f)
    report["num_params_19_blocks"] = sum(p.numel() for p in net19.parameters())
    del net19

    net, sd = build_ref(refmod, T_CFG)
    sd64 = {k: v.double() for k, v in sd.items()}
    for case, h, w, seed, tvals, nulls in CASES:
        x, c, cp = make_inputs(seed, 2, h, w, text_scale=30.0)
        t = torch.tensor(tvals)
        nl = [None] * 3 if nulls is None else [torch.tensor(n).bool() for n in nulls]
        taps, hooks = {}, []
        for bi, blk in enumerate(net.blocks):
            hooks.append(blk.register_forward_hook(lambda m, i, o, bi=bi: taps.update({f"block{bi}_X": o[0].detach().clone(), f"block{bi}_c": o[1].detach().clone()})))
        cc, cpc = c.clone(), cp.clone()
        with torch.no_grad():
            v = net(x.clone(), t, cc, cpc, *nl)
        for hk in hooks:
            hk.remove()
        out = {"inputs_checksum": np.array(checksum(x, c, cp)), "v": v.numpy(), "c_after": np.array(checksum(cc, cpc))}
        for k, val in taps.items():
            # small fixtures: 8 seeded token rows of every block output (both streams) + whole-tensor checksums
            rows = torch.randperm(val.shape[1], generator=torch.Generator().manual_seed(17))[:8].sort().values
            out["taprows_" + k] = rows.numpy()
            out["tap_" + k] = val[:, rows].numpy()
            out["tapsum_" + k] = np.array(checksum(val))
        np.savez_compressed(os.path.join(GOLD, f"forward_{case}.npz"), **out)
        with torch.no_grad():
            vo = O.forward(sd, O.OracleConfig(**T_CFG), x.clone(), t, c.clone(), cp.clone(), *nl)
            vf = O.forward(sd, O.OracleConfig(**T_CFG, attn_core="flash_bf16", gemm="bf16"), x.clone(), t, c.clone(), cp.clone(), *nl)
        report[case] = {"oracle_vs_ref": rel_l2(vo, v), "fast_rounding_vs_ref": rel_l2(vf, v), "v_std": float(v.std())}
        # yardstick for the block taps: the same algorithm and rounding points in EXACT (float64) arithmetic vs the reference's fp32 run.
        # The text stream is ~1e-3 from the reference after block 1 whatever evaluates it (the bf16 roundings of the attention core flip
        # under fp32 summation-order noise; the stream is small, so its update dominates it).
        t64 = {}
        with torch.no_grad():
            v64 = O.forward(sd64, O.OracleConfig(**T_CFG, dtype=torch.float64), x.double(), t.double(), c.double(), cp.double(), *nl, taps=t64)
        report[case]["exact_vs_ref"] = rel_l2(v64, v)
        for bi, (X64, C64) in enumerate(t64["blocks"]):
            for nm, val in (("X", X64), ("c", C64)):
                k = f"block{bi}_{nm}"
                report[case]["exact_vs_ref_tap_" + k] = rel_l2(val[:, torch.from_numpy(out["taprows_" + k])], torch.from_numpy(out["tap_" + k]))
        print(case, report[case], flush=True)

    # stage 3 constructs the model with max_res = 1024, max_res_orig = 256 (src/train.py:47-48 -> RoPE_Scale 0.25 -> interpolate_factor 4):
    # pin that this changes NOTHING in the RoPE2d branch (get_axial_freqs takes the raw positions), i.e. the fixture above is the stage-3 model too
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        net4 = refmod.diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device="cpu",
                                 positional_encoding="RoPE2d", checkpoint_MLP=False, checkpoint_attn=False, max_res_orig=256, max_res=1024,
                                 update_max_res=True, **T_CFG)
    net4.load_state_dict(sd, strict=True)
    x, c, cp = make_inputs(70, 2, 32, 32, text_scale=30.0)
    with torch.no_grad():
        va = net(x.clone(), torch.tensor([0.25, 0.8]), c.clone(), cp.clone())
        vb = net4(x.clone(), torch.tensor([0.25, 0.8]), c.clone(), cp.clone())
    report["max_res_1024_ctor_changes_output"] = not torch.equal(va, vb)
    assert torch.equal(va, vb), "RoPE_Scale does reach the RoPE2d branch: the HIP path ignores it"
    del net4

    # gradients
    x, c, cp = make_inputs(72, 2, 32, 32, text_scale=30.0)
    t = torch.tensor([0.4, 0.9])
    nl = [torch.tensor(n).bool() for n in ([0, 1], [0, 0], [1, 0])]
    net.zero_grad()
    v = net(x.clone(), t, c.clone(), cp.clone(), *nl)
    loss = v.pow(2).mean()
    loss.backward()
    out = {"loss": np.array(float(loss.detach()))}
    names, norms, samples = [], [], []
    gs = torch.Generator().manual_seed(11)
    for n, p in net.named_parameters():
        if p.grad is None:
            continue
        names.append(n)
        norms.append(float(p.grad.double().norm()))
        idx = torch.randint(0, p.numel(), (8,), generator=gs)
        samples.append(p.grad.flatten()[idx].numpy())
    out["grad_names"], out["grad_norms"], out["grad_samples"] = np.array(names), np.array(norms), np.stack(samples)
    np.savez_compressed(os.path.join(GOLD, "grads_trained.npz"), **out)

    with open(os.path.join(GOLD, "generation_report_trained.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
