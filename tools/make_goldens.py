"""Generate golden fixtures under tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference).  It imports the
reference's diff_model through tools/ref_import.py, loads the seeded synthetic
weights of oracle/weights.py into it and records inputs-checksums and outputs of
the reference's own code for the hot path.  Only data is written (npz / json):
no reference source or bytecode.

Usage:  python tools/make_goldens.py            (writes tests/golden/*)
"""
import contextlib
import io
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from oracle import mmdit_oracle as O  # noqa: E402
from oracle.weights import make_inputs, make_state_dict, state_dict_spec  # noqa: E402
from ref_import import import_reference  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)
torch.set_num_threads(8)

CONFIGS = {
    # name: (model kwargs, latent h, w)
    "micro": (dict(dim=128, num_heads=2, num_blocks=3), 16, 16),
    "xs": (dict(dim=256, num_heads=4, num_blocks=2), 64, 64),       # BASELINE.json configs[0]
    "b": (dict(dim=768, num_heads=12, num_blocks=12), 32, 32),      # BASELINE.json configs[1] depth/shape
}


def build_ref(refmod, cfg, MLP_type="swiglu", seed=0):
    with contextlib.redirect_stdout(io.StringIO()):
        net = refmod.diff_model(inCh=16, class_dim=768, patch_size=2, dim=cfg["dim"], hidden_scale=4.0,
                                num_heads=cfg["num_heads"], attn_type="softmax_flash", MLP_type=MLP_type,
                                num_blocks=cfg["num_blocks"], device="cpu", positional_encoding="RoPE2d",
                                checkpoint_MLP=False, checkpoint_attn=False)
    sd = make_state_dict(seed, MLP_type=MLP_type, **cfg)
    ref_sd = net.state_dict()
    assert list(ref_sd.keys()) == list(sd.keys()), "state_dict key order differs from oracle/weights.py spec"
    for k in sd:
        assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), k
    # rotary freqs: our closed form must equal the reference's parameter
    for k in sd:
        if k.endswith("freqs"):
            assert torch.equal(sd[k], ref_sd[k]), k
    net.load_state_dict(sd, strict=True)
    return net, sd


def checksum(*ts):
    return [float(t.double().sum()) for t in ts] + [float(t.double().abs().sum()) for t in ts]


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main():
    refmod = import_reference()
    report = {}

    # ---- 1. state-dict layout -------------------------------------------------
    for name, (cfg, _, _) in CONFIGS.items():
        for mt in (["swiglu", "gelu"] if name == "micro" else ["swiglu"]):
            net, _ = build_ref(refmod, cfg, mt)
            spec = [[k, list(v.shape), str(v.dtype)] for k, v in net.state_dict().items()]
            named = [n for n, _ in net.named_parameters()]
            nograd = [n for n, p in net.named_parameters() if not p.requires_grad]
            with open(os.path.join(GOLD, f"state_dict_spec_{name}_{mt}.json"), "w") as f:
                json.dump({"state_dict": spec, "named_parameters": named, "no_grad": nograd,
                           "num_params": sum(p.numel() for p in net.parameters())}, f)

    # ---- 2. leaf functions -----------------------------------------------------------------
    leaf = {}
    pe_mod = refmod.PositionalEncoding(8, device="cpu")
    leaf["pe8_500"] = pe_mod(torch.tensor([500.0])).numpy()
    pe_mod = refmod.PositionalEncoding(256, device="cpu")
    tt = torch.tensor([0.02, 0.5, 0.98]) * 1000.0
    leaf["pe256_t"] = tt.numpy()
    leaf["pe256"] = pe_mod(tt).numpy()
    from src.blocks.rotary_embedding import RotaryEmbedding, apply_rotary_emb
    rot = RotaryEmbedding(32, use_xpos=False, interpolate_factor=1.0)
    leaf["axial_3_5"] = rot.get_axial_freqs(3, 5).detach().numpy()
    leaf["axial_32_32"] = rot.get_axial_freqs(32, 32).detach().numpy()
    g = torch.Generator().manual_seed(7)
    tq = torch.randn((2, 3, 3, 5, 64), generator=g)
    leaf["rope_in"] = tq.numpy()
    leaf["rope_out"] = apply_rotary_emb(rot.get_axial_freqs(3, 5), tq).detach().numpy()
    from src.blocks.patchify import unpatchify
    ramp = torch.arange(2 * 15 * 64, dtype=torch.float32).reshape(2, 15, 64)
    leaf["unpatchify_ramp_6_10"] = unpatchify(ramp, (2, 2), (6, 10)).numpy()
    from src.blocks.ImagePositionalEncoding import PatchEmbed
    pemb = PatchEmbed(height=256, width=256, patch_size=2, in_channels=16, embed_dim=8, layer_norm=False, flatten=True,
                      bias=False, interpolation_scale=1, pos_embed_type="RoPE2d", pos_embed_max_size=256)
    wpe = torch.randn((8, 16, 2, 2), generator=g)
    pemb.proj.weight.data.copy_(wpe)
    xin = torch.randn((2, 16, 6, 10), generator=g)
    leaf["patch_w"], leaf["patch_in"] = wpe.numpy(), xin.numpy()
    leaf["patch_out"] = pemb(xin).detach().numpy()
    from src.blocks.Norm import Norm
    nm = Norm(32, 32)
    xs, ys = torch.randn((2, 5, 32), generator=g), torch.randn((2, 32), generator=g)
    leaf["norm_x"], leaf["norm_y"] = xs.numpy(), ys.numpy()
    leaf["norm_wscale"], leaf["norm_wshift"] = nm.c_scale.weight.detach().numpy(), nm.c_shift.weight.detach().numpy()
    leaf["norm_out"] = nm(xs, ys).detach().numpy()
    from src.helpers.TimeSampler import TimeSampler  # noqa: F401  (import check only)
    np.savez_compressed(os.path.join(GOLD, "leaf_functions.npz"), **leaf)

    # ---- 3. forward goldens -----------------------------------------------------------------
    cases = [
        # (case name, config, h, w, seed, t, text_scale, nulls (pooled, gemma, bert))
        ("micro_plain", "micro", 16, 16, 0, [0.3, 0.7], 1.0, None),
        ("micro_nulls", "micro", 16, 16, 1, [0.02, 0.98], 30.0, ([1, 0], [0, 1], [1, 1])),
        ("micro_nonsquare", "micro", 12, 20, 2, [0.5, 0.5], 30.0, ([0, 0], [1, 0], [0, 0])),
        ("xs_plain", "xs", 64, 64, 0, [0.02, 0.98], 1.0, None),
        ("xs_gemma30_nulls", "xs", 64, 64, 1, [0.5, 0.3], 30.0, ([1, 0], [1, 0], [1, 0])),
        ("xs_nonsquare", "xs", 48, 80, 2, [0.98, 0.5], 30.0, ([1, 1], [1, 1], [1, 1])),
        ("b_plain", "b", 32, 32, 0, [0.25, 0.8], 30.0, ([0, 1], [0, 0], [1, 0])),
    ]
    nets = {}
    for case, cname, h, w, seed, tvals, tscale, nulls in cases:
        cfg = CONFIGS[cname][0]
        if cname not in nets:
            nets[cname] = build_ref(refmod, cfg)
        net, sd = nets[cname]
        x, c, cp = make_inputs(seed, 2, h, w, text_scale=tscale)
        t = torch.tensor(tvals)
        nl = [None] * 3 if nulls is None else [torch.tensor(n).bool() for n in nulls]
        out = {"inputs_checksum": np.array(checksum(x, c, cp))}
        taps = {}
        hooks = []
        if case == "micro_plain":
            b0 = net.blocks[0]
            for nm_, mod in [("y_proj", b0.y_proj), ("norm1_x", b0.norm1_x), ("norm1_c", b0.norm1_c),
                             ("mlp_x", b0.MLP_x), ("q_norm_x", b0.attn.q_norm_x), ("k_norm_c", b0.attn.k_norm_c)]:
                hooks.append(mod.register_forward_hook(lambda m, i, o, nm_=nm_: taps.__setitem__(nm_, o.detach().clone())))
            hooks.append(b0.attn.register_forward_hook(lambda m, i, o: taps.update(attn_x=o[0].detach().clone(), attn_c=o[1].detach().clone())))
            for bi, blk in enumerate(net.blocks):
                hooks.append(blk.register_forward_hook(lambda m, i, o, bi=bi: taps.update({f"block{bi}_X": o[0].detach().clone(), f"block{bi}_c": o[1].detach().clone()})))
        cc, cpc = c.clone(), cp.clone()
        with torch.no_grad():
            v = net(x.clone(), t, cc, cpc, *nl)
        for hk in hooks:
            hk.remove()
        out["v"] = v.numpy()
        out["c_after"] = np.array(checksum(cc, cpc))  # in-place null masking is part of the contract
        for k, val in taps.items():
            out["tap_" + k] = val.numpy()
        np.savez_compressed(os.path.join(GOLD, f"forward_{case}.npz"), **out)

        # cross-check the oracle right here so a drift is caught at generation time
        ocfg = O.OracleConfig(**cfg)
        with torch.no_grad():
            vo = O.forward(sd, ocfg, x.clone(), t, c.clone(), cp.clone(), *nl)
            vx = O.forward(sd, O.OracleConfig(**cfg, attn_core="fp32"), x.clone(), t, c.clone(), cp.clone(), *nl)
            vf = O.forward(sd, O.OracleConfig(**cfg, attn_core="flash_bf16", gemm="bf16"), x.clone(), t, c.clone(), cp.clone(), *nl)
        report[case] = {"oracle_vs_ref": rel_l2(vo, v), "exact_attn_vs_ref": rel_l2(vx, v), "fast_rounding_vs_ref": rel_l2(vf, v),
                        "v_std": float(v.std()), "max_abs": float((vo - v).abs().max())}
        print(case, report[case], flush=True)

    # ---- 4. gradients (loss = v.pow(2).mean()) ------------------------------------------------
    for cname, h, w in [("micro", 16, 16), ("xs", 64, 64)]:
        net, sd = nets[cname]
        x, c, cp = make_inputs(5, 2, h, w, text_scale=30.0)
        t = torch.tensor([0.4, 0.9])
        nl = [torch.tensor(n).bool() for n in ([0, 1], [0, 0], [1, 0])]
        net.zero_grad()
        v = net(x.clone(), t, c.clone(), cp.clone(), *nl)
        loss = v.pow(2).mean()
        loss.backward()
        out = {"loss": np.array(float(loss)), "v": v.detach().numpy()}
        names, norms, samples = [], [], []
        gs = torch.Generator().manual_seed(11)
        for n, p in net.named_parameters():
            if p.grad is None:
                continue
            names.append(n)
            norms.append(float(p.grad.double().norm()))
            idx = torch.randint(0, p.numel(), (8,), generator=gs)
            samples.append(p.grad.flatten()[idx].numpy())
            if cname == "micro" and p.numel() <= 4096:
                out["grad__" + n] = p.grad.numpy()
        out["grad_names"] = np.array(names)
        out["grad_norms"] = np.array(norms)
        out["grad_samples"] = np.stack(samples)
        np.savez_compressed(os.path.join(GOLD, f"grads_{cname}.npz"), **out)
        net.zero_grad()

    # ---- 5. two full train steps (model_trainer.py:378-503 restated around the real model) ----
    cfg = CONFIGS["micro"][0]
    net, sd = build_ref(refmod, cfg)
    optim = torch.optim.AdamW(net.parameters(), lr=1e-3, eps=1e-8, weight_decay=0.01, betas=(0.9, 0.999))
    from transformers import get_constant_schedule_with_warmup
    sched = get_constant_schedule_with_warmup(optimizer=optim, num_warmup_steps=2)
    losses, lrs = [], []
    for step in range(3):
        x0, c, cp = make_inputs(20 + step, 2, 16, 16, text_scale=30.0)
        g = torch.Generator().manual_seed(300 + step)
        eps = torch.randn(x0.shape, generator=g)
        t = torch.sigmoid(torch.randn((2,), generator=g))
        nl = [(torch.rand((2,), generator=g) < p) for p in (0.1, 0.316, 0.316)]
        tt = t[:, None, None, None]
        x_t = (1 - tt) * x0 + tt * eps
        lrs.append(optim.param_groups[0]["lr"])
        v = net(x_t, t, c, cp, *nl)
        loss = torch.nn.MSELoss(reduction="none")(v, (eps - x0)).flatten(1, -1).mean()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
        optim.step()
        sched.step(step + 1)
        optim.zero_grad()
        losses.append(float(loss))
    fp = {"losses": np.array(losses), "lrs": np.array(lrs)}
    gs = torch.Generator().manual_seed(13)
    names, sums, samples = [], [], []
    for n, p in net.named_parameters():
        names.append(n)
        sums.append(float(p.detach().double().sum()))
        idx = torch.randint(0, p.numel(), (4,), generator=gs)
        samples.append(p.detach().flatten()[idx].numpy())
    fp["param_names"], fp["param_sums"], fp["param_samples"] = np.array(names), np.array(sums), np.stack(samples)
    np.savez_compressed(os.path.join(GOLD, "train_steps_micro.npz"), **fp)

    # ---- 6. checkpoint layout (diff_model.py:489-536, 553-578) ---------------------------------
    with tempfile.TemporaryDirectory() as td:
        net.saveModel(td, EMA_state_dict=net.state_dict(), optimizer=optim, scheduler=sched,
                      grad_scalar=torch.amp.GradScaler("cuda", enabled=False), step=7)
        listing = sorted(os.listdir(td))
        with open(os.path.join(td, "model_params_7s.json")) as f:
            params_json = json.load(f)
        optim_sd = torch.load(os.path.join(td, "optim_7s.pkl"), weights_only=False)
        layout = {"files": listing, "model_params": params_json,
                  "optim_param_group_keys": sorted(optim_sd["param_groups"][0].keys()),
                  "optim_num_params": len(optim_sd["param_groups"][0]["params"]),
                  "optim_state_keys": sorted(next(iter(optim_sd["state"].values())).keys())}
    with open(os.path.join(GOLD, "checkpoint_layout.json"), "w") as f:
        json.dump(layout, f, indent=1)

    # ---- 7. sampler: the reference's own sample_imgs loop with stand-in text/VAE objects -------
    class _Cfg:
        latent_channels, shift_factor, scaling_factor = 16, 0.1159, 0.3611

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

    net, sd = nets["micro"]
    _, th, tp = make_inputs(40, 1, 16, 16, text_scale=30.0)
    net.text_encoders = _Enc(th, tp)
    gen = torch.Generator().manual_seed(99)
    img = net.sample_imgs(2, 4, ["x"], cfg_scale=3.0, width=128, height=128, sampler="euler", generator=gen)
    gen = torch.Generator().manual_seed(99)
    noise = torch.randn((2, 16, 16, 16), generator=gen)
    lat = O.euler_cfg_sample(sd, O.OracleConfig(**CONFIGS["micro"][0]), noise, th, tp, 4, 3.0)
    report["sampler"] = {"oracle_vs_ref": rel_l2(((lat - 0.1159) / 0.3611).clamp(-1, 1), img)}
    np.savez_compressed(os.path.join(GOLD, "sampler_micro.npz"), noise=noise.numpy(), out=img.numpy())
    del net.text_encoders

    # ---- 8. gelu MLP variant ---------------------------------------------------------------------
    net, sd = build_ref(refmod, CONFIGS["micro"][0], "gelu")
    x, c, cp = make_inputs(3, 2, 16, 16, text_scale=30.0)
    t = torch.tensor([0.1, 0.6])
    with torch.no_grad():
        v = net(x.clone(), t, c.clone(), cp.clone())
        vo = O.forward(sd, O.OracleConfig(**CONFIGS["micro"][0], MLP_type="gelu"), x.clone(), t, c.clone(), cp.clone())
    report["micro_gelu"] = {"oracle_vs_ref": rel_l2(vo, v)}
    np.savez_compressed(os.path.join(GOLD, "forward_micro_gelu.npz"), v=v.numpy(), inputs_checksum=np.array(checksum(x, c, cp)))

    with open(os.path.join(GOLD, "generation_report.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))
    tot = sum(os.path.getsize(os.path.join(GOLD, f)) for f in os.listdir(GOLD))
    print("golden dir bytes:", tot)


if __name__ == "__main__":
    main()
