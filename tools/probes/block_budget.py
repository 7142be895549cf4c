"""Where a single HIP block (bf16 mode) leaves the rounding-matched oracle: block 0 fed with the oracle's own stage inputs, stage by stage.
    python tools/probes/block_budget.py [b|trained]"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sd3_amd  # noqa: E402,F401
from oracle import mmdit_oracle as O  # noqa: E402
from oracle.weights import make_inputs, make_state_dict  # noqa: E402
from sd3_amd import ops  # noqa: E402
from sd3_amd.models.diff_model import diff_model  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "b"
cfg = dict(dim=768, num_heads=12, num_blocks=12) if which == "b" else dict(dim=1216, num_heads=19, num_blocks=3)
seed = 0 if which == "b" else 70


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


rb = lambda t: t.to(torch.bfloat16).float()
net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=torch.device("cuda:0"),
                 positional_encoding="RoPE2d", checkpoint_MLP=False, checkpoint_attn=False, **cfg)
sd = make_state_dict(0, **cfg)
net.load_state_dict(sd)
net.set_precision("fast")
x, c, cp = make_inputs(seed, 2, 32, 32, text_scale=30.0)
t = torch.tensor([0.25, 0.8])
for core in ("flash_bf16", "flash_bf16_tiled"):
    ocfg = O.OracleConfig(**cfg, attn_core=core, gemm="bf16")
    tp = {}
    with torch.no_grad():
        O.forward(sd, ocfg, x.clone(), t, c.clone(), cp.clone(), taps=tp)
        b0, bt = net.blocks[0], tp["block0"]
        X0, C0, y = tp["x0"], tp["c0"], tp["y"]
        Xr, Cr = tp["blocks"][0]
        Xo, Co = b0(X0.cuda(), C0.cuda(), y.cuda(), x.shape)
        print(f"[{which} / oracle core {core}] block 0: X {rel(Xo, Xr):.2e}  c {rel(Co, Cr):.2e}   updates alone: X {rel(Xo.float().cpu() - X0, Xr - X0):.2e}  c {rel(Co.float().cpu() - C0, Cr - C0):.2e}")
        yp = bt["y_proj"].cuda()
        print(f"   norm1_x {rel(b0.norm1_x(X0.cuda(), yp), bt['norm1_x']):.2e}   norm1_c {rel(b0.norm1_c(C0.cuda(), yp), bt['norm1_c']):.2e}")
        ax, ac = b0.attn(bt["norm1_x"].cuda(), bt["norm1_c"].cuda(), x.shape)
        print(f"   attention module (oracle norm1 in): a_x {rel(rb(ax.float()), rb(bt['attn_x'])):.2e}  a_c {rel(rb(ac.float()), rb(bt['attn_c'])):.2e}")
        Q, K, V = [bt[k].to(torch.bfloat16).cuda().contiguous() for k in ("q", "k", "v")]
        N = 256
        Ox, Oc, _ = ops.attn_fwd(Q, K, V, N, 0.125, 0)
        B, H, S, _ = Q.shape
        core_ref = bt["attn_core"].permute(0, 2, 1, 3).reshape(B, S, H * 64)
        ex = O.attention_core(bt["q"], bt["k"], bt["v"], 0.125, "fp32").permute(0, 2, 1, 3).reshape(B, S, H * 64)
        mine = torch.cat([Ox, Oc], 1).float().cpu()
        print(f"   attention core (oracle q k v in): vs oracle core {rel(mine, core_ref):.2e} (image rows {rel(mine[:, :N], core_ref[:, :N]):.2e}, text rows {rel(mine[:, N:], core_ref[:, N:]):.2e}); "
              f"vs exact fp32 {rel(mine, ex):.2e}; oracle core vs exact {rel(core_ref, ex):.2e}")
        # the MLP on the oracle's own input
        p = "blocks.0."
        gate = lambda n: O._lin(ocfg, bt["y_proj"], sd[p + n + ".weight"])[:, None, :]
        Xa = O._act(ocfg, bt["attn_x"]) * gate("scale1_x") + X0
        n2 = O._act(ocfg, O.norm_modulate(Xa, bt["y_proj"], sd[p + "norm2_x.c_scale.weight"], sd[p + "norm2_x.c_shift.weight"], ocfg))
        print(f"   norm2_x {rel(b0.norm2_x(Xa.cuda(), yp), n2):.2e}   MLP_x (oracle norm2 in) {rel(rb(b0.MLP_x(n2.cuda()).float()), rb(bt['mlp_x'])):.2e}")
        Ca = O._act(ocfg, bt["attn_c"]) * gate("scale1_c") + C0
        n2c = O._act(ocfg, O.norm_modulate(Ca, bt["y_proj"], sd[p + "norm2_c.c_scale.weight"], sd[p + "norm2_c.c_shift.weight"], ocfg))
        mc = O.mlp(n2c, sd, p + "MLP_c.", ocfg)
        print(f"   norm2_c {rel(b0.norm2_c(Ca.cuda(), yp), n2c):.2e}   MLP_c (oracle norm2 in) {rel(rb(b0.MLP_c(n2c.cuda()).float()), rb(mc)):.2e}")
        print(f"   magnitudes: |X0| {float(X0.std()):.3f} |C0| {float(C0.std()):.4f} |Xr-X0| {float((Xr - X0).std()):.3f} |Cr-C0| {float((Cr - C0).std()):.3f} attn_c {float(bt['attn_c'].std()):.3f} mlp_c {float(mc.std()):.3f}")
