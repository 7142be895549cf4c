"""Where does the B-depth parity margin go?  HIP parity mode vs (a) the reference golden, (b) exact arithmetic with the reference's
rounding points (the oracle run in float64: tools/scratch_b_plain_fp64.npz, written in the build container)."""
import os
import sys

import numpy as np
import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import sd3_amd  # noqa: E402,F401
from oracle.weights import make_inputs, make_state_dict  # noqa: E402
from sd3_amd.models.diff_model import diff_model  # noqa: E402


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm())


cfg = dict(dim=768, num_heads=12, num_blocks=12)
net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=torch.device("cuda:0"),
                 positional_encoding="RoPE2d", **cfg)
net.load_state_dict(make_state_dict(0, **cfg))
net.set_precision("parity")
x, c, cp = make_inputs(0, 2, 32, 32, text_scale=30.0)
t = torch.tensor([0.25, 0.8])
nulls = [torch.tensor(m).bool() for m in ([0, 1], [0, 0], [1, 0])]
with torch.no_grad():
    v = net(x.cuda(), t, c.clone().cuda(), cp.clone().cuda(), *nulls)
gold = torch.from_numpy(np.load(os.path.join(ROOT, "tests", "golden", "forward_b_plain.npz"))["v"])
exact = torch.from_numpy(np.load(os.path.join(ROOT, "tools", "scratch_b_plain_fp64.npz"))["v64"])
print(f"HIP parity vs reference golden (8 CPU threads): {rel(v, gold):.4e}")
print(f"HIP parity vs exact arithmetic (fp64 oracle):    {rel(v, exact):.4e}")
print(f"reference golden vs exact arithmetic:            {rel(gold, exact):.4e}")
