"""Import the reference's MMDiT (src.models.diff_model) in THIS container only.

The reference is pure Python but three of its imports are absent here
(SURVEY.md 8c): xformers.ops.swiglu_op (MLP.py:3, Transformer_Block_Dual.py:10),
colorama (PositionalEncoding.py:3) and src.helpers.VAE_T5_CLIP_inference
(diff_model.py:15 -> open_clip/diffusers).  They are replaced by minimal stubs in
sys.modules.  The SwiGLU stub restates xformers==0.0.29.post3's documented eager
semantics (SwiGLUEagerOp):  w3( silu(x@w1.T+b1) * (x@w2.T+b2) ) + b3  with
[w1;w2] packed as w12 (2h x d).  Nothing here travels to the GPU box; this
module is only used by tools/make_goldens.py.
"""
import sys
import types

import torch
from torch import nn

REFERENCE_ROOT = "/root/reference"


def _install_stubs():
    if "xformers.ops.swiglu_op" in sys.modules:
        return

    class SwiGLU(nn.Module):
        def __init__(self, in_features, hidden_features, out_features=None, bias=True, *, _pack_weights=True):
            super().__init__()
            out_features = out_features or in_features
            self.w12 = nn.Linear(in_features, 2 * hidden_features, bias=bias)
            self.w3 = nn.Linear(hidden_features, out_features, bias=bias)

        def forward(self, x):
            x1, x2 = self.w12(x).chunk(2, dim=-1)
            return self.w3(torch.nn.functional.silu(x1) * x2)

    xf = types.ModuleType("xformers")
    xops = types.ModuleType("xformers.ops")
    xsw = types.ModuleType("xformers.ops.swiglu_op")
    xsw.SwiGLU = SwiGLU
    xf.ops = xops
    xops.swiglu_op = xsw
    sys.modules["xformers"] = xf
    sys.modules["xformers.ops"] = xops
    sys.modules["xformers.ops.swiglu_op"] = xsw

    col = types.ModuleType("colorama")

    class _Fore:
        def __getattr__(self, k):
            return ""

    col.Fore = _Fore()
    sys.modules["colorama"] = col

    vae = types.ModuleType("src.helpers.VAE_T5_CLIP_inference")

    class VAE_T5_CLIP_inference:  # never instantiated by the goldens
        def __init__(self, *a, **k):
            raise RuntimeError("text encoders / VAE are not available in this container")

    vae.VAE_T5_CLIP_inference = VAE_T5_CLIP_inference
    sys.modules["src.helpers.VAE_T5_CLIP_inference"] = vae


def import_reference():
    """Returns the reference's src.models.diff_model module."""
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import importlib
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        mod = importlib.import_module("src.models.diff_model")
    return mod
