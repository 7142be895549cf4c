"""MLP: mirror of the reference's src/blocks/MLP.py (lines 7-39) and of the xformers SwiGLU module
it wraps (xformers==0.0.29.post3: packed w12 (2h,d)+bias, w3 (d,h)+bias)."""
from types import SimpleNamespace as NS

import torch
from torch import nn

from .. import engine, ops
from ..packing import Pack


class SwiGLU(nn.Module):
    """Parameter container with xformers.ops.swiglu_op.SwiGLU's attribute names (w12, w3)."""

    def __init__(self, in_features, hidden_features, out_features=None, bias=True):
        super().__init__()
        out_features = out_features or in_features
        self.w12 = nn.Linear(in_features, 2 * hidden_features, bias=bias)
        self.w3 = nn.Linear(hidden_features, out_features, bias=bias)


class _MLPFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, X, *params):
        m = mod._mode()
        w = mod.weights(m)
        shp = X.shape
        xa = m.act(X.reshape(-1, shp[-1]).contiguous())
        gu, h = engine.mlp_core_fwd(m, w, xa)
        out = ops.gemm(h, w.Wdown, bias=w.bdown, out_dtype=torch.float32, precision=m.prec)
        ctx.mod, ctx.m, ctx.shp = mod, m, shp
        ctx.save_for_backward(xa, gu, h)
        return out.view(*shp[:-1], out.shape[-1])

    @staticmethod
    def backward(ctx, dout):
        xa, gu, h = ctx.saved_tensors
        m, mod = ctx.m, ctx.mod
        w = mod.weights(m)
        d2 = dout.reshape(-1, dout.shape[-1]).contiguous()
        dbdown = torch.zeros(d2.shape[1], dtype=torch.float32, device=d2.device)
        ops.colsum(d2, dbdown)
        dx, g = engine.mlp_core_bwd(m, w, m.act(d2), xa, gu, h, d2.device)
        engine.wgrad_join(d2.device)
        return None, dx.float().view(ctx.shp), g.Wup, g.bup, g.Wdown, dbdown


class MLP(nn.Module):
    def __init__(self, dim, hidden_scale=4.0, act="swiglu"):
        super().__init__()
        self.proj_size = int(dim * hidden_scale)
        self.act_ = act
        if act == "swiglu":
            self.MLP = SwiGLU(dim, self.proj_size, dim)
        elif act == "gelu":
            self.lin_up = nn.Linear(dim, self.proj_size)
            self.lin_down = nn.Linear(self.proj_size, dim)
        else:
            raise RuntimeError(f"MLP act must be 'swiglu' or 'gelu', got {act}")
        self._pup, self._pdown = Pack([self._up.weight]), Pack([self._down.weight])
        self.precision = "fast"

    # the two Linears under whichever names the reference's MLP_type registers them (not re-registered)
    @property
    def _up(self):
        return self.MLP.w12 if self.act_ == "swiglu" else self.lin_up

    @property
    def _down(self):
        return self.MLP.w3 if self.act_ == "swiglu" else self.lin_down

    def _mode(self):
        return engine.FAST if self.precision == "fast" else engine.PARITY

    def weights(self, m):
        return NS(Wup=self._pup.get(m), bup=self._up.bias.detach(), Wdown=self._pdown.get(m), bdown=self._down.bias.detach(),
                  hidden=self.proj_size, gelu=self.act_ == "gelu")

    def scatter_grads(self, g, out: dict):
        self._pup.split_grad(g.Wup, out)
        self._pdown.split_grad(g.Wdown, out)
        out[id(self._up.bias)] = g.bup
        out[id(self._down.bias)] = g.bdown

    def forward(self, X):
        return _MLPFn.apply(self, X, self._up.weight, self._up.bias, self._down.weight, self._down.bias)
