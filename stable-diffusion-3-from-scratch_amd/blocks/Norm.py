"""adaLN norm: mirror of the reference's src/blocks/Norm.py (class Norm, lines 5-22)."""
import torch
from torch import nn

from .. import engine, ops
from ..packing import Pack


class _NormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, X, y, w_shift, w_scale):
        m = mod._mode()
        B, L, d = X.shape
        W = mod._pack.get(m)                       # [c_shift | c_scale] rows
        ya = m.act(y.contiguous())
        modv = ops.gemm(ya, W, out_dtype=torch.float32, precision=m.prec)
        x2 = X.reshape(B * L, d).float().contiguous()
        out, mu, rs = ops.ln_modulate_fwd(x2, modv[:, d:], modv[:, :d], L, m.T)
        ctx.mod, ctx.m, ctx.shape = mod, m, (B, L, d)
        ctx.save_for_backward(x2, mu, rs, modv, ya, W)
        return out.view(B, L, d)

    @staticmethod
    def backward(ctx, dout):
        x2, mu, rs, modv, ya, W = ctx.saved_tensors
        m, (B, L, d) = ctx.m, ctx.shape
        dmod = torch.zeros_like(modv)
        dx = ops.ln_modulate_bwd(m.act(dout.reshape(B * L, d).contiguous()), x2, mu, rs, modv[:, d:], None, L, dmod[:, d:], dmod[:, :d])
        da = m.act(dmod)
        dy = ops.gemm(da, W, b_kmajor=True, out_dtype=torch.float32, precision=m.prec)
        gW = ops.gemm(da, ya, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, precision=m.prec)
        return None, dx.view(B, L, d), dy, gW[:d], gW[d:]


class Norm(nn.Module):
    """LayerNorm(no affine, eps 1e-5) * (1 + c_scale(y)) + c_shift(y)   (Norm.py:16-22).
    Inside diff_model the two Linears are folded into the block's modulation GEMM; forward() here is
    the standalone drop-in (same signature as the reference)."""

    def __init__(self, dim, c_dim):
        super().__init__()
        self.norm = nn.LayerNorm(dim, elementwise_affine=False)   # parameter-free; kept for attribute parity
        self.c_shift = nn.Linear(c_dim, dim, bias=False)
        self.c_scale = nn.Linear(c_dim, dim, bias=False)
        self._pack = Pack([self.c_shift.weight, self.c_scale.weight])
        self.precision = "fast"

    def _mode(self):
        return engine.FAST if self.precision == "fast" else engine.PARITY

    def forward(self, X, y=None):
        return _NormFn.apply(self, X, y, self.c_shift.weight, self.c_scale.weight)
