"""MMDiT block: mirror of the reference's src/blocks/Transformer_Block_Dual.py (ctor 15-53, forward 56-77)."""
from types import SimpleNamespace as NS

import torch
from torch import nn

from .. import engine
from ..packing import Pack
from .Attention import Attention
from .MLP import MLP
from .Norm import Norm


class _BlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, blk, orig_shape, X, c, y, *params):
        m = blk._mode()
        w = blk.weights(m)
        B, N, d = X.shape
        Mt = c.shape[1]
        dims = (B, N, Mt, blk.attn.num_heads, d)
        rope = blk.attn.rotary_emb.tables(orig_shape[-2] // 2, orig_shape[-1] // 2, X.device)
        X2, C2, sv = engine.block_fwd(m, w, X.reshape(B * N, d).float().contiguous(), c.reshape(B * Mt, d).float().contiguous(),
                                      m.act(y.contiguous()), dims, rope)
        ctx.blk, ctx.m, ctx.sv, ctx.dims, ctx.rope = blk, m, sv, dims, rope
        return X2.view(B, N, d), C2.view(B, Mt, d)

    @staticmethod
    def backward(ctx, dX, dC):
        blk, m, sv = ctx.blk, ctx.m, ctx.sv
        B, N, Mt, H, d = ctx.dims
        w = blk.weights(m)
        dX_, dC_, dy, g = engine.block_bwd(m, w, sv, dX.reshape(B * N, d).float().contiguous(),
                                           None if dC is None else dC.reshape(B * Mt, d).float().contiguous(), None, ctx.dims, ctx.rope)
        engine.wgrad_join(dX.device)
        ctx.sv = None
        out = {}
        blk.scatter_grads(g, out)
        return (None, None, dX_.view(B, N, d), dC_.view(B, Mt, d), dy) + tuple(out.get(id(p)) for p in blk._param_list())


class Transformer_Block_Dual(nn.Module):
    def __init__(self, dim, c_dim, hidden_scale=4.0, num_heads=8, attn_type="softmax", MLP_type="gelu", causal=False,
                 positional_encoding="absolute", RoPE_Scale=1, kv_merge_attn=False, qk_half_dim=False,
                 checkpoint_MLP=True, checkpoint_attn=True, layer_idx=None, last=False):
        super().__init__()
        if c_dim != dim:
            raise RuntimeError("Transformer_Block_Dual: c_dim must equal dim (as constructed by diff_model.py:151)")
        if MLP_type == "swiglu_old":
            raise RuntimeError("MLP_type 'swiglu_old' (legacy checkpoints) is not implemented; use 'swiglu' or 'gelu'")
        # Activation checkpointing flags are accepted for signature parity; with 288 GB of HBM the
        # HIP path keeps activations resident (forward results are bit-identical either way).
        self.checkpoint_MLP, self.checkpoint_attn = checkpoint_MLP, checkpoint_attn
        self.last = last
        self.y_proj = nn.Sequential(nn.Linear(c_dim, c_dim), nn.SiLU())
        self.MLP_x = MLP(dim, hidden_scale, act=MLP_type)
        if not self.last:
            self.MLP_c = MLP(dim, hidden_scale, act=MLP_type)
        self.attn = Attention(dim, num_heads=num_heads, attn_type=attn_type, causal=causal, positional_encoding=positional_encoding,
                              RoPE_Scale=RoPE_Scale, kv_merge_attn=kv_merge_attn, qk_half_dim=qk_half_dim, layer_idx=layer_idx, dual=True, last=last)
        self.norm1_x = Norm(dim, c_dim)
        self.norm2_x = Norm(dim, c_dim)
        self.norm1_c = Norm(dim, c_dim)
        if not self.last:
            self.norm2_c = Norm(dim, c_dim)
        self.scale1_x = nn.Linear(c_dim, dim, bias=False)
        self.scale2_x = nn.Linear(c_dim, dim, bias=False)
        if not self.last:
            self.scale1_c = nn.Linear(c_dim, dim, bias=False)
            self.scale2_c = nn.Linear(c_dim, dim, bias=False)
        # one GEMM produces every modulation vector of the block; row order = engine.MOD_NAMES_FULL
        mods = [self.norm1_x.c_shift, self.norm1_x.c_scale, self.scale1_x, self.norm2_x.c_shift, self.norm2_x.c_scale, self.scale2_x,
                self.norm1_c.c_shift, self.norm1_c.c_scale]
        if not self.last:
            mods += [self.scale1_c, self.norm2_c.c_shift, self.norm2_c.c_scale, self.scale2_c]
        self._pmod = Pack([l.weight for l in mods])
        self._py = Pack([self.y_proj[0].weight])
        self.precision = "fast"

    def _mode(self):
        return engine.FAST if self.precision == "fast" else engine.PARITY

    def _param_list(self):
        return [p for p in self.parameters() if p.requires_grad]

    def weights(self, m):
        w = self.attn.weights(m)
        w.Wy, w.by, w.Wmod, w.last = self._py.get(m), self.y_proj[0].bias.detach(), self._pmod.get(m), self.last
        w.mlp_x = self.MLP_x.weights(m)
        w.mlp_c = None if self.last else self.MLP_c.weights(m)
        return w

    def scatter_grads(self, g, out: dict):
        self.attn.scatter_grads(g, out)
        self._py.split_grad(g.Wy, out)
        out[id(self.y_proj[0].bias)] = g.by
        self._pmod.split_grad(g.Wmod, out)
        self.MLP_x.scatter_grads(g.mlp_x, out)
        if not self.last:
            self.MLP_c.scatter_grads(g.mlp_c, out)

    def forward(self, X, c, y, orig_shape):
        return _BlockFn.apply(self, tuple(orig_shape), X, c, y, *self._param_list())
