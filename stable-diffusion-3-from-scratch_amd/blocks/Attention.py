"""Joint image+text attention: mirror of the reference's src/blocks/Attention.py for the trained
configuration (dual stream, attn_type softmax / softmax_flash, RoPE2d, non-causal; ctor 16-114,
forward 118-135, 174-194, 258-293, 410-425).  Experimental attention types of the reference
(cosine*, relu, silu, exp, both, kv_merge_attn, qk_half_dim, 1-D RoPE, RoPE2dV2) are out of scope and
raise."""
from types import SimpleNamespace as NS

import torch
from torch import nn

from .. import engine, ops
from ..packing import Pack
from .rotary_embedding import RotaryEmbedding

F32, BF16 = torch.float32, torch.bfloat16


class _AttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, orig_shape, x, c, *params):
        m = mod._mode()
        w = mod.weights(m)
        B, N, d = x.shape
        Mt, H, S, dev = c.shape[1], mod.num_heads, x.shape[1] + c.shape[1], x.device
        rope = mod.rotary_emb.tables(orig_shape[-2] // 2, orig_shape[-1] // 2, dev)
        xa, ca = m.act(x.reshape(B * N, d).contiguous()), m.act(c.reshape(B * Mt, d).contiguous())
        qkv_x = ops.gemm(xa, w.Wqkv_x, out_dtype=m.T, precision=m.prec)
        qkv_c = ops.gemm(ca, w.Wqkv_c, out_dtype=m.T, precision=m.prec)
        Q = torch.empty((B, H, S, 64), dtype=BF16, device=dev)
        K, V = torch.empty_like(Q), torch.empty_like(Q)
        ops.qk_norm_rope_fwd(qkv_x, w.wq_x, w.wk_x, rope[0], rope[1], B, N, H, S, 0, Q, K, V)
        ops.qk_norm_rope_fwd(qkv_c, w.wq_c, w.wk_c, None, None, B, Mt, H, S, N, Q, K, V)
        Ox, Oc, lse = ops.attn_fwd(Q, K, V, N, mod.scale, m.attn_mode)
        Oxa = m.act(Ox.view(B * N, d))
        out_x = ops.gemm(Oxa, w.Wo_x, out_dtype=F32, precision=m.prec).view(B, N, d)
        if mod.last:
            Oca, out_c = None, ops.cast(Oc, F32)   # raw head-merged attention output (Attention.py:425)
        else:
            Oca = m.act(Oc.view(B * Mt, d))
            out_c = ops.gemm(Oca, w.Wo_c, out_dtype=F32, precision=m.prec).view(B, Mt, d)
        ctx.mod, ctx.m, ctx.rope, ctx.dims = mod, m, rope, (B, N, Mt, H, d)
        ctx.save_for_backward(xa, ca, qkv_x, qkv_c, Q, K, V, Ox, Oc, lse, Oxa, Oca)
        return out_x, out_c

    @staticmethod
    def backward(ctx, dox, doc):
        xa, ca, qkv_x, qkv_c, Q, K, V, Ox, Oc, lse, Oxa, Oca = ctx.saved_tensors
        mod, m, rope = ctx.mod, ctx.m, ctx.rope
        B, N, Mt, H, d = ctx.dims
        S, dev = N + Mt, xa.device
        w = mod.weights(m)
        gout = {}
        dax = m.act(dox.reshape(B * N, d).contiguous())
        dOx = ops.gemm(dax, w.Wo_x, b_kmajor=True, out_dtype=BF16, precision=m.prec)
        mod._po_x.split_grad(ops.gemm(dax, Oxa, a_kmajor=True, b_kmajor=True, out_dtype=F32, precision=m.prec), gout)
        if mod.last:
            dOc = None if doc is None else ops.cast(doc.reshape(B, Mt, d).float().contiguous(), BF16)
        else:
            dac = m.act(doc.reshape(B * Mt, d).contiguous())
            dOc = ops.gemm(dac, w.Wo_c, b_kmajor=True, out_dtype=BF16, precision=m.prec)
            mod._po_c.split_grad(ops.gemm(dac, Oca, a_kmajor=True, b_kmajor=True, out_dtype=F32, precision=m.prec), gout)
        dQ, dK, dV = ops.attn_bwd(Q, K, V, Ox, Oc, dOx, dOc, lse, N, mod.scale, m.T)
        gq_x, gk_x, gq_c, gk_c = [torch.zeros(64, dtype=F32, device=dev) for _ in range(4)]
        dqkv_x = ops.qk_norm_rope_bwd(dQ, dK, dV, qkv_x, w.wq_x, w.wk_x, rope[0], rope[1], B, N, H, S, 0, gq_x, gk_x, m.T)
        dqkv_c = ops.qk_norm_rope_bwd(dQ, dK, dV, qkv_c, w.wq_c, w.wk_c, None, None, B, Mt, H, S, N, gq_c, gk_c, m.T)
        dx = ops.gemm(dqkv_x, w.Wqkv_x, b_kmajor=True, out_dtype=F32, precision=m.prec).view(B, N, d)
        dc = ops.gemm(dqkv_c, w.Wqkv_c, b_kmajor=True, out_dtype=F32, precision=m.prec).view(B, Mt, d)
        mod._pqkv_x.split_grad(ops.gemm(dqkv_x, xa, a_kmajor=True, b_kmajor=True, out_dtype=F32, precision=m.prec), gout)
        mod._pqkv_c.split_grad(ops.gemm(dqkv_c, ca, a_kmajor=True, b_kmajor=True, out_dtype=F32, precision=m.prec), gout)
        gout[id(mod.q_norm_x.weight)], gout[id(mod.k_norm_x.weight)] = gq_x, gk_x
        gout[id(mod.q_norm_c.weight)], gout[id(mod.k_norm_c.weight)] = gq_c, gk_c
        return (None, None, dx, dc) + tuple(gout.get(id(p)) for p in mod._param_list())


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, attn_type="cosine", causal=False, emb_dim=None, positional_encoding="absolute", RoPE_Scale=1,
                 kv_merge_attn=False, qk_half_dim=False, layer_idx=None, dual=False, last=False):
        super().__init__()
        if attn_type not in ("softmax", "softmax_flash"):
            raise RuntimeError(f"attn_type must be 'softmax' or 'softmax_flash' on the HIP path, but got {attn_type}")
        if not dual or causal or kv_merge_attn or qk_half_dim or emb_dim is not None:
            raise RuntimeError("Attention: only the dual-stream, non-causal configuration of the trained model is implemented")
        if positional_encoding != "RoPE2d":
            raise RuntimeError("Attention: only positional_encoding='RoPE2d' is implemented")
        if dim // num_heads != 64 or dim % num_heads:
            raise RuntimeError("Attention: head_dim must be 64 (the reference's dim = 64*num_heads convention, train.py:37-39)")
        self.positional_encoding, self.kv_merge_attn, self.RoPE_Scale = positional_encoding, kv_merge_attn, RoPE_Scale
        self.layer_idx, self.dual, self.last = layer_idx, dual, last
        self.query_proj_x = nn.Linear(dim, dim, bias=False)
        self.key_proj_x = nn.Linear(dim, dim, bias=False)
        self.value_proj_x = nn.Linear(dim, dim, bias=False)
        self.out_proj_x = nn.Linear(dim, dim, bias=False)
        self.query_proj_c = nn.Linear(dim, dim, bias=False)
        self.key_proj_c = nn.Linear(dim, dim, bias=False)
        self.value_proj_c = nn.Linear(dim, dim, bias=False)
        if not self.last:
            self.out_proj_c = nn.Linear(dim, dim, bias=False)
        self.dim, self.num_heads = dim, num_heads
        self.head_dim_qk = self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.q_norm_x = nn.RMSNorm(self.head_dim_qk)
        self.k_norm_x = nn.RMSNorm(self.head_dim_qk)
        self.q_norm_c = nn.RMSNorm(self.head_dim_qk)
        self.k_norm_c = nn.RMSNorm(self.head_dim_qk)
        self.attn_type = attn_type   # both names run the same HIP flash kernel
        self.causal = causal
        self.rotary_emb = RotaryEmbedding(self.head_dim_qk // 2, use_xpos=False, interpolate_factor=1 / RoPE_Scale)
        self._pqkv_x = Pack([self.query_proj_x.weight, self.key_proj_x.weight, self.value_proj_x.weight])
        self._pqkv_c = Pack([self.query_proj_c.weight, self.key_proj_c.weight, self.value_proj_c.weight])
        self._po_x = Pack([self.out_proj_x.weight])
        self._po_c = None if self.last else Pack([self.out_proj_c.weight])
        self.precision = "fast"

    def _mode(self):
        return engine.FAST if self.precision == "fast" else engine.PARITY

    def _param_list(self):
        return [p for p in self.parameters() if p.requires_grad]

    def weights(self, m):
        return NS(Wqkv_x=self._pqkv_x.get(m), Wqkv_c=self._pqkv_c.get(m), Wo_x=self._po_x.get(m),
                  Wo_c=None if self.last else self._po_c.get(m),
                  wq_x=self.q_norm_x.weight.detach(), wk_x=self.k_norm_x.weight.detach(),
                  wq_c=self.q_norm_c.weight.detach(), wk_c=self.k_norm_c.weight.detach())

    def scatter_grads(self, g, out: dict):
        self._pqkv_x.split_grad(g.Wqkv_x, out)
        self._pqkv_c.split_grad(g.Wqkv_c, out)
        self._po_x.split_grad(g.Wo_x, out)
        if not self.last:
            self._po_c.split_grad(g.Wo_c, out)
        out[id(self.q_norm_x.weight)], out[id(self.k_norm_x.weight)] = g.wq_x, g.wk_x
        out[id(self.q_norm_c.weight)], out[id(self.k_norm_c.weight)] = g.wq_c, g.wk_c

    def forward(self, x, c=None, orig_shape=None):
        assert c is not None, "Dual attention requires context tensor c"
        return _AttentionFn.apply(self, tuple(orig_shape), x, c, *self._param_list())
