"""Explicit forward / backward kernel schedules of the MMDiT path.

Instead of composing many small autograd ops, each stage below is a fixed sequence of launches of
the HIP kernels in libmmdit_hip.so with hand-written backward schedules, so that residual adds,
gating, bias and modulation gradients are fused into the kernels that already touch the data.
PyTorch only owns the buffers.  Reference call stack being replaced: diff_model.forward
(models/diff_model.py:264-346) -> Transformer_Block_Dual.forward (blocks/Transformer_Block_Dual.py:56-77)
-> Attention.forward (blocks/Attention.py:118-427) / MLP.forward (blocks/MLP.py:25-40) / Norm.forward
(blocks/Norm.py:16-22), and their autograd mirrors.

Precision modes:
  fast   : bf16 activations / bf16 MFMA operands, fp32 accumulation, fp32 residual stream, flash attention.
  parity : fp32 activations, split-bf16 (3-pass) MFMA GEMMs, attention core that reproduces the rounding
           points of the reference's CPU branch (Attention.py:277-284).  Used for the 1e-3 golden check.
"""
from types import SimpleNamespace as NS

import torch

from . import ops
from ._lib import ACT_SILU, PREC_BF16, PREC_SPLIT

F32 = torch.float32
BF16 = torch.bfloat16


class Mode:
    def __init__(self, fast: bool = True):
        self.fast = fast
        self.T = BF16 if fast else F32
        self.prec = PREC_BF16 if fast else PREC_SPLIT
        self.attn_mode = 0 if fast else 1

    def act(self, t):
        """Tensor as a GEMM A operand in this mode's activation dtype."""
        return t if t.dtype == self.T else ops.cast(t, self.T)


FAST = Mode(True)
PARITY = Mode(False)

MOD_NAMES_FULL = ["shift1x", "scale1x", "gate1x", "shift2x", "scale2x", "gate2x", "shift1c", "scale1c", "gate1c", "shift2c", "scale2c", "gate2c"]


def _mod_views(mod, d, last):
    names = MOD_NAMES_FULL[:8] if last else MOD_NAMES_FULL
    return NS(**{n: mod[:, i * d:(i + 1) * d] for i, n in enumerate(names)})


def _gemm(m, A, B, **kw):
    return ops.gemm(A, B, precision=m.prec, **kw)


def _wgrad(m, dY, X):
    """dW[N,K] = dY[M,N]^T X[M,K] (fp32)."""
    return ops.gemm(dY, X, a_kmajor=True, b_kmajor=True, out_dtype=F32, precision=m.prec)


def _dgrad(m, dY, W, out_dtype, **kw):
    """dX[M,K] = dY[M,N] W[N,K]."""
    return ops.gemm(dY, W, b_kmajor=True, out_dtype=out_dtype, precision=m.prec, **kw)


# ----------------------------------------------------------------------------------------------
# MLP stage (one stream):  Y = X + gate * (W_down act(W_up LNmod(X) + b_up) + b_down)
# ----------------------------------------------------------------------------------------------
def mlp_core_fwd(m, w, x_act):
    """w: NS(Wup, bup, Wdown, bdown, hidden, gelu).  Returns (pre-activation, hidden activation)."""
    gu = _gemm(m, x_act, w.Wup, bias=w.bup, out_dtype=m.T)
    h = ops.mlp_act_fwd(gu, w.hidden, w.gelu)
    return gu, h


def mlp_core_bwd(m, w, dacc, x_act, gu, h, dev):
    """dacc: grad of the down-projection output (T).  Returns (dx_act, grads)."""
    dh = _dgrad(m, dacc, w.Wdown, m.T)
    gWdown = _wgrad(m, dacc, h)
    dbup = torch.zeros(gu.shape[1], dtype=F32, device=dev)
    dgu = ops.mlp_act_bwd(dh, gu, w.hidden, dbup, w.gelu)
    dx = _dgrad(m, dgu, w.Wup, m.T)
    gWup = _wgrad(m, dgu, x_act)
    return dx, NS(Wup=gWup, bup=dbup, Wdown=gWdown)


# ----------------------------------------------------------------------------------------------
# Transformer_Block_Dual
# ----------------------------------------------------------------------------------------------
def block_fwd(m, w, X, C, y, dims, rope):
    """X (B*N,d) fp32, C (B*M,d) fp32, y (B,d) in m.T.  Returns X2, C2, saved."""
    B, N, Mt, H, d = dims
    S, dev = N + Mt, X.device
    sv = NS(X=X, C=C, y=y)
    sv.pre = torch.empty((B, d), dtype=F32, device=dev)
    sv.yp = _gemm(m, y, w.Wy, bias=w.by, act=ACT_SILU, aux=sv.pre, out_dtype=m.T)
    sv.mod = _gemm(m, sv.yp, w.Wmod, out_dtype=F32)
    ms = _mod_views(sv.mod, d, w.last)

    sv.ln1x, sv.mu1x, sv.rs1x = ops.ln_modulate_fwd(X, ms.scale1x, ms.shift1x, N, m.T)
    sv.ln1c, sv.mu1c, sv.rs1c = ops.ln_modulate_fwd(C, ms.scale1c, ms.shift1c, Mt, m.T)
    sv.qkv_x = _gemm(m, sv.ln1x, w.Wqkv_x, out_dtype=m.T)
    sv.qkv_c = _gemm(m, sv.ln1c, w.Wqkv_c, out_dtype=m.T)
    sv.Q = torch.empty((B, H, S, 64), dtype=BF16, device=dev)
    sv.K, sv.V = torch.empty_like(sv.Q), torch.empty_like(sv.Q)
    ops.qk_norm_rope_fwd(sv.qkv_x, w.wq_x, w.wk_x, rope[0], rope[1], B, N, H, S, 0, sv.Q, sv.K, sv.V)
    ops.qk_norm_rope_fwd(sv.qkv_c, w.wq_c, w.wk_c, None, None, B, Mt, H, S, N, sv.Q, sv.K, sv.V)
    sv.Ox, sv.Oc, sv.lse = ops.attn_fwd(sv.Q, sv.K, sv.V, N, 64 ** -0.5, m.attn_mode)

    sv.Oxa = m.act(sv.Ox.view(B * N, d))
    sv.acc_ox = torch.empty((B * N, d), dtype=m.T, device=dev)
    X1 = _gemm(m, sv.Oxa, w.Wo_x, gate=ms.gate1x, rows_per_batch=N, residual=X, aux=sv.acc_ox, out_dtype=F32)
    if not w.last:
        sv.Oca = m.act(sv.Oc.view(B * Mt, d))
        sv.acc_oc = torch.empty((B * Mt, d), dtype=m.T, device=dev)
        C1 = _gemm(m, sv.Oca, w.Wo_c, gate=ms.gate1c, rows_per_batch=Mt, residual=C, aux=sv.acc_oc, out_dtype=F32)
    else:
        C1 = C
    sv.X1, sv.C1 = X1, C1

    sv.ln2x, sv.mu2x, sv.rs2x = ops.ln_modulate_fwd(X1, ms.scale2x, ms.shift2x, N, m.T)
    sv.gu_x, sv.h_x = mlp_core_fwd(m, w.mlp_x, sv.ln2x)
    sv.acc_mx = torch.empty((B * N, d), dtype=m.T, device=dev)
    X2 = _gemm(m, sv.h_x, w.mlp_x.Wdown, bias=w.mlp_x.bdown, gate=ms.gate2x, rows_per_batch=N, residual=X1, aux=sv.acc_mx, out_dtype=F32)
    if not w.last:
        sv.ln2c, sv.mu2c, sv.rs2c = ops.ln_modulate_fwd(C1, ms.scale2c, ms.shift2c, Mt, m.T)
        sv.gu_c, sv.h_c = mlp_core_fwd(m, w.mlp_c, sv.ln2c)
        sv.acc_mc = torch.empty((B * Mt, d), dtype=m.T, device=dev)
        C2 = _gemm(m, sv.h_c, w.mlp_c.Wdown, bias=w.mlp_c.bdown, gate=ms.gate2c, rows_per_batch=Mt, residual=C1, aux=sv.acc_mc, out_dtype=F32)
    else:
        C2 = C1
    return X2, C2, sv


def block_bwd(m, w, sv, dX2, dC2, dy_acc, dims, rope):
    """dX2 (B*N,d) fp32, dC2 (B*M,d) fp32 or None, dy_acc (B,d) fp32 or None.
    Returns dX, dC, dy_acc', grads (NS keyed like the packed weights)."""
    B, N, Mt, H, d = dims
    S, dev = N + Mt, dX2.device
    ms = _mod_views(sv.mod, d, w.last)
    dmod = torch.zeros_like(sv.mod)
    dms = _mod_views(dmod, d, w.last)
    g = NS()

    def mlp_stream(wm, dY, acc, gate, dgate, ln2, gu, h, X1, mu, rs, scale, dscale, dshift, rpb):
        db_down = torch.zeros(d, dtype=F32, device=dev)
        dacc = ops.gate_residual_bwd(dY, acc, gate, rpb, dgate, db_down, m.T)
        dln2, gm = mlp_core_bwd(m, wm, dacc, ln2, gu, h, dev)
        gm.bdown = db_down
        dX1 = ops.ln_modulate_bwd(dln2, X1, mu, rs, scale, dY, rpb, dscale, dshift)
        return dX1, gm

    dX1, g.mlp_x = mlp_stream(w.mlp_x, dX2, sv.acc_mx, ms.gate2x, dms.gate2x, sv.ln2x, sv.gu_x, sv.h_x, sv.X1, sv.mu2x, sv.rs2x,
                              ms.scale2x, dms.scale2x, dms.shift2x, N)
    if not w.last:
        dC1, g.mlp_c = mlp_stream(w.mlp_c, dC2, sv.acc_mc, ms.gate2c, dms.gate2c, sv.ln2c, sv.gu_c, sv.h_c, sv.C1, sv.mu2c, sv.rs2c,
                                  ms.scale2c, dms.scale2c, dms.shift2c, Mt)
    else:
        dC1 = dC2

    # attention output projections (gated residual)
    dacc = ops.gate_residual_bwd(dX1, sv.acc_ox, ms.gate1x, N, dms.gate1x, None, m.T)
    dOx = _dgrad(m, dacc, w.Wo_x, BF16)
    g.Wo_x = _wgrad(m, dacc, sv.Oxa)
    if not w.last:
        dacc_c = ops.gate_residual_bwd(dC1, sv.acc_oc, ms.gate1c, Mt, dms.gate1c, None, m.T)
        dOc = _dgrad(m, dacc_c, w.Wo_c, BF16)
        g.Wo_c = _wgrad(m, dacc_c, sv.Oca)
    else:
        dOc = None  # the last block's text attention output is discarded (Attention.py:425)

    # attention core + QK norm / RoPE
    dQ, dK, dV = ops.attn_bwd(sv.Q, sv.K, sv.V, sv.Ox, sv.Oc, dOx, dOc, sv.lse, N, 64 ** -0.5, m.T)
    g.wq_x, g.wk_x = torch.zeros(64, dtype=F32, device=dev), torch.zeros(64, dtype=F32, device=dev)
    g.wq_c, g.wk_c = torch.zeros(64, dtype=F32, device=dev), torch.zeros(64, dtype=F32, device=dev)
    dqkv_x = ops.qk_norm_rope_bwd(dQ, dK, dV, sv.qkv_x, w.wq_x, w.wk_x, rope[0], rope[1], B, N, H, S, 0, g.wq_x, g.wk_x, m.T)
    dqkv_c = ops.qk_norm_rope_bwd(dQ, dK, dV, sv.qkv_c, w.wq_c, w.wk_c, None, None, B, Mt, H, S, N, g.wq_c, g.wk_c, m.T)
    dln1x = _dgrad(m, dqkv_x, w.Wqkv_x, m.T)
    g.Wqkv_x = _wgrad(m, dqkv_x, sv.ln1x)
    dln1c = _dgrad(m, dqkv_c, w.Wqkv_c, m.T)
    g.Wqkv_c = _wgrad(m, dqkv_c, sv.ln1c)
    dX = ops.ln_modulate_bwd(dln1x, sv.X, sv.mu1x, sv.rs1x, ms.scale1x, dX1, N, dms.scale1x, dms.shift1x)
    dC = ops.ln_modulate_bwd(dln1c, sv.C, sv.mu1c, sv.rs1c, ms.scale1c, dC1, Mt, dms.scale1c, dms.shift1c)

    # modulation vectors and y_proj
    dmod_a = m.act(dmod)
    dyp = _dgrad(m, dmod_a, w.Wmod, F32)
    g.Wmod = _wgrad(m, dmod_a, sv.yp)
    g.by = torch.zeros(d, dtype=F32, device=dev)
    dpre = ops.silu_bwd(dyp, sv.pre, m.T, g.by)
    dy_acc = _dgrad(m, dpre, w.Wy, F32, residual=dy_acc)
    g.Wy = _wgrad(m, dpre, sv.y)
    return dX, dC, dy_acc, g


# ----------------------------------------------------------------------------------------------
# whole model
# ----------------------------------------------------------------------------------------------
def model_fwd(m, W, x_t, t, c, c_pooled, rope):
    """W: packed weights (see models/diff_model.py: WeightCache).  x_t (B,Cin,H,W); t (B,) fp32;
    c (B,tokens,2304); c_pooled (B,class_dim).  Returns v (B,Cin,H,W) fp32 and the saved state."""
    B, Cin, Hh, Ww = x_t.shape
    d, H = W.dim, W.heads
    N, Mt = (Hh // 2) * (Ww // 2), c.shape[1]
    dims = (B, N, Mt, H, d)
    sv = NS(dims=dims, img=(B, Cin, Hh, Ww), c=c, t=t)
    sv.pe = ops.time_embed_fwd(t, W.time_scale, W.denom, m.T)
    temb = _gemm(m, sv.pe, W.Wt, out_dtype=F32)
    sv.cp = m.act(c_pooled)
    sv.y = _gemm(m, sv.cp, W.Wcond, residual=temb, out_dtype=m.T)

    sv.cn1, sv.cn2 = ops.text_rmsnorm_fwd(c, W.wn1, W.wn2, W.s1, W.s2, W.split, m.T)
    c1 = _gemm(m, sv.cn1, W.Wc1, out_dtype=F32)
    c2 = _gemm(m, sv.cn2, W.Wc2, out_dtype=F32)
    C = torch.cat([c1.view(B, W.split, d), c2.view(B, Mt - W.split, d)], dim=1).view(B * Mt, d)

    sv.patches = ops.patchify(x_t, m.T)
    sv.X0 = _gemm(m, sv.patches, W.Wpatch, out_dtype=m.T)
    X = _gemm(m, sv.X0, W.Wpe, bias=W.bpe, out_dtype=F32)

    sv.blocks = []
    for wb in W.blocks:
        X, C, bs = block_fwd(m, wb, X, C, sv.y, dims, rope)
        sv.blocks.append(bs)

    sv.Xf = X
    sv.modo = _gemm(m, sv.y, W.Wmod_out, out_dtype=F32)  # [shift | scale]
    sv.lnf, sv.muf, sv.rsf = ops.ln_modulate_fwd(X, sv.modo[:, d:], sv.modo[:, :d], N, m.T)
    Z = _gemm(m, sv.lnf, W.Wout, bias=W.bout, out_dtype=F32)
    v = ops.unpatchify(Z, B, Cin, Hh, Ww, F32)
    return v, sv


def _grad_tensors(g):
    out = []
    for v in vars(g).values():
        if isinstance(v, NS):
            out += _grad_tensors(v)
        elif torch.is_tensor(v):
            out.append(v)
    return out


def model_bwd(m, W, sv, dv, rope, on_grads=None):
    """Returns grads: NS with top-level packed-weight grads and .blocks = [block grads].
    on_grads(list_of_tensors) is called as soon as a group of parameter gradients is final (block by
    block, last block first) so a data-parallel reducer can overlap its collective with the rest."""
    B, N, Mt, H, d = sv.dims
    _, Cin, Hh, Ww = sv.img
    dev = dv.device
    g = NS()
    dZ = ops.patchify(dv.contiguous(), m.T)
    g.bout = torch.zeros(W.Wout.shape[0], dtype=F32, device=dev)
    ops.colsum(dZ, g.bout)
    dlnf = _dgrad(m, dZ, W.Wout, m.T)
    g.Wout = _wgrad(m, dZ, sv.lnf)
    dmodo = torch.zeros_like(sv.modo)
    dX = ops.ln_modulate_bwd(dlnf, sv.Xf, sv.muf, sv.rsf, sv.modo[:, d:], None, N, dmodo[:, d:], dmodo[:, :d])
    dmodo_a = m.act(dmodo)
    dy_acc = _dgrad(m, dmodo_a, W.Wmod_out, F32)
    g.Wmod_out = _wgrad(m, dmodo_a, sv.y)

    dC = None
    g.blocks = [None] * len(W.blocks)
    for i in range(len(W.blocks) - 1, -1, -1):
        dX, dC, dy_acc, g.blocks[i] = block_bwd(m, W.blocks[i], sv.blocks[i], dX, dC, dy_acc, sv.dims, rope)
        sv.blocks[i] = None  # free saved activations as we go
        if on_grads is not None:
            on_grads(_grad_tensors(g.blocks[i]))

    # patch embedding
    g.bpe = torch.zeros(d, dtype=F32, device=dev)
    ops.colsum(dX, g.bpe)
    dX_a = m.act(dX)
    dX0 = _dgrad(m, dX_a, W.Wpe, m.T)
    g.Wpe = _wgrad(m, dX_a, sv.X0)
    g.Wpatch = _wgrad(m, dX0, sv.patches)

    # text embedding
    dCv = dC.view(B, Mt, d)
    dc1 = m.act(dCv[:, :W.split].reshape(B * W.split, d))
    dc2 = m.act(dCv[:, W.split:].reshape(B * (Mt - W.split), d))
    dcn1 = _dgrad(m, dc1, W.Wc1, m.T)
    dcn2 = _dgrad(m, dc2, W.Wc2, m.T)
    g.Wc1 = _wgrad(m, dc1, sv.cn1)
    g.Wc2 = _wgrad(m, dc2, sv.cn2)
    g.wn1, g.wn2, g.s1, g.s2 = ops.text_rmsnorm_bwd(dcn1, dcn2, sv.c, W.wn1, W.wn2, W.s1, W.s2, W.split)

    # conditioning vector
    dy_a = m.act(dy_acc)
    g.Wcond = _wgrad(m, dy_a, sv.cp)
    dpe = _dgrad(m, dy_a, W.Wt, m.T)
    g.Wt = _wgrad(m, dy_a, sv.pe)
    g.time_scale = ops.time_embed_bwd(dpe, sv.t, W.time_scale, W.denom)
    if on_grads is not None:
        on_grads([v for v in vars(g).values() if torch.is_tensor(v)])
    return g
