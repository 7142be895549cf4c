"""CPU ORACLE for the MMDiT flow-matching hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain torch-CPU ops, the algorithm of the reference
(gmongaras/Stable-Diffusion-3-From-Scratch @ /root/reference) for the one hot
path this repo accelerates.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it; the product path (the package + libmmdit_hip.so)
never does and fails loudly when the HIP library is missing.

Parity pin: every function here is checked against outputs of the reference
itself, imported in the build container by tools/make_goldens.py (three
sys.modules stubs) and committed as fixtures under tests/golden/.
The xformers SwiGLU boundary (un-vendored third party, xformers==0.0.29.post3,
reference README.md:69) is restated from its documented eager semantics and is
"parity unpinned" at that boundary only; the FLUX VAE (diffusers==0.30.3) is not
restated here at all.

Reference citations (file:line are into /root/reference/src):
  forward            models/diff_model.py:264-346
  block              blocks/Transformer_Block_Dual.py:56-77
  attention          blocks/Attention.py:118-135, 174-194, 258-284, 410-425
  rope               blocks/rotary_embedding.py:36-76, 269-321
  mlp                blocks/MLP.py:25-40 (+ xformers SwiGLU eager semantics)
  norm               blocks/Norm.py:16-22
  time embedding     blocks/PositionalEncoding.py:15-30
  patch embed        blocks/ImagePositionalEncoding.py:114-116, 175-187
  unpatchify         blocks/patchify.py:41-72
  noise / loss / opt model_trainer.py:378-503, models/diff_model.py:229-241
  sampler            models/diff_model.py:367-429

Rounding modes (OracleConfig):
  attn_core = "oracle_bf16": the reference CPU branch, Attention.py:277-284
              (QK^T rounded to bf16, scaled in bf16, softmax in bf16, PV in bf16)
            = "fp32":        exact softmax attention
            = "flash_bf16":  the HIP fast path's rounding points (bf16 Q/K/V,
              fp32 scores, P=exp(s-m) rounded to bf16, fp32 row sum, O/l -> bf16)
  gemm      = "fp32":  reference arithmetic (fp32 weights and activations)
            = "bf16":  operands (weights and GEMM inputs) rounded to bf16 at the
              same points as the HIP fast path, fp32 accumulation; intermediate
              GEMM outputs that the fast path stores as bf16 are rounded too.
            = "fp8":   "bf16", and the four big GEMMs of every block (QKV, attention
              out-projection, MLP up / down; both streams) take OCP e4m3 operands with
              per-tensor scales exactly as the HIP fp8 inference mode does on the first
              call of a site: s = amax / 448, q = e4m3_rne(clamp(x / s)), the packed
              weight ([Wq; Wk; Wv], w12, ...) shares ONE scale, C = s_a s_b (A_q B_q^T).
              This is test infrastructure for BASELINE config 5; the reference itself
              has no fp8 path.
            = "mxfp8": the same four GEMM sites with MX (OCP microscaling) operands: every 32
              consecutive K values share the smallest power-of-two scale with amax / scale <= 448, values
              e4m3_rne(clamp(x / scale)) -- the HIP "mxfp8" inference mode (mmdit_mxfp8_quantize,
              mmdit_gemm_args.scale_mode 1).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional

import torch
import torch.nn.functional as F

FP32_EPS = torch.finfo(torch.float32).eps  # nn.RMSNorm(eps=None) on fp32 input


@dataclass
class OracleConfig:
    dim: int = 256
    num_heads: int = 4
    num_blocks: int = 2
    hidden_scale: float = 4.0
    patch_size: int = 2
    inCh: int = 16
    class_dim: int = 768
    MLP_type: str = "swiglu"
    text_hidden: int = 2304
    attn_core: str = "oracle_bf16"
    gemm: str = "fp32"
    # arithmetic dtype: float32 = the reference; float64 (with a float64 state dict and inputs) = the same algorithm and the same
    # bf16 rounding points of the attention core in exact arithmetic -- the yardstick that separates the reference's own fp32
    # rounding noise from an implementation's (tools/probes/parity_budget.py).  Tables the reference computes in fp32 (time-embedding
    # denominators, RoPE angles and their cos / sin) keep their fp32 values.
    dtype: torch.dtype = torch.float32
    # backward of the bf16 mode: the HIP fast path stores every gradient that feeds a GEMM (dY of each Linear, the data gradients the
    # GEMMs write) in bf16, as the reference's autocast backward does.  grad_round=True rounds the gradient to bf16 at exactly those
    # points (the outputs of every Linear and every stored activation) when autograd runs through this restatement; False (default)
    # leaves the backward in the arithmetic dtype (straight-through roundings), which is what the committed goldens were made with.
    grad_round: bool = False

    @property
    def head_dim(self):
        return self.dim // self.num_heads

    @property
    def hidden(self):
        return int(self.dim * self.hidden_scale)


# ----------------------------------------------------------------------------
# rounding helpers
# ----------------------------------------------------------------------------
def _rb(x: torch.Tensor) -> torch.Tensor:
    """Round to bf16 and back, straight-through for autograd."""
    return x + (x.to(torch.bfloat16).to(x.dtype) - x).detach()


class _RoundGrad(torch.autograd.Function):
    """Identity in forward; the incoming gradient is rounded to bf16 in backward (a gradient the HIP path stores as bf16)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def _rg(cfg: "OracleConfig", x):
    return _RoundGrad.apply(x) if cfg.grad_round and cfg.gemm != "fp32" and x.requires_grad else x


def _lin(cfg: OracleConfig, x, w, b=None):
    """nn.Linear; in gemm=bf16 / fp8 mode both operands are rounded to bf16 first."""
    if cfg.gemm in ("bf16", "fp8", "mxfp8"):
        x = _rb(x)
        w = _rb(w)
    return _rg(cfg, F.linear(x, w, b))


def _q8(x: torch.Tensor):
    """Per-tensor OCP e4m3 quantisation as csrc/rowops.hip fp8_quant_kernel does it: returns (values of the e4m3 codes as
    fp32, dequantisation scale amax/448)."""
    a = x.detach().abs().amax().clamp_min(1e-12)
    q = (x * (448.0 / a)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32)
    return q, a / 448.0


def _qmx(x: torch.Tensor):
    """MX (OCP microscaling) e4m3 quantisation along the last dimension as csrc/rowops.hip mxfp8_quant_kernel does it: blocks of 32
    values share the smallest power-of-two scale with amax / scale <= 448 (2^(floor(log2 amax) - 8) or twice that; 2^-127 for an all-zero block); returns the DEQUANTISED values
    (e4m3 code x scale, exact in fp32), which is what the block-scaled matrix instruction multiplies."""
    shp = x.shape
    xb = x.detach().to(torch.float32).reshape(-1, shp[-1] // 32, 32)
    amax = xb.abs().amax(-1, keepdim=True)
    _, ex = torch.frexp(amax)                                     # amax = m 2^ex, m in [0.5, 1): floor(log2(amax)) = ex - 1
    e = torch.where(amax > 0, ex - 1 - 8, torch.full_like(ex, -127))
    e = torch.where(amax > 448.0 * torch.ldexp(torch.ones_like(amax), e), e + 1, e).clamp(-127, 127)     # amax / scale <= 448: nothing saturates
    scale = torch.ldexp(torch.ones_like(amax), e)
    q = (xb / scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32)
    return (q * scale).reshape(shp)


def _lin8(cfg: OracleConfig, x, ws, b=None):
    """One of the fp8 GEMM sites of a block: `ws` are the weights the HIP path packs row-wise into ONE operand (so they share
    one e4m3 scale); returns the list of outputs.  Outside fp8 mode (or K % 128 != 0, which the HIP path keeps in bf16) these
    are plain _lin calls."""
    if cfg.gemm not in ("fp8", "mxfp8") or x.shape[-1] % 128:
        assert b is None or len(ws) == 1
        return [_lin(cfg, x, w, b) for w in ws]      # (bias inside F.linear: the reference's op, bit for bit)
    if cfg.gemm == "mxfp8":     # block scales: no shared per-tensor scale, the packing of the weights does not matter
        xq = _qmx(_rb(x))
        outs = [F.linear(xq, _qmx(_rb(w))) for w in ws]
        if b is not None:
            outs[0] = outs[0] + b
        return outs
    xq, sa = _q8(_rb(x))
    wq, sb = _q8(_rb(torch.cat(list(ws), dim=0)))
    out = F.linear(xq, wq) * (sa * sb)
    if b is not None:
        out = out + b
    return list(out.split([w.shape[0] for w in ws], dim=-1))


def _act(cfg: OracleConfig, x):
    """Activation tensor that the HIP fast path stores as bf16 (and, with grad_round, whose gradient it stores as bf16)."""
    return _rg(cfg, _rb(x)) if cfg.gemm in ("bf16", "fp8", "mxfp8") else x


# ----------------------------------------------------------------------------
# leaf functions
# ----------------------------------------------------------------------------
def positional_encoding(time: torch.Tensor, dim: int) -> torch.Tensor:
    """PositionalEncoding.forward (blocks/PositionalEncoding.py:15-30).
    denom_i = 10000^(2i/dim) for i = 0..dim-1 (not dim/2); output is
    cat(sin(e[:, 0::2]), cos(e[:, 1::2]))."""
    denom = (torch.tensor(10000.0) ** ((2 * torch.arange(dim)) / dim)).to(torch.float32)
    e = time[:, None] / denom[None, :].to(time.dtype)
    return torch.cat((e[:, ::2].sin(), e[:, 1::2].cos()), dim=1)


def rms_norm(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """nn.RMSNorm(dim) with eps=None on fp32 input -> eps = finfo(fp32).eps."""
    return F.rms_norm(x, (x.shape[-1],), w, FP32_EPS)


def norm_modulate(x, y, w_scale, w_shift, cfg: OracleConfig):
    """Norm.forward (blocks/Norm.py:16-22): LN(no affine, eps 1e-5)*(1+W_s y)+W_h y."""
    xn = F.layer_norm(x, (x.shape[-1],))
    return xn * (1 + _lin(cfg, y, w_scale)[:, None, :]) + _lin(cfg, y, w_shift)[:, None, :]


def rope_inv_freq(head_dim: int) -> torch.Tensor:
    """RotaryEmbedding(dim=head_dim//2) 'lang' frequencies
    (blocks/rotary_embedding.py:120; Attention.py:98)."""
    d = head_dim // 2
    return 1.0 / (10000 ** (torch.arange(0, d, 2)[: d // 2].float() / d))


def axial_freqs(height: int, width: int, inv_freq: torch.Tensor) -> torch.Tensor:
    """get_axial_freqs(height, width) (blocks/rotary_embedding.py:269-288):
    (h, w, head_dim) angles; first half <- row index, second half <- column
    index, each frequency repeated twice (interleaved pairs)."""
    fh = torch.arange(height).float()[:, None] * inv_freq[None, :]
    fw = torch.arange(width).float()[:, None] * inv_freq[None, :]
    fh = fh.repeat_interleave(2, dim=-1)[:, None, :].expand(height, width, -1)
    fw = fw.repeat_interleave(2, dim=-1)[None, :, :].expand(height, width, -1)
    return torch.cat([fh, fw], dim=-1)


def rotate_half(x):
    """(x0,x1,x2,x3,..) -> (-x1,x0,-x3,x2,..) (blocks/rotary_embedding.py:36-40)."""
    x = x.reshape(*x.shape[:-1], -1, 2)
    x1, x2 = x.unbind(dim=-1)
    return torch.stack((-x2, x1), dim=-1).reshape(*x.shape[:-2], -1)


def apply_rope(freqs, t):
    """apply_rotary_emb (blocks/rotary_embedding.py:43-76), full-width rotation."""
    return t * freqs.cos().to(t.dtype) + rotate_half(t) * freqs.sin().to(t.dtype)


def attention_core(q, k, v, scale: float, mode: str):
    """softmax(q k^T * scale) v on (B,H,S,hd) tensors.
    oracle_bf16 reproduces the op sequence of Attention.py:277-284."""
    if mode == "oracle_bf16":
        attn = (q.to(torch.bfloat16) @ k.to(torch.bfloat16).mT) * scale
        attn = attn.softmax(dim=-1)
        return (attn @ v.to(torch.bfloat16)).to(q.dtype)
    if mode == "fp32":
        attn = ((q @ k.mT) * scale).softmax(dim=-1)
        return attn @ v
    if mode == "flash_bf16":
        qb, kb, vb = _rb(q), _rb(k), _rb(v)
        s = (qb @ kb.mT) * scale
        m = s.amax(dim=-1, keepdim=True)
        p = torch.exp(s - m)
        l = p.sum(dim=-1, keepdim=True)
        o = (_rb(p) @ vb) / l
        return _rb(o)
    if mode == "flash_bf16_tiled":
        # the same rounding points as "flash_bf16", placed as a TILED online softmax places them (csrc/attention.hip attn_fwd_dma_kernel:
        # 64-key tiles; the running maximum of a query row moves tile by tile; p = exp2(s c - m_running) is rounded to bf16 for the P V
        # product while the row sum adds the unrounded p; accumulator and sum are rescaled by exp2(m_old - m_new)).  Rounding p relative
        # to the running instead of the final maximum is not the same rounding (exp(m_run - m_final) is no power of two), which is the
        # whole difference to "flash_bf16": ~1e-3 of the output where the attention is peaked, nothing where it is diffuse.
        qb, kb, vb = _rb(q), _rb(k), _rb(v)
        c2 = scale * 1.4426950408889634
        S = k.shape[-2]
        m = torch.full(q.shape[:-1] + (1,), float("-inf"), dtype=q.dtype)
        l = torch.zeros_like(m)
        o = torch.zeros_like(q)
        for j in range(0, S, 64):
            sj = (qb @ kb[..., j:j + 64, :].mT) * c2
            mn = torch.maximum(m, sj.amax(dim=-1, keepdim=True))
            alpha = torch.exp2(m - mn)
            p = torch.exp2(sj - mn)
            l = l * alpha + p.sum(dim=-1, keepdim=True)
            o = o * alpha + _rb(p) @ vb[..., j:j + 64, :]
            m = mn
        return _rb(o / l)
    raise ValueError(mode)


def patch_embed(x_t, w_patch, cfg: OracleConfig):
    """PatchEmbed.forward with pos_embed=None: Conv2d(k=p, stride=p, no bias),
    flatten(2).transpose(1,2)  (blocks/ImagePositionalEncoding.py:114-116, 181-183)."""
    if cfg.gemm in ("bf16", "fp8", "mxfp8"):
        x_t, w_patch = _rb(x_t), _rb(w_patch)
    y = F.conv2d(x_t, w_patch, None, stride=cfg.patch_size)
    return y.flatten(2).transpose(1, 2)


def unpatchify(patches, patch_size: int, hw):
    """blocks/patchify.py:41-72; patch vector order is (C, ph, pw)."""
    n, _, pd = patches.shape
    h, w = hw
    nh, nw = (h + patch_size - 1) // patch_size, (w + patch_size - 1) // patch_size
    ch = pd // (patch_size * patch_size)
    x = patches.view(n, nh, nw, ch, patch_size, patch_size).permute(0, 3, 1, 4, 2, 5).contiguous()
    return x.view(n, ch, nh * patch_size, nw * patch_size)[:, :, :h, :w]


def mlp(x, sd, prefix: str, cfg: OracleConfig):
    """MLP.forward (blocks/MLP.py:25-40).  swiglu = xformers SwiGLU eager
    semantics: w3(silu(x W1^T + b1) * (x W2^T + b2)) + b3, [W1;W2] = w12."""
    if cfg.MLP_type == "swiglu":
        gu = _act(cfg, _lin8(cfg, x, [sd[prefix + "MLP.w12.weight"]], sd[prefix + "MLP.w12.bias"])[0])
        g, u = gu.chunk(2, dim=-1)
        h = _act(cfg, F.silu(g) * u)
        return _lin8(cfg, h, [sd[prefix + "MLP.w3.weight"]], sd[prefix + "MLP.w3.bias"])[0]
    if cfg.MLP_type == "gelu":
        u = _act(cfg, _lin(cfg, x, sd[prefix + "lin_up.weight"], sd[prefix + "lin_up.bias"]))
        h = _act(cfg, F.gelu(u))
        return _lin(cfg, h, sd[prefix + "lin_down.weight"], sd[prefix + "lin_down.bias"])
    raise ValueError(cfg.MLP_type)


def attention(x, c, sd, prefix: str, cfg: OracleConfig, hw, last: bool, taps: Optional[dict] = None):
    """Attention.forward, dual / softmax / RoPE2d branch (blocks/Attention.py)."""
    B, N, d = x.shape
    M = c.shape[1]
    H, hd = cfg.num_heads, cfg.head_dim

    def heads(t, L):
        return t.reshape(B, L, H, hd).permute(0, 2, 1, 3)

    qkv_x = _lin8(cfg, x, [sd[prefix + n + "_proj_x.weight"] for n in ("query", "key", "value")])
    qkv_c = _lin8(cfg, c, [sd[prefix + n + "_proj_c.weight"] for n in ("query", "key", "value")])
    q_x = rms_norm(heads(_act(cfg, qkv_x[0]), N), sd[prefix + "q_norm_x.weight"])
    k_x = rms_norm(heads(_act(cfg, qkv_x[1]), N), sd[prefix + "k_norm_x.weight"])
    v_x = heads(_act(cfg, qkv_x[2]), N)
    q_c = rms_norm(heads(_act(cfg, qkv_c[0]), M), sd[prefix + "q_norm_c.weight"])
    k_c = rms_norm(heads(_act(cfg, qkv_c[1]), M), sd[prefix + "k_norm_c.weight"])
    v_c = heads(_act(cfg, qkv_c[2]), M)

    # RoPE2d on image tokens only; patch size hard-coded to 2 (Attention.py:178-179)
    h2, w2 = hw[0] // 2, hw[1] // 2
    freqs = axial_freqs(h2, w2, sd[prefix + "rotary_emb.freqs"])
    q_x = apply_rope(freqs, q_x.reshape(B, H, h2, w2, hd)).reshape(B, H, -1, hd)
    k_x = apply_rope(freqs, k_x.reshape(B, H, h2, w2, hd)).reshape(B, H, -1, hd)

    q = torch.cat([q_x, q_c], dim=2)
    k = torch.cat([k_x, k_c], dim=2)
    v = torch.cat([v_x, v_c], dim=2)
    if taps is not None:
        taps["q"], taps["k"], taps["v"] = q.detach(), k.detach(), v.detach()
    o = _rg(cfg, attention_core(q, k, v, hd ** -0.5, cfg.attn_core))
    if taps is not None:
        taps["attn_core"] = o.detach()
    o_x = o[:, :, :N].permute(0, 2, 1, 3).reshape(B, N, -1)
    o_c = o[:, :, N:].permute(0, 2, 1, 3).reshape(B, M, -1)
    a_x = _lin8(cfg, o_x, [sd[prefix + "out_proj_x.weight"]])[0]
    a_c = o_c if last else _lin8(cfg, o_c, [sd[prefix + "out_proj_c.weight"]])[0]
    return a_x, a_c


def block(X, c, y, sd, i: int, cfg: OracleConfig, hw, taps: Optional[dict] = None):
    """Transformer_Block_Dual.forward (blocks/Transformer_Block_Dual.py:56-77)."""
    p = f"blocks.{i}."
    last = i == cfg.num_blocks - 1
    y = F.silu(_lin(cfg, y, sd[p + "y_proj.0.weight"], sd[p + "y_proj.0.bias"]))

    def nrm(Z, name):
        return _act(cfg, norm_modulate(Z, y, sd[p + name + ".c_scale.weight"], sd[p + name + ".c_shift.weight"], cfg))

    def gate(name):
        return _lin(cfg, y, sd[p + name + ".weight"])[:, None, :]

    n1x, n1c = nrm(X, "norm1_x"), nrm(c, "norm1_c")
    a_x, a_c = attention(n1x, n1c, sd, p + "attn.", cfg, hw, last, taps)
    if taps is not None:
        taps.update(y_proj=y.detach(), norm1_x=n1x.detach(), norm1_c=n1c.detach(), attn_x=a_x.detach(), attn_c=a_c.detach())
    # (_act: in the bf16 / fp8 rounding-matched modes the HIP fast path stores the projection outputs in bf16 and forms the gated
    #  residual update from the stored value, in the consumer's adaLN kernel; identity in the reference-exact fp32 mode)
    X = _act(cfg, a_x) * gate("scale1_x") + X
    if not last:
        c = _act(cfg, a_c) * gate("scale1_c") + c
    m_x = mlp(nrm(X, "norm2_x"), sd, p + "MLP_x.", cfg)
    if taps is not None:
        taps["mlp_x"] = m_x.detach()
    X = _act(cfg, m_x) * gate("scale2_x") + X
    if not last:
        c = _act(cfg, mlp(nrm(c, "norm2_c"), sd, p + "MLP_c.", cfg)) * gate("scale2_c") + c
    return X, c


def forward(sd: Dict[str, torch.Tensor], cfg: OracleConfig, x_t, t, c, c_pooled,
            nullCls_pooled=None, nullCls_gemma=None, nullCls_bert=None,
            taps: Optional[dict] = None):
    """diff_model.forward (models/diff_model.py:264-346).  Like the reference it
    zeroes the caller's c / c_pooled rows IN PLACE for null-masked samples."""
    with torch.no_grad():
        if nullCls_pooled is not None:
            c_pooled[nullCls_pooled] *= 0
        if nullCls_gemma is not None:
            c[nullCls_gemma, :77] *= 0
        if nullCls_bert is not None:
            c[nullCls_bert, 77:] *= 0
    d = cfg.dim
    pe = positional_encoding(t.to(cfg.dtype) * sd["time_scale"], d)
    t_emb = _lin(cfg, pe, sd["t_emb2.weight"])
    y = t_emb + _lin(cfg, c_pooled.to(cfg.dtype), sd["cond_MLP.weight"])
    hw = x_t.shape[-2:]
    cf = c.to(cfg.dtype)
    ctx = torch.cat([
        _lin(cfg, sd["learnable_scalar"] * rms_norm(cf[:, :77], sd["pre_c_norm.weight"]), sd["c_proj.weight"]),
        _lin(cfg, sd["learnable_scalar2"] * rms_norm(cf[:, 77:], sd["pre_c_norm2.weight"]), sd["c_proj2.weight"]),
    ], dim=1)
    X = patch_embed(x_t.to(cfg.dtype), sd["pos_enc.proj.weight"], cfg)
    X = _lin(cfg, X, sd["patch_emb.weight"], sd["patch_emb.bias"])
    if taps is not None:
        taps.update(y=y.detach(), c0=ctx.detach(), x0=X.detach(), blocks=[])
    for i in range(cfg.num_blocks):
        bt = {} if (taps is not None and i == 0) else None
        X, ctx = block(X, ctx, y, sd, i, cfg, hw, bt)
        if taps is not None:
            taps["blocks"].append((X.detach(), ctx.detach()))
            if bt is not None:
                taps["block0"] = bt
    Z = _act(cfg, norm_modulate(X, y, sd["out_norm.c_scale.weight"], sd["out_norm.c_shift.weight"], cfg))
    Z = _lin(cfg, Z, sd["out_proj.weight"], sd["out_proj.bias"])
    return unpatchify(Z, cfg.patch_size, hw)


# ----------------------------------------------------------------------------
# training step (model_trainer.py:378-503) and sampler (diff_model.py:367-429)
# ----------------------------------------------------------------------------
def noise_batch(x0, t, eps):
    """diff_model.noise_batch with the noise passed in (diff_model.py:229-241)."""
    tt = t[:, None, None, None]
    return (1 - tt) * x0 + tt * eps


def rectified_flow_loss(v_pred, x0, eps):
    """model_trainer.py:429-446: MSE(v_pred, eps - x0), per-sample flatten, global mean."""
    return F.mse_loss(v_pred, (eps - x0), reduction="none").flatten(1, -1).mean()


def warmup_lr(step: int, warmup_steps: int) -> float:
    """HF get_constant_schedule_with_warmup lambda (model_trainer.py:25-41)."""
    return 1.0 if step >= warmup_steps else float(step) / float(max(1.0, warmup_steps))


class OracleTrainer:
    """fwd + bwd + clip(1.0) + AdamW(lr, (0.9,0.999), 1e-8, wd 0.01) + constant-with-warmup
    schedule, restating model_trainer.py:260-263, 463-503 on the oracle forward."""

    def __init__(self, sd, cfg: OracleConfig, lr=1e-4, warmup_steps=0):
        self.cfg = cfg
        self.sd = {k: v.clone().requires_grad_(k.split(".")[-1] != "freqs") for k, v in sd.items()}
        self.params = [v for v in self.sd.values() if v.requires_grad]
        self.optim = torch.optim.AdamW(self.params, lr=lr, eps=1e-8, weight_decay=0.01, betas=(0.9, 0.999))
        self.base_lr, self.warmup_steps, self.step_idx = lr, warmup_steps, 0
        for g in self.optim.param_groups:
            g["lr"] = lr * warmup_lr(0, warmup_steps)

    def step(self, x0, eps, t, c, c_pooled, nulls=(None, None, None)):
        x_t = noise_batch(x0, t, eps)
        v = forward(self.sd, self.cfg, x_t, t, c, c_pooled, *nulls)
        loss = rectified_flow_loss(v, x0, eps)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(self.params, 1.0)
        self.optim.step()
        self.step_idx += 1
        for g in self.optim.param_groups:
            g["lr"] = self.base_lr * warmup_lr(self.step_idx, self.warmup_steps)
        self.optim.zero_grad()
        return loss.detach()


@torch.no_grad()
def cfg_sample(sd, cfg: OracleConfig, noise, text_hidden, text_pooled, num_steps: int, cfg_scale: float, sampler: str = "euler", generator=None):
    """sample_imgs (diff_model.py:367-460) up to (not including) the VAE decode: "euler" (430-432), "euler_stochastic" (434-449:
    per-step noise drawn on the CPU from `generator`, sigma = t(1-t)/(1-t+0.008), scaled by sqrt(dt)) and "heun" (451-462: second
    velocity at t - dt on the Euler prediction, trapezoidal update)."""
    B = noise.shape[0]
    out = noise.clone()
    null = torch.tensor([0] * B + [1] * B).bool()
    th = text_hidden.repeat(2 * B, 1, 1)
    tp = text_pooled.repeat(2 * B, 1)
    dt = 1 / num_steps

    def velocity(x, tt):
        v = forward(sd, cfg, x.repeat(2, 1, 1, 1), tt, th, tp, null, null, null)
        return (1 + cfg_scale) * v[:B] - cfg_scale * v[B:]

    for t in torch.linspace(1, 1.0 / num_steps, num_steps):
        tt = t.repeat(2 * B)
        v = velocity(out, tt)
        if sampler == "euler":
            out = out - v * dt
        elif sampler == "euler_stochastic":
            sigma = (tt * (1 - tt) / (1 - tt + 0.008))[:B, None, None, None]
            out = out - v * dt + sigma * torch.randn(v.shape, generator=generator) * (dt ** 0.5)
        elif sampler == "heun":
            v2 = velocity(out - v * dt, tt - dt)
            out = out - (dt / 2) * (v + v2)
        else:
            raise ValueError("Invalid sampler specified. Choose 'euler', 'euler_stochastic', or 'heun'.")
    return out


def euler_cfg_sample(sd, cfg: OracleConfig, noise, text_hidden, text_pooled, num_steps: int, cfg_scale: float):
    """sample_imgs, sampler='euler' (diff_model.py:384-432), up to (not including) VAE decode."""
    return cfg_sample(sd, cfg, noise, text_hidden, text_pooled, num_steps, cfg_scale, "euler")
