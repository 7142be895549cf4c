"""FLUX.1-schnell VAE (`diffusers.AutoencoderKL`) on the MI355X kernels -- inference only, as the reference
uses it: frozen, bf16, `encode(x).latent_dist.sample()` on the data side (helpers/VAE_T5_CLIP.py:176-182,
helpers/VAE_T5_CLIP_inference.py:25-43) and `decode(z).sample` in the sampler (models/diff_model.py:467-477).

Mirror of the diffusers interface that those call sites touch: `.config.{latent_channels, scaling_factor, shift_factor}`,
`.dtype`, `.encode(x).latent_dist.sample()/.mode()`, `.decode(z).sample`, parameters under diffusers' state_dict keys
(a real `diffusion_pytorch_model.safetensors` loads with `load_state_dict`).  Every 3x3 convolution is an implicit GEMM
inside the LDS-DMA MFMA kernel (zero-bordered NHWC bf16 operand, bf16 operands, fp32 accumulate; conv_in with its 3 / 16
channels zero-padded to 64), GroupNorm+SiLU a statistics pass + an apply pass that writes the next convolution's operand,
the mid-block attention GEMM + row softmax + GEMM.  No CPU / PyTorch-math fallback."""
import math
from types import SimpleNamespace

import torch
from torch import nn

from . import _lib, ops

BF16, F32 = torch.bfloat16, torch.float32


def _pad8(n):
    return (n + 7) // 8 * 8


_H_BF16 = _lib.experiment("MMDIT_VAE_H_BF16", "1") != "0"     # A/B switch: conv1 outputs of the ResNet blocks in bf16
_CIN_PAD = 64     # conv_in operand channels (the implicit-GEMM K = 9 * C must be a multiple of the 64-wide K tile)


class _Conv(nn.Module):
    """nn.Conv2d parameter holder (weight (Cout, Cin, k, k), bias) + the packed bf16 GEMM operand [Cout_p][k*k*Cin_p]."""

    def __init__(self, cin, cout, k, cin_pad=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.empty(cout))
        self.cin, self.cout, self.k = cin, cout, k
        self.cin_pad = cin_pad      # channel count of the (zero-padded) operand: conv_in runs as an implicit GEMM on 64 channels
        self._packed = None

    def packed(self):
        key = (self.weight._version, self.bias._version, self.weight.device)
        if self._packed is None or self._packed[0] != key:
            cin_p, cout_p = self.cin_pad or _pad8(self.cin), _pad8(self.cout)
            w = torch.zeros((cout_p, self.k, self.k, cin_p), dtype=F32, device=self.weight.device)
            w[:self.cout, :, :, :self.cin] = self.weight.detach().permute(0, 2, 3, 1)
            b = torch.zeros(cout_p, dtype=F32, device=self.weight.device)
            b[:self.cout] = self.bias.detach()
            self._packed = (key, w.reshape(cout_p, -1).to(BF16).contiguous(), b)
        return self._packed[1], self._packed[2]


class _Norm(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(c))
        self.bias = nn.Parameter(torch.empty(c))


class _Linear(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(c, c))
        self.bias = nn.Parameter(torch.empty(c))


class _Resnet(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm1, self.conv1 = _Norm(cin), _Conv(cin, cout, 3)
        self.norm2, self.conv2 = _Norm(cout), _Conv(cout, cout, 3)
        if cin != cout:
            self.conv_shortcut = _Conv(cin, cout, 1)


class _Attn(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.group_norm = _Norm(c)
        self.to_q, self.to_k, self.to_v = _Linear(c), _Linear(c), _Linear(c)
        self.to_out = nn.ModuleList([_Linear(c)])


class _Mid(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.attentions = nn.ModuleList([_Attn(c)])
        self.resnets = nn.ModuleList([_Resnet(c, c), _Resnet(c, c)])


class _Sampler(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = _Conv(c, c, 3)


class _Down(nn.Module):
    def __init__(self, cin, cout, n, down):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(cin if j == 0 else cout, cout) for j in range(n)])
        if down:
            self.downsamplers = nn.ModuleList([_Sampler(cout)])


class _Up(nn.Module):
    def __init__(self, cin, cout, n, up):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(cin if j == 0 else cout, cout) for j in range(n)])
        if up:
            self.upsamplers = nn.ModuleList([_Sampler(cout)])


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        ch = cfg.block_out_channels
        self.conv_in = _Conv(cfg.in_channels, ch[0], 3, cin_pad=_CIN_PAD)
        self.down_blocks = nn.ModuleList([_Down(ch[max(i - 1, 0)], c, cfg.layers_per_block, i != len(ch) - 1) for i, c in enumerate(ch)])
        self.mid_block = _Mid(ch[-1])
        self.conv_norm_out = _Norm(ch[-1])
        self.conv_out = _Conv(ch[-1], 2 * cfg.latent_channels, 3)


class _Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        rev = list(reversed(cfg.block_out_channels))
        self.conv_in = _Conv(cfg.latent_channels, rev[0], 3, cin_pad=_CIN_PAD)
        self.mid_block = _Mid(rev[0])
        self.up_blocks = nn.ModuleList([_Up(rev[max(i - 1, 0)], c, cfg.layers_per_block + 1, i != len(rev) - 1) for i, c in enumerate(rev)])
        self.conv_norm_out = _Norm(rev[-1])
        self.conv_out = _Conv(rev[-1], cfg.out_channels, 3)


class _Act:
    """An activation on the device: fp32 rows (B*H*W, C) = NHWC pixels, plus its geometry."""

    def __init__(self, x, B, H, W):
        self.x, self.B, self.H, self.W = x, B, H, W


class DiagonalGaussianDistribution:
    """diffusers' class of the same name: moments (B, 2*latent, h, w) -> mean, clamped logvar; sample(), mode()."""

    def __init__(self, moments):
        self.mean, logvar = moments.chunk(2, dim=1)
        self.logvar = logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None):
        return self.mean + self.std * torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)

    def mode(self):
        return self.mean


class AutoencoderKL(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, latent_channels=16, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 norm_num_groups=32, scaling_factor=0.3611, shift_factor=0.1159, device="cuda"):
        super().__init__()
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels, latent_channels=latent_channels,
                                      block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                                      norm_num_groups=norm_num_groups, scaling_factor=scaling_factor, shift_factor=shift_factor,
                                      use_quant_conv=False, use_post_quant_conv=False, force_upcast=True)
        self.encoder = _Encoder(self.config)
        self.decoder = _Decoder(self.config)
        self.eps = 1e-6
        self._pad_cache = {}
        self.to(device)
        for p in self.parameters():
            p.requires_grad_(False)

    @property
    def dtype(self):
        return BF16     # what the reference casts the VAE to (VAE_T5_CLIP_inference.py:32); the sampler feeds `output.to(VAE.dtype)`

    # ---- building blocks ----------------------------------------------------------------------
    def _padded(self, B, H, W, C, dev):
        """Zero-bordered bf16 (B, H+2, W+2, C) operand buffer of the implicit-GEMM convolution.  One buffer per shape is
        recycled: producers only ever write the interior, so the border stays zero, and producer / consumer alternate on
        one stream."""
        key = (B, H, W, C, dev)
        buf = self._pad_cache.get(key)
        if buf is None:
            buf = self._pad_cache[key] = torch.zeros((B, H + 2, W + 2, C), dtype=BF16, device=dev)
        return buf

    def _gn(self, a, norm, silu, padded=False):
        out = self._padded(a.B, a.H, a.W, a.x.shape[1], a.x.device) if padded else None
        return ops.vae_groupnorm(a.x, norm.weight, norm.bias, a.B, a.H, a.W, self.config.norm_num_groups, self.eps, silu, out)

    def _conv3(self, xin, a, conv, mode=0, residual=None, out_dtype=F32):
        """3x3 convolution of the activation `a` whose conv operand is `xin`:
        a zero-bordered bf16 (B, H+2, W+2, Cin) tensor -> implicit GEMM (Cin % 64 == 0: every layer, conv_in through _conv_in);
        bf16 rows (B*H*W, Cin_p) -> materialised im2col + GEMM (kept for operands that are not channel-padded).
        mode 0: stride 1 / 1: Downsample2D / 2: Upsample2D (nearest x2 first).  Returns _Act with fp32 (B*Ho*Wo, Cout_p)."""
        w, b = conv.packed()
        if xin.dim() == 4:
            Hin, Win = xin.shape[1] - 2, xin.shape[2] - 2           # (already upsampled for mode 2)
            y = ops.gemm(xin, w, bias=b, residual=residual, out_dtype=out_dtype, conv=(2 if mode == 1 else 1, Hin, Win, xin.shape[3]))
            return _Act(y, a.B, Hin // 2 if mode == 1 else Hin, Win // 2 if mode == 1 else Win)
        cols, Ho, Wo = ops.vae_im2col3x3(xin.view(a.B, a.H, a.W, xin.shape[1]), mode)
        y = ops.gemm(cols, w, bias=b, residual=residual, out_dtype=F32)
        return _Act(y, a.B, Ho, Wo)

    def _conv_in(self, x, conv):
        """First convolution (3 image / 16 latent channels): NCHW -> NHWC bf16 with the channels zero-padded to 64, into the
        zero-bordered operand of the implicit GEMM (K = 9 x 64; the padded channels multiply zero weight columns)."""
        B, C, H, W = x.shape
        xb = ops.vae_nchw_to_nhwc(x.contiguous() if x.dtype in (F32, BF16) else x.float().contiguous(), _CIN_PAD)
        xin = ops.vae_pad_cast(xb.view(B * H * W, _CIN_PAD), B, H, W, self._padded(B, H, W, _CIN_PAD, x.device))
        return self._conv3(xin, _Act(None, B, H, W), conv)

    def _resample_operand(self, a, upsample):
        """fp32 activation -> zero-bordered bf16 conv operand (nearest x2 upsampled for Upsample2D)."""
        s = 2 if upsample else 1
        return ops.vae_pad_cast(a.x, a.B, a.H, a.W, self._padded(a.B, a.H * s, a.W * s, a.x.shape[1], a.x.device), upsample)

    def _resnet(self, a, r):
        # conv1's output only feeds norm2: bf16 (the reference runs the whole VAE in bf16, VAE_T5_CLIP.py:155-171) -- the bf16 epilogue of
        # the GEMM stores half the bytes and both GroupNorm passes read half; the residual stream between the blocks stays fp32
        h = self._conv3(self._gn(a, r.norm1, True, padded=True), a, r.conv1, out_dtype=BF16 if _H_BF16 else F32)
        sc = a.x
        if hasattr(r, "conv_shortcut"):
            w, b = r.conv_shortcut.packed()
            sc = ops.gemm(ops.cast(a.x, BF16), w, bias=b, out_dtype=F32)
        return self._conv3(self._gn(h, r.norm2, True, padded=True), h, r.conv2, residual=sc)

    def _attn(self, a, at):
        B, HW, C = a.B, a.H * a.W, a.x.shape[1]
        h = self._gn(a, at.group_norm, False)
        wb = lambda l: (l.weight.detach().to(BF16), l.bias.detach())
        (wq, bq), (wk, bk), (wv, bv), (wo, bo) = wb(at.to_q), wb(at.to_k), wb(at.to_v), wb(at.to_out[0])
        q = ops.gemm(h, wq, bias=bq, out_dtype=BF16)
        k = ops.gemm(h, wk, bias=bk, out_dtype=BF16)
        v = ops.gemm(h, wv, bias=bv, out_dtype=BF16)
        o = torch.empty((B * HW, C), dtype=BF16, device=h.device)
        HWp = _pad8(HW)         # GEMM operands need 16-byte rows: keys / values are zero-padded to a multiple of 8 tokens
        kp, vp = torch.zeros((HWp, C), dtype=BF16, device=h.device), torch.zeros((HWp, C), dtype=BF16, device=h.device)
        for i in range(B):      # one head of width C: plain GEMMs per image
            kp[:HW].copy_(k[i * HW:(i + 1) * HW])
            vp[:HW].copy_(v[i * HW:(i + 1) * HW])
            s = ops.gemm(q[i * HW:(i + 1) * HW], kp, out_dtype=F32)                 # (HW, HWp)
            p = ops.vae_softmax_rows(s, 1.0 / math.sqrt(C), cols=HW)               # padding columns = 0
            ops.gemm(p, vp, b_kmajor=True, out=o[i * HW:(i + 1) * HW])
        return _Act(ops.gemm(o, wo, bias=bo, residual=a.x, out_dtype=F32), a.B, a.H, a.W)

    def _mid(self, a, m):
        a = self._resnet(a, m.resnets[0])
        a = self._attn(a, m.attentions[0])
        return self._resnet(a, m.resnets[1])

    # ---- public interface -----------------------------------------------------------------------
    @torch.no_grad()
    def encode(self, x):
        """x: (B, 3, H, W) image in [-1, 1] (any float dtype), H and W multiples of 8."""
        if not x.is_cuda:
            raise RuntimeError("the VAE runs on the HIP kernels only (no CPU fallback)")
        B, C, H, W = x.shape
        e = self.encoder
        a = self._conv_in(x, e.conv_in)
        for blk in e.down_blocks:
            for r in blk.resnets:
                a = self._resnet(a, r)
            if hasattr(blk, "downsamplers"):
                a = self._conv3(self._resample_operand(a, False), a, blk.downsamplers[0].conv, mode=1)
        a = self._mid(a, e.mid_block)
        a = self._conv3(self._gn(a, e.conv_norm_out, True, padded=True), a, e.conv_out)
        moments = ops.vae_nhwc_to_nchw(a.x, B, 2 * self.config.latent_channels, a.H, a.W)
        return SimpleNamespace(latent_dist=DiagonalGaussianDistribution(moments))

    @torch.no_grad()
    def decode(self, z):
        """z: (B, latent, h, w) in VAE space (the caller applied (z - shift_factor) / scaling_factor)."""
        if not z.is_cuda:
            raise RuntimeError("the VAE runs on the HIP kernels only (no CPU fallback)")
        B, C, H, W = z.shape
        d = self.decoder
        a = self._conv_in(z, d.conv_in)
        a = self._mid(a, d.mid_block)
        for blk in d.up_blocks:
            for r in blk.resnets:
                a = self._resnet(a, r)
            if hasattr(blk, "upsamplers"):
                a = self._conv3(self._resample_operand(a, True), a, blk.upsamplers[0].conv, mode=2)
        a = self._conv3(self._gn(a, d.conv_norm_out, True, padded=True), a, d.conv_out)
        return SimpleNamespace(sample=ops.vae_nhwc_to_nchw(a.x, B, self.config.out_channels, a.H, a.W))
