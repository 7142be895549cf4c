"""PatchEmbed: mirror of the reference's src/blocks/ImagePositionalEncoding.py PatchEmbed
(ctor 78-141, forward 175-187) for the trained configuration: pos_embed disabled (RoPE2d), no
layer norm, flatten=True, bias=False.  The stride-p conv is executed as gather + MFMA GEMM."""
import torch
from torch import nn

from .. import engine, ops
from ..packing import Pack


class _PatchEmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, latent, weight):
        m = mod._mode()
        B, C, H, W = latent.shape
        patches = ops.patchify(latent.contiguous(), m.T)
        out = ops.gemm(patches, mod._pack.get(m), out_dtype=torch.float32, precision=m.prec)
        ctx.m, ctx.mod = m, mod
        ctx.save_for_backward(patches)
        return out.view(B, (H // 2) * (W // 2), -1)

    @staticmethod
    def backward(ctx, dout):
        (patches,) = ctx.saved_tensors
        m = ctx.m
        d2 = m.act(dout.reshape(-1, dout.shape[-1]).contiguous())
        gW = ops.gemm(d2, patches, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, precision=m.prec)
        return None, None, gW.view(ctx.mod.proj.weight.shape)


class PatchEmbed(nn.Module):
    def __init__(self, height=224, width=224, patch_size=16, in_channels=3, embed_dim=768, layer_norm=False, flatten=True,
                 bias=True, interpolation_scale=1, pos_embed_type="sincos", pos_embed_max_size=None):
        super().__init__()
        if patch_size != 2:
            raise RuntimeError("PatchEmbed: the HIP path implements the reference's patch_size=2 configuration")
        if layer_norm or bias or not flatten:
            raise RuntimeError("PatchEmbed: only layer_norm=False, bias=False, flatten=True (the reference's call, diff_model.py:193-205) is implemented")
        if pos_embed_type in ("absolute", "sincos"):
            raise RuntimeError("PatchEmbed: absolute sin-cos position tables are not part of the trained configuration (RoPE2d)")
        self.patch_size = patch_size
        self.flatten, self.layer_norm = flatten, layer_norm
        self.height, self.width = height // patch_size, width // patch_size
        self.proj = nn.Conv2d(in_channels, embed_dim, kernel_size=(patch_size, patch_size), stride=patch_size, bias=False)
        self.pos_embed = None
        self._pack = Pack([self.proj.weight])
        self.precision = "fast"

    def _mode(self):
        return engine.FAST if self.precision == "fast" else engine.PARITY

    def forward(self, latent):
        if latent.shape[-1] % 2 or latent.shape[-2] % 2:
            latent = latent[..., : latent.shape[-2] // 2 * 2, : latent.shape[-1] // 2 * 2]  # the conv silently floors
        return _PatchEmbedFn.apply(self, latent, self.proj.weight)
