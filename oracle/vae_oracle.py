"""CPU oracle for SURVEY row V: the FLUX.1-schnell VAE (`diffusers.AutoencoderKL`, pin diffusers==0.30.3,
reference requirements.txt:15) that the reference calls at helpers/VAE_T5_CLIP.py:176-182,
helpers/VAE_T5_CLIP_inference.py:25-43 (encode -> latent_dist.sample() * scaling_factor + shift_factor) and
models/diff_model.py:467-477 (decode((z - shift_factor) / scaling_factor).sample.clamp(-1, 1)).

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg may import it).

**Parity against diffusers itself: UNPINNED** -- diffusers is not installed in the build container, there is no network and the
reference holds no test or golden vector for the VAE.  **Architecture pinned against an independent implementation**: HF transformers
(installed) carries its own implementation of the latent-diffusion encoder / decoder that `AutoencoderKL` was converted from
(modeling_janus.py); with the FLUX geometry and this file's diffusers-keyed seeded weights loaded through a pure key renaming it agrees
with this restatement to 1e-6 (oracle/vae_crosscheck.py, tests/test_vae.py).  The file restates the published algorithm of
`AutoencoderKL` / `Encoder` / `Decoder` / `ResnetBlock2D` / `Downsample2D` /
`Upsample2D` / `Attention` (diffusers 0.30.3, models/autoencoders/vae.py, models/resnet.py, models/downsampling.py,
models/upsampling.py, models/attention_processor.py) for the FLUX VAE config:
  in/out_channels 3, latent_channels 16, block_out_channels (128, 256, 512, 512), layers_per_block 2, norm_num_groups 32,
  act_fn silu, mid_block_add_attention, use_quant_conv False, use_post_quant_conv False, scaling_factor 0.3611,
  shift_factor 0.1159, force_upcast True.
state_dict keys and shapes follow diffusers' module tree so that a real checkpoint would load (`state_dict_spec`).
"""
import math
from dataclasses import dataclass, field
from typing import Tuple

import torch
import torch.nn.functional as F


@dataclass
class VAEConfig:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 16
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.3611
    shift_factor: float = 0.1159
    eps: float = 1e-6


# ---------------------------------------------------------------------------------------------
# state_dict layout (diffusers module tree)
# ---------------------------------------------------------------------------------------------
def _resnet_spec(prefix, cin, cout, out):
    out[f"{prefix}.norm1.weight"] = (cin,)
    out[f"{prefix}.norm1.bias"] = (cin,)
    out[f"{prefix}.conv1.weight"] = (cout, cin, 3, 3)
    out[f"{prefix}.conv1.bias"] = (cout,)
    out[f"{prefix}.norm2.weight"] = (cout,)
    out[f"{prefix}.norm2.bias"] = (cout,)
    out[f"{prefix}.conv2.weight"] = (cout, cout, 3, 3)
    out[f"{prefix}.conv2.bias"] = (cout,)
    if cin != cout:
        out[f"{prefix}.conv_shortcut.weight"] = (cout, cin, 1, 1)
        out[f"{prefix}.conv_shortcut.bias"] = (cout,)


def _mid_spec(prefix, c, out):
    _resnet_spec(f"{prefix}.resnets.0", c, c, out)
    a = f"{prefix}.attentions.0"
    out[f"{a}.group_norm.weight"] = (c,)
    out[f"{a}.group_norm.bias"] = (c,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        out[f"{a}.{n}.weight"] = (c, c)
        out[f"{a}.{n}.bias"] = (c,)
    _resnet_spec(f"{prefix}.resnets.1", c, c, out)


def state_dict_spec(cfg: VAEConfig = VAEConfig()):
    """Ordered {key: shape} of AutoencoderKL(**FLUX config).state_dict()."""
    out = {}
    ch = cfg.block_out_channels
    # encoder (vae.py Encoder)
    out["encoder.conv_in.weight"] = (ch[0], cfg.in_channels, 3, 3)
    out["encoder.conv_in.bias"] = (ch[0],)
    cin = ch[0]
    for i, cout in enumerate(ch):
        for j in range(cfg.layers_per_block):
            _resnet_spec(f"encoder.down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout, out)
        if i != len(ch) - 1:
            out[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            out[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (cout,)
        cin = cout
    _mid_spec("encoder.mid_block", ch[-1], out)
    out["encoder.conv_norm_out.weight"] = (ch[-1],)
    out["encoder.conv_norm_out.bias"] = (ch[-1],)
    out["encoder.conv_out.weight"] = (2 * cfg.latent_channels, ch[-1], 3, 3)
    out["encoder.conv_out.bias"] = (2 * cfg.latent_channels,)
    # decoder (vae.py Decoder)
    rev = list(reversed(ch))
    out["decoder.conv_in.weight"] = (rev[0], cfg.latent_channels, 3, 3)
    out["decoder.conv_in.bias"] = (rev[0],)
    _mid_spec("decoder.mid_block", rev[0], out)
    cin = rev[0]
    for i, cout in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            _resnet_spec(f"decoder.up_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout, out)
        if i != len(rev) - 1:
            out[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            out[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (cout,)
        cin = cout
    out["decoder.conv_norm_out.weight"] = (rev[-1],)
    out["decoder.conv_norm_out.bias"] = (rev[-1],)
    out["decoder.conv_out.weight"] = (cfg.out_channels, rev[-1], 3, 3)
    out["decoder.conv_out.bias"] = (cfg.out_channels,)
    return out


def make_state_dict(seed: int, cfg: VAEConfig = VAEConfig()):
    """Seeded random weights with sane magnitudes (fan-in scaled convs, norm weights near 1)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shape in state_dict_spec(cfg).items():
        if ".norm" in k or "group_norm" in k or "conv_norm_out" in k:
            sd[k] = (1.0 + 0.1 * torch.randn(shape, generator=g)) if k.endswith("weight") else 0.1 * torch.randn(shape, generator=g)
        elif k.endswith("bias"):
            sd[k] = 0.05 * torch.randn(shape, generator=g)
        else:
            fan_in = math.prod(shape[1:])
            sd[k] = torch.randn(shape, generator=g) / math.sqrt(fan_in)
    return sd


# ---------------------------------------------------------------------------------------------
# modules (functional)
# ---------------------------------------------------------------------------------------------
def _gn(x, sd, p, cfg):
    return F.group_norm(x, cfg.norm_num_groups, sd[f"{p}.weight"], sd[f"{p}.bias"], eps=cfg.eps)


def resnet(x, sd, p, cfg):
    """ResnetBlock2D.forward (resnet.py): norm1 -> silu -> conv1 -> norm2 -> silu -> (dropout 0) -> conv2; + shortcut; / 1."""
    h = F.conv2d(F.silu(_gn(x, sd, f"{p}.norm1", cfg)), sd[f"{p}.conv1.weight"], sd[f"{p}.conv1.bias"], padding=1)
    h = F.conv2d(F.silu(_gn(h, sd, f"{p}.norm2", cfg)), sd[f"{p}.conv2.weight"], sd[f"{p}.conv2.bias"], padding=1)
    if f"{p}.conv_shortcut.weight" in sd:
        x = F.conv2d(x, sd[f"{p}.conv_shortcut.weight"], sd[f"{p}.conv_shortcut.bias"])
    return x + h


def attention(x, sd, p, cfg):
    """Attention with AttnProcessor2_0 as configured by UNetMidBlock2D (heads = 1, residual_connection, spatial group norm)."""
    B, C, H, W = x.shape
    h = _gn(x, sd, f"{p}.group_norm", cfg).view(B, C, H * W).transpose(1, 2)          # (B, HW, C)
    q = F.linear(h, sd[f"{p}.to_q.weight"], sd[f"{p}.to_q.bias"])
    k = F.linear(h, sd[f"{p}.to_k.weight"], sd[f"{p}.to_k.bias"])
    v = F.linear(h, sd[f"{p}.to_v.weight"], sd[f"{p}.to_v.bias"])
    a = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(C), dim=-1) @ v
    a = F.linear(a, sd[f"{p}.to_out.0.weight"], sd[f"{p}.to_out.0.bias"])
    return x + a.transpose(1, 2).reshape(B, C, H, W)


def mid_block(x, sd, p, cfg):
    x = resnet(x, sd, f"{p}.resnets.0", cfg)
    x = attention(x, sd, f"{p}.attentions.0", cfg)
    return resnet(x, sd, f"{p}.resnets.1", cfg)


def encode_moments(x, sd, cfg: VAEConfig = VAEConfig()):
    """Encoder.forward: image (B,3,H,W) in [-1,1] -> moments (B, 2*latent, H/8, W/8)."""
    n = len(cfg.block_out_channels)
    h = F.conv2d(x, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    for i in range(n):
        for j in range(cfg.layers_per_block):
            h = resnet(h, sd, f"encoder.down_blocks.{i}.resnets.{j}", cfg)
        if i != n - 1:   # Downsample2D(padding=0): pad (0,1,0,1) then stride-2 conv
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"],
                         sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"], stride=2)
    h = mid_block(h, sd, "encoder.mid_block", cfg)
    h = F.silu(_gn(h, sd, "encoder.conv_norm_out", cfg))
    return F.conv2d(h, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)


def sample_latent(moments, noise, cfg: VAEConfig = VAEConfig()):
    """DiagonalGaussianDistribution.sample() followed by the reference's normalisation (VAE_T5_CLIP_inference.py:41)."""
    mean, logvar = moments.chunk(2, dim=1)
    std = torch.exp(0.5 * logvar.clamp(-30.0, 20.0))
    return (mean + std * noise) * cfg.scaling_factor + cfg.shift_factor


def decode(z, sd, cfg: VAEConfig = VAEConfig()):
    """Decoder.forward on the VAE-space latent (the caller has already applied (z - shift) / scale, diff_model.py:467)."""
    rev = list(reversed(cfg.block_out_channels))
    h = F.conv2d(z, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    h = mid_block(h, sd, "decoder.mid_block", cfg)
    for i in range(len(rev)):
        for j in range(cfg.layers_per_block + 1):
            h = resnet(h, sd, f"decoder.up_blocks.{i}.resnets.{j}", cfg)
        if i != len(rev) - 1:   # Upsample2D: nearest x2, then conv
            h = F.conv2d(F.interpolate(h, scale_factor=2.0, mode="nearest"), sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"],
                         sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
    h = F.silu(_gn(h, sd, "decoder.conv_norm_out", cfg))
    return F.conv2d(h, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)
