"""Cross-check of oracle/vae_oracle.py against an INDEPENDENT published implementation of the same autoencoder.  TEST INFRASTRUCTURE ONLY.

diffusers (the reference's dependency for the FLUX VAE) is not installed here, but HF `transformers` is -- here and on the GPU box -- and
its `modeling_janus.py` carries an independent implementation of the latent-diffusion ("taming") encoder / decoder that diffusers'
`AutoencoderKL` was converted from: `JanusVQVAEEncoder` / `JanusVQVAEDecoder` = conv_in, per level `num_res_blocks` (+1 in the decoder)
ResnetBlocks [GroupNorm(32, eps 1e-6) -> x*sigmoid(x) -> 3x3 conv, twice, 1x1 `nin_shortcut` on a channel change], stride-2 3x3
downsampling behind a (0,1,0,1) zero pad / nearest x2 upsampling + 3x3 conv, a mid block ResnetBlock - single-head AttnBlock
(1x1-conv q/k/v, softmax(q k / sqrt(C)), 1x1 proj_out, residual) - ResnetBlock, GroupNorm -> swish -> conv_out with `double_latent`
(mean | logvar).  Built with the FLUX geometry (base 128, multipliers (1,2,4,4), 2 res blocks, 16 latent channels, double latent) and
with the extra attention blocks Janus puts into its last resolution level removed, it is the architecture diffusers' Encoder / Decoder
run for the FLUX config; diffusers' state_dict keys map onto it by pure renaming (the inverse of diffusers' own LDM-checkpoint
conversion: down_blocks.i.resnets.j <-> down.i.block.j, conv_shortcut <-> nin_shortcut, mid_block.resnets.{0,1} <-> mid.block_{1,2},
attentions.0.{group_norm,to_q,to_k,to_v,to_out.0} <-> attn_1.{norm,q,k,v,proj_out} with (C,C) <-> (C,C,1,1), downsamplers.0.conv <->
downsample.conv, up_blocks.i <-> up.i in Janus' lowest-resolution-first list, conv_norm_out <-> norm_out).

What this pins: the block arithmetic and the assembly of the restatement (and hence of the HIP path tested against it) equal an
implementation written by someone else.  What it cannot pin: that diffusers 0.30.3 itself computes this for the FLUX config (the
statement above is from its published source, not checked here), and the pretrained weights.
"""
import torch
from torch import nn

from . import vae_oracle as V


def _janus(cfg: V.VAEConfig):
    from transformers.models.janus.configuration_janus import JanusVQVAEConfig
    from transformers.models.janus.modeling_janus import JanusVQVAEDecoder, JanusVQVAEEncoder
    ch = cfg.block_out_channels
    jc = JanusVQVAEConfig(double_latent=True, latent_channels=cfg.latent_channels, in_channels=cfg.in_channels, out_channels=cfg.out_channels,
                          base_channels=ch[0], channel_multiplier=tuple(c // ch[0] for c in ch), num_res_blocks=cfg.layers_per_block, dropout=0.0)
    enc, dec = JanusVQVAEEncoder(jc).eval(), JanusVQVAEDecoder(jc).eval()
    enc.down[-1].attn = nn.ModuleList()      # FLUX / SD autoencoders attend in the mid block only
    dec.up[0].attn = nn.ModuleList()
    return enc, dec


def _rename(k: str) -> str:
    side, rest = k.split(".", 1)
    r = rest
    r = r.replace("conv_norm_out", "norm_out")
    r = r.replace("mid_block.resnets.0", "mid.block_1").replace("mid_block.resnets.1", "mid.block_2")
    r = r.replace("mid_block.attentions.0.group_norm", "mid.attn_1.norm")
    for a, b in (("to_q", "q"), ("to_k", "k"), ("to_v", "v"), ("to_out.0", "proj_out")):
        r = r.replace(f"mid_block.attentions.0.{a}", f"mid.attn_1.{b}")
    r = r.replace("down_blocks.", "down.").replace("up_blocks.", "up.").replace(".resnets.", ".block.")
    r = r.replace("downsamplers.0.conv", "downsample.conv").replace("upsamplers.0.conv", "upsample.conv")
    r = r.replace("conv_shortcut", "nin_shortcut")
    return r


def load_into_janus(sd, cfg: V.VAEConfig = V.VAEConfig()):
    """Janus encoder / decoder modules holding the diffusers-keyed state dict `sd`."""
    enc, dec = _janus(cfg)
    for mod, side in ((enc, "encoder."), (dec, "decoder.")):
        want = mod.state_dict()
        got = {}
        for k, v in sd.items():
            if not k.startswith(side):
                continue
            nk = _rename(k)
            if v.dim() == 2 and want[nk].dim() == 4:      # Linear (C, C) <-> 1x1 conv (C, C, 1, 1)
                v = v[:, :, None, None]
            got[nk] = v
        missing = set(want) - set(got)
        assert not missing and set(got) == set(want), (sorted(missing)[:5], sorted(set(got) - set(want))[:5])
        mod.load_state_dict(got, strict=True)
    return enc, dec


@torch.no_grad()
def crosscheck(seed=0, hw=(32, 48), cfg: V.VAEConfig = V.VAEConfig()):
    """rel-L2 of (encoder moments, decoder output) between the restatement and the Janus modules on seeded weights and inputs."""
    sd = V.make_state_dict(seed, cfg)
    enc, dec = load_into_janus(sd, cfg)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.rand((2, cfg.in_channels) + tuple(hw), generator=g) * 2 - 1
    z = torch.randn((2, cfg.latent_channels, hw[0] // 8, hw[1] // 8), generator=g)
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    return rel(V.encode_moments(x, sd, cfg), enc(x.clone())), rel(V.decode(z, sd, cfg), dec(z.clone()))
