"""The VAE half of the reference's `VAE_T5_CLIP_inference` / `VAE_T5_CLIP` holders (helpers/VAE_T5_CLIP_inference.py:19-43,
helpers/VAE_T5_CLIP.py:150-182): owns the frozen bf16 FLUX VAE and `forward_VAE_and_sample`.  The text encoders of those
classes (Gemma-2-2b, ModernBERT, MetaCLIP: pretrained third-party models) are out of this build's scope -- assign any
object with `text_to_embedding(text)` to `.text_encoder` to use `diff_model.sample_imgs`."""
import torch

from ..vae import AutoencoderKL


class VAE_inference:
    def __init__(self, device, state_dict=None):
        self.device = device
        self.VAE = AutoencoderKL(device=device).eval()            # reference: AutoencoderKL.from_pretrained("black-forest-labs/FLUX.1-schnell", subfolder="vae")
        if state_dict is not None:
            self.VAE.load_state_dict(state_dict, strict=True)
        self.VAE_downsample = 8
        self.text_encoder = None

    def forward_VAE_and_sample(self, x, generator=None):
        """image (B,3,H,W) in [-1,1] -> normalised latent (VAE_T5_CLIP_inference.py:37-41): encode, sample, * scaling_factor + shift_factor."""
        z = self.VAE.encode(x).latent_dist.sample(generator=generator)
        return z * self.VAE.config.scaling_factor + self.VAE.config.shift_factor

    @torch.no_grad()
    def text_to_embedding(self, text):
        if self.text_encoder is None:
            raise RuntimeError("text encoders (Gemma-2-2b / ModernBERT / MetaCLIP) are out of this build's scope: assign .text_encoder")
        return self.text_encoder.text_to_embedding(text)
