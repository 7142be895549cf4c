"""On-GPU data preparation inside the training rank: raw images -> FLUX-VAE latents on the same GPU that
trains, instead of the reference's dedicated loader GPUs (helpers/VAE_T5_CLIP.py:176-182 runs the VAE on 2 of 8 GPUs and
ships padded latents to the model ranks, model_trainer.py:353-370).  Use as `model_trainer(..., data_source=ImageLatentSource(...))`.

`image_source()` must return (images (B,3,H,W) in [-1,1] on the device, text (B,154,2304) bf16, pooled (B,768) bf16): the
text side (Gemma / ModernBERT / MetaCLIP embeddings) is produced elsewhere -- those encoders are out of this build's scope."""
import torch

from .VAE_inference import VAE_inference


class ImageLatentSource:
    def __init__(self, image_source, vae: VAE_inference, generator=None):
        self.image_source, self.vae, self.generator = image_source, vae, generator

    @torch.no_grad()
    def __call__(self):
        images, text, pooled = self.image_source()
        if images.shape[-1] % 16 or images.shape[-2] % 16:
            raise RuntimeError("image sides must be multiples of 16 (8x VAE downsampling, 2x2 patches)")
        latents = self.vae.forward_VAE_and_sample(images, generator=self.generator)      # (B,16,H/8,W/8), normalised
        return latents.to(torch.bfloat16), text, pooled                                  # the trainer's wire format (model_trainer.py:353-355)
