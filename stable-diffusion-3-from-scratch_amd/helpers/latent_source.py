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

    @classmethod
    def synthetic(cls, batch, res, class_dim, device, text_len=154, text_dim=2304, seed=0):
        """Fresh U(-1,1) images (batch, 3, res, res) and random text embeddings from a device generator every call, encoded by a FLUX-shaped
        VAE with its constructor's random weights (the pretrained ones are not on the box): the data path of BASELINE.json configs[3]
        for throughput runs (train.py --vae-in-rank, tools/config4_bench.py)."""
        device = torch.device(device)
        g = torch.Generator(device=device).manual_seed(seed)

        def images():
            x = torch.rand((batch, 3, res, res), generator=g, device=device) * 2 - 1
            text = torch.randn((batch, text_len, text_dim), generator=g, device=device).to(torch.bfloat16)
            pooled = torch.randn((batch, class_dim), generator=g, device=device).to(torch.bfloat16)
            return x, text, pooled

        vae = VAE_inference(device)
        # the VAE's constructor leaves its parameters uninitialised (the reference always loads the pretrained file): seeded fan-in
        # scaled weights, unit norm scales, zero biases -- activations stay O(1) through the encoder
        gw = torch.Generator(device="cpu").manual_seed(seed + 2)
        with torch.no_grad():
            for name, prm in vae.VAE.named_parameters():
                if prm.dim() >= 2:
                    fan_in = prm[0].numel()
                    prm.copy_((torch.randn(prm.shape, generator=gw) / fan_in ** 0.5).to(prm.device))
                elif name.endswith("weight"):
                    prm.fill_(1.0)
                else:
                    prm.zero_()
        return cls(images, vae, generator=torch.Generator(device=device).manual_seed(seed + 1))
