"""The reference's latent wire format between loader and model ranks: a batch of one aspect-ratio bucket,
(B, C, h, w) bf16 latents, is padded at the bottom / right with +inf to the fixed (max_res/8, max_res/8) so that every message
has one shape (helpers/VAE_T5_CLIP.py:438), and the receiver recovers (h, w) by counting the inf entries of the first channel
of the first sample and gathers the finite values (model_trainer.py:362-370).

`pad_latents` / `unpad_latents` are that format.  The unpad reads the two extents back with ONE host transfer (the reference
calls .item() twice and then runs a boolean-mask gather, i.e. a nonzero + a third synchronisation) and returns a strided view
of the padded batch -- padding is bottom/right only, so slicing is the gather."""
import torch
import torch.nn.functional as F


def pad_latents(x: torch.Tensor, max_res: int) -> torch.Tensor:
    """(B, C, h, w) -> (B, C, max_res/8, max_res/8), +inf at the bottom / right (VAE_T5_CLIP.py:438)."""
    side = max_res // 8
    if x.shape[-1] > side or x.shape[-2] > side:
        raise RuntimeError(f"latent {tuple(x.shape[-2:])} larger than the wire format's {side}x{side}")
    return F.pad(x, (0, side - x.shape[-1], 0, side - x.shape[-2]), value=float("inf"))


def unpad_latents(x: torch.Tensor) -> torch.Tensor:
    """Inverse of pad_latents (model_trainer.py:362-370): the leading (h, w) block that holds no +inf padding."""
    first = x[0, 0] == float("inf")
    pad_h, pad_w = torch.stack((first.sum(-2)[0], first.sum(-1)[0])).tolist()     # one device -> host transfer
    h, w = x.shape[2] - int(pad_h), x.shape[3] - int(pad_w)
    return x[:, :, :h, :w]
