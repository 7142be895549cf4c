"""patchify / unpatchify: mirror of src/blocks/patchify.py (patchify 4-38, unpatchify 41-72)."""
import torch

from .. import ops


def patchify(images, patch_size):
    """(N,C,H,W) -> (N, num_patches, C*ph*pw), patch vector order (C, ph, pw).  2x2 patches, even H/W."""
    if tuple(patch_size) != (2, 2):
        raise RuntimeError("patchify: the HIP path implements the reference's patch_size=2 configuration")
    N, C, H, W = images.shape
    if H % 2 or W % 2:
        raise RuntimeError("patchify: latent height and width must be even")
    tok = ops.patchify(images.contiguous(), images.dtype if images.dtype in (torch.float32, torch.bfloat16) else torch.float32)
    return tok.view(N, (H // 2) * (W // 2), C * 4)


def unpatchify(patches, patch_size, original_shape):
    """(N, num_patches, C*ph*pw) -> (N, C, H, W)."""
    if tuple(patch_size) != (2, 2):
        raise RuntimeError("unpatchify: the HIP path implements the reference's patch_size=2 configuration")
    N, num_patches, patch_dim = patches.shape
    H, W = original_shape
    if H % 2 or W % 2 or (H // 2) * (W // 2) != num_patches:
        raise RuntimeError(f"unpatchify: shape '{[N, H // 2, W // 2, patch_dim // 4, 2, 2]}' is invalid for input of size {patches.numel()}")
    return ops.unpatchify(patches.contiguous().view(N * num_patches, patch_dim), N, patch_dim // 4, H, W, patches.dtype)
