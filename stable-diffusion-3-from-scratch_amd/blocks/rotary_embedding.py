"""Axial 2-D RoPE tables: the subset of src/blocks/rotary_embedding.py the trained configuration uses
(RotaryEmbedding 'lang' frequencies 91-166, get_axial_freqs 269-288, apply_rotary_emb 43-76,
rotate_half 36-40).  The rotation itself runs inside the fused QK-norm+RoPE HIP kernel; this module
owns the `freqs` parameter (it is part of the reference's state_dict) and builds the cos/sin tables."""
import torch
from torch import nn


def rotate_half(x):
    x = x.reshape(*x.shape[:-1], -1, 2)
    x1, x2 = x.unbind(dim=-1)
    return torch.stack((-x2, x1), dim=-1).reshape(*x.shape[:-2], -1)


def apply_rotary_emb(freqs, t, start_index=0, scale=1.0, seq_dim=-2, freqs_seq_dim=None):
    """Host-side utility with the reference's semantics (full-width rotation of t by the angle table)."""
    dtype = t.dtype
    rot = freqs.shape[-1]
    mid = t[..., start_index:start_index + rot].float()
    out = (mid * freqs.cos() * scale) + (rotate_half(mid) * freqs.sin() * scale)
    return torch.cat((t[..., :start_index].float(), out, t[..., start_index + rot:].float()), dim=-1).type(dtype)


class RotaryEmbedding(nn.Module):
    def __init__(self, dim, use_xpos=False, interpolate_factor=1.0, theta=10000):
        super().__init__()
        if use_xpos:
            raise RuntimeError("xpos is not part of the trained configuration (not implemented)")
        assert interpolate_factor >= 1.0
        self.interpolate_factor = interpolate_factor  # no effect on get_axial_freqs (as in the reference)
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: (dim // 2)].float() / dim))
        self.freqs = nn.Parameter(freqs, requires_grad=False)
        self._tables = {}

    def get_axial_freqs(self, *dims):
        """(d0, d1, ..., sum over axes of 2*len(freqs)) angle table; axis k rotates its own feature slab."""
        fr = self.freqs.detach().float().cpu()
        all_freqs = []
        for ind, dim in enumerate(dims):
            f = (torch.arange(dim).float()[:, None] * fr[None, :]).repeat_interleave(2, dim=-1)
            shape = [1] * len(dims) + [f.shape[-1]]
            shape[ind] = dim
            all_freqs.append(f.view(shape).expand(*dims, f.shape[-1]))
        return torch.cat(all_freqs, dim=-1)

    def tables(self, height, width, device):
        """cos/sin (height*width, head_dim) fp32 on `device`, cached per shape and parameter version."""
        key = (height, width, str(device), self.freqs._version, self.freqs.data_ptr())
        tb = self._tables.get(key)
        if tb is None:
            ang = self.get_axial_freqs(height, width).reshape(height * width, -1)
            tb = (ang.cos().contiguous().to(device), ang.sin().contiguous().to(device))
            self._tables = {key: tb}
        return tb

    def __deepcopy__(self, memo):
        import copy
        new = RotaryEmbedding.__new__(RotaryEmbedding)
        nn.Module.__init__(new)
        new.interpolate_factor = self.interpolate_factor
        new.freqs = copy.deepcopy(self.freqs, memo)
        new._tables = {}
        return new
