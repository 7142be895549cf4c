"""Sinusoidal timestep embedding: mirror of src/blocks/PositionalEncoding.py (lines 8-30)."""
import torch
from torch import nn

from .. import ops


class PositionalEncoding(nn.Module):
    def __init__(self, dim, device):
        super().__init__()
        self.dim = dim
        # same expression as the reference (PositionalEncoding.py:15-16): i = 0..dim-1, not dim/2
        self.denom = (torch.tensor(10000.0) ** ((2 * torch.arange(self.dim)) / self.dim)).to(dtype=torch.float, device=device)
        self._ones = None

    def _denom_on(self, device):
        if self.denom.device != device:
            self.denom = self.denom.to(device)
        return self.denom

    def forward(self, time):
        """time (N,) -> (N, dim): cat(sin(e[:, 0::2]), cos(e[:, 1::2])), e = time / denom."""
        t = time.float().contiguous()
        one = torch.ones(1, dtype=torch.float32, device=t.device)
        return ops.time_embed_fwd(t, one, self._denom_on(t.device), torch.float32)
