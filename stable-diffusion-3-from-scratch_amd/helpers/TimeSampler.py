"""Logit-normal timestep sampler: mirror of src/helpers/TimeSampler.py (lines 5-21)."""
import torch


class TimeSampler:
    def __init__(self, weighted=True, m=0.0, s=1.0):
        self.weighted, self.m, self.s = weighted, m, s

    def __call__(self, n, generator=None, device=None):
        return self.sample(n, generator=generator, device=device)

    def sample(self, n, generator=None, device=None):
        """sigmoid(N(m, s)) (weighted) or U(0,1).  The reference draws from the CPU default generator
        (generator=None, device=None); the synthetic benchmark passes a device generator to avoid host syncs."""
        if self.weighted:
            u = torch.randn(n, generator=generator, device=device) * self.s + self.m
            return torch.sigmoid(u)
        return torch.rand(n, generator=generator, device=device)
