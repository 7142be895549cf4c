"""Packed low-precision shadow copies of the fp32 master weights.

Parameters keep the reference's names, shapes and fp32 dtype (state_dict / optimizer layout,
SURVEY.md 8b).  For the MFMA GEMMs several of them are concatenated row-wise (q|k|v, the 12
modulation matrices of a block, ...) into one operand in the activation dtype of the precision
mode; the reference gets its bf16 copies from torch.autocast's weight cache
(model_trainer.py:416).  Copies are refreshed only when a parameter's version / storage changes.
"""
import torch

from . import ops


class Pack:
    def __init__(self, params):
        self.params = list(params)
        self.rows = [p.shape[0] for p in self.params]
        self.cols = self.params[0].numel() // self.params[0].shape[0]
        self._key = None
        self._val = None

    def _state(self, mode):
        return (mode.fast,) + tuple((p._version, p.data_ptr()) for p in self.params)

    def get(self, mode):
        key = self._state(mode)
        if key != self._key:
            views = [p.detach().view(r, self.cols) for p, r in zip(self.params, self.rows)]
            if not mode.fast:
                self._val = views[0] if len(views) == 1 else torch.cat(views, 0)
            else:
                out = torch.empty((sum(self.rows), self.cols), dtype=mode.T, device=views[0].device)
                a = 0
                for v, r in zip(views, self.rows):
                    ops.cast(v, mode.T, out=out[a:a + r])
                    a += r
                self._val = out
            self._key = key
        return self._val

    def split_grad(self, G, out: dict):
        """G: packed fp32 gradient (sum rows, cols) -> per-parameter views."""
        a = 0
        for p, r in zip(self.params, self.rows):
            out[id(p)] = G[a:a + r].view(p.shape)
            a += r

    def __deepcopy__(self, memo):
        import copy
        return Pack([copy.deepcopy(p, memo) for p in self.params])
