"""Packed low-precision shadow copies of the fp32 master weights.

Parameters keep the reference's names, shapes and fp32 dtype (state_dict / optimizer layout,
SURVEY.md 8b).  For the MFMA GEMMs several of them are concatenated row-wise (q|k|v, the 12
modulation matrices of a block, ...) into one operand in the activation dtype of the precision
mode; the reference gets its bf16 copies from torch.autocast's weight cache
(model_trainer.py:416).  Copies are refreshed -- all stale ones of a device in one launch -- when a parameter's version / storage
changed or an optimizer stepped (see _EPOCH below).
"""
import weakref

import torch

from . import ops

# Staleness.  Tensor._version is NOT enough: torch's fused AdamW (and this package's HIP optimizer step, which writes through
# raw pointers) update parameters without bumping it -- measured on ROCm: version 1 -> 1 across optimizer.step() -- so a cache
# keyed on versions alone keeps multiplying with the initial weights.  Every optimizer step therefore advances a global epoch
# (a global torch.optim post-step hook, so the reference's own trainer is covered; ClipAdamW.step_clipped calls bump_epoch()
# itself) and the epoch is part of every cache key.
_EPOCH = [0]
_REGISTRY = weakref.WeakSet()


def bump_epoch(*_a, **_k):
    """Parameters may have changed behind Tensor._version: every Pack re-derives its copy at the next use."""
    _EPOCH[0] += 1


from torch.optim.optimizer import register_optimizer_step_post_hook as _register_post_hook  # noqa: E402

_register_post_hook(bump_epoch)


class Pack:
    def __init__(self, params):
        self.params = list(params)
        self.rows = [p.shape[0] for p in self.params]
        self.cols = self.params[0].numel() // self.params[0].shape[0]
        self._key = None
        self._val = None
        self._dsts = None
        self._buf = None        # persistent bf16 buffer of the fast mode (refreshed in place: its address is part of cached launch tables)
        self.generation = 0     # refresh count (consumers that derive further copies, e.g. the fp8 weight cache, key on it)
        _REGISTRY.add(self)

    def _state(self, mode):
        return self._state_of(mode.fast)

    def _state_of(self, fast):
        return (bool(fast), _EPOCH[0]) + tuple((p._version, p.data_ptr()) for p in self.params)

    def _dst_views(self):
        dev = self.params[0].device
        if self._buf is None or self._buf.device != dev:
            self._buf = torch.empty((sum(self.rows), self.cols), dtype=torch.bfloat16, device=dev)
            self._dsts, a = [], 0
            for r in self.rows:
                self._dsts.append(self._buf[a:a + r])
                a += r
        return self._dsts

    def _pairs(self):
        return [(p.detach().view(r, self.cols), d) for p, r, d in zip(self.params, self.rows, self._dst_views())]

    def _mark(self, mode=None):
        """The bf16 buffer holds the current parameters (mode None: the fast mode)."""
        self.generation += 1
        self._val = self._buf
        self._val._mmdit_gen = self.generation
        self._key = self._state_of(True if mode is None else mode.fast)

    def get(self, mode):
        key = self._state(mode)
        if key != self._key:
            if not mode.fast:
                views = [p.detach().view(r, self.cols) for p, r in zip(self.params, self.rows)]
                self._val = views[0] if len(views) == 1 else torch.cat(views, 0)
                self._key = key
            elif self.params[0].is_cuda:
                # one launch refreshes this pack and every other stale bf16 copy on the device (all of them after an optimizer step)
                dev = self.params[0].device
                stale = [self] + [q for q in list(_REGISTRY) if q is not self and q._buf is not None and q.params[0].device == dev and q._state(mode) != q._key]
                pairs = [pr for q in stale for pr in q._pairs()]
                ops.cast_multi(pairs)
                for q in stale:
                    q._mark(mode)
            else:
                raise RuntimeError("bf16 weight copies are produced by the HIP library on an MI355X only (no CPU fallback): move the model to the GPU")
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


def shadow_targets(device):
    """For an optimizer that rewrites the bf16 copies itself (optim.ClipAdamW): {id(param): bf16 view of its copy} over the
    live packs that are in the bf16 mode on `device`; parameters that sit in more than one pack are left to the regular
    refresh.  Returns (targets, packs)."""
    count, target, packs = {}, {}, []
    for q in list(_REGISTRY):
        if q._buf is None or q._key is None or not q._key[0] or q._buf.device != device or q.params[0].device != device:
            continue
        packs.append(q)
        for p, dst in zip(q.params, q._dst_views()):
            count[id(p)] = count.get(id(p), 0) + 1
            target[id(p)] = dst
    return {k: t for k, t in target.items() if count[k] == 1}, packs


def mark_rewritten(packs, rewritten_ids):
    """After bump_epoch(): the packs whose every parameter had its bf16 copy rewritten by the optimizer are current again."""
    for q in packs:
        if all(id(p) in rewritten_ids for p in q.params):
            q._mark()
