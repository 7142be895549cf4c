"""Canary-guarded device allocations: a debug mode that finds out-of-bounds writes of the HIP kernels.

GPU AddressSanitizer is not available on the MI355X pool, and an out-of-bounds store of a training-path kernel is invisible to every
output-comparing test as long as it lands in allocator slack or in a buffer that is rewritten later.  `install()` replaces the name
`torch` inside the modules that allocate kernel outputs (ops, engine, optim, vae) by a proxy whose empty / zeros / empty_like /
zeros_like over-allocate every CUDA tensor by PAD bytes on each side, fill the two guards with a sentinel and hand out the view in
between (byte-exact: the back guard starts at the first byte behind the last element).  `verify()` checks every guard handed out
so far; with `per_launch=True` the C-ABI status check that follows every launch (`_lib.check`) also synchronises and verifies, so
the first violating launch is named.  Nothing of this is active unless install() is called (tools/probes/guard_step.py,
tests/test_guard_gpu.py): the product path allocates with plain torch.
"""
import traceback

import torch as _torch

PAD = 512
SENTINEL = 0xA5


class Violation(RuntimeError):
    pass


class _Registry:
    def __init__(self):
        self.bufs = []          # (uint8 buffer incl. guards, data bytes, tag)
        self.seq = 0
        self.per_launch = False
        self.launches = 0
        self.found = []
        self.watch = None       # set of allocation sequence numbers to verify per launch (None: all)

    def alloc(self, shape, dtype, device, zero):
        if isinstance(shape, int):
            shape = (shape,)
        shape = tuple(int(s) for s in shape)
        n = 1
        for s in shape:
            n *= s
        nbytes = n * _torch.empty((), dtype=dtype).element_size()
        buf = _torch.empty(PAD + nbytes + PAD, dtype=_torch.uint8, device=device)
        buf[:PAD].fill_(SENTINEL)
        buf[PAD + nbytes:].fill_(SENTINEL)
        data = buf[PAD:PAD + nbytes]
        if zero:
            data.zero_()
        fr = [f for f in traceback.extract_stack(limit=8)[:-2] if "debug_guard" not in f.filename][-3:]
        tag = (self.seq, tuple(shape), str(dtype), " <- ".join(f"{f.filename.rsplit('/', 1)[-1]}:{f.lineno}:{f.name}" for f in reversed(fr)))
        self.seq += 1
        self.bufs.append((buf, nbytes, tag))
        return data.view(dtype).view(shape)

    def verify(self, what="", only=None):
        """Check the guards (of the allocations in `only`, or all); returns the list of violations found by this call."""
        bad = []
        sel = [b for b in self.bufs if only is None or b[2][0] in only]
        if not sel:
            return bad
        # one pass: count the non-sentinel bytes of every guard
        fronts = _torch.stack([b[0][:PAD] for b in sel])
        backs = _torch.stack([b[0][PAD + b[1]:] for b in sel])
        cnt = _torch.stack([(fronts != SENTINEL).sum(1), (backs != SENTINEL).sum(1)], 1).cpu()
        for (buf, nbytes, tag), (cf, cb) in zip(sel, cnt.tolist()):
            for side, c, g in (("front", cf, buf[:PAD]), ("back", cb, buf[PAD + nbytes:])):
                if c:
                    idx = (g != SENTINEL).nonzero().flatten().cpu()
                    rec = dict(after=what, side=side, bytes=int(c), first=int(idx[0]), last=int(idx[-1]), alloc=tag,
                               sample=bytes(g[idx[:16]].cpu().tolist()).hex())
                    bad.append(rec)
                    g.fill_(SENTINEL)      # re-arm: report every violating launch once
        self.found += bad
        return bad

    def reset(self):
        self.bufs, self.found = [], []


REG = _Registry()


class _GuardedTorch:
    """Stands in for the `torch` module inside ops / engine / optim / vae: CUDA allocations get guards, the rest is torch."""

    def __init__(self, real):
        self.__dict__["_real"] = real

    def __getattr__(self, k):
        return getattr(self._real, k)

    @staticmethod
    def _cuda(device):
        return device is not None and _torch.device(device).type == "cuda"

    def empty(self, *shape, dtype=None, device=None, **kw):
        if len(shape) == 1 and not isinstance(shape[0], int):
            shape = tuple(shape[0])
        if not self._cuda(device) or kw:
            return self._real.empty(shape, dtype=dtype, device=device, **kw)
        return REG.alloc(shape, dtype or _torch.float32, device, False)

    def zeros(self, *shape, dtype=None, device=None, **kw):
        if len(shape) == 1 and not isinstance(shape[0], int):
            shape = tuple(shape[0])
        if not self._cuda(device) or kw:
            return self._real.zeros(shape, dtype=dtype, device=device, **kw)
        return REG.alloc(shape, dtype or _torch.float32, device, True)

    def empty_like(self, t, dtype=None, **kw):
        if not t.is_cuda or kw or not t.is_contiguous():
            return self._real.empty_like(t, dtype=dtype, **kw)
        return REG.alloc(tuple(t.shape), dtype or t.dtype, t.device, False)

    def zeros_like(self, t, dtype=None, **kw):
        if not t.is_cuda or kw or not t.is_contiguous():
            return self._real.zeros_like(t, dtype=dtype, **kw)
        return REG.alloc(tuple(t.shape), dtype or t.dtype, t.device, True)


_installed = {}


def install(per_launch=False, watch=None):
    """Guard every kernel-output allocation of the package from now on.  per_launch: synchronise and verify after every launch
    (slow; names the violating launch).  watch: allocation sequence numbers to restrict the per-launch check to."""
    from . import _lib, engine, ops, optim, vae
    REG.per_launch, REG.watch = per_launch, (set(watch) if watch is not None else None)
    if _installed:
        return REG
    proxy = _GuardedTorch(_torch)
    for mod in (ops, engine, optim, vae):
        _installed[mod] = mod.torch
        mod.torch = proxy
    ops.NO_POOL = True       # zero-pool slices become allocations of their own
    real_check = _lib.check

    def check(status, what):
        real_check(status, what)
        REG.launches += 1
        if REG.per_launch:
            _torch.cuda.synchronize()
            bad = REG.verify(f"launch #{REG.launches} {what}", REG.watch)
            for b in bad:
                print(f"[guard] {b}", flush=True)

    _installed["check"] = real_check
    _lib.check = check
    ops.check = check
    return REG


def uninstall():
    from . import _lib, ops
    for mod, real in list(_installed.items()):
        if mod == "check":
            _lib.check = real
            ops.check = real
            ops.NO_POOL = False
        else:
            mod.torch = real
    _installed.clear()
    REG.reset()
