"""AdamW whose step runs GradScaler.unscale_ + clip_grad_norm_ + the AdamW update as three HIP launches over the whole
parameter list (csrc/optim.hip, include/mmdit_hip.h `mmdit_grad_sumsq / mmdit_clip_coef / mmdit_adamw_step`).

It IS a `torch.optim.AdamW` (reference model_trainer.py:260-269): same constructor, same `param_groups`, and the same per-parameter
state (`step` device scalar, `exp_avg`, `exp_avg_sq`) as torch's fused implementation, so `state_dict()` / `load_state_dict()`
and the reference's `optim.pkl` checkpoints are interchangeable with it and the LR scheduler drives it unchanged.  `step()` is
torch's own; `step_clipped()` is the fused sequence.  There is no CPU fallback: it needs the HIP library and CUDA tensors.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib, packing

_REC = np.dtype([("param", "<u8"), ("grad", "<u8"), ("exp_avg", "<u8"), ("exp_avg_sq", "<u8"), ("numel", "<i8"), ("shadow", "<u8")])   # mmdit_adamw_tensor


class ClipAdamW(torch.optim.AdamW):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, **kw):
        kw.setdefault("fused", True)   # state layout of torch's fused implementation (device-side `step`)
        if kw.get("amsgrad") or kw.get("maximize"):
            raise ValueError("ClipAdamW: amsgrad / maximize are not implemented")
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, **kw)
        self._table = None      # device pointer table + chunk map of the current parameter list
        self.table_builds = 0   # pointer-table uploads so far
        self._steps_flat, self._step_views = None, None
        # the update kernel also rewrites the bf16 GEMM-operand copies (packing.shadow_targets); MMDIT_ADAMW_SHADOWS=0: A/B switch
        self.write_shadows = _lib.experiment("MMDIT_ADAMW_SHADOWS", "1") != "0"
        # learning rates as DEVICE doubles, one per param group (the update launch reads them: a launch captured into a hipGraph
        # must not bake the scheduler's current value in); refreshed by one tiny fill only when a group's lr changed
        self._lr_dev, self._lr_host = None, None
        self._capture_staging = None
        self._last_shadows = ([], set())   # (packs, parameter ids) whose bf16 copies the last step_clipped() rewrote (replay bookkeeping)

    # ------------------------------------------------------------------------------------------
    def _state_of(self, p):
        st = self.state[p]
        if len(st) == 0:   # same lazy initialisation as torch.optim.adam._init_group (fused: fp32 device scalar)
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def _bind_steps(self, params):
        """Every `step` scalar becomes a view of ONE flat device tensor, so that the counters advance with one tiny launch
        (`torch._foreach_add_(steps, device_scalar)` reads the scalar back on the host: a device synchronisation per step).
        Re-bound when the states were replaced (load_state_dict) or the set of parameters with gradients changed."""
        views = self._step_views
        if views is not None and len(views) == len(params) and all(self.state[p]["step"] is v for p, v in zip(params, views)):
            return self._steps_flat
        flat = torch.stack([self.state[p]["step"].to(torch.float32).reshape(()) for p in params])
        views = [flat[i] for i in range(len(params))]
        for p, v in zip(params, views):
            self.state[p]["step"] = v
        self._steps_flat, self._step_views = flat, views
        return flat

    def _build_table(self, groups, dev):
        """Device pointer table + chunk map for `groups` = [[(p, g, m, v), ...] per param group].  The chunk map depends only
        on the sizes (built once); the pointer table (40 B per tensor) is re-uploaded whenever a pointer moved -- the gradient
        arenas of the backward pass come from the caching allocator and do move -- through a small ring of pinned staging
        buffers and an asynchronous copy: no allocation and no host wait in the steady state."""
        ptrs = tuple((p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), s.data_ptr() if s is not None else 0)
                     for grp in groups for p, g, m, v, s in grp)
        layout = (tuple(len(grp) for grp in groups), tuple(k[4] for k in ptrs), dev)
        t = self._table
        if t is None or t["layout"] != layout:
            ct, co, ranges = [], [], []
            i = n = 0
            for grp in groups:
                c0 = n
                for p, _, _, _, _ in grp:
                    offs = np.arange(0, p.numel(), _lib.ADAMW_CHUNK, dtype=np.int64)
                    ct.append(np.full(len(offs), i, dtype=np.int32))
                    co.append(offs)
                    i += 1
                    n += len(offs)
                ranges.append((c0, n))   # chunk range of this param group (its own lr / betas / eps / weight decay)
            host = [torch.from_numpy(np.concatenate(ct)).pin_memory(), torch.from_numpy(np.concatenate(co)).pin_memory()]
            t = self._table = dict(layout=layout, host=host, chunk_tensor=host[0].to(dev, non_blocking=True), chunk_off=host[1].to(dev, non_blocking=True),
                                   n_chunks=n, ranges=ranges, ptrs=None, tensors=torch.empty(len(ptrs) * _REC.itemsize, dtype=torch.uint8, device=dev),
                                   staging=[torch.empty(len(ptrs) * _REC.itemsize, dtype=torch.uint8).pin_memory() for _ in range(4)],
                                   staged=[None] * 4, slot=0,
                                   partials=torch.empty(n, dtype=torch.float32, device=dev), out3=torch.empty(3, dtype=torch.float32, device=dev))
        if t["ptrs"] != ptrs:
            self.table_builds += 1
            slot = t["slot"]
            t["slot"] = (slot + 1) % 4
            capturing = torch.cuda.is_current_stream_capturing()
            if capturing:
                # the captured copy node re-reads its pinned source at every replay: it gets a buffer of its own that nothing
                # else ever writes (allocated by prepare_capture(): no allocation inside the capture)
                if self._capture_staging is None or self._capture_staging.numel() != len(ptrs) * _REC.itemsize:
                    raise RuntimeError("ClipAdamW: call prepare_capture() before capturing a step into a graph")
                staging = self._capture_staging
            else:
                if t["staged"][slot] is not None:
                    t["staged"][slot].synchronize()      # the copy that last used this staging buffer (4 uploads ago): long finished
                staging = t["staging"][slot]
            rec = staging.numpy().view(_REC)
            rec[:] = np.array(ptrs, dtype=np.int64).view(_REC).reshape(-1)
            t["tensors"].copy_(staging, non_blocking=True)
            if not capturing:
                t["staged"][slot] = torch.cuda.Event()
                t["staged"][slot].record(torch.cuda.current_stream(dev))
            t["ptrs"] = ptrs
        return t

    def after_replay(self):
        """Host bookkeeping of a step that ran as a graph replay (the launches of step_clipped() were captured, its host side was not):
        the parameters moved behind Tensor._version -> every derived copy is stale (packing epoch) except the bf16 copies the captured
        update kernel rewrote, whose generation advances (the fp8 / mxfp8 weight caches key on it); and the replay re-uploaded the
        CAPTURE's gradient pointers into the device table, so the host-side cache of an eager step's pointers is void."""
        packing.bump_epoch()
        if self._last_shadows[0]:
            packing.mark_rewritten(*self._last_shadows)
        if self._table is not None:
            self._table["ptrs"] = None

    def prepare_capture(self):
        """Call right before capturing step_clipped() into a graph: a pinned pointer-table source that belongs to the capture."""
        n = sum(1 for g in self.param_groups for p in g["params"] if p.requires_grad)
        self._capture_staging = torch.empty(n * _REC.itemsize, dtype=torch.uint8).pin_memory()

    def sync_lr(self, dev):
        """Bring the device copies of the learning rates up to date (outside any graph capture: the scheduler changes
        `param_groups[i]["lr"]` on the host between steps)."""
        lrs = [float(g["lr"]) for g in self.param_groups]
        if self._lr_dev is None or self._lr_dev.device != dev or len(self._lr_host) != len(lrs):
            self._lr_dev = torch.zeros(len(lrs), dtype=torch.float64, device=dev)
            self._lr_host = [None] * len(lrs)
        for i, lr in enumerate(lrs):
            if self._lr_host[i] != lr:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("ClipAdamW: the learning rate changed inside a graph capture; call sync_lr() before capturing / replaying")
                self._lr_dev[i].fill_(lr)
                self._lr_host[i] = lr
        return self._lr_dev

    @torch.no_grad()
    def step_clipped(self, loss_scale=None, max_norm=1.0):
        """grads *= (1/loss_scale) * min(1, max_norm / (||grads / loss_scale|| + 1e-6));  AdamW update unless a gradient is
        inf/nan;  every `step` += 1 - found_inf.  `loss_scale`: the GradScaler's device scalar or None; `max_norm` None: no clip.
        Returns (found_inf, grad_norm) as 0-dim device tensors (nothing is synchronised).  The stored .grad tensors are left
        as backward produced them (the reference zeroes them right after the step, model_trainer.py:503)."""
        groups, dev = [], None
        shadows, shadow_packs = {}, []
        if self.write_shadows:
            p0 = next((p for g in self.param_groups for p in g["params"] if p.grad is not None), None)
            if p0 is not None and p0.is_cuda:
                shadows, shadow_packs = packing.shadow_targets(p0.device)
        for group in self.param_groups:
            rows = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError("ClipAdamW.step_clipped needs contiguous fp32 CUDA parameters and gradients")
                st = self._state_of(p)
                rows.append((p, p.grad, st["exp_avg"], st["exp_avg_sq"], shadows.get(id(p))))
                dev = p.device
            groups.append(rows)
        if dev is None:
            raise RuntimeError("ClipAdamW.step_clipped: no parameter has a gradient")
        t = self._build_table(groups, dev)
        steps_flat = self._bind_steps([r[0] for rows in groups for r in rows])
        L = _lib.lib()
        s = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        vp = lambda x: ctypes.c_void_p(x.data_ptr())
        _lib.check(L.mmdit_grad_sumsq(vp(t["tensors"]), vp(t["chunk_tensor"]), vp(t["chunk_off"]), t["n_chunks"], vp(t["partials"]), s), "mmdit_grad_sumsq")
        _lib.check(L.mmdit_clip_coef(vp(t["partials"]), t["n_chunks"], vp(loss_scale) if loss_scale is not None else None,
                                     float(max_norm) if max_norm is not None else 0.0, vp(t["out3"]), s), "mmdit_clip_coef")
        lr_dev = self.sync_lr(dev)
        for gi, (group, rows, (c0, c1)) in enumerate(zip(self.param_groups, groups, t["ranges"])):
            if not rows:
                continue
            step0 = self.state[rows[0][0]]["step"]
            b1, b2 = group["betas"]
            _lib.check(L.mmdit_adamw_step_dlr(vp(t["tensors"]), ctypes.c_void_p(t["chunk_tensor"].data_ptr() + 4 * c0), ctypes.c_void_p(t["chunk_off"].data_ptr() + 8 * c0), c1 - c0,
                                              vp(t["out3"]), vp(step0), ctypes.c_void_p(lr_dev.data_ptr() + 8 * gi), float(b1), float(b2), float(group["eps"]),
                                              float(group["weight_decay"]), s), "mmdit_adamw_step_dlr")
        steps_flat.add_(1.0 - t["out3"][1])
        packing.bump_epoch()     # the kernels wrote the parameters through raw pointers: the bf16 operand copies are stale now ...
        self._last_shadows = (shadow_packs, {id(r[0]) for rows in groups for r in rows if r[4] is not None})
        if shadow_packs:         # ... except the ones the update kernel has just rewritten
            packing.mark_rewritten(*self._last_shadows)
        # out3 is overwritten by the next call: hand out copies
        res = t["out3"].clone()
        return res[1], res[2]
