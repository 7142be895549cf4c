"""Data-parallel gradient averaging over RCCL/xGMI (backend "nccl" on ROCm), replacing the DDP Reducer
behind model_trainer.py:224/467 of the reference.

Design for MI355X: one process per GPU; gradients are averaged in flat fp32 buckets on a dedicated side
HIP stream so the collective of block i overlaps the backward kernels of block i-1.  The engine hands
over each block's gradients as soon as they are final (engine.model_bwd(on_grads=...)); xGMI is
point-to-point (7 links x ~153 GB/s), so buckets are large (one per transformer block, ~105 MB fp32
for MMDiT-B) to stay bandwidth- rather than latency-bound.  No data-path collective other than this one.
"""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, group=None, bucket_bytes=32 << 20, force=False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bucket_bytes = bucket_bytes
        self.enabled = self.world > 1 or (force and dist.is_initialized())   # force: exercise the path with one rank (tests)
        self._cuda = torch.cuda.is_available()
        self._stream = None
        self._pending = []       # tensors waiting for a bucket
        self._pending_bytes = 0
        self._inflight = []      # (flat, tensors, work)
        self._after = None
        self.skip = False        # True on non-final gradient-accumulation micro-steps (no_sync semantics)
        self._views_wanted = False

    def _side(self, device):
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=device)
        return self._stream

    # ---- called by the engine (or the hook fallback) as gradients become final -----------------
    def _reduce_in_place(self, arenas):
        """All-reduce flat buffers where they are (no gather, no copy-back)."""
        if arenas[0].is_cuda:
            side = self._side(arenas[0].device)
            side.wait_stream(torch.cuda.current_stream())
            if self._after is not None:
                side.wait_stream(self._after)
            ctx = torch.cuda.stream(side)
        else:
            ctx = _null()
        nccl = dist.get_backend(self.group) == "nccl"
        with ctx:
            for a in arenas:
                work = dist.all_reduce(a, op=dist.ReduceOp.AVG if nccl else dist.ReduceOp.SUM, group=self.group, async_op=True)
                self._inflight.append((a, None, work, not nccl))
        # no record_stream: the buffers are released on the main stream, which has waited for this stream in finish()

    def add_bucket(self, tensors, after=None, arenas=None):
        """Engine path (zero copy-back): `tensors` become ONE bucket right away; returns replacement tensors -- views of
        the flat bucket, same shapes and order -- that hold the averaged gradients once finish() has run.  The caller
        uses them in place of `tensors` (which are not written back), so the ~400 per-parameter copy-back launches of
        the generic path and a second pass over the gradients disappear.  With `arenas` (flat buffers that ARE the storage
        of `tensors`) even the gather copy goes away.  Returns None when the caller keeps using `tensors`."""
        if after is not None:
            self._after = after
        if not self.enabled or self.skip:
            return None
        tensors = [t for t in tensors if t is not None]
        if not tensors:
            return None
        self.flush()                      # keep the launch order of anything queued through add()
        if arenas:
            # the producer guarantees that `tensors` are exactly the contents of these flat buffers (engine.block_bwd):
            # average them where they are -- no gather copy, no replacement
            self._reduce_in_place(arenas)
            return None
        self._pending, self._pending_bytes = tensors, 1
        self._views_wanted = True
        self.flush()
        flat = self._inflight[-1][0]
        views, off = [], 0
        for t in tensors:
            n = t.numel()
            views.append(flat[off:off + n].view(t.shape))
            off += n
        return views

    def add(self, tensors, after=None):
        """tensors: gradients that are final once the current stream (and `after`, the stream that produced
        the weight gradients, if given) reach this point."""
        if after is not None:
            self._after = after
        if not self.enabled or self.skip:
            return
        for t in tensors:
            if t is None:
                continue
            self._pending.append(t)
            self._pending_bytes += t.numel() * t.element_size()
        if self._pending_bytes >= self.bucket_bytes:
            self.flush()

    def flush(self):
        if not self._pending:
            return
        tensors, self._pending, self._pending_bytes = self._pending, [], 0
        if tensors[0].is_cuda:
            side = self._side(tensors[0].device)
            side.wait_stream(torch.cuda.current_stream())
            if self._after is not None:
                side.wait_stream(self._after)
            with torch.cuda.stream(side):
                self._launch(tensors)
        else:
            self._launch(tensors)

    def _launch(self, tensors):
        flat = torch.cat([t.reshape(-1) for t in tensors])
        backend = dist.get_backend(self.group)
        if backend == "nccl":
            work = dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            div = False
        else:
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            div = True
        # (flat, tensors to copy back into | None when the caller took views of flat, work, divide-after)
        self._inflight.append((flat, None if self._views_wanted else tensors, work, div))
        self._views_wanted = False

    # ---- called by the trainer after backward(), before clip / optimizer step --------------------
    def finish(self):
        if not self.enabled or self.skip:
            return
        self.flush()
        for flat, tensors, work, div in self._inflight:
            cuda = flat.is_cuda
            ctx = torch.cuda.stream(self._stream) if cuda else _null()
            with ctx:
                work.wait()
                if div:
                    flat.div_(self.world)
                if tensors is not None:
                    outs, off = [], 0
                    for t in tensors:
                        n = t.numel()
                        outs.append(flat[off:off + n].view_as(t))
                        off += n
                    torch._foreach_copy_(tensors, outs)
        self._inflight = []
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)

    def attach_hooks(self, params):
        """Fallback for modules without an engine callback: reduce each parameter's gradient as soon as
        autograd has accumulated it (reverse registration order fills the buckets)."""
        for p in params:
            if p.requires_grad:
                p.register_post_accumulate_grad_hook(lambda q: self.add([q.grad]))


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def broadcast_parameters(module, group=None, src=0):
    """DDP's initial parameter broadcast (model_trainer.py:224) so every replica starts identical."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for p in module.parameters():
        dist.broadcast(p.data, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
    from . import packing
    packing.bump_epoch()     # collectives write through .data: any bf16 weight copy derived before the broadcast is stale
