"""Data-parallel gradient averaging over RCCL/xGMI (backend "nccl" on ROCm), replacing the DDP Reducer
behind model_trainer.py:224/467 of the reference.

Design for MI355X: one process per GPU; gradients are averaged in flat fp32 buckets on a dedicated side
HIP stream so the collective of block i overlaps the backward kernels of block i-1.  The engine hands
over each block's gradients as soon as they are final (engine.model_bwd(on_grads=...)); xGMI is
point-to-point (7 links x ~153 GB/s), so buckets are large (one per transformer block, ~105 MB fp32
for MMDiT-B) to stay bandwidth- rather than latency-bound.  No data-path collective other than this one.

Three selectable algorithms per bucket (constructor `algorithm`, env MMDIT_REDUCE_ALGO; DESIGN.md 5 for the expected times):
  allreduce  dist.all_reduce(AVG): RCCL's own choice of ring / tree over its channels (default: the path every RCCL release tunes).
  rs_ag      dist.reduce_scatter_tensor + dist.all_gather_into_tensor: the same traffic split in its two phases (the seam a sharded
             optimizer step goes into).
  direct     reduce-scatter as ONE all-to-all (rank r sends shard j to rank j over the direct xGMI link r-j: all 7 links of a GPU carry
             1/8 of the bucket at the same time) + local fp32 sum of the 8 received shards + all-gather as one all-to-all of the
             averaged shard.  Optional bf16 wire format (`wire_dtype=torch.bfloat16`: half the bytes per link, fp32 accumulation of
             the received shards; the gathered average is bf16-rounded).
Every algorithm needs the bucket length to be a multiple of the world size only for rs_ag / direct; the engine's arenas are padded to
multiples of 1024 elements (engine.ARENA_QUANTUM), other buckets fall back to allreduce.
"""
import os

import torch
import torch.distributed as dist

ALGORITHMS = ("allreduce", "rs_ag", "direct")


class GradReducer:
    def __init__(self, group=None, bucket_bytes=32 << 20, force=False, algorithm=None, wire_dtype=None, blocks_per_fork=None):
        self.group = group
        # engine path: how many blocks' arenas are handed to the side stream at once.  Every hand-over is a fork of the captured step (the side stream waits for the
        # main stream's last kernel, and on replay the main stream's next kernel starts ~17 us late: profiles/r06_claiming_forcedist_contention.txt); k blocks per fork
        # mean 1 / k as many forks and the first collective of a group starting up to k - 1 blocks later.  The collectives themselves stay one per arena.
        self.blocks_per_fork = max(1, int(blocks_per_fork if blocks_per_fork is not None else os.environ.get("MMDIT_REDUCE_BLOCKS_PER_FORK", "1")))
        self._held, self._held_blocks = [], 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bucket_bytes = bucket_bytes
        self.enabled = self.world > 1 or (force and dist.is_initialized())   # force: exercise the path with one rank (tests)
        self.algorithm = algorithm or os.environ.get("MMDIT_REDUCE_ALGO", "allreduce")
        if self.algorithm not in ALGORITHMS:
            raise ValueError(f"GradReducer: algorithm must be one of {ALGORITHMS}, not {self.algorithm!r}")
        if wire_dtype is None and os.environ.get("MMDIT_REDUCE_WIRE", "") == "bf16":
            wire_dtype = torch.bfloat16
        if wire_dtype not in (None, torch.float32, torch.bfloat16):
            raise ValueError("GradReducer: wire_dtype is torch.float32 (default) or torch.bfloat16")
        self.wire_dtype = None if wire_dtype == torch.float32 else wire_dtype
        self._cuda = torch.cuda.is_available()
        self._stream = None
        self._pending = []       # tensors waiting for a bucket
        self._pending_bytes = 0
        self._inflight = []      # (flat, tensors to copy back into | None)
        self._works = []         # non-RCCL backends: (Work, flat) of all-reduces still running on the backend's own threads
        self._after = None
        self.skip = False        # True on non-final gradient-accumulation micro-steps (no_sync semantics)
        self._views_wanted = False
        self.buckets = 0         # buckets averaged so far (tests / logs)
        # rs_ag / direct: the shard / receive / send buffers, ONE set per (role, length, dtype, device), reused by every bucket of that
        # size in every step (the collectives of a rank are ordered on the side stream, so a buffer is free again when the next bucket
        # of its size starts); nothing is allocated per bucket per step -- at MMDiT-B that was ~300 MB of allocator traffic per block
        self._bufs = {}
        # timing (eager steps only -- events recorded inside a captured step cannot be read): set `timing = True`, run steps, read
        # timing_summary(): per-bucket HIP events around each collective on the side stream + one pair around the main stream's wait
        self.timing = False
        self._tev, self._texposed = [], []

    def describe(self):
        """One line for the logs: what this rank's reducer will do."""
        backend = dist.get_backend(self.group) if dist.is_initialized() else "none"
        wire = "bf16" if self.wire_dtype == torch.bfloat16 else "fp32"
        return f"GradReducer(world={self.world}, enabled={self.enabled}, backend={backend}, algorithm={self.algorithm}, wire={wire}, blocks_per_fork={self.blocks_per_fork})"

    def _buf(self, role, n, dtype, device):
        key = (role, n, dtype, device)
        b = self._bufs.get(key)
        if b is None:
            b = self._bufs[key] = torch.empty(n, dtype=dtype, device=device)
        return b

    def timing_summary(self, steps=1):
        """{"allreduce_total_ms", "allreduce_exposed_ms", "buckets_per_step"} per step over the steps run since `timing` was set: total =
        sum of the collectives' own durations on the side stream, exposed = how long the main stream had to wait for the side stream
        in finish() (0 when the last bucket's collective ended before the backward did).  Synchronises the device."""
        torch.cuda.synchronize()
        total = sum(e0.elapsed_time(e1) for e0, e1 in self._tev)
        exposed = sum(max(0.0, e0.elapsed_time(e1)) for e0, e1 in self._texposed)
        out = {"allreduce_total_ms": round(total / max(1, steps), 3), "allreduce_exposed_ms": round(exposed / max(1, steps), 3),
               "buckets_per_step": len(self._tev) // max(1, steps)}
        self._tev, self._texposed = [], []
        return out

    def _side(self, device):
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=device)
        return self._stream

    # ---- the collective -------------------------------------------------------------------------
    def _average(self, flat):
        """Average the flat fp32 buffer over the ranks, in place.  Called inside the side-stream context on a GPU: the (synchronous)
        torch.distributed calls enqueue behind the stream's work and make the stream -- not the host -- wait for their results."""
        self.buckets += 1
        W, n = self.world, flat.numel()
        nccl = dist.get_backend(self.group) == "nccl"
        if self.timing and flat.is_cuda and not torch.cuda.is_current_stream_capturing():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            try:
                self._average_untimed(flat, W, n, nccl)
            finally:
                e1.record()
                self._tev.append((e0, e1))
            return
        self._average_untimed(flat, W, n, nccl)

    def _average_untimed(self, flat, W, n, nccl):
        algo = self.algorithm if (n % W == 0 and n > 0) else "allreduce"
        if algo == "allreduce":
            if nccl:
                dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group)
            else:
                # gloo (the CPU tests, CUDA tensors over gloo): a synchronous call would block the HOST in the middle of the backward;
                # keep the overlap by waiting in finish() (the division by the world size follows the wait)
                self._works.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True), flat))
            return
        s = n // W
        if algo == "rs_ag":
            shard = self._buf("shard", s, flat.dtype, flat.device)
            dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.AVG if nccl else dist.ReduceOp.SUM, group=self.group)
            if not nccl:
                shard.div_(W)
            dist.all_gather_into_tensor(flat, shard, group=self.group)
            return
        # (rs_ag / direct on a non-RCCL backend are multi-step and stay synchronous: test-only there)
        # direct: shard j of every rank -> rank j (one all-to-all), fp32 sum there, averaged shard -> everyone (one all-to-all)
        wire = self.wire_dtype or flat.dtype
        send = flat if wire == flat.dtype else self._buf("send", n, wire, flat.device).copy_(flat)
        recv = self._buf("recv", n, wire, flat.device)
        dist.all_to_all_single(recv, send, group=self.group)
        shard = self._buf("sum", s, torch.float32, flat.device)
        torch.sum(recv.view(W, s), dim=0, dtype=torch.float32, out=shard).div_(W)    # (rank order: the same summation on every rank)
        send2 = self._buf("send2", n, wire, flat.device)
        send2.view(W, s).copy_(shard.unsqueeze(0).expand(W, s))                      # (rounds to the wire format)
        if wire == flat.dtype:
            dist.all_to_all_single(flat, send2, group=self.group)
        else:
            dist.all_to_all_single(recv, send2, group=self.group)
            flat.copy_(recv)

    # ---- called by the engine (or the hook fallback) as gradients become final -----------------
    def _reduce_in_place(self, arenas):
        """Average flat buffers where they are (no gather, no copy-back)."""
        if arenas[0].is_cuda:
            side = self._side(arenas[0].device)
            side.wait_stream(torch.cuda.current_stream())
            if self._after is not None:
                side.wait_stream(self._after)
            ctx = torch.cuda.stream(side)
        else:
            ctx = _null()
        with ctx:
            for a in arenas:
                self._average(a)
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
            self._held.extend(arenas)
            self._held_blocks += 1
            if self._held_blocks >= self.blocks_per_fork:
                self._flush_held()
            return None
        self._flush_held()
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

    def _flush_held(self):
        if self._held:
            held, self._held, self._held_blocks = self._held, [], 0
            self._reduce_in_place(held)

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
        parts = [t.reshape(-1) for t in tensors]
        n = sum(p.numel() for p in parts)
        pad = -n % self.world if self.algorithm != "allreduce" else 0      # (rs_ag / direct: equal shards)
        if pad:
            parts.append(torch.zeros(pad, dtype=parts[0].dtype, device=parts[0].device))
        flat = torch.cat(parts)
        self._average(flat)
        # (flat, tensors to copy back into | None when the caller took views of flat)
        self._inflight.append((flat, None if self._views_wanted else tensors))
        self._views_wanted = False

    # ---- called by the trainer after backward(), before clip / optimizer step --------------------
    def finish(self):
        if not self.enabled or self.skip:
            return
        self._flush_held()
        self.flush()
        for work, flat in self._works:
            work.wait()
            flat.div_(self.world)
        self._works = []
        for flat, tensors in self._inflight:
            if tensors is None:
                continue
            ctx = torch.cuda.stream(self._stream) if flat.is_cuda else _null()
            with ctx:
                outs, off = [], 0
                for t in tensors:
                    n = t.numel()
                    outs.append(flat[off:off + n].view_as(t))
                    off += n
                torch._foreach_copy_(tensors, outs)
        self._inflight = []
        if self._stream is not None:
            if self.timing and not torch.cuda.is_current_stream_capturing():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()                      # main stream: the backward is done here
                e1.record(self._stream)          # side stream: the last collective is done here
                self._texposed.append((e0, e1))
            torch.cuda.current_stream().wait_stream(self._stream)

    def reset(self):
        """Forget everything queued or in flight (a graph capture that raised: its collectives were never launched)."""
        self._pending, self._pending_bytes, self._inflight, self._views_wanted, self._works = [], 0, [], False, []
        self._held, self._held_blocks = [], 0

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
