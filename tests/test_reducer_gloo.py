"""N>1 data-parallel path on CPU: world_size-2 gloo processes exercise sd3_amd.reducer.GradReducer (bucketed
all-reduce, hook fallback, accumulation skip, parameter broadcast) and the trainer's optimizer-step logic on a
small stand-in module (the HIP model itself has no CPU fallback).  After k steps every rank must hold
bit-identical parameters, equal to a single-process run on the concatenated batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _net():
    torch.manual_seed(0)
    return nn.Sequential(nn.Linear(16, 32), nn.SiLU(), nn.Linear(32, 8))


def _data(step, rank, world):
    g = torch.Generator().manual_seed(100 + step)
    x, y = torch.randn((4 * world, 16), generator=g), torch.randn((4 * world, 8), generator=g)
    return x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4], x, y


def _worker(rank, world, port, accum, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sd3_amd  # noqa: F401
    from sd3_amd.reducer import GradReducer, broadcast_parameters
    net = _net()
    if rank == 1:
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)       # diverge on purpose: the broadcast must repair it
    broadcast_parameters(net)
    red = GradReducer(bucket_bytes=1024)
    red.attach_hooks(net.parameters())
    opt = torch.optim.AdamW(net.parameters(), lr=1e-2)
    step = 0
    for it in range(3):
        for k in range(accum):
            x, y, _, _ = _data(step, rank, world)
            step += 1
            red.skip = k != accum - 1
            if red.skip:
                # hooks must not reduce on non-final micro-steps; grads keep accumulating locally
                pass
            loss = nn.functional.mse_loss(net(x), y) / accum
            loss.backward()
            if not red.skip:
                # on the final micro-step the hook saw grads that already include the earlier micro-steps
                red.finish()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
        opt.step()
        opt.zero_grad()
    q.put((rank, [p.detach().numpy().copy() for p in net.parameters()]))   # by value: the process exits right after
    dist.barrier()
    dist.destroy_process_group()


def _reference(world, accum):
    net = _net()
    opt = torch.optim.AdamW(net.parameters(), lr=1e-2)
    step = 0
    for it in range(3):
        for k in range(accum):
            _, _, x, y = _data(step, 0, world)
            step += 1
            # mean over the global batch == average over ranks of per-rank means (equal shard sizes)
            (nn.functional.mse_loss(net(x), y) / accum).backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
        opt.step()
        opt.zero_grad()
    return [p.detach().clone() for p in net.parameters()]


@pytest.mark.parametrize("accum", [1, 2])
def test_two_rank_gloo_matches_single_process(accum):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, accum, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for a, b in zip(res[0], res[1]):
        assert (a == b).all(), "ranks diverged"
    for a, r in zip(res[0], _reference(world, accum)):
        assert torch.allclose(torch.from_numpy(a), r, atol=1e-6)


def _bucket_worker(rank, world, port, q):
    """Engine path: add_bucket() hands back views of the flat bucket that hold the averaged gradients after finish();
    with gradient accumulation every micro-step is averaged and the views are accumulated like ordinary gradients."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sd3_amd  # noqa: F401
    from sd3_amd.reducer import GradReducer
    red = GradReducer()
    acc = None
    for micro in range(2):
        g = torch.Generator().manual_seed(10 * micro + rank)
        grads = [torch.randn(5, 7, generator=g), None, torch.randn(11, generator=g), torch.randn(3, 4, 2, generator=g)]
        views = red.add_bucket(grads)
        more = red.add_bucket([torch.full((6,), float(rank + micro))])      # a second bucket in flight
        arena = torch.arange(10, dtype=torch.float32) * (rank + 1)          # gradients that already live in one flat buffer:
        inplace = [arena[:4].view(2, 2), arena[4:]]                        # averaged where they are, nothing handed back
        assert red.add_bucket(inplace, arenas=[arena]) is None
        red.finish()
        assert torch.allclose(arena, torch.arange(10, dtype=torch.float32) * 1.5) and torch.equal(inplace[1], arena[4:])
        assert len(views) == 3 and [v.shape for v in views] == [grads[0].shape, grads[2].shape, grads[3].shape]
        cur = views + more
        acc = cur if acc is None else [a + c for a, c in zip(acc, cur)]
    q.put((rank, [a.numpy().copy() for a in acc]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_bucket_views():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = None
    for micro in range(2):
        per_rank = []
        for rank in range(world):
            g = torch.Generator().manual_seed(10 * micro + rank)
            per_rank.append([torch.randn(5, 7, generator=g), torch.randn(11, generator=g), torch.randn(3, 4, 2, generator=g),
                             torch.full((6,), float(rank + micro))])
        mean = [sum(ts) / world for ts in zip(*per_rank)]
        want = mean if want is None else [w + m for w, m in zip(want, mean)]
    for rank in range(world):
        for a, w in zip(res[rank], want):
            assert torch.allclose(torch.from_numpy(a), w, atol=1e-6)


# ---------------------------------------------------------------------------------------------------------------------
# The TRAINER's micro-step / accumulation logic (model_trainer.micro_step) on the engine path, with a stand-in module that
# behaves like diff_model towards the trainer: it owns a `grad_reducer` slot and hands the current micro-step's gradients to
# reducer.add_bucket() from inside backward (engine.model_bwd's on_grads), using the returned views as its gradients.
class _EngineLikeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, w, b):
        ctx.net = net
        ctx.save_for_backward(x, w)
        return x.flatten(1) @ w + b

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        gw, gb = x.flatten(1).t() @ dy, dy.sum(0)
        red = ctx.net.grad_reducer
        if red is not None:
            repl = red.add_bucket([gw, gb])
            if repl is not None:
                gw, gb = repl
        return None, None, gw, gb


class _EngineLike(nn.Module):
    inCh, class_dim, wandb_id, start_step = 2, 8, None, 0

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        self.w = nn.Parameter(torch.randn(2 * 4 * 4, 2 * 4 * 4, generator=g) * 0.1)
        self.b = nn.Parameter(torch.zeros(2 * 4 * 4))
        self.device, self.dev, self.grad_reducer = torch.device("cpu"), "cpu", None
        self.calls = 0

    def noise_batch(self, X, t):
        eps = torch.ones_like(X) * 0.5
        t = t[:, None, None, None]
        return (1 - t) * X + t * eps, eps

    def forward(self, x_t, t, c, cp, *masks):
        return _EngineLikeFn.apply(self, x_t, self.w, self.b).view(x_t.shape)


def _trainer_data(step, rank, world, per):
    g = torch.Generator().manual_seed(500 + step)
    x = torch.randn((per * world, 2, 4, 4), generator=g)
    t = torch.sigmoid(torch.randn((per * world,), generator=g))
    sl = slice(rank * per, (rank + 1) * per)
    return x[sl], t[sl]


def _make_trainer(net, accum, per):
    from sd3_amd.model_trainer import model_trainer
    tr = model_trainer(net, batchSize=per, accumulation_steps=accum, totalSteps=10, lr=1e-2, ema_update_freq=10, ema_decay=0.9, warmup_steps=0,
                       use_lr_scheduler=False, device=torch.device("cpu"), saveDir="/tmp/_gl", numSaveSteps=100, use_amp=False, max_res=32,
                       use_ema=False, hip_optimizer=False, async_checkpoint=False)
    return tr


def _trainer_worker(rank, world, port, accum, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sd3_amd  # noqa: F401
    net = _EngineLike()
    tr = _make_trainer(net, accum, 3)
    assert tr.reducer.enabled and net.grad_reducer is tr.reducer
    n_reduce = [0]
    real = dist.all_reduce

    def counting(*a, **k):
        n_reduce[0] += 1
        return real(*a, **k)

    dist.all_reduce = counting
    micro = [0]

    def source():
        x, t = _trainer_data(micro[0], rank, world, 3)
        tr._t = t
        micro[0] += 1
        return x, torch.zeros(3, 154, 4), torch.zeros(3, 8)

    tr.data_source = source
    tr._sample_conditioning = lambda n: (tr._t, None, None, None)
    losses = [float(tr.train_step(s + 1)) for s in range(3)]
    q.put((rank, [p.detach().numpy().copy() for p in net.parameters()], losses, n_reduce[0]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("accum", [1, 2])
def test_trainer_accumulation_reduces_once_per_optimizer_step(accum):
    """model_trainer on the engine path, 2 gloo ranks: ranks stay bit-identical, match a single process that sees the whole
    batch, and with accumulation the collective runs once per OPTIMIZER step (DDP.no_sync, reference model_trainer.py:463-480)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_trainer_worker, args=(r, world, port, accum, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r[0]: r[1:] for r in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for a, b in zip(res[0][0], res[1][0]):
        assert (a == b).all(), "ranks diverged"
    # one bucket per optimizer step (the late path flushes one flat bucket; the overlapped path one bucket per add_bucket call)
    assert res[0][2] == 3, res[0][2]
    # single process, whole batch
    sys.path.insert(0, ROOT)
    import sd3_amd  # noqa: F401
    net = _EngineLike()
    tr = _make_trainer(net, accum, 3 * world)
    micro = [0]

    def source():
        x, t = _trainer_data(micro[0], 0, 1, 3 * world)
        tr._t = t
        micro[0] += 1
        return x, torch.zeros(3 * world, 154, 4), torch.zeros(3 * world, 8)

    tr.data_source = source
    tr._sample_conditioning = lambda n: (tr._t, None, None, None)
    losses = [float(tr.train_step(s + 1)) for s in range(3)]
    for a, p in zip(res[0][0], net.parameters()):
        assert torch.allclose(torch.from_numpy(a), p.detach(), atol=1e-6)
    assert abs(0.5 * (res[0][1][0] + res[1][1][0]) - losses[0]) < 1e-6


# ---------------------------------------------------------------------------------------------------------------------
# The selectable bucket algorithms (reducer.ALGORITHMS): RCCL's all-reduce, reduce-scatter + all-gather, and the "direct" one
# (reduce-scatter and all-gather as one all-to-all each, local fp32 sum of the received shards, optional bf16 wire format).
def _algo_worker(rank, world, port, algo, wire, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sd3_amd  # noqa: F401
    from sd3_amd.reducer import GradReducer
    red = GradReducer(algorithm=algo, wire_dtype=torch.bfloat16 if wire == "bf16" else None)
    g = torch.Generator().manual_seed(40 + rank)
    arena = torch.randn(2048, generator=g)                     # an engine arena (a multiple of engine.ARENA_QUANTUM): averaged in place
    views = [arena[:1000].view(10, 100), arena[1000:2040]]
    assert red.add_bucket(views, arenas=[arena]) is None
    odd = [torch.randn(5, 7, generator=g), torch.randn(11, generator=g)]     # 46 elements + padding to the world size inside the reducer
    repl = red.add_bucket(odd)
    tiny = torch.randn(3, generator=g)                          # 3 elements, world 2: not divisible -> that bucket falls back to all-reduce
    assert red.add_bucket([tiny[:2], tiny[2:]], arenas=[tiny]) is None
    red.finish()
    assert red.buckets == 3
    q.put((rank, [arena.numpy().copy(), repl[0].numpy().copy(), repl[1].numpy().copy(), tiny.numpy().copy()]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("algo,wire", [("allreduce", "f32"), ("rs_ag", "f32"), ("direct", "f32"), ("direct", "bf16")])
def test_two_rank_gloo_bucket_algorithms(algo, wire):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_algo_worker, args=(r, world, port, algo, wire, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    per_rank = []
    for rank in range(world):
        g = torch.Generator().manual_seed(40 + rank)
        per_rank.append([torch.randn(2048, generator=g), torch.randn(5, 7, generator=g), torch.randn(11, generator=g), torch.randn(3, generator=g)])
    want = [(a + b) / 2 for a, b in zip(*per_rank)]
    for i, (a, b, w) in enumerate(zip(res[0], res[1], want)):
        assert (a == b).all(), "ranks diverged"
        got = torch.from_numpy(a)
        if wire == "f32" or i == 3:      # fp32 wire: the two-rank mean is exact in every algorithm (one addition, one halving)
            assert torch.equal(got, w), (algo, i)
        else:                             # bf16 wire: operands and the gathered mean are bf16-rounded (2^-9 relative each)
            assert torch.allclose(got, w, rtol=2 ** -7, atol=2e-2) and not torch.equal(got, w)


def test_engine_arenas_are_shardable():
    """The flat gradient arenas the engine hands to the reducer are multiples of ARENA_QUANTUM elements (equal shards for every world
    size that divides it), also the prefix of the small-gradient arena."""
    import sd3_amd  # noqa: F401
    from sd3_amd import engine
    views, prefix = engine._zeros_views([(7,), (130,), (64,), (3, 5)], torch.device("cpu"), prefix=3)
    assert prefix.numel() % engine.ARENA_QUANTUM == 0 and prefix.numel() >= 7 + 130 + 64
    assert views[3].data_ptr() == prefix.data_ptr() + 4 * prefix.numel()       # scratch views start right behind the padded prefix
    assert all(float(v.abs().sum()) == 0 for v in views)
    assert engine.ARENA_QUANTUM % 8 == 0


# ---------------------------------------------------------------------------------------------------------------------
# Eight ranks (the node size the driver's SCALE run uses), arenas with the MMDiT-B block's REAL gradient element counts.
def _b_block_arena_sizes():
    """Per-block flat gradient arena sizes of MMDiT-B as the engine lays them out (engine.block_bwd: the block's weight gradients in one
    arena padded to engine.ARENA_QUANTUM elements), for a middle block and the last block (no text MLP / out-projection / gates)."""
    from oracle.weights import state_dict_spec
    import sd3_amd  # noqa: F401
    from sd3_amd import engine
    spec = state_dict_spec(dim=768, num_heads=12, num_blocks=12)
    out = []
    for blk in (5, 11):
        n = sum(int(np.prod(shape)) for name, shape in spec if name.startswith(f"blocks.{blk}.") and not name.endswith("freqs"))
        out.append(-(-n // engine.ARENA_QUANTUM) * engine.ARENA_QUANTUM)
    return out


def _eight_worker(rank, world, port, algo, wire, sizes, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sd3_amd  # noqa: F401
    from sd3_amd.reducer import GradReducer
    red = GradReducer(algorithm=algo, wire_dtype=torch.bfloat16 if wire == "bf16" else None)
    g = torch.Generator().manual_seed(900 + rank)
    arenas = [torch.randn(n, generator=g) for n in sizes]
    for a in arenas:                       # one bucket per block, handed over as the backward reaches it
        assert red.add_bucket([a[: a.numel() // 2], a[a.numel() // 2:]], arenas=[a]) is None
    odd = torch.randn(1001, generator=g)   # not a multiple of 8: that bucket falls back to all-reduce inside any algorithm
    assert red.add_bucket([odd], arenas=[odd]) is None
    red.finish()
    assert red.buckets == len(sizes) + 1
    h = [float(a.double().sum()) for a in arenas + [odd]] + [float(a.double().abs().sum()) for a in arenas + [odd]]
    q.put((rank, h, arenas[0][:4096].numpy().copy(), odd.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("algo,wire", [("allreduce", "f32"), ("rs_ag", "f32"), ("direct", "f32"), ("direct", "bf16")])
def test_eight_rank_gloo_bucket_algorithms_on_b_block_arenas(algo, wire):
    """world_size 8 (one node of MI355X): every bucket algorithm on arenas whose lengths have the divisibility of the real MMDiT-B
    block arenas (the real element counts -- 7.9 M for a middle block, 4.3 M for the last -- scaled down 64x in bytes: the counts are
    multiples of ARENA_QUANTUM = 1024, the scaled ones keep (count / 1024) mod 8, i.e. the same shard geometry for 8 ranks), plus
    a bucket that 8 does not divide.  All ranks end bit-identical; fp32 wire equals the fp64 mean to fp32 rounding; bf16 wire to 2^-7."""
    real = _b_block_arena_sizes()
    assert all(n % 1024 == 0 for n in real) and real[0] > 7_000_000 and real[1] > 4_000_000, real
    sizes = [1024 * ((n // 1024) % 8 + 8 * ((n // 1024) // 512)) for n in real]        # ~64x smaller, same (n / 1024) mod 8
    assert all(s % 8 == 0 and (s // 1024) % 8 == (n // 1024) % 8 for s, n in zip(sizes, real))
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_eight_worker, args=(r, world, port, algo, wire, sizes, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r[0]: r[1:] for r in (q.get(timeout=300) for _ in range(world))}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in range(1, world):
        assert res[r][0] == res[0][0], "ranks diverged (checksums)"
        assert (res[r][1] == res[0][1]).all() and (res[r][2] == res[0][2]).all(), "ranks diverged"
    gens = [torch.Generator().manual_seed(900 + r) for r in range(world)]
    per = [[torch.randn(n, generator=g) for n in sizes] + [torch.randn(1001, generator=g)] for g in gens]
    want0 = torch.stack([p[0][:4096].double() for p in per]).mean(0)
    want_odd = torch.stack([p[-1].double() for p in per]).mean(0)
    got0, got_odd = torch.from_numpy(res[0][1]).double(), torch.from_numpy(res[0][2]).double()
    tol = 2 ** -7 if wire == "bf16" else 1e-6
    assert float((got0 - want0).abs().max()) <= tol * float(want0.abs().max()) + 1e-7
    assert float((got_odd - want_odd).abs().max()) <= 1e-6 * float(want_odd.abs().max()) + 1e-7      # (the fallback bucket is always fp32 all-reduce)


# ---------------------------------------------------------------------------------------------------------------------
# The launch mode of a data-parallel run is a collective decision (model_trainer.capture_graph_agreed; bench.py and train() use it).
def _agree_worker(rank, world, port, fail_rank, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sd3_amd  # noqa: F401
    net = _EngineLike()
    tr = _make_trainer(net, 1, 3)
    micro = [0]

    def source():
        x, t = _trainer_data(micro[0], rank, world, 3)
        tr._t = t
        micro[0] += 1
        return x, torch.zeros(3, 154, 4), torch.zeros(3, 8)

    tr.data_source = source
    tr._sample_conditioning = lambda n: (tr._t, None, None, None)
    tr.train_step(1)
    # a stand-in for the GPU capture (no GPU here): succeeds by installing a graph object, or raises on the chosen rank
    tr.can_capture = lambda: None
    sentinel = object()

    def fake_capture(step):
        if rank == fail_rank:
            raise RuntimeError("hipErrorStreamCaptureInvalidated (simulated)")
        tr._graph = sentinel
        return sentinel

    tr.capture_graph = fake_capture
    agreed = tr.capture_graph_agreed(2)
    kept = tr._graph is sentinel
    if fail_rank is None:
        # second half of the decision: the replay check.  Every rank fine -> the graph stays; rank 1 reports a bad replay -> BOTH drop it
        assert tr.keep_graph_if_agreed(True) and tr._graph is sentinel
        assert not tr.keep_graph_if_agreed(rank != 1) and tr._graph is None
        assert not tr.keep_graph_if_agreed(True)          # (no graph: nothing to keep, and no collective is issued)
        tr._graph = sentinel if kept else None
    strict_raised = False
    if fail_rank is not None:
        tr._graph = None
        try:                       # strict: the failing rank raises -- AFTER the agreement, so nobody is left waiting in the all-reduce
            tr.capture_graph_agreed(2, strict=True)
        except RuntimeError:
            strict_raised = True
        assert tr._graph is None
    tr._graph = None               # (the sentinel cannot replay) -- the eager path must still work on every rank, in step
    losses = [float(tr.train_step(s)) for s in (2, 3)]
    q.put((rank, agreed, kept, strict_raised, losses, [p.detach().numpy().copy() for p in net.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_rank", [None, 1])
def test_launch_mode_is_agreed_across_ranks_and_falls_back_to_eager(fail_rank):
    """2 gloo ranks: when the capture fails on one rank (simulated), capture_graph_agreed() returns False on BOTH, the rank whose
    capture succeeded drops its graph, nobody hangs, strict mode raises only on the failing rank and only after the agreement, and
    the eager steps that follow keep the ranks bit-identical; when it succeeds everywhere, both keep their graph."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_agree_worker, args=(r, world, port, fail_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r[0]: r[1:] for r in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        agreed, kept, strict_raised, losses, _ = res[r]
        assert agreed == (fail_rank is None) and kept == (fail_rank is None)
        assert strict_raised == (fail_rank == r)
        assert all(l == l for l in losses)
    for a, b in zip(res[0][4], res[1][4]):
        assert (a == b).all(), "ranks diverged after the fallback"


# ---------------------------------------------------------------------------------------------------------------------
# The data-parallel settings nobody could measure in advance are chosen by a warm-up A/B whose result is a COLLECTIVE decision
# (model_trainer.vote_fastest / autotune_reducer; bench.py and train() call it before the step is captured).
def _autotune_worker(rank, world, port, q):
    import time
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sd3_amd  # noqa: F401
    from sd3_amd.model_trainer import vote_fastest
    # the vote alone: rank r finds candidate (r mod 3) fastest, but candidate 1 has the smallest WORST time over the ranks
    local = [10.0 + 5.0 * ((rank + k) % 3) for k in range(3)]
    local[1] = 12.0 + 0.1 * rank
    best, worst = vote_fastest(local)
    net = _EngineLike()
    tr = _make_trainer(net, 1, 3)
    micro = [0]

    def source():
        x, t = _trainer_data(micro[0], rank, world, 3)
        tr._t = t
        micro[0] += 1
        return x, torch.zeros(3, 154, 4), torch.zeros(3, 8)

    tr.data_source = source
    tr._sample_conditioning = lambda n: (tr._t, None, None, None)
    tr.train_step(1)
    # the whole A/B on the CPU trainer: every step is a real optimizer step; a rank- and setting-dependent sleep stands in for the link / CU effects.
    # rs_ag is slowest on rank 0 only, direct is uniformly middling, allreduce is fastest on most ranks but very slow on rank 5 -> direct must win;
    # reserves: 16 is best by the max over ranks although rank 2 prefers 0
    algo_ms = {"allreduce": 2.0 + (30.0 if rank == 5 else 0.0), "rs_ag": 6.0 + (40.0 if rank == 0 else 0.0), "direct": 12.0}
    res_ms = {0: 9.0 - (6.0 if rank == 2 else 0.0), 16: 4.0, 32: 8.0}

    def run(step):
        loss = tr.train_step(step)
        time.sleep((algo_ms[tr.reducer.algorithm] + res_ms[tr.reserved_cus]) * 1e-3)
        return loss

    tr.reserved_cus = 32
    step, table = tr.autotune_reducer(1, steps_each=2, run_step=run)
    losses = [float(tr.train_step(s)) for s in (step + 1, step + 2)]      # the ranks go on in step with the chosen setting
    q.put((rank, best, worst, step, table, tr.reducer.algorithm, tr.reserved_cus, losses, [p.detach().numpy().copy() for p in net.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_gloo_reducer_autotune_is_a_collective_decision():
    """world_size 8: vote_fastest returns the same winner (the smallest maximum over the ranks, not anybody's local favourite) and the same table on every
    rank; autotune_reducer -- two timed blocks of two optimizer steps per candidate (the faster block counts) on the CPU trainer, three algorithms then three reserves -- leaves all eight ranks on
    the same algorithm and reserve, counts the steps it ran, and the replicas stay bit-identical through and after it."""
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_autotune_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r[0]: r[1:] for r in (q.get(timeout=240) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    best0, worst0, step0, table0 = res[0][:4]
    assert best0 == 1 and abs(worst0[1] - 12.7) < 1e-9 and worst0[0] == 20.0 and worst0[2] == 20.0
    for r in range(world):
        best, worst, step, table, algo, reserve, losses, params = res[r]
        assert (best, worst, step, table) == (best0, worst0, step0, table0), "every rank holds the same table and winner"
        assert algo == "direct" and reserve == 16 and table["chosen"] == {"algorithm": "direct", "reserved_cus": 16, "blocks_per_fork": table["chosen"]["blocks_per_fork"], "steps_each": 2}
        assert table["chosen"]["blocks_per_fork"] in (1, 2, 3) and set(table["blocks_per_fork"]) == {"1", "2", "3"}
        assert step == 1 + 3 * 5 + 3 * 5 + 3 * 5          # (1 settling + 2 blocks of 2 timed steps for each of 3 algorithms, 3 reserves and 3 fork widths)
        assert all(l == l for l in losses)
        for a, b in zip(params, res[0][7]):
            assert (a == b).all(), "ranks diverged"


def _train_loop_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sd3_amd  # noqa: F401
    from sd3_amd.model_trainer import model_trainer
    net = _EngineLike()
    tr = model_trainer(net, batchSize=3, accumulation_steps=1, totalSteps=64, lr=1e-2, ema_update_freq=1000, ema_decay=0.9, warmup_steps=0, use_lr_scheduler=False,
                       device=torch.device("cpu"), saveDir="/tmp/_gl_train", numSaveSteps=10 ** 6, use_amp=False, max_res=32, use_ema=False, hip_optimizer=False,
                       async_checkpoint=False, log_steps=10 ** 6)
    assert tr.autotune and tr.reducer.enabled
    micro = [0]

    def source():
        x, t = _trainer_data(micro[0], rank, world, 3)
        tr._t = t
        micro[0] += 1
        return x, torch.zeros(3, 154, 4), torch.zeros(3, 8)

    tr.data_source = source
    tr._sample_conditioning = lambda n: (tr._t, None, None, None)
    tr.keep_losses = True
    tr.train()
    q.put((rank, micro[0], len(tr.loss_history), tr.autotune_table, tr.reducer.algorithm, tr.reserved_cus, [float(l) for l in tr.loss_history[-3:]],
           [p.detach().numpy().copy() for p in net.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_train_loop_with_the_reducer_autotune():
    """model_trainer.train() on 2 gloo ranks (the CPU trainer): after its first step the loop runs autotune_reducer -- (1 + 2 + 2) optimizer steps for each of three bucket
    algorithms, three reserves and three fork widths, real training steps on fresh batches -- and goes on; exactly totalSteps batches are consumed and totalSteps losses recorded, both ranks
    hold the same table and setting, and the replicas end bit-identical."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_loop_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r[0]: r[1:] for r in (q.get(timeout=240) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        batches, nloss, table, algo, reserve, last, params = res[r]
        assert batches == 64 and nloss == 64, (batches, nloss)
        assert table is not None and table == res[0][2] and set(table["algorithm"]) == {"allreduce", "rs_ag", "direct"} and set(table["reserved_cus"]) == {"0", "16", "32"}
        assert table["chosen"]["algorithm"] == algo and table["chosen"]["reserved_cus"] == reserve and (algo, reserve) == (res[0][3], res[0][4])
        assert all(l == l for l in last)
        for a, b in zip(params, res[0][6]):
            assert (a == b).all(), "ranks diverged"


def _fork_worker(rank, world, port, k, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sd3_amd  # noqa: F401
    from sd3_amd.reducer import GradReducer
    red = GradReducer(blocks_per_fork=k)
    g = torch.Generator().manual_seed(40 + rank)
    arenas = [torch.randn(2048, generator=g) for _ in range(5)]
    launched = []
    for i, a in enumerate(arenas):
        assert red.add_bucket([a[:1024], a[1024:]], arenas=[a]) is None
        launched.append(red.buckets)                     # collectives issued so far: held buckets are not
    tail = torch.randn(333, generator=g)
    views = red.add_bucket([tail])                        # the last, gathered bucket: everything held goes first
    red.finish()
    q.put((rank, launched, red.buckets, [a.clone() for a in arenas], views[0].clone()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("k", [1, 2, 3])
def test_two_rank_gloo_blocks_per_fork(k):
    """GradReducer(blocks_per_fork=k): block arenas are handed to the collective stream k at a time (one fork of the captured step per hand-over instead of one per
    block); whatever is still held goes out before a gathered bucket and at finish().  Same averages for every k, every bucket reduced exactly once."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fork_worker, args=(r, world, port, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r[0]: r[1:] for r in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_launched = [(i + 1) // k * k for i in range(5)]
    gens = [torch.Generator().manual_seed(40 + r) for r in range(world)]
    local = [[torch.randn(2048, generator=g) for _ in range(5)] + [torch.randn(333, generator=g)] for g in gens]
    mean = [(local[0][i].double() + local[1][i].double()) / 2 for i in range(6)]
    for r in range(world):
        launched, total, arenas, tail = res[r]
        assert launched == want_launched and total == 6
        for a, m in zip(arenas, mean[:5]):
            assert float((a.double() - m).abs().max()) <= 1e-6
        assert float((tail.double() - mean[5]).abs().max()) <= 1e-6
