"""hipGraph replay of the optimizer step (model_trainer.capture_graph) IS the eager step -- at MMDiT-B size with the replay pattern that
lost the loss in round 2, with the gradient collectives inside the graph (1-rank RCCL group), with static input slots (any data
source), and across the eager -> replay -> sample (fp8 weight caches) -> replay boundary.

Root cause pinned by test_b_size_replay_with_synchronize_between_replays ("final_loss 0.0"): a hipMemsetAsync captured
into a hipGraph is not kept in stream order by the ROCm 7 runtime when the graph is launched on an idle stream
(tools/probes/graph_memset_order.py); torch's multi-block mean() zeroes a 4-byte semaphore that way, in a block of the graph's private
pool that earlier temporaries of the same step reuse, so the last-block election failed and the mean was never written.  The loss is
now produced by ops.flow_loss (no workspace to clear) and leaves the graph through a persistent buffer."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.weights import make_inputs, make_state_dict  # noqa: E402

CONFIGS = {"micro": dict(dim=128, num_heads=2, num_blocks=3), "b": dict(dim=768, num_heads=12, num_blocks=12)}


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


class _CpuSource:
    """A data source that is NOT the trainer's SyntheticData: batches drawn on the CPU and moved (what a loader would hand over)."""

    def __init__(self, batch, hw, seed):
        self.g, self.batch, self.hw = torch.Generator().manual_seed(seed), batch, hw

    def __call__(self):
        x0 = torch.randn((self.batch, 16, self.hw, self.hw), generator=self.g).to(torch.bfloat16)
        c = torch.randn((self.batch, 154, 2304), generator=self.g)
        c[:, :77] *= 30.0
        c[:, 77:, 1024:] = 0
        cp = torch.randn((self.batch, 768), generator=self.g).to(torch.bfloat16)
        return x0.cuda(), c.to(torch.bfloat16).cuda(), cp.cuda()


def graph_vs_eager(cname, batch, max_res, n_steps=3, pattern="each", force_reducer=False, slots=False, lr=1e-3, accum=1, keep_graph=False, hip_loss=True,
                   drop_after=None):
    """One trainer: three eager warm-up steps, snapshot of everything (parameters, AdamW state, loss scale, every RNG stream),
    n_steps eager; restore; capture; n_steps replayed with the given host pattern between replays.  Returns (eager losses,
    replayed losses, eager final parameters, replayed final parameters, trainer)."""
    import sd3_amd  # noqa: F401
    from sd3_amd import engine, packing
    from sd3_amd.model_trainer import model_trainer
    from sd3_amd.models.diff_model import diff_model

    overlap = engine._WG_OVERLAP
    try:
        engine._WG_OVERLAP = False
        torch.manual_seed(0)
        dev = torch.device("cuda:0")
        net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu",
                         device=dev, positional_encoding="RoPE2d", checkpoint_MLP=False, checkpoint_attn=False, **CONFIGS[cname])
        net.load_state_dict(make_state_dict(0, **CONFIGS[cname]))
        src = _CpuSource(batch, max_res // 8, 77) if slots else None
        tr = model_trainer(net, batchSize=batch, accumulation_steps=accum, totalSteps=100, lr=lr, ema_update_freq=1, ema_decay=0.9, warmup_steps=8,
                           use_lr_scheduler=False, device=dev, saveDir="/tmp/_t", numSaveSteps=100, null_prob_pooled=0.1,
                           null_prob_gemma=0.316, null_prob_bert=0.316, max_res=max_res, device_rng=not slots, use_ema=False,
                           force_reducer=force_reducer, data_source=src, hip_loss=hip_loss)
        tr.keep_graph = keep_graph
        assert tr.reducer.enabled == force_reducer
        for s in (1, 2, 3):
            tr.train_step(s)
        torch.cuda.synchronize()
        params = [p for p in net.parameters()]
        snap_p = [p.detach().clone() for p in params]
        snap_o = {id(p): {k: v.clone() for k, v in tr.optim.state[p].items()} for p in params if p in tr.optim.state}
        snap_s = (tr.grad_scaler._scale.clone(), tr.grad_scaler._growth_tracker.clone())

        def rng_get():
            return (torch.get_rng_state(), torch.cuda.get_rng_state(), tr._gen.get_state() if tr._gen is not None else None,
                    (src.g if slots else tr.data_source.g).get_state())

        def rng_set(st):
            torch.set_rng_state(st[0])
            torch.cuda.set_rng_state(st[1])
            if tr._gen is not None:
                tr._gen.set_state(st[2])
            (src.g if slots else tr.data_source.g).set_state(st[3])

        snap_r = rng_get()

        def steps():
            out = []
            for s in range(4, 4 + n_steps):
                if drop_after is not None and tr._graph is not None and s - 4 == drop_after:
                    # the ranks vote the graph down (model_trainer.keep_graph_if_agreed): a REAL graph is dropped, eager launches continue
                    assert tr.keep_graph_if_agreed(False) is False and tr._graph is None and tr._slots is None
                l = tr.train_step(s)
                if pattern == "each":
                    out.append(float(l))
                elif pattern == "sync":      # the host waits for the GPU between replays and reads nothing: the graph is launched on an idle stream
                    torch.cuda.synchronize()
                    out.append(l.clone())
                else:                        # burst: replays back to back
                    out.append(l.clone())
            torch.cuda.synchronize()
            return [float(x) for x in out], [p.detach().clone() for p in params]

        l0, p0 = steps()
        with torch.no_grad():
            for p, q in zip(params, snap_p):
                p.copy_(q)
                for k, v in snap_o.get(id(p), {}).items():
                    tr.optim.state[p][k].copy_(v)
            tr.grad_scaler._scale.copy_(snap_s[0])
            tr.grad_scaler._growth_tracker.copy_(snap_s[1])
        packing.bump_epoch()                       # the parameters were rewritten: their bf16 operand copies are stale
        tr.scheduler.step(3)
        hw = max_res // 8
        x, c, cp = [a.cuda() for a in make_inputs(3, 2, hw, hw)]
        net(x, torch.tensor([0.3, 0.7]), c, cp).sum().backward()      # (one eager pass refreshes the bf16 copies outside the capture)
        tr.optim.zero_grad()
        rng_set(snap_r)
        # (static slots: the capture draws one batch per micro-step to fill its input slots; the first replay trains on exactly those,
        #  so the data / RNG order of an eager-then-replayed run is the eager run's -- nothing drawn is dropped)
        tr.capture_graph(4)
        assert tr._graph is not None
        l1, p1 = steps()
    finally:
        engine._WG_OVERLAP = overlap
    return l0, l1, p0, p1, tr


def check(l0, l1, p0, p1, tag):
    print(f"[graph {tag}] eager losses {l0}  replayed {l1}")
    # step 4: same parameters, same batch, deterministic forward and loss reduction -> bit-identical; later steps to the run-to-run
    # noise of the backward (fp32 atomic column sums, tools/probes/determinism.py)
    assert l0[0] == l1[0] and np.allclose(l0, l1, rtol=1e-3), (l0, l1)
    assert all(1e-3 < x < 10 for x in l1)
    for a, b in zip(p0, p1):
        assert rel(a, b) < 1e-3 and float((a - b).abs().max()) <= 2e-3 * float(b.abs().max()) + 1e-6


def test_b_size_replay_with_synchronize_between_replays():
    """MMDiT-B, batch 64 (the bench's step), the host pattern that printed `final_loss: 0.0` in round 2: a device synchronisation
    after every replay and one read at the end."""
    l0, l1, p0, p1, tr = graph_vs_eager("b", 64, 256, n_steps=4, pattern="sync", lr=1e-4)
    check(l0, l1, p0, p1, "B sync")
    assert float(tr.last_loss) == l1[-1]


def test_captured_step_has_no_memset_nodes():
    """The hazard behind round 2's lost loss is structural: any hipMemsetAsync inside the captured step is a memset NODE, and those do
    not reliably keep their stream order on replay.  The captured step (HIP loss) must contain kernel and memcpy nodes only; with the
    torch loss expression (hip_loss=False) torch's multi-block mean() brings one in -- which is how this test knows it can see them."""
    _, _, _, _, tr = graph_vs_eager("b", 64, 256, n_steps=1, pattern="each", lr=1e-4, keep_graph=True)
    hist = tr.graph_node_types()
    print(f"[graph nodes] MMDiT-B step, HIP loss: {hist}")
    assert hist.get("kernel", 0) > 300 and hist.get("memset", 0) == 0, hist
    _, _, _, _, tr = graph_vs_eager("b", 64, 256, n_steps=1, pattern="each", lr=1e-4, keep_graph=True, hip_loss=False)
    hist2 = tr.graph_node_types()
    print(f"[graph nodes] MMDiT-B step, torch loss expression: {hist2}")
    assert hist2.get("memset", 0) >= 1, hist2


def test_b_size_replay_burst():
    l0, l1, p0, p1, _ = graph_vs_eager("b", 64, 256, n_steps=4, pattern="burst", lr=1e-4)
    check(l0, l1, p0, p1, "B burst")


def test_b_size_replay_with_tile_claiming():
    """The bench's step with dynamic tile claiming on (csrc/gemm8p.hip; what model_trainer switches on when gradients are reduced): ~50 claimed launches per step are
    captured with their scheduler slots and replayed; eager and replayed steps agree as they do with the static walk, and the workspace's ticket / scheduler words are
    zero after the last replay (every launch leaves its queue heads clean)."""
    import sd3_amd  # noqa: F401
    from sd3_amd import _lib, ops
    L = _lib.lib()
    before = L.mmdit_gemm_get_claiming()
    assert L.mmdit_gemm_set_claiming(1) == 0
    try:
        l0, l1, p0, p1, _ = graph_vs_eager("b", 64, 256, n_steps=3, pattern="burst", lr=1e-4)
        check(l0, l1, p0, p1, "B burst, tile claiming")
        torch.cuda.synchronize()
        ws = ops._GEMM_WS[torch.device("cuda", 0)]
        assert int(ws[:8192].view(torch.int32).abs().sum()) == 0
    finally:
        assert L.mmdit_gemm_set_claiming(before) == 0


def test_replay_with_static_input_slots_and_cpu_conditioning():
    """Any data source + the reference's CPU draw of t / null masks (device_rng=False): inputs are copied into static slots before
    every replay; two accumulation micro-steps per optimizer step."""
    l0, l1, p0, p1, tr = graph_vs_eager("micro", 4, 128, n_steps=3, pattern="each", slots=True, accum=2)
    assert tr._slots is not None and len(tr._slots) == 2
    check(l0, l1, p0, p1, "micro slots accum2")


def test_replay_with_other_bucket_shape_runs_eager():
    """A batch whose shapes differ from the captured ones (aspect-ratio buckets) is an eager step; the graph stays for its shape."""
    _, _, _, _, tr = graph_vs_eager("micro", 4, 128, n_steps=1, pattern="each", slots=True)
    src = tr.data_source
    tr.data_source = _CpuSource(4, 12, 5)           # 96^2 images: 12x12 latents instead of 16x16
    before = [p.detach().clone() for p in tr.model.parameters()]
    l = float(tr.train_step(9))
    assert 1e-3 < l < 10 and tr._graph is not None
    assert any(not torch.equal(a, b) for a, b in zip(before, tr.model.parameters()))
    tr.data_source = src
    assert 1e-3 < float(tr.train_step(10)) < 10


def test_replay_with_gradient_collectives_one_rank_rccl():
    """Data-parallel step captured WITH its collectives: per-block all-reduce on the side stream inside the graph (fork / join edges).
    One-rank RCCL group (the boxes of this pool have one GPU): the same code path as N ranks, and AVG over one rank is the identity,
    so the replayed steps must equal the eager steps of the same trainer."""
    import torch.distributed as dist
    import socket
    with socket.socket() as _s:      # a free rendezvous port (a fixed one can be taken on a shared box)
        _s.bind(("127.0.0.1", 0))
        _port = _s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port), RANK="0", WORLD_SIZE="1")
    # a caller that creates the process group itself must switch the flight recorder on first (model_trainer.init_distributed does):
    # capture_graph() polls its per-group status for the watchdog hand-off instead of sleeping ("Embedding")
    os.environ.setdefault("TORCH_FR_BUFFER_SIZE", "2000")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    tr = None
    try:
        l0, l1, p0, p1, tr = graph_vs_eager("micro", 4, 128, n_steps=3, pattern="burst", force_reducer=True)
        assert tr.reducer.enabled and tr.model.grad_reducer is tr.reducer and tr.reducer.buckets > 0
        assert tr.capture_handoff == "polled", tr.capture_handoff      # (not the fixed-delay heuristic)
    finally:
        if tr is not None:          # the graph holds the captured collectives: release it before the communicator goes away
            tr._graph = None
        import gc
        gc.collect()
        torch.cuda.synchronize()
        dist.destroy_process_group()
    check(l0, l1, p0, p1, "micro 1-rank RCCL")


def test_dropping_a_real_graph_with_collectives_then_eager_steps():
    """the drop-the-graph path on the real thing.  One-rank RCCL group, the step captured WITH its per-block all-reduces; two
    replays, then the second launch-mode vote fails (keep_graph_if_agreed(False)): the hipGraph, its private pool and its registered RNG
    generators are released while the trainer goes on with eager steps -- which must continue the run exactly where an all-eager run
    of the same trainer would be (same data / RNG order, same parameters to the backward's run-to-run noise)."""
    import torch.distributed as dist
    import socket
    with socket.socket() as _s:
        _s.bind(("127.0.0.1", 0))
        _port = _s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("TORCH_FR_BUFFER_SIZE", "2000")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    tr = None
    try:
        # (four steps, as the other replay tests: the backward's fp32 atomics make two runs of the SAME code drift apart step by step,
        #  tools/probes/determinism.py, and check()'s parameter bars are sized for three to four steps)
        l0, l1, p0, p1, tr = graph_vs_eager("micro", 4, 128, n_steps=4, pattern="burst", force_reducer=True, drop_after=2)
        assert tr._graph is None and tr.replayed_steps == 2 and tr.reducer.enabled
        more = [float(tr.train_step(s)) for s in (9, 10)]      # and it keeps training
        assert all(1e-3 < x < 10 for x in more)
    finally:
        if tr is not None:
            tr._graph = None
        import gc
        gc.collect()
        torch.cuda.synchronize()
        dist.destroy_process_group()
    check(l0, l1, p0, p1, "micro 1-rank RCCL, graph dropped after two replays")


def test_graph_dropped_before_its_first_replay_keeps_the_drawn_batches():
    """a capture that the ranks vote down before any replay (capture_graph_agreed's mismatch path) has already drawn one batch
    per micro-step into the static input slots.  Those batches are the next eager step's inputs -- nothing drawn is dropped -- so the
    run equals the all-eager run step by step (CPU data source + the reference's CPU conditioning draw, two accumulation micro-steps)."""
    l0, l1, p0, p1, tr = graph_vs_eager("micro", 4, 128, n_steps=3, pattern="each", slots=True, accum=2, drop_after=0)
    assert tr._graph is None and tr.replayed_steps == 0 and tr._carry_inputs is None
    check(l0, l1, p0, p1, "micro slots accum2, graph dropped before its first replay")


def test_fp8_weight_caches_follow_replays():
    """the captured AdamW rewrites the bf16 weight copies at every replay; the fp8 / mxfp8 weight caches (keyed on the
    copies' generation) must notice.  capture -> replay -> sample mxfp8 -> replay -> sample again == fresh model with the same
    parameters."""
    import sd3_amd  # noqa: F401
    from sd3_amd.models.diff_model import diff_model
    _, _, _, _, tr = graph_vs_eager("micro", 4, 128, n_steps=1, pattern="each")
    net = tr.model
    x, c, cp = [a.cuda() for a in make_inputs(11, 2, 16, 16, text_scale=30.0)]
    t = torch.tensor([0.4, 0.6])
    outs = []
    for k in range(2):
        tr.train_step(20 + k)                       # replay
        net.eval()
        if k == 0:
            net.set_precision("mxfp8")      # (clears the quantised-weight caches)
        else:
            net.precision = "mxfp8"          # (does not: the caches must notice the replay by themselves)
        with torch.no_grad():
            v = net(x, t, c.clone(), cp.clone())
        fresh = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu",
                           device=torch.device("cuda:0"), positional_encoding="RoPE2d", **CONFIGS["micro"])
        fresh.load_state_dict(net.state_dict())
        fresh.set_precision("mxfp8")
        fresh.eval()
        with torch.no_grad():
            vf = fresh(x, t, c.clone(), cp.clone())
        outs.append((v.clone(), vf.clone()))
        net.precision = "fast"
        net.train()
    for k, (v, vf) in enumerate(outs):
        assert torch.equal(v, vf), f"mxfp8 forward after replay {k} used stale quantised weights (rel {rel(v, vf):.3e})"
    assert not torch.equal(outs[0][0], outs[1][0])      # (the replay in between did change the weights)
    net.set_precision("fast")
