"""Data-parallel training over RCCL (backend "nccl" on ROCm), one process per GPU: the real diff_model + model_trainer +
GradReducer, gradients averaged block by block from inside the backward schedule (reference: DDP at model_trainer.py:224,
all-reduce inside backward at :467, torchrun launch runjob_SLURM.sh:37-43).

world = 2 needs two visible GPUs (skipped on the 1-GPU boxes); the same worker also runs with world = 1 (a one-rank nccl group
with the reducer forced on), so that every line of it is exercised on a single MI355X.  Asserted: ranks end bit-identical;
the mean of the per-rank losses equals the loss of ONE process that trains on the concatenated batch, step by step; the
parameters follow that single-process run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = dict(dim=256, num_heads=4, num_blocks=2)     # MMDiT-XS
PER, STEPS, HW = 2, 2, 16


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _global_batch(step, world):
    """The global batch of optimizer step `step` (CPU generator: identical in every process), plus its t / eps."""
    from oracle.weights import make_inputs
    x0, c, cp = make_inputs(700 + step, PER * world, HW, HW, text_scale=30.0)
    g = torch.Generator().manual_seed(900 + step)
    eps = torch.randn(x0.shape, generator=g)
    t = torch.sigmoid(torch.randn((PER * world,), generator=g))
    return x0, c, cp, eps, t


def _train(rank, world, data_world, accum, force):
    """STEPS optimizer steps on rows [rank*n, (rank+1)*n) of the global batches built for `data_world` ranks."""
    import sd3_amd  # noqa: F401
    from oracle.weights import make_state_dict
    from sd3_amd.model_trainer import model_trainer
    from sd3_amd.models.diff_model import diff_model
    dev = torch.device(f"cuda:{rank}")
    net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=dev,
                     positional_encoding="RoPE2d", **CFG)
    net.load_state_dict(make_state_dict(0, **CFG))
    if rank == 1:
        with torch.no_grad():
            net.blocks[0].attn.query_proj_x.weight.add_(0.5)     # diverge on purpose: the initial broadcast must repair it
    n = PER * data_world // world
    tr = model_trainer(net, batchSize=n, accumulation_steps=accum, totalSteps=10, lr=1e-3, ema_update_freq=10, ema_decay=0.9, warmup_steps=0,
                       use_lr_scheduler=False, device=dev, saveDir="/tmp/_mg", numSaveSteps=100, max_res=8 * HW, use_ema=False, force_reducer=force)
    state = {"micro": 0}

    def source():
        x0, c, cp, eps, t = _global_batch(state["micro"], data_world)
        sl = slice(rank * n, (rank + 1) * n)
        state.update(eps=eps[sl].to(dev), t=t[sl].to(dev), micro=state["micro"] + 1)
        return x0[sl].to(dev), c[sl].to(dev), cp[sl].to(dev)

    tr.data_source = source
    tr._sample_conditioning = lambda k: (state["t"], None, None, None)
    net.noise_batch = lambda X, t: ((1 - t)[:, None, None, None] * X + t[:, None, None, None] * state["eps"], state["eps"])
    net.train()
    # first a backward WITHOUT an optimizer step, at the (broadcast) initial parameters: the gradients as the reducer leaves them
    # (averaged over the ranks), compared directly -- a reducer bug cannot hide behind Adam's sign sensitivity
    tr.optim.zero_grad()
    for k in range(accum):
        tr.micro_step(final=(k == accum - 1))
    torch.cuda.synchronize(dev)
    tr.grads = [None if p.grad is None else p.grad.detach().cpu().numpy().copy() for p in net.parameters()]
    tr.optim.zero_grad()
    losses = [float(tr.train_step(s + 1)) for s in range(STEPS)]
    torch.cuda.synchronize(dev)
    return losses, [p.detach().cpu().numpy().copy() for p in net.parameters()], tr


def _worker(rank, world, port, accum, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                      TORCH_FR_BUFFER_SIZE=os.environ.get("TORCH_FR_BUFFER_SIZE", "2000"))      # (capture hand-off polls the flight recorder)
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", init_method="env://", rank=rank, world_size=world, device_id=torch.device(f"cuda:{rank}"))
    losses, params, tr = _train(rank, world, world, accum, force=True)
    assert tr.reducer.enabled and tr.model.grad_reducer is tr.reducer
    q.put((rank, losses, params, tr.grads))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, accum):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, accum, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r[0]: r[1:] for r in (q.get(timeout=600) for _ in range(world))}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single process on the concatenated batch (this process, GPU 0, no process group)
    ref_losses, ref_params, ref_tr = _train(0, 1, world, accum, force=False)
    for r in range(1, world):
        for a, b in zip(res[0][1], res[r][1]):
            assert (a == b).all(), "ranks diverged"
        for a, b in zip(res[0][2], res[r][2]):
            assert (a is None and b is None) or (a == b).all(), "averaged gradients differ between ranks"
    # the averaged gradient arenas themselves: mean over ranks of the shard gradients == gradient of the global mean loss, up to
    # the fp32 summation order of the weight-gradient GEMMs (two reductions over n rows averaged vs one over 2n rows)
    worst = 0.0
    for a, b in zip(res[0][2], ref_tr.grads):
        if a is None:
            assert b is None
            continue
        err = float(np.linalg.norm(a.astype(np.float64) - b) / (np.linalg.norm(b) + 1e-30))
        worst = max(worst, err)
    print(f"[dp{world} accum{accum}] averaged gradients vs single process: worst per-parameter rel-L2 {worst:.2e}")
    assert worst < 2e-3      # (bf16 activations / gradients downstream of the loss: the two runs round identically row by row; what differs is the reduction order)
    mean_losses = np.mean([res[r][0] for r in range(world)], axis=0)
    print(f"[dp{world} accum{accum}] mean of per-rank losses {mean_losses.tolist()} vs single process {ref_losses}")
    # step 1: same weights, the global mean is the mean of the equal-sized shard means (bf16 kernels are batch-size independent
    # row by row: equality up to the fp32 summation order of the loss); step 2 also checks that the averaged update was applied
    assert np.allclose(mean_losses, ref_losses, rtol=2e-4)
    moved = 0.0
    from oracle.weights import make_state_dict
    init = [v.numpy() for k, v in make_state_dict(0, **CFG).items()]
    for a, b, p0 in zip(res[0][1], ref_params, init):
        da, db = a.astype(np.float64) - p0, b.astype(np.float64) - p0
        moved = max(moved, float(np.abs(db).max()))
        # Adam normalises the gradient, so where a gradient entry is ~0 a last-bit difference in the reduction order flips the
        # sign of a full-size step: compare the updates in the L2 sense
        assert np.linalg.norm(da - db) <= 0.1 * np.linalg.norm(db) + 1e-7
    assert moved > 1e-4


def test_data_parallel_one_rank_rccl_group():
    _run(1, 1)


@pytest.mark.parametrize("accum", [1, 2])
def test_data_parallel_two_ranks_rccl(accum):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two MI355X (the GPU boxes of this pool expose one)")
    _run(2, accum)


def _graph_worker(rank, world, port, q):
    """The data-parallel step captured with its collectives (model_trainer.capture_graph) on `world` ranks: three eager steps, capture,
    three replays, on the trainer's own synthetic data (seed 1234 + rank)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                      TORCH_FR_BUFFER_SIZE=os.environ.get("TORCH_FR_BUFFER_SIZE", "2000"))      # (capture hand-off polls the flight recorder)
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", init_method="env://", rank=rank, world_size=world, device_id=torch.device(f"cuda:{rank}"))
    import sd3_amd  # noqa: F401
    from oracle.weights import make_state_dict
    from sd3_amd.model_trainer import model_trainer
    from sd3_amd.models.diff_model import diff_model
    dev = torch.device(f"cuda:{rank}")
    torch.manual_seed(7)
    net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=dev,
                     positional_encoding="RoPE2d", **CFG)
    net.load_state_dict(make_state_dict(0, **CFG))
    tr = model_trainer(net, batchSize=PER, accumulation_steps=1, totalSteps=10, lr=1e-3, ema_update_freq=10, ema_decay=0.9, warmup_steps=0,
                       use_lr_scheduler=False, device=dev, saveDir="/tmp/_mg", numSaveSteps=100, max_res=8 * HW, use_ema=False, force_reducer=True,
                       device_rng=True)
    net.train()
    losses = [float(tr.train_step(s + 1)) for s in range(3)]
    tr.capture_graph(4)
    assert tr.capture_handoff == "polled", tr.capture_handoff     # the watchdog hand-off polled the flight recorder (not the fixed delay)
    losses += [float(tr.train_step(s + 4)) for s in range(3)]
    torch.cuda.synchronize(dev)
    q.put((rank, losses, [p.detach().cpu().numpy().copy() for p in net.parameters()], tr.reducer.buckets))
    tr._graph = None
    dist.barrier()
    dist.destroy_process_group()


def _run_graph(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_graph_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r[0]: r[1:] for r in (q.get(timeout=600) for _ in range(world))}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in range(world):
        assert all(np.isfinite(res[r][0])) and all(1e-3 < l < 20 for l in res[r][0]) and res[r][2] > 0
    for r in range(1, world):
        for a, b in zip(res[0][1], res[r][1]):
            assert (a == b).all(), "ranks diverged under graph replay"
    return res


def test_data_parallel_graph_replay_one_rank_rccl():
    _run_graph(1)


def test_data_parallel_graph_replay_two_ranks_rccl():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two MI355X (the GPU boxes of this pool expose one)")
    _run_graph(2)


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` starts its own two ranks (no torchrun) and prints one JSON line with n_gpus = 2."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two MI355X (the GPU boxes of this pool expose one)")
    import json
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["value"] > 0
