"""Benchmark of the MMDiT-B flow-matching TRAINING step on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: starts the N ranks itself, one child process per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one optimizer step on one synthetic batch per GPU (per-GPU batch 64, weak scaling):
synthetic bf16 latents (64,16,32,32) + Gemma-2-2b-shaped text embeds (64,154,2304) + pooled (64,768)
already resident in HBM -> noise -> MMDiT-B forward (HIP) -> rectified-flow MSE -> backward (HIP) ->
RCCL gradient all-reduce (N>1) -> unscale, clip(1.0), AdamW, scheduler.  Nothing is skipped or cached.
Rank 0 prints ONE JSON line.  value = N * 64 * K / max-over-ranks wall time.

Launch mode: after the warm-up the optimizer step (with its collectives) is captured into a hipGraph and the K timed steps are
replays (one host call each); the same trainer's eager rate is measured first and reported as `ms_per_step_eager`; `--eager` times
host launches instead.  The line proves that the timed steps trained: every timed step's loss is cloned on the device and must lie in
(1e-3, 10), the parameter norms must have moved (`loss_first`, `final_loss`, `param_norm_sum`); otherwise the run fails.

roofline: the dominant kernel (by total time) is the MFMA GEMM; every GEMM launch of three extra
steps after the timed region is bracketed with HIP events on its launch stream, achieved = its
algorithmic FLOPs (2*M*N*K per launch) / average launch duration; peak = 2.5 PFLOP/s dense bf16.
cpu_baseline: the CPU oracle restatement of the reference (oracle/, validated bit-exact against the
reference in the build container) timed on this box's host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_CFG = dict(dim=768, num_heads=12, num_blocks=12)
TRAIN_GFLOP_PER_IMG = 293.3   # 3 x 97.78 GFLOP forward (GEMM + attention matmuls, MAC = 2 FLOP)
PEAK_BF16 = 2.5e15            # dense MFMA bf16 peak per MI355X (MI355X_MICROARCH.md)


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed PMC passes (profiles/*_pmc_summary.json, newest round):
    (2*FETCH_SIZE + WRITE_SIZE) KiB as MI355X_MICROARCH.md prescribes for gfx950; collected by tools/pmc_collect.sh
    in separate rocprofv3 --pmc runs of this same bench (counters cannot be read inside the timed run).  None if absent."""
    import glob
    name = kernel.split("+")[0]
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")), reverse=True):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        # the profiler prints every template argument: <..., FP8, SWIGLU> follow the ones ops._variant names
        sw = "+swiglu" in kernel
        lean = name.startswith(("gemm_lean_kernel", "gemm_wide_kernel"))   # <..., B_KM, SWIGLU>; the general kernel: <..., FP8, SWIGLU>
        qk = "+qk" in kernel
        cands = [name[:-1] + ((",1,0>" if sw else ",0,1>" if qk else ",0,0>") if lean else (",0,1>" if sw else ",0,0>")),
                 name[:-1] + ((",1>" if sw else ",0>") if lean else ""), name, name.split("<")[0]]
        if name.startswith("gemm8_kernel"):     # tools/pmc_summary.py names the 8-phase kernel's instantiations exactly as ops._variant does
            cands = [name]
        for key in cands:
            if key in d:
                return d[key]["hbm_bytes_per_launch"], d[key]["mfma_util"], os.path.relpath(f, ROOT)
    return None, None, None


def host_cpu():
    """(model name, physical core count) of the host from /proc/cpuinfo: distinct (physical id, core id) pairs; (None, None) when unreadable."""
    try:
        model, cores, phys = None, set(), None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name" and model is None:
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                cores.add((phys, v))
        return model, (len(cores) or None)
    except OSError:
        return None, None


def cpu_baseline(seconds_budget=25.0):
    """fwd + bwd + clip + AdamW of the oracle on MMDiT-B, batch 8, host cores only."""
    import torch
    from oracle import mmdit_oracle as O
    from oracle.weights import make_state_dict
    threads = torch.get_num_threads()
    cfg = O.OracleConfig(**B_CFG)
    tr = O.OracleTrainer(make_state_dict(0, **B_CFG), cfg, lr=1e-4, warmup_steps=0)
    g = torch.Generator().manual_seed(0)
    bs = 8

    def batch():
        x0 = torch.randn((bs, 16, 32, 32), generator=g)
        c = torch.randn((bs, 154, 2304), generator=g)
        c[:, :77] *= 30
        c[:, 77:, 1024:] = 0
        return x0, torch.randn(x0.shape, generator=g), torch.sigmoid(torch.randn((bs,), generator=g)), c, torch.randn((bs, 768), generator=g)

    tr.step(*batch())  # warm-up
    n, t0 = 0, time.time()
    while n < 3 or (time.time() - t0 < seconds_budget and n < 20):
        tr.step(*batch())
        n += 1
    dt = time.time() - t0
    model, physical = host_cpu()
    return {"value": round(n * bs / dt, 3), "unit": "images/s", "cores": threads, "cpu_model": model, "physical_cores": physical, "kind": "port",
            "sample": f"{n} optimizer steps of MMDiT-B (fp32 weights, reference CPU attention branch), batch {bs}, fwd+bwd+clip+AdamW, torch CPU {threads} threads"}


def self_launch(n):
    """Start one bench.py process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment, what
    torch.distributed.run would set) and wait for them; rank 0's JSON line goes to this process's stdout.  Returns the exit
    code: 0, or the first non-zero child status (the remaining ranks are then terminated so that nobody hangs in a collective)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                   TORCH_FR_BUFFER_SIZE=os.environ.get("TORCH_FR_BUFFER_SIZE", "2000"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    code = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            rc = p.poll()
            if rc is None:
                continue
            alive.remove(p)
            if rc != 0 and code == 0:
                code = rc
                for q in alive:
                    q.terminate()
        time.sleep(0.05)
    return code


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--torch-optimizer", action="store_true", help="development A/B: torch's multi-tensor unscale/clip/AdamW kernels instead of the three HIP launches")
    ap.add_argument("--no-autotune", action="store_true", help="N > 1: keep the reducer's default settings instead of the voted warm-up A/B")
    ap.add_argument("--force-dist", action="store_true", help="development: run the N>1 code path (RCCL group + gradient reducer) with one rank")
    ap.add_argument("--graph", action="store_true", help="(default) replay the optimizer step from a hipGraph captured after the warm-up; with this flag a failed capture is an error instead of an eager fallback")
    ap.add_argument("--eager", action="store_true", help="issue every launch of every step from the host")
    ap.add_argument("--no-eager-leg", action="store_true", help="profiling runs: skip the untimed-for-the-metric eager timing leg (ms_per_step_eager) in front of the capture")
    ap.add_argument("--precision", default="fast", choices=["fast", "parity"], help="development: parity = the 1e-3 mode (fp32 activations, 3-term split-bf16 GEMMs, "
                    "reference rounding points in attention); the headline metric is quoted on fast (= the reference's bf16 autocast)")
    ap.add_argument("--check-launch", action="store_true", help="rendezvous check only (gloo, no GPU): every rank joins the group, one all-reduce, rank 0 prints the world size")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing in this process has touched the GPU yet (no torch import,
        # no HIP call), the N ranks are CHILD processes (one per GPU, env:// rendezvous on 127.0.0.1) and this process only waits.
        sys.exit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.check_launch:
        assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
        if world > 1:
            dist.init_process_group("gloo", init_method="env://", world_size=world, rank=rank)
        tsum = torch.tensor([float(rank + 1)])
        if world > 1:
            dist.all_reduce(tsum)
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"launch_ok": float(tsum) == world * (world + 1) / 2, "n_gpus": world, "local_rank": local_rank}), flush=True)
        return
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X: the MMDiT hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1 or args.force_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # the flight recorder must be on BEFORE the process group exists: model_trainer.capture_graph polls its per-group status to know
        # that RCCL's watchdog has retired every eager collective (deterministic capture hand-off, model_trainer._wait_for_watchdog)
        os.environ.setdefault("TORCH_FR_BUFFER_SIZE", "2000")
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
        from importlib import import_module
        import sd3_amd  # noqa: F401
        opts = import_module("sd3_amd.model_trainer").rccl_options()      # MMDIT_RCCL_MAX_CTAS: workgroups a collective may use (see there)
        dist.init_process_group("nccl", init_method="env://", world_size=world, rank=rank, device_id=torch.device(f"cuda:{local_rank}"),
                                **({"pg_options": opts} if opts is not None else {}))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"

    import sd3_amd  # noqa: F401
    from sd3_amd import ops
    from sd3_amd.model_trainer import loss_is_plausible, model_trainer
    from sd3_amd.models.diff_model import diff_model

    BENCH_WINDOW = (1e-3, 10.0)      # N(0, 1) synthetic latents: the loss starts near 2; model_trainer.LOSS_WINDOW is the general one
    dev = torch.device(f"cuda:{local_rank}")
    torch.manual_seed(1234)
    net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=dev,
                     positional_encoding="RoPE2d", checkpoint_MLP=False, checkpoint_attn=False, **B_CFG)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):     # (the trainer prints the reference's "Number of parameters" line: stdout carries the JSON line only)
        trainer = model_trainer(net, batchSize=args.batch, accumulation_steps=1, totalSteps=10 ** 9, lr=1e-4, ema_update_freq=10 ** 9, ema_decay=0.999,
                                warmup_steps=1000, use_lr_scheduler=False, device=dev, saveDir="/tmp/bench_ckpt", numSaveSteps=10 ** 9,
                                null_prob_pooled=0.1, null_prob_gemma=0.316, null_prob_bert=0.316, use_amp=True, max_res=256,
                                device_rng=True, use_ema=False, force_reducer=args.force_dist, hip_optimizer=not args.torch_optimizer)
    if args.precision != "fast":
        net.set_precision(args.precision)
    net.train()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step = 0
    for _ in range(args.warmup):
        step += 1
        trainer.train_step(step)
    # data parallel: the first N > 1 run tunes itself -- bucket algorithm and CU reserve by a voted A/B over eager warm-up steps (model_trainer.autotune_reducer);
    # the table goes into the JSON line (`reducer_autotune`).  --no-autotune keeps the defaults (allreduce, fp32 wire, 32 reserved CUs).
    autotune = None
    if trainer.reducer.enabled and not args.no_autotune:
        step, autotune = trainer.autotune_reducer(step)
    # The timed steps replay a hipGraph of the whole optimizer step (forward + backward + per-block gradient all-reduce on its side
    # stream + clip + AdamW, ~430 launches, one host call per step): the eager step needs ~26 ms of host enqueue per 30 ms of GPU
    # work, so any host jitter stalls the GPU -- and with N ranks every stall is propagated to all of them by the next collective.
    # The eager rate of the same trainer is measured first and reported next to it (`ms_per_step_eager`); --eager times host launches.
    use_graph = not args.eager
    launch, eager_ms = "eager", None
    params = [p for p in net.parameters() if p.requires_grad]

    def param_checksum():
        return float(torch.stack(torch._foreach_norm([p.detach() for p in params])).double().sum())

    if use_graph:
        while step < 3:              # capture needs the steady state (bf16 weight copies, zero pool, optimizer state)
            step += 1
            trainer.train_step(step)
        if not args.no_eager_leg:
            sync()
            t0 = time.perf_counter()
            n_eager = min(args.steps, 10)
            for _ in range(n_eager):
                step += 1
                trainer.train_step(step)
            sync()
            eager_ms = (time.perf_counter() - t0) / n_eager * 1e3
        # the launch mode is a collective decision (model_trainer.capture_graph_agreed): every rank tries the capture, the ranks agree
        # (MIN over "my capture succeeded") before anything else is enqueued, a failure anywhere means eager launches everywhere --
        # the measurement is never lost to the capture.  --graph makes this rank's own capture error fatal (after the agreement).
        if trainer.capture_graph_agreed(step + 1, strict=args.graph):
            launch = "hipGraph replay"
        c_before = param_checksum()
        warm = []
        for _ in range(2):              # (the first replays also warm the graph's own memory; eager fallback: the same step count on every rank)
            step += 1
            warm.append(trainer.train_step(step).clone())
        if launch == "hipGraph replay":
            # the replays must have trained on EVERY rank, or every rank goes back to eager launches (model_trainer.keep_graph_if_agreed)
            ok = all(loss_is_plausible(l, BENCH_WINDOW) for l in warm) and param_checksum() != c_before
            if not trainer.keep_graph_if_agreed(ok):
                launch = "eager (replay check failed on some rank)"
                for _ in range(2):
                    step += 1
                    trainer.train_step(step)
    warmup_effective = step
    sync()
    check0 = param_checksum()
    sync()
    t0 = time.perf_counter()
    losses = []
    for _ in range(args.steps):
        step += 1
        losses.append(trainer.train_step(step).clone())     # (a device-side copy: nothing is read back inside the timed region)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        et = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(et, op=dist.ReduceOp.MAX)
        elapsed = float(et)
    loss_first, loss_val = float(losses[0]), float(losses[-1])
    check1 = param_checksum()
    # the line must prove that the timed steps trained: a finite, plausible loss in every timed step and parameters that moved
    if not all(loss_is_plausible(l, BENCH_WINDOW) for l in losses) or check1 == check0:
        raise RuntimeError(f"bench: the timed steps did not train (losses {[float(l) for l in losses]}, parameter norm sum {check0} -> {check1})")

    # shader clock and board power under the step (sysfs hwmon of this GPU, polled from a thread over extra UNTIMED steps in the launch mode of the
    # timed ones): the MFMA peak of the roofline is quoted at 2.4 GHz, the GEMMs run at the board's power limit and clock lower
    # (tools/probes/clock_under_load.py, profiles/r05_clock_under_load.txt)
    clocks = None
    if rank == 0 and world == 1 and not args.no_roofline:
        try:
            from tools.gpu_sensors import GpuSensors
            sens = GpuSensors(dev)
            if sens.available:
                n_s = max(10, min(args.steps, 40))
                sens.start()
                for _ in range(n_s):
                    step += 1
                    trainer.train_step(step)
                sync()
                r = sens.stop(skip=0.2)
                if r.get("clock_mhz"):
                    clocks = {"shader_mhz_under_step": round(r["clock_mhz"]), "shader_mhz_range": [round(r["clock_min"]), round(r["clock_max"])],
                              "power_w_under_step": round(r["power_w"]) if "power_w" in r else None, "samples": r["samples"],
                              "source": f"sysfs hwmon ({sens.how}), 20 ms polls over {n_s} untimed steps"}
        except Exception as e:      # noqa: BLE001 -- sensors are optional evidence, never a reason to lose the line
            clocks = {"error": repr(e)}

    roofline, allreduce = None, None
    if not args.no_roofline:
        # three extra (untimed) steps with every GEMM launch bracketed by HIP events.  EVERY rank runs them -- the steps contain
        # the gradient all-reduce, so a rank-0-only loop would leave the other ranks out of the collectives -- rank 0 reports.
        from sd3_amd import engine
        overlap, engine._WG_OVERLAP = engine._WG_OVERLAP, False   # serialise the side-stream wgrad launches: clean per-kernel durations
        graph, trainer._graph = trainer._graph, None              # (eager: every launch is bracketed individually)
        ops.PROFILE = [] if rank == 0 else None
        for _ in range(3):
            step += 1
            trainer.train_step(step)
        torch.cuda.synchronize()
        engine._WG_OVERLAP = overlap
        profile, ops.PROFILE = ops.PROFILE, None
        if trainer.reducer.enabled:
            # three more eager steps with the side-stream overlap back on and HIP events around every bucket's collective: how long the
            # collectives take and how much of that the main stream waits for after the backward (reducer.timing_summary)
            trainer.reducer.timing = True
            for _ in range(3):
                step += 1
                trainer.train_step(step)
            allreduce = trainer.reducer.timing_summary(steps=3)
            trainer.reducer.timing = False
        trainer._graph = graph
    if rank == 0 and not args.no_roofline:
        stats = {}
        for name, flops, e0, e1 in profile:
            s = stats.setdefault(name, [0, 0.0, 0.0])
            s[0] += 1
            s[1] += flops
            s[2] += e0.elapsed_time(e1) * 1e-3
        tot_t = sum(s[2] for s in stats.values())
        tot_f = sum(s[1] for s in stats.values())
        dom = max(stats.items(), key=lambda kv: kv[1][2])
        roofline = {"bound": "mfma", "kernel": dom[0], "achieved": round(dom[1][1] / dom[1][2] / 1e12, 1), "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                    "frac": round(dom[1][1] / dom[1][2] / PEAK_BF16, 4), "traffic": pmc_traffic(dom[0])[0],
                    "traffic_unit": "bytes/launch (L2<->fabric, 2*FETCH_SIZE+WRITE_SIZE)", "traffic_source": pmc_traffic(dom[0])[2], "mfma_util_pmc": pmc_traffic(dom[0])[1],
                    "avg_launch_us": round(dom[1][2] / dom[1][0] * 1e6, 2), "launches_per_step": dom[1][0] // 3,
                    "all_gemm": {"achieved": round(tot_f / tot_t / 1e12, 1), "frac": round(tot_f / tot_t / PEAK_BF16, 4),
                                 "ms_per_step": round(tot_t / 3 * 1e3, 3), "gflop_per_step": round(tot_f / 3 / 1e9, 1),
                                 # launches whose epilogue also runs a former row kernel ("+qk": QK-norm + RoPE + joint-layout store; "swiglu_bwd":
                                 # the SwiGLU backward, an HBM-bound pass of 644 MB per launch) are priced with their GEMM FLOPs only; the same
                                 # figure over the plain GEMM launches:
                                 "frac_without_fused_row_work": round(sum(v[1] for k, v in stats.items() if "+qk" not in k and "swiglu_bwd" not in k) /
                                                                      max(1e-12, sum(v[2] for k, v in stats.items() if "+qk" not in k and "swiglu_bwd" not in k)) / PEAK_BF16, 4)},
                    "by_kernel": {k: {"tflops": round(v[1] / v[2] / 1e12, 1), "ms_per_step": round(v[2] / 3 * 1e3, 3), "launches_per_step": v[0] // 3} for k, v in sorted(stats.items())}}
    if world > 1:
        dist.barrier()

    if rank == 0:
        value = world * args.batch * args.steps / elapsed
        out = {"metric": "train images/sec MMDiT-B 256^2 bf16", "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "bf16" if args.precision == "fast" else "f32 (split-bf16 MFMA)", "data": "synthetic",
               "config": {"workload": "MMDiT-B (12 blocks, d=768, 12 heads, SwiGLU 4x, RoPE2d) 256^2 images -> 32x32x16 latents, synthetic "
                                      "Gemma-2-2b-shaped text embeds (154x2304) + pooled (768); fwd+bwd+grad-allreduce+clip+AdamW (fp32 master weights)",
                          "per_gpu_batch": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}"},
               "mfma_roofline_frac_step": round(value * TRAIN_GFLOP_PER_IMG * 1e9 / (world * PEAK_BF16), 4),
               "loss_first": round(loss_first, 5), "final_loss": round(loss_val, 5), "param_norm_sum": [round(check0, 6), round(check1, 6)],
               "warmup_effective": warmup_effective, "ms_per_step_eager": None if eager_ms is None else round(eager_ms, 3),
               "optimizer": "hip (unscale+clip+AdamW, 3 launches)" if trainer.hip_optimizer else "torch (multi-tensor)",
               "launch": launch}
        if trainer.hip_optimizer:
            out["optimizer_table_builds"] = trainer.optim.table_builds
        if roofline is not None:
            out["roofline"] = roofline
        if clocks is not None:
            out["clocks"] = clocks
        if trainer.reducer.enabled:
            out["reducer"] = {"algorithm": trainer.reducer.algorithm, "wire": "bf16" if trainer.reducer.wire_dtype == torch.bfloat16 else "fp32",
                              "reserved_cus": trainer.reserved_cus}
        if autotune is not None:
            out["reducer_autotune"] = autotune
        if allreduce is not None:
            out.update(allreduce)      # allreduce_total_ms / allreduce_exposed_ms / buckets_per_step (eager steps, per step)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1 or args.force_dist:
        dist.barrier()
        trainer._graph = None        # (the graph holds the captured collectives: release it before the communicator)
        torch.cuda.synchronize()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
