"""Premise check for running two half-batch pipelines of the training step BESIDE each other (the GEMMs are power-limited, the row kernels are not: two
independent streams of work let the hardware run an HBM-bound kernel of one next to a GEMM of the other).  Two INDEPENDENT MMDiT-B trainers of batch 32
(own weights, own optimizer, own split-tail workspace), each step captured in its own hipGraph, replayed on two streams -- against one trainer of batch 64.
The pair runs AdamW twice (two models): the real form would share the weights and run it once.
    python tools/probes/two_pipelines.py [cu budget for the pair's GEMM grids, default 256]"""
import contextlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sd3_amd  # noqa: E402,F401
from sd3_amd import _lib, ops  # noqa: E402
from sd3_amd.model_trainer import model_trainer  # noqa: E402
from sd3_amd.models.diff_model import diff_model  # noqa: E402
from tools.gpu_sensors import GpuSensors  # noqa: E402

budget = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
L = _lib.lib()
sens = GpuSensors(dev)
B_CFG = dict(dim=768, num_heads=12, num_blocks=12)


def build(batch, seed):
    torch.manual_seed(seed)
    net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=dev,
                     positional_encoding="RoPE2d", checkpoint_MLP=False, checkpoint_attn=False, **B_CFG)
    with contextlib.redirect_stdout(sys.stderr):
        tr = model_trainer(net, batchSize=batch, accumulation_steps=1, totalSteps=10 ** 9, lr=1e-4, ema_update_freq=10 ** 9, ema_decay=0.999,
                           warmup_steps=1000, use_lr_scheduler=False, device=dev, saveDir="/tmp/two_ckpt", numSaveSteps=10 ** 9,
                           null_prob_pooled=0.1, null_prob_gemma=0.316, null_prob_bert=0.316, use_amp=True, max_res=256,
                           device_rng=True, use_ema=False, hip_optimizer=True)
    net.train()
    return tr


def warm_and_capture(tr, ws=None):
    if ws is not None:      # this trainer's own split-tail workspace: the pointer is read when a launch is enqueued, i.e. baked into its graph
        assert L.mmdit_gemm_set_workspace(ws.data_ptr(), ws.numel()) == 0
    for s in range(1, 5):
        tr.train_step(s)
    assert tr.capture_graph_agreed(5)
    for s in range(2):
        tr.train_step(6)
    torch.cuda.synchronize()


def timed(fn, n=20):
    torch.cuda.synchronize()
    sens.start()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    r = sens.stop(skip=0.2)
    return dt, r.get("clock_mhz", 0), r.get("power_w", 0)


one = build(64, 1)
warm_and_capture(one)
dt, mhz, w = timed(lambda: one.train_step(7))
print(f"one trainer, batch 64:                         {dt:7.2f} ms per 64 images = {64 / dt * 1e3:7.1f} img/s   clock {mhz:5.0f} MHz  power {w:5.0f} W", flush=True)
del one
torch.cuda.empty_cache()

L.mmdit_set_cu_budget(budget)
wsA = torch.zeros(8192 + 256 * 65536 * 4, dtype=torch.uint8, device=dev)
wsB = torch.zeros(8192 + 256 * 65536 * 4, dtype=torch.uint8, device=dev)
a, b = build(32, 2), build(32, 3)
warm_and_capture(a, wsA)
warm_and_capture(b, wsB)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def seq():
    a.train_step(7); b.train_step(7)


def conc():
    with torch.cuda.stream(s1):
        a.train_step(7)
    with torch.cuda.stream(s2):
        b.train_step(7)


dt, mhz, w = timed(seq)
print(f"two trainers of batch 32, one stream:          {dt:7.2f} ms per 64 images = {64 / dt * 1e3:7.1f} img/s   clock {mhz:5.0f} MHz  power {w:5.0f} W   (GEMM grids <= {budget} CUs)", flush=True)
dt, mhz, w = timed(conc)
print(f"two trainers of batch 32, two streams:         {dt:7.2f} ms per 64 images = {64 / dt * 1e3:7.1f} img/s   clock {mhz:5.0f} MHz  power {w:5.0f} W   (GEMM grids <= {budget} CUs)", flush=True)
print("losses:", float(a.last_loss) if a.last_loss is not None else None, float(b.last_loss) if b.last_loss is not None else None)
L.mmdit_set_cu_budget(256)
