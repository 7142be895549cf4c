"""The reference's OWN training stages as model steps (README.md:251-254, src/train.py:11-13,35-47): 19 blocks, d = 1216, 19 heads at
  stage 1: 256^2,  per-GPU batch 140 x 2 accumulation steps
  stage 2: 512^2,  per-GPU batch 40
  stage 3: 1024^2, per-GPU batch 13 x 2 accumulation steps (S = 4250)
One GPU's slice of each: synthetic latents / text embeddings, random-init weights, bf16 operands, the whole optimizer step (micro-steps,
clip, AdamW) replayed from a hipGraph.  One JSON line per stage: images/s, ms per optimizer step, fraction of the bf16 MFMA roofline
(SURVEY 8d FLOPs, tools/flops.py), peak memory -- the number that decides whether `checkpoint_MLP / checkpoint_attn` may stay no-ops.
    python tools/stage_bench.py [--stages 2,3] [--steps 5]"""
import argparse
import contextlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import sd3_amd  # noqa: E402,F401
from flops import train_flops  # noqa: E402
from sd3_amd.model_trainer import model_trainer  # noqa: E402
from sd3_amd.models.diff_model import diff_model  # noqa: E402

STAGES = {1: (256, 140, 2), 2: (512, 40, 1), 3: (1024, 13, 2)}
PEAK = 2.5e15

ap = argparse.ArgumentParser()
ap.add_argument("--stages", default="2,3")
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--blocks", type=int, default=19)
ap.add_argument("--eager", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
d = 64 * a.blocks

for st in [int(s) for s in a.stages.split(",")]:
    res, batch, accum = STAGES[st]
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    torch.manual_seed(1234)
    net = diff_model(inCh=16, class_dim=768, patch_size=2, dim=d, hidden_scale=4.0, num_heads=a.blocks, attn_type="softmax_flash", MLP_type="swiglu",
                     num_blocks=a.blocks, device=dev, positional_encoding="RoPE2d", max_res_orig=256, max_res=res, update_max_res=True)
    with contextlib.redirect_stdout(sys.stderr):
        tr = model_trainer(net, batchSize=batch, accumulation_steps=accum, totalSteps=10 ** 9, lr=1e-4, ema_update_freq=10 ** 9, ema_decay=0.99, warmup_steps=1000,
                           use_lr_scheduler=False, device=dev, saveDir="/tmp/_stage", numSaveSteps=10 ** 9, null_prob_pooled=0.1, null_prob_gemma=0.316,
                           null_prob_bert=0.316, use_amp=True, max_res=res, device_rng=True, use_ema=False, hip_optimizer=True)
    net.train()
    step = 0
    for _ in range(3):
        step += 1
        tr.train_step(step)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        step += 1
        tr.train_step(step)
    torch.cuda.synchronize()
    eager_ms = (time.perf_counter() - t0) / 2 * 1e3
    launch = "eager"
    if not a.eager and tr.capture_graph_agreed(step + 1):
        launch = "hipGraph replay"
    for _ in range(2):
        step += 1
        tr.train_step(step)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = []
    for _ in range(a.steps):
        step += 1
        losses.append(tr.train_step(step).clone())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    imgs = batch * accum
    flop = train_flops(d, a.blocks, (res // 16) ** 2) * imgs
    rec = {"stage": st, "workload": f"{a.blocks} blocks, d = {d}, {a.blocks} heads, {res}^2 images (S = {(res // 16) ** 2 + 154}), per-GPU batch {batch} x {accum} accumulation",
           "images_per_s": round(imgs / dt, 2), "ms_per_optimizer_step": round(dt * 1e3, 2), "ms_per_optimizer_step_eager": round(eager_ms, 2), "launch": launch,
           "tflop_per_step": round(flop / 1e12, 1), "mfma_roofline_frac_step": round(flop / dt / PEAK, 4),
           "peak_mem_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), "peak_reserved_gib": round(torch.cuda.max_memory_reserved() / 2 ** 30, 1),
           "losses": [round(float(l), 4) for l in losses], "replayed_steps": tr.replayed_steps}
    assert all(1e-3 < l < 20 for l in rec["losses"]), rec
    print(json.dumps(rec), flush=True)
    tr._graph = None
    del tr, net
