"""BASELINE.json configs[3] ("config 4"), the single-GPU slice: MMDiT-L (24 blocks, d = 1024, 16 heads) on 512^2 images -> 64x64x16 latents,
bf16, per-GPU batch 16, with the FLUX-VAE encode running ON the training GPU in front of every step (helpers/VAE_T5_CLIP.py:176-182 runs
it on dedicated loader GPUs).  Reports one JSON line: end-to-end images/s (encode + step), the step alone, the encode alone, and the
MFMA roofline fraction of the step's GEMM launches.  Synthetic images / text embeddings, random-init weights.
python tools/config4_bench.py [--batch 16] [--steps 8] [--warmup 3]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sd3_amd  # noqa: E402,F401
from sd3_amd import ops  # noqa: E402
from sd3_amd.helpers.latent_source import ImageLatentSource  # noqa: E402
from sd3_amd.model_trainer import model_trainer  # noqa: E402
from sd3_amd.models.diff_model import diff_model  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--eager", action="store_true", help="host launches instead of replaying the step from a hipGraph (the encode is always launched from the host)")
a = ap.parse_args()
dev = torch.device("cuda:0")
PEAK = 2.5e15


def build(data_source):
    net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=dev,
                     positional_encoding="RoPE2d", dim=1024, num_heads=16, num_blocks=24)
    return model_trainer(net, batchSize=a.batch, accumulation_steps=1, totalSteps=1000, lr=1e-4, ema_update_freq=10 ** 9, ema_decay=0.999, warmup_steps=10,
                         use_lr_scheduler=True, device=dev, saveDir="/tmp/_c4", numSaveSteps=10 ** 9, max_res=512, device_rng=True, use_ema=False,
                         hip_optimizer=True, data_source=data_source)


LAUNCH = {}


def timed_steps(tr, first, tag):
    for s in range(first, first + a.warmup):
        loss = tr.train_step(s)
    LAUNCH[tag] = "eager"
    if not a.eager and tr.capture_graph_agreed(first + a.warmup):      # the step replays from a hipGraph; a data source's batch is copied into its input slots
        LAUNCH[tag] = "hipGraph replay"
        for s in range(2):
            tr.train_step(first + a.warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(first + a.warmup, first + a.warmup + a.steps):
        loss = tr.train_step(s)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.steps, float(loss)


# (1) end to end: synthetic 512^2 images -> in-rank VAE encode -> MMDiT-L step
src = ImageLatentSource.synthetic(a.batch, 512, 768, dev)
tr = build(src)
t_e2e, loss_e2e = timed_steps(tr, 1, "end_to_end")
# the encode alone, on the same object
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    src()
torch.cuda.synchronize()
t_enc = (time.perf_counter() - t0) / a.steps
def gemm_profile(trainer):
    """(FLOPs, seconds) per step of every GEMM launch (HIP events around each launch of two extra steps, as bench.py does)."""
    from sd3_amd import engine
    overlap, engine._WG_OVERLAP = engine._WG_OVERLAP, False   # serialise the side-stream weight-gradient launches: clean per-launch durations
    ops.PROFILE = []
    graph, trainer._graph = trainer._graph, None              # (eager: every launch is bracketed individually)
    for s in range(100, 102):
        trainer.train_step(s)
    torch.cuda.synchronize()
    trainer._graph = graph
    engine._WG_OVERLAP = overlap
    f = sum(x[1] for x in ops.PROFILE) / 2
    t = sum(x[2].elapsed_time(x[3]) for x in ops.PROFILE) * 1e-3 / 2
    ops.PROFILE = None
    return f, t


f_all, t_all = gemm_profile(tr)      # MMDiT GEMMs + the VAE's implicit-GEMM convolutions
peak_mem = torch.cuda.max_memory_allocated() / 2 ** 30
del tr, src
torch.cuda.empty_cache()
# (2) the step alone (latents already in HBM: SyntheticData)
tr = build(None)
t_step, _ = timed_steps(tr, 1, "step_only")
f_mm, t_mm = gemm_profile(tr)
roof = lambda f, t: {"bound": "mfma", "achieved_tflops": round(f / t / 1e12, 1), "peak_tflops": PEAK / 1e12, "frac": round(f / t / PEAK, 4), "gemm_ms_per_step": round(t * 1e3, 2),
                     "gemm_tflop_per_step": round(f / 1e12, 1)}
print(json.dumps({"config": "MMDiT-L (24 blocks, d=1024, 16 heads) 512^2 images -> in-rank FLUX-VAE encode -> 64x64x16 latents -> fwd+bwd+clip+AdamW, bf16, 1 GPU slice of BASELINE configs[3]",
                  "per_gpu_batch": a.batch, "images_per_s_end_to_end": round(a.batch / t_e2e, 1), "ms_per_step_end_to_end": round(t_e2e * 1e3, 2),
                  "images_per_s_step_only": round(a.batch / t_step, 1), "ms_per_step_only": round(t_step * 1e3, 2),
                  "vae_encode_ms_per_batch": round(t_enc * 1e3, 2), "loss_last": round(loss_e2e, 5),
                  "mmdit_gemm_roofline": roof(f_mm, t_mm), "vae_conv_gemm_roofline": roof(f_all - f_mm, max(1e-9, t_all - t_mm)),
                  "mfma_roofline_frac_of_step_time": round(f_mm / t_step / PEAK, 4), "launch": LAUNCH, "data": "synthetic", "peak_mem_gib": round(peak_mem, 1)}))
