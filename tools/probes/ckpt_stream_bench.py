"""Stall of a checkpoint on the MMDiT-B training loop: reference-style blocking saveModel vs the streamed one.
Usage (GPU box): python tools/probes/ckpt_stream_bench.py"""
import os, shutil, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sd3_amd  # noqa
from sd3_amd.model_trainer import model_trainer
from sd3_amd.models.diff_model import diff_model

dev = torch.device("cuda:0")
out = "/tmp/ckpt_bench"
shutil.rmtree(out, ignore_errors=True)
net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=dev,
                 positional_encoding="RoPE2d", checkpoint_MLP=False, checkpoint_attn=False, dim=768, num_heads=12, num_blocks=12)
tr = model_trainer(net, batchSize=64, accumulation_steps=1, totalSteps=10 ** 9, lr=1e-4, ema_update_freq=1, ema_decay=0.999, warmup_steps=10,
                   use_lr_scheduler=False, device=dev, saveDir=out, numSaveSteps=10 ** 9, max_res=256, device_rng=True, use_ema=True)
net.train()
step = 0
def run(n):
    global step
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        step += 1; tr.train_step(step); tr.update_ema()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
run(5)
base = run(20)
for mode in ("streamed", "streamed (pinned buffers warm)", "blocking"):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if mode == "blocking":
        stream, tr.ckpt_stream = tr.ckpt_stream, None
    tr.save_checkpoint(step)
    t_call = (time.perf_counter() - t0) * 1e3
    t20 = run(20) * 20
    if tr.ckpt_stream is not None:
        t1 = time.perf_counter(); tr.ckpt_stream.wait(); t_wait = (time.perf_counter() - t1) * 1e3
    else:
        t_wait = 0.0
    print(f"{mode:<32} save call {t_call:8.1f} ms   next 20 steps {t20:8.1f} ms (undisturbed {base * 20:.1f})   then wait() {t_wait:8.1f} ms   "
          f"loop stall {t_call + t20 - base * 20:8.1f} ms")
sz = sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out)) / 2 ** 30
print(f"checkpoint files: {sorted(os.listdir(out))[:6]} ... {sz:.2f} GiB in {out}")
shutil.rmtree(out, ignore_errors=True)
