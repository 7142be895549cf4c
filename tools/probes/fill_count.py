"""Which host call sites launch fill / memset / copy kernels in a steady-state training step?  (torch profiler with stacks)"""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sd3_amd  # noqa
from sd3_amd.model_trainer import model_trainer
from sd3_amd.models.diff_model import diff_model
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
torch.manual_seed(1234)
net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=dev,
                 positional_encoding="RoPE2d", checkpoint_MLP=False, checkpoint_attn=False, dim=768, num_heads=12, num_blocks=12)
tr = model_trainer(net, batchSize=64, accumulation_steps=1, totalSteps=10 ** 9, lr=1e-4, ema_update_freq=10 ** 9, ema_decay=0.999,
                   warmup_steps=1000, use_lr_scheduler=False, device=dev, saveDir="/tmp/bench_ckpt", numSaveSteps=10 ** 9,
                   null_prob_pooled=0.1, null_prob_gemma=0.316, null_prob_bert=0.316, use_amp=True, max_res=256,
                   device_rng=True, use_ema=False, hip_optimizer=True)
net.train()
for s in range(5):
    tr.train_step(s + 1)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.train_step(6)
    torch.cuda.synchronize()
cnt, kern = collections.Counter(), collections.Counter()
for e in prof.events():
    if e.name.startswith("aten::") and any(k in e.name for k in ("zero", "fill", "full", "copy_", "clone", "contiguous", "cat")):
        site = next((f for f in (e.stack or []) if "sd3_amd" in f or "stable-diffusion" in f), "?")
        cnt[(e.name, site.strip()[-110:])] += 1
    if "Fill" in e.name or "fill" in e.name.lower() and "aten" not in e.name:
        kern[e.name[:80]] += 1
for (name, site), n in cnt.most_common(25):
    print(f"{n:5d}  {name:<18} {site}")
print("aten zero/fill calls:", sum(cnt.values()))
for k, n in kern.most_common(8):
    print(f"{n:5d}  {k}")
