"""How far ahead of the GPU does the host run?  Enqueue time (host loop without synchronisation) vs wall time per step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import sd3_amd  # noqa: E402,F401
from sd3_amd.model_trainer import model_trainer  # noqa: E402
from sd3_amd.models.diff_model import diff_model  # noqa: E402

dev = torch.device("cuda:0")
force = "--force-dist" in sys.argv
if force:
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=dev,
                 positional_encoding="RoPE2d", **bench.B_CFG)
tr = model_trainer(net, batchSize=64, accumulation_steps=1, totalSteps=1000, lr=1e-4, ema_update_freq=10**9, ema_decay=0.999, warmup_steps=10,
                   use_lr_scheduler=True, device=dev, saveDir="/tmp/_b", numSaveSteps=10**9, max_res=256, device_rng=True, use_ema=False, force_reducer=force)
for s in range(1, 4):
    tr.train_step(s)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for s in range(4, 4 + n):
    tr.train_step(s)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
# un-throttled host cost: one step at a time from an idle GPU (the launch queue never fills, the host never blocks)
single = []
for s in range(20, 26):
    torch.cuda.synchronize()
    a = time.perf_counter()
    tr.train_step(s)
    single.append(time.perf_counter() - a)
torch.cuda.synchronize()
print(f"host enqueue from idle GPU: {1e3 * min(single):.2f} ms/step (min of {len(single)})")
print(f"host enqueue {1e3 * (t1 - t0) / n:.2f} ms/step, wall {1e3 * (t2 - t0) / n:.2f} ms/step")
if force:
    dist.destroy_process_group()
