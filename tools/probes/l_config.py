"""MMDiT-L (24 blocks, d=1024, 16 heads, 64x64x16 latents -> 1024 image tokens) smoke + throughput on one GPU."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sd3_amd
from sd3_amd.model_trainer import model_trainer
from sd3_amd.models.diff_model import diff_model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=dev,
                 positional_encoding="RoPE2d", dim=1024, num_heads=16, num_blocks=24)
tr = model_trainer(net, batchSize=B, accumulation_steps=1, totalSteps=1000, lr=1e-4, ema_update_freq=10**9, ema_decay=0.999, warmup_steps=10,
                   use_lr_scheduler=True, device=dev, saveDir="/tmp/_b", numSaveSteps=10**9, max_res=512, device_rng=True, use_ema=False,
                   hip_optimizer=os.environ.get("HIPOPT", "1") != "0")
losses = [float(tr.train_step(s)) for s in range(1, 4)]
torch.cuda.synchronize()
mode = "eager"
if os.environ.get("L_GRAPH", "1") != "0":      # replay the whole step from a hipGraph, as bench.py does (L_GRAPH=0: eager launches)
    mode = "hipGraph replay" if tr.capture_graph_agreed(4) else "eager (capture failed)"
    for s in range(4, 6):
        tr.train_step(s)
    torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for s in range(6, 6 + n):
    tr.train_step(s)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"MMDiT-L 512^2 batch {B} [{mode}]: losses {losses}, {dt * 1e3:.1f} ms/step, {B / dt:.1f} img/s, peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
