"""Per-phase timing of the MMDiT-B training step, un-traced: GPU time between HIP events at the phase boundaries and host
enqueue time of each phase (is the GPU ever waiting for the host?).  Usage: python tools/probes/phase_times.py [--torch-optimizer]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sd3_amd  # noqa
from sd3_amd.model_trainer import model_trainer
from sd3_amd.models.diff_model import diff_model

dev = torch.device("cuda:0")
torch.manual_seed(1234)
net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=dev,
                 positional_encoding="RoPE2d", checkpoint_MLP=False, checkpoint_attn=False, dim=768, num_heads=12, num_blocks=12)
tr = model_trainer(net, batchSize=64, accumulation_steps=1, totalSteps=10 ** 9, lr=1e-4, ema_update_freq=10 ** 9, ema_decay=0.999,
                   warmup_steps=1000, use_lr_scheduler=False, device=dev, saveDir="/tmp/bench_ckpt", numSaveSteps=10 ** 9,
                   null_prob_pooled=0.1, null_prob_gemma=0.316, null_prob_bert=0.316, use_amp=True, max_res=256,
                   device_rng=True, use_ema=False, hip_optimizer="--torch-optimizer" not in sys.argv)
net.train()
for s in range(5):
    tr.train_step(s + 1)
torch.cuda.synchronize()
N = 12
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(N + 1)]
host = []
t_start = time.perf_counter()
for s in range(N + 1):
    h0 = time.perf_counter(); ev[s][0].record()
    loss = tr.micro_step(final=True)
    h1 = time.perf_counter(); ev[s][1].record()
    tr.optimizer_step(6 + s)
    h2 = time.perf_counter(); ev[s][2].record()
    host.append((h0, h1, h2))
torch.cuda.synchronize()
t_end = time.perf_counter()
fb = [ev[s][0].elapsed_time(ev[s][1]) for s in range(N)]
op = [ev[s][1].elapsed_time(ev[s][2]) for s in range(N)]
gap = [ev[s][2].elapsed_time(ev[s + 1][0]) for s in range(N)]
hfb = [(h[1] - h[0]) * 1e3 for h in host[:N]]
hop = [(h[2] - h[1]) * 1e3 for h in host[:N]]
avg = lambda x: sum(x[2:]) / len(x[2:])
print(f"GPU  fwd+bwd {avg(fb):7.3f} ms   optimizer phase {avg(op):7.3f} ms   between steps {avg(gap):6.3f} ms   => step {avg(fb) + avg(op) + avg(gap):7.3f} ms")
print(f"host fwd+bwd {avg(hfb):7.3f} ms   optimizer phase {avg(hop):7.3f} ms   (enqueue time; wall {(t_end - t_start) / (N + 1) * 1e3:.3f} ms/step)")
print("host lead at end of enqueue of step k over GPU completion: per-step host enqueue total", [round(a + b, 1) for a, b in zip(hfb, hop)])
