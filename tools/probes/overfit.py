"""End-to-end learning check: MMDiT-B on ONE fixed batch (fixed noise / timesteps): the loss must fall steadily, memory must stay
flat.  Usage (GPU box): python tools/probes/overfit.py [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sd3_amd  # noqa
from sd3_amd.model_trainer import model_trainer
from sd3_amd.models.diff_model import diff_model

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=dev,
                 positional_encoding="RoPE2d", checkpoint_MLP=False, checkpoint_attn=False, dim=768, num_heads=12, num_blocks=12)
g = torch.Generator(device="cuda").manual_seed(1)
B = 16
x0 = torch.randn((B, 16, 32, 32), generator=g, device=dev).to(torch.bfloat16)
c = torch.randn((B, 154, 2304), generator=g, device=dev).to(torch.bfloat16)
cp = torch.randn((B, 768), generator=g, device=dev).to(torch.bfloat16)
eps = torch.randn((B, 16, 32, 32), generator=g, device=dev)
t = torch.sigmoid(torch.randn((B,), generator=g, device=dev))
tr = model_trainer(net, batchSize=B, accumulation_steps=1, totalSteps=10 ** 6, lr=2e-4, ema_update_freq=10, ema_decay=0.99, warmup_steps=10,
                   use_lr_scheduler=False, device=dev, saveDir="/tmp/_o", numSaveSteps=10 ** 9, max_res=256, device_rng=True, use_ema=True)
net.train()
x_t = (1 - t)[:, None, None, None] * x0.float() + t[:, None, None, None] * eps
target = eps - x0.float()
t0 = time.perf_counter()
for s in range(1, steps + 1):
    v = net(x_t, t, c.clone(), cp.clone())
    loss = torch.nn.functional.mse_loss(v.float(), target, reduction="none").flatten(1, -1).mean()
    tr.grad_scaler.scale(loss).backward()
    tr.optimizer_step(s)
    if s % 10 == 0:
        tr.update_ema()
    if s % 20 == 0 or s == 1:
        print(f"step {s:4d}  loss {float(loss):.4f}  grad-norm {float(tr.last_grad_norm):.3f}  allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB  "
              f"reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB  pointer-table uploads {tr.optim.table_builds}", flush=True)
print(f"{steps} steps in {time.perf_counter() - t0:.1f} s")
