"""Second probe for `final_loss: 0.0`: which replay pattern loses the loss?  python tools/probes/graph_loss_probe2.py MODE [n]
MODE: each = float(loss) after every replay; burst = n replays back to back, one read at the end; sync = torch.cuda.synchronize()
after every replay (no read), one read at the end; bench = the bench's sequence (2 replays, sync, n replays, sync, read)."""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")))
import sd3_amd  # noqa: E402,F401
from sd3_amd.model_trainer import model_trainer  # noqa: E402
from sd3_amd.models.diff_model import diff_model  # noqa: E402

mode = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 22
dev = torch.device("cuda:0")
torch.manual_seed(1234)
net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=dev,
                 positional_encoding="RoPE2d", checkpoint_MLP=False, checkpoint_attn=False, dim=768, num_heads=12, num_blocks=12)
tr = model_trainer(net, batchSize=64, accumulation_steps=1, totalSteps=10 ** 9, lr=1e-4, ema_update_freq=10 ** 9, ema_decay=0.999,
                   warmup_steps=1000, use_lr_scheduler=False, device=dev, saveDir="/tmp/bench_ckpt", numSaveSteps=10 ** 9,
                   null_prob_pooled=0.1, null_prob_gemma=0.316, null_prob_bert=0.316, use_amp=True, max_res=256,
                   device_rng=True, use_ema=False)
net.train()
step = 0
for _ in range(5):
    step += 1
    tr.train_step(step)
tr.capture_graph(step + 1)
out = []
if mode == "each":
    for _ in range(n):
        step += 1
        out.append(round(float(tr.train_step(step)), 4))
elif mode == "burst":
    for _ in range(n):
        step += 1
        l = tr.train_step(step)
    torch.cuda.synchronize()
    out.append(float(l))
elif mode == "sync":
    for _ in range(n):
        step += 1
        l = tr.train_step(step)
        torch.cuda.synchronize()
    out.append(float(l))
elif mode == "bench":
    for _ in range(2):
        step += 1
        tr.train_step(step)
    torch.cuda.synchronize()
    out.append(float(tr._graph_loss))
    for _ in range(n):
        step += 1
        l = tr.train_step(step)
    torch.cuda.synchronize()
    out.append(float(l))
    step += 1
    out.append(float(tr.train_step(step)))
print(f"[{mode} n={n}] losses {out}  capture-time loss tensor of the private pool (round-2 read-back): {float(tr._graph_loss)!r}  scale {float(tr.grad_scaler._scale)} grad_norm {float(tr.last_grad_norm)}")
