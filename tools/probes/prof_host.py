import cProfile, pstats, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, sd3_amd
from sd3_amd.model_trainer import model_trainer
from sd3_amd.models.diff_model import diff_model
dev = torch.device("cuda:0")
net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=dev, positional_encoding="RoPE2d", **bench.B_CFG)
tr = model_trainer(net, batchSize=64, accumulation_steps=1, totalSteps=1000, lr=1e-4, ema_update_freq=10**9, ema_decay=0.999, warmup_steps=10,
                   use_lr_scheduler=True, device=dev, saveDir="/tmp/_b", numSaveSteps=10**9, max_res=256, device_rng=True, use_ema=False)
for s in range(1, 4): tr.train_step(s)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for s in range(4, 9): tr.train_step(s)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
