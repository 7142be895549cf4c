import os, sys, torch
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/tests") else os.environ.get("GRAFT_REPO_ROOT", "."))
import sd3_amd
from oracle.weights import make_state_dict
from sd3_amd.model_trainer import model_trainer
from sd3_amd.models.diff_model import diff_model
CFG = dict(dim=128, num_heads=2, num_blocks=3)
def run(graph):
    torch.manual_seed(0)
    net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=torch.device("cuda:0"), positional_encoding="RoPE2d", **CFG)
    net.load_state_dict(make_state_dict(0, **CFG))
    tr = model_trainer(net, batchSize=4, accumulation_steps=1, totalSteps=100, lr=1e-3, ema_update_freq=1, ema_decay=0.9, warmup_steps=8, use_lr_scheduler=False, device=torch.device("cuda:0"), saveDir="/tmp/_t", numSaveSteps=100, null_prob_pooled=0.1, null_prob_gemma=0.316, null_prob_bert=0.316, max_res=128, device_rng=True, use_ema=False)
    seen = {}
    orig = tr.data_source.__call__
    ds = tr.data_source
    class Wrap(type(ds)):
        pass
    real_noise = net.noise_batch
    def noise(X, t):
        xt, eps = real_noise(X, t)
        seen["x0"], seen["t"], seen["eps"], seen["xt"] = X, t, eps, xt
        return xt, eps
    net.noise_batch = noise
    for s in (1, 2, 3):
        tr.train_step(s)
    p3 = [p.detach().clone() for p in net.parameters()]
    if graph:
        tr.capture_graph(4)
    l4 = float(tr.train_step(4))
    out = {k: v.detach().float().clone() for k, v in seen.items()}
    return l4, out, p3
l0, d0, p0 = run(False)
l1, d1, p1 = run(True)
print("loss", l0, l1)
print("params after 3 eager steps identical:", all(torch.equal(a, b) for a, b in zip(p0, p1)))
for k in d0:
    print(k, "equal" if torch.equal(d0[k], d1[k]) else f"DIFF max {float((d0[k]-d1[k]).abs().max()):.3e}")
