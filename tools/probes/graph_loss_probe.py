"""Fifth probe: (a) minimal torch-only repro candidates (a multi-block reduction captured into a hipGraph, replayed with host
synchronisation between replays), (b) DOT dump of the trainer's captured graph (node kinds: kernel / memset / memcpy, edges)."""
import os
import sys

import torch

dev = torch.device("cuda:0")

# ---- (a) torch only -------------------------------------------------------------------------------------------------------------
for n in (1 << 20, 1 << 14):
    x = torch.randn(n, device=dev)
    junk = torch.full((128,), 7.0, device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        m = torch.ones(64, device=dev, dtype=torch.bool)
        k = (~m).to(torch.bfloat16)          # small temporaries whose blocks are reused below
        del m, k
        y = (x * x).mean()
        z = y / 1
    res = []
    for i in range(6):
        x.normal_()
        g.replay()
        torch.cuda.synchronize()
        res.append((round(float(z), 5), round(float((x * x).mean()), 5)))
    print(f"[torch-only n={n}] (graph, eager) per replay: {res}")

# ---- (b) the trainer's graph ----------------------------------------------------------------------------------------------------
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")))
import sd3_amd  # noqa: E402,F401
from sd3_amd.model_trainer import model_trainer  # noqa: E402
from sd3_amd.models.diff_model import diff_model  # noqa: E402

torch.manual_seed(1234)
net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=dev,
                 positional_encoding="RoPE2d", checkpoint_MLP=False, checkpoint_attn=False, dim=768, num_heads=12, num_blocks=12)
tr = model_trainer(net, batchSize=64, accumulation_steps=1, totalSteps=10 ** 9, lr=1e-4, ema_update_freq=10 ** 9, ema_decay=0.999,
                   warmup_steps=1000, use_lr_scheduler=False, device=dev, saveDir="/tmp/bench_ckpt", numSaveSteps=10 ** 9,
                   null_prob_pooled=0.1, null_prob_gemma=0.316, null_prob_bert=0.316, use_amp=True, max_res=256,
                   device_rng=True, use_ema=False)
net.train()
for s in range(1, 6):
    tr.train_step(s)
real_graph = torch.cuda.CUDAGraph


class Dbg(real_graph):
    def __new__(cls, *a, **k):
        return real_graph.__new__(cls, keep_graph=False) if False else real_graph.__new__(cls)


g0 = real_graph()
try:
    g0.enable_debug_mode()
    torch.cuda.CUDAGraph = lambda *a, **k: g0
    tr.capture_graph(6)
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r03_step_graph.dot")
    g0.debug_dump(out)
    print("dot written:", os.path.getsize(out), "bytes")
except Exception as e:
    print("debug dump failed:", type(e).__name__, e)
finally:
    torch.cuda.CUDAGraph = real_graph
