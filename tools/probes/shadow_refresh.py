"""Do the bf16 shadow copies of the weights follow the optimizer?  For both optimizer paths: parameter versions before / after
a step, number of ops.cast calls in the next forward, and whether the forward output moves after a large-lr step."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sd3_amd  # noqa
from sd3_amd import ops
from sd3_amd.model_trainer import model_trainer
from sd3_amd.models.diff_model import diff_model

dev = torch.device("cuda:0")
for hip in (True, False):
    torch.manual_seed(0)
    net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=dev,
                     positional_encoding="RoPE2d", dim=128, num_heads=2, num_blocks=3)
    tr = model_trainer(net, batchSize=4, accumulation_steps=1, totalSteps=10, lr=1e-2, ema_update_freq=1, ema_decay=0.9, warmup_steps=0,
                       use_lr_scheduler=False, device=dev, saveDir="/tmp/_t", numSaveSteps=100, max_res=128, device_rng=True, use_ema=False, hip_optimizer=hip)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((2, 16, 16, 16), generator=g, device=dev)
    c = torch.randn((2, 154, 2304), generator=g, device=dev)
    cp = torch.randn((2, 768), generator=g, device=dev)
    t = torch.tensor([0.3, 0.7], device=dev)
    w = net.blocks[0].attn.query_proj_x.weight
    with torch.no_grad():
        y0 = net(x, t, c.clone(), cp.clone()).float().clone()
    v0, w0 = w._version, w.detach().clone()
    tr.train_step(1)
    n = [0]
    real = ops.cast
    def counting(*a, **k):
        n[0] += 1
        return real(*a, **k)
    ops.cast = counting
    import sd3_amd.packing as pk
    with torch.no_grad():
        y1 = net(x, t, c.clone(), cp.clone()).float().clone()
    ops.cast = real
    print(f"hip_optimizer={hip}: version {v0} -> {w._version}; master moved {float((w.detach() - w0).abs().max()):.3e}; casts in next forward {n[0]}; "
          f"forward output moved rel {float((y1 - y0).norm() / y0.norm()):.3e}")
