"""Run one training step (and the samplers / VAE) under canary-guarded allocations (sd3_amd.debug_guard): every kernel output is
over-allocated by 512 B on each side, the guards are checked at the end of the step and -- for a buffer that was hit -- per launch.

  python tools/probes/guard_step.py [b|l|micro] [batch]
"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")))
import sd3_amd  # noqa: E402,F401
from sd3_amd import debug_guard  # noqa: E402
from sd3_amd.model_trainer import model_trainer  # noqa: E402
from sd3_amd.models.diff_model import diff_model  # noqa: E402

CFG = {"b": (dict(dim=768, num_heads=12, num_blocks=12), 256, 64), "l": (dict(dim=1024, num_heads=16, num_blocks=24), 512, 4),
       "micro": (dict(dim=128, num_heads=2, num_blocks=3), 128, 4)}
name = sys.argv[1] if len(sys.argv) > 1 else "b"
cfg, res, batch = CFG[name]
if len(sys.argv) > 2:
    batch = int(sys.argv[2])
dev = torch.device("cuda:0")
torch.manual_seed(1234)
net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=dev,
                 positional_encoding="RoPE2d", checkpoint_MLP=False, checkpoint_attn=False, **cfg)
tr = model_trainer(net, batchSize=batch, accumulation_steps=1, totalSteps=10 ** 9, lr=1e-4, ema_update_freq=10 ** 9, ema_decay=0.999,
                   warmup_steps=1000, use_lr_scheduler=False, device=dev, saveDir="/tmp/bench_ckpt", numSaveSteps=10 ** 9,
                   null_prob_pooled=0.1, null_prob_gemma=0.316, null_prob_bert=0.316, use_amp=True, max_res=res,
                   device_rng=True, use_ema=False)
net.train()
for s in (1, 2):
    tr.train_step(s)
torch.cuda.synchronize()

reg = debug_guard.install(per_launch=False)
loss = tr.train_step(3)
torch.cuda.synchronize()
bad = reg.verify("end of training step")
print(f"[guard {name} batch {batch}] training step: loss {float(loss):.5f}, {reg.seq} guarded allocations, {reg.launches} launches, {len(bad)} guard violations")
for b in bad:
    print("   ", b)
if bad:      # name the launches
    watch = {b["alloc"][0] for b in bad}
    reg.reset()
    reg.seq = 0
    debug_guard.install(per_launch=True, watch=watch)
    tr.train_step(4)
    torch.cuda.synchronize()
debug_guard.uninstall()

# inference paths: bf16 / fp8 / mxfp8 forward (the sampler's step) under the guards
x = torch.randn((batch, 16, res // 8, res // 8), device=dev)
c = torch.randn((batch, 154, 2304), device=dev)
cp = torch.randn((batch, 768), device=dev)
t = torch.rand((batch,), device=dev)
net.eval()
for prec in ("fast", "fp8", "mxfp8", "parity"):
    reg = debug_guard.install(per_launch=False)
    net.set_precision(prec)
    with torch.no_grad():
        v = net(x, t, c.clone(), cp.clone())
    torch.cuda.synchronize()
    bad = reg.verify(f"end of {prec} forward")
    print(f"[guard {name}] {prec} forward: finite {bool(torch.isfinite(v).all())}, {reg.seq} guarded allocations, {len(bad)} guard violations")
    for b in bad:
        print("   ", b)
    debug_guard.uninstall()
net.set_precision("fast")
