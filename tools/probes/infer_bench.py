"""Forward-only (sampler) throughput of MMDiT-B, bf16 vs fp8 operands: one CFG step = a batch of 2 x B forwards."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sd3_amd, bench
from sd3_amd.models.diff_model import diff_model
dev = torch.device("cuda:0")
cfg = dict(dim=1024, num_heads=16, num_blocks=24) if "--L" in sys.argv else bench.B_CFG
side = 64 if "--L" in sys.argv else 32
B = 128
net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=dev, positional_encoding="RoPE2d", **cfg)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(B, 16, side, side, generator=g, device=dev)
c = torch.randn(B, 154, 2304, generator=g, device=dev).to(torch.bfloat16)
cp = torch.randn(B, 768, generator=g, device=dev).to(torch.bfloat16)
t = torch.rand(B, generator=g, device=dev)
for prec in ("fast", "fp8"):
    net.set_precision(prec)
    with torch.no_grad():
        for _ in range(3): net(x, t, c.clone(), cp.clone())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 10
        for _ in range(n): net(x, t, c.clone(), cp.clone())
        e1.record(); torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) / n * 1e-3
    print(f"{'MMDiT-L 512^2' if '--L' in sys.argv else 'MMDiT-B 256^2'} forward batch {B} [{prec}]: {dt*1e3:.2f} ms, {B/dt:.0f} forward img/s, {B/2/dt/28:.1f} images/s at 28 CFG steps")
