"""Optimizer-step micro-benchmark on the MMDiT-B parameter list (random gradients): ClipAdamW.step_clipped (three HIP launches)
vs GradScaler-style unscale + clip_grad_norm_ + torch fused AdamW.  HIP events.  Usage (GPU box): python tools/optim_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sd3_amd  # noqa: E402,F401
from sd3_amd.models.diff_model import diff_model  # noqa: E402
from sd3_amd.optim import ClipAdamW  # noqa: E402


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda:0")
    net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=dev,
                     positional_encoding="RoPE2d", checkpoint_MLP=False, checkpoint_attn=False, dim=768, num_heads=12, num_blocks=12)
    params = [p for p in net.parameters() if p.requires_grad]
    n = sum(p.numel() for p in params)
    arena = torch.randn(n, device=dev) * 1e-3 * 1024.0
    off = 0
    for p in params:
        p.grad = arena[off:off + p.numel()].view(p.shape)
        off += p.numel()
    scale = torch.tensor(1024.0, device=dev)
    oa = ClipAdamW(params, lr=1e-4, weight_decay=0.01)
    ob = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01, fused=True)
    ta = timed(lambda: oa.step_clipped(scale, 1.0))

    def torch_path():
        grads = [p.grad for p in params]
        total = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(grads, 2.0)), 2.0)
        inv = scale.double().reciprocal().float()
        torch._foreach_mul_(grads, inv * torch.clamp(1.0 / (total * inv + 1e-6), max=1.0))
        ob.step()
        torch._foreach_mul_(grads, 1024.0)   # (restore the magnitude for the next repetition; not part of the real path)

    tb = timed(torch_path)
    print(f"{len(params)} tensors, {n / 1e6:.1f} M parameters")
    print(f"ClipAdamW.step_clipped   {ta:7.3f} ms   ({n * 32 / ta / 1e6:.0f} GB/s of the 32 B/param algorithmic traffic)")
    print(f"torch unscale+clip+AdamW {tb:7.3f} ms   (includes one extra foreach_mul)")


if __name__ == "__main__":
    main()
