"""Attention micro-benchmark on the MMDiT-B shape (batch 64, 12 heads, 256 image + 154 text tokens, head_dim 64) or `reps B H N M`.
Times forward and backward (prep + dQ + dK/dV launches) with HIP events; checks the outputs against a torch fp32 reference.
python tools/attn_bench.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sd3_amd  # noqa: E402,F401
from sd3_amd import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, H, N, M, hd = 64, 12, 256, 154, 64
if len(sys.argv) > 5:      # python tools/attn_bench.py reps B H N M   (e.g. 20 16 16 1024 154: MMDiT-L 512^2 batch 16)
    B, H, N, M = (int(v) for v in sys.argv[2:6])
S = N + M
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(s, generator=g, device="cuda").to(torch.bfloat16)
Q, K, V = rnd(B, H, S, hd), rnd(B, H, S, hd), rnd(B, H, S, hd)
dOx, dOc = rnd(B, N, H * hd), rnd(B, M, H * hd)
scale = hd ** -0.5


def timed(fn):
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return out, e0.elapsed_time(e1) / reps * 1e-3


(Ox, Oc, lse), tf = timed(lambda: ops.attn_fwd(Q, K, V, N, scale, 0))
(dQ, dK, dV), tb = timed(lambda: ops.attn_bwd(Q, K, V, Ox, Oc, dOx, dOc, lse, N, scale, torch.bfloat16))
fl = 4.0 * B * H * S * S * hd
print(f"attn fwd {tf * 1e6:8.1f} us  {fl / tf / 1e12:7.1f} TF   bwd {tb * 1e6:8.1f} us  {2.5 * fl / tb / 1e12:7.1f} TF   (NW={os.environ.get('MMDIT_ATTN_NW', '2')})")

# reference on a slice of the batch (fp32 math on the bf16 inputs)
nb = 4
q, k, v = (t[:nb].float().requires_grad_(True) for t in (Q, K, V))
p = torch.softmax(q @ k.transpose(-1, -2) * scale, dim=-1)
o = p @ v                                                     # (nb, H, S, hd)
do = torch.cat([dOx[:nb].view(nb, N, H, hd), dOc[:nb].view(nb, M, H, hd)], 1).permute(0, 2, 1, 3).float()
o.backward(do)
of = torch.cat([Ox[:nb].view(nb, N, H, hd), Oc[:nb].view(nb, M, H, hd)], 1).permute(0, 2, 1, 3).float()
rel = lambda a, b: float((a - b).norm() / b.norm())
print(f"rel err: O {rel(of, o.detach()):.2e}  dQ {rel(dQ[:nb].float(), q.grad):.2e}  dK {rel(dK[:nb].float(), k.grad):.2e}  dV {rel(dV[:nb].float(), v.grad):.2e}")

# attention backward + QK-RMSNorm / RoPE backward: two passes (mmdit_attn_bwd + mmdit_qk_norm_rope_bwd_pair) vs the fused epilogues
d = H * hd
qkv_x, qkv_c = rnd(B * N, 3 * d), rnd(B * M, 3 * d)
wts = [1 + 0.1 * torch.randn(64, generator=g, device="cuda") for _ in range(4)]
ang = torch.rand(N, 64, generator=g, device="cuda") * 6.28
cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
dw = [torch.zeros(64, device="cuda") for _ in range(4)]
dw4 = torch.zeros(256, device="cuda")


def two_pass():
    a, b_, c = ops.attn_bwd(Q, K, V, Ox, Oc, dOx, dOc, lse, N, scale, torch.bfloat16)
    return ops.qk_norm_rope_bwd_pair(a, b_, c, (qkv_x, wts[0], wts[1], cos, sin, N, 0, dw[0], dw[1]), (qkv_c, wts[2], wts[3], None, None, M, N, dw[2], dw[3]), B, H, S, torch.bfloat16)


_, t2 = timed(two_pass)
_, t1 = timed(lambda: ops.attn_bwd_qk(Q, K, V, Ox, Oc, dOx, dOc, lse, N, scale, qkv_x, qkv_c, *wts, cos, sin, dw4))
print(f"attn bwd + qk-norm/rope bwd: two passes {t2 * 1e6:8.1f} us   fused {t1 * 1e6:8.1f} us")
