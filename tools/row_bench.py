"""Row-kernel micro-benchmark on the MMDiT-B image-stream shapes (16384 rows, d = 768): time and algorithmic GB/s.
python tools/row_bench.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sd3_amd  # noqa: E402,F401
from sd3_amd import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
BF, F32 = torch.bfloat16, torch.float32
B, N, d, H, h = 64, 256, 768, 12, 3072
M = B * N
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s, dt=F32: torch.randn(s, generator=g, device="cuda").to(dt)


def timed(name, fn, nbytes):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    print(f"{name:<22}{t * 1e6:9.1f} us {nbytes / t / 1e9:9.0f} GB/s  ({nbytes / 1e6:.0f} MB)")


# gate_residual_bwd: dy f32 + acc bf16 -> dacc bf16
dy, acc, gate = rnd(M, d), rnd(M, d, dt=BF), rnd(B, d)
dgate, bpart = torch.zeros(B, d, device="cuda"), torch.zeros(B, d, device="cuda")
timed("gate_res_bwd", lambda: ops.gate_residual_bwd(dy, acc, gate, N, dgate, bpart, BF), M * d * 8)
# ln_modulate fwd / bwd
x, sc, sh = rnd(M, d), rnd(B, d), rnd(B, d)
timed("ln_mod_fwd", lambda: ops.ln_modulate_fwd(x, sc, sh, N, BF), M * d * 6)
y = ops.ln_modulate_fwd(x, sc, sh, N, BF)
xn, mean, rstd = y if isinstance(y, tuple) else (y, None, None)
dout, dres = rnd(M, d, dt=BF), rnd(M, d)
dsc, dsh = torch.zeros(B, d, device="cuda"), torch.zeros(B, d, device="cuda")
if mean is not None:
    timed("ln_mod_bwd", lambda: ops.ln_modulate_bwd(dout, x, mean, rstd, sc, dres, N, dsc, dsh), M * d * 14)
# swiglu fwd / bwd
gu = rnd(M, 2 * h, dt=BF)
timed("swiglu_fwd", lambda: ops.mlp_act_fwd(gu, h), M * h * 6)
dh, dbup = rnd(M, h, dt=BF), torch.zeros(2 * h, device="cuda")
timed("swiglu_bwd", lambda: ops.mlp_act_bwd(dh, gu, h, dbup), M * h * 10)
# qk_norm_rope fwd / bwd (image stream: RoPE on)
qkv = rnd(M, 3 * d, dt=BF)
wq, wk = rnd(64), rnd(64)
cos, sin = rnd(N, 64), rnd(N, 64)
S = N + 154
Q, K, V = (torch.empty(B, H, S, 64, dtype=BF, device="cuda") for _ in range(3))
timed("qk_norm_rope_fwd", lambda: ops.qk_norm_rope_fwd(qkv, wq, wk, cos, sin, B, N, H, S, 0, Q, K, V), M * 3 * d * 4)
dQ, dK, dV = (rnd(B, H, S, 64, dt=BF) for _ in range(3))
dwq, dwk = torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda")
timed("qk_norm_rope_bwd", lambda: ops.qk_norm_rope_bwd(dQ, dK, dV, qkv, wq, wk, cos, sin, B, N, H, S, 0, dwq, dwk, BF), M * 3 * d * 6)
