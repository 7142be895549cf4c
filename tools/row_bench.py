"""Row-kernel micro-benchmark on the MMDiT-B image-stream shapes (16384 rows, d = 768): time and algorithmic GB/s.
python tools/row_bench.py [reps] [--cold]
--cold: a 1.5 GB copy runs before every timed call, so the kernel's operands come from HBM as they do inside a training step
(back-to-back calls on the same buffers are partly served by the 256 MB Infinity Cache and flatter the number)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sd3_amd  # noqa: E402,F401
from sd3_amd import ops  # noqa: E402

COLD = "--cold" in sys.argv
_args = [a for a in sys.argv[1:] if not a.startswith("--")]
reps = int(_args[0]) if _args else 30
_flush_src = torch.empty(768 * 2 ** 20, dtype=torch.uint8, device="cuda") if COLD else None
_flush_dst = torch.empty_like(_flush_src) if COLD else None
BF, F32 = torch.bfloat16, torch.float32
B, N, d, H, h = 64, 256, 768, 12, 3072
M = B * N
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s, dt=F32: torch.randn(s, generator=g, device="cuda").to(dt)


def timed(name, fn, nbytes):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if COLD:
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for e0, e1 in ev:
            _flush_dst.copy_(_flush_src)
            e0.record()
            fn()
            e1.record()
        torch.cuda.synchronize()
        t = sum(e0.elapsed_time(e1) for e0, e1 in ev) / reps * 1e-3
    else:
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
    # gated form (the one the training step runs): + acc read, dacc write, dgate / dbias column sums
    accb, gate_, dgate_, dbias_ = rnd(M, d, dt=BF), rnd(B, d), torch.zeros(B, d, device="cuda"), torch.zeros(B, d, device="cuda")
    timed("ln_mod_bwd gated", lambda: ops.ln_modulate_bwd(dout, x, mean, rstd, sc, dres, N, dsc, dsh, gated=(accb, gate_, dgate_, dbias_)), M * d * 18)
    Mt_ = B * 154
    xt_, doutt_, drest_, acct_ = rnd(Mt_, d), rnd(Mt_, d, dt=BF), rnd(Mt_, d), rnd(Mt_, d, dt=BF)
    yt_ = ops.ln_modulate_fwd(xt_, sc, sh, 154, BF)
    pa_ = dict(dout=dout, x=x, mean=mean, rstd=rstd, scale=sc, dres=dres, rpb=N, dscale=dsc, dshift=dsh, gated=(accb, gate_, dgate_, dbias_))
    pb_ = dict(dout=doutt_, x=xt_, mean=yt_[1], rstd=yt_[2], scale=sc, dres=drest_, rpb=154, dscale=dsc, dshift=dsh, gated=(acct_, gate_, dgate_, dbias_))
    timed("ln_mod_bwd gated img+txt pair", lambda: ops.ln_modulate_bwd_pair(pa_, pb_), (M + Mt_) * d * 18)
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
# one launch over image + text rows vs two launches (how much of a short row kernel is ramp-up / tail?)
Mc = B * 154
xa, xb, xab = rnd(M, d), rnd(Mc, d), rnd(M + Mc, d)
sc2 = rnd(2 * B, d)
timed("ln_mod_fwd img+txt, 2 launches", lambda: (ops.ln_modulate_fwd(xa, sc, sh, N, BF), ops.ln_modulate_fwd(xb, sc, sh, 154, BF)), (M + Mc) * d * 6)
timed("ln_mod_fwd img+txt, 1 launch", lambda: ops.ln_modulate_fwd(xab, sc2[:, :], sc2[:, :], (M + Mc) // (2 * B), BF), (M + Mc) * d * 6)
# the same two launches forked onto two streams (event fork / join on the GPU): do their ramp-ups and tails overlap?
_side = torch.cuda.Stream()
def _two_streams():
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(main)
    ops.ln_modulate_fwd(xa, sc, sh, N, BF)
    with torch.cuda.stream(_side):
        _side.wait_event(ev)
        ops.ln_modulate_fwd(xb, sc, sh, 154, BF)
        ev2 = torch.cuda.Event()
        ev2.record(_side)
    main.wait_event(ev2)
timed("ln_mod_fwd img+txt, 2 streams", _two_streams, (M + Mc) * d * 6)
# text pre-norm backward (both halves of the 154 tokens, d = 2304; once per step)
xt = rnd(B, 154, 2304, dt=BF)
w1_, w2_ = rnd(2304), rnd(2304)
s1_, s2_ = torch.ones(1, device="cuda"), torch.ones(1, device="cuda")
g1_, g2_ = rnd(B * 77, 2304, dt=BF), rnd(B * 77, 2304, dt=BF)
timed("text_rmsnorm_bwd (2 halves)", lambda: ops.text_rmsnorm_bwd(g1_, g2_, xt, w1_, w2_, s1_, s2_, 77), B * 154 * 2304 * 4)
