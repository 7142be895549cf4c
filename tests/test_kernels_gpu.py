"""GPU parity tests of each HIP kernel (through the C ABI) against plain torch fp32 references of the
same op, on seeded inputs.  Tolerances: fp32 / split-bf16 paths 1e-5..1e-4 relative, bf16 paths
bounded by bf16 rounding of inputs/outputs (stated per test)."""
import math

import numpy as np

import os

import pytest
import torch
import torch.nn.functional as F

from oracle.weights import make_state_dict  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import sd3_amd  # noqa: F401
    from sd3_amd import ops as _ops
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return _ops


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def rnd(*shape, seed=0, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).cuda()


# ------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (300, 200, 72), (64, 9216, 768), (1000, 64, 768), (308, 256, 2304)])
def test_gemm_nt_bf16(ops, M, N, K):
    A, B = rnd(M, K, seed=1, dtype=torch.bfloat16), rnd(N, K, seed=2, dtype=torch.bfloat16)
    ref = A.float() @ B.float().T
    out = ops.gemm(A, B, out_dtype=torch.float32)
    assert rel(out, ref) < 1e-5
    outb = ops.gemm(A, B, out_dtype=torch.bfloat16)
    assert rel(outb, ref) < 4e-3  # bf16 output rounding


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 72), (256, 768, 768)])
def test_gemm_split_fp32(ops, M, N, K):
    A, B = rnd(M, K, seed=3), rnd(N, K, seed=4)
    ref = (A.double() @ B.double().T)
    out = ops.gemm(A, B, out_dtype=torch.float32, precision=ops.PREC_SPLIT)
    assert rel(out, ref) < 1e-6  # 3-term split-bf16: fp32-exact products


@pytest.mark.parametrize("prec", ["bf16", "split"])
def test_gemm_layouts_dgrad_wgrad(ops, prec):
    M, N, K = 308, 256, 192  # ragged reduction length for wgrad (M = 308 = 2*154)
    dt = torch.bfloat16 if prec == "bf16" else torch.float32
    p = ops.PREC_BF16 if prec == "bf16" else ops.PREC_SPLIT
    tol = 1e-5 if prec == "bf16" else 1e-6
    dY, W, X = rnd(M, N, seed=5, dtype=dt), rnd(N, K, seed=6, dtype=dt), rnd(M, K, seed=7, dtype=dt)
    # dgrad: dX[M,K] = dY[M,N] @ W[N,K]   (B k-major)
    dX = ops.gemm(dY, W, b_kmajor=True, out_dtype=torch.float32, precision=p)
    assert rel(dX, dY.double() @ W.double()) < tol
    # wgrad: dW[N,K] = dY^T[N,M] @ X[M,K]  (both k-major)
    dW = ops.gemm(dY, X, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, precision=p)
    assert rel(dW, dY.double().T @ X.double()) < tol


def test_gemm_epilogue(ops):
    Bt, rpb, N, K = 3, 100, 256, 128
    M = Bt * rpb
    A, W = rnd(M, K, seed=8, dtype=torch.bfloat16), rnd(N, K, seed=9, dtype=torch.bfloat16)
    bias, res = rnd(N, seed=10), rnd(M, N, seed=11)
    mod = rnd(Bt, 3 * N, seed=12)
    gate = mod[:, N:2 * N]
    aux = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    out = ops.gemm(A, W, out_dtype=torch.float32, bias=bias, gate=gate, rows_per_batch=rpb, residual=res, aux=aux)
    acc = A.float() @ W.float().T + bias
    ref = res + gate.repeat_interleave(rpb, 0) * acc
    assert rel(out, ref) < 1e-5
    assert rel(aux, acc) < 4e-3
    # silu + aux (pre-activation) + accumulate
    pre = torch.empty((M, N), dtype=torch.float32, device="cuda")
    o2 = ops.gemm(A, W, out_dtype=torch.bfloat16, bias=bias, act=ops.ACT_SILU, aux=pre)
    assert rel(pre, acc) < 1e-5
    assert rel(o2, F.silu(acc)) < 4e-3
    c = res.clone()
    ops.gemm(A, W, out=c, accumulate=True)
    assert rel(c, res + A.float() @ W.float().T) < 1e-5
    # residual without gate
    o3 = ops.gemm(A, W, out_dtype=torch.float32, residual=res)
    assert rel(o3, res + A.float() @ W.float().T) < 1e-5


# ------------------------------------------------------------------------------------ row kernels
@pytest.mark.parametrize("d,rpb,Bt", [(768, 256, 3), (256, 154, 2), (128, 64, 2), (1024, 77, 2)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_ln_modulate(ops, d, rpb, Bt, dt):
    rows = Bt * rpb
    x = rnd(rows, d, seed=1) * 2 + 0.5
    mod = rnd(Bt, 4 * d, seed=2, scale=0.3)
    scale, shift = mod[:, d:2 * d], mod[:, 3 * d:]
    out, mean, rstd = ops.ln_modulate_fwd(x, scale, shift, rpb, dt)
    xr = x.clone().requires_grad_(True)
    sr, hr = scale.clone().requires_grad_(True), shift.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (d,)) * (1 + sr.repeat_interleave(rpb, 0)) + hr.repeat_interleave(rpb, 0)
    tol = 1e-5 if dt == torch.float32 else 4e-3
    assert rel(out, ref) < tol
    assert rel(mean, x.mean(-1)) < 1e-5
    dout = rnd(rows, d, seed=3).to(dt)
    dres = rnd(rows, d, seed=4)
    ref.backward(dout.float())
    dmod = torch.zeros((Bt, 2 * d), device="cuda")
    dx = ops.ln_modulate_bwd(dout, x, mean, rstd, scale, dres, rpb, dmod[:, :d], dmod[:, d:])
    assert rel(dx, xr.grad + dres) < 2e-5
    assert rel(dmod[:, :d], sr.grad) < 2e-5
    assert rel(dmod[:, d:], hr.grad) < 2e-5


@pytest.mark.parametrize("xdt", [torch.float32, torch.bfloat16])
def test_text_rmsnorm(ops, xdt):
    Bt, d = 3, 2304
    x = (rnd(Bt, 154, d, seed=1) * 30).to(xdt)
    w1, w2 = 1 + 0.1 * rnd(d, seed=2), 1 + 0.1 * rnd(d, seed=3)
    s1, s2 = torch.tensor([0.01], device="cuda"), torch.tensor([0.02], device="cuda")
    o1, o2 = ops.text_rmsnorm_fwd(x, w1, w2, s1, s2, 77, torch.float32)
    xf = x.float()
    w1r, w2r, s1r, s2r = [t.clone().requires_grad_(True) for t in (w1, w2, s1, s2)]
    r1 = s1r * F.rms_norm(xf[:, :77], (d,), w1r, torch.finfo(torch.float32).eps)
    r2 = s2r * F.rms_norm(xf[:, 77:], (d,), w2r, torch.finfo(torch.float32).eps)
    assert rel(o1, r1.reshape(-1, d)) < 1e-5 and rel(o2, r2.reshape(-1, d)) < 1e-5
    g1, g2 = rnd(Bt * 77, d, seed=4), rnd(Bt * 77, d, seed=5)
    (r1.reshape(-1, d) * g1).sum().backward()
    (r2.reshape(-1, d) * g2).sum().backward()
    dw1, dw2, ds1, ds2 = ops.text_rmsnorm_bwd(g1, g2, x, w1, w2, s1, s2, 77)
    assert rel(dw1, w1r.grad) < 2e-5 and rel(dw2, w2r.grad) < 2e-5
    assert rel(ds1, s1r.grad) < 2e-5 and rel(ds2, s2r.grad) < 2e-5


def _rope_tables(h2, w2):
    inv = 1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))
    fh = (torch.arange(h2).float()[:, None] * inv[None]).repeat_interleave(2, -1)[:, None, :].expand(h2, w2, -1)
    fw = (torch.arange(w2).float()[:, None] * inv[None]).repeat_interleave(2, -1)[None, :, :].expand(h2, w2, -1)
    fr = torch.cat([fh, fw], -1).reshape(h2 * w2, 64)
    return fr.cos().contiguous().cuda(), fr.sin().contiguous().cuda()


def _rot_half(x):
    x = x.reshape(*x.shape[:-1], -1, 2)
    a, b = x.unbind(-1)
    return torch.stack((-b, a), -1).reshape(*x.shape[:-2], -1)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Bt,H,h2,w2,Mtxt", [(2, 3, 4, 6, 10), (16, 12, 16, 16, 154), (16, 16, 32, 32, 154)])    # (>= 2048 rows: several row lanes per workgroup in the backward; the last: MMDiT-L at 512^2)
def test_qk_norm_rope(ops, dt, Bt, H, h2, w2, Mtxt):
    N = h2 * w2
    S = N + Mtxt
    d = H * 64
    cos, sin = _rope_tables(h2, w2)
    wq, wk = 1 + 0.1 * rnd(64, seed=1), 1 + 0.1 * rnd(64, seed=2)
    qkv_x = rnd(Bt * N, 3 * d, seed=3).to(dt)
    qkv_c = rnd(Bt * Mtxt, 3 * d, seed=4).to(dt)
    Q = torch.zeros((Bt, H, S, 64), dtype=torch.bfloat16, device="cuda")
    K, V = torch.zeros_like(Q), torch.zeros_like(Q)
    ops.qk_norm_rope_fwd(qkv_x, wq, wk, cos, sin, Bt, N, H, S, 0, Q, K, V)
    ops.qk_norm_rope_fwd(qkv_c, wq, wk, None, None, Bt, Mtxt, H, S, N, Q, K, V)

    def ref(qkv, L, rope, wq_, wk_):
        q, k, v = qkv.float().reshape(Bt, L, 3, H, 64).permute(2, 0, 3, 1, 4)
        q = F.rms_norm(q, (64,), wq_, torch.finfo(torch.float32).eps)
        k = F.rms_norm(k, (64,), wk_, torch.finfo(torch.float32).eps)
        if rope:
            q = q * cos + _rot_half(q) * sin
            k = k * cos + _rot_half(k) * sin
        return q, k, v

    xr, cr = qkv_x.float().requires_grad_(True), qkv_c.float().requires_grad_(True)
    wqr, wkr = wq.clone().requires_grad_(True), wk.clone().requires_grad_(True)
    qx, kx, vx = ref(xr, N, True, wqr, wkr)
    qc, kc, vc = ref(cr, Mtxt, False, wqr, wkr)
    Qr, Kr, Vr = torch.cat([qx, qc], 2), torch.cat([kx, kc], 2), torch.cat([vx, vc], 2)
    assert rel(Q, Qr) < 4e-3 and rel(K, Kr) < 4e-3 and rel(V, Vr) < 4e-3
    gdt = dt
    dQ, dK, dV = rnd(Bt, H, S, 64, seed=5).to(gdt), rnd(Bt, H, S, 64, seed=6).to(gdt), rnd(Bt, H, S, 64, seed=7).to(gdt)
    (Qr * dQ.float()).sum().backward(retain_graph=True)
    (Kr * dK.float()).sum().backward(retain_graph=True)
    (Vr * dV.float()).sum().backward()
    dwq, dwk = torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda")
    dx = ops.qk_norm_rope_bwd(dQ, dK, dV, qkv_x, wq, wk, cos, sin, Bt, N, H, S, 0, dwq, dwk, dt)
    dc = ops.qk_norm_rope_bwd(dQ, dK, dV, qkv_c, wq, wk, None, None, Bt, Mtxt, H, S, N, dwq, dwk, dt)
    tol = 2e-5 if dt == torch.float32 else 6e-3
    assert rel(dx, xr.grad) < tol and rel(dc, cr.grad) < tol
    assert rel(dwq, wqr.grad) < tol and rel(dwk, wkr.grad) < tol


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Bt,H,h2,w2,Mtxt", [(2, 3, 4, 6, 10), (16, 12, 16, 16, 154), (16, 16, 32, 32, 154)])
def test_qk_norm_rope_pair_launch_equals_two_launches(ops, dt, Bt, H, h2, w2, Mtxt):
    """mmdit_qk_norm_rope_{fwd,bwd}_pair (image + text rows of a block in one launch) == the two single-stream launches: outputs
    bit-identical, the atomically accumulated norm-weight gradients equal up to the order of the atomics."""
    N = h2 * w2
    S, d = N + Mtxt, H * 64
    cos, sin = _rope_tables(h2, w2)
    wqx, wkx, wqc, wkc = (1 + 0.1 * rnd(64, seed=i) for i in (1, 2, 3, 4))
    qkv_x, qkv_c = rnd(Bt * N, 3 * d, seed=5).to(dt), rnd(Bt * Mtxt, 3 * d, seed=6).to(dt)
    outs = []
    for pair in (False, True):
        Q = torch.zeros((Bt, H, S, 64), dtype=torch.bfloat16, device="cuda")
        K, V = torch.zeros_like(Q), torch.zeros_like(Q)
        if pair:
            ops.qk_norm_rope_fwd_pair((qkv_x, wqx, wkx, cos, sin, N, 0), (qkv_c, wqc, wkc, None, None, Mtxt, N), Bt, H, S, Q, K, V)
        else:
            ops.qk_norm_rope_fwd(qkv_x, wqx, wkx, cos, sin, Bt, N, H, S, 0, Q, K, V)
            ops.qk_norm_rope_fwd(qkv_c, wqc, wkc, None, None, Bt, Mtxt, H, S, N, Q, K, V)
        outs.append((Q, K, V))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    dQ, dK, dV = (rnd(Bt, H, S, 64, seed=i).to(dt) for i in (7, 8, 9))
    res = []
    for pair in (False, True):
        dw = [torch.zeros(64, device="cuda") for _ in range(4)]
        if pair:
            dx, dc = ops.qk_norm_rope_bwd_pair(dQ, dK, dV, (qkv_x, wqx, wkx, cos, sin, N, 0, dw[0], dw[1]), (qkv_c, wqc, wkc, None, None, Mtxt, N, dw[2], dw[3]), Bt, H, S, dt)
        else:
            dx = ops.qk_norm_rope_bwd(dQ, dK, dV, qkv_x, wqx, wkx, cos, sin, Bt, N, H, S, 0, dw[0], dw[1], dt)
            dc = ops.qk_norm_rope_bwd(dQ, dK, dV, qkv_c, wqc, wkc, None, None, Bt, Mtxt, H, S, N, dw[2], dw[3], dt)
        res.append((dx, dc, dw))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for a, b in zip(res[0][2], res[1][2]):
        assert rel(b, a) < 1e-5


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("gelu", [False, True])
def test_mlp_act_bwd_pair_launch_equals_two_launches(ops, dt, gelu):
    hidden = 1536
    P = []
    for i, rows in enumerate((300, 77)):
        P.append((rnd(rows, hidden, seed=10 + i).to(dt), rnd(rows, hidden if gelu else 2 * hidden, seed=20 + i).to(dt)))
    db1 = [torch.zeros(P[0][1].shape[1], device="cuda") for _ in P]
    single = [ops.mlp_act_bwd(dh, gu, hidden, db, gelu) for (dh, gu), db in zip(P, db1)]
    db2 = [torch.zeros(P[0][1].shape[1], device="cuda") for _ in P]
    pair = ops.mlp_act_bwd_pair((P[0][0], P[0][1], db2[0]), (P[1][0], P[1][1], db2[1]), hidden, gelu)
    for a, b in zip(single, pair):
        assert torch.equal(a, b)
    for a, b in zip(db1, db2):
        assert rel(b, a) < 1e-5


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("gelu", [False, True])
def test_mlp_act(ops, dt, gelu):
    rows, hidden = 300, 1536
    gu = rnd(rows, hidden if gelu else 2 * hidden, seed=1).to(dt)
    gr = gu.float().requires_grad_(True)
    if gelu:
        ref = F.gelu(gr)
    else:
        g, u = gr.chunk(2, -1)
        ref = F.silu(g) * u
    h = ops.mlp_act_fwd(gu, hidden, gelu)
    tol = 1e-5 if dt == torch.float32 else 4e-3
    assert rel(h, ref) < tol
    dh = rnd(rows, hidden, seed=2).to(dt)
    ref.backward(dh.float())
    db = torch.zeros(gu.shape[1], device="cuda")
    dgu = ops.mlp_act_bwd(dh, gu, hidden, db, gelu)
    assert rel(dgu, gr.grad) < (2e-5 if dt == torch.float32 else 5e-3)
    assert rel(db, gr.grad.sum(0)) < (2e-5 if dt == torch.float32 else 5e-3)


@pytest.mark.parametrize("d,rpb,Bt", [(768, 256, 3), (256, 154, 2), (1024, 77, 2)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_ln_modulate_fwd_res_equals_residual_then_norm(ops, d, rpb, Bt, dt):
    """mmdit_ln_modulate_fwd_res == (x + gate[b] * acc) followed by mmdit_ln_modulate_fwd, and both against torch fp32;
    mmdit_gate_residual_fwd is the update on its own."""
    rows = Bt * rpb
    x = rnd(rows, d, seed=1) * 2 + 0.5
    acc = rnd(rows, d, seed=2).to(dt)
    mod = rnd(Bt, 6 * d, seed=3, scale=0.3)
    scale, shift, gate = mod[:, d:2 * d], mod[:, 3 * d:4 * d], mod[:, 5 * d:]
    x1_ref = x + gate.repeat_interleave(rpb, 0) * acc.float()
    x1, out, mean, rstd = ops.ln_modulate_fwd_res(x, acc, gate, scale, shift, rpb, dt)
    assert rel(x1, x1_ref) < 1e-6 and rel(ops.gate_residual_fwd(x, acc, gate, rpb), x1_ref) < 1e-6
    out0, mean0, rstd0 = ops.ln_modulate_fwd(x1, scale, shift, rpb, dt)
    assert rel(mean, mean0) < 1e-6 and rel(rstd, rstd0) < 1e-6 and rel(out, out0) < (1e-6 if dt == torch.float32 else 5e-4)
    ref = F.layer_norm(x1_ref, (d,)) * (1 + scale.repeat_interleave(rpb, 0)) + shift.repeat_interleave(rpb, 0)
    assert rel(out, ref) < (1e-5 if dt == torch.float32 else 4e-3)


@pytest.mark.parametrize("d,rpb,Bt", [(768, 256, 3), (256, 154, 2), (1024, 77, 2)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_ln_modulate_bwd_gated_equals_unfused_pair(ops, d, rpb, Bt, dt):
    """mmdit_ln_modulate_bwd_gated == mmdit_ln_modulate_bwd followed by mmdit_gate_residual_bwd on its dx, to fp32 round-off (the
    compiler contracts a*b+c differently in the two instantiations, so dx may differ in the last bit and dacc by one rounding of
    the activation dtype on a few elements); the atomically accumulated column sums likewise."""
    rows = Bt * rpb
    x = rnd(rows, d, seed=1) * 2 + 0.5
    mod = rnd(Bt, 6 * d, seed=2, scale=0.3)
    scale, shift, gate = mod[:, d:2 * d], mod[:, 3 * d:4 * d], mod[:, 5 * d:]
    _, mean, rstd = ops.ln_modulate_fwd(x, scale, shift, rpb, dt)
    dout, dres, acc = rnd(rows, d, seed=3).to(dt), rnd(rows, d, seed=4), rnd(rows, d, seed=5).to(dt)
    dm0, dg0, db0 = torch.zeros((Bt, 2 * d), device="cuda"), torch.zeros((Bt, 2 * d), device="cuda"), torch.zeros((Bt, d), device="cuda")
    dx0 = ops.ln_modulate_bwd(dout, x, mean, rstd, scale, dres, rpb, dm0[:, :d], dm0[:, d:])
    dacc0 = ops.gate_residual_bwd(dx0, acc, gate, rpb, dg0[:, d:], db0, dt)
    for with_bias in (True, False):
        dm1, dg1, db1 = torch.zeros((Bt, 2 * d), device="cuda"), torch.zeros((Bt, 2 * d), device="cuda"), torch.zeros((Bt, d), device="cuda")
        dx1, dacc1 = ops.ln_modulate_bwd(dout, x, mean, rstd, scale, dres, rpb, dm1[:, :d], dm1[:, d:], gated=(acc, gate, dg1[:, d:], db1 if with_bias else None))
        assert dacc1.dtype == dt and rel(dx1, dx0) < 1e-6 and rel(dacc1, dacc0) < (1e-6 if dt == torch.float32 else 5e-4)
        assert float((dacc1.float() - dacc0.float()).abs().max()) <= 2.0 ** -7 * float(dacc0.float().abs().max())     # at most one bf16 ulp anywhere
        assert rel(dm1, dm0) < 1e-6 and rel(dg1, dg0) < 2e-6 and float(dg1[:, :d].abs().max()) == 0.0
        assert rel(db1, db0) < 2e-6 if with_bias else float(db1.abs().max()) == 0.0


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gate_residual_bwd_colsum_silu(ops, dt):
    Bt, rpb, d = 3, 154, 768
    rows = Bt * rpb
    dy, acc = rnd(rows, d, seed=1), rnd(rows, d, seed=2).to(dt)
    mod = rnd(Bt, 2 * d, seed=3)
    gate = mod[:, d:]
    dmod = torch.zeros((Bt, 2 * d), device="cuda")
    db = torch.zeros(d, device="cuda")
    dacc = ops.gate_residual_bwd(dy, acc, gate, rpb, dmod[:, :d], db, dt)
    ref = dy * gate.repeat_interleave(rpb, 0)
    tol = 2e-5 if dt == torch.float32 else 4e-3
    assert rel(dacc, ref) < tol
    assert rel(dmod[:, :d], (dy * acc.float()).reshape(Bt, rpb, d).sum(1)) < 2e-5
    assert rel(db, ref.sum(0)) < 2e-5
    cs = torch.zeros(d, device="cuda")
    ops.colsum(acc, cs)
    assert rel(cs, acc.float().sum(0)) < 2e-5
    for r_small in (5, 64, 200):         # few rows: the 8-row-chunk launch shape (per-sample partial sums of a block)
        part = rnd(r_small, 2 * d, seed=6)
        cs2 = torch.zeros(2 * d, device="cuda")
        ops.colsum(part, cs2)
        assert rel(cs2, part.sum(0)) < 2e-5
    pre = rnd(Bt, d, seed=4)
    dyy = rnd(Bt, d, seed=5)
    pr = pre.clone().requires_grad_(True)
    F.silu(pr).backward(dyy)
    dbb = torch.zeros(d, device="cuda")
    dpre = ops.silu_bwd(dyy, pre, torch.float32, dbb)
    assert rel(dpre, pr.grad) < 2e-5 and rel(dbb, pr.grad.sum(0)) < 2e-5


def test_patchify_unpatchify_time_embed_cast(ops):
    x = rnd(2, 16, 6, 10, seed=1)
    tok = ops.patchify(x, torch.float32)
    ref = x.reshape(2, 16, 3, 2, 5, 2).permute(0, 2, 4, 1, 3, 5).reshape(2 * 15, 64)
    assert torch.equal(tok, ref)
    img = ops.unpatchify(tok, 2, 16, 6, 10, torch.float32)
    assert torch.equal(img, x)
    dim = 256
    denom = (torch.tensor(10000.0) ** ((2 * torch.arange(dim)) / dim)).float().cuda()
    t = torch.tensor([0.02, 0.5, 0.98], device="cuda")
    ts = torch.tensor([1000.0], device="cuda", requires_grad=True)
    e = (t * ts)[:, None] / denom[None]
    ref = torch.cat((e[:, ::2].sin(), e[:, 1::2].cos()), 1)
    out = ops.time_embed_fwd(t, ts.detach(), denom, torch.float32)
    assert (out - ref).abs().max() < 2e-4  # sin/cos of arguments up to 1e3 in fp32
    g = rnd(3, dim, seed=2)
    ref.backward(g)
    dts = ops.time_embed_bwd(g, t, ts.detach(), denom)
    assert abs(float(dts) - float(ts.grad)) < 1e-3 * (abs(float(ts.grad)) + 1e-3)
    w = rnd(1003, seed=3)
    assert torch.equal(ops.cast(w, torch.bfloat16), w.to(torch.bfloat16))


# -------------------------------------------------------------------------------------- attention
def _attn_ref(Q, K, V, scale, oracle):
    q, k, v = Q.float(), K.float(), V.float()
    if oracle:  # Attention.py:277-284
        a = (Q @ K.mT) * scale
        a = a.softmax(-1)
        return (a @ V).float()
    return ((q @ k.mT) * scale).softmax(-1) @ v


@pytest.mark.parametrize("Bt,H,N,Mt", [(2, 3, 64, 30), (1, 2, 256, 154), (2, 2, 24, 154), (1, 1, 100, 0)])
@pytest.mark.parametrize("mode", [0, 1])
def test_attention_fwd(ops, Bt, H, N, Mt, mode):
    S = N + Mt
    Q, K, V = [rnd(Bt, H, S, 64, seed=s).to(torch.bfloat16) for s in (1, 2, 3)]
    Ox, Oc, lse = ops.attn_fwd(Q, K, V, N, 0.125, mode)
    ref = _attn_ref(Q, K, V, 0.125, mode == 1)
    refm = ref.permute(0, 2, 1, 3).reshape(Bt, S, H * 64)
    out = torch.cat([Ox, Oc], 1) if Mt else Ox
    assert rel(out, refm) < (3e-3 if mode == 1 else 5e-3)
    lse_ref = torch.logsumexp((Q.float() @ K.float().mT) * 0.125, -1)
    assert (lse - lse_ref).abs().max() < 2e-2


@pytest.mark.parametrize("Bt,H,N,Mt,last", [(2, 3, 64, 30, False), (1, 2, 256, 154, False), (2, 2, 24, 154, True)])
def test_attention_bwd(ops, Bt, H, N, Mt, last):
    S = N + Mt
    Q, K, V = [rnd(Bt, H, S, 64, seed=s).to(torch.bfloat16) for s in (1, 2, 3)]
    Ox, Oc, lse = ops.attn_fwd(Q, K, V, N, 0.125, 0)
    dOx = rnd(Bt, N, H * 64, seed=4).to(torch.bfloat16)
    dOc = None if last else rnd(Bt, Mt, H * 64, seed=5).to(torch.bfloat16)
    dQ, dK, dV = ops.attn_bwd(Q, K, V, Ox, Oc, dOx, dOc, lse, N, 0.125, torch.float32)
    qr, kr, vr = [t.float().requires_grad_(True) for t in (Q, K, V)]
    ref = (((qr @ kr.mT) * 0.125).softmax(-1) @ vr).permute(0, 2, 1, 3).reshape(Bt, S, H * 64)
    dO = torch.cat([dOx.float(), torch.zeros(Bt, Mt, H * 64, device="cuda") if last else dOc.float()], 1)
    ref.backward(dO)
    # bf16 P / dS operands: ~1e-2 relative
    assert rel(dQ, qr.grad) < 1.5e-2 and rel(dK, kr.grad) < 1.5e-2 and rel(dV, vr.grad) < 1.5e-2


def test_attention_fwd_bwd_at_mmdit_l_sequence_length(ops):
    """S = 1178 (MMDiT-L: 1024 image + 154 text tokens; 18 full 64-key tiles + a 26-key tail, five 256-query workgroups with a 154-query
    tail): forward in both modes and backward against the CPU oracle's attention core (oracle/mmdit_oracle.py attention_core:
    "oracle_bf16" = the reference's CPU branch, Attention.py:277-284; "fp32" = exact softmax attention under torch autograd) for
    every (batch, head) of a (1, 2) problem; also the last block's shape (no text output gradient)."""
    from oracle.mmdit_oracle import attention_core
    Bt, H, N, Mt = 1, 2, 1024, 154
    S = N + Mt
    Q, K, V = [rnd(Bt, H, S, 64, seed=s) for s in (11, 12, 13)]
    Qb, Kb, Vb = Q.to(torch.bfloat16), K.to(torch.bfloat16), V.to(torch.bfloat16)
    merge = lambda o: o.permute(0, 2, 1, 3).reshape(Bt, S, H * 64)
    # forward, reference-rounding mode vs the oracle's restatement of the reference's CPU branch
    Ox, Oc, _ = ops.attn_fwd(Qb, Kb, Vb, N, 0.125, 1)
    ref1 = merge(attention_core(Q.cpu(), K.cpu(), V.cpu(), 0.125, "oracle_bf16"))
    r1 = rel(torch.cat([Ox, Oc], 1).cpu(), ref1)
    # forward (flash) + backward vs exact attention on the bf16-rounded operands (fp32 autograd on the CPU oracle function)
    Ox, Oc, lse = ops.attn_fwd(Qb, Kb, Vb, N, 0.125, 0)
    qr, kr, vr = [t.float().cpu().requires_grad_(True) for t in (Qb, Kb, Vb)]
    ref0 = merge(attention_core(qr, kr, vr, 0.125, "fp32"))
    r0 = rel(torch.cat([Ox, Oc], 1).cpu(), ref0.detach())
    assert r1 < 3e-3 and r0 < 5e-3, (r1, r0)
    assert (lse.cpu() - torch.logsumexp((qr.detach() @ kr.detach().mT) * 0.125, -1)).abs().max() < 2e-2
    res = []
    for last in (False, True):
        dOx = rnd(Bt, N, H * 64, seed=14).to(torch.bfloat16)
        dOc = None if last else rnd(Bt, Mt, H * 64, seed=15).to(torch.bfloat16)
        dQ, dK, dV = ops.attn_bwd(Qb, Kb, Vb, Ox, Oc, dOx, dOc, lse, N, 0.125, torch.float32)
        for t in (qr, kr, vr):
            t.grad = None
        dO = torch.cat([dOx.float().cpu(), torch.zeros(Bt, Mt, H * 64) if last else dOc.float().cpu()], 1)
        ref0.backward(dO, retain_graph=True)
        errs = (rel(dQ.cpu(), qr.grad), rel(dK.cpu(), kr.grad), rel(dV.cpu(), vr.grad))
        res.append(errs)
        # bf16 P / dS operands: ~1e-2 relative (the same bar as at S <= 410)
        assert max(errs) < 1.5e-2, (last, errs)
    print(f"[attention S=1178] fwd oracle-rounding mode {r1:.2e}, flash {r0:.2e}; bwd (dQ, dK, dV) {res[0]} / last block {res[1]}")


def test_attention_fwd_bwd_at_the_1024px_stage_sequence_length(ops):
    """S = 4250 (the reference's 1024^2 training stage, README.md:251-252 / src/train.py:47: 64 x 64 image tokens + 154 text tokens;
    66 full 64-key tiles + a 26-key tail, seventeen 256-query workgroups with a 154-query tail), head count 19-odd-like (3 heads, batch
    1): forward in both modes and backward against the CPU oracle's attention core, as at S = 1178 above; also the last block's
    shape (no text output gradient).  The long sequence is where the online-softmax rescale branch and the running sums see the
    most tiles."""
    from oracle.mmdit_oracle import attention_core
    Bt, H, N, Mt = 1, 3, 4096, 154
    S = N + Mt
    Q, K, V = [rnd(Bt, H, S, 64, seed=s) for s in (21, 22, 23)]
    # a few keys that dominate late in the sequence force accumulator rescales deep into the tile loop (guide rule 26)
    K[:, :, 4000:4003] *= 4.0
    Qb, Kb, Vb = Q.to(torch.bfloat16), K.to(torch.bfloat16), V.to(torch.bfloat16)
    merge = lambda o: o.permute(0, 2, 1, 3).reshape(Bt, S, H * 64)
    Ox, Oc, _ = ops.attn_fwd(Qb, Kb, Vb, N, 0.125, 1)
    ref1 = merge(attention_core(Q.cpu(), K.cpu(), V.cpu(), 0.125, "oracle_bf16"))
    r1 = rel(torch.cat([Ox, Oc], 1).cpu(), ref1)
    Ox, Oc, lse = ops.attn_fwd(Qb, Kb, Vb, N, 0.125, 0)
    qr, kr, vr = [t.float().cpu().requires_grad_(True) for t in (Qb, Kb, Vb)]
    ref0 = merge(attention_core(qr, kr, vr, 0.125, "fp32"))
    r0 = rel(torch.cat([Ox, Oc], 1).cpu(), ref0.detach())
    assert r1 < 3e-3 and r0 < 5e-3, (r1, r0)
    assert (lse.cpu() - torch.logsumexp((qr.detach() @ kr.detach().mT) * 0.125, -1)).abs().max() < 2e-2
    res = []
    for last in (False, True):
        dOx = rnd(Bt, N, H * 64, seed=24).to(torch.bfloat16)
        dOc = None if last else rnd(Bt, Mt, H * 64, seed=25).to(torch.bfloat16)
        dQ, dK, dV = ops.attn_bwd(Qb, Kb, Vb, Ox, Oc, dOx, dOc, lse, N, 0.125, torch.float32)
        for t in (qr, kr, vr):
            t.grad = None
        dO = torch.cat([dOx.float().cpu(), torch.zeros(Bt, Mt, H * 64) if last else dOc.float().cpu()], 1)
        ref0.backward(dO, retain_graph=True)
        errs = (rel(dQ.cpu(), qr.grad), rel(dK.cpu(), kr.grad), rel(dV.cpu(), vr.grad))
        res.append(errs)
        assert max(errs) < 1.5e-2, (last, errs)
    print(f"[attention S=4250] fwd oracle-rounding mode {r1:.2e}, flash {r0:.2e}; bwd (dQ, dK, dV) {res[0]} / last block {res[1]}")


@pytest.mark.parametrize("Bt,H,h2,w2,Mt,last", [(2, 3, 8, 8, 30, False), (1, 2, 16, 16, 154, False), (2, 2, 16, 16, 154, True), (1, 2, 32, 32, 154, False)])
def test_attention_bwd_with_fused_qk_norm_rope_backward(ops, Bt, H, h2, w2, Mt, last):
    """mmdit_attn_bwd_qk (attention backward whose epilogues run the RoPE + QK-RMSNorm backward on the fp32 accumulators and write
    the gradient of the raw QKV GEMM outputs) against (a) the two-pass composition mmdit_attn_bwd + mmdit_qk_norm_rope_bwd_pair and
    (b) torch autograd through norm -> RoPE -> exact softmax attention on the same bf16 operands."""
    N = h2 * w2
    S, d = N + Mt, H * 64
    assert ops.attn_bwd_qk_ok(torch.empty((Bt, H, S, 64), dtype=torch.bfloat16, device="cuda"), N, torch.empty((Bt * N, 3 * d), dtype=torch.bfloat16, device="cuda"))
    cos, sin = _rope_tables(h2, w2)
    wqx, wkx, wqc, wkc = (1 + 0.1 * rnd(64, seed=i) for i in (1, 2, 3, 4))
    qkv_x, qkv_c = rnd(Bt * N, 3 * d, seed=5).to(torch.bfloat16), rnd(Bt * Mt, 3 * d, seed=6).to(torch.bfloat16)
    Q = torch.zeros((Bt, H, S, 64), dtype=torch.bfloat16, device="cuda")
    K, V = torch.zeros_like(Q), torch.zeros_like(Q)
    ops.qk_norm_rope_fwd_pair((qkv_x, wqx, wkx, cos, sin, N, 0), (qkv_c, wqc, wkc, None, None, Mt, N), Bt, H, S, Q, K, V)
    Ox, Oc, lse = ops.attn_fwd(Q, K, V, N, 0.125, 0)
    dOx = rnd(Bt, N, d, seed=7).to(torch.bfloat16)
    dOc = None if last else rnd(Bt, Mt, d, seed=8).to(torch.bfloat16)
    # (a) two passes
    dQ, dK, dV = ops.attn_bwd(Q, K, V, Ox, Oc, dOx, dOc, lse, N, 0.125, torch.bfloat16)
    dw2 = [torch.zeros(64, device="cuda") for _ in range(4)]
    dx2, dc2 = ops.qk_norm_rope_bwd_pair(dQ, dK, dV, (qkv_x, wqx, wkx, cos, sin, N, 0, dw2[0], dw2[1]), (qkv_c, wqc, wkc, None, None, Mt, N, dw2[2], dw2[3]), Bt, H, S, torch.bfloat16)
    # fused
    dw4 = torch.zeros(256, device="cuda")
    dx1, dc1 = ops.attn_bwd_qk(Q, K, V, Ox, Oc, dOx, dOc, lse, N, 0.125, qkv_x, qkv_c, wqx, wkx, wqc, wkc, cos, sin, dw4)
    assert dx1.shape == dx2.shape and dc1.shape == dc2.shape and dx1.dtype == torch.bfloat16

    # (b) autograd
    def chain(qkv, L, rope, wq_, wk_):
        q, k, v = qkv.reshape(Bt, L, 3, H, 64).permute(2, 0, 3, 1, 4)
        q = F.rms_norm(q, (64,), wq_, torch.finfo(torch.float32).eps)
        k = F.rms_norm(k, (64,), wk_, torch.finfo(torch.float32).eps)
        if rope:
            q, k = q * cos + _rot_half(q) * sin, k * cos + _rot_half(k) * sin
        return q, k, v
    xr, cr = qkv_x.float().requires_grad_(True), qkv_c.float().requires_grad_(True)
    wr = [w.clone().requires_grad_(True) for w in (wqx, wkx, wqc, wkc)]
    qx, kx, vx = chain(xr, N, True, wr[0], wr[1])
    qc, kc, vc = chain(cr, Mt, False, wr[2], wr[3])
    bf = lambda t: t + (t.to(torch.bfloat16).float() - t).detach()          # the forward rounds Q, K, V to bf16 (straight-through)
    Qr, Kr, Vr = bf(torch.cat([qx, qc], 2)), bf(torch.cat([kx, kc], 2)), bf(torch.cat([vx, vc], 2))
    out = (((Qr @ Kr.mT) * 0.125).softmax(-1) @ Vr).permute(0, 2, 1, 3).reshape(Bt, S, d)
    out.backward(torch.cat([dOx.float(), torch.zeros(Bt, Mt, d, device="cuda") if last else dOc.float()], 1))
    ref = [xr.grad, cr.grad] + [w.grad for w in wr]
    fused = [dx1, dc1] + list(dw4.reshape(4, 64))
    two = [dx2, dc2] + dw2
    e1 = [rel(a, r) for a, r in zip(fused, ref)]
    e2 = [rel(a, r) for a, r in zip(two, ref)]
    e12 = [rel(a, b) for a, b in zip(fused, two)]
    print(f"[attn bwd + qk fused] S={S} vs autograd: fused {['%.2e' % e for e in e1]}, two-pass {['%.2e' % e for e in e2]}; fused vs two-pass {['%.2e' % e for e in e12]}")
    # bf16 P / dS operands: ~1e-2 relative, as for mmdit_attn_bwd; the fused form skips one bf16 rounding (dQ, dK), so it may not be worse
    assert max(e1) < 1.5e-2, e1
    assert max(e12) < 1.2e-2, e12
    for a, b in zip(e1, e2):
        assert a < b * 1.1 + 1e-4, (e1, e2)


def test_attention_oracle_mode_matches_cpu_oracle(ops):
    """mode 1 must reproduce the reference's CPU attention branch far below the 1e-3 parity bar."""
    from oracle.mmdit_oracle import attention_core
    Bt, H, S = 1, 2, 410
    Q, K, V = [rnd(Bt, H, S, 64, seed=s) for s in (1, 2, 3)]
    Qb, Kb, Vb = Q.to(torch.bfloat16), K.to(torch.bfloat16), V.to(torch.bfloat16)
    Ox, Oc, _ = ops.attn_fwd(Qb, Kb, Vb, 256, 0.125, 1)
    ref = attention_core(Q.cpu(), K.cpu(), V.cpu(), 0.125, "oracle_bf16").permute(0, 2, 1, 3).reshape(Bt, S, H * 64)
    out = torch.cat([Ox, Oc], 1).float().cpu()
    r = rel(out, ref)
    print(f"[attn oracle-mode] rel-L2 vs CPU oracle core = {r:.3e}")
    assert r < 3e-4


@pytest.mark.parametrize("kmajor", [False, True])
def test_gemm_lean_kernel_320x256_and_256x256(ops, kmajor):
    """The lean hot-path kernel (csrc/gemm_lean.hip) on the shapes it is chosen for: image + text rows grouped, N = 768 (320x256 tiles:
    one round; ragged last row tiles 16384 = 51.2 x 320, 9856 = 30.8 x 320) and N = 2304, forward (row-major weight, bias + SiLU) and
    data-gradient (k-major weight) layouts, against torch on the same bf16 operands (fp32 accumulate; bf16 output rounding 4e-3)."""
    if os.environ.get("MMDIT_GEMM_LEAN") == "0":
        pytest.skip("the lean / wide kernels are switched off (MMDIT_GEMM_LEAN=0)")
    from sd3_amd._lib import ACT_SILU
    for N, K in ((768, 768), (2304, 768), (768, 3072)):
        Ax, Ac = rnd(16384, K, seed=1, dtype=torch.bfloat16), rnd(9856, K, seed=2, dtype=torch.bfloat16)
        W = rnd(K, N, seed=3, scale=0.05, dtype=torch.bfloat16) if kmajor else rnd(N, K, seed=3, scale=0.05, dtype=torch.bfloat16)
        bias = None if kmajor else rnd(N, seed=4)
        kw = dict(b_kmajor=True) if kmajor else dict(bias=bias, act=ACT_SILU)
        probs = [dict(A=Ax, B=W, out_dtype=torch.bfloat16, **kw), dict(A=Ac, B=W, out_dtype=torch.bfloat16, **kw)]
        arr = (ops.GemmArgs * 2)()
        for i in range(2):
            ops._fill_gemm(arr[i], **probs[i])
        plan = ops._lib.lib().mmdit_gemm_plan(arr, 2)
        assert plan & 128 and (plan & 15) == (3 if N == 768 else plan & 15), plan      # lean kernel; 320x256 tiles for N = 768
        ox, oc = ops.gemm_grouped(probs)
        for o, A in ((ox, Ax), (oc, Ac)):
            ref = A.float() @ (W.float() if kmajor else W.float().T)
            if not kmajor:
                ref = F.silu(ref + bias)
            assert rel(o, ref) < 4e-3
            assert float((o.float() - ref).abs().max()) < 3e-2 * float(ref.abs().max())


@pytest.mark.parametrize("layout", ["nt", "dgrad", "wgrad"])
def test_gemm_dma_path_and_split_k(ops, layout):
    """K % 64 == 0 takes the global_load_lds kernel; ragged M/N are clamped; split-K slices add atomically."""
    M, N, K = 200, 136, 256
    if layout == "nt":
        A, B = rnd(M, K, seed=21, dtype=torch.bfloat16), rnd(N, K, seed=22, dtype=torch.bfloat16)
        ref, kw = A.double() @ B.double().T, {}
    elif layout == "dgrad":
        A, B = rnd(M, K, seed=21, dtype=torch.bfloat16), rnd(K, N, seed=22, dtype=torch.bfloat16)
        ref, kw = A.double() @ B.double(), dict(b_kmajor=True)
    else:
        A, B = rnd(K, M, seed=21, dtype=torch.bfloat16), rnd(K, N, seed=22, dtype=torch.bfloat16)
        ref, kw = A.double().T @ B.double(), dict(a_kmajor=True, b_kmajor=True)
    assert rel(ops.gemm(A, B, out_dtype=torch.float32, **kw), ref) < 1e-5
    assert rel(ops.gemm(A, B, out_dtype=torch.bfloat16, **kw), ref) < 4e-3
    bias = rnd(N, seed=23)
    res = rnd(M, N, seed=24)
    out = ops.gemm(A, B, bias=bias, residual=res, split_k=3, **kw)
    assert rel(out, ref + bias.double() + res.double()) < 1e-5


def test_gemm_stream_k_grouped_wgrad(ops):
    """Weight-gradient group (long reductions, few tiles) through the stream-K decomposition vs fp64 references."""
    probs, refs = [], []
    for i, (Mr, N, K) in enumerate([(1024, 256, 384), (640, 128, 128), (64, 512, 256), (1024, 72, 200)]):
        dY, X = rnd(Mr, N, seed=40 + i, dtype=torch.bfloat16), rnd(Mr, K, seed=50 + i, dtype=torch.bfloat16)
        probs.append(dict(A=dY, B=X, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, stream_k=True))
        refs.append(dY.double().T @ X.double())
    outs = ops.gemm_grouped(probs)
    for o, r in zip(outs, refs):
        assert rel(o, r) < 1e-5


@pytest.mark.parametrize("claiming", [False, True])
@pytest.mark.parametrize("Bt,H,h2,w2,Mt,K", [(64, 12, 16, 16, 154, 768), (16, 16, 32, 32, 154, 1024), (3, 4, 4, 6, 10, 256)])
def test_gemm_qkv_epilogue_with_qk_norm_rope_equals_gemm_plus_row_kernel(ops, Bt, H, h2, w2, Mt, K, claiming):
    """mmdit_gemm_qkv_norm_rope (QKV projection whose epilogue applies the per-head QK RMSNorm + axial RoPE and writes Q, K, V in the
    joint attention layout; Attention.py:118-135, 174-194, 258-261) against the two launches it replaces, mmdit_gemm_grouped +
    mmdit_qk_norm_rope_fwd_pair: the q / k columns of the raw projections bit-identical, Q / K / V equal up to the last bf16 bit on a vanishing fraction of
    the elements (the same arithmetic on the same rounded values; only instruction selection may differ), and both against an fp32
    torch reference.  MMDiT-B and MMDiT-L block shapes and a small ragged one (rows not a multiple of the tile, fewer than 32 tokens per sample).
    claiming: with mmdit_gemm_set_claiming(1) -- the data-parallel trainer's setting -- the planner gives the fused launch to the 8-phase kernel (epi8_qk:
    factors requested a pass ahead, FULL tiles as a compile-time variant) instead of the wide-slot kernel; the conftest fixture switches it off again."""
    from sd3_amd import _lib
    assert _lib.lib().mmdit_gemm_set_claiming(1 if claiming else 0) == 0
    N = h2 * w2
    S, d = N + Mt, H * 64
    cos, sin = _rope_tables(h2, w2)
    wqx, wkx, wqc, wkc = (1 + 0.1 * rnd(64, seed=i) for i in (1, 2, 3, 4))
    X, C = rnd(Bt * N, K, seed=5, dtype=torch.bfloat16), rnd(Bt * Mt, K, seed=6, dtype=torch.bfloat16)
    Wx, Wc = rnd(3 * d, K, seed=7, scale=0.05, dtype=torch.bfloat16), rnd(3 * d, K, seed=8, scale=0.05, dtype=torch.bfloat16)
    probs = lambda: [dict(A=X, B=Wx, out_dtype=torch.bfloat16), dict(A=C, B=Wc, out_dtype=torch.bfloat16)]
    # two launches
    qkv_x, qkv_c = ops.gemm_grouped(probs())
    Q2 = torch.zeros((Bt, H, S, 64), dtype=torch.bfloat16, device="cuda")
    K2, V2 = torch.zeros_like(Q2), torch.zeros_like(Q2)
    ops.qk_norm_rope_fwd_pair((qkv_x, wqx, wkx, cos, sin, N, 0), (qkv_c, wqc, wkc, None, None, Mt, N), Bt, H, S, Q2, K2, V2)
    # one launch
    Q1 = torch.full((Bt, H, S, 64), float("nan"), dtype=torch.bfloat16, device="cuda")
    K1, V1 = Q1.clone(), Q1.clone()
    raw = ops.gemm_qkv_norm_rope(probs(), [(wqx, wkx, cos, sin, N, 0), (wqc, wkc, None, None, Mt, N)], H, S, Q1, K1, V1)
    if raw is None:
        assert Bt * N < 2048, "the MMDiT block shapes must take the fused path"
        pytest.skip("the planner keeps this small problem off the lean kernel: the caller runs the two launches")
    assert torch.equal(raw[0][:, :2 * d], qkv_x[:, :2 * d]) and torch.equal(raw[1][:, :2 * d], qkv_c[:, :2 * d])      # (q and k columns; v is not written to the raw buffer)
    for a, b_, name in ((Q1, Q2, "Q"), (K1, K2, "K"), (V1, V2, "V")):
        assert torch.isfinite(a.float()).all(), name
        frac = float((a.view(torch.int16) != b_.view(torch.int16)).float().mean())
        assert frac < 1e-4 and rel(a, b_) < 1e-4, (name, frac, rel(a, b_))       # (instruction selection: a handful of last-bit differences)
    assert torch.equal(V1, V2)
    # fp32 reference of the chain on the bf16 operands
    def ref(A, W, L, wq_, wk_, rope):
        q, k, v = (A.float() @ W.float().T).to(torch.bfloat16).float().reshape(Bt, L, 3, H, 64).permute(2, 0, 3, 1, 4)
        q = F.rms_norm(q, (64,), wq_, torch.finfo(torch.float32).eps)
        k = F.rms_norm(k, (64,), wk_, torch.finfo(torch.float32).eps)
        if rope:
            q, k = q * cos + _rot_half(q) * sin, k * cos + _rot_half(k) * sin
        return q, k, v
    qx, kx, vx = ref(X, Wx, N, wqx, wkx, True)
    qc, kc, vc = ref(C, Wc, Mt, wqc, wkc, False)
    assert rel(Q1, torch.cat([qx, qc], 2)) < 6e-3 and rel(K1, torch.cat([kx, kc], 2)) < 6e-3 and rel(V1, torch.cat([vx, vc], 2)) < 6e-3
    # raw=False (C = NULL): only the 8-phase kernel's epilogue can drop the raw columns -- refused (None) on the wide-slot kernel, the same Q / K / V to the bit otherwise
    Q3 = torch.full((Bt, H, S, 64), float("nan"), dtype=torch.bfloat16, device="cuda")
    K3, V3 = Q3.clone(), Q3.clone()
    r3 = ops.gemm_qkv_norm_rope(probs(), [(wqx, wkx, cos, sin, N, 0), (wqc, wkc, None, None, Mt, N)], H, S, Q3, K3, V3, raw=False)
    if claiming:
        assert r3 == [None, None] and torch.equal(Q3, Q1) and torch.equal(K3, K1) and torch.equal(V3, V1)
    else:
        assert r3 is None


@pytest.mark.parametrize("Bt,H,h2,w2,Mt,K", [(16, 16, 32, 32, 154, 1024), (8, 12, 16, 16, 154, 768)])
def test_gemm_qkv_epilogue_on_mx_operands_equals_gemm_plus_row_kernel(ops, Bt, H, h2, w2, Mt, K):
    """The same fusion on MX e4m3 operands (mxfp8 inference: the QKV projection of the 8-phase MX kernel with the QK-norm / RoPE epilogue) against
    the MX GEMM followed by mmdit_qk_norm_rope_fwd_pair: raw q / k columns bit-identical, Q / K / V up to instruction selection."""
    N = h2 * w2
    S, d = N + Mt, H * 64
    cos, sin = _rope_tables(h2, w2)
    wqx, wkx, wqc, wkc = (1 + 0.1 * rnd(64, seed=i) for i in (1, 2, 3, 4))
    X, C = rnd(Bt * N, K, seed=5, dtype=torch.bfloat16), rnd(Bt * Mt, K, seed=6, dtype=torch.bfloat16)
    Wx, Wc = rnd(3 * d, K, seed=7, scale=0.05, dtype=torch.bfloat16), rnd(3 * d, K, seed=8, scale=0.05, dtype=torch.bfloat16)
    (qx, sx), (qc, sc), (qwx, swx), (qwc, swc) = (ops.quant_mxfp8(t) for t in (X, C, Wx, Wc))
    probs = lambda: [dict(A=qx, B=qwx, out_dtype=torch.bfloat16, scale_a=sx, scale_b=swx, scale_mode=1), dict(A=qc, B=qwc, out_dtype=torch.bfloat16, scale_a=sc, scale_b=swc, scale_mode=1)]
    qkv_x, qkv_c = ops.gemm_grouped(probs())
    Q2 = torch.zeros((Bt, H, S, 64), dtype=torch.bfloat16, device="cuda")
    K2, V2 = torch.zeros_like(Q2), torch.zeros_like(Q2)
    ops.qk_norm_rope_fwd_pair((qkv_x, wqx, wkx, cos, sin, N, 0), (qkv_c, wqc, wkc, None, None, Mt, N), Bt, H, S, Q2, K2, V2)
    Q1 = torch.full((Bt, H, S, 64), float("nan"), dtype=torch.bfloat16, device="cuda")
    K1, V1 = Q1.clone(), Q1.clone()
    raw = ops.gemm_qkv_norm_rope(probs(), [(wqx, wkx, cos, sin, N, 0), (wqc, wkc, None, None, Mt, N)], H, S, Q1, K1, V1)
    assert raw is not None, "MX operands at the MMDiT block shapes must take the fused path"
    assert torch.equal(raw[0][:, :2 * d], qkv_x[:, :2 * d]) and torch.equal(raw[1][:, :2 * d], qkv_c[:, :2 * d])
    for a, b_, name in ((Q1, Q2, "Q"), (K1, K2, "K"), (V1, V2, "V")):
        assert torch.isfinite(a.float()).all(), name
        frac = float((a.view(torch.int16) != b_.view(torch.int16)).float().mean())
        assert frac < 1e-4 and rel(a, b_) < 1e-4, (name, frac, rel(a, b_))
    assert torch.equal(V1, V2)
    # raw=False (inference): C = NULL, the raw q / k columns are not written -- the same Q, K, V to the bit
    Q3 = torch.full((Bt, H, S, 64), float("nan"), dtype=torch.bfloat16, device="cuda")
    K3, V3 = Q3.clone(), Q3.clone()
    assert ops.gemm_qkv_norm_rope(probs(), [(wqx, wkx, cos, sin, N, 0), (wqc, wkc, None, None, Mt, N)], H, S, Q3, K3, V3, raw=False) == [None, None]
    assert torch.equal(Q3, Q1) and torch.equal(K3, K1) and torch.equal(V3, V1)


def test_gemm_lean_weight_gradient_kernel(ops):
    """gemm_kk_kernel (csrc/gemm_lean.hip: both operands k-major, fp32 out, 256x256 tiles) on the schedules it runs: whole-K rounds
    only, rounds + a split tail (atomic partial tiles into the pre-zeroed output), the balanced tail of a block's mixed image + text
    launch, a caller-requested split-K, accumulation into an existing gradient, and ragged M / N edges -- against fp64 references of the
    same bf16 operands, and against the general kernel (MMDIT_GEMM_KK=0 is read once per process, so that comparison is the reference)."""
    def make(shapes, seed, **kw):
        probs, refs = [], []
        for i, (rows, M, N) in enumerate(shapes):
            dY, X = rnd(rows, M, seed=seed + 2 * i, dtype=torch.bfloat16), rnd(rows, N, seed=seed + 2 * i + 1, dtype=torch.bfloat16)
            probs.append(dict(A=dY, B=X, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, **kw))
            refs.append(dY.double().T @ X.double())
        return probs, refs

    def plan(probs):
        arr = (ops.GemmArgs * len(probs))()
        for i, p in enumerate(probs):
            ops._fill_gemm(arr[i], **p)
        return ops._lib.lib().mmdit_gemm_plan(arr, len(probs))

    if os.environ.get("MMDIT_GEMM_KK") == "0":
        pytest.skip("the lean weight-gradient kernel is switched off (MMDIT_GEMM_KK=0)")
    # (a) a block-like group: 4 "image" + 4 "text" problems, 288 tiles of 256x256 -> one round + balanced split tail
    shapes = [(4096, 2304, 768), (2432, 2304, 768), (4096, 768, 768), (2432, 768, 768), (4096, 6144, 768), (2432, 6144, 768), (4096, 768, 3072), (2432, 768, 3072)]
    probs, refs = make(shapes, 100, stream_k=True)
    code = plan(probs)
    assert code & 128 and code & 32 and (code & 15) == 2, code          # lean kernel, round + tail schedule, 256x256
    outs = ops.gemm_grouped(probs)
    for o, r in zip(outs, refs):
        assert rel(o, r) < 1e-5
    # (b) one round exactly / ragged edges (M = 200, N = 328: partial tiles in both directions) / K = 64 (a single K tile)
    for shapes in ([(1024, 2048, 2048)], [(1024, 200, 328), (640, 520, 72)], [(64, 512, 256)]):
        probs, refs = make(shapes, 200, stream_k=True)
        if shapes[0][0] >= 512:
            assert plan(probs) & 128, (shapes, plan(probs))
        for o, r in zip(ops.gemm_grouped(probs), refs):
            assert rel(o, r) < 1e-5, shapes
    # (c) caller-requested split-K (every tile in three atomic slices)
    probs, refs = make([(1536, 512, 512)], 300, split_k=3)
    for o, r in zip(ops.gemm_grouped(probs), refs):
        assert rel(o, r) < 1e-5
    # (d) accumulation into an existing gradient (second micro-batch of an accumulated step)
    probs, refs = make([(2048, 2304, 3072), (2048, 6144, 768)], 400)          # 180 tiles of 256x256: one round
    base = [rnd(2304, 3072, seed=410), rnd(6144, 768, seed=411)]
    for p, b in zip(probs, base):
        p["out"], p["accumulate"] = b.clone(), True
    assert plan(probs) & 128, plan(probs)
    for o, r, b in zip(ops.gemm_grouped(probs), refs, base):
        assert rel(o, r + b.double()) < 1e-5


@pytest.mark.parametrize("M,h,K,bias", [(1000, 256, 128, True), (16384, 3072, 768, True), (300, 128, 64, False)])
def test_gemm_swiglu_epilogue_equals_gemm_plus_row_kernel(ops, M, h, K, bias):
    """act=ACT_SWIGLU (activation formed in the w12 GEMM's epilogue) must be BIT-identical to the plain bf16 GEMM followed by
    mmdit_swiglu_fwd: same pre-activations [g | u], same h = silu(g) * u; and both agree with an fp32 torch reference of
    F.linear + silu * (MLP.py:15-40) within bf16 rounding (4e-3).  Ragged M, grouped launch of two problems."""
    probs, refs = [], []
    for s in range(2):
        Mr = M if s == 0 else M // 2 + 3
        X, W = rnd(Mr, K, seed=70 + s, dtype=torch.bfloat16), rnd(2 * h, K, seed=72 + s, scale=0.05, dtype=torch.bfloat16)
        b = rnd(2 * h, seed=74 + s) if bias else None
        gu = ops.gemm(X, W, bias=b, out_dtype=torch.bfloat16)
        refs.append((gu, ops.mlp_act_fwd(gu, h, False), X, W, b))
        probs.append(dict(A=X, B=W, bias=b, act=ops.ACT_SWIGLU, aux=torch.empty((Mr, 2 * h), dtype=torch.bfloat16, device="cuda")))
    outs = ops.gemm_grouped(probs)
    for p, o, (gu, hh, X, W, b) in zip(probs, outs, refs):
        assert o.shape == hh.shape and torch.equal(p["aux"], gu) and torch.equal(o, hh)
        pre = X.float() @ W.float().T + (b if b is not None else 0.0)
        assert rel(o, F.silu(pre[:, :h]) * pre[:, h:]) < 4e-3
    # inference form: no pre-activation output, same activation
    o2 = ops.gemm_grouped([dict(A=p["A"], B=p["B"], bias=p["bias"], act=ops.ACT_SWIGLU) for p in probs])
    assert all(torch.equal(a, b) for a, b in zip(o2, outs))
    if K % 128 == 0:   # e4m3 operands (fp8 inference mode): again bit-identical to the fp8 GEMM followed by the row kernel
        for p in probs:
            qa, sa = ops.quant_fp8(p["A"])
            qw, sw = ops.quant_fp8(p["B"])
            gu8 = ops.gemm(qa, qw, bias=p["bias"], out_dtype=torch.bfloat16, scale_a=sa, scale_b=sw)
            h8 = ops.gemm(qa, qw, bias=p["bias"], act=ops.ACT_SWIGLU, scale_a=sa, scale_b=sw)
            assert torch.equal(h8, ops.mlp_act_fwd(gu8, h, False))
    with pytest.raises(RuntimeError):     # hidden not a multiple of 128: unsupported shape, the caller keeps the two-kernel path
        ops.gemm(rnd(64, 64, dtype=torch.bfloat16), rnd(2 * 72, 64, dtype=torch.bfloat16), act=ops.ACT_SWIGLU, aux=torch.empty((64, 144), dtype=torch.bfloat16, device="cuda"))


def test_gemm_wgrad_reduction_length_not_a_multiple_of_the_k_tile(ops):
    """Weight gradients dW = dY^T X whose reduction length (rows = tokens x batch) is NOT a multiple of the 64-deep K tile -- the text stream:
    154 x 16 = 2464 rows at MMDiT-L batch 16, 154 x 13 = 2002 at the reference's own batch (src/train.py:13) -- stay on the 8-phase weight-gradient
    kernel (its K-tail instantiation zero-fills the k-rows beyond K on their way into LDS) instead of the register-staged kernel: plan bit 256, results
    against float64 products of the same bf16 operands.  Grouped with whole-K-tile problems as in a block's launch; K-decomposed (stream_k) and plain;
    accumulation into an existing gradient; a K below one tile falls back (no fast path, still correct)."""
    def make(shapes, seed, **kw):
        probs, refs = [], []
        for i, (rows, M, N) in enumerate(shapes):
            dY, X = rnd(rows, M, seed=seed + 2 * i, dtype=torch.bfloat16), rnd(rows, N, seed=seed + 2 * i + 1, dtype=torch.bfloat16)
            probs.append(dict(A=dY, B=X, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, **kw))
            refs.append(dY.double().T @ X.double())
        return probs, refs

    def plan(probs):
        arr = (ops.GemmArgs * len(probs))()
        for i, p in enumerate(probs):
            ops._fill_gemm(arr[i], **p)
        return ops._lib.lib().mmdit_gemm_plan(arr, len(probs))

    # (a) an MMDiT-L block's group: image rows 16384, text rows 2464 (= 38 K tiles + 32 rows), K-decomposed
    shapes = [(16384, 1024, 1024), (2464, 1024, 1024), (16384, 3072, 1024), (2464, 3072, 1024), (2464, 1024, 4096), (2464, 8192, 1024)]
    probs, refs = make(shapes, 500, stream_k=True)
    assert plan(probs) & 256, plan(probs)
    for o, r in zip(ops.gemm_grouped(probs), refs):
        assert rel(o, r) < 1e-5
    # (b) the reference's batch (154 x 13 = 2002 rows: 31 tiles + 18 rows) at its trained width, plain (not K-decomposed) and grouped with image rows
    probs, refs = make([(3328, 1216, 1216), (2002, 1216, 1216), (3328, 4864, 1216), (2002, 4864, 1216), (2002, 1216, 9728)], 520)
    assert plan(probs) & 256, plan(probs)
    for o, r in zip(ops.gemm_grouped(probs), refs):
        assert rel(o, r) < 1e-5
    # ... and shapes for which the planner does not pick 256 x 256 tiles: whatever kernel runs, the result is right (ragged output, one tile + 36 rows)
    for rows, M, N in [(2002, 456, 264), (100, 512, 256), (2464, 1024, 1024)]:
        probs, refs = make([(rows, M, N)], 525)
        assert rel(ops.gemm_grouped(probs)[0], refs[0]) < 1e-5
    # (c) accumulation into an existing gradient
    probs, refs = make([(2464, 1024, 1024)], 530, stream_k=True)
    base = rnd(1024, 1024, seed=531)
    probs[0]["out"], probs[0]["accumulate"] = base.clone(), True
    assert rel(ops.gemm_grouped(probs)[0], refs[0] + base.double()) < 1e-5
    # (c2) few output tiles, so that the K decomposition cuts every tile into MORE slices than it has K tiles (empty slices behind the last one)
    probs, refs = make([(2464, 256, 128), (2464, 384, 256), (2100, 128, 128)], 535, stream_k=True)
    assert plan(probs) & 256, plan(probs)
    for o, r in zip(ops.gemm_grouped(probs), refs):
        assert rel(o, r) < 1e-5
    # (d) fewer rows than one K tile: not the LDS-DMA kernels, still right
    probs, refs = make([(40, 256, 256)], 540)
    assert not plan(probs) & 256
    assert rel(ops.gemm_grouped(probs)[0], refs[0]) < 1e-5


@pytest.mark.parametrize("M,h,K", [(16384, 3072, 768), (3000, 512, 128), (700, 264, 64)])
def test_gemm_swiglu_bwd_epilogue_equals_gemm_plus_row_kernel(ops, M, h, K):
    """act=ACT_SWIGLU_BWD (the SwiGLU backward formed in the epilogue of the down-projection's data-gradient GEMM) must be BIT-identical
    to the plain bf16 GEMM dh = dY W3 followed by mmdit_swiglu_bwd, the bias gradients equal up to the order of the fp32 atomics, and
    both agree with autograd through silu(g) * u -> F.linear (MLP.py:15-40) within bf16 rounding.  Ragged M and h, grouped launch."""
    probs, refs = [], []
    for s in range(2):
        Mr = M if s == 0 else M // 2 + 5
        dY, W3 = rnd(Mr, K, seed=80 + s, dtype=torch.bfloat16), rnd(K, h, seed=82 + s, scale=0.05, dtype=torch.bfloat16)   # nn.Linear(h, K).weight
        gu = rnd(Mr, 2 * h, seed=84 + s, dtype=torch.bfloat16)
        dh = ops.gemm(dY, W3, b_kmajor=True, out_dtype=torch.bfloat16)
        db = torch.zeros(2 * h, device="cuda")
        refs.append((ops.mlp_act_bwd(dh, gu, h, db, False), db))
        probs.append(dict(A=dY, B=W3, aux=gu, dbias=torch.zeros(2 * h, device="cuda")))
    outs = ops.gemm_swiglu_bwd(probs)
    assert outs is not None
    for p, o, (dgu, db) in zip(probs, outs, refs):
        assert o.shape == dgu.shape and torch.equal(o, dgu)
        assert rel(p["dbias"], db) < 1e-5
        gr = p["aux"].float().requires_grad_(True)
        g, u = gr.chunk(2, -1)
        F.linear(F.silu(g) * u, p["B"].float()).backward(p["A"].float())
        assert rel(o, gr.grad) < 6e-3 and rel(p["dbias"], gr.grad.sum(0)) < 6e-3
    # a shape the LDS-DMA kernels do not take (K not a multiple of 64): None, the caller keeps the two passes
    assert ops.gemm_swiglu_bwd([dict(A=rnd(64, 72, dtype=torch.bfloat16), B=rnd(72, 128, dtype=torch.bfloat16), aux=rnd(64, 256, dtype=torch.bfloat16))]) is None


def test_gemm_block_wgrads_balanced_tail(ops):
    """All eight weight gradients of one MMDiT-B block at batch 64 (image K = 16384, text K = 9856; 288 tiles of 256x256 on 256
    workgroups): one full round plus a split tail that goes to the workgroups holding the short (text) tiles.  Given in
    image/text-interleaved order (the library sorts by K); every product vs an fp32 reference of the same bf16 operands."""
    Mx, Mc, d, h = 16384, 9856, 768, 3072
    probs, refs = [], []
    for i, (N, K) in enumerate([(3 * d, d), (d, d), (2 * h, d), (d, h)]):
        for Mr in (Mx, Mc):
            dY, X = rnd(Mr, N, seed=60 + 2 * i + (Mr == Mc), dtype=torch.bfloat16), rnd(Mr, K, seed=80 + 2 * i + (Mr == Mc), dtype=torch.bfloat16)
            probs.append(dict(A=dY, B=X, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, stream_k=True))
            refs.append(dY.float().T @ X.float())
    for _ in range(2):   # twice: the partial tiles are added atomically into freshly zeroed outputs each time
        outs = ops.gemm_grouped(probs)
        for o, r in zip(outs, refs):
            assert rel(o, r) < 5e-5


@pytest.mark.parametrize("budget", [248, 224, 64])
def test_gemm_cu_budget(ops, budget):
    """mmdit_set_cu_budget (data parallel: compute units left to the collectives' kernels): the planner and every persistent grid count on `budget`
    CUs instead of 256.  The results do not depend on it: a forward launch (bit-identical: same tiles, same K order), the SwiGLU-fused launch, and a
    block's eight weight gradients (round + balanced split tail, whose decomposition follows the budget: compared with the fp32 reference)."""
    from sd3_amd import _lib
    L = _lib.lib()
    assert L.mmdit_get_cu_budget() == 256
    M, d, h = 26240, 768, 3072
    A, W, W12, b12 = rnd(M, d, seed=1, dtype=torch.bfloat16), rnd(3 * d, d, seed=2, dtype=torch.bfloat16), rnd(2 * h, d, seed=3, dtype=torch.bfloat16), rnd(2 * h, seed=4)
    y0 = ops.gemm(A, W, out_dtype=torch.bfloat16)
    h0 = ops.gemm(A, W12, bias=b12, act=ops.ACT_SWIGLU)
    probs, refs = [], []
    for i, (N, K) in enumerate([(3 * d, d), (d, d), (2 * h, d), (d, h)]):
        for Mr in (16384, 9856):
            dY, X = rnd(Mr, N, seed=60 + 2 * i + (Mr == 9856), dtype=torch.bfloat16), rnd(Mr, K, seed=80 + 2 * i + (Mr == 9856), dtype=torch.bfloat16)
            probs.append(dict(A=dY, B=X, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, stream_k=True))
            refs.append(dY.float().T @ X.float())
    try:
        assert L.mmdit_set_cu_budget(250) != 0 and L.mmdit_set_cu_budget(32) != 0      # a multiple of 8 in [64, 256]
        assert L.mmdit_set_cu_budget(budget) == 0 and L.mmdit_get_cu_budget() == budget
        assert torch.equal(ops.gemm(A, W, out_dtype=torch.bfloat16), y0)
        assert torch.equal(ops.gemm(A, W12, bias=b12, act=ops.ACT_SWIGLU), h0)
        for _ in range(2):
            for o, r in zip(ops.gemm_grouped(probs), refs):
                assert rel(o, r) < 5e-5
    finally:
        assert L.mmdit_set_cu_budget(256) == 0


def _claiming_cases(ops):
    """The persistent launches of a block at MMDiT-B batch 64 (more tiles than compute units): forward Linear, SwiGLU up-projection, SwiGLU-backward data
    gradient, a data gradient, and the block's eight weight gradients as one grouped launch.  Returns name -> thunk returning the launch's outputs."""
    M, d, h = 26240, 768, 3072
    A, W, W12, b12 = rnd(M, d, seed=1, dtype=torch.bfloat16), rnd(3 * d, d, seed=2, dtype=torch.bfloat16), rnd(2 * h, d, seed=3, scale=0.05, dtype=torch.bfloat16), rnd(2 * h, seed=4)
    dY, W3, GU = rnd(M, d, seed=5, dtype=torch.bfloat16), rnd(d, h, seed=6, scale=0.05, dtype=torch.bfloat16), rnd(M, 2 * h, seed=7, dtype=torch.bfloat16)
    dH = rnd(M, 2 * h, seed=8, dtype=torch.bfloat16)
    probs = []
    for i, (N, K) in enumerate([(3 * d, d), (d, d), (2 * h, d), (d, h)]):
        for Mr in (16384, 9856):
            probs.append(dict(A=rnd(Mr, N, seed=60 + 2 * i + (Mr == 9856), dtype=torch.bfloat16), B=rnd(Mr, K, seed=80 + 2 * i + (Mr == 9856), dtype=torch.bfloat16),
                              a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, stream_k=True))
    # the fused QKV launch (QK-norm + RoPE + joint-layout store in the epilogue), image + text rows: with claiming on the planner gives it to the
    # 8-phase kernel (927 claimed tiles), with claiming off to the wide-slot kernel
    B, Ni, Mt, H = 64, 256, 154, 12
    Ac = rnd(B * Mt, d, seed=9, dtype=torch.bfloat16)
    Wc = rnd(3 * d, d, seed=10, dtype=torch.bfloat16)
    wqk = [rnd(64, seed=11 + i).abs() + 0.5 for i in range(4)]
    ang = rnd(Ni, 64, seed=15)
    rc, rs = torch.cos(ang).contiguous(), torch.sin(ang).contiguous()

    def qkv():
        Q = torch.zeros((B, H, Ni + Mt, 64), dtype=torch.bfloat16, device="cuda")
        K, V = torch.zeros_like(Q), torch.zeros_like(Q)
        raw = ops.gemm_qkv_norm_rope([dict(A=A[:B * Ni], B=W, out_dtype=torch.bfloat16), dict(A=Ac, B=Wc, out_dtype=torch.bfloat16)],
                                     [(wqk[0], wqk[1], rc, rs, Ni, 0), (wqk[2], wqk[3], None, None, Mt, Ni)], H, Ni + Mt, Q, K, V)
        assert raw is not None
        return [raw[0][:, :2 * d], raw[1][:, :2 * d], Q, K, V]      # (the v columns of the raw projection are not written)

    return {
        "qkv": qkv,
        "linear": lambda: [ops.gemm(A, W, out_dtype=torch.bfloat16)],
        "swiglu": lambda: [ops.gemm(A, W12, bias=b12, act=ops.ACT_SWIGLU)],
        "swiglu_bwd": lambda: ops.gemm_swiglu_bwd([dict(A=dY, B=W3, aux=GU)]),
        "dgrad": lambda: [ops.gemm(dH, W12, b_kmajor=True, out_dtype=torch.bfloat16)],
        "wgrad": lambda: ops.gemm_grouped(probs),
        "wgrad224": lambda: _with_wgrad_budget(ops, 224, probs),      # the data-parallel plan: 224 whole-K tiles + 64 split tail tiles on the whole-device grid
    }


def _with_wgrad_budget(ops, budget, probs):
    before, ops.WGRAD_CU_BUDGET = ops.WGRAD_CU_BUDGET, budget
    try:
        return ops.gemm_grouped(probs)
    finally:
        ops.WGRAD_CU_BUDGET = before


def test_gemm_dynamic_tile_claiming_equals_the_static_walk(ops):
    """csrc/gemm8p.hip "dynamic tile claiming": with the workspace registered the persistent 8-phase launches pop their tiles from per-XCD queues instead of
    walking them in a fixed stride.  Which tiles exist and what each computes is the planner's: the bf16 outputs are BIT-IDENTICAL to the static walk
    (no workspace), the weight gradients (whose split tail goes through fp32 atomics without the workspace) equal to rounding; launch after launch (the
    queue heads are reset by the last workgroup to leave: thirty launches in a row through one scheduler-slot ring) the results do not change."""
    from sd3_amd import _lib
    L = _lib.lib()
    cases = _claiming_cases(ops)
    dev = torch.device("cuda", torch.cuda.current_device())
    before = L.mmdit_gemm_get_claiming()      # (per-device setting: a data-parallel trainer of an earlier test leaves it on)
    assert L.mmdit_gemm_set_claiming(0) == 0 and L.mmdit_gemm_get_claiming() == 0
    static = {k: f() for k, f in cases.items() if not k.startswith("wgrad")}      # claiming off: the static walk (ops registers the workspace at its first launch)
    assert L.mmdit_gemm_set_claiming(1) == 0 and L.mmdit_gemm_get_claiming() == 1
    try:
        _claiming_equals_static(ops, L, cases, dev, static)
    finally:
        torch.cuda.synchronize()
        assert L.mmdit_gemm_set_claiming(before) == 0


def _claiming_equals_static(ops, L, cases, dev, static):
    dyn = {k: f() for k, f in cases.items()}
    for a, b in zip(dyn["wgrad"], dyn["wgrad224"]):      # two decompositions of the same weight gradients (288 = 256 + 32 x S and 224 + 64 x S tiles)
        assert rel(a, b) < 1e-5
    for k, outs in static.items():
        for a, b in zip(outs, dyn[k]):      # (the fused QKV launch changes KERNEL with the switch -- wide-slot vs 8-phase: equal to bf16 rounding, not bit for bit)
            assert (torch.equal(a, b) if k != "qkv" else rel(a.float(), b.float()) < 4e-3), k
    ws = ops._GEMM_WS[dev]
    words = ws[:8192].view(torch.int32)
    left = [(i, int(words[i])) for i in torch.nonzero(words).flatten().tolist()]
    assert not left, f"tickets / queue heads are left zero by every launch; nonzero (word, value): {left[:16]}"
    for _ in range(30):
        for k, f in cases.items():
            for a, b in zip(f(), dyn[k]):
                assert torch.equal(a, b), k
    assert int(ws[:8192].view(torch.int32).abs().sum()) == 0
    torch.cuda.synchronize()
    assert L.mmdit_gemm_set_workspace(None, 0) == 0
    try:
        for k, f in cases.items():      # (ops does not register again: it has done so once for this device)
            for a, b in zip(f(), dyn[k]):
                assert (torch.equal(a, b) if not k.startswith("wgrad") else rel(a, b) < 1e-5), k
    finally:
        torch.cuda.synchronize()
        assert L.mmdit_gemm_set_workspace(ws.data_ptr(), ws.numel()) == 0


@pytest.mark.parametrize("held", [8, 32])
def test_gemm_dynamic_tile_claiming_beside_an_occupant_kernel(ops, held):
    """A kernel on another stream holds `held` compute units (mmdit_debug_occupy: one-wave workgroups with a little LDS, the stand-in for a collective's
    channels) while the persistent launches run: the workgroups that cannot become resident find the queues empty when they finally start.  Results are
    bit-identical to the undisturbed launches (weight gradients included: their split tail is summed in slice order), and the heads are clean afterwards."""
    from sd3_amd import _lib
    L = _lib.lib()
    cases = _claiming_cases(ops)
    before = L.mmdit_gemm_get_claiming()
    assert L.mmdit_gemm_set_claiming(1) == 0
    try:
        ref = {k: f() for k, f in cases.items()}
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        for k, f in cases.items():
            assert L.mmdit_debug_occupy(held, int(3e-3 * 2.0e9), side.cuda_stream) == 0      # ~3 ms at ~2 GHz: longer than any of these launches
            out = f()
            torch.cuda.synchronize()
            for a, b in zip(out, ref[k]):
                assert torch.equal(a, b), k
        dev = torch.device("cuda", torch.cuda.current_device())
        assert int(ops._GEMM_WS[dev][:8192].view(torch.int32).abs().sum()) == 0
    finally:
        torch.cuda.synchronize()
        assert L.mmdit_gemm_set_claiming(before) == 0


@pytest.mark.parametrize("M,N,K,out_dtype", [(256, 256, 128, torch.float32), (1000, 768, 768, torch.bfloat16), (16384, 2304, 768, torch.bfloat16),
                                             (9000, 1304, 256, torch.bfloat16)])     # (the last: ragged M and N tiles of the 8-phase kernel's per-tensor form)
def test_gemm_fp8_operands(ops, M, N, K, out_dtype):
    """fp8 (e4m3, per-tensor scale) operand GEMM of the inference path vs the same quantised values multiplied in fp32."""
    g = torch.Generator(device="cuda").manual_seed(0)
    A = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
    W = (torch.randn((N, K), generator=g, device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device="cuda")
    qa, sa = ops.quant_fp8(A)
    qw, sw = ops.quant_fp8(W)
    # quantisation itself: dequantised values within e4m3 precision of the input, scale = amax / 448
    assert abs(float(sa) - float(A.float().abs().max()) / 448.0) < 1e-6 * float(sa) + 1e-12
    assert rel(qa.float() * sa, A.float()) < 4e-2
    y = ops.gemm(qa, qw, bias=bias, out_dtype=out_dtype, scale_a=sa, scale_b=sw)
    ref = (qa.float() * sa) @ (qw.float() * sw).t() + bias
    assert rel(y.float(), ref) < (5e-3 if out_dtype == torch.bfloat16 else 1e-5)
    # and it is a faithful approximation of the bf16 product
    assert rel(y.float(), A.float() @ W.float().t() + bias) < 6e-2


@pytest.mark.parametrize("workspace", [False, True])
def test_gemm_zero_mask_marks_exactly_the_outputs_that_receive_atomics(ops, workspace):
    """mmdit_gemm_zero_mask: in a K-decomposed grouped launch (a block's weight gradients) only the problems that own tiles of the
    split tail are accumulated atomically.  Every other output is pre-filled with NaN here and must come out fully overwritten and
    correct; the flagged ones start from zero.  With the split-tail workspace registered (mmdit_gemm_set_workspace: partial tiles through
    per-slice slots, the last slice to arrive sums them) NO output needs a zero-fill, the result is the same and it is deterministic."""
    import ctypes
    from sd3_amd import _lib
    ws = torch.zeros(8192 + 256 * 65536 * 4, dtype=torch.uint8, device="cuda") if workspace else None
    prev = ops._GEMM_WS.get(torch.device("cuda", torch.cuda.current_device()))
    assert _lib.lib().mmdit_gemm_set_workspace(ws.data_ptr() if workspace else None, ws.numel() if workspace else 0) == 0
    try:
        _zero_mask_case(ops, workspace)
    finally:        # back to the process-wide workspace of ops (or none)
        torch.cuda.synchronize()
        assert _lib.lib().mmdit_gemm_set_workspace(prev.data_ptr() if prev is not None else None, prev.numel() if prev is not None else 0) == 0


def _zero_mask_case(ops, workspace):
    import ctypes
    from sd3_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(9)
    rnd = lambda *s: torch.randn(s, generator=g, device="cuda").to(torch.bfloat16)
    Mx, Mc, d = 4096, 2432, 768
    probs = []
    for N, K in ((3 * d, d), (d, d), (8 * d, d), (d, 4 * d)):
        for Mr in (Mx, Mc):
            probs.append(dict(A=rnd(Mr, N), B=rnd(Mr, K), a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, stream_k=True))
    n = len(probs)
    arr = (_lib.GemmArgs * n)()
    outs = [torch.empty((p["A"].shape[1], p["B"].shape[1]), dtype=torch.float32, device="cuda") for p in probs]
    for i, p in enumerate(probs):
        ops._fill_gemm(arr[i], out=outs[i], **p)
    mask = ctypes.c_uint(0)
    assert _lib.lib().mmdit_gemm_zero_mask(arr, n, ctypes.byref(mask)) == 0
    flagged = [(mask.value >> i) & 1 for i in range(n)]
    if workspace:
        assert sum(flagged) == 0, flagged                   # nothing is accumulated in place
    else:
        assert 0 < sum(flagged) < n, flagged                # 288 tiles on 256 workgroups: a tail exists, and it is not everything
    runs = []
    for _ in range(2):
        for o, f in zip(outs, flagged):
            o.fill_(0.0 if f else float("nan"))
        # (straight through the C ABI: ops.gemm_grouped would register the process-wide workspace)
        assert _lib.lib().mmdit_gemm_grouped(arr, n, torch.cuda.current_stream().cuda_stream) == 0
        runs.append([o.clone() for o in outs])
    for o, p in zip(outs, probs):
        ref = p["A"].float().t() @ p["B"].float()
        assert torch.isfinite(o).all() and rel(o, ref) < 2e-5
    if workspace:       # fixed summation order of the slices: bit-identical from launch to launch (the atomics are not)
        assert all(torch.equal(a, b) for a, b in zip(*runs))


@pytest.mark.parametrize("shape", [(4, 64, 39, 256), (64, 256, 154, 768), (16, 1024, 154, 1024)])
@pytest.mark.parametrize("res", [False, True])
def test_ln_modulate_pair_launch_equals_two_launches(ops, res, shape):
    """mmdit_ln_modulate_fwd_pair / _bwd_pair (image + text stream of a block in one launch) are the single-problem kernels run on two
    problems: outputs bit-identical, the atomically accumulated per-sample sums equal up to the order of the atomics.  The MMDiT-B batch-64
    and MMDiT-L batch-16 shapes take the backward launcher's one-round sizing (28 rows per workgroup instead of 16)."""
    g = torch.Generator(device="cuda").manual_seed(11)
    B, N, Mt, d = shape
    rnd = lambda *s: torch.randn(s, generator=g, device="cuda")
    P = []
    for rpb in (N, Mt):
        rows = B * rpb
        P.append(dict(x=rnd(rows, d) * 2, scale=rnd(B, d) * 0.3, shift=rnd(B, d) * 0.3, rpb=rpb, acc=rnd(rows, d).to(torch.bfloat16), gate=rnd(B, d),
                      dout=rnd(rows, d).to(torch.bfloat16), dres=rnd(rows, d)))
    fw = lambda p: (ops.ln_modulate_fwd_res(p["x"], p["acc"], p["gate"], p["scale"], p["shift"], p["rpb"], torch.bfloat16) if res
                    else (p["x"],) + ops.ln_modulate_fwd(p["x"], p["scale"], p["shift"], p["rpb"], torch.bfloat16))
    single = [fw(p) for p in P]
    sel = lambda p: {k: p[k] for k in (("x", "scale", "shift", "rpb", "acc", "gate") if res else ("x", "scale", "shift", "rpb"))}
    pair = ops.ln_modulate_fwd_pair(sel(P[0]), sel(P[1]), torch.bfloat16)
    for a, b in zip(single, pair):
        for u, v in zip(a, b):
            assert torch.equal(u, v)
    # backward
    outs = {}
    for mode in ("single", "pair"):
        args = []
        for p, f in zip(P, single):
            x1, _, mean, rstd = f
            a = dict(dout=p["dout"], x=x1, mean=mean, rstd=rstd, scale=p["scale"], dres=p["dres"], rpb=p["rpb"],
                     dscale=torch.zeros_like(p["scale"]), dshift=torch.zeros_like(p["scale"]))
            if res:
                a["gated"] = (p["acc"], p["gate"], torch.zeros_like(p["gate"]), torch.zeros_like(p["gate"]))
            args.append(a)
        if mode == "single":
            r = [ops.ln_modulate_bwd(a["dout"], a["x"], a["mean"], a["rstd"], a["scale"], a["dres"], a["rpb"], a["dscale"], a["dshift"], gated=a.get("gated")) for a in args]
        else:
            r = ops.ln_modulate_bwd_pair(args[0], args[1])
        outs[mode] = (r, args)
    for (ra, aa), (rb, ab) in zip(zip(*outs["single"]), zip(*outs["pair"])):
        for u, v in zip(ra if isinstance(ra, tuple) else (ra,), rb if isinstance(rb, tuple) else (rb,)):
            assert torch.equal(u, v)
        for k in ("dscale", "dshift"):
            assert rel(ab[k], aa[k]) < 1e-5
        if res:
            assert rel(ab["gated"][2], aa["gated"][2]) < 1e-5 and rel(ab["gated"][3], aa["gated"][3]) < 1e-5


def _mx_reference(x):
    """torch restatement of mmdit_mxfp8_quantize: (e4m3 codes as uint8 (rows, K), E8M0 bytes in the GEMM layout [K/64][rows][2], dequantised fp32)."""
    rows, K = x.shape
    xb = x.float().reshape(rows, K // 32, 32)
    amax = xb.abs().amax(-1, keepdim=True)
    _, ex = torch.frexp(amax)
    e = torch.where(amax > 0, ex - 1 - 8, torch.full_like(ex, -127))
    e = torch.where(amax > 448.0 * torch.ldexp(torch.ones_like(amax), e), e + 1, e).clamp(-127, 127)     # amax / scale <= 448: nothing saturates
    scale = torch.ldexp(torch.ones_like(amax), e)
    q = (xb / scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    sc = (e + 127).to(torch.uint8).reshape(rows, K // 32)                  # per (row, 32-block); the device layout is compared through ops.mx_scales_to_rows
    return q.reshape(rows, K).view(torch.uint8), sc, (q.float() * scale).reshape(rows, K)


@pytest.mark.parametrize("M,N,K,out_dtype", [(256, 256, 128, torch.float32), (1000, 768, 768, torch.bfloat16), (308, 2304, 768, torch.float32),
                                             (16384, 2304, 768, torch.bfloat16), (9000, 1304, 256, torch.bfloat16)])     # (the last: ragged M and N tiles of the 8-phase MX kernel)
def test_gemm_mxfp8_operands(ops, M, N, K, out_dtype):
    """MX (block-scaled e4m3) operand GEMM: the quantiser is bit-identical to its torch restatement (codes and E8M0 scales, in the
    GEMM's scale layout), and the GEMM -- block scales applied by v_mfma_scale_f32_32x32x64_f8f6f4 -- equals the fp32 product of
    the dequantised operands.  Blocks get magnitudes spread over 2^-20 .. 2^20 (plus an all-zero block) to exercise the scales."""
    g = torch.Generator(device="cuda").manual_seed(0)
    A = torch.randn((M, K), generator=g, device="cuda")
    A = (A.reshape(M, K // 32, 32) * torch.exp2(torch.randint(-20, 21, (M, K // 32, 1), generator=g, device="cuda").float())).reshape(M, K)
    A[3, 32:64] = 0.0
    A = A.to(torch.bfloat16)
    W = (torch.randn((N, K), generator=g, device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device="cuda")
    qa, sa = ops.quant_mxfp8(A)
    qw, sw = ops.quant_mxfp8(W)
    for (q, sc, x) in ((qa, sa, A), (qw, sw, W)):
        q_ref, sc_ref, deq = _mx_reference(x)
        assert torch.equal(q.view(torch.uint8), q_ref) and torch.equal(ops.mx_scales_to_rows(sc, *x.shape), sc_ref)
        assert rel(deq, x.float()) < 4e-2
    y = ops.gemm(qa, qw, bias=bias, out_dtype=out_dtype, scale_a=sa, scale_b=sw, scale_mode=1)
    ref = _mx_reference(A)[2].double() @ _mx_reference(W)[2].double().t() + bias.double()
    r = rel(y.double(), ref)
    print(f"[mxfp8 gemm] {M}x{N}x{K} -> {out_dtype}: rel err vs the fp64 product of the dequantised operands = {r:.3e}")
    assert r < (5e-3 if out_dtype == torch.bfloat16 else 5e-5)      # (fp32 accumulation over blocks whose magnitudes span 2^40)
    assert rel(y.float(), A.float() @ W.float().t() + bias) < 6e-2


def test_gemm_mxfp8_swiglu_epilogue(ops):
    """The packed w12 GEMM with the SwiGLU epilogue on MX operands (the gate / up rows of a tile are two separate runs of scale rows)."""
    g = torch.Generator(device="cuda").manual_seed(2)
    M, h, K = 1024, 512, 256
    A = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
    W = (torch.randn((2 * h, K), generator=g, device="cuda") * torch.exp2(torch.randint(-6, 3, (2 * h, 1), generator=g, device="cuda").float()) * 0.1).to(torch.bfloat16)
    bias = torch.randn(2 * h, generator=g, device="cuda")
    qa, sa = ops.quant_mxfp8(A)
    qw, sw = ops.quant_mxfp8(W)
    y = ops.gemm(qa, qw, bias=bias, act=ops.ACT_SWIGLU, scale_a=sa, scale_b=sw, scale_mode=1)
    gu = (_mx_reference(A)[2] @ _mx_reference(W)[2].t() + bias).to(torch.bfloat16).float()
    ref = torch.nn.functional.silu(gu[:, :h]) * gu[:, h:]
    assert rel(y.float(), ref) < 6e-3
    # MX in, MX out: the same activation leaving the epilogue as e4m3 codes + block scales == quantising the bf16 output afterwards
    q, sc = ops._mx_buffers(M, h, "cuda")
    ops.gemm(qa, qw, bias=bias, act=ops.ACT_SWIGLU, scale_a=sa, scale_b=sw, scale_mode=1, out=q, out_scales=sc)
    q_ref, sc_ref = ops.quant_mxfp8(y)
    assert torch.equal(q.view(torch.uint8), q_ref.view(torch.uint8))
    assert torch.equal(ops.mx_scales_to_rows(sc, M, h), ops.mx_scales_to_rows(sc_ref, M, h))


def test_mx_producers_match_bf16_output_plus_quantise_pass(ops):
    """The MX-producing variants of adaLN (plain and with the pending gated residual), SwiGLU and attention forward emit exactly
    the codes and scale bytes that their bf16 outputs followed by mmdit_mxfp8_quantize give (the out-projection / QKV / MLP GEMMs
    of the "mxfp8" mode then run without quantise passes)."""
    g = torch.Generator(device="cuda").manual_seed(5)
    B, N, d, hid = 4, 64, 256, 512
    rows = B * N
    x = torch.randn((rows, d), generator=g, device="cuda") * 3
    sc, sh = torch.randn((B, d), generator=g, device="cuda") * 0.3, torch.randn((B, d), generator=g, device="cuda") * 0.3
    acc = torch.randn((rows, d), generator=g, device="cuda").to(torch.bfloat16)
    gate = torch.randn((B, d), generator=g, device="cuda")

    def same(mx, ref_bf16):
        q, s = ops.quant_mxfp8(ref_bf16)
        assert torch.equal(mx.q.view(torch.uint8), q.view(torch.uint8))
        assert torch.equal(ops.mx_scales_to_rows(mx.sc, *q.shape), ops.mx_scales_to_rows(s, *q.shape))
        assert rel(mx.dequant(), ref_bf16.float()) < 4e-2

    out, mean, rstd = ops.ln_modulate_fwd(x, sc, sh, N, torch.bfloat16)
    x1, mx, mean2, rstd2 = ops.ln_modulate_fwd_mx(x, sc, sh, N)
    same(mx, out)
    assert x1 is x and torch.equal(mean, mean2) and torch.equal(rstd, rstd2)
    xr, outr, _, _ = ops.ln_modulate_fwd_res(x, acc, gate, sc, sh, N, torch.bfloat16)
    xr2, mxr, _, _ = ops.ln_modulate_fwd_mx(x, sc, sh, N, acc=acc, gate=gate)
    same(mxr, outr)
    assert torch.equal(xr, xr2)
    gu = torch.randn((rows, 2 * hid), generator=g, device="cuda").to(torch.bfloat16)
    same(ops.swiglu_fwd_mx(gu, hid), ops.mlp_act_fwd(gu, hid, False))
    H, M = 4, 27                                                   # S = 91: ragged tiles, 108 text rows (no multiple of 8)
    Q, K, V = (torch.randn((B, H, N + M, 64), generator=g, device="cuda").to(torch.bfloat16) for _ in range(3))
    Ox, Oc, _ = ops.attn_fwd(Q, K, V, N, 0.125, 0)
    mxx, mxc = ops.attn_fwd_mx(Q, K, V, N, 0.125)
    same(mxx, Ox.view(B * N, H * 64))
    same(mxc, Oc.view(B * M, H * 64))


def test_fp8_delayed_scaling_site(ops):
    """One-pass quantiser with delayed scaling: call k uses margin x amax(call k-1); values above that range saturate at +-448."""
    g = torch.Generator(device="cuda").manual_seed(1)
    site = ops.Fp8Site(margin=1.5)
    x0 = torch.randn((512, 256), generator=g, device="cuda").to(torch.bfloat16)
    q0, s0 = site.quantise(x0)                       # first call: measured amax, no margin
    assert abs(float(s0) * 448.0 - float(x0.float().abs().max())) < 1e-3 and rel(q0.float() * s0, x0.float()) < 4e-2
    x1 = x0 * 1.3                                    # grows, but stays inside the 1.5x margin of the previous amax
    q1, s1 = site.quantise(x1)
    assert abs(float(s1) - 1.5 * float(x0.float().abs().max()) / 448.0) < 1e-6 and rel(q1.float() * s1, x1.float()) < 5e-2
    x2 = x1 * 4.0                                    # outgrows the margin: saturates instead of overflowing to NaN
    q2, s2 = site.quantise(x2)
    assert abs(float(s2) - 1.5 * float(x1.float().abs().max()) / 448.0) < 1e-6
    d2 = q2.float() * s2
    assert torch.isfinite(d2).all() and float(d2.abs().max()) <= 1.5 * float(x1.float().abs().max()) * (1 + 1e-6)
    q3, s3 = site.quantise(x0)                       # the ring has moved on to amax(x2)
    assert abs(float(s3) - 1.5 * float(x2.float().abs().max()) / 448.0) < 1e-5


# ------------------------------------------------------------------------------------ optimizer
def test_clip_adamw_matches_torch_sequence(ops):
    """ClipAdamW.step_clipped (three HIP launches) vs the reference's sequence on the same scaled gradients: unscale (divide
    by the power-of-two loss scale), torch.nn.utils.clip_grad_norm_(1.0), torch.optim.AdamW(fused).step(), skipped when a
    gradient is inf.  Two param groups (different lr / weight decay), ragged and unaligned sizes, lr changing per step.
    Tolerance 2e-6 relative on parameters and moments after 5 steps; step counters and found_inf exact."""
    from sd3_amd.optim import ClipAdamW
    shapes = [(1,), (7,), (768, 768), (65536 + 3,), (200001,), (3, 65536), (64, 768)]
    base = [rnd(*s, seed=100 + i) for i, s in enumerate(shapes)]
    # views at odd element offsets of one arena: the gradients of the engine are such views (no 16-byte alignment guaranteed)
    arena = torch.zeros(sum(b.numel() for b in base) + 1, device="cuda")

    def make(cls):
        ps = [torch.nn.Parameter(b.clone()) for b in base]
        return ps, cls([dict(params=ps[:4], weight_decay=0.01), dict(params=ps[4:], weight_decay=0.1, lr=3e-4)], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, fused=True)

    pa, oa = make(ClipAdamW)
    pb, ob = make(torch.optim.AdamW)
    scale = torch.tensor(1024.0, device="cuda")
    for step, (mag, poison) in enumerate([(1e-3, False), (2.0, False), (1.0, True), (0.5, False), (1e-2, False)]):
        for grp_a, grp_b in zip(oa.param_groups, ob.param_groups):
            grp_a["lr"] = grp_b["lr"] = grp_b["lr"] * 0.9
        off = 1
        for i, (qa, qb) in enumerate(zip(pa, pb)):
            g = rnd(*qa.shape, seed=1000 + 10 * step + i) * mag * 1024.0
            if poison and i == 2:
                g.view(-1)[5] = float("inf")
            qa.grad = arena[off:off + g.numel()].view(g.shape)
            qa.grad.copy_(g)
            off += g.numel()
            qb.grad = g / 1024.0
        found_inf, norm = oa.step_clipped(scale, 1.0)
        total = torch.nn.utils.clip_grad_norm_(pb, 1.0)
        if bool(torch.isfinite(total)):
            ob.step()
        assert float(found_inf) == (1.0 if poison else 0.0)
        if not poison:
            assert abs(float(norm) - float(total)) < 1e-5 * float(total)
        for qa, qb in zip(pa, pb):
            sa, sb = oa.state[qa], ob.state[qb]
            assert float(sa["step"]) == float(sb["step"]) if len(sb) else float(sa["step"]) == 0.0
    for qa, qb in zip(pa, pb):
        assert rel(qa, qb) < 2e-6
        assert rel(oa.state[qa]["exp_avg"], ob.state[qb]["exp_avg"]) < 2e-6
        assert rel(oa.state[qa]["exp_avg_sq"], ob.state[qb]["exp_avg_sq"]) < 2e-6
    assert float(oa.state[pa[0]]["step"]) == 4.0
    # interchangeable state: torch's state_dict loads into ClipAdamW and the next step agrees again
    pc, oc = make(ClipAdamW)
    with torch.no_grad():
        for qc, qb in zip(pc, pb):
            qc.copy_(qb)
    import copy
    oc.load_state_dict(copy.deepcopy(ob.state_dict()))   # (load_state_dict keeps same-dtype/device tensors by reference)
    assert set(oa.state_dict()["state"][0].keys()) == set(ob.state_dict()["state"][0].keys())
    for i, (qc, qb) in enumerate(zip(pc, pb)):
        qb.grad = rnd(*qb.shape, seed=5000 + i) * 0.1
        qc.grad = qb.grad.clone()
    oc.step_clipped(None, None)      # no loss scale, no clipping: plain AdamW
    ob.step()
    for qc, qb in zip(pc, pb):
        assert rel(qc, qb) < 2e-6
        assert float(oc.state[qc]["step"]) == float(ob.state[qb]["step"]) == 5.0
    # the step counters are views of one flat tensor and survive a state_dict round trip
    sd = copy.deepcopy(oc.state_dict())
    assert all(float(s["step"]) == 5.0 and s["step"].dim() == 0 for s in sd["state"].values())


def test_clip_adamw_rewrites_bf16_operand_copies(ops):
    """The update kernel also rewrites the packed bf16 GEMM-operand copy of every parameter that lives in exactly one pack
    (mmdit_adamw_tensor.shadow_bf16): after step_clipped those packs are current without a refresh pass and hold exactly
    bf16(parameter); a parameter shared by two packs is left to the regular refresh."""
    from sd3_amd import engine, packing
    from sd3_amd.optim import ClipAdamW
    ps = [torch.nn.Parameter(rnd(768, 768, seed=1, scale=0.02)), torch.nn.Parameter(rnd(64, 768, seed=2, scale=0.02)),
          torch.nn.Parameter(rnd(40, 72, seed=3, scale=0.02))]
    pk1, pk2, pk3 = packing.Pack(ps[:2]), packing.Pack([ps[2]]), packing.Pack([ps[2]])
    for pk in (pk1, pk2, pk3):
        pk.get(engine.FAST)
    opt = ClipAdamW(ps, lr=1e-2)
    before = [p.detach().clone() for p in ps]
    for i, p in enumerate(ps):
        p.grad = rnd(*p.shape, seed=10 + i)
    opt.step_clipped(None, 1.0)
    assert all(float((p.detach() - b).abs().max()) > 1e-3 for p, b in zip(ps, before))
    assert pk1._key == pk1._state_of(True) and pk2._key != pk2._state_of(True) and pk3._key != pk3._state_of(True)
    assert torch.equal(pk1.get(engine.FAST), torch.cat([ps[0].detach(), ps[1].detach()], 0).to(torch.bfloat16))
    assert torch.equal(pk2.get(engine.FAST), ps[2].detach().to(torch.bfloat16)) and pk3._key == pk3._state_of(True)   # one refresh served both
    assert torch.equal(pk3.get(engine.FAST), ps[2].detach().to(torch.bfloat16))
    # an inf step leaves parameters and copies untouched
    ps[0].grad.view(-1)[0] = float("inf")
    snap = pk1._buf.clone()
    found_inf, _ = opt.step_clipped(None, 1.0)
    assert float(found_inf) == 1.0 and torch.equal(pk1.get(engine.FAST), snap)


def test_flow_loss_matches_torch_expression():
    """ops.flow_loss (mmdit_flow_loss) == the trainer's torch expression (reference model_trainer.py:429-446):
    mean((v - (eps - x0))^2) / accumulation_steps with the bf16 label rounding of torch's bf16 subtraction, and its gradient;
    bit-reproducible from call to call (fixed-order reduction, no atomics)."""
    from sd3_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    for shape, dt, accum in (((64, 16, 32, 32), torch.bfloat16, 1), ((4, 16, 16, 16), torch.bfloat16, 2), ((3, 16, 12, 20), torch.float32, 1)):
        v = torch.randn(shape, generator=g, device="cuda")
        x0 = torch.randn(shape, generator=g, device="cuda").to(dt)
        eps = torch.randn(shape, generator=g, device="cuda").to(dt)
        vr = v.clone().requires_grad_(True)
        ref = torch.nn.MSELoss(reduction="none")(vr, (eps - x0).to(torch.float32)).flatten(1, -1).mean() / accum
        ref.backward()
        loss, dv = ops.flow_loss(v, x0, eps, accum)
        loss2, dv2 = ops.flow_loss(v, x0, eps, accum)
        assert torch.equal(loss, loss2) and torch.equal(dv, dv2)
        assert abs(float(loss) - float(ref)) <= 2e-6 * abs(float(ref)), (float(loss), float(ref))
        assert float((dv - vr.grad).abs().max()) <= 1e-6 * float(vr.grad.abs().max())
        assert ops.flow_loss(v, x0, eps, accum, need_grad=False)[1] is None
    with pytest.raises(RuntimeError):
        ops.flow_loss(v, x0.to(torch.bfloat16), eps, 1)


def test_trainer_hip_loss_equals_torch_loss_step():
    """model_trainer(hip_loss=True) and (hip_loss=False) take the same two optimizer steps (same seeds): equal losses to fp32
    summation order, parameters to the backward's own run-to-run noise."""
    import sd3_amd  # noqa: F401
    from sd3_amd.model_trainer import model_trainer
    from sd3_amd.models.diff_model import diff_model
    res = []
    for hip_loss in (True, False):
        torch.manual_seed(0)
        net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu",
                         device=torch.device("cuda:0"), positional_encoding="RoPE2d", dim=128, num_heads=2, num_blocks=3)
        net.load_state_dict(make_state_dict(0, dim=128, num_heads=2, num_blocks=3))
        tr = model_trainer(net, batchSize=4, accumulation_steps=2, totalSteps=10, lr=1e-3, ema_update_freq=1, ema_decay=0.9, warmup_steps=2,
                           use_lr_scheduler=False, device=torch.device("cuda:0"), saveDir="/tmp/_t", numSaveSteps=100, max_res=128,
                           device_rng=True, use_ema=False, hip_loss=hip_loss)
        losses = [float(tr.train_step(s)) for s in (1, 2)]
        torch.cuda.synchronize()
        res.append((losses, [p.detach().clone() for p in net.parameters()]))
    assert np.allclose(res[0][0], res[1][0], rtol=1e-5), (res[0][0], res[1][0])
    for a, b in zip(res[0][1], res[1][1]):
        assert float((a - b).abs().max()) <= 2e-3 * float(b.abs().max()) + 1e-6


def test_weight_gradient_flush_groups_by_k_alignment():
    """engine._wgrad_flush: a block's weight gradients dW = dY^T X in grouped launches.  The text stream's 154 * batch rows are not a
    multiple of the 64-deep K tile for most batch sizes; such problems get a launch of their own (the 8-phase kernel's K-tail instantiation
    since round 4, the register-staged kernel before).  All outputs must equal the fp32 products of the bf16 operands."""
    from sd3_amd import engine
    got = {}
    shapes = [(16384, 256, 128), (2464, 256, 128), (2464, 384, 256), (4096, 128, 384), (2100, 128, 128), (64, 256, 128)]
    ops_ = []
    for i, (rows, n, k) in enumerate(shapes):
        dY, X = rnd(rows, n, seed=20 + i).to(torch.bfloat16), rnd(rows, k, seed=40 + i).to(torch.bfloat16)
        ops_.append((dY, X))
    pending = [((lambda o, i=i: got.__setitem__(i, o)), engine._wg(dY, X)) for i, (dY, X) in enumerate(ops_)]
    arena = engine._wgrad_flush(engine.FAST, pending)
    torch.cuda.synchronize()
    assert arena is not None and arena.numel() % engine.ARENA_QUANTUM == 0
    for i, (dY, X) in enumerate(ops_):
        ref = dY.float().t() @ X.float()
        assert got[i].shape == ref.shape and got[i].data_ptr() >= arena.data_ptr()
        assert rel(got[i], ref) < 2e-5, (shapes[i], rel(got[i], ref))
