"""Explicit forward / backward kernel schedules of the MMDiT path.

Instead of composing many small autograd ops, each stage below is a fixed sequence of launches of
the HIP kernels in libmmdit_hip.so with hand-written backward schedules, so that residual adds,
gating, bias and modulation gradients are fused into the kernels that already touch the data.
PyTorch only owns the buffers.  Reference call stack being replaced: diff_model.forward
(models/diff_model.py:264-346) -> Transformer_Block_Dual.forward (blocks/Transformer_Block_Dual.py:56-77)
-> Attention.forward (blocks/Attention.py:118-427) / MLP.forward (blocks/MLP.py:25-40) / Norm.forward
(blocks/Norm.py:16-22), and their autograd mirrors.

Launch shaping for 256 CUs: the image and text streams of a block go through ONE grouped GEMM launch
per stage, and all weight-gradient GEMMs of a block (tiny output grids, long reductions) are deferred
and issued as ONE grouped launch at the end of the block's backward.

Precision modes:
  fast   : bf16 activations / bf16 MFMA operands, fp32 accumulation, fp32 residual stream, flash attention.
  parity : fp32 activations, 3-term split-bf16 (fp32-exact) MFMA GEMMs, attention core that reproduces the
           rounding points of the reference's CPU branch (Attention.py:277-284).  For the 1e-3 golden check.
  mxfp8  : as fp8 with MX block scales (one E8M0 scale per 32 K values, applied by the matrix instruction; stateless one-pass
           quantisation of the activations).
  fp8    : fast, plus e4m3 operands (per-tensor scales, delayed activation scaling) for the QKV / out / MLP GEMMs of every
           block.  Forward only (the sampler, BASELINE config 5).
"""
from types import SimpleNamespace as NS

import torch

from . import _lib, ops
from ._lib import ACT_SILU, PREC_BF16, PREC_SPLIT

F32 = torch.float32
BF16 = torch.bfloat16


class Mode:
    def __init__(self, fast: bool = True, fp8: bool = False, mx: bool = False):
        self.fast = fast
        self.fp8 = fp8          # inference only: the four big GEMMs of a block take e4m3 operands (per-tensor scales)
        self.mx = mx            # ... with MX block scales (E8M0 per 32 K values) instead of per-tensor scales
        self.T = BF16 if fast else F32
        self.prec = PREC_BF16 if fast else PREC_SPLIT
        self.attn_mode = 0 if fast else 1
        self._q = {}            # id(packed bf16 weight) -> (weight, fp8 copy, scale); weights are static while sampling

    def act(self, t):
        """Tensor as a GEMM A operand in this mode's activation dtype."""
        return t if t.dtype == self.T else ops.cast(t, self.T)


FAST = Mode(True)
PARITY = Mode(False)
FP8 = Mode(True, fp8=True)
MXFP8 = Mode(True, fp8=True, mx=True)

MOD_NAMES_FULL = ["shift1x", "scale1x", "gate1x", "shift2x", "scale2x", "gate2x", "shift1c", "scale1c", "gate1c", "shift2c", "scale2c", "gate2c"]


def _mod_views(mod, d, last):
    names = MOD_NAMES_FULL[:8] if last else MOD_NAMES_FULL
    return NS(**{n: mod[:, i * d:(i + 1) * d] for i, n in enumerate(names)})


def _gemm(m, A, B, **kw):
    return ops.gemm(A, B, precision=m.prec, **kw)


def _to_fp8(m, p):
    """fp8 inference mode: quantise the activation on the fly (per-tensor amax -> e4m3), take the cached e4m3 copy of the
    weight, drop the backward-only aux output.  Problems whose K is not a multiple of the 128-wide fp8 K-tile stay bf16."""
    A, B = p["A"], p["B"]
    pre = isinstance(A, ops.MxAct)       # the producer already emitted MX e4m3 (block_fwd checked the site's eligibility)
    if not pre and (A.dtype != BF16 or B.dtype != BF16 or A.shape[1] % 128 or p.get("a_kmajor") or p.get("b_kmajor")):
        return p
    ent, gen = m._q.get(id(B)), getattr(B, "_mmdit_gen", 0)     # (the bf16 copy is refreshed in place: packing.Pack.generation)
    if m.mx:    # MX: block scales, stateless one-pass quantisation of the activation
        if not pre and not A.is_contiguous():
            return p
        if ent is None or ent[4] != gen:
            qb, sb = ops.quant_mxfp8(B)
            ent = m._q[id(B)] = (B, qb, sb, None, gen)
        qa, sa = (A.q, A.sc) if pre else ops.quant_mxfp8(A)
        q = {k: v for k, v in p.items() if k != "aux"}
        q.update(A=qa, B=ent[1], scale_a=sa, scale_b=ent[2], scale_mode=1)
        return q
    if ent is None or ent[4] != gen:
        qb, sb = ops.quant_fp8(B)
        ent = m._q[id(B)] = (B, qb, sb, ent[3] if ent is not None else ops.Fp8Site(), gen)
    qa, sa = ent[3].quantise(A)      # delayed scaling: consecutive sampler steps see nearly the same activation range
    q = {k: v for k, v in p.items() if k != "aux"}
    q.update(A=qa, B=ent[1], scale_a=sa, scale_b=ent[2])
    return q


def _group(m, problems, fp8=False):
    """One grouped launch; problems = list of dicts (A, B, + gemm kwargs).  fp8: eligible for e4m3 operands in FP8 mode."""
    if fp8 and m.fp8:
        q = [_to_fp8(m, p) for p in problems]
        if all(x is not p for x, p in zip(q, problems)):     # one kernel variant per launch: all problems in e4m3, or none
            problems = q
    for p in problems:
        p["precision"] = m.prec
    return ops.gemm_grouped(problems)


def _wg(dY, X):
    """Problem descriptor of dW[N,K] = dY[M,N]^T X[M,K] (fp32 out, zero-initialised: stream-K adds partial tiles)."""
    # long reductions (rows >= 2048) are decomposed over K (MMDIT_WGRAD_MODE: streamk (default; measured 10.2 ms/step vs 14.4 splitk4 / 23.7 splitk8: fp32 atomics dominate split-K) | splitk<N> | plain);
    # the per-sample GEMMs (rows = batch) stay regular
    d = dict(A=dY, B=X, a_kmajor=True, b_kmajor=True, out_dtype=F32)
    if dY.shape[0] >= 2048 and dY.dtype == BF16:
        if _WG_MODE == "streamk":
            d["stream_k"] = True
        elif _WG_MODE.startswith("splitk"):
            d["split_k"] = int(_WG_MODE[6:])
    return d


def _wgrad(m, dY, X):
    return _group(m, [_wg(dY, X)])[0]


def _dgrad(m, dY, W, out_dtype, **kw):
    """dX[M,K] = dY[M,N] W[N,K]."""
    return ops.gemm(dY, W, b_kmajor=True, out_dtype=out_dtype, precision=m.prec, **kw)


# Weight gradients are off the critical path of the backward pass (only the optimizer consumes them): the deferred grouped
# wgrad launch of a block CAN run on a side HIP stream (MMDIT_WGRAD_STREAM=1).  Measured in round 2 (same box, A/B): 37.45 vs
# 37.41 ms/step before and 1909 vs 1922 img/s after the 320x256-tile kernel -- no gain: the persistent 256x256 wgrad kernel owns
# every CU (128 KB LDS, 2 x 243 VGPRs per SIMD), so main-stream kernels queue behind it instead of overlapping with it (the row
# kernels showed 17 -> 116 us in the profile).  Default: one stream (clean per-kernel durations, graph-capturable step).
_WG_MODE = _lib.experiment("MMDIT_WGRAD_MODE", "streamk")
_WG_OVERLAP = _lib.experiment("MMDIT_WGRAD_STREAM", "0") == "1"
_FUSE_SWIGLU = _lib.experiment("MMDIT_FUSE_SWIGLU", "1") != "0"   # SwiGLU activation in the up-projection GEMM's epilogue (A/B switch)
_wg_streams = {}


def wgrad_stream(device):
    st = _wg_streams.get(device)
    if st is None:
        st = _wg_streams[device] = torch.cuda.Stream(device=device)
    return st


def wgrad_join(device):
    """Make the current stream wait for every weight-gradient launch issued so far."""
    st = _wg_streams.get(device)
    if st is not None:
        torch.cuda.current_stream(device).wait_stream(st)


ARENA_QUANTUM = 1024     # flat gradient arenas are multiples of this many elements: a reducer can cut them into equal shards for any
                         # world size that divides it (reducer.GradReducer: reduce-scatter / all-gather algorithms)


def _zeros_views(shapes, dev, prefix=None):
    """One zero-filled fp32 arena (one memset launch) carved into views of the given shapes.
    prefix=k: also return the flat sub-arena that holds exactly the first k views (a data-parallel reducer averages the
    parameter gradients of a block in place through it)."""
    sizes = [int(torch.Size(s).numel()) for s in shapes]
    pad = [(n + 3) // 4 * 4 for n in sizes]           # keep every view 16-byte aligned
    if prefix is not None:                            # the sub-arena of the first `prefix` views: a multiple of ARENA_QUANTUM (zero padding)
        pad[prefix - 1] += -sum(pad[:prefix]) % ARENA_QUANTUM
    flat = ops.zeros(sum(pad), dev)                   # (a slice of the pass-wide zero pool on a GPU)
    out, off = [], 0
    for s, n, p in zip(shapes, sizes, pad):
        out.append(flat[off:off + n].view(s))
        off += p
    if prefix is not None:
        return out, flat[:sum(pad[:prefix])]
    return out


def _wgrad_flush(m, pending):
    """pending: list of (setter, descriptor).  Grouped launches (<= 12 problems each; on the side stream with MMDIT_WGRAD_STREAM=1).
    On a GPU every output is a view of ONE flat fp32 arena, in `pending` order (a data-parallel reducer averages the block's weight
    gradients in place through it: no gather copy); the arena is NOT zero-filled -- ops.gemm_grouped asks the planner which
    outputs receive atomic partial tiles and zeroes only those."""
    arena = None
    if not pending:
        return arena
    gpu = pending[0][1]["A"].is_cuda
    overlap = _WG_OVERLAP and gpu
    if gpu:
        dev = pending[0][1]["A"].device
        shapes = [(d["A"].shape[1], d["B"].shape[1]) for _, d in pending]
        sizes = [(a * b + 3) // 4 * 4 for a, b in shapes]          # 16-byte aligned slices
        arena = torch.empty((sum(sizes) + ARENA_QUANTUM - 1) // ARENA_QUANTUM * ARENA_QUANTUM, dtype=F32, device=dev)   # (the tail padding is never read)
        off = 0
        for (_, d), (a, b), n in zip(pending, shapes, sizes):
            d["out"] = arena[off:off + a * b].view(a, b)
            d["out"]._mmdit_zero_check = arena     # (the arena itself: neighbouring outputs that need a zero-fill share one launch)
            off += n
    if overlap:
        # inputs must outlive the side-stream reads (record_stream below)
        side = wgrad_stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        ctx = torch.cuda.stream(side)
    else:
        ctx = _NullCtx()
    with ctx:
        # one launch per kernel variant: K-decomposed (long reductions) / regular, and reductions whose length is a multiple of the 64-deep K
        # tile apart from the others (the text stream's 154 * batch rows are not, for a per-GPU batch that is not a multiple of 32): the
        # unaligned group runs on the 8-phase kernel's K-tail instantiation (csrc/gemm8p.hip KT; round 4 -- before, it fell to the
        # register-staged kernel), the aligned group on the plain one.  Merging the two launches gains nothing: MMDiT-L's 256 image tiles are
        # exactly one round, the 256 short text tiles a second one either way.
        for flag in (True, False):
            for aligned in (True, False):
                part = [pd for pd in pending if (bool(pd[1].get("stream_k")) or pd[1].get("split_k", 1) > 1) == flag and (pd[1]["A"].shape[0] % 64 == 0) == aligned]
                for i in range(0, len(part), 12):
                    chunk = part[i:i + 12]
                    outs = _group(m, [d for _, d in chunk])
                    for (setter, _), o in zip(chunk, outs):
                        setter(o)

    if overlap:
        for _, d in pending:
            d["A"].record_stream(side)
            d["B"].record_stream(side)
    return arena   # the flat buffer that holds every output of this flush (None: outputs were allocated one by one)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


# ----------------------------------------------------------------------------------------------
# MLP core (standalone MLP module):  h = act(x Wup^T + bup);  y = h Wdown^T + bdown
# ----------------------------------------------------------------------------------------------
def mlp_core_fwd(m, w, x_act):
    gu = _gemm(m, x_act, w.Wup, bias=w.bup, out_dtype=m.T)
    h = ops.mlp_act_fwd(gu, w.hidden, w.gelu)
    return gu, h


def mlp_core_bwd(m, w, dacc, x_act, gu, h, dev):
    dh = _dgrad(m, dacc, w.Wdown, m.T)
    dbup = ops.zeros(gu.shape[1], dev)
    dgu = ops.mlp_act_bwd(dh, gu, w.hidden, dbup, w.gelu)
    dx = _dgrad(m, dgu, w.Wup, m.T)
    ds = [_wg(dacc, h), _wg(dgu, x_act)]
    same = (ds[0].get("stream_k"), ds[0].get("split_k")) == (ds[1].get("stream_k"), ds[1].get("split_k"))
    gWdown, gWup = _group(m, ds) if same else (_group(m, ds[:1])[0], _group(m, ds[1:])[0])
    return dx, NS(Wup=gWup, bup=dbup, Wdown=gWdown)


# ----------------------------------------------------------------------------------------------
# Transformer_Block_Dual
# ----------------------------------------------------------------------------------------------
def cond_fwd_all(m, blocks, y):
    """y' = SiLU(W_y y + b) and the adaLN modulation vectors of ALL transformer blocks in two grouped launches: they only
    depend on the conditioning vector y (Transformer_Block_Dual.py:56-60), and as 2 x n_blocks separate M = batch GEMMs
    they are pure launch latency on the critical path (~55 us per block).  Returns [(pre, yp, mod)] per block; pre / yp of
    all blocks are row-stacked in one buffer each (cond_bwd_all runs one SiLU-backward launch over the stack)."""
    B, d = y.shape
    nb, dev = len(blocks), y.device
    pre_all = torch.empty((nb * B, d), dtype=F32, device=dev)
    yp_all = torch.empty((nb * B, d), dtype=m.T, device=dev)
    pres = [pre_all[i * B:(i + 1) * B] for i in range(nb)]
    yps = [yp_all[i * B:(i + 1) * B] for i in range(nb)]
    mods = []
    for i0 in range(0, nb, 12):
        sl = range(i0, min(nb, i0 + 12))
        _group(m, [dict(A=y, B=blocks[i].Wy, bias=blocks[i].by, act=ACT_SILU, aux=pres[i], out=yps[i]) for i in sl])
        mods += _group(m, [dict(A=yps[i], B=blocks[i].Wmod, out_dtype=F32) for i in sl])
    return [(pres[i], yps[i], mods[i]) for i in range(nb)], pre_all


class Pending:
    """A residual stream whose last gated update is still pending: value = x + gate[b] * acc (acc None: value = x).  The update is
    formed inside the next adaLN forward kernel that reads the stream (ops.ln_modulate_fwd_res), not in the fp32 epilogue of the
    projection GEMM that produced acc."""
    __slots__ = ("x", "acc", "gate", "rpb")

    def __init__(self, x, acc=None, gate=None, rpb=0):
        self.x, self.acc, self.gate, self.rpb = x, acc, gate, rpb

    def value(self):
        return self.x if self.acc is None else ops.gate_residual_fwd(self.x, self.acc, self.gate, self.rpb)


def _norm(m, P, scale, shift, rpb, mx=False):
    """adaLN of a (possibly pending) residual stream: returns (materialised stream fp32, normed, mean, rstd).  mx: the normed rows
    leave as MX e4m3 (ops.MxAct) for the fp8 GEMM that consumes them."""
    pend = isinstance(P, Pending) and P.acc is not None
    if mx:
        return ops.ln_modulate_fwd_mx(P.x if isinstance(P, Pending) else P, scale, shift, rpb, acc=P.acc if pend else None, gate=P.gate if pend else None)
    if pend:
        return ops.ln_modulate_fwd_res(P.x, P.acc, P.gate, scale, shift, rpb, m.T)
    x = P.x if isinstance(P, Pending) else P
    return (x,) + ops.ln_modulate_fwd(x, scale, shift, rpb, m.T)


def _norm_pair(m, a, b, mx=False):
    """adaLN of the image stream (a) and of the text stream (b) = (P, scale, shift, rpb) each: ONE launch when both are plain or both
    carry a pending gated residual update (ops.ln_modulate_fwd_pair; ~5 us less per pair than two launches), two launches otherwise."""
    pend = [isinstance(P, Pending) and P.acc is not None for P, *_ in (a, b)]
    dev = (a[0].x if isinstance(a[0], Pending) else a[0]).device
    if mx or pend[0] != pend[1] or dev.type != "cuda" or not _LN_PAIR:
        return _norm(m, *a, mx=mx), _norm(m, *b, mx=mx)
    args = []
    for (P, scale, shift, rpb), pe in zip((a, b), pend):
        d = dict(x=P.x if isinstance(P, Pending) else P, scale=scale, shift=shift, rpb=rpb)
        if pe:
            d.update(acc=P.acc, gate=P.gate)
        args.append(d)
    ra, rb = ops.ln_modulate_fwd_pair(args[0], args[1], m.T)
    return ra, rb


def _ln_bwd_pair(a, b):
    """adaLN backward of the image (a) and the text (b) stream: dicts of ops.ln_modulate_bwd_pair; one launch when both are gated or
    both are plain, two otherwise."""
    ga, gb = a.get("gated") is not None, b.get("gated") is not None
    if ga != gb or not a["x"].is_cuda or not _LN_PAIR:
        return [ops.ln_modulate_bwd(p["dout"], p["x"], p["mean"], p["rstd"], p["scale"], p["dres"], p["rpb"], p["dscale"], p["dshift"], gated=p.get("gated")) for p in (a, b)]
    return ops.ln_modulate_bwd_pair(a, b)


def block_fwd(m, w, X, C, y, dims, rope, cond=None, keep=True, lazy=False):
    """X (B*N,d) fp32, C (B*M,d) fp32 (or Pending streams), y (B,d) in m.T.  cond: this block's (pre, yp, mod) from cond_fwd_all.
    keep=False (inference): the backward-only GEMM side outputs (SwiGLU pre-activations) are not written.
    lazy=True: return the outputs as Pending streams (their MLP residual update is left to the consumer's adaLN kernel).
    Returns X2, C2, saved."""
    B, N, Mt, H, d = dims
    S, dev = N + Mt, (X.x if isinstance(X, Pending) else X).device
    both = not w.last
    sv = NS(y=y)
    if cond is not None:
        sv.pre, sv.yp, sv.mod = cond
    else:
        sv.pre = torch.empty((B, d), dtype=F32, device=dev)
        sv.yp = _gemm(m, y, w.Wy, bias=w.by, act=ACT_SILU, aux=sv.pre, out_dtype=m.T)
        sv.mod = _gemm(m, sv.yp, w.Wmod, out_dtype=F32)
    ms = _mod_views(sv.mod, d, w.last)
    # "mxfp8" inference: the activations that feed the four fp8 GEMM sites leave their producers (adaLN, attention, SwiGLU) as MX e4m3
    # -- no quantise passes -- when every site of the block is eligible (K % 128; SwiGLU MLPs)
    mxf = (m.mx and not keep and dev.type == "cuda" and d % 128 == 0 and not w.mlp_x.gelu
           and w.mlp_x.hidden % 128 == 0 and (w.last or (not w.mlp_c.gelu and w.mlp_c.hidden % 128 == 0)) and _MX_FUSE)

    (sv.X, sv.ln1x, sv.mu1x, sv.rs1x), (sv.C, sv.ln1c, sv.mu1c, sv.rs1c) = _norm_pair(m, (X, ms.scale1x, ms.shift1x, N), (C, ms.scale1c, ms.shift1c, Mt), mx=mxf)
    sv.Q = torch.empty((B, H, S, 64), dtype=BF16, device=dev)
    sv.K, sv.V = torch.empty_like(sv.Q), torch.empty_like(sv.Q)
    fused = None
    if _QKV_FUSE and m.fast and dev.type == "cuda" and m.T == BF16 and (not m.fp8 or mxf) and d == H * 64:
        # QK-norm + RoPE + joint-layout store in the QKV GEMM's epilogue (one launch, no second pass over the raw projection); None: the
        # planner would not give these problems to a kernel with that epilogue (bf16: the lean wide-slot kernel; MX operands: the 8-phase kernel)
        qp = [dict(A=sv.ln1x, B=w.Wqkv_x, out_dtype=m.T, precision=m.prec), dict(A=sv.ln1c, B=w.Wqkv_c, out_dtype=m.T, precision=m.prec)]
        if mxf:
            qq = [_to_fp8(m, p) for p in qp]
            qp = qq if all(x is not p for x, p in zip(qq, qp)) else None
        if qp is not None:
            # (MX operands = inference: the raw q / k columns, 2/3 of the projection's bytes, are saved for a backward that never comes -- not written)
            fused = ops.gemm_qkv_norm_rope(qp, [(w.wq_x, w.wk_x, rope[0], rope[1], N, 0), (w.wq_c, w.wk_c, None, None, Mt, N)], H, S, sv.Q, sv.K, sv.V, raw=_QKV_RAW or not mxf)
    if fused is not None:
        sv.qkv_x, sv.qkv_c = fused
    else:
        sv.qkv_x, sv.qkv_c = _group(m, [dict(A=sv.ln1x, B=w.Wqkv_x, out_dtype=m.T), dict(A=sv.ln1c, B=w.Wqkv_c, out_dtype=m.T)], fp8=True)
    if fused is not None:
        pass
    elif _LN_PAIR and dev.type == "cuda":      # image + text rows in one launch
        ops.qk_norm_rope_fwd_pair((sv.qkv_x, w.wq_x, w.wk_x, rope[0], rope[1], N, 0), (sv.qkv_c, w.wq_c, w.wk_c, None, None, Mt, N), B, H, S, sv.Q, sv.K, sv.V)
    else:
        ops.qk_norm_rope_fwd(sv.qkv_x, w.wq_x, w.wk_x, rope[0], rope[1], B, N, H, S, 0, sv.Q, sv.K, sv.V)
        ops.qk_norm_rope_fwd(sv.qkv_c, w.wq_c, w.wk_c, None, None, B, Mt, H, S, N, sv.Q, sv.K, sv.V)
    if mxf:
        sv.Oxa, sv.Oca = ops.attn_fwd_mx(sv.Q, sv.K, sv.V, N, 64 ** -0.5)
    else:
        sv.Ox, sv.Oc, sv.lse = ops.attn_fwd(sv.Q, sv.K, sv.V, N, 64 ** -0.5, m.attn_mode)
        sv.Oxa = m.act(sv.Ox.view(B * N, d))
        if both:
            sv.Oca = m.act(sv.Oc.view(B * Mt, d))

    # attention output projections: the GEMM writes acc in the activation dtype; X1 = X + gate1 * acc is formed by norm2's kernel
    probs = [dict(A=sv.Oxa, B=w.Wo_x, out_dtype=m.T)]
    if both:
        probs.append(dict(A=sv.Oca, B=w.Wo_c, out_dtype=m.T))
    outs = _group(m, probs, fp8=True)
    sv.acc_ox = outs[0]
    if both:
        sv.acc_oc = outs[1]
        (sv.X1, sv.ln2x, sv.mu2x, sv.rs2x), (sv.C1, sv.ln2c, sv.mu2c, sv.rs2c) = _norm_pair(
            m, (Pending(sv.X, sv.acc_ox, ms.gate1x, N), ms.scale2x, ms.shift2x, N), (Pending(sv.C, sv.acc_oc, ms.gate1c, Mt), ms.scale2c, ms.shift2c, Mt), mx=mxf)
    else:
        sv.C1 = sv.C
        sv.X1, sv.ln2x, sv.mu2x, sv.rs2x = _norm(m, Pending(sv.X, sv.acc_ox, ms.gate1x, N), ms.scale2x, ms.shift2x, N, mx=mxf)
    # SwiGLU in the up-projection's epilogue (bf16 mode, hidden % 128 == 0, K % 64 == 0): the GEMM writes the pre-activations and
    # the activation; otherwise (GELU, parity / fp8 mode, odd sizes) the activation is a row kernel over the GEMM output
    fuse = _FUSE_SWIGLU and m.fast and dev.type == "cuda" and not w.mlp_x.gelu and w.mlp_x.hidden % 128 == 0 and d % (128 if m.fp8 else 64) == 0

    def up(xn, mw, rows):
        if fuse and mxf:     # MX in, MX out: the activation leaves the epilogue as e4m3 codes + block scales (no quantise pass before w3)
            q, sc = ops._mx_buffers(rows, mw.hidden, dev)
            return dict(A=xn, B=mw.Wup, bias=mw.bup, act=ops.ACT_SWIGLU, out=q, out_scales=sc)
        if fuse:
            return dict(A=xn, B=mw.Wup, bias=mw.bup, act=ops.ACT_SWIGLU, aux=torch.empty((rows, 2 * mw.hidden), dtype=m.T, device=dev) if keep else None)
        return dict(A=xn, B=mw.Wup, bias=mw.bup, out_dtype=m.T)

    probs = [up(sv.ln2x, w.mlp_x, B * N)]
    if both:
        probs.append(up(sv.ln2c, w.mlp_c, B * Mt))
    outs = _group(m, probs, fp8=True)
    pre = [p.get("aux") for p in probs]
    if fuse and mxf:
        outs = [ops.MxAct(p["out"], p["out_scales"]) for p in probs]
    if fuse:
        sv.gu_x, sv.h_x = pre[0], outs[0]
    else:
        sv.gu_x = outs[0]
        sv.h_x = ops.swiglu_fwd_mx(sv.gu_x, w.mlp_x.hidden) if mxf else ops.mlp_act_fwd(sv.gu_x, w.mlp_x.hidden, w.mlp_x.gelu)
    probs = [dict(A=sv.h_x, B=w.mlp_x.Wdown, bias=w.mlp_x.bdown, out_dtype=m.T)]
    if both:
        if fuse:
            sv.gu_c, sv.h_c = pre[1], outs[1]
        else:
            sv.gu_c = outs[1]
            sv.h_c = ops.swiglu_fwd_mx(sv.gu_c, w.mlp_c.hidden) if mxf else ops.mlp_act_fwd(sv.gu_c, w.mlp_c.hidden, w.mlp_c.gelu)
        probs.append(dict(A=sv.h_c, B=w.mlp_c.Wdown, bias=w.mlp_c.bdown, out_dtype=m.T))
    outs = _group(m, probs, fp8=True)
    sv.acc_mx = outs[0]
    X2 = Pending(sv.X1, sv.acc_mx, ms.gate2x, N)
    if both:
        sv.acc_mc = outs[1]
        C2 = Pending(sv.C1, sv.acc_mc, ms.gate2c, Mt)
    else:
        C2 = Pending(sv.C1)
    if not lazy:
        X2, C2 = X2.value(), C2.value()
    return X2, C2, sv


_LN_PAIR = _lib.experiment("MMDIT_LN_PAIR", "1") != "0"      # image + text rows of the adaLN / QK-norm+RoPE / MLP-activation-backward kernels in one launch (A/B switch)
_MX_FUSE = _lib.experiment("MMDIT_MX_FUSE", "1") != "0"      # 0: quantise passes in front of the fp8 GEMMs (A/B measurements, tests)
_BATCH_WMOD = _lib.experiment("MMDIT_BATCH_WMOD", "1") != "0"   # single-rank backward: the modulation-matrix weight gradients of all blocks in one grouped launch (A/B switch)
_QKV_FUSE = _lib.experiment("MMDIT_QKV_FUSE", "1") != "0"        # QK-norm + RoPE + joint-layout store inside the QKV GEMM epilogue (A/B switch)
_QKV_RAW = _lib.experiment("MMDIT_QKV_RAW", "0") == "1"      # experiment: the fused MX QKV launch writes the raw q / k columns even when nobody reads them
_QK_FUSE = _lib.experiment("MMDIT_ATTN_QK_FUSE", "1") != "0"     # QK-norm + RoPE backward inside the attention backward kernels (A/B switch)
_FUSE_SWIGLU_BWD = _lib.experiment("MMDIT_FUSE_SWIGLU_BWD", "1") != "0"   # SwiGLU backward in the epilogue of the down-projection's data gradient (A/B switch)
_FUSE_GATE = _lib.experiment("MMDIT_FUSE_GATE", "1") != "0"   # gated-residual backward inside the adaLN backward that produces its input (A/B switch)


def _qk_bwd_two_pass(m, w, sv, g, dOx, dOc, dims, rope, dev):
    """Attention backward, then the QK-norm + RoPE backward as a pass of its own."""
    B, N, Mt, H, d = dims
    S = N + Mt
    dQ, dK, dV = ops.attn_bwd(sv.Q, sv.K, sv.V, sv.Ox, sv.Oc, dOx, dOc, sv.lse, N, 64 ** -0.5, m.T)
    if _LN_PAIR and dev.type == "cuda":
        return ops.qk_norm_rope_bwd_pair(dQ, dK, dV, (sv.qkv_x, w.wq_x, w.wk_x, rope[0], rope[1], N, 0, g.wq_x, g.wk_x),
                                         (sv.qkv_c, w.wq_c, w.wk_c, None, None, Mt, N, g.wq_c, g.wk_c), B, H, S, m.T)
    dqkv_x = ops.qk_norm_rope_bwd(dQ, dK, dV, sv.qkv_x, w.wq_x, w.wk_x, rope[0], rope[1], B, N, H, S, 0, g.wq_x, g.wk_x, m.T)
    dqkv_c = ops.qk_norm_rope_bwd(dQ, dK, dV, sv.qkv_c, w.wq_c, w.wk_c, None, None, B, Mt, H, S, N, g.wq_c, g.wk_c, m.T)
    return dqkv_x, dqkv_c


def block_bwd_begin(m, w, sv, dims, dev, defer_cond=False):
    """First half of a block's backward set-up: the zeroed arena of its small gradients and its modulation-gradient buffer.
    Split from block_bwd so that the PRODUCER of this block's incoming residual gradients (the adaLN backward of the following
    block, or of the output head) can already run the backward of this block's MLP gated-residual update, fused
    (st.req_x / st.req_c are the operands it needs: see ops.ln_modulate_bwd(gated=...))."""
    B, N, Mt, H, d = dims
    both = not w.last
    ms = _mod_views(sv.mod, d, w.last)
    g = NS(mlp_x=NS(), mlp_c=NS() if both else None)
    # every atomically-accumulated small gradient of the block comes from ONE zeroed arena (one memset launch)
    hx = sv.gu_x.shape[1]
    nb = 2 if both else 1
    # parameter gradients first (they form one flat sub-arena a reducer can average in place), scratch after them
    shapes = [(nb * d,), (hx,), (64,), (64,), (64,), (64,)] + ([(sv.gu_c.shape[1],)] if both else []) + ([] if defer_cond else [(d,)])
    npar = len(shapes)
    shapes += [tuple(sv.mod.shape), (B, nb * d)]
    zs, small_arena = _zeros_views(shapes, dev, prefix=npar)
    st = NS(g=g, ms=ms, zs=zs, npar=npar, small_arena=small_arena, both=both)
    st.bdown, g.mlp_x.bup, g.wq_x, g.wk_x, g.wq_c, g.wk_c = zs[:6]
    if not defer_cond:
        g.by = zs[npar - 1]
    st.dmod, st.bpart = zs[npar], zs[npar + 1]   # bpart: per-batch partial rows of the down-proj bias grads
    g.mlp_x.bdown = st.bdown[:d]
    if both:
        g.mlp_c.bdown, g.mlp_c.bup = st.bdown[d:], zs[6]
    st.dms = _mod_views(st.dmod, d, w.last)
    st.req_x = (sv.acc_mx, ms.gate2x, st.dms.gate2x, st.bpart[:, :d])
    st.req_c = (sv.acc_mc, ms.gate2c, st.dms.gate2c, st.bpart[:, d:]) if both else None
    return st


def block_bwd(m, w, sv, dX2, dC2, dy_acc, dims, rope, defer_cond=False, st=None, dacc=None, nxt=None, defer_wmod=False):
    """dX2 (B*N,d) fp32, dC2 (B*M,d) fp32 or None, dy_acc (B,d) fp32 or None.
    Returns dX, dC, dy_acc', grads (NS keyed like the packed weights).
    defer_cond: leave the backward of y' = SiLU(W_y y + b) to cond_bwd_all (one launch set for all blocks): the third
    return value is then this block's modulation gradient in the activation dtype, and grads has no Wy / by.
    Fused gate chain (model_bwd): st = block_bwd_begin(...) of this block, dacc = (dacc_x, dacc_c) already formed by the producer of
    dX2 / dC2, nxt = the begin-state of the block that consumes dX / dC; a fifth value, that block's (dacc_x, dacc_c), is returned."""
    B, N, Mt, H, d = dims
    S, dev = N + Mt, dX2.device
    both = not w.last
    if st is None:
        st = block_bwd_begin(m, w, sv, dims, dev, defer_cond)
    g, ms, dms, bpart, bdown, dmod = st.g, st.ms, st.dms, st.bpart, st.bdown, st.dmod
    pending = []   # deferred weight-gradient GEMMs: (setter, descriptor)
    fuse = _FUSE_GATE

    def defer(ns, name, dY, Xa):
        pending.append((lambda o, ns=ns, name=name: setattr(ns, name, o), _wg(dY, Xa)))

    # ---- MLP: gated residual -> down-proj -> activation -> up-proj -> adaLN
    dacc_x = dacc[0] if dacc is not None else ops.gate_residual_bwd(dX2, sv.acc_mx, ms.gate2x, N, dms.gate2x, bpart[:, :d], m.T)
    probs = [dict(A=dacc_x, B=w.mlp_x.Wdown, b_kmajor=True, out_dtype=m.T)]
    if both:
        dacc_c = dacc[1] if dacc is not None else ops.gate_residual_bwd(dC2, sv.acc_mc, ms.gate2c, Mt, dms.gate2c, bpart[:, d:], m.T)
        probs.append(dict(A=dacc_c, B=w.mlp_c.Wdown, b_kmajor=True, out_dtype=m.T))
    ops.colsum(bpart, bdown)   # finish both bias gradients: sum the per-batch partial rows
    # SwiGLU: the activation backward runs in the epilogue of this data-gradient GEMM (no dh round trip, no row-kernel launch) when the
    # planner gives the launch to the 8-phase kernel; otherwise GEMM + mlp_act_bwd
    dgu = None
    if (_FUSE_SWIGLU_BWD and m.fast and dev.type == "cuda" and m.prec == PREC_BF16 and dacc_x.dtype == BF16 and sv.gu_x.dtype == BF16
            and not w.mlp_x.gelu and (not both or not w.mlp_c.gelu)):
        fp = [dict(A=dacc_x, B=w.mlp_x.Wdown, aux=sv.gu_x, dbias=g.mlp_x.bup, precision=m.prec)]
        if both:
            fp.append(dict(A=dacc_c, B=w.mlp_c.Wdown, aux=sv.gu_c, dbias=g.mlp_c.bup, precision=m.prec))
        dgu = ops.gemm_swiglu_bwd(fp)
    if dgu is None:
        dh = _group(m, probs)
    defer(g.mlp_x, "Wdown", dacc_x, sv.h_x)
    pair = both and _LN_PAIR and dev.type == "cuda" and w.mlp_x.hidden == w.mlp_c.hidden and w.mlp_x.gelu == w.mlp_c.gelu
    if dgu is not None:
        dgu_x, dgu_c = dgu[0], (dgu[1] if both else None)
    elif pair:     # image + text activation backward in one launch
        dgu_x, dgu_c = ops.mlp_act_bwd_pair((dh[0], sv.gu_x, g.mlp_x.bup), (dh[1], sv.gu_c, g.mlp_c.bup), w.mlp_x.hidden, w.mlp_x.gelu)
    else:
        dgu_x = ops.mlp_act_bwd(dh[0], sv.gu_x, w.mlp_x.hidden, g.mlp_x.bup, w.mlp_x.gelu)
    probs = [dict(A=dgu_x, B=w.mlp_x.Wup, b_kmajor=True, out_dtype=m.T)]
    defer(g.mlp_x, "Wup", dgu_x, sv.ln2x)
    if both:
        defer(g.mlp_c, "Wdown", dacc_c, sv.h_c)
        if not pair and dgu is None:
            dgu_c = ops.mlp_act_bwd(dh[1], sv.gu_c, w.mlp_c.hidden, g.mlp_c.bup, w.mlp_c.gelu)
        probs.append(dict(A=dgu_c, B=w.mlp_c.Wup, b_kmajor=True, out_dtype=m.T))
        defer(g.mlp_c, "Wup", dgu_c, sv.ln2c)
    dln2 = _group(m, probs)

    # ---- adaLN backward (fused with the backward of the attention-output gated residual) -> attention output projections
    ax = dict(dout=dln2[0], x=sv.X1, mean=sv.mu2x, rstd=sv.rs2x, scale=ms.scale2x, dres=dX2, rpb=N, dscale=dms.scale2x, dshift=dms.shift2x)
    if fuse:
        ax["gated"] = (sv.acc_ox, ms.gate1x, dms.gate1x, None)
    dC1 = dC2
    if both:
        ac = dict(dout=dln2[1], x=sv.C1, mean=sv.mu2c, rstd=sv.rs2c, scale=ms.scale2c, dres=dC2, rpb=Mt, dscale=dms.scale2c, dshift=dms.shift2c)
        if fuse:
            ac["gated"] = (sv.acc_oc, ms.gate1c, dms.gate1c, None)
        rx, rc = _ln_bwd_pair(ax, ac)
    else:
        rx = ops.ln_modulate_bwd(ax["dout"], ax["x"], ax["mean"], ax["rstd"], ax["scale"], ax["dres"], N, ax["dscale"], ax["dshift"], gated=ax.get("gated"))
    if fuse:
        dX1, dacc_x = rx
    else:
        dX1 = rx
        dacc_x = ops.gate_residual_bwd(dX1, sv.acc_ox, ms.gate1x, N, dms.gate1x, None, m.T)
    probs = [dict(A=dacc_x, B=w.Wo_x, b_kmajor=True, out_dtype=BF16)]
    defer(g, "Wo_x", dacc_x, sv.Oxa)
    if both:
        if fuse:
            dC1, dacc_c = rc
        else:
            dC1 = rc
            dacc_c = ops.gate_residual_bwd(dC1, sv.acc_oc, ms.gate1c, Mt, dms.gate1c, None, m.T)
        probs.append(dict(A=dacc_c, B=w.Wo_c, b_kmajor=True, out_dtype=BF16))
        defer(g, "Wo_c", dacc_c, sv.Oca)
    dO = _group(m, probs)
    dOx, dOc = dO[0], (dO[1] if both else None)   # the last block's text attention output is discarded (Attention.py:425)

    # ---- attention core + QK norm / RoPE
    # the QK-norm + RoPE backward runs in the epilogues of the attention backward kernels (ops.attn_bwd_qk) when the shapes allow it:
    # no dQ / dK / dV round trip, one pass less (MMDIT_ATTN_QK_FUSE=0: the two-pass form)
    if _QK_FUSE and m.fast and dev.type == "cuda" and ops.attn_bwd_qk_ok(sv.Q, N, sv.qkv_x) and g.wq_x.data_ptr() + 768 == g.wk_c.data_ptr():
        dqkv_x, dqkv_c = ops.attn_bwd_qk(sv.Q, sv.K, sv.V, sv.Ox, sv.Oc, dOx, dOc, sv.lse, N, 64 ** -0.5, sv.qkv_x, sv.qkv_c, w.wq_x, w.wk_x, w.wq_c, w.wk_c,
                                         rope[0], rope[1], torch.as_strided(g.wq_x, (256,), (1,)))
    else:
        dqkv_x, dqkv_c = _qk_bwd_two_pass(m, w, sv, g, dOx, dOc, dims, rope, dev)
    dln1 = _group(m, [dict(A=dqkv_x, B=w.Wqkv_x, b_kmajor=True, out_dtype=m.T), dict(A=dqkv_c, B=w.Wqkv_c, b_kmajor=True, out_dtype=m.T)])
    defer(g, "Wqkv_x", dqkv_x, sv.ln1x)
    defer(g, "Wqkv_c", dqkv_c, sv.ln1c)
    # the adaLN backward that produces dX / dC also runs the gated-residual backward of the block that consumes them
    nacc = None
    ax = dict(dout=dln1[0], x=sv.X, mean=sv.mu1x, rstd=sv.rs1x, scale=ms.scale1x, dres=dX1, rpb=N, dscale=dms.scale1x, dshift=dms.shift1x)
    ac = dict(dout=dln1[1], x=sv.C, mean=sv.mu1c, rstd=sv.rs1c, scale=ms.scale1c, dres=dC1, rpb=Mt, dscale=dms.scale1c, dshift=dms.shift1c)
    if nxt is not None and fuse:
        ax["gated"], ac["gated"] = nxt.req_x, nxt.req_c
        (dX, nax), (dC, nac) = _ln_bwd_pair(ax, ac)
        nacc = (nax, nac)
    else:
        dX, dC = _ln_bwd_pair(ax, ac)

    # ---- modulation vectors and y_proj
    dmod_a = m.act(dmod)
    if defer_cond:
        # defer_wmod: the M = batch weight gradient of the modulation matrix is left to the caller as well -- model_bwd issues those of
        # all blocks in ONE grouped launch at the end (twelve 21-us launches of a few tiles each are pure ramp-up and tail).  Not with a
        # data-parallel reducer: there every block's gradients must be complete when the block is handed over, so that their all-reduce
        # overlaps the rest of the backward instead of landing in the last, exposed bucket.
        if not defer_wmod:
            defer(g, "Wmod", dmod_a, sv.yp)
        dy_acc = dmod_a
    else:
        defer(g, "Wmod", dmod_a, sv.yp)
        dyp = _dgrad(m, dmod_a, w.Wmod, F32, **({"split_k": 16} if m.fast else {}))   # M = batch: 6 tiles, K = 12 d
        dpre = ops.silu_bwd(dyp, sv.pre, m.T, g.by)
        dy_acc = _dgrad(m, dpre, w.Wy, F32, residual=dy_acc)
        defer(g, "Wy", dpre, sv.y)

    wg_arena = _wgrad_flush(m, pending)   # all weight gradients of the block in one grouped launch
    # every parameter gradient of the block lives in one of these two flat buffers (None: not the case, e.g. overlap off)
    g.arenas = [wg_arena, st.small_arena] if wg_arena is not None else None
    if nxt is not None:
        return dX, dC, dy_acc, g, nacc
    return dX, dC, dy_acc, g


# ----------------------------------------------------------------------------------------------
# whole model
# ----------------------------------------------------------------------------------------------
def text_fwd(m, W, c):
    """Text tokens entering the blocks (diff_model.py:164-172, 323-326): cat[c_proj(s1 RMSNorm(c[:, :77])), c_proj2(s2 RMSNorm(c[:, 77:]))]
    as one (B*tokens, d) fp32 buffer.  Independent of the timestep: the sampler computes it once for all its steps.  Returns C and the two normalised operands (saved for backward)."""
    B, Mt = c.shape[0], c.shape[1]
    d = W.dim
    cn1, cn2 = ops.text_rmsnorm_fwd(c, W.wn1, W.wn2, W.s1, W.s2, W.split, m.T)
    c1, c2 = _group(m, [dict(A=cn1, B=W.Wc1, out_dtype=F32), dict(A=cn2, B=W.Wc2, out_dtype=F32)])
    C = torch.cat([c1.view(B, W.split, d), c2.view(B, Mt - W.split, d)], dim=1).view(B * Mt, d)
    return C, cn1, cn2


def model_fwd(m, W, x_t, t, c, c_pooled, rope, keep=True, C=None):
    """W: packed weights (see models/diff_model.py).  x_t (B,Cin,H,W); t (B,) fp32;
    c (B,tokens,2304); c_pooled (B,class_dim).  Returns v (B,Cin,H,W) fp32 and the saved state.
    C: the text tokens from text_fwd() when the caller already has them (inference only: nothing is saved for their backward)."""
    B, Cin, Hh, Ww = x_t.shape
    d, H = W.dim, W.heads
    N, Mt = (Hh // 2) * (Ww // 2), c.shape[1]
    dims = (B, N, Mt, H, d)
    sv = NS(dims=dims, img=(B, Cin, Hh, Ww), c=c, t=t)
    sv.pe = ops.time_embed_fwd(t, W.time_scale, W.denom, m.T)
    temb = _gemm(m, sv.pe, W.Wt, out_dtype=F32)
    sv.cp = m.act(c_pooled)
    sv.y = _gemm(m, sv.cp, W.Wcond, residual=temb, out_dtype=m.T)

    if C is None:
        C, sv.cn1, sv.cn2 = text_fwd(m, W, c)

    sv.patches = ops.patchify(x_t, m.T)
    sv.X0 = _gemm(m, sv.patches, W.Wpatch, out_dtype=m.T)
    X = _gemm(m, sv.X0, W.Wpe, bias=W.bpe, out_dtype=F32)

    sv.blocks = []
    conds, sv.pre_all = cond_fwd_all(m, W.blocks, sv.y)
    for wb, cond in zip(W.blocks, conds):
        X, C, bs = block_fwd(m, wb, X, C, sv.y, dims, rope, cond, keep, lazy=True)   # (Pending streams between the blocks)
        sv.blocks.append(bs)

    sv.modo = _gemm(m, sv.y, W.Wmod_out, out_dtype=F32)  # [shift | scale]
    sv.Xf, sv.lnf, sv.muf, sv.rsf = _norm(m, X, sv.modo[:, d:], sv.modo[:, :d], N)
    Z = _gemm(m, sv.lnf, W.Wout, bias=W.bout, out_dtype=F32)
    v = ops.unpatchify(Z, B, Cin, Hh, Ww, F32)
    return v, sv


def _grad_tensors(g):
    out = []
    for v in vars(g).values():
        if isinstance(v, NS):
            out += _grad_tensors(v)
        elif torch.is_tensor(v):
            out.append(v)
    return out


def _set_grad_tensors(g, new, pos=0):
    """Replace the tensors of g (same traversal order as _grad_tensors) by `new`; returns the next position."""
    for k, v in list(vars(g).items()):
        if isinstance(v, NS):
            pos = _set_grad_tensors(v, new, pos)
        elif torch.is_tensor(v):
            setattr(g, k, new[pos])
            pos += 1
    return pos


def model_bwd(m, W, sv, dv, rope, on_grads=None):
    """Returns grads: NS with top-level packed-weight grads and .blocks = [block grads].
    on_grads(list_of_tensors) is called as soon as a group of parameter gradients is final (block by
    block, last block first) so a data-parallel reducer can overlap its collective with the rest."""
    B, N, Mt, H, d = sv.dims
    _, Cin, Hh, Ww = sv.img
    dev = dv.device
    if dev.type == "cuda":
        ops.zero_pool_begin(dev)
    g = NS()
    pending = []

    def defer(name, dY, Xa):
        pending.append((lambda o, name=name: setattr(g, name, o), _wg(dY, Xa)))

    dZ = ops.patchify(dv.contiguous(), m.T)
    g.bout = ops.zeros(W.Wout.shape[0], dev)
    ops.colsum(dZ, g.bout)
    dlnf = _dgrad(m, dZ, W.Wout, m.T)
    defer("Wout", dZ, sv.lnf)
    dmodo = ops.zeros(sv.modo.shape, dev) if sv.modo.dtype == F32 else torch.zeros_like(sv.modo)
    nblk = len(W.blocks)
    # fused gate chain: every adaLN backward that produces a residual-stream gradient also runs the gated-residual backward of
    # its consumer (block_bwd_begin / block_bwd): the output head's for the last block's MLP update, block i's for block i-1's
    sts = [None] * nblk
    sts[nblk - 1] = block_bwd_begin(m, W.blocks[-1], sv.blocks[-1], sv.dims, dev, defer_cond=True)
    dacc = None
    if _FUSE_GATE:
        dX, dax = ops.ln_modulate_bwd(dlnf, sv.Xf, sv.muf, sv.rsf, sv.modo[:, d:], None, N, dmodo[:, d:], dmodo[:, :d], gated=sts[-1].req_x)
        dacc = (dax, None)
    else:
        dX = ops.ln_modulate_bwd(dlnf, sv.Xf, sv.muf, sv.rsf, sv.modo[:, d:], None, N, dmodo[:, d:], dmodo[:, :d])
    dmodo_a = m.act(dmodo)
    dy_acc = _dgrad(m, dmodo_a, W.Wmod_out, F32, **({"split_k": 4} if m.fast else {}))
    defer("Wmod_out", dmodo_a, sv.y)

    dC = None
    g.blocks = [None] * nblk
    dmods = [None] * nblk
    yps = [None] * nblk
    batch_wmod = on_grads is None and _BATCH_WMOD
    for i in range(nblk - 1, -1, -1):
        if i > 0:
            sts[i - 1] = block_bwd_begin(m, W.blocks[i - 1], sv.blocks[i - 1], sv.dims, dev, defer_cond=True)
            dX, dC, dmods[i], g.blocks[i], dacc = block_bwd(m, W.blocks[i], sv.blocks[i], dX, dC, None, sv.dims, rope, defer_cond=True,
                                                            st=sts[i], dacc=dacc, nxt=sts[i - 1], defer_wmod=batch_wmod)
        else:
            dX, dC, dmods[i], g.blocks[i] = block_bwd(m, W.blocks[i], sv.blocks[i], dX, dC, None, sv.dims, rope, defer_cond=True, st=sts[i], dacc=dacc,
                                                      defer_wmod=batch_wmod)
        sts[i] = None
        yps[i] = sv.blocks[i].yp
        sv.blocks[i] = None  # free saved activations as we go
        if on_grads is not None:
            # the reducer may hand back views of its flat bucket (zero copy-back): use them as this block's gradients
            repl = on_grads(_grad_tensors(g.blocks[i]), _wg_streams.get(dev), g.blocks[i].arenas)
            if repl is not None:
                _set_grad_tensors(g.blocks[i], repl)

    # conditioning path of all blocks at once (see cond_fwd_all): d(mod) -> d(y') -> SiLU backward -> d(y)
    dyp_all = ops.zeros((nblk * B, d), dev)
    by_all = ops.zeros((nblk, d), dev)
    for i0 in range(0, nblk, 12):
        _group(m, [dict(A=dmods[i], B=W.blocks[i].Wmod, b_kmajor=True, out=dyp_all[i * B:(i + 1) * B], **({"split_k": 4} if m.fast else {}))
                   for i in range(i0, min(nblk, i0 + 12))])
    dpre_all = ops.silu_bwd(dyp_all, sv.pre_all, m.T, by_all, rows_per_bias=B)
    dyc = torch.empty((nblk + 1, B, d), dtype=F32, device=dev)
    dyc[nblk].copy_(dy_acc)
    for i0 in range(0, nblk, 12):
        _group(m, [dict(A=dpre_all[i * B:(i + 1) * B], B=W.blocks[i].Wy, b_kmajor=True, out=dyc[i]) for i in range(i0, min(nblk, i0 + 12))])
    dy_acc = dyc.sum(0)
    late = []   # (block index, name): block gradients produced here, after that block's gradients were handed to on_grads
    for i in range(nblk):
        g.blocks[i].by = by_all[i]
        pending.append((lambda o, i=i: setattr(g.blocks[i], "Wy", o), _wg(dpre_all[i * B:(i + 1) * B], sv.y)))
        late += [(i, "by"), (i, "Wy")]
        if batch_wmod:
            pending.append((lambda o, i=i: setattr(g.blocks[i], "Wmod", o), _wg(dmods[i], yps[i])))

    # patch embedding
    g.bpe = ops.zeros(d, dev)
    ops.colsum(dX, g.bpe)
    dX_a = m.act(dX)
    dX0 = _dgrad(m, dX_a, W.Wpe, m.T)
    defer("Wpe", dX_a, sv.X0)
    defer("Wpatch", dX0, sv.patches)

    # text embedding
    dCv = dC.view(B, Mt, d)
    dc1 = m.act(dCv[:, :W.split].reshape(B * W.split, d))
    dc2 = m.act(dCv[:, W.split:].reshape(B * (Mt - W.split), d))
    dcn1, dcn2 = _group(m, [dict(A=dc1, B=W.Wc1, b_kmajor=True, out_dtype=m.T), dict(A=dc2, B=W.Wc2, b_kmajor=True, out_dtype=m.T)])
    defer("Wc1", dc1, sv.cn1)
    defer("Wc2", dc2, sv.cn2)
    g.wn1, g.wn2, g.s1, g.s2 = ops.text_rmsnorm_bwd(dcn1, dcn2, sv.c, W.wn1, W.wn2, W.s1, W.s2, W.split)

    # conditioning vector
    dy_a = m.act(dy_acc)
    defer("Wcond", dy_a, sv.cp)
    dpe = _dgrad(m, dy_a, W.Wt, m.T)
    defer("Wt", dy_a, sv.pe)
    g.time_scale = ops.time_embed_bwd(dpe, sv.t, W.time_scale, W.denom)
    _wgrad_flush(m, pending)
    if on_grads is not None:
        keys = [k for k, v in vars(g).items() if torch.is_tensor(v)]
        repl = on_grads([getattr(g, k) for k in keys] + [getattr(g.blocks[i], n) for i, n in late], _wg_streams.get(dev))
        if repl is not None:
            for k, t in zip(keys, repl):
                setattr(g, k, t)
            for (i, n), t in zip(late, repl[len(keys):]):
                setattr(g.blocks[i], n, t)
    wgrad_join(dev)
    return g
