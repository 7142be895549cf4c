"""Thin tensor-level wrappers over the C ABI (include/mmdit_hip.h).

Every function launches on torch's current HIP stream and never synchronises.
PyTorch is used only to own device memory; all arithmetic happens in libmmdit_hip.so.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_SILU, ACT_SWIGLU, ACT_SWIGLU_BWD, BF16, F32, FP8, PREC_BF16, PREC_SPLIT, GemmArgs, check

_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float8_e4m3fn: FP8}

# bench.py sets this to a list to time every GEMM launch with HIP events on the launch stream
# (entries: (kernel variant, algorithmic FLOPs, start event, end event)); None = no instrumentation.
PROFILE = None
_ZERO_ALL = _lib.experiment("MMDIT_ZERO_ALL", "0") == "1"     # A/B: zero-fill every K-decomposed GEMM output (from the pass's zero pool)


class _ZeroPool:
    """Zero-initialised fp32 scratch for one backward pass (atomic accumulation targets: split-K / bias / norm-weight
    gradients).  The demand of a pass is learned from the previous one, so that from the second step on ONE memset serves
    every request of the pass (~200 tiny fill launches per step otherwise).  Slices stay valid for as long as they are
    referenced: every pass gets a fresh buffer."""

    def __init__(self):
        self.buf, self.off, self.cap, self.need = None, 0, 0, 0

    def begin(self, device):
        self.cap = max(self.cap, self.need)
        self.need, self.off = 0, 0
        self.buf = torch.zeros(self.cap, dtype=torch.float32, device=device) if self.cap else None

    def take(self, numel, device):
        n = (numel + 3) // 4 * 4            # keep every slice 16-byte aligned
        self.need += n
        if self.buf is not None and self.buf.device == device and self.off + n <= self.cap:
            v = self.buf[self.off:self.off + numel]
            self.off += n
            return v
        return torch.zeros(numel, dtype=torch.float32, device=device)


_pools = {}
NO_POOL = False     # debug_guard.install() sets it: every zeros() request becomes an allocation of its own (with its own guards)


def zero_pool_begin(device):
    """Call at the start of a backward pass (engine.model_bwd)."""
    _pools.setdefault(device, _ZeroPool()).begin(device)


def zeros(shape, device):
    """fp32 zeros of `shape`: a slice of the pass's pool on a GPU, plain torch.zeros elsewhere."""
    shape = tuple(shape) if not isinstance(shape, int) else (shape,)
    pool = None if NO_POOL else _pools.get(device)
    if pool is None:
        return torch.zeros(shape, dtype=torch.float32, device=device)
    return pool.take(int(torch.Size(shape).numel()), device).view(shape)


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise RuntimeError(f"unsupported dtype {t.dtype} (the HIP path takes float32 / bfloat16)")


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("mmdit HIP op called with a CPU tensor: the MMDiT hot path has no CPU fallback")
    return t.data_ptr()


def _s():
    return torch.cuda.current_stream().cuda_stream


def _c(t):
    if t is not None and not t.is_contiguous():
        raise RuntimeError("mmdit HIP ops need contiguous tensors")
    return t


def torch_dtype(code: int):
    return torch.float32 if code == F32 else torch.bfloat16


# ---------------------------------------------------------------------------------------------
def _fill_gemm(a, A, B, *, a_kmajor=False, b_kmajor=False, out=None, out_dtype=None, bias=None, act=ACT_NONE,
               gate=None, rows_per_batch=0, residual=None, aux=None, accumulate=False, precision=PREC_BF16, split_k=1, stream_k=False, conv=None, scale_a=None, scale_b=None, scale_mode=0, out_scales=None, dbias=None):
    """conv = (mode, H, W, C): A is a zero-bordered NHWC bf16 tensor (batch, H+2, W+2, C) -- implicit-GEMM 3x3 convolution."""
    if conv is not None:
        mode, cH, cW, cC = conv
        if A.dim() != 4 or tuple(A.shape[1:]) != (cH + 2, cW + 2, cC) or A.dtype != torch.bfloat16:
            raise RuntimeError("conv operand must be a zero-bordered bf16 (batch, H+2, W+2, C) tensor")
        M, K_ = A.shape[0] * (cH * cW if mode == 1 else (cH // 2) * (cW // 2)), 9 * cC
    elif a_kmajor:
        K_, M = A.shape
    else:
        M, K_ = A.shape
    if b_kmajor:
        Kb, N = B.shape
    else:
        N, Kb = B.shape
    if K_ != Kb:
        raise RuntimeError(f"gemm: inner dimensions differ ({K_} vs {Kb})")
    if act == ACT_SWIGLU:   # packed SwiGLU up-projection: out (M, h) = silu(g) * u, aux (M, 2h) = the pre-activations [g | u]
        if (aux is not None and (aux.dtype != torch.bfloat16 or tuple(aux.shape) != (M, N))) or (out is not None and tuple(out.shape) != (M, N // 2)):
            raise RuntimeError("gemm(act=ACT_SWIGLU): aux (optional) is the bf16 (M, 2h) pre-activation buffer, out (M, h)")
        if (out_scales is not None) != (out is not None and out.dtype == torch.float8_e4m3fn):
            raise RuntimeError("gemm(act=ACT_SWIGLU): an MX output needs out (float8_e4m3fn) AND out_scales")
        if out is None:
            out = torch.empty((M, N // 2), dtype=torch.bfloat16, device=A.device)
    if act == ACT_SWIGLU_BWD:   # dgrad of the SwiGLU down-projection + activation backward: aux (M, 2h) = saved [g | u] (input), out (M, 2h) = d[g | u]
        if aux is None or aux.dtype != torch.bfloat16 or tuple(aux.shape) != (M, 2 * N) or (out is not None and tuple(out.shape) != (M, 2 * N)):
            raise RuntimeError("gemm(act=ACT_SWIGLU_BWD): aux is the saved bf16 (M, 2h) pre-activation buffer, out (M, 2h)")
        if out is None:
            out = torch.empty((M, 2 * N), dtype=torch.bfloat16, device=A.device)
    if out is None:
        if split_k > 1 or stream_k:
            # K-decomposed launch: partial tiles may be added atomically.  gemm_grouped asks the planner which outputs really
            # receive atomics and zero-fills only those (whole-K tiles are stored): no memset of every weight gradient
            if _ZERO_ALL:
                out = zeros((M, N), A.device)
            else:
                out = torch.empty((M, N), dtype=torch.float32, device=A.device)
                out._mmdit_zero_check = True
        else:
            out = torch.empty((M, N), dtype=out_dtype or torch.float32, device=A.device)
    a.A, a.a_dtype, a.a_kmajor, a.lda = _p(A), _dt(A), int(a_kmajor), (K_ if conv is not None else A.stride(0))
    if conv is not None:
        a.conv_mode, a.conv_H, a.conv_W, a.conv_C = conv
    a.scale_a, a.scale_b, a.scale_mode, a.c_scales = _p(scale_a), _p(scale_b), int(scale_mode), _p(out_scales)
    a.B, a.b_dtype, a.b_kmajor, a.ldb = _p(B), _dt(B), int(b_kmajor), B.stride(0)
    a.C, a.c_dtype, a.ldc = _p(out), _dt(out), out.stride(0)
    a.M, a.N, a.K = M, N, K_
    a.bias = _p(bias)
    a.act = act
    if gate is not None:
        a.gate, a.ld_gate, a.rows_per_batch = _p(gate), gate.stride(0), rows_per_batch
    if residual is not None:
        a.residual, a.ld_res = _p(residual), residual.stride(0)
    if aux is not None:
        a.aux, a.aux_dtype, a.ld_aux = _p(aux), _dt(aux), aux.stride(0)
    a.accumulate = int(accumulate)
    a.precision = precision
    a.split_k = split_k
    a.stream_k = int(stream_k)
    a.dbias = _p(dbias)
    return out


_CFG = {0: "2,2,2,2", 1: "4,2,2,2", 2: "2,4,4,2", 3: "2,4,5,2"}


def _variant(arr, n, outs):
    """Name of the kernel symbol (template instantiation) mmdit_gemm_grouped launches for these problems."""
    a = arr[0]
    plan = _lib.lib().mmdit_gemm_plan(arr, n)
    tc = "f" if _dt(outs[0]) == F32 else "t"
    aux = [arr[i].aux_dtype for i in range(n) if arr[i].aux]
    ta = ("f" if aux[0] == F32 else "t") if aux else tc
    km = f"{int(bool(a.a_kmajor))},{int(bool(a.b_kmajor))}"
    if plan == 64:
        ab = "t,t" if a.precision == PREC_BF16 else "f,f"
        return f"gemm_kernel<{ab},{km},{int(a.precision == PREC_SPLIT)},{tc},{ta}>"
    if plan & 256:      # the 8-phase kernel (csrc/gemm8p.hip): <a_kmajor, b_kmajor, epilogue>
        epi = "f32" if a.a_kmajor else "swiglu" if a.act == ACT_SWIGLU else "swiglu_bwd" if a.act == ACT_SWIGLU_BWD else "bf16"
        return f"gemm8_kernel<{320 if plan & 15 == 3 else 256},{km},{epi}>" + ("+ktail" if plan & 32 else "")
    if plan & 128 and a.a_kmajor:
        return "gemm_kk_kernel<2,4,4,2>" + ("+ktail" if plan & 32 else "")
    if plan & 128:
        kern = "gemm_lean_kernel" if _lib.experiment("MMDIT_GEMM_WIDE", "1") == "0" else "gemm_wide_kernel"
        return f"{kern}<{_CFG[plan & 15]},{int(bool(a.b_kmajor))}>" + ("+swiglu" if a.act == ACT_SWIGLU else "")
    return f"gemm_dma_kernel<{_CFG[plan & 15]},{km},{tc},{ta}>" + ("+streamK" if plan & 16 else "") + ("+ktail" if plan & 32 else "") + ("+swiglu" if a.act == ACT_SWIGLU else "")


# Workspace of the GEMM launches (mmdit_gemm_set_workspace): 4 KiB of zeroed tickets (split tails), 4 KiB of zeroed scheduler words (the heads of the
# per-XCD tile queues the persistent 8-phase launches claim their tiles from -- csrc/gemm8p.hip "dynamic tile claiming"), then 512 slots of 256 KiB for
# partial tiles: one per DEVICE, allocated at the first GEMM launch on that GPU and kept for the life of the process (a captured hipGraph holds its
# address).  MMDIT_GEMM_WS=0: none -- static tile walk, fp32 atomics for the split tail (the round-2 path).
_GEMM_WS = {}
_GEMM_WS_ON = _lib.experiment("MMDIT_GEMM_WS", "1") != "0" and _lib.experiment("MMDIT_WGRAD_STREAM", "0") != "1"   # (one workspace: its launches must be stream-ordered -- not with the weight-gradient side stream)
GEMM_WS_BYTES = 8192 + 512 * 65536 * 4


def _ensure_gemm_workspace(device):
    if not _GEMM_WS_ON or device in _GEMM_WS:
        return
    if torch.cuda.is_current_stream_capturing():
        return          # (never allocate the workspace inside a capture: the eager warm-up steps do it)
    ws = torch.zeros(GEMM_WS_BYTES, dtype=torch.uint8, device=device)
    with torch.cuda.device(device):       # the library keys the registration by the CURRENT device
        check(_lib.lib().mmdit_gemm_set_workspace(ws.data_ptr(), ws.numel()), "mmdit_gemm_set_workspace")
    _GEMM_WS[device] = ws


# Data parallel (diff_model._MMDiTFn.backward): compute units the K-decomposed (weight-gradient) launches are planned for, None = the device's
WGRAD_CU_BUDGET = None


def _gemm_grouped(problems):
    """problems: list of dicts of gemm() keyword arguments (plus 'A', 'B'), all of one kernel variant.
    One launch; returns the list of outputs."""
    n = len(problems)
    if problems[0]["A"].is_cuda:
        _ensure_gemm_workspace(problems[0]["A"].device)
    arr = (GemmArgs * n)()
    outs = [_fill_gemm(arr[i], **problems[i]) for i in range(n)]
    if any(getattr(o, "_mmdit_zero_check", None) is not None for o in outs):
        mask = ctypes.c_uint(0)
        check(_lib.lib().mmdit_gemm_zero_mask(arr, n, ctypes.byref(mask)), "mmdit_gemm_zero_mask")
        spans = []          # [arena, first element, end element): flagged outputs that are neighbours in one arena share a fill launch
        for i, o in enumerate(outs):
            arena = getattr(o, "_mmdit_zero_check", None)
            if arena is None:
                continue
            del o._mmdit_zero_check
            if not (mask.value >> i) & 1:
                continue
            if arena is True:
                o.zero_()
                continue
            b = (o.data_ptr() - arena.data_ptr()) // 4
            e = (b + o.numel() + 3) // 4 * 4
            if spans and spans[-1][0] is arena and spans[-1][2] == b:
                spans[-1][2] = e
            else:
                spans.append([arena, b, e])
        for arena, b, e in spans:
            arena[b:min(e, arena.numel())].zero_()
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_lib.lib().mmdit_gemm_grouped(arr, n, _s()), "mmdit_gemm_grouped")
        e1.record()
        PROFILE.append((_variant(arr, n, outs), sum(2.0 * arr[i].M * arr[i].N * arr[i].K for i in range(n)), e0, e1))
        return outs
    check(_lib.lib().mmdit_gemm_grouped(arr, n, _s()), "mmdit_gemm_grouped")
    return outs


def gemm_grouped(problems):
    """problems: list of dicts of gemm() keyword arguments (plus 'A', 'B'), all of one kernel variant.  One launch; returns the list of outputs.
    K-decomposed launches (stream_k: the weight gradients) are planned for WGRAD_CU_BUDGET compute units when that is set (zero mask and launch alike)."""
    planned = WGRAD_CU_BUDGET if (WGRAD_CU_BUDGET and problems[0].get("stream_k") and problems[0]["A"].is_cuda) else None
    if not planned:
        return _gemm_grouped(problems)
    L = _lib.lib()
    whole = L.mmdit_get_cu_budget()
    check(L.mmdit_set_cu_budget(min(planned, whole)), "mmdit_set_cu_budget")
    try:
        return _gemm_grouped(problems)
    finally:
        check(L.mmdit_set_cu_budget(whole), "mmdit_set_cu_budget")


def gemm_swiglu_bwd(problems):
    """Data gradient of the SwiGLU down-projection with the activation backward in its epilogue (MMDIT_ACT_SWIGLU_BWD): one launch for
    the image and the text MLP.  problems: dicts with A = d(out) (M, d) bf16, B = w3 (d, h) [b_kmajor is set here], aux = the saved (M, 2h)
    [g | u], dbias = fp32 (2h,) accumulating bias gradient or None.  Returns the list of d[g | u] (M, 2h) bf16 -- the bits of gemm() followed
    by mlp_act_bwd() -- or None when the planner would not run these problems on the 8-phase 256 x 256 kernel (caller: the two passes)."""
    n = len(problems)
    # what the fused launch needs beyond the plain GEMM (csrc/gemm.hip: 16-byte aligned [g | u] and d[g | u] rows, hidden width a multiple of 8):
    # an arena slice that does not offer it takes the two-pass path like a shape the planner refuses, instead of raising
    for q in problems:
        aux, out = q["aux"], q.get("out")
        if aux.data_ptr() % 16 or aux.stride(0) % 8 or aux.shape[-1] % 16 or (out is not None and (out.data_ptr() % 16 or out.stride(0) % 8)):
            return None
    _ensure_gemm_workspace(problems[0]["A"].device)
    arr = (GemmArgs * n)()
    outs = [_fill_gemm(arr[i], **dict(problems[i], b_kmajor=True, act=ACT_SWIGLU_BWD)) for i in range(n)]
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = _lib.lib().mmdit_gemm_grouped(arr, n, _s())
    if rc == _lib.ERR_SHAPE:
        return None
    check(rc, "mmdit_gemm_grouped(swiglu_bwd)")
    if PROFILE is not None:
        e1.record()
        PROFILE.append((_variant(arr, n, outs), sum(2.0 * arr[i].M * arr[i].N * arr[i].K for i in range(n)), e0, e1))
    return outs


def gemm_qkv_norm_rope(problems, streams, heads, s_total, Q, K, V, raw=True):
    """The QKV projection with QK-RMSNorm + RoPE + joint-layout store in the GEMM epilogue (mmdit_gemm_qkv_norm_rope): one launch.
    problems: 1 or 2 dicts of gemm() arguments (A = normed stream rows, B = packed [q | k | v] weight, out_dtype bf16; image stream first);
    streams[i] = (wq, wk, rope_cos | None, rope_sin | None, tokens per sample, joint position of token 0).  Returns the raw projections
    (kept for backward: q and k columns; the v columns are not written, V holds them) or None when the planner would not run these problems on the lean wide-slot kernel (caller: GEMM + row kernel).
    raw=False (inference: nobody reads the raw projection): C = NULL, the q / k columns are not written either -- 8-phase kernel only (MX operands, or bf16 with tile claiming on),
    otherwise None as above; returns [None] * n."""
    n = len(problems)
    _ensure_gemm_workspace(problems[0]["A"].device)
    arr = (GemmArgs * n)()
    if raw:
        outs = [_fill_gemm(arr[i], **problems[i]) for i in range(n)]
    else:
        for i in range(n):      # (a one-row stand-in gives the planner the dtype and a legal row pitch; the pointer itself is NULL)
            N_ = problems[i]["B"].shape[0]
            _fill_gemm(arr[i], **dict(problems[i], out=torch.empty((1, N_), dtype=problems[i].get("out_dtype") or torch.bfloat16, device=problems[i]["A"].device)))
            arr[i].C = None
        outs = [None] * n
    qk = (_lib.QkEpilogue * n)()
    for i, (wq, wk, rc, rs, tokens, tok0) in enumerate(streams):
        qk[i].wq, qk[i].wk, qk[i].rope_cos, qk[i].rope_sin, qk[i].tokens, qk[i].tok0 = _p(wq), _p(wk), _p(rc), _p(rs), int(tokens), int(tok0)
    L = _lib.lib()
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = L.mmdit_gemm_qkv_norm_rope(arr, qk, n, int(heads), int(s_total), _p(Q), _p(K), _p(V), _s())
    if rc == _lib.ERR_SHAPE:
        return None
    check(rc, "mmdit_gemm_qkv_norm_rope")
    if PROFILE is not None:
        e1.record()
        # (mmdit_gemm_plan knows nothing of the epilogue request: the fused launch runs on the wide-slot kernel -- bf16 -- or, MX operands / tile claiming on, on the 8-phase kernel)
        on8 = arr[0].a_dtype == _lib.FP8 or bool(L.mmdit_gemm_get_claiming())
        PROFILE.append((("gemm8_kernel<256,0,0,bf16>+qk" if on8 else "gemm_wide_kernel<2,4,5,2,0>+qk"), sum(2.0 * arr[i].M * arr[i].N * arr[i].K for i in range(n)), e0, e1))
    return outs


def gemm(A, B, **kw):
    """C[M,N] = epilogue(A[M,K] B[N,K]^T).  A: (M,K) or k-major (K,M); B: (N,K) or k-major (K,N).
    gate may be a strided 2-D view (rows = batch) with unit inner stride."""
    return gemm_grouped([dict(A=A, B=B, **kw)])[0]


def quant_fp8(x):
    """Per-tensor e4m3 quantisation: returns (q: torch.float8_e4m3fn like x, scale: fp32 (1,) dequantisation scale on the device)."""
    n = x.numel()
    buf = torch.zeros(2, dtype=torch.float32, device=x.device)       # [amax, scale]
    q = torch.empty(x.shape, dtype=torch.float8_e4m3fn, device=x.device)
    L = _lib.lib()
    check(L.mmdit_fp8_amax(_p(_c(x)), _dt(x), n, _p(buf), _s()), "mmdit_fp8_amax")
    check(L.mmdit_fp8_quantize(_p(x), _dt(x), n, _p(buf), _p(q), _p(buf[1:]), _s()), "mmdit_fp8_quantize")
    return q, buf[1:]


class MxAct:
    """An activation that left its producer in MX e4m3: q (rows, K) float8_e4m3fn codes + sc, the E8M0 block scales in the GEMM's
    layout (see quant_mxfp8).  engine._to_fp8 hands it to the GEMM as is."""
    __slots__ = ("q", "sc")

    def __init__(self, q, sc):
        self.q, self.sc = q, sc

    @property
    def shape(self):
        return self.q.shape

    def dequant(self):
        """fp32 values (tests)."""
        rows, K = self.q.shape
        e = mx_scales_to_rows(self.sc, rows, K).to(torch.int32) - 127
        return (self.q.float().view(rows, K // 32, 32) * torch.ldexp(torch.ones((), device=self.q.device), e).unsqueeze(-1)).view(rows, K)


def mx_scale_bytes(rows, K):
    """Size of the E8M0 scale buffer of a (rows, K) MX operand: rows padded to 128, 2 bytes per row and 64-wide K half, 512 spare."""
    return (K // 64) * ((rows + 127) // 128 * 128) * 2 + 512


def mx_scales_to_rows(sc, rows, K):
    """The scale bytes as a (rows, K/32) uint8 tensor (inverse of the GEMM layout: per 64-wide K half the rows in groups of 128, inside
    a group byte (row & 31) * 8 + h * 4 + ((row >> 5) & 3) for half-block h)."""
    rp = (rows + 127) // 128 * 128
    v = sc[:(K // 64) * rp * 2].view(K // 64, rp // 128, 32, 2, 4)          # [k64][group][r31][h][rb]
    return v.permute(1, 4, 2, 0, 3).reshape(rp, K // 32)[:rows]              # row = group*128 + rb*32 + r31, blk = k64*2 + h


def _mx_buffers(rows, K, device):
    return (torch.empty((rows, K), dtype=torch.float8_e4m3fn, device=device), torch.empty(mx_scale_bytes(rows, K), dtype=torch.uint8, device=device))


def ln_modulate_fwd_mx(x, scale, shift, rows_per_batch, acc=None, gate=None):
    """adaLN with the output in MX e4m3 (mmdit_ln_modulate_fwd_mx); acc / gate: the pending gated residual update, as in
    ln_modulate_fwd_res.  Returns (x1 fp32 (x itself without acc), MxAct, mean, rstd)."""
    rows, d = x.shape
    q, sc = _mx_buffers(rows, d, x.device)
    mean = torch.empty((rows,), dtype=torch.float32, device=x.device)
    rstd = torch.empty((rows,), dtype=torch.float32, device=x.device)
    x1 = torch.empty((rows, d), dtype=torch.float32, device=x.device) if acc is not None else x
    check(_lib.lib().mmdit_ln_modulate_fwd_mx(_p(_c(x)), _p(acc), _dt(acc) if acc is not None else BF16, _p(gate), gate.stride(0) if gate is not None else 0,
                                              _p(x1) if acc is not None else None, _p(scale), _p(shift), scale.stride(0), rows, d, rows_per_batch,
                                              _p(q), _p(sc), _p(mean), _p(rstd), _s()), "mmdit_ln_modulate_fwd_mx")
    return x1, MxAct(q, sc), mean, rstd


def swiglu_fwd_mx(gu, hidden):
    """SwiGLU activation of the bf16 pre-activations (rows, 2 hidden) with the output in MX e4m3 (mmdit_swiglu_fwd_mx)."""
    rows = gu.shape[0]
    q, sc = _mx_buffers(rows, hidden, gu.device)
    check(_lib.lib().mmdit_swiglu_fwd_mx(_p(_c(gu)), _dt(gu), rows, hidden, _p(q), _p(sc), _s()), "mmdit_swiglu_fwd_mx")
    return MxAct(q, sc)


def attn_fwd_mx(Q, K, V, n_img, scale):
    """Flash attention forward with the head-merged outputs in MX e4m3 (mmdit_attn_fwd_mx): returns (MxAct (B*N, H*64), MxAct (B*M, H*64) or None)."""
    B, H, S, hd = Q.shape
    n_txt = S - n_img
    qx, sx = _mx_buffers(B * n_img, H * hd, Q.device)
    qc, scc = _mx_buffers(B * n_txt, H * hd, Q.device) if n_txt else (None, None)
    check(_lib.lib().mmdit_attn_fwd_mx(_p(Q), _p(K), _p(V), B, H, S, n_img, float(scale), _p(qx), _p(qc), _p(sx), _p(scc), _s()), "mmdit_attn_fwd_mx")
    return MxAct(qx, sx), (MxAct(qc, scc) if n_txt else None)


def quant_mxfp8(x):
    """MX (OCP microscaling) e4m3 quantisation of a row-major 2-D operand: returns (q: float8_e4m3fn like x, scales: uint8 E8M0 block
    scales in the GEMM's layout, see mx_scales_to_rows; 512 spare bytes behind them) -- mmdit_mxfp8_quantize; pass scale_mode=1 to gemm()."""
    rows, K = x.shape
    q = torch.empty((rows, K), dtype=torch.float8_e4m3fn, device=x.device)
    sc = torch.empty(mx_scale_bytes(rows, K), dtype=torch.uint8, device=x.device)
    check(_lib.lib().mmdit_mxfp8_quantize(_p(x), _dt(x), rows, K, x.stride(0), _p(q), _p(sc), _s()), "mmdit_mxfp8_quantize")
    return q, sc


class Fp8Site:
    """Delayed-scaling state of one activation call site (one GEMM A operand): the first call measures amax with the two-pass
    quantiser, later calls quantise in ONE pass with margin x the previous call's amax while collecting their own."""

    def __init__(self, margin=1.5):
        self.state, self.phase, self.margin = None, 0, margin

    def quantise(self, x):
        L = _lib.lib()
        q = torch.empty(x.shape, dtype=torch.float8_e4m3fn, device=x.device)
        if self.state is None:
            self.state = torch.zeros(4, dtype=torch.float32, device=x.device)
            check(L.mmdit_fp8_amax(_p(_c(x)), _dt(x), x.numel(), _p(self.state), _s()), "mmdit_fp8_amax")      # -> state[0] = amax
        check(L.mmdit_fp8_quantize_delayed(_p(_c(x)), _dt(x), x.numel(), _p(self.state), self.phase, float(self.margin if self.phase else 1.0),
                                           _p(q), _s()), "mmdit_fp8_quantize_delayed")
        self.phase += 1
        return q, self.state[3:]


def cast(src, dtype, out=None):
    if out is None:
        out = torch.empty(src.shape, dtype=dtype, device=src.device)
    check(_lib.lib().mmdit_cast(_p(_c(src)), _dt(src), _p(_c(out)), _dt(out), src.numel(), _s()), "mmdit_cast")
    return out


_CAST_TABLES = {}


def cast_multi(pairs):
    """fp32 -> bf16 for a list of (src, dst) CUDA tensor pairs in ONE launch (mmdit_cast_multi).  The device pointer table and
    chunk map are cached per set of pointers (weights and their bf16 copies are persistent buffers)."""
    import numpy as np
    dev = pairs[0][0].device
    key = tuple((s.data_ptr(), d.data_ptr(), s.numel()) for s, d in pairs)
    t = _CAST_TABLES.get(key)
    if t is None:
        for s, d in pairs:
            if s.dtype != torch.float32 or d.dtype != torch.bfloat16 or not s.is_contiguous() or not d.is_contiguous() or s.numel() != d.numel() or d.device != dev:
                raise RuntimeError("cast_multi: contiguous fp32 sources and bf16 destinations of equal size on one device")
        rec = np.array(key, dtype=np.int64).reshape(-1)
        ct = np.concatenate([np.full((n + _lib.ADAMW_CHUNK - 1) // _lib.ADAMW_CHUNK, i, dtype=np.int32) for i, (_, _, n) in enumerate(key)])
        co = np.concatenate([np.arange(0, n, _lib.ADAMW_CHUNK, dtype=np.int64) for _, _, n in key])
        host = [torch.from_numpy(a).pin_memory() for a in (rec, ct, co)]
        if len(_CAST_TABLES) >= 8:
            _CAST_TABLES.pop(next(iter(_CAST_TABLES)))
        t = _CAST_TABLES[key] = dict(host=host, dev=[h.to(dev, non_blocking=True) for h in host], n=len(ct))
    check(_lib.lib().mmdit_cast_multi(_p(t["dev"][0]), _p(t["dev"][1]), _p(t["dev"][2]), t["n"], _s()), "mmdit_cast_multi")


def ln_modulate_fwd(x, scale, shift, rows_per_batch, out_dtype):
    rows, d = x.shape
    out = torch.empty((rows, d), dtype=out_dtype, device=x.device)
    mean = torch.empty((rows,), dtype=torch.float32, device=x.device)
    rstd = torch.empty((rows,), dtype=torch.float32, device=x.device)
    check(_lib.lib().mmdit_ln_modulate_fwd(_p(_c(x)), _p(scale), _p(shift), scale.stride(0), rows, d, rows_per_batch,
                                           _p(out), _dt(out), _p(mean), _p(rstd), _s()), "mmdit_ln_modulate_fwd")
    return out, mean, rstd


def ln_modulate_fwd_res(x, acc, gate, scale, shift, rows_per_batch, out_dtype):
    """x1 = x + gate[b] * acc, then adaLN of x1: returns (x1 fp32, out, mean, rstd) -- mmdit_ln_modulate_fwd_res."""
    rows, d = x.shape
    x1 = torch.empty((rows, d), dtype=torch.float32, device=x.device)
    out = torch.empty((rows, d), dtype=out_dtype, device=x.device)
    mean = torch.empty((rows,), dtype=torch.float32, device=x.device)
    rstd = torch.empty((rows,), dtype=torch.float32, device=x.device)
    check(_lib.lib().mmdit_ln_modulate_fwd_res(_p(_c(x)), _p(_c(acc)), _dt(acc), _p(gate), gate.stride(0), _p(x1), _p(scale), _p(shift), scale.stride(0),
                                               rows, d, rows_per_batch, _p(out), _dt(out), _p(mean), _p(rstd), _s()), "mmdit_ln_modulate_fwd_res")
    return x1, out, mean, rstd


def ln_modulate_fwd_pair(pa, pb, out_dtype):
    """Two adaLN forward problems of the same width in ONE launch (the image and the text stream of a block; mmdit_ln_modulate_fwd_pair).
    pa / pb: dicts with x, scale, shift, rpb and -- both or neither -- acc, gate (the pending gated residual update, as in
    ln_modulate_fwd_res).  Returns [(x1, out, mean, rstd), (x1, out, mean, rstd)] (x1 is x itself without acc)."""
    res = pa.get("acc") is not None
    assert (pb.get("acc") is not None) == res
    d = pa["x"].shape[1]
    probs, outs = (_lib.LnFwdProblem * 2)(), []
    for q, p in zip(probs, (pa, pb)):
        x = _c(p["x"])
        rows = x.shape[0]
        dev = x.device
        x1 = torch.empty((rows, d), dtype=torch.float32, device=dev) if res else x
        out = torch.empty((rows, d), dtype=out_dtype, device=dev)
        mean = torch.empty((rows,), dtype=torch.float32, device=dev)
        rstd = torch.empty((rows,), dtype=torch.float32, device=dev)
        q.x, q.scale, q.shift, q.ld_mod, q.rows, q.rows_per_batch = _p(x), _p(p["scale"]), _p(p["shift"]), p["scale"].stride(0), rows, p["rpb"]
        q.out, q.mean, q.rstd = _p(out), _p(mean), _p(rstd)
        if res:
            acc = _c(p["acc"])
            q.acc, q.gate, q.ld_gate, q.x_out = _p(acc), _p(p["gate"]), p["gate"].stride(0), _p(x1)
            p["_keep"] = (x, acc)
        else:
            p["_keep"] = (x,)
        outs.append((x1, out, mean, rstd))
    acc_dt = _dt(pa["acc"]) if res else _DT[out_dtype]
    check(_lib.lib().mmdit_ln_modulate_fwd_pair(ctypes.byref(probs[0]), ctypes.byref(probs[1]), d, acc_dt, _DT[out_dtype], _s()), "mmdit_ln_modulate_fwd_pair")
    return outs


def ln_modulate_bwd_pair(pa, pb):
    """Two adaLN backward problems in one launch (mmdit_ln_modulate_bwd_pair).  pa / pb: dicts with dout, x, mean, rstd, scale, dres (or
    None), rpb, dscale, dshift and -- both or neither -- gated = (acc, gate, dgate, dbias | None).  Returns [dx or (dx, dacc)] * 2."""
    gated = pa.get("gated") is not None
    assert (pb.get("gated") is not None) == gated
    d = pa["x"].shape[1]
    probs, outs, keep = (_lib.LnBwdProblem * 2)(), [], []
    for q, p in zip(probs, (pa, pb)):
        dout, x = _c(p["dout"]), p["x"]
        rows, dev = x.shape[0], x.device
        dx = torch.empty((rows, d), dtype=torch.float32, device=dev)
        q.dout, q.x, q.mean, q.rstd, q.scale, q.ld_mod = _p(dout), _p(x), _p(p["mean"]), _p(p["rstd"]), _p(p["scale"]), p["scale"].stride(0)
        q.dres, q.rows, q.rows_per_batch = _p(p.get("dres")), rows, p["rpb"]
        q.dx, q.dscale, q.dshift, q.ld_dmod = _p(dx), _p(p["dscale"]), _p(p["dshift"]), p["dscale"].stride(0)
        keep.append(dout)
        if gated:
            acc, gate, dgate, dbias = p["gated"]
            acc = _c(acc)
            dacc = torch.empty((rows, d), dtype=acc.dtype, device=dev)
            q.acc, q.gate, q.ld_gate, q.dacc, q.dgate, q.ld_dgate = _p(acc), _p(gate), gate.stride(0), _p(dacc), _p(dgate), dgate.stride(0)
            q.dbias, q.ld_dbias = _p(dbias), (dbias.stride(0) if dbias is not None else 0)
            keep.append(acc)
            outs.append((dx, dacc))
        else:
            outs.append(dx)
    check(_lib.lib().mmdit_ln_modulate_bwd_pair(ctypes.byref(probs[0]), ctypes.byref(probs[1]), d, _dt(pa["dout"]), _s()), "mmdit_ln_modulate_bwd_pair")
    return outs


def gate_residual_fwd(x, acc, gate, rows_per_batch):
    """x + gate[b] * acc (fp32)."""
    rows, d = x.shape
    out = torch.empty((rows, d), dtype=torch.float32, device=x.device)
    check(_lib.lib().mmdit_gate_residual_fwd(_p(_c(x)), _p(_c(acc)), _dt(acc), _p(gate), gate.stride(0), rows, d, rows_per_batch, _p(out), _s()), "mmdit_gate_residual_fwd")
    return out


def ln_modulate_bwd(dout, x, mean, rstd, scale, dres, rows_per_batch, dscale, dshift, gated=None):
    """Returns dx (fp32) = dres + LN-backward; accumulates into the dscale / dshift views (same leading dim).
    gated = (acc, gate, dgate, dbias | None): also run the backward of the gated residual update that consumes dx
    (mmdit_ln_modulate_bwd_gated) and return (dx, dacc) with dacc = dx * gate in acc's dtype."""
    rows, d = x.shape
    dx = torch.empty((rows, d), dtype=torch.float32, device=x.device)
    if gated is None:
        check(_lib.lib().mmdit_ln_modulate_bwd(_p(_c(dout)), _dt(dout), _p(x), _p(mean), _p(rstd), _p(scale), scale.stride(0), _p(dres),
                                               rows, d, rows_per_batch, _p(dx), _p(dscale), _p(dshift), dscale.stride(0), _s()), "mmdit_ln_modulate_bwd")
        return dx
    acc, gate, dgate, dbias = gated
    dacc = torch.empty((rows, d), dtype=acc.dtype, device=x.device)
    check(_lib.lib().mmdit_ln_modulate_bwd_gated(_p(_c(dout)), _dt(dout), _p(x), _p(mean), _p(rstd), _p(scale), scale.stride(0), _p(dres),
                                                 rows, d, rows_per_batch, _p(dx), _p(dscale), _p(dshift), dscale.stride(0),
                                                 _p(_c(acc)), _dt(acc), _p(gate), gate.stride(0), _p(dacc), _p(dgate), dgate.stride(0),
                                                 _p(dbias), dbias.stride(0) if dbias is not None else 0, _s()), "mmdit_ln_modulate_bwd_gated")
    return dx, dacc


def text_rmsnorm_fwd(x, w1, w2, s1, s2, split, out_dtype):
    batch, tokens, d = x.shape
    out1 = torch.empty((batch * split, d), dtype=out_dtype, device=x.device)
    out2 = torch.empty((batch * (tokens - split), d), dtype=out_dtype, device=x.device)
    check(_lib.lib().mmdit_text_rmsnorm_fwd(_p(_c(x)), _dt(x), _p(w1), _p(w2), _p(s1), _p(s2), batch, tokens, split, d,
                                            _p(out1), _p(out2), _dt(out1), _s()), "mmdit_text_rmsnorm_fwd")
    return out1, out2


def text_rmsnorm_bwd(dout1, dout2, x, w1, w2, s1, s2, split):
    batch, tokens, d = x.shape
    dw1, dw2 = zeros(w1.shape, w1.device), zeros(w2.shape, w2.device)
    ds1, ds2 = zeros(s1.shape, s1.device), zeros(s2.shape, s2.device)
    check(_lib.lib().mmdit_text_rmsnorm_bwd(_p(_c(dout1)), _p(_c(dout2)), _dt(dout1), _p(x), _dt(x), _p(w1), _p(w2), _p(s1), _p(s2),
                                            batch, tokens, split, d, _p(dw1), _p(dw2), _p(ds1), _p(ds2), _s()), "mmdit_text_rmsnorm_bwd")
    return dw1, dw2, ds1, ds2


def qk_norm_rope_fwd(qkv, wq, wk, rope_cos, rope_sin, batch, tokens, heads, s_total, tok0, Q, K, V):
    check(_lib.lib().mmdit_qk_norm_rope_fwd(_p(_c(qkv)), _dt(qkv), _p(wq), _p(wk), _p(rope_cos), _p(rope_sin), batch, tokens, heads, s_total, tok0,
                                            _p(Q), _p(K), _p(V), _s()), "mmdit_qk_norm_rope_fwd")


def qk_norm_rope_bwd(dQ, dK, dV, qkv, wq, wk, rope_cos, rope_sin, batch, tokens, heads, s_total, tok0, dwq, dwk, out_dtype):
    dqkv = torch.empty(qkv.shape, dtype=out_dtype, device=qkv.device)
    check(_lib.lib().mmdit_qk_norm_rope_bwd(_p(dQ), _p(dK), _p(dV), _dt(dQ), _p(qkv), _dt(qkv), _p(wq), _p(wk), _p(rope_cos), _p(rope_sin),
                                            batch, tokens, heads, s_total, tok0, _p(dqkv), _dt(dqkv), _p(dwq), _p(dwk), _s()), "mmdit_qk_norm_rope_bwd")
    return dqkv


def qk_norm_rope_fwd_pair(img, txt, batch, heads, s_total, Q, K, V):
    """The image and the text rows of a block in one launch (mmdit_qk_norm_rope_fwd_pair).  img / txt = (qkv, wq, wk, rope_cos, rope_sin,
    tokens, tok0); both write the joint Q / K / V."""
    probs, keep = (_lib.QkProblem * 2)(), []
    for q, (qkv, wq, wk, rc, rs, tokens, tok0) in zip(probs, (img, txt)):
        qkv = _c(qkv)
        keep.append(qkv)
        q.qkv, q.wq, q.wk, q.rope_cos, q.rope_sin, q.tokens, q.tok0 = _p(qkv), _p(wq), _p(wk), _p(rc), _p(rs), tokens, tok0
    check(_lib.lib().mmdit_qk_norm_rope_fwd_pair(ctypes.byref(probs[0]), ctypes.byref(probs[1]), _dt(img[0]), batch, heads, s_total, _p(Q), _p(K), _p(V), _s()),
          "mmdit_qk_norm_rope_fwd_pair")


def qk_norm_rope_bwd_pair(dQ, dK, dV, img, txt, batch, heads, s_total, out_dtype):
    """Backward of both streams in one launch (mmdit_qk_norm_rope_bwd_pair).  img / txt = (qkv, wq, wk, rope_cos, rope_sin, tokens, tok0, dwq, dwk);
    returns (dqkv_img, dqkv_txt)."""
    probs, outs = (_lib.QkProblem * 2)(), []
    for q, (qkv, wq, wk, rc, rs, tokens, tok0, dwq, dwk) in zip(probs, (img, txt)):
        dqkv = torch.empty(qkv.shape, dtype=out_dtype, device=qkv.device)
        q.qkv, q.wq, q.wk, q.rope_cos, q.rope_sin, q.tokens, q.tok0 = _p(qkv), _p(wq), _p(wk), _p(rc), _p(rs), tokens, tok0
        q.dqkv, q.dwq, q.dwk = _p(dqkv), _p(dwq), _p(dwk)
        outs.append(dqkv)
    check(_lib.lib().mmdit_qk_norm_rope_bwd_pair(ctypes.byref(probs[0]), ctypes.byref(probs[1]), _p(dQ), _p(dK), _p(dV), _dt(dQ), _dt(img[0]), _DT[out_dtype],
                                                 batch, heads, s_total, _s()), "mmdit_qk_norm_rope_bwd_pair")
    return outs


def attn_fwd(Q, K, V, n_img, scale, mode):
    batch, heads, S, hd = Q.shape
    if hd != 64:
        raise RuntimeError("attention kernels are built for head_dim 64 (the reference's dim = 64*num_heads convention)")
    D = heads * hd
    Ox = torch.empty((batch, n_img, D), dtype=torch.bfloat16, device=Q.device)
    Oc = torch.empty((batch, S - n_img, D), dtype=torch.bfloat16, device=Q.device) if S > n_img else None
    lse = torch.empty((batch, heads, S), dtype=torch.float32, device=Q.device)
    check(_lib.lib().mmdit_attn_fwd(_p(Q), _p(K), _p(V), batch, heads, S, n_img, float(scale), mode, _p(Ox), _p(Oc), _p(lse), _s()), "mmdit_attn_fwd")
    return Ox, Oc, lse


def attn_bwd(Q, K, V, Ox, Oc, dOx, dOc, lse, n_img, scale, out_dtype):
    batch, heads, S, hd = Q.shape
    delta = torch.empty((batch, heads, S), dtype=torch.float32, device=Q.device)
    dQ = torch.empty(Q.shape, dtype=out_dtype, device=Q.device)
    dK = torch.empty(Q.shape, dtype=out_dtype, device=Q.device)
    dV = torch.empty(Q.shape, dtype=out_dtype, device=Q.device)
    check(_lib.lib().mmdit_attn_bwd(_p(Q), _p(K), _p(V), _p(Ox), _p(Oc), _p(_c(dOx)), _p(_c(dOc)), _p(lse), _p(delta), batch, heads, S, n_img,
                                    float(scale), _p(dQ), _p(dK), _p(dV), _dt(dQ), _s()), "mmdit_attn_bwd")
    return dQ, dK, dV


def attn_bwd_qk(Q, K, V, Ox, Oc, dOx, dOc, lse, n_img, scale, qkv_x, qkv_c, wq_x, wk_x, wq_c, wk_c, rope_cos, rope_sin, dw4):
    """Attention backward with the QK-norm + RoPE backward in its epilogues (mmdit_attn_bwd_qk): returns (dqkv_x, dqkv_c), the gradients
    of the raw QKV projections, and ADDS the norm-weight gradients into dw4 = the contiguous fp32 [wq_x | wk_x | wq_c | wk_c] (256).
    bf16 only; needs n_img % 32 == 0 (attn_bwd_qk_ok)."""
    batch, heads, S, hd = Q.shape
    delta = torch.empty((batch, heads, S), dtype=torch.float32, device=Q.device)
    dqkv_x = torch.empty(qkv_x.shape, dtype=torch.bfloat16, device=Q.device)
    dqkv_c = torch.empty(qkv_c.shape, dtype=torch.bfloat16, device=Q.device)
    if dw4.dtype != torch.float32 or dw4.numel() != 256 or not dw4.is_contiguous():
        raise RuntimeError("attn_bwd_qk: dw4 is the contiguous fp32 [wq_x | wk_x | wq_c | wk_c] accumulator (256)")
    check(_lib.lib().mmdit_attn_bwd_qk(_p(Q), _p(K), _p(V), _p(Ox), _p(Oc), _p(_c(dOx)), _p(_c(dOc)), _p(lse), _p(delta), batch, heads, S, n_img, float(scale),
                                       _p(_c(qkv_x)), _p(_c(qkv_c)), _p(wq_x), _p(wk_x), _p(wq_c), _p(wk_c), _p(rope_cos), _p(rope_sin),
                                       _p(dqkv_x), _p(dqkv_c), _p(dw4), _s()), "mmdit_attn_bwd_qk")
    return dqkv_x, dqkv_c


def attn_bwd_qk_ok(Q, n_img, qkv_x):
    """The fused form's preconditions: bf16 operands, head_dim 64, image tokens a multiple of the 32-row wave blocks."""
    return Q.is_cuda and Q.shape[-1] == 64 and n_img % 32 == 0 and n_img < Q.shape[2] and qkv_x.dtype == torch.bfloat16


def mlp_act_fwd(gu, hidden, gelu=False):
    rows = gu.shape[0]
    h = torch.empty((rows, hidden), dtype=gu.dtype, device=gu.device)
    fn = _lib.lib().mmdit_gelu_fwd if gelu else _lib.lib().mmdit_swiglu_fwd
    check(fn(_p(_c(gu)), _p(h), _dt(gu), rows, hidden, _s()), "mmdit_mlp_act_fwd")
    return h


def mlp_act_bwd(dh, gu, hidden, dbias, gelu=False):
    rows = gu.shape[0]
    dgu = torch.empty_like(gu)
    fn = _lib.lib().mmdit_gelu_bwd if gelu else _lib.lib().mmdit_swiglu_bwd
    check(fn(_p(_c(dh)), _p(gu), _p(dgu), _dt(gu), rows, hidden, _p(dbias), _s()), "mmdit_mlp_act_bwd")
    return dgu


def mlp_act_bwd_pair(a, b, hidden, gelu=False):
    """mlp_act_bwd of two problems of the same hidden width in one launch (mmdit_mlp_act_bwd_pair).  a / b = (dh, gu, dbias); returns [dgu, dgu]."""
    probs, outs, keep = (_lib.MlpBwdProblem * 2)(), [], []
    for q, (dh, gu, dbias) in zip(probs, (a, b)):
        dh = _c(dh)
        dgu = torch.empty_like(gu)
        q.dh, q.gu, q.dgu, q.rows, q.dbias = _p(dh), _p(gu), _p(dgu), gu.shape[0], _p(dbias)
        keep.append(dh)
        outs.append(dgu)
    check(_lib.lib().mmdit_mlp_act_bwd_pair(ctypes.byref(probs[0]), ctypes.byref(probs[1]), _dt(a[1]), hidden, int(gelu), _s()), "mmdit_mlp_act_bwd_pair")
    return outs


def silu_bwd(dy, pre, out_dtype, dbias, rows_per_bias=0):
    """rows_per_bias > 0: dbias is (rows / rows_per_bias, cols): one bias-gradient row per group of rows (stacked blocks)."""
    rows, cols = pre.shape
    dpre = torch.empty((rows, cols), dtype=out_dtype, device=pre.device)
    check(_lib.lib().mmdit_silu_bwd(_p(_c(dy)), _dt(dy), _p(pre), _p(dpre), _dt(dpre), rows, cols, _p(dbias), int(rows_per_bias), _s()), "mmdit_silu_bwd")
    return dpre


def gate_residual_bwd(dy, acc, gate, rows_per_batch, dgate, dbias, out_dtype):
    """dbias: None, a (d,) vector, or a (batch, d) view of per-batch partial rows (column-sum it afterwards)."""
    rows, d = dy.shape
    dacc = torch.empty((rows, d), dtype=out_dtype, device=dy.device)
    ld_dbias = dbias.stride(0) if (dbias is not None and dbias.dim() == 2) else 0
    check(_lib.lib().mmdit_gate_residual_bwd(_p(_c(dy)), _p(acc), _dt(acc), _p(gate), gate.stride(0), rows, d, rows_per_batch,
                                             _p(dacc), _dt(dacc), _p(dgate), dgate.stride(0), _p(dbias), ld_dbias, _s()), "mmdit_gate_residual_bwd")
    return dacc


def flow_loss(v, x0, eps, accumulation_steps=1, need_grad=True):
    """Rectified-flow loss mean((v - (eps - x0))^2) / accumulation_steps (mmdit_flow_loss): returns (loss 0-dim fp32, dv = d loss / d v
    fp32 like v, or None).  v fp32, x0 / eps of one dtype (bf16: the label is rounded to bf16 as torch's bf16 subtraction does)."""
    n = v.numel()
    if v.dtype != torch.float32 or x0.dtype != eps.dtype or x0.numel() != n or eps.numel() != n or n % 8:
        raise RuntimeError("flow_loss: v fp32, x0 and eps of one dtype and v's size, numel % 8 == 0")
    loss = torch.empty((), dtype=torch.float32, device=v.device)
    part = torch.empty(256, dtype=torch.float32, device=v.device)
    dv = torch.empty(v.shape, dtype=torch.float32, device=v.device) if need_grad else None
    check(_lib.lib().mmdit_flow_loss(_p(_c(v)), _p(_c(x0)), _p(_c(eps)), _dt(x0), n, 1.0 / (n * accumulation_steps), _p(dv), _p(part), _p(loss), _s()), "mmdit_flow_loss")
    return loss, dv


def colsum(x, out):
    rows, cols = x.shape
    check(_lib.lib().mmdit_colsum(_p(x), _dt(x), rows, cols, x.stride(0), _p(out), _s()), "mmdit_colsum")
    return out


def patchify(img, out_dtype):
    batch, ch, H, W = img.shape
    tokens = torch.empty((batch * (H // 2) * (W // 2), ch * 4), dtype=out_dtype, device=img.device)
    check(_lib.lib().mmdit_patchify(_p(_c(img)), _dt(img), batch, ch, H, W, _p(tokens), _dt(tokens), _s()), "mmdit_patchify")
    return tokens


def unpatchify(tokens, batch, ch, H, W, out_dtype):
    img = torch.empty((batch, ch, H, W), dtype=out_dtype, device=tokens.device)
    check(_lib.lib().mmdit_unpatchify(_p(_c(tokens)), _dt(tokens), batch, ch, H, W, _p(img), _dt(img), _s()), "mmdit_unpatchify")
    return img


def time_embed_fwd(t, time_scale, denom, out_dtype):
    batch, dim = t.shape[0], denom.shape[0]
    out = torch.empty((batch, dim), dtype=out_dtype, device=t.device)
    check(_lib.lib().mmdit_time_embed_fwd(_p(t), _p(time_scale), _p(denom), batch, dim, _p(out), _dt(out), _s()), "mmdit_time_embed_fwd")
    return out


def time_embed_bwd(dout, t, time_scale, denom):
    batch, dim = t.shape[0], denom.shape[0]
    dts = zeros(time_scale.shape, time_scale.device)
    check(_lib.lib().mmdit_time_embed_bwd(_p(_c(dout)), _dt(dout), _p(t), _p(time_scale), _p(denom), batch, dim, _p(dts), _s()), "mmdit_time_embed_bwd")
    return dts


# ---------------------------------------------------------------------------------------------
# FLUX VAE building blocks (csrc/vae.hip); activations NHWC bf16
# ---------------------------------------------------------------------------------------------
def vae_nchw_to_nhwc(x, c_padded, scale=1.0, shift=0.0):
    B, C, H, W = x.shape
    out = torch.empty((B, H, W, c_padded), dtype=torch.bfloat16, device=x.device)
    check(_lib.lib().mmdit_vae_nchw_to_nhwc(_p(_c(x)), _dt(x), B, C, H, W, c_padded, float(scale), float(shift), _p(out), _s()), "mmdit_vae_nchw_to_nhwc")
    return out


def vae_nhwc_to_nchw(x2d, B, C, H, W, lo=-float("inf"), hi=float("inf")):
    """x2d: fp32 (B*H*W, ld) rows = pixels; returns fp32 (B, C, H, W) of the first C columns, clamped."""
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=x2d.device)
    check(_lib.lib().mmdit_vae_nhwc_to_nchw(_p(_c(x2d)), B, C, H, W, x2d.shape[1], float(lo), float(hi), _p(out), _s()), "mmdit_vae_nhwc_to_nchw")
    return out


def vae_im2col3x3(x, mode=0):
    """x: bf16 (B, H, W, C) -> bf16 (B*Ho*Wo, 9*C); mode 0 same / 1 stride-2 downsample / 2 nearest-x2 upsample then conv."""
    B, H, W, C = x.shape
    Ho, Wo = (H // 2, W // 2) if mode == 1 else (2 * H, 2 * W) if mode == 2 else (H, W)
    out = torch.empty((B * Ho * Wo, 9 * C), dtype=torch.bfloat16, device=x.device)
    check(_lib.lib().mmdit_vae_im2col3x3(_p(_c(x)), B, H, W, C, mode, _p(out), _s()), "mmdit_vae_im2col3x3")
    return out, Ho, Wo


def vae_groupnorm(x, gamma, beta, B, H, W, groups, eps, silu, padded_out=None):
    """x: (B*H*W, C) fp32 or bf16 -> bf16 (B*H*W, C), or into the interior of `padded_out` (B, H+2, W+2, C) with zero borders."""
    C = x.shape[1]
    sums = torch.zeros(B * groups * 2, dtype=torch.float32, device=x.device)
    out = padded_out if padded_out is not None else torch.empty((B * H * W, C), dtype=torch.bfloat16, device=x.device)
    check(_lib.lib().mmdit_vae_groupnorm(_p(_c(x)), _dt(x), _p(gamma), _p(beta), B, H, W, C, groups, float(eps), int(silu), _p(sums), _p(out),
                                         int(padded_out is not None), _s()), "mmdit_vae_groupnorm")
    return out


def vae_pad_cast(x, B, H, W, padded_out, upsample=False):
    """x: (B*H*W, C) fp32 or bf16 -> interior of the zero-bordered bf16 `padded_out` (B, uH+2, uW+2, C); nearest x2 if upsample."""
    check(_lib.lib().mmdit_vae_pad_cast(_p(_c(x)), _dt(x), B, H, W, x.shape[1], int(upsample), _p(padded_out), _s()), "mmdit_vae_pad_cast")
    return padded_out


def vae_softmax_rows(x, scale, cols=None):
    """softmax over the first `cols` columns of every row of x (fp32); the remaining (padding) columns come out as 0."""
    rows, ld = x.shape
    cols = ld if cols is None else cols
    out = torch.empty((rows, ld), dtype=torch.bfloat16, device=x.device)
    check(_lib.lib().mmdit_vae_softmax_rows(_p(_c(x)), rows, cols, ld, float(scale), _p(out), _s()), "mmdit_vae_softmax_rows")
    return out
