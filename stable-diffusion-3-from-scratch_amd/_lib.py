"""ctypes binding of libmmdit_hip.so (C ABI: include/mmdit_hip.h).

There is no fallback: if the library is missing or a call fails, a RuntimeError is
raised.  The product path never routes through PyTorch math or the CPU oracle.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
def experiment(name: str, default: str) -> str:
    """A/B experiment switch (the fused-vs-unfused, schedule and layout switches listed in tools/README.md).  They are read ONLY when
    MMDIT_EXPERIMENTS=1 is set: the product has one path per operation, the switches exist for same-box A/B measurements."""
    return os.environ.get(name, default) if os.environ.get("MMDIT_EXPERIMENTS") == "1" else default


LIB_PATH = os.environ.get("MMDIT_LIB") or os.path.join(_HERE, "libmmdit_hip.so")   # MMDIT_LIB: A/B a scratch build
HEADER_PATH = os.path.join(_HERE, "..", "include", "mmdit_hip.h")

F32, BF16 = 0, 1
ACT_NONE, ACT_SILU, ACT_SWIGLU, ACT_SWIGLU_BWD = 0, 1, 2, 3
PREC_BF16, PREC_SPLIT = 0, 1
FP8 = 2   # dtype code of OCP e4m3 GEMM operands (stored in torch.uint8 / float8_e4m3fn tensors)

_vp, _i, _i64, _f, _d = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double


class GemmArgs(ctypes.Structure):
    _fields_ = [
        ("A", _vp), ("a_dtype", _i), ("a_kmajor", _i), ("lda", _i64),
        ("B", _vp), ("b_dtype", _i), ("b_kmajor", _i), ("ldb", _i64),
        ("C", _vp), ("c_dtype", _i), ("ldc", _i64),
        ("M", _i), ("N", _i), ("K", _i),
        ("bias", _vp),
        ("act", _i),
        ("gate", _vp), ("ld_gate", _i64), ("rows_per_batch", _i),
        ("residual", _vp), ("ld_res", _i64),
        ("aux", _vp), ("aux_dtype", _i), ("ld_aux", _i64),
        ("accumulate", _i),
        ("precision", _i),
        ("split_k", _i),
        ("stream_k", _i),
        ("conv_mode", _i), ("conv_H", _i), ("conv_W", _i), ("conv_C", _i),
        ("scale_a", _vp), ("scale_b", _vp), ("scale_mode", _i), ("c_scales", _vp),
        ("dbias", _vp),
    ]


# argument types of every entry point, in header order
class LnFwdProblem(ctypes.Structure):      # mirrors mmdit_ln_fwd_problem (include/mmdit_hip.h)
    _fields_ = [("x", ctypes.c_void_p), ("acc", ctypes.c_void_p), ("gate", ctypes.c_void_p), ("ld_gate", ctypes.c_int64), ("x_out", ctypes.c_void_p),
                ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p), ("ld_mod", ctypes.c_int64), ("rows", ctypes.c_int), ("rows_per_batch", ctypes.c_int),
                ("out", ctypes.c_void_p), ("mean", ctypes.c_void_p), ("rstd", ctypes.c_void_p)]


class LnBwdProblem(ctypes.Structure):      # mirrors mmdit_ln_bwd_problem
    _fields_ = [("dout", ctypes.c_void_p), ("x", ctypes.c_void_p), ("mean", ctypes.c_void_p), ("rstd", ctypes.c_void_p), ("scale", ctypes.c_void_p),
                ("ld_mod", ctypes.c_int64), ("dres", ctypes.c_void_p), ("rows", ctypes.c_int), ("rows_per_batch", ctypes.c_int),
                ("dx", ctypes.c_void_p), ("dscale", ctypes.c_void_p), ("dshift", ctypes.c_void_p), ("ld_dmod", ctypes.c_int64),
                ("acc", ctypes.c_void_p), ("gate", ctypes.c_void_p), ("ld_gate", ctypes.c_int64), ("dacc", ctypes.c_void_p), ("dgate", ctypes.c_void_p),
                ("ld_dgate", ctypes.c_int64), ("dbias", ctypes.c_void_p), ("ld_dbias", ctypes.c_int64)]


class QkProblem(ctypes.Structure):        # mirrors mmdit_qk_problem
    _fields_ = [("qkv", ctypes.c_void_p), ("wq", ctypes.c_void_p), ("wk", ctypes.c_void_p), ("rope_cos", ctypes.c_void_p), ("rope_sin", ctypes.c_void_p),
                ("tokens", ctypes.c_int), ("tok0", ctypes.c_int), ("dqkv", ctypes.c_void_p), ("dwq", ctypes.c_void_p), ("dwk", ctypes.c_void_p)]


class MlpBwdProblem(ctypes.Structure):     # mirrors mmdit_mlp_bwd_problem
    _fields_ = [("dh", ctypes.c_void_p), ("gu", ctypes.c_void_p), ("dgu", ctypes.c_void_p), ("rows", ctypes.c_int), ("dbias", ctypes.c_void_p)]


class QkEpilogue(ctypes.Structure):
    """mmdit_qk_epilogue"""
    _fields_ = [("wq", _vp), ("wk", _vp), ("rope_cos", _vp), ("rope_sin", _vp), ("tokens", _i), ("tok0", _i)]


_SIGNATURES = {
    "mmdit_abi_version": ([], _i),
    "mmdit_struct_size": ([_i], _i),
    "mmdit_build_arch": ([], ctypes.c_char_p),
    "mmdit_gemm": ([ctypes.POINTER(GemmArgs), _vp], _i),
    "mmdit_gemm_grouped": ([ctypes.POINTER(GemmArgs), _i, _vp], _i),
    "mmdit_gemm_plan": ([ctypes.POINTER(GemmArgs), _i], _i),
    "mmdit_gemm_set_workspace": ([_vp, ctypes.c_longlong], _i),
    "mmdit_debug_occupy": ([_i, ctypes.c_longlong, _vp], _i),
    "mmdit_gemm_set_claiming": ([_i], _i),
    "mmdit_gemm_get_claiming": ([], _i),
    "mmdit_set_cu_budget": ([_i], _i),
    "mmdit_get_cu_budget": ([], _i),
    "mmdit_gemm_qkv_norm_rope": ([ctypes.POINTER(GemmArgs), ctypes.POINTER(QkEpilogue), _i, _i, _i, _vp, _vp, _vp, _vp], _i),
    "mmdit_gemm_zero_mask": ([ctypes.POINTER(GemmArgs), _i, ctypes.POINTER(ctypes.c_uint)], _i),
    "mmdit_fp8_amax": ([_vp, _i, _i64, _vp, _vp], _i),
    "mmdit_fp8_quantize": ([_vp, _i, _i64, _vp, _vp, _vp, _vp], _i),
    "mmdit_fp8_quantize_delayed": ([_vp, _i, _i64, _vp, _i, ctypes.c_float, _vp, _vp], _i),
    "mmdit_mxfp8_quantize": ([_vp, _i, _i, _i, _i64, _vp, _vp, _vp], _i),
    "mmdit_ln_modulate_fwd_mx": ([_vp, _vp, _i, _vp, _i64, _vp, _vp, _vp, _i64, _i, _i, _i, _vp, _vp, _vp, _vp, _vp], _i),
    "mmdit_swiglu_fwd_mx": ([_vp, _i, _i, _i, _vp, _vp, _vp], _i),
    "mmdit_attn_fwd_mx": ([_vp, _vp, _vp, _i, _i, _i, _i, ctypes.c_float, _vp, _vp, _vp, _vp, _vp], _i),
    "mmdit_cast": ([_vp, _i, _vp, _i, _i64, _vp], _i),
    "mmdit_ln_modulate_fwd": ([_vp, _vp, _vp, _i64, _i, _i, _i, _vp, _i, _vp, _vp, _vp], _i),
    "mmdit_ln_modulate_fwd_res": ([_vp, _vp, _i, _vp, _i64, _vp, _vp, _vp, _i64, _i, _i, _i, _vp, _i, _vp, _vp, _vp], _i),
    "mmdit_ln_modulate_fwd_pair": ([ctypes.POINTER(LnFwdProblem), ctypes.POINTER(LnFwdProblem), _i, _i, _i, _vp], _i),
    "mmdit_ln_modulate_bwd_pair": ([ctypes.POINTER(LnBwdProblem), ctypes.POINTER(LnBwdProblem), _i, _i, _vp], _i),
    "mmdit_qk_norm_rope_fwd_pair": ([ctypes.POINTER(QkProblem), ctypes.POINTER(QkProblem), _i, _i, _i, _i, _vp, _vp, _vp, _vp], _i),
    "mmdit_qk_norm_rope_bwd_pair": ([ctypes.POINTER(QkProblem), ctypes.POINTER(QkProblem), _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp], _i),
    "mmdit_mlp_act_bwd_pair": ([ctypes.POINTER(MlpBwdProblem), ctypes.POINTER(MlpBwdProblem), _i, _i, _i, _vp], _i),
    "mmdit_gate_residual_fwd": ([_vp, _vp, _i, _vp, _i64, _i, _i, _i, _vp, _vp], _i),
    "mmdit_ln_modulate_bwd": ([_vp, _i, _vp, _vp, _vp, _vp, _i64, _vp, _i, _i, _i, _vp, _vp, _vp, _i64, _vp], _i),
    "mmdit_ln_modulate_bwd_gated": ([_vp, _i, _vp, _vp, _vp, _vp, _i64, _vp, _i, _i, _i, _vp, _vp, _vp, _i64, _vp, _i, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp], _i),
    "mmdit_text_rmsnorm_fwd": ([_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp], _i),
    "mmdit_text_rmsnorm_bwd": ([_vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp], _i),
    "mmdit_qk_norm_rope_fwd": ([_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp], _i),
    "mmdit_qk_norm_rope_bwd": ([_vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp], _i),
    "mmdit_attn_fwd": ([_vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp, _vp], _i),
    "mmdit_attn_bwd_qk": ([_vp] * 9 + [_i, _i, _i, _i, _f] + [_vp] * 11 + [_vp], _i),
    "mmdit_attn_bwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _i, _vp], _i),
    "mmdit_swiglu_fwd": ([_vp, _vp, _i, _i, _i, _vp], _i),
    "mmdit_swiglu_bwd": ([_vp, _vp, _vp, _i, _i, _i, _vp, _vp], _i),
    "mmdit_gelu_fwd": ([_vp, _vp, _i, _i, _i, _vp], _i),
    "mmdit_gelu_bwd": ([_vp, _vp, _vp, _i, _i, _i, _vp, _vp], _i),
    "mmdit_silu_bwd": ([_vp, _i, _vp, _vp, _i, _i, _i, _vp, _i, _vp], _i),
    "mmdit_gate_residual_bwd": ([_vp, _vp, _i, _vp, _i64, _i, _i, _i, _vp, _i, _vp, _i64, _vp, _i64, _vp], _i),
    "mmdit_flow_loss": ([_vp, _vp, _vp, _i, _i64, _f, _vp, _vp, _vp, _vp], _i),
    "mmdit_colsum": ([_vp, _i, _i, _i, _i64, _vp, _vp], _i),
    "mmdit_vae_nchw_to_nhwc": ([_vp, _i, _i, _i, _i, _i, _i, ctypes.c_float, ctypes.c_float, _vp, _vp], _i),
    "mmdit_vae_nhwc_to_nchw": ([_vp, _i, _i, _i, _i, _i, ctypes.c_float, ctypes.c_float, _vp, _vp], _i),
    "mmdit_vae_im2col3x3": ([_vp, _i, _i, _i, _i, _i, _vp, _vp], _i),
    "mmdit_vae_groupnorm": ([_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, ctypes.c_float, _i, _vp, _vp, _i, _vp], _i),
    "mmdit_vae_pad_cast": ([_vp, _i, _i, _i, _i, _i, _i, _vp, _vp], _i),
    "mmdit_vae_softmax_rows": ([_vp, _i, _i, _i, ctypes.c_float, _vp, _vp], _i),
    "mmdit_patchify": ([_vp, _i, _i, _i, _i, _i, _vp, _i, _vp], _i),
    "mmdit_unpatchify": ([_vp, _i, _i, _i, _i, _i, _vp, _i, _vp], _i),
    "mmdit_time_embed_fwd": ([_vp, _vp, _vp, _i, _i, _vp, _i, _vp], _i),
    "mmdit_time_embed_bwd": ([_vp, _i, _vp, _vp, _vp, _i, _i, _vp, _vp], _i),
    "mmdit_grad_sumsq": ([_vp, _vp, _vp, _i, _vp, _vp], _i),
    "mmdit_clip_coef": ([_vp, _i, _vp, _f, _vp, _vp], _i),
    "mmdit_adamw_step": ([_vp, _vp, _vp, _i, _vp, _vp, _d, _d, _d, _d, _d, _vp], _i),
    "mmdit_adamw_step_dlr": ([_vp, _vp, _vp, _i, _vp, _vp, _vp, _d, _d, _d, _d, _vp], _i),
    "mmdit_cast_multi": ([_vp, _vp, _vp, _i, _vp], _i),
}
ADAMW_CHUNK = 65536   # MMDIT_ADAMW_CHUNK
ABI_VERSION = 8       # MMDIT_ABI_VERSION of include/mmdit_hip.h this binding mirrors
# struct ids of mmdit_struct_size() -> ctypes mirrors (None: laid out with numpy record dtypes in optim.py / ops.py: 48 / 24 bytes)
_STRUCTS = [("mmdit_gemm_args", GemmArgs), ("mmdit_ln_fwd_problem", LnFwdProblem), ("mmdit_ln_bwd_problem", LnBwdProblem),
            ("mmdit_qk_problem", QkProblem), ("mmdit_mlp_bwd_problem", MlpBwdProblem), ("mmdit_adamw_tensor", None), ("mmdit_cast_tensor", None), ("mmdit_qk_epilogue", QkEpilogue)]

_lib = None


def declared_symbols():
    """Entry points declared by include/mmdit_hip.h (parsed from the header text)."""
    with open(HEADER_PATH) as f:
        txt = f.read()
    return sorted(set(re.findall(r"\b(mmdit_[a-z0-9_]+)\s*\(", txt)) - {"mmdit_stream_t"})


def lib():
    """Load the HIP library (once).  Raises RuntimeError if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python __graft_entry__.py build` "
                "(hipcc --offload-arch=gfx950).  There is no PyTorch/CPU fallback for the MMDiT hot path.")
        try:
            # The library must share PyTorch's HIP runtime instance and device context: initialise torch's first when a GPU
            # is present (loading this library before torch has touched the GPU left its launches with hipErrorNoDevice).
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        for name, (argtypes, restype) in _SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError -> symbol missing: fail loudly
            fn.argtypes = argtypes
            fn.restype = restype
        # a library built for another ABI (a stale scratch build behind MMDIT_LIB, an old .so next to a new binding) must not be called
        if L.mmdit_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH}: ABI version {L.mmdit_abi_version()}, this binding is written for {ABI_VERSION}: rebuild (python __graft_entry__.py build)")
        for which, (name, struct) in enumerate(_STRUCTS):
            if struct is not None and L.mmdit_struct_size(which) != ctypes.sizeof(struct):
                raise RuntimeError(f"{LIB_PATH}: sizeof({name}) = {L.mmdit_struct_size(which)}, the binding's ctypes mirror has {ctypes.sizeof(struct)} bytes")
        _lib = L
    return _lib


ERR_ARG, ERR_DTYPE, ERR_SHAPE = -1, -2, -3      # MMDIT_ERR_* of include/mmdit_hip.h


def check(status: int, what: str):
    if status != 0:
        kind = {-1: "invalid argument", -2: "dtype combination not built", -3: "unsupported shape"}.get(status, f"hipError {status}")
        raise RuntimeError(f"{what} failed: {kind}")
