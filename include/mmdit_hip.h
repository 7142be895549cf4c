/*
 * mmdit_hip.h -- C ABI of libmmdit_hip.so: the MI355X (gfx950) kernels behind the
 * MMDiT flow-matching training path of gmongaras/Stable-Diffusion-3-From-Scratch.
 *
 * The reference has no native code: its GPU arithmetic is reached through
 * torch / flash-attn / xformers call sites in src/blocks and src/models
 * (SURVEY.md 2b).  Each entry point below replaces one of those call sites and
 * cites it (file:line into the reference's src/).  Conventions:
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers
 *     owned by the caller (workspaces and saved-for-backward buffers included);
 *   - the library allocates nothing, never synchronises the device and launches
 *     only on the stream it is given; its only mutable state are three per-device
 *     settings made by the caller: mmdit_gemm_set_workspace (a pointer),
 *     mmdit_gemm_set_claiming (a flag) and mmdit_set_cu_budget (an integer);
 *   - re-entrant: safe to call from the autograd worker thread;
 *   - return value: 0 on success, MMDIT_ERR_* (<0) for invalid arguments, or a
 *     positive hipError_t from the launch.
 * dtype codes: MMDIT_F32 / MMDIT_BF16.  "T" below = activation dtype of the
 * caller's precision mode (bf16 = fast mode, f32 = parity mode).
 */
#ifndef MMDIT_HIP_H
#define MMDIT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMDIT_F32 0
#define MMDIT_BF16 1
#define MMDIT_FP8 2   /* OCP e4m3fn, GEMM operands only (inference path): see mmdit_fp8_quantize */

#define MMDIT_ERR_ARG (-1)      /* null pointer / bad size / misaligned */
#define MMDIT_ERR_DTYPE (-2)    /* dtype combination not built */
#define MMDIT_ERR_SHAPE (-3)    /* unsupported shape (e.g. head_dim != 64) */

#define MMDIT_ACT_NONE 0
#define MMDIT_ACT_SILU 1
#define MMDIT_ACT_SWIGLU 2      /* packed SwiGLU up-projection (MLP.py:15-40 / xformers SwiGLU w12): B = w12 [2h, K], N = 2h;
                                 * aux[M, 2h] (bf16, optional: training keeps it for backward) = [g | u] = A B^T + bias, C[M, h] (bf16) =
                                 * silu(g) * u computed from the bf16-rounded [g | u] (== mmdit_swiglu_fwd of the plain GEMM's output).
                                 * Row-major bf16 operands with K % 64 == 0 or e4m3 operands with K % 128 == 0, h % 128 == 0 (else
                                 * MMDIT_ERR_SHAPE: run the GEMM and mmdit_swiglu_fwd separately); no gate / residual / split. */

#define MMDIT_ACT_SWIGLU_BWD 3  /* data gradient of the SwiGLU down-projection fused with the activation backward (MLP.py:32 backward / xformers
                                 * SwiGLU w3, then silu(g) * u): A = dY [M, K] row-major bf16, B = w3 [K, N] (b_kmajor = 1, N = h), aux [M, 2h] (bf16,
                                 * INPUT) = the [g | u] kept by MMDIT_ACT_SWIGLU, C [M, 2h] (bf16, ldc >= 2h) = d[g | u] computed from the
                                 * bf16-rounded dh = A B (== mmdit_gemm followed by mmdit_swiglu_bwd, bit for bit, without the dh round trip);
                                 * dbias (optional, fp32 [2h], PRE-ZEROED or accumulating) += column sums of d[g | u].  h % 8 == 0, K % 64 == 0;
                                 * MMDIT_ERR_SHAPE when the planner would not give the launch to the 8-phase 256 x 256 kernel (run the two
                                 * passes instead); no bias / gate / residual / split. */

#define MMDIT_PREC_BF16 0       /* single-pass bf16 MFMA operands, fp32 accumulate */
#define MMDIT_PREC_SPLIT 1      /* 3-term split-bf16, 6-pass MFMA: fp32-exact products (parity mode) */

typedef void* mmdit_stream_t;   /* hipStream_t */

/* library/ABI version and build target ("gfx950").  MMDIT_ABI_VERSION changes with every struct-layout or signature change; the
 * Python binding (_lib.py) refuses a library whose version or struct sizes differ from what it was written for, so a scratch build
 * (MMDIT_LIB=...) with another layout fails loudly instead of reading past a struct.  mmdit_struct_size(which): sizeof of
 * 0 mmdit_gemm_args, 1 mmdit_ln_fwd_problem, 2 mmdit_ln_bwd_problem, 3 mmdit_qk_problem, 4 mmdit_mlp_bwd_problem,
 * 5 mmdit_adamw_tensor, 6 mmdit_cast_tensor, 7 mmdit_qk_epilogue; -1 for an unknown id. */
#define MMDIT_ABI_VERSION 8
int mmdit_abi_version(void);
int mmdit_struct_size(int which);
const char* mmdit_build_arch(void);

/* ---------------------------------------------------------------------------
 * GEMM:  C[M,N] = epilogue( A[M,K] * B[N,K]^T )         fp32 accumulation on MFMA
 * Replaces every nn.Linear on the path (Attention.py:130-135, 424-425;
 * MLP.py:19,32 / xformers SwiGLU w12,w3; Norm.py:13-14; Transformer_Block_Dual.py
 * :25-28,49-53; diff_model.py:306-332,339) and their autograd dgrad/wgrad.
 *   a_kmajor=0: A stored (M,K) row-major, lda >= K.   a_kmajor=1: A stored (K,M), lda >= M.
 *   b_kmajor=0: B stored (N,K) row-major (nn.Linear weight), ldb >= K.  b_kmajor=1: B stored (K,N).
 *   forward = (0,0); dgrad dX = dY * W uses (0,1) with B=W; wgrad dW = dY^T X uses (1,1).
 * Epilogue, in order:  v = acc (+ bias[n]);  aux = v (optional, raw pre-activation copy);
 *   v = act(v);  v = residual[m,n] + gate[(m / rows_per_batch), n] * v   (gate optional -> residual + v);
 *   C = (accumulate ? C + v : v).
 * Requirements: K%8==0 for row-major operands, M%8==0 / N%8==0 for k-major operands,
 * 16-byte aligned bases and leading dimensions.
 * ------------------------------------------------------------------------- */
typedef struct {
  const void* A; int a_dtype; int a_kmajor; int64_t lda;
  const void* B; int b_dtype; int b_kmajor; int64_t ldb;
  void* C; int c_dtype; int64_t ldc;
  int M, N, K;
  const float* bias;
  int act;
  const float* gate; int64_t ld_gate; int rows_per_batch;
  const float* residual; int64_t ld_res;
  void* aux; int aux_dtype; int64_t ld_aux;
  int accumulate;
  int precision;
  int split_k;    /* >1: K is split over split_k workgroups per tile which add atomically into a PRE-ZEROED fp32 C
                     (skinny problems, e.g. the per-sample modulation GEMMs with M = batch); bf16 precision, K%64==0,
                     no act/aux/gate/accumulate */
  int stream_k;   /* !=0: stream-K decomposition allowed: C is fp32 and PRE-ZEROED, the (tile, K-tile) units of the whole
                     launch are split evenly over the resident workgroups and partial tiles are added atomically (weight
                     gradients: few output tiles, very long reductions).  Ignored when the fast path does not apply. */
  /* Implicit-GEMM 3x3 convolution on the fast path (FLUX VAE, nn.Conv2d(C, N, 3)): conv_mode != 0 makes A a ZERO-BORDERED
   * NHWC bf16 tensor (batch, conv_H + 2, conv_W + 2, conv_C) and K = 9 * conv_C ordered (kh, kw, c) -- B is the weight re-laid
   * as [N][kh][kw][C].  conv_mode 1: stride 1, padding 1, M = batch * conv_H * conv_W.  conv_mode 2: stride 2 after padding
   * (0,1,0,1) (diffusers Downsample2D), M = batch * (conv_H/2) * (conv_W/2).  Output rows are the output pixels in NHWC
   * order.  Needs bf16 operands and conv_C % 32 == 0; lda is ignored. */
  int conv_mode, conv_H, conv_W, conv_C;
  /* fp8 operands (a_dtype == b_dtype == MMDIT_FP8, row-major, K % 128 == 0), the matrix instruction is the 64-wide
   * v_mfma_scale_f32_32x32x64_f8f6f4.
   *   scale_mode 0 (per-tensor): C = scale_a[0] * scale_b[0] * (A_q B_q^T), then the usual epilogue; scale_*: device pointers to the
   *     fp32 dequantisation scales written by mmdit_fp8_quantize / mmdit_fp8_quantize_delayed (unit block scales in the MFMA).
   *   scale_mode 1 (MX, OCP microscaling): every 32 consecutive K values of a row share one E8M0 scale 2^(e-127); scale_* point to
   *     the byte tensors written by mmdit_mxfp8_quantize (rows = M for A, N for B).  Layout: for every 64-wide
   *     K half the rows in groups of 128 (rows padded to a multiple of 128), inside a group the byte of (row, half-block h) at
   *     (row & 31) * 8 + h * 4 + ((row >> 5) & 3): a 256-row tile's scales of one half are one contiguous 512-byte run (one LDS-DMA
   *     piece), and the bytes of the four 32-row blocks a wave multiplies sit in one dword per lane (op_sel picks the block).  Size
   *     (K / 64) * rows_pad * 2 + 512 bytes (tile loads of the last rows read, and ignore, what follows).  The block scales are
   *     applied by the matrix instruction itself. */
  const void* scale_a;
  const void* scale_b;
  int scale_mode;
  /* MMDIT_ACT_SWIGLU with MX operands only: c_dtype = MMDIT_FP8 and c_scales != NULL make the activation leave as MX e4m3 codes
   * (C: (M, N/2) bytes, ldc in bytes) + E8M0 block scales (layout / size as scale_mode 1 for an (M, N/2) operand), so that the
   * down-projection GEMM reads it without a quantise pass; aux must be NULL (inference). */
  void* c_scales;
  /* MMDIT_ACT_SWIGLU_BWD: bias gradient of the packed up-projection (column sums of C), added atomically; NULL: not wanted */
  float* dbias;
} mmdit_gemm_args;
int mmdit_gemm(const mmdit_gemm_args* args, mmdit_stream_t stream);
/* Grouped launch: count (1..12) independent problems of the SAME kernel variant (dtypes, layouts,
 * precision, act, accumulate, aux dtype) share one grid, e.g. the image and the text stream of a block
 * (Attention.py:130-135 issues them as separate Linears) or all weight-gradient GEMMs of a block. */
int mmdit_gemm_grouped(const mmdit_gemm_args* args, int count, mmdit_stream_t stream);
/* Which outputs of a K-decomposed launch (stream_k / split_k requests: the weight gradients) must be ZERO when the launch starts,
 * i.e. receive atomically added partial tiles: bit i of *mask = problem i.  With the round + tail schedule these are only the
 * problems that own tiles of the split tail (MMDiT-B: one of a block's eight weight gradients); whole-K tiles are stored, not added,
 * and need no zero-fill.  Same planner as the launch itself (nothing is launched). */
int mmdit_gemm_zero_mask(const mmdit_gemm_args* args, int count, unsigned* mask);
/* The QKV projection of an attention block with the per-head QK RMSNorm, the axial RoPE and the joint-layout store in the GEMM's epilogue
 * (Attention.py:118-135, 174-194, 258-261): one launch for the image and the text stream (count = 1 or 2 problems, in that order).
 * args[i]: bf16 row-major A (rows, K), bf16 row-major packed weight B (3 * heads * 64, K) = [q | k | v] rows, bf16 C (rows, 3 * heads * 64)
 * = the raw projection, kept for backward: its q and k columns exactly as mmdit_gemm writes them, its v columns NOT written (V below is
 * the same data; the backward kernels read only q and k of C); no bias, no activation.  In addition the rows are written
 * to Q, K, V (batch, heads, s_total, 64) bf16: token n of sample b of stream i (row b * tokens + n) at position tok0 + n; q and k are
 * RMS-normalised over the 64 features of their head (weights wq / wk, eps = finfo(float32).eps) and, when rope_cos / rope_sin (tokens, 64)
 * are given, rotated -- the arithmetic of mmdit_qk_norm_rope_fwd on the ROUNDED raw values, i.e. the same results without its pass.
 * MMDIT_ERR_SHAPE when the planner would not give these problems to the lean wide-slot kernel (run mmdit_gemm_grouped +
 * mmdit_qk_norm_rope_fwd_pair instead).  args[i].C may be NULL (inference: nobody reads the raw projection): the q / k columns are then not written
 * either -- by the 8-phase kernel's epilogue only (e4m3 operands with MX scales, or bf16 operands with tile claiming on), MMDIT_ERR_SHAPE otherwise. */
typedef struct mmdit_qk_epilogue {
  const float* wq; const float* wk;
  const float* rope_cos; const float* rope_sin;
  int tokens, tok0;
} mmdit_qk_epilogue;
int mmdit_gemm_qkv_norm_rope(const mmdit_gemm_args* args, const mmdit_qk_epilogue* qk, int count, int heads, int s_total,
                             void* Q, void* K, void* V, mmdit_stream_t stream);
/* Optional device workspace of the GEMM launches, owned by the caller.  Layout: bytes [0, 4096) tickets and [4096, 8192) scheduler words, both
 * ZERO-FILLED by the caller once (every launch leaves them zero again), then 256 KiB slots for partial tiles.  With it
 *   - (with mmdit_gemm_set_claiming(1)) the persistent 8-phase launches (256-row tiles, more tiles than compute units) CLAIM their tiles from eight per-XCD queues whose heads are the
 *     scheduler words (one returning atomic per tile, issued a tile ahead) instead of walking them in a fixed stride: a workgroup that becomes
 *     resident late -- another kernel, e.g. a collective's channels on the reducer's stream, holds its compute unit -- finds the queues empty and
 *     leaves, where the fixed stride would run its whole share as a second round (measured with mmdit_debug_occupy: DESIGN.md 5).  Which tiles exist
 *     and what each computes does not depend on who claims it: results are bit-identical to the static walk;
 *   - the partial tiles of a weight-gradient launch's split tail (k-major x k-major, fp32 out, K-decomposed) are stored to per-slice slots and summed
 *     in slice order by the last slice to arrive (ticket counters) instead of being added with fp32 atomics -- faster (an atomic 256x256 partial costs
 *     ~0.6 us of launch time), deterministic, and the outputs need no zero-fill (mmdit_gemm_zero_mask reports none); a launch that needs more slots
 *     than fit falls back to atomics.
 * This registration, mmdit_gemm_set_claiming and mmdit_set_cu_budget are the library's only mutable state, all per device (hipSetDevice first).  Launches that use the
 * workspace must be stream-ordered among themselves (one workspace per device; the scheduler words are handed out in a ring of 64 launches).
 * ptr = NULL, bytes = 0 removes it (static tile walk, atomics).  bytes >= 8192 + 262144.  The library never allocates. */
int mmdit_gemm_set_workspace(void* ptr, long long bytes);
/* Dynamic tile claiming of the persistent 8-phase launches (above) on / off, per device, default OFF: it costs a launch ~3 us (the first claim, the LDS
 * hand-over of each claimed position, one look at the other queues at the end: +0.12 ms = +0.4 % on the MMDiT-B step) and pays when another kernel holds compute units
 * while the GEMMs run -- model_trainer turns it on when gradients are reduced (RCCL's channels on the reducer's stream).  Needs the workspace.
 * bf16-operand launches only (the training step): e4m3-operand and convolution launches keep the static walk whatever the switch says.  With claiming
 * on, mmdit_gemm_qkv_norm_rope runs bf16 problems on the 8-phase kernel (claimed) instead of the wide-slot kernel: the same values to bf16 rounding. */
int mmdit_gemm_set_claiming(int on);
int mmdit_gemm_get_claiming(void);
/* Test / measurement aid: `wgs` (1..256) one-wave workgroups with 1 KiB of LDS each sleep-spin for `cycles` shader cycles on `stream` -- a stand-in for a
 * long-running kernel (a collective's channels) that keeps 160-KiB GEMM workgroups off `wgs` compute units.  tests/test_kernels_gpu.py, tools/probes/cu_contention.py. */
int mmdit_debug_occupy(int wgs, long long cycles, mmdit_stream_t stream);
/* Compute units the GEMM PLANNER counts on (default: all of the device's, hipDeviceAttributeMultiprocessorCount -- 256 on an MI355X): rounds, tile
 * configurations and split tails are sized for that number, and a launch that is not claimed (see above) also limits its persistent grid to it.  A claimed
 * launch covers the whole device whatever the budget: a workgroup whose compute unit is taken by another kernel starts late and finds nothing, one whose
 * compute unit is free does its share.  The data-parallel trainer uses exactly that for the block weight-gradient launches (budget = CUs - reserved_cus
 * around those launches only: ops.WGRAD_CU_BUDGET; measured beside a stand-in occupant in DESIGN.md 5); every other launch keeps the whole-chip plan -- a
 * smaller budget makes the one-round launches (the N = 768 projections of MMDiT-B: 249 tiles of 320 x 256) two rounds of smaller tiles.  The reference has
 * no counterpart (DDP leaves the split to the CUDA scheduler, model_trainer.py:224).
 * n: a multiple of 8 in [64, CUs of the device] (the XCD round-robin stays even); per device (hipSetDevice first); takes effect with the next launch --
 * change it only while no captured graph of earlier launches is replayed. */
int mmdit_set_cu_budget(int n);
int mmdit_get_cu_budget(void);
/* Which kernel mmdit_gemm_grouped would launch for these problems (no launch): 64 = register-staged kernel (gemm.hip);
 * otherwise the LDS-DMA kernel (gemm_dma.hip) with tile configuration (value & 15): 0 = 128x128, 1 = 256x128,
 * 2 = 256x256, 3 = 320x256 (lean kernel only), plus 16 if the stream-K decomposition is used, plus 32 for the full-rounds +
 * split-K-tail schedule (the default K decomposition of stream_k launches), plus 128 when the lean hot-path kernel
 * (csrc/gemm_lean.hip: bf16 in / bf16 out, bias / SiLU only; with k-major A: its weight-gradient kernel, fp32 out) takes the launch.
 * + 256 when the launch goes to the 8-phase kernel (csrc/gemm8p.hip: 256x256 tiles on the de-phased main loop, 16x16x32 MFMA).
 * Negative = the MMDIT_ERR_* the launch would return.
 * Lets profilers / benchmarks attribute timings to the exact kernel symbol. */
int mmdit_gemm_plan(const mmdit_gemm_args* args, int count);

/* Per-tensor fp8 (e4m3) quantisation for the inference GEMMs (BASELINE config 5): amax[0] = max(amax[0], max|x|) (caller zeroes
 * it), then q = round_to_e4m3(x * 448 / amax) and scale[0] = amax / 448 (the dequantisation scale mmdit_gemm_args.scale_*).
 * Two launches on the same stream; nothing is read back by the host. */
int mmdit_fp8_amax(const void* x, int x_dtype, int64_t n, float* amax, mmdit_stream_t stream);
int mmdit_fp8_quantize(const void* x, int x_dtype, int64_t n, const float* amax, void* q_fp8, float* scale, mmdit_stream_t stream);
/* Delayed scaling, one pass: quantise with margin * (amax of the previous call at this call site) and collect this call's amax.
 * state: 4 floats {amax ring [3], dequantisation scale (output)}; phase = call counter of the call site (the caller initialises
 * state[phase % 3] with a two-pass mmdit_fp8_amax before the first call). */
int mmdit_fp8_quantize_delayed(const void* x, int x_dtype, int64_t n, float* state, int phase, float margin, void* q_fp8, mmdit_stream_t stream);
/* MX (OCP microscaling) e4m3 quantisation of a row-major operand x[rows][K] (leading dimension ldx, K % 64 == 0):
 * one pass, no state.  Every 32 consecutive K values share the smallest E8M0 scale 2^e with amax / 2^e <= 448 (e = floor(log2 amax) - 8,
 * or one more when the mantissa of amax exceeds 1.75; an all-zero block gets 2^-127); q[rows][K] = saturate_e4m3(x / scale), scales in the layout mmdit_gemm_args.scale_mode 1 reads
 * (allocate (K / 64) * rows_pad128 * 2 + 512 bytes).  Replaces the per-tensor amax / delayed-scaling passes in front of an fp8 GEMM
 * (reference: none -- the reference runs bf16 autocast; BASELINE.json config 5 asks for an fp8 inference path). */
int mmdit_mxfp8_quantize(const void* x, int x_dtype, int rows, int K, int64_t ldx, void* q_fp8, void* scales_e8m0, mmdit_stream_t stream);
/* MX-producing variants of the three kernels whose outputs feed the fp8 GEMMs of a block (inference, "mxfp8" precision): the
 * activation leaves its producer as e4m3 codes + E8M0 block scales (layout and size as for mmdit_gemm_args.scale_mode 1), bit-identical to the bf16 output followed by mmdit_mxfp8_quantize -- no quantise pass in front of the
 * QKV / out-projection / MLP GEMMs.
 *   mmdit_ln_modulate_fwd_mx: adaLN (Norm.py:16-22), optionally with the pending gated residual update of mmdit_ln_modulate_fwd_res
 *     (acc != NULL: bf16 acc, writes x_out); d % 64 == 0.
 *   mmdit_swiglu_fwd_mx: SwiGLU activation of the bf16 pre-activations [g | u] (MLP.py:25-40); hidden % 64 == 0.
 *   mmdit_attn_fwd_mx: flash attention forward (mode 0 of mmdit_attn_fwd) writing Ox / Oc as e4m3 (B, tokens, heads * 64) with scales
 *     in that layout (K = heads * 64); no lse (inference). */
int mmdit_ln_modulate_fwd_mx(const float* x, const void* acc, int acc_dtype, const float* gate, int64_t ld_gate, float* x_out,
                             const float* scale, const float* shift, int64_t ld_mod, int rows, int d, int rows_per_batch,
                             void* q_fp8, void* scales_e8m0, float* mean, float* rstd, mmdit_stream_t stream);
int mmdit_swiglu_fwd_mx(const void* gu, int dtype, int rows, int hidden, void* q_fp8, void* scales_e8m0, mmdit_stream_t stream);
int mmdit_attn_fwd_mx(const void* Q, const void* K, const void* V, int batch, int heads, int S, int n_img, float scale,
                      void* Ox_fp8, void* Oc_fp8, void* scales_x, void* scales_c, mmdit_stream_t stream);

/* dtype conversion of n elements (bf16 shadow copies of the fp32 master weights; the
 * reference gets these from torch.autocast, model_trainer.py:416). n%8==0 not required. */
int mmdit_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, mmdit_stream_t stream);

/* ---------------------------------------------------------------------------
 * adaLN: out = LayerNorm(x; eps=1e-5, no affine) * (1 + scale[b,:]) + shift[b,:]
 * Norm.forward, blocks/Norm.py:16-22.  x fp32 (rows, d) residual stream;
 * scale/shift fp32 with leading dimension ld_mod (slices of the per-block
 * modulation matrix); b = row / rows_per_batch.  Saves mean/rstd per row.
 * bwd: dx = dres + LN-backward(dout * (1+scale));  dscale[b,:] += sum_rows dout*xhat;
 *      dshift[b,:] += sum_rows dout   (atomic fp32 accumulation, caller zero-inits).
 * d%4==0, d<=4096.
 * ------------------------------------------------------------------------- */
int mmdit_ln_modulate_fwd(const float* x, const float* scale, const float* shift, int64_t ld_mod,
                          int rows, int d, int rows_per_batch,
                          void* out, int out_dtype, float* mean, float* rstd, mmdit_stream_t stream);
/* adaLN forward fused with the gated residual update that PRODUCES its input (blocks/Transformer_Block_Dual.py:64-76):
 *   x_out = x + gate[b,:] * acc   (acc = the projection GEMM's output in the activation dtype, gate fp32 (batch, ld_gate));
 *   out = LayerNorm(x_out) * (1 + scale[b,:]) + shift[b,:].
 * The projection GEMM then needs no fp32 gate/residual epilogue.  mmdit_gate_residual_fwd is the update on its own. */
int mmdit_ln_modulate_fwd_res(const float* x, const void* acc, int acc_dtype, const float* gate, int64_t ld_gate, float* x_out,
                              const float* scale, const float* shift, int64_t ld_mod,
                              int rows, int d, int rows_per_batch,
                              void* out, int out_dtype, float* mean, float* rstd, mmdit_stream_t stream);
/* The image and the text stream of a block run the same adaLN kernels on different rows, modulation vectors and rows-per-sample: the
 * *_pair entry points take both problems (same d, same dtypes) in ONE launch -- these 20-45 us kernels pay ~8 us of ramp-up and tail
 * per launch.  Field meaning as in mmdit_ln_modulate_fwd_res / mmdit_ln_modulate_bwd_gated (acc == NULL in BOTH problems: the plain
 * mmdit_ln_modulate_fwd / _bwd arithmetic; dbias may be NULL). */
typedef struct mmdit_ln_fwd_problem {
  const float* x; const void* acc; const float* gate; int64_t ld_gate; float* x_out;
  const float* scale; const float* shift; int64_t ld_mod; int rows, rows_per_batch;
  void* out; float* mean; float* rstd;
} mmdit_ln_fwd_problem;
int mmdit_ln_modulate_fwd_pair(const mmdit_ln_fwd_problem* p0, const mmdit_ln_fwd_problem* p1, int d, int acc_dtype, int out_dtype, mmdit_stream_t stream);
typedef struct mmdit_ln_bwd_problem {
  const void* dout; const float* x; const float* mean; const float* rstd; const float* scale; int64_t ld_mod; const float* dres; int rows, rows_per_batch;
  float* dx; float* dscale; float* dshift; int64_t ld_dmod;
  const void* acc; const float* gate; int64_t ld_gate; void* dacc; float* dgate; int64_t ld_dgate; float* dbias; int64_t ld_dbias;
} mmdit_ln_bwd_problem;
int mmdit_ln_modulate_bwd_pair(const mmdit_ln_bwd_problem* p0, const mmdit_ln_bwd_problem* p1, int d, int dout_dtype, mmdit_stream_t stream);
int mmdit_gate_residual_fwd(const float* x, const void* acc, int acc_dtype, const float* gate, int64_t ld_gate,
                            int rows, int d, int rows_per_batch, float* out, mmdit_stream_t stream);
int mmdit_ln_modulate_bwd(const void* dout, int dout_dtype, const float* x, const float* mean, const float* rstd,
                          const float* scale, int64_t ld_mod, const float* dres,
                          int rows, int d, int rows_per_batch,
                          float* dx, float* dscale, float* dshift, int64_t ld_dmod, mmdit_stream_t stream);
/* The same, fused with the backward of the gated residual update that consumes dx next in the backward order
 * (X_out = acc * gate[b,:] + X_in, blocks/Transformer_Block_Dual.py:64-76; dx = d(X_out)):
 *   dacc = dx * gate[b,:]  (acc's dtype = dout's dtype),  dgate[b,:] += sum_rows dx * acc,
 *   dbias[b,:] += sum_rows dacc  (optional: per-batch partial rows of the producing projection's bias gradient).
 * Replaces ln_modulate_bwd followed by gate_residual_bwd on the same rows. */
int mmdit_ln_modulate_bwd_gated(const void* dout, int dout_dtype, const float* x, const float* mean, const float* rstd,
                                const float* scale, int64_t ld_mod, const float* dres,
                                int rows, int d, int rows_per_batch,
                                float* dx, float* dscale, float* dshift, int64_t ld_dmod,
                                const void* acc, int acc_dtype, const float* gate, int64_t ld_gate,
                                void* dacc, float* dgate, int64_t ld_dgate, float* dbias, int64_t ld_dbias, mmdit_stream_t stream);

/* ---------------------------------------------------------------------------
 * Text pre-norm: out = scalar * RMSNorm_w(x), eps = FLT_EPSILON (nn.RMSNorm(eps=None) on fp32)
 * diff_model.py:164-172, 323-326.  Rows are (b, j) with j in [0,154): rows with j < split use
 * (w1, s1), the others (w2, s2).  out is written as two contiguous halves:
 * out1[(b*split + j), :] and out2[(b*(tokens-split) + j - split), :].
 * bwd accumulates dw1,dw2 (d) and ds1,ds2 (1) atomically (x has no gradient in the reference).
 * ------------------------------------------------------------------------- */
int mmdit_text_rmsnorm_fwd(const void* x, int x_dtype, const float* w1, const float* w2, const float* s1, const float* s2,
                           int batch, int tokens, int split, int d, void* out1, void* out2, int out_dtype, mmdit_stream_t stream);
int mmdit_text_rmsnorm_bwd(const void* dout1, const void* dout2, int dout_dtype, const void* x, int x_dtype,
                           const float* w1, const float* w2, const float* s1, const float* s2,
                           int batch, int tokens, int split, int d,
                           float* dw1, float* dw2, float* ds1, float* ds2, mmdit_stream_t stream);

/* ---------------------------------------------------------------------------
 * Per-head QK RMSNorm + axial 2-D RoPE + head split into the joint [image;text] buffers.
 * Attention.py:130-135 (q/k_norm_*), 178-194 + rotary_embedding.py:36-76,269-288 (RoPE2d),
 * 259-261 (cat).  qkv: (batch*tokens, 3*dim) rows of one stream = [q | k | v], head_dim = 64.
 * Writes bf16 Q,K,V of shape (batch, heads, S_total, 64) at token offset tok0.
 * rope_cos/rope_sin: fp32 (tokens, 64) tables or NULL (text stream: no rotation).
 * bwd: dqkv from dQ,dK,dV (dtype dq_dtype), dwq/dwk (64) accumulated atomically.
 * ------------------------------------------------------------------------- */
int mmdit_qk_norm_rope_fwd(const void* qkv, int qkv_dtype, const float* wq, const float* wk,
                           const float* rope_cos, const float* rope_sin,
                           int batch, int tokens, int heads, int s_total, int tok0,
                           void* Q, void* K, void* V, mmdit_stream_t stream);
int mmdit_qk_norm_rope_bwd(const void* dQ, const void* dK, const void* dV, int dq_dtype,
                           const void* qkv, int qkv_dtype, const float* wq, const float* wk,
                           const float* rope_cos, const float* rope_sin,
                           int batch, int tokens, int heads, int s_total, int tok0,
                           void* dqkv, int dqkv_dtype, float* dwq, float* dwk, mmdit_stream_t stream);
/* The image and the text rows of a block in ONE launch (both write / read the same joint Q, K, V at their own token offset; the text
 * problem has rope_cos = rope_sin = NULL): same arithmetic as two mmdit_qk_norm_rope_fwd / _bwd calls, one ramp-up and tail. */
typedef struct mmdit_qk_problem {
  const void* qkv; const float* wq; const float* wk; const float* rope_cos; const float* rope_sin; int tokens, tok0;
  void* dqkv; float* dwq; float* dwk;      /* backward only */
} mmdit_qk_problem;
int mmdit_qk_norm_rope_fwd_pair(const mmdit_qk_problem* p0, const mmdit_qk_problem* p1, int qkv_dtype, int batch, int heads, int s_total,
                                void* Q, void* K, void* V, mmdit_stream_t stream);
int mmdit_qk_norm_rope_bwd_pair(const mmdit_qk_problem* p0, const mmdit_qk_problem* p1, const void* dQ, const void* dK, const void* dV, int dq_dtype,
                                int qkv_dtype, int dqkv_dtype, int batch, int heads, int s_total, mmdit_stream_t stream);

/* ---------------------------------------------------------------------------
 * Joint softmax attention core, non-causal, head_dim 64, bf16 operands, fp32 accumulate.
 * Replaces flash_attn_func (Attention.py:293) and its CPU twin (Attention.py:277-284).
 * Q,K,V: bf16 (batch, heads, S, 64).  Output is written head-merged and split by stream:
 * Ox (batch, n_img, heads*64), Oc (batch, S-n_img, heads*64), bf16.  lse: fp32 (batch,heads,S).
 * mode 0: flash (fp32 scores, online softmax).  mode 1: reproduces the rounding points of the
 * reference's CPU branch (scores->bf16, *scale->bf16, softmax->bf16, PV->bf16; two passes).
 * bwd: dOx/dOc bf16 (dOc may be NULL = zeros, last block), delta fp32 workspace (batch,heads,S) -- rowsum(dO * O), produced
 * by the dQ pass and consumed by the dK/dV pass --, dQ,dK,dV written (not accumulated) in dq_dtype.
 * ------------------------------------------------------------------------- */
int mmdit_attn_fwd(const void* Q, const void* K, const void* V, int batch, int heads, int S, int n_img,
                   float scale, int mode, void* Ox, void* Oc, float* lse, mmdit_stream_t stream);
/* mmdit_attn_bwd with the backward of the per-head QK RMSNorm + axial RoPE (Attention.py:130-135, 178-194 under autograd) fused into the
 * epilogues of its two kernels: the dQ / dK rows go from the accumulators straight to the gradient of the raw QKV projections
 * dqkv_x (batch * n_img, 3 * heads * 64) / dqkv_c (batch * (S - n_img), 3 * heads * 64) [q | k | v per row, bf16], dV rows are stored into
 * their v part -- no dQ / dK / dV tensors, no separate mmdit_qk_norm_rope_bwd pass.  qkv_x / qkv_c: the saved raw projections (bf16, same
 * geometry); wq_* / wk_*: the norm weights (64); rope_cos / rope_sin (n_img, 64).  dw: (256) fp32, the norm-weight gradients
 * [wq_x | wk_x | wq_c | wk_c], ACCUMULATED (every workgroup adds its LDS-reduced sums with one atomic per feature: zero or hold the
 * running gradient on entry).  n_img % 32 == 0 (else MMDIT_ERR_SHAPE: use the two-pass form). */
int mmdit_attn_bwd_qk(const void* Q, const void* K, const void* V, const void* Ox, const void* Oc, const void* dOx, const void* dOc,
                      const float* lse, float* delta, int batch, int heads, int S, int n_img, float scale,
                      const void* qkv_x, const void* qkv_c, const float* wq_x, const float* wk_x, const float* wq_c, const float* wk_c,
                      const float* rope_cos, const float* rope_sin, void* dqkv_x, void* dqkv_c, float* dw, mmdit_stream_t stream);
int mmdit_attn_bwd(const void* Q, const void* K, const void* V, const void* Ox, const void* Oc,
                   const void* dOx, const void* dOc, const float* lse, float* delta,
                   int batch, int heads, int S, int n_img, float scale,
                   void* dQ, void* dK, void* dV, int dq_dtype, mmdit_stream_t stream);

/* ---------------------------------------------------------------------------
 * MLP activations.  swiglu: h = silu(g) * u with [g | u] = gu (rows, 2*hidden)  (xformers SwiGLU
 * eager semantics behind MLP.py:19,32).  gelu: h = gelu_erf(u) (MLP.py:21-23,36-40).
 * bwd also accumulates the column sums of the produced gradient into dbias (bias grad of the
 * preceding Linear), atomically; dbias may be NULL.
 * ------------------------------------------------------------------------- */
int mmdit_swiglu_fwd(const void* gu, void* h, int dtype, int rows, int hidden, mmdit_stream_t stream);
int mmdit_swiglu_bwd(const void* dh, const void* gu, void* dgu, int dtype, int rows, int hidden, float* dbias, mmdit_stream_t stream);
int mmdit_gelu_fwd(const void* u, void* h, int dtype, int rows, int hidden, mmdit_stream_t stream);
int mmdit_gelu_bwd(const void* dh, const void* u, void* du, int dtype, int rows, int hidden, float* dbias, mmdit_stream_t stream);
/* mmdit_swiglu_bwd / mmdit_gelu_bwd (gelu != 0) of two problems of the same hidden width -- the image and the text MLP of a block -- in one launch */
typedef struct mmdit_mlp_bwd_problem { const void* dh; const void* gu; void* dgu; int rows; float* dbias; } mmdit_mlp_bwd_problem;
int mmdit_mlp_act_bwd_pair(const mmdit_mlp_bwd_problem* p0, const mmdit_mlp_bwd_problem* p1, int dtype, int hidden, int gelu, mmdit_stream_t stream);
/* d(pre) = dy * silu'(pre): y_proj's SiLU (Transformer_Block_Dual.py:25-28). dbias accumulates column sums. */
int mmdit_silu_bwd(const void* dy, int dy_dtype, const float* pre, void* dpre, int dpre_dtype, int rows, int cols, float* dbias, int rows_per_bias,
                   mmdit_stream_t stream);   /* rows_per_bias > 0: dbias is (rows / rows_per_bias, cols), one row per group of rows (stacked blocks) */

/* Backward of  Y = X + gate[b,:] * acc  (Transformer_Block_Dual.py:64-66,70-76):
 * dacc = dy * gate[b,:] (dtype dacc_dtype); dgate[b,:] += sum_rows dy*acc; dbias[b*ld_dbias + :] += sum_rows dacc (optional;
 * ld_dbias = 0: one shared row, ld_dbias > 0: per-batch partial rows that the caller column-sums -- far less atomic contention). */
int mmdit_gate_residual_bwd(const float* dy, const void* acc, int acc_dtype, const float* gate, int64_t ld_gate,
                            int rows, int d, int rows_per_batch, void* dacc, int dacc_dtype,
                            float* dgate, int64_t ld_dgate, float* dbias, int64_t ld_dbias, mmdit_stream_t stream);

/* Rectified-flow loss of the training step (model_trainer.py:429-446): label = eps - x0 (rounded to bf16 when x0 / eps are bf16, as
 * torch's bf16 subtraction rounds it), loss[0] = coef * sum_i (v[i] - label[i])^2 with coef = 1 / (n * accumulation_steps) from the
 * caller, and -- for the backward -- dv[i] = 2 * coef * (v[i] - label[i]) (NULL: not written).  Two launches: <= 256 workgroups write one
 * partial sum each into partials[256] (fixed order inside a workgroup), one workgroup adds them in index order: bit-reproducible, no
 * atomics, no zero-initialised workspace (torch's multi-block mean() clears a semaphore with hipMemsetAsync, which a hipGraph replay on
 * ROCm 7 does not keep in stream order: DESIGN.md 5).  n % 8 == 0. */
int mmdit_flow_loss(const float* v, const void* x0, const void* eps, int in_dtype, int64_t n, float coef, float* dv, float* partials,
                    float* loss, mmdit_stream_t stream);

/* Column sums: out[c] += sum_r x[r,c]  (bias gradients). */
int mmdit_colsum(const void* x, int dtype, int rows, int cols, int64_t ld, float* out, mmdit_stream_t stream);

/* ---------------------------------------------------------------------------
 * Patchify / unpatchify layout kernels (ImagePositionalEncoding.py:114-116,181-183 conv-as-GEMM gather;
 * patchify.py:41-72).  Image is NCHW (batch, ch, H, W), H and W even; tokens are row-major over
 * (H/2, W/2) and the patch vector order is (ch, ph, pw).
 * ------------------------------------------------------------------------- */
int mmdit_patchify(const void* img, int img_dtype, int batch, int ch, int H, int W, void* tokens, int tok_dtype, mmdit_stream_t stream);
int mmdit_unpatchify(const void* tokens, int tok_dtype, int batch, int ch, int H, int W, void* img, int img_dtype, mmdit_stream_t stream);

/* Sinusoidal timestep embedding, PositionalEncoding.py:15-30 with diff_model.py:306's time_scale:
 * tau = t*time_scale; e_i = tau / 10000^(2i/dim), i=0..dim-1; out = [sin(e_0),sin(e_2),..,cos(e_1),cos(e_3),..].
 * denom: the fp32 table 10000^(2i/dim), i=0..dim-1 (the reference precomputes it too, PositionalEncoding.py:15-16).
 * bwd: dtime_scale += sum dout * d(out)/d(time_scale). */
int mmdit_time_embed_fwd(const float* t, const float* time_scale, const float* denom, int batch, int dim, void* out, int out_dtype, mmdit_stream_t stream);
int mmdit_time_embed_bwd(const void* dout, int dout_dtype, const float* t, const float* time_scale, const float* denom, int batch, int dim, float* dtime_scale, mmdit_stream_t stream);

/* ---------------------------------------------------------------------------
 * FLUX VAE (diffusers AutoencoderKL) building blocks.  Call sites in the reference:
 * helpers/VAE_T5_CLIP.py:176-182 and helpers/VAE_T5_CLIP_inference.py:25-43 (encode), models/diff_model.py:467-477 (decode).
 * Activations are NHWC bf16 with channel counts that are multiples of 8; a 3x3 convolution is
 * mmdit_vae_im2col3x3 + mmdit_gemm with the weight re-laid as [Cout][kh][kw][Cin].
 * ------------------------------------------------------------------------- */
/* NCHW (fp32 | bf16) -> NHWC bf16, channels zero-padded to C_padded; value = (x + shift) * scale
 * (decode entry: (z - shift_factor) / scaling_factor, diff_model.py:467). */
int mmdit_vae_nchw_to_nhwc(const void* src, int src_dtype, int batch, int C, int H, int W, int C_padded, float scale, float shift,
                           void* dst_bf16, mmdit_stream_t stream);
/* NHWC fp32 (row pitch ld >= C) -> NCHW fp32, clamped to [lo, hi] (.sample.clamp(-1, 1), diff_model.py:467-477). */
int mmdit_vae_nhwc_to_nchw(const float* src, int batch, int C, int H, int W, int ld, float lo, float hi, float* dst, mmdit_stream_t stream);
/* im2col of a 3x3 convolution: dst[(b,yo,xo), (kh*3+kw)*C + c].  mode 0: stride 1, padding 1 (nn.Conv2d(…, 3, padding=1));
 * mode 1: stride 2 after padding (0,1,0,1) (diffusers Downsample2D, padding=0); mode 2: nearest x2 upsampling then
 * stride 1, padding 1 (diffusers Upsample2D) -- output (2H, 2W), the upsampled tensor is never materialised. */
int mmdit_vae_im2col3x3(const void* src_bf16, int batch, int H, int W, int C, int mode, void* dst_bf16, mmdit_stream_t stream);
/* GroupNorm(groups, C, eps, affine) over NHWC x (fp32 | bf16), optional SiLU, bf16 out (ResnetBlock2D norm1/norm2,
 * Attention.group_norm, conv_norm_out).  sums_zeroed: batch*groups*2 fp32 scratch, zero on entry.
 * pad != 0: y is a zero-bordered (batch, H+2, W+2, C) tensor (borders zeroed by the caller) whose interior is written --
 * the A operand of the implicit-GEMM convolution (mmdit_gemm_args.conv_mode). */
int mmdit_vae_groupnorm(const void* x, int x_dtype, const float* gamma, const float* beta, int batch, int H, int W, int C, int groups, float eps, int silu,
                        float* sums_zeroed, void* y_bf16, int pad, mmdit_stream_t stream);
/* NHWC (fp32 | bf16) -> interior of a zero-bordered bf16 (batch, uH+2, uW+2, C); upsample = 1: nearest x2 first (Upsample2D). */
int mmdit_vae_pad_cast(const void* x, int x_dtype, int batch, int H, int W, int C, int upsample, void* y_bf16, mmdit_stream_t stream);
/* y = softmax(scale * x) over the first `cols` entries of each row (pitch ld, padding written as 0), fp32 -> bf16
 * (single-head mid-block attention, attention_processor.py). */
int mmdit_vae_softmax_rows(const float* x, int rows, int cols, int ld, float scale, void* y_bf16, mmdit_stream_t stream);

/* ---------------------------------------------------------------------------
 * Optimizer step: GradScaler.unscale_ + torch.nn.utils.clip_grad_norm_ + AdamW.step of the reference
 * trainer (model_trainer.py:260-269 builds AdamW(lr, eps=1e-8, weight_decay=0.01, betas=(0.9, 0.999)); 463-503 runs
 * unscale_ / clip / step / update) as three launches over the whole parameter list.  All tensors fp32.
 * The parameter list is a DEVICE array of mmdit_adamw_tensor and a DEVICE chunk map: chunk c covers elements
 * [chunk_off[c], min(numel, chunk_off[c] + MMDIT_ADAMW_CHUNK)) of tensor chunk_tensor[c] (one workgroup per chunk).
 * ------------------------------------------------------------------------- */
#define MMDIT_ADAMW_CHUNK 65536
typedef struct mmdit_adamw_tensor {
  float* param;
  const float* grad;      /* as produced by backward: still multiplied by the loss scale */
  float* exp_avg;
  float* exp_avg_sq;
  int64_t numel;
  void* shadow_bf16;      /* optional: the bf16 GEMM-operand copy of this parameter, rewritten by mmdit_adamw_step together with
                           * the fp32 master (saves the separate fp32 -> bf16 refresh pass); NULL = none */
} mmdit_adamw_tensor;
/* partials[c] = sum of grad^2 over chunk c (no atomics: deterministic). */
int mmdit_grad_sumsq(const mmdit_adamw_tensor* tensors, const int* chunk_tensor, const int64_t* chunk_off, int n_chunks, float* partials, mmdit_stream_t stream);
/* out3[0] = gradient multiplier (1/loss_scale) * min(1, max_norm / (norm + 1e-6)), norm = sqrt(sum partials) / loss_scale
 * (max_norm <= 0: no clipping); out3[1] = found_inf (1.0 if the norm is not finite, else 0.0); out3[2] = norm.
 * loss_scale: device scalar of the GradScaler, or NULL for 1. */
int mmdit_clip_coef(const float* partials, int n_chunks, const float* loss_scale, float max_norm, float* out3, mmdit_stream_t stream);
/* AdamW update of every chunk with grad * coef_found[0] (coef_found = out3 above, or NULL for multiplier 1); the step is
 * skipped entirely when coef_found[1] != 0.  step_count: device scalar holding the number of steps taken so far (the
 * caller adds 1 - found_inf afterwards, as torch's fused AdamW does with its per-parameter `step` tensors):
 *   p *= 1 - lr*wd;  m += (1-b1)(g - m);  v = b2 v + (1-b2) g^2;  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps),  t = step_count+1. */
int mmdit_adamw_step(const mmdit_adamw_tensor* tensors, const int* chunk_tensor, const int64_t* chunk_off, int n_chunks, const float* coef_found,
                     const float* step_count, double lr, double beta1, double beta2, double eps, double weight_decay, mmdit_stream_t stream);
/* The same with the learning rate read from DEVICE memory (one double): a launch captured into a hipGraph must not bake the
 * scheduler's current value in (model_trainer.py:25-41, 496: the rate changes every step during warm-up / cosine decay). */
int mmdit_adamw_step_dlr(const mmdit_adamw_tensor* tensors, const int* chunk_tensor, const int64_t* chunk_off, int n_chunks, const float* coef_found,
                         const float* step_count, const double* lr_dev, double beta1, double beta2, double eps, double weight_decay, mmdit_stream_t stream);

/* fp32 master weights -> bf16 GEMM operand copies for a whole parameter list in one launch (the reference gets its bf16 copies
 * from torch.autocast's weight cache, model_trainer.py:416).  Same chunk map as above: chunk c covers elements
 * [chunk_off[c], +MMDIT_ADAMW_CHUNK) of tensors[chunk_tensor[c]]; tensors / maps are DEVICE arrays. */
typedef struct mmdit_cast_tensor {
  const float* src;
  void* dst;              /* bf16 */
  int64_t numel;
} mmdit_cast_tensor;
int mmdit_cast_multi(const mmdit_cast_tensor* tensors, const int* chunk_tensor, const int64_t* chunk_off, int n_chunks, mmdit_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MMDIT_HIP_H */
