// MFMA GEMM for gfx950:  C[M,N] = epilogue(A[M,K] * B[N,K]^T), fp32 accumulate.
//
// v1 structure: 128x128x64 block tile, 256 threads = 4 waves (2x2), each wave a 64x64
// sub-tile = 2x2 v_mfma_f32_32x32x16_bf16 accumulators.  Operands are staged
// global -> registers -> LDS (double buffered, one barrier per K-tile).
//   row-major operand tile  : LDS [128 rows][64 k] bf16, pitch 144 B (conflict-free ds_read_b128)
//   k-major operand tile    : LDS [64 k][128 rows] bf16, pitch 320 B, fragments via ds_read_b64_tr_b16
// SPLIT precision (parity mode): fp32 operands are split exactly into three bf16 pieces
// a = a0 + a1 + a2 (8+8+8 significand bits) while staging and the product is formed from the six
// MFMA passes with weight >= 2^-16 (a0b0 + a0b1 + a1b0 + a0b2 + a1b1 + a2b0): fp32-exact products,
// fp32 accumulation.  A 2-term split (1e-5 relative) is not enough: the bf16 rounding points of the
// attention core amplify an upstream error d to ~sqrt(d * 2^-8).
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int RM_PITCH = 144;            // bytes per row of a row-major tile (64 bf16 + 16 B pad)
constexpr int KM_PITCH = 320;            // bytes per k-row of a k-major tile (128 bf16 + 64 B pad)
constexpr int TILE_BYTES = 20480;        // max(128*144, 64*320)
constexpr int CHUNKS = 4;                // 16-byte (8 x bf16) chunks per thread per operand tile

struct GemmParams {
  const void* A; const void* B; void* C; void* aux;
  const float* bias; const float* gate; const float* residual;
  int64_t lda, ldb, ldc, ld_gate, ld_res, ld_aux;
  int M, N, K;
  int rows_per_batch, act, accumulate;
};

// ---- staging registers -----------------------------------------------------------------------
template <typename T, bool SPLIT> struct Stage;
template <> struct Stage<bf16_t, false> { u32x4 v[CHUNKS]; };
template <> struct Stage<float, false> { f32x4 v[CHUNKS][2]; };
template <> struct Stage<float, true> { f32x4 v[CHUNKS][2]; };

template <bool KM>
__device__ __forceinline__ void chunk_coords(int c, int& r, int& kc) {
  // returns tile row index r (0..127) and k index kc (0..63, multiple of 8 for row-major; any for k-major)
  if (!KM) { r = c >> 3; kc = (c & 7) * 8; }        // 8 chunks of 8 k per row
  else { kc = c >> 4; r = (c & 15) * 8; }           // 16 chunks of 8 rows per k-row
}

template <typename T, bool KM, bool SPLIT>
__device__ __forceinline__ void load_tile(Stage<T, SPLIT>& st, const T* base, int64_t ld, int row0, int k0, int rows, int K, int tid) {
#pragma unroll
  for (int i = 0; i < CHUNKS; i++) {
    int c = tid + 256 * i, r, kc;
    chunk_coords<KM>(c, r, kc);
    bool ok = (row0 + r < rows) && (k0 + kc < K);
    const T* p = KM ? base + (int64_t)(k0 + kc) * ld + row0 + r : base + (int64_t)(row0 + r) * ld + k0 + kc;
    if constexpr (sizeof(T) == 2) {
      st.v[i] = ok ? *(const u32x4*)p : (u32x4){0, 0, 0, 0};
    } else {
      st.v[i][0] = ok ? *(const f32x4*)p : (f32x4){0, 0, 0, 0};
      st.v[i][1] = ok ? *(const f32x4*)(p + 4) : (f32x4){0, 0, 0, 0};
    }
  }
}

template <typename T, bool KM, bool SPLIT>
__device__ __forceinline__ void store_tile(const Stage<T, SPLIT>& st, char* hi, char* mid, char* lo, int tid) {
#pragma unroll
  for (int i = 0; i < CHUNKS; i++) {
    int c = tid + 256 * i, r, kc;
    chunk_coords<KM>(c, r, kc);
    int off = KM ? kc * KM_PITCH + r * 2 : r * RM_PITCH + kc * 2;
    if constexpr (sizeof(T) == 2) {
      *LDS_PTR(u32x4, hi + off) = st.v[i];
    } else {
      float f[8] = {st.v[i][0][0], st.v[i][0][1], st.v[i][0][2], st.v[i][0][3], st.v[i][1][0], st.v[i][1][1], st.v[i][1][2], st.v[i][1][3]};
      uint32_t h[4], md[4], l[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        bf16_t h0 = f2bf(f[2 * j]), h1 = f2bf(f[2 * j + 1]);
        h[j] = (uint32_t)h0 | ((uint32_t)h1 << 16);
        if constexpr (SPLIT) {
          const float r0 = f[2 * j] - bf2f(h0), r1 = f[2 * j + 1] - bf2f(h1);     // exact
          const bf16_t m0 = f2bf(r0), m1 = f2bf(r1);
          md[j] = (uint32_t)m0 | ((uint32_t)m1 << 16);
          l[j] = pack_bf2(r0 - bf2f(m0), r1 - bf2f(m1));                           // exact residual, 8 bits left
        }
      }
      *LDS_PTR(u32x4, hi + off) = (u32x4){h[0], h[1], h[2], h[3]};
      if constexpr (SPLIT) {
        *LDS_PTR(u32x4, mid + off) = (u32x4){md[0], md[1], md[2], md[3]};
        *LDS_PTR(u32x4, lo + off) = (u32x4){l[0], l[1], l[2], l[3]};
      }
    }
  }
}

// fragment for a 32-row block starting at tile row r0, k-step ks (16 k): lane holds row r0+(l&31), k = ks*16 + (l>>5)*8 .. +7
template <bool KM>
__device__ __forceinline__ bf16x8 load_frag(const char* tile, int r0, int ks, int lane) {
  if (!KM) {
    const char* p = tile + (r0 + (lane & 31)) * RM_PITCH + (ks * 16 + (lane >> 5) * 8) * 2;
    return *LDS_PTR(const bf16x8, p);
  } else {
    int kr = ks * 16 + (lane >> 5) * 8 + ((lane & 15) >> 2);
    int col = r0 + 16 * ((lane >> 4) & 1) + (lane & 3) * 4;
    const char* p = tile + kr * KM_PITCH + col * 2;
    s16x4 lo = lds_tr16(p), hi = lds_tr16(p + 4 * KM_PITCH);
    s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
  }
}

template <typename TA, typename TB, bool A_KM, bool B_KM, bool SPLIT, typename TC, typename TAUX>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
  constexpr int NT = SPLIT ? 6 : 2;                  // tiles per stage: A0, B0, (A1, B1, A2, B2)
  constexpr int NSTAGE = SPLIT ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const TA* A = (const TA*)p.A;
  const TB* B = (const TB*)p.B;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  Stage<TA, SPLIT> sa;
  Stage<TB, SPLIT> sb;
  const int nk = (p.K + BK - 1) / BK;

  auto tile_ptr = [&](int stage, int which) { return smem + (stage * NT + which) * TILE_BYTES; };

  load_tile<TA, A_KM, SPLIT>(sa, A, p.lda, m0, 0, p.M, p.K, tid);
  load_tile<TB, B_KM, SPLIT>(sb, B, p.ldb, n0, 0, p.N, p.K, tid);
  store_tile<TA, A_KM, SPLIT>(sa, tile_ptr(0, 0), tile_ptr(0, 2), tile_ptr(0, 4), tid);
  store_tile<TB, B_KM, SPLIT>(sb, tile_ptr(0, 1), tile_ptr(0, 3), tile_ptr(0, 5), tid);
  __syncthreads();

  int cur = 0;
  for (int kt = 0; kt < nk; kt++) {
    const bool more = kt + 1 < nk;
    if (more) {
      load_tile<TA, A_KM, SPLIT>(sa, A, p.lda, m0, (kt + 1) * BK, p.M, p.K, tid);
      load_tile<TB, B_KM, SPLIT>(sb, B, p.ldb, n0, (kt + 1) * BK, p.N, p.K, tid);
    }
    const char* ta = tile_ptr(cur, 0);
    const char* tb = tile_ptr(cur, 1);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ks++) {
      bf16x8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; i++) a[i] = load_frag<A_KM>(ta, wm * 64 + i * 32, ks, lane);
#pragma unroll
      for (int j = 0; j < 2; j++) b[j] = load_frag<B_KM>(tb, wn * 64 + j * 32, ks, lane);
      if constexpr (SPLIT) {
        bf16x8 a1[2], b1[2], a2[2], b2[2];
#pragma unroll
        for (int i = 0; i < 2; i++) { a1[i] = load_frag<A_KM>(tile_ptr(cur, 2), wm * 64 + i * 32, ks, lane); a2[i] = load_frag<A_KM>(tile_ptr(cur, 4), wm * 64 + i * 32, ks, lane); }
#pragma unroll
        for (int j = 0; j < 2; j++) { b1[j] = load_frag<B_KM>(tile_ptr(cur, 3), wn * 64 + j * 32, ks, lane); b2[j] = load_frag<B_KM>(tile_ptr(cur, 5), wn * 64 + j * 32, ks, lane); }
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 2; j++) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[i], b[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b1[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b2[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b1[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
          }
      } else {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    if (more) {
      if constexpr (NSTAGE == 1) __syncthreads();
      const int nxt = NSTAGE == 1 ? 0 : cur ^ 1;
      store_tile<TA, A_KM, SPLIT>(sa, tile_ptr(nxt, 0), tile_ptr(nxt, 2), tile_ptr(nxt, 4), tid);
      store_tile<TB, B_KM, SPLIT>(sb, tile_ptr(nxt, 1), tile_ptr(nxt, 3), tile_ptr(nxt, 5), tid);
      __syncthreads();
      cur = nxt;
    }
  }

  // ---- epilogue ---------------------------------------------------------------------------------
  TC* C = (TC*)p.C;
  TAUX* AUX = (TAUX*)p.aux;
#pragma unroll
  for (int i = 0; i < 2; i++) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int col = n0 + wn * 64 + j * 32 + (lane & 31);
      if (col >= p.N) continue;
      const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= p.M) continue;
        float v = acc[i][j][r] + bv;
        if (AUX) io<TAUX>::st(AUX + (int64_t)row * p.ld_aux + col, v);
        if (p.act == MMDIT_ACT_SILU) v = silu_f(v);
        if (p.residual) {
          float g = p.gate ? p.gate[(int64_t)(row / p.rows_per_batch) * p.ld_gate + col] : 1.f;
          v = p.residual[(int64_t)row * p.ld_res + col] + g * v;
        }
        TC* cp = C + (int64_t)row * p.ldc + col;
        if (p.accumulate) v += io<TC>::ld(cp);
        io<TC>::st(cp, v);
      }
    }
  }
}

template <typename TA, typename TB, bool A_KM, bool B_KM, bool SPLIT, typename TC, typename TAUX>
int launch(const GemmParams& p, hipStream_t s) {
  constexpr int smem = (SPLIT ? 6 : 4) * TILE_BYTES;  // 2 stages x 2 tiles, or 1 stage x 6 tiles
  auto k = gemm_kernel<TA, TB, A_KM, B_KM, SPLIT, TC, TAUX>;
  static bool attr_done = false;  // idempotent; a benign race only repeats the call
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM);
  hipLaunchKernelGGL(k, grid, dim3(256), smem, s, p);
  return mmdit_launch_status();
}

template <typename TA, typename TB, bool SPLIT, typename TC, typename TAUX>
int dispatch_layout(const mmdit_gemm_args* a, const GemmParams& p, hipStream_t s) {
  if (!a->a_kmajor && !a->b_kmajor) return launch<TA, TB, false, false, SPLIT, TC, TAUX>(p, s);
  if (!a->a_kmajor && a->b_kmajor) return launch<TA, TB, false, true, SPLIT, TC, TAUX>(p, s);
  if (a->a_kmajor && a->b_kmajor) return launch<TA, TB, true, true, SPLIT, TC, TAUX>(p, s);
  return MMDIT_ERR_DTYPE;  // (k-major A, row-major B) is not used on the path
}

template <typename TA, typename TB, bool SPLIT>
int dispatch_out(const mmdit_gemm_args* a, const GemmParams& p, hipStream_t s) {
  const int aux_dt = a->aux ? a->aux_dtype : a->c_dtype;
  if (a->c_dtype == MMDIT_F32 && aux_dt == MMDIT_F32) return dispatch_layout<TA, TB, SPLIT, float, float>(a, p, s);
  if (a->c_dtype == MMDIT_F32 && aux_dt == MMDIT_BF16) return dispatch_layout<TA, TB, SPLIT, float, bf16_t>(a, p, s);
  if (a->c_dtype == MMDIT_BF16 && aux_dt == MMDIT_BF16) return dispatch_layout<TA, TB, SPLIT, bf16_t, bf16_t>(a, p, s);
  if (a->c_dtype == MMDIT_BF16 && aux_dt == MMDIT_F32) return dispatch_layout<TA, TB, SPLIT, bf16_t, float>(a, p, s);
  return MMDIT_ERR_DTYPE;
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" int mmdit_gemm(const mmdit_gemm_args* a, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(a && a->A && a->B && a->C);
  MMDIT_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0);
  const int esa = a->a_dtype == MMDIT_F32 ? 4 : 2, esb = a->b_dtype == MMDIT_F32 ? 4 : 2;
  MMDIT_CHECK_ARG(aligned16(a->A) && aligned16(a->B));
  MMDIT_CHECK_ARG((a->lda * esa) % 16 == 0 && (a->ldb * esb) % 16 == 0);
  if (a->a_kmajor) { MMDIT_CHECK_ARG(a->M % 8 == 0 && a->lda >= a->M); } else { MMDIT_CHECK_ARG(a->K % 8 == 0 && a->lda >= a->K); }
  if (a->b_kmajor) { MMDIT_CHECK_ARG(a->N % 8 == 0 && a->ldb >= a->N); } else { MMDIT_CHECK_ARG(a->K % 8 == 0 && a->ldb >= a->K); }
  MMDIT_CHECK_ARG(a->ldc >= a->N);
  if (a->gate) MMDIT_CHECK_ARG(a->residual && a->rows_per_batch > 0);
  if (a->accumulate) MMDIT_CHECK_ARG(a->c_dtype == MMDIT_F32);
  GemmParams p;
  p.A = a->A; p.B = a->B; p.C = a->C; p.aux = a->aux;
  p.bias = a->bias; p.gate = a->gate; p.residual = a->residual;
  p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc; p.ld_gate = a->ld_gate; p.ld_res = a->ld_res; p.ld_aux = a->ld_aux;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.rows_per_batch = a->rows_per_batch > 0 ? a->rows_per_batch : 1; p.act = a->act; p.accumulate = a->accumulate;
  hipStream_t s = (hipStream_t)stream;
  if (a->precision == MMDIT_PREC_BF16 && a->a_dtype == MMDIT_BF16 && a->b_dtype == MMDIT_BF16) return dispatch_out<bf16_t, bf16_t, false>(a, p, s);
  if (a->precision == MMDIT_PREC_SPLIT && a->a_dtype == MMDIT_F32 && a->b_dtype == MMDIT_F32) return dispatch_out<float, float, true>(a, p, s);
  return MMDIT_ERR_DTYPE;
}
