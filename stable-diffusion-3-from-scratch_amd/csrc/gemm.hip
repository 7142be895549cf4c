// MFMA GEMM for gfx950:  C[M,N] = epilogue(A[M,K] * B[N,K]^T), fp32 accumulate, grouped launches.
//
// Structure: 128x128x64 block tile, 256 threads = 4 waves (2x2), each wave a 64x64 sub-tile =
// 2x2 v_mfma_f32_32x32x16_bf16 accumulators.  Operands are staged global -> registers -> LDS
// (double buffered, one barrier per K-tile).
//   row-major operand tile : LDS [128 rows][64 k] bf16, pitch 144 B (conflict-free ds_read_b128)
//   k-major operand tile   : LDS [64 k][128 rows] bf16, pitch 320 B, fragments via ds_read_b64_tr_b16
// The MFMA is issued as D^T = B_frag * A_frag so that each lane ends up owning ONE output row and
// groups of 4 consecutive output columns: the epilogue (bias, SiLU, gate*acc+residual, aux copy,
// accumulate) then runs on 8/16-byte vectors instead of scalars.
// Grouped launch: up to MAXG independent problems of the same variant share one grid (image + text
// stream of a block, or all weight-gradient GEMMs of a block) so that small problems still fill the
// 256 CUs; the linear tile id is remapped so that each XCD (private L2) gets a contiguous tile range.
// SPLIT precision (parity mode): fp32 operands are split exactly into three bf16 pieces
// a = a0 + a1 + a2 (8+8+8 significand bits) while staging and the product is formed from the six
// MFMA passes with weight >= 2^-16 (a0b0 + a0b1 + a1b0 + a0b2 + a1b1 + a2b0): fp32-exact products,
// fp32 accumulation.  A 2-term split (1e-5 relative) is not enough: the bf16 rounding points of the
// attention core amplify an upstream error d to ~sqrt(d * 2^-8).
#include "gemm_common.h"
#include <stdlib.h>

using namespace gemm;

namespace {

constexpr int BM = 128, BN = 128;
constexpr int RM_PITCH = 144;            // bytes per row of a row-major tile (64 bf16 + 16 B pad)
constexpr int KM_PITCH = 320;            // bytes per k-row of a k-major tile (128 bf16 + 64 B pad)
constexpr int TILE_BYTES = 20480;        // max(128*144, 64*320)
constexpr int CHUNKS = 4;                // 16-byte (8 x bf16) chunks per thread per operand tile

// ---- staging registers -----------------------------------------------------------------------
template <typename T> struct Stage;
template <> struct Stage<bf16_t> { u32x4 v[CHUNKS]; };
template <> struct Stage<float> { f32x4 v[CHUNKS][2]; };

template <bool KM>
__device__ __forceinline__ void chunk_coords(int c, int& r, int& kc) {
  if (!KM) { r = c >> 3; kc = (c & 7) * 8; }        // 8 chunks of 8 k per row
  else { kc = c >> 4; r = (c & 15) * 8; }           // 16 chunks of 8 rows per k-row
}

template <typename T, bool KM>
__device__ __forceinline__ void load_tile(Stage<T>& st, const T* base, int64_t ld, int row0, int k0, int rows, int K, int tid) {
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
__device__ __forceinline__ void store_tile(const Stage<T>& st, char* hi, char* mid, char* lo, int tid) {
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
__global__ __launch_bounds__(256) void gemm_kernel(GroupParams gp) {
  constexpr int NT = SPLIT ? 6 : 2;                  // tiles per stage: A0, B0, (A1, B1, A2, B2)
  constexpr int NSTAGE = SPLIT ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  int tm, tn, sk;
  const Problem& p = locate_tile(gp, blockIdx.x, tm, tn, sk);   // split_k is always 1 on this path
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const TA* A = (const TA*)p.A;
  const TB* B = (const TB*)p.B;
  const int M = p.M, N = p.N, K = p.K;
  const int64_t lda = p.lda, ldb = p.ldb;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  Stage<TA> sa;
  Stage<TB> sb;
  const int nk = (K + BK - 1) / BK;
  auto tile_ptr = [&](int stage, int which) { return smem + (stage * NT + which) * TILE_BYTES; };

  load_tile<TA, A_KM>(sa, A, lda, m0, 0, M, K, tid);
  load_tile<TB, B_KM>(sb, B, ldb, n0, 0, N, K, tid);
  store_tile<TA, A_KM, SPLIT>(sa, tile_ptr(0, 0), tile_ptr(0, 2), tile_ptr(0, 4), tid);
  store_tile<TB, B_KM, SPLIT>(sb, tile_ptr(0, 1), tile_ptr(0, 3), tile_ptr(0, 5), tid);
  __syncthreads();

  int cur = 0;
  for (int kt = 0; kt < nk; kt++) {
    const bool more = kt + 1 < nk;
    if (more) {
      load_tile<TA, A_KM>(sa, A, lda, m0, (kt + 1) * BK, M, K, tid);
      load_tile<TB, B_KM>(sb, B, ldb, n0, (kt + 1) * BK, N, K, tid);
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
      // transposed product: acc[i][j] = (B_j A_i^T) -> rows = n, columns (lanes) = m
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
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a2[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1[j], a1[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b2[j], a[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a1[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1[j], a[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
          }
      } else {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
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

  __syncthreads();   // every wave is done with the operand tiles: their LDS is reused by the epilogue
  epilogue<TC, TAUX>(acc, p, gp, m0, n0, wm, wn, lane, sk, smem + wave * EP_WAVE_BYTES);
}

template <typename TA, typename TB, bool A_KM, bool B_KM, bool SPLIT, typename TC, typename TAUX>
int launch(const GroupParams& gp, hipStream_t s) {
  constexpr int smem = (SPLIT ? 6 : 4) * TILE_BYTES;  // 2 stages x 2 tiles, or 1 stage x 6 tiles
  auto k = gemm_kernel<TA, TB, A_KM, B_KM, SPLIT, TC, TAUX>;
  static unsigned long long attr_done = 0;  // one bit per device; idempotent, a benign race only repeats the call
  if (!mmdit_device_once(attr_done)) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    mmdit_device_mark(attr_done);
  }
  hipLaunchKernelGGL(k, dim3(gp.total_tiles), dim3(256), smem, s, gp);
  return mmdit_launch_status();
}

template <typename TA, typename TB, bool SPLIT, typename TC, typename TAUX>
int dispatch_layout(const mmdit_gemm_args* a, const GroupParams& p, hipStream_t s) {
  if (!a->a_kmajor && !a->b_kmajor) return launch<TA, TB, false, false, SPLIT, TC, TAUX>(p, s);
  if (!a->a_kmajor && a->b_kmajor) return launch<TA, TB, false, true, SPLIT, TC, TAUX>(p, s);
  if (a->a_kmajor && a->b_kmajor) return launch<TA, TB, true, true, SPLIT, TC, TAUX>(p, s);
  return MMDIT_ERR_DTYPE;  // (k-major A, row-major B) is not used on the path
}

template <typename TA, typename TB, bool SPLIT>
int dispatch_out(const mmdit_gemm_args* a, int aux_dt, const GroupParams& p, hipStream_t s) {
  if (a->c_dtype == MMDIT_F32 && aux_dt == MMDIT_F32) return dispatch_layout<TA, TB, SPLIT, float, float>(a, p, s);
  if (a->c_dtype == MMDIT_F32 && aux_dt == MMDIT_BF16) return dispatch_layout<TA, TB, SPLIT, float, bf16_t>(a, p, s);
  if (a->c_dtype == MMDIT_BF16 && aux_dt == MMDIT_BF16) return dispatch_layout<TA, TB, SPLIT, bf16_t, bf16_t>(a, p, s);
  if (a->c_dtype == MMDIT_BF16 && aux_dt == MMDIT_F32) return dispatch_layout<TA, TB, SPLIT, bf16_t, float>(a, p, s);
  return MMDIT_ERR_DTYPE;
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

int check_problem(const mmdit_gemm_args* a) {
  MMDIT_CHECK_ARG(a->A && a->B && a->C);
  MMDIT_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0 && a->N % 4 == 0);
  const int esa = a->a_dtype == MMDIT_F32 ? 4 : a->a_dtype == MMDIT_FP8 ? 1 : 2, esb = a->b_dtype == MMDIT_F32 ? 4 : a->b_dtype == MMDIT_FP8 ? 1 : 2, esc = a->c_dtype == MMDIT_F32 ? 4 : 2;
  MMDIT_CHECK_ARG(a->c_dtype == MMDIT_F32 || a->c_dtype == MMDIT_BF16 || (a->c_dtype == MMDIT_FP8 && a->act == MMDIT_ACT_SWIGLU && a->c_scales));   // (FP8: MX output of the SwiGLU epilogue)
  MMDIT_CHECK_ARG(aligned16(a->A) && aligned16(a->B) && ((uintptr_t)a->C & (4 * esc - 1)) == 0);
  MMDIT_CHECK_ARG((a->lda * esa) % 16 == 0 && (a->ldb * esb) % 16 == 0 && a->ldc % 4 == 0);
  if (a->a_kmajor) { MMDIT_CHECK_ARG(a->M % 8 == 0 && a->lda >= a->M); } else { MMDIT_CHECK_ARG(a->K % 8 == 0 && a->lda >= a->K); }
  if (a->b_kmajor) { MMDIT_CHECK_ARG(a->N % 8 == 0 && a->ldb >= a->N); } else { MMDIT_CHECK_ARG(a->K % 8 == 0 && a->ldb >= a->K); }
  MMDIT_CHECK_ARG(a->ldc >= (a->act == MMDIT_ACT_SWIGLU ? a->N / 2 : a->act == MMDIT_ACT_SWIGLU_BWD ? 2 * a->N : a->N));
  if (a->bias) MMDIT_CHECK_ARG(aligned16(a->bias));
  if (a->gate) MMDIT_CHECK_ARG(a->residual && a->rows_per_batch > 0 && aligned16(a->gate) && a->ld_gate % 4 == 0);
  if (a->residual) MMDIT_CHECK_ARG(aligned16(a->residual) && a->ld_res % 4 == 0);
  if (a->aux) MMDIT_CHECK_ARG(a->ld_aux % 4 == 0 && ((uintptr_t)a->aux & 7) == 0);
  if (a->accumulate) MMDIT_CHECK_ARG(a->c_dtype == MMDIT_F32);
  return 0;
}

}  // namespace

// Tile-configuration heuristic of the LDS-DMA path.  Bigger tiles halve the L2->CU traffic per FLOP (a 128x128
// tile needs ~64 B/clk/CU at full MFMA rate, about what the L2 can deliver) but need enough tiles to fill 256 CUs.
// workspace of the lean weight-gradient kernel's split tail (device memory owned by the caller; first 4 KiB: zero-initialised tickets)
// (one registration per DEVICE: the tickets and slots are device memory, and a ticket left non-zero by a launch on one GPU must not be
//  seen by another)
static void* g_ws[64] = {};
static long long g_ws_bytes[64] = {};
// Scheduler page of the workspace (bytes [4096, 8192): 64 slots of 16 ints, zero-filled by the caller with the tickets): a launch that claims its
// tiles dynamically (gemm8p.hip) takes the next slot round robin and leaves it zeroed.  Launches that use the workspace are stream-ordered (header),
// so one slot would do; the ring keeps a launch on another stream from sharing the words of its 63 predecessors.
static unsigned g_sched_next[64] = {};
static int g_claiming[64] = {};      // mmdit_gemm_set_claiming
extern "C" int mmdit_gemm_set_claiming(int on) { g_claiming[mmdit_current_device()] = on != 0; return 0; }
extern "C" int mmdit_gemm_get_claiming(void) { return g_claiming[mmdit_current_device()]; }
int* mmdit_gemm_sched_slot() {
  const int dev = mmdit_current_device();
  if (!g_ws[dev] || !g_claiming[dev]) return nullptr;
  return (int*)((char*)g_ws[dev] + 4096) + (__atomic_fetch_add(&g_sched_next[dev], 1u, __ATOMIC_RELAXED) & 63u) * 16;      // (launchers may run on several host threads)
}
// compute units the persistent launches may count on, per device (0 = not set: all of the device's); mmdit_set_cu_budget
static int g_cu_budget[64] = {};
// compute units of the current device (hipDeviceAttributeMultiprocessorCount, read once per device; 256 = an MI355X when no device answers: the planner
// also runs without one, mmdit_gemm_plan in the CPU tests)
int mmdit_device_cus() {
  static int cus[64] = {};
  const int dev = mmdit_current_device();
  if (!cus[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) { (void)hipGetLastError(); n = 256; }
    cus[dev] = n / 8 * 8;      // (the XCD round-robin stays even)
  }
  return cus[dev];
}
extern "C" int mmdit_get_cu_budget(void) { const int n = g_cu_budget[mmdit_current_device()]; return n ? n : mmdit_device_cus(); }
extern "C" int mmdit_set_cu_budget(int n) {
  MMDIT_CHECK_ARG(n >= 64 && n <= mmdit_device_cus() && n % 8 == 0);
  g_cu_budget[mmdit_current_device()] = n == mmdit_device_cus() ? 0 : n;
  return 0;
}

static int pick_dma_cfg(const mmdit_gemm_args* args, int count, int split_k, bool stream_k, bool lean_ok) {
  static const char* force = mmdit_exp_env("MMDIT_GEMM_CFG");
  if (force) return atoi(force) == CFG_320x256 && !lean_ok ? CFG_256x256 : atoi(force);
  static const char* force_epi = mmdit_exp_env("MMDIT_GEMM_CFG_EPI");   // experiments: tile configuration of the gated-residual (fp32 C) launches only
  if (force_epi && args[0].gate) return atoi(force_epi);
  if (stream_k) return CFG_256x256;   // no quantisation with stream-K: take the fewest bytes per FLOP
  // Wave quantisation decides (measured, tools/gemm_bench.py): a "round" of 128x128 tiles (2 workgroups per CU)
  // costs 1.0, a round of 256x256 tiles (1 per CU, 4x the FLOPs each) 1.58.
  long t128 = 0, t256 = 0;
  for (int i = 0; i < count; i++) {
    t128 += (long)((args[i].M + 127) / 128) * ((args[i].N + 127) / 128);
    t256 += (long)((args[i].M + 255) / 256) * ((args[i].N + 255) / 256);
  }
  // (e4m3 operands: a 256x256 round costs 1.67 rounds of 128x128 tiles -- tools/probes/fp8_bench.py: qkv 80 vs 96 us, w12 190 vs 226 us
  // in favour of 256x256, out-proj / w3 (N = 768) 37 vs 39 us, 98 vs 107 us in favour of 128x128)
  // (round 5: MX operands with a bf16 / SwiGLU output run the 8-phase kernel at 256x256 (gemm8p.hip MX) -- out-proj 32.0 vs 38.0 us, w3 76.4 vs
  //  100.1 us in favour of 256x256 now: a 256x256 round costs ~1.2 rounds of 128x128 tiles there)
  const bool mx8 = args[0].a_dtype == MMDIT_FP8 && args[0].K % 128 == 0 && !args[0].gate && !args[0].residual && !args[0].aux &&
                   (args[0].c_dtype == MMDIT_BF16 || args[0].act == MMDIT_ACT_SWIGLU);
  const double r256 = mx8 ? 1.2 : args[0].a_dtype == MMDIT_FP8 ? 1.67 : 1.58;
  const long cu = mmdit_get_cu_budget();
  const double c128 = (double)((t128 * split_k + 2 * cu - 1) / (2 * cu)), c256 = r256 * (double)((t256 * split_k + cu - 1) / cu);
  if (lean_ok) {
    // 320x256 tiles (lean kernel): a round costs 1.25x a 256x256 round (tile area); MMDiT-B's N = 768 GEMMs at batch 64 fit ONE round
    long t320 = 0;
    for (int i = 0; i < count; i++) t320 += (long)((args[i].M + 319) / 320) * ((args[i].N + 255) / 256);
    const double c320 = 1.25 * 1.58 * (double)((t320 + cu - 1) / cu);
    if (c320 < c256 && c320 < c128) return CFG_320x256;
  }
  return c256 < c128 ? CFG_256x256 : CFG_128x128;
}

static inline int split_k_of(const mmdit_gemm_args* a) { return a->split_k > 1 ? a->split_k : 1; }

// QKV epilogue request of mmdit_gemm_qkv_norm_rope (nullptr: a plain launch)
struct QkRequest {
  const mmdit_qk_epilogue* qk;
  int heads, s_total;
  void* Q; void* K; void* V;
  unsigned no_raw;      // bit i: problem i was given C == NULL (the raw q / k columns are not wanted)
};

static int gemm_grouped_impl(const mmdit_gemm_args* args, int count, mmdit_stream_t stream, bool plan_only, unsigned* zero_mask = nullptr, const QkRequest* qkr = nullptr,
                             bool no_dma_override = false) {
  MMDIT_CHECK_ARG(args && count >= 1 && count <= MAXG);
  const mmdit_gemm_args* a0 = &args[0];
  GroupParams gp;
  gp.tail_first = -1; gp.tail_rounds = 0; gp.tail_G = 0;
  int aux_dt = -1;
  // LDS-DMA fast path: bf16 operands, every K a multiple of the 64-wide K-tile (MMDIT_GEMM_NO_DMA=1 forces
  // the register-staged kernel, for A/B measurements)
  static const bool no_dma = mmdit_exp_env("MMDIT_GEMM_NO_DMA") != nullptr;
  static const char* raster_env = mmdit_exp_env("MMDIT_GEMM_RASTER");
  // fp8 (e4m3) operands: DMA kernel only, row-major x row-major, K a multiple of the 128-wide fp8 K-tile, per-tensor scales
  const bool fp8 = a0->a_dtype == MMDIT_FP8 || a0->b_dtype == MMDIT_FP8;
  bool dma = !no_dma && !no_dma_override && a0->precision == MMDIT_PREC_BF16 && a0->a_dtype == MMDIT_BF16 && a0->b_dtype == MMDIT_BF16;
  // weight gradients whose reduction length is not a multiple of the K tile: the 8-phase kernel's K-tail instantiation takes them (gemm8p.hip KT);
  // if the planner ends up elsewhere the launch is re-planned without the LDS-DMA kernels (below)
  bool ktail_any = false;
  if (fp8) {
    MMDIT_CHECK_ARG(a0->a_dtype == MMDIT_FP8 && a0->b_dtype == MMDIT_FP8 && a0->precision == MMDIT_PREC_BF16 && split_k_of(a0) == 1 && !a0->stream_k);
    dma = true;
  }
  const int bk = fp8 ? 2 * BK : BK;   // elements per K-tile (two 64-byte ring halves per row)
  bool conv = false;
  const int split_k = a0->split_k > 1 ? a0->split_k : 1;
  for (int i = 0; i < count; i++) {
    const mmdit_gemm_args* a = &args[i];
    int rc = check_problem(a);
    if (rc) return rc;
    // one kernel variant per launch: dtypes, layouts, precision, activation and accumulate must agree
    MMDIT_CHECK_ARG(a->a_dtype == a0->a_dtype && a->b_dtype == a0->b_dtype && a->c_dtype == a0->c_dtype && a->a_kmajor == a0->a_kmajor &&
                    a->b_kmajor == a0->b_kmajor && a->precision == a0->precision && a->act == a0->act && a->accumulate == a0->accumulate);
    if (a->aux) { MMDIT_CHECK_ARG(aux_dt < 0 || aux_dt == a->aux_dtype); aux_dt = a->aux_dtype; }
    MMDIT_CHECK_ARG((a->split_k > 1 ? a->split_k : 1) == split_k);
    if (fp8) MMDIT_CHECK_ARG(!a->a_kmajor && !a->b_kmajor && a->K % bk == 0 && a->scale_a && a->scale_b && !a->conv_mode && a->scale_mode == a0->scale_mode &&
                             (a->scale_mode == 0 || (a->scale_mode == 1 && aligned16(a->scale_a) && aligned16(a->scale_b))));
    if (a->K % bk != 0) {
      if (!fp8 && a->a_kmajor && a->b_kmajor && a->c_dtype == MMDIT_F32 && a->K > bk && !a->conv_mode) ktail_any = true;
      else dma = false;
    }
    if (a->a_kmajor && a->M < 8) dma = false;
    if (a->b_kmajor && a->N < 8) dma = false;
    if (a->conv_mode) {
      // implicit-GEMM convolution: DMA path only (bf16, conv_C % 32 == 0 so that a K half never straddles a tap), plain row-major B
      MMDIT_CHECK_ARG((a->conv_mode == 1 || a->conv_mode == 2) && !a->a_kmajor && !a->b_kmajor && a->conv_C > 0 && a->conv_C % 32 == 0 && a->K == 9 * a->conv_C && a->K % BK == 0);
      MMDIT_CHECK_ARG(a->conv_H > 0 && a->conv_W > 0 && (a->conv_mode == 1 || (a->conv_H % 2 == 0 && a->conv_W % 2 == 0)));
      const int64_t px = a->conv_mode == 1 ? (int64_t)a->conv_H * a->conv_W : (int64_t)(a->conv_H / 2) * (a->conv_W / 2);
      MMDIT_CHECK_ARG(a->M % px == 0 && split_k == 1 && !a->stream_k && a->a_dtype == MMDIT_BF16 && a->b_dtype == MMDIT_BF16 && a->precision == MMDIT_PREC_BF16);
      MMDIT_CHECK_ARG((a->M / px) * (int64_t)(a->conv_H + 2) * (a->conv_W + 2) * a->conv_C * 2 < (1ll << 32));
      conv = true;
      continue;
    }
    // the DMA kernel addresses an operand as a wave-uniform 64-bit base + a 32-bit per-lane byte offset
    if ((int64_t)(a->a_kmajor ? a->K : a->M) * a->lda * 2 >= (1ll << 32) || (int64_t)(a->b_kmajor ? a->K : a->N) * a->ldb * 2 >= (1ll << 32)) {
      MMDIT_CHECK_ARG(!fp8);
      dma = false;
    }
  }
  if (conv) MMDIT_CHECK_ARG(dma);   // no register-staged fallback for the implicit-GEMM convolution
  const bool swiglu = a0->act == MMDIT_ACT_SWIGLU;
  if (swiglu) {
    // C[M, N/2] = silu(g) * u with [g | u] = A B^T + bias the two halves of the N = 2h columns, aux[M, N] = [g | u]: DMA kernel only
    if (!dma) return MMDIT_ERR_SHAPE;
    for (int i = 0; i < count; i++) {
      const mmdit_gemm_args* a = &args[i];
      // (c_dtype FP8: the activation leaves as MX e4m3 codes + c_scales -- MX operands, no pre-activation output)
      MMDIT_CHECK_ARG(a->c_dtype == MMDIT_BF16 || (a->c_dtype == MMDIT_FP8 && a->c_scales && !a->aux && fp8 && a->scale_mode == 1 && a->c_dtype == a0->c_dtype));
      MMDIT_CHECK_ARG(!a->a_kmajor && !a->b_kmajor && (!a->aux || a->aux_dtype == MMDIT_BF16) && !a->gate && !a->residual &&
                      !a->accumulate && split_k == 1 && !a->stream_k && !a->conv_mode);
      if (a->N % 256 != 0) return MMDIT_ERR_SHAPE;
      MMDIT_CHECK_ARG(a->ldc >= a->N / 2 && a->ldc % 8 == 0 && aligned16(a->C));
      if (a->aux) MMDIT_CHECK_ARG(a->ld_aux >= a->N && a->ld_aux % 8 == 0 && aligned16(a->aux));
    }
  }
  const bool swiglu_bwd = a0->act == MMDIT_ACT_SWIGLU_BWD;
  if (swiglu_bwd) {
    // C[M, 2N] = d[g | u] from dh = A B (N = h hidden columns) and aux[M, 2N] = [g | u]: the 8-phase kernel's 256 x 256 launch only
    if (!dma) return MMDIT_ERR_SHAPE;
    for (int i = 0; i < count; i++) {
      const mmdit_gemm_args* a = &args[i];
      MMDIT_CHECK_ARG(a->c_dtype == MMDIT_BF16 && !a->a_kmajor && a->b_kmajor && a->aux && a->aux_dtype == MMDIT_BF16 && !a->bias && !a->gate && !a->residual &&
                      !a->accumulate && split_k == 1 && !a->stream_k && !a->conv_mode && !fp8);
      MMDIT_CHECK_ARG(a->N % 8 == 0 && a->ldc >= 2 * a->N && a->ldc % 8 == 0 && aligned16(a->C) && a->ld_aux >= 2 * a->N && a->ld_aux % 8 == 0 && aligned16(a->aux));
      if (a->dbias) MMDIT_CHECK_ARG(((uintptr_t)a->dbias & 3) == 0);
    }
  }
  int bm = BM, bn = BN, cfg = CFG_128x128;
  // stream-K for the weight gradients (k-major A): fp32 C must be pre-zeroed by the caller (a0->stream_k)
  static const bool no_sk = mmdit_exp_env("MMDIT_GEMM_NO_STREAMK") != nullptr;
  const bool stream_k = dma && !no_sk && a0->stream_k && a0->c_dtype == MMDIT_F32 && split_k == 1 && a0->act == MMDIT_ACT_NONE && !a0->accumulate;
  // lean hot-path kernel (gemm_lean.hip): bf16 row-major A, bf16 output, bias / SiLU epilogue only.  MMDIT_GEMM_LEAN=0: never;
  // 2: only where it offers the 320x256 tile; 1 (default): also for 256x256 launches
  static const char* lean_env = mmdit_exp_env("MMDIT_GEMM_LEAN");
  static const int lean_mode = lean_env ? atoi(lean_env) : 1;
  bool lean_ok = dma && lean_mode > 0 && !fp8 && !conv && !stream_k && split_k == 1 && !a0->a_kmajor && a0->c_dtype == MMDIT_BF16 &&
                 (a0->act == MMDIT_ACT_NONE || a0->act == MMDIT_ACT_SILU || swiglu || swiglu_bwd) && !a0->accumulate;
  for (int i = 0; i < count && lean_ok; i++) {
    const mmdit_gemm_args* a = &args[i];
    lean_ok = (!a->aux || swiglu || swiglu_bwd) && !a->gate && !a->residual && a->N % 8 == 0 && a->ldc % 8 == 0 && aligned16(a->C) && (!a->b_kmajor || a->N >= 8);
  }
  if (dma) {
    cfg = pick_dma_cfg(args, count, split_k, stream_k, lean_ok);
    if (swiglu && cfg != CFG_320x256) cfg = CFG_256x256;   // the activation pairs gate / up columns inside a 256-column tile
    if (swiglu_bwd) cfg = CFG_256x256;                     // (the fused backward epilogue exists at 256 rows)
    static const char* qk8_env = mmdit_exp_env("MMDIT_QK_8PHASE");   // experiment (probes builds): bf16 QKV launches on the 8-phase kernel without claiming
    static const bool qk8 = qk8_env && atoi(qk8_env) > 0;
    // (MX operands + the QKV epilogue: the 8-phase kernel's 256-row tile; bf16: the same kernel when tiles are CLAIMED -- since round 6 it takes
    // the wide-slot kernel's time, 131 us at MMDiT-B, and its 927 tiles are then immune to held compute units like the other multi-round launches)
    if (qkr && (fp8 || qk8 || g_claiming[mmdit_current_device()])) cfg = CFG_256x256;
    dma_cfg_tile(cfg, bm, bn);
  }
  const bool lean = lean_ok && (cfg == CFG_320x256 || (cfg == CFG_256x256 && lean_mode == 1));
  int tiles = 0, units = 0;
  // K-decomposed launches take the problems longest-K first: the tiles of the first round are then ordered long -> short and the
  // balanced tail below can hand the split leftovers to the workgroups that finish their first tile early
  static const char* kdec_env = mmdit_exp_env("MMDIT_GEMM_KDEC");
  static const bool kdec_streamk = kdec_env && kdec_env[0] == 's', kdec_plain = kdec_env && kdec_env[0] == 'p';
  int order[MAXG];
  for (int i = 0; i < count; i++) order[i] = i;
  if (stream_k && !kdec_streamk)
    for (int i = 1; i < count; i++)
      for (int j = i; j > 0 && args[order[j]].K > args[order[j - 1]].K; j--) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
  for (int i = 0; i < count; i++) {
    const mmdit_gemm_args* a = &args[order[i]];
    Problem& p = gp.p[i];
    p.A = a->A; p.B = a->B; p.C = a->C; p.aux = a->aux;
    p.bias = a->bias; p.gate = a->gate; p.residual = a->residual;
    p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc; p.ld_gate = a->ld_gate; p.ld_res = a->ld_res; p.ld_aux = a->ld_aux;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.rows_per_batch = a->rows_per_batch > 0 ? a->rows_per_batch : 1;
    p.conv_mode = a->conv_mode; p.cC = a->conv_C; p.cHp = a->conv_H + 2; p.cWp = a->conv_W + 2;
    p.cHo = a->conv_mode == 2 ? a->conv_H / 2 : a->conv_H; p.cWo = a->conv_mode == 2 ? a->conv_W / 2 : a->conv_W;
    p.tiles_n = (a->N + bn - 1) / bn;
    p.tiles_m = (a->M + bm - 1) / bm;
    p.tile_start = tiles;
    p.nk = a->K / bk;
    p.scale_a = (const float*)a->scale_a; p.scale_b = (const float*)a->scale_b; p.c_scales = (unsigned char*)a->c_scales;
    p.dbias = a->dbias;
    p.unit_start = units;
    tiles += p.tiles_n * p.tiles_m;
    units += p.tiles_n * p.tiles_m * p.nk;
    if (stream_k) MMDIT_CHECK_ARG(!a->aux && !a->gate && a->stream_k);
  }
  gp.stream_k = stream_k; gp.total_units = units;
  gp.mx = fp8 && a0->scale_mode == 1;
  // K-decomposition for the weight gradients (few output tiles, very long reductions; a0->stream_k = "C is pre-zeroed fp32,
  // decompose along K as you like").  Default "tail": R full rounds of one-tile-per-workgroup over the whole K -- the 32
  // workgroups of an XCD then walk K together and share operand panels through its L2 (stream-K's contiguous unit ranges
  // never do: measured 3.8x operand over-fetch, HBM-bound) -- and only the T % slots leftover tiles are cut S ways along K.
  // MMDIT_GEMM_KDEC=streamk selects the stream-K schedule instead.
  // MMDIT_GEMM_KDEC=plain keeps the tail schedule but spreads the tail units over all workgroups.
  int full_tiles = split_k > 1 ? 0 : tiles, tail_split = split_k;
  bool tail_mode = false;
  if (stream_k && !kdec_streamk) {
    gp.stream_k = 0;
    tail_mode = true;
    const int G = mmdit_get_cu_budget() * (cfg == CFG_128x128 ? 2 : 1), r = tiles % G;
    int nk_min = 1 << 30;
    for (int i = 0; i < count; i++) nk_min = gp.p[i].nk < nk_min ? gp.p[i].nk : nk_min;
    full_tiles = tiles - r;
    tail_split = 1;
    if (r) {
      double best = 1e30;
      for (int S = 1; S <= 64 && S * 2 <= nk_min; S++) {
        // rounds of full-K tile time, plus the atomic adds of the r*S partial tiles (measured ~0.6 us of whole-GPU time
        // per 256x256 fp32 partial = 0.0028 rounds)
        const double c = (double)((r * S + G - 1) / G) / S + (S > 1 ? 0.0028 * r * S : 0.0);
        if (c < best) { best = c; tail_split = S; }
      }
      static const char* ts_env = mmdit_exp_env("MMDIT_GEMM_TAIL_S");   // experiments: force the K split of the tail tiles
      const int ts_force = ts_env ? atoi(ts_env) : 0;
      if (ts_force > 0 && ts_force * 2 <= nk_min) { tail_split = ts_force; best = -1.0; }
      // Balanced tail (one full round, problems of different K): in the first round the tiles of the shorter problems finish
      // early; give the tail units to exactly those workgroups (E of them) instead of stacking them on top of the longest tiles.
      if (full_tiles == G && !kdec_plain) {
        const int nk_max = gp.p[0].nk;
        int first_short = G, nk_short = 0, nk_tail = 0;
        for (int i = 0; i < count; i++) {
          const Problem& q = gp.p[i];
          if (q.nk < nk_max && q.tile_start < G && first_short == G) { first_short = q.tile_start; nk_short = q.nk; }
          if (q.tile_start + q.tiles_m * q.tiles_n > full_tiles) nk_tail = q.nk > nk_tail ? q.nk : nk_tail;
        }
        const int E = G - first_short;
        if (E > 0) {
          double bbest = 1e30;
          int bS = 0, brounds = 0;
          for (int S = 1; S <= 64 && S * 2 <= nk_min; S++) {
            const int rounds = (r * S + E - 1) / E, extra = rounds * ((nk_tail + S - 1) / S);
            const double load = (double)(nk_short + extra) / nk_max, c = (load > 1.0 ? load - 1.0 : 0.0) + (S > 1 ? 0.0028 * r * S : 0.0);
            if (c < bbest) { bbest = c; bS = S; brounds = rounds; }
          }
          // (round 6) with tile claiming on (mmdit_gemm_set_claiming + the workspace) the 8-phase kernel CLAIMS its positions (gemm8p.hip): the workgroups
          // whose first tile is short reach the tail first by themselves -- the split of this model is kept, its static assignment is not
          if (bbest < best) {
            tail_split = bS;
            if (!(g_ws[mmdit_current_device()] && g_claiming[mmdit_current_device()])) { gp.tail_first = first_short; gp.tail_rounds = brounds; gp.tail_G = G; }
          }
        }
      }
    }
  }
  static const bool no_persist = mmdit_exp_env("MMDIT_GEMM_NO_PERSIST") != nullptr;
  gp.persistent = !no_persist;
  if (aux_dt < 0) aux_dt = a0->c_dtype == MMDIT_FP8 ? MMDIT_BF16 : a0->c_dtype;
  gp.count = count; gp.total_tiles = tiles; gp.act = a0->act; gp.accumulate = a0->accumulate; gp.split_k = tail_split; gp.full_tiles = full_tiles;
  static const char* debug_env = mmdit_exp_env("MMDIT_GEMM_DEBUG");   // ablation bits (tools/gemm_ablate.py): 2 = operand stream only (no LDS reads / MFMA), 8 = no epilogue, 64 = no bf16 fast epilogue
  gp.debug = debug_env ? atoi(debug_env) : 0;
  static const char* epi_env = mmdit_exp_env("MMDIT_GEMM_EPI");
  gp.epi_direct = epi_env ? (atoi(epi_env) == 0) : 0;
  gp.raster = raster_env ? atoi(raster_env) : 8;   // n-tiles per rasterization group (see locate_tile)
  if (split_k > 1) {
    // split-K slices accumulate atomically into a pre-zeroed fp32 C: only on the DMA path, plain epilogue
    MMDIT_CHECK_ARG(dma && a0->c_dtype == MMDIT_F32 && a0->act == MMDIT_ACT_NONE && !a0->accumulate && split_k <= 64);
    for (int i = 0; i < count; i++) MMDIT_CHECK_ARG(!args[i].aux && !args[i].gate);
  }
  // lean weight-gradient kernel (gemm_lean.hip, gemm_kk_kernel): both operands k-major, fp32 C, 256x256 tiles, the round + tail (or
  // caller-split) schedule, nothing but store / accumulate / atomic add in the epilogue.  MMDIT_GEMM_KK=0: the general kernel.
  static const char* kk_env = mmdit_exp_env("MMDIT_GEMM_KK");
  bool kk = dma && (!kk_env || atoi(kk_env)) && !fp8 && !conv && !gp.stream_k && cfg == CFG_256x256 && a0->a_kmajor && a0->b_kmajor &&
            a0->c_dtype == MMDIT_F32 && a0->act == MMDIT_ACT_NONE;
  for (int i = 0; i < count && kk; i++) {
    const mmdit_gemm_args* a = &args[i];
    kk = !a->bias && !a->gate && !a->residual && !a->aux && a->N % 4 == 0 && a->ldc % 4 == 0 && aligned16(a->C);
  }
  // ... whose split tail goes through the registered workspace (mmdit_gemm_set_workspace) instead of fp32 atomics when it is large enough:
  // 4 KiB of tickets + one 256x256 fp32 slot per (tail tile, K slice)
  gp.ws_slots = nullptr; gp.ws_count = nullptr; gp.sched = nullptr;
  gp.qk_on = 0; gp.qkQ = gp.qkK = gp.qkV = nullptr; gp.qk_heads = 0; gp.qk_s_total = 0;
  if (qkr) {
    // the fused QKV epilogue exists in the wide-slot lean kernel only, for one or two streams in the caller's order, [q | k | v] columns
    // (round 5: ... and, for MX e4m3 operands, in the 8-phase kernel at 256 rows)
    const bool mxqk = dma && fp8 && gp.mx && cfg == CFG_256x256 && !a0->a_kmajor && a0->c_dtype == MMDIT_BF16;
    if (!(lean || mxqk) || count > 2 || a0->act != MMDIT_ACT_NONE || a0->b_kmajor) return MMDIT_ERR_SHAPE;
    if (qkr->no_raw && !(cfg == CFG_256x256 && (mxqk || g_claiming[mmdit_current_device()]))) return MMDIT_ERR_SHAPE;      // (only the 8-phase kernel's epilogue can drop the raw columns)
    for (int i = 0; i < count; i++) {
      if (order[i] != i || args[i].bias || args[i].N != 3 * qkr->heads * 64 || qkr->qk[i].tokens <= 0 || args[i].M % qkr->qk[i].tokens) return MMDIT_ERR_SHAPE;
      gp.qk[i].wq = qkr->qk[i].wq; gp.qk[i].wk = qkr->qk[i].wk;
      gp.qk[i].rcos = qkr->qk[i].rope_cos; gp.qk[i].rsin = qkr->qk[i].rope_sin;
      gp.qk[i].tokens = qkr->qk[i].tokens; gp.qk[i].tok0 = qkr->qk[i].tok0;
    }
    if (count == 1) gp.qk[1] = gp.qk[0];
    for (int i = 0; i < count; i++)
      if ((qkr->no_raw >> i) & 1) gp.p[i].C = nullptr;
    gp.qk_on = 1; gp.qk_heads = qkr->heads; gp.qk_s_total = qkr->s_total;
    gp.qkQ = (bf16_t*)qkr->Q; gp.qkK = (bf16_t*)qkr->K; gp.qkV = (bf16_t*)qkr->V;
  }
  if (kk && gp.split_k > 1) {
    const int dev = mmdit_current_device();
    const long long tail_tiles = tiles - full_tiles;
    if (g_ws[dev] && tail_tiles <= 1024 && 8192 + tail_tiles * gp.split_k * 65536LL * 4 <= g_ws_bytes[dev]) {
      gp.ws_count = (int*)g_ws[dev];
      gp.ws_slots = (float*)((char*)g_ws[dev] + 8192);
    }
  }
  if (zero_mask) {
    // which outputs receive ATOMIC partial tiles (and therefore must be zero when the launch starts): every problem under stream-K or
    // a caller-requested split-K; with the round + tail schedule only the problems that own tiles of the split tail; none otherwise
    unsigned mask = 0;
    for (int i = 0; i < count; i++) {
      const Problem& q = gp.p[i];
      const bool atomic = dma && !gp.ws_slots && (gp.stream_k || (gp.split_k > 1 && q.tile_start + q.tiles_m * q.tiles_n > gp.full_tiles));
      if (atomic) mask |= 1u << order[i];
    }
    *zero_mask = mask;
  }
  // the 8-phase kernel (gemm8p.hip) takes every 256x256 launch of the lean kernels; MMDIT_GEMM_8P=0: the round-2/3 kernels of gemm_lean.hip
  static const char* p8_env = mmdit_exp_env("MMDIT_GEMM_8P");
  static const int p8_mode = p8_env ? atoi(p8_env) : 1;      // 1: every lean launch; 2: only the 256x256 ones
  // (the QKV launch with the QK-norm / RoPE epilogue stays on the wide kernel at 320 rows: with that epilogue's registers the 320-row 8-phase
  //  variant measured slower, 1.73 vs 1.59 ms per step)
  // e4m3 operands (E8M0 block scales or per-tensor scales) on the 8-phase loop: 256 x 256 tiles, bf16 output (bias allowed) or the SwiGLU epilogue (bf16 or MX output)
  bool mx8 = p8_mode > 0 && dma && fp8 && cfg == CFG_256x256 && (!qkr || gp.mx) && !stream_k && split_k == 1 && (a0->act == MMDIT_ACT_NONE || swiglu) && !a0->accumulate;
  for (int i = 0; i < count && mx8; i++) {
    const mmdit_gemm_args* a = &args[i];
    mx8 = (a->c_dtype == MMDIT_BF16 || (swiglu && a->c_dtype == MMDIT_FP8)) && (!a->aux || (swiglu && a->c_dtype == MMDIT_BF16)) && !a->gate && !a->residual && a->K % 128 == 0 &&
          a->N % 8 == 0 && a->ldc % 8 == 0 && aligned16(a->C) && (int64_t)a->M * a->lda < (1ll << 32) && (int64_t)a->N * a->ldb < (1ll << 32);
  }
  const bool p8 = mx8 || (p8_mode > 0 && ((kk && cfg == CFG_256x256) || (lean && (cfg == CFG_256x256 || (cfg == CFG_320x256 && p8_mode == 1 && !qkr)))));
  // implicit-GEMM convolutions on the 8-phase loop (round 5): 256 x 256 tiles, every problem a convolution with C % 64 == 0, bf16 output (bias) or fp32 output
  // (bias + residual), nothing else in the epilogue
  bool conv8 = p8_mode > 0 && conv && dma && !fp8 && cfg == CFG_256x256 && !stream_k && split_k == 1 && a0->act == MMDIT_ACT_NONE && !a0->accumulate &&
               (a0->c_dtype == MMDIT_BF16 || a0->c_dtype == MMDIT_F32);
  for (int i = 0; i < count && conv8; i++) {
    const mmdit_gemm_args* a = &args[i];
    conv8 = a->conv_mode && a->conv_C % 64 == 0 && !a->aux && !a->gate && (a->c_dtype == MMDIT_F32 || !a->residual) && a->N % 8 == 0 && a->ldc % 8 == 0 && aligned16(a->C) &&
            (!a->residual || (a->ld_res % 4 == 0 && aligned16(a->residual))) && (!a->bias || aligned16(a->bias));
  }
  if (swiglu_bwd && !p8) return MMDIT_ERR_SHAPE;
  if (dma && ktail_any && !(kk && p8)) return gemm_grouped_impl(args, count, stream, plan_only, zero_mask, qkr, true);   // (only that kernel adds a K tail)
  if (plan_only) return dma ? (cfg | (gp.stream_k ? 16 : 0) | (tail_mode ? 32 : 0) | (lean || kk ? 128 : 0) | (p8 || conv8 ? 256 : 0)) : 64;   // see mmdit_gemm_plan (128 with k-major A: the lean weight-gradient kernel)
  hipStream_t s = (hipStream_t)stream;
  if (conv8) return launch_gemm8_conv(a0->c_dtype == MMDIT_F32, gp, s);
  if (p8) return launch_gemm8(cfg, a0->a_kmajor, a0->b_kmajor, gp, s, ktail_any, mx8);
  if (lean) return launch_lean_cfg(cfg, a0->b_kmajor, gp, s);
  if (kk) return launch_lean_wgrad(gp, s);
  if (dma) return launch_dma(cfg, a0->a_kmajor, a0->b_kmajor, a0->c_dtype, aux_dt, fp8, gp, s);
  if (a0->precision == MMDIT_PREC_BF16 && a0->a_dtype == MMDIT_BF16 && a0->b_dtype == MMDIT_BF16) return dispatch_out<bf16_t, bf16_t, false>(a0, aux_dt, gp, s);
  if (a0->precision == MMDIT_PREC_SPLIT && a0->a_dtype == MMDIT_F32 && a0->b_dtype == MMDIT_F32) return dispatch_out<float, float, true>(a0, aux_dt, gp, s);
  return MMDIT_ERR_DTYPE;
}

extern "C" int mmdit_gemm_grouped(const mmdit_gemm_args* args, int count, mmdit_stream_t stream) { return gemm_grouped_impl(args, count, stream, false); }

extern "C" int mmdit_gemm_qkv_norm_rope(const mmdit_gemm_args* args, const mmdit_qk_epilogue* qk, int count, int heads, int s_total,
                                        void* Q, void* K, void* V, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(args && qk && count >= 1 && count <= 2 && heads > 0 && s_total > 0 && Q && K && V);
  for (int i = 0; i < count; i++) {
    MMDIT_CHECK_ARG(qk[i].wq && qk[i].wk && qk[i].tokens > 0 && qk[i].tok0 >= 0 && qk[i].tok0 + qk[i].tokens <= s_total);
    MMDIT_CHECK_ARG((qk[i].rope_cos == nullptr) == (qk[i].rope_sin == nullptr));
  }
  // C == NULL: the raw q / k columns are not wanted (inference).  The planner sees a stand-in pointer (Q: valid, aligned, never written through C).
  mmdit_gemm_args tmp[2];
  unsigned no_raw = 0;
  for (int i = 0; i < count; i++) {
    tmp[i] = args[i];
    if (!tmp[i].C) { tmp[i].C = Q; no_raw |= 1u << i; }
  }
  const QkRequest r{qk, heads, s_total, Q, K, V, no_raw};
  return gemm_grouped_impl(no_raw ? tmp : args, count, stream, false, nullptr, &r);
}

extern "C" int mmdit_gemm_set_workspace(void* ptr, long long bytes) {
  MMDIT_CHECK_ARG((ptr == nullptr && bytes == 0) || (ptr != nullptr && bytes >= 8192 + 65536 * 4 && ((uintptr_t)ptr & 15) == 0));
  const int dev = mmdit_current_device();      // registered for the CURRENT device (hipSetDevice before the call)
  g_ws[dev] = ptr;
  g_ws_bytes[dev] = bytes;
  return 0;
}

// mmdit_debug_occupy: a stand-in for a long-running kernel on another stream (a collective's channels) -- `wgs` one-wave workgroups with 1 KiB of LDS each
// (enough that a 160 KiB GEMM workgroup cannot share their compute unit) sleep-spin for `cycles` shader cycles.  tests / tools/probes/cu_contention.py.
namespace {
__global__ void occupy_kernel(long long cycles, int* sink) {
  __shared__ int pad[256];
  pad[threadIdx.x & 255] = threadIdx.x;
  const long long t0 = __builtin_readcyclecounter();
  while ((long long)__builtin_readcyclecounter() - t0 < cycles) __builtin_amdgcn_s_sleep(32);
  if (cycles < 0) sink[0] = pad[0];
}
}  // namespace
extern "C" int mmdit_debug_occupy(int wgs, long long cycles, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(wgs >= 1 && wgs <= 256 && cycles >= 0 && cycles <= (1ll << 36));
  hipLaunchKernelGGL(occupy_kernel, dim3(wgs), dim3(64), 0, (hipStream_t)stream, cycles, (int*)nullptr);
  return mmdit_launch_status();
}

extern "C" int mmdit_gemm_plan(const mmdit_gemm_args* args, int count) { return gemm_grouped_impl(args, count, nullptr, true); }

extern "C" int mmdit_gemm_zero_mask(const mmdit_gemm_args* args, int count, unsigned* mask) {
  MMDIT_CHECK_ARG(mask);
  *mask = 0;
  const int rc = gemm_grouped_impl(args, count, nullptr, true, mask);
  return rc < 0 || rc > 511 ? rc : 0;   // (plan codes are small non-negative integers (9 bits); anything else is a status)
}

extern "C" int mmdit_gemm(const mmdit_gemm_args* a, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(a);
  return mmdit_gemm_grouped(a, 1, stream);
}
