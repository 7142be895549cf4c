// Shared declarations of the GEMM translation units (gemm.hip: register-staged general / parity kernel
// and host dispatch; gemm_dma.hip: LDS-DMA fast kernels).
#pragma once
#include "common.h"

namespace gemm {

constexpr int BK = 64;
constexpr int MAXG = 12;                 // problems per grouped launch
constexpr int NXCD = 8;

struct Problem {
  const void* A; const void* B; void* C; void* aux;
  const float* bias; const float* gate; const float* residual;
  int64_t lda, ldb, ldc, ld_gate, ld_res, ld_aux;
  int M, N, K;
  int rows_per_batch, tiles_n, tiles_m, tile_start;
  int nk, unit_start;   // stream-K: K-tiles per output tile, first (tile, K-tile) unit of this problem
  // implicit-GEMM 3x3 convolution (conv_mode != 0): A is a zero-bordered NHWC bf16 tensor (batch, cHp, cWp, cC); output row
  // m = (b, yo, xo) reads pixel (s*yo + kh + o, s*xo + kw + o) for K index (kh*3 + kw)*cC + c  (mode 1: s=1, o=0; mode 2: s=2, o=1)
  int conv_mode, cHo, cWo, cHp, cWp, cC;
  const float* scale_a; const float* scale_b;   // fp8 operands: per-tensor dequantisation scales (device scalars), C = sa*sb*(A_q B_q^T); MX mode: E8M0 bytes (mx_scale_index)
  unsigned char* c_scales;                      // SwiGLU epilogue with an MX e4m3 output: E8M0 scales of C (C then holds e4m3 codes)
  float* dbias;                                 // SwiGLU-backward epilogue: column sums of C are added here (nullptr: not wanted)
};
// QKV projection with the per-head QK RMSNorm + axial RoPE + joint-layout store in its epilogue (gemm_lean.hip, mmdit_gemm_qkv_norm_rope)
struct QkEpi {
  const float* wq; const float* wk;          // (64) norm weights of this stream
  const float* rcos; const float* rsin;      // (tokens, 64) RoPE factors, nullptr: no rotation (text stream)
  int tokens, tok0;                          // row = b * tokens + n;  joint position = tok0 + n
};
struct GroupParams {
  Problem p[MAXG];
  int count, total_tiles, act, accumulate, split_k, raster, debug, epi_direct;
  int full_tiles;   // tiles [0, full_tiles) are multiplied over their whole K by one workgroup; tiles [full_tiles, total_tiles) are cut
                    // split_k ways along K (partials added atomically into a pre-zeroed fp32 C)
  // Balanced tail (tail_first >= 0; needs full_tiles == tail_G, problems sorted by K descending): the split tail units go only to
  // the workgroups whose first-round tile is a SHORT one (its position in the tile order >= tail_first), tail_rounds units each.
  int tail_first, tail_rounds, tail_G;
  int stream_k, total_units, persistent;   // stream-K: every workgroup gets an equal share of the (tile, K-tile) units
  int mx;                                  // fp8 operands: scale_a / scale_b are E8M0 block-scale tensors (MX), not per-tensor fp32 scalars
  // lean weight-gradient kernel: the partial tiles of the split tail go through a workspace instead of fp32 atomics -- slice sk of tail
  // tile tt is stored to ws_slots + (tt * split_k + sk) * 65536, the LAST of the tile's split_k slices to arrive (ticket ws_count[tt])
  // sums the slots in slice order and writes C (deterministic; C needs no zero-fill).  nullptr: atomics.
  float* ws_slots;
  int* ws_count;
  // Dynamic tile claiming (gemm8p.hip; round 6): 16 ints of the registered workspace -- [0, 8) the heads of the per-XCD queues (queue q holds the
  // work positions q, q + 8, q + 16, ... < total_work in that order), [8] the count of workgroups that have left the launch (the last one zeroes all
  // nine words: no memset node, nothing for the host to do between launches).  nullptr: the static walk (workgroup b takes b, b + G, b + 2 G, ...).
  int* sched;
  // QKV epilogue (qk_on): problems 0 / 1 = image / text stream, C = the raw projection [q | k | v] as always, plus Q / K / V (batch, heads,
  // s_total, 64) bf16 written from the rounded raw values (Attention.py:130-135, 178-194, 258-261)
  QkEpi qk[2];
  bf16_t* qkQ; bf16_t* qkK; bf16_t* qkV;
  int qk_on, qk_heads, qk_s_total;
};

// block id -> (problem, m-tile, n-tile, split-K slice).
//  * XCD chunking: block b is observed to run on XCD b % 8 (private 4 MB L2 each); every XCD gets a contiguous
//    range of the tile sequence (used for L2 locality only, never for correctness).
//  * Rasterization inside a problem: tiles are walked in column groups of gp.raster n-tiles, m fastest between
//    groups' rows, so that one group of B panels (raster*BN rows of the weight) stays L2-resident while the
//    A row panels stream past it once.
// work items of a launch: the unsplit tiles first, then (tile, K-slice) pairs of the split tail
__device__ __host__ __forceinline__ int total_work(const GroupParams& gp) {
  return gp.tail_first >= 0 ? gp.full_tiles + gp.tail_rounds * gp.tail_G : gp.full_tiles + (gp.total_tiles - gp.full_tiles) * gp.split_k;
}
__device__ __forceinline__ bool is_split_work(const GroupParams& gp, int work) { return work >= gp.full_tiles && gp.split_k > 1; }

// work index -> contiguous range per XCD of a sequence of W items (workgroup b runs on XCD b % 8; full_tiles is a multiple of 8
// whenever a split tail follows it, so the same holds for the tail's local index)
__device__ __forceinline__ int xcd_chunk(int w, int W) {
  const int q = W / NXCD, r = W % NXCD, xcd = w % NXCD, j = w / NXCD;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

__device__ __forceinline__ const Problem& locate_in_problem(const GroupParams& gp, int t, int& tm, int& tn);

__device__ __forceinline__ int work_tile(const GroupParams& gp, int work, int& sk) {
  // Inside the split tail the K-slice is the SLOW index (k-chunk-major): an XCD owns (mostly) one K-slice of ALL tail
  // tiles, so each operand panel of that slice is fetched once into its L2 instead of once per tile.
  int t;
  if (work < gp.full_tiles) {
    sk = 0;
    t = xcd_chunk(work, gp.full_tiles);
  } else {
    const int Tt = gp.total_tiles - gp.full_tiles, lin = xcd_chunk(work - gp.full_tiles, Tt * gp.split_k);
    sk = lin / Tt;
    t = gp.full_tiles + lin - sk * Tt;
  }
  return t;
}

__device__ __forceinline__ const Problem& locate_tile(const GroupParams& gp, int work, int& tm, int& tn, int& sk) {
  return locate_in_problem(gp, work_tile(gp, work, sk), tm, tn);
}

__device__ __forceinline__ const Problem& locate_in_problem(const GroupParams& gp, int t, int& tm, int& tn) {
  int pi = 0;
#pragma unroll 1
  for (int i = 1; i < gp.count; i++) pi = (t >= gp.p[i].tile_start) ? i : pi;
  const Problem& p = gp.p[pi];
  t -= p.tile_start;
  const int gw = gp.raster, per_group = gw * p.tiles_m, grp = t / per_group, rem = t - grp * per_group;
  const int w = min(gw, p.tiles_n - grp * gw);   // width of this (possibly last, narrower) column group
  tm = rem / w;
  tn = grp * gw + rem - tm * w;
  return p;
}

// Epilogue.  The accumulators hold C^T fragments (lane = output row, 4 consecutive columns per register
// group), which would make every global store touch 32 different rows.  Each wave therefore stages its
// 32-row x 64-column block through a wave-private LDS region (pitch 272 B: conflict-free 16-B writes and
// reads) and then walks it row-contiguously: one wave instruction covers 4 rows x 256 B (fp32) / 128 B (bf16)
// so C / aux stores and the residual / gate / bias loads are fully coalesced.
// split-K slices (gp.split_k > 1) add their partial sums atomically into a pre-zeroed fp32 C; bias / residual
// are contributed by slice 0 only.
constexpr int EP_PITCH = 272;
constexpr int EP_WAVE_BYTES = 32 * EP_PITCH;   // LDS bytes each wave needs for the epilogue

template <typename TC, typename TAUX, int MI = 2, int NJ = 2>
__device__ __forceinline__ void epilogue(const f32x16 (&acc)[MI][NJ], const Problem& p, const GroupParams& gp, int m0, int n0, int wm, int wn,
                                         int lane, int sk, char* stage /* wave-private, EP_WAVE_BYTES */, bool atomic_out = false) {
  static_assert(NJ == 2, "wave sub-tile must be 64 columns wide");
  TC* C = (TC*)p.C;
  TAUX* AUX = (TAUX*)p.aux;
  const bool first = sk == 0;
  const float* bias = first ? p.bias : nullptr;
  const float* gate = p.gate;
  const float* res = first ? p.residual : nullptr;
  const int c4 = (lane & 15) * 4, col = n0 + wn * 64 + c4;
  float b4[4] = {0.f, 0.f, 0.f, 0.f};
  if (bias && col < p.N) ld4(bias + col, b4);
#pragma unroll
  for (int i = 0; i < MI; i++) {
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
      for (int g = 0; g < 4; g++)
        *LDS_PTR(f32x4, stage + (lane & 31) * EP_PITCH + (j * 32 + 8 * g + 4 * (lane >> 5)) * 4) =
            (f32x4){acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private region: program order is enough
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int r = it * 4 + (lane >> 4);
      const f32x4 t = *LDS_PTR(const f32x4, stage + r * EP_PITCH + c4 * 4);
      const int row = m0 + wm * (MI * 32) + i * 32 + r;
      if (row >= p.M || col >= p.N) continue;
      float v[4] = {t[0] + b4[0], t[1] + b4[1], t[2] + b4[2], t[3] + b4[3]};
      if (AUX) st4(AUX + (int64_t)row * p.ld_aux + col, v);
      if (gp.act == MMDIT_ACT_SILU) {
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = silu_f(v[e]);
      }
      if (res) {
        float r4[4];
        ld4(res + (int64_t)row * p.ld_res + col, r4);
        if (gate) {
          float g4[4];
          ld4(gate + (int64_t)(row / p.rows_per_batch) * p.ld_gate + col, g4);
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = r4[e] + g4[e] * v[e];
        } else {
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] += r4[e];
        }
      }
      TC* cp = C + (int64_t)row * p.ldc + col;
      if constexpr (sizeof(TC) == 4) {
        if (gp.split_k > 1 || atomic_out) {
#pragma unroll
          for (int e = 0; e < 4; e++) atomicAdd((float*)cp + e, v[e]);
          continue;
        }
      }
      if (gp.accumulate) {
        float c4v[4];
        ld4(cp, c4v);
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] += c4v[e];
      }
      st4(cp, v);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next block overwrites the region
  }
}

// previous form kept for A/B measurements (MMDIT_GEMM_EPI=0): every lane stores its own row directly
template <typename TC, typename TAUX, int MI = 2, int NJ = 2>
__device__ __forceinline__ void epilogue_direct(const f32x16 (&acc)[MI][NJ], const Problem& p, const GroupParams& gp, int m0, int n0, int wm, int wn, int lane, int sk) {
  TC* C = (TC*)p.C;
  TAUX* AUX = (TAUX*)p.aux;
  const bool first = sk == 0;
  const float* bias = first ? p.bias : nullptr;
  const float* gate = p.gate;
  const float* res = first ? p.residual : nullptr;
#pragma unroll
  for (int i = 0; i < MI; i++) {
    const int row = m0 + wm * (MI * 32) + i * 32 + (lane & 31);
    if (row >= p.M) continue;
    const float* grow = gate ? gate + (int64_t)(row / p.rows_per_batch) * p.ld_gate : nullptr;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int col = n0 + wn * (NJ * 32) + j * 32 + 8 * g + 4 * (lane >> 5);
        if (col >= p.N) continue;
        float v[4] = {acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]};
        if (bias) {
          float b4[4];
          ld4(bias + col, b4);
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] += b4[e];
        }
        if (AUX) st4(AUX + (int64_t)row * p.ld_aux + col, v);
        if (gp.act == MMDIT_ACT_SILU) {
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = silu_f(v[e]);
        }
        if (res) {
          float r4[4];
          ld4(res + (int64_t)row * p.ld_res + col, r4);
          if (grow) {
            float g4[4];
            ld4(grow + col, g4);
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = r4[e] + g4[e] * v[e];
          } else {
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] += r4[e];
          }
        }
        TC* cp = C + (int64_t)row * p.ldc + col;
        if constexpr (sizeof(TC) == 4) {
          if (gp.split_k > 1) {
#pragma unroll
            for (int e = 0; e < 4; e++) atomicAdd((float*)cp + e, v[e]);
            continue;
          }
        }
        if (gp.accumulate) {
          float c4[4];
          ld4(cp, c4);
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] += c4[e];
        }
        st4(cp, v);
      }
    }
  }
}


// tile configurations of the LDS-DMA kernel (gemm_dma.hip)
enum DmaCfg { CFG_128x128 = 0, CFG_256x128 = 1, CFG_256x256 = 2, CFG_320x256 = 3 };   // 320x256: lean kernel only (gemm_lean.hip)
inline void dma_cfg_tile(int cfg, int& bm, int& bn) { bm = cfg == CFG_128x128 ? 128 : cfg == CFG_320x256 ? 320 : 256; bn = cfg >= CFG_256x256 ? 256 : 128; }
int launch_dma(int cfg, bool a_km, bool b_km, int c_dtype, int aux_dtype, bool fp8, const GroupParams& gp, hipStream_t s);
// lean hot-path kernel: bf16 row-major A, bf16 B (row- or k-major), bf16 C, bias / SiLU only; cfg CFG_256x256 or CFG_320x256
int launch_lean_cfg(int cfg, bool b_km, const GroupParams& gp, hipStream_t s);
int launch_lean_wgrad(const GroupParams& gp, hipStream_t s);   // gemm_lean.hip: k-major x k-major -> fp32, 256x256, K-decomposed schedule
// gemm8p.hip: the de-phased 8-phase main loop on 256x256 tiles (16x16x32 MFMA): every launch the lean kernels take at that tile size
int launch_gemm8_fp8(bool b_km, const GroupParams& gp, hipStream_t s);       // gemm8p_inf.hip: e4m3 operands (gp.mx: E8M0 block scales, else per-tensor), 256x256 tiles
int launch_gemm8_conv(bool f32_out, const GroupParams& gp, hipStream_t s);   // gemm8p.hip CONV: implicit-GEMM 3x3 convolution, 256x256 tiles, bf16 (+ bias) or fp32 (+ bias + residual) output
int launch_gemm8(int cfg, bool a_km, bool b_km, const GroupParams& gp, hipStream_t s, bool ktail = false, bool fp8 = false);   // ktail: some weight-gradient K is not a multiple of 64   // cfg: CFG_256x256 / CFG_320x256 (bf16 outputs only)

}  // namespace gemm
