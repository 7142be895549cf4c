// Joint [image;text] softmax attention for gfx950, head_dim 64, bf16 MFMA operands, fp32 accumulate.
//
// All matmuls are issued "transposed" so that the softmax row statistic of a query lives in ONE lane:
//   S^T = K Q^T  (v_mfma_f32_32x32x16_bf16: lane owns query l&31, 16 keys per accumulator)
//   O^T = V^T P^T: the accumulator registers of S^T are already the B operand of this MFMA (the
//   key order inside a 16-key slab is a fixed permutation which the V^T operand reproduces by
//   addressing), V^T fragments come from a row-major [key][d] LDS tile via ds_read_b64_tr_b16.
// Forward: one wave = 32 queries, KV tiles of 64 keys double-buffered through LDS.
// Backward: two kernels (dQ: query-stationary; dK/dV: key-stationary) so nothing is atomically reduced.
#include <type_traits>
#include "common.h"

namespace {

constexpr int HD = 64;
constexpr int KT = 64;          // keys (or queries) per LDS tile
constexpr int P144 = 144;       // row pitch for ds_read_b128 fragments (conflict-free)
constexpr int P192 = 192;       // row pitch for ds_read_b64_tr_b16 fragments (conflict-free)
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// B-operand fragment from 8 consecutive accumulator registers (fp32 -> bf16), regs [h8*8, h8*8+8)
__device__ __forceinline__ bf16x8 pack_frag(const f32x16& p, int h8) {
  u32x4 w;
#pragma unroll
  for (int i = 0; i < 4; i++) w[i] = pack_bf2(p[h8 * 8 + 2 * i], p[h8 * 8 + 2 * i + 1]);
  return __builtin_bit_cast(bf16x8, w);
}

// A-operand fragment of X^T where X is a row-major [row][64] LDS tile (pitch bytes): lane owns column
// cb*32 + (lane&31); its 8 k-slots are rows  rb + 16*h8 + 4*(lane>>5) + {0..3}  and  + 8 + {0..3}
// (the row order produced by pack_frag on an accumulator whose rows are these tile rows).
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int pitch, int rb, int h8, int cb, int lane) {
  const int row = rb + 16 * h8 + 4 * (lane >> 5) + ((lane & 15) >> 2);
  const int col = cb * 32 + 16 * ((lane >> 4) & 1) + (lane & 3) * 4;
  const char* p = tile + row * pitch + col * 2;
  s16x4 lo = lds_tr16(p), hi = lds_tr16(p + 8 * pitch);
  s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}

// A-operand fragment of a row-major [row][64] LDS tile: lane owns row rb + (lane&31), k = ks*16 + (lane>>5)*8 ..+7
__device__ __forceinline__ bf16x8 row_frag(const char* tile, int pitch, int rb, int ks, int lane) {
  return *LDS_PTR(const bf16x8, tile + (rb + (lane & 31)) * pitch + (ks * 16 + (lane >> 5) * 8) * 2);
}

// copy a [64 rows][64] bf16 tile global -> LDS (512 16-byte chunks), rows >= nvalid zero-filled.  NT threads (any multiple of 64).
template <int NT> constexpr int tile_chunks() { return (512 + NT - 1) / NT; }
template <int NT>
__device__ __forceinline__ void tile_g2r(u32x4 (&st)[tile_chunks<NT>()], const bf16_t* g, int row0, int nrows_total, int tid) {
#pragma unroll
  for (int i = 0; i < tile_chunks<NT>(); i++) {
    const int c = tid + i * NT, row = c >> 3, kc = c & 7;
    st[i] = (c < 512 && row0 + row < nrows_total) ? *(const u32x4*)(g + (int64_t)(row0 + row) * HD + kc * 8) : (u32x4){0, 0, 0, 0};
  }
}
template <int NT>
__device__ __forceinline__ void tile_r2s(const u32x4 (&st)[tile_chunks<NT>()], char* tile, int pitch, int tid) {
#pragma unroll
  for (int i = 0; i < tile_chunks<NT>(); i++) {
    const int c = tid + i * NT, row = c >> 3, kc = c & 7;
    if (512 % NT == 0 || c < 512) *LDS_PTR(u32x4, tile + row * pitch + kc * 16) = st[i];
  }
}

// Unpadded [64 rows][128 B] tile that is read BOTH ways (ds_read_b128 row fragments and ds_read_b64_tr_b16 transposed fragments,
// e.g. K in the dQ kernel, Q / dO in the dK/dV kernel): no single row pitch is conflict-free for both (144 B: 2-way conflicts on the
// transposed reads, PMC: 21-23 % of the LDS cycles of the backward kernels; 192 B: 4-way on the row reads), an XOR swizzle is:
// 16-byte piece p of row r lives at slot p ^ sw2(r), sw2(r) = ((r >> 1) & 7) ^ (((r >> 1) & 1) << 2).
__device__ __forceinline__ int sw2(int r) { return ((r >> 1) & 7) ^ (((r >> 1) & 1) << 2); }
template <int NT>
__device__ __forceinline__ void tile_r2s_sw(const u32x4 (&st)[tile_chunks<NT>()], char* tile, int tid) {
#pragma unroll
  for (int i = 0; i < tile_chunks<NT>(); i++) {
    const int c = tid + i * NT, row = c >> 3, kc = c & 7;
    if (512 % NT == 0 || c < 512) *LDS_PTR(u32x4, tile + row * 128 + ((kc ^ sw2(row)) << 4)) = st[i];
  }
}
__device__ __forceinline__ bf16x8 row_frag_d(const char* tile, int rb, int ks, int lane) {
  const int r = rb + (lane & 31);
  return *LDS_PTR(const bf16x8, tile + r * 128 + (((ks * 2 + (lane >> 5)) ^ sw2(r)) << 4));
}
__device__ __forceinline__ bf16x8 tr_frag_d(const char* tile, int rb, int h8, int cb, int lane) {
  const int row = rb + 16 * h8 + 4 * (lane >> 5) + ((lane & 15) >> 2);
  const int cbyte = (cb * 32 + 16 * ((lane >> 4) & 1) + (lane & 3) * 4) * 2;
  const char* p0 = tile + row * 128 + ((((cbyte >> 4) ^ sw2(row)) << 4) | (cbyte & 15));
  const char* p1 = tile + (row + 8) * 128 + ((((cbyte >> 4) ^ sw2(row + 8)) << 4) | (cbyte & 15));
  s16x4 lo = lds_tr16(p0), hi = lds_tr16(p1);
  s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}

// pointer to the 64-wide head slice of token s (joint index) in the stream-split (B, L, H*64) tensors
__device__ __forceinline__ const bf16_t* tok_ptr(const bf16_t* x_img, const bf16_t* x_txt, int64_t b, int s, int n_img, int n_txt, int D, int h) {
  if (s < n_img) return x_img + ((b * n_img + s) * (int64_t)D + h * HD);
  return x_txt ? x_txt + ((b * n_txt + (s - n_img)) * (int64_t)D + h * HD) : nullptr;
}

// 1-D grid -> (tile, batch*head).  Workgroup b is observed to run on XCD b % 8 (private L2 each): all tiles of one
// (batch, head) are placed on ONE XCD so that its K/V (or Q/dO) is fetched from HBM once instead of once per XCD
// (measured: 577 MiB fetched per forward launch against 120 MB of Q,K,V before this mapping).  Locality only.
__device__ __forceinline__ void map_block(int ntile, int BH, int& tile, int& bh) {
  const int id = blockIdx.x;
  if (BH % 8 == 0) {
    const int xcd = id & 7, j = id >> 3;
    tile = j % ntile;
    bh = (j / ntile) * 8 + xcd;
  } else {
    tile = id % ntile;
    bh = id / ntile;
  }
}

// ------------------------------------------------------------------------------------------------
// Backward of the per-head QK RMSNorm (+ axial RoPE on image tokens) fused into the epilogues of the attention backward kernels
// (mmdit_attn_bwd_qk): the dQ / dK rows a wave holds in its accumulators go straight to the gradient of the raw QKV projection, the
// stand-alone mmdit_qk_norm_rope_bwd pass (and the dQ / dK / dV round trip through HBM) disappears.  Same arithmetic as
// qk_norm_rope_bwd_body in rowops.hip (reference: Attention.py:130-135, 178-194 under autograd), on the accumulator layout:
// lane = row (l & 31), its 32 features f(db, g, e) = 32 db + 8 g + 4 (l >> 5) + e; the other 32 live in lane l ^ 32.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float partner32(float x, int lane);   // (defined with the forward kernels below)
constexpr float QK_RMS_EPS = 1.1920929e-07f;     // nn.RMSNorm(eps=None) on fp32 input: finfo(float32).eps (= RMS_EPS of rowops.hip)

struct QkFuse {
  const bf16_t* qkv_x; const bf16_t* qkv_c;      // saved raw projections, row-major (rows, 3 * H * 64): [q | k | v] per row
  const float* wq_x; const float* wk_x; const float* wq_c; const float* wk_c;   // norm weights (64)
  const float* rcos; const float* rsin;          // RoPE factors (n_img, 64)
  bf16_t* dqkv_x; bf16_t* dqkv_c;                // outputs, same geometry as qkv_x / qkv_c
  float* dw;                                     // (256) norm-weight gradients [wq_x | wk_x | wq_c | wk_c], accumulated (one atomic add per workgroup and feature)
};

// The epilogue goes through the LDS: a wave parks its 32 x 64 fp32 accumulator tile row-major (stride QK_ROW_F floats: b128 writes of
// the accumulator layout and b128 reads of the row layout are both conflict-free), then 8 lanes own one row (8 features = one 16-byte
// piece of the bf16 row each), 8 rows per pass, 4 passes.  Every global access is then a full 128-byte row segment per 8 lanes
// (the first version worked on the accumulator layout directly: 8-byte pieces of 32 different rows per instruction, 56 such
// instructions per wave -- the texture addresser made the fused form SLOWER than the two-pass one, 401 vs 336 us at MMDIT-B).
constexpr int QK_ROW_F = 68;
constexpr int QK_WAVE_BYTES = 32 * QK_ROW_F * 4;          // 8704 B per wave
template <int NW> constexpr int qk_lds_bytes() { return NW * QK_WAVE_BYTES + 128 * 4; }     // tiles + [image | text][64] weight-gradient sums

__device__ __forceinline__ void acc_to_lds(const f32x16 (&acc)[2], float* tile, int lane) {
  float* p = tile + (lane & 31) * QK_ROW_F + 4 * (lane >> 5);
#pragma unroll
  for (int db = 0; db < 2; db++)
#pragma unroll
    for (int g = 0; g < 4; g++)
      *LDS_PTR(f32x4, p + db * 32 + 8 * g) = (f32x4){acc[db][g * 4], acc[db][g * 4 + 1], acc[db][g * 4 + 2], acc[db][g * 4 + 3]};
  __builtin_amdgcn_wave_barrier();
}
// rows 16 half .. 16 half + 15 of the accumulator tile only, parked as rows 0 .. 15 of `tile16` (16 x QK_ROW_F floats)
__device__ __forceinline__ void acc_to_lds_half(const f32x16 (&acc)[2], float* tile16, int lane, int half) {
  if (((lane & 31) >> 4) == half) {
    float* p = tile16 + (lane & 15) * QK_ROW_F + 4 * (lane >> 5);
#pragma unroll
    for (int db = 0; db < 2; db++)
#pragma unroll
      for (int g = 0; g < 4; g++)
        *LDS_PTR(f32x4, p + db * 32 + 8 * g) = (f32x4){acc[db][g * 4], acc[db][g * 4 + 1], acc[db][g * 4 + 2], acc[db][g * 4 + 3]};
  }
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void lds_row8(const float* tile, int r, int c, float (&v)[8]) {
  const f32x4 a = *LDS_PTR(const f32x4, tile + r * QK_ROW_F + 8 * c), b = *LDS_PTR(const f32x4, tile + r * QK_ROW_F + 8 * c + 4);
#pragma unroll
  for (int e = 0; e < 4; e++) { v[e] = a[e]; v[4 + e] = b[e]; }
}
// tile: the wave's parked accumulators = gradient w.r.t. the normalised (and rotated) rows, to be multiplied by `mul`.  x0 / o0: raw
// features / output slot of the wave's row 0 (this head, this part), `pitch` elements between rows; nvalid rows of the 32 exist.
// cs0 / sn0: RoPE factors of row 0's token (image rows; rows are consecutive tokens), nullptr for text rows.  Adds this wave's
// norm-weight gradient into sdw[64] (LDS atomics from 8 lanes).
// the raw rows qk_bwd_tile needs (4 passes x 16 bytes per lane), requested ahead of it
__device__ __forceinline__ void qk_bwd_rows(u32x4 (&xr)[4], int lane, int nvalid, const bf16_t* x0, int64_t pitch) {
  const int c = lane & 7, rsub = lane >> 3;
#pragma unroll
  for (int it = 0; it < 4; it++) xr[it] = __builtin_nontemporal_load((const u32x4*)(x0 + min(it * 8 + rsub, nvalid - 1) * pitch + 8 * c));
}
template <int IT0 = 0, int IT1 = 4>      // passes [IT0, IT1) of 8 rows each (the one-pass backward parks 16 rows at a time: `tile` then points 16 * (IT0 / 2) rows in front of the parked ones)
__device__ __forceinline__ void qk_bwd_tile(const float* tile, float mul, int lane, int nvalid, const bf16_t* x0, bf16_t* o0, int64_t pitch,
                                            const float* w, const float* cs0, const float* sn0, float* sdw, const u32x4* xpre = nullptr) {
  const int c = lane & 7, rsub = lane >> 3;
  float w8[8], dw[8];
  ld8(w + 8 * c, w8);
#pragma unroll
  for (int e = 0; e < 8; e++) dw[e] = 0.f;
  // the four passes' raw rows are requested up front (16 registers as packed bf16): their latency overlaps the first pass
  u32x4 xr[4];
  if (xpre) {      // (compile-time at every call site: the caller requested the rows earlier -- qk_bwd_rows -- so that their latency lies under its barriers)
#pragma unroll
    for (int it = IT0; it < IT1; it++) xr[it] = xpre[it];
  } else {
#pragma unroll
    for (int it = IT0; it < IT1; it++) xr[it] = __builtin_nontemporal_load((const u32x4*)(x0 + min(it * 8 + rsub, nvalid - 1) * pitch + 8 * c));   // (last use of the saved projection)
  }
  // Round 6 (profiles/r06_epilogue_waits.txt).  Loads and stores of a wave retire through ONE in-order counter: a pass that requests its RoPE factors
  // behind the previous pass's store waits for that store, and the compiler's count is exact only in straight-line code.  FULL (all 32 rows exist:
  // every wave but the last one or two of a sequence) is therefore a compile-time variant without a conditional store, in which the factors of pass
  // p + 1 are requested into the SAME registers right after pass p's last use of them and BEFORE its store: pass p + 1 then waits with vmcnt(1).
  // Text rows (no rotation) read the norm weights -- 64 valid floats -- and ignore them, so that the load is unconditional too.
  const bool rope = cs0 != nullptr;
  const float* tc = rope ? cs0 : w;
  const float* ts = rope ? sn0 : w;
  auto body = [&](auto full_t) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_t)::value;
    float c8[8], s8[8];
    auto request = [&](int it) __attribute__((always_inline)) {
      const int rq = rope ? (FULL ? it * 8 + rsub : min(it * 8 + rsub, nvalid - 1)) : 0;
      ld8(tc + rq * 64 + 8 * c, c8);
      ld8(ts + rq * 64 + 8 * c, s8);
    };
    request(IT0);
#pragma unroll
    for (int it = IT0; it < IT1; it++) {
      const int r = it * 8 + rsub;
      const bool valid = FULL || r < nvalid;
      float dz[8], x[8];
#pragma unroll
      for (int e = 0; e < 4; e++) { x[2 * e] = __builtin_bit_cast(float, xr[it][e] << 16); x[2 * e + 1] = __builtin_bit_cast(float, xr[it][e] & 0xffff0000u); }
      lds_row8(tile, r, c, dz);
#pragma unroll
      for (int e = 0; e < 8; e++) dz[e] *= mul;
      if (rope) {   // transpose of the rotation: pairs (2p, 2p+1)
#pragma unroll
        for (int p = 0; p < 4; p++) {
          const float da = dz[2 * p], dbv = dz[2 * p + 1];
          dz[2 * p] = da * c8[2 * p] + dbv * s8[2 * p + 1];
          dz[2 * p + 1] = dbv * c8[2 * p + 1] - da * s8[2 * p];
        }
      }
      if (it + 1 < IT1) request(it + 1);      // (the factors of this pass are dead: same registers; before this pass's store)
      float ss = 0.f;
#pragma unroll
      for (int e = 0; e < 8; e++) ss += x[e] * x[e];
      const float rinv = rsqrtf(sum8(ss) * (1.f / 64.f) + QK_RMS_EPS);
      float dot = 0.f;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        x[e] *= rinv;                                      // xhat
        dw[e] += valid ? dz[e] * x[e] : 0.f;               // d(norm weight)
        dz[e] *= w8[e];                                    // d(xhat)
        dot += dz[e] * x[e];
      }
      dot = sum8(dot) * (1.f / 64.f);
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] = rinv * (dz[e] - x[e] * dot);
      if (valid) st8(o0 + r * pitch + 8 * c, o);
    }
  };
  if (nvalid >= 8 * IT1) body(std::integral_constant<bool, true>{});
  else body(std::integral_constant<bool, false>{});
  // the wave's 32 rows: lanes with equal c hold partial sums for features 8c .. 8c+7 (lane bits 5:3 = row lane)
#pragma unroll
  for (int e = 0; e < 8; e++) {
    float v = dw[e];
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (lane < 8) atomicAdd(&sdw[8 * c + e], v);
  }
}
// dV rows: no arithmetic, the LDS round trip only makes the stores whole 128-byte rows
template <int IT0 = 0, int IT1 = 4>
__device__ __forceinline__ void rows_from_tile(const float* tile, int lane, int nvalid, bf16_t* o0, int64_t pitch) {
  const int c = lane & 7, rsub = lane >> 3;
#pragma unroll
  for (int it = IT0; it < IT1; it++) {
    const int r = it * 8 + rsub;
    float v[8];
    lds_row8(tile, r, c, v);
    if (r < nvalid) st8(o0 + r * pitch + 8 * c, v);
  }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int NW, bool ORACLE>
__global__ __launch_bounds__(NW * 64) void attn_fwd_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
                                                            int BH, int H, int S, int n_img, float scale,
                                                            bf16_t* __restrict__ Ox, bf16_t* __restrict__ Oc, float* __restrict__ lse) {
  constexpr int NT = NW * 64;
  __shared__ __attribute__((aligned(16))) char smem[2 * KT * P144 + 2 * KT * P192];
  auto kt = [&](int i) { return smem + i * KT * P144; };
  auto vt = [&](int i) { return smem + 2 * KT * P144 + i * KT * P192; };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  MMDIT_YOUNG_HALF_PRIO();
  int qtile, bh;
  map_block((S + 32 * NW - 1) / (32 * NW), BH, qtile, bh);
  const int h = bh % H;
  const int64_t b = bh / H;
  const bf16_t* Qb = Q + (int64_t)bh * S * HD;
  const bf16_t* Kb = K + (int64_t)bh * S * HD;
  const bf16_t* Vb = V + (int64_t)bh * S * HD;
  const int q = qtile * 32 * NW + wave * 32 + (lane & 31);
  const int qc = min(q, S - 1);
  const bool active = qtile * 32 * NW + wave * 32 < S;   // wave-uniform: a wave whose 32 queries are all padding only helps with the tile copies

  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ks++) qf[ks] = *(const bf16x8*)(Qb + (int64_t)qc * HD + ks * 16 + (lane >> 5) * 8);

  f32x16 o[2];
#pragma unroll
  for (int db = 0; db < 2; db++)
#pragma unroll
    for (int r = 0; r < 16; r++) o[db][r] = 0.f;
  float m = -INFINITY, l = 0.f;
  const int nkv = (S + KT - 1) / KT;
  u32x4 sk[tile_chunks<NT>()], sv[tile_chunks<NT>()];

  // scores of tile j from LDS buffer kb_ -> s[2] (masked keys = -inf).  fast: log2 domain; oracle: natural, bf16-rounded
  auto scores = [&](int j, const char* ktile, f32x16 (&s)[2]) {
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
      if (j * KT + kb * 32 >= S) {   // (wave-uniform) the whole 32-key block is padding: no MFMAs, all scores masked
#pragma unroll
        for (int r = 0; r < 16; r++) s[kb][r] = -INFINITY;
        continue;
      }
#pragma unroll
      for (int r = 0; r < 16; r++) s[kb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ks++) s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(ktile, P144, kb * 32, ks, lane), qf[ks], s[kb], 0, 0, 0);
      if (ORACLE) {
#pragma unroll
        for (int r = 0; r < 16; r++) s[kb][r] = bf2f(f2bf(bf2f(f2bf(s[kb][r])) * scale));   // Attention.py:277: bf16 matmul, then bf16 * scale
      }
      if ((j + 1) * KT > S) {   // only the last (ragged) tile has keys to mask: wave-uniform branch
#pragma unroll
        for (int r = 0; r < 16; r++)
          if (j * KT + kb * 32 + acc_row(r, lane) >= S) s[kb][r] = -INFINITY;
      }
    }
  };
  auto tile_max = [&](const f32x16 (&s)[2]) {
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int r = 0; r < 16; r++) mx = fmaxf(mx, s[kb][r]);
    return fmaxf(mx, __shfl_xor(mx, 32, 64));
  };
  auto pv = [&](int j, const f32x16 (&p)[2], const char* vtile) {
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
      if (j * KT + kb * 32 < S)   // (the probabilities of a padding block are all zero)
#pragma unroll
      for (int h8 = 0; h8 < 2; h8++) {
        const bf16x8 pf = pack_frag(p[kb], h8);
#pragma unroll
        for (int db = 0; db < 2; db++) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(vtile, P192, kb * 32, h8, db, lane), pf, o[db], 0, 0, 0);
      }
  };

  for (int pass = 0; pass < (ORACLE ? 2 : 1); pass++) {
    tile_g2r<NT>(sk, Kb, 0, S, tid);
    tile_g2r<NT>(sv, Vb, 0, S, tid);
    __syncthreads();  // previous pass done with the buffers
    tile_r2s<NT>(sk, kt(0), P144, tid);
    tile_r2s<NT>(sv, vt(0), P192, tid);
    __syncthreads();
    for (int j = 0; j < nkv; j++) {
      const int cur = j & 1;
      if (j + 1 < nkv) {
        tile_g2r<NT>(sk, Kb, (j + 1) * KT, S, tid);
        tile_g2r<NT>(sv, Vb, (j + 1) * KT, S, tid);
      }
      f32x16 s[2];
      if (active) scores(j, kt(cur), s);
      if (!active) {
      } else if (!ORACLE) {
        // raw scores; m is tracked in the scaled log2 domain: p = exp2(s*c - m) as one fma + v_exp_f32
        const float c = scale * LOG2E;
        const float mn = fmaxf(m, tile_max(s) * c);
        const float alpha = fast_exp2(m - mn);
        float rs = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
          for (int r = 0; r < 16; r++) { const float p = fast_exp2(fmaf(s[kb][r], c, -mn)); s[kb][r] = p; rs += p; }
        rs += __shfl_xor(rs, 32, 64);
        l = l * alpha + rs; m = mn;
#pragma unroll
        for (int db = 0; db < 2; db++)
#pragma unroll
          for (int r = 0; r < 16; r++) o[db][r] *= alpha;
        pv(j, s, vt(cur));
      } else if (pass == 0) {
        const float mn = fmaxf(m, tile_max(s));
        float rs = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
          for (int r = 0; r < 16; r++) rs += expf(s[kb][r] - mn);
        rs += __shfl_xor(rs, 32, 64);
        l = l * expf(m - mn) + rs; m = mn;
      } else {
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
          for (int r = 0; r < 16; r++) s[kb][r] = expf(s[kb][r] - m) / l;   // softmax output, rounded to bf16 by pack_frag
        pv(j, s, vt(cur));
      }
      if (j + 1 < nkv) {
        tile_r2s<NT>(sk, kt(cur ^ 1), P144, tid);
        tile_r2s<NT>(sv, vt(cur ^ 1), P192, tid);
        __syncthreads();
      }
    }
  }

  if (q < S) {
    const float inv = ORACLE ? 1.f : 1.f / l;
    const int n_txt = S - n_img, D = H * HD;
    bf16_t* dst = q < n_img ? Ox + ((b * n_img + q) * (int64_t)D + h * HD) : Oc + ((b * n_txt + (q - n_img)) * (int64_t)D + h * HD);
#pragma unroll
    for (int db = 0; db < 2; db++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        float v4[4] = {o[db][g * 4] * inv, o[db][g * 4 + 1] * inv, o[db][g * 4 + 2] * inv, o[db][g * 4 + 3] * inv};
        st4(dst + db * 32 + 8 * g + 4 * (lane >> 5), v4);
      }
    if (lane < 32) lse[(int64_t)bh * S + q] = ORACLE ? m + logf(l) : m * LN2 + logf(l);
  }
}

// ------------------------------------------------------------------------------------------------
// forward, LDS-DMA variant (fast mode): K/V tiles go global -> LDS with global_load_lds (no VGPR staging, no ds_write) into a
// 4-stage ring, three tiles ahead of the MFMAs -- with one tile of look-ahead the loop was bound by the L2 latency of the
// tile copy, not by MFMA or VALU work.  The LDS image of a DMA is lane-linear, so tiles are unpadded [64 keys][128 B] and the
// bank swizzle is applied to the per-lane SOURCE address and mirrored by the fragment reads:
//   K (ds_read_b128 row fragments):      16-B piece p of key row r lives at slot p ^ ((r >> 1) & 7)
//   V (ds_read_b64_tr_b16 fragments):    16-B piece p of key row r lives at slot p ^ (4 * ((r >> 1) & 1))
// Rows beyond S are clamped to the last valid row (their scores are masked / their probabilities are zero).
// One s_waitcnt vmcnt + one s_barrier per tile; every wave issues exactly two pieces per tile (8 waves = 8 + 8 KiB).
// Round 4: nothing is issued past the last tile (the ring used to reload clamped copies: 3 of 7 + 3 tile copies at S = 410), the Q loads
// go out first and share the first tile's counted wait, the first MFMA of a score block takes a constant-zero C, scale-and-shift and row
// sums are packed fp32 pairs: 70 -> 67.5 us.  Workgroup geometry, same box (tools/probes/attn_fwd_geo.sh, waves x ring stages):
// 8x4 70.2, 8x3 69.6, 4x4 104, 4x3 101, 4x2 101 us -- four independent 4-wave workgroups per CU de-phase MFMA and VALU work but copy every
// K/V tile twice as often, and the LDS-DMA stream (64 KiB per tile step and CU) becomes the limit.
// ------------------------------------------------------------------------------------------------
constexpr int ANS = 4;   // ring stages

__device__ __forceinline__ void attn_glds16(const void* gptr, uint32_t lds_dst_) {
  const uint32_t lds_dst = __builtin_amdgcn_readfirstlane(lds_dst_);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ void attn_glds16s(uint32_t voff, const char* sbase_, uint32_t lds_dst_) {   // wave-uniform base + 32-bit lane offset
  const uint64_t a = (uint64_t)(uintptr_t)sbase_;
  const char* sbase = (const char*)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a));
  const uint32_t lds_dst = __builtin_amdgcn_readfirstlane(lds_dst_);
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
// value of lane ^ 32 without an LDS round trip: v_permlane32_swap exchanges the upper half of its first operand with the lower
// half of the second; whichever way the compiler allocates the two copies of x (one register: both results are the partner's
// value; two registers: [own | partner] and [partner | own]), the select below picks the partner
__device__ __forceinline__ float partner32(float x, int lane) {
  const auto pr = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
  return __builtin_bit_cast(float, lane < 32 ? pr[1] : pr[0]);
}
// fragments of the swizzled unpadded tiles
__device__ __forceinline__ bf16x8 row_frag_sw(const char* tile, int rb, int ks, int lane) {
  const int r = rb + (lane & 31), kp = ks * 2 + (lane >> 5);
  return *LDS_PTR(const bf16x8, tile + r * 128 + ((kp ^ ((r >> 1) & 7)) << 4));
}
__device__ __forceinline__ bf16x8 tr_frag_sw(const char* tile, int rb, int h8, int cb, int lane) {
  const int row = rb + 16 * h8 + 4 * (lane >> 5) + ((lane & 15) >> 2);
  const int cbyte = (cb * 32 + 16 * ((lane >> 4) & 1) + (lane & 3) * 4) * 2;
  const char* p = tile + row * 128 + ((((cbyte >> 4) ^ (4 * ((row >> 1) & 1))) << 4) | (cbyte & 15));
  s16x4 lo = lds_tr16(p), hi = lds_tr16(p + 8 * 128);     // row + 8: same swizzle bit
  s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}

// DBG (ablation builds, tools/probes/attn_ablate.py): 1 no exponentials (p = s), 2 no P V MFMAs, 4 no K Q^T MFMAs, 8 no LDS fragment reads
// (constant operands), 16 no per-tile wait / barrier / DMA, 32 no rescale / row sums
// MXO (fp8 inference): the output leaves as MX e4m3 -- codes in place of the bf16 rows (same (B, tokens, H * 64) geometry, one byte per
// feature) plus E8M0 block scales in the GEMM's layout (mx_scale_index; a head's 64 features are two 32-blocks) -- so the out-projection
// GEMM needs no quantise pass; bit-identical to the bf16 output followed by mmdit_mxfp8_quantize.
template <int DBG = 0, bool MXO = false, int NW = 8, int NS = ANS>
__global__ __launch_bounds__(64 * NW) void attn_fwd_dma_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
                                                           int BH, int H, int S, int n_img, float scale,
                                                           bf16_t* __restrict__ Ox, bf16_t* __restrict__ Oc, float* __restrict__ lse,
                                                           unsigned char* __restrict__ scx = nullptr, unsigned char* __restrict__ scc = nullptr) {
  constexpr int PPW = 8 / NW;   // 1 KiB pieces (8 key rows) of each operand tile this wave copies
  static_assert(NW * PPW == 8 && NS >= 2 && NW * 4096 <= NS * 2 * KT * 128, "tile pieces / epilogue staging");
  __shared__ __attribute__((aligned(16))) char smem[NS * 2 * KT * 128];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  MMDIT_YOUNG_HALF_PRIO();
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  int qtile, bh;
  map_block((S + 32 * NW - 1) / (32 * NW), BH, qtile, bh);
  const int h = bh % H;
  const int64_t b = bh / H;
  const bf16_t* Qb = Q + (int64_t)bh * S * HD;
  const bf16_t* Kb = K + (int64_t)bh * S * HD;
  const bf16_t* Vb = V + (int64_t)bh * S * HD;
  const int q = qtile * 32 * NW + wave * 32 + (lane & 31);
  const int qc = min(q, S - 1);
  const bool active = qtile * 32 * NW + wave * 32 < S;

  if (DBG & 64) return;   // launch floor
  const int nkv = (S + KT - 1) / KT;
  // this lane's part of a tile: piece p = PPW * wave + i holds key rows 8 p .. 8 p + 7; row 8 p + (lane >> 3), 16-byte slot lane & 7
  const int slot = lane & 7;
  auto issue = [&](int j, int stage) {
#pragma unroll
    for (int i = 0; i < PPW; i++) {
      const int pc = wave * PPW + i, rl = 8 * pc + (lane >> 3);
      const int kcol = (slot ^ ((rl >> 1) & 7)) * 8, vcol = (slot ^ (4 * ((rl >> 1) & 1))) * 8;
      const int row = min(j * KT + rl, S - 1);
      attn_glds16(Kb + (int64_t)row * HD + kcol, lds0 + stage * (2 * KT * 128) + pc * 1024);
      attn_glds16(Vb + (int64_t)row * HD + vcol, lds0 + stage * (2 * KT * 128) + KT * 128 + pc * 1024);
    }
  };
  // The Q fragments are requested FIRST and the ring's first NS - 1 tiles behind them; nothing waits here: the first tile's counted wait
  // in the loop covers the Q loads too (they are older).  The Q loads are asm so that the compiler does not put a vmcnt(0) of its own
  // in front of their first use (it cannot see the DMA queue); the empty asm after the loop's wait carries the dependence.
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ks++) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qf[ks]) : "v"(Qb + (int64_t)qc * HD + ks * 16 + (lane >> 5) * 8) : "memory");
#pragma unroll
  for (int st = 0; st < NS - 1; st++)
    if (st < nkv) issue(st, st);

  f32x16 o[2];
#pragma unroll
  for (int db = 0; db < 2; db++)
#pragma unroll
    for (int r = 0; r < 16; r++) o[db][r] = 0.f;
  float m = -INFINITY, l = 0.f;
  const float c = scale * LOG2E;
  // Per-lane LDS byte offsets of the fragment reads, computed ONCE: everything that depends on the tile / key block / k-step is
  // a compile-time constant added as the ds_read immediate (the swizzles only involve lane bits), so a tile costs 6 address
  // adds instead of one per read.
  //   K row fragment (kb, ks): row 32 kb + (lane & 31), piece (2 ks + (lane >> 5)) ^ ((row >> 1) & 7)  =  [(ks ^ f) << 5] + per-lane part
  //   V^T fragment (kb, h8, db): row 32 kb + 16 h8 + 4 (lane >> 5) + rr, piece ^ 4 ((rr >> 1) & 1)      =  [(db ^ b) << 6] + per-lane part
  uint32_t kofs[4], vofs[2];
  {
    const int l31 = lane & 31, hi5 = lane >> 5, swz = (l31 >> 1) & 7, f = swz >> 1, e = hi5 ^ (swz & 1);
#pragma unroll
    for (int ks = 0; ks < 4; ks++) kofs[ks] = l31 * 128 + ((ks ^ f) << 5) + (e << 4);
    const int rr = (lane & 15) >> 2, x = (lane >> 4) & 1, y = lane & 3, bb = (rr >> 1) & 1;
#pragma unroll
    for (int db = 0; db < 2; db++) vofs[db] = KT * 128 + (4 * hi5 + rr) * 128 + ((db ^ bb) << 6) + (x << 5) + ((y >> 1) << 4) + ((y & 1) << 3);
  }
  int stage = 0;
  for (int j = 0; j < nkv; j++) {
    if (!(DBG & 16) || j == 0) {
    // tile j has landed; the min(NS - 2, nkv - 1 - j) younger tiles may be in flight (nothing is issued past the last tile)
    const int ahead = nkv - 1 - j;
    if (NS >= 3 && ahead >= NS - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW * (NS - 2)) : "memory");
    else if (NS >= 4 && ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]));
    __builtin_amdgcn_s_barrier();                                          // ... for every wave; and everyone has left tile j-1
    if (j + NS - 1 < nkv) issue(j + NS - 1, stage == 0 ? NS - 1 : stage - 1);   // refill the stage of tile j-1
    }
    const char* tile = smem + stage * (2 * KT * 128);
    stage = stage + 1 == NS ? 0 : stage + 1;
    if (!active) continue;
    const char* kp[4] = {tile + kofs[0], tile + kofs[1], tile + kofs[2], tile + kofs[3]};
    const char* vp[2] = {tile + vofs[0], tile + vofs[1]};
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
      // (a fully padded second block of the last tile is multiplied too -- its rows are copies of the last key -- and masked below: a
      //  skip would cost every tile 16 speculative v_mov of -inf)
      if (DBG & 4) {
#pragma unroll
        for (int r = 0; r < 16; r++) s[kb][r] = (float)qf[r & 3][0];
      } else {
        // the first MFMA of the chain takes the constant 0 as its C operand: no 16 v_mov per block to clear the accumulator
        constexpr f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
          s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((DBG & 8) ? qf[(ks + 1) & 3] : *LDS_PTR(const bf16x8, kp[ks] + kb * 32 * 128), qf[ks], ks == 0 ? zero16 : s[kb], 0, 0, 0);
      }
      if ((j + 1) * KT > S) {
#pragma unroll
        for (int r = 0; r < 16; r++)
          if (j * KT + kb * 32 + acc_row(r, lane) >= S) s[kb][r] = -INFINITY;
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int r = 0; r < 16; r++) mx = fmaxf(mx, s[kb][r]);
    mx = fmaxf(mx, partner32(mx, lane)) * c;   // the two lanes of a query (l, l + 32)
    // the accumulators are rescaled only when some query's running maximum grows (exact: alpha == 1 otherwise); after the first
    // tiles that is rare, and it takes 32 multiplies per lane out of most iterations
    if (!(DBG & 32) && !__all(mx <= m)) {
      const float mn = fmaxf(m, mx);
      const float alpha = fast_exp2(m - mn);
      l *= alpha;
      m = mn;
#pragma unroll
      for (int db = 0; db < 2; db++)
#pragma unroll
        for (int r = 0; r < 16; r++) o[db][r] *= alpha;
    }
    // p = 2^(s c - m): the scale-and-shift and the row sums as packed fp32 pairs (v_pk_fma_f32 / v_pk_add_f32: two values per issue slot);
    // the 32 v_exp_f32 stay scalar -- they are the VALU floor of a tile
#ifdef MMDIT_ATTN_SCALAR_MATH      // experiment: one-value fp32 instructions (they co-issue with the other waves' MFMAs; v_pk_* do not: tools/probes/mfma_valu_mix.hip)
    float rs = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float t = fmaf(s[kb][r], c, -m);       // (compile with -fno-slp-vectorize, or the compiler packs these again)
        const float pe = (DBG & 1) ? t : fast_exp2(t);
        s[kb][r] = pe;
        if (!(DBG & 32)) rs += pe;
      }
    l += rs + partner32(rs, lane);
#else
    f32x2 rs2 = {0.f, 0.f};
    const f32x2 c2 = {c, c}, nm2 = {-m, -m};
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 sv = {s[kb][r], s[kb][r + 1]};
        const f32x2 t = (DBG & 1) ? sv : __builtin_elementwise_fma(sv, c2, nm2);
        const f32x2 pv = {(DBG & 1) ? t[0] : fast_exp2(t[0]), (DBG & 1) ? t[1] : fast_exp2(t[1])};
        s[kb][r] = pv[0];
        s[kb][r + 1] = pv[1];
        if (!(DBG & 32)) rs2 += pv;
      }
    const float rs = rs2[0] + rs2[1];
    l += rs + partner32(rs, lane);
#endif
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
      if (j * KT + kb * 32 < S)
#pragma unroll
        for (int h8 = 0; h8 < 2; h8++) {
          const bf16x8 pf = pack_frag(s[kb], h8);
#pragma unroll
          for (int db = 0; db < 2; db++) {
            if (DBG & 2) { o[db][h8] += (float)pf[0]; continue; }
            if (DBG & 8) { o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[db + h8], pf, o[db], 0, 0, 0); continue; }
            const char* q0 = vp[db] + (kb * 32 + 16 * h8) * 128;
            const s16x4 lo = lds_tr16(q0), hi = lds_tr16(q0 + 8 * 128);
            const s16x8 vr = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vr), pf, o[db], 0, 0, 0);
          }
        }
  }
  __builtin_amdgcn_s_barrier();                      // every wave has left the last tile (no DMA is in flight: the last wait was vmcnt(0))
  if (DBG & 128) { if (o[0][0] + o[1][3] + l == 12345.f) lse[0] = 1.f; return; }
  // Epilogue: the accumulators hold O^T fragments (lane = query, 4 consecutive features per register group): stored directly, every
  // instruction would touch 32 different rows with 8 bytes each (16 instructions per wave, store-issue-bound: the empty-loop
  // ablation of this kernel takes 32 of its 73 us).  Each wave stages its 32 x 64 bf16 block through a private 4 KB of the (now
  // idle) ring -- 16-byte chunk c of row r at chunk c ^ (r & 7), conflict-free both ways -- and writes it as 8 rows x 128 B per
  // instruction (whole lines).
  {
    const float inv = 1.f / l;
    char* stg = smem + wave * 4096;
    const int wr = lane & 31, wc = lane >> 5;
#pragma unroll
    for (int db = 0; db < 2; db++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const u32x2 pk = {pack_bf2(o[db][g * 4] * inv, o[db][g * 4 + 1] * inv), pack_bf2(o[db][g * 4 + 2] * inv, o[db][g * 4 + 3] * inv)};
        *LDS_PTR(u32x2, stg + wr * 128 + (((db * 4 + g) ^ (wr & 7)) << 4) + wc * 8) = pk;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private region: program order is enough
    const int n_txt = S - n_img, D = H * HD;
    const int rr = lane >> 3, rc = lane & 7;
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const int r = it * 8 + rr, qq = qtile * 32 * NW + wave * 32 + r;
      const u32x4 t = *LDS_PTR(const u32x4, stg + r * 128 + ((rc ^ (r & 7)) << 4));
      if constexpr (MXO) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; e++) { v[2 * e] = __builtin_bit_cast(float, t[e] << 16); v[2 * e + 1] = __builtin_bit_cast(float, t[e] & 0xffff0000u); }
        float amax = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) amax = fmaxf(amax, fabsf(v[e]));
        amax = fmaxf(amax, __shfl_xor(amax, 1, 64));     // the 32-block: four 16-byte chunks = four adjacent lanes
        amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
        float inv;
        const int ex = mx_exponent(amax, inv);
        if (qq < S) {
          const bool img = qq < n_img;
          const int64_t tok = img ? b * n_img + qq : b * n_txt + (qq - n_img);
          unsigned char* dst = (unsigned char*)(img ? Ox : Oc) + tok * D + h * HD + rc * 8;
          *(uint2*)dst = make_uint2(mx_pack4(v, inv), mx_pack4(v + 4, inv));
          if ((rc & 3) == 0) (img ? scx : scc)[mx_scale_index((int)tok, h * 2 + (rc >> 2), (BH / H) * (img ? n_img : n_txt))] = (unsigned char)(ex + 127);
        }
      } else if (qq < S) {
        bf16_t* dst = qq < n_img ? Ox + ((b * n_img + qq) * (int64_t)D + h * HD) : Oc + ((b * n_txt + (qq - n_img)) * (int64_t)D + h * HD);
        *(u32x4*)(dst + rc * 8) = t;
      }
    }
    if (!MXO && q < S && lane < 32) lse[(int64_t)bh * S + q] = m * LN2 + logf(l);
  }
}

// (Two forward variants of round 2 -- a persistent one, 2 workgroups per CU walking the work items, 71.6 vs 69.7 us, and one with 64 queries
//  per wave on 4 waves, 80 vs 74 us -- were measured slower and are not part of the library any more; their text is in the history of this
//  file, commit cd8583f.)

// ------------------------------------------------------------------------------------------------
// backward dQ: query-stationary, loops over KV tiles.  dQ^T[hd][q] += K^T[hd][key] dS^T[key][q]
// ------------------------------------------------------------------------------------------------
#ifndef MMDIT_DQ_WAVES
#define MMDIT_DQ_WAVES 4
#endif
template <int NW, typename TG, bool FUSE = false>
__global__ __launch_bounds__(NW * 64, NW == 8 ? MMDIT_DQ_WAVES : 1) void attn_bwd_dq_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
                                                               const bf16_t* __restrict__ Ox, const bf16_t* __restrict__ Oc,
                                                               const bf16_t* __restrict__ dOx, const bf16_t* __restrict__ dOc,
                                                               const float* __restrict__ lse, float* __restrict__ delta,
                                                               int BH, int H, int S, int n_img, float scale, TG* __restrict__ dQ, QkFuse F = QkFuse()) {
  // delta[q] = sum_d dO[q,d] O[q,d] is formed here from the query's own dO / O rows (each lane holds half of the 64 features of its
  // query) and written out for the dK/dV kernel that follows on the same stream -- no separate preparation pass.
  constexpr int NT = NW * 64;
  // (fused epilogue: dynamic LDS of qk_lds_bytes<NW>() > 64 KB, whose head is the K / V tile pair of the main loop)
  // NW == 8: the K / V tiles arrive through a DQ_ST-stage LDS-DMA ring (global_load_lds, no staging registers, two tiles in flight behind
  // the one being multiplied) -- the register-staged copy of the other widths pays a tile's round trip between two barriers every tile.
  constexpr bool DMA = NW == 8;
#ifndef MMDIT_DQ_STAGES
#define MMDIT_DQ_STAGES 3
#endif
  constexpr int DQ_ST = MMDIT_DQ_STAGES;
  __shared__ __attribute__((aligned(16))) char smem_static[FUSE ? 16 : (DMA ? DQ_ST : 1) * 2 * KT * 128];
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  char* smem = FUSE ? smem_dyn : smem_static;
  char* ktile = smem;
  char* vtile = smem + KT * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  MMDIT_YOUNG_HALF_PRIO();
  int qtile, bh;
  map_block((S + 32 * NW - 1) / (32 * NW), BH, qtile, bh);
  const int h = bh % H;
  const int64_t b = bh / H;
  const bf16_t* Qb = Q + (int64_t)bh * S * HD;
  const bf16_t* Kb = K + (int64_t)bh * S * HD;
  const bf16_t* Vb = V + (int64_t)bh * S * HD;
  const int q = qtile * 32 * NW + wave * 32 + (lane & 31);
  const int qc = min(q, S - 1);
  const bool active = qtile * 32 * NW + wave * 32 < S;
  const bf16_t* dop = tok_ptr(dOx, dOc, b, qc, n_img, S - n_img, H * HD, h);

  bf16x8 qf[4], dof[4];
#pragma unroll
  for (int ks = 0; ks < 4; ks++) {
    qf[ks] = *(const bf16x8*)(Qb + (int64_t)qc * HD + ks * 16 + (lane >> 5) * 8);
    u32x4 z = {0, 0, 0, 0};
    dof[ks] = dop ? *(const bf16x8*)(dop + ks * 16 + (lane >> 5) * 8) : __builtin_bit_cast(bf16x8, z);
  }
  float dq_delta = 0.f;
  {
    const bf16_t* op = tok_ptr(Ox, Oc, b, qc, n_img, S - n_img, H * HD, h);
    if (dop && op) {
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        const u32x4 o4 = *(const u32x4*)(op + ks * 16 + (lane >> 5) * 8), d4 = __builtin_bit_cast(u32x4, dof[ks]);
#pragma unroll
        for (int e = 0; e < 4; e++)
          dq_delta += __builtin_bit_cast(float, o4[e] << 16) * __builtin_bit_cast(float, d4[e] << 16) +
                      __builtin_bit_cast(float, o4[e] & 0xffff0000u) * __builtin_bit_cast(float, d4[e] & 0xffff0000u);
      }
    }
    dq_delta += __shfl_xor(dq_delta, 32, 64);
    if (lane < 32 && q < S) delta[(int64_t)bh * S + q] = dq_delta;
  }
  const float lq = lse[(int64_t)bh * S + qc] * LOG2E;
  f32x16 acc[2];
#pragma unroll
  for (int db = 0; db < 2; db++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[db][r] = 0.f;

  const int nkv = (S + KT - 1) / KT;
  u32x4 sk[DMA ? 1 : tile_chunks<NT>()], sv[DMA ? 1 : tile_chunks<NT>()];
  // DMA: this lane's 16 bytes of a tile -- key row 8 * wave + (lane >> 3), LDS slot lane & 7 holds chunk slot ^ sw2(row) (the layout
  // tile_r2s_sw writes and row_frag_d / tr_frag_d read); rows past the end re-read the last key (masked below in the ragged tile)
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  const int rl = 8 * wave + (lane >> 3), dcol = ((lane & 7) ^ sw2(rl)) * 8;
  auto issue = [&](int j, int stage) {     // wave-uniform bases (SGPR pairs) + one 32-bit lane offset: the kernel sits at its 128-VGPR cap
    const uint32_t voff = (uint32_t)min(min(j, nkv - 1) * KT + rl, S - 1) * (HD * 2) + dcol * 2;
    attn_glds16s(voff, (const char*)Kb, lds0 + stage * (2 * KT * 128) + wave * 1024);
    attn_glds16s(voff, (const char*)Vb, lds0 + stage * (2 * KT * 128) + KT * 128 + wave * 1024);
  };
  int stage = 0;
  if constexpr (DMA) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the Q / dO / O / lse loads above: from here on only DMA pieces are counted)
#pragma unroll
    for (int st = 0; st < DQ_ST - 1; st++) issue(st, st);
  }
  for (int j = 0; j < nkv; j++) {
    if constexpr (DMA) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (DQ_ST - 2)) : "memory");   // tile j has landed (DQ_ST - 2 younger tiles may be in flight)
      __builtin_amdgcn_s_barrier();                                             // ... for every wave; and everyone has left tile j - 1
      issue(j + DQ_ST - 1, stage == 0 ? DQ_ST - 1 : stage - 1);                 // refill the stage of tile j - 1
      ktile = smem + stage * (2 * KT * 128);
      vtile = ktile + KT * 128;
      stage = stage + 1 == DQ_ST ? 0 : stage + 1;
    } else {
      tile_g2r<NT>(sk, Kb, j * KT, S, tid);
      tile_g2r<NT>(sv, Vb, j * KT, S, tid);
      __syncthreads();
      tile_r2s_sw<NT>(sk, ktile, tid);
      tile_r2s_sw<NT>(sv, vtile, tid);
      __syncthreads();
    }
    if (!active) continue;
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
      if (j * KT + kb * 32 >= S) continue;   // (wave-uniform) 32 padding keys contribute nothing
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; r++) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_d(ktile, kb * 32, ks, lane), qf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_d(vtile, kb * 32, ks, lane), dof[ks], dp, 0, 0, 0);
      }
      // per 8-row half: softmax arithmetic, pack, the half's two dQ MFMAs -- s / dp die as they are consumed (the kernel sits at its
      // 128-VGPR cap), and the second half's VALU work issues behind the first half's MFMAs.
      // Padding keys exist in the last K/V tile only: a wave-uniform branch.  (Written as a per-element `ragged && ...` select the
      // compiler evaluated index, compare and two selects for EVERY score of EVERY tile: 4 of the loop's ~11 VALU slots per score.)
      const bool ragged = (j + 1) * KT > S;
#pragma unroll
      for (int h8 = 0; h8 < 2; h8++) {
        __builtin_amdgcn_sched_barrier(0);
        f32x16 d8;      // (only elements [8 h8, 8 h8 + 8) are used: pack_frag's register window)
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const int r = 8 * h8 + i;
          const float p = fast_exp2(fmaf(s[r], scale * LOG2E, -lq));
          d8[r] = p * (dp[r] - dq_delta);
        }
        if (ragged) {
#pragma unroll
          for (int i = 0; i < 8; i++)
            if (j * KT + kb * 32 + acc_row(8 * h8 + i, lane) >= S) d8[8 * h8 + i] = 0.f;
        }
        const bf16x8 dsf = pack_frag(d8, h8);
#pragma unroll
        for (int db = 0; db < 2; db++) acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_d(ktile, kb * 32, h8, db, lane), dsf, acc[db], 0, 0, 0);
      }
    }
  }
  if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing (clamped, unused) pieces must land before the LDS is reused / released
  if constexpr (FUSE) {
    // dQ rows -> gradient of the raw q projection (RMSNorm + RoPE backward, qk_bwd_tile), norm-weight gradient partials of this workgroup
    __syncthreads();                       // every wave has left the last K/V tile: the LDS is free
    float* sdw = (float*)(smem + NW * QK_WAVE_BYTES);             // [image | text][64]
    if (tid < 128) sdw[tid] = 0.f;
    __syncthreads();
    if (active) {
      const int q0 = qtile * 32 * NW + wave * 32;
      const bool img = q0 < n_img;                                   // wave-uniform: n_img % 32 == 0 (checked by the launcher)
      const int n_txt = S - n_img, tok0 = img ? q0 : q0 - n_img, nvalid = min(32, (img ? n_img : S) - q0);
      const int64_t pitch = 3 * (int64_t)(H * HD), off = (img ? b * n_img + tok0 : b * n_txt + tok0) * pitch + h * HD;
      float* tile = (float*)(smem + wave * QK_WAVE_BYTES);
      acc_to_lds(acc, tile, lane);
      qk_bwd_tile(tile, scale, lane, nvalid, (img ? F.qkv_x : F.qkv_c) + off, (img ? F.dqkv_x : F.dqkv_c) + off, pitch, img ? F.wq_x : F.wq_c,
                  img ? F.rcos + (int64_t)tok0 * 64 : nullptr, img ? F.rsin + (int64_t)tok0 * 64 : nullptr, sdw + (img ? 0 : 64));
    }
    __syncthreads();
    if (tid < 128 && sdw[tid] != 0.f) atomicAdd(F.dw + (tid >> 6) * 128 + (tid & 63), sdw[tid]);    // [wq_x | . | wq_c | .]  (a workgroup is all image or mostly one stream: half of the adds are exact zeros)
  } else if (q < S) {
    TG* dst = dQ + ((int64_t)bh * S + q) * HD;
#pragma unroll
    for (int db = 0; db < 2; db++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        float v4[4] = {acc[db][g * 4] * scale, acc[db][g * 4 + 1] * scale, acc[db][g * 4 + 2] * scale, acc[db][g * 4 + 3] * scale};
        st4(dst + db * 32 + 8 * g + 4 * (lane >> 5), v4);
      }
  }
}

// ------------------------------------------------------------------------------------------------
// backward dK/dV: key-stationary (each wave owns 32 keys), loops over query tiles of 64.
//   S[q][key] = Q K^T, dP[q][key] = dO V^T  (lane owns key l&31, rows = queries)
//   dV^T[d][key] += dO^T[d][q] P[q][key],  dK^T[hd][key] += Q^T[hd][q] dS[q][key]
// ------------------------------------------------------------------------------------------------
// TRACE (probes build, tools/probes/attn_bwd_trace.py): lane 0 of every wave of the first 2048 workgroups records the cycle counter at
// the phase boundaries (72 slots per wave)
template <int NW, typename TG, bool FUSE = false, bool TRACE = false>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
                                                           const bf16_t* __restrict__ dOx, const bf16_t* __restrict__ dOc,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           int BH, int H, int S, int n_img, float scale, TG* __restrict__ dK, TG* __restrict__ dV, QkFuse F = QkFuse(),
                                                           unsigned long long* __restrict__ trace = nullptr) {
  int tpos = 0;
  auto stamp = [&]() {
    if constexpr (TRACE) {
      if (blockIdx.x < 2048 && (threadIdx.x & 63) == 0 && tpos < 72) trace[((int64_t)blockIdx.x * NW + (threadIdx.x >> 6)) * 72 + tpos] = __builtin_readcyclecounter();
      tpos++;
    }
  };
  stamp();                                   // 0: start
  constexpr int NT = NW * 64;
  __shared__ __attribute__((aligned(16))) char smem_static[FUSE ? 16 : 2 * KT * 128 + 2 * KT * 4];
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];      // fused epilogue: qk_lds_bytes<NW>()
  char* smem = FUSE ? smem_dyn : smem_static;
  char* qtile = smem;
  char* dotile = smem + KT * 128;
  float* lse_s = (float*)(smem + 2 * KT * 128);
  float* del_s = lse_s + KT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  MMDIT_YOUNG_HALF_PRIO();
  int ktile, bh;
  map_block((S + 32 * NW - 1) / (32 * NW), BH, ktile, bh);
  const int h = bh % H;
  const int64_t b = bh / H;
  const bf16_t* Qb = Q + (int64_t)bh * S * HD;
  const bf16_t* Kb = K + (int64_t)bh * S * HD;
  const bf16_t* Vb = V + (int64_t)bh * S * HD;
  const int key = ktile * 32 * NW + wave * 32 + (lane & 31);
  const bool active = ktile * 32 * NW + wave * 32 < S;   // wave-uniform: a wave whose 32 keys are all padding only helps with the tile copies
  const int keyc = min(key, S - 1);
  const int n_txt = S - n_img, D = H * HD;

  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ks++) {
    kf[ks] = *(const bf16x8*)(Kb + (int64_t)keyc * HD + ks * 16 + (lane >> 5) * 8);
    vf[ks] = *(const bf16x8*)(Vb + (int64_t)keyc * HD + ks * 16 + (lane >> 5) * 8);
  }
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int db = 0; db < 2; db++)
#pragma unroll
    for (int r = 0; r < 16; r++) { dk[db][r] = 0.f; dv[db][r] = 0.f; }

  const int nq = (S + KT - 1) / KT;
  // The next Q / dO tile (+ its lse / delta) is requested into registers BEFORE this tile's arithmetic and parked in LDS after it: with
  // one workgroup per CU (204 VGPRs) nothing else hides the ~2 us of a tile's round trip.  (NW = 8: one 16-byte chunk of each per thread.)
  u32x4 sq[tile_chunks<NT>()], sd[tile_chunks<NT>()];
  float lv = 0.f, dl = 0.f;
  auto request = [&](int jq) {
    tile_g2r<NT>(sq, Qb, jq * KT, S, tid);
#pragma unroll
    for (int i = 0; i < tile_chunks<NT>(); i++) {
      const int c = tid + i * NT, row = c >> 3, kc = c & 7, s = jq * KT + row;
      const bf16_t* p = (c < 512 && s < S) ? tok_ptr(dOx, dOc, b, s, n_img, n_txt, D, h) : nullptr;
      sd[i] = p ? *(const u32x4*)(p + kc * 8) : (u32x4){0, 0, 0, 0};
    }
    lv = 0.f, dl = 0.f;
    if (tid < KT && jq * KT + tid < S) { lv = lse[(int64_t)bh * S + jq * KT + tid]; dl = delta[(int64_t)bh * S + jq * KT + tid]; }   // (no arithmetic on the loaded values here: it would put the wait in front of the tile's MFMAs)
  };
  request(0);
  stamp();                                   // 1: K / V fragments and the first tile requested
  for (int jq = 0; jq < nq; jq++) {
    __syncthreads();
    stamp();                                 // per tile +0: everyone has left the previous tile
    tile_r2s_sw<NT>(sq, qtile, tid);
    tile_r2s_sw<NT>(sd, dotile, tid);
    if (tid < KT) { lse_s[tid] = lv * LOG2E; del_s[tid] = dl; }
    stamp();                                 // +1: this tile's chunks have landed and are parked in LDS
    __syncthreads();
    stamp();                                 // +2: ... everybody's
#ifndef MMDIT_DKV_NO_PREFETCH
    if (jq + 1 < nq) request(jq + 1);
#endif
    if (active) {
#pragma unroll
    for (int qb = 0; qb < 2; qb++) {
      if (jq * KT + qb * 32 >= S) continue;   // (wave-uniform) 32 padding queries contribute nothing
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; r++) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_d(qtile, qb * 32, ks, lane), kf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_d(dotile, qb * 32, ks, lane), vf[ks], dp, 0, 0, 0);
      }
      if constexpr (TRACE) stamp();            // +3 / +6: S / dP MFMAs issued
      f32x16 ds;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int r0 = qb * 32 + 8 * g + 4 * (lane >> 5);
        const f32x4 l4 = *LDS_PTR(const f32x4, lse_s + r0);
        const f32x4 d4 = *LDS_PTR(const f32x4, del_s + r0);
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int r = g * 4 + e;
          const float p = fast_exp2(fmaf(s[r], scale * LOG2E, -l4[e]));
          s[r] = p;
          ds[r] = p * (dp[r] - d4[e]);
        }
      }
      // No per-score masks in the steady state: a lane whose key is padding (key >= S) works on the clamped last key and its dK / dV
      // column is never stored; padding QUERIES exist in the last Q / dO tile only (their rows are zero-filled, lse = delta = 0, so
      // they would contribute exact zeros anyway; masked in a wave-uniform branch to keep P and dS themselves zero there).
      if ((jq + 1) * KT > S) {
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
          for (int e = 0; e < 4; e++)
            if (jq * KT + qb * 32 + 8 * g + 4 * (lane >> 5) + e >= S) { s[g * 4 + e] = 0.f; ds[g * 4 + e] = 0.f; }
      }
      if constexpr (TRACE) stamp();            // +4 / +7: softmax arithmetic issued
#pragma unroll
      for (int h8 = 0; h8 < 2; h8++) {
        const bf16x8 pf = pack_frag(s, h8), dsf = pack_frag(ds, h8);
#pragma unroll
        for (int db = 0; db < 2; db++) {
          dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_d(dotile, qb * 32, h8, db, lane), pf, dv[db], 0, 0, 0);
          dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_d(qtile, qb * 32, h8, db, lane), dsf, dk[db], 0, 0, 0);
        }
      }
      if constexpr (TRACE) stamp();            // +5 / +8: dV / dK MFMAs issued
    }
    }
    if constexpr (TRACE) { const int done = !active ? 0 : (jq * KT + 32 >= S ? 1 : 2); for (int i = done * 3; i < 6; i++) stamp(); }   // (fixed slot count per tile)
#ifdef MMDIT_DKV_NO_PREFETCH
    if (jq + 1 < nq) request(jq + 1);
#endif
  }
  stamp();                                   // loop end
  if constexpr (FUSE) {
    // dK rows -> gradient of the raw k projection (qk_bwd_tile), dV rows into the v part of the same output rows
    const int k0 = ktile * 32 * NW + wave * 32;
    const bool img = k0 < n_img;                                   // wave-uniform: n_img % 32 == 0
    const int tok0 = img ? k0 : k0 - n_img, nvalid = min(32, (img ? n_img : S) - k0);
    const int64_t pitch = 3 * (int64_t)D, off = (img ? b * n_img + tok0 : b * n_txt + tok0) * pitch + D + h * HD;
    u32x4 xk[4];                           // the raw k rows: requested here, used behind the two barriers and the parking of dK
    if (active) qk_bwd_rows(xk, lane, nvalid, (img ? F.qkv_x : F.qkv_c) + off, pitch);
    __syncthreads();                       // every wave has left the last Q / dO tile
    float* sdw = (float*)(smem + NW * QK_WAVE_BYTES);             // [image | text][64]
    if (tid < 128) sdw[tid] = 0.f;
    __syncthreads();
    if (active) {
      bf16_t* ob = (img ? F.dqkv_x : F.dqkv_c) + off;
      float* tile = (float*)(smem + wave * QK_WAVE_BYTES);
      acc_to_lds(dk, tile, lane);
      qk_bwd_tile(tile, scale, lane, nvalid, (img ? F.qkv_x : F.qkv_c) + off, ob, pitch, img ? F.wk_x : F.wk_c,
                  img ? F.rcos + (int64_t)tok0 * 64 : nullptr, img ? F.rsin + (int64_t)tok0 * 64 : nullptr, sdw + (img ? 0 : 64), xk);
      __builtin_amdgcn_wave_barrier();
      acc_to_lds(dv, tile, lane);
      rows_from_tile(tile, lane, nvalid, ob + D, pitch);
    }
    __syncthreads();
    if (tid < 128 && sdw[tid] != 0.f) atomicAdd(F.dw + (tid >> 6) * 128 + 64 + (tid & 63), sdw[tid]);   // [. | wk_x | . | wk_c]
  } else if (key < S) {
    TG* pk = dK + ((int64_t)bh * S + key) * HD;
    TG* pv = dV + ((int64_t)bh * S + key) * HD;
#pragma unroll
    for (int db = 0; db < 2; db++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        float k4[4] = {dk[db][g * 4] * scale, dk[db][g * 4 + 1] * scale, dk[db][g * 4 + 2] * scale, dk[db][g * 4 + 3] * scale};
        float v4[4] = {dv[db][g * 4], dv[db][g * 4 + 1], dv[db][g * 4 + 2], dv[db][g * 4 + 3]};
        st4(pk + db * 32 + 8 * g + 4 * (lane >> 5), k4);
        st4(pv + db * 32 + 8 * g + 4 * (lane >> 5), v4);
      }
  }
  if constexpr (TRACE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stamp();                                   // stores done
}

#ifdef MMDIT_PROBES
// ------------------------------------------------------------------------------------------------
// backward in ONE pass per (batch, head) for S <= 416 (round 6; MMDiT-B at 256^2: S = 410): one workgroup owns the whole problem.
// EXPERIMENT (probes build, MMDIT_ATTN_ONEPASS=1) -- built, correct (tests/test_kernels_gpu.py::test_attention_bwd_with_fused_qk_norm_rope_backward passes on
// it at S = 94 and 410, closer to autograd than the two kernels), and SLOWER: profiles/r06_attn_onepass_ab.txt.  MMDiT-B batch 64 (768 problems), one box:
//     two kernels (dQ 110 + dK/dV 153)                           264 us
//     this kernel                                                1508 us
//     ... without the 32 ds_add_f32 per wave and 32-query block   237 us      (MMDIT_OP_DBG=1: the reduction of the eight waves' dQ partials is the whole loss)
//     ... without dS^T park / transposed read / dQ MFMAs too      227 us      (MMDIT_OP_DBG=2)
// LDS float atomics retire ~1 lane per 2 clocks on gfx950 (7168 wave instructions per workgroup = 423 us: ~140 clocks each), so the in-LDS reduction the
// design rests on costs 5 x the kernel.  And the ceiling without ANY reduction is 237 us: 5 instead of 7 GEMM units buys 27 us of 264 once the dQ
// epilogue, delta and the second pass's idle waves (13 key blocks on 8 waves) are paid -- every non-atomic reduction (partials staged through LDS and
// added by owners: 150 KB of LDS traffic per 32-query step; full-K dQ by four owner waves from exchanged dS: +14 us and 32 KB of K tiles that do not
// fit beside the accumulator) gives that back.  The two kernels stay the product.
// Key-stationary like the dK/dV kernel above (a wave owns 32 keys: S / dP with lane = key, P / dS packed straight from the accumulators into the
// dV^T / dK^T MFMAs), in two passes over the query tiles (13 key blocks on 8 waves) -- and dQ from the SAME dS instead of a second kernel that
// recomputes S and dP: the wave parks its dS block transposed in a private LDS tile ([key][query], 80-byte rows), reads it back with
// ds_read_b64_tr_b16 as the A operand of dQ[q][d] += dS[q][key] K[key][d] (B = the wave's K rows, transposed fragments held in registers), and adds
// the 32 x 64 partial -- lane = feature, so an instruction covers 32 consecutive floats of two rows: conflict-free -- into an fp32 accumulator of the
// whole dQ in LDS (416 x 68 floats, the layout qk_bwd_tile reads) with ds_add_f32.  No S / dP recomputation (5 instead of 7 GEMM units), Q / K / V /
// dO / O are read once per pass from L2 instead of twice from HBM, delta = rowsum(dO O) is formed where the dO tile is parked.  The QK-norm / RoPE
// backward epilogues are those of the two kernels above: dK / dV per pass (parked 16 rows at a time: the LDS left beside the dQ accumulator), dQ at the end
// straight out of the accumulator.
// ------------------------------------------------------------------------------------------------
constexpr int OP_MAXB = 13;                                    // 32-row blocks of S
constexpr int OP_DST = 80;                                     // byte pitch of a wave's dS^T tile (32 queries x 2 B + 16)
constexpr int OP_DQ_BYTES = OP_MAXB * QK_WAVE_BYTES;           // 113152: the dQ accumulator
constexpr int OP_R_DST0 = 2 * KT * 128 + 2 * KT * 4;           // 16896: Q tile, dO tile, lse, delta in front of the dS^T tiles
constexpr int OP_R_BYTES = OP_R_DST0 + 8 * 32 * OP_DST;        // 37376 (>= 8 x 4 KB of K rows, >= 8 x 16 parked rows)
constexpr int OP_LDS = OP_DQ_BYTES + OP_R_BYTES + 256 * 4;     // 151552

__global__ __launch_bounds__(512) void attn_bwd_onepass_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
                                                               const bf16_t* __restrict__ Ox, const bf16_t* __restrict__ Oc,
                                                               const bf16_t* __restrict__ dOx, const bf16_t* __restrict__ dOc, const float* __restrict__ lse,
                                                               int BH, int H, int S, int n_img, float scale, QkFuse F, int dbg) {
  constexpr int NT = 512;
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  float* dqa = (float*)smem_dyn;
  char* R = smem_dyn + OP_DQ_BYTES;
  char* qtile = R;
  char* dotile = R + KT * 128;
  float* lse_s = (float*)(R + 2 * KT * 128);
  float* del_s = lse_s + KT;
  float* sdw = (float*)(smem_dyn + OP_DQ_BYTES + OP_R_BYTES);      // [q: image | text][64], [k: image | text][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* dst_w = R + OP_R_DST0 + wave * (32 * OP_DST);
  int tile0, bh;
  map_block(1, BH, tile0, bh);
  const int h = bh % H;
  const int64_t b = bh / H;
  const bf16_t* Qb = Q + (int64_t)bh * S * HD;
  const bf16_t* Kb = K + (int64_t)bh * S * HD;
  const bf16_t* Vb = V + (int64_t)bh * S * HD;
  const int n_txt = S - n_img, D = H * HD;
  const int64_t pitch = 3 * (int64_t)D;
  const int nkb = (S + 31) >> 5, nq = (S + KT - 1) / KT;

  for (int i = tid; i < OP_DQ_BYTES / 16; i += NT) *LDS_PTR(f32x4, dqa + 4 * i) = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (tid < 256) sdw[tid] = 0.f;

  u32x4 sq[1], sd[1], so[1];
  float lv = 0.f;
  auto request = [&](int jq) {
    tile_g2r<NT>(sq, Qb, jq * KT, S, tid);
    const int row = tid >> 3, kc = tid & 7, sidx = jq * KT + row;
    const bf16_t* pd = sidx < S ? tok_ptr(dOx, dOc, b, sidx, n_img, n_txt, D, h) : nullptr;
    const bf16_t* po = sidx < S ? tok_ptr(Ox, Oc, b, sidx, n_img, n_txt, D, h) : nullptr;
    sd[0] = pd ? *(const u32x4*)(pd + kc * 8) : (u32x4){0, 0, 0, 0};
    so[0] = (pd && po) ? *(const u32x4*)(po + kc * 8) : (u32x4){0, 0, 0, 0};
    lv = 0.f;
    if (tid < KT && jq * KT + tid < S) lv = lse[(int64_t)bh * S + jq * KT + tid];
  };

#pragma unroll 1
  for (int pass = 0; pass * 8 < nkb; pass++) {
    const int kblk = pass * 8 + wave;
    const bool active = kblk < nkb;                                  // (wave-uniform)
    const int key = kblk * 32 + (lane & 31), keyc = min(key, S - 1);
    const bool ragged_keys = kblk * 32 + 32 > S;                     // (wave-uniform) this wave's block holds padding keys
    __syncthreads();                                                 // R is free: the zero fill / the previous pass's epilogue is done
    bf16x8 kf[4], vf[4], kt[2][2];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      kf[ks] = *(const bf16x8*)(Kb + (int64_t)keyc * HD + ks * 16 + (lane >> 5) * 8);
      vf[ks] = *(const bf16x8*)(Vb + (int64_t)keyc * HD + ks * 16 + (lane >> 5) * 8);
    }
    {   // the wave's 32 K rows as a (swizzled) tile: its transposed fragments are the B operand of the dQ MFMAs
      char* kw = R + wave * 4096;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int c = lane + 64 * i, row = c >> 3, kc = c & 7;
        const u32x4 v = *(const u32x4*)(Kb + (int64_t)min(kblk * 32 + row, S - 1) * HD + kc * 8);
        *LDS_PTR(u32x4, kw + row * 128 + ((kc ^ sw2(row)) << 4)) = v;
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int h8 = 0; h8 < 2; h8++)
#pragma unroll
        for (int db = 0; db < 2; db++) kt[h8][db] = tr_frag_d(kw, 0, h8, db, lane);
    }
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int db = 0; db < 2; db++)
#pragma unroll
      for (int r = 0; r < 16; r++) { dk[db][r] = 0.f; dv[db][r] = 0.f; }
    request(0);
#pragma unroll 1
    for (int jq = 0; jq < nq; jq++) {
      __syncthreads();                                               // everyone has left the previous tile (jq = 0: has taken its K fragments out of R)
      tile_r2s_sw<NT>(sq, qtile, tid);
      tile_r2s_sw<NT>(sd, dotile, tid);
      {   // delta = rowsum(dO * O): 8 lanes hold a row
        float dl = 0.f;
#pragma unroll
        for (int e = 0; e < 4; e++)
          dl += __builtin_bit_cast(float, so[0][e] << 16) * __builtin_bit_cast(float, sd[0][e] << 16) +
                __builtin_bit_cast(float, so[0][e] & 0xffff0000u) * __builtin_bit_cast(float, sd[0][e] & 0xffff0000u);
        dl = sum8(dl);
        if ((tid & 7) == 0) del_s[tid >> 3] = dl;
      }
      if (tid < KT) lse_s[tid] = lv * LOG2E;
      __syncthreads();
      if (jq + 1 < nq) request(jq + 1);
      if (active) {
#pragma unroll 1
        for (int qb = 0; qb < 2; qb++) {
          if (jq * KT + qb * 32 >= S) continue;                      // (wave-uniform) 32 padding queries contribute nothing
          f32x16 sacc, dp;
#pragma unroll
          for (int r = 0; r < 16; r++) { sacc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
          for (int ks = 0; ks < 4; ks++) {
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_d(qtile, qb * 32, ks, lane), kf[ks], sacc, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_d(dotile, qb * 32, ks, lane), vf[ks], dp, 0, 0, 0);
          }
          f32x16 ds;
#pragma unroll
          for (int g = 0; g < 4; g++) {
            const int r0 = qb * 32 + 8 * g + 4 * (lane >> 5);
            const f32x4 l4 = *LDS_PTR(const f32x4, lse_s + r0);
            const f32x4 d4 = *LDS_PTR(const f32x4, del_s + r0);
#pragma unroll
            for (int e = 0; e < 4; e++) {
              const int r = g * 4 + e;
              const float pe = fast_exp2(fmaf(sacc[r], scale * LOG2E, -l4[e]));
              sacc[r] = pe;
              ds[r] = pe * (dp[r] - d4[e]);
            }
          }
          if ((jq + 1) * KT > S) {                                   // padding queries: the last tile only
#pragma unroll
            for (int g = 0; g < 4; g++)
#pragma unroll
              for (int e = 0; e < 4; e++)
                if (jq * KT + qb * 32 + 8 * g + 4 * (lane >> 5) + e >= S) { sacc[g * 4 + e] = 0.f; ds[g * 4 + e] = 0.f; }
          }
          if (ragged_keys && key >= S) {                             // padding keys must not reach dQ (their dK / dV columns are simply not stored)
#pragma unroll
            for (int r = 0; r < 16; r++) ds[r] = 0.f;
          }
          // dS^T into the wave's tile: row = key, 4 consecutive queries per 8-byte write
#pragma unroll
          for (int g = 0; g < 4; g++) {
            const u32x2 w2 = {pack_bf2(ds[g * 4], ds[g * 4 + 1]), pack_bf2(ds[g * 4 + 2], ds[g * 4 + 3])};
            *LDS_PTR(u32x2, dst_w + (lane & 31) * OP_DST + (8 * g + 4 * (lane >> 5)) * 2) = w2;
          }
#pragma unroll
          for (int h8 = 0; h8 < 2; h8++) {
            const bf16x8 pf = pack_frag(sacc, h8), dsf = pack_frag(ds, h8);
#pragma unroll
            for (int db = 0; db < 2; db++) {
              dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_d(dotile, qb * 32, h8, db, lane), pf, dv[db], 0, 0, 0);
              dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_d(qtile, qb * 32, h8, db, lane), dsf, dk[db], 0, 0, 0);
            }
          }
          __builtin_amdgcn_wave_barrier();
          if (dbg & 2) continue;
          const bf16x8 a0 = tr_frag(dst_w, OP_DST, 0, 0, 0, lane), a1 = tr_frag(dst_w, OP_DST, 0, 1, 0, lane);      // lane = query, k-slots = this wave's keys
          // dq: lane = feature 32 db + (lane & 31), register r = query acc_row(r, lane) of the block
          float* dst = dqa + (jq * KT + qb * 32 + 4 * (lane >> 5)) * QK_ROW_F + (lane & 31);
#pragma unroll
          for (int db = 0; db < 2; db++) {
            f32x16 dq;
#pragma unroll
            for (int r = 0; r < 16; r++) dq[r] = 0.f;
            dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, kt[0][db], dq, 0, 0, 0);
            dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, kt[1][db], dq, 0, 0, 0);
            if (dbg & 1) { asm volatile("" ::"v"(dq[0]), "v"(dq[5]), "v"(dq[15])); continue; }
#pragma unroll
            for (int r = 0; r < 16; r++)
              __hip_atomic_fetch_add(LDS_PTR(float, dst + ((r & 3) + 8 * (r >> 2)) * QK_ROW_F + 32 * db), dq[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          __builtin_amdgcn_wave_barrier();                           // (the dS^T tile is rewritten by the next block)
        }
      }
    }
    // ---- dK / dV of this pass: QK-norm / RoPE backward from 16 parked rows at a time ------------------------------------------------------
    __syncthreads();                                                 // every wave has left the last Q / dO tile: R is free
    if (active) {
      float* t16 = (float*)(R + wave * (16 * QK_ROW_F * 4));
      const int k0 = kblk * 32;
      const bool img = k0 < n_img;                                   // wave-uniform: n_img % 32 == 0
      const int tok0 = img ? k0 : k0 - n_img, nvalid = min(32, (img ? n_img : S) - k0);
      const int64_t off = (img ? b * n_img + tok0 : b * n_txt + tok0) * pitch + D + h * HD;
      const bf16_t* xb = (img ? F.qkv_x : F.qkv_c) + off;
      bf16_t* ob = (img ? F.dqkv_x : F.dqkv_c) + off;
      const float* wk = img ? F.wk_x : F.wk_c;
      const float* cs = img ? F.rcos + (int64_t)tok0 * 64 : nullptr;
      const float* sn = img ? F.rsin + (int64_t)tok0 * 64 : nullptr;
      float* sk = sdw + 128 + (img ? 0 : 64);
      acc_to_lds_half(dk, t16, lane, 0);
      qk_bwd_tile<0, 2>(t16, scale, lane, nvalid, xb, ob, pitch, wk, cs, sn, sk);
      __builtin_amdgcn_wave_barrier();
      acc_to_lds_half(dk, t16, lane, 1);
      qk_bwd_tile<2, 4>(t16 - 16 * QK_ROW_F, scale, lane, nvalid, xb, ob, pitch, wk, cs, sn, sk);
      __builtin_amdgcn_wave_barrier();
      acc_to_lds_half(dv, t16, lane, 0);
      rows_from_tile<0, 2>(t16, lane, nvalid, ob + D, pitch);
      __builtin_amdgcn_wave_barrier();
      acc_to_lds_half(dv, t16, lane, 1);
      rows_from_tile<2, 4>(t16 - 16 * QK_ROW_F, lane, nvalid, ob + D, pitch);
    }
  }
  // ---- dQ: straight out of the accumulator -----------------------------------------------------------------------------------------------
  __syncthreads();
#pragma unroll 1
  for (int blk = wave; blk < nkb; blk += 8) {
    const int q0 = blk * 32;
    const bool img = q0 < n_img;
    const int tok0 = img ? q0 : q0 - n_img, nvalid = min(32, (img ? n_img : S) - q0);
    const int64_t off = (img ? b * n_img + tok0 : b * n_txt + tok0) * pitch + h * HD;
    qk_bwd_tile(dqa + q0 * QK_ROW_F, scale, lane, nvalid, (img ? F.qkv_x : F.qkv_c) + off, (img ? F.dqkv_x : F.dqkv_c) + off, pitch, img ? F.wq_x : F.wq_c,
                img ? F.rcos + (int64_t)tok0 * 64 : nullptr, img ? F.rsin + (int64_t)tok0 * 64 : nullptr, sdw + (img ? 0 : 64));
  }
  __syncthreads();
  if (tid < 256 && sdw[tid] != 0.f) atomicAdd(F.dw + ((tid >> 6) & 1) * 128 + (tid >> 7) * 64 + (tid & 63), sdw[tid]);      // [wq_x | wk_x | wq_c | wk_c]
}

#endif

#ifdef MMDIT_PROBES
// ------------------------------------------------------------------------------------------------
// backward dK/dV, DE-PHASED (round 4 experiment, probes build only: MMDIT_ATTN_DKV_DP=1).  Same arithmetic and fragment conventions as
// attn_bwd_dkv_kernel above, another schedule.  MEASURED SLOWER than that kernel (backward of one block, S = 410, 64 x 12 heads, same box:
// 264.9 vs 248.5 us; correct: the attention tests pass with it), so it is not the product path.  Ablations of this kernel (same box, backward =
// dQ kernel ~105 us + this): full 268; without the stagger 272; no exponentials 262; no MFMAs 243; no LDS fragment reads 263; no DMA in the loop
// 260; no MFMAs + no LDS reads + no exponentials 188 -- the pieces are small and nearly additive, and ~80 us remain without them: the fused
// QK-norm / RoPE epilogue (HBM-bound row traffic, one workgroup per CU: nothing to overlap with), the launch and first-tile latency of six
// rounds of workgroups, barriers and the non-transcendental VALU work.  De-phasing attacks MFMA / VALU serialisation, which is not where this
// kernel's time is.  There a tile's phases -- S / dP MFMAs, softmax-backward VALU work, dV / dK MFMAs, each behind LDS fragment reads -- run
// one after the other, because two workgroup barriers per tile put all eight waves into the same phase (cycle stamps, round 3: 5700 cycles
// per tile against 2048 of matrix-pipe time).  Here:
//  * the unit is a BLOCK of 32 queries (two per 64-query tile) with a V phase (the arithmetic of block b: P, dS packed to bf16 fragments) and
//    an M phase (the dV / dK MFMAs of block b, then the S / dP MFMAs of block b + 1): 16 MFMAs against 16 exponentials + ~50 packed VALU slots;
//  * the waves form two groups (wave >> 2: the two waves of every SIMD are in different groups) and group 1 runs ONE barrier behind group 0:
//    between any two consecutive barriers one group is in a V phase and the other in an M phase, so on every SIMD the matrix pipe of one wave
//    runs beside the VALU work of the other (the 8-phase GEMM's trick, csrc/gemm8p.hip);
//  * Q / dO tiles (+ their lse / delta rows) go global -> LDS by LDS-DMA into a 4-stage ring (swizzle sw2 on the source address: the layout
//    tile_r2s_sw writes), two tiles ahead, with counted vmcnt waits; no staging registers, no ds_write.
// Hazards (barrier intervals: group 0 has V(b) in interval 2b and M(b) in 2b + 1, group 1 one later).  Tile j is read from M(2j - 1) (S / dP of
// its first block) to M(2j + 1) (dV / dK of its second), i.e. until interval 4j + 4: its stage is refilled (tile j + 4) in M(2j + 2), interval
// 4j + 5 / 4j + 6.  All LDS fragment reads sit in the M phases (row fragments of block b + 1 first, transposed fragments of block b + 1 behind
// the dV / dK MFMAs of block b); tile j is first read at the start of M(2j - 1), and the wait for it sits at the end of V(2j - 2) of every
// wave (interval <= 4j - 3), two barriers earlier.
// ------------------------------------------------------------------------------------------------
constexpr int DKV_ST = 4;                            // ring stages (a power of two)
constexpr int DKV_STB = 2 * KT * 128 + 2 * KT * 4;   // 16896 B: Q tile, dO tile, lse row, delta row
template <bool FUSE> constexpr int dkv_lds_bytes() { return FUSE && qk_lds_bytes<8>() > DKV_ST * DKV_STB ? qk_lds_bytes<8>() : DKV_ST * DKV_STB; }

__device__ __forceinline__ void attn_glds4(const void* gptr, uint32_t lds_dst_) {    // 4 bytes per lane: 256 B per instruction
  const uint32_t lds_dst = __builtin_amdgcn_readfirstlane(lds_dst_);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gptr), "s"(lds_dst) : "memory", "m0");
}

template <typename TG, bool FUSE, bool TRACE = false>
__global__ __launch_bounds__(512) void attn_bwd_dkv_dp_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
                                                              const bf16_t* __restrict__ dOx, const bf16_t* __restrict__ dOc,
                                                              const float* __restrict__ lse, const float* __restrict__ delta,
                                                              int BH, int H, int S, int n_img, float scale, TG* __restrict__ dK, TG* __restrict__ dV, QkFuse F = QkFuse(),
                                                              unsigned long long* __restrict__ trace = nullptr) {
  constexpr int NW = 8;
  // TRACE (tools/probes/attn_bwd_trace.py --dp): lane 0 of every wave stamps the cycle counter: 0 start, 1 prologue done (first S / dP issued);
  // per block: +0 V-phase arithmetic issued, +1 past the barrier, +2 row reads + DMA issued, +3 dV / dK MFMAs issued, +4 S / dP MFMAs issued and past
  // the second barrier
  int tpos = 0;
  auto stamp = [&]() {
    if constexpr (TRACE) {
      if (blockIdx.x < 2048 && (threadIdx.x & 63) == 0 && tpos < 72) trace[((int64_t)blockIdx.x * NW + (threadIdx.x >> 6)) * 72 + tpos] = __builtin_readcyclecounter();
      tpos++;
    }
  };
  stamp();
  extern __shared__ __attribute__((aligned(16))) char smem[];      // dkv_lds_bytes<FUSE>(): the ring; afterwards the fused epilogue's tiles
#ifdef MMDIT_DKV_GRP_LSB     // experiment: which waves share a SIMD?
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wave & 1;
#else
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wave >> 2;
#endif
  MMDIT_YOUNG_HALF_PRIO();
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  int ktile, bh;
  map_block((S + 32 * NW - 1) / (32 * NW), BH, ktile, bh);
  const int h = bh % H;
  const int64_t b = bh / H;
  const bf16_t* Qb = Q + (int64_t)bh * S * HD;
  const bf16_t* Kb = K + (int64_t)bh * S * HD;
  const bf16_t* Vb = V + (int64_t)bh * S * HD;
  const int key = ktile * 32 * NW + wave * 32 + (lane & 31);
  const bool active = ktile * 32 * NW + wave * 32 < S;   // wave-uniform: a wave whose 32 keys are all padding only helps with the tile copies
  const int keyc = min(key, S - 1);
  const int n_txt = S - n_img, D = H * HD;
  const int nq = (S + KT - 1) / KT, nblk = (S + 31) / 32;

  // K / V fragments of this lane's key: requested first (asm: the compiler must not put a vmcnt(0) of its own in front of their first use --
  // it cannot see the DMA queue); the first tile's counted wait covers them, the empty asm behind it carries the dependence
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ks++) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(kf[ks]) : "v"(Kb + (int64_t)keyc * HD + ks * 16 + (lane >> 5) * 8) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(vf[ks]) : "v"(Vb + (int64_t)keyc * HD + ks * 16 + (lane >> 5) * 8) : "memory");
  }
  // this lane's 16 bytes of a Q / dO tile: row 8 * wave + (lane >> 3), LDS slot lane & 7 holds chunk slot ^ sw2(row); rows past the end re-read
  // the last query (masked in the ragged block).  Wave 0 also copies the tile's 64 lse and delta values (4 bytes per lane).
  const int rl = 8 * wave + (lane >> 3), dcol = ((lane & 7) ^ sw2(rl)) * 8;
  auto issue = [&](int t) {
    const uint32_t base = lds0 + (t & (DKV_ST - 1)) * DKV_STB;
    const int row = min(t * KT + rl, S - 1);
    attn_glds16(Qb + (int64_t)row * HD + dcol, base + wave * 1024);
    attn_glds16(tok_ptr(dOx, dOc, b, row, n_img, n_txt, D, h) + dcol, base + KT * 128 + wave * 1024);
    if (wave == 0) {
      const int64_t r1 = (int64_t)bh * S + min(t * KT + lane, S - 1);
      attn_glds4(lse + r1, base + 2 * KT * 128);
      attn_glds4(delta + r1, base + 2 * KT * 128 + KT * 4);
    }
  };
  auto wait_tiles = [&](int n) {      // all but the n youngest tiles of this wave's requests have landed (n = 0, 1, 2)
    if (wave == 0) {
      if (n >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else if (n == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (n >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else if (n == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
#pragma unroll
  for (int t = 0; t < DKV_ST - 1; t++)
    if (t < nq) issue(t);

  f32x16 dk[2], dv[2];
#pragma unroll
  for (int db = 0; db < 2; db++)
#pragma unroll
    for (int r = 0; r < 16; r++) { dk[db][r] = 0.f; dv[db][r] = 0.f; }
  // per-lane LDS offsets of the fragment reads (tile / block / k-step parts are immediates: sw2 only involves row bits 1..3)
  uint32_t rowoff[4], troff[2][2];
  {
    const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) rowoff[ks] = l31 * 128 + (((ks * 2 + hi) ^ sw2(l31)) << 4);
    const int row0 = 4 * hi + ((lane & 15) >> 2);
#pragma unroll
    for (int cb = 0; cb < 2; cb++) {
      const int cbyte = (cb * 32 + 16 * ((lane >> 4) & 1) + (lane & 3) * 4) * 2;
      troff[cb][0] = row0 * 128 + ((((cbyte >> 4) ^ sw2(row0)) << 4) | (cbyte & 15));
      troff[cb][1] = (row0 + 8) * 128 + ((((cbyte >> 4) ^ sw2(row0 + 8)) << 4) | (cbyte & 15));
    }
  }
  const float c = scale * LOG2E;
  f32x16 s, dp;
  bf16x8 pf[2], dsf[2];
  constexpr f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto tile_of = [&](int bq) { return smem + ((bq >> 1) & (DKV_ST - 1)) * DKV_STB; };
  // fragment registers of an M phase: ALL LDS reads of the phase are requested before its first MFMA (the compiler's own order -- read two,
  // wait, multiply -- left ~600 cycles of LDS latency per phase exposed, during which the SIMD's matrix pipe idled: the other wave is in
  // its V phase)
  bf16x8 fq[4], fo[4];       // row fragments of Q / dO (block bq + 1)
  s16x4 tq4[2][2][2], to4[2][2][2];   // transposed fragments of Q / dO (block bq): [h8][db][low | high half]
  auto read_rows = [&](int bq) {
#ifdef MMDIT_DKV_NOLDS           // ablation: no LDS fragment reads (wrong results)
#pragma unroll
    for (int ks = 0; ks < 4; ks++) { fq[ks] = kf[ks]; fo[ks] = vf[ks]; }
    return;
#endif
    const char* tq = tile_of(bq) + (bq & 1) * 32 * 128;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      fq[ks] = *LDS_PTR(const bf16x8, tq + rowoff[ks]);
      fo[ks] = *LDS_PTR(const bf16x8, tq + KT * 128 + rowoff[ks]);
    }
  };
  auto read_tr = [&](int bq) {
#ifdef MMDIT_DKV_NOLDS
#pragma unroll
    for (int h8 = 0; h8 < 2; h8++)
#pragma unroll
      for (int db = 0; db < 2; db++)
#pragma unroll
        for (int w = 0; w < 2; w++) { to4[h8][db][w] = (s16x4){1, 2, 3, (short)bq}; tq4[h8][db][w] = (s16x4){4, 3, 2, (short)bq}; }
    return;
#endif
    const char* tq = tile_of(bq) + (bq & 1) * 32 * 128;
#pragma unroll
    for (int h8 = 0; h8 < 2; h8++)
#pragma unroll
      for (int db = 0; db < 2; db++)
#pragma unroll
        for (int w = 0; w < 2; w++) {
          to4[h8][db][w] = lds_tr16(tq + h8 * 16 * 128 + KT * 128 + troff[db][w]);
          tq4[h8][db][w] = lds_tr16(tq + h8 * 16 * 128 + troff[db][w]);
        }
  };
  auto mfma_sdp = [&]() {        // S = Q K^T, dP = dO V^T of the block whose row fragments were read (rows = queries, lane = key)
#ifdef MMDIT_DKV_NOMFMA          // ablation: no MFMAs (wrong results)
#pragma unroll
    for (int ks = 0; ks < 4; ks++) { s[ks] = (float)fq[ks][0]; dp[ks] = (float)fo[ks][0]; }
    return;
#endif
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[ks], kf[ks], ks == 0 ? zero16 : s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fo[ks], vf[ks], ks == 0 ? zero16 : dp, 0, 0, 0);
    }
  };
  auto softmax_bwd = [&](int bq) {     // P = exp2(S c - lse), dS = P (dP - delta): packed fp32 pairs; -> bf16 fragments
    const char* tl = tile_of(bq) + 2 * KT * 128;
    const int qb = bq & 1, hi = lane >> 5;
    f32x16 ds;
    f32x4 lq[4], dq4[4];       // all eight (broadcast) reads first: one exposed LDS latency per phase instead of one per group
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const int r0 = qb * 32 + 8 * g + 4 * hi;
      lq[g] = *LDS_PTR(const f32x4, tl + r0 * 4);
      dq4[g] = *LDS_PTR(const f32x4, tl + KT * 4 + r0 * 4);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const f32x4 l4 = lq[g], d4 = dq4[g];
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        const int r = g * 4 + e;
        const f32x2 sv = {s[r], s[r + 1]}, dpv = {dp[r], dp[r + 1]};
        const f32x2 nl = (f32x2){l4[e], l4[e + 1]} * (f32x2){-LOG2E, -LOG2E};
        const f32x2 t = __builtin_elementwise_fma(sv, (f32x2){c, c}, nl);
#ifdef MMDIT_DKV_NOEXP          // ablation: no exponentials (wrong results)
        const f32x2 pv = t;
#else
        const f32x2 pv = {fast_exp2(t[0]), fast_exp2(t[1])};
#endif
        const f32x2 dsv = pv * (dpv - (f32x2){d4[e], d4[e + 1]});
        s[r] = pv[0]; s[r + 1] = pv[1];
        ds[r] = dsv[0]; ds[r + 1] = dsv[1];
      }
    }
    // Padding QUERIES exist in the last block only (their tile rows are copies of the last query): P and dS are zeroed there in a
    // wave-uniform branch.  A lane whose key is padding works on the clamped last key and its dK / dV column is never stored.
    if ((bq + 1) * 32 > S) {
#pragma unroll
      for (int g = 0; g < 4; g++)
#pragma unroll
        for (int e = 0; e < 4; e++)
          if (bq * 32 + 8 * g + 4 * hi + e >= S) { s[g * 4 + e] = 0.f; ds[g * 4 + e] = 0.f; }
    }
#pragma unroll
    for (int h8 = 0; h8 < 2; h8++) { pf[h8] = pack_frag(s, h8); dsf[h8] = pack_frag(ds, h8); }
  };
  auto mfma_dkv = [&]() {        // dV^T += dO^T P, dK^T += Q^T dS of the block whose transposed fragments were read
#ifdef MMDIT_DKV_NOMFMA
#pragma unroll
    for (int h8 = 0; h8 < 2; h8++)
#pragma unroll
      for (int db = 0; db < 2; db++) { dv[db][h8] += (float)to4[h8][db][0][0] + (float)to4[h8][db][1][0] + (float)pf[h8][0]; dk[db][h8] += (float)tq4[h8][db][0][0] + (float)tq4[h8][db][1][0] + (float)dsf[h8][0]; }
    return;
#endif
#pragma unroll
    for (int h8 = 0; h8 < 2; h8++)
#pragma unroll
      for (int db = 0; db < 2; db++) {
        const s16x4 olo = to4[h8][db][0], ohi = to4[h8][db][1], qlo = tq4[h8][db][0], qhi = tq4[h8][db][1];
        const s16x8 of = {olo[0], olo[1], olo[2], olo[3], ohi[0], ohi[1], ohi[2], ohi[3]};
        const s16x8 qf = {qlo[0], qlo[1], qlo[2], qlo[3], qhi[0], qhi[1], qhi[2], qhi[3]};
        dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, of), pf[h8], dv[db], 0, 0, 0);
        dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qf), dsf[h8], dk[db], 0, 0, 0);
      }
  };
  auto bar = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); };

  wait_tiles(min(2, nq - 1));
  asm volatile("" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]), "+v"(vf[0]), "+v"(vf[1]), "+v"(vf[2]), "+v"(vf[3]));
  bar();                                   // tile 0 has landed for every wave
  if (active) { read_rows(0); mfma_sdp(); read_tr(0); }
  stamp();
#ifndef MMDIT_DKV_NOSTAGGER    // (experiment: both groups in the same phase -- the same code as a lockstep kernel)
  if (grp == 1) bar();                     // group 1 runs one barrier behind from here on
#endif
  for (int bq = 0; bq < nblk; bq++) {
    // ---- V phase: the arithmetic of block bq (pure VALU work); at its end the wait for tile (bq >> 1) + 1, first read in the NEXT M phase
    if (active) softmax_bwd(bq);
    stamp();
#ifdef MMDIT_DKV_NODMA
    if (bq == 0) wait_tiles(0);
#else
    if (!(bq & 1)) wait_tiles(max(0, min(1, nq - 2 - (bq >> 1))));
#endif
    __builtin_amdgcn_s_barrier();
    stamp();
    // ---- M phase.  ALL LDS fragment traffic lives here, between the MFMAs (the V phase is pure VALU work): the row fragments of block
    // bq + 1 are requested first and arrive under the dV / dK MFMAs of block bq (whose transposed fragments were requested in the previous
    // M phase); the transposed fragments of block bq + 1 are requested behind those MFMAs and arrive under the S / dP MFMAs and the V phase.
    if (active && bq + 1 < nblk) read_rows(bq + 1);
    const int t2 = (bq >> 1) + DKV_ST - 1;
#ifndef MMDIT_DKV_NODMA          // (ablation: the loop re-reads the first three tiles)
    if (!(bq & 1) && t2 < nq) issue(t2);   // refill the stage of tile (bq >> 1) - 1: every wave left it two barriers ago
#endif
    stamp();
    if (active) {
      __builtin_amdgcn_sched_barrier(0);
      mfma_dkv();
      __builtin_amdgcn_sched_barrier(0);
      stamp();
      if (bq + 1 < nblk) { read_tr(bq + 1); __builtin_amdgcn_sched_barrier(0); mfma_sdp(); }
    } else stamp();
    __builtin_amdgcn_s_barrier();
    stamp();
  }
#ifndef MMDIT_DKV_NOSTAGGER
  if (grp == 0) bar();                     // rejoin
#endif
  __syncthreads();                         // every wave has left the ring
  if constexpr (FUSE) {
    // dK rows -> gradient of the raw k projection (qk_bwd_tile), dV rows into the v part of the same output rows
    float* sdw = (float*)(smem + NW * QK_WAVE_BYTES);             // [image | text][64]
    if (tid < 128) sdw[tid] = 0.f;
    __syncthreads();
    if (active) {
      const int k0 = ktile * 32 * NW + wave * 32;
      const bool img = k0 < n_img;                                   // wave-uniform: n_img % 32 == 0
      const int tok0 = img ? k0 : k0 - n_img, nvalid = min(32, (img ? n_img : S) - k0);
      const int64_t pitch = 3 * (int64_t)D, off = (img ? b * n_img + tok0 : b * n_txt + tok0) * pitch + D + h * HD;
      bf16_t* ob = (img ? F.dqkv_x : F.dqkv_c) + off;
      float* tile = (float*)(smem + wave * QK_WAVE_BYTES);
      acc_to_lds(dk, tile, lane);
      qk_bwd_tile(tile, scale, lane, nvalid, (img ? F.qkv_x : F.qkv_c) + off, ob, pitch, img ? F.wk_x : F.wk_c,
                  img ? F.rcos + (int64_t)tok0 * 64 : nullptr, img ? F.rsin + (int64_t)tok0 * 64 : nullptr, sdw + (img ? 0 : 64));
      __builtin_amdgcn_wave_barrier();
      acc_to_lds(dv, tile, lane);
      rows_from_tile(tile, lane, nvalid, ob + D, pitch);
    }
    __syncthreads();
    if (tid < 128 && sdw[tid] != 0.f) atomicAdd(F.dw + (tid >> 6) * 128 + 64 + (tid & 63), sdw[tid]);   // [. | wk_x | . | wk_c]
  } else if (key < S) {
    TG* pk = dK + ((int64_t)bh * S + key) * HD;
    TG* pv = dV + ((int64_t)bh * S + key) * HD;
#pragma unroll
    for (int db = 0; db < 2; db++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        float k4[4] = {dk[db][g * 4] * scale, dk[db][g * 4 + 1] * scale, dk[db][g * 4 + 2] * scale, dk[db][g * 4 + 3] * scale};
        float v4[4] = {dv[db][g * 4], dv[db][g * 4 + 1], dv[db][g * 4 + 2], dv[db][g * 4 + 3]};
        st4(pk + db * 32 + 8 * g + 4 * (lane >> 5), k4);
        st4(pv + db * 32 + 8 * g + 4 * (lane >> 5), v4);
      }
  }
}
#endif   // MMDIT_PROBES

// waves (32 queries / keys each) per workgroup: all waves of a workgroup share one stream of 64-row K/V (or Q/dO) tiles, so
// 8 waves cut the tile copies and barriers per (batch, head) from 7x to 2x (measured at S = 410: forward 115 -> 70 us,
// backward 372 -> 271 us).  7-wave workgroups would pad S = 410 less (448 instead of 512 rows) but measured slower (forward 96 vs
// 70 us: 448 threads copy a 512-chunk tile in two unbalanced passes).  MMDIT_ATTN_NW = 2 | 4 | 6 | 7 | 8 overrides for A/B runs.
int attn_waves(int S) {
  (void)S;
#ifdef MMDIT_PROBES
  static const char* e = getenv("MMDIT_ATTN_NW");
  if (e) return atoi(e);
#endif
  return 8;
}

}  // namespace

extern "C" int mmdit_attn_fwd(const void* Q, const void* K, const void* V, int batch, int heads, int S, int n_img, float scale, int mode,
                              void* Ox, void* Oc, float* lse, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(Q && K && V && Ox && lse && batch > 0 && heads > 0 && S > 0 && n_img > 0 && n_img <= S);
  MMDIT_CHECK_ARG(Oc || n_img == S);
  hipStream_t s = (hipStream_t)stream;
  const int nw = attn_waves(S);
#define MMDIT_FWD(NW, OR) hipLaunchKernelGGL((attn_fwd_kernel<NW, OR>), dim3(((S + 32 * NW - 1) / (32 * NW)) * batch * heads), dim3(NW * 64), 0, s, (const bf16_t*)Q, (const bf16_t*)K, \
                                             (const bf16_t*)V, batch * heads, heads, S, n_img, scale, (bf16_t*)Ox, (bf16_t*)Oc, lse)
  if (mode == 1) MMDIT_FWD(2, true);                 // the reference's rounding points (parity mode)
  else if (mode != 0) return MMDIT_ERR_ARG;
#ifdef MMDIT_PROBES                                   // experiments: the register-staged kernel at a forced workgroup width (MMDIT_ATTN_DMA=0 / MMDIT_ATTN_NW)
  else if ((getenv("MMDIT_ATTN_DMA") && atoi(getenv("MMDIT_ATTN_DMA")) == 0) || getenv("MMDIT_ATTN_NW")) {
    if (nw == 8) MMDIT_FWD(8, false);
    else if (nw == 7) MMDIT_FWD(7, false);
    else if (nw == 6) MMDIT_FWD(6, false);
    else if (nw == 4) MMDIT_FWD(4, false);
    else MMDIT_FWD(2, false);
  }
#endif
#define MMDIT_FWD_DMA(NW, NS) hipLaunchKernelGGL((attn_fwd_dma_kernel<0, false, NW, NS>), dim3(((S + 32 * NW - 1) / (32 * NW)) * batch * heads), dim3(64 * NW), 0, s, (const bf16_t*)Q, \
                                                (const bf16_t*)K, (const bf16_t*)V, batch * heads, heads, S, n_img, scale, (bf16_t*)Ox, (bf16_t*)Oc, lse)
#ifdef MMDIT_PROBES                                   // experiments: waves per workgroup / ring depth of the DMA kernel (MMDIT_ATTN_FWD_GEO = 84 | 83 | 43 | 42 | 44)
  else if (mmdit_exp_env("MMDIT_ATTN_FWD_GEO") && atoi(mmdit_exp_env("MMDIT_ATTN_FWD_GEO")) != 84) {
    switch (atoi(mmdit_exp_env("MMDIT_ATTN_FWD_GEO"))) {
      case 83: MMDIT_FWD_DMA(8, 3); break;
      case 44: MMDIT_FWD_DMA(4, 4); break;
      case 43: MMDIT_FWD_DMA(4, 3); break;
      case 42: MMDIT_FWD_DMA(4, 2); break;
      default: return MMDIT_ERR_ARG;
    }
  }
#endif
  else MMDIT_FWD_DMA(8, 4);
#undef MMDIT_FWD_DMA
#undef MMDIT_FWD
  return mmdit_launch_status();
}


extern "C" int mmdit_attn_fwd_mx(const void* Q, const void* K, const void* V, int batch, int heads, int S, int n_img, float scale,
                                 void* Ox_fp8, void* Oc_fp8, void* scales_x, void* scales_c, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(Q && K && V && Ox_fp8 && scales_x && batch > 0 && heads > 0 && S > 0 && n_img > 0 && n_img <= S);
  MMDIT_CHECK_ARG((Oc_fp8 && scales_c) || n_img == S);
  hipLaunchKernelGGL((attn_fwd_dma_kernel<0, true>), dim3(((S + 255) / 256) * batch * heads), dim3(512), 0, (hipStream_t)stream, (const bf16_t*)Q, (const bf16_t*)K,
                     (const bf16_t*)V, batch * heads, heads, S, n_img, scale, (bf16_t*)Ox_fp8, (bf16_t*)Oc_fp8, nullptr, (unsigned char*)scales_x, (unsigned char*)scales_c);
  return mmdit_launch_status();
}

#ifdef MMDIT_PROBES
// measurement aid (tools/probes/attn_ablate.py; not part of include/mmdit_hip.h): the forward kernel with parts of its loop removed
extern "C" int mmdit_probe_attn_fwd_dbg(const void* Q, const void* K, const void* V, int batch, int heads, int S, int n_img, float scale,
                                        void* Ox, void* Oc, float* lse, int dbg, mmdit_stream_t stream) {
#define MMDIT_DBG(D) case D: hipLaunchKernelGGL(attn_fwd_dma_kernel<D>, dim3(((S + 255) / 256) * batch * heads), dim3(512), 0, (hipStream_t)stream, (const bf16_t*)Q, \
                                                 (const bf16_t*)K, (const bf16_t*)V, batch * heads, heads, S, n_img, scale, (bf16_t*)Ox, (bf16_t*)Oc, lse); break;
  switch (dbg) {
    MMDIT_DBG(0) MMDIT_DBG(1) MMDIT_DBG(2) MMDIT_DBG(4) MMDIT_DBG(8) MMDIT_DBG(16) MMDIT_DBG(32) MMDIT_DBG(33) MMDIT_DBG(6) MMDIT_DBG(39) MMDIT_DBG(24) MMDIT_DBG(57) MMDIT_DBG(63) MMDIT_DBG(64) MMDIT_DBG(191) MMDIT_DBG(128)
    default: return MMDIT_ERR_ARG;
  }
#undef MMDIT_DBG
  return mmdit_launch_status();
}

#endif

#ifdef MMDIT_PROBES
// measurement aid (tools/probes/attn_bwd_trace.py; not declared in include/mmdit_hip.h): the dK/dV kernel (plain or fused epilogue) with per-phase
// cycle stamps, trace = 2048 workgroups x 8 waves x 72 slots of 8 bytes.  delta must already hold rowsum(dO * O) (run mmdit_attn_bwd first).
extern "C" int mmdit_probe_attn_bwd_dkv_trace(const void* Q, const void* K, const void* V, const void* dOx, const void* dOc, const float* lse, const float* delta,
                                              int batch, int heads, int S, int n_img, float scale, void* dK, void* dV, void* trace, mmdit_stream_t stream) {
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<8, bf16_t, false, true>), dim3(((S + 255) / 256) * batch * heads), dim3(512), 0, (hipStream_t)stream, (const bf16_t*)Q, (const bf16_t*)K,
                     (const bf16_t*)V, (const bf16_t*)dOx, (const bf16_t*)dOc, lse, delta, batch * heads, heads, S, n_img, scale, (bf16_t*)dK, (bf16_t*)dV, QkFuse(),
                     (unsigned long long*)trace);
  return mmdit_launch_status();
}
#endif

#ifdef MMDIT_PROBES
// the same stamps for the de-phased kernel (plain epilogue): trace = 2048 workgroups x 8 waves x 72 slots
extern "C" int mmdit_probe_attn_bwd_dkv_dp_trace(const void* Q, const void* K, const void* V, const void* dOx, const void* dOc, const float* lse, const float* delta,
                                                 int batch, int heads, int S, int n_img, float scale, void* dK, void* dV, void* trace, mmdit_stream_t stream) {
  static unsigned long long raised = 0;
  if (!mmdit_device_once(raised)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_dp_kernel<bf16_t, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, dkv_lds_bytes<false>());
    if (e != hipSuccess) return (int)e;
    mmdit_device_mark(raised);
  }
  hipLaunchKernelGGL((attn_bwd_dkv_dp_kernel<bf16_t, false, true>), dim3(((S + 255) / 256) * batch * heads), dim3(512), dkv_lds_bytes<false>(), (hipStream_t)stream, (const bf16_t*)Q,
                     (const bf16_t*)K, (const bf16_t*)V, (const bf16_t*)dOx, (const bf16_t*)dOc, lse, delta, batch * heads, heads, S, n_img, scale, (bf16_t*)dK, (bf16_t*)dV, QkFuse(),
                     (unsigned long long*)trace);
  return mmdit_launch_status();
}
#endif

extern "C" int mmdit_attn_bwd_qk(const void* Q, const void* K, const void* V, const void* Ox, const void* Oc, const void* dOx, const void* dOc,
                                 const float* lse, float* delta, int batch, int heads, int S, int n_img, float scale,
                                 const void* qkv_x, const void* qkv_c, const float* wq_x, const float* wk_x, const float* wq_c, const float* wk_c,
                                 const float* rope_cos, const float* rope_sin, void* dqkv_x, void* dqkv_c, float* dw, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(Q && K && V && Ox && dOx && lse && delta && batch > 0 && heads > 0 && S > 0 && n_img > 0 && n_img <= S);
  MMDIT_CHECK_ARG(Oc || n_img == S);
  MMDIT_CHECK_ARG(qkv_x && wq_x && wk_x && rope_cos && rope_sin && dqkv_x && dw && (n_img == S || (qkv_c && wq_c && wk_c && dqkv_c)));
  if (n_img % 32) return MMDIT_ERR_SHAPE;       // a wave's 32 rows must belong to one stream
  const QkFuse F{(const bf16_t*)qkv_x, (const bf16_t*)qkv_c, wq_x, wk_x, wq_c, wk_c, rope_cos, rope_sin, (bf16_t*)dqkv_x, (bf16_t*)dqkv_c, dw};
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(((S + 255) / 256) * batch * heads);
  constexpr int lds = qk_lds_bytes<8>();        // 70144 B: above the 64 KB default cap of dynamic LDS
  static unsigned long long raised = 0;            // one bit per device (the attribute is a per-device property)
  if (!mmdit_device_once(raised)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<8, bf16_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<8, bf16_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
#ifdef MMDIT_PROBES
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_dp_kernel<bf16_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, dkv_lds_bytes<true>());
#endif
    if (e != hipSuccess) return (int)e;             // (positive: a HIP error code, as from mmdit_launch_status)
    mmdit_device_mark(raised);
  }
#ifdef MMDIT_PROBES      // experiment (MMDIT_ATTN_ONEPASS=1): one pass per (batch, head), S <= 416 -- measured slower, see attn_bwd_onepass_kernel
  if (S <= 32 * OP_MAXB && mmdit_exp_env("MMDIT_ATTN_ONEPASS") && atoi(mmdit_exp_env("MMDIT_ATTN_ONEPASS")) == 1) {
    static unsigned long long raised1 = 0;
    if (!mmdit_device_once(raised1)) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_onepass_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, OP_LDS);
      if (e != hipSuccess) return (int)e;
      mmdit_device_mark(raised1);
    }
    hipLaunchKernelGGL(attn_bwd_onepass_kernel, dim3(batch * heads), dim3(512), OP_LDS, s, (const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)V, (const bf16_t*)Ox,
                       (const bf16_t*)Oc, (const bf16_t*)dOx, (const bf16_t*)dOc, lse, batch * heads, heads, S, n_img, scale, F,
                       mmdit_exp_env("MMDIT_OP_DBG") ? atoi(mmdit_exp_env("MMDIT_OP_DBG")) : 0);
    return mmdit_launch_status();
  }
#endif
  hipLaunchKernelGGL((attn_bwd_dq_kernel<8, bf16_t, true>), grid, dim3(512), lds, s, (const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)V, (const bf16_t*)Ox,
                     (const bf16_t*)Oc, (const bf16_t*)dOx, (const bf16_t*)dOc, lse, delta, batch * heads, heads, S, n_img, scale, (bf16_t*)nullptr, F);
#ifdef MMDIT_PROBES      // experiment: the de-phased dK/dV kernel (MMDIT_ATTN_DKV_DP=1; needs a text output gradient when there are text tokens)
  if (mmdit_exp_env("MMDIT_ATTN_DKV_DP") && atoi(mmdit_exp_env("MMDIT_ATTN_DKV_DP")) == 1 && (dOc || n_img == S)) {
    hipLaunchKernelGGL((attn_bwd_dkv_dp_kernel<bf16_t, true>), grid, dim3(512), dkv_lds_bytes<true>(), s, (const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)V, (const bf16_t*)dOx,
                       (const bf16_t*)dOc, lse, delta, batch * heads, heads, S, n_img, scale, (bf16_t*)nullptr, (bf16_t*)nullptr, F);
    return mmdit_launch_status();
  }
#endif
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<8, bf16_t, true>), grid, dim3(512), lds, s, (const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)V, (const bf16_t*)dOx,
                     (const bf16_t*)dOc, lse, delta, batch * heads, heads, S, n_img, scale, (bf16_t*)nullptr, (bf16_t*)nullptr, F);
  return mmdit_launch_status();
}

extern "C" int mmdit_attn_bwd(const void* Q, const void* K, const void* V, const void* Ox, const void* Oc, const void* dOx, const void* dOc,
                              const float* lse, float* delta, int batch, int heads, int S, int n_img, float scale,
                              void* dQ, void* dK, void* dV, int dq_dtype, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(Q && K && V && Ox && dOx && lse && delta && dQ && dK && dV && batch > 0 && heads > 0 && S > 0 && n_img > 0 && n_img <= S);
  MMDIT_CHECK_ARG(Oc || n_img == S);
  hipStream_t s = (hipStream_t)stream;
  // (delta = rowsum(dO * O) is produced by the dQ kernel, which runs first)
  const int nw = attn_waves(S);
#define MMDIT_DKV(NW, TG) hipLaunchKernelGGL((attn_bwd_dkv_kernel<NW, TG>), dim3(((S + 32 * NW - 1) / (32 * NW)) * batch * heads), dim3(NW * 64), 0, s, (const bf16_t*)Q, (const bf16_t*)K, \
                                             (const bf16_t*)V, (const bf16_t*)dOx, (const bf16_t*)dOc, lse, delta, batch * heads, heads, S, n_img, scale, (TG*)dK, (TG*)dV)
#define MMDIT_DQ(NW, TG) hipLaunchKernelGGL((attn_bwd_dq_kernel<NW, TG>), dim3(((S + 32 * NW - 1) / (32 * NW)) * batch * heads), dim3(NW * 64), 0, s, (const bf16_t*)Q, (const bf16_t*)K, \
                                            (const bf16_t*)V, (const bf16_t*)Ox, (const bf16_t*)Oc, (const bf16_t*)dOx, (const bf16_t*)dOc, lse, delta, batch * heads, heads, S, n_img, scale, (TG*)dQ)
  if (dq_dtype == MMDIT_BF16) {
#ifdef MMDIT_PROBES      // experiments: other workgroup widths (MMDIT_ATTN_NW)
    if (nw == 7) { MMDIT_DQ(7, bf16_t); MMDIT_DKV(7, bf16_t); }
    else if (nw == 6) { MMDIT_DQ(6, bf16_t); MMDIT_DKV(6, bf16_t); }
    else if (nw == 4) { MMDIT_DQ(4, bf16_t); MMDIT_DKV(4, bf16_t); }
    else if (nw == 2) { MMDIT_DQ(2, bf16_t); MMDIT_DKV(2, bf16_t); }
    else
#endif
    { MMDIT_DQ(8, bf16_t); MMDIT_DKV(8, bf16_t); }
  } else if (dq_dtype == MMDIT_F32) {
    MMDIT_DQ(2, float);
    MMDIT_DKV(2, float);
  } else return MMDIT_ERR_DTYPE;
#undef MMDIT_DQ
#undef MMDIT_DKV
  return mmdit_launch_status();
}
