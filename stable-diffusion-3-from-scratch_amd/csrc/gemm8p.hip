// 256 x 256 x 64 bf16 GEMM on the DE-PHASED ("8-phase") main loop -- cdna_hip_programming.md 5 "The 256^2 8-phase template", T2, T3+T4; built and
// measured as tools/probes/gemm8p.hip first (8192^3 on random operands: 1.51 PFLOP/s against hipBLASLt's 1.46 and 1.33 for the
// one-barrier-per-K-step kernels of gemm_lean.hip on the same box; 2260 shader cycles per K tile = 96 % of the matrix-pipe floor).
//
// Structure.  8 waves = 2 groups (wr = wave >> 2) x 4 column waves (wc = wave & 3).  Group 1 runs ONE s_barrier behind group 0, so between any
// two consecutive barriers one group issues its LDS fragment reads + LDS-DMA while the other issues MFMAs: on every SIMD the matrix pipe of
// one wave runs beside the LDS / VMEM issue of the other.  A K tile (64 deep) is four phases, one quadrant of the wave's 128 x 64 outputs each:
//     P1  read B0 (4 fragments), A0 (8)   C00 += A0 B0          P3  read A1 (8)    C11 += A1 B1
//     P2  read B1 (4)                     C01 += A0 B1          P4  (B0 kept)      C10 += A1 B0
// with v_mfma_f32_16x16x32_bf16 (16 per phase): on random operands the 16x16x32 form runs 8 % more FLOPs per joule than 32x32x16 (a quarter of
// the accumulator traffic per FLOP), and this loop is clock-(power-)bound, not issue-bound.
// LDS image of a K tile: four half-tiles A0 A1 B0 B1 of 16 KB (128 rows x 64 k); half-tile A[q] holds the 64-row halves q of BOTH wave rows
// (local row wr * 64 + c  <->  tile row wr * 128 + q * 64 + c), B[q] the 32-column halves q of the four wave columns -- so a wave's outputs are a
// contiguous 128 x 64 block (epilogues store whole lines) while every half-tile has exactly ONE reading phase per K tile.  Two K tiles are
// resident (128 KB) + 32 KB of epilogue staging = the whole 160 KB.
// Every phase issues the two LDS-DMA instructions (per wave) of one half-tile, almost two K tiles ahead:
//     P1(t): A1(t+1)     P2(t): A0(t+2)     P3(t): B0(t+2)     P4(t): B1(t+2)
// and the loop has ONE counted wait per K tile, s_waitcnt vmcnt(6) in P4 (three half-tiles stay in flight), never 0.
// Hazards (the guide's rules for two groups staggered by a barrier): data waited for in phase p is read in phase p+1 or later; a half-tile is
// restaged >= 1 phase after the phase whose reads were retired (lgkmcnt(0)) BEFORE that phase's first barrier.
// Layouts: row-major operands as 128-byte rows, 16-byte chunk c of local row r at chunk c ^ ((r >> 1) & 7) (conflict-free ds_read_b128 of
// 16-row fragments); k-major operands (data / weight gradients) as 64 k-rows of 256 bytes, chunk c of k-row k at c ^ f(k),
// f(k) = ((k & 3) << 2) ^ (((k >> 3) & 1) << 1) (conflict-free ds_read_b64_tr_b16 of the 4 x 16 blocks a 16x16x32 operand is made of).
#include "gemm_tile.h"

using namespace gemm;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;

namespace {

// tile MT x 256 (MT = 256 or 320).  A half-tile holds the q-th quadrant rows (QR = MT / 4 of them) of BOTH wave rows: 2 QR rows x 128 B.
template <int MT> struct Geo {
  static constexpr int QR = MT / 4, FI = QR / 16;          // rows of a wave's quadrant; its 16-row fragments
  static constexpr int HTA = 2 * QR * 128, HTB = 16384;    // bytes of an A / B half-tile
  static constexpr int KBUF = 2 * HTA + 2 * HTB;           // one K tile: A0 A1 B0 B1
  static constexpr int XA0 = 0, XA1 = HTA, XB0 = 2 * HTA, XB1 = 2 * HTA + HTB;
  static constexpr int PA = 2 * QR / 8;                    // 1-KiB DMA pieces of an A half-tile (16 / 20): waves 0..PA-17 issue 3, the others 2
  static constexpr int PAW = (PA + 7) / 8;
  // epilogue staging: 8 waves x 4 KiB -- behind the two K tiles (256: 128 + 32 = the whole 160 KB), inside the idle operand buffers (320: 144 KB)
  static constexpr int STAGE0 = MT == 256 ? 2 * KBUF : 0;
  static constexpr int SMEM = MT == 256 ? 2 * KBUF + 8 * EP32_WAVE_BYTES : 2 * KBUF;
  static constexpr int NB32 = MT / 64;                     // 32-row blocks of a wave's rows (4 / 5)
};

enum Epi8 { EPI_BF16 = 0, EPI_SWIGLU = 1, EPI_QK = 2, EPI_F32 = 3, EPI_SWIGLU_BWD = 4, EPI_F32R = 5 };

#define SWZ_R(r) (((r) >> 1) & 7)
#define SWZ_K(k) ((((k) & 3) << 2) ^ ((((k) >> 3) & 1) << 1))

// accumulators: acc[qm][qn][i * 2 + j], i < FI, j < 2: C^T fragments of 16 x 16 (lane & 15 = output row, 4 consecutive columns per lane)
template <int MT> struct AccT { f32x4 a[2][2][2 * Geo<MT>::FI]; };

// the 4-value group q (0..7) of 32-row block i32 of a wave's (MT / 2) x 64 outputs: rows i32 * 32 + (q >> 2) * 16 + (lane & 15),
// columns (q & 3) * 16 + (lane >> 4) * 4 + [0, 4).  (MT = 320: block 2 straddles the two row quadrants of 80.)
template <int MT>
__device__ __forceinline__ f32x4& grp(AccT<MT>& acc, int i32, int q) {
  constexpr int FI = Geo<MT>::FI;
  const int f16 = i32 * 2 + (q >> 2), j16 = q & 3;
  return acc.a[f16 / FI][j16 >> 1][(f16 % FI) * 2 + (j16 & 1)];
}

// ---- epilogues (the arithmetic and the read-back / store side are those of gemm_tile.h; only the write side knows the fragment layout) ------
// bf16 output (+ bias, + SiLU): the wave's 32 x 64 block is converted first, staged as 32 rows x 128 B (chunk c of row r at c ^ (r & 7)) and
// leaves as 8 rows x 128 B per wave instruction
template <int MT, bool SC = false, bool LAUNDER = SC>      // SC (e4m3 operands with per-tensor scales): the accumulators are multiplied by alpha = scale_a * scale_b first
__device__ __forceinline__ void epi8_bf16(AccT<MT>& acc, const Problem& p, const GroupParams& gp, int m0, int n0, int wm, int wn, int lane, char* stage, float alpha = 1.f) {
  bf16_t* C = (bf16_t*)p.C;
  const float* bias = p.bias;
  const int wr = lane & 15, wq = lane >> 4;            // write side: row within the 16-row fragment, 4-column group
  const int rr = lane >> 3, rc = lane & 7;             // read side: row within the 8-row pass, 16-B chunk (8 columns)
  const int colw = n0 + wn * 64;
  float bv[4][4];
  if (bias) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int c = colw + j * 16 + wq * 4;
      ld4(bias + (c < p.N ? c : 0), bv[j]);
      if (c >= p.N) bv[j][0] = bv[j][1] = bv[j][2] = bv[j][3] = 0.f;
    }
  }
#ifdef MMDIT_PROBES      // ablations / experiments of the epilogue (MMDIT_GEMM_DEBUG bits 16, 32, 64): not in the product -- the 320-row variants sit at the
                         // 256-VGPR edge, and code that is never executed still moves their register allocation (in-loop spills)
  if (gp.debug & 64) {      // experiment: no staging -- every lane stores its 4 consecutive columns (8 bytes) straight from the accumulator layout
#pragma unroll
    for (int i = 0; i < Geo<MT>::NB32; i++)
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const f32x4& a = grp(acc, i, q);
        float v[4] = {a[0], a[1], a[2], a[3]};
        if (bias) {
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] += bv[q & 3][e];
        }
        if (gp.act == MMDIT_ACT_SILU) {
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = silu_f(v[e]);
        }
        const int row = m0 + wm * (MT / 2) + i * 32 + (q >> 2) * 16 + wr, col = colw + (q & 3) * 16 + wq * 4;
        if (row < p.M && col < p.N) *(u32x2*)(C + (int64_t)row * p.ldc + col) = (u32x2){pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
      }
    return;
  }
  if (gp.debug & 32) {      // ablation: the global stores alone (no conversion, no staging): 8 rows x 128 B per instruction, as the real epilogue
#pragma unroll
    for (int i = 0; i < Geo<MT>::NB32; i++)
#pragma unroll
      for (int it = 0; it < 4; it++) {
        const f32x4& a = grp(acc, i, it);
        const int row = m0 + wm * (MT / 2) + i * 32 + it * 8 + rr, col = colw + rc * 8;
        if (row < p.M && col < p.N) *(u32x4*)(C + (int64_t)row * p.ldc + col) = __builtin_bit_cast(u32x4, a);
      }
    return;
  }
#endif
#pragma unroll
  for (int i = 0; i < Geo<MT>::NB32; i++) {
    int ln = lane;
    // (per-block address VALU instead of spilled offsets: see epi8_swiglu.  LAUNDER also for the 320-row data-gradient kernel -- 14 reloads -> 1, its launches
    //  2-3 % faster --, NOT for the 320-row forward kernel: there it moves the remaining reloads into the K loop, tools/check_spills.py)
    if constexpr (LAUNDER) asm volatile("" : "+v"(ln));
    const int wr = ln & 15, wq = ln >> 4, rr = ln >> 3, rc = ln & 7;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const f32x4& a = grp(acc, i, q);
      float v[4] = {a[0], a[1], a[2], a[3]};
      if constexpr (SC) {
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] *= alpha;
      }
      if (bias) {
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] += bv[q & 3][e];
      }
      if (gp.act == MMDIT_ACT_SILU) {
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = silu_f(v[e]);
      }
      const int row = (q >> 2) * 16 + wr, chunk = (q & 3) * 2 + (wq >> 1);
      *LDS_PTR(u32x2, stage + row * 128 + ((chunk ^ (row & 7)) << 4) + (wq & 1) * 8) = (u32x2){pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private region: program order is enough
    const int col = colw + rc * 8;
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const int r = it * 8 + rr;
      const u32x4 t = *LDS_PTR(const u32x4, stage + r * 128 + ((rc ^ (r & 7)) << 4));
      const int row = m0 + wm * (MT / 2) + i * 32 + r;
#ifdef MMDIT_PROBES
      if (gp.debug & 16) { if (t[0] == 0x12345678u) *(u32x4*)(C + (int64_t)row * p.ldc + col) = t; continue; }   // ablation: staging without the global stores
#endif
#ifdef MMDIT_PROBES
      if (gp.debug & 1024) { if (row < p.M && col < p.N) __builtin_nontemporal_store(t, (u32x4*)(C + (int64_t)row * p.ldc + col)); continue; }      // experiment: streaming stores
#endif
      if (row < p.M && col < p.N) *(u32x4*)(C + (int64_t)row * p.ldc + col) = t;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next block overwrites the region
  }
}

// SwiGLU backward in the epilogue of the down-projection's data gradient (MMDIT_ACT_SWIGLU_BWD; MLP.py:32 backward / xformers SwiGLU): the tile
// is dh = dY W3 for the hidden columns [n0, n0 + 256).  It is staged as bf16 exactly as epi8_bf16 would store it; every lane then takes 8
// consecutive columns of a row, loads the saved pre-activations g, u of those columns (aux = [g | u], 2 N columns) and writes d[g | u] to C:
// the arithmetic of rowops.hip mlp_act_bwd_body on the ROUNDED dh, i.e. the bits of the GEMM followed by mmdit_swiglu_bwd without the dh round
// trip (322 of that pass's 806 MB at MMDiT-B) and without its launch.  Bias gradient (column sums of d[g | u]): one atomic per column and wave.
template <int MT>
__device__ __forceinline__ void epi8_swiglu_bwd(AccT<MT>& acc, const Problem& p, int m0, int n0, int wm, int wn, int lane, char* stage) {
  constexpr int NB = Geo<MT>::NB32;
  const bf16_t* GU = (const bf16_t*)p.aux;
  bf16_t* D = (bf16_t*)p.C;
  const int wr = lane & 15, wq = lane >> 4;            // write side: row within the 16-row fragment, 4-column group
  const int rr = lane >> 3, rc = lane & 7;             // read side: row within the 8-row pass, 16-B chunk (8 columns)
  const int col = n0 + wn * 64 + rc * 8;
  const bool cok = col < p.N;
  const int row0 = m0 + wm * (MT / 2) + rr;
  // The pre-activation rows of the next 32-row block are requested before the current block is converted (double buffer: 2 x 8 quads).
  // (Requesting all four blocks up front -- 32 KB per wave in flight, the retired blocks parked in packed form -- measured SLOWER, 246 vs 225 us
  //  per launch at MMDiT-B: 30 spilled registers, and the epilogue phases are HBM-bound anyway: 256 CUs x 512 KB per round.)
  // Round 6: the loads are UNCONDITIONAL (rows beyond M read row M - 1, columns beyond N the last 8: values nobody uses) and the stores of the
  // passes are not counted (st8_uncounted): before, every pass ended in s_waitcnt vmcnt(0) (profiles/r06_epilogue_waits.txt).
  u32x4 gq[2][4], uq[2][4];
  const int colc = cok ? col : p.N - 8;
  auto request = [&](int i, int b) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const int row = min(row0 + i * 32 + it * 8, p.M - 1);
      const bf16_t* src = GU + (int64_t)row * p.ld_aux + colc;      // (saved pre-activations at their last use: streaming loads)
      gq[b][it] = __builtin_nontemporal_load((const u32x4*)src);
      uq[b][it] = __builtin_nontemporal_load((const u32x4*)(src + p.N));
    }
  };
  float sg[8], su[8];
#pragma unroll
  for (int e = 0; e < 8; e++) sg[e] = su[e] = 0.f;
  request(0, 0);
#pragma unroll
  for (int i = 0; i < NB; i++) {
    if (i + 1 < NB) request(i + 1, (i + 1) & 1);
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const f32x4& a = grp(acc, i, q);
      const int row = (q >> 2) * 16 + wr, chunk = (q & 3) * 2 + (wq >> 1);
      *LDS_PTR(u32x2, stage + row * 128 + ((chunk ^ (row & 7)) << 4) + (wq & 1) * 8) = (u32x2){pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3])};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private region: program order is enough
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const int r = it * 8 + rr;
      const u32x4 t = *LDS_PTR(const u32x4, stage + r * 128 + ((rc ^ (r & 7)) << 4));
      const int row = row0 + i * 32 + it * 8;
      const u32x4 gw = gq[i & 1][it], uw = uq[i & 1][it];
      float og[8], ou[8];
#pragma unroll
      for (int e = 0; e < 4; e++) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const float d = __builtin_bit_cast(float, h ? t[e] & 0xffff0000u : t[e] << 16);
          const float g = __builtin_bit_cast(float, h ? gw[e] & 0xffff0000u : gw[e] << 16);
          const float u = __builtin_bit_cast(float, h ? uw[e] & 0xffff0000u : uw[e] << 16);
          swiglu_bwd_f(d, g, u, og[2 * e + h], ou[2 * e + h]);
        }
      }
      if (row < p.M && cok) {
#pragma unroll
        for (int e = 0; e < 8; e++) { sg[e] += og[e]; su[e] += ou[e]; }
        bf16_t* dst = D + (int64_t)row * p.ldc + col;
        st8_uncounted(dst, og);
        st8_uncounted(dst + p.N, ou);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next block overwrites the region
  }
  if (p.dbias) {      // the 8 lanes with the same chunk (rr = 0..7) hold partial sums of the same 8 columns
#pragma unroll
    for (int e = 0; e < 8; e++) {
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) { sg[e] += __shfl_xor(sg[e], o, 64); su[e] += __shfl_xor(su[e], o, 64); }
    }
    if (rr == 0 && cok) {
#pragma unroll
      for (int e = 0; e < 8; e++) { atomicAdd(p.dbias + col + e, sg[e]); atomicAdd(p.dbias + p.N + col + e, su[e]); }
    }
  }
}

// QKV projection: the raw q / k columns + Q / K / V in the joint attention layout (gemm_tile.h epilogue_bf16_qk: same read-back side, same arithmetic).
// Round 6: the RoPE factors of a pass (8 rows x 128 B; 16 passes per wave and tile) are REQUESTED ONE PASS AHEAD, before the previous pass's stores.
// Loads and stores of a wave retire through one in-order counter on gfx9 (vmcnt): a load issued behind a store is not back before that store has
// been acknowledged.  For the compiler to emit the exact count, the loop must contain neither a store inside a conditional region (the number in
// flight at the join is then unknown and the wait degrades to "everything but the newest request" -- the stores included) nor a conditional load
// (a merge of old and new registers = copies that wait where the load was issued): FULL tiles are a compile-time variant, the factor loads are
// unconditional (rows beyond M read the table's last row; a stream without rotation reads the norm weights -- 64 valid floats -- and ignores
// them), the v columns have their own loop; C == nullptr (inference: nobody reads the raw q / k columns) skips the raw store.  MX operands, MMDiT-L QKV (75392 x 3072 x 1024): 455 -> 420 us; bf16, MMDiT-B: 153 -> 131 us, the
// wide-slot kernel's time (profiles/r06_epilogue_waits.txt).  What remains is the compute unit's memory path: per tile 512 KB of operands, 213 KB
// of stores and 341 KB of factors (the four waves of a row block read the same table rows).
template <int MT>
__device__ __forceinline__ void epi8_qk(AccT<MT>& acc, const Problem& p, const GroupParams& gp, const QkEpi& e, int m0, int n0, int wm, int wn, int lane, char* stage) {
  bf16_t* C = (bf16_t*)p.C;
  const int colw = n0 + wn * 64;
  const int D = gp.qk_heads * 64, part = colw / D, head = (colw - part * D) >> 6;
  if (colw >= p.N) return;      // (wave-uniform; the staging is wave-private and the epilogue has no barrier)
  bf16_t* obase = part == 0 ? gp.qkQ : part == 1 ? gp.qkK : gp.qkV;
  const int rowt = m0 + wm * (MT / 2);
  const bool rope = e.rcos != nullptr;
  const float* tc = rope ? e.rcos : e.wq;
  const float* ts = rope ? e.rsin : e.wq;
  float w8[8], cb[2][8], sb[2][8];      // the factors of two passes: one in use, one on its way (indices are constants once the loops are unrolled: no copies)
#pragma unroll
  for (int q = 0; q < 8; q++) w8[q] = 0.f;
  if (part < 2) ld8((part == 0 ? e.wq : e.wk) + (lane & 7) * 8, w8);
  // (sample, token) of a row: the block's first row by ONE wave-uniform division; its 32 rows are consecutive and a sample has more than 32 tokens or
  // the rows wrap more than once -- the per-row division then
  auto token_of = [&](int rowb, int b0, int r, int& b, int& n) __attribute__((always_inline)) {
    b = b0; n = rowb - b0 * e.tokens + r;
    if (e.tokens >= 32) { if (n >= e.tokens) { n -= e.tokens; b++; } }
    else { b = (rowb + r) / e.tokens; n = rowb + r - b * e.tokens; }
  };
  auto request = [&](int i, int it, int ln, float (&cn)[8], float (&sn)[8]) __attribute__((always_inline)) {
    const int rowb = rowt + i * 32;
    const int b0 = __builtin_amdgcn_readfirstlane(rowb / e.tokens);
    int b, n;
    token_of(rowb, b0, it * 8 + (ln >> 3), b, n);
    n = rope ? min(n, e.tokens - 1) : 0;
    ld8(tc + (int64_t)n * 64 + (ln & 7) * 8, cn);
    ld8(ts + (int64_t)n * 64 + (ln & 7) * 8, sn);
  };
  auto stage_block = [&](int i, int ln) __attribute__((always_inline)) {
    const int wr = ln & 15, wq = ln >> 4;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const f32x4& a = grp(acc, i, q);
      const int row = (q >> 2) * 16 + wr, chunk = (q & 3) * 2 + (wq >> 1);
      *LDS_PTR(u32x2, stage + row * 128 + ((chunk ^ (row & 7)) << 4) + (wq & 1) * 8) = (u32x2){pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3])};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  if (part == 2) {      // v: the attention operand IS the raw projection -- one store per pass (backward never reads the v columns of C)
#pragma unroll
    for (int i = 0; i < Geo<MT>::NB32; i++) {
      int ln = lane;
      asm volatile("" : "+v"(ln));      // (per-block address VALU instead of spilled offsets: see epi8_swiglu)
      stage_block(i, ln);
      const int rr = ln >> 3, rc = ln & 7;
      const int rowb = rowt + i * 32;
      const int b0 = __builtin_amdgcn_readfirstlane(rowb / e.tokens);
#pragma unroll
      for (int it = 0; it < 4; it++) {
        const int r = it * 8 + rr;
        const u32x4 t = *LDS_PTR(const u32x4, stage + r * 128 + ((rc ^ (r & 7)) << 4));
        int b, n;
        token_of(rowb, b0, r, b, n);
        if (rowb + r < p.M) *(u32x4*)(obase + (((int64_t)b * gp.qk_heads + head) * gp.qk_s_total + e.tok0 + n) * 64 + rc * 8) = t;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    return;
  }
  auto qk_loop = [&](auto full_t, auto raw_t) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_t)::value;
    constexpr bool RAW = decltype(raw_t)::value;      // the raw q / k columns are wanted (C != nullptr: training saves them for the backward; inference passes nullptr)
    request(0, 0, lane, cb[0], sb[0]);
#pragma unroll
    for (int i = 0; i < Geo<MT>::NB32; i++) {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      stage_block(i, ln);
      const int rr = ln >> 3, rc = ln & 7;
      const int col = colw + rc * 8;
      const int rowb = rowt + i * 32;
      const int b0 = __builtin_amdgcn_readfirstlane(rowb / e.tokens);
#pragma unroll
      for (int it = 0; it < 4; it++) {
        const int r = it * 8 + rr;
        const u32x4 t = *LDS_PTR(const u32x4, stage + r * 128 + ((rc ^ (r & 7)) << 4));
        const int row = rowb + r;
        int b, n;
        token_of(rowb, b0, r, b, n);
        bf16_t* dst = obase + (((int64_t)b * gp.qk_heads + head) * gp.qk_s_total + e.tok0 + n) * 64 + rc * 8;
        const int cur = (i * 4 + it) & 1;
        float (&c8)[8] = cb[cur];
        float (&s8)[8] = sb[cur];
        if (it < 3) request(i, it + 1, ln, cb[cur ^ 1], sb[cur ^ 1]);
        else if (i + 1 < Geo<MT>::NB32) request(i + 1, 0, ln, cb[cur ^ 1], sb[cur ^ 1]);
        float x[8];
#pragma unroll
        for (int q = 0; q < 4; q++) { x[2 * q] = __builtin_bit_cast(float, t[q] << 16); x[2 * q + 1] = __builtin_bit_cast(float, t[q] & 0xffff0000u); }
        float ss = 0.f;
#pragma unroll
        for (int q = 0; q < 8; q++) ss += x[q] * x[q];
        ss = sum8(ss);      // (the 8 lanes of a row; DPP: the same pairs as the xor-1 / 2 / 4 butterfly of the stand-alone kernel, without its three LDS round trips)
        const float rinv = rsqrtf(ss * (1.f / 64.f) + 1.1920929e-07f);
#pragma unroll
        for (int q = 0; q < 8; q++) x[q] = x[q] * rinv * w8[q];
        if (rope) {
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const float a = x[2 * q], bb = x[2 * q + 1];
            x[2 * q] = a * c8[2 * q] - bb * s8[2 * q];
            x[2 * q + 1] = bb * c8[2 * q + 1] + a * s8[2 * q + 1];
          }
        }
        if (FULL || row < p.M) {
          if constexpr (RAW) *(u32x4*)(C + (int64_t)row * p.ldc + col) = t;
          st8(dst, x);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  };
  using std::integral_constant;
  const bool full = m0 + MT <= p.M;
  if (C) {
    if (full) qk_loop(integral_constant<bool, true>{}, integral_constant<bool, true>{});
    else qk_loop(integral_constant<bool, false>{}, integral_constant<bool, true>{});
  } else {
    if (full) qk_loop(integral_constant<bool, true>{}, integral_constant<bool, false>{});
    else qk_loop(integral_constant<bool, false>{}, integral_constant<bool, false>{});
  }
}

// SwiGLU-fused w12 GEMM: the wave's columns 0..31 are gate rows, 32..63 up rows of the SAME 32 hidden indices 128 tn + 32 wn + c (the DMA
// source mapping interleaves them), so group q (columns (q & 3) * 16 ..) pairs with group q + 2; aux[M, 2h] (if given) gets the bf16
// pre-activations, C[M, h] = silu(g) * u formed from the ROUNDED values (bit-identical to the GEMM followed by mmdit_swiglu_fwd)
template <int MT, bool MXOUT = false>      // MXOUT: the MX-operand kernel (only there may the activation leave as e4m3 + block scales)
__device__ __forceinline__ void epi8_swiglu(AccT<MT>& acc, const Problem& p, int m0, int tn, int wm, int wn, int lane, char* stage, float alpha = 1.f) {
  bf16_t* Hout = (bf16_t*)p.C;
  bf16_t* GU = (bf16_t*)p.aux;
  const float* bias = p.bias;
  const int h = p.N >> 1;
  const int wr = lane & 15, wq = lane >> 4;
  const int hc = tn * 128 + wn * 32;
  float bgv[2][4], buv[2][4];      // (zeros without a bias: the add below is unconditional -- inside the unrolled loops `if (bias)` became a select per accumulator value)
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int e = 0; e < 4; e++) bgv[j][e] = buv[j][e] = 0.f;
  if (bias) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      ld4(bias + hc + j * 16 + wq * 4, bgv[j]);
      ld4(bias + h + hc + j * 16 + wq * 4, buv[j]);
    }
  }
#pragma unroll
  for (int i = 0; i < Geo<MT>::NB32; i++) {
    u32x2 pa[2][2];
    // (MX kernels sit at 256 VGPRs with the next item's prologue state live: the per-lane staging offsets / row indices of this loop, hoisted, are
    // SPILLED, and every reload is followed by s_waitcnt vmcnt(0) -- which waits for the stores just issued and for the next item's operand requests.
    // The lane id is laundered per block so that the handful of address VALU is redone instead: tools/check_spills.py counts the epilogue too.)
    int ln = lane;
    if constexpr (MXOUT) asm volatile("" : "+v"(ln));
    const int wr = ln & 15, wq = ln >> 4;
#pragma unroll
    for (int il = 0; il < 2; il++)
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const f32x4& ag = grp(acc, i, il * 4 + j);
        const f32x4& au = grp(acc, i, il * 4 + 2 + j);
        float vg[4] = {ag[0], ag[1], ag[2], ag[3]}, vu[4] = {au[0], au[1], au[2], au[3]};
        if constexpr (MXOUT) {      // (the e4m3-operand kernel: per-tensor scales arrive as alpha; 1 with block scales)
#pragma unroll
          for (int e = 0; e < 4; e++) { vg[e] *= alpha; vu[e] *= alpha; }
        }
#pragma unroll
        for (int e = 0; e < 4; e++) { vg[e] += bgv[j][e]; vu[e] += buv[j][e]; }
        const u32x2 pg = {pack_bf2(vg[0], vg[1]), pack_bf2(vg[2], vg[3])}, pu = {pack_bf2(vu[0], vu[1]), pack_bf2(vu[2], vu[3])};
        const int row = il * 16 + wr;
        if (GU) {   // staged as 32 rows x 128 B: chunks 0..3 = gate columns, 4..7 = up columns
          *LDS_PTR(u32x2, stage + row * 128 + (((j * 2 + (wq >> 1)) ^ (row & 7)) << 4) + (wq & 1) * 8) = pg;
          *LDS_PTR(u32x2, stage + row * 128 + (((4 + j * 2 + (wq >> 1)) ^ (row & 7)) << 4) + (wq & 1) * 8) = pu;
        }
        float a[4];
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const float g0 = __builtin_bit_cast(float, pg[e] << 16), g1 = __builtin_bit_cast(float, pg[e] & 0xffff0000u);
          const float u0 = __builtin_bit_cast(float, pu[e] << 16), u1 = __builtin_bit_cast(float, pu[e] & 0xffff0000u);
          a[2 * e] = silu_f(g0) * u0;
          a[2 * e + 1] = silu_f(g1) * u1;
        }
        pa[il][j] = u32x2{pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3])};
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (GU) {
      const int rr = ln >> 3, rc = ln & 7;
      const int col = (rc & 4 ? h : 0) + hc + (rc & 3) * 8;
#pragma unroll
      for (int it = 0; it < 4; it++) {
        const int r = it * 8 + rr;
        const u32x4 t = *LDS_PTR(const u32x4, stage + r * 128 + ((rc ^ (r & 7)) << 4));
        const int row = m0 + wm * (MT / 2) + i * 32 + r;
        if (row < p.M) __builtin_nontemporal_store(t, (u32x4*)(GU + (int64_t)row * p.ld_aux + col));   // read again only in backward: streaming store
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (MXOUT && p.c_scales) {
      // MX output (mxfp8 inference: the down-projection reads e4m3 + E8M0 block scales): the wave's 32 hidden indices of a row are ONE 32-block --
      // a lane holds 8 of them (fragments j = 0, 1), the lanes wr, wr + 16, wr + 32, wr + 48 the rest.  Codes staged as 32 rows x 32 B and stored
      // 16 B per lane; the arithmetic (mx_exponent / mx_pack4 on the bf16-ROUNDED activation) is that of gemm_tile.h epilogue_swiglu and of
      // mmdit_mxfp8_quantize: bit-identical to the bf16 output followed by the quantise pass.
#pragma unroll
      for (int il = 0; il < 2; il++) {
        float hv[8];
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
          for (int e = 0; e < 2; e++) { hv[4 * j + 2 * e] = __builtin_bit_cast(float, pa[il][j][e] << 16); hv[4 * j + 2 * e + 1] = __builtin_bit_cast(float, pa[il][j][e] & 0xffff0000u); }
        float amax = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) amax = fmaxf(amax, fabsf(hv[e]));
        amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
        amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
        float inv;
        const int ex = mx_exponent(amax, inv);
        const int row = il * 16 + wr;
#pragma unroll
        for (int j = 0; j < 2; j++) *LDS_PTR(unsigned, stage + row * 32 + j * 16 + wq * 4) = mx_pack4(hv + 4 * j, inv);
        const int myrow = m0 + wm * (MT / 2) + i * 32 + row;
        if (wq == 0 && myrow < p.M) p.c_scales[mx_scale_index(myrow, hc >> 5, p.M)] = (unsigned char)(ex + 127);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int r = ln >> 1, row = m0 + wm * (MT / 2) + i * 32 + r;
      const u32x4 t = *LDS_PTR(const u32x4, stage + r * 32 + (ln & 1) * 16);
      if (row < p.M) *(u32x4*)((unsigned char*)p.C + (int64_t)row * p.ldc + hc + (ln & 1) * 16) = t;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      continue;
    }
    // the activation: 32 rows x 64 B (chunk c of row r at c ^ (r & 3))
#pragma unroll
    for (int il = 0; il < 2; il++)
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int row = il * 16 + wr;
        *LDS_PTR(u32x2, stage + row * 64 + (((j * 2 + (wq >> 1)) ^ (row & 3)) << 4) + (wq & 1) * 8) = pa[il][j];
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
      const int rr = ln >> 2, rc = ln & 3;
#pragma unroll
      for (int it = 0; it < 2; it++) {
        const int r = it * 16 + rr;
        const u32x4 t = *LDS_PTR(const u32x4, stage + r * 64 + ((rc ^ (r & 3)) << 4));
        const int row = m0 + wm * (MT / 2) + i * 32 + r;
        if (row < p.M) *(u32x4*)(Hout + (int64_t)row * p.ldc + hc + rc * 8) = t;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// fp32 output of the weight gradients: plain / accumulating streaming store or atomic add; staged per 32 x 32 block (32 rows x 128 B)
template <int MT>
__device__ __forceinline__ void epi8_f32(AccT<MT>& acc, const Problem& p, int m0, int n0, int wm, int wn, int lane, char* stage, bool atomic_out, bool accumulate) {
  float* C = (float*)p.C;
  const int wr = lane & 15, wq = lane >> 4;
  const int rr = lane >> 3, rc = lane & 7;
#pragma unroll
  for (int i = 0; i < Geo<MT>::NB32; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int col = n0 + wn * 64 + j * 32 + rc * 4;
      const int row0 = m0 + wm * (MT / 2) + i * 32 + rr;
#pragma unroll
      for (int il = 0; il < 2; il++)
#pragma unroll
        for (int jj = 0; jj < 2; jj++) {
          const int row = il * 16 + wr;
          *LDS_PTR(f32x4, stage + row * 128 + (((jj * 4 + wq) ^ (row & 7)) << 4)) = grp(acc, i, il * 4 + j * 2 + jj);
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < 4; it++) {
        const int r = it * 8 + rr;
        f32x4 t = *LDS_PTR(const f32x4, stage + r * 128 + ((rc ^ (r & 7)) << 4));
        const int row = row0 + it * 8;
        if (row >= p.M || col >= p.N) continue;
        float* cp = C + (int64_t)row * p.ldc + col;
        if (atomic_out) {
#pragma unroll
          for (int e = 0; e < 4; e++) atomicAdd(cp + e, t[e]);
          continue;
        }
        if (accumulate) t += *(const f32x4*)cp;
        __builtin_nontemporal_store(t, (f32x4*)cp);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// fp32 output + bias + residual (the FLUX-VAE convolutions, vae.py _conv3: y = conv(x) + b (+ shortcut)): staged per 32 x 32 block like epi8_f32, the read-back
// side (8 lanes per row, 4 columns each) adds the bias and the residual row and stores 16 bytes
template <int MT>
__device__ __forceinline__ void epi8_f32r(AccT<MT>& acc, const Problem& p, int m0, int n0, int wm, int wn, int lane, char* stage) {
  float* C = (float*)p.C;
  constexpr int NBLK = Geo<MT>::NB32 * 2;      // 32 x 32 blocks of the wave's 128 x 64: block k = (i = k / 2, j = k % 2)
  const int wr = lane & 15, wq = lane >> 4;
  const int rr = lane >> 3, rc = lane & 7;
  const int rowt = m0 + wm * (MT / 2) + rr;
  const int colj[2] = {n0 + wn * 64 + rc * 4, n0 + wn * 64 + 32 + rc * 4};
  f32x4 b4[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (p.bias) {
#pragma unroll
    for (int j = 0; j < 2; j++) b4[j] = *(const f32x4*)(p.bias + min(colj[j], p.N - 4));      // (columns beyond N: a valid address, a value nobody stores)
  }
  auto stage_block = [&](int k) __attribute__((always_inline)) {
    const int i = k >> 1, j = k & 1;
#pragma unroll
    for (int il = 0; il < 2; il++)
#pragma unroll
      for (int jj = 0; jj < 2; jj++) {
        const int row = il * 16 + wr;
        *LDS_PTR(f32x4, stage + row * 128 + (((jj * 4 + wq) ^ (row & 7)) << 4)) = grp(acc, i, il * 4 + j * 2 + jj);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  if (!p.residual) {      // (wave-uniform)
#pragma unroll
    for (int k = 0; k < NBLK; k++) {
      stage_block(k);
      const int col = colj[k & 1];
#pragma unroll
      for (int it = 0; it < 4; it++) {
        const int r = it * 8 + rr;
        f32x4 t = *LDS_PTR(const f32x4, stage + r * 128 + ((rc ^ (r & 7)) << 4));
        const int row = rowt + (k >> 1) * 32 + it * 8;
        if (row >= p.M || col >= p.N) continue;
        t += b4[k & 1];
        *(f32x4*)(C + (int64_t)row * p.ldc + col) = t;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    return;
  }
  // With a residual (the second convolution of a FLUX-VAE ResNet block) every pass used to load its 16 bytes of shortcut, wait for them -- behind the
  // stores of the pass before, through the one in-order counter -- and store: one 1 KB request in flight per wave, ~3 GB/s per compute unit for the
  // tile's 256 KB of shortcut.  Round 6: the shortcut rows of block k + 1 are requested before block k is staged (unconditional loads, clamped
  // addresses) and the stores are not counted by the compiler (st16_uncounted): 4-8 requests in flight per wave, waits on loads only.
  f32x4 res[2][4];
  auto request = [&](int k, f32x4 (&dst)[4]) __attribute__((always_inline)) {
    const int col = min(colj[k & 1], p.N - 4);
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const int row = min(rowt + (k >> 1) * 32 + it * 8, p.M - 1);
      dst[it] = *(const f32x4*)(p.residual + (int64_t)row * p.ld_res + col);
    }
  };
  request(0, res[0]);
#pragma unroll
  for (int k = 0; k < NBLK; k++) {
    if (k + 1 < NBLK) request(k + 1, res[(k + 1) & 1]);
    stage_block(k);
    const int col = colj[k & 1];
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const int r = it * 8 + rr;
      f32x4 t = *LDS_PTR(const f32x4, stage + r * 128 + ((rc ^ (r & 7)) << 4));
      const int row = rowt + (k >> 1) * 32 + it * 8;
      t += b4[k & 1];
      t += res[k & 1][it];
      if (row < p.M && col < p.N) st16_uncounted(C + (int64_t)row * p.ldc + col, __builtin_bit_cast(u32x4, t));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// a partial tile of the split tail into its workspace slot (row-major [256][256] fp32, device-scope write-through stores)
template <int MT>
__device__ __forceinline__ void epi8_f32_slot(AccT<MT>& acc, float* slot, int wm, int wn, int lane, char* stage) {
#pragma unroll
  for (int i = 0; i < Geo<MT>::NB32; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      // (the lane id laundered per block: hoisted, the staging offsets of the 32 passes were spilled, and each reload -- scratch_load + vmcnt(0) -- waited for
      //  the write-through store before it: profiles/r06_epilogue_waits.txt)
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const int wr = ln & 15, wq = ln >> 4;
      const int rr = ln >> 3, rc = ln & 7;
#pragma unroll
      for (int il = 0; il < 2; il++)
#pragma unroll
        for (int jj = 0; jj < 2; jj++) {
          const int row = il * 16 + wr;
          *LDS_PTR(f32x4, stage + row * 128 + (((jj * 4 + wq) ^ (row & 7)) << 4)) = grp(acc, i, il * 4 + j * 2 + jj);
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < 4; it++) {
        const int r = it * 8 + rr;
        const f32x4 t = *LDS_PTR(const f32x4, stage + r * 128 + ((rc ^ (r & 7)) << 4));
        float* dst = slot + (int64_t)(wm * (MT / 2) + i * 32 + r) * 256 + wn * 64 + j * 32 + rc * 4;
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(t) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

#define BAR8() asm volatile("s_barrier" ::: "memory")
#define LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define VMCNT8(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// KT (weight gradients only): the reduction length need not be a multiple of the 64-deep K tile (text rows = 154 x batch: 2464 at MMDiT-L batch 16,
// 2002 at the reference's own batch 13).  The main loop runs the whole K tiles; the item that ends at the end of K then adds the remaining
// K % 64 rows from one more tile whose k-rows beyond K are zero-filled on their way into LDS (a plain load + ds_write pass in the LDS-DMA image,
// un-pipelined: once per output tile).  A separate instantiation: launches whose K are all multiples of 64 run the code without it.
// MX (round 5; inference, BASELINE configs[4]): e4m3 operands with E8M0 block scales on the SAME loop.  A row of 128 e4m3 values is 128 bytes, so
// the LDS image, the DMA pieces, the swizzle and the fragment reads are those of the bf16 kernel byte for byte: a lane's two 16-byte fragments of a
// K tile (chunks g and 4 + g of its row, g = lane >> 4: the bf16 k-steps 0 and 1) are exactly registers 0-3 and 4-7 of the 16x16x128 operand
// (k = 16 g + [0, 16) and 64 + 16 g + [0, 16): tools/probes/mx16_probe.hip), ONE v_mfma_scale_f32_16x16x128_f8f6f4 replaces the two bf16 MFMAs
// and a K tile is 128 deep.  The accumulator layout is that of 16x16x32, so the epilogues are shared.  Scales: lane (r, g) supplies the E8M0 byte
// of row r, 32-block g of the K tile; in the tensor's layout (mx_scale_index) the bytes of the four 32-row blocks of a 128-row group share a dword,
// so a lane loads 2 dwords per operand and K tile straight from global memory (L2-resident, a few KB per tile) with inline-asm loads one K tile
// ahead -- they are older than every staging the counted wait of P4 leaves in flight, no extra wait, no LDS -- and op_sel picks the byte.
// CONV (round 5; BASELINE configs[3]): A is the zero-bordered NHWC operand of an implicit-GEMM 3 x 3 convolution (gemm_common.h Problem::conv_mode): output
// row m = (b, yo, xo) reads pixel (s yo + kh + o, s xo + kw + o), K index (kh * 3 + kw) * C + c.  Inside a kernel row the 3 C channels of the three taps are
// contiguous, so the K tile base advances by 128 bytes and jumps by (Wp - 3) C elements every 3 C / 64 K tiles; the per-lane part of the address is the
// pixel offset of the lane's output row, one register per (half-tile, piece) instead of the row * ld product.
// PT (with MX): e4m3 operands with PER-TENSOR scales -- unit block scales (E8M0 127 in every byte, no scale loads), alpha = scale_a * scale_b in the epilogue.
template <int MT, bool A_KM, bool B_KM, int EPI, bool KT = false, bool MX = false, bool CONV = false, bool PT = false>
__global__ __launch_bounds__(512) void gemm8_kernel(GroupParams gp) {
  static_assert(!PT || MX, "per-tensor scales: the e4m3-operand kernel");
  static_assert(MT == 256 || MT == 320, "tile rows");
  static_assert(!CONV || (!A_KM && !B_KM && MT == 256 && !MX && (EPI == EPI_BF16 || EPI == EPI_F32R)), "implicit-GEMM convolution: row-major operands, 256-row tiles");
  static_assert(EPI != EPI_F32R || (!A_KM && MT == 256), "fp32 + residual epilogue: row-major A, 256-row tiles");
  static_assert(!MX || (!A_KM && !B_KM && MT == 256 && (EPI == EPI_BF16 || EPI == EPI_SWIGLU || EPI == EPI_QK)), "MX: row-major e4m3 operands, 256-row tiles, bf16 / SwiGLU / QKV epilogue");
  static_assert(!KT || (A_KM && B_KM && EPI == EPI_F32), "K tail: the weight-gradient kernel");
  static_assert(EPI == EPI_F32 ? (A_KM && B_KM && MT == 256) : !A_KM, "fp32 epilogue = weight gradients (both operands k-major, 256 rows); bf16 epilogues take a row-major A");
  static_assert(!(EPI == EPI_SWIGLU || EPI == EPI_QK) || !B_KM, "fused epilogues: row-major weight");
  static_assert(EPI != EPI_SWIGLU_BWD || (B_KM && MT == 256), "SwiGLU backward epilogue: the data-gradient layout, 256-row tiles");
  using GE = Geo<MT>;
  constexpr int QR = GE::QR, FI = GE::FI, KBUF = GE::KBUF, XA0 = GE::XA0, XA1 = GE::XA1, XB0 = GE::XB0, XB1 = GE::XB1, PAW = GE::PAW;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  const uint32_t ldsw = lds0 + wave * 1024;
  const int G = (int)gridDim.x;
  const bool hiw = wave + 8 * (PAW - 1) < GE::PA;      // (wave-uniform) this wave carries PAW A pieces per half-tile, the others PAW - 1

  int pos = (int)blockIdx.x, end = total_work(gp);
  if (gp.tail_first >= 0) {   // balanced tail: this workgroup's share of the tail units (gemm_lean.hip gemm_kk_kernel)
    const int e = xcd_chunk((int)blockIdx.x, gp.full_tiles) - gp.tail_first, E = gp.tail_G - gp.tail_first;
    const int left = (gp.total_tiles - gp.full_tiles) * gp.split_k - e;
    end = gp.full_tiles + (e < 0 || left <= 0 ? 0 : (left + E - 1) / E) * gp.tail_G;
  }
  end = __builtin_amdgcn_readfirstlane(end);

  // ---- dynamic tile claiming (round 6; gp.sched != nullptr: every 256-row launch with more work positions than workgroups) -------------------
  // A persistent workgroup needs a whole CU (its 8 waves hold all 512 registers of every SIMD), so while another kernel (a collective's channels)
  // holds C compute units C of the G workgroups of this grid become resident only when others leave.  With the static walk (workgroup b: positions
  // b, b + G, ...) their whole share then runs as a second round.  Here the positions are CLAIMED: queue q (one per XCD, so that the L2-contiguous
  // tile ranges survive) holds the positions q, q + 8, ... in order; a workgroup pops a queue with one returning agent-scope atomic add on its head
  // (sched[q]) -- its own XCD's first, then whichever it last found work in -- and looks at the others' heads once that one is empty, so a late
  // workgroup finds nothing left and leaves.  Which positions exist, and what they compute, is fixed by the planner: results do not depend on who
  // claims what.
  //   * the atomic is issued by one lane from inline asm (the compiler neither counts nor waits for it) at the START OF THE LAST PAIR OF K TILES of
  //     the current item (~3 us before its end: MI355X_MICROARCH.md "dequeue" 0.3 - 1.1 us) and its return sits in one register until the main loop's
  //     final vmcnt(0): no exposed latency, and a workgroup commits to its next position only ~3 us before it can start it;
  //   * the first position is claimed by the kernel's first instructions and taken after the set-up arithmetic;
  //   * the claimed position reaches the other waves through one LDS word of the (then idle) operand buffers, two barriers;
  //   * a workgroup that finds nothing counts itself out (sched[8]) before its last epilogue; the last one to leave zeroes the nine words: the next
  //     launch that uses this slot finds it clean (no memset node, nothing for the host to do).
  // (320-row tiles keep the static walk: the scalars this adds to what is live across the main loop cost the <320, swiglu> kernel six scratch
  //  reloads inside it.)
  int* const sq = gp.sched;
  // (kernel-uniform; 320-row tiles: never -- see launch8.  The e4m3-operand / convolution kernels -- inference -- never: claiming exists for the data-parallel
  //  training step, and the per-tensor fp8 SwiGLU kernel has no register to give it: there the compiler COPIES the claim register while the atomic that
  //  fills it may be in flight (tools/check_spills.py audits that in every kernel that claims))
  const bool dyn = MT == 256 && !MX && !CONV && sq != nullptr;
  const int xq = (int)blockIdx.x & (NXCD - 1);
  char* const s_next = smem + 16;      // (byte 0: the split tail's ticket)
  uint32_t claim_v = 0;                // wave 0, lane 0: what the last claim_issue returned (the word's value before the add)
  auto claim_issue = [&](int q) {      // q (wave-uniform): a queue, or NXCD = the count of workgroups that have left
    if (wave == 0) {
      uint64_t sv;
      asm volatile("s_nop 4\n\ts_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %0, %2, %3, %4 sc0\n\ts_mov_b64 exec, %1"
                   : "=&v"(claim_v), "=&s"(sv)
                   : "v"((uint32_t)(q * 4)), "v"(1u), "s"(sq)
                   : "memory");
    }
  };
  auto claim_take = [&](int q) -> int {      // every thread calls it, operand buffers idle; returns a position or `end`
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(claim_v)::"memory");      // (behind the main loop's own vmcnt(0): free)
    if (tid == 0) {
      int got = (int)claim_v * NXCD + q;
      if (got >= end) {      // queue q is empty: the others, after a look at their heads (an exhausted queue stays exhausted: it costs a load, not an atomic)
        got = end;
        int hd[NXCD];
#pragma unroll
        for (int t = 1; t < NXCD; t++) hd[t] = __hip_atomic_load(sq + ((q + t) & (NXCD - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int t = 1; t < NXCD; t++) {
          const int qq = (q + t) & (NXCD - 1);
          if (got == end && hd[t] * NXCD + qq < end) {
            const int p = atomicAdd(sq + qq, 1) * NXCD + qq;
            if (p < end) got = p;
          }
        }
      }
      *LDS_PTR(int, s_next) = got;
    }
    LGKM0(); BAR8();
    const int r = *LDS_PTR(const volatile int, s_next);
    LGKM0(); BAR8();
    return __builtin_amdgcn_readfirstlane(r);
  };
  if (dyn) claim_issue(xq);            // the first position: its latency lies under the set-up arithmetic below

  // ---- fragment read offsets inside a half-tile (bytes) ---------------------------------------------------------------------------------
  // row-major: fragment (16 rows x 32 k) = row (lane & 15), 16 B at k chunk 4 ks + (lane >> 4): one ds_read_b128.  The chunk permutation only
  //            involves the row's low four bits, so fragment i of a quadrant is fragment 0 + i * 2048: ONE address register per k step.
  // k-major:   two ds_read_b64_tr_b16 of 4 k-rows x 16 columns: k-rows 32 ks + 8 (lane >> 4) + ((lane & 15) >> 2) (+ 4), columns c0 + 4 (lane & 3)
  //            (the permutation mixes the column chunk into the k-row's bits: one address register per fragment).
  constexpr int NAO = A_KM ? 2 * FI : 2, NBO = B_KM ? 4 : 2;
  uint32_t aoff[NAO], boff[NBO];
#pragma unroll
  for (int ks = 0; ks < 2; ks++) {
    if (A_KM) {
#pragma unroll
      for (int i = 0; i < FI; i++) { const int kr = ks * 32 + 8 * (lane >> 4) + ((lane & 15) >> 2), c = wr * QR + i * 16 + 4 * (lane & 3); aoff[i * 2 + ks] = kr * 256 + (((c >> 3) ^ SWZ_K(kr)) << 4) + (c & 7) * 2; }
    } else {
      const int r = wr * QR + (lane & 15), kp = ks * 4 + (lane >> 4);
      aoff[ks] = r * 128 + ((kp ^ SWZ_R(r)) << 4);
    }
    if (B_KM) {
#pragma unroll
      for (int j = 0; j < 2; j++) { const int kr = ks * 32 + 8 * (lane >> 4) + ((lane & 15) >> 2), c = wc * 32 + j * 16 + 4 * (lane & 3); boff[j * 2 + ks] = kr * 256 + (((c >> 3) ^ SWZ_K(kr)) << 4) + (c & 7) * 2; }
    } else {
      const int r = wc * 32 + (lane & 15), kp = ks * 4 + (lane >> 4);
      boff[ks] = r * 128 + ((kp ^ SWZ_R(r)) << 4);
    }
  }
  auto frag = [&](bool km, const char* p) -> bf16x8 {
    if (!km) return *LDS_PTR(const bf16x8, p);
    const s16x4 lo = lds_tr16(p), hi = lds_tr16(p + 4 * 256);
    return __builtin_bit_cast(bf16x8, (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
  };

  AccT<MT> acc;
  bf16x8 fa[2 * FI], fb0[4], fb1s[4];
  bf16x8 (&fb1)[4] = MT == 256 ? fb1s : fb0;      // (320 rows: ONE set of B fragments, see REREAD below)
#define FB1 fb1
#ifdef MMDIT_G8_SPLITWAIT     // experiment (-DMMDIT_G8_SPLITWAIT, NOT the product): fragments in k-step order and a COUNTED lgkmcnt in front of the phase's first barrier,
                              // so that the first k-step's MFMAs start while the second one's reads are in flight.  Measured (same box, tools/gemm_bench.py):
                              // 8192^3 NT 1470 -> 1511, NN 1420 -> 1444, TN 1318 -> 1316 TF, SwiGLU up-projection 280 -> 268 us, block weight gradients
                              // 462 -> 455 us; all GEMM tests pass.  Not shipped: it breaks the hazard rule the schedule is built on -- the other
                              // group restages this half-tile one barrier later, and a read that has not retired by then is protected only by the
                              // DMA's latency (>= 500 cycles against <= 250 of queued LDS reads): a race with a margin, for 1.5-3 %.
  auto readA1 = [&](int buf, int x, int ks) {
#pragma unroll
    for (int i = 0; i < FI; i++) fa[i * 2 + ks] = frag(A_KM, smem + buf * KBUF + x + (A_KM ? aoff[A_KM ? i * 2 + ks : 0] : aoff[ks] + i * 2048));
  };
  auto readB1 = [&](bf16x8 (&fb)[4], int buf, int x, int ks) {
#pragma unroll
    for (int j = 0; j < 2; j++) fb[j * 2 + ks] = frag(B_KM, smem + buf * KBUF + x + (B_KM ? boff[B_KM ? j * 2 + ks : 0] : boff[ks] + j * 2048));
  };
#endif
  auto readA = [&](int buf, int x) {
#pragma unroll
    for (int i = 0; i < FI; i++)
#pragma unroll
      for (int ks = 0; ks < 2; ks++) fa[i * 2 + ks] = frag(A_KM, smem + buf * KBUF + x + (A_KM ? aoff[A_KM ? i * 2 + ks : 0] : aoff[ks] + i * 2048));
  };
  auto readB = [&](bf16x8 (&fb)[4], int buf, int x) {
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int ks = 0; ks < 2; ks++) fb[j * 2 + ks] = frag(B_KM, smem + buf * KBUF + x + (B_KM ? boff[B_KM ? j * 2 + ks : 0] : boff[ks] + j * 2048));
  };
  auto mma = [&](f32x4 (&c)[2 * FI], const bf16x8 (&fb)[4]) {
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
      for (int i = 0; i < FI; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) c[i * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j * 2 + ks], fa[i * 2 + ks], c[i * 2 + j], 0, 0, 0);
  };

  // first K tile of an item whose accumulators are NOT zeroed (drain schedule below): the first k-step starts from a constant-zero C
  auto mma0 = [&](f32x4 (&c)[2 * FI], const bf16x8 (&fb)[4]) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < FI; i++)
#pragma unroll
      for (int j = 0; j < 2; j++) c[i * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j * 2], fa[i * 2], z, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < FI; i++)
#pragma unroll
      for (int j = 0; j < 2; j++) c[i * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j * 2 + 1], fa[i * 2 + 1], c[i * 2 + j], 0, 0, 0);
  };

  // ---- staging of the current item.  The per-lane source offset of a DMA piece is formed when the piece is issued (four VALU operations in a
  // phase whose vector ALU is otherwise idle) from TWO registers per operand -- holding one offset per (half-tile, piece) would cost 8 / 10
  // registers this kernel does not have:
  //   row-major: row0 = the lane's row in piece 0 of half-tile 0, chk = 16 * its chunk; offset = min(row0 + d, R - 1) * ld * 2 + chk, where the
  //              row delta d of (half-tile, piece) is a compile-time constant (the chunk permutation depends on the piece's parity = the wave's);
  //   k-major:   col0 = the lane's first column in half-tile 0, kof = k-row * ld * 2; offset = min(col0 + d, R - 8) * 2 + kof + piece * 32 * ld * 2.
  uint32_t rowA = 0, chkA = 0, rowB = 0, chkB = 0;
  uint32_t ld2A = 0, ld2B = 0;             // bytes per operand row (wave-uniform)
  int limA = 0, limB = 0;                  // clamp: R - 1 (row-major) / R - 8 (k-major)
  // wave-uniform source pointers: curA / curB = the K tile being multiplied; stagings take it + 0 / 1 / 2 K steps (clamped to the item's
  // last K tile: past it the last one is requested again and never read) -- running pointers, no 64-bit multiplies by variables in the loop
  uint64_t curA = 0, curB = 0, stepA = 0, stepB = 0;
  int nkt = 0, krem = 0;                   // K tiles of the current item; K tiles behind the one being multiplied
  // STREAM (round 5): the operand stream does not stop at an item boundary.  The stagings of an item's last two K tiles that used to repeat its
  // last K tile (requested, never read) request the NEXT item's K tile 0 and three half-tiles of its K tile 1 instead -- exactly the state the
  // per-item prologue leaves -- so the next item's main loop starts on data that is already in LDS, with no fill latency between two tiles.
  // Needs the deferred epilogue's own staging (256-row tiles: the operand buffers belong to the next item by then), an even K tile count (the next
  // item's K tile 0 must land in buffer 0) and the same problem on both sides (the per-lane chunk / row-pitch state is shared).
  // The K tiles whose stagings stay inside the item run a loop without clamps or selects (gen = false); only an item's last two or three K tiles take
  // the general form.  MEASURED (round 5, profiles/r05_gemm_stream_ab.txt; correct: the GEMM / model / inference tests pass with it; same box, alternating
  // libraries): training step 27.91 / 27.97 ms without, 27.85 / 27.93 with; mxfp8 sampler 19.20 / 19.16 vs 19.23 / 19.25 img/s; the MX launches of MMDiT-L
  // (K = 1024, 8 K tiles per item) 714 / 731 vs 705 / 710 us.  Within the noise: the launches are power-limited and the fill latency it removes
  // was idle time the clock had been using.  NOT the product: builds with -DMMDIT_G8_STREAM only.
#ifdef MMDIT_G8_STREAM
  constexpr bool STREAM = MT == 256 && EPI != EPI_F32 && !CONV && !KT;
#else
  constexpr bool STREAM = false;
#endif
  uint32_t rowA_n = 0, rowB_n = 0;         // the next item's lane rows
  uint64_t curA_n = 0, curB_n = 0;         // ... and K tile 0
  uint64_t scurA_n = 0, scurB_n = 0;       // ... and (MX) the scale bytes of its K tile 0
  int nkt_n = 0;
  bool strm = false;                       // (workgroup-uniform) the current item's trailing stagings belong to the next item
  int swig_h = 0;                          // SwiGLU: rows between the gate and the up half of the packed weight
  // ---- MX operands: 8-register fragments (lo = chunk g, hi = chunk 4 + g of the row) and the scale dwords of two K tiles -----------------------
  constexpr int NSC = !MX ? 1 : EPI == EPI_SWIGLU ? 6 : 4;      // A (rows i & 1 = 0, 1), B (j = 0, 1) -- SwiGLU: gate and up rows
  i32x8 xa[MX ? FI : 1], xb0[MX ? 2 : 1], xb1[MX ? 2 : 1];
  uint32_t sc[2][NSC];
  uint32_t soffA = 0, soffB = 0;            // per-lane byte offsets into the scale tensors (this item)
  uint64_t scurA = 0, scurB = 0, sstepA = 0, sstepB = 0, supB = 0;      // wave-uniform: scale bytes of the current K tile; bytes per K tile; gate -> up rows
  auto xreadA = [&](int buf, int x) {
#pragma unroll
    for (int i = 0; i < (MX ? FI : 1); i++) {
      xa[i].lo = *LDS_PTR(const i32x4_t, smem + buf * KBUF + x + aoff[0] + i * 2048);
      xa[i].hi = *LDS_PTR(const i32x4_t, smem + buf * KBUF + x + aoff[1] + i * 2048);
    }
  };
  auto xreadB = [&](i32x8 (&xb)[MX ? 2 : 1], int buf, int x) {
#pragma unroll
    for (int j = 0; j < (MX ? 2 : 1); j++) {
      xb[j].lo = *LDS_PTR(const i32x4_t, smem + buf * KBUF + x + boff[0] + j * 2048);
      xb[j].hi = *LDS_PTR(const i32x4_t, smem + buf * KBUF + x + boff[1] + j * 2048);
    }
  };
  // scale dwords of K tile (current + d) into set `set` (inline asm: the compiler must neither count nor wait for these loads)
  auto sload = [&](int set, int d, bool gen = true) {      // gen = false: a K tile that is known to exist in this item (no clamp, no next item)
    if constexpr (MX && !PT) {
      const bool nx = gen && STREAM && strm && d > krem;
      const int dd = !gen ? d : nx ? min(d - krem - 1, nkt_n - 1) : min(d, krem);
      const uint64_t pa = (nx ? scurA_n : scurA) + (uint64_t)(uint32_t)dd * sstepA, pb = (nx ? scurB_n : scurB) + (uint64_t)(uint32_t)dd * sstepB;
      const uint64_t ua = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pa >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pa);
      const uint64_t ub = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pb >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pb);
      asm volatile("s_nop 4\n\tglobal_load_dword %0, %2, %3\n\tglobal_load_dword %1, %2, %3 offset:128" : "=v"(sc[set][0]), "=v"(sc[set][1]) : "v"(soffA), "s"(ua) : "memory");
      asm volatile("s_nop 4\n\tglobal_load_dword %0, %2, %3\n\tglobal_load_dword %1, %2, %3 offset:128" : "=v"(sc[set][2]), "=v"(sc[set][3]) : "v"(soffB), "s"(ub) : "memory");
      if constexpr (EPI == EPI_SWIGLU) {
        const uint64_t pu = pb + supB;
        const uint64_t uu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pu >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pu);
        asm volatile("s_nop 4\n\tglobal_load_dword %0, %2, %3\n\tglobal_load_dword %1, %2, %3 offset:128" : "=v"(sc[set][4]), "=v"(sc[set][5]) : "v"(soffB), "s"(uu) : "memory");
      }
    }
  };
  // the loads of set `set` have been waited for (counted vmcnt in front of this point): from here on the registers hold the data
  auto spin = [&](int set) {
    if constexpr (MX && !PT) {
#pragma unroll
      for (int k = 0; k < NSC; k++) asm volatile("" : "+v"(sc[set][k]));
    }
  };
  // quadrant (qm, qn) of the K tile whose scales are in set `set`: c^T += B-fragment x A-fragment (first operand = the weight rows = output columns)
  auto mmax = [&](f32x4 (&c)[2 * FI], const i32x8 (&xb)[MX ? 2 : 1], int set, int qm, int qn) {
    if constexpr (MX) {
      // B bytes: plain -- byte (wc & 1) * 2 + qn of the dword (shift by 16 (wc & 1), op_sel qn); SwiGLU -- byte wc of the gate (qn = 0) / up (qn = 1) dword
      uint32_t sb[2];
#pragma unroll
      for (int j = 0; j < 2; j++) sb[j] = PT ? 0x7f7f7f7fu : EPI == EPI_SWIGLU ? sc[set][2 + 2 * qn + j] >> (8 * wc) : sc[set][2 + j] >> (16 * (wc & 1));
#pragma unroll
      for (int i = 0; i < FI; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
          c[i * 2 + j] = mx16_mfma(xb[j], xa[i], c[i * 2 + j], EPI == EPI_SWIGLU ? 0 : qn, (int)sb[j], qm * 2 + (i >> 1), PT ? 0x7f7f7f7f : (int)sc[set][i & 1]);
    }
  };

  uint32_t cvA[2][CONV ? PAW : 1];        // CONV: pixel byte offset of this lane's row in piece i of half-tile h
  int cseg_left = 0, cseg_len = 0;        // CONV: K tiles left in / per kernel row
  uint64_t cjump = 0;                     // CONV: extra bytes when the K index moves to the next kernel row
  // the lane's operand rows of an item (the part of the staging state that changes from tile to tile of one problem)
  auto item_rows = [&](const Item& it, uint32_t& ra, uint32_t& rb) {
    const int m0 = it.tm * MT, n0 = it.tn * 256;
    if (A_KM) { const int kr = wave * 4 + (lane >> 4), lr0 = ((lane & 15) ^ SWZ_K(kr)) * 8; ra = m0 + (lr0 >> 6) * 128 + (lr0 & 63); }
    else ra = m0 + wave * 8 + (lane >> 3);
    if (B_KM) { const int kr = wave * 4 + (lane >> 4), lr0 = ((lane & 15) ^ SWZ_K(kr)) * 8; rb = n0 + (lr0 >> 5) * 64 + (lr0 & 31); }
    else { const int lr = wave * 8 + (lane >> 3); rb = EPI == EPI_SWIGLU ? it.tn * 128 + lr : n0 + (lr >> 5) * 64 + (lr & 31); }
  };
  auto item_setup = [&](const Item& it) {
    const Problem& q = gp.p[it.pi];
    const int m0 = it.tm * MT, n0 = it.tn * 256;
    ld2A = (uint32_t)q.lda * (MX ? 1 : 2); ld2B = (uint32_t)q.ldb * (MX ? 1 : 2);      // bytes per operand row (e4m3: one byte per value)
    item_rows(it, rowA, rowB);
    if (A_KM) {   // (MT = 256 only) piece i: k-rows (8 i + wave) * 4 + (lane >> 4); LDS chunk (lane & 15) holds global chunk (lane & 15) ^ SWZ_K(k-row)
      const int kr = wave * 4 + (lane >> 4);
      chkA = kr * ld2A; limA = q.M - 8;
    } else {      // piece pc = wave + 8 i: local rows 8 pc + (lane >> 3) of the half-tile = wave row lr / QR, quadrant row lr % QR
      const int lr = wave * 8 + (lane >> 3);
      chkA = ((lane & 7) ^ SWZ_R(lr)) * 16; limA = q.M - 1;
    }
    if (B_KM) {
      const int kr = wave * 4 + (lane >> 4);
      chkB = kr * ld2B; limB = q.N - 8;
    } else {
      const int lr = wave * 8 + (lane >> 3);      // piece 0: wave columns lr >> 5 (0, 1), column lr & 31; piece 1: wave columns 2, 3
      chkB = ((lane & 7) ^ SWZ_R(lr)) * 16;
      if (EPI == EPI_SWIGLU) swig_h = q.N >> 1;   // gate / up rows of the packed weight
      limB = q.N - 1;
    }
    const int kt0 = it.h0 >> 1;
    nkt = (it.h1 - it.h0) >> 1;
    krem = nkt - 1;
    stepA = A_KM ? (uint64_t)64 * q.lda * 2 : 128;
    stepB = B_KM ? (uint64_t)64 * q.ldb * 2 : 128;
    curA = (uint64_t)(uintptr_t)q.A + kt0 * stepA;
    curB = (uint64_t)(uintptr_t)q.B + kt0 * stepB;
    if constexpr (CONV) {
      const int hw = q.cHo * q.cWo, sdn = q.conv_mode == 2 ? 2 : 1, o = q.conv_mode == 2 ? 1 : 0;
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int i = 0; i < PAW; i++) {
          const int lr = 64 * i + wave * 8 + (lane >> 3);                               // local row of the half-tile: wave row lr / 64, quadrant row lr % 64
          const int m = min(m0 + (lr >> 6) * 128 + h * 64 + (lr & 63), q.M - 1);
          const int bimg = m / hw, rem = m - bimg * hw, yo = rem / q.cWo, xo = rem - yo * q.cWo;
          cvA[h][i] = (uint32_t)((((int64_t)bimg * q.cHp + sdn * yo) * q.cWp + sdn * xo) * q.cC * 2);
        }
      cseg_len = 3 * (q.cC / 64);
      const int kh = kt0 / cseg_len, within = kt0 - kh * cseg_len;
      cjump = (uint64_t)((int64_t)q.cWp - 3) * q.cC * 2;
      curA = (uint64_t)(uintptr_t)q.A + (uint64_t)((((int64_t)(kh + o) * q.cWp + o) * q.cC) * 2) + (uint64_t)within * 128;
      cseg_left = cseg_len - within;
    }
    if constexpr (MX && !PT) {
      // scale bytes of (row, K half) at ((half * rows_pad + (row & ~127)) * 2 + (row & 31) * 8 + (blk & 1) * 4 + ((row >> 5) & 3): mx_scale_index
      const int64_t padA = ((int64_t)q.M + 127) & ~(int64_t)127, padB = ((int64_t)q.N + 127) & ~(int64_t)127;
      sstepA = (uint64_t)padA * 4; sstepB = (uint64_t)padB * 4;                 // two 64-wide halves per K tile
      const uint32_t lane_part = (uint32_t)((lane & 15) * 8 + ((lane >> 4) & 1) * 4);
      soffA = (uint32_t)(wr * 256) + lane_part + (uint32_t)(lane >> 5) * (uint32_t)(padA * 2);
      scurA = (uint64_t)(uintptr_t)q.scale_a + (uint64_t)kt0 * sstepA + (uint64_t)m0 * 2;
      if constexpr (EPI == EPI_SWIGLU) {
        soffB = lane_part + (uint32_t)(lane >> 5) * (uint32_t)(padB * 2);
        scurB = (uint64_t)(uintptr_t)q.scale_b + (uint64_t)kt0 * sstepB + (uint64_t)(it.tn * 128) * 2;
        supB = (uint64_t)(q.N >> 1) * 2;
      } else {
        soffB = (uint32_t)((wc >> 1) * 256) + lane_part + (uint32_t)(lane >> 5) * (uint32_t)(padB * 2);
        scurB = (uint64_t)(uintptr_t)q.scale_b + (uint64_t)kt0 * sstepB + (uint64_t)n0 * 2;
      }
    }
  };
  // half-tile h of K tile (current + d), d = 0 / 1 / 2 -> buffer buf
  auto stageA = [&](int h, int d, int buf, bool gen = true) {
    const bool nx = gen && STREAM && strm && d > krem;      // (workgroup-uniform) a K tile of the next item
    const int dd = !gen ? d : nx ? min(d - krem - 1, nkt_n - 1) : min(d, krem);
    const char* src = (const char*)(uintptr_t)((nx ? curA_n : curA) + (uint64_t)(uint32_t)dd * stepA + (CONV && dd >= cseg_left ? cjump : 0));      // (CONV: d <= 2 < a kernel row's K tiles)
    const uint32_t rA = nx ? rowA_n : rowA;
    const uint32_t dst = ldsw + buf * KBUF + (h ? XA1 : XA0);
#pragma unroll
    for (int i = 0; i < PAW; i++) {
      if (!(i < PAW - 1 || GE::PA % 8 == 0 || hiw)) continue;
      uint32_t voff;
      if (A_KM) voff = (uint32_t)min((int)rA + h * 64, limA) * 2 + chkA + (uint32_t)(i * 32) * ld2A;
      else {      // local row 8 (wave + 8 i) + ..: wave row (lr / QR), quadrant row lr % QR -> tile row (lr / QR) * 2 QR + h QR + lr % QR
        const int lr0 = 64 * i;                                 // + wave * 8 + (lane >> 3) < 64: same wave row as long as 64 i + 63 < QR ... handled per case
        int drow;
        if (MT == 256) drow = i * 128 + h * 64;                  // QR = 64: piece i is wave row i
        else drow = 0;                                           // (320: see below)
        if (CONV) voff = cvA[h][CONV ? i : 0] + chkA;
        else if (MT == 256) voff = (uint32_t)min((int)rA + drow, limA) * ld2A + chkA;
        else {
          // QR = 80: local row lr = 64 i + 8 wave + (lane >> 3) (< 160); tile row = (lr >= 80 ? 160 : 0) + h * 80 + (lr >= 80 ? lr - 80 : lr) = lr + (lr >= 80 ? 80 : 0) + h * 80
          const int lr = lr0 + (int)rA;                        // rowA carries m0 + 8 wave + (lane >> 3)
          const int lrl = lr0 + wave * 8;                        // wave-uniform part: pieces never straddle the wave rows (80 = 10 pieces)
          voff = (uint32_t)min(lr + (lrl >= 80 ? 80 : 0) + h * 80, limA) * ld2A + chkA;
        }
      }
      glds16(voff, src, dst + i * 8192);
    }
  };
  auto stageB = [&](int h, int d, int buf, bool gen = true) {
    const bool nx = gen && STREAM && strm && d > krem;
    const int dd = !gen ? d : nx ? min(d - krem - 1, nkt_n - 1) : min(d, krem);
    const char* src = (const char*)(uintptr_t)((nx ? curB_n : curB) + (uint64_t)(uint32_t)dd * stepB);
    const uint32_t rB = nx ? rowB_n : rowB;
    const uint32_t dst = ldsw + buf * KBUF + (h ? XB1 : XB0);
#pragma unroll
    for (int i = 0; i < 2; i++) {
      uint32_t voff;
      if (B_KM) voff = (uint32_t)min((int)rB + h * 32, limB) * 2 + chkB + (uint32_t)(i * 32) * ld2B;
      else if (EPI == EPI_SWIGLU) voff = (uint32_t)((int)rB + i * 64 + h * swig_h) * ld2B + chkB;      // hidden index 128 tn + 64 i + ..: gate (h = 0) / up (h = 1)
      else voff = (uint32_t)min((int)rB + i * 128 + h * 32, limB) * ld2B + chkB;                       // piece i: wave columns 2 i, 2 i + 1
      glds16(voff, src, dst + i * 8192);
    }
  };
  auto next_ktile = [&]() {
    curA += stepA; curB += stepB; krem--;
    if constexpr (MX && !PT) { scurA += sstepA; scurB += sstepB; }
    if constexpr (CONV) { if (--cseg_left == 0) { curA += cjump; cseg_left = cseg_len; } }
  };
  // Two schedules.  KEEP (256 rows): B0 stays in registers from P1 to P4; stagings P1(t): A1(t+1), P2(t): A0(t+2), P3(t): B0(t+2), P4(t): B1(t+2);
  // the three youngest half-tiles at the counted wait are A, B, B.  REREAD (320 rows: 160 accumulator + 40 A-fragment registers leave room for ONE
  // set of B fragments): P4 reads B0 again, so B0 is restaged last -- P1(t): B0(t+1), P2(t): A0(t+2), P3(t): B1(t+2), P4(t): A1(t+2); youngest A, B, A.
  constexpr bool REREAD = MT != 256;
  auto wait3 = [&]() {
    constexpr int NA = REREAD ? 2 : 1, NB = REREAD ? 1 : 2;
    if (GE::PA % 8 == 0 || hiw) VMCNT8(NA * PAW + NB * 2);
    else VMCNT8(NA * (PAW - 1) + NB * 2);
  };

#ifdef MMDIT_G8_SPLITWAIT
#define P1_READS(cur) readB1(fb0, cur, XB0, 0); readA1(cur, XA0, 0); __builtin_amdgcn_sched_barrier(0); readB1(fb0, cur, XB0, 1); readA1(cur, XA0, 1);
#define P1_WAIT() asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(((A_KM ? 2 : 1) * FI + (B_KM ? 4 : 2)) > 15 ? 15 : ((A_KM ? 2 : 1) * FI + (B_KM ? 4 : 2))) : "memory")
#define P3_READS(cur) readA1(cur, XA1, 0); __builtin_amdgcn_sched_barrier(0); readA1(cur, XA1, 1);
#define P3_WAIT() asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"((A_KM ? 2 : 1) * FI) : "memory")
#else
#define P1_READS(cur) readB(fb0, cur, XB0); __builtin_amdgcn_sched_barrier(0); readA(cur, XA0);
#define P1_WAIT() LGKM0()
#define P3_READS(cur) readA(cur, XA1);
#define P3_WAIT() LGKM0()
#endif
#ifdef MMDIT_PROBES      // timing experiment (debug bit 256, results are garbage): K tiles 1 and 2 behind a drain do not wait for the drain's stores
#define WAIT3X() do { if (DRAIN && relax > 0) { relax--; VMCNT8(1 * PAW + 2 * 2 + 4 * FI); } else wait3(); } while (0)
#else
#define WAIT3X() wait3()
#endif
#define KTILE8(cur, gen)                                                                                          \
  {                                                                                                          \
    /* P1 */                                                                                                 \
    P1_READS(cur)                                                                                            \
    if (REREAD) stageB(0, 1, (cur) ^ 1, gen); else stageA(1, 1, (cur) ^ 1, gen);                           \
    P1_WAIT(); BAR8(); __builtin_amdgcn_sched_barrier(0);                                                    \
    mma(acc.a[0][0], fb0);                                                                                   \
    __builtin_amdgcn_sched_barrier(0); BAR8();                                                               \
    /* P2 */                                                                                                 \
    readB(FB1, cur, XB1);                                                                                    \
    stageA(0, 2, cur, gen);                                                                                 \
    LGKM0(); BAR8(); __builtin_amdgcn_sched_barrier(0);                                                      \
    mma(acc.a[0][1], FB1);                                                                                   \
    __builtin_amdgcn_sched_barrier(0); BAR8();                                                               \
    /* P3 */                                                                                                 \
    P3_READS(cur)                                                                                            \
    if (REREAD) stageB(1, 2, cur, gen); else stageB(0, 2, cur, gen);                                       \
    P3_WAIT(); BAR8(); __builtin_amdgcn_sched_barrier(0);                                                    \
    mma(acc.a[1][1], FB1);                                                                                   \
    __builtin_amdgcn_sched_barrier(0); BAR8();                                                               \
    /* P4 */                                                                                                 \
    if (REREAD) { readB(fb0, cur, XB0); stageA(1, 2, cur, gen); } else stageB(1, 2, cur, gen);             \
    WAIT3X();                                                                                                \
    LGKM0(); BAR8(); __builtin_amdgcn_sched_barrier(0);                                                      \
    mma(acc.a[1][0], fb0);                                                                                   \
    __builtin_amdgcn_sched_barrier(0); BAR8();                                                               \
    next_ktile();                                                                                            \
  }

  // MX K tile (KEEP schedule): the phases of KTILE8 with 8-register fragments and one scaled MFMA per 16 x 16 block; P1 requests the NEXT K tile's
  // scale dwords into the other register set (older than every staging P4's counted wait leaves in flight)
#define KTILE8X(cur, gen)                                                                                         \
  {                                                                                                          \
    xreadB(xb0, cur, XB0); __builtin_amdgcn_sched_barrier(0); xreadA(cur, XA0);                              \
    sload((cur) ^ 1, 1, gen);                                                                                     \
    stageA(1, 1, (cur) ^ 1, gen);                                                                                 \
    LGKM0(); BAR8(); __builtin_amdgcn_sched_barrier(0);                                                      \
    mmax(acc.a[0][0], xb0, cur, 0, 0);                                                                       \
    __builtin_amdgcn_sched_barrier(0); BAR8();                                                               \
    xreadB(xb1, cur, XB1);                                                                                   \
    stageA(0, 2, cur, gen);                                                                                       \
    LGKM0(); BAR8(); __builtin_amdgcn_sched_barrier(0);                                                      \
    mmax(acc.a[0][1], xb1, cur, 0, 1);                                                                       \
    __builtin_amdgcn_sched_barrier(0); BAR8();                                                               \
    xreadA(cur, XA1);                                                                                        \
    stageB(0, 2, cur, gen);                                                                                       \
    LGKM0(); BAR8(); __builtin_amdgcn_sched_barrier(0);                                                      \
    mmax(acc.a[1][1], xb1, cur, 1, 1);                                                                       \
    __builtin_amdgcn_sched_barrier(0); BAR8();                                                               \
    stageB(1, 2, cur, gen);                                                                                       \
    wait3();                                                                                                 \
    spin((cur) ^ 1);                                                                                         \
    LGKM0(); BAR8(); __builtin_amdgcn_sched_barrier(0);                                                      \
    mmax(acc.a[1][0], xb0, cur, 1, 0);                                                                       \
    __builtin_amdgcn_sched_barrier(0); BAR8();                                                               \
    next_ktile();                                                                                            \
  }

  // the first K tile of an item (KEEP schedule, 256 rows): always from a constant-zero C (no zero-fill of the accumulators); with `dr` the previous
  // tile drains in its load sections.  The counted wait of P4 must then leave the stagings of P2, P3, P4 in flight AND the 3 x FI stores issued
  // between them (each phase: stores first, then its staging).
#define KTILE8D(cur)                                                                                         \
  {                                                                                                          \
    P1_READS(cur)                                                                                            \
    if (dr) drain_q(acc.a[0][0], 0, 0);                                                                      \
    stageA(1, 1, (cur) ^ 1);                                                                                 \
    P1_WAIT(); BAR8(); __builtin_amdgcn_sched_barrier(0);                                                    \
    mma0(acc.a[0][0], fb0);                                                                                  \
    __builtin_amdgcn_sched_barrier(0); BAR8();                                                               \
    readB(FB1, cur, XB1);                                                                                    \
    if (dr) drain_q(acc.a[0][1], 0, 1);                                                                      \
    stageA(0, 2, cur);                                                                                       \
    LGKM0(); BAR8(); __builtin_amdgcn_sched_barrier(0);                                                      \
    mma0(acc.a[0][1], FB1);                                                                                  \
    __builtin_amdgcn_sched_barrier(0); BAR8();                                                               \
    P3_READS(cur)                                                                                            \
    if (dr) drain_q(acc.a[1][1], 1, 1);                                                                      \
    stageB(0, 2, cur);                                                                                       \
    P3_WAIT(); BAR8(); __builtin_amdgcn_sched_barrier(0);                                                    \
    mma0(acc.a[1][1], FB1);                                                                                  \
    __builtin_amdgcn_sched_barrier(0); BAR8();                                                               \
    if (dr) drain_q(acc.a[1][0], 1, 0);                                                                      \
    stageB(1, 2, cur);                                                                                       \
    if (dr) VMCNT8(1 * PAW + 2 * 2 + 3 * FI); else wait3();                                                  \
    LGKM0(); BAR8(); __builtin_amdgcn_sched_barrier(0);                                                      \
    mma0(acc.a[1][0], fb0);                                                                                  \
    __builtin_amdgcn_sched_barrier(0); BAR8();                                                               \
    next_ktile();                                                                                            \
  }

  char* stage = smem + GE::STAGE0 + wave * EP32_WAVE_BYTES;
  int* s_ticket = (int*)smem;      // (the operand buffers are idle while an epilogue runs)

  // The epilogue of an item.  DEFER (256 rows, bf16 outputs: the staging has its own 32 KB): it runs AFTER the next item's prologue requests
  // have been issued, i.e. under their latency (~1.4 us per tile of a K = 768 launch); otherwise (320 rows: the staging aliases the operand
  // buffers; weight gradients: ticket + long reductions) right behind the item's main loop.
  constexpr bool DEFER = MT == 256 && EPI != EPI_F32;
  // DRAIN (round 5): the epilogue of a FULL tile runs inside the FIRST K tile of the workgroup's next item.  That K tile's MFMAs start from a
  // constant-zero C (mma0), so quadrant q's accumulator registers still hold the previous tile until phase q's MFMAs overwrite them: each
  // quadrant is converted and stored in the load section of the phase that precedes its first MFMA, while the other wave group's MFMAs run.
  // No LDS staging: a v_permlane16_swap pair turns two 16 x 16 C^T fragments (lane = row, 4 consecutive columns) into 8 consecutive columns
  // per lane, one global_store_dwordx4 per 16 rows x 64 B.  The stores are issued UNCONDITIONALLY (full tiles only; partial tiles keep the
  // staged epilogue) because the counted vmcnt of that K tile's P4 counts them: 3 quadrants x FI stores are younger than the staging it waits for.
  // MEASURED (round 5, profiles/r05_gemm_drain_ab.txt; same box, N = 6144, K = 768, 256 x 256 tiles): staged deferred epilogue 272 us, drain 275 us,
  // drain WITHOUT its stores 231 us -- the stores cost the same ~41 us per launch (322 MB) whether they leave in a burst behind the main loop or
  // inside the next tile's first K tile, with the following waits relaxed (debug 256), as streaming stores (debug 1024), or with the workgroups'
  // start times spread over a tile period (debug >> 16): correct, bit-identical, not faster.  NOT the product: probes builds only.
#ifdef MMDIT_PROBES
  constexpr bool DRAIN = DEFER && EPI == EPI_BF16 && !MX;
#else
  constexpr bool DRAIN = false;
#endif
  char* dr_ptr = nullptr;      // this lane's first output element of the previous tile (row lane & 15, 8-column group of its 16-lane row)
  uint32_t dr_ld = 0;          // bytes per output row
  bool prev_drain = false;     // the pending item is drained inside the next item's first K tile
  auto drain_q = [&](f32x4 (&c)[2 * FI], int qm, int qn) {
    char* pq = dr_ptr + (int64_t)(qm * QR) * dr_ld + qn * 64;
#pragma unroll
    for (int i = 0; i < FI; i++) {
      const f32x4 a = c[i * 2], b = c[i * 2 + 1];
      const uint32_t a0 = pack_bf2(a[0], a[1]), a1 = pack_bf2(a[2], a[3]), b0 = pack_bf2(b[0], b[1]), b1 = pack_bf2(b[2], b[3]);
      // rows of 16 lanes: fragment j = 0 columns {0-3, 4-7, 8-11, 12-15}; after the swap row 0 holds j = 0 columns 0-7, row 1 j = 1 columns 0-7,
      // row 2 j = 0 columns 8-15, row 3 j = 1 columns 8-15
      const auto x = __builtin_amdgcn_permlane16_swap(a0, b0, false, false), y = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
#ifdef MMDIT_PROBES
      if (gp.debug & 512) { asm volatile("" ::"v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1])); continue; }      // ablation: conversion + swaps, no stores
#endif
#ifdef MMDIT_PROBES
      if (gp.debug & 1024) { __builtin_nontemporal_store((u32x4){x[0], y[0], x[1], y[1]}, (u32x4*)(pq + (int64_t)(i * 16) * dr_ld)); continue; }      // experiment: streaming stores
#endif
      *(u32x4*)(pq + (int64_t)(i * 16) * dr_ld) = (u32x4){x[0], y[0], x[1], y[1]};
    }
  };
  auto run_epilogue = [&](const Item& it) __attribute__((always_inline)) {
    // ---- epilogue ---------------------------------------------------------------------------------------------------------------------
  const Problem& q = gp.p[it.pi];
  const int m0 = it.tm * MT, n0 = it.tn * 256;
#ifdef MMDIT_PROBES
  if (gp.debug & 8) { if (acc.a[0][0][0][0] == 12345.f) ((float*)q.C)[0] = 1.f; return; }      // ablation (MMDIT_GEMM_DEBUG=8): no epilogue
#endif
  float alpha = 1.f;
  if constexpr (PT) alpha = q.scale_a[0] * q.scale_b[0];      // e4m3 operands with per-tensor scales (device scalars)
  if constexpr (EPI == EPI_BF16) epi8_bf16<MT, MX, MX || (MT == 320 && B_KM)>(acc, q, gp, m0, n0, wr, wc, lane, stage, alpha);
  else if constexpr (EPI == EPI_SWIGLU) epi8_swiglu<MT, MX>(acc, q, m0, it.tn, wr, wc, lane, stage, alpha);
  else if constexpr (EPI == EPI_QK) epi8_qk<MT>(acc, q, gp, gp.qk[it.pi & 1], m0, n0, wr, wc, lane, stage);
  else if constexpr (EPI == EPI_SWIGLU_BWD) epi8_swiglu_bwd<MT>(acc, q, m0, n0, wr, wc, lane, stage);
  else if constexpr (EPI == EPI_F32R) epi8_f32r<MT>(acc, q, m0, n0, wr, wc, lane, stage);
  else {
    if (it.atomic && gp.ws_slots) {
      // partial tile of the split tail through the workspace (gemm_lean.hip gemm_kk_kernel: slot store, ticket, the last slice sums)
      constexpr int TE = 256 * 256, NW = 8;
      const int tt = it.tile - gp.full_tiles, S = gp.split_k;
      float* slots = gp.ws_slots + (int64_t)tt * S * TE;
      epi8_f32_slot<MT>(acc, slots + (int64_t)it.sk * TE, wr, wc, lane, stage);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) *s_ticket = atomicAdd(gp.ws_count + tt, 1);
      __syncthreads();
      if (*s_ticket == S - 1) {
        float* C = (float*)q.C;
        constexpr int NCH = TE / 4 / (64 * NW);
#pragma unroll 1
        for (int b0 = 0; b0 < NCH; b0 += 4) {
          f32x4 tsum[4];
#pragma unroll
          for (int u = 0; u < 4; u++) tsum[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
          for (int s0 = 0; s0 < S; s0 += 4) {
            f32x4 v[4][4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const int idx = tid + (b0 + u) * (64 * NW), r = idx / 64, c = (idx % 64) * 4;
#pragma unroll
              for (int k = 0; k < 4; k++) {
                const float* src = slots + (int64_t)min(s0 + k, S - 1) * TE + (int64_t)r * 256 + c;
                asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[u][k]) : "v"(src) : "memory");
              }
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
              asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[u][0]), "+v"(v[u][1]), "+v"(v[u][2]), "+v"(v[u][3])::"memory");
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
              for (int k = 0; k < 4; k++)
                if (s0 + k < S) tsum[u] += v[u][k];
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int idx = tid + (b0 + u) * (64 * NW), r = idx / 64, c = (idx % 64) * 4;
            if (m0 + r < q.M && n0 + c < q.N) {
              float* cp = C + (int64_t)(m0 + r) * q.ldc + n0 + c;
              if (gp.accumulate) tsum[u] += *(const f32x4*)cp;
              __builtin_nontemporal_store(tsum[u], (f32x4*)cp);
            }
          }
        }
        if (tid == 0) gp.ws_count[tt] = 0;
      }
      __syncthreads();
    } else {
      epi8_f32<MT>(acc, q, m0, n0, wr, wc, lane, stage, it.atomic, gp.accumulate != 0);
    }
  }
  };

#ifdef MMDIT_PROBES      // experiment: start-time skew -- workgroup b waits ((b / 8) % 32) / 32 of a span of (debug >> 16) * 256 cycles (spread inside each XCD)
  if (gp.debug >> 16) {
    const long long span = (long long)(gp.debug >> 16) * 256, until = __builtin_readcyclecounter() + span * (((int)blockIdx.x / 8) % 32) / 32;
    while ((long long)__builtin_readcyclecounter() < until) __builtin_amdgcn_s_sleep(8);
  }
#endif
  if (dyn) pos = claim_take(xq);
  bool left = false;      // (workgroup-uniform) this workgroup has counted itself out
  Item item = item_at(gp, pos, end), prev = item;
  bool pending = false;
  bool streamed = false;      // (workgroup-uniform) this item's first stagings were requested by the previous item's main loop
  while (item.valid) {
    if (STREAM && streamed) {      // the staging state is the one the previous item's loop has been using for its trailing requests
      rowA = rowA_n; rowB = rowB_n; curA = curA_n; curB = curB_n; nkt = nkt_n; krem = nkt - 1;
      if constexpr (MX && !PT) { scurA = scurA_n; scurB = scurB_n; }
    } else {
      item_setup(item);
    }
    if constexpr (STREAM) {
      const Item nxt = item_at(gp, item.pos + G, end);      // (found again at the bottom of the loop: not kept alive across the main loop)
      strm = !dyn && nxt.valid && nxt.pi == item.pi && nkt >= 2 && (nkt & 1) == 0 && nxt.h1 > nxt.h0;
#ifdef MMDIT_PROBES
      if (gp.debug & 4096) strm = false;      // A/B: a prologue per item
#endif
      if (strm) {
        const Problem& q = gp.p[nxt.pi];
        item_rows(nxt, rowA_n, rowB_n);
        curA_n = (uint64_t)(uintptr_t)q.A + (uint64_t)(nxt.h0 >> 1) * stepA;
        curB_n = (uint64_t)(uintptr_t)q.B + (uint64_t)(nxt.h0 >> 1) * stepB;
        nkt_n = (nxt.h1 - nxt.h0) >> 1;
        if constexpr (MX && !PT) {
          scurA_n = (uint64_t)(uintptr_t)q.scale_a + (uint64_t)(nxt.h0 >> 1) * sstepA + (uint64_t)(nxt.tm * MT) * 2;
          scurB_n = (uint64_t)(uintptr_t)q.scale_b + (uint64_t)(nxt.h0 >> 1) * sstepB + (uint64_t)(EPI == EPI_SWIGLU ? nxt.tn * 128 : nxt.tn * 256) * 2;
        }
      }
    }
    // (dynamic claiming) the next position is claimed at the start of the item's last pair of K tiles (pair index tcl), from the queue this item came from
    constexpr bool LATE = !DRAIN && !STREAM;      // (the experiment builds' loops have no claim point: they claim at the item's start)
    const int qcl = item.pos & (NXCD - 1), tcl = dyn ? (LATE && nkt >= 2 ? (nkt - 2) & ~1 : -1) : -2;      // (workgroup-uniform; -2: never)
    if (tcl == -1) claim_issue(qcl);      // (an item of fewer than two K tiles: now)
    if (nkt > 0 && !(STREAM && streamed)) {
      if constexpr (MX) sload(0, 0);      // (first: the scale dwords of K tile 0 are older than the stagings the first counted wait leaves in flight)
      // prologue: K tile 0 whole, three half-tiles of K tile 1 (in the order the loop continues; second argument: K tiles ahead)
      if (REREAD) { stageA(0, 0, 0); stageB(1, 0, 0); stageA(1, 0, 0); stageB(0, 0, 0); stageA(0, 1, 1); stageB(1, 1, 1); stageA(1, 1, 1); }
      else { stageA(0, 0, 0); stageB(0, 0, 0); stageB(1, 0, 0); stageA(1, 0, 0); stageA(0, 1, 1); stageB(0, 1, 1); stageB(1, 1, 1); }
    }
    const bool dr = DRAIN && pending && prev_drain && nkt > 0;      // (workgroup-uniform)
#ifdef MMDIT_PROBES
    int relax = (dr && (gp.debug & 256)) ? 2 : 0;
#endif
    if (DEFER && pending && !dr) run_epilogue(prev);
    if (!DRAIN || nkt == 0) {      // (DRAIN: the first K tile starts from a constant-zero C)
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
          for (int i = 0; i < 2 * FI; i++) acc.a[a][b][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if (nkt > 0) {
      if (!(STREAM && streamed)) wait3();      // (DEFER: the epilogue's stores are younger than every request -- this also waits for all but a few of them.
                                               //  STREAM: K tile 0 has been complete since the last counted wait of the previous item's loop)
      BAR8();
      if (wr == 1) BAR8();         // group 1 runs one barrier behind from here on
      int t = 0;
      if constexpr (DRAIN) {       // the first K tile peeled: with or without the previous tile's drain
        KTILE8D(0)
#pragma unroll 1
        for (t = 1; t + 1 < nkt; t += 2) {
          KTILE8(1, true)
          KTILE8(0, true)
        }
        if (t < nkt) KTILE8(1, true)
      } else if constexpr (MX) {
        spin(0);
        if constexpr (STREAM) {      // the K tiles whose stagings stay inside the item: no clamps, no selects
#pragma unroll 1
          for (; t + 3 < nkt; t += 2) {
            KTILE8X(0, false)
            KTILE8X(1, false)
          }
        }
#pragma unroll 1
        for (; t + 1 < nkt; t += 2) {
          if (t == tcl) claim_issue(qcl);
          KTILE8X(0, true)
          KTILE8X(1, true)
        }
        if (t < nkt) KTILE8X(0, true)
      } else {
        if constexpr (STREAM) {
#pragma unroll 1
          for (; t + 3 < nkt; t += 2) {
            KTILE8(0, false)
            KTILE8(1, false)
          }
        }
#pragma unroll 1
        for (; t + 1 < nkt; t += 2) {
          if (t == tcl) claim_issue(qcl);
          KTILE8(0, true)
          KTILE8(1, true)
        }
        if (t < nkt) KTILE8(0, true)
      }
      if (!(STREAM && strm)) VMCNT8(0);      // the trailing (unused) requests have landed (STREAM: they are the next item's and stay in flight)
      if (wr == 0) BAR8();         // rejoin
      BAR8();                      // every wave's requests have landed and every wave has left the operand buffers
    }
    if constexpr (KT) {
      const Problem& q = gp.p[item.pi];
      const int kfull = q.nk * 64, ktail = q.K - kfull;
      if (ktail > 0 && item.h1 == 2 * q.nk && item.h0 < item.h1) {      // (workgroup-uniform) the ONE item of the tile that ends at the end of K (an empty
                                                                         //  split-K slice behind it has h0 == h1 == 2 nk as well)
        // the tail tile into buffer 0, in the image the LDS-DMA writes (piece i of a half-tile: k-rows 32 i + 4 wave + (lane >> 4), 16 bytes per lane)
        const bf16_t* Ab = (const bf16_t*)q.A + (int64_t)kfull * q.lda;
        const bf16_t* Bb = (const bf16_t*)q.B + (int64_t)kfull * q.ldb;
        const int kr0 = wave * 4 + (lane >> 4);
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
          for (int i = 0; i < 2; i++) {
            const int kr = kr0 + 32 * i;
            u32x4 va = {0u, 0u, 0u, 0u}, vb = {0u, 0u, 0u, 0u};
            if (kr < ktail) {
              va = *(const u32x4*)(Ab + (int64_t)kr * q.lda + min((int)rowA + h * 64, limA));
              vb = *(const u32x4*)(Bb + (int64_t)kr * q.ldb + min((int)rowB + h * 32, limB));
            }
            *LDS_PTR(u32x4, smem + (h ? XA1 : XA0) + wave * 1024 + lane * 16 + i * 8192) = va;
            *LDS_PTR(u32x4, smem + (h ? XB1 : XB0) + wave * 1024 + lane * 16 + i * 8192) = vb;
          }
        __syncthreads();
        readB(fb0, 0, XB0); readB(fb1s, 0, XB1); readA(0, XA0);
        mma(acc.a[0][0], fb0); mma(acc.a[0][1], fb1s);
        readA(0, XA1);
        mma(acc.a[1][1], fb1s); mma(acc.a[1][0], fb0);
        __syncthreads();             // every wave has left the buffers (epilogue staging / the next prologue reuse them)
      }
    }
    // (dynamic claiming) the next position -- before the epilogue: a workgroup that finds none counts itself out now, under its last epilogue
    int npos = item.pos + G;
    if (dyn) {
      npos = claim_take(qcl);
      if (npos == end) { claim_issue(NXCD); left = true; }
    }
    if (DEFER) {
      prev = item; pending = true;
      if constexpr (DRAIN) {
        const Problem& q = gp.p[item.pi];
        const int m0 = item.tm * MT, n0 = item.tn * 256;
        prev_drain = m0 + MT <= q.M && n0 + 256 <= q.N && !q.bias && gp.act == MMDIT_ACT_NONE && (q.ldc & 7) == 0 && ((uintptr_t)q.C & 15) == 0;
#ifdef MMDIT_PROBES
        if (gp.debug & 128) prev_drain = false;      // A/B: the staged deferred epilogue everywhere
#endif
        dr_ld = (uint32_t)q.ldc * 2;
#ifdef MMDIT_PROBES
        const int m0w = (gp.debug & 2048) ? 0 : m0;      // experiment: every tile row block writes to rows [0, 256) -- the output stays L2-resident
#else
        const int m0w = m0;
#endif
        dr_ptr = (char*)q.C + ((int64_t)(m0w + wr * (MT / 2) + (lane & 15)) * q.ldc + n0 + wc * 64 + ((lane >> 4) & 1) * 16 + (lane >> 5) * 8) * 2;
      }
    } else {
      run_epilogue(item);
      if (MT != 256) BAR8();       // (the staging lives in the operand buffers the next prologue overwrites)
    }
    streamed = STREAM && strm;
    item = item_at(gp, npos, end);
  }
  if (DEFER && pending) run_epilogue(prev);
  if (dyn) {      // the last workgroup to leave resets the queues (every other one has made its last claim before it counted itself out)
    if (!left) claim_issue(NXCD);      // (a workgroup that never found a position)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(claim_v)::"memory");
    if (tid == 0 && (int)claim_v == G - 1) {
#pragma unroll
      for (int q = 0; q <= NXCD; q++) atomicExch(sq + q, 0);
    }
  }
}

template <int MT, bool A_KM, bool B_KM, int EPI, bool KT = false, bool MX = false, bool CONV = false, bool PT = false>
int launch8(const GroupParams& gp, hipStream_t s) {
  auto k = gemm8_kernel<MT, A_KM, B_KM, EPI, KT, MX, CONV, PT>;
  constexpr int smem = Geo<MT>::SMEM;
  static unsigned long long attr_done = 0;   // one bit per device
  if (!mmdit_device_once(attr_done)) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    mmdit_device_mark(attr_done);
  }
  const int work = total_work(gp);
  const int cu = mmdit_get_cu_budget();      // what the planner counted on
  const bool persistent = gp.persistent && work > cu;
  if (MT == 256 && !MX && !CONV && persistent && gp.tail_first < 0) {
    // more positions than the budget's workgroups: claimed dynamically when the workspace is registered (mmdit_gemm_set_workspace) -- on the WHOLE device:
    // a workgroup whose compute unit is taken starts late, finds the queues empty and leaves; one whose compute unit is free does its share
    GroupParams gq = gp;
    gq.sched = mmdit_gemm_sched_slot();
    const int all = mmdit_device_cus(), grid = gq.sched ? (work < all ? work : all) : cu;
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), smem, s, gq);
    return mmdit_launch_status();
  }
  hipLaunchKernelGGL(k, dim3(persistent ? cu : work), dim3(512), smem, s, gp);
  return mmdit_launch_status();
}

// The instantiations are split over two translation units (the file is compiled twice: gemm8p.hip = the training kernels, gemm8p_inf.hip = this file with
// MMDIT_G8_PART 2 = the e4m3-operand and convolution kernels of the inference / VAE paths), so that the two halves compile in parallel.
#ifndef MMDIT_G8_PART
#define MMDIT_G8_PART 1
#endif

#if MMDIT_G8_PART == 2
// e4m3 operands, E8M0 block scales (gp.mx) or per-tensor scales (inference): 256-row tiles, row-major weight, bf16 output or the SwiGLU epilogue
}  // namespace
int gemm::launch_gemm8_fp8(bool b_km, const GroupParams& gp, hipStream_t s) {
  if (b_km || gp.act == MMDIT_ACT_SWIGLU_BWD || gp.act == MMDIT_ACT_SILU) return MMDIT_ERR_ARG;
  if (gp.qk_on) return gp.mx && gp.act == MMDIT_ACT_NONE ? launch8<256, false, false, EPI_QK, false, true>(gp, s) : MMDIT_ERR_ARG;      // QKV projection + QK-norm / RoPE epilogue
  if (!gp.mx) return gp.act == MMDIT_ACT_SWIGLU ? launch8<256, false, false, EPI_SWIGLU, false, true, false, true>(gp, s) : launch8<256, false, false, EPI_BF16, false, true, false, true>(gp, s);
  return gp.act == MMDIT_ACT_SWIGLU ? launch8<256, false, false, EPI_SWIGLU, false, true>(gp, s) : launch8<256, false, false, EPI_BF16, false, true>(gp, s);
}

// implicit-GEMM 3 x 3 convolution (every problem of the launch has conv_mode != 0), 256 x 256 tiles: bf16 output (+ bias / SiLU) or fp32 output + bias + residual
int gemm::launch_gemm8_conv(bool f32_out, const GroupParams& gp, hipStream_t s) {
  return f32_out ? launch8<256, false, false, EPI_F32R, false, false, true>(gp, s) : launch8<256, false, false, EPI_BF16, false, false, true>(gp, s);
}

#else

template <int MT>
int launch8_bf16(bool b_km, const GroupParams& gp, hipStream_t s) {
  if (gp.qk_on) {      // (at 320 rows the QKV epilogue does not fit the register budget without spills in the K loop: gemm.hip keeps that launch on the wide kernel)
    if constexpr (MT == 256) return b_km ? MMDIT_ERR_ARG : launch8<MT, false, false, EPI_QK>(gp, s);
    else return MMDIT_ERR_SHAPE;
  }
  if (gp.act == MMDIT_ACT_SWIGLU) return b_km ? MMDIT_ERR_ARG : launch8<MT, false, false, EPI_SWIGLU>(gp, s);
  if (gp.act == MMDIT_ACT_SWIGLU_BWD) {
    if constexpr (MT == 256) return b_km ? launch8<MT, false, true, EPI_SWIGLU_BWD>(gp, s) : MMDIT_ERR_ARG;
    else return MMDIT_ERR_SHAPE;
  }
  return b_km ? launch8<MT, false, true, EPI_BF16>(gp, s) : launch8<MT, false, false, EPI_BF16>(gp, s);
}

}  // namespace

// MT x 256 tiles (cfg CFG_256x256 or CFG_320x256).  a_km && b_km: fp32 weight gradients (256 rows; the K-decomposed schedule of gemm.hip);
// otherwise bf16 output with the bias / SiLU, SwiGLU (gp.act) or QKV (gp.qk_on) epilogue; fp8: the e4m3-operand kernels of gemm8p_inf.hip.  gemm.hip has checked the rest.
int gemm::launch_gemm8(int cfg, bool a_km, bool b_km, const GroupParams& gp, hipStream_t s, bool ktail, bool fp8) {
  if (fp8) return cfg == CFG_256x256 && !a_km && !ktail ? launch_gemm8_fp8(b_km, gp, s) : MMDIT_ERR_ARG;
  if (a_km) {
    if (!b_km || cfg != CFG_256x256) return MMDIT_ERR_ARG;
    return ktail ? launch8<256, true, true, EPI_F32, true>(gp, s) : launch8<256, true, true, EPI_F32>(gp, s);
  }
  if (ktail) return MMDIT_ERR_ARG;
  if (cfg == CFG_320x256) return launch8_bf16<320>(b_km, gp, s);
  if (cfg == CFG_256x256) return launch8_bf16<256>(b_km, gp, s);
  return MMDIT_ERR_ARG;
}
#endif
