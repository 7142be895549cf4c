// Lean LDS-DMA MFMA GEMM for the hot shapes of the MMDiT step: bf16 A[M,K] (row-major) x bf16 B ([N,K] row-major: forward;
// [K,N] k-major: data gradients) -> bf16 C[M,N] (+ bias, + SiLU), K % 64 == 0, grouped problems, persistent tile loop.
//
// Same machinery as gemm_dma.hip (4-slot ring of 32-wide K halves filled by global_load_lds, DMA cursor three halves ahead and
// running across tile boundaries, pieces issued between the MFMA rows, one counted vmcnt + one barrier per half, fragments
// software-pipelined in registers, epilogue of a tile deferred behind the first half of the next one) without the general kernel's
// work-item machinery (no split-K / stream-K / tails / implicit convolution / fp8 / fp32 epilogues; the SwiGLU epilogue of the
// packed w12 GEMM is a template variant): the whole schedule state is
// wave-uniform and lives in SGPRs, which frees the registers for a
//   320 x 256 tile (8 waves 2x4, 5x2 accumulators = 160 VGPRs, 144 KB LDS):
// the N = 768 GEMMs of MMDiT-B at per-GPU batch 64 (out-proj, MLP down, and the data gradients of QKV / out / MLP up; image +
// text rows grouped: 26 240 rows) are 249 tiles of 320 x 256 -- ONE round on the 256 CUs -- against 309 tiles of 256 x 256 (two
// rounds, the second one 21 % full) or 1230 of 128 x 128 (2.4 rounds on 512 slots).  320 rows are 20 DMA pieces of 16 rows per
// half: waves 0..3 issue three of them, waves 4..7 two (the vmcnt immediates differ per wave half accordingly).
#include "gemm_tile.h"

using namespace gemm;

namespace {

struct TileRef {
  int pi, tm, tn, nh;
  bool valid;
};

__device__ __forceinline__ TileRef tile_at(const GroupParams& gp, int pos) {
  TileRef t;
  t.valid = pos < gp.total_tiles;
  t.pi = t.tm = t.tn = 0;
  t.nh = 0;
  if (!t.valid) return t;
  const Problem& p = locate_in_problem(gp, xcd_chunk(pos, gp.total_tiles), t.tm, t.tn);
  t.pi = (int)(&p - &gp.p[0]);
  t.nh = 2 * p.nk;
  return t;
}

__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }

template <int WM, int WN, int MI, int NJ, bool B_KM, bool SWIGLU = false>
__global__ __launch_bounds__(64 * WM * WN) void gemm_lean_kernel(GroupParams gp) {
  static_assert(!SWIGLU || (!B_KM && WN == 4 && NJ == 2), "SwiGLU epilogue: row-major packed weight, 256-column tile");
  constexpr int TBM = WM * MI * 32, TBN = WN * NJ * 32, NW = WM * WN;
  constexpr int HA = TBM * 64, HB = TBN * 64, H = HA + HB;          // bytes of one ring slot
  constexpr int NA = TBM / 16, NB = TBN / 16;                        // 1-KiB DMA pieces per half and operand
  constexpr int PA_LO = NA / NW, PA_REM = NA % NW, PA_HI = PA_LO + (PA_REM ? 1 : 0), PB = NB / NW;
  constexpr int PPH = PA_HI + PB;                                    // most pieces a wave issues per half
  static_assert(NB % NW == 0 && PA_LO >= 1 && PPH <= 2 * MI, "piece schedule");
  static_assert(NW * EP32_WAVE_BYTES <= H && NJ == 2, "epilogue staging lives in a ring slot; wave sub-tile is 64 columns wide");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
  MMDIT_YOUNG_HALF_PRIO();
  const int wm = wave / WN, wn = wave % WN;
  const bool hi = PA_REM && wave < PA_REM;                           // this wave carries PA_HI A pieces per half
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  const int G = (int)gridDim.x;

  // ---- DMA cursor: (tile, half) of the next half to issue; wave-uniform ----------------------------------------
  TileRef ct = tile_at(gp, (int)blockIdx.x);
  int cpos = (int)blockIdx.x, ch = 0;
  uint32_t va[PA_HI], vb[PB];
  const char* sa = nullptr;
  const char* sb = nullptr;
  int64_t stepb = 0;
  auto cursor_setup = [&]() {
    const Problem& q = gp.p[ct.pi];
#pragma unroll
    for (int i = 0; i < PA_HI; i++) va[i] = piece_voff<false, TBM>(min(wave + NW * i, NA - 1), lane, q.lda, ct.tm * TBM, q.M);
#pragma unroll
    for (int i = 0; i < PB; i++) {
      if constexpr (SWIGLU) vb[i] = swiglu_voff<2>(wave + NW * i, lane, q.ldb, ct.tn, q.N >> 1);   // gate / up rows interleaved per wave (gemm_tile.h)
      else vb[i] = piece_voff<B_KM, TBN>(wave + NW * i, lane, q.ldb, ct.tn * TBN, q.N);
    }
    stepb = B_KM ? (int64_t)BKH * q.ldb * 2 : BKH * 2;
    sa = (const char*)q.A;
    sb = (const char*)q.B;
  };
  // after the last half of the stream the cursor stays where it is: the steady-state loop keeps issuing (re-reading that half
  // into a free slot), so the loop body is branch-free on data and every wave issues a fixed number of pieces per half
  auto cursor_advance = [&]() {
    if (!ct.valid) return;
    if (ch + 1 < ct.nh) {
      ch++;
      sa += BKH * 2;
      sb += stepb;
      return;
    }
    const TileRef nx = tile_at(gp, cpos + G);
    if (nx.valid) {
      ct = nx;
      cpos += G;
      ch = 0;
      cursor_setup();
    } else {
      ct.valid = false;
    }
  };
  auto issue_piece = [&](int q, int slot) {   // q: compile-time index among this wave's pieces of the cursor's half
    const uint32_t dst = lds0 + slot * H;
    if (q < PA_HI) {
      if (q < PA_LO || hi) glds16(va[q], sa, dst + (wave + NW * q) * 1024);
    } else {
      glds16(vb[q - PA_HI], sb, dst + HA + (wave + NW * (q - PA_HI)) * 1024);
    }
  };

  f32x16 acc[MI][NJ];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
      for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  };
  int cslot = 0, dslot = 0;   // ring slots of the half being multiplied / of the half being issued
  auto bump = [](int s) { return s + 1 == RING ? 0 : s + 1; };
  auto run_epilogue = [&](const TileRef& t) {
    char* stage = smem + dslot * H + wave * EP32_WAVE_BYTES;   // dslot: free until the next issue
    if constexpr (SWIGLU) epilogue_swiglu<MI, false>(acc, gp.p[t.pi], t.tm * TBM, t.tn, wm, wn, lane, stage, 1.f);
    else epilogue_bf16<MI, NJ>(acc, gp.p[t.pi], gp, t.tm * TBM, t.tn * TBN, wm, wn, lane, stage);
  };

  if (ct.valid) {
    cursor_setup();
#pragma unroll 1
    for (int s = 0; s < RING - 1; s++) {
#pragma unroll
      for (int q = 0; q < PPH; q++) issue_piece(q, dslot);
      dslot = bump(dslot);
      cursor_advance();
    }
  }

  bf16x8 a[MI], b[2][NJ];
  constexpr int DSTRIDE = (2 * MI) / PPH > 0 ? (2 * MI) / PPH : 1;   // MFMA rows between two pieces
  auto half_body = [&]() {
    const int nslot = bump(cslot);
    const char* ta = smem + cslot * H;
    const char* tb = ta + HA;
    const char* na = smem + nslot * H;   // (after the last half of the stream: read, never used)
    const char* nb = na + HA;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      const int c = ks & 1, nx = c ^ 1;
      const bool last = ks == 1;
#pragma unroll
      for (int j = 0; j < NJ; j++) b[nx][j] = load_frag_h<B_KM, TBN>(last ? nb : tb, wn * (NJ * 32) + j * 32, last ? 0 : ks + 1, lane);
#pragma unroll
      for (int i = 0; i < MI; i++) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[c][j], a[i], acc[i][j], 0, 0, 0);
        a[i] = load_frag_h<false, TBM>(last ? na : ta, wm * (MI * 32) + i * 32, last ? 0 : ks + 1, lane);
        const int q = ks * MI + i;   // compile-time after unrolling
        if (q % DSTRIDE == 0 && q / DSTRIDE < PPH) {
          __builtin_amdgcn_sched_barrier(0);
          issue_piece(q / DSTRIDE, dslot);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    dslot = bump(dslot);
    cslot = nslot;
    cursor_advance();
  };
  auto half_sync = [&]() {
    // RING-1 halves are in flight: the current half and the next one have landed once only the pieces of the youngest one
    // may still be outstanding (loads retire in order); a wave's own count per half is PA_HI + PB or PA_LO + PB
    if (hi) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 3) * (PA_HI + PB)) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 3) * (PA_LO + PB)) : "memory");
    __builtin_amdgcn_s_barrier();
  };

  TileRef tile = tile_at(gp, (int)blockIdx.x), prev = tile;
  int pos = (int)blockIdx.x;
  bool pending = false, first = true;
  while (tile.valid) {
    half_sync();
    if (pending) {   // the previous tile's epilogue, deferred to here: its stores drain under the MFMAs that follow
      run_epilogue(prev);
      __builtin_amdgcn_s_barrier();   // staging reads done before the DMA below refills that slot
    }
    zero_acc();
    if (first) {   // fragments of a tile's first half are carried over from the previous tile, except at the start of the stream
      first = false;
#pragma unroll
      for (int j = 0; j < NJ; j++) b[0][j] = load_frag_h<B_KM, TBN>(smem + HA, wn * (NJ * 32) + j * 32, 0, lane);
#pragma unroll
      for (int i = 0; i < MI; i++) a[i] = load_frag_h<false, TBM>(smem, wm * (MI * 32) + i * 32, 0, lane);
    }
    half_body();
#pragma unroll 1
    for (int u = 1; u < tile.nh; u++) {
      half_sync();
      half_body();
    }
    pending = true;
    prev = tile;
    pos += G;
    tile = tile_at(gp, pos);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing (unused) DMA pieces must land before the LDS is reused / released
  __builtin_amdgcn_s_barrier();                      // every wave has left the last half (its slot is the staging slot)
  if (pending) run_epilogue(prev);
}

template <int WM, int WN, int MI, int NJ, bool B_KM, bool SWIGLU = false>
int launch_lean(const GroupParams& gp, hipStream_t s) {
  constexpr int smem = RING * (WM * MI * 32 + WN * NJ * 32) * 64;
  auto k = gemm_lean_kernel<WM, WN, MI, NJ, B_KM, SWIGLU>;
  static bool attr_done = false;  // idempotent; a benign race only repeats the call
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  const int grid = gp.total_tiles < 256 ? gp.total_tiles : 256;   // one persistent workgroup per CU
  hipLaunchKernelGGL(k, dim3(grid), dim3(64 * WM * WN), smem, s, gp);
  return mmdit_launch_status();
}


// ------------------------------------------------------------------------------------------------------------------------------
// Wide-slot variant of the lean kernel: same tiles / waves / epilogues, but a ring slot is a whole 64-wide K step and the ring is
// a double buffer.  Row-major operands are fetched as 128-byte rows (whole cache lines per DMA lane group); k-major B keeps the
// two 32-row half images of the kernel above back to back.  The DMA cursor runs ONE slot ahead (its pieces are issued between the
// first MFMA rows of the slot being multiplied), fragments are pipelined inside a slot only.
// ------------------------------------------------------------------------------------------------------------------------------
template <int WM, int WN, int MI, int NJ, bool B_KM, bool SWIGLU = false>
__global__ __launch_bounds__(64 * WM * WN) void gemm_wide_kernel(GroupParams gp) {
  static_assert(!SWIGLU || (!B_KM && WN == 4 && NJ == 2), "SwiGLU epilogue: row-major packed weight, 256-column tile");
  constexpr int TBM = WM * MI * 32, TBN = WN * NJ * 32, NW = WM * WN;
  constexpr int HA = TBM * 128, HB = TBN * 128, H = HA + HB;        // bytes of one slot (64-wide K step)
  constexpr int PA = TBM / 8 / NW, PB = TBN / 8 / NW, PPS = PA + PB; // 1-KiB DMA pieces per wave and slot
  static_assert(TBM % (8 * NW) == 0 && TBN % (8 * NW) == 0 && PPS <= 4 * MI, "piece schedule");
  static_assert(NW * EP32_WAVE_BYTES <= H && NJ == 2, "epilogue staging lives in the idle slot; wave sub-tile is 64 columns wide");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
  MMDIT_YOUNG_HALF_PRIO();
  const int wm = wave / WN, wn = wave % WN;
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  const int G = (int)gridDim.x;

  // ---- DMA cursor: (tile, slot) of the next 64-wide K step to request; wave-uniform ---------------------------------
  TileRef ct = tile_at(gp, (int)blockIdx.x);
  int cpos = (int)blockIdx.x, ch = 0;
  uint32_t va[PA], vb[PB];
  const char* sa = nullptr;
  const char* sb = nullptr;
  int64_t stepb = 0;
  auto cursor_setup = [&]() {
    const Problem& q = gp.p[ct.pi];
#pragma unroll
    for (int i = 0; i < PA; i++) va[i] = wide_voff(wave * PA + i, lane, q.lda, ct.tm * TBM, q.M);
#pragma unroll
    for (int i = 0; i < PB; i++) {
      if constexpr (B_KM) {   // two 32-row half images: pieces 0 .. TBN/16-1 are k rows 0..31, the rest k rows 32..63
        const int c = wave * PB + i, hsel = c / (TBN / 16);
        vb[i] = piece_voff<true, TBN>(c % (TBN / 16), lane, q.ldb, ct.tn * TBN, q.N) + (uint32_t)(hsel * 32 * q.ldb * 2);
      } else if constexpr (SWIGLU) {
        const int c = wave * PB + i, r = 8 * c + (lane >> 3), chunk = (lane & 7) ^ (r & 7);   // tile-local row r -> gate / up row of the packed weight
        const int row = ((r >> 5) & 1) * (q.N >> 1) + ct.tn * 128 + (r >> 6) * 32 + (r & 31);
        vb[i] = (uint32_t)((int64_t)row * q.ldb * 2 + chunk * 16);
      } else {
        vb[i] = wide_voff(wave * PB + i, lane, q.ldb, ct.tn * TBN, q.N);
      }
    }
    stepb = B_KM ? (int64_t)64 * q.ldb * 2 : 128;
    sa = (const char*)q.A;
    sb = (const char*)q.B;
  };
  auto cursor_advance = [&]() {   // past the end of the stream the last slot is requested again (into the idle buffer; never read)
    if (!ct.valid) return;
    if (ch + 1 < (ct.nh >> 1)) {
      ch++;
      sa += 128;
      sb += stepb;
      return;
    }
    const TileRef nx = tile_at(gp, cpos + G);
    if (nx.valid) {
      ct = nx;
      cpos += G;
      ch = 0;
      cursor_setup();
    } else {
      ct.valid = false;
    }
  };
  auto issue_piece = [&](int q, int buf) {   // q: compile-time index among this wave's pieces of the cursor's slot
    const uint32_t dst = lds0 + buf * H;
    if (q < PA) glds16(va[q], sa, dst + (wave * PA + q) * 1024);
    else glds16(vb[q - PA], sb, dst + HA + (wave * PB + (q - PA)) * 1024);
  };

  f32x16 acc[MI][NJ];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
      for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  };
  int cbuf = 0;   // buffer being multiplied; the DMA fills cbuf ^ 1
  auto run_epilogue = [&](const TileRef& t) {
    char* stage = smem + (cbuf ^ 1) * H + wave * EP32_WAVE_BYTES;   // the idle buffer: free until the next issue
    if constexpr (SWIGLU) epilogue_swiglu<MI, false>(acc, gp.p[t.pi], t.tm * TBM, t.tn, wm, wn, lane, stage, 1.f);
    else epilogue_bf16<MI, NJ>(acc, gp.p[t.pi], gp, t.tm * TBM, t.tn * TBN, wm, wn, lane, stage);
  };
  auto ldB = [&](const char* tb, int j, int ks) -> bf16x8 {
    if constexpr (B_KM) return load_frag_h<true, TBN>(tb + (ks >> 1) * (TBN * 64), wn * (NJ * 32) + j * 32, ks & 1, lane);
    else return load_frag_w(tb, wn * (NJ * 32) + j * 32, ks, lane);
  };

  if (ct.valid) {
    cursor_setup();
#pragma unroll
    for (int q = 0; q < PPS; q++) issue_piece(q, 0);
    cursor_advance();
  }

  bf16x8 a[MI], b[2][NJ];
  constexpr int DSTRIDE = (4 * MI) / PPS > 1 ? 1 : 1;   // one piece behind each of the first PPS MFMA rows: the late rows cover its latency
  auto slot_body = [&]() {
    const char* ta = smem + cbuf * H;
    const char* tb = ta + HA;
#pragma unroll
    for (int j = 0; j < NJ; j++) b[0][j] = ldB(tb, j, 0);
#pragma unroll
    for (int i = 0; i < MI; i++) a[i] = load_frag_w(ta, wm * (MI * 32) + i * 32, 0, lane);
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const int c = ks & 1, nx = c ^ 1;
      const bool last = ks == 3;
      if (!last) {
#pragma unroll
        for (int j = 0; j < NJ; j++) b[nx][j] = ldB(tb, j, ks + 1);
      }
#pragma unroll
      for (int i = 0; i < MI; i++) {
        __builtin_amdgcn_sched_barrier(0);
        MMDIT_PRIO(1);
#pragma unroll
        for (int j = 0; j < NJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[c][j], a[i], acc[i][j], 0, 0, 0);
        MMDIT_PRIO(0);
        if (!last) a[i] = load_frag_w(ta, wm * (MI * 32) + i * 32, ks + 1, lane);
        const int q = ks * MI + i;   // compile-time after unrolling
        if (q % DSTRIDE == 0 && q / DSTRIDE < PPS) {
          __builtin_amdgcn_sched_barrier(0);
          issue_piece(q / DSTRIDE, cbuf ^ 1);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    cursor_advance();
  };
  auto slot_sync = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the slot (and any older store) have landed
    __builtin_amdgcn_s_barrier();                      // ... everybody's; and everyone has left the other buffer
  };

  TileRef tile = tile_at(gp, (int)blockIdx.x), prev = tile;
  int pos = (int)blockIdx.x;
  bool pending = false;
  while (tile.valid) {
    slot_sync();
    if (pending) {   // the previous tile's epilogue, deferred to here: its stores drain under the MFMAs that follow
      run_epilogue(prev);
      __builtin_amdgcn_s_barrier();   // staging reads done before the DMA below refills that buffer
    }
    zero_acc();
    slot_body();
    cbuf ^= 1;
#pragma unroll 1
    for (int u = 1; u < (tile.nh >> 1); u++) {
      slot_sync();
      slot_body();
      cbuf ^= 1;
    }
    pending = true;
    prev = tile;
    pos += G;
    tile = tile_at(gp, pos);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing (unused) pieces must land before the LDS is reused / released
  __builtin_amdgcn_s_barrier();
  if (pending) run_epilogue(prev);
}

template <int WM, int WN, int MI, int NJ, bool B_KM, bool SWIGLU = false>
int launch_wide(const GroupParams& gp, hipStream_t s) {
  constexpr int smem = 2 * (WM * MI * 32 + WN * NJ * 32) * 128;
  auto k = gemm_wide_kernel<WM, WN, MI, NJ, B_KM, SWIGLU>;
  static bool attr_done = false;  // idempotent; a benign race only repeats the call
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  const int grid = gp.total_tiles < 256 ? gp.total_tiles : 256;   // one persistent workgroup per CU
  hipLaunchKernelGGL(k, dim3(grid), dim3(64 * WM * WN), smem, s, gp);
  return mmdit_launch_status();
}

}  // namespace

int gemm::launch_lean_cfg(int cfg, bool b_km, const GroupParams& gp, hipStream_t s) {
  // wide-slot variant (128-byte DMA rows, double buffer): default; MMDIT_GEMM_WIDE=0 selects the 4-slot ring of 32-wide halves.
  // Measured on the MMDiT-B shapes (tools/gemm_bench.py): out-proj 37.6 -> 34.7 us, w3 113 -> 106 / 123 -> 117 us, qkv 117.5 -> 114 us,
  // w12 dgrad 228 -> 222 us, 8192^3 1218 -> 1247 TF; the step 31.36 -> 31.19 ms.
  static const char* wide_env = getenv("MMDIT_GEMM_WIDE");
  if (!wide_env || atoi(wide_env)) {
    if (gp.act == MMDIT_ACT_SWIGLU) {
      if (b_km) return MMDIT_ERR_ARG;
      if (cfg == CFG_320x256) return launch_wide<2, 4, 5, 2, false, true>(gp, s);
      if (cfg == CFG_256x256) return launch_wide<2, 4, 4, 2, false, true>(gp, s);
      return MMDIT_ERR_ARG;
    }
    if (cfg == CFG_320x256) return b_km ? launch_wide<2, 4, 5, 2, true>(gp, s) : launch_wide<2, 4, 5, 2, false>(gp, s);
    if (cfg == CFG_256x256) return b_km ? launch_wide<2, 4, 4, 2, true>(gp, s) : launch_wide<2, 4, 4, 2, false>(gp, s);
    return MMDIT_ERR_ARG;
  }
  if (gp.act == MMDIT_ACT_SWIGLU) {   // packed w12 GEMM with the activation in the epilogue (gemm.hip has checked the rest)
    if (b_km) return MMDIT_ERR_ARG;
    if (cfg == CFG_320x256) return launch_lean<2, 4, 5, 2, false, true>(gp, s);
    if (cfg == CFG_256x256) return launch_lean<2, 4, 4, 2, false, true>(gp, s);
    return MMDIT_ERR_ARG;
  }
  if (cfg == CFG_320x256) return b_km ? launch_lean<2, 4, 5, 2, true>(gp, s) : launch_lean<2, 4, 5, 2, false>(gp, s);
  if (cfg == CFG_256x256) return b_km ? launch_lean<2, 4, 4, 2, true>(gp, s) : launch_lean<2, 4, 4, 2, false>(gp, s);
  return MMDIT_ERR_ARG;
}
