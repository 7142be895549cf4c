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

#ifdef MMDIT_PROBES      // the ring-of-halves kernel of round 2: superseded by the wide-slot kernel below and by gemm8p.hip; A/B builds only
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
  static unsigned long long attr_done = 0;  // one bit per device; idempotent, a benign race only repeats the call
  if (!mmdit_device_once(attr_done)) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    mmdit_device_mark(attr_done);
  }
  const int cu = mmdit_get_cu_budget();
  const int grid = gp.total_tiles < cu ? gp.total_tiles : cu;   // one persistent workgroup per CU (of the budget)
  hipLaunchKernelGGL(k, dim3(grid), dim3(64 * WM * WN), smem, s, gp);
  return mmdit_launch_status();
}


#endif   // MMDIT_PROBES

// ------------------------------------------------------------------------------------------------------------------------------
// Wide-slot variant of the lean kernel: same tiles / waves / epilogues, but a ring slot is a whole 64-wide K step and the ring is
// a double buffer.  Row-major operands are fetched as 128-byte rows (whole cache lines per DMA lane group); k-major B keeps the
// two 32-row half images of the kernel above back to back.  The DMA cursor runs ONE slot ahead (its pieces are issued between the
// first MFMA rows of the slot being multiplied), fragments are pipelined inside a slot only.
// ------------------------------------------------------------------------------------------------------------------------------
template <int WM, int WN, int MI, int NJ, bool B_KM, bool SWIGLU = false, bool QK = false>
__global__ __launch_bounds__(64 * WM * WN) void gemm_wide_kernel(GroupParams gp) {
  static_assert(!SWIGLU || (!B_KM && WN == 4 && NJ == 2), "SwiGLU epilogue: row-major packed weight, 256-column tile");
  static_assert(!QK || (!B_KM && !SWIGLU && NJ == 2), "QKV epilogue: row-major weight, one head per wave");
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
        const int c = wave * PB + i, r = 8 * c + (lane >> 3), chunk = (lane & 7) ^ MMDIT_WIDE_SWZ(r);   // tile-local row r -> gate / up row of the packed weight
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
    else if constexpr (QK) epilogue_bf16_qk<MI, NJ>(acc, gp.p[t.pi], gp, gp.qk[t.pi & 1], t.tm * TBM, t.tn * TBN, wm, wn, lane, stage);
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

template <int WM, int WN, int MI, int NJ, bool B_KM, bool SWIGLU = false, bool QK = false>
int launch_wide(const GroupParams& gp, hipStream_t s) {
  constexpr int smem = 2 * (WM * MI * 32 + WN * NJ * 32) * 128;
  auto k = gemm_wide_kernel<WM, WN, MI, NJ, B_KM, SWIGLU, QK>;
  static unsigned long long attr_done = 0;  // one bit per device; idempotent, a benign race only repeats the call
  if (!mmdit_device_once(attr_done)) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    mmdit_device_mark(attr_done);
  }
  const int cu = mmdit_get_cu_budget();
  const int grid = gp.total_tiles < cu ? gp.total_tiles : cu;   // one persistent workgroup per CU (of the budget)
  hipLaunchKernelGGL(k, dim3(grid), dim3(64 * WM * WN), smem, s, gp);
  return mmdit_launch_status();
}

#ifdef MMDIT_PROBES      // the weight-gradient kernel of round 3: superseded by gemm8_kernel<256, true, true, f32> (gemm8p.hip); A/B builds only
// ------------------------------------------------------------------------------------------------------------------------------
// Lean weight-gradient kernel: C[M,N] (fp32) = A^T B with BOTH operands k-major (A [K,M], B [K,N] row-major: dW = dY^T X), the
// K-decomposed schedule of gemm.hip (whole-K tiles in rounds + a split tail whose partial tiles are added atomically into a
// pre-zeroed C, optionally the balanced tail), plain / accumulating / atomic fp32 epilogue -- and nothing else.  The general
// kernel (gemm_dma_kernel<2,4,4,2,true,true,float,float>) carries the same loop, but with every feature compiled in its
// wave-uniform state no longer fits the scalar registers (106 SGPRs + 28 spilled to VGPR lanes, 18 k lines of ISA); here the state
// is the cursor's item, the current item and five words of the previous one.
// ------------------------------------------------------------------------------------------------------------------------------
template <int MI, int NJ>
__device__ __forceinline__ void epilogue_f32_plain(f32x16 (&acc)[MI][NJ], const Problem& p, int m0, int n0, int wm, int wn, int lane, char* stage,
                                                   bool atomic_out, bool accumulate) {
  float* C = (float*)p.C;
  const int wr = lane & 31, wc = lane >> 5;          // write side: row, 16-B chunk parity
  const int rr = lane >> 3, rc = lane & 7;           // read side: row within the 8-row pass, 16-B chunk
#pragma unroll
  for (int i = 0; i < MI; i++) {
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const int col = n0 + wn * (NJ * 32) + j * 32 + rc * 4;
      const int row0 = m0 + wm * (MI * 32) + i * 32 + rr;
#pragma unroll
      for (int g = 0; g < 4; g++)
        *LDS_PTR(f32x4, stage + wr * 128 + (((2 * g + wc) ^ (wr & 7)) << 4)) =
            (f32x4){acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]};
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private region: program order is enough
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
        if (accumulate) {
          const f32x4 c4 = *(const f32x4*)cp;
          t += c4;
        }
        __builtin_nontemporal_store(t, (f32x4*)cp);     // next read by the optimizer, a whole backward later: streaming store
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next block overwrites the region
    }
  }
}

// a partial tile of the split tail into its workspace slot (row-major [TBM][TBN] fp32, whole tile, no bounds: the slot is private)
template <int MI, int NJ>
__device__ __forceinline__ void epilogue_f32_slot(f32x16 (&acc)[MI][NJ], float* slot, int ld, int wm, int wn, int lane, char* stage) {
  const int wr = lane & 31, wc = lane >> 5;
  const int rr = lane >> 3, rc = lane & 7;
#pragma unroll
  for (int i = 0; i < MI; i++) {
#pragma unroll
    for (int j = 0; j < NJ; j++) {
#pragma unroll
      for (int g = 0; g < 4; g++)
        *LDS_PTR(f32x4, stage + wr * 128 + (((2 * g + wc) ^ (wr & 7)) << 4)) =
            (f32x4){acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]};
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < 4; it++) {
        const int r = it * 8 + rr;
        const f32x4 t = *LDS_PTR(const f32x4, stage + r * 128 + ((rc ^ (r & 7)) << 4));
        // device-scope write-through store (sc1): the slot is read by a workgroup on another XCD, whose L2 is not coherent with this one;
        // a release FENCE would write back this XCD's whole L2 instead (measured: the fenced version was slower than the atomics)
        float* dst = slot + (int64_t)(wm * (MI * 32) + i * 32 + r) * ld + wn * (NJ * 32) + j * 32 + rc * 4;
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(t) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
}

template <int WM, int WN, int MI, int NJ>
__global__ __launch_bounds__(64 * WM * WN) void gemm_kk_kernel(GroupParams gp) {
  constexpr int TBM = WM * MI * 32, TBN = WN * NJ * 32, NW = WM * WN;
  constexpr int HA = TBM * 64, HB = TBN * 64, H = HA + HB;           // bytes of one ring slot (a 32-deep K half)
  constexpr int PA = TBM / 16 / NW, PB = TBN / 16 / NW, PP = PA + PB; // 1-KiB DMA pieces per wave and half
  constexpr int DSTRIDE = (2 * MI) / PP > 0 ? (2 * MI) / PP : 1;
  static_assert(PA >= 1 && PB >= 1 && PP <= 2 * MI && NW * EP32_WAVE_BYTES <= H && NJ == 2, "piece schedule / staging in a ring slot");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
  MMDIT_YOUNG_HALF_PRIO();
  const int wm = wave / WN, wn = wave % WN;
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  const int G = (int)gridDim.x;

  int pos = (int)blockIdx.x, end = total_work(gp);
  if (gp.tail_first >= 0) {   // balanced tail: this workgroup's share of the tail units (none if its first tile is a long one)
    const int e = xcd_chunk((int)blockIdx.x, gp.full_tiles) - gp.tail_first, E = gp.tail_G - gp.tail_first;
    const int left = (gp.total_tiles - gp.full_tiles) * gp.split_k - e;
    end = gp.full_tiles + (e < 0 || left <= 0 ? 0 : (left + E - 1) / E) * gp.tail_G;
  }
  end = uni(end);

  // ---- DMA cursor: (item, half) the next issued half belongs to; wave-uniform ---------------------------------------
  Item cit = item_at(gp, pos, end);
  while (cit.valid && cit.h0 >= cit.h1) cit = item_at(gp, cit.pos + G, end);
  int ch = cit.h0;
  uint32_t va[PA], vb[PB];
  const char* sa = nullptr;
  const char* sb = nullptr;
  int64_t stepa = 0, stepb = 0;
  auto cursor_setup = [&]() {
    const Problem& q = gp.p[cit.pi];
#pragma unroll
    for (int i = 0; i < PA; i++) va[i] = piece_voff<true, TBM>(wave * PA + i, lane, q.lda, cit.tm * TBM, q.M);
#pragma unroll
    for (int i = 0; i < PB; i++) vb[i] = piece_voff<true, TBN>(wave * PB + i, lane, q.ldb, cit.tn * TBN, q.N);
    stepa = (int64_t)BKH * q.lda * 2;
    stepb = (int64_t)BKH * q.ldb * 2;
    sa = (const char*)q.A + ch * stepa;
    sb = (const char*)q.B + ch * stepb;
  };
  auto cursor_advance = [&]() {   // past the end of the stream the last half is requested again (into a free slot; never read)
    if (!cit.valid) return;
    if (ch + 1 < cit.h1) {
      ch++;
      sa += stepa;
      sb += stepb;
      return;
    }
    Item nx = cit;
    do nx = item_at(gp, nx.pos + G, end); while (nx.valid && nx.h0 >= nx.h1);
    if (nx.valid) {
      cit = nx;
      ch = cit.h0;
      cursor_setup();
    } else {
      cit.valid = false;
    }
  };
  auto issue_piece = [&](int q, int slot) {
    const uint32_t dst = lds0 + slot * H;
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
  int cslot = 0, dslot = 0;
  auto bump = [](int s) { return s + 1 == RING ? 0 : s + 1; };

  if (cit.valid) {
    cursor_setup();
#pragma unroll 1
    for (int s = 0; s < RING - 1; s++) {
#pragma unroll
      for (int q = 0; q < PP; q++) issue_piece(q, dslot);
      dslot = bump(dslot);
      cursor_advance();
    }
  }

  bf16x8 a[MI], b[2][NJ];
  auto half_sync = [&]() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 3) * PP) : "memory");
    __builtin_amdgcn_s_barrier();
  };
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
      for (int j = 0; j < NJ; j++) b[nx][j] = load_frag_h<true, TBN>(last ? nb : tb, wn * (NJ * 32) + j * 32, last ? 0 : ks + 1, lane);
#pragma unroll
      for (int i = 0; i < MI; i++) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[c][j], a[i], acc[i][j], 0, 0, 0);
        a[i] = load_frag_h<true, TBM>(last ? na : ta, wm * (MI * 32) + i * 32, last ? 0 : ks + 1, lane);
        const int q = ks * MI + i;   // compile-time after unrolling
        if (q % DSTRIDE == 0 && q / DSTRIDE < PP) {
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

  // the previous item, as far as its epilogue needs it
  int p_pi = 0, p_tm = 0, p_tn = 0, p_sk = 0, p_tile = 0;
  bool p_atomic = false, pending = false, first = true;
  __shared__ int s_ticket;
  auto run_epilogue = [&]() {
    char* stage = smem + dslot * H + wave * EP32_WAVE_BYTES;   // dslot: free until the next issue
    const Problem& q = gp.p[p_pi];
    if (p_atomic && gp.ws_slots) {
      // Partial tile of the split tail: no fp32 atomics (each 256x256 partial costs ~0.6 us of L2 atomic throughput for the WHOLE
      // launch).  Store it to the slice's workspace slot, publish (release fence + ticket); the last of the tile's slices to arrive sums
      // the slots in slice order -- deterministic -- and writes C.  No workgroup ever waits for another one.
      constexpr int TE = TBM * TBN;
      const int tt = p_tile - gp.full_tiles, S = gp.split_k;
      float* slots = gp.ws_slots + (int64_t)tt * S * TE;
      epilogue_f32_slot<MI, NJ>(acc, slots + (int64_t)p_sk * TE, TBN, wm, wn, lane, stage);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the write-through stores of this wave have reached memory
      __syncthreads();                                   // ... of every wave
      if (tid == 0) s_ticket = atomicAdd(gp.ws_count + tt, 1);   // (device-scope atomic, performed at the memory side)
      __syncthreads();
      if (s_ticket == S - 1) {               // (workgroup-uniform) every slice of this tile has been published
        float* C = (float*)q.C;
        const int m0 = p_tm * TBM, n0 = p_tn * TBN;
        // device-scope (sc1) loads past this XCD's L2, 8 chunks x up to 4 slices in flight per lane; the loads are issued from asm (the
        // compiler has no sc1 load), so their destinations are handed to it only through the wait that follows them
        constexpr int NCH = TE / 4 / (64 * NW);     // 16-byte chunks per lane (32 for the 256x256 tile)
        static_assert(NCH % 4 == 0, "chunk batches");
#pragma unroll 1
        for (int b0 = 0; b0 < NCH; b0 += 4) {
          f32x4 t[4];
#pragma unroll
          for (int u = 0; u < 4; u++) t[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
          for (int s0 = 0; s0 < S; s0 += 4) {
            f32x4 v[4][4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const int idx = tid + (b0 + u) * (64 * NW), r = idx / (TBN / 4), c = (idx % (TBN / 4)) * 4;
#pragma unroll
              for (int k = 0; k < 4; k++) {
                const float* src = slots + (int64_t)min(s0 + k, S - 1) * TE + (int64_t)r * TBN + c;
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
                if (s0 + k < S) t[u] += v[u][k];
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int idx = tid + (b0 + u) * (64 * NW), r = idx / (TBN / 4), c = (idx % (TBN / 4)) * 4;
            if (m0 + r < q.M && n0 + c < q.N) {
              float* cp = C + (int64_t)(m0 + r) * q.ldc + n0 + c;
              if (gp.accumulate) t[u] += *(const f32x4*)cp;
              __builtin_nontemporal_store(t[u], (f32x4*)cp);
            }
          }
        }
        if (tid == 0) gp.ws_count[tt] = 0;   // ready for the next launch (stream order)
      }
      __syncthreads();                       // (s_ticket is reused by the next partial tile of this workgroup)
      return;
    }
    epilogue_f32_plain<MI, NJ>(acc, q, p_tm * TBM, p_tn * TBN, wm, wn, lane, stage, p_atomic, gp.accumulate != 0);
  };
  Item item = item_at(gp, pos, end);
  while (item.valid) {
    const int n = item.h1 - item.h0;
    if (n > 0) half_sync();
    else __builtin_amdgcn_s_barrier();
    if (pending) {   // the previous tile's epilogue, deferred to here: its stores drain under the MFMAs that follow
      run_epilogue();
      __builtin_amdgcn_s_barrier();   // staging reads done before the DMA below refills that slot
    }
    zero_acc();
    if (n > 0) {
      if (first) {   // fragments of a tile's first half are carried over from the previous tile, except at the start of the stream
        first = false;
#pragma unroll
        for (int j = 0; j < NJ; j++) b[0][j] = load_frag_h<true, TBN>(smem + HA, wn * (NJ * 32) + j * 32, 0, lane);
#pragma unroll
        for (int i = 0; i < MI; i++) a[i] = load_frag_h<true, TBM>(smem, wm * (MI * 32) + i * 32, 0, lane);
      }
      half_body();
#pragma unroll 1
      for (int u = 1; u < n; u++) {
        half_sync();
        half_body();
      }
    }
    pending = true;   // (an empty split-K slice still reaches the epilogue: it adds zeros)
    p_pi = item.pi; p_tm = item.tm; p_tn = item.tn; p_atomic = item.atomic; p_sk = item.sk; p_tile = item.tile;
    item = item_at(gp, item.pos + G, end);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing (unused) DMA pieces must land before the LDS is reused / released
  __builtin_amdgcn_s_barrier();                      // every wave has left the last half (its slot is the staging slot)
  if (pending) run_epilogue();
}

int launch_kk(const GroupParams& gp, hipStream_t s) {
  constexpr int smem = RING * (256 + 256) * 64;
  auto k = gemm_kk_kernel<2, 4, 4, 2>;
  static unsigned long long attr_done = 0;  // one bit per device; idempotent, a benign race only repeats the call
  if (!mmdit_device_once(attr_done)) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    mmdit_device_mark(attr_done);
  }
  const int work = total_work(gp);
  const int cu = mmdit_get_cu_budget();
  int grid = gp.persistent && work > cu ? cu : work;   // one persistent workgroup per CU (of the budget)
  // experiments (MMDIT_WGRAD_STREAM=1): leave CUs to the kernels of the main stream; only without the balanced tail (MMDIT_GEMM_KDEC=plain),
  // whose unit -> workgroup map is built for 256 workgroups
  static const char* g_env = mmdit_exp_env("MMDIT_GEMM_KK_GRID");
  if (g_env && gp.tail_first < 0 && atoi(g_env) > 0 && atoi(g_env) < grid) grid = atoi(g_env);
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), smem, s, gp);
  return mmdit_launch_status();
}

#endif   // MMDIT_PROBES

}  // namespace

// The product library launches ONE kernel of this file: the wide-slot kernel at 320 x 256 with the QKV epilogue (QK-RMSNorm + RoPE + joint-layout
// store) -- every other launch of the lean family goes to the 8-phase kernels of gemm8p.hip.  -DMMDIT_PROBES builds (tools/build_variant.sh) keep the
// kernels of rounds 2-3 selectable for same-box A/B runs (MMDIT_GEMM_8P=0 / 2, MMDIT_GEMM_WIDE=0, MMDIT_GEMM_KK=0).
int gemm::launch_lean_wgrad(const GroupParams& gp, hipStream_t s) {
#ifdef MMDIT_PROBES
  return launch_kk(gp, s);
#else
  (void)gp; (void)s;
  return MMDIT_ERR_SHAPE;
#endif
}

int gemm::launch_lean_cfg(int cfg, bool b_km, const GroupParams& gp, hipStream_t s) {
  if (gp.qk_on && cfg == CFG_320x256 && !b_km && gp.act == MMDIT_ACT_NONE) return launch_wide<2, 4, 5, 2, false, false, true>(gp, s);
#ifdef MMDIT_PROBES
  static const char* wide_env = getenv("MMDIT_GEMM_WIDE");
  if (gp.qk_on) {
    if (b_km || gp.act != MMDIT_ACT_NONE || (wide_env && !atoi(wide_env))) return MMDIT_ERR_SHAPE;
    if (cfg == CFG_256x256) return launch_wide<2, 4, 4, 2, false, false, true>(gp, s);
    return MMDIT_ERR_SHAPE;
  }
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
  if (gp.act == MMDIT_ACT_SWIGLU) {
    if (b_km) return MMDIT_ERR_ARG;
    if (cfg == CFG_320x256) return launch_lean<2, 4, 5, 2, false, true>(gp, s);
    if (cfg == CFG_256x256) return launch_lean<2, 4, 4, 2, false, true>(gp, s);
    return MMDIT_ERR_ARG;
  }
  if (cfg == CFG_320x256) return b_km ? launch_lean<2, 4, 5, 2, true>(gp, s) : launch_lean<2, 4, 5, 2, false>(gp, s);
  if (cfg == CFG_256x256) return b_km ? launch_lean<2, 4, 4, 2, true>(gp, s) : launch_lean<2, 4, 4, 2, false>(gp, s);
#endif
  return MMDIT_ERR_SHAPE;
}
