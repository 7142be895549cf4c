// LDS-DMA MFMA GEMM kernels for gfx950 (bf16 operands, K % 64 == 0): the fast path of mmdit_gemm*.
//
// Operands go HBM/L2 -> LDS directly with global_load_lds (16 B per lane): no VGPR staging, no ds_write traffic.
// The LDS image of a DMA is lane-linear (wave-uniform base + lane*16), so tiles are unpadded and the
// bank-conflict swizzle is applied to the per-lane SOURCE address and mirrored in the fragment reads:
//   row-major half-tile [R][32] bf16 (64 B rows):    16-B piece p of row r   lives at slot p ^ ((r>>2)&3)
//   k-major  half-tile [32][R] bf16 (2R-byte k-rows): 16-B piece p of k-row k lives at slot p ^ ((k&3)<<2)
// (ds_read_b128 fragments of the first and ds_read_b64_tr_b16 fragments of the second are then conflict-free.)
// Rows beyond M / N are clamped to valid addresses (their results are never stored).
//
// Pipeline.  A workgroup consumes a STREAM of 32-wide K "halves" that runs across its output tiles (persistent
// tile loop, or stream-K segments).  The halves live in a 4-slot LDS ring; the DMA cursor runs 3 halves ahead of
// the MFMAs.  One LDS-DMA piece costs its wave ~60-180 issue cycles, so the pieces of half g+3 are spread
// BETWEEN the MFMA rows of half g (issued in a burst they serialise with the MFMAs: measured 81 us of DMA +
// 130 us of MFMA = 219 us for the w12 GEMM).  Per half: s_waitcnt vmcnt(<pieces of the younger halves>) + one
// s_barrier.  Fragment reads are software-pipelined in registers (B double-buffered, A reloaded behind its
// last MFMA) and get counted lgkmcnt waits.  The epilogue of a tile is deferred until the first half of the
// next tile has landed, so its stores drain under the following MFMAs; it stages C through a wave-private,
// XOR-swizzled 4-KiB block of the ring slot that is free at that point (one extra barrier per tile).
//
// Tile configurations (threads = 64 * WM * WN, each wave owns an (MI*32) x (NJ*32) sub-tile):
//   128x128: 4 waves 2x2, 2x2 accumulators,  64 KB LDS (2 workgroups / CU)
//   256x128: 8 waves 4x2, 2x2 accumulators, 128 KB LDS
//   256x256: 8 waves 2x4, 4x2 accumulators, 128 KB LDS  -- half the L2->CU bytes per FLOP of 128x128
#include "gemm_tile.h"

using namespace gemm;

namespace {


template <int WM, int WN, int MI, int NJ, bool A_KM, bool B_KM, typename TC, typename TAUX, int FP8 = 0, bool SWIGLU = false>
__global__ __launch_bounds__(64 * WM * WN) void gemm_dma_kernel(GroupParams gp) {
  // FP8: e4m3 operands (row-major only).  A ring slot still holds 64 B per row = 64 fp8 values, so the whole DMA / ring /
  // barrier machinery is byte-identical; a half is ONE 64-wide k-step of the MX matrix instruction
  // v_mfma_scale_f32_32x32x64_f8f6f4 (twice the bf16 MFMA rate; 32 K bytes per lane and operand) with unit block scales (E8M0
  // 127): the per-tensor scales of the operands are applied in the epilogue.  Half the operand bytes per FLOP and half the
  // MFMA time of the bf16 kernel.
  // FP8 = 1: per-tensor scales (applied in the epilogue, unit block scales); FP8 = 2 (MX): E8M0 block scales fed to the instruction.
  constexpr bool MX = FP8 == 2;
  static_assert(!FP8 || (!A_KM && !B_KM), "fp8 operands are row-major");
  static_assert(!SWIGLU || (!A_KM && !B_KM && WN == 4 && NJ == 2 && sizeof(TC) == 2 && sizeof(TAUX) == 2), "SwiGLU epilogue: bf16, row-major, 256-column tile");
  constexpr int ESZ = FP8 ? 1 : 2, KSTEPS = FP8 ? 1 : 2;   // MFMA k-steps per half
  using frag_t = typename std::conditional<(FP8 != 0), i32x8, bf16x8>::type;
  constexpr int TBM = WM * MI * 32, TBN = WN * NJ * 32, NW = WM * WN;
  constexpr int SCB = MX ? 2048 : 0;                           // MX: E8M0 block scales of the half, A rows at +0, B rows at +1024 (layout: mx_scale_index)
  constexpr int HA = TBM * 64, HB = TBN * 64, H = HA + HB + SCB;   // bytes of one ring slot
  constexpr int PA = TBM / 16 / NW, PB = TBN / 16 / NW, PP = PA + PB;   // 1-KiB DMA pieces per wave and half
  constexpr int DSTRIDE = (2 * MI) / PP > 0 ? (2 * MI) / PP : 1;   // bf16: MFMA rows between two pieces
  static_assert(PA >= 1 && PB >= 1 && PP <= 2 * MI && PP <= 4, "piece schedule");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  MMDIT_YOUNG_HALF_PRIO();
  const int wm = wave / WN, wn = wave % WN;
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  // epilogue staging: the ring slot that is free between the barrier of half_sync and the next DMA issue (when a slot
  // is large enough for all waves), else a dedicated region behind the ring
  constexpr bool STAGE_IN_RING = NW * EP32_WAVE_BYTES <= H;

  int pos, end;
  if (gp.stream_k) {
    const long long U = gp.total_units, G = gridDim.x;
    pos = (int)(blockIdx.x * U / G);
    end = (int)((blockIdx.x + 1) * U / G);
  } else {
    pos = blockIdx.x;
    end = total_work(gp);
    if (gp.tail_first >= 0) {   // balanced tail: this workgroup's share of the tail units (none if its first tile is a long one)
      const int e = xcd_chunk((int)blockIdx.x, gp.full_tiles) - gp.tail_first, E = gp.tail_G - gp.tail_first;
      const int left = (gp.total_tiles - gp.full_tiles) * gp.split_k - e;
      end = gp.full_tiles + (e < 0 || left <= 0 ? 0 : (left + E - 1) / E) * gp.tail_G;
    }
  }

  // ---- DMA cursor: (item, half) the next issued half belongs to --------------------------------------------
  Item cit = item_at(gp, pos, end);
  while (cit.valid && cit.h0 >= cit.h1) cit = item_at(gp, next_pos(gp, cit), end);
  int ch = cit.h0;
  uint32_t va[PA], vb[PB];
  const char* sa = nullptr;
  const char* sb = nullptr;
  int64_t stepa = 0, stepb = 0;
  // MX mode (fp8 operands with E8M0 block scales): two extra DMA pieces per half, issued by wave 0 -- the A tile's and the B tile's
  // scale bytes of the half (TBM * 2 resp. TBN * 2 contiguous bytes of the [K/64][rows_pad][..] tensors; the lanes beyond that
  // re-read the last 16 bytes).  Wave-uniform bases in SGPRs, advanced by rows_pad * 2 bytes per half.
  constexpr bool mx = MX;
  const bool mx0 = MX && wave == 0;
  const char* ssa = nullptr;
  const char* ssb = nullptr;
  int64_t sstepa = 0, sstepb = 0;
  uint32_t vsb = 0;   // per-lane byte offset of the B scale piece (SwiGLU: two 256-byte runs, gate and up rows)
  int cseg = 0, cseg_len = 0;      // implicit-GEMM convolution: halves left in / per kernel row (3 taps x cC/32 halves are contiguous)
  int64_t crow_jump = 0;           // extra bytes when the K index moves to the next kernel row
  auto cursor_setup = [&]() {
    const Problem& q = gp.p[cit.pi];
    if (!A_KM && q.conv_mode) {
#pragma unroll
      for (int i = 0; i < PA; i++) va[i] = conv_voff(wave * PA + i, lane, q, cit.tm * TBM);
    } else {
#pragma unroll
      for (int i = 0; i < PA; i++) va[i] = piece_voff<A_KM, TBM, ESZ>(wave * PA + i, lane, q.lda, cit.tm * TBM, q.M);
    }
#pragma unroll
    for (int i = 0; i < PB; i++) {
      if constexpr (SWIGLU) vb[i] = swiglu_voff<ESZ>(wave * PB + i, lane, q.ldb, cit.tn, q.N >> 1);
      else vb[i] = piece_voff<B_KM, TBN, ESZ>(wave * PB + i, lane, q.ldb, cit.tn * TBN, q.N);
    }
    stepa = A_KM ? (int64_t)BKH * q.lda * 2 : BKH * 2;
    stepb = B_KM ? (int64_t)BKH * q.ldb * 2 : BKH * 2;
    sa = (const char*)q.A + ch * stepa;
    sb = (const char*)q.B + ch * stepb;
    if constexpr (MX) {
      sstepa = (((int64_t)q.M + 127) & ~(int64_t)127) * 2;
      sstepb = (((int64_t)q.N + 127) & ~(int64_t)127) * 2;
      ssa = (const char*)q.scale_a + ch * sstepa + (int64_t)cit.tm * TBM * 2;
      ssb = (const char*)q.scale_b + ch * sstepb + (SWIGLU ? 0 : (int64_t)cit.tn * TBN * 2);
      if constexpr (SWIGLU) vsb = (uint32_t)(((lane & 16 ? (q.N >> 1) : 0) + cit.tn * 128) * 2 + (lane & 15) * 16);   // gate rows [128 tn, +128), up rows [h + 128 tn, +128)
      else vsb = (uint32_t)(lane * 16 < TBN * 2 ? lane * 16 : TBN * 2 - 16);
    }
    cseg = 0;
    if (!A_KM && q.conv_mode) {
      cseg_len = 3 * (q.cC / BKH);
      const int kh = ch / cseg_len, within = ch - kh * cseg_len, o = q.conv_mode == 2 ? 1 : 0;
      crow_jump = ((int64_t)q.cWp - 3) * q.cC * 2;
      sa = (const char*)q.A + (((int64_t)(kh + o) * q.cWp + o) * q.cC) * 2 + (int64_t)within * (BKH * 2);
      cseg = cseg_len - within;
    }
  };
  // after the last half of the stream the cursor stays where it is: the steady-state loop keeps issuing (it re-reads
  // that half into a free slot) so that the loop body is branch-free and exactly PP pieces are issued per half
  auto cursor_advance = [&]() {
    if (!cit.valid) return;
    if (ch + 1 < cit.h1) {
      ch++;
      sa += stepa;
      sb += stepb;
      if constexpr (MX) { ssa += sstepa; ssb += sstepb; }
      if (cseg && --cseg == 0) {   // convolution: next kernel row
        sa += crow_jump;
        cseg = cseg_len;
      }
      return;
    }
    Item nx = cit;
    do nx = item_at(gp, next_pos(gp, nx), end); while (nx.valid && nx.h0 >= nx.h1);
    if (nx.valid) {
      cit = nx;
      ch = cit.h0;
      cursor_setup();
    } else {
      cit.valid = false;
    }
  };
  auto issue_piece = [&](int q, int slot) {   // q: compile-time piece index of the cursor's half
    const uint32_t dst = lds0 + slot * H;
    if (q < PA) glds16(va[q], sa, dst + (wave * PA + q) * 1024);
    else glds16(vb[q - PA], sb, dst + HA + (wave * PB + (q - PA)) * 1024);
  };
  auto issue_scales = [&](int slot) {         // MX mode, wave 0: the scale bytes of the cursor's half
    if constexpr (MX)
      if (mx0) {
        const uint32_t dst = lds0 + slot * H + HA + HB;
        glds16((uint32_t)(lane * 16 < TBM * 2 ? lane * 16 : TBM * 2 - 16), ssa, dst);
        glds16(vsb, ssb, dst + 1024);
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
  auto run_epilogue = [&](const Item& it) {   // reads the accumulators only (the zeroing that follows is unconditional code)
    char* stage = smem + (STAGE_IN_RING ? dslot * H : RING * H) + wave * EP32_WAVE_BYTES;   // dslot: free until the next issue
    if (!(gp.debug & 8)) {
      const Problem& q = gp.p[it.pi];
      float alpha = 1.f;
      if constexpr (FP8 == 1) alpha = q.scale_a[0] * q.scale_b[0];   // (MX: the block scales went into the MFMAs)
      if constexpr (SWIGLU) {
        epilogue_swiglu<MI, FP8 == 1>(acc, q, it.tm * TBM, it.tn, wm, wn, lane, stage, alpha);
        return;
      }
      bool fast = false;
      if constexpr (sizeof(TC) == 2 && NJ == 2)
        fast = !q.aux && !q.residual && !q.gate && !gp.accumulate && !it.atomic && (q.N & 7) == 0 && (q.ldc & 7) == 0 && ((uintptr_t)q.C & 15) == 0 && !(gp.debug & 64);
      if (fast) epilogue_bf16<MI, NJ>(acc, q, gp, it.tm * TBM, it.tn * TBN, wm, wn, lane, stage, alpha);
      else epilogue32<TC, TAUX, MI, NJ>(acc, q, gp, it.tm * TBM, it.tn * TBN, wm, wn, lane, it.sk, stage, it.atomic, alpha, A_KM && B_KM);
    }
    else if (acc[0][0][0] == 12345.678f) ((float*)gp.p[it.pi].C)[0] = 0.f;   // ablation: keep the accumulators live
  };

  const bool any = cit.valid;   // false: this workgroup only has empty split-K slices (or nothing)
  if (any) {
    cursor_setup();
#pragma unroll 1
    for (int s = 0; s < RING - 1; s++) {
#pragma unroll
      for (int q = 0; q < PP; q++) issue_piece(q, dslot);
      issue_scales(dslot);
      dslot = bump(dslot);
      cursor_advance();
    }
  }

  // Fragment registers are carried from half to half: the last k-step of a half already reads the first fragments
  // of the NEXT half (which half_sync made visible one half early), so no half starts with an exposed LDS burst.
  // transposed product: acc[i][j] = (B_j A_i^T) -> rows = n, lanes = m
  frag_t a[MI], b[2][NJ];
  // MX: the E8M0 bytes of this lane's 32 K bytes, for all of the wave's A row blocks in ONE register (the layout puts the four
  // 32-row blocks of a 128-row group next to each other; op_sel picks the block in the MFMA), likewise for B.  Per-tensor mode: 1.0
  // (E8M0 127) in every byte.
  constexpr int SCBN = SWIGLU ? 2 : 1;
  int sca = 0x7f7f7f7f, scb[SCBN] = {0x7f7f7f7f};   // (unit scales in every byte: op_sel picks a byte)
  auto ldSA = [&](const char* slot) -> int {
    if constexpr (!MX) return 0x7f7f7f7f;
    const char* p = slot + HA + HB + ((wm * MI) >> 2) * 256 + (lane & 31) * 8 + (lane >> 5) * 4 + ((wm * MI) & 3);
    if constexpr (MI == 4) return *LDS_PTR(const int, p);
    else return *LDS_PTR(const unsigned short, p);
  };
  auto ldSB = [&](const char* slot, int j) -> int {
    if constexpr (!MX) return 0x7f7f7f7f;
    if constexpr (SWIGLU) return *LDS_PTR(const unsigned char, slot + HA + HB + 1024 + j * 256 + (lane & 31) * 8 + (lane >> 5) * 4 + wn);
    else return *LDS_PTR(const unsigned short, slot + HA + HB + 1024 + ((wn * NJ) >> 2) * 256 + (lane & 31) * 8 + (lane >> 5) * 4 + ((wn * NJ) & 3));
  };
  static_assert(!MX || ((MI == 4 || MI == 2) && NJ == 2), "scale packing");
  auto ldA = [&](const char* t, int r0, int ks) -> frag_t { if constexpr (FP8) return load_frag8(t, r0, lane); else return load_frag_h<A_KM, TBM>(t, r0, ks, lane); };
  auto ldB = [&](const char* t, int r0, int ks) -> frag_t { if constexpr (FP8) return load_frag8(t, r0, lane); else return load_frag_h<B_KM, TBN>(t, r0, ks, lane); };
  // one K half: multiply slot done&3 while the PP pieces of the cursor's half go into slot issued&3 (= (done-1)&3,
  // which every wave left before the barrier of half_sync).  No data-dependent branch inside.
  auto half_sync = [&]() {
    // exactly RING-1 halves are in flight here: the current half and the next one have landed once only the pieces of
    // the RING-3 youngest halves may still be outstanding (loads retire in order; stores only make the wait conservative)
    if (MX && mx0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 3) * (PP + 2)) : "memory");   // (wave 0 also carries the two scale pieces)
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 3) * PP) : "memory");
    __builtin_amdgcn_s_barrier();   // everyone's pieces landed; everyone left half done-1, whose slot the DMA below refills
  };
  // one K half: multiply slot cslot while the PP pieces of the cursor's half go into slot dslot (which every wave left before
  // the barrier of half_sync).  No data-dependent branch inside.
  auto half_body = [&]() {
    const int nslot = bump(cslot);
    const char* ta = smem + cslot * H;
    const char* tb = ta + HA;
    const char* na = smem + nslot * H;   // (after the last half of the stream: read, never used)
    const char* nb = na + HA;
    if constexpr (FP8) {
      // fp8 (one 64-wide k-step per half): fragments are NOT carried from half to half.  With 32-byte fragments the carried (in
      // place reloaded) registers come out of the compiler as two sets plus copies (loop phis of 8-register tuples built from two
      // ds_read_b128 are not coalesced): 128 accumulators + 2 x 48 spill accumulators inside the K loop once the MX scale
      // registers are added (3x slower), and even without them the copies cost (per-tensor w12 247 us).  Instead every half loads
      // its own fragments: both B fragments and the first A fragment up front (their latency is covered by the other wave of the
      // SIMD), the next A fragment under the MFMAs of the current one -- 32 fragment registers live, no phis (w12 190 us).
      static_assert(!FP8 || MI * NJ >= PP, "piece schedule (fp8)");
      constexpr int S = (MI * NJ) / PP > 0 ? (MI * NJ) / PP : 1;
      frag_t bq[NJ], acur = ldA(ta, wm * (MI * 32), 0), anxt;
#pragma unroll
      for (int j = 0; j < NJ; j++) bq[j] = ldB(tb, wn * (NJ * 32) + j * 32, 0);
      sca = ldSA(ta);
#pragma unroll
      for (int j = 0; j < SCBN; j++) scb[j] = ldSB(ta, j);
#pragma unroll
      for (int i = 0; i < MI; i++) {
        if (i + 1 < MI) anxt = ldA(ta, wm * (MI * 32) + (i + 1) * 32, 0);
#pragma unroll
        for (int j = 0; j < NJ; j++) {
          __builtin_amdgcn_sched_barrier(0);
          acc[i][j] = mx_mfma(bq[j], acur, acc[i][j], SWIGLU ? 0 : j, scb[SWIGLU ? j : 0], i, sca);
          const int q = i * NJ + j;   // compile-time after unrolling
          if (q % S == 0 && q / S < PP) {
            __builtin_amdgcn_sched_barrier(0);
            issue_piece(q / S, dslot);
          }
        }
        acur = anxt;
      }
      __builtin_amdgcn_sched_barrier(0);
      issue_scales(dslot);
    } else {
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ks++) {
        const int c = ks & 1, nx = c ^ 1;
        const bool last = ks + 1 == KSTEPS;
        // B fragments are double-buffered, each A fragment is reloaded right after the last MFMA that reads it; the
        // order is pinned so every ds_read has MFMAs of cover and gets a counted lgkmcnt wait.
#pragma unroll
        for (int j = 0; j < NJ; j++) b[nx][j] = ldB(last ? nb : tb, wn * (NJ * 32) + j * 32, last ? 0 : ks + 1);
#pragma unroll
        for (int i = 0; i < MI; i++) {
          __builtin_amdgcn_sched_barrier(0);
          MMDIT_PRIO(1);
#pragma unroll
          for (int j = 0; j < NJ; j++)
            if constexpr (!FP8) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[c][j], a[i], acc[i][j], 0, 0, 0);
          MMDIT_PRIO(0);
          a[i] = ldA(last ? na : ta, wm * (MI * 32) + i * 32, last ? 0 : ks + 1);
          const int q = ks * MI + i;   // compile-time after unrolling
          if (q % DSTRIDE == 0 && q / DSTRIDE < PP) {
            __builtin_amdgcn_sched_barrier(0);
            issue_piece(q / DSTRIDE, dslot);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    dslot = bump(dslot);
    cslot = nslot;
    cursor_advance();
  };
  auto half_body_nocompute = [&]() {   // ablation (MMDIT_GEMM_DEBUG & 2): DMA stream only
#pragma unroll
    for (int q = 0; q < PP; q++) issue_piece(q, dslot);
    issue_scales(dslot);
    dslot = bump(dslot);
    cslot = bump(cslot);
    cursor_advance();
  };

  Item item = item_at(gp, pos, end), prev = item;
  bool pending = false;
  const bool compute = !(gp.debug & 2);
  bool first = true;
  while (item.valid) {
    const int n = item.h1 - item.h0;
    if (n > 0) half_sync();
    else if (STAGE_IN_RING) __builtin_amdgcn_s_barrier();
    // the previous tile's epilogue is deferred to here: its stores drain under the MFMAs that follow
    if (pending) {
      run_epilogue(prev);
      if (STAGE_IN_RING) __builtin_amdgcn_s_barrier();   // staging reads done before the DMA below refills that slot
    }
    zero_acc();
    if (n > 0) {
      if (compute) {
        // the fragments of a tile's first half: carried over from the previous tile, except at the start of the stream
        if (first) {
          first = false;
          const char* ta = smem;
          if constexpr (!FP8) {
#pragma unroll
            for (int j = 0; j < NJ; j++) b[0][j] = ldB(ta + HA, wn * (NJ * 32) + j * 32, 0);
#pragma unroll
            for (int i = 0; i < MI; i++) a[i] = ldA(ta, wm * (MI * 32) + i * 32, 0);
          }
        }
        half_body();
#pragma unroll 1
        for (int u = 1; u < n; u++) {
          half_sync();
          half_body();
        }
      } else {
#pragma unroll 1
        for (int u = 0; u < n; u++) {
          if (u) half_sync();
          half_body_nocompute();
        }
      }
    }
    pending = true;   // (an empty split-K slice still contributes its epilogue terms)
    prev = item;
    item = item_at(gp, next_pos(gp, item), end);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing (unused) DMA pieces must land before the LDS is reused / released
  if (STAGE_IN_RING) __builtin_amdgcn_s_barrier();   // every wave has left the last half (its slot is the staging slot)
  if (pending) run_epilogue(prev);
}

template <int WM, int WN, int MI, int NJ, bool A_KM, bool B_KM, typename TC, typename TAUX, int FP8 = 0, bool SWIGLU = false>
int launch_cfg(const GroupParams& gp, hipStream_t s) {
  constexpr int slot = (WM * MI * 32 + WN * NJ * 32) * 64 + (FP8 == 2 ? 2048 : 0);
  constexpr int smem = RING * slot + (WM * WN * EP32_WAVE_BYTES <= slot ? 0 : WM * WN * EP32_WAVE_BYTES);
  auto k = gemm_dma_kernel<WM, WN, MI, NJ, A_KM, B_KM, TC, TAUX, FP8, SWIGLU>;
  static unsigned long long attr_done = 0;  // one bit per device; idempotent, a benign race only repeats the call
  if (!mmdit_device_once(attr_done)) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    mmdit_device_mark(attr_done);
  }
  // one resident workgroup per slot (256 CUs x workgroups that fit per CU by LDS)
  const int slots = mmdit_get_cu_budget() * (smem <= 80 * 1024 ? 2 : 1);
  const int work = total_work(gp);
  const int grid = gp.stream_k ? slots : (gp.persistent && work > slots ? slots : work);
  hipLaunchKernelGGL(k, dim3(grid), dim3(64 * WM * WN), smem, s, gp);
  return mmdit_launch_status();
}

template <bool A_KM, bool B_KM, typename TC, typename TAUX>
int by_cfg(int cfg, const GroupParams& gp, hipStream_t s) {
  if (cfg == CFG_128x128) return launch_cfg<2, 2, 2, 2, A_KM, B_KM, TC, TAUX>(gp, s);
  if (cfg == CFG_256x128) return launch_cfg<4, 2, 2, 2, A_KM, B_KM, TC, TAUX>(gp, s);
  if (cfg == CFG_256x256) return launch_cfg<2, 4, 4, 2, A_KM, B_KM, TC, TAUX>(gp, s);
  return MMDIT_ERR_ARG;
}

template <typename TC, typename TAUX>
int by_layout(int cfg, bool a_km, bool b_km, const GroupParams& gp, hipStream_t s) {
  if (!a_km && !b_km) return by_cfg<false, false, TC, TAUX>(cfg, gp, s);
  if (!a_km && b_km) return by_cfg<false, true, TC, TAUX>(cfg, gp, s);
  if (a_km && b_km) return by_cfg<true, true, TC, TAUX>(cfg, gp, s);
  return MMDIT_ERR_DTYPE;
}

}  // namespace

int gemm::launch_dma(int cfg, bool a_km, bool b_km, int c_dtype, int aux_dtype, bool fp8, const GroupParams& gp, hipStream_t s) {
  if (gp.act == MMDIT_ACT_SWIGLU) {   // packed w12 GEMM with the activation in the epilogue (gemm.hip has checked the rest)
    if (a_km || b_km || (c_dtype != MMDIT_BF16 && !(c_dtype == MMDIT_FP8 && fp8 && gp.mx)) || aux_dtype != MMDIT_BF16 || cfg != CFG_256x256) return MMDIT_ERR_DTYPE;
    if (fp8 && gp.mx) return launch_cfg<2, 4, 4, 2, false, false, bf16_t, bf16_t, 2, true>(gp, s);
    return fp8 ? launch_cfg<2, 4, 4, 2, false, false, bf16_t, bf16_t, 1, true>(gp, s) : launch_cfg<2, 4, 4, 2, false, false, bf16_t, bf16_t, 0, true>(gp, s);
  }
  if (fp8) {   // e4m3 operands: row-major x row-major, bf16 or fp32 output (aux, if any, in the output dtype)
    if (a_km || b_km || aux_dtype != c_dtype || cfg == CFG_256x128) return MMDIT_ERR_DTYPE;
    if (gp.mx) {
      if (c_dtype == MMDIT_BF16) return cfg == CFG_128x128 ? launch_cfg<2, 2, 2, 2, false, false, bf16_t, bf16_t, 2>(gp, s) : launch_cfg<2, 4, 4, 2, false, false, bf16_t, bf16_t, 2>(gp, s);
      if (c_dtype == MMDIT_F32) return cfg == CFG_128x128 ? launch_cfg<2, 2, 2, 2, false, false, float, float, 2>(gp, s) : launch_cfg<2, 4, 4, 2, false, false, float, float, 2>(gp, s);
      return MMDIT_ERR_DTYPE;
    }
    if (c_dtype == MMDIT_BF16) return cfg == CFG_128x128 ? launch_cfg<2, 2, 2, 2, false, false, bf16_t, bf16_t, 1>(gp, s) : launch_cfg<2, 4, 4, 2, false, false, bf16_t, bf16_t, 1>(gp, s);
    if (c_dtype == MMDIT_F32) return cfg == CFG_128x128 ? launch_cfg<2, 2, 2, 2, false, false, float, float, 1>(gp, s) : launch_cfg<2, 4, 4, 2, false, false, float, float, 1>(gp, s);
    return MMDIT_ERR_DTYPE;
  }
  if (c_dtype == MMDIT_F32 && aux_dtype == MMDIT_F32) return by_layout<float, float>(cfg, a_km, b_km, gp, s);
  if (c_dtype == MMDIT_F32 && aux_dtype == MMDIT_BF16) return by_layout<float, bf16_t>(cfg, a_km, b_km, gp, s);
  if (c_dtype == MMDIT_BF16 && aux_dtype == MMDIT_BF16) return by_layout<bf16_t, bf16_t>(cfg, a_km, b_km, gp, s);
  if (c_dtype == MMDIT_BF16 && aux_dtype == MMDIT_F32) return by_layout<bf16_t, float>(cfg, a_km, b_km, gp, s);
  return MMDIT_ERR_DTYPE;
}
