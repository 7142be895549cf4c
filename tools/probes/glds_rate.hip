// Operand-stream rate probe: how fast can ONE CU pull GEMM operand tiles, by path and by access shape?
//   path 0: global_load_lds_dwordx4 (LDS-DMA, what gemm_dma.hip uses)      path 1: global_load_dwordx4 -> VGPR (no LDS)
//   path 2: global_load_dwordx4 -> VGPR -> ds_write_b128
//   shape: bytes of one matrix row covered by one step: 64 (two 32-wide bf16 K halves, the current ring slot), 128 (one 64-wide
//          K step = a full cache line per row), 256, or 0 = fully contiguous 1-KiB pieces.
// 256 workgroups x 8 waves, one per CU; a step moves 32 KiB per workgroup (256 rows x 128 B or the equivalent).  `share` = number
// of distinct row blocks the 32 workgroups of an XCD walk (1: everything but the first touch is an L2 hit; 32: pure streaming).
// hipcc --offload-arch=gfx950 -O3 -o glds_rate glds_rate.hip && ./glds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

__device__ __forceinline__ void glds16(const char* gptr, uint32_t lds_dst_) {
  const uint32_t lds_dst = __builtin_amdgcn_readfirstlane(lds_dst_);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_dst) : "memory", "m0");
}

template <int PATH, int ROWB, int DEPTH, bool BARRIER = false, int NMFMA = 0>
__global__ __launch_bounds__(512) void stream(const char* __restrict__ base, long ld, int kbytes, int nblocks, int share, int steps, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  // steps per 256-row block: every step moves 32 KiB
  const int spb = (int)((256 * (long)kbytes) / 32768);
  int blk = (xcd * 13 + (j % share)) % nblocks, pos = 0, slot = 0;
  u32x4 acc = {0, 0, 0, 0};
  typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
  typedef __attribute__((ext_vector_type(16))) float f32x16;
  f32x16 macc[4];
  bf16x8 fa, fb;
  for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) macc[i][r] = 0.f;
  for (int e = 0; e < 8; e++) { fa[e] = (__bf16)(float)(threadIdx.x + e); fb[e] = (__bf16)(float)(blockIdx.x + e); }
  auto addr = [&](int q) -> const char* {
    const char* rowbase = base + (long)blk * 256 * ld;
    long off;
    if (ROWB == 64) off = (long)(wave * 32 + (q & 1) * 16 + (lane >> 2)) * ld + (long)pos * 128 + (q >> 1) * 64 + (lane & 3) * 16;
    else if (ROWB == 128) off = (long)(wave * 32 + q * 8 + (lane >> 3)) * ld + (long)pos * 128 + (lane & 7) * 16;
    else if (ROWB == 256) off = (long)((pos & 1) * 128 + wave * 16 + q * 4 + (lane >> 4)) * ld + (long)(pos >> 1) * 256 + (lane & 15) * 16;
    else off = (long)pos * 32768 + (wave * 4 + q) * 1024 + lane * 16;     // the block's rows as one contiguous slab (ld == kbytes)
    return rowbase + off;
  };
  auto advance = [&]() {
    slot = (slot + 1) & 3;
    if (++pos >= spb) { pos = 0; blk += share; if (blk >= nblocks) blk -= nblocks; }
  };
  if constexpr (PATH == 0) {
    for (int s = 0; s < steps; s++) {
#pragma unroll
      for (int q = 0; q < 4; q++) glds16(addr(q), lds0 + slot * 32768 + (wave * 4 + q) * 1024);
      if constexpr (NMFMA > 0) {   // a compute segment between the issue and the wait (register-only MFMAs)
#pragma unroll
        for (int i = 0; i < NMFMA; i++) macc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, macc[i & 3], 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (DEPTH - 1)) : "memory");
      if constexpr (BARRIER) __builtin_amdgcn_s_barrier();
      advance();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    u32x4 va[4], vb[4];
    auto load = [&](u32x4 (&v)[4]) {
#pragma unroll
      for (int q = 0; q < 4; q++) v[q] = *(const u32x4*)addr(q);
      advance();
    };
    auto consume = [&](u32x4 (&v)[4]) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (PATH == 1) { acc[0] ^= v[q][0]; acc[1] ^= v[q][1]; acc[2] ^= v[q][2]; acc[3] ^= v[q][3]; }
        else *LDS_PTR(u32x4, smem + slot * 32768 + (wave * 4 + q) * 1024 + lane * 16) = v[q];
      }
    };
    load(va);
    for (int s = 1; s + 1 < steps; s += 2) {
      load(vb);
      consume(va);
      load(va);
      consume(vb);
    }
    consume(va);
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u || macc[0][0] + macc[1][1] + macc[2][2] + macc[3][3] == 12345.f) sink[0] = acc[0];
}

// ---- GEMM-shaped skeleton: the real main loop's resource mix without the tile bookkeeping ---------------------------------
// One step = a 32-wide K half of a 256x256 tile: 32 KiB of operands by LDS-DMA into a 4-slot ring (ROWB 64: 16 rows x 64 B per
// piece, the current layout; 128: 8 rows x 128 B per piece, same bytes), NW waves in a (WM x WN) grid, each with an (MI*32) x (NJ*32)
// accumulator tile, fragments by ds_read_b128 from the slot that landed two steps ago (real data dependence, conflict-free
// swizzled addresses), one vmcnt + s_barrier per step.  READS = false drops the LDS reads (operands stay in registers).
template <int WM, int WN, int MI, int NJ, int ROWB, bool READS, int DEPTH, int ABLOCKS = 0, int PF = 0>
__global__ __launch_bounds__(64 * WM * WN) void gemm_like(const char* __restrict__ base, long ld, int kbytes, int nblocks, int share, int steps, float* sink) {
  typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
  typedef __attribute__((ext_vector_type(16))) float f32x16;
  constexpr int NW = WM * WN, PP = 32 / NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  const int spb = ABLOCKS > 0 ? 24 : (int)((256 * (long)kbytes) / 32768);
  int blk = (xcd * 13 + (j % share)) % nblocks, pos = 0, slot = 0;
  f32x16 acc[MI][NJ];
  for (int i = 0; i < MI; i++) for (int jj = 0; jj < NJ; jj++) for (int r = 0; r < 16; r++) acc[i][jj][r] = 0.f;
  bf16x8 a[MI], b[2][NJ];
  for (int i = 0; i < MI; i++) for (int e = 0; e < 8; e++) a[i][e] = (__bf16)(float)(lane + e + i);
  for (int jj = 0; jj < NJ; jj++) for (int e = 0; e < 8; e++) { b[0][jj][e] = (__bf16)(float)(lane - e + jj); b[1][jj][e] = b[0][jj][e]; }
  // ABLOCKS > 0: realistic operand sharing of a rasterised tile schedule -- an XCD's 32 workgroups are 4 m-tiles x 8 n-tiles; the A half
  // (pieces 0..15, 128 rows x 64-B... here: rows 0..127 of the step) comes from a panel shared by the 8 workgroups of an m-tile and
  // NEW for every tile (cycling through ABLOCKS panels of 256 rows: 102 = a 40 MB activation tensor, L2 misses served by the
  // Infinity Cache; 1000 = 390 MB, HBM), the B half from a panel shared by 4 workgroups out of a small L2-resident weight set.
  int tile = 0;
  const int tmi = j >> 3, tni = j & 7;
  auto panel = [&](int p, int t) -> long {
    if (ABLOCKS == 0) return blk;
    if (p < 16) return ((long)t * 32 + xcd * 4 + tmi) % ABLOCKS;
    return 1010 + tni + 8 * (t % 3);
  };
  auto addr = [&](int q) -> const char* {
    const int p = wave * PP + q;     // piece 0..31 of the step
    long off;
    if (ABLOCKS > 0) {
      // faithful K = 768 geometry: a panel is [256 rows][1536 B]; ROWB 64: every step takes bytes [64 pos, 64 pos + 64) of the rows
      // of the A panel (pieces 0..15) and of the B panel (16..31), 24 steps per tile; ROWB 128: even steps take 128 B per row of
      // the A panel, odd steps of the B panel (the same bytes per step and per tile, whole cache lines per row)
      if (ROWB == 64) return base + panel(p, tile) * 256 * ld + (long)((p & 15) * 16 + (lane >> 2)) * ld + (long)pos * 64 + (lane & 3) * 16;
      return base + panel((pos & 1) * 16, tile) * 256 * ld + (long)(p * 8 + (lane >> 3)) * ld + (long)(pos >> 1) * 128 + (lane & 7) * 16;
    }
    const char* rowbase = base + (long)blk * 256 * ld;
    if (ROWB == 64) off = (long)((p & 15) * 16 + (lane >> 2)) * ld + (long)pos * 128 + (p >> 4) * 64 + (lane & 3) * 16;
    else off = (long)(p * 8 + (lane >> 3)) * ld + (long)pos * 128 + (lane & 7) * 16;
    return rowbase + off;
  };
  unsigned pfv = 0;
  auto frag = [&](int s, int r0, int ks) -> bf16x8 {   // the kernel's row-major fragment read (64-byte rows, XOR-swizzled 16-B pieces)
    const int r = r0 + (lane & 31), kp = ks * 2 + (lane >> 5);
    return *LDS_PTR(const bf16x8, smem + s * 32768 + r * 64 + ((kp ^ ((r >> 2) & 3)) << 4));
  };
  for (int s = 0; s < steps; s++) {
    const int rs = (slot + 2) & 3;     // a slot whose DMA has certainly landed
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      const int c = ks & 1, nx = c ^ 1;
      if constexpr (READS) {
#pragma unroll
        for (int jj = 0; jj < NJ; jj++) b[nx][jj] = frag(rs, 256 * 0 + wn * (NJ * 32) + jj * 32, ks ^ 1);
      }
#pragma unroll
      for (int i = 0; i < MI; i++) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 0; jj < NJ; jj++) {
#ifdef SKEL_MFMA16   // the same FLOPs as two 16x16x32 instructions per 32x32x16 slot (power / rate experiment only: the fragment layouts are not those of a real 16x16 kernel)
          typedef __attribute__((ext_vector_type(4))) float f32x4_;
          f32x4_ lo = {acc[i][jj][0 + 8 * ks], acc[i][jj][1 + 8 * ks], acc[i][jj][2 + 8 * ks], acc[i][jj][3 + 8 * ks]};
          f32x4_ hi = {acc[i][jj][4 + 8 * ks], acc[i][jj][5 + 8 * ks], acc[i][jj][6 + 8 * ks], acc[i][jj][7 + 8 * ks]};
          lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[c][jj], a[i], lo, 0, 0, 0);
          hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[c][jj], hi, 0, 0, 0);
#pragma unroll
          for (int e = 0; e < 4; e++) { acc[i][jj][e + 8 * ks] = lo[e]; acc[i][jj][4 + e + 8 * ks] = hi[e]; }
#else
          acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[c][jj], a[i], acc[i][jj], 0, 0, 0);
#endif
        }
        if constexpr (READS) a[i] = frag(rs, (wm * (MI * 32) + i * 32) & 255, ks ^ 1);
        const int q = ks * MI + i;
        constexpr int DS = (2 * MI) / PP > 0 ? (2 * MI) / PP : 1;
        if (q % DS == 0 && q / DS < PP) {
          __builtin_amdgcn_sched_barrier(0);
          glds16(addr(q / DS), lds0 + slot * 32768 + (wave * PP + q / DS) * 1024);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (PF > 0) {
      // L2 prefetch: waves 0..3 touch one dword of each 128-byte line of the A rows PF steps ahead (a step covers every line
      // of its 256 rows x 128 B = 256 lines: 4 waves x 64 lanes); the loaded value is never waited for in the loop
      if (wave < 4) {
        int pp = pos + PF, tt = tile;
        if (pp >= spb) { pp -= spb; tt++; }
        const char* pa = base + panel(0, tt) * 256 * ld + (long)(wave * 64 + lane) * ld + (long)pp * 64;
        // the destination is a register the compiler must keep for us until the end of the kernel (asm loads are invisible
        // to its scoreboard: a temporary would be reused before the data lands)
        asm volatile("global_load_dword %0, %1, off sc1" : "+v"(pfv) : "v"(pa) : "memory");
      }
      if (wave < 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PP * (DEPTH - 1) + DEPTH - 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PP * (DEPTH - 1)) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PP * (DEPTH - 1)) : "memory");
    }
    __builtin_amdgcn_s_barrier();
    slot = (slot + 1) & 3;
    if (++pos >= spb) { pos = 0; tile++; blk += share; if (blk >= nblocks) blk -= nblocks; }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float t = pfv == 0x1234567u ? 1.f : 0.f;
  for (int i = 0; i < MI; i++) for (int jj = 0; jj < NJ; jj++) t += acc[i][jj][0];
  if (t == 12345.f) sink[0] = t;
}

template <int WM, int WN, int MI, int NJ, int ROWB, bool READS, int DEPTH, int ABLOCKS = 0, int PF = 0>
void run_gemm_like(const char* name, const char* d, long ld, int share, float* sink) {
  const int steps = 4000, nblocks = 102;
  hipFuncSetAttribute((const void*)gemm_like<WM, WN, MI, NJ, ROWB, READS, DEPTH, ABLOCKS, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  float best = 1e30f;
  for (int rep = 0; rep < 4; rep++) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((gemm_like<WM, WN, MI, NJ, ROWB, READS, DEPTH, ABLOCKS, PF>), dim3(256), dim3(64 * WM * WN), 131072, 0, d, ld, (int)ld, nblocks, share, steps, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double fl = 256.0 * steps * 2.0 * 256 * 256 * 32;
  printf("%-40s waves %dx%d tile/wave %3dx%3d rows %3dB reads=%d depth=%d apanels=%4d pf=%d: %7.3f ms  %7.1f TFLOP/s  %5.1f GB/s/CU\n", name, WM, WN, MI * 32, NJ * 32, ROWB, (int)READS, DEPTH, ABLOCKS, PF,
         best, fl / best / 1e9, 256.0 * steps * 32768.0 / best / 1e6 / 256);
}

template <int PATH, int ROWB, int DEPTH, bool BARRIER = false, int NMFMA = 0>
void run(const char* name, const char* d, long ld, int kbytes, int rows, int share, unsigned* sink) {
  const int steps = 4000, nblocks = rows / 256;
  hipFuncSetAttribute((const void*)stream<PATH, ROWB, DEPTH, BARRIER, NMFMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  float best = 1e30f;
  for (int rep = 0; rep < 4; rep++) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream<PATH, ROWB, DEPTH, BARRIER, NMFMA>), dim3(256), dim3(512), 131072, 0, d, ld, kbytes, nblocks, share, steps, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double bytes = 256.0 * steps * 32768.0;
  printf("%-34s ld=%6ld B share=%2d depth=%d barrier=%d mfma/step=%2d: %7.3f ms  %6.2f TB/s  %5.1f GB/s/CU", name, ld, share, DEPTH, (int)BARRIER, NMFMA, best, bytes / best / 1e9, bytes / best / 1e6 / 256);
  if (NMFMA) printf("  %6.1f TFLOP/s of MFMA", 256.0 * 8 * steps * NMFMA * 32768.0 / best / 1e9);
  printf("\n");
}

int main(int argc, char** argv) {
  const bool only_gemm = argc > 1 && argv[1][0] == 'g';
  setvbuf(stdout, nullptr, _IONBF, 0);
  const long bytes = 26240L * 16384 + (1 << 20);
  char* d; hipMalloc(&d, bytes); hipMemset(d, 1, bytes);
  if (getenv("GLDS_RANDOM")) {   // pseudo-random bf16 operands (N(0,1)-ish) instead of the constant fill: what the matrix pipes' power draw depends on
    const size_t n = bytes / 2;
    unsigned short* h = (unsigned short*)malloc(n * 2);
    unsigned x = 12345u;
    for (size_t i = 0; i < n; i++) {
      float s = 0.f;
      for (int t = 0; t < 4; t++) { x = x * 1664525u + 1013904223u; s += (float)(x >> 8) * (1.f / 16777216.f) - 0.5f; }
      s *= 1.7f;
      unsigned u; memcpy(&u, &s, 4);
      h[i] = (unsigned short)(u >> 16);
    }
    hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice);
    free(h);
    printf("operands: pseudo-random bf16\n");
  }
  unsigned* sink; hipMalloc(&sink, 64);
  for (long ld : {1536L, 16384L}) {
    if (only_gemm) break;
    const int kb = (int)ld;
    for (int share : {1, 4, 32}) {
      run<0, 64, 3>("glds  16 rows x 64 B / piece", d, ld, kb, 26240 / 256 * 256, share, sink);
      run<0, 128, 3>("glds   8 rows x 128 B / piece", d, ld, kb, 26240 / 256 * 256, share, sink);
      run<0, 256, 3>("glds   4 rows x 256 B / piece", d, ld, kb, 26240 / 256 * 256, share, sink);
      run<0, 0, 3>("glds  contiguous 1 KiB / piece", d, ld, kb, 26240 / 256 * 256, share, sink);
      run<1, 64, 3>("vgpr  16 rows x 64 B", d, ld, kb, 26240 / 256 * 256, share, sink);
      run<1, 128, 3>("vgpr   8 rows x 128 B", d, ld, kb, 26240 / 256 * 256, share, sink);
      run<2, 128, 3>("vgpr+ds_write 8 rows x 128 B", d, ld, kb, 26240 / 256 * 256, share, sink);
    }
  }
  // the GEMM kernel's structure: wait + workgroup barrier every step (32 KiB = one 32-wide K half of a 256x256 tile), with and
  // without a compute segment of 16 MFMAs per wave and step (= what a 256x256 tile needs per half)
  const int R = 26240 / 256 * 256;
  if (!only_gemm) {
  run<0, 64, 2, true>("glds 64B rows + barrier", d, 1536, 1536, R, 4, sink);
  run<0, 64, 3, true>("glds 64B rows + barrier", d, 1536, 1536, R, 4, sink);
  run<0, 64, 4, true>("glds 64B rows + barrier", d, 1536, 1536, R, 4, sink);
  run<0, 128, 2, true>("glds 128B rows + barrier", d, 1536, 1536, R, 4, sink);
  run<0, 128, 3, true>("glds 128B rows + barrier", d, 1536, 1536, R, 4, sink);
  run<0, 128, 4, true>("glds 128B rows + barrier", d, 1536, 1536, R, 4, sink);
  run<0, 64, 2, true, 16>("glds 64B rows + barrier + mfma", d, 1536, 1536, R, 4, sink);
  run<0, 64, 3, true, 16>("glds 64B rows + barrier + mfma", d, 1536, 1536, R, 4, sink);
  run<0, 64, 4, true, 16>("glds 64B rows + barrier + mfma", d, 1536, 1536, R, 4, sink);
  run<0, 128, 2, true, 16>("glds 128B rows + barrier + mfma", d, 1536, 1536, R, 4, sink);
  run<0, 128, 3, true, 16>("glds 128B rows + barrier + mfma", d, 1536, 1536, R, 4, sink);
  run<0, 128, 4, true, 16>("glds 128B rows + barrier + mfma", d, 1536, 1536, R, 4, sink);
  run<0, 64, 3, false, 16>("glds 64B rows, no barrier + mfma", d, 1536, 1536, R, 4, sink);
  run<0, 128, 3, false, 16>("glds 128B rows, no barrier + mfma", d, 1536, 1536, R, 4, sink);
  run<0, 64, 3, true, 16>("glds 64B rows + barrier + mfma", d, 16384, 16384, R, 4, sink);
  run<0, 128, 3, true, 16>("glds 128B rows + barrier + mfma", d, 16384, 16384, R, 4, sink);
  run<0, 0, 3, true, 16>("glds contiguous + barrier + mfma", d, 1536, 1536, R, 4, sink);
  }
  printf("---- GEMM-shaped skeleton (256x256 tile, one 32-wide K half per step) ----\n");
  float* fs = (float*)sink;
  const bool only_pf = argc > 1 && argv[1][1] == 'p';
  if (!only_pf) {
  run_gemm_like<2, 4, 4, 2, 64, false, 2>("8 waves, no LDS reads", d, 1536, 4, fs);
  run_gemm_like<2, 4, 4, 2, 64, true, 2>("8 waves (the kernel today)", d, 1536, 4, fs);
  run_gemm_like<2, 4, 4, 2, 64, true, 3>("8 waves (the kernel today)", d, 1536, 4, fs);
  run_gemm_like<2, 4, 4, 2, 128, true, 2>("8 waves, 128-byte rows", d, 1536, 4, fs);
  run_gemm_like<2, 4, 4, 2, 128, true, 3>("8 waves, 128-byte rows", d, 1536, 4, fs);
  run_gemm_like<2, 2, 4, 4, 64, false, 2>("4 waves 128x128, no LDS reads", d, 1536, 4, fs);
  run_gemm_like<2, 2, 4, 4, 64, true, 2>("4 waves 128x128", d, 1536, 4, fs);
  run_gemm_like<2, 2, 4, 4, 64, true, 3>("4 waves 128x128", d, 1536, 4, fs);
  run_gemm_like<2, 2, 4, 4, 128, true, 2>("4 waves 128x128, 128-byte rows", d, 1536, 4, fs);
  run_gemm_like<2, 2, 4, 4, 128, true, 3>("4 waves 128x128, 128-byte rows", d, 1536, 4, fs);
  run_gemm_like<2, 4, 4, 2, 64, true, 2, 102>("8 waves, A panels stream (40 MB)", d, 1536, 4, fs);
  run_gemm_like<2, 4, 4, 2, 64, true, 3, 102>("8 waves, A panels stream (40 MB)", d, 1536, 4, fs);
  run_gemm_like<2, 4, 4, 2, 64, true, 2, 1000>("8 waves, A panels stream (390 MB)", d, 1536, 4, fs);
  run_gemm_like<2, 4, 4, 2, 64, true, 3, 1000>("8 waves, A panels stream (390 MB)", d, 1536, 4, fs);
  }
  run_gemm_like<2, 4, 4, 2, 64, true, 2, 102, 6>("8 waves, 40 MB, L2 prefetch 6 ahead", d, 1536, 4, fs);
  run_gemm_like<2, 4, 4, 2, 64, true, 2, 1000, 6>("8 waves, 390 MB, L2 prefetch 6 ahead", d, 1536, 4, fs);
  run_gemm_like<2, 4, 4, 2, 64, true, 2, 1000, 12>("8 waves, 390 MB, L2 prefetch 12 ahead", d, 1536, 4, fs);
  run_gemm_like<2, 4, 4, 2, 128, true, 2, 1000>("8 waves, 128B rows, 390 MB", d, 1536, 4, fs);
  run_gemm_like<2, 2, 4, 4, 128, true, 3, 1000>("4 waves, 128B rows, 390 MB", d, 1536, 4, fs);
  run_gemm_like<2, 4, 4, 2, 64, true, 2>("8 waves (the kernel today) K=8192", d, 16384, 4, fs);
  run_gemm_like<2, 2, 4, 4, 128, true, 3>("4 waves 128x128, 128-byte rows K=8192", d, 16384, 4, fs);
  if (only_gemm) return 0;
  run<0, 64, 2>("glds  16 rows x 64 B / piece", d, 1536, 1536, 26240 / 256 * 256, 1, sink);
  run<0, 64, 4>("glds  16 rows x 64 B / piece", d, 1536, 1536, 26240 / 256 * 256, 1, sink);
  run<0, 128, 2>("glds   8 rows x 128 B / piece", d, 1536, 1536, 26240 / 256 * 256, 1, sink);
  run<0, 128, 4>("glds   8 rows x 128 B / piece", d, 1536, 1536, 26240 / 256 * 256, 1, sink);
  return 0;
}
