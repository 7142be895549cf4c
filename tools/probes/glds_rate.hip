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

int main() {
  const long bytes = 26240L * 16384 + (1 << 20);
  char* d; hipMalloc(&d, bytes); hipMemset(d, 1, bytes);
  unsigned* sink; hipMalloc(&sink, 64);
  for (long ld : {1536L, 16384L}) {
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
  run<0, 64, 2>("glds  16 rows x 64 B / piece", d, 1536, 1536, 26240 / 256 * 256, 1, sink);
  run<0, 64, 4>("glds  16 rows x 64 B / piece", d, 1536, 1536, 26240 / 256 * 256, 1, sink);
  run<0, 128, 2>("glds   8 rows x 128 B / piece", d, 1536, 1536, 26240 / 256 * 256, 1, sink);
  run<0, 128, 4>("glds   8 rows x 128 B / piece", d, 1536, 1536, 26240 / 256 * 256, 1, sink);
  return 0;
}
