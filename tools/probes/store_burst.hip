// Epilogue-store probe: is the GEMM epilogue's store rate (~16 B/clk/CU) a per-CU limit or the aggregate limit of the
// L2 -> fabric write path when all 256 persistent workgroups reach their epilogue together?
// N workgroups (8 waves, one per CU) each write BURSTS of one output tile (320 rows x 512 B, row stride LD bytes) with
// global_store_dwordx4, wait for the stores (s_waitcnt vmcnt(0)), then idle for ~IDLE us (the next tile's main loop); the shader-clock
// duration of every burst is averaged.  Compare N = 256 (every CU bursts at once) with N = 32 / 64 / 128, and with bursts that are
// de-phased across workgroups (phase = 1: workgroup i starts i/N of a period later).
// hipcc --offload-arch=gfx950 -O3 -o store_burst store_burst.hip && ./store_burst
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <bool NT>
__global__ __launch_bounds__(512) void burst(char* out, long ld, int tiles_per_wg, int reps, int idle_sleeps, int phase, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (phase) for (int i = 0; i < (int)((long)idle_sleeps * blockIdx.x / gridDim.x); i++) __builtin_amdgcn_s_sleep(127);
  unsigned long long tot = 0;
  for (int r = 0; r < reps; r++) {
    // tile r % tiles_per_wg of this workgroup: 320 rows x 512 B; wave w owns rows w*40 .. w*40+39, two rows per store instruction
    char* base = out + ((long)blockIdx.x * tiles_per_wg + r % tiles_per_wg) * 320 * ld;
    __syncthreads();
    const unsigned long long t0 = clock64();
    u32x4 v = {(unsigned)r, (unsigned)lane, 3u, 4u};
#pragma unroll 4
    for (int i = 0; i < 20; i++) {
      char* p = base + (long)(wave * 40 + i * 2 + (lane >> 5)) * ld + (lane & 31) * 16;
      if constexpr (NT) __builtin_nontemporal_store(v, (u32x4*)p); else *(u32x4*)p = v;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    tot += clock64() - t0;
    for (int i = 0; i < idle_sleeps; i++) __builtin_amdgcn_s_sleep(127);
  }
  if (threadIdx.x == 0) cyc[blockIdx.x] = tot;
}

int main() {
  const long ld = 4608;                       // 2304 bf16 columns
  const int tiles = 3, reps = 30;
  char* out; unsigned long long* cyc;
  hipMalloc(&out, (size_t)256 * tiles * 320 * ld + (1 << 20));
  hipMalloc(&cyc, 256 * 8);
  printf("burst = 160 KiB per workgroup; cycles are shader-clock ticks of clock64()\n");
  for (int nt = 0; nt < 2; nt++)
  for (int phase = 0; phase < 2; phase++)
    for (int n : {8, 32, 64, 128, 256}) {
      const int idle = 6;   // 6 x s_sleep(127) ~ 6 x 8128 cycles
      hipMemset(cyc, 0, 256 * 8);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (nt) hipLaunchKernelGGL(burst<true>, dim3(n), dim3(512), 0, 0, out, ld, tiles, reps, idle, phase, cyc);
      else hipLaunchKernelGGL(burst<false>, dim3(n), dim3(512), 0, 0, out, ld, tiles, reps, idle, phase, cyc);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> h(n);
      hipMemcpy(h.data(), cyc, n * 8, hipMemcpyDeviceToHost);
      double s = 0; for (auto c : h) s += (double)c;
      const double per = s / n / reps;
      printf("nt %d  workgroups %3d  phase %d : %8.0f ticks per burst  = %6.1f B/tick/CU   (kernel %.3f ms)\n", nt, n, phase, per, 163840.0 / per, ms);
    }
  return 0;
}
