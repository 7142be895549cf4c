// Store-instruction SHAPE probe (round 5): what does a global_store_dwordx4 wave instruction cost as a function of how its 1 KiB is laid out over
// output rows?  256 workgroups (8 waves, one per CU) each write bursts of one 256 x 256 bf16 output tile (256 rows x 512 B, row stride LD bytes):
//   R = rows per store instruction: 1 KiB / R contiguous bytes per row.  R = 8 is the GEMM epilogue of rounds 2-4 (8 rows x 128 B: a wave owns a
//   128 B column block), R = 16 the direct (permlane16_swap) form of round 5 (16 rows x 64 B), R = 2 / 4 what a cross-wave staging could issue
//   (2 rows x 512 B = whole tile rows, 4 rows x 256 B).
// hipcc --offload-arch=gfx950 -O3 -o store_shape store_shape.hip && ./store_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int R, bool NT>
__global__ __launch_bounds__(512) void burst(char* out, long ld, int tiles_per_wg, int reps, int idle_sleeps, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int BPR = 1024 / R;            // bytes per row and instruction
  constexpr int CB = 512 / BPR;            // column blocks of the tile
  constexpr int LPR = BPR / 16;            // lanes per row
  unsigned long long tot = 0;
  for (int r = 0; r < reps; r++) {
    char* base = out + ((long)blockIdx.x * tiles_per_wg + r % tiles_per_wg) * 256 * ld;
    __syncthreads();
    const unsigned long long t0 = clock64();
    u32x4 v = {(unsigned)r, (unsigned)lane, 3u, 4u};
    // 128 instructions per workgroup = 16 per wave; instruction k of wave w: column block (w % CB), rows ((w / CB) * 16 + k) * R ... + R
    const int cb = wave % CB, rg = wave / CB;
#pragma unroll 4
    for (int k = 0; k < 16; k++) {
      const int row = ((rg * 16 + k) * R) % 256 + lane / LPR;      // (CB < 8: several waves share a column block and split the rows)
      const int rowc = (rg * 16 + k) * R * CB / 8 * 0 + row;       // keep simple: rows wrap inside the tile
      char* p = base + (long)rowc * ld + cb * BPR + (lane % LPR) * 16;
      if constexpr (NT) __builtin_nontemporal_store(v, (u32x4*)p); else *(u32x4*)p = v;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    tot += clock64() - t0;
    for (int i = 0; i < idle_sleeps; i++) __builtin_amdgcn_s_sleep(127);
  }
  if (threadIdx.x == 0) cyc[blockIdx.x] = tot;
}

template <int R, bool NT>
void run(char* out, unsigned long long* cyc, long ld, int n) {
  const int tiles = 3, reps = 30, idle = 6;
  hipMemset(cyc, 0, 256 * 8);
  hipLaunchKernelGGL((burst<R, NT>), dim3(n), dim3(512), 0, 0, out, ld, tiles, reps, idle, cyc);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(n);
  hipMemcpy(h.data(), cyc, n * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto c : h) s += (double)c;
  const double per = s / n / reps;
  printf("ld %6ld  rows/instr %2d (%4d B per row)  nt %d  workgroups %3d : %8.0f ticks per 128 KiB tile = %6.1f B/tick/CU\n", ld, R, 1024 / R, (int)NT, n, per, 131072.0 / per);
}

int main() {
  char* out; unsigned long long* cyc;
  hipMalloc(&out, (size_t)256 * 3 * 256 * 12288 + (1 << 20));
  hipMalloc(&cyc, 256 * 8);
  for (long ld : {12288L, 4608L, 1536L})
    for (int n : {256, 16}) {
      run<1, false>(out, cyc, ld, n); run<2, false>(out, cyc, ld, n); run<4, false>(out, cyc, ld, n); run<8, false>(out, cyc, ld, n); run<16, false>(out, cyc, ld, n);
      run<2, true>(out, cyc, ld, n); run<8, true>(out, cyc, ld, n);
    }
  return 0;
}
