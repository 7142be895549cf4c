// MFMA issue-rate probe: pure register-resident v_mfma_f32_32x32x16_bf16 loops (no memory traffic).
// hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; i++)
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; e++) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)(float)(blockIdx.x + e); }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; i++) s += acc[i][0];
  if (s == 12345.f) out[0] = s;
}

// the same loop on operands that CHANGE between MFMAs: NOP pairs of pseudo-random bf16 fragments (normal-ish values from a hash of
// lane / block / index) rotate through the instruction stream -- data toggling is what a real GEMM's matrix pipes see, and the chip's
// power management prices it (a constant-operand loop holds the top clock, see main)
template <int NACC, int NOP>
__global__ __launch_bounds__(512) void krand(float* out, int iters, float scale) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; i++)
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  bf16x8 a[NOP], b[NOP];
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int p = 0; p < NOP; p++)
    for (int e = 0; e < 8; e++) {
      float s = 0.f;
      for (int t = 0; t < 4; t++) { h = h * 1664525u + 1013904223u; s += (float)(h >> 8) * (1.f / 16777216.f) - 0.5f; }   // ~N(0, 1/3)
      a[p][e] = (__bf16)(s * scale);
      for (int t = 0; t < 4; t++) { h = h * 1664525u + 1013904223u; s += (float)(h >> 8) * (1.f / 16777216.f) - 0.5f; }
      b[p][e] = (__bf16)(s * scale);
    }
  for (int it = 0; it < iters; it += NOP) {
#pragma unroll
    for (int r = 0; r < NOP; r++)       // (compile-time rotation: the fragment arrays stay in registers)
#pragma unroll
      for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i % NOP], b[(i + r) % NOP], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; i++) s += acc[i][0];
  if (s == 12345.f) out[0] = s;
}

// operand ORDER on random data: MODE 0 = every MFMA a new (a, b) pair; 1 = the A fragment held for 2 consecutive MFMAs (the kernels'
// j-inner order at NJ = 2); 2 = A held for 4; 3 = A held for 8 (b rotates); 4 = 16x16x32 instructions, every one a new pair
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int MODE>
__global__ __launch_bounds__(512) void korder(float* out, int iters) {
  constexpr int NOP = 8;
  f32x16 acc[8];
  f32x4 acc4[16];
  for (int i = 0; i < 8; i++)
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  for (int i = 0; i < 16; i++)
    for (int r = 0; r < 4; r++) acc4[i][r] = 0.f;
  bf16x8 a[NOP], b[NOP];
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int p = 0; p < NOP; p++)
    for (int e = 0; e < 8; e++) {
      float s = 0.f;
      for (int t = 0; t < 4; t++) { h = h * 1664525u + 1013904223u; s += (float)(h >> 8) * (1.f / 16777216.f) - 0.5f; }
      a[p][e] = (__bf16)s;
      for (int t = 0; t < 4; t++) { h = h * 1664525u + 1013904223u; s += (float)(h >> 8) * (1.f / 16777216.f) - 0.5f; }
      b[p][e] = (__bf16)s;
    }
  for (int it = 0; it < iters; it += NOP) {
#pragma unroll
    for (int r = 0; r < NOP; r++)
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if constexpr (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i % NOP], b[(i + r) % NOP], acc[i], 0, 0, 0);
        if constexpr (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i / 2 + r) % NOP], b[(i + r) % NOP], acc[i], 0, 0, 0);
        if constexpr (MODE == 2) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i / 4 + r) % NOP], b[(i + r) % NOP], acc[i], 0, 0, 0);
        if constexpr (MODE == 3) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[r], b[i], acc[i], 0, 0, 0);
        if constexpr (MODE == 4) {
          acc4[2 * i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i % NOP], b[(i + r) % NOP], acc4[2 * i], 0, 0, 0);
          acc4[2 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + 3) % NOP], b[(i + r + 5) % NOP], acc4[2 * i + 1], 0, 0, 0);
        }
      }
  }
  float s = 0.f;
  for (int i = 0; i < 8; i++) s += acc[i][0];
  for (int i = 0; i < 16; i++) s += acc4[i][0];
  if (s == 12345.f) out[0] = s;
}
template <int MODE>
void run_order(const char* what, int iters, float* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(korder<MODE>, dim3(256), dim3(512), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  double fl = 256.0 * 8 * (double)iters * 8 * 32768.0;
  printf("random operands, %-58s: %.3f ms  %.1f TFLOP/s\n", what, best, fl / best / 1e9);
}

template <int NACC, int NOP>
void run_rand(int waves_per_cu, int iters, float scale, float* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int threads = 64 * waves_per_cu;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((krand<NACC, NOP>), dim3(256), dim3(threads), 0, 0, d, iters, scale);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = 256.0 * waves_per_cu * (double)iters * NACC * 32768.0;
    printf("RANDOM operands (%d rotating pairs, scale %g) nacc=%d waves/CU=%d iters=%d: %.3f ms  %.1f TFLOP/s\n", NOP, scale, NACC, waves_per_cu, iters, ms, fl / ms / 1e9);
  }
}

template <int NACC>
void run(int waves_per_cu, int iters, float* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int threads = 64 * waves_per_cu;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = 256.0 * waves_per_cu * (double)iters * NACC * 32768.0;
    printf("nacc=%d waves/CU=%d iters=%d: %.3f ms  %.1f TFLOP/s\n", NACC, waves_per_cu, iters, ms, fl / ms / 1e9);
  }
}

int main() {
  float* d; hipMalloc(&d, 4);
  run<8>(4, 20000, d);
  run<8>(8, 20000, d);
  run<8>(8, 200000, d);
  run<4>(8, 40000, d);
  run<2>(8, 80000, d);
  run_rand<8, 4>(8, 200000, 1.f, d);
  run_rand<8, 8>(8, 200000, 1.f, d);
  run_rand<8, 8>(8, 200000, 1e-3f, d);
  run_rand<8, 8>(8, 200000, 0.f, d);      // same instruction stream, all-zero operands
  run_rand<8, 8>(4, 200000, 1.f, d);
  run<8>(8, 200000, d);
  run_order<0>("32x32x16, every MFMA a new (a, b) pair", 200000, d);
  run_order<1>("32x32x16, A fragment held for 2 consecutive MFMAs", 200000, d);
  run_order<2>("32x32x16, A fragment held for 4 consecutive MFMAs", 200000, d);
  run_order<3>("32x32x16, A fragment held for 8 consecutive MFMAs", 200000, d);
  run_order<4>("16x16x32 (two per 32x32x16 slot), every one a new pair", 200000, d);
  return 0;
}
