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
  return 0;
}
