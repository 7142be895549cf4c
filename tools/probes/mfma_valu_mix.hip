// (round 4) Which VALU instructions co-issue with the matrix pipe of the same SIMD?  256 workgroups x 8 waves (2 per SIMD), register-only work:
// waves 0-3 issue 8 independent v_mfma_f32_32x32x16_bf16 per iteration, waves 4-7 issue 32 VALU instructions of ONE kind per iteration.
// Printed per kind: the VALU waves alone, the MFMA waves alone (the other half idle), both together; "together" close to max(alone, alone)
// means the two pipes overlap, close to the sum means they serialise.
// hipcc --offload-arch=gfx950 -O3 -o mfma_valu_mix mfma_valu_mix.hip && ./mfma_valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// KIND: 0 v_fma_f32, 1 v_exp_f32, 2 v_pk_fma_f32, 3 v_cvt_pk_bf16_f32, 4 v_max3_f32, 5 v_pk_mul_f32, 6 v_sub + v_mul (unpacked pair), 7 v_add_f32, 8 v_mul_f32, 9 v_and / v_or, 10 v_mov
// WHO: 1 = only the MFMA waves work, 2 = only the VALU waves work, 3 = both
template <int KIND, int WHO>
__global__ __launch_bounds__(512) void k(float* out, int iters, float c) {
  f32x16 acc[8];
  for (int i = 0; i < 8; i++)
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; e++) { a[e] = (__bf16)(float)((threadIdx.x + e) & 7); b[e] = (__bf16)(float)((blockIdx.x + e) & 3); }
  float v[32];
  for (int i = 0; i < 32; i++) v[i] = (float)(threadIdx.x + i) * 1e-3f;
  const int wave = threadIdx.x >> 6;
  for (int it = 0; it < iters; it++) {
    if (wave < 4) {
      if (WHO & 1) {
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
      }
    } else if (WHO & 2) {
#pragma unroll
      for (int i = 0; i < 32; i += 2) {
        if (KIND == 0) { asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v[i]) : "v"(v[i]), "v"(c), "v"(v[(i + 3) & 31])); asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v[i + 1]) : "v"(v[i + 1]), "v"(c), "v"(v[(i + 5) & 31])); }
        else if (KIND == 7) { asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(c)); asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[i + 1]) : "v"(v[i + 1]), "v"(c)); }
        else if (KIND == 8) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(c)); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i + 1]) : "v"(v[i + 1]), "v"(c)); }
        else if (KIND == 9) { asm volatile("v_and_b32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(c)); asm volatile("v_or_b32 %0, %1, %2" : "=v"(v[i + 1]) : "v"(v[i + 1]), "v"(c)); }
        else if (KIND == 10) { asm volatile("v_mov_b32 %0, %1" : "=v"(v[i]) : "v"(v[i + 1])); asm volatile("v_mov_b32 %0, %1" : "=v"(v[i + 1]) : "v"(v[(i + 2) & 31])); }
        else if (KIND == 1) { v[i] = __builtin_amdgcn_exp2f(v[i]); v[i + 1] = __builtin_amdgcn_exp2f(v[i + 1]); }
        else if (KIND == 2) { f32x2 t = {v[i], v[i + 1]}; t = __builtin_elementwise_fma(t, (f32x2){c, c}, (f32x2){1.f, 1.f}); v[i] = t[0]; v[i + 1] = t[1];
                              f32x2 u = {v[(i + 16) & 31], v[(i + 17) & 31]}; u = __builtin_elementwise_fma(u, (f32x2){c, c}, (f32x2){1.f, 1.f}); v[(i + 16) & 31] = u[0]; v[(i + 17) & 31] = u[1]; }
        else if (KIND == 3) { unsigned r0, r1; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r0) : "v"(v[i]), "v"(v[i + 1])); asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r1) : "v"(v[i + 1]), "v"(v[i]));
                              v[i] = __builtin_bit_cast(float, r0); v[i + 1] = __builtin_bit_cast(float, r1); }
        else if (KIND == 4) { v[i] = __builtin_fmaxf(__builtin_fmaxf(v[i], v[i + 1]), c); v[i + 1] = __builtin_fmaxf(__builtin_fmaxf(v[i + 1], v[(i + 2) & 31]), c); }
        else if (KIND == 5) { f32x2 t = {v[i], v[i + 1]}; t = t * (f32x2){c, c}; v[i] = t[0]; v[i + 1] = t[1];
                              f32x2 u = {v[(i + 16) & 31], v[(i + 17) & 31]}; u = u * (f32x2){c, c}; v[(i + 16) & 31] = u[0]; v[(i + 17) & 31] = u[1]; }
        else { v[i] = (v[i] - c) * v[i + 1]; v[i + 1] = (v[i + 1] - c) * c; }
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; i++) s += acc[i][0];
  for (int i = 0; i < 32; i++) s += v[i];
  if (s == 12345.f) out[0] = s;
}

template <int KIND, int WHO>
float run(float* d) {
  const int iters = 50000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, WHO>), dim3(256), dim3(512), 0, 0, d, iters, 0.999f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  return best * 1e6f / iters;
}

template <int KIND>
void kind(const char* what, float* d) {
  const float m = run<KIND, 1>(d), v = run<KIND, 2>(d), both = run<KIND, 3>(d);
  printf("%-46s MFMA waves alone %6.1f ns   VALU waves alone %6.1f ns   together %6.1f ns   (max %6.1f, sum %6.1f)\n", what, m, v, both, m > v ? m : v, m + v);
}

int main() {
  float* d; hipMalloc(&d, 4);
  printf("per iteration: waves 0-3 issue 8 x v_mfma_f32_32x32x16_bf16, waves 4-7 issue 32 VALU instructions (16 packed ones for the pk kinds x 2 sets)\n");
  kind<0>("v_fma_f32", d);
  kind<1>("v_exp_f32", d);
  kind<2>("v_pk_fma_f32 (32 per iteration)", d);
  kind<3>("v_cvt_pk_bf16_f32", d);
  kind<4>("v_max3_f32 / v_max", d);
  kind<5>("v_pk_mul_f32 (32 per iteration)", d);
  kind<6>("v_sub_f32 + v_mul_f32 (dependent pairs)", d);
  kind<7>("v_add_f32", d);
  kind<8>("v_mul_f32", d);
  kind<9>("v_and_b32 / v_or_b32", d);
  kind<10>("v_mov_b32", d);
  return 0;
}
