// Can the VALU and the matrix pipe of a SIMD work at the same time?  256 workgroups x 8 waves (2 per SIMD), register-only work.
//   mode 0: every wave MFMAs only (8 per iteration)            mode 1: every wave VALU only (16 x {v_fma, v_exp} per iteration)
//   mode 2: waves 0-3 MFMAs only, waves 4-7 VALU only (the two waves of a SIMD run different pipes)
//   mode 3: every wave both, phase by phase (8 MFMAs, then the VALU block; order pinned with sched_barrier)
//   mode 4: every wave both, interleaved in source order (1 MFMA, 2 x {fma, exp}, ...; pinned)
//   mode 5: as 3, but waves 4-7 run the VALU block FIRST (the two waves of a SIMD in opposite phases)
// hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, float c) {
  f32x16 acc[8];
  for (int i = 0; i < 8; i++)
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; e++) { a[e] = (__bf16)(float)((threadIdx.x + e) & 7); b[e] = (__bf16)(float)((blockIdx.x + e) & 3); }
  float v[16];
  for (int i = 0; i < 16; i++) v[i] = (float)(threadIdx.x + i) * 1e-3f;
  const int wave = threadIdx.x >> 6;
  auto mf = [&](int i) { acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0); };
  auto va = [&](int i) { v[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(v[i], c, -1.f)); };
  for (int it = 0; it < iters; it++) {
    if constexpr (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 8; i++) mf(i);
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 16; i++) va(i);
    } else if constexpr (MODE == 2) {
      if (wave < 4) {
#pragma unroll
        for (int i = 0; i < 8; i++) mf(i);
      } else {
#pragma unroll
        for (int i = 0; i < 16; i++) va(i);
      }
    } else if constexpr (MODE == 3 || MODE == 5) {
      const bool valu_first = MODE == 5 && wave >= 4;
      if (valu_first) {
#pragma unroll
        for (int i = 0; i < 16; i++) va(i);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < 8; i++) mf(i);
      __builtin_amdgcn_sched_barrier(0);
      if (!valu_first) {
#pragma unroll
        for (int i = 0; i < 16; i++) va(i);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        mf(i);
        __builtin_amdgcn_sched_barrier(0);
        va(2 * i);
        va(2 * i + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; i++) s += acc[i][0];
  for (int i = 0; i < 16; i++) s += v[i];
  if (s == 12345.f) out[0] = s;
}

template <int MODE>
void run(const char* what, float* d) {
  const int iters = 100000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  printf("%-86s %8.3f ms  = %6.1f ns per iteration\n", what, best, best * 1e6 / iters);
}

int main() {
  float* d; hipMalloc(&d, 4);
  run<0>("0: every wave 8 MFMAs per iteration", d);
  run<1>("1: every wave 16 x {v_fma, v_exp} per iteration", d);
  run<2>("2: waves 0-3 the MFMAs, waves 4-7 the VALU block (one of each per SIMD)", d);
  run<3>("3: every wave both, phase by phase (MFMAs, then VALU)", d);
  run<4>("4: every wave both, interleaved in the instruction stream (1 MFMA : 2 x {fma, exp})", d);
  run<5>("5: every wave both, phase by phase, waves 4-7 in the opposite phase order", d);
  return 0;
}
