// Scheduling experiment for the attention backward inner loop, register-only (no LDS, no barriers): per "tile" a wave runs, for two
// 32-row blocks q in {0, 1}:  S_q, dP_q = 8 MFMAs  ->  softmax-backward arithmetic on the 16 + 16 accumulator registers (fma, exp, sub, mul,
// bf16 pack)  ->  8 MFMAs that take the packed result as an operand (dV, dK).  256 workgroups x 8 waves (2 per SIMD), like the dK/dV kernel.
//   mode 0: program order  M_S0 V0 M_D0 M_S1 V1 M_D1   (what the kernel does today)
//   mode 1: software-pipelined  M_S0 | M_S1 interleaved with V0 | M_D0 interleaved with V1 | M_D1
//   mode 2: the MFMAs alone (both waves)        mode 3: the VALU arithmetic alone
// hipcc --offload-arch=gfx950 -O3 -o attn_sched attn_sched.hip && ./attn_sched
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ unsigned pk(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
  bf2 r = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, r);
}

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, float c, float lq, float delta) {
  f32x16 dk[2], dv[2];
  for (int i = 0; i < 2; i++)
    for (int r = 0; r < 16; r++) { dk[i][r] = 0.f; dv[i][r] = 0.f; }
  bf16x8 kf[4], vf[4], qa[4], da[4];
  for (int i = 0; i < 4; i++)
    for (int e = 0; e < 8; e++) {
      kf[i][e] = (__bf16)(float)((threadIdx.x + e + i) & 3);
      vf[i][e] = (__bf16)(float)((threadIdx.x + 2 * e + i) & 3);
      qa[i][e] = (__bf16)(float)((blockIdx.x + e + i) & 1);
      da[i][e] = (__bf16)(float)((blockIdx.x + e + 3 * i) & 1);
    }
  auto mfma = [](bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); };
  auto zero = [](f32x16& x) { for (int r = 0; r < 16; r++) x[r] = 0.f; };
  // softmax-backward arithmetic of 8 rows: p = exp2(s * c - lq); ds = p * (dp - delta); pack p and ds
  auto valu8 = [&](const f32x16& s, const f32x16& dp, int h8, bf16x8& pf, bf16x8& dsf) {
    u32x4 wp, wd;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int r = 8 * h8 + 2 * i;
      const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c, -lq)), p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r + 1], c, -lq));
      wp[i] = pk(p0, p1);
      wd[i] = pk(p0 * (dp[r] - delta), p1 * (dp[r + 1] - delta));
    }
    pf = __builtin_bit_cast(bf16x8, wp);
    dsf = __builtin_bit_cast(bf16x8, wd);
  };
  float sink = 0.f;
  for (int it = 0; it < iters; it++) {
    f32x16 s0, p0, s1, p1;
    bf16x8 pf[2][2], dsf[2][2];
    if constexpr (MODE == 0) {
#pragma unroll
      for (int q = 0; q < 2; q++) {
        f32x16 s, dp;
        zero(s); zero(dp);
#pragma unroll
        for (int ks = 0; ks < 4; ks++) { s = mfma(qa[ks], kf[ks], s); dp = mfma(da[ks], vf[ks], dp); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h8 = 0; h8 < 2; h8++) valu8(s, dp, h8, pf[q][h8], dsf[q][h8]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h8 = 0; h8 < 2; h8++)
#pragma unroll
          for (int db = 0; db < 2; db++) { dv[db] = mfma(da[h8 + db], pf[q][h8], dv[db]); dk[db] = mfma(qa[h8 + db], dsf[q][h8], dk[db]); }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (MODE == 1) {
      zero(s0); zero(p0); zero(s1); zero(p1);
#pragma unroll
      for (int ks = 0; ks < 4; ks++) { s0 = mfma(qa[ks], kf[ks], s0); p0 = mfma(da[ks], vf[ks], p0); }
      __builtin_amdgcn_sched_barrier(0);
      // M_S1 interleaved with V0: two MFMAs, then a quarter of the arithmetic
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        s1 = mfma(qa[ks], kf[(ks + 1) & 3], s1);
        p1 = mfma(da[ks], vf[(ks + 1) & 3], p1);
        __builtin_amdgcn_sched_barrier(0);
        if (ks & 1) valu8(s0, p0, ks >> 1, pf[0][ks >> 1], dsf[0][ks >> 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      // M_D0 interleaved with V1
#pragma unroll
      for (int h8 = 0; h8 < 2; h8++) {
#pragma unroll
        for (int db = 0; db < 2; db++) { dv[db] = mfma(da[h8 + db], pf[0][h8], dv[db]); dk[db] = mfma(qa[h8 + db], dsf[0][h8], dk[db]); }
        __builtin_amdgcn_sched_barrier(0);
        valu8(s1, p1, h8, pf[1][h8], dsf[1][h8]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int h8 = 0; h8 < 2; h8++)
#pragma unroll
        for (int db = 0; db < 2; db++) { dv[db] = mfma(da[h8 + db], pf[1][h8], dv[db]); dk[db] = mfma(qa[h8 + db], dsf[1][h8], dk[db]); }
      __builtin_amdgcn_sched_barrier(0);
    } else if constexpr (MODE == 2) {
#pragma unroll
      for (int q = 0; q < 2; q++) {
#pragma unroll
        for (int ks = 0; ks < 4; ks++) { dv[0] = mfma(qa[ks], kf[ks], dv[0]); dk[0] = mfma(da[ks], vf[ks], dk[0]); }
#pragma unroll
        for (int h8 = 0; h8 < 2; h8++)
#pragma unroll
          for (int db = 0; db < 2; db++) { dv[db] = mfma(da[h8 + db], kf[h8], dv[db]); dk[db] = mfma(qa[h8 + db], vf[h8], dk[db]); }
      }
    } else {
#pragma unroll
      for (int q = 0; q < 2; q++)
#pragma unroll
        for (int h8 = 0; h8 < 2; h8++) { valu8(dk[q], dv[q], h8, pf[q][h8], dsf[q][h8]); sink += (float)pf[q][h8][0] + (float)dsf[q][h8][1]; }
    }
  }
  float t = sink;
  for (int i = 0; i < 2; i++) t += dk[i][0] + dv[i][0];
  if (t == 12345.f) out[0] = t;
}

template <int MODE>
void run(const char* what, float* d) {
  const int iters = 50000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters, 0.18f, 1.f, 0.25f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  printf("%-80s %8.3f ms  = %7.1f ns per tile (2 waves per SIMD)\n", what, best, best * 1e6 / iters);
}

int main() {
  float* d; hipMalloc(&d, 4);
  run<2>("2: the 32 MFMAs of a tile alone", d);
  run<3>("3: the softmax-backward arithmetic of a tile alone", d);
  run<0>("0: program order  M_S0 V0 M_D0 M_S1 V1 M_D1", d);
  run<1>("1: software-pipelined  M_S0 | M_S1+V0 | M_D0+V1 | M_D1", d);
  return 0;
}
