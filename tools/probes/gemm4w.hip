// (round 6) 256 x 256 x 32-deep-step bf16 GEMM main loop with FOUR waves of 128 x 128 outputs each, one wave per SIMD, 512 registers per lane (VERDICT r05 item 4,
// DESIGN 6.1): a wave's 8 A and 8 B fragments of a 32-deep k step feed 64 v_mfma_f32_16x16x32_bf16 -- 32 KB of LDS fragment reads per wave and K tile against the
// 24 KB of each of the EIGHT 128 x 64 waves of the 8-phase kernel (tools/probes/gemm8p.hip, csrc/gemm8p.hip): 128 KB instead of 192 KB per K tile and compute unit.
// The 8-phase loop is limited by the board's power, not by issue slots, so the question this probe asks is energy: does a third fewer LDS bytes per FLOP
// buy clock?   C[M][N] = A[M][K] B[N][K]^T, bf16 in, bf16 out.
//   hipcc --offload-arch=gfx950 -O3 -o gemm4w gemm4w.hip && ./gemm4w [n = 8192] [reps]
//
// Structure.  Waves (wr = wave >> 1, wc = wave & 1) own rows wr * 128 + [0, 128), columns wc * 128 + [0, 128).  The operands move in 32-deep STEPS: a step's
// A and B rows (256 x 64 B each = 32 KB) occupy one of FIVE ring slots (160 KB), filled by LDS-DMA four steps ahead.  One barrier per step:
//     top of step s:  vmcnt(24)   my pieces of step s + 1 have landed (three younger steps stay in flight, 8 pieces each)
//                     lgkmcnt(0)  my fragments of step s are in registers
//                     s_barrier   => everyone's: slot s is free, slot s + 1 is complete
//                     issue the DMA of step s + 5 into slot s, the 16 fragment reads of step s + 1 (second register set), the 64 MFMAs of step s
// so the matrix pipe of a SIMD has one wave to feed it, and that wave's LDS / VMEM issue runs between its own MFMAs (they are independent: 64 accumulators).
// LDS image of a step: row r (A rows first, then B rows) = 64 B = four 16-byte chunks, chunk c at slot c ^ ((r >> 2) & 3) (conflict-free ds_read_b128 of
// 16 rows x 4 chunk columns).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef uint16_t bf16_t;
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
  typedef __attribute__((ext_vector_type(2))) float f2;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f2){lo, hi}, bf2));
}
__device__ __forceinline__ void glds16(uint32_t voff, uint64_t sbase, uint32_t lds_dst) {
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ uint64_t uniform64(uint64_t a) {
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a);
}

constexpr int SLOT = 32768;        // one 32-deep step: A rows [0, 256) then B rows [0, 256), 64 B each
constexpr int NSLOT = 5;
constexpr int BUF = SLOT;          // (host harness: LDS = NSLOT * SLOT)
#define SWZ4(r) (((r) >> 2) & 3)

// MODE 0: full kernel; 1: no MFMAs; 2: no fragment reads; 3: no LDS-DMA inside the loop; 4: MFMAs + barriers only
template <int MODE, int VER>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm8p_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C,
                                                                                               int M, int N, int K, int tiles_m, int tiles_n, unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  int tm, tn;
  {
    const int nt = tiles_m * tiles_n, b = blockIdx.x;
    const int q = nt / 8, r = nt % 8, xcd = b % 8;
    int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
    const int GW = 8, per = GW * tiles_m, grp = id / per, rem = id - grp * per;
    const int gw = min(GW, tiles_n - grp * GW);
    tm = rem / gw;
    tn = grp * GW + rem % gw;
  }
  const int ns = K / 32;      // steps

  // ---- staging: a DMA piece = 16 rows x 64 B (lane >> 2 = row, lane & 3 = 16-byte slot holding chunk slot ^ SWZ4(row)); a step has 16 A + 16 B pieces,
  //      wave w issues pieces w, w + 4, w + 8, w + 12 of each operand: rows 64 i + 16 w + (lane >> 2)
  const int r0 = wave * 16 + (lane >> 2), ch0 = (lane & 3) ^ SWZ4(r0);
  const uint32_t voff0 = (uint32_t)((int64_t)r0 * K * 2 + ch0 * 16), vstep = (uint32_t)((int64_t)64 * K * 2);
  const uint64_t baseA = uniform64((uint64_t)(uintptr_t)(A + (int64_t)(tm * 256) * K)), baseB = uniform64((uint64_t)(uintptr_t)(B + (int64_t)(tn * 256) * K));
  const uint32_t ldsw = lds0 + wave * 1024;
  auto stage = [&](int s, int slot) {      // step s -> ring slot
    if ((MODE == 3 || MODE == 4) && s >= NSLOT) return;
    const uint64_t off = (uint64_t)min(s, ns - 1) * 64;
    const uint32_t dst = ldsw + slot * SLOT;
#pragma unroll
    for (int i = 0; i < 4; i++) glds16(voff0 + i * vstep, baseA + off, dst + i * 4096);
#pragma unroll
    for (int i = 0; i < 4; i++) glds16(voff0 + i * vstep, baseB + off, dst + 16384 + i * 4096);
  };

  // ---- fragment read offsets inside a slot: 16x16x32 operand = row (lane & 15), 16 B = k chunk (lane >> 4); fragment i of a wave = fragment 0 + i * 1024
  const int ra = wr * 128 + (lane & 15), rb = wc * 128 + (lane & 15);
  const uint32_t aoff = ra * 64 + (((lane >> 4) ^ SWZ4(ra)) << 4), boff = 16384 + rb * 64 + (((lane >> 4) ^ SWZ4(rb)) << 4);

  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 8; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[2][8], fb[2][8];
  auto reads = [&](int set, int slot) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      fa[set][i] = (MODE == 2 || MODE == 4) ? fa[set][i] : *LDS_PTR(const bf16x8, smem + slot * SLOT + aoff + i * 1024);
      fb[set][i] = (MODE == 2 || MODE == 4) ? fb[set][i] : *LDS_PTR(const bf16x8, smem + slot * SLOT + boff + i * 1024);
    }
  };
  auto mma = [&](int set) {
    if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; i++) asm volatile("" ::"v"(fa[set][i]), "v"(fb[set][i]));
      return;
    }
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
      for (int j = 0; j < 8; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[set][j], fa[set][i], acc[i][j], 0, 0, 0);
  };
#define BAR() asm volatile("s_barrier" ::: "memory")
#define LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
  if (MODE == 2 || MODE == 4) {
#pragma unroll
    for (int set = 0; set < 2; set++)
#pragma unroll
      for (int f = 0; f < 8; f++)
        for (int e = 0; e < 8; e++) { fa[set][f][e] = (__bf16)(float)(lane + e - f); fb[set][f][e] = (__bf16)(float)(wave + e + f); }
  }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0t = __builtin_amdgcn_s_memrealtime();
  // ---- prologue: steps 0 .. 4 requested, step 0 landed and read
#pragma unroll
  for (int s = 0; s < NSLOT; s++) stage(s, s);
  VMCNT(32);
  BAR();
  reads(0, 0);
  // one step: `set` holds its fragments, slot = s % 5 (kept as a running index)
#define STEP(set, s, slot, nslot)                                                                                       \
  {                                                                                                                     \
    VMCNT(24);                                                                                                          \
    LGKM0();                                                                                                            \
    BAR(); __builtin_amdgcn_sched_barrier(0);                                                                           \
    stage((s) + NSLOT, slot);                                                                                           \
    reads((set) ^ 1, nslot);                                                                                            \
    mma(set);                                                                                                           \
  }
  int slot = 0;
  for (int s = 0; s < ns; s += 2) {
    const int n1 = slot + 1 == NSLOT ? 0 : slot + 1, n2 = n1 + 1 == NSLOT ? 0 : n1 + 1;
    STEP(0, s, slot, n1)
    STEP(1, s + 1, n1, n2)
    slot = n2;
  }
  VMCNT(0);
  if (stamps && tid == 0) { stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0; stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0t; }

  // ---- epilogue (plain): acc[i][j] = C^T fragment: row wr * 128 + i * 16 + (lane & 15), columns wc * 128 + j * 16 + 4 (lane >> 4) + [0, 4)
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int row = tm * 256 + wr * 128 + i * 16 + (lane & 15), col = tn * 256 + wc * 128 + j * 16 + 4 * (lane >> 4);
      const f32x4& c = acc[i][j];
      *(u32x2*)(C + (int64_t)row * N + col) = (u32x2){pack_bf2(c[0], c[1]), pack_bf2(c[2], c[3])};
    }
}

// ------------------------------------------------------------------------------------------------ host
static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static inline float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

static unsigned long long* g_stamps = nullptr;
static double g_cyc_per_ktile = 0, g_mhz = 0;
template <int MODE, int VER>
static float run(const bf16_t* A, const bf16_t* B, bf16_t* C, int M, int N, int K, int reps) {
  auto k = gemm8p_kernel<MODE, VER>;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * SLOT));
  const int tmn = M / 256, tnn = N / 256;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  if (!g_stamps) CK(hipMalloc(&g_stamps, 16 * 65536));
  for (int i = 0; i < 2; i++) hipLaunchKernelGGL(k, dim3(tmn * tnn), dim3(256), NSLOT * SLOT, 0, A, B, C, M, N, K, tmn, tnn, (unsigned long long*)nullptr);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k, dim3(tmn * tnn), dim3(256), NSLOT * SLOT, 0, A, B, C, M, N, K, tmn, tnn, (unsigned long long*)nullptr);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  // one more launch with per-workgroup stamps of the main loop: shader cycles (s_memtime) and 100 MHz ticks (s_memrealtime)
  hipLaunchKernelGGL(k, dim3(tmn * tnn), dim3(256), NSLOT * SLOT, 0, A, B, C, M, N, K, tmn, tnn, g_stamps);
  std::vector<unsigned long long> st(2 * tmn * tnn);
  CK(hipMemcpy(st.data(), g_stamps, st.size() * 8, hipMemcpyDeviceToHost));
  double cyc = 0, real = 0;
  for (int i = 0; i < tmn * tnn; i++) { cyc += (double)st[2 * i]; real += (double)st[2 * i + 1]; }
  g_cyc_per_ktile = cyc / (tmn * tnn) / (K / 64);
  g_mhz = cyc / real * 100.0;
  return ms / reps;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 8192, reps = argc > 2 ? atoi(argv[2]) : 10;
  const int M = n, N = n, K = argc > 3 ? atoi(argv[3]) : n;
  if (M % 256 || N % 256 || K % 128) { printf("M, N multiples of 256, K of 128\n"); return 1; }
  std::vector<uint16_t> hA((size_t)M * K), hB((size_t)N * K);
  uint32_t s = 12345;
  auto rnd = [&]() { float a = 0; for (int t = 0; t < 4; t++) { s = s * 1664525u + 1013904223u; a += (float)(s >> 8) * (1.f / 16777216.f) - 0.5f; } return a * 1.7f; };   // ~N(0, 1)
  for (auto& x : hA) x = f2bf(rnd());
  for (auto& x : hB) x = f2bf(rnd());
  bf16_t *A, *B, *C;
  CK(hipMalloc(&A, hA.size() * 2)); CK(hipMalloc(&B, hB.size() * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
  CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(C, 0, (size_t)M * N * 2));
  const double fl = 2.0 * M * N * K;
  std::vector<uint16_t> hC((size_t)M * N);
  auto check = [&](const char* tag) {
    CK(hipMemcpy(hC.data(), C, hC.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0;
    int nbad = 0;
    uint32_t s2 = 777;
    for (int it = 0; it < 2048; it++) {
      s2 = s2 * 1664525u + 1013904223u; const int m = (s2 >> 8) % M;
      s2 = s2 * 1664525u + 1013904223u; const int nn = (s2 >> 8) % N;
      double ref = 0;
      for (int k = 0; k < K; k++) ref += (double)bf2f(hA[(size_t)m * K + k]) * (double)bf2f(hB[(size_t)nn * K + k]);
      const double got = bf2f(hC[(size_t)m * N + nn]), err = fabs(got - ref) / (fabs(ref) + sqrt((double)K) * 0.05);
      if (err > worst) worst = err;
      if (err > 1.5e-2) { if (nbad < 5) printf("  MISMATCH C[%d][%d] = %g, reference %g\n", m, nn, got, ref); nbad++; }
    }
    printf("check %s: worst relative error %.3e over 2048 samples, %d bad\n", tag, worst, nbad);
    CK(hipMemset(C, 0, (size_t)M * N * 2));
    return nbad;
  };
  int bad = 0;
  run<0, 1>(A, B, C, M, N, K, 3);      // clock warm-up
  for (int round = 0; round < 3; round++) {
    const float t1 = run<0, 1>(A, B, C, M, N, K, reps), t2 = run<0, 2>(A, B, C, M, N, K, reps);
    printf("round %d   VER 1 %8.1f us %7.1f TFLOP/s     VER 2 %8.1f us %7.1f TFLOP/s  (%.0f cycles per K tile at %.0f MHz)\n", round, t1 * 1e3, fl / t1 / 1e9, t2 * 1e3, fl / t2 / 1e9, g_cyc_per_ktile, g_mhz);
  }
  run<0, 1>(A, B, C, M, N, K, 1); bad += check("VER 1");
  run<0, 2>(A, B, C, M, N, K, 1); bad += check("VER 2");
  auto abl = [&](const char* tag, float t) { printf("   %-28s %7.1f us  %5.0f cycles per K tile at %4.0f MHz\n", tag, t * 1e3, g_cyc_per_ktile, g_mhz); };
  printf("ablations (VER 1 / VER 2):\n");
  abl("VER 1 full", run<0, 1>(A, B, C, M, N, K, reps));                 abl("VER 2 full", run<0, 2>(A, B, C, M, N, K, reps));
  abl("VER 1 no MFMAs", run<1, 1>(A, B, C, M, N, K, reps));             abl("VER 2 no MFMAs", run<1, 2>(A, B, C, M, N, K, reps));
  abl("VER 1 no fragment reads", run<2, 1>(A, B, C, M, N, K, reps));    abl("VER 2 no fragment reads", run<2, 2>(A, B, C, M, N, K, reps));
  abl("VER 1 no DMA in the loop", run<3, 1>(A, B, C, M, N, K, reps));   abl("VER 2 no DMA in the loop", run<3, 2>(A, B, C, M, N, K, reps));
  abl("VER 1 MFMAs + barriers", run<4, 1>(A, B, C, M, N, K, reps));     abl("VER 2 MFMAs + barriers", run<4, 2>(A, B, C, M, N, K, reps));
  return bad ? 2 : 0;
}
