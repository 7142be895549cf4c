// 256 x 256 x 64 bf16 GEMM main loop on the de-phased 8-phase structure (cdna_hip_programming.md 5, "The 256^2 8-phase template", T3+T4, T5),
// written from the guide's description (its example source is not in this image).  C[M][N] = A[M][K] B[N][K]^T, bf16 in, bf16 out.
//   hipcc --offload-arch=gfx950 -O3 -o gemm8p gemm8p.hip && ./gemm8p [n = 8192] [reps]
//
// Structure.  8 waves = 2 groups (wr = wave >> 2) x 4 column waves (wc = wave & 3).  Group 1 runs ONE barrier behind group 0, so between any two
// consecutive barriers one group issues its LDS fragment reads + LDS-DMA while the other issues MFMAs (one wave of each group per SIMD:
// matrix pipe beside LDS / VMEM issue, and s_setprio has roles to arbitrate).  A K tile (64 deep) is four phases, one output quadrant each:
//   P1  read B0 (4 x ds_read_b128), A0 (8)   C00 += A0 B0          P3  read A1 (8)    C11 += A1 B1
//   P2  read B1 (4)                          C01 += A0 B1          P4  (B0 kept)      C10 += A1 B0
// A wave's rows are {qm * 128 + wr * 64 + [0, 64)}, its columns {qn * 128 + wc * 32 + [0, 32)}: quadrant (qm, qn) reads half-tile A[qm] / B[qn]
// (128 rows x 64 k = 16 KB each), so every half-tile has ONE reading phase per K tile and can be restaged one phase later.  Every phase
// issues the two LDS-DMA instructions of one half-tile, almost two K tiles ahead:  P2(t): A0(t+2)  P3(t): B0(t+2)  P4(t): B1(t+2)
// P1(t+1): A1(t+2).  One counted wait per K tile, s_waitcnt vmcnt(6) in P4 (three half-tiles stay in flight), never 0 in the loop.
// Hazards (guide): data waited for in phase p is read in phase p+1 or later; a half-tile is restaged >= 1 phase after the phase whose
// reads were retired (lgkmcnt(0)) BEFORE that phase's first barrier.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef uint16_t bf16_t;
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

#ifndef SWZ
#define SWZ(r) (((r) >> 1) & 7)
#endif
#ifndef PRIO
#define PRIO 1
#endif
#ifndef STAGGER
#define STAGGER 1
#endif
#ifndef MF16
#define MF16 0      // 1: v_mfma_f32_16x16x32_bf16 (16 per phase) instead of v_mfma_f32_32x32x16_bf16 (8 per phase)
#endif

__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
  typedef __attribute__((ext_vector_type(2))) float f2;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f2){lo, hi}, bf2));
}

// one LDS-DMA instruction: 64 lanes x 16 B from (wave-uniform 64-bit base + per-lane 32-bit offset) to LDS [m0, m0 + 1024)
__device__ __forceinline__ void glds16(uint32_t voff, uint64_t sbase, uint32_t lds_dst) {
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ uint64_t uniform64(uint64_t a) {
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a);
}

constexpr int HT = 16384;          // bytes of a half-tile (128 rows x 128 B)
constexpr int BUF = 4 * HT;        // A0, A1, B0, B1 of one K tile
constexpr int XA0 = 0, XA1 = HT, XB0 = 2 * HT, XB1 = 3 * HT;

// VER 1: reads 12 / 4 / 8 / 0 per phase, lgkmcnt(0) BEFORE the phase's first barrier, restage one phase after the read, vmcnt(6) in P4.
// VER 2: reads 8 / 4 / 8 / 4 (the next K tile's B0 is read in P4 into a second register set), lgkmcnt(0) AFTER the first barrier (the wave
//        arrives at the barrier as soon as its reads and DMA are ISSUED), restage two phases after the read, every half-tile is staged six
//        phases before it is read, vmcnt(8) in P1 and P3 (four half-tiles stay in flight).
// MODE 0: full kernel; 1: no MFMAs (operand stream + fragment reads only); 2: no fragment reads; 3: no LDS-DMA inside the loop; 4: MFMAs + barriers only
template <int MODE, int VER>
__global__ __launch_bounds__(512) void gemm8p_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C, int M, int N, int K,
                                                     int tiles_m, int tiles_n, unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(char, smem);

  // XCD-contiguous tile ranges (block b runs on XCD b % 8), inside a range column groups of 8 n-tiles walked m-major
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
  const int nk = K / 64;

  // ---- staging addresses: piece i (0, 1) of a half-tile: LDS rows i * 64 + wave * 8 + (lane >> 3), 16-B slot lane & 7 holds chunk slot ^ SWZ(row)
  uint32_t voffA[2], voffB[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int r = i * 64 + wave * 8 + (lane >> 3), chunk = (lane & 7) ^ SWZ(r);
    voffA[i] = (uint32_t)((int64_t)r * K * 2 + chunk * 16);
    voffB[i] = voffA[i];
  }
  const uint64_t baseA0 = uniform64((uint64_t)(uintptr_t)(A + (int64_t)(tm * 256) * K)), baseA1 = uniform64((uint64_t)(uintptr_t)(A + (int64_t)(tm * 256 + 128) * K));
  const uint64_t baseB0 = uniform64((uint64_t)(uintptr_t)(B + (int64_t)(tn * 256) * K)), baseB1 = uniform64((uint64_t)(uintptr_t)(B + (int64_t)(tn * 256 + 128) * K));
  const uint32_t ldsw = lds0 + wave * 1024;
  auto stage = [&](uint64_t base, const uint32_t (&voff)[2], int x, int t, int buf) {     // half-tile x (byte offset in the buffer) of K tile t -> buffer buf
    const uint64_t src = base + (uint64_t)min(t, nk - 1) * 128;
    const uint32_t dst = ldsw + buf * BUF + x;
    if ((MODE == 3 || MODE == 4) && t >= 2) return;
    glds16(voff[0], src, dst);
    glds16(voff[1], src, dst + 8192);
  };

  // ---- fragment read offsets (bytes inside a half-tile).  32x32x16 operand: row (lane & 31), 16 B = k chunk 2 ks + (lane >> 5), ks < 4;
  //      16x16x32 operand: row (lane & 15), 16 B = k chunk 4 ks + (lane >> 4), ks < 2.  fa[8] / fb[4] either way.
  uint32_t aoff[8], boff[4];
#if MF16
#pragma unroll
  for (int ks = 0; ks < 2; ks++) {
    const int kp = ks * 4 + (lane >> 4);
#pragma unroll
    for (int i = 0; i < 4; i++) { const int r = wr * 64 + i * 16 + (lane & 15); aoff[i * 2 + ks] = r * 128 + ((kp ^ SWZ(r)) << 4); }
#pragma unroll
    for (int j = 0; j < 2; j++) { const int r = wc * 32 + j * 16 + (lane & 15); boff[j * 2 + ks] = r * 128 + ((kp ^ SWZ(r)) << 4); }
  }
#else
#pragma unroll
  for (int ks = 0; ks < 4; ks++) {
    const int kp = ks * 2 + (lane >> 5);
#pragma unroll
    for (int i = 0; i < 2; i++) { const int r = wr * 64 + i * 32 + (lane & 31); aoff[i * 4 + ks] = r * 128 + ((kp ^ SWZ(r)) << 4); }
    const int rb = wc * 32 + (lane & 31);
    boff[ks] = rb * 128 + ((kp ^ SWZ(rb)) << 4);
  }
#endif

#if MF16
  f32x4 acc[2][2][8];      // [qm][qn][i * 2 + j]: C^T fragments of 16 x 16 (lane & 15 = output row, 4 consecutive columns per lane)
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int i = 0; i < 8; i++) acc[a][b][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#else
  f32x16 acc[2][2][2];     // [qm][qn][i]: C^T fragments (lane & 31 = output row, registers = 4-column groups)
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[a][b][i][r] = 0.f;
#endif

  bf16x8 fa[8], fb0[4], fb1[4], fb0n[4];
  auto readA = [&](int buf, int x) {
#pragma unroll
    for (int f = 0; f < 8; f++) fa[f] = (MODE == 2 || MODE == 4) ? fa[f] : *LDS_PTR(const bf16x8, smem + buf * BUF + x + aoff[f]);
  };
  auto readB = [&](bf16x8 (&fb)[4], int buf, int x) {
#pragma unroll
    for (int f = 0; f < 4; f++) fb[f] = (MODE == 2 || MODE == 4) ? fb[f] : *LDS_PTR(const bf16x8, smem + buf * BUF + x + boff[f]);
  };
  auto mma = [&](auto& c, const bf16x8 (&fb)[4]) {
    if (MODE == 1) {
#pragma unroll
      for (int f = 0; f < 4; f++) asm volatile("" ::"v"(fa[f]), "v"(fa[4 + f]), "v"(fb[f]));
      return;
    }
    if (PRIO) __builtin_amdgcn_s_setprio(1);
#if MF16
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) c[i * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j * 2 + ks], fa[i * 2 + ks], c[i * 2 + j], 0, 0, 0);
#else
#pragma unroll
    for (int ks = 0; ks < 4; ks++)
#pragma unroll
      for (int i = 0; i < 2; i++) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks], fa[i * 4 + ks], c[i], 0, 0, 0);
#endif
    if (PRIO) __builtin_amdgcn_s_setprio(0);
  };
#define BAR() asm volatile("s_barrier" ::: "memory")
#define LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

  if (MODE == 2 || MODE == 4) {
#pragma unroll
    for (int f = 0; f < 4; f++)
      for (int e = 0; e < 8; e++) { fa[f][e] = (__bf16)(float)(lane + e); fa[4 + f][e] = (__bf16)(float)(lane - e); fb0[f][e] = (__bf16)(float)(wave + e); fb1[f][e] = (__bf16)(float)(f + e); fb0n[f][e] = (__bf16)(float)(f - e); }
  }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if constexpr (VER == 1) {
  // ---- prologue: K tile 0 whole, three half-tiles of K tile 1
  stage(baseA0, voffA, XA0, 0, 0); stage(baseB0, voffB, XB0, 0, 0); stage(baseB1, voffB, XB1, 0, 0); stage(baseA1, voffA, XA1, 0, 0);
  stage(baseA0, voffA, XA0, 1, 1); stage(baseB0, voffB, XB0, 1, 1); stage(baseB1, voffB, XB1, 1, 1);
  VMCNT(6);
  BAR();
  if (STAGGER && wr == 1) BAR();       // group 1 runs one barrier behind from here on

  // one K tile = four phases; `t` is the K tile being multiplied, cur = its buffer (compile-time in the unrolled pair)
#define KTILE(t, cur)                                                                                                   \
  {                                                                                                                     \
    /* P1 */                                                                                                            \
    readB(fb0, cur, XB0); __builtin_amdgcn_sched_barrier(0); readA(cur, XA0);                                            \
    stage(baseA1, voffA, XA1, (t) + 1, (cur) ^ 1);                                                                                 \
    LGKM0(); BAR(); __builtin_amdgcn_sched_barrier(0);                                                                  \
    mma(acc[0][0], fb0);                                                                                                \
    __builtin_amdgcn_sched_barrier(0); BAR();                                                                           \
    /* P2 */                                                                                                            \
    readB(fb1, cur, XB1);                                                                                               \
    stage(baseA0, voffA, XA0, (t) + 2, cur);                                                                                 \
    LGKM0(); BAR(); __builtin_amdgcn_sched_barrier(0);                                                                  \
    mma(acc[0][1], fb1);                                                                                                \
    __builtin_amdgcn_sched_barrier(0); BAR();                                                                           \
    /* P3 */                                                                                                            \
    readA(cur, XA1);                                                                                                    \
    stage(baseB0, voffB, XB0, (t) + 2, cur);                                                                                 \
    LGKM0(); BAR(); __builtin_amdgcn_sched_barrier(0);                                                                  \
    mma(acc[1][1], fb1);                                                                                                \
    __builtin_amdgcn_sched_barrier(0); BAR();                                                                           \
    /* P4 */                                                                                                            \
    stage(baseB1, voffB, XB1, (t) + 2, cur);                                                                                 \
    VMCNT(6);                                                                                                           \
    BAR(); __builtin_amdgcn_sched_barrier(0);                                                                           \
    mma(acc[1][0], fb0);                                                                                                \
    __builtin_amdgcn_sched_barrier(0); BAR();                                                                           \
  }
  for (int t = 0; t < nk; t += 2) {
    KTILE(t, 0)
    KTILE(t + 1, 1)
  }
  } else {
  // ---- VER 2.  Stagings in flight at the loop entry (as if phases had run before): tile 0 whole, B0 A0 B1 of tile 1; B0(0) is read here.
  stage(baseA0, voffA, XA0, 0, 0); stage(baseB0, voffB, XB0, 0, 0); stage(baseB1, voffB, XB1, 0, 0); stage(baseA1, voffA, XA1, 0, 0);
  stage(baseB0, voffB, XB0, 1, 1); stage(baseA0, voffA, XA0, 1, 1); stage(baseB1, voffB, XB1, 1, 1);
  VMCNT(6);
  BAR();
  readB(fb0, 0, XB0);
  LGKM0();
  if (STAGGER && wr == 1) BAR();
  // phase g stages (six phases ahead of its read):  P1(t): A1(t+1)   P2(t): B0(t+2)   P3(t): A0(t+2)   P4(t): B1(t+2)
#define KTILE2(t, cur, FB0, FB0N)                                                                                       \
  {                                                                                                                     \
    /* P1: read A0(t) */                                                                                                \
    readA(cur, XA0);                                                                                                    \
    stage(baseA1, voffA, XA1, (t) + 1, (cur) ^ 1);                                                                      \
    VMCNT(8);                                                                                                           \
    BAR(); LGKM0(); __builtin_amdgcn_sched_barrier(0);                                                                  \
    mma(acc[0][0], FB0);                                                                                                \
    __builtin_amdgcn_sched_barrier(0); BAR();                                                                           \
    /* P2: read B1(t) */                                                                                                \
    readB(fb1, cur, XB1);                                                                                               \
    stage(baseB0, voffB, XB0, (t) + 2, cur);                                                                            \
    BAR(); LGKM0(); __builtin_amdgcn_sched_barrier(0);                                                                  \
    mma(acc[0][1], fb1);                                                                                                \
    __builtin_amdgcn_sched_barrier(0); BAR();                                                                           \
    /* P3: read A1(t) */                                                                                                \
    readA(cur, XA1);                                                                                                    \
    stage(baseA0, voffA, XA0, (t) + 2, cur);                                                                            \
    VMCNT(8);                                                                                                           \
    BAR(); LGKM0(); __builtin_amdgcn_sched_barrier(0);                                                                  \
    mma(acc[1][1], fb1);                                                                                                \
    __builtin_amdgcn_sched_barrier(0); BAR();                                                                           \
    /* P4: read B0(t+1) into the other register set */                                                                  \
    readB(FB0N, (cur) ^ 1, XB0);                                                                                        \
    stage(baseB1, voffB, XB1, (t) + 2, cur);                                                                            \
    BAR(); LGKM0(); __builtin_amdgcn_sched_barrier(0);                                                                  \
    mma(acc[1][0], FB0);                                                                                                \
    __builtin_amdgcn_sched_barrier(0); BAR();                                                                           \
  }
  for (int t = 0; t < nk; t += 2) {
    KTILE2(t, 0, fb0, fb0n)
    KTILE2(t + 1, 1, fb0n, fb0)
  }
  }
  VMCNT(0);
  if (stamps && tid == 0) { stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0; stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
  if (STAGGER && wr == 0) BAR();       // rejoin

  // ---- epilogue (plain).  32x32: acc[qm][qn][i] is a C^T fragment: row = qm*128 + wr*64 + i*32 + (lane & 31), cols qn*128 + wc*32 + 8 g + 4 (lane >> 5) + [0, 4);
  //      16x16: acc[qm][qn][i*2+j]: row = qm*128 + wr*64 + i*16 + (lane & 15), cols qn*128 + wc*32 + j*16 + 4 (lane >> 4) + [0, 4)
#pragma unroll
  for (int qm = 0; qm < 2; qm++)
#pragma unroll
    for (int qn = 0; qn < 2; qn++) {
#if MF16
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const int row = tm * 256 + qm * 128 + wr * 64 + i * 16 + (lane & 15), col = tn * 256 + qn * 128 + wc * 32 + j * 16 + 4 * (lane >> 4);
          const f32x4& c = acc[qm][qn][i * 2 + j];
          *(u32x2*)(C + (int64_t)row * N + col) = (u32x2){pack_bf2(c[0], c[1]), pack_bf2(c[2], c[3])};
        }
#else
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int row = tm * 256 + qm * 128 + wr * 64 + i * 32 + (lane & 31);
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const int col = tn * 256 + qn * 128 + wc * 32 + 8 * g + 4 * (lane >> 5);
          const f32x16& c = acc[qm][qn][i];
          *(u32x2*)(C + (int64_t)row * N + col) = (u32x2){pack_bf2(c[4 * g], c[4 * g + 1]), pack_bf2(c[4 * g + 2], c[4 * g + 3])};
        }
      }
#endif
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
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF));
  const int tmn = M / 256, tnn = N / 256;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  if (!g_stamps) CK(hipMalloc(&g_stamps, 16 * 65536));
  for (int i = 0; i < 2; i++) hipLaunchKernelGGL(k, dim3(tmn * tnn), dim3(512), 2 * BUF, 0, A, B, C, M, N, K, tmn, tnn, (unsigned long long*)nullptr);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k, dim3(tmn * tnn), dim3(512), 2 * BUF, 0, A, B, C, M, N, K, tmn, tnn, (unsigned long long*)nullptr);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  // one more launch with per-workgroup stamps of the main loop: shader cycles (s_memtime) and 100 MHz ticks (s_memrealtime)
  hipLaunchKernelGGL(k, dim3(tmn * tnn), dim3(512), 2 * BUF, 0, A, B, C, M, N, K, tmn, tnn, g_stamps);
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
