// Layout probe for gfx950: verifies the MFMA operand / accumulator lane maps,
// the ds_read_b64_tr_b16 transpose semantics and global_load_lds placement that
// the kernels in csrc/ assume.  Build: hipcc --offload-arch=gfx950 -O2 layout_probe.hip -o layout_probe
// Prints PASS/FAIL per assumption plus raw dumps for the transpose read.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include <cstdint>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); uint32_t r = u + 0x7fff + ((u >> 16) & 1); return (uint16_t)(r >> 16); }
static float bf2f(uint16_t h) { uint32_t u = ((uint32_t)h) << 16; float f; memcpy(&f, &u, 4); return f; }

// A: [32][16] row-major bf16, B: [16][32] row-major bf16 ; C [32][32]
__global__ void k_mfma32(const uint16_t* A, const uint16_t* B, float* C) {
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int e = 0; e < 8; e++) {
    int k = (l >> 5) * 8 + e;
    uint16_t av = A[(l & 31) * 16 + k];
    uint16_t bv = B[k * 32 + (l & 31)];
    a[e] = __builtin_bit_cast(__bf16, av);
    b[e] = __builtin_bit_cast(__bf16, bv);
  }
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 16; r++) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    int col = l & 31;
    C[row * 32 + col] = acc[r];
  }
}

// A: [16][32], B: [32][16], C [16][16]
__global__ void k_mfma16(const uint16_t* A, const uint16_t* B, float* C) {
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int e = 0; e < 8; e++) {
    int k = (l >> 4) * 8 + e;
    a[e] = __builtin_bit_cast(__bf16, A[(l & 15) * 32 + k]);
    b[e] = __builtin_bit_cast(__bf16, B[k * 16 + (l & 15)]);
  }
  f32x4 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; r++) {
    int row = (l >> 4) * 4 + r;
    int col = l & 15;
    C[row * 16 + col] = acc[r];
  }
}

// f32 mfma 32x32x2: A [32][2], B [2][32]
__global__ void k_mfma32f(const float* A, const float* B, float* C) {
  int l = threadIdx.x;
  float a = A[(l & 31) * 2 + (l >> 5)];
  float b = B[(l >> 5) * 32 + (l & 31)];
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 16; r++) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    C[row * 32 + (l & 31)] = acc[r];
  }
}

// transpose read dump: LDS filled with lds[i] = i (short). lane l address = base + l*8 bytes (4 shorts)
__global__ void k_tr(short* out, int mode) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  int l = threadIdx.x;
  for (int i = l; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  int off;
  if (mode == 0) off = l * 4;                     // contiguous 8B per lane
  else {
    // tile [rows][64 cols] shorts, pitch 64: lane i in 16-group reads row (i>>2), cols (i&3)*4 ; group g -> col block g*16
    int i = l & 15, g = l >> 4;
    off = (i >> 2) * 64 + g * 16 + (i & 3) * 4;
  }
  s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + off));
  for (int j = 0; j < 4; j++) out[l * 4 + j] = t[j];
}

// global_load_lds 16B: LDS dest = base + lane*16 ?  src per lane = g + perm(lane)*8 shorts
__global__ void k_glds(const short* g, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[2048];
  int l = threadIdx.x;
  for (int i = l; i < 2048; i += 64) lds[i] = -1;
  __syncthreads();
  int src = (l ^ 5);  // permuted source
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + src * 8),
                                   (__attribute__((address_space(3))) void*)(lds + 512), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = l; i < 2048; i += 64) out[i] = lds[i];
}

int main() {
  srand(1);
  bool all = true;
  {
    std::vector<uint16_t> A(32 * 16), B(16 * 32);
    for (auto& v : A) v = f2bf((rand() % 17 - 8) / 4.0f);
    for (auto& v : B) v = f2bf((rand() % 13 - 6) / 2.0f);
    uint16_t *dA, *dB; float* dC;
    CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dC, 32 * 32 * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
    k_mfma32<<<1, 64>>>(dA, dB, dC);
    std::vector<float> C(32 * 32);
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    double maxe = 0;
    for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) {
      float r = 0; for (int k = 0; k < 16; k++) r += bf2f(A[i * 16 + k]) * bf2f(B[k * 32 + j]);
      maxe = fmax(maxe, fabs(r - C[i * 32 + j]));
    }
    printf("mfma_32x32x16_bf16 layout: maxerr %.3g %s\n", maxe, maxe < 1e-3 ? "PASS" : "FAIL");
    all &= maxe < 1e-3;
  }
  {
    std::vector<uint16_t> A(16 * 32), B(32 * 16);
    for (auto& v : A) v = f2bf((rand() % 17 - 8) / 4.0f);
    for (auto& v : B) v = f2bf((rand() % 13 - 6) / 2.0f);
    uint16_t *dA, *dB; float* dC;
    CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dC, 16 * 16 * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
    k_mfma16<<<1, 64>>>(dA, dB, dC);
    std::vector<float> C(16 * 16);
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    double maxe = 0;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
      float r = 0; for (int k = 0; k < 32; k++) r += bf2f(A[i * 32 + k]) * bf2f(B[k * 16 + j]);
      maxe = fmax(maxe, fabs(r - C[i * 16 + j]));
    }
    printf("mfma_16x16x32_bf16 layout: maxerr %.3g %s\n", maxe, maxe < 1e-3 ? "PASS" : "FAIL");
    all &= maxe < 1e-3;
  }
  {
    std::vector<float> A(32 * 2), B(2 * 32);
    for (auto& v : A) v = (rand() % 1000) / 777.0f;
    for (auto& v : B) v = (rand() % 1000) / 333.0f;
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, 32 * 32 * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    k_mfma32f<<<1, 64>>>(dA, dB, dC);
    std::vector<float> C(32 * 32);
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    double maxe = 0;
    for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) {
      float r = fmaf(A[i * 2 + 1], B[32 + j], A[i * 2] * B[j]);
      maxe = fmax(maxe, fabs(r - C[i * 32 + j]));
    }
    printf("mfma_32x32x2_f32 layout: maxerr %.3g %s\n", maxe, maxe < 1e-5 ? "PASS" : "FAIL");
    all &= maxe < 1e-5;
  }
  for (int mode = 0; mode < 2; mode++) {
    short* d; CK(hipMalloc(&d, 64 * 4 * 2));
    k_tr<<<1, 64>>>(d, mode);
    std::vector<short> o(256);
    CK(hipMemcpy(o.data(), d, 512, hipMemcpyDeviceToHost));
    printf("tr16_b64 mode %d dump (lane: 4 values):\n", mode);
    for (int l = 0; l < 64; l++) { printf(" L%02d:%4d %4d %4d %4d", l, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]); if ((l & 3) == 3) printf("\n"); }
    // expectation: out[l][j] = in[lane (l&~15) + 4j + ((l&15)>>2)][(l&3)]
    bool ok = true;
    for (int l = 0; l < 64; l++) for (int j = 0; j < 4; j++) {
      int srcl = (l & ~15) + 4 * j + ((l & 15) >> 2), e = l & 3;
      int off;
      if (mode == 0) off = srcl * 4; else { int i = srcl & 15, g = srcl >> 4; off = (i >> 2) * 64 + g * 16 + (i & 3) * 4; }
      if (o[l * 4 + j] != (short)(off + e)) ok = false;
    }
    printf("tr16_b64 semantic hypothesis A (out[l][j]=in[4j+(l>>2)][l&3]) mode %d: %s\n", mode, ok ? "PASS" : "FAIL");
    if (mode == 1) {
      // derived expectation: lane l gets tile[row j][col g*16 + (l&15)]
      bool ok2 = true;
      for (int l = 0; l < 64; l++) for (int j = 0; j < 4; j++) if (o[l * 4 + j] != (short)(j * 64 + (l >> 4) * 16 + (l & 15))) ok2 = false;
      printf("tr16_b64 tile view (lane l elem j = tile[j][16g + (l&15)]): %s\n", ok2 ? "PASS" : "FAIL");
      all &= ok2;
    }
  }
  {
    std::vector<short> g(64 * 8);
    for (int i = 0; i < 512; i++) g[i] = (short)i;
    short *dg, *dout; CK(hipMalloc(&dg, 1024)); CK(hipMalloc(&dout, 4096));
    CK(hipMemcpy(dg, g.data(), 1024, hipMemcpyHostToDevice));
    k_glds<<<1, 64>>>(dg, dout);
    std::vector<short> o(2048);
    CK(hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost));
    bool ok = true;
    for (int l = 0; l < 64; l++) for (int e = 0; e < 8; e++) if (o[512 + l * 8 + e] != (short)((l ^ 5) * 8 + e)) ok = false;
    for (int i = 0; i < 512; i++) if (o[i] != -1) ok = false;
    printf("global_load_lds b128: dest = base + lane*16, per-lane src: %s\n", ok ? "PASS" : "FAIL");
    all &= ok;
  }
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d clock %d kHz LDS/block %zu\n", p.name, p.multiProcessorCount, p.clockRate, p.sharedMemPerBlock);
  printf("ALL %s\n", all ? "PASS" : "FAIL");
  return 0;
}
