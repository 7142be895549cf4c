// Layout probe for v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 operands, E8M0 block scales): which K index does byte p of lane l hold, and which
// (row, 32-block) does lane l's scale byte scale?  One wave, D = A B^T with A[16][128], B[16][128] small-integer e4m3 values, all hypotheses
// checked against a host product.   hipcc --offload-arch=gfx950 -O3 -o mx16_probe mx16_probe.hip && ./mx16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const unsigned char* A, const unsigned char* B, const unsigned* SA, const unsigned* SB, float* D, int hyp, int osa, int osb) {
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  unsigned char ab[32], bb[32];
  for (int p = 0; p < 32; p++) {
    int kk = hyp == 0 ? 32 * g + p : (p < 16 ? 16 * g + p : 64 + 16 * g + (p - 16));
    ab[p] = A[r * 128 + kk];
    bb[p] = B[r * 128 + kk];
  }
  i32x8 a, b;
  for (int i = 0; i < 8; i++) {
    a[i] = ab[4 * i] | (ab[4 * i + 1] << 8) | (ab[4 * i + 2] << 16) | (ab[4 * i + 3] << 24);
    b[i] = bb[4 * i] | (bb[4 * i + 1] << 8) | (bb[4 * i + 2] << 16) | (bb[4 * i + 3] << 24);
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  const int sa = (int)SA[l], sb = (int)SB[l];
  if (osa == 0 && osb == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
  else if (osa == 1 && osb == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 1, sa, 0, sb);
  else if (osa == 2 && osb == 3) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 2, sa, 3, sb);
  for (int i = 0; i < 4; i++) D[l * 4 + i] = c[i];
}

static unsigned char e4m3(int v) {   // small integers -8..8 exactly
  if (v == 0) return 0;
  unsigned char s = v < 0 ? 0x80 : 0; v = abs(v);
  int e = 0; while ((1 << (e + 1)) <= v) e++;          // v = 2^e * (1 + m/8)
  int m = (v - (1 << e)) * 8 / (1 << e);
  return s | ((e + 7) << 3) | m;
}

int main() {
  std::vector<unsigned char> A(16 * 128), B(16 * 128);
  std::vector<int> Ai(16 * 128), Bi(16 * 128);
  srand(1);
  for (int i = 0; i < 16 * 128; i++) { Ai[i] = rand() % 9 - 4; Bi[i] = rand() % 9 - 4; A[i] = e4m3(Ai[i]); B[i] = e4m3(Bi[i]); }
  unsigned char *dA, *dB; unsigned *dSA, *dSB; float* dD;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dSA, 256); hipMalloc(&dSB, 256); hipMalloc(&dD, 1024);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  std::vector<float> D(256);
  // ---- 1. K layout with unit scales (127 in every byte)
  std::vector<unsigned> S1(64, 0x7f7f7f7fu);
  hipMemcpy(dSA, S1.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dSB, S1.data(), 256, hipMemcpyHostToDevice);
  for (int hyp = 0; hyp < 2; hyp++) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dSA, dSB, dD, hyp, 0, 0);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    // C layout hypothesis (16x16 family): lane l, reg i: row (l >> 4) * 4 + i of A ... col l & 15 of B (D = A B^T with a = first operand)
    int bad = 0, badT = 0;
    for (int l = 0; l < 64; l++) for (int i = 0; i < 4; i++) {
      const int ra = (l >> 4) * 4 + i, rb = l & 15;
      double ref = 0, refT = 0;
      for (int kk = 0; kk < 128; kk++) { ref += Ai[ra * 128 + kk] * Bi[rb * 128 + kk]; refT += Ai[rb * 128 + kk] * Bi[ra * 128 + kk]; }
      bad += fabs(D[l * 4 + i] - ref) > 1e-3; badT += fabs(D[l * 4 + i] - refT) > 1e-3;
    }
    printf("K layout hypothesis %d (%s): mismatches %d of 256 with D[(l>>4)*4+i][l&15] = A-row x B-row, %d transposed\n", hyp,
           hyp == 0 ? "lane group g holds k = 32 g + [0, 32)" : "regs 0-3: k = 16 g + [0,16), regs 4-7: k = 64 + 16 g + [0,16)", bad, badT);
  }
  // ---- 2. scales: lane (r, g) of operand A carries in byte `sel` the E8M0 exponent 127 + (r % 3) + 2 * g; B unit.  Expect block g of row r scaled.
  for (int trial = 0; trial < 3; trial++) {
    const int osa = trial == 0 ? 0 : trial == 1 ? 1 : 2, osb = trial == 2 ? 3 : 0;
    std::vector<unsigned> SA(64), SB(64);
    for (int l = 0; l < 64; l++) {
      const int r = l & 15, g = l >> 4;
      unsigned ea = 127 + (r % 3) + 2 * g, eb = 127 + (trial == 2 ? (r % 2) + g : 0);
      SA[l] = 0x7f7f7f7fu; SB[l] = 0x7f7f7f7fu;
      SA[l] = (SA[l] & ~(0xffu << (8 * osa))) | (ea << (8 * osa));
      SB[l] = (SB[l] & ~(0xffu << (8 * osb))) | (eb << (8 * osb));
    }
    hipMemcpy(dSA, SA.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dSB, SB.data(), 256, hipMemcpyHostToDevice);
    for (int hyp = 0; hyp < 2; hyp++) {
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dSA, dSB, dD, hyp, osa, osb);
      hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
      int bad = 0;
      for (int l = 0; l < 64; l++) for (int i = 0; i < 4; i++) {
        const int ra = (l >> 4) * 4 + i, rb = l & 15;
        double ref = 0;
        for (int kk = 0; kk < 128; kk++) {
          const int blk = kk / 32;
          const double sa = ldexp(1.0, (ra % 3) + 2 * blk), sb = ldexp(1.0, trial == 2 ? (rb % 2) + blk : 0);
          ref += Ai[ra * 128 + kk] * sa * Bi[rb * 128 + kk] * sb;
        }
        bad += fabs(D[l * 4 + i] - ref) > 1e-3 * (1 + fabs(ref));
      }
      printf("scales trial %d (op_sel a %d b %d), K hypothesis %d: lane (r, g) scales row r block g -> mismatches %d of 256\n", trial, osa, osb, hyp, bad);
    }
  }
  return 0;
}
