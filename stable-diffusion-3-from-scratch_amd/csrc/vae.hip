// FLUX-VAE (diffusers AutoencoderKL) building blocks for gfx950 -- SURVEY row V, first correct HIP path.
// Activations are NHWC (channels innermost, multiples of 8) so that every access is a 16-byte vector and a 3x3
// convolution becomes  im2col (this file)  x  the MFMA GEMM of gemm_dma.hip / gemm.hip  with the weight re-laid as
// [Cout][kh][kw][Cin].  GroupNorm(+SiLU) is two passes over the tensor (statistics with one atomic pair per group and
// workgroup, then normalise / affine / SiLU / convert), the mid-block attention is GEMM + row softmax + GEMM.
#include "common.h"

namespace {

inline int grid_cap(int64_t n, int bs) { int64_t g = (n + bs - 1) / bs; return (int)(g < 1 ? 1 : g > 65535 ? 65535 : g); }

// NCHW (fp32 or bf16) -> NHWC bf16 with the channel count padded to Cp (zero fill).  One thread per (pixel, 8 channels).
template <typename TI>
__global__ void nchw_to_nhwc_kernel(const TI* __restrict__ src, bf16_t* __restrict__ dst, int B, int C, int HW, int Cp, float scale, float shift) {
  const int cg = Cp / 8;
  const int64_t total = (int64_t)B * HW * cg;
  for (int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (int64_t)gridDim.x * blockDim.x) {
    const int g = (int)(gid % cg);
    const int64_t pix = gid / cg;
    const int p = (int)(pix % HW);
    const int64_t b = pix / HW;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int c = g * 8 + e;
      v[e] = c < C ? (io<TI>::ld(src + (b * C + c) * HW + p) + shift) * scale : 0.f;
    }
    st8(dst + pix * Cp + g * 8, v);
  }
}

// NHWC fp32 (row pitch ld >= C) -> NCHW fp32, optional clamp to [lo, hi].  One thread per output element (C is tiny here).
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int C, int HW, int ld, float lo, float hi) {
  const int64_t total = (int64_t)B * C * HW;
  for (int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(gid % HW);
    const int64_t bc = gid / HW;
    const int c = (int)(bc % C);
    const int64_t b = bc / C;
    dst[gid] = fminf(fmaxf(src[(b * HW + p) * ld + c], lo), hi);
  }
}

// im2col for a 3x3 convolution on NHWC bf16.  Output row m = (b, yo, xo), column (kh*3 + kw)*C + c.
//   mode 0: stride 1, padding 1                           (Ho, Wo) = (H, W)
//   mode 1: stride 2, padding (0,1,0,1) -- Downsample2D   (Ho, Wo) = (H/2, W/2)
//   mode 2: nearest x2 upsample, then stride 1, padding 1 (Ho, Wo) = (2H, 2W); the upsampled tensor is never materialised
__global__ void im2col3x3_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int B, int H, int W, int C, int mode, int Ho, int Wo) {
  const int cg = C / 8;
  const int64_t total = (int64_t)B * Ho * Wo * 9 * cg;
  for (int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (int64_t)gridDim.x * blockDim.x) {
    const int g = (int)(gid % cg);
    int64_t r = gid / cg;
    const int tap = (int)(r % 9);
    r /= 9;
    const int xo = (int)(r % Wo);
    r /= Wo;
    const int yo = (int)(r % Ho);
    const int64_t b = r / Ho;
    const int kh = tap / 3, kw = tap - kh * 3;
    int yi, xi;
    bool ok;
    if (mode == 1) {
      yi = 2 * yo + kh; xi = 2 * xo + kw;
      ok = yi < H && xi < W;
    } else if (mode == 2) {
      const int yu = yo + kh - 1, xu = xo + kw - 1;
      ok = yu >= 0 && xu >= 0 && yu < Ho && xu < Wo;
      yi = yu >> 1; xi = xu >> 1;
    } else {
      yi = yo + kh - 1; xi = xo + kw - 1;
      ok = yi >= 0 && xi >= 0 && yi < H && xi < W;
    }
    u32x4 v = {0, 0, 0, 0};
    if (ok) v = *(const u32x4*)(src + ((b * H + yi) * W + xi) * (int64_t)C + g * 8);
    *(u32x4*)(dst + gid * 8) = v;
  }
}

// GroupNorm statistics: sums[b][g] += (sum x, sum x^2) over the HW x (C/G) elements of the group.
// block = 256 threads = (C/8) column threads x row lanes; grid = (row chunks, B).
template <typename TI>
__global__ __launch_bounds__(256) void gn_stats_kernel(const TI* __restrict__ x, int HW, int C, int G, int rows_per_block, float* __restrict__ sums) {
  __shared__ float s[2 * 64];   // G <= 64
  const int cg = C / 8, tx = threadIdx.x % cg, ty = threadIdx.x / cg, nty = 256 / cg;
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < 2 * G; i += 256) s[i] = 0.f;
  __syncthreads();
  float a1[8], a2[8];
#pragma unroll
  for (int e = 0; e < 8; e++) { a1[e] = 0.f; a2[e] = 0.f; }
  const int r0 = blockIdx.x * rows_per_block, r1 = min(HW, r0 + rows_per_block);
  if (ty < nty) {
    // four rows in flight per thread, loaded unconditionally from a clamped row (a row past the end re-reads the last one and is
    // dropped): under `if (r < r1)` the loads of an iteration were serialised round trips
    for (int r = r0 + ty; r < r1; r += 4 * nty) {
      float v[4][8];
#pragma unroll
      for (int k = 0; k < 4; k++) ld8_nt(x + ((int64_t)b * HW + min(r + k * nty, r1 - 1)) * C + tx * 8, v[k]);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (r + k * nty >= r1) break;
#pragma unroll
        for (int e = 0; e < 8; e++) { a1[e] += v[k][e]; a2[e] += v[k][e] * v[k][e]; }
      }
    }
    const int cpg = C / G;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int g = (tx * 8 + e) / cpg;
      atomicAdd(&s[2 * g], a1[e]);
      atomicAdd(&s[2 * g + 1], a2[e]);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * G; i += 256) atomicAdd(sums + (int64_t)b * 2 * G + i, s[i]);
}

// GroupNorm apply: y = (x - mean_g) * rstd_g * gamma_c + beta_c, optional SiLU, bf16 out.
// pad != 0: y is a zero-bordered (B, H+2, W+2, C) tensor whose interior is written (operand of the implicit-GEMM convolution).
// grid = (pixel chunks, B); a thread owns one 8-channel group for all its pixels, so the group statistics, gamma and beta are
// folded into a per-channel scale / offset once and the loop is load - fma - (SiLU) - store with two pixels in flight.
template <typename TI>
__global__ __launch_bounds__(256) void gn_apply_kernel(const TI* __restrict__ x, const float* __restrict__ sums, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int H, int W, int C, int G, float eps, int silu, int pad, int pix_per_block,
                                                       bf16_t* __restrict__ y) {
  const int cg = C / 8, cpg = C / G, HW = H * W;
  const int t = threadIdx.x % cg, ty = threadIdx.x / cg, nty = 256 / cg;
  const int64_t b = blockIdx.y;
  if (ty >= nty) return;
  float sc[8], of[8];
  {
    float ga[8], be[8];
    ld8(gamma + t * 8, ga);
    ld8(beta + t * 8, be);
    const float inv_n = 1.f / ((float)HW * (float)cpg);
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int g = (t * 8 + e) / cpg;
      const float m = sums[(b * G + g) * 2] * inv_n;
      const float var = fmaxf(sums[(b * G + g) * 2 + 1] * inv_n - m * m, 0.f);
      sc[e] = rsqrtf(var + eps) * ga[e];
      of[e] = be[e] - m * sc[e];
    }
  }
  const int p0 = blockIdx.x * pix_per_block, p1 = min(HW, p0 + pix_per_block);
  for (int p = p0 + ty; p < p1; p += 2 * nty) {
    float v0[8], v1[8];
    const int q = p + nty;
    ld8(x + ((b * HW + p) * (int64_t)C) + t * 8, v0);
    if (q < p1) ld8(x + ((b * HW + q) * (int64_t)C) + t * 8, v1);
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int pp = k ? q : p;
      if (pp >= p1) break;
      float (&v)[8] = k ? v1 : v0;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const float o = fmaf(v[e], sc[e], of[e]);
        v[e] = silu ? silu_f(o) : o;
      }
      int64_t o = ((b * HW + pp) * (int64_t)C) + t * 8;
      if (pad) {
        const int yy = pp / W, xx = pp - yy * W;
        o = (((b * (H + 2) + yy + 1) * (W + 2) + xx + 1) * (int64_t)C) + t * 8;
      }
      st8(y + o, v);
    }
  }
}

// fp32 | bf16 NHWC (B,H,W,C) -> interior of a zero-bordered bf16 (B, uH+2, uW+2, C), uH = H << up (nearest x2 upsampling when up)
template <typename TI>
__global__ void pad_cast_kernel(const TI* __restrict__ x, bf16_t* __restrict__ y, int B, int H, int W, int C, int up) {
  const int cg = C / 8, Ho = H << up, Wo = W << up;
  const int64_t total = (int64_t)B * Ho * Wo * cg;
  for (int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(gid % cg);
    int64_t r = gid / cg;
    const int xo = (int)(r % Wo);
    r /= Wo;
    const int yo = (int)(r % Ho);
    const int64_t b = r / Ho;
    float v[8];
    ld8(x + (((b * H + (yo >> up)) * W + (xo >> up)) * (int64_t)C) + t * 8, v);
    st8(y + (((b * (Ho + 2) + yo + 1) * (Wo + 2) + xo + 1) * (int64_t)C) + t * 8, v);
  }
}

// row softmax of scale * x over the first `cols` columns (fp32, row pitch ld) -> bf16 (same pitch, padding columns = 0);
// one wave per row
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int rows, int cols, int ld, float scale) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (int64_t)row * ld;
  float m = -INFINITY;
  for (int c = lane; c < cols; c += 64) m = fmaxf(m, xr[c] * scale);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) s += __expf(xr[c] * scale - m);
  s = wave_sum(s);
  const float inv = 1.f / s;
  for (int c = lane; c < ld; c += 64) y[(int64_t)row * ld + c] = c < cols ? f2bf(__expf(xr[c] * scale - m) * inv) : (bf16_t)0;
}

}  // namespace

extern "C" int mmdit_vae_nchw_to_nhwc(const void* src, int src_dtype, int batch, int C, int H, int W, int C_padded, float scale, float shift,
                                      void* dst_bf16, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(src && dst_bf16 && batch > 0 && C > 0 && H > 0 && W > 0 && C_padded >= C && C_padded % 8 == 0);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(grid_cap((int64_t)batch * H * W * (C_padded / 8), 256));
  if (src_dtype == MMDIT_F32) hipLaunchKernelGGL((nchw_to_nhwc_kernel<float>), grid, dim3(256), 0, s, (const float*)src, (bf16_t*)dst_bf16, batch, C, H * W, C_padded, scale, shift);
  else if (src_dtype == MMDIT_BF16) hipLaunchKernelGGL((nchw_to_nhwc_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst_bf16, batch, C, H * W, C_padded, scale, shift);
  else return MMDIT_ERR_DTYPE;
  return mmdit_launch_status();
}

extern "C" int mmdit_vae_nhwc_to_nchw(const float* src, int batch, int C, int H, int W, int ld, float lo, float hi, float* dst, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(src && dst && batch > 0 && C > 0 && H > 0 && W > 0 && ld >= C);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_cap((int64_t)batch * C * H * W, 256)), dim3(256), 0, s, src, dst, batch, C, H * W, ld, lo, hi);
  return mmdit_launch_status();
}

extern "C" int mmdit_vae_im2col3x3(const void* src_bf16, int batch, int H, int W, int C, int mode, void* dst_bf16, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(src_bf16 && dst_bf16 && batch > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && mode >= 0 && mode <= 2);
  MMDIT_CHECK_ARG(mode != 1 || (H % 2 == 0 && W % 2 == 0));
  const int Ho = mode == 1 ? H / 2 : mode == 2 ? 2 * H : H, Wo = mode == 1 ? W / 2 : mode == 2 ? 2 * W : W;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(im2col3x3_kernel, dim3(grid_cap((int64_t)batch * Ho * Wo * 9 * (C / 8), 256)), dim3(256), 0, s, (const bf16_t*)src_bf16, (bf16_t*)dst_bf16, batch, H, W, C,
                     mode, Ho, Wo);
  return mmdit_launch_status();
}

extern "C" int mmdit_vae_groupnorm(const void* x, int x_dtype, const float* gamma, const float* beta, int batch, int H, int W, int C, int groups, float eps, int silu,
                                   float* sums_zeroed, void* y_bf16, int pad, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(x && gamma && beta && sums_zeroed && y_bf16 && batch > 0 && H > 0 && W > 0 && C % 8 == 0 && C / 8 <= 256 && groups > 0 && groups <= 64 && C % groups == 0);
  hipStream_t s = (hipStream_t)stream;
  const int HW = H * W;
  // statistics: ~2048 workgroups per launch, whatever the tensor size -- every workgroup ends with 2 * groups global atomics onto the
  // batch * 2 * groups sums (at 128 rows per workgroup a 512^2 x 128 x 16 tensor made 2 M atomics onto 1024 addresses: 1.2 TB/s)
  int rpb = 128;
  while ((int64_t)((HW + rpb - 1) / rpb) * batch > 2048 && rpb < HW) rpb *= 2;
  const int ppb = 128;   // pixels per apply workgroup
  dim3 g1((HW + rpb - 1) / rpb, batch), g2((HW + ppb - 1) / ppb, batch);
  if (x_dtype == MMDIT_F32) {
    hipLaunchKernelGGL((gn_stats_kernel<float>), g1, dim3(256), 0, s, (const float*)x, HW, C, groups, rpb, sums_zeroed);
    hipLaunchKernelGGL((gn_apply_kernel<float>), g2, dim3(256), 0, s, (const float*)x, sums_zeroed, gamma, beta, H, W, C, groups, eps, silu, pad, ppb, (bf16_t*)y_bf16);
  } else if (x_dtype == MMDIT_BF16) {
    hipLaunchKernelGGL((gn_stats_kernel<bf16_t>), g1, dim3(256), 0, s, (const bf16_t*)x, HW, C, groups, rpb, sums_zeroed);
    hipLaunchKernelGGL((gn_apply_kernel<bf16_t>), g2, dim3(256), 0, s, (const bf16_t*)x, sums_zeroed, gamma, beta, H, W, C, groups, eps, silu, pad, ppb, (bf16_t*)y_bf16);
  } else return MMDIT_ERR_DTYPE;
  return mmdit_launch_status();
}

extern "C" int mmdit_vae_pad_cast(const void* x, int x_dtype, int batch, int H, int W, int C, int upsample, void* y_bf16, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(x && y_bf16 && batch > 0 && H > 0 && W > 0 && C % 8 == 0 && (upsample == 0 || upsample == 1));
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(grid_cap((int64_t)batch * (H << upsample) * (W << upsample) * (C / 8), 256));
  if (x_dtype == MMDIT_F32) hipLaunchKernelGGL((pad_cast_kernel<float>), grid, dim3(256), 0, s, (const float*)x, (bf16_t*)y_bf16, batch, H, W, C, upsample);
  else if (x_dtype == MMDIT_BF16) hipLaunchKernelGGL((pad_cast_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y_bf16, batch, H, W, C, upsample);
  else return MMDIT_ERR_DTYPE;
  return mmdit_launch_status();
}

extern "C" int mmdit_vae_softmax_rows(const float* x, int rows, int cols, int ld, float scale, void* y_bf16, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(x && y_bf16 && rows > 0 && cols > 0 && ld >= cols);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, (bf16_t*)y_bf16, rows, cols, ld, scale);
  return mmdit_launch_status();
}
