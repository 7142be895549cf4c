// HBM-bound row / elementwise kernels of the MMDiT path for gfx950.
// All loads/stores are 8-16 B per lane; reductions over a row live in one 64-lane wave;
// per-batch / per-column gradient sums are accumulated in registers over a row chunk and
// flushed with fp32 atomics (caller zero-initialises the destination).
#include "common.h"
#include <float.h>

namespace {

constexpr float LN_EPS = 1e-5f;           // nn.LayerNorm default (Norm.py:10)
constexpr float RMS_EPS = FLT_EPSILON;    // nn.RMSNorm(eps=None) on fp32 input

// -------------------------------------------------------------------------------------------
// adaLN forward: one wave per row, the row is kept in registers (NIT float4 per lane).
// -------------------------------------------------------------------------------------------
// RES: the row entering the norm is x + gate[b] * acc -- the gated residual update that PRODUCES the normed tensor
// (X_out = acc * gate + X_in, Transformer_Block_Dual.py:64-76) is formed here, where the row is in registers anyway, instead of
// in the fp32 epilogue of the projection GEMM (which then only writes acc in the activation dtype); the updated residual
// stream row is written to xo (it is the next residual input and the x of the backward pass).
// round to bf16 and back: a fused producer quantises exactly what the unfused path would have stored as bf16 and quantised afterwards
__device__ __forceinline__ float bf16_round(float x) { return __builtin_bit_cast(float, pack_bf2(x, 0.f) << 16); }

// MX (TO = unsigned char, d % 32 == 0): the normalised row leaves as e4m3 codes + E8M0 block scales (a 32-block is the 4 values of 8
// adjacent lanes of one iteration), bit-identical to the bf16 output followed by mmdit_mxfp8_quantize.
template <int NITX, typename TO, typename TA, bool RES, bool MX = false>
__device__ __forceinline__ void ln_mod_fwd_body(int block, const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                int64_t ld_mod, int rows, int d, int rpb, TO* __restrict__ out,
                                                float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                const TA* __restrict__ acc, const float* __restrict__ gate, int64_t ld_gate, float* __restrict__ xo,
                                                unsigned char* __restrict__ mx_scales) {
  constexpr int NIT = NITX < 0 ? -NITX : NITX;      // NITX < 0: d == 256 * NIT exactly, no lane of any iteration is out of range
  constexpr bool EXACT = NITX < 0;
  const int lane = threadIdx.x & 63, row = block * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = d >> 2;
  const int b = row / rpb;
  const float* xr = x + (int64_t)row * d;
  float v[NIT][4];
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    int ch = lane + 64 * it;
    if (EXACT || ch < nch) {
      ld4(xr + ch * 4, v[it]);
      if constexpr (RES) {
        float av[4], g[4];
        ld4(acc + (int64_t)row * d + ch * 4, av);
        ld4(gate + (int64_t)b * ld_gate + ch * 4, g);
#pragma unroll
        for (int e = 0; e < 4; e++) v[it][e] = v[it][e] + g[e] * av[e];
        __builtin_nontemporal_store(f32x4{v[it][0], v[it][1], v[it][2], v[it][3]}, (f32x4*)(xo + (int64_t)row * d + ch * 4));   // next read a GEMM, an attention and a GEMM later
      }
      s += v[it][0] + v[it][1] + v[it][2] + v[it][3];
    }
    else { v[it][0] = v[it][1] = v[it][2] = v[it][3] = 0.f; }
  }
  const float mean = wave_sum(s) / d;
  float q = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    int ch = lane + 64 * it;
    if (EXACT || ch < nch) {
#pragma unroll
      for (int e = 0; e < 4; e++) { float c = v[it][e] - mean; q += c * c; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / d + LN_EPS);
  if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
  const float* sc = scale + (int64_t)b * ld_mod;
  const float* sh = shift + (int64_t)b * ld_mod;
  TO* orow = out + (int64_t)row * d;
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    int ch = lane + 64 * it;
    if (EXACT || ch < nch) {
      float a[4], h[4], o[4];
      ld4(sc + ch * 4, a); ld4(sh + ch * 4, h);
#pragma unroll
      for (int e = 0; e < 4; e++) o[e] = (v[it][e] - mean) * rstd * (1.f + a[e]) + h[e];
      if constexpr (MX) {
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = bf16_round(o[e]);
        float amax = fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3])));
        amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
        amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
        amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
        float inv;
        const int ex = mx_exponent(amax, inv);
        *(unsigned*)((unsigned char*)out + (int64_t)row * d + ch * 4) = mx_pack4(o, inv);
        if ((lane & 7) == 0) mx_scales[mx_scale_index(row, ch >> 3, rows)] = (unsigned char)(ex + 127);
      } else {
        st4(orow + ch * 4, o);
      }
    }
  }
}

template <int NIT, typename TO, typename TA, bool RES, bool MX = false>
__global__ __launch_bounds__(256) void ln_mod_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                         int64_t ld_mod, int rows, int d, int rpb, TO* __restrict__ out,
                                                         float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                         const TA* __restrict__ acc, const float* __restrict__ gate, int64_t ld_gate, float* __restrict__ xo,
                                                         unsigned char* __restrict__ mx_scales = nullptr) {
  ln_mod_fwd_body<NIT, TO, TA, RES, MX>((int)blockIdx.x, x, scale, shift, ld_mod, rows, d, rpb, out, mean_o, rstd_o, acc, gate, ld_gate, xo, mx_scales);
}

// Two independent adaLN problems of the same width in ONE launch (the image and the text stream of a block): these kernels run 20-45 us
// on 75-180 MB, of which ~8 us is ramp-up and tail -- one launch over both streams instead of two saves ~5 us per pair
// (tools/row_bench.py --cold).  Blocks [0, nblk0) work on problem 0, the rest on problem 1.
struct LnFwdProb {
  const float* x; const float* scale; const float* shift; int64_t ld_mod; int rows, rpb; void* out; float* mean; float* rstd;
  const void* acc; const float* gate; int64_t ld_gate; float* xo;
};
template <int NIT, typename TO, typename TA, bool RES>
__global__ __launch_bounds__(256) void ln_mod_fwd_pair_kernel(LnFwdProb p0, LnFwdProb p1, int nblk0, int d) {
  const bool first = (int)blockIdx.x < nblk0;     // (workgroup-uniform)
  const LnFwdProb& p = first ? p0 : p1;
  ln_mod_fwd_body<NIT, TO, TA, RES, false>(first ? (int)blockIdx.x : (int)blockIdx.x - nblk0, p.x, p.scale, p.shift, p.ld_mod, p.rows, d, p.rpb, (TO*)p.out, p.mean, p.rstd,
                                           (const TA*)p.acc, p.gate, p.ld_gate, p.xo, nullptr);
}

// out = x + gate[b] * acc (the gated residual update on its own: used where no norm consumes the result)
template <typename TA>
__global__ __launch_bounds__(256) void gate_residual_fwd_kernel(const float* __restrict__ x, const TA* __restrict__ acc, const float* __restrict__ gate, int64_t ld_gate,
                                                                int rows, int d, int rpb, float* __restrict__ out) {
  const int nch = d >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)rows * nch; i += (int64_t)gridDim.x * 256) {
    const int row = (int)(i / nch), ch = (int)(i - (int64_t)row * nch);
    float xv[4], av[4], g[4];
    ld4(x + (int64_t)row * d + ch * 4, xv);
    ld4(acc + (int64_t)row * d + ch * 4, av);
    ld4(gate + (int64_t)(row / rpb) * ld_gate + ch * 4, g);
#pragma unroll
    for (int e = 0; e < 4; e++) xv[e] = xv[e] + g[e] * av[e];
    st4(out + (int64_t)row * d + ch * 4, xv);
  }
}

// adaLN backward: block = (batch b, chunk of RCH rows); wave w takes rows w, w+4, ...
// GATED: the kernel also runs the backward of the gated residual update that CONSUMES dx in the backward order
// (X_out = acc * gate[b] + X_in, Transformer_Block_Dual.py:64-76: dx is d(X_out)): dacc = dx * gate[b] for the producing GEMM's
// dgrad / wgrad, dgate[b] += sum_rows dx * acc, dbias[b] += sum_rows dacc (per-batch partial rows of the projection's bias
// gradient) -- dx is in registers here, so the separate pass over it (mmdit_gate_residual_bwd) disappears.
#ifndef MMDIT_LN_BWD_RCH
#define MMDIT_LN_BWD_RCH 16
#endif
constexpr int LN_BWD_RCH = MMDIT_LN_BWD_RCH;   // rows per block: B * rows_per_batch / 16 blocks keep every CU busy with several waves
template <int NITX, typename TG, typename TA, bool GATED>
__device__ __forceinline__ void ln_mod_bwd_body(int block, const TG* __restrict__ dout, const float* __restrict__ x, const float* __restrict__ mean_i,
                                                const float* __restrict__ rstd_i, const float* __restrict__ scale, int64_t ld_mod,
                                                const float* __restrict__ dres, int d, int rpb, int nchunk,
                                                float* __restrict__ dx, float* __restrict__ dscale, float* __restrict__ dshift, int64_t ld_dmod,
                                                const TA* __restrict__ acc, const float* __restrict__ gate, int64_t ld_gate, TA* __restrict__ dacc,
                                                float* __restrict__ dgate, int64_t ld_dgate, float* __restrict__ dbias, int64_t ld_dbias, int rch = LN_BWD_RCH) {
  constexpr int NIT = NITX < 0 ? -NITX : NITX;      // NITX < 0: d == 256 * NIT exactly, no lane of any iteration is out of range
  constexpr bool EXACT = NITX < 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = block / nchunk, chunk = block % nchunk;
  const int nch = d >> 2;
  const float* sc = scale + (int64_t)b * ld_mod;
  // the sample's (1 + scale) and gate rows live in LDS and are re-read per row (2 x NIT ds_read_b128): as loop-invariant registers they
  // cost 8 * NIT VGPRs, which put the gated d = 768 kernel at 150 VGPRs = 3 waves per SIMD; without them it fits 4
  __shared__ float s_a1[NIT * 256], s_gt[GATED ? NIT * 256 : 4];
  for (int c = threadIdx.x; c < NIT * 256; c += 256) {
    s_a1[c] = c < d ? sc[c] + 1.f : 0.f;
    if constexpr (GATED) s_gt[c] = c < d ? gate[(int64_t)b * ld_gate + c] : 0.f;
  }
  __syncthreads();
  float ds[NIT][4], dh[NIT][4];
  float sg[GATED ? NIT : 1][4], sb[GATED ? NIT : 1][4];
#pragma unroll
  for (int it = 0; it < NIT; it++) {
#pragma unroll
    for (int e = 0; e < 4; e++) { ds[it][e] = 0.f; dh[it][e] = 0.f; }
    if constexpr (GATED) {
#pragma unroll
      for (int e = 0; e < 4; e++) { sg[it][e] = 0.f; sb[it][e] = 0.f; }
    }
  }
  const int rend = min(rpb, (chunk + 1) * rch);
  for (int rl = chunk * rch + wave; rl < rend; rl += 4) {
    const int64_t row = (int64_t)b * rpb + rl;
    const float mean = mean_i[row], rstd = rstd_i[row];
    float g[NIT][4], xh[NIT][4];
    float c1 = 0.f, c2 = 0.f;
    asm volatile("" ::: "memory");      // (keeps the LDS reads of s_a1 / s_gt inside the row loop)
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      int ch = lane + 64 * it;
      if (EXACT || ch < nch) {
        float dy[4], xv[4], a1[4];
        ld4_nt(dout + row * d + ch * 4, dy);      // (dy, the saved residual row and acc are at their last use: streaming loads)
        ld4_nt(x + row * d + ch * 4, xv);
        ld4(s_a1 + ch * 4, a1);
#pragma unroll
        for (int e = 0; e < 4; e++) {
          xh[it][e] = (xv[e] - mean) * rstd;
          g[it][e] = dy[e] * a1[e];
          c1 += g[it][e]; c2 += g[it][e] * xh[it][e];
          ds[it][e] += dy[e] * xh[it][e]; dh[it][e] += dy[e];
        }
      }
    }
    c1 = wave_sum(c1) / d; c2 = wave_sum(c2) / d;
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      int ch = lane + 64 * it;
      if (EXACT || ch < nch) {
        float o[4];
        if (dres) ld4(dres + row * d + ch * 4, o); else { o[0] = o[1] = o[2] = o[3] = 0.f; }
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] += rstd * (g[it][e] - c1 - xh[it][e] * c2);
        st4(dx + row * d + ch * 4, o);
        if constexpr (GATED) {
          float av[4], da[4], gt[4];
          ld4_nt(acc + row * d + ch * 4, av);
          ld4(s_gt + ch * 4, gt);
#pragma unroll
          for (int e = 0; e < 4; e++) { da[e] = o[e] * gt[e]; sg[it][e] += o[e] * av[e]; sb[it][e] += da[e]; }
          st4(dacc + row * d + ch * 4, da);
        }
      }
    }
  }
  if constexpr (NIT <= 6) {
    // combine the 4 waves in LDS (one quantity at a time: <= 24 KB), then one atomic per column per block
    __shared__ float red[4][NIT * 256];
    auto flush = [&](const float (&v)[NIT][4], float* dst) {
#pragma unroll
      for (int it = 0; it < NIT; it++)
#pragma unroll
        for (int e = 0; e < 4; e++) red[wave][(it * 64 + lane) * 4 + e] = v[it][e];
      __syncthreads();
      for (int c = threadIdx.x; c < d; c += 256) atomicAdd(dst + c, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
      __syncthreads();
    };
    flush(ds, dscale + (int64_t)b * ld_dmod);
    flush(dh, dshift + (int64_t)b * ld_dmod);
    if constexpr (GATED) {
      flush(sg, dgate + (int64_t)b * ld_dgate);
      if (dbias) flush(sb, dbias + (int64_t)b * ld_dbias);   // uniform
    }
  } else {
    auto flush = [&](const float (&v)[NIT][4], float* dst) {
#pragma unroll
      for (int it = 0; it < NIT; it++) {
        int ch = lane + 64 * it;
        if (EXACT || ch < nch) {
#pragma unroll
          for (int e = 0; e < 4; e++) atomicAdd(dst + ch * 4 + e, v[it][e]);
        }
      }
    };
    flush(ds, dscale + (int64_t)b * ld_dmod);
    flush(dh, dshift + (int64_t)b * ld_dmod);
    if constexpr (GATED) {
      flush(sg, dgate + (int64_t)b * ld_dgate);
      if (dbias) flush(sb, dbias + (int64_t)b * ld_dbias);
    }
  }
}

template <int NIT, typename TG, typename TA, bool GATED>
__global__ __launch_bounds__(256) void ln_mod_bwd_kernel(const TG* __restrict__ dout, const float* __restrict__ x, const float* __restrict__ mean_i,
                                                         const float* __restrict__ rstd_i, const float* __restrict__ scale, int64_t ld_mod,
                                                         const float* __restrict__ dres, int d, int rpb, int nchunk,
                                                         float* __restrict__ dx, float* __restrict__ dscale, float* __restrict__ dshift, int64_t ld_dmod,
                                                         const TA* __restrict__ acc, const float* __restrict__ gate, int64_t ld_gate, TA* __restrict__ dacc,
                                                         float* __restrict__ dgate, int64_t ld_dgate, float* __restrict__ dbias, int64_t ld_dbias) {
  ln_mod_bwd_body<NIT, TG, TA, GATED>((int)blockIdx.x, dout, x, mean_i, rstd_i, scale, ld_mod, dres, d, rpb, nchunk, dx, dscale, dshift, ld_dmod, acc, gate, ld_gate, dacc,
                                      dgate, ld_dgate, dbias, ld_dbias);
}

// two adaLN backward problems of the same width in one launch (see ln_mod_fwd_pair_kernel)
struct LnBwdProb {
  const void* dout; const float* x; const float* mean; const float* rstd; const float* scale; int64_t ld_mod; const float* dres; int rpb, nchunk;   // nchunk chunks of rch rows per sample
  float* dx; float* dscale; float* dshift; int64_t ld_dmod;
  const void* acc; const float* gate; int64_t ld_gate; void* dacc; float* dgate; int64_t ld_dgate; float* dbias; int64_t ld_dbias;
};
template <int NIT, typename TG, typename TA, bool GATED>
__global__ __launch_bounds__(256) void ln_mod_bwd_pair_kernel(LnBwdProb p0, LnBwdProb p1, int nblk0, int d, int rch) {
  const bool first = (int)blockIdx.x < nblk0;     // (workgroup-uniform)
  const LnBwdProb& p = first ? p0 : p1;
  ln_mod_bwd_body<NIT, TG, TA, GATED>(first ? (int)blockIdx.x : (int)blockIdx.x - nblk0, (const TG*)p.dout, p.x, p.mean, p.rstd, p.scale, p.ld_mod, p.dres, d, p.rpb, p.nchunk,
                                      p.dx, p.dscale, p.dshift, p.ld_dmod, (const TA*)p.acc, p.gate, p.ld_gate, (TA*)p.dacc, p.dgate, p.ld_dgate, p.dbias, p.ld_dbias, rch);
}

// -------------------------------------------------------------------------------------------
// text RMSNorm (one half of the 154 tokens per launch): src row = (m/cnt)*tokens + off + m%cnt
// -------------------------------------------------------------------------------------------
template <int NITX, typename TI, typename TO>
__global__ __launch_bounds__(256) void text_rms_fwd_kernel(const TI* __restrict__ x, const float* __restrict__ w, const float* __restrict__ sp,
                                                           int rows, int cnt, int tokens, int off, int d, TO* __restrict__ out) {
  constexpr int NIT = NITX < 0 ? -NITX : NITX;      // NITX < 0: d == 256 * NIT exactly, no lane of any iteration is out of range
  constexpr bool EXACT = NITX < 0;
  const int lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= rows) return;
  const int nch = d >> 2;
  const TI* xr = x + ((int64_t)(m / cnt) * tokens + off + m % cnt) * d;
  float v[NIT][4];
  float q = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    int ch = lane + 64 * it;
    if (EXACT || ch < nch) { ld4(xr + ch * 4, v[it]); q += v[it][0] * v[it][0] + v[it][1] * v[it][1] + v[it][2] * v[it][2] + v[it][3] * v[it][3]; }
  }
  const float rinv = rsqrtf(wave_sum(q) / d + RMS_EPS);
  const float s = sp[0];
  TO* orow = out + (int64_t)m * d;
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    int ch = lane + 64 * it;
    if (EXACT || ch < nch) {
      float wv[4], o[4];
      ld4(w + ch * 4, wv);
#pragma unroll
      for (int e = 0; e < 4; e++) o[e] = s * (v[it][e] * rinv * wv[e]);
      st4(orow + ch * 4, o);
    }
  }
}

// Both halves of the 154 tokens in one launch (blockIdx.y = half); a workgroup is 8 waves x 4 rows.  Every workgroup ends with d
// global atomics on the SAME d addresses (the weight gradient is not per sample), so the waves are first combined in LDS
// (ds_add_f32) and the workgroups are as fat as the register budget allows: 154 atomics per address and launch instead of 2 x 308.
#ifndef TRB_ROWS
#define TRB_ROWS 64
#endif
struct TextRmsBwdHalf { const void* dout; const float* w; const float* sp; int rows, cnt, off; float* dw; float* dsp; };
template <int NITX, typename TI, typename TG>
__global__ __launch_bounds__(512) void text_rms_bwd_kernel(TextRmsBwdHalf h0, TextRmsBwdHalf h1, const TI* __restrict__ x, int tokens, int d) {
  constexpr int NIT = NITX < 0 ? -NITX : NITX;      // NITX < 0: d == 256 * NIT exactly, no lane of any iteration is out of range
  constexpr bool EXACT = NITX < 0;
  const TextRmsBwdHalf& hp = blockIdx.y == 0 ? h0 : h1;     // (workgroup-uniform)
  const TG* __restrict__ dout = (const TG*)hp.dout;
  const float* __restrict__ w = hp.w;
  const float* __restrict__ sp = hp.sp;
  const int rows = hp.rows, cnt = hp.cnt, off = hp.off;
  float* __restrict__ dw = hp.dw;
  float* __restrict__ dsp = hp.dsp;
  __shared__ float red[NIT * 256];
  for (int c = threadIdx.x; c < NIT * 256; c += 512) red[c] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = d >> 2;
  const float s = sp[0];
  float accw[NIT][4], wv[NIT][4];
  float accs = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    int ch = lane + 64 * it;
    if (EXACT || ch < nch) ld4(w + ch * 4, wv[it]); else { wv[it][0] = wv[it][1] = wv[it][2] = wv[it][3] = 0.f; }
#pragma unroll
    for (int e = 0; e < 4; e++) accw[it][e] = 0.f;
  }
  for (int m = blockIdx.x * TRB_ROWS + wave; m < min(rows, (int)(blockIdx.x + 1) * TRB_ROWS); m += 8) {   // TRB_ROWS rows per workgroup, 8 waves
    const TI* xr = x + ((int64_t)(m / cnt) * tokens + off + m % cnt) * d;
    float v[NIT][4];
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      int ch = lane + 64 * it;
      if (EXACT || ch < nch) { ld4(xr + ch * 4, v[it]); q += v[it][0] * v[it][0] + v[it][1] * v[it][1] + v[it][2] * v[it][2] + v[it][3] * v[it][3]; }
    }
    const float rinv = rsqrtf(wave_sum(q) / d + RMS_EPS);
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      int ch = lane + 64 * it;
      if (EXACT || ch < nch) {
        float dy[4];
        ld4(dout + (int64_t)m * d + ch * 4, dy);
#pragma unroll
        for (int e = 0; e < 4; e++) { float xn = v[it][e] * rinv; accw[it][e] += dy[e] * s * xn; accs += dy[e] * xn * wv[it][e]; }
      }
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    const int ch = lane + 64 * it;
    if (EXACT || ch < nch) {
#pragma unroll
      for (int e = 0; e < 4; e++) atomicAdd(&red[ch * 4 + e], accw[it][e]);
    }
  }
  __syncthreads();
#ifndef TRB_NO_GLOBAL
  for (int c = threadIdx.x; c < d; c += 512) atomicAdd(dw + c, red[c]);
#else
  if (red[threadIdx.x] == 123.456f) dw[0] = 1.f;
#endif
  accs = wave_sum(accs);
#ifndef TRB_NO_DSP
  if (lane == 0) atomicAdd(dsp, accs);
#else
  if (accs == 123.456f) dsp[0] = 1.f;
#endif
}

// -------------------------------------------------------------------------------------------
// per-head QK RMSNorm + RoPE2d + head split.  8 lanes own one 64-wide head vector (8 elements each).
// flat index over [row][part q/k/v][head][chunk] == memory order of a qkv row.
// -------------------------------------------------------------------------------------------
__device__ __forceinline__ float group8_sum(float v) {
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
  return v;
}

// Forward: same thread mapping as the backward kernel below -- a workgroup of 24 * heads threads is one qkv row (thread = (part,
// head, 8-element chunk), no per-element index divisions, norm weights in registers), rows blockIdx.x, + gridDim.x, ... two in
// flight; with the grid a multiple of the tokens per sample all rows of a workgroup are the same token and the RoPE factors are
// loaded once.
// (bid, rstride) = this workgroup's index and the number of workgroups of ITS problem (the pair kernel runs two problems in one grid)
template <typename TI>
__device__ __forceinline__ void qk_norm_rope_fwd_body(int bid, int rstride, const TI* __restrict__ qkv, const float* __restrict__ wq, const float* __restrict__ wk,
                                                      const float* __restrict__ rcos, const float* __restrict__ rsin,
                                                      int rows, int tokens, int heads, int s_total, int tok0,
                                                      bf16_t* __restrict__ Q, bf16_t* __restrict__ K, bf16_t* __restrict__ V) {
  const int hc = threadIdx.x % (8 * heads), part = threadIdx.x / (8 * heads), head = hc >> 3, chunk = hc & 7;
  bf16_t* obase = part == 0 ? Q : part == 1 ? K : V;
  float w[8];
#pragma unroll
  for (int e = 0; e < 8; e++) w[e] = 0.f;
  if (part < 2) ld8((part == 0 ? wq : wk) + chunk * 8, w);
  const bool same_token = rcos && rstride % tokens == 0;
  float cs[2][8], sn[2][8];
  if (same_token && part < 2) {
    const int n = bid % tokens;
    ld8(rcos + (int64_t)n * 64 + chunk * 8, cs[0]);
    ld8(rsin + (int64_t)n * 64 + chunk * 8, sn[0]);
#pragma unroll
    for (int e = 0; e < 8; e++) { cs[1][e] = cs[0][e]; sn[1][e] = sn[0][e]; }
  }
  for (int row0 = bid; row0 < rows; row0 += 2 * rstride) {
    float x[2][8];
    // both rows' loads are issued unconditionally (a row past the end re-reads the last row and is dropped below): inside
    // `if (row < rows)` each load sat in its own branch and was waited for before the next one was issued
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int row = min(row0 + k * rstride, rows - 1);
      ld8(qkv + (((int64_t)row * 3 + part) * heads + head) * 64 + chunk * 8, x[k]);
    }
    if (part < 2 && rcos && !same_token) {
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int n = min(row0 + k * rstride, rows - 1) % tokens;
        ld8(rcos + (int64_t)n * 64 + chunk * 8, cs[k]);
        ld8(rsin + (int64_t)n * 64 + chunk * 8, sn[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int row = row0 + k * rstride;
      if (row >= rows) break;
      const int n = row % tokens, b = row / tokens;
      if (part < 2) {
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) ss += x[k][e] * x[k][e];
        const float rinv = rsqrtf(group8_sum(ss) * (1.f / 64.f) + RMS_EPS);
#pragma unroll
        for (int e = 0; e < 8; e++) x[k][e] = x[k][e] * rinv * w[e];
        if (rcos) {
#pragma unroll
          for (int p = 0; p < 4; p++) {
            float a = x[k][2 * p], bb = x[k][2 * p + 1];
            x[k][2 * p] = a * cs[k][2 * p] - bb * sn[k][2 * p];
            x[k][2 * p + 1] = bb * cs[k][2 * p + 1] + a * sn[k][2 * p + 1];
          }
        }
      }
      st8(obase + (((int64_t)b * heads + head) * s_total + tok0 + n) * 64 + chunk * 8, x[k]);
    }
  }
}
template <typename TI>
__global__ __launch_bounds__(1024) void qk_norm_rope_fwd_kernel(const TI* __restrict__ qkv, const float* __restrict__ wq, const float* __restrict__ wk,
                                                                const float* __restrict__ rcos, const float* __restrict__ rsin,
                                                                int rows, int tokens, int heads, int s_total, int tok0,
                                                                bf16_t* __restrict__ Q, bf16_t* __restrict__ K, bf16_t* __restrict__ V) {
  qk_norm_rope_fwd_body<TI>((int)blockIdx.x, (int)gridDim.x, qkv, wq, wk, rcos, rsin, rows, tokens, heads, s_total, tok0, Q, K, V);
}
// the image and the text rows of a block in one launch (workgroups [0, g0) = problem 0): same Q / K / V, different token ranges
struct QkProb {
  const void* qkv; const float* wq; const float* wk; const float* rcos; const float* rsin; int rows, tokens, tok0;
  const void* dQ; const void* dK; const void* dV; void* dqkv; float* dwq; float* dwk;      // (backward only)
};
template <typename TI>
__global__ __launch_bounds__(1024) void qk_norm_rope_fwd_pair_kernel(QkProb p0, QkProb p1, int g0, int heads, int s_total,
                                                                     bf16_t* __restrict__ Q, bf16_t* __restrict__ K, bf16_t* __restrict__ V) {
  const bool first = (int)blockIdx.x < g0;     // (workgroup-uniform)
  const QkProb& p = first ? p0 : p1;
  qk_norm_rope_fwd_body<TI>(first ? (int)blockIdx.x : (int)blockIdx.x - g0, first ? g0 : (int)gridDim.x - g0, (const TI*)p.qkv, p.wq, p.wk, p.rcos, p.rsin,
                            p.rows, p.tokens, heads, s_total, p.tok0, Q, K, V);
}

// Backward: a workgroup has 24 * heads threads = one qkv row (thread = (part, head, 8-element chunk): no per-element index
// divisions, the q/k norm weights stay in registers) and walks rows blockIdx.x, + gridDim.x, ... two at a time; the grid is
// small (2 workgroups per CU) because every workgroup ends with 128 global atomics on the same two cache lines.
// FAST: every problem of the launch either has no RoPE or walks ONE token per row lane (lanes % tokens == 0): a single copy of the RoPE
// factors, loaded once -- 16 registers fewer than the general path, which kept the kernel at the 128-VGPR cap with spills
template <bool FAST, typename TG, typename TI, typename TO>
__device__ __forceinline__ void qk_norm_rope_bwd_body(int bid, int rstride, const TG* __restrict__ dQ, const TG* __restrict__ dK, const TG* __restrict__ dV,
                                                      const TI* __restrict__ qkv, const float* __restrict__ wq, const float* __restrict__ wk,
                                                      const float* __restrict__ rcos, const float* __restrict__ rsin,
                                                      int rows, int tokens, int heads, int s_total, int tok0,
                                                      TO* __restrict__ dqkv, float* __restrict__ dwq, float* __restrict__ dwk) {
  __shared__ float sdw[2][64];
  for (int i = threadIdx.x; i < 128; i += blockDim.x) sdw[i >> 6][i & 63] = 0.f;
  __syncthreads();
  // a workgroup is RL row lanes of 24 * heads threads: RL rows per iteration share ONE set of 128 global atomics at the end
  const int per_row = 24 * heads, rlane = threadIdx.x / per_row, t = threadIdx.x - rlane * per_row;
  { const int RL = blockDim.x / per_row; bid = bid * RL + rlane; rstride *= RL; }
  const int hc = t % (8 * heads), part = t / (8 * heads), head = hc >> 3, chunk = hc & 7;
  const TG* gbase = part == 0 ? dQ : part == 1 ? dK : dV;
  float w[8], aw[8];
#pragma unroll
  for (int e = 0; e < 8; e++) { w[e] = 0.f; aw[e] = 0.f; }
  if (part < 2) ld8((part == 0 ? wq : wk) + chunk * 8, w);
  // grid a multiple of the tokens per sample: every row of this workgroup is the same token, its RoPE factors are loaded once
  const bool same_token = FAST ? rcos != nullptr : (rcos && rstride % tokens == 0);
  float cs[FAST ? 1 : 2][8], sn[FAST ? 1 : 2][8];
  if (same_token && part < 2) {
    const int n = bid % tokens;
    ld8(rcos + (int64_t)n * 64 + chunk * 8, cs[0]);
    ld8(rsin + (int64_t)n * 64 + chunk * 8, sn[0]);
    if constexpr (!FAST) {
#pragma unroll
      for (int e = 0; e < 8; e++) { cs[1][e] = cs[0][e]; sn[1][e] = sn[0][e]; }
    }
  }
  for (int row0 = bid; row0 < rows; row0 += 2 * rstride) {
    float dz[2][8], x[2][8];
    // loads of both rows issued back to back, unconditionally (a row past the end re-reads the last row and is dropped below): under
    // `if (row < rows)` the two rows "in flight" were two serialized round trips
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int row = min(row0 + k * rstride, rows - 1);
      const int n = row % tokens, b = row / tokens;
      ld8_nt(gbase + (((int64_t)b * heads + head) * s_total + tok0 + n) * 64 + chunk * 8, dz[k]);      // (single use; the saved qkv below: last use)
    }
    if (part < 2) {
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int row = min(row0 + k * rstride, rows - 1);
        ld8_nt(qkv + (((int64_t)row * 3 + part) * heads + head) * 64 + chunk * 8, x[k]);
        if constexpr (!FAST) {
          if (rcos && !same_token) {
            const int n = row % tokens;
            ld8(rcos + (int64_t)n * 64 + chunk * 8, cs[k]);
            ld8(rsin + (int64_t)n * 64 + chunk * 8, sn[k]);
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int row = row0 + k * rstride;
      if (row >= rows) break;
      if (part < 2) {
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) ss += x[k][e] * x[k][e];
        const float rinv = rsqrtf(group8_sum(ss) * (1.f / 64.f) + RMS_EPS);
        if (rcos) {
#pragma unroll
          for (int p = 0; p < 4; p++) {
            float da = dz[k][2 * p], db = dz[k][2 * p + 1];
            dz[k][2 * p] = da * cs[FAST ? 0 : k][2 * p] + db * sn[FAST ? 0 : k][2 * p + 1];
            dz[k][2 * p + 1] = db * cs[FAST ? 0 : k][2 * p + 1] - da * sn[FAST ? 0 : k][2 * p];
          }
        }
        float dot = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) {
          float xh = x[k][e] * rinv;
          aw[e] += dz[k][e] * xh;
          dz[k][e] *= w[e];                 // d(xhat)
          dot += dz[k][e] * xh;
          x[k][e] = xh;
        }
        dot = group8_sum(dot) * (1.f / 64.f);
#pragma unroll
        for (int e = 0; e < 8; e++) dz[k][e] = rinv * (dz[k][e] - x[k][e] * dot);
      }
      st8(dqkv + (((int64_t)row * 3 + part) * heads + head) * 64 + chunk * 8, dz[k]);
    }
  }
  // weight gradients: sum over rows (done) and heads: LDS atomics per (part, column), then 128 global atomics per workgroup
  if (part < 2) {
#pragma unroll
    for (int e = 0; e < 8; e++) atomicAdd(&sdw[part][chunk * 8 + e], aw[e]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 128; i += blockDim.x) atomicAdd((i < 64 ? dwq : dwk) + (i & 63), sdw[i >> 6][i & 63]);
}
template <bool FAST, typename TG, typename TI, typename TO>
__global__ __launch_bounds__(1024) void qk_norm_rope_bwd_kernel(const TG* __restrict__ dQ, const TG* __restrict__ dK, const TG* __restrict__ dV,
                                                                const TI* __restrict__ qkv, const float* __restrict__ wq, const float* __restrict__ wk,
                                                                const float* __restrict__ rcos, const float* __restrict__ rsin,
                                                                int rows, int tokens, int heads, int s_total, int tok0,
                                                                TO* __restrict__ dqkv, float* __restrict__ dwq, float* __restrict__ dwk) {
  qk_norm_rope_bwd_body<FAST, TG, TI, TO>((int)blockIdx.x, (int)gridDim.x, dQ, dK, dV, qkv, wq, wk, rcos, rsin, rows, tokens, heads, s_total, tok0, dqkv, dwq, dwk);
}
template <bool FAST, typename TG, typename TI, typename TO>
__global__ __launch_bounds__(1024) void qk_norm_rope_bwd_pair_kernel(QkProb p0, QkProb p1, int g0, int heads, int s_total) {
  const bool first = (int)blockIdx.x < g0;     // (workgroup-uniform)
  const QkProb& p = first ? p0 : p1;
  qk_norm_rope_bwd_body<FAST, TG, TI, TO>(first ? (int)blockIdx.x : (int)blockIdx.x - g0, first ? g0 : (int)gridDim.x - g0, (const TG*)p.dQ, (const TG*)p.dK, (const TG*)p.dV,
                                    (const TI*)p.qkv, p.wq, p.wk, p.rcos, p.rsin, p.rows, p.tokens, heads, s_total, p.tok0, (TO*)p.dqkv, p.dwq, p.dwk);
}

// -------------------------------------------------------------------------------------------
// column-owner elementwise kernels: block = 2 row-lanes x 128 column threads (8 columns each);
// grid = (column slabs of 1024, row chunks of CO_RCH rows).
// -------------------------------------------------------------------------------------------
constexpr int CO_RCH = 64;

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

// MX: h leaves as e4m3 codes + E8M0 block scales (a 32-block is the 8 values of 4 adjacent column lanes), hidden % 32 == 0
template <typename T, bool GELU, bool MX = false>
__global__ __launch_bounds__(256) void mlp_act_fwd_kernel(const T* __restrict__ gu, T* __restrict__ h, int rows, int hidden, unsigned char* __restrict__ mx_scales = nullptr) {
  // 256 columns x CO_RCH rows per workgroup (32 column groups of 8 x 8 row lanes), two rows in flight per thread
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c = blockIdx.x * 256 + tx * 8;
  if (c >= hidden) return;
  const int64_t ldi = GELU ? hidden : 2 * (int64_t)hidden;
  const int rend = min(rows, (int)(blockIdx.y + 1) * CO_RCH);
  for (int r0 = blockIdx.y * CO_RCH + ty; r0 < rend; r0 += 16) {
    float g[2][8], u[2][8];
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int r = r0 + 8 * k;
      if (r < rend) {
        ld8(gu + r * ldi + c, g[k]);
        if constexpr (!GELU) ld8(gu + r * ldi + hidden + c, u[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int r = r0 + 8 * k;
      if (r >= rend) break;
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] = GELU ? gelu_f(g[k][e]) : silu_f(g[k][e]) * u[k][e];
      if constexpr (MX) {
        float amax = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) { o[e] = bf16_round(o[e]); amax = fmaxf(amax, fabsf(o[e])); }
        amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
        amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
        float inv;
        const int ex = mx_exponent(amax, inv);
        *(uint2*)((unsigned char*)h + (int64_t)r * hidden + c) = make_uint2(mx_pack4(o, inv), mx_pack4(o + 4, inv));
        if ((tx & 3) == 0) mx_scales[mx_scale_index(r, c >> 5, rows)] = (unsigned char)(ex + 127);
      } else {
        st8(h + (int64_t)r * hidden + c, o);
      }
    }
  }
}

__device__ __forceinline__ void colsum_flush(float (&acc)[8], float* sbuf /*[256*8]*/, float* dst, int c, int ncols) {
  // combine the two row-lanes (ty = 0/1) through LDS, then one atomic per column
  const int tx = threadIdx.x & 127, ty = threadIdx.x >> 7;
  if (ty == 1) {
#pragma unroll
    for (int e = 0; e < 8; e++) sbuf[tx * 8 + e] = acc[e];
  }
  __syncthreads();
  if (ty == 0 && c < ncols) {
#pragma unroll
    for (int e = 0; e < 8; e++) atomicAdd(dst + c + e, acc[e] + sbuf[tx * 8 + e]);
  }
  __syncthreads();
}

constexpr int MB_RCH = 128;
// activation backward: a workgroup covers 256 columns x MB_RCH rows (32 column groups of 8 x 8 row lanes, two rows in flight
// per thread); the bias-gradient column sums cost one atomic per column and workgroup.  grid = (hidden / 256, rows / MB_RCH)
template <typename T, bool GELU>
__device__ __forceinline__ void mlp_act_bwd_body(int by, const T* __restrict__ dh, const T* __restrict__ gu, T* __restrict__ dgu,
                                                 int rows, int hidden, float* __restrict__ dbias) {
  __shared__ float sbuf[8 * 256];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c = blockIdx.x * 256 + tx * 8;
  const bool act = c < hidden;
  const int64_t ldi = GELU ? hidden : 2 * (int64_t)hidden;
  float sg[8], su[8];
#pragma unroll
  for (int e = 0; e < 8; e++) { sg[e] = 0.f; su[e] = 0.f; }
  const int rend = min(rows, (by + 1) * MB_RCH);
  if (act) {
    for (int r0 = by * MB_RCH + ty; r0 < rend; r0 += 16) {
      float d[2][8], g[2][8], u[2][8];
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int r = r0 + 8 * k;
        if (r < rend) {
          ld8_nt(dh + (int64_t)r * hidden + c, d[k]);       // (dh and the saved pre-activations are at their last use)
          ld8_nt(gu + r * ldi + c, g[k]);
          if constexpr (!GELU) ld8_nt(gu + r * ldi + hidden + c, u[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int r = r0 + 8 * k;
        if (r >= rend) break;
        float og[8], ou[8];
        if constexpr (GELU) {
#pragma unroll
          for (int e = 0; e < 8; e++) { og[e] = d[k][e] * gelu_grad_f(g[k][e]); sg[e] += og[e]; }
          st8(dgu + r * ldi + c, og);
        } else {
#pragma unroll
          for (int e = 0; e < 8; e++) {
            swiglu_bwd_f(d[k][e], g[k][e], u[k][e], og[e], ou[e]);
            sg[e] += og[e]; su[e] += ou[e];
          }
          st8(dgu + r * ldi + c, og);
          st8(dgu + r * ldi + hidden + c, ou);
        }
      }
    }
  }
  if (dbias) {
    const int col = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int which = 0; which < (GELU ? 1 : 2); which++) {
#pragma unroll
      for (int e = 0; e < 8; e++) sbuf[ty * 256 + tx * 8 + e] = which == 0 ? sg[e] : su[e];
      __syncthreads();
      if (col < hidden) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 8; r++) s += sbuf[r * 256 + threadIdx.x];
        atomicAdd(dbias + which * hidden + col, s);
      }
      __syncthreads();
    }
  }
}
template <typename T, bool GELU>
__global__ __launch_bounds__(256) void mlp_act_bwd_kernel(const T* __restrict__ dh, const T* __restrict__ gu, T* __restrict__ dgu,
                                                          int rows, int hidden, float* __restrict__ dbias) {
  mlp_act_bwd_body<T, GELU>((int)blockIdx.y, dh, gu, dgu, rows, hidden, dbias);
}
// two problems of the same hidden width (the image and the text MLP of a block) in one launch: blockIdx.y in [0, nby0) = problem 0
struct MlpBwdProb { const void* dh; const void* gu; void* dgu; int rows; float* dbias; };
template <typename T, bool GELU>
__global__ __launch_bounds__(256) void mlp_act_bwd_pair_kernel(MlpBwdProb p0, MlpBwdProb p1, int nby0, int hidden) {
  const bool first = (int)blockIdx.y < nby0;     // (workgroup-uniform)
  const MlpBwdProb& p = first ? p0 : p1;
  mlp_act_bwd_body<T, GELU>(first ? (int)blockIdx.y : (int)blockIdx.y - nby0, (const T*)p.dh, (const T*)p.gu, (T*)p.dgu, p.rows, hidden, p.dbias);
}

constexpr int GR_RCH = 64;
// dacc = dy * gate[b]; dgate[b] += sum dy*acc; dbias += sum dacc.  grid = (d / 256 column slabs, batch * row chunks).
// A workgroup covers 256 columns x GR_RCH rows: 32 column groups of 8 x 8 row lanes, two rows in flight per thread; the
// row lanes are combined through LDS and every column costs ONE atomic per workgroup (atomics are the expensive part).
template <typename TA, typename TO>
__global__ __launch_bounds__(256) void gate_res_bwd_kernel(const float* __restrict__ dy, const TA* __restrict__ acc, const float* __restrict__ gate, int64_t ld_gate,
                                                           int d, int rpb, int nchunk, TO* __restrict__ dacc,
                                                           float* __restrict__ dgate, int64_t ld_dgate, float* __restrict__ dbias, int64_t ld_dbias) {
  __shared__ float sbuf[8 * 256];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c = blockIdx.x * 256 + tx * 8;
  const bool act = c < d;
  const int b = blockIdx.y / nchunk, chunk = blockIdx.y % nchunk;
  float g[8], sgate[8], sb[8];
#pragma unroll
  for (int e = 0; e < 8; e++) { g[e] = 0.f; sgate[e] = 0.f; sb[e] = 0.f; }
  if (act) {
    ld8(gate + (int64_t)b * ld_gate + c, g);
    const int rend = min(rpb, (chunk + 1) * GR_RCH);
    for (int rl = chunk * GR_RCH + ty; rl < rend; rl += 16) {
      const int64_t row0 = (int64_t)b * rpb + rl, row1 = row0 + 8;
      const bool two = rl + 8 < rend;
      float dv0[8], av0[8], dv1[8], av1[8], o[8];
      ld8(dy + row0 * d + c, dv0);
      ld8(acc + row0 * d + c, av0);
      if (two) {
        ld8(dy + row1 * d + c, dv1);
        ld8(acc + row1 * d + c, av1);
      }
#pragma unroll
      for (int e = 0; e < 8; e++) { o[e] = dv0[e] * g[e]; sgate[e] += dv0[e] * av0[e]; sb[e] += o[e]; }
      st8(dacc + row0 * d + c, o);
      if (two) {
#pragma unroll
        for (int e = 0; e < 8; e++) { o[e] = dv1[e] * g[e]; sgate[e] += dv1[e] * av1[e]; sb[e] += o[e]; }
        st8(dacc + row1 * d + c, o);
      }
    }
  }
  // combine the 8 row lanes through LDS: thread t owns column slab_base + t
  const int col = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
  for (int which = 0; which < 2; which++) {
    float* dst = which == 0 ? dgate + (int64_t)b * ld_dgate : (dbias ? dbias + (int64_t)b * ld_dbias : nullptr);
    if (!dst) break;   // uniform
#pragma unroll
    for (int e = 0; e < 8; e++) sbuf[ty * 256 + tx * 8 + e] = which == 0 ? sgate[e] : sb[e];
    __syncthreads();
    if (col < d) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 8; r++) s += sbuf[r * 256 + threadIdx.x];
      atomicAdd(dst + col, s);
    }
    __syncthreads();
  }
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int rows, int cols, int64_t ld, float* __restrict__ out, int rch) {
  __shared__ float sbuf[1024];
  const int tx = threadIdx.x & 127, ty = threadIdx.x >> 7;
  const int c = blockIdx.x * 1024 + tx * 8;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; e++) s[e] = 0.f;
  if (c < cols) {
    const int rend = min(rows, (int)(blockIdx.y + 1) * rch);
#pragma unroll 4
    for (int r = blockIdx.y * rch + ty; r < rend; r += 2) {
      float v[8];
      ld8(x + r * ld + c, v);
#pragma unroll
      for (int e = 0; e < 8; e++) s[e] += v[e];
    }
  }
  colsum_flush(s, sbuf, out, c, cols);
}

// y_proj SiLU backward on small (rows, cols) matrices: 64 columns x 4 row lanes per block, the row lanes are combined
// through LDS and the column owner adds into dbias (one writer per column: no atomics).  blockIdx.y = row group: rows
// [g * rpg, (g+1) * rpg) feed dbias row g (the y_proj matrices of all transformer blocks stacked: one launch for all).
template <typename TG, typename TO>
__global__ __launch_bounds__(256) void silu_bwd_kernel(const TG* __restrict__ dy, const float* __restrict__ pre, TO* __restrict__ dpre, int rpg, int cols, float* __restrict__ dbias) {
  __shared__ float sb[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int r0 = blockIdx.y * rpg;
  float s = 0.f;
  if (c < cols) {
#pragma unroll 4
    for (int r = r0 + rl; r < r0 + rpg; r += 4) {
      const int64_t i = (int64_t)r * cols + c;
      const float p = pre[i], sig = sigmoid_f(p);
      const float g = io<TG>::ld(dy + i) * sig * (1.f + p * (1.f - sig));
      io<TO>::st(dpre + i, g);
      s += g;
    }
  }
  sb[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < cols && dbias) dbias[(int64_t)blockIdx.y * cols + c] += sb[0][cl] + sb[1][cl] + sb[2][cl] + sb[3][cl];
}

template <typename TI, typename TO>
__global__ void cast_kernel(const TI* __restrict__ src, TO* __restrict__ dst, int64_t n) {
  const int64_t n8 = n >> 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float v[8];
    ld8(src + i * 8, v);
    st8(dst + i * 8, v);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) io<TO>::st(dst + n8 * 8 + threadIdx.x, io<TI>::ld(src + n8 * 8 + threadIdx.x));
}

// patchify: one thread per (b, ch, y, j) moves two horizontally adjacent pixels
template <typename TI, typename TO, bool TO_TOKENS>
__global__ void patch_kernel(const TI* __restrict__ src, TO* __restrict__ dst, int batch, int ch, int H, int W) {
  const int W2 = W >> 1, H2 = H >> 1;
  const int64_t total = (int64_t)batch * ch * H * W2;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(g % W2); int64_t r = g / W2;
    const int y = (int)(r % H); r /= H;
    const int c = (int)(r % ch); const int64_t b = r / ch;
    const int64_t img_off = ((b * ch + c) * H + y) * W + 2 * j;
    const int64_t tok_off = ((b * H2 + (y >> 1)) * W2 + j) * (int64_t)(ch * 4) + c * 4 + (y & 1) * 2;
    if (TO_TOKENS) { io<TO>::st(dst + tok_off, io<TI>::ld(src + img_off)); io<TO>::st(dst + tok_off + 1, io<TI>::ld(src + img_off + 1)); }
    else { io<TO>::st(dst + img_off, io<TI>::ld(src + tok_off)); io<TO>::st(dst + img_off + 1, io<TI>::ld(src + tok_off + 1)); }
  }
}

template <typename TO>
__global__ void time_embed_fwd_kernel(const float* __restrict__ t, const float* __restrict__ ts, const float* __restrict__ denom, int batch, int dim, TO* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (i >= dim) return;
  const float tau = t[b] * ts[0];
  const int half = dim >> 1;
  const float v = i < half ? sinf(tau / denom[2 * i]) : cosf(tau / denom[2 * (i - half) + 1]);
  io<TO>::st(out + (int64_t)b * dim + i, v);
}

template <typename TG>
__global__ __launch_bounds__(256) void time_embed_bwd_kernel(const TG* __restrict__ dout, const float* __restrict__ t, const float* __restrict__ ts,
                                                             const float* __restrict__ denom, int batch, int dim, float* __restrict__ dts) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const float tau = t[b] * ts[0];
  const int half = dim >> 1;
  float s = 0.f;
  for (int i = threadIdx.x; i < dim; i += 256) {
    const float g = io<TG>::ld(dout + (int64_t)b * dim + i);
    if (i < half) { const float dn = denom[2 * i]; s += g * cosf(tau / dn) / dn; }
    else { const float dn = denom[2 * (i - half) + 1]; s -= g * sinf(tau / dn) / dn; }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(dts, (red[0] + red[1] + red[2] + red[3]) * t[b]);
}

inline int nit_for(int d) { return (d / 4 + 63) / 64; }
inline int grid_cap(int64_t n, int bs) { int64_t g = (n + bs - 1) / bs; return (int)(g < 1 ? 1 : g > 4096 ? 4096 : g); }

}  // namespace

// NIT = iterations of 64 lanes x 4 columns that cover a row; NEGATIVE when d == 256 * |NIT| exactly (the kernels then drop their
// per-iteration range checks: with the checks every load sits in its own branch and is waited for before the next one is issued --
// 6 serialized round trips per row in the adaLN backward instead of 2).  `d` must be in scope.
#define NIT_CASE(n, ...)                                                        \
  { if (d == (n) * 256) { constexpr int NIT = -(n); __VA_ARGS__; } else { constexpr int NIT = (n); __VA_ARGS__; } }
#define NIT_SWITCH(nit, ...)                                    \
  do {                                                          \
    if (nit <= 1) NIT_CASE(1, __VA_ARGS__)                      \
    else if (nit <= 2) NIT_CASE(2, __VA_ARGS__)                 \
    else if (nit <= 3) NIT_CASE(3, __VA_ARGS__)                 \
    else if (nit <= 4) NIT_CASE(4, __VA_ARGS__)                 \
    else if (nit <= 6) NIT_CASE(6, __VA_ARGS__)                 \
    else if (nit <= 9) NIT_CASE(9, __VA_ARGS__)                 \
    else if (nit <= 12) NIT_CASE(12, __VA_ARGS__)               \
    else NIT_CASE(16, __VA_ARGS__)                              \
  } while (0)

// The adaLN forward keeps its range checks as well: its unguarded variant is 8 % faster (33.6 -> 31 us for image + text), but the
// different instruction selection rounds a few outputs differently, and the MX-emitting variant of the kernel must stay bit-identical
// to "plain kernel + quantise pass" (test_mxfp8_mode_vs_mx_oracle) while the parity mode sits 1.3 % under its 1e-3 bar.
#define LN_FWD_NIT (NIT < 0 ? -NIT : NIT)      // (also the text RMSNorm forward, 12 us per launch: same reason, nothing to gain)

extern "C" int mmdit_abi_version(void) { return MMDIT_ABI_VERSION; }
extern "C" int mmdit_struct_size(int which) {
  switch (which) {
    case 0: return (int)sizeof(mmdit_gemm_args);
    case 1: return (int)sizeof(mmdit_ln_fwd_problem);
    case 2: return (int)sizeof(mmdit_ln_bwd_problem);
    case 3: return (int)sizeof(mmdit_qk_problem);
    case 4: return (int)sizeof(mmdit_mlp_bwd_problem);
    case 5: return (int)sizeof(mmdit_adamw_tensor);
    case 6: return (int)sizeof(mmdit_cast_tensor);
    case 7: return (int)sizeof(mmdit_qk_epilogue);
    default: return -1;
  }
}
extern "C" const char* mmdit_build_arch(void) { return "gfx950"; }

extern "C" int mmdit_ln_modulate_fwd(const float* x, const float* scale, const float* shift, int64_t ld_mod, int rows, int d, int rpb,
                                     void* out, int out_dtype, float* mean, float* rstd, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(x && scale && shift && out && mean && rstd && rows > 0 && d > 0 && d % 4 == 0 && d <= 4096 && rpb > 0 && ld_mod % 4 == 0);
  hipStream_t s = (hipStream_t)stream;
  const int nit = nit_for(d);
  dim3 grid((rows + 3) / 4);
  if (out_dtype == MMDIT_BF16) { NIT_SWITCH(nit, hipLaunchKernelGGL((ln_mod_fwd_kernel<LN_FWD_NIT, bf16_t, bf16_t, false>), grid, dim3(256), 0, s, x, scale, shift, ld_mod, rows, d, rpb, (bf16_t*)out, mean, rstd, nullptr, nullptr, 0, nullptr)); }
  else if (out_dtype == MMDIT_F32) { NIT_SWITCH(nit, hipLaunchKernelGGL((ln_mod_fwd_kernel<LN_FWD_NIT, float, float, false>), grid, dim3(256), 0, s, x, scale, shift, ld_mod, rows, d, rpb, (float*)out, mean, rstd, nullptr, nullptr, 0, nullptr)); }
  else return MMDIT_ERR_DTYPE;
  return mmdit_launch_status();
}

extern "C" int mmdit_ln_modulate_fwd_res(const float* x, const void* acc, int acc_dtype, const float* gate, int64_t ld_gate, float* x_out,
                                         const float* scale, const float* shift, int64_t ld_mod, int rows, int d, int rpb,
                                         void* out, int out_dtype, float* mean, float* rstd, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(x && acc && gate && x_out && scale && shift && out && mean && rstd && rows > 0 && d > 0 && d % 4 == 0 && d <= 4096 && rpb > 0 && ld_mod % 4 == 0 && ld_gate % 4 == 0);
  MMDIT_CHECK_ARG(acc_dtype == out_dtype);
  hipStream_t s = (hipStream_t)stream;
  const int nit = nit_for(d);
  dim3 grid((rows + 3) / 4);
  if (out_dtype == MMDIT_BF16) { NIT_SWITCH(nit, hipLaunchKernelGGL((ln_mod_fwd_kernel<LN_FWD_NIT, bf16_t, bf16_t, true>), grid, dim3(256), 0, s, x, scale, shift, ld_mod, rows, d, rpb, (bf16_t*)out, mean, rstd, (const bf16_t*)acc, gate, ld_gate, x_out)); }
  else if (out_dtype == MMDIT_F32) { NIT_SWITCH(nit, hipLaunchKernelGGL((ln_mod_fwd_kernel<LN_FWD_NIT, float, float, true>), grid, dim3(256), 0, s, x, scale, shift, ld_mod, rows, d, rpb, (float*)out, mean, rstd, (const float*)acc, gate, ld_gate, x_out)); }
  else return MMDIT_ERR_DTYPE;
  return mmdit_launch_status();
}

extern "C" int mmdit_ln_modulate_fwd_mx(const float* x, const void* acc, int acc_dtype, const float* gate, int64_t ld_gate, float* x_out,
                                        const float* scale, const float* shift, int64_t ld_mod, int rows, int d, int rpb,
                                        void* q_fp8, void* scales_e8m0, float* mean, float* rstd, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(x && scale && shift && q_fp8 && scales_e8m0 && mean && rstd && rows > 0 && d > 0 && d % 64 == 0 && d <= 4096 && rpb > 0 && ld_mod % 4 == 0);
  MMDIT_CHECK_ARG(!acc || (gate && x_out && acc_dtype == MMDIT_BF16 && ld_gate % 4 == 0));
  hipStream_t s = (hipStream_t)stream;
  const int nit = nit_for(d);
  dim3 grid((rows + 3) / 4);
  unsigned char* q = (unsigned char*)q_fp8;
  unsigned char* sc = (unsigned char*)scales_e8m0;
  if (acc) { NIT_SWITCH(nit, hipLaunchKernelGGL((ln_mod_fwd_kernel<LN_FWD_NIT, unsigned char, bf16_t, true, true>), grid, dim3(256), 0, s, x, scale, shift, ld_mod, rows, d, rpb, q, mean, rstd, (const bf16_t*)acc, gate, ld_gate, x_out, sc)); }
  else { NIT_SWITCH(nit, hipLaunchKernelGGL((ln_mod_fwd_kernel<LN_FWD_NIT, unsigned char, bf16_t, false, true>), grid, dim3(256), 0, s, x, scale, shift, ld_mod, rows, d, rpb, q, mean, rstd, nullptr, nullptr, 0, nullptr, sc)); }
  return mmdit_launch_status();
}

extern "C" int mmdit_gate_residual_fwd(const float* x, const void* acc, int acc_dtype, const float* gate, int64_t ld_gate, int rows, int d, int rpb,
                                       float* out, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(x && acc && gate && out && rows > 0 && d > 0 && d % 4 == 0 && rpb > 0 && ld_gate % 4 == 0);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(grid_cap((int64_t)rows * (d / 4), 256));
  if (acc_dtype == MMDIT_BF16) hipLaunchKernelGGL((gate_residual_fwd_kernel<bf16_t>), grid, dim3(256), 0, s, x, (const bf16_t*)acc, gate, ld_gate, rows, d, rpb, out);
  else if (acc_dtype == MMDIT_F32) hipLaunchKernelGGL((gate_residual_fwd_kernel<float>), grid, dim3(256), 0, s, x, (const float*)acc, gate, ld_gate, rows, d, rpb, out);
  else return MMDIT_ERR_DTYPE;
  return mmdit_launch_status();
}

template <typename TG, typename TA, bool GATED>
static int ln_mod_bwd_launch(const void* dout, const float* x, const float* mean, const float* rstd, const float* scale, int64_t ld_mod, const float* dres,
                             int rows, int d, int rpb, float* dx, float* dscale, float* dshift, int64_t ld_dmod, const void* acc, const float* gate,
                             int64_t ld_gate, void* dacc, float* dgate, int64_t ld_dgate, float* dbias, int64_t ld_dbias, hipStream_t s) {
  const int nit = nit_for(d), nchunk = (rpb + LN_BWD_RCH - 1) / LN_BWD_RCH;
  dim3 grid((rows / rpb) * nchunk);
  NIT_SWITCH(nit, hipLaunchKernelGGL((ln_mod_bwd_kernel<(NIT < 0 ? -NIT : NIT), TG, TA, GATED>), grid, dim3(256), 0, s, (const TG*)dout, x, mean, rstd, scale, ld_mod, dres, d, rpb, nchunk,
                                     dx, dscale, dshift, ld_dmod, (const TA*)acc, gate, ld_gate, (TA*)dacc, dgate, ld_dgate, dbias, ld_dbias));
  return mmdit_launch_status();
}

extern "C" int mmdit_ln_modulate_bwd(const void* dout, int dout_dtype, const float* x, const float* mean, const float* rstd, const float* scale, int64_t ld_mod,
                                     const float* dres, int rows, int d, int rpb, float* dx, float* dscale, float* dshift, int64_t ld_dmod, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(dout && x && mean && rstd && scale && dx && dscale && dshift && rows > 0 && d % 4 == 0 && d <= 4096 && rpb > 0 && rows % rpb == 0 && ld_mod % 4 == 0);
  hipStream_t s = (hipStream_t)stream;
  if (dout_dtype == MMDIT_BF16) return ln_mod_bwd_launch<bf16_t, bf16_t, false>(dout, x, mean, rstd, scale, ld_mod, dres, rows, d, rpb, dx, dscale, dshift, ld_dmod, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, s);
  if (dout_dtype == MMDIT_F32) return ln_mod_bwd_launch<float, float, false>(dout, x, mean, rstd, scale, ld_mod, dres, rows, d, rpb, dx, dscale, dshift, ld_dmod, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, s);
  return MMDIT_ERR_DTYPE;
}

extern "C" int mmdit_ln_modulate_bwd_gated(const void* dout, int dout_dtype, const float* x, const float* mean, const float* rstd, const float* scale, int64_t ld_mod,
                                           const float* dres, int rows, int d, int rpb, float* dx, float* dscale, float* dshift, int64_t ld_dmod,
                                           const void* acc, int acc_dtype, const float* gate, int64_t ld_gate, void* dacc, float* dgate, int64_t ld_dgate,
                                           float* dbias, int64_t ld_dbias, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(dout && x && mean && rstd && scale && dx && dscale && dshift && rows > 0 && d % 4 == 0 && d <= 4096 && rpb > 0 && rows % rpb == 0 && ld_mod % 4 == 0);
  MMDIT_CHECK_ARG(acc && gate && dacc && dgate && ld_gate % 4 == 0 && acc_dtype == dout_dtype);
  hipStream_t s = (hipStream_t)stream;
  if (dout_dtype == MMDIT_BF16) return ln_mod_bwd_launch<bf16_t, bf16_t, true>(dout, x, mean, rstd, scale, ld_mod, dres, rows, d, rpb, dx, dscale, dshift, ld_dmod, acc, gate, ld_gate, dacc, dgate, ld_dgate, dbias, ld_dbias, s);
  if (dout_dtype == MMDIT_F32) return ln_mod_bwd_launch<float, float, true>(dout, x, mean, rstd, scale, ld_mod, dres, rows, d, rpb, dx, dscale, dshift, ld_dmod, acc, gate, ld_gate, dacc, dgate, ld_dgate, dbias, ld_dbias, s);
  return MMDIT_ERR_DTYPE;
}

extern "C" int mmdit_ln_modulate_fwd_pair(const mmdit_ln_fwd_problem* a, const mmdit_ln_fwd_problem* b, int d, int acc_dtype, int out_dtype, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(a && b && d > 0 && d % 4 == 0 && d <= 4096);
  const bool res = a->acc != nullptr;
  MMDIT_CHECK_ARG((b->acc != nullptr) == res && (!res || acc_dtype == out_dtype));
  LnFwdProb q[2];
  const mmdit_ln_fwd_problem* src[2] = {a, b};
  for (int i = 0; i < 2; i++) {
    const mmdit_ln_fwd_problem* p = src[i];
    MMDIT_CHECK_ARG(p->x && p->scale && p->shift && p->out && p->mean && p->rstd && p->rows > 0 && p->rows_per_batch > 0 && p->ld_mod % 4 == 0);
    if (res) MMDIT_CHECK_ARG(p->gate && p->x_out && p->ld_gate % 4 == 0);
    q[i] = LnFwdProb{p->x, p->scale, p->shift, p->ld_mod, p->rows, p->rows_per_batch, p->out, p->mean, p->rstd, p->acc, p->gate, p->ld_gate, p->x_out};
  }
  hipStream_t s = (hipStream_t)stream;
  const int nit = nit_for(d), nb0 = (a->rows + 3) / 4, nb1 = (b->rows + 3) / 4;
  dim3 grid(nb0 + nb1);
#define LNP(TO, RES) NIT_SWITCH(nit, hipLaunchKernelGGL((ln_mod_fwd_pair_kernel<LN_FWD_NIT, TO, TO, RES>), grid, dim3(256), 0, s, q[0], q[1], nb0, d))
  if (out_dtype == MMDIT_BF16) { if (res) { LNP(bf16_t, true); } else { LNP(bf16_t, false); } }
  else if (out_dtype == MMDIT_F32) { if (res) { LNP(float, true); } else { LNP(float, false); } }
  else return MMDIT_ERR_DTYPE;
#undef LNP
  return mmdit_launch_status();
}

extern "C" int mmdit_ln_modulate_bwd_pair(const mmdit_ln_bwd_problem* a, const mmdit_ln_bwd_problem* b, int d, int dout_dtype, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(a && b && d > 0 && d % 4 == 0 && d <= 4096 && a->rows_per_batch > 0 && b->rows_per_batch > 0);
  const bool gated = a->acc != nullptr;
  MMDIT_CHECK_ARG((b->acc != nullptr) == gated);
  LnBwdProb q[2];
  const mmdit_ln_bwd_problem* src[2] = {a, b};
  int nb[2];
  // rows per workgroup: 16, or more when 16 would give more workgroups than fit the chip at once (4 of these 256-thread workgroups per CU): the
  // image + text launch of MMDiT-B at batch 64 is 1640 workgroups of 16 rows = 1.6 rounds, 832 of 32 rows or 1024 of 28 rows = one round -- measured
  // 68.4 (16) -> 63.8 us (32) (single-stream launches, one round either way, are faster with 16: 42 vs 47 us)
  // (generalised: the smallest multiple of 4 rows -- one row per wave and step -- for which the launch fits the chip in ONE round; the resident
  //  workgroups per CU follow from the kernel's registers: 4 up to d = 768, 3 at d = 1024, 2 above)
  auto wgs = [&](int r) { return (long)(a->rows / a->rows_per_batch) * ((a->rows_per_batch + r - 1) / r) + (long)(b->rows / b->rows_per_batch) * ((b->rows_per_batch + r - 1) / r); };
  const int nit_ = nit_for(d), nit_abs = nit_ < 0 ? -nit_ : nit_;
  // resident workgroups = CUs of THIS device x workgroups per CU.  Per CU: the measured table (4 / 3 / 2 by row width), capped by what the
  // occupancy query says for the exact instantiation being launched -- the query follows the kernel's registers if they ever grow, but on ROCm 7.2
  // it over-reports by one workgroup in some SGPR bands (MI355X_MICROARCH.md, "Residency"), so it is an upper bound, never the value.
  // Queried once per (device, instantiation); a wrong answer costs speed (two rounds instead of one), never correctness.
  int per_cu = nit_abs <= 3 ? 4 : nit_abs == 4 ? 3 : 2, cus = 256;
  {
    static int occ[64][17][2][2], cu_count[64];      // 0 = not asked yet
    const int dev = mmdit_current_device(), ti = dout_dtype == MMDIT_BF16 ? 0 : 1, gi = gated ? 1 : 0, ni = nit_abs > 16 ? 16 : nit_abs;
    if (!cu_count[dev]) {
      int n = 0;
      cu_count[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    cus = cu_count[dev];
    if (!occ[dev][ni][ti][gi]) {
      int nbq = 0;
      hipError_t e = hipErrorUnknown;
#define LNQ(T, G) NIT_SWITCH(nit_, e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbq, (const void*)ln_mod_bwd_pair_kernel<(NIT < 0 ? -NIT : NIT), T, T, G>, 256, 0))
      if (dout_dtype == MMDIT_BF16) { if (gated) { LNQ(bf16_t, true); } else { LNQ(bf16_t, false); } }
      else if (dout_dtype == MMDIT_F32) { if (gated) { LNQ(float, true); } else { LNQ(float, false); } }
#undef LNQ
      occ[dev][ni][ti][gi] = (e == hipSuccess && nbq > 0) ? nbq : 8;
    }
    per_cu = per_cu < occ[dev][ni][ti][gi] ? per_cu : occ[dev][ni][ti][gi];
  }
  const long slots = (long)cus * per_cu;
  int rch = LN_BWD_RCH;
  if (wgs(LN_BWD_RCH) > slots && wgs(LN_BWD_RCH) <= 2 * slots)
    for (int r = LN_BWD_RCH + 4; r <= 64; r += 4)
      if (wgs(r) <= slots) { rch = r; break; }
  for (int i = 0; i < 2; i++) {
    const mmdit_ln_bwd_problem* p = src[i];
    MMDIT_CHECK_ARG(p->dout && p->x && p->mean && p->rstd && p->scale && p->dx && p->dscale && p->dshift && p->rows > 0 && p->rows_per_batch > 0 &&
                    p->rows % p->rows_per_batch == 0 && p->ld_mod % 4 == 0);
    if (gated) MMDIT_CHECK_ARG(p->gate && p->dacc && p->dgate && p->ld_gate % 4 == 0);
    const int nchunk = (p->rows_per_batch + rch - 1) / rch;
    nb[i] = (p->rows / p->rows_per_batch) * nchunk;
    q[i] = LnBwdProb{p->dout, p->x, p->mean, p->rstd, p->scale, p->ld_mod, p->dres, p->rows_per_batch, nchunk, p->dx, p->dscale, p->dshift, p->ld_dmod,
                     p->acc, p->gate, p->ld_gate, p->dacc, p->dgate, p->ld_dgate, p->dbias, p->ld_dbias};
  }
  hipStream_t s = (hipStream_t)stream;
  const int nit = nit_for(d);
  dim3 grid(nb[0] + nb[1]);
// (the backward keeps its range checks: without them the batched loads cost 24 more VGPRs = one wave per SIMD less, measured 79 -> 84 us)
#define LNP(T, G) NIT_SWITCH(nit, hipLaunchKernelGGL((ln_mod_bwd_pair_kernel<(NIT < 0 ? -NIT : NIT), T, T, G>), grid, dim3(256), 0, s, q[0], q[1], nb[0], d, rch))
  if (dout_dtype == MMDIT_BF16) { if (gated) { LNP(bf16_t, true); } else { LNP(bf16_t, false); } }
  else if (dout_dtype == MMDIT_F32) { if (gated) { LNP(float, true); } else { LNP(float, false); } }
  else return MMDIT_ERR_DTYPE;
#undef LNP
  return mmdit_launch_status();
}

extern "C" int mmdit_text_rmsnorm_fwd(const void* x, int x_dtype, const float* w1, const float* w2, const float* s1, const float* s2,
                                      int batch, int tokens, int split, int d, void* out1, void* out2, int out_dtype, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(x && w1 && w2 && s1 && s2 && out1 && out2 && batch > 0 && split > 0 && split < tokens && d % 4 == 0 && d <= 4096);
  hipStream_t s = (hipStream_t)stream;
  const int nit = nit_for(d);
  for (int half = 0; half < 2; half++) {
    const int cnt = half ? tokens - split : split, off = half ? split : 0, rows = batch * cnt;
    const float* w = half ? w2 : w1; const float* sp = half ? s2 : s1; void* out = half ? out2 : out1;
    dim3 grid((rows + 3) / 4);
#define TRF(TI, TO) NIT_SWITCH(nit, hipLaunchKernelGGL((text_rms_fwd_kernel<LN_FWD_NIT, TI, TO>), grid, dim3(256), 0, s, (const TI*)x, w, sp, rows, cnt, tokens, off, d, (TO*)out))
    if (x_dtype == MMDIT_F32 && out_dtype == MMDIT_F32) { TRF(float, float); }
    else if (x_dtype == MMDIT_F32 && out_dtype == MMDIT_BF16) { TRF(float, bf16_t); }
    else if (x_dtype == MMDIT_BF16 && out_dtype == MMDIT_BF16) { TRF(bf16_t, bf16_t); }
    else if (x_dtype == MMDIT_BF16 && out_dtype == MMDIT_F32) { TRF(bf16_t, float); }
    else return MMDIT_ERR_DTYPE;
#undef TRF
  }
  return mmdit_launch_status();
}

extern "C" int mmdit_text_rmsnorm_bwd(const void* dout1, const void* dout2, int dout_dtype, const void* x, int x_dtype,
                                      const float* w1, const float* w2, const float* s1, const float* s2, int batch, int tokens, int split, int d,
                                      float* dw1, float* dw2, float* ds1, float* ds2, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(dout1 && dout2 && x && w1 && w2 && s1 && s2 && dw1 && dw2 && ds1 && ds2 && batch > 0 && split > 0 && split < tokens && d % 4 == 0 && d <= 4096);
  hipStream_t s = (hipStream_t)stream;
  const int nit = nit_for(d);
  TextRmsBwdHalf h[2];
  int maxrows = 0;
  for (int half = 0; half < 2; half++) {
    const int cnt = half ? tokens - split : split, rows = batch * cnt;
    h[half] = TextRmsBwdHalf{half ? dout2 : dout1, half ? w2 : w1, half ? s2 : s1, rows, cnt, half ? split : 0, half ? dw2 : dw1, half ? ds2 : ds1};
    maxrows = rows > maxrows ? rows : maxrows;
  }
  dim3 grid((maxrows + TRB_ROWS - 1) / TRB_ROWS, 2);     // (a half with fewer rows: its surplus workgroups find an empty row range)
#define TRB(TI, TG) NIT_SWITCH(nit, hipLaunchKernelGGL((text_rms_bwd_kernel<NIT, TI, TG>), grid, dim3(512), 0, s, h[0], h[1], (const TI*)x, tokens, d))
  if (x_dtype == MMDIT_F32 && dout_dtype == MMDIT_F32) { TRB(float, float); }
  else if (x_dtype == MMDIT_F32 && dout_dtype == MMDIT_BF16) { TRB(float, bf16_t); }
  else if (x_dtype == MMDIT_BF16 && dout_dtype == MMDIT_BF16) { TRB(bf16_t, bf16_t); }
  else if (x_dtype == MMDIT_BF16 && dout_dtype == MMDIT_F32) { TRB(bf16_t, float); }
  else return MMDIT_ERR_DTYPE;
#undef TRB
  return mmdit_launch_status();
}

extern "C" int mmdit_qk_norm_rope_fwd(const void* qkv, int qkv_dtype, const float* wq, const float* wk, const float* rope_cos, const float* rope_sin,
                                      int batch, int tokens, int heads, int s_total, int tok0, void* Q, void* K, void* V, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(qkv && wq && wk && Q && K && V && batch > 0 && tokens > 0 && heads > 0 && tok0 >= 0 && tok0 + tokens <= s_total);
  MMDIT_CHECK_ARG((rope_cos == nullptr) == (rope_sin == nullptr));
  hipStream_t s = (hipStream_t)stream;
  MMDIT_CHECK_ARG(24 * heads <= 1024);   // one row per workgroup of 24*heads threads
  const int rows = batch * tokens;
  int g = rows < 1024 ? rows : 1024;
  if (rope_cos && tokens <= 1024 && rows >= tokens) g = (g / tokens > 0 ? g / tokens : 1) * tokens;
  dim3 grid(g);
  if (qkv_dtype == MMDIT_BF16) hipLaunchKernelGGL((qk_norm_rope_fwd_kernel<bf16_t>), grid, dim3(24 * heads), 0, s, (const bf16_t*)qkv, wq, wk, rope_cos, rope_sin, rows, tokens, heads, s_total, tok0, (bf16_t*)Q, (bf16_t*)K, (bf16_t*)V);
  else if (qkv_dtype == MMDIT_F32) hipLaunchKernelGGL((qk_norm_rope_fwd_kernel<float>), grid, dim3(24 * heads), 0, s, (const float*)qkv, wq, wk, rope_cos, rope_sin, rows, tokens, heads, s_total, tok0, (bf16_t*)Q, (bf16_t*)K, (bf16_t*)V);
  else return MMDIT_ERR_DTYPE;
  return mmdit_launch_status();
}

// Launch shape of the QK-norm/RoPE backward: RL row lanes per workgroup (as many as fit 1024 threads; every workgroup ends with 128
// global atomics on the same two cache lines, so fewer, fatter workgroups), `lanes` rows in flight per iteration pair, a multiple of the
// tokens per sample when RoPE applies (same token for all rows of a lane: factors loaded once).
static int qk_bwd_rl(int heads, int min_rows) {
  static const int env = [] { const char* e = mmdit_exp_env("MMDIT_QK_RL"); return e ? atoi(e) : 0; }();
  int rl = 1024 / (24 * heads);
  if (env > 0 && env < rl) rl = env;
  if (rl < 1 || min_rows < 2048) rl = 1;
  return rl;
}
static int qk_bwd_grid(int rows, int tokens, bool rope, int& rl) {
  static const int lanes_max = [] { const char* e = mmdit_exp_env("MMDIT_QK_LANES"); return e ? atoi(e) : 768; }();
  if (rl == 1) {
    int g = rows < 512 ? rows : 512;
    if (rope && tokens <= 1024 && rows >= tokens) g = (g / tokens > 0 ? g / tokens : 1) * tokens;
    return g;
  }
  if (!rope) return lanes_max / rl;
  if (tokens > lanes_max) return tokens <= 2048 && tokens % rl == 0 ? tokens / rl : lanes_max / rl;      // (one lane per token: 512^2 images, 1024 tokens)
  int k = lanes_max / tokens;
  while (k > 1 && (k * tokens) % rl) k--;
  if ((k * tokens) % rl) return lanes_max / rl;       // (lanes then walk all tokens: factors reloaded per row)
  return k * tokens / rl;
}

extern "C" int mmdit_qk_norm_rope_bwd(const void* dQ, const void* dK, const void* dV, int dq_dtype, const void* qkv, int qkv_dtype,
                                      const float* wq, const float* wk, const float* rope_cos, const float* rope_sin,
                                      int batch, int tokens, int heads, int s_total, int tok0, void* dqkv, int dqkv_dtype, float* dwq, float* dwk, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(dQ && dK && dV && qkv && wq && wk && dqkv && dwq && dwk && batch > 0 && tokens > 0 && heads > 0 && tok0 >= 0 && tok0 + tokens <= s_total);
  MMDIT_CHECK_ARG((rope_cos == nullptr) == (rope_sin == nullptr));
  hipStream_t s = (hipStream_t)stream;
  MMDIT_CHECK_ARG(heads >= 1 && 24 * heads <= 1024);   // one row per workgroup of 24*heads threads
  const int rows = batch * tokens;
  int rl = qk_bwd_rl(heads, rows);
  dim3 grid(qk_bwd_grid(rows, tokens, rope_cos != nullptr, rl));
  const bool fast = !rope_cos || ((int)grid.x * rl) % tokens == 0;
#define QKB(TG, TI, TO) do { if (fast) hipLaunchKernelGGL((qk_norm_rope_bwd_kernel<true, TG, TI, TO>), grid, dim3(24 * heads * rl), 0, s, (const TG*)dQ, (const TG*)dK, (const TG*)dV, (const TI*)qkv, wq, wk, rope_cos, rope_sin, rows, tokens, heads, s_total, tok0, (TO*)dqkv, dwq, dwk); \
  else hipLaunchKernelGGL((qk_norm_rope_bwd_kernel<false, TG, TI, TO>), grid, dim3(24 * heads * rl), 0, s, (const TG*)dQ, (const TG*)dK, (const TG*)dV, (const TI*)qkv, wq, wk, rope_cos, rope_sin, rows, tokens, heads, s_total, tok0, (TO*)dqkv, dwq, dwk); } while (0)
  if (dq_dtype == MMDIT_BF16 && qkv_dtype == MMDIT_BF16 && dqkv_dtype == MMDIT_BF16) QKB(bf16_t, bf16_t, bf16_t);
  else if (dq_dtype == MMDIT_F32 && qkv_dtype == MMDIT_F32 && dqkv_dtype == MMDIT_F32) QKB(float, float, float);
  else if (dq_dtype == MMDIT_BF16 && qkv_dtype == MMDIT_F32 && dqkv_dtype == MMDIT_F32) QKB(bf16_t, float, float);
  else return MMDIT_ERR_DTYPE;
#undef QKB
  return mmdit_launch_status();
}

extern "C" int mmdit_qk_norm_rope_fwd_pair(const mmdit_qk_problem* a, const mmdit_qk_problem* b, int qkv_dtype, int batch, int heads, int s_total,
                                           void* Q, void* K, void* V, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(a && b && Q && K && V && batch > 0 && heads > 0 && 24 * heads <= 1024);
  const mmdit_qk_problem* src[2] = {a, b};
  QkProb q[2];
  int g[2];
  for (int i = 0; i < 2; i++) {
    const mmdit_qk_problem* p = src[i];
    MMDIT_CHECK_ARG(p->qkv && p->wq && p->wk && p->tokens > 0 && p->tok0 >= 0 && p->tok0 + p->tokens <= s_total && (p->rope_cos == nullptr) == (p->rope_sin == nullptr));
    const int rows = batch * p->tokens;
    g[i] = rows < 1024 ? rows : 1024;
    if (p->rope_cos && p->tokens <= 1024) g[i] = (g[i] / p->tokens > 0 ? g[i] / p->tokens : 1) * p->tokens;
    q[i] = QkProb{p->qkv, p->wq, p->wk, p->rope_cos, p->rope_sin, rows, p->tokens, p->tok0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  }
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(g[0] + g[1]);
  if (qkv_dtype == MMDIT_BF16) hipLaunchKernelGGL((qk_norm_rope_fwd_pair_kernel<bf16_t>), grid, dim3(24 * heads), 0, s, q[0], q[1], g[0], heads, s_total, (bf16_t*)Q, (bf16_t*)K, (bf16_t*)V);
  else if (qkv_dtype == MMDIT_F32) hipLaunchKernelGGL((qk_norm_rope_fwd_pair_kernel<float>), grid, dim3(24 * heads), 0, s, q[0], q[1], g[0], heads, s_total, (bf16_t*)Q, (bf16_t*)K, (bf16_t*)V);
  else return MMDIT_ERR_DTYPE;
  return mmdit_launch_status();
}

extern "C" int mmdit_qk_norm_rope_bwd_pair(const mmdit_qk_problem* a, const mmdit_qk_problem* b, const void* dQ, const void* dK, const void* dV, int dq_dtype,
                                           int qkv_dtype, int dqkv_dtype, int batch, int heads, int s_total, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(a && b && dQ && dK && dV && batch > 0 && heads > 0 && 24 * heads <= 1024);
  const mmdit_qk_problem* src[2] = {a, b};
  QkProb q[2];
  int g[2];
  for (int i = 0; i < 2; i++) {
    const mmdit_qk_problem* p = src[i];
    MMDIT_CHECK_ARG(p->qkv && p->wq && p->wk && p->dqkv && p->dwq && p->dwk && p->tokens > 0 && p->tok0 >= 0 && p->tok0 + p->tokens <= s_total &&
                    (p->rope_cos == nullptr) == (p->rope_sin == nullptr));
    q[i] = QkProb{p->qkv, p->wq, p->wk, p->rope_cos, p->rope_sin, batch * p->tokens, p->tokens, p->tok0, dQ, dK, dV, p->dqkv, p->dwq, p->dwk};
  }
  int rl = qk_bwd_rl(heads, q[0].rows < q[1].rows ? q[0].rows : q[1].rows);     // (one block shape for both problems)
  for (int i = 0; i < 2; i++) g[i] = qk_bwd_grid(q[i].rows, q[i].tokens, q[i].rcos != nullptr, rl);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(g[0] + g[1]);
  bool fast = true;
  for (int i = 0; i < 2; i++) fast = fast && (!q[i].rcos || (g[i] * rl) % q[i].tokens == 0);
#define QKP(TG, TI, TO) do { if (fast) hipLaunchKernelGGL((qk_norm_rope_bwd_pair_kernel<true, TG, TI, TO>), grid, dim3(24 * heads * rl), 0, s, q[0], q[1], g[0], heads, s_total); \
  else hipLaunchKernelGGL((qk_norm_rope_bwd_pair_kernel<false, TG, TI, TO>), grid, dim3(24 * heads * rl), 0, s, q[0], q[1], g[0], heads, s_total); } while (0)
  if (dq_dtype == MMDIT_BF16 && qkv_dtype == MMDIT_BF16 && dqkv_dtype == MMDIT_BF16) QKP(bf16_t, bf16_t, bf16_t);
  else if (dq_dtype == MMDIT_F32 && qkv_dtype == MMDIT_F32 && dqkv_dtype == MMDIT_F32) QKP(float, float, float);
  else if (dq_dtype == MMDIT_BF16 && qkv_dtype == MMDIT_F32 && dqkv_dtype == MMDIT_F32) QKP(bf16_t, float, float);
  else return MMDIT_ERR_DTYPE;
#undef QKP
  return mmdit_launch_status();
}

extern "C" int mmdit_mlp_act_bwd_pair(const mmdit_mlp_bwd_problem* a, const mmdit_mlp_bwd_problem* b, int dtype, int hidden, int gelu, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(a && b && hidden > 0 && hidden % 8 == 0);
  const mmdit_mlp_bwd_problem* src[2] = {a, b};
  MlpBwdProb q[2];
  int nby[2];
  for (int i = 0; i < 2; i++) {
    MMDIT_CHECK_ARG(src[i]->dh && src[i]->gu && src[i]->dgu && src[i]->rows > 0);
    q[i] = MlpBwdProb{src[i]->dh, src[i]->gu, src[i]->dgu, src[i]->rows, src[i]->dbias};
    nby[i] = (src[i]->rows + MB_RCH - 1) / MB_RCH;
  }
  dim3 grid((hidden + 255) / 256, nby[0] + nby[1]);
  hipStream_t s = (hipStream_t)stream;
#define MBP(T, G) hipLaunchKernelGGL((mlp_act_bwd_pair_kernel<T, G>), grid, dim3(256), 0, s, q[0], q[1], nby[0], hidden)
  if (dtype == MMDIT_BF16) { if (gelu) MBP(bf16_t, true); else MBP(bf16_t, false); }
  else if (dtype == MMDIT_F32) { if (gelu) MBP(float, true); else MBP(float, false); }
  else return MMDIT_ERR_DTYPE;
#undef MBP
  return mmdit_launch_status();
}

template <bool GELU>
static int mlp_act_fwd(const void* gu, void* h, int dtype, int rows, int hidden, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(gu && h && rows > 0 && hidden > 0 && hidden % 8 == 0);
  dim3 grid((hidden + 255) / 256, (rows + CO_RCH - 1) / CO_RCH);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MMDIT_BF16) hipLaunchKernelGGL((mlp_act_fwd_kernel<bf16_t, GELU>), grid, dim3(256), 0, s, (const bf16_t*)gu, (bf16_t*)h, rows, hidden);
  else if (dtype == MMDIT_F32) hipLaunchKernelGGL((mlp_act_fwd_kernel<float, GELU>), grid, dim3(256), 0, s, (const float*)gu, (float*)h, rows, hidden);
  else return MMDIT_ERR_DTYPE;
  return mmdit_launch_status();
}
template <bool GELU>
static int mlp_act_bwd(const void* dh, const void* gu, void* dgu, int dtype, int rows, int hidden, float* dbias, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(dh && gu && dgu && rows > 0 && hidden > 0 && hidden % 8 == 0);
  dim3 grid((hidden + 255) / 256, (rows + MB_RCH - 1) / MB_RCH);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MMDIT_BF16) hipLaunchKernelGGL((mlp_act_bwd_kernel<bf16_t, GELU>), grid, dim3(256), 0, s, (const bf16_t*)dh, (const bf16_t*)gu, (bf16_t*)dgu, rows, hidden, dbias);
  else if (dtype == MMDIT_F32) hipLaunchKernelGGL((mlp_act_bwd_kernel<float, GELU>), grid, dim3(256), 0, s, (const float*)dh, (const float*)gu, (float*)dgu, rows, hidden, dbias);
  else return MMDIT_ERR_DTYPE;
  return mmdit_launch_status();
}
extern "C" int mmdit_swiglu_fwd_mx(const void* gu, int dtype, int rows, int hidden, void* q_fp8, void* scales_e8m0, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(gu && q_fp8 && scales_e8m0 && rows > 0 && hidden > 0 && hidden % 64 == 0);
  dim3 grid((hidden + 255) / 256, (rows + CO_RCH - 1) / CO_RCH);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MMDIT_BF16) hipLaunchKernelGGL((mlp_act_fwd_kernel<bf16_t, false, true>), grid, dim3(256), 0, s, (const bf16_t*)gu, (bf16_t*)q_fp8, rows, hidden, (unsigned char*)scales_e8m0);
  else return MMDIT_ERR_DTYPE;
  return mmdit_launch_status();
}
extern "C" int mmdit_swiglu_fwd(const void* gu, void* h, int dtype, int rows, int hidden, mmdit_stream_t st) { return mlp_act_fwd<false>(gu, h, dtype, rows, hidden, st); }
extern "C" int mmdit_swiglu_bwd(const void* dh, const void* gu, void* dgu, int dtype, int rows, int hidden, float* dbias, mmdit_stream_t st) { return mlp_act_bwd<false>(dh, gu, dgu, dtype, rows, hidden, dbias, st); }
extern "C" int mmdit_gelu_fwd(const void* u, void* h, int dtype, int rows, int hidden, mmdit_stream_t st) { return mlp_act_fwd<true>(u, h, dtype, rows, hidden, st); }
extern "C" int mmdit_gelu_bwd(const void* dh, const void* u, void* du, int dtype, int rows, int hidden, float* dbias, mmdit_stream_t st) { return mlp_act_bwd<true>(dh, u, du, dtype, rows, hidden, dbias, st); }

extern "C" int mmdit_silu_bwd(const void* dy, int dy_dtype, const float* pre, void* dpre, int dpre_dtype, int rows, int cols, float* dbias, int rows_per_bias,
                              mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(dy && pre && dpre && rows > 0 && cols > 0);
  const int rpg = rows_per_bias > 0 ? rows_per_bias : rows;
  MMDIT_CHECK_ARG(rows % rpg == 0);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((cols + 63) / 64, rows / rpg);
  if (dy_dtype == MMDIT_BF16 && dpre_dtype == MMDIT_BF16) hipLaunchKernelGGL((silu_bwd_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)dy, pre, (bf16_t*)dpre, rpg, cols, dbias);
  else if (dy_dtype == MMDIT_F32 && dpre_dtype == MMDIT_F32) hipLaunchKernelGGL((silu_bwd_kernel<float, float>), grid, dim3(256), 0, s, (const float*)dy, pre, (float*)dpre, rpg, cols, dbias);
  else if (dy_dtype == MMDIT_F32 && dpre_dtype == MMDIT_BF16) hipLaunchKernelGGL((silu_bwd_kernel<float, bf16_t>), grid, dim3(256), 0, s, (const float*)dy, pre, (bf16_t*)dpre, rpg, cols, dbias);
  else return MMDIT_ERR_DTYPE;
  return mmdit_launch_status();
}

extern "C" int mmdit_gate_residual_bwd(const float* dy, const void* acc, int acc_dtype, const float* gate, int64_t ld_gate, int rows, int d, int rpb,
                                       void* dacc, int dacc_dtype, float* dgate, int64_t ld_dgate, float* dbias, int64_t ld_dbias, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(dy && acc && gate && dacc && dgate && rows > 0 && d % 8 == 0 && rpb > 0 && rows % rpb == 0 && ld_gate % 4 == 0);
  hipStream_t s = (hipStream_t)stream;
  const int nchunk = (rpb + GR_RCH - 1) / GR_RCH;
  dim3 grid((d + 255) / 256, (rows / rpb) * nchunk);
  if (acc_dtype == MMDIT_BF16 && dacc_dtype == MMDIT_BF16) hipLaunchKernelGGL((gate_res_bwd_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, s, dy, (const bf16_t*)acc, gate, ld_gate, d, rpb, nchunk, (bf16_t*)dacc, dgate, ld_dgate, dbias, ld_dbias);
  else if (acc_dtype == MMDIT_F32 && dacc_dtype == MMDIT_F32) hipLaunchKernelGGL((gate_res_bwd_kernel<float, float>), grid, dim3(256), 0, s, dy, (const float*)acc, gate, ld_gate, d, rpb, nchunk, (float*)dacc, dgate, ld_dgate, dbias, ld_dbias);
  else return MMDIT_ERR_DTYPE;
  return mmdit_launch_status();
}

namespace {
// rectified-flow loss: see mmdit_flow_loss in the header.  Grid-stride over groups of 8 elements, per-lane fp32 sums, wave reduction,
// the four wave sums of a workgroup added in wave order by lane 0.
template <typename T>
__global__ __launch_bounds__(256) void flow_loss_partial_kernel(const float* __restrict__ v, const T* __restrict__ x0, const T* __restrict__ eps, int64_t n8, float coef2,
                                                                float* __restrict__ dv, float* __restrict__ partials) {
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    float a[8], b[8], c[8], g[8];
    ld8(v + i * 8, a);
    ld8(x0 + i * 8, b);
    ld8(eps + i * 8, c);
#pragma unroll
    for (int e = 0; e < 8; e++) {
      float lab = c[e] - b[e];
      if (sizeof(T) == 2) lab = bf2f(f2bf(lab));      // torch computes eps - x0 in bf16 (one rounding), then casts to v's dtype
      const float d = a[e] - lab;
      s += d * d;
      g[e] = coef2 * d;
    }
    if (dv) st8(dv + i * 8, g);
  }
  __shared__ float sm[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = ((sm[0] + sm[1]) + sm[2]) + sm[3];
}
__global__ __launch_bounds__(64) void flow_loss_final_kernel(const float* __restrict__ partials, int count, float coef, float* __restrict__ loss) {
  // 64 lanes x 4 partials in a fixed order, then the butterfly: the same summation tree in every run
  float s = 0.f;
  for (int i = threadIdx.x; i < count; i += 64) s += partials[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) loss[0] = s * coef;
}
}  // namespace

extern "C" int mmdit_flow_loss(const float* v, const void* x0, const void* eps, int in_dtype, int64_t n, float coef, float* dv, float* partials,
                               float* loss, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(v && x0 && eps && partials && loss && n > 0 && n % 8 == 0);
  const int64_t n8 = n / 8;
  const int grid = (int)((n8 + 255) / 256 < 256 ? (n8 + 255) / 256 : 256);
  hipStream_t s = (hipStream_t)stream;
  if (in_dtype == MMDIT_BF16)
    hipLaunchKernelGGL((flow_loss_partial_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, v, (const bf16_t*)x0, (const bf16_t*)eps, n8, 2.f * coef, dv, partials);
  else if (in_dtype == MMDIT_F32)
    hipLaunchKernelGGL((flow_loss_partial_kernel<float>), dim3(grid), dim3(256), 0, s, v, (const float*)x0, (const float*)eps, n8, 2.f * coef, dv, partials);
  else
    return MMDIT_ERR_DTYPE;
  hipLaunchKernelGGL(flow_loss_final_kernel, dim3(1), dim3(64), 0, s, (const float*)partials, grid, coef, loss);
  return mmdit_launch_status();
}

extern "C" int mmdit_colsum(const void* x, int dtype, int rows, int cols, int64_t ld, float* out, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(x && out && rows > 0 && cols > 0 && cols % 8 == 0 && ld % 8 == 0);
  hipStream_t s = (hipStream_t)stream;
  // few rows (the per-sample partial sums of a block: 64 x 2d): 8-row chunks, so that the loads of a thread are all in flight at once --
  // the 64-row chunk walked them as one latency-bound loop in two workgroups (10-17 us for 400 KB)
  const int rch = rows <= 256 ? 8 : CO_RCH;
  dim3 grid((cols + 1023) / 1024, (rows + rch - 1) / rch);
  if (dtype == MMDIT_BF16) hipLaunchKernelGGL((colsum_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)x, rows, cols, ld, out, rch);
  else if (dtype == MMDIT_F32) hipLaunchKernelGGL((colsum_kernel<float>), grid, dim3(256), 0, s, (const float*)x, rows, cols, ld, out, rch);
  else return MMDIT_ERR_DTYPE;
  return mmdit_launch_status();
}

namespace {
template <typename TI>
__global__ __launch_bounds__(256) void fp8_amax_kernel(const TI* __restrict__ x, int64_t n8, float* __restrict__ amax) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    float v[8];
    ld8(x + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; e++) m = fmaxf(m, fabsf(v[e]));
  }
  __shared__ float sm[4];
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  // one atomic per workgroup (they all hit the same address); non-negative floats order like their bit patterns
  if (threadIdx.x == 0) atomicMax((unsigned int*)amax, __float_as_uint(fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]))));
}
template <typename TI>
__global__ __launch_bounds__(256) void fp8_quant_kernel(const TI* __restrict__ x, int64_t n8, const float* __restrict__ amax, uint2* __restrict__ q, float* __restrict__ scale) {
  const float a = fmaxf(amax[0], 1e-12f), s = 448.f / a;
  if (blockIdx.x == 0 && threadIdx.x == 0) scale[0] = a / 448.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    float v[8];
    ld8(x + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = fminf(fmaxf(v[e] * s, -448.f), 448.f);
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
    q[i] = make_uint2((unsigned)lo, (unsigned)hi);
  }
}
// Delayed scaling (one pass): quantise with the amax of the PREVIOUS call at this call site (x margin) while collecting this
// call's amax for the next one.  state = {amax[0], amax[1], amax[2], scale}: call k reads amax[k % 3], maximises into
// amax[(k+1) % 3] and clears amax[(k+2) % 3] (the target of call k+1, last read by call k-1) -- stream order makes that race-free.
template <typename TI>
__global__ __launch_bounds__(256) void fp8_quant_delayed_kernel(const TI* __restrict__ x, int64_t n8, float* __restrict__ state, int phase, float margin,
                                                                uint2* __restrict__ q) {
  __shared__ float sm[4];
  const float a = fmaxf(state[phase % 3] * margin, 1e-12f), s = 448.f / a;
  if (blockIdx.x == 0 && threadIdx.x == 0) { state[3] = a / 448.f; state[(phase + 2) % 3] = 0.f; }
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    float v[8];
    ld8(x + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; e++) { m = fmaxf(m, fabsf(v[e])); v[e] = fminf(fmaxf(v[e] * s, -448.f), 448.f); }
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
    q[i] = make_uint2((unsigned)lo, (unsigned)hi);
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax((unsigned int*)(state + (phase + 1) % 3), __float_as_uint(fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]))));
}
}  // namespace

namespace {
// MX quantisation: one thread per 32-element block (64 B of bf16 / 128 B of fp32 in, 32 B + 1 scale byte out); consecutive
// threads take consecutive blocks of a row, so the loads and the e4m3 stores of a wave are contiguous.
template <typename TI>
__global__ __launch_bounds__(256) void mxfp8_quant_kernel(const TI* __restrict__ x, int rows, int K, int64_t ldx, uint4* __restrict__ q, unsigned char* __restrict__ sc) {
  const int bpr = K >> 5;                                             // blocks per row
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)rows * bpr) return;
  const int row = (int)(t / bpr), blk = (int)(t - (int64_t)row * bpr);
  const TI* src = x + (int64_t)row * ldx + blk * 32;
  float v[32];
#pragma unroll
  for (int c = 0; c < 4; c++) ld8(src + c * 8, *(float(*)[8])(v + c * 8));
  float amax = 0.f;
#pragma unroll
  for (int e = 0; e < 32; e++) amax = fmaxf(amax, fabsf(v[e]));
  float inv;
  const int ex = mx_exponent(amax, inv);
  unsigned w[8];
#pragma unroll
  for (int c = 0; c < 8; c++) w[c] = mx_pack4(v + 4 * c, inv);
  uint4* dst = q + ((int64_t)row * K + blk * 32) / 16;
  dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
  dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
  sc[mx_scale_index(row, blk, rows)] = (unsigned char)(ex + 127);
}
}  // namespace

extern "C" int mmdit_mxfp8_quantize(const void* x, int x_dtype, int rows, int K, int64_t ldx, void* q_fp8, void* scales, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(x && q_fp8 && scales && rows > 0 && K > 0 && K % 64 == 0 && ldx >= K && ldx % 8 == 0);
  hipStream_t s = (hipStream_t)stream;
  const int64_t nblk = (int64_t)rows * (K / 32);
  const dim3 grid((unsigned)((nblk + 255) / 256));
  if (x_dtype == MMDIT_F32) hipLaunchKernelGGL((mxfp8_quant_kernel<float>), grid, dim3(256), 0, s, (const float*)x, rows, K, ldx, (uint4*)q_fp8, (unsigned char*)scales);
  else if (x_dtype == MMDIT_BF16) hipLaunchKernelGGL((mxfp8_quant_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)x, rows, K, ldx, (uint4*)q_fp8, (unsigned char*)scales);
  else return MMDIT_ERR_DTYPE;
  return mmdit_launch_status();
}

extern "C" int mmdit_fp8_quantize_delayed(const void* x, int x_dtype, int64_t n, float* state, int phase, float margin, void* q_fp8, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(x && state && q_fp8 && n > 0 && n % 8 == 0 && phase >= 0 && margin >= 1.f);
  hipStream_t s = (hipStream_t)stream;
  const int g0 = grid_cap(n / 8, 256);
  dim3 grid(g0 > 1024 ? 1024 : g0);
  if (x_dtype == MMDIT_F32) hipLaunchKernelGGL((fp8_quant_delayed_kernel<float>), grid, dim3(256), 0, s, (const float*)x, n / 8, state, phase, margin, (uint2*)q_fp8);
  else if (x_dtype == MMDIT_BF16) hipLaunchKernelGGL((fp8_quant_delayed_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)x, n / 8, state, phase, margin, (uint2*)q_fp8);
  else return MMDIT_ERR_DTYPE;
  return mmdit_launch_status();
}

extern "C" int mmdit_fp8_amax(const void* x, int x_dtype, int64_t n, float* amax, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(x && amax && n > 0 && n % 8 == 0);
  hipStream_t s = (hipStream_t)stream;
  const int g0 = grid_cap(n / 8, 256);
  dim3 grid(g0 > 1024 ? 1024 : g0);
  if (x_dtype == MMDIT_F32) hipLaunchKernelGGL((fp8_amax_kernel<float>), grid, dim3(256), 0, s, (const float*)x, n / 8, amax);
  else if (x_dtype == MMDIT_BF16) hipLaunchKernelGGL((fp8_amax_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)x, n / 8, amax);
  else return MMDIT_ERR_DTYPE;
  return mmdit_launch_status();
}
extern "C" int mmdit_fp8_quantize(const void* x, int x_dtype, int64_t n, const float* amax, void* q_fp8, float* scale, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(x && amax && q_fp8 && scale && n > 0 && n % 8 == 0);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(grid_cap(n / 8, 256));
  if (x_dtype == MMDIT_F32) hipLaunchKernelGGL((fp8_quant_kernel<float>), grid, dim3(256), 0, s, (const float*)x, n / 8, amax, (uint2*)q_fp8, scale);
  else if (x_dtype == MMDIT_BF16) hipLaunchKernelGGL((fp8_quant_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)x, n / 8, amax, (uint2*)q_fp8, scale);
  else return MMDIT_ERR_DTYPE;
  return mmdit_launch_status();
}

extern "C" int mmdit_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(src && dst && n > 0);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(grid_cap(n / 8 + 1, 256));
  if (src_dtype == MMDIT_F32 && dst_dtype == MMDIT_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), grid, dim3(256), 0, s, (const float*)src, (bf16_t*)dst, n);
  else if (src_dtype == MMDIT_BF16 && dst_dtype == MMDIT_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), grid, dim3(256), 0, s, (const bf16_t*)src, (float*)dst, n);
  else return MMDIT_ERR_DTYPE;
  return mmdit_launch_status();
}

template <bool TO_TOKENS>
static int patch_dispatch(const void* src, int sdt, void* dst, int ddt, int batch, int ch, int H, int W, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(src && dst && batch > 0 && ch > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(grid_cap((int64_t)batch * ch * H * (W / 2), 256));
#define PK(TI, TO) hipLaunchKernelGGL((patch_kernel<TI, TO, TO_TOKENS>), grid, dim3(256), 0, s, (const TI*)src, (TO*)dst, batch, ch, H, W)
  if (sdt == MMDIT_F32 && ddt == MMDIT_F32) PK(float, float);
  else if (sdt == MMDIT_F32 && ddt == MMDIT_BF16) PK(float, bf16_t);
  else if (sdt == MMDIT_BF16 && ddt == MMDIT_BF16) PK(bf16_t, bf16_t);
  else if (sdt == MMDIT_BF16 && ddt == MMDIT_F32) PK(bf16_t, float);
  else return MMDIT_ERR_DTYPE;
#undef PK
  return mmdit_launch_status();
}
extern "C" int mmdit_patchify(const void* img, int img_dtype, int batch, int ch, int H, int W, void* tokens, int tok_dtype, mmdit_stream_t st) {
  return patch_dispatch<true>(img, img_dtype, tokens, tok_dtype, batch, ch, H, W, st);
}
extern "C" int mmdit_unpatchify(const void* tokens, int tok_dtype, int batch, int ch, int H, int W, void* img, int img_dtype, mmdit_stream_t st) {
  return patch_dispatch<false>(tokens, tok_dtype, img, img_dtype, batch, ch, H, W, st);
}

extern "C" int mmdit_time_embed_fwd(const float* t, const float* time_scale, const float* denom, int batch, int dim, void* out, int out_dtype, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(t && time_scale && denom && out && batch > 0 && dim > 0 && dim % 2 == 0);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((dim + 255) / 256, batch);
  if (out_dtype == MMDIT_BF16) hipLaunchKernelGGL((time_embed_fwd_kernel<bf16_t>), grid, dim3(256), 0, s, t, time_scale, denom, batch, dim, (bf16_t*)out);
  else if (out_dtype == MMDIT_F32) hipLaunchKernelGGL((time_embed_fwd_kernel<float>), grid, dim3(256), 0, s, t, time_scale, denom, batch, dim, (float*)out);
  else return MMDIT_ERR_DTYPE;
  return mmdit_launch_status();
}
extern "C" int mmdit_time_embed_bwd(const void* dout, int dout_dtype, const float* t, const float* time_scale, const float* denom, int batch, int dim, float* dtime_scale, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(dout && t && time_scale && denom && dtime_scale && batch > 0 && dim > 0 && dim % 2 == 0);
  hipStream_t s = (hipStream_t)stream;
  if (dout_dtype == MMDIT_BF16) hipLaunchKernelGGL((time_embed_bwd_kernel<bf16_t>), dim3(batch), dim3(256), 0, s, (const bf16_t*)dout, t, time_scale, denom, batch, dim, dtime_scale);
  else if (dout_dtype == MMDIT_F32) hipLaunchKernelGGL((time_embed_bwd_kernel<float>), dim3(batch), dim3(256), 0, s, (const float*)dout, t, time_scale, denom, batch, dim, dtime_scale);
  else return MMDIT_ERR_DTYPE;
  return mmdit_launch_status();
}
