// Optimizer step for gfx950 -- SURVEY 8(f) row 4: GradScaler.unscale_ + clip_grad_norm_ + AdamW (reference
// model_trainer.py:260-269 builds the optimizer, 463-503 runs unscale_/clip/step/update) as three launches over the
// whole parameter list instead of ~50 multi-tensor launches in three passes.  Everything here is HBM-bound: the
// gradients are read twice (norm, update), parameters and both moments once, 28 B of traffic per parameter for the update.
//
// The parameter list is described by a DEVICE table of mmdit_adamw_tensor plus a chunk map (chunk c = elements
// [chunk_off[c], chunk_off[c] + CHUNK) of tensor chunk_tensor[c]); one workgroup per chunk.  The squared-norm partials are
// written per chunk and reduced by one workgroup in a fixed order (deterministic, no atomics).
#include "common.h"

#include "../../include/mmdit_hip.h"

namespace {

constexpr int CHUNK = MMDIT_ADAMW_CHUNK, TPB = 256;

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < TPB / 64; w++) t += red[w];
  return t;
}

__global__ __launch_bounds__(TPB) void grad_sumsq_kernel(const mmdit_adamw_tensor* __restrict__ tensors, const int* __restrict__ chunk_tensor,
                                                          const int64_t* __restrict__ chunk_off, float* __restrict__ partials) {
  __shared__ float red[TPB / 64];
  const int c = blockIdx.x;
  const mmdit_adamw_tensor t = tensors[chunk_tensor[c]];
  const int64_t off = chunk_off[c];
  const int n = (int)min((int64_t)CHUNK, t.numel - off);
  const float* g = t.grad + off;
  float a0 = 0.f, a1 = 0.f;
  if (((uintptr_t)g & 15) == 0) {
    const int n4 = n >> 2;
    int i = threadIdx.x;
    for (; i + TPB < n4; i += 2 * TPB) {   // two 16-byte loads in flight per thread
      const float4 x = ((const float4*)g)[i], y = ((const float4*)g)[i + TPB];
      a0 += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
      a1 += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
    }
    if (i < n4) {
      const float4 x = ((const float4*)g)[i];
      a0 += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
    }
    for (int j = (n4 << 2) + threadIdx.x; j < n; j += TPB) a1 += g[j] * g[j];
  } else {
    for (int j = threadIdx.x; j < n; j += TPB) a0 += g[j] * g[j];
  }
  const float s = block_sum(a0 + a1, red);
  if (threadIdx.x == 0) partials[c] = s;
}

// out[0] = multiplier for the stored gradients = inv_scale * min(1, max_norm / (||g|| * inv_scale + 1e-6))   (unscale_ then
// clip_grad_norm_, torch/nn/utils/clip_grad.py: clip_coef clamped to 1);  out[1] = found_inf (1.0 when the norm is not
// finite -- an inf/nan gradient anywhere, which is GradScaler.unscale_'s test);  out[2] = the unscaled gradient norm.
__global__ __launch_bounds__(TPB) void clip_coef_kernel(const float* __restrict__ partials, int n, const float* __restrict__ loss_scale, float max_norm, float* __restrict__ out) {
  __shared__ double red[TPB];
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += TPB) a += (double)partials[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int s = TPB / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float total = (float)sqrt(red[0]);
    const float inv_scale = loss_scale ? (float)(1.0 / (double)loss_scale[0]) : 1.0f;
    const float norm = total * inv_scale;
    const float clip = max_norm > 0.f ? fminf(max_norm / (norm + 1e-6f), 1.0f) : 1.0f;
    const bool finite = isfinite(total);
    out[0] = inv_scale * clip;
    out[1] = finite ? 0.f : 1.f;
    out[2] = norm;
  }
}

struct AdamConst {
  float decay;        // 1 - lr * weight_decay
  float one_m_b1;     // 1 - beta1
  float b2, one_m_b2;
  float eps;
  double lr, b1d, b2d, wd;
  const double* lr_dev;   // not NULL: the learning rate is read from device memory (a captured launch must not bake it in)
};

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamConst& k, float step_size, float bc2_sqrt) {
  // torch fused AdamW (ATen fused_adam_utils.cuh, ADAMW mode): decoupled decay, lerp'ed first moment, bias-corrected step
  p *= k.decay;
  m = m + k.one_m_b1 * (g - m);
  v = k.b2 * v + k.one_m_b2 * g * g;
  const float denom = sqrtf(v) / bc2_sqrt + k.eps;
  p -= step_size * m / denom;
}

typedef __attribute__((ext_vector_type(4))) float f32x4_nt;
__device__ __forceinline__ float4 ntl4(const float* p, int i) { const f32x4_nt v = __builtin_nontemporal_load((const f32x4_nt*)p + i); return make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void nts4(float* p, int i, const float4& v) { __builtin_nontemporal_store(f32x4_nt{v.x, v.y, v.z, v.w}, (f32x4_nt*)p + i); }
#define NTL(P_, i_) ntl4(P_, i_)
#define NTS(P_, i_, v_) nts4(P_, i_, v_)

__global__ __launch_bounds__(TPB) void adamw_kernel(const mmdit_adamw_tensor* __restrict__ tensors, const int* __restrict__ chunk_tensor, const int64_t* __restrict__ chunk_off,
                                                     const float* __restrict__ coef_found, const float* __restrict__ step_count, AdamConst k) {   // (k is a by-value copy: lr / decay may be replaced below)
  float coef = 1.f;
  if (coef_found) {
    if (coef_found[1] != 0.f) return;   // inf/nan gradients: the whole step is skipped (GradScaler.step)
    coef = coef_found[0];
  }
  const double step = (double)step_count[0] + 1.0;
  if (k.lr_dev) {
    k.lr = k.lr_dev[0];
    k.decay = (float)(1.0 - k.lr * k.wd);
  }
  const float step_size = (float)(k.lr / (1.0 - pow(k.b1d, step)));
  const float bc2_sqrt = (float)sqrt(1.0 - pow(k.b2d, step));
  const int c = blockIdx.x;
  const mmdit_adamw_tensor t = tensors[chunk_tensor[c]];
  const int64_t off = chunk_off[c];
  const int n = (int)min((int64_t)CHUNK, t.numel - off);
  float* P = t.param + off;
  const float* G = t.grad + off;
  float* M = t.exp_avg + off;
  float* V = t.exp_avg_sq + off;
  bf16_t* S = t.shadow_bf16 ? (bf16_t*)t.shadow_bf16 + off : nullptr;
  int done = 0;
  if ((((uintptr_t)P | (uintptr_t)G | (uintptr_t)M | (uintptr_t)V) & 15) == 0 && ((uintptr_t)S & 7) == 0) {
    const int n4 = n >> 2;
    for (int i = threadIdx.x; i < n4; i += 2 * TPB) {
      const bool two = i + TPB < n4;
      const int i2 = two ? i + TPB : i;
      // every byte of the 10 GB this kernel moves is touched exactly once per step: streaming (non-temporal) loads and stores keep it
      // out of the way of what the next kernels want in L2 / Infinity Cache
      float4 p0 = NTL(P, i), g0 = NTL(G, i), m0 = NTL(M, i), v0 = NTL(V, i);
      float4 p1 = NTL(P, i2), g1 = NTL(G, i2), m1 = NTL(M, i2), v1 = NTL(V, i2);
      adam1(p0.x, g0.x * coef, m0.x, v0.x, k, step_size, bc2_sqrt);
      adam1(p0.y, g0.y * coef, m0.y, v0.y, k, step_size, bc2_sqrt);
      adam1(p0.z, g0.z * coef, m0.z, v0.z, k, step_size, bc2_sqrt);
      adam1(p0.w, g0.w * coef, m0.w, v0.w, k, step_size, bc2_sqrt);
      NTS(P, i, p0); NTS(M, i, m0); NTS(V, i, v0);
      if (S) __builtin_nontemporal_store(u32x2{pack_bf2(p0.x, p0.y), pack_bf2(p0.z, p0.w)}, (u32x2*)S + i);
      if (two) {
        adam1(p1.x, g1.x * coef, m1.x, v1.x, k, step_size, bc2_sqrt);
        adam1(p1.y, g1.y * coef, m1.y, v1.y, k, step_size, bc2_sqrt);
        adam1(p1.z, g1.z * coef, m1.z, v1.z, k, step_size, bc2_sqrt);
        adam1(p1.w, g1.w * coef, m1.w, v1.w, k, step_size, bc2_sqrt);
        NTS(P, i2, p1); NTS(M, i2, m1); NTS(V, i2, v1);
        if (S) __builtin_nontemporal_store(u32x2{pack_bf2(p1.x, p1.y), pack_bf2(p1.z, p1.w)}, (u32x2*)S + i2);
      }
    }
    done = n4 << 2;
  }
  for (int j = done + threadIdx.x; j < n; j += TPB) {
    float p = P[j], m = M[j], v = V[j];
    adam1(p, G[j] * coef, m, v, k, step_size, bc2_sqrt);
    P[j] = p; M[j] = m; V[j] = v;
    if (S) S[j] = f2bf(p);
  }
}

// fp32 master weights -> bf16 operand copies, every tensor of the model in one launch (same chunk map scheme)
__global__ __launch_bounds__(TPB) void cast_multi_kernel(const mmdit_cast_tensor* __restrict__ tensors, const int* __restrict__ chunk_tensor,
                                                          const int64_t* __restrict__ chunk_off) {
  const int c = blockIdx.x;
  const mmdit_cast_tensor t = tensors[chunk_tensor[c]];
  const int64_t off = chunk_off[c];
  const int n = (int)min((int64_t)CHUNK, t.numel - off);
  const float* s = t.src + off;
  bf16_t* d = (bf16_t*)t.dst + off;
  int done = 0;
  if ((((uintptr_t)s | (uintptr_t)d) & 15) == 0) {
    const int n8 = n >> 3;
    for (int i = threadIdx.x; i < n8; i += TPB) {
      const float4 a = ((const float4*)s)[2 * i], b = ((const float4*)s)[2 * i + 1];
      ((uint4*)d)[i] = make_uint4(pack_bf2(a.x, a.y), pack_bf2(a.z, a.w), pack_bf2(b.x, b.y), pack_bf2(b.z, b.w));
    }
    done = n8 << 3;
  }
  for (int j = done + threadIdx.x; j < n; j += TPB) d[j] = (bf16_t)(pack_bf2(s[j], 0.f) & 0xffffu);
}

}  // namespace

extern "C" int mmdit_cast_multi(const mmdit_cast_tensor* tensors, const int* chunk_tensor, const int64_t* chunk_off, int n_chunks, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(tensors && chunk_tensor && chunk_off && n_chunks > 0);
  hipLaunchKernelGGL(cast_multi_kernel, dim3(n_chunks), dim3(TPB), 0, (hipStream_t)stream, tensors, chunk_tensor, chunk_off);
  return mmdit_launch_status();
}

extern "C" int mmdit_grad_sumsq(const mmdit_adamw_tensor* tensors, const int* chunk_tensor, const int64_t* chunk_off, int n_chunks, float* partials, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(tensors && chunk_tensor && chunk_off && partials && n_chunks > 0);
  hipLaunchKernelGGL(grad_sumsq_kernel, dim3(n_chunks), dim3(TPB), 0, (hipStream_t)stream, tensors, chunk_tensor, chunk_off, partials);
  return mmdit_launch_status();
}

extern "C" int mmdit_clip_coef(const float* partials, int n_chunks, const float* loss_scale, float max_norm, float* out3, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(partials && out3 && n_chunks > 0);
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(TPB), 0, (hipStream_t)stream, partials, n_chunks, loss_scale, max_norm, out3);
  return mmdit_launch_status();
}

extern "C" int mmdit_adamw_step(const mmdit_adamw_tensor* tensors, const int* chunk_tensor, const int64_t* chunk_off, int n_chunks, const float* coef_found,
                                const float* step_count, double lr, double beta1, double beta2, double eps, double weight_decay, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(tensors && chunk_tensor && chunk_off && step_count && n_chunks > 0);
  MMDIT_CHECK_ARG(lr >= 0.0 && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0 && weight_decay >= 0.0);
  AdamConst k;
  k.decay = (float)(1.0 - lr * weight_decay);
  k.one_m_b1 = (float)(1.0 - beta1);
  k.b2 = (float)beta2;
  k.one_m_b2 = (float)(1.0 - beta2);
  k.eps = (float)eps;
  k.lr = lr; k.b1d = beta1; k.b2d = beta2; k.wd = weight_decay; k.lr_dev = nullptr;
  hipLaunchKernelGGL(adamw_kernel, dim3(n_chunks), dim3(TPB), 0, (hipStream_t)stream, tensors, chunk_tensor, chunk_off, coef_found, step_count, k);
  return mmdit_launch_status();
}

extern "C" int mmdit_adamw_step_dlr(const mmdit_adamw_tensor* tensors, const int* chunk_tensor, const int64_t* chunk_off, int n_chunks, const float* coef_found,
                                    const float* step_count, const double* lr_dev, double beta1, double beta2, double eps, double weight_decay, mmdit_stream_t stream) {
  MMDIT_CHECK_ARG(tensors && chunk_tensor && chunk_off && step_count && lr_dev && n_chunks > 0);
  MMDIT_CHECK_ARG(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0 && weight_decay >= 0.0);
  AdamConst k;
  k.decay = 1.f;
  k.one_m_b1 = (float)(1.0 - beta1);
  k.b2 = (float)beta2;
  k.one_m_b2 = (float)(1.0 - beta2);
  k.eps = (float)eps;
  k.lr = 0.0; k.b1d = beta1; k.b2d = beta2; k.wd = weight_decay; k.lr_dev = lr_dev;
  hipLaunchKernelGGL(adamw_kernel, dim3(n_chunks), dim3(TPB), 0, (hipStream_t)stream, tensors, chunk_tensor, chunk_off, coef_found, step_count, k);
  return mmdit_launch_status();
}
