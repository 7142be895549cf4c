// Shared device helpers for the gfx950 kernels (wave = 64 lanes, MFMA 32x32x16 bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mmdit_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef uint16_t bf16_t;  // raw storage type for bf16 in global memory

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

// round-to-nearest-even fp32 -> bf16 on the hardware converter (v_cvt_pk_bf16_f32: one instruction per pair)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ float bf2f(bf16_t h) { return __builtin_bit_cast(float, ((uint32_t)h) << 16); }
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  bf16x2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // raw v_exp_f32

// generic scalar load/store by dtype tag
template <typename T> struct io;
template <> struct io<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct io<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// load / store 8 consecutive elements as fp32 (16 B bf16 vector or 2 x 16 B fp32 vectors)
__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
  float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void ld8(const bf16_t* p, float (&v)[8]) {
  uint4 a = *(const uint4*)p;
  uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
  for (int i = 0; i < 4; i++) { v[2 * i] = __builtin_bit_cast(float, w[i] << 16); v[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u); }
}
// streaming (non-temporal) variants for data at its LAST use (saved activations read by a backward kernel): no point keeping it in L2 / MALL
__device__ __forceinline__ void ld8_nt(const bf16_t* p, float (&v)[8]) {
  const u32x4 a = __builtin_nontemporal_load((const u32x4*)p);
#pragma unroll
  for (int i = 0; i < 4; i++) { v[2 * i] = __builtin_bit_cast(float, a[i] << 16); v[2 * i + 1] = __builtin_bit_cast(float, a[i] & 0xffff0000u); }
}
__device__ __forceinline__ void ld8_nt(const float* p, float (&v)[8]) { ld8(p, v); }
__device__ __forceinline__ void ld4_nt(const float* p, float (&v)[4]) { const f32x4 a = __builtin_nontemporal_load((const f32x4*)p); v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; }
__device__ __forceinline__ void ld4_nt(const bf16_t* p, float (&v)[4]) {
  const u32x2 a = __builtin_nontemporal_load((const u32x2*)p);
  v[0] = __builtin_bit_cast(float, a[0] << 16); v[1] = __builtin_bit_cast(float, a[0] & 0xffff0000u);
  v[2] = __builtin_bit_cast(float, a[1] << 16); v[3] = __builtin_bit_cast(float, a[1] & 0xffff0000u);
}
__device__ __forceinline__ void st8(float* p, const float (&v)[8]) {
  *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
  *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void st8(bf16_t* p, const float (&v)[8]) {
  *(uint4*)p = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
}
// A 16-byte global store the compiler does NOT count (inline asm).  gfx9 retires loads and stores of a wave through one in-order counter (vmcnt) and
// the compiler keeps its model of that queue exact only along straight-line code: a store inside `if (row < M)` makes the number of requests in
// flight unknown at the join, and every later wait for a LOAD then degrades to vmcnt(0) -- it waits for the stores just issued.  With the stores
// of an epilogue pass hidden, the compiler counts the (unconditional) loads alone: vmcnt(n) with n = the loads issued since can only wait for MORE
// than the load it is after (the hidden stores sit in the same queue), never for less -- safe -- and in practice it waits for stores that are at
// least a pass old.  Store data is read from the registers when the instruction issues (no wait before they are overwritten on gfx9).
__device__ __forceinline__ void st16_uncounted(void* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st8_uncounted(bf16_t* p, const float (&v)[8]) {
  st16_uncounted(p, (u32x4){pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])});
}

__device__ __forceinline__ void ld4(const float* p, float (&v)[4]) { float4 a = *(const float4*)p; v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; }
__device__ __forceinline__ void ld4(const bf16_t* p, float (&v)[4]) {
  uint2 a = *(const uint2*)p;
  v[0] = __builtin_bit_cast(float, a.x << 16); v[1] = __builtin_bit_cast(float, a.x & 0xffff0000u);
  v[2] = __builtin_bit_cast(float, a.y << 16); v[3] = __builtin_bit_cast(float, a.y & 0xffff0000u);
}
__device__ __forceinline__ void st4(float* p, const float (&v)[4]) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void st4(bf16_t* p, const float (&v)[4]) { *(uint2*)p = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])); }

// sum over the 8 lanes of a row (lane bits 2:0) on the DPP path (quad_perm 1032, quad_perm 2301, row_half_mirror): three dependent
// VALU moves instead of three LDS round trips (ds_bpermute) -- the epilogue is a latency chain, not a throughput problem
template <int CTRL> __device__ __forceinline__ float dpp_f(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float sum8(float x) {
  x += dpp_f<0xB1>(x);
  x += dpp_f<0x4E>(x);
  return x + dpp_f<0x141>(x);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// v_rcp_f32 (1 ulp) instead of the IEEE division sequence (a dozen VALU instructions per element: the SwiGLU GEMM epilogue evaluates
// 64 of these per lane and tile); __expf is the hardware exponential already, so the quotient was never correctly rounded anyway
#ifdef MMDIT_PRECISE_PROBE   // experiment: correctly rounded transcendentals everywhere (does the parity margin depend on them?)
#define rsqrtf(x) (1.0f / sqrtf(x))
#define __expf(x) expf(x)
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }
#else
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
#endif
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
// backward of h = silu(g) * u for one element: dg = d u s (1 + g (1 - s)), du = d g s, s = sigmoid(g).  ONE spelling (explicit fma) for the row
// kernel (rowops.hip mlp_act_bwd_body) and the GEMM epilogue (gemm8p.hip epi8_swiglu_bwd): the two paths must produce the same bits, and the
// compiler's choice to contract 1 + g (1 - s) depends on the surrounding code
__device__ __forceinline__ void swiglu_bwd_f(float d, float g, float u, float& dg, float& du) {
  const float s = sigmoid_f(g);
  dg = d * u * s * __builtin_fmaf(g, 1.f - s, 1.f);
  du = d * g * s;
}

// ---- MX (OCP microscaling) e4m3 helpers ------------------------------------------------------------------------------------
// E8M0 exponent of a 32-block with absolute maximum amax: the smallest power of two with amax / 2^ex <= 448 (floor(log2 amax) - 8
// from the float's exponent field, plus one when the mantissa exceeds 448 / 256; zero / denormal amax -> 2^-127), and 2^-ex.
__device__ __forceinline__ int mx_exponent(float amax, float& inv) {
  int ex = (int)((__float_as_uint(amax) >> 23) & 0xff) - 127 - 8;
  if (amax > 448.f * __uint_as_float((unsigned)(ex + 127 > 0 ? ex + 127 : 0) << 23)) ex++;
  ex = ex < -127 ? -127 : (ex > 127 ? 127 : ex);
  const unsigned f = (unsigned)(127 - ex) << 23;
  inv = __uint_as_float(f ? f : 0x00400000u);   // (ex = 127: 2^-127 is a denormal)
  return ex;
}
__device__ __forceinline__ unsigned mx_pack4(const float* v, float inv) {
  const float a0 = fminf(fmaxf(v[0] * inv, -448.f), 448.f), a1 = fminf(fmaxf(v[1] * inv, -448.f), 448.f);
  const float a2 = fminf(fmaxf(v[2] * inv, -448.f), 448.f), a3 = fminf(fmaxf(v[3] * inv, -448.f), 448.f);
  int pk = 0;
  pk = __builtin_amdgcn_cvt_pk_fp8_f32(a0, a1, pk, false);
  pk = __builtin_amdgcn_cvt_pk_fp8_f32(a2, a3, pk, true);
  return (unsigned)pk;
}
// Scale byte of (row, 32-block blk of K) in the layout the GEMM reads (mmdit_gemm_args.scale_mode 1): for every 64-wide K half the
// rows in groups of 128, inside a group the byte of (row, half-block h = blk & 1) at (row & 31) * 8 + h * 4 + ((row >> 5) & 3) --
// so the E8M0 bytes of the four 32-row blocks a wave's fragments cover sit in ONE dword per lane (selected by op_sel in the MFMA).
// rows are padded to a multiple of 128: buffer = (K / 64) * rows_pad * 2 + 512 bytes.
__device__ __forceinline__ int64_t mx_scale_index(int row, int blk, int rows) {
  const int64_t rows_pad = ((int64_t)rows + 127) & ~(int64_t)127;
  return ((int64_t)(blk >> 1) * rows_pad + (row & ~127)) * 2 + (row & 31) * 8 + (blk & 1) * 4 + ((row >> 5) & 3);
}
// transpose read: lane gets 4 bf16 = column (lane&15) of the 4x16 block whose rows are addressed by the 16-lane group
__device__ __forceinline__ s16x4 lds_tr16(const void* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, p)); }

// Static priority for the second-dispatched half of an 8-wave workgroup (MI355X_MICROARCH.md "Two waves per SIMD", item 4): the two waves
// of a SIMD arbitrate VALU issue by priority, then age, so waves 4-7 lose every contended slot; ONE s_setprio 1 for that half, no
// per-segment flips.  The condition must be provably wave-uniform (s_setprio ignores EXEC).  Experiment switch: -DMMDIT_STATIC_PRIO.
#ifdef MMDIT_STATIC_PRIO
#define MMDIT_YOUNG_HALF_PRIO() do { if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1); } while (0)
#else
#define MMDIT_YOUNG_HALF_PRIO() do { } while (0)
#endif

// Experiment switch -DMMDIT_MFMA_PRIO: raise the wave's priority around each MFMA row of the GEMM main loops (cdna_hip_programming.md T5:
// +21-25 % on a phase-split schedule, ~0 on a lockstep one).
#ifdef MMDIT_MFMA_PRIO
#define MMDIT_PRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define MMDIT_PRIO(x) do { } while (0)
#endif

int mmdit_device_cus();                         // gemm.hip: compute units of the current device (a multiple of 8; 256 when no device answers)
int* mmdit_gemm_sched_slot();                   // gemm.hip: the next 16-int slot of the registered workspace's scheduler page (nullptr: no workspace -> static tile walk)
extern "C" int mmdit_get_cu_budget(void);      // compute units the persistent GEMM launches may count on (gemm.hip; include/mmdit_hip.h mmdit_set_cu_budget)
static inline int mmdit_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
#define MMDIT_CHECK_ARG(c) do { if (!(c)) return MMDIT_ERR_ARG; } while (0)

// Experiment switches (tile configuration, schedule and ablation overrides of the GEMM / row kernels: the MMDIT_GEMM_* / MMDIT_QK_* variables of
// tools/README.md) exist only in -DMMDIT_PROBES builds (tools/build_variant.sh); the product library has ONE path per launch.
#include <stdlib.h>
static inline const char* mmdit_exp_env(const char* name) {
#ifdef MMDIT_PROBES
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

// Per-device "done once" cache for host-side launch set-up (hipFuncSetAttribute of > 64 KB dynamic LDS is a per-device property; one
// process normally drives one GPU, but a second device must not inherit the first one's flag).  Returns true when `done` already
// holds the current device's bit; mark with mmdit_device_mark.  A benign race only repeats the idempotent call.
static inline int mmdit_current_device() { int d = 0; (void)hipGetDevice(&d); return d < 0 ? 0 : (d > 63 ? 63 : d); }
static inline bool mmdit_device_once(const unsigned long long& done) { return (done >> mmdit_current_device()) & 1ull; }
static inline void mmdit_device_mark(unsigned long long& done) { done |= 1ull << mmdit_current_device(); }
