// Stand-in for a long-running collective: `wgs` small workgroups (one wave, 1 KiB of LDS -- enough that a 160 KiB GEMM workgroup cannot share the CU) spin for
// `cycles` shader cycles on the given stream.  tools/probes/cu_contention.py runs the training step beside it.
// hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o liboccupy.so occupy.hip
#include <hip/hip_runtime.h>
__global__ void occupy_kernel(long long cycles, int* sink) {
  __shared__ int pad[256];
  pad[threadIdx.x & 255] = threadIdx.x;
  const long long t0 = __builtin_readcyclecounter();
  while ((long long)__builtin_readcyclecounter() - t0 < cycles) __builtin_amdgcn_s_sleep(32);
  if (cycles < 0) sink[0] = pad[0];
}
extern "C" int occupy(int wgs, long long cycles, void* stream) {
  hipLaunchKernelGGL(occupy_kernel, dim3(wgs), dim3(64), 0, (hipStream_t)stream, cycles, (int*)nullptr);
  return (int)hipGetLastError();
}
