// Second translation unit of the 8-phase GEMM (csrc/gemm8p.hip): the e4m3-operand (MX / per-tensor) and implicit-GEMM convolution instantiations of the
// inference and VAE paths, compiled in parallel with the training kernels.  All code lives in gemm8p.hip.
#define MMDIT_G8_PART 2
#include "gemm8p.hip"
