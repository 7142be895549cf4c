#!/bin/bash
# Build a scratch variant of the HIP library with extra compiler flags for same-box A/B runs:
#   bash tools/build_variant.sh prio -DMMDIT_STATIC_PRIO     ->  tools/scratch/prio/libmmdit_hip.so   (select with MMDIT_LIB=...)
# The scratch directory is git-ignored but travels to the GPU box with gpurun.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=tools/scratch/$name
mkdir -p $out
for f in gemm gemm_dma gemm_lean gemm8p gemm8p_inf rowops attention vae optim; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c stable-diffusion-3-from-scratch_amd/csrc/$f.hip -o $out/$f.o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libmmdit_hip.so $out/*.o
rm -f $out/*.o
ls -la $out/libmmdit_hip.so
