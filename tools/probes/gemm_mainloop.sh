#!/bin/bash
# Steady-state main-loop rates of the GEMM kernels on full-chip problems (1024 tiles of 256^2, K = 16384: prologue / epilogue negligible),
# row-major vs k-major operands, with the DMA-only (MMDIT_GEMM_DEBUG=2) and no-epilogue (8) ablations of the general kernel.
cd "$GRAFT_REPO_ROOT" || exit 1
for kind in fwd dgrad wgrad; do
  for dbg in 0 2 8; do
    MMDIT_GEMM_DEBUG=$dbg python tools/gemm_ablate.py $kind 8192 8192 16384 5 2>&1 | grep TF
  done
  MMDIT_GEMM_LEAN=0 python tools/gemm_ablate.py $kind 8192 8192 16384 5 2>&1 | grep TF | sed 's/$/  (general kernel forced: MMDIT_GEMM_LEAN=0)/'
done
