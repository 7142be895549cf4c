#!/bin/bash
# One table of (shape x tile configuration x ablation) timings of the LDS-DMA GEMM on the MMDiT-B shapes (image + text rows as one
# problem of M = 26240).  Ablation bits (MMDIT_GEMM_DEBUG): 0 full kernel, 8 no epilogue, 2 operand stream only (no LDS reads / MFMA).
# Usage (GPU box, repo root): bash tools/gemm_sweep.sh > gpurun_out/gemm_sweep.txt
M=26240
for shape in "fwd $M 2304 768" "fwd $M 768 768" "fwd $M 768 3072" "fwd $M 6144 768" "dgrad $M 768 2304" "dgrad $M 768 768" "dgrad $M 768 6144" "dgrad $M 3072 768" "wgrad 2304 768 $M" "wgrad 768 3072 $M" "fwd 8192 8192 8192"; do
  for cfg in 0 1 2; do
    for dbg in 0 8 2; do
      MMDIT_GEMM_CFG=$cfg MMDIT_GEMM_DEBUG=$dbg python3 tools/gemm_ablate.py $shape 20 2>&1 | tail -1
    done
  done
done
