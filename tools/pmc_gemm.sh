#!/bin/bash
# PMC passes over ONE GEMM shape (tools/gemm_one.py): where do the wave cycles go?   bash tools/pmc_gemm.sh fwd 8192 8192 8192
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_LDS" "GRBM_GUI_ACTIVE"; do
  i=$((i + 1))
  rm -rf gpurun_out/pmc_gemm_$i
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc_gemm_$i -o run --output-format csv -- python3 tools/gemm_one.py "$@" > gpurun_out/pmc_gemm_$i.log 2>&1
  f=$(find gpurun_out/pmc_gemm_$i -name "*counter_collection.csv" | head -1)
  python3 tools/pmc_summary.py "$f" 2>/dev/null | grep gemm_dma | head -2
  rm -rf gpurun_out/pmc_gemm_$i
done
