#!/bin/bash
# PMC passes over the attention micro-benchmark (tools/attn_bench.py): where do the wave cycles of the three kernels go?
#   bash tools/pmc_attn.sh > gpurun_out/pmc_attn.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_INSTS_SALU" "GRBM_GUI_ACTIVE"; do
  i=$((i + 1))
  rm -rf gpurun_out/pmc_attn_$i
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc_attn_$i -o run --output-format csv -- python3 tools/attn_bench.py 3 > gpurun_out/pmc_attn_$i.log 2>&1
  f=$(find gpurun_out/pmc_attn_$i -name "*counter_collection.csv" | head -1)
  python3 tools/pmc_summary.py "$f" 2>/dev/null | grep attn_ | head -4
  rm -rf gpurun_out/pmc_attn_$i
done
