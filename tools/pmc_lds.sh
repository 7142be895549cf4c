#!/bin/bash
# LDS bank-conflict and MFMA-busy counters of ONE GEMM shape for the default library and variants (MMDIT_GEMM_8P=0, tools/scratch/<v>):
#   bash tools/pmc_lds.sh "wgrad 768 2304 16384" 8p0      -> gpurun_out/pmc_lds.txt style lines on stdout
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
shape=${1:-"fwd 8192 8192 8192"}; shift
for v in default "$@"; do
  unset MMDIT_LIB MMDIT_GEMM_8P
  if [ "$v" = 8p0 ]; then export MMDIT_GEMM_8P=0; elif [ "$v" != default ]; then export MMDIT_LIB=$GRAFT_REPO_ROOT/tools/scratch/$v/libmmdit_hip.so; fi
  for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "GRBM_GUI_ACTIVE"; do
    rm -rf gpurun_out/pmc_lds
    timeout 200 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc_lds -o run --output-format csv -- python3 tools/gemm_one.py $shape > /dev/null 2>&1
    f=$(find gpurun_out/pmc_lds -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$v" <<'PY'
import csv, sys, collections
d = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "gemm" not in k: continue
    k = k.split("(")[0][-60:]
    d[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, v in d.items():
    print(sys.argv[2], k, {a: int(b / n[(k, a)]) for a, b in v.items()})
PY
    rm -rf gpurun_out/pmc_lds
  done
done
