#!/bin/bash
# Four separate rocprofv3 --pmc passes over the same short bench.py run (run on the GPU box, from the repo root):
#   bash tools/pmc_collect.sh            -> gpurun_out/pmc_bench_{1..4}.json   (then: python tools/pmc_merge.py r01)
# Counters are collected in their own passes (no sys/runtime trace) as MI355X_MICROARCH.md prescribes; the wgrad side
# stream is disabled so that kernels do not overlap and per-kernel counters are attributable; --eager: every launch a host dispatch
# (the same kernels with the same arguments as the default hipGraph replay, one counter record per dispatch).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export MMDIT_WGRAD_STREAM=0
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i + 1))
  rm -rf gpurun_out/pmc_bench_$i
  timeout 400 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc_bench_$i -o run --output-format csv -- \
    python3 bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline --no-roofline > gpurun_out/pmc_bench_$i.log 2>&1
  f=$(find gpurun_out/pmc_bench_$i -name "*counter_collection.csv" | head -1)
  python3 tools/pmc_summary.py "$f" --json gpurun_out/pmc_bench_$i.json > gpurun_out/pmc_bench_$i.txt
  rm -rf gpurun_out/pmc_bench_$i     # keep the merged-back payload small
done
