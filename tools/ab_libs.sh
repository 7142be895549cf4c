#!/bin/bash
# Same-box A/B of library variants built with tools/build_variant.sh:  bash tools/ab_libs.sh default prio mfmaprio
# For every variant: the GEMM / attention micro-benchmarks and two bench.py lines (interleaved rounds: variance between rounds shows up).
cd "$GRAFT_REPO_ROOT" || exit 1
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = default ]; then unset MMDIT_LIB; else export MMDIT_LIB=$GRAFT_REPO_ROOT/tools/scratch/$v/libmmdit_hip.so; fi
    echo "== round $round variant $v"
    python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('bench', d['value'], 'img/s', d['ms_per_step'], 'ms/step  eager', d['ms_per_step_eager'], ' loss', d['final_loss'])"
    if [ $round = 1 ]; then
      python tools/gemm_bench.py 2>&1 | grep "grouped\|square\|SwiGLU"
      python tools/attn_bench.py 30 2>&1 | grep "attn fwd"
    fi
  done
done
