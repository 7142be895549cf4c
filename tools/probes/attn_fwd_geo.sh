#!/bin/bash
# (round 4) forward attention: waves per workgroup x LDS-DMA ring depth (probes build: MMDIT_ATTN_FWD_GEO = <waves><stages>), same box.
cd "$(dirname "$0")/../.."
[ -f tools/scratch/probes/libmmdit_hip.so ] || bash tools/build_variant.sh probes -DMMDIT_PROBES > /dev/null 2>&1
for geo in 84 83 44 43 42; do
  echo "== MMDIT_ATTN_FWD_GEO=$geo"
  MMDIT_LIB=tools/scratch/probes/libmmdit_hip.so MMDIT_ATTN_FWD_GEO=$geo python tools/attn_bench.py 30 2>&1 | grep "attn fwd\|rel err"
done
