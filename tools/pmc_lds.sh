cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for v in default swzold; do
  if [ "$v" = default ]; then unset MMDIT_LIB; else export MMDIT_LIB=$GRAFT_REPO_ROOT/tools/scratch/$v/libmmdit_hip.so; fi
  rm -rf gpurun_out/pmc_lds
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d gpurun_out/pmc_lds -o run --output-format csv -- python3 tools/gemm_one.py fwd 8192 8192 8192 > gpurun_out/pmc_lds_$v.log 2>&1
  f=$(find gpurun_out/pmc_lds -name "*counter_collection.csv" | head -1)
  echo "== $v"; python3 tools/pmc_summary.py "$f" 2>/dev/null | grep -i "gemm\|kernel " | head -4
  python3 - "$f" <<'PY'
import csv,sys,collections
d=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"][:40]; d[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
for k,v in d.items():
    if "gemm" in k: print(k, {a:int(b) for a,b in v.items()}, "conflict/active = %.3f"%(v.get("SQ_LDS_BANK_CONFLICT",0)/max(1,v.get("SQ_LDS_IDX_ACTIVE",1))))
PY
  rm -rf gpurun_out/pmc_lds
done
