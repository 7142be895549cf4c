"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: python tools/pmc_summary.py <counter_collection.csv> [--json out.json]
Kernel names of the GEMM templates are normalised to the names bench.py uses (gemm_dma_kernel<2,4,4,2,0,1,t,t>)."""
import collections
import csv
import json
import re
import sys


_T = {"unsigned short": "t", "float": "f", "true": "1", "false": "0"}


def norm(name):
    m = re.search(r"(gemm_dma_kernel|gemm_lean_kernel|gemm_wide_kernel|gemm_kernel)<([^>]*)>", name)   # demangled (csv output)
    if m:
        return m.group(1) + "<" + ",".join(_T.get(a.strip(), a.strip()) for a in m.group(2).split(",")) + ">"
    _E8 = {"0": "bf16", "1": "swiglu", "2": "qk", "3": "f32", "4": "swiglu_bwd"}
    m = re.search(r"gemm8_kernel<(\d+), (true|false), (true|false), (\d)", name)   # demangled: the names ops._variant gives the 8-phase kernel
    if m:
        return "gemm8_kernel<%s,%s,%s,%s>" % (m.group(1), _T[m.group(2)], _T[m.group(3)], _E8.get(m.group(4), m.group(4)))
    m = re.search(r"gemm8_kernelILi(\d+)ELb([01])ELb([01])ELi(\d)E", name)
    if m:
        return "gemm8_kernel<%s,%s,%s,%s>" % (m.group(1), m.group(2), m.group(3), _E8.get(m.group(4), m.group(4)))
    m = re.search(r"::(\w+_kernel)", name)
    if m:
        return m.group(1)
    m = re.search(r"gemm_dma_kernelILi(\d)ELi(\d)ELi(\d)ELi(\d)ELb([01])ELb([01])E(\w)(\w)E", name)
    if m:
        return "gemm_dma_kernel<%s,%s,%s,%s,%s,%s,%s,%s>" % m.groups()
    m = re.search(r"(gemm_lean_kernel|gemm_wide_kernel)ILi(\d)ELi(\d)ELi(\d)ELi(\d)ELb([01])ELb([01])E", name)
    if m:
        return "%s<%s,%s,%s,%s,%s>" % m.groups()[:6] + ("+swiglu" if m.group(7) == "1" else "")
    m = re.search(r"gemm_kernelI(\w)(\w)Lb([01])ELb([01])ELb([01])E(\w)(\w)E", name)
    if m:
        return "gemm_kernel<%s,%s,%s,%s,%s,%s,%s>" % m.groups()
    m = re.search(r"N_1\d+(\w+?_kernel)", name)
    if m:
        return m.group(1)
    return name[:60]


def main():
    rows = csv.DictReader(open(sys.argv[1]))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[norm(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, cs in agg.items():
        out[k] = {c: {"avg": sum(v) / len(v), "n": len(v), "sum": sum(v)} for c, v in cs.items()}
    keys = sorted(out, key=lambda k: -max(c["sum"] for c in out[k].values()))
    for k in keys[:30]:
        print(f"{k:<52}" + "  ".join(f"{c}: avg {v['avg']:.4g} (n={v['n']})" for c, v in sorted(out[k].items())))
    if "--json" in sys.argv:
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
