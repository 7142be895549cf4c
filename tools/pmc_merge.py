"""Merge the four per-pass PMC summaries of tools/pmc_collect.sh into profiles/<round>_pmc_summary.{json,txt}.
python tools/pmc_merge.py r01   (reads gpurun_out/pmc_bench_{1..4}.json)

Derived columns (MI355X_MICROARCH.md, HBM / rocprofv3 section):
  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)
  hbm bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB   (gfx950: FETCH_SIZE reports half of a wide streaming read;
                         Infinity-Cache hits are included, so this is L2<->fabric traffic, an upper bound of HBM traffic)"""
import json
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
p = [json.load(open(f"gpurun_out/pmc_bench_{i}.json")) for i in (1, 2, 3, 4)]
out = {}
for k in p[1]:
    g = p[1][k].get("GRBM_GUI_ACTIVE")
    if not g:
        continue
    mf = p[0].get(k, {}).get("SQ_VALU_MFMA_BUSY_CYCLES", {"avg": 0.0})["avg"]
    fe = p[2].get(k, {}).get("FETCH_SIZE", {"avg": 0.0})["avg"]
    wr = p[3].get(k, {}).get("WRITE_SIZE", {"avg": 0.0})["avg"]
    cyc = g["avg"] / 8.0
    out[k] = {"launches": g["n"], "gui_active_cycles_per_xcd": round(cyc), "mfma_util": round(mf / (cyc * 1024.0), 4) if cyc else 0.0,
              "fetch_kib": round(fe), "write_kib": round(wr), "hbm_bytes_per_launch": int((2 * fe + wr) * 1024)}
json.dump(out, open(f"profiles/{rnd}_pmc_summary.json", "w"), indent=1, sort_keys=True)
keys = sorted(out, key=lambda k: -out[k]["gui_active_cycles_per_xcd"] * out[k]["launches"])
with open(f"profiles/{rnd}_pmc_summary.txt", "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc <set> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline (MMDIT_WGRAD_STREAM=0)\n")
    f.write("# four separate passes: {SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES} {GRBM_GUI_ACTIVE} {FETCH_SIZE} {WRITE_SIZE}; per-launch averages\n")
    f.write("# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); hbm bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB (gfx950 FETCH_SIZE correction)\n")
    f.write(f"{'kernel':<40} {'launches':>8} {'cycles/xcd':>11} {'mfma_util':>9} {'fetch MiB':>10} {'write MiB':>10} {'hbm GB/s@2.4GHz':>16}\n")
    for k in keys[:40]:
        o = out[k]
        t = o["gui_active_cycles_per_xcd"] / 2.4e9
        f.write(f"{k:<40} {o['launches']:>8} {o['gui_active_cycles_per_xcd']:>11} {o['mfma_util']:>9.3f} {o['fetch_kib'] / 1024:>10.1f} {o['write_kib'] / 1024:>10.1f} "
                f"{(o['hbm_bytes_per_launch'] / t / 1e9 if t else 0):>16.0f}\n")
print(open(f"profiles/{rnd}_pmc_summary.txt").read())
