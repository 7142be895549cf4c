"""List every dispatch of the kernels matching a substring in a rocprofv3 rocpd database: start offset, duration, grid, and
the kernel dispatched just before it (to spot in-situ slowdowns a micro-benchmark does not show).
Usage: python tools/rocpd_calls.py <results.db> <substring> [max_rows]     (max_rows < 0: the LAST |max_rows| matching dispatches)"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    sub = sys.argv[2]
    limit = int(sys.argv[3]) if len(sys.argv) > 3 else 120
    cols = [r[1] for r in db.execute("pragma table_info(rocpd_kernel_dispatch)")]
    grid = "d.grid_size_x" if "grid_size_x" in cols else "0"
    wg = "d.workgroup_size_x" if "workgroup_size_x" in cols else "0"
    rows = db.execute(f"""select s.kernel_name, d.start, d.end, {grid}, {wg}
                          from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id and d.guid = s.guid
                          order by d.start""").fetchall()
    t0 = rows[0][1]
    shown = 0
    if limit < 0:
        hits = [i for i, r in enumerate(rows) if sub in r[0]]
        skip = set(hits[:limit])
        limit = -limit
    else:
        skip = set()
    for i, (name, st, en, g, w) in enumerate(rows):
        if sub in name and shown < limit and i not in skip:
            prev = rows[i - 1] if i else None
            gap = (st - prev[2]) / 1e3 if prev else 0.0
            print(f"{(st - t0) / 1e6:10.3f} ms  dur {(en - st) / 1e3:9.2f} us  grid {g:>8} wg {w:>5}  gap_after_prev {gap:8.2f} us  prev {prev[0][:60] if prev else ''}")
            shown += 1


if __name__ == "__main__":
    main()
