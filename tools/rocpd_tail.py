"""Per-kernel histogram of the LAST n dispatches of a rocprofv3 kernel trace (rocpd SQLite) -- e.g. the replayed steps at the end of a
bench.py run: python tools/rocpd_tail.py <results.db> <dispatches per step> <steps>"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
per, steps = int(sys.argv[2]), int(sys.argv[3])
rows = db.execute("""select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s
                     on d.kernel_id = s.id and d.guid = s.guid order by d.start desc limit ?""", (per * steps,)).fetchall()
hist = {}
for name, t0, t1 in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", name)[:100]
    h = hist.setdefault(name, [0, 0])
    h[0] += 1
    h[1] += t1 - t0
print(f"last {len(rows)} dispatches = {steps} steps of {per}")
print(f"{'kernel':<102} {'per step':>9} {'us/step':>9}")
for name, (n, t) in sorted(hist.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:<102} {n / steps:>9.1f} {t / steps / 1e3:>9.1f}")
