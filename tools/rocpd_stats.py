"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) into a per-kernel stats table.
Usage: python tools/rocpd_stats.py <results.db> [steps]   (steps = optimizer steps in the trace, for per-step columns)"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    m = re.match(r"([\w:]+)<(.*)>\(", name)
    if m:
        return f"{m.group(1)}<{m.group(2)[:70]}>"
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
    rows = db.execute("""select s.kernel_name, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
                         from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id and d.guid = s.guid
                         group by s.kernel_name order by 3 desc""").fetchall()
    tot = sum(r[2] for r in rows)
    t0, t1 = db.execute("select min(start), max(end) from rocpd_kernel_dispatch").fetchone()
    print(f"total kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches; trace span {(t1 - t0) / 1e6:.1f} ms")
    print(f"{'kernel':<112} {'calls':>7} {'total_ms':>10} {'avg_us':>9} {'min_us':>8} {'max_us':>8} {'pct':>6}" + ("  ms/step" if steps else ""))
    for name, n, t, mn, mx in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
        line = f"{short(name):<112} {n:>7} {t / 1e6:>10.3f} {t / n / 1e3:>9.2f} {mn / 1e3:>8.2f} {mx / 1e3:>8.2f} {100 * t / tot:>6.2f}"
        if steps:
            line += f" {t / 1e6 / steps:>8.3f}"
        print(line)


if __name__ == "__main__":
    main()
