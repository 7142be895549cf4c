"""Dump the dispatch sequence of ONE training step (between the last two adamw_kernel launches) from a rocprofv3 rocpd database:
start offset, duration, short kernel name.  Usage: python tools/rocpd_sequence.py <results.db> [marker_substring]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.search(r"N_1\d+(\w+?_kernel)I?([\w]*)", name)
    if m:
        return m.group(1) + ("<" + m.group(2)[:40] + ">" if m.group(2) else "")
    m = re.search(r"(\w+Functor|\w+_kernel\w*|copyBuffer|\w+Kernel)", name)
    return (m.group(1) if m else name)[:70]


def main():
    db = sqlite3.connect(sys.argv[1])
    marker = sys.argv[2] if len(sys.argv) > 2 else "adamw_kernel"
    rows = db.execute("""select s.kernel_name, d.start, d.end, d.grid_size_x from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s
                         on d.kernel_id = s.id and d.guid = s.guid order by d.start""").fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    lo, hi = (marks[-2] + 1, marks[-1] + 1) if len(marks) >= 2 else (0, len(rows))
    t0 = rows[lo][1]
    for name, st, en, g in rows[lo:hi]:
        print(f"{(st - t0) / 1e3:10.1f} us  {(en - st) / 1e3:8.2f} us  grid {g:>9}  {short(name)}")


if __name__ == "__main__":
    main()
