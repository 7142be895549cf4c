"""Idle-time analysis of a rocprofv3 rocpd kernel trace: over a time window (default: between the last two AdamW launches
groups, i.e. one optimizer step) report the union of kernel-busy time, the idle time, and the largest idle gaps with
the kernels on either side.
Usage: python tools/rocpd_gaps.py <results.db> [top_n]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    rows = db.execute("""select s.kernel_name, d.start, d.end
                         from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id and d.guid = s.guid
                         order by d.start""").fetchall()
    # step boundaries: first FusedOptimizer launch after a gap of non-optimizer kernels
    marks, in_opt = [], False
    for name, st, en in rows:
        is_opt = "FusedOptimizer" in name or "adamw_kernel" in name
        if is_opt and not in_opt:
            marks.append(st)
        if not is_opt and "multi_tensor_apply" not in name and "grad_sumsq" not in name and "clip_coef" not in name:
            in_opt = False
        elif is_opt:
            in_opt = True
    if len(marks) < 3:
        print("fewer than 3 optimizer steps in the trace")
        return
    w0, w1 = marks[-3], marks[-2]
    win = [(n, s, e) for n, s, e in rows if s >= w0 and s < w1]
    busy, cur_s, cur_e, gaps = 0, None, None, []
    last_name = ""
    for n, s, e in win:
        if cur_e is None:
            cur_s, cur_e, last_name = s, e, n
            continue
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, last_name, n, (cur_e - w0) / 1e6))
            cur_s, cur_e, last_name = s, e, n
        else:
            if e > cur_e:
                cur_e, last_name = e, n
    busy += cur_e - cur_s
    span = w1 - w0
    print(f"step window {span / 1e6:.3f} ms: busy (union of kernels) {busy / 1e6:.3f} ms, idle {(span - busy) / 1e6:.3f} ms in {len(gaps)} gaps; "
          f"sum of kernel durations {sum(e - s for _, s, e in win) / 1e6:.3f} ms over {len(win)} dispatches")
    hist = {}
    for g, a, b, t in gaps:
        key = "<2us" if g < 2000 else "<5us" if g < 5000 else "<10us" if g < 10000 else "<50us" if g < 50000 else ">=50us"
        c = hist.setdefault(key, [0, 0])
        c[0] += 1
        c[1] += g
    print("gap histogram: " + ", ".join(f"{k}: {v[0]} gaps / {v[1] / 1e6:.3f} ms" for k, v in hist.items()))
    for g, a, b, t in sorted(gaps, reverse=True)[:top]:
        print(f"  {g / 1e3:9.2f} us at +{t:8.3f} ms  after {a[:58]:<58}  before {b[:58]}")


if __name__ == "__main__":
    main()
