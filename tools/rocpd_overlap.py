"""How much do the kernels of a rocprofv3 trace overlap?  Prints the span, the union of the kernel intervals (time
with at least one kernel running), the sum of the kernel durations, and both per queue/stream.

usage: python tools/rocpd_overlap.py r_results.db [t_from_fraction t_to_fraction]
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    sel = f"select start, end, {qcol if qcol else 0} from kernels order by start"
    rows = db.execute(sel).fetchall()
    lo, hi = rows[0][0], max(r[1] for r in rows)
    f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    a, b = lo + (hi - lo) * f0, lo + (hi - lo) * f1
    rows = [r for r in rows if r[0] >= a and r[1] <= b]
    span = rows[-1][1] - rows[0][0]
    total = sum(e - s for s, e, _ in rows)
    union, cur_s, cur_e = 0, None, None
    for s, e, _ in rows:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    print(f"kernels {len(rows)}  span {span / 1e6:.2f} ms  busy (union) {union / 1e6:.2f} ms = {100 * union / span:.1f} %  "
          f"sum of durations {total / 1e6:.2f} ms = {total / union:.2f} x the union")
    per = {}
    for s, e, q in rows:
        per.setdefault(q, [0, 0])
        per[q][0] += 1
        per[q][1] += e - s
    for q, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"  queue/stream {q}: {n} kernels, {t / 1e6:.2f} ms")


if __name__ == "__main__":
    main()
