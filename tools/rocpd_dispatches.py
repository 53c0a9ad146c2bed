"""Per-dispatch listing (start, duration, grid, kernel) of a time window of a rocprofv3 `*_results.db`.

usage: python tools/rocpd_dispatches.py results.db out.csv [skip_fraction [count]]
Lists `count` consecutive dispatches (default 400) starting at `skip_fraction` (default 0.8) of the trace.
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    grid = [c for c in cols if c.lower() in ("grid_x", "grid_size_x", "grid_size")]
    wg = [c for c in cols if c.lower() in ("workgroup_x", "workgroup_size_x", "workgroup_size")]
    sel = f"start, end, {name_col}" + (f", {grid[0]}" if grid else ", 0") + (f", {wg[0]}" if wg else ", 0")
    rows = db.execute(f"select {sel} from kernels order by start").fetchall()
    skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.8
    count = int(sys.argv[4]) if len(sys.argv) > 4 else 400
    i0 = int(len(rows) * skip)
    t0 = rows[i0][0]
    lines = ["start_us,dur_us,gap_before_us,grid,wg,kernel"]
    prev_end = None
    for s, e, n, g, w in rows[i0:i0 + count]:
        short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].replace(",", ";")[:60]
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        lines.append(f"{(s - t0) / 1e3:.1f},{(e - s) / 1e3:.1f},{gap:.1f},{g},{w},{short}")
        prev_end = e
    open(sys.argv[2], "w").write("\n".join(lines) + "\n")
    print("columns:", cols)


if __name__ == "__main__":
    main()
