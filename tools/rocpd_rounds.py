"""Round-by-round kernel times of the LAST tree in a rocprofv3 `*_results.db` (rocpd SQLite output): one line per
growth round with the time of its partition kernel(s), its histogram kernel and its split search.

usage: python tools/rocpd_rounds.py r_results.db [out.csv]
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {name_col}, start, end, grid_x, workgroup_x from kernels order by start").fetchall() \
        if "grid_x" in cols else [(*r, 0, 1) for r in db.execute(f"select {name_col}, start, end from kernels order by start")]
    roots = [i for i, r in enumerate(rows) if "grow_kernel<true" in r[0]]
    if len(roots) < 2:
        # data with several feature tiles: a tree starts with the first histogram launch after the gradients
        roots, armed = [], False
        for i, r in enumerate(rows):
            if "gradients" in r[0] or "quantize_kernel" in r[0]:
                armed = True
            elif armed and "hist_kernel" in r[0]:
                roots.append(i)
                armed = False
    if len(roots) < 2:
        raise SystemExit("fewer than two trees in the trace")
    tree = rows[roots[-2]:roots[-1]]
    lines = ["round,partition_us,partition_workgroups,hist_us,hist_workgroups,scan_us"]
    cur = None
    out = []
    for name, s, e, gx, wx in tree:
        us = (e - s) / 1e3
        wgs = gx // max(wx, 1)
        if "grow_kernel<" in name or "hist_kernel" in name:
            if cur is None:
                cur = {"p": 0.0, "pw": 0}
            cur["h"], cur["hw"] = us, wgs
        elif "grow_flags" in name or "partition_scatter" in name or "grow_partition" in name or "partition_count" in name:
            if cur is None:
                cur = {"p": 0.0, "pw": 0}
            cur["p"] += us
            cur["pw"] = max(cur["pw"], wgs)
        elif "scan_" in name and cur is not None and "h" in cur:
            cur["s"] = us
            out.append(cur)
            cur = None
    for i, c in enumerate(out):
        lines.append(f"{i},{c['p']:.2f},{c['pw']},{c['h']:.2f},{c['hw']},{c.get('s', 0):.2f}")
    lines.append(f"# totals: partition {sum(c['p'] for c in out):.1f} us, hist {sum(c['h'] for c in out):.1f} us, scan {sum(c.get('s', 0) for c in out):.1f} us")
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
