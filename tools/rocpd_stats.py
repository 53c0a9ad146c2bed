"""Per-kernel statistics from a rocprofv3 `*_results.db` (rocpd SQLite output).

usage: python tools/rocpd_stats.py gpurun_out/prof/r_results.db [out.csv]
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = db.execute(
        f"select {name_col}, count(*), sum(end-start), min(end-start), max(end-start) from kernels group by {name_col} "
        "order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["kernel,calls,total_us,avg_us,min_us,max_us,percent"]
    for n, c, t, mn, mx in rows:
        short = n.replace("(anonymous namespace)::", "").replace("void ", "")
        if short.startswith("lga::"):
            short = short.split("(")[0]
        short = short.replace(",", ";")[:90]
        lines.append(f"{short},{c},{t/1e3:.1f},{t/c/1e3:.2f},{mn/1e3:.2f},{mx/1e3:.2f},{100.0*t/total:.2f}")
    span = db.execute("select min(start), max(end) from kernels").fetchone()
    lines.append(f"# kernel busy time {total/1e6:.2f} ms over a span of {(span[1]-span[0])/1e6:.2f} ms")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
