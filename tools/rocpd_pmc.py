"""Per-kernel hardware-counter totals from a rocprofv3 --pmc `*_results.db` (rocpd SQLite output).

usage: python tools/rocpd_pmc.py results.db [out.csv]
One line per (kernel, counter): dispatches, summed value, value per dispatch, plus the largest single
dispatch of each kernel (its grid, duration and counter value) - for the histogram kernel that is the
root pass over all rows.
"""
import sqlite3
import sys


def short(n):
    s = n.replace("(anonymous namespace)::", "").replace("void ", "")
    if s.startswith("lga::"):
        s = s.split("(")[0]
    return s.replace(",", ";")[:80]


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute(
        "select kernel_name, counter_name, count(*), sum(value), sum(duration), max(grid_size) from counters_collection "
        "group by kernel_name, counter_name order by 5 desc").fetchall()
    lines = ["kernel,counter,dispatches,total,per_dispatch,total_kernel_us"]
    for k, c, n, v, d, g in rows:
        if not k.startswith("lga::") and "lga::" not in k:
            continue
        lines.append(f"{short(k)},{c},{n},{v:.0f},{v/n:.1f},{d/1e3:.1f}")
    lines.append("# largest dispatch per kernel")
    lines.append("kernel,counter,grid_size,workgroup_size,lds_bytes,vgprs,duration_us,value")
    big = db.execute(
        "select kernel_name, counter_name, grid_size, workgroup_size, lds_block_size, vgpr_count, duration, value, "
        "dispatch_id from counters_collection c where duration = (select max(duration) from counters_collection d "
        "where d.kernel_name = c.kernel_name) order by kernel_name, counter_name").fetchall()
    for k, c, g, w, l, vg, d, v, _ in big:
        if "lga::" not in k:
            continue
        lines.append(f"{short(k)},{c},{g},{w},{l},{vg},{d/1e3:.1f},{v:.0f}")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
