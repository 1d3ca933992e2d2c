#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (--kernel-trace --stats, optionally --pmc) into the text summary kept under profiles/.

usage: tools/rocprof_summary.py <results.db> [--pmc]
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    print("# rocprofv3 kernel summary from", sys.argv[1].split("/")[-1])
    print("%-90s %8s %14s %14s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        print("%-90s %8d %14.1f %14.1f %8.3f" % (name[:90], calls, total, avg, pct))
    row = db.execute("select grid_x, workgroup_x, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size from kernels "
                     "where name like '%render_kernel%' limit 1").fetchone()
    if row:
        print("\nrender_kernel dispatch: grid_x=%d workgroup_x=%d vgpr=%d agpr=%d sgpr=%d lds=%d scratch=%d" % row)
    if "--pmc" in sys.argv:
        q = ("select name, counter_name, count(distinct dispatch_id), sum(counter_value) from pmc_events "
             "where name like '%render_kernel%' group by name, counter_name")
        try:
            print("\n%-50s %-28s %10s %22s" % ("kernel", "counter (summed over SE/XCC)", "dispatches", "avg_per_dispatch"))
            for kname, cname, n, tot in db.execute(q):
                print("%-50s %-28s %10d %22.1f" % (kname[:50], cname, n, tot / max(n, 1)))
        except sqlite3.Error as e:
            print("pmc query failed:", e)


if __name__ == "__main__":
    main()
