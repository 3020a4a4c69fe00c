#!/usr/bin/env python3
"""Turns a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats`, default
output format of ROCm 7.2) into the kernel-stats summary kept under profiles/.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_x_kernel_stats.csv

Columns follow rocprofv3's own *_kernel_stats.csv (durations in ns), plus the
launch geometry and register/LDS footprint of the kernel's first dispatch.
If the DB holds PMC samples (a --pmc pass) they are appended per kernel and counter.
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "min(grid_x), min(workgroup_x), max(lds_size), max(vgpr_count), max(accum_vgpr_count), "
        "max(sgpr_count), max(scratch_size) from kernels group by name order by sum(duration) desc"
    ).fetchall()
    total = sum(r[2] for r in rows) or 1
    print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,GridX,WorkgroupX,LDS,VGPR,AGPR,SGPR,Scratch")
    for r in rows:
        print('"%s",%d,%d,%.0f,%d,%d,%.4f,%d,%d,%d,%d,%d,%d,%d' % (
            r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / total, r[6], r[7], r[8], r[9],
            r[10], r[11], r[12]))
    try:
        pmc = cur.execute(
            "select k.name, p.counter_name, count(*), sum(p.value), avg(p.value) "
            "from counters_collection p join kernels k on k.dispatch_id = p.dispatch_id "
            "group by k.name, p.counter_name order by k.name, p.counter_name").fetchall()
    except sqlite3.Error:
        pmc = []
    if pmc:
        print()
        print("Name,Counter,Dispatches,Sum,AveragePerDispatch")
        for r in pmc:
            print('"%s",%s,%d,%.0f,%.1f' % r)


if __name__ == "__main__":
    main(sys.argv[1])
