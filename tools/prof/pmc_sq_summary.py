#!/usr/bin/env python3
"""Per-kernel sums of every counter found in the rocprofv3 rocpd databases under a directory."""
import glob
import sqlite3
import sys
from collections import defaultdict

vals = defaultdict(dict)
for path in sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True)):
    db = sqlite3.connect(path)
    try:
        rows = db.execute(
            "select k.name, p.counter_name, count(*), sum(p.value) from counters_collection p join kernels k "
            "on k.dispatch_id = p.dispatch_id group by k.name, p.counter_name").fetchall()
    except sqlite3.Error as e:
        print(path, e)
        continue
    for name, ctr, n, v in rows:
        name = name.split("(")[0].split("<")[0].replace("void ", "").strip()
        if name.startswith("zh_"):
            vals[name][ctr] = v / n
for name in sorted(vals):
    print(name)
    for ctr in sorted(vals[name]):
        print("   %-40s %18.0f" % (ctr, vals[name][ctr]))
