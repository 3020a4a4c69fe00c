#!/usr/bin/env python3
"""Per-kernel counters (per launch) from the rocprofv3 rocpd databases under the given directories, with the ratios
DESIGN.md §4 quotes:

  valu busy      SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES      share of a resident wave's time in which it issues a vector op
  salu busy      SQ_ACTIVE_INST_SCA  / SQ_WAVE_CYCLES
  lds busy       SQ_ACTIVE_INST_LDS  / SQ_WAVE_CYCLES
  vmem busy      SQ_ACTIVE_INST_VMEM / SQ_WAVE_CYCLES
  wait any       SQ_WAIT_ANY / SQ_WAVE_CYCLES              parked at s_waitcnt / a barrier
  wait inst      SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES         ready to issue, the pipe is not
  lane util      SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)    active lanes of the vector instructions issued
  simd valu      4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE)   share of the chip's vector issue slots (if the
                 GRBM counter is there; SQ counters are summed over every XCD / SE instance a dispatch reports)
  l2 hit         TCC_HIT / (TCC_HIT + TCC_MISS)
  bytes read     32 * RDREQ_32B + 64 * RDREQ_64B + 128 * RDREQ_128B  (requests at the L2's memory side, by size)
  FETCH_SIZE     as rocprofv3 reports it (KiB -> bytes): tallies a 128-byte request at 64 on gfx950
  bytes written  32 * (WRREQ - WRREQ_64B) + 64 * WRREQ_64B
"""
import glob
import sqlite3
import sys
from collections import defaultdict

vals = defaultdict(dict)
launches = {}
for root in sys.argv[1:]:
    for path in sorted(glob.glob(root + "/**/*.db", recursive=True)):
        db = sqlite3.connect(path)
        try:
            rows = db.execute(
                "select k.name, p.counter_name, count(distinct p.dispatch_id), sum(p.value) from counters_collection p "
                "join kernels k on k.dispatch_id = p.dispatch_id group by k.name, p.counter_name").fetchall()
        except sqlite3.Error as e:
            print(path, e)
            continue
        agg = defaultdict(lambda: [0, 0.0])
        for name, ctr, n, v in rows:  # template instances of one kernel together
            name = name.split("(")[0].split("<")[0].replace("void ", "").strip()
            if name.startswith("zh_"):
                agg[(name, ctr)][0] += n
                agg[(name, ctr)][1] += v
        for (name, ctr), (n, v) in agg.items():
            vals[name][ctr] = v / n
            launches[name] = n


def ratio(d, a, b, scale=1.0):
    return scale * d[a] / d[b] if a in d and b in d and d[b] else None


for name in sorted(vals, key=lambda k: -vals[k].get("SQ_WAVE_CYCLES", 0)):
    d = vals[name]
    print("%s   (%d launches in the command; per launch:)" % (name, launches[name]))
    derived = [
        ("valu busy", ratio(d, "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES")),
        ("salu busy", ratio(d, "SQ_ACTIVE_INST_SCA", "SQ_WAVE_CYCLES")),
        ("lds busy", ratio(d, "SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES")),
        ("vmem busy", ratio(d, "SQ_ACTIVE_INST_VMEM", "SQ_WAVE_CYCLES")),
        ("wait any", ratio(d, "SQ_WAIT_ANY", "SQ_WAVE_CYCLES")),
        ("wait inst", ratio(d, "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES")),
        ("lane util", ratio(d, "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", 1.0 / 64.0)),
        ("simd valu", ratio(d, "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", 4.0 / 1024.0)),
        ("valu per wave-kcycle", ratio(d, "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", 1000.0)),
        ("l2 hit", d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
         if d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0) else None),
    ]
    for k, v in derived:
        if v is not None:
            print("   %-22s %8.3f" % (k, v))
    if "TCC_EA0_RDREQ_sum" in d:
        n32, n64, n128 = (d.get("TCC_EA0_RDREQ_%s_sum" % s, 0.0) for s in ("32B", "64B", "128B"))
        print("   %-22s %14.0f   (%.0f x 32 B, %.0f x 64 B, %.0f x 128 B; RDREQ %.0f)" %
              ("bytes read", 32 * n32 + 64 * n64 + 128 * n128, n32, n64, n128, d["TCC_EA0_RDREQ_sum"]))
    if "FETCH_SIZE" in d:
        print("   %-22s %14.0f" % ("bytes FETCH_SIZE", d["FETCH_SIZE"] * 1024.0))
    if "TCC_EA0_WRREQ_sum" in d:
        w64 = d.get("TCC_EA0_WRREQ_64B_sum", 0.0)
        print("   %-22s %14.0f" % ("bytes written", 32 * (d["TCC_EA0_WRREQ_sum"] - w64) + 64 * w64))
    for ctr in sorted(d):
        print("      %-38s %18.0f" % (ctr, d[ctr]))
