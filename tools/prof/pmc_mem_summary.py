#!/usr/bin/env python3
"""One line a kernel from tools/prof/pmc_mem.sh's text (gpurun_out/<tag>_pmc_mem.txt): average request latencies
(LEVEL or LATENCY sum / requests, cycles), translation misses, L1 tag lookups a CU-cycle, stall shares.
GRBM_GUI_ACTIVE is summed over the eight XCDs: / 8 = the launch's cycles."""
import re
import sys

cur, sec, data = None, None, {}
for line in open(sys.argv[1]).read().splitlines():
    if line.startswith("===="):
        sec = line.split()[1].rstrip(":")
        continue
    m = re.match(r"^(zh_\w+)\s+\((\d+) launches", line)
    if m:
        cur = (sec, m.group(1))
        data[cur] = {}
        continue
    m = re.match(r"^\s+([A-Z0-9_a-z]+)\s+(\d+)$", line)
    if m and cur:
        data[cur][m.group(1)] = int(m.group(2))


def r(d, a, b):
    return d[a] / d[b] if a in d and b in d and d[b] else float("nan")


print("workload kernel | L1->L2 read cyc (M req) write cyc | L2->fabric read cyc (M req) write cyc | UTCL1 miss % | "
      "L1 cyc a read instr | tag lookups a CU-cycle | stall %: translation in-flight limit, read tag conflict")
for (w, k), d in data.items():
    cyc = d.get("GRBM_GUI_ACTIVE", 0) / 8
    if not cyc or d.get("TCP_TCC_READ_REQ_sum", 0) < 1e6:
        continue
    print("%s %s | %.0f (%.0f) %.0f | %.0f (%.0f) %.0f | %.4f | %.0f | %.2f | %.1f %.1f" % (
        w, k, r(d, "TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum"), d.get("TCP_TCC_READ_REQ_sum", 0) / 1e6,
        r(d, "TCP_TCC_WRITE_REQ_LATENCY_sum", "TCP_TCC_WRITE_REQ_sum"),
        r(d, "TCC_EA0_RDREQ_LEVEL_sum", "TCC_EA0_RDREQ_sum"), d.get("TCC_EA0_RDREQ_sum", 0) / 1e6,
        r(d, "TCC_EA0_WRREQ_LEVEL_sum", "TCC_EA0_WRREQ_sum"),
        100 * r(d, "TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_REQUEST_sum"),
        r(d, "TCP_TCP_LATENCY_sum", "TCP_TA_TCP_STATE_READ_sum"),
        d.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / 256 / cyc,
        100 * d.get("TCP_UTCL1_STALL_INFLIGHT_MAX_sum", 0) / 256 / cyc,
        100 * d.get("TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", 0) / 256 / cyc))
