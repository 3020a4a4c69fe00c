#!/usr/bin/env python3
"""Memory-side traffic per kernel launch from rocprofv3 PMC passes of one command (--pmc FETCH_SIZE, --pmc
WRITE_SIZE and -- round 5 -- the read requests by size: they do not fit one pass on gfx950), merged into
profiles/hbm_traffic.json under a workload key -- the headline step and BASELINE configs 2-5 each have their own
passes (tools/prof/pmc_passes.sh):

    python tools/pmc_traffic.py fetch_results.db write_results.db --sized-db rdreq_results.db --workload headline \
        --known-bytes 4294967296 --merge-into gpurun_out/hbm_traffic.json

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE / WRITE_SIZE count KiB at
the L2's memory side (Infinity Cache hits included); on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced
read stream, because it tallies a 128-byte request at 64.  Round 5 settles what that means for gathers: gfx950 has the
request counters by size (TCC_EA0_RDREQ_32B / _64B / _128B), and EVERY kernel of this library -- 16-byte streams and
2-byte table probes alike -- fetches 128-byte requests almost only (profiles/r05_*_pmc_sq.txt): a miss brings a 128-byte
line whatever the lane asked for.  So bytes read = 32 n32 + 64 n64 + 128 n128 exactly (`fetch_by_request_size`, = 2 x
FETCH_SIZE), `hbm_bytes_per_launch` uses it where the sized pass is there and the calibrated factor (zh_checksum_pieces_
kernel reads a known number of bytes once, --known-bytes) otherwise; `hbm_bytes_per_launch_uncorrected` is FETCH_SIZE +
WRITE_SIZE as reported.  These are bytes across the L2's memory side: the share the Infinity Cache serves never reaches
HBM, which is how a gather kernel can show more than the 6.29 TB/s a copy gets from HBM.
"""
import argparse
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    rows = db.execute(
        "select k.name, count(*), sum(p.value) from counters_collection p join kernels k "
        "on k.dispatch_id = p.dispatch_id where p.counter_name = ? group by k.name", (counter,)).fetchall()
    out = {}
    for name, calls, total in rows:  # "void zh_x_kernel<true>(...)" -> "zh_x_kernel"
        key = name.split("(")[0].split("<")[0].replace("void ", "").strip()
        c, t = out.get(key, (0, 0))
        out[key] = (c + calls, t + total)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_db")
    ap.add_argument("write_db")
    ap.add_argument("--workload", required=True, help="key in the file: headline, c2, c3_zlib6, c4_share, c5, ...")
    ap.add_argument("--known-bytes", type=int, default=0,
                    help="bytes ONE launch of zh_checksum_pieces_kernel reads in this command (0: no calibration here)")
    ap.add_argument("--command", default="", help="the profiled command, for the record")
    ap.add_argument("--sized-db", default=None, help="a pass with TCC_EA0_RDREQ_sum / _32B_sum / _64B_sum / _128B_sum")
    ap.add_argument("--merge-into", required=True)
    a = ap.parse_args()
    from bench import source_sha
    fetch = per_kernel(a.fetch_db, "FETCH_SIZE")
    write = per_kernel(a.write_db, "WRITE_SIZE")
    sized = {}
    if a.sized_db:
        by = {c: per_kernel(a.sized_db, "TCC_EA0_RDREQ_%s_sum" % c) for c in ("32B", "64B", "128B")}
        for name in set().union(*[set(v) for v in by.values()]):
            calls = max(by[c].get(name, (0, 0))[0] for c in by)
            if calls:
                sized[name] = sum(w * by[c].get(name, (0, 0))[1] for c, w in (("32B", 32), ("64B", 64), ("128B", 128))) / calls
    try:
        with open(a.merge_into) as fh:
            doc = json.load(fh)
        if doc.get("source_sha") != source_sha() or "workloads" not in doc:
            doc = None  # other sources (or the old single-workload layout): start over
    except (OSError, ValueError):
        doc = None
    if doc is None:
        doc = {"source_sha": source_sha(),  # bench.py quotes these numbers only while the kernel sources are the ones measured
               "calibration": "zh_checksum_pieces_kernel reads a known number of bytes once (dwordx4, coalesced): the factor "
                              "is only used for a kernel that has no sized pass",
               "note": "hbm_bytes_per_launch = read requests counted by size (32 n32 + 64 n64 + 128 n128, "
                       "fetch_by_request_size) + WRITE_SIZE as reported; hbm_bytes_per_launch_uncorrected = FETCH_SIZE + "
                       "WRITE_SIZE as reported (FETCH_SIZE tallies a 128-byte request at 64 on gfx950).  Bytes across the "
                       "L2's memory side, Infinity Cache hits included",
               "workloads": {}}
    calls, kib = fetch.get("zh_checksum_pieces_kernel", (0, 0))
    if a.known_bytes and calls and kib:
        factor, calibrated = a.known_bytes / (kib * 1024.0 / calls), True
    else:
        factor = doc["workloads"].get("headline", {}).get("fetch_correction_factor", 2.0)
        calibrated = False
    w = {"command": a.command, "fetch_correction_factor": round(factor, 4), "calibrated_here": calibrated, "kernels": {}}
    for name in sorted(set(fetch) | set(write)):
        if not name.startswith("zh_"):
            continue
        fc, fk = fetch.get(name, (0, 0))
        wc, wk = write.get(name, (0, 0))
        fb = fk * 1024.0 / fc * factor if fc else 0.0
        if name in sized:
            fb = sized[name]
        wb = wk * 1024.0 / wc if wc else 0.0
        w["kernels"][name] = {
            "fetch_by_request_size": int(sized[name]) if name in sized else None,
            "launches": fc or wc,
            "fetch_kib_raw_per_launch": round(fk / fc, 1) if fc else 0,
            "write_kib_raw_per_launch": round(wk / wc, 1) if wc else 0,
            "hbm_bytes_per_launch": int(fb + wb),
            "hbm_bytes_per_launch_uncorrected": int((fk * 1024.0 / fc if fc else 0.0) + wb),
        }
    doc["workloads"][a.workload] = w
    with open(a.merge_into, "w") as fh:
        json.dump(doc, fh, indent=1)
    dom = sorted(w["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:4]
    print(a.workload, "factor %.3f%s" % (factor, "" if calibrated else " (headline's)"),
          {k: round(v["hbm_bytes_per_launch"] / 1e9, 3) for k, v in dom})


if __name__ == "__main__":
    main()
