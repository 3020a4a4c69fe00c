#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 PMC passes of one command (one with
--pmc FETCH_SIZE, one with --pmc WRITE_SIZE: they do not fit one pass on gfx950), merged into
profiles/hbm_traffic.json under a workload key -- the headline step and BASELINE configs 2-5 each
have their own pair of passes (tools/prof/pmc_passes.sh):

    python tools/pmc_traffic.py fetch_results.db write_results.db --workload headline \
        --known-bytes 4294967296 --merge-into gpurun_out/hbm_traffic.json

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes:
FETCH_SIZE / WRITE_SIZE count KiB at the L2's memory side; on gfx950 FETCH_SIZE reports
half the bytes of a wide (16 B/lane) coalesced read stream.  The factor is calibrated on
zh_checksum_pieces_kernel, which reads every input byte exactly once with dwordx4 loads
(--known-bytes = what one launch of it reads in this command); narrower patterns are reported
with the same factor and flagged as uncalibrated.  A workload whose command gives no clean
calibration (--known-bytes 0) takes the factor the file already holds for "headline".
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
    ap.add_argument("--merge-into", required=True)
    a = ap.parse_args()
    from bench import source_sha
    fetch = per_kernel(a.fetch_db, "FETCH_SIZE")
    write = per_kernel(a.write_db, "WRITE_SIZE")
    try:
        with open(a.merge_into) as fh:
            doc = json.load(fh)
        if doc.get("source_sha") != source_sha() or "workloads" not in doc:
            doc = None  # other sources (or the old single-workload layout): start over
    except (OSError, ValueError):
        doc = None
    if doc is None:
        doc = {"source_sha": source_sha(),  # bench.py quotes these numbers only while the kernel sources are the ones measured
               "calibration": "zh_checksum_pieces_kernel reads a known number of bytes once (dwordx4, coalesced)",
               "note": "FETCH corrected by the dwordx4-stream factor (an upper bound for narrow gathers, whose requests are "
                       "64 B and counted as such); WRITE_SIZE as reported (uncalibrated)",
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
        wb = wk * 1024.0 / wc if wc else 0.0
        w["kernels"][name] = {
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
