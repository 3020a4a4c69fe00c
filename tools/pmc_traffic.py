#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 PMC passes of `python bench.py ...`
(one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE: they do not fit one pass on gfx950).

    python tools/pmc_traffic.py fetch_results.db write_results.db --buffers 4096 --size 1048576 \
        > profiles/hbm_traffic.json

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes:
FETCH_SIZE / WRITE_SIZE count KiB at the L2's memory side; on gfx950 FETCH_SIZE reports
half the bytes of a wide (16 B/lane) coalesced read stream.  The factor is calibrated here
on zh_checksum_pieces_kernel, which reads every input byte exactly once with dwordx4 loads
(known byte count = buffers x size); narrower patterns are reported with the same factor
and flagged as uncalibrated.
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
    ap.add_argument("--buffers", type=int, required=True)
    ap.add_argument("--size", type=int, required=True)
    a = ap.parse_args()
    fetch = per_kernel(a.fetch_db, "FETCH_SIZE")
    write = per_kernel(a.write_db, "WRITE_SIZE")
    known = a.buffers * a.size
    calls, kib = fetch.get("zh_checksum_pieces_kernel", (0, 0))
    factor = known / (kib * 1024.0 / calls) if calls and kib else 2.0
    from bench import source_sha
    out = {"buffers": a.buffers, "buffer_bytes": a.size,
           # bench.py quotes these numbers only while the kernel sources are the ones measured
           "source_sha": source_sha(),
           "fetch_correction_factor": round(factor, 4),
           "calibration": "zh_checksum_pieces_kernel reads buffers x size bytes once (dwordx4, coalesced)",
           "kernels": {}}
    for name in sorted(set(fetch) | set(write)):
        if not name.startswith("zh_"):
            continue
        fc, fk = fetch.get(name, (0, 0))
        wc, wk = write.get(name, (0, 0))
        fb = fk * 1024.0 / fc * factor if fc else 0.0
        wb = wk * 1024.0 / wc if wc else 0.0
        out["kernels"][name] = {
            "fetch_kib_raw_per_launch": round(fk / fc, 1) if fc else 0,
            "write_kib_raw_per_launch": round(wk / wc, 1) if wc else 0,
            "hbm_bytes_per_launch": int(fb + wb),
            "hbm_bytes_per_launch_uncorrected": int((fk * 1024.0 / fc if fc else 0.0) + wb),
            "note": "FETCH corrected by the dwordx4-stream factor (an upper bound for narrow gathers, whose "
                    "requests are 64 B and counted as such); WRITE_SIZE as reported (uncalibrated)",
        }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
