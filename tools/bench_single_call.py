#!/usr/bin/env python3
"""Latency of ONE host-buffer call (what `compress(s)` / `uncompress(s)` of the reference become
through the shim): python tools/bench_single_call.py [--size 1048576] [--reps 30]"""
import argparse
import ctypes as c
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1 << 20)
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    import torch  # noqa: F401
    import synth
    from zippy_amd import api
    eng = api.engine()
    eng.set_gzip_fname_len(0)
    src = synth.gen_batch("mix", 1, args.size)[0].tobytes()

    def call(fn, data, *mid):
        dst, dlen = c.c_void_p(), c.c_size_t()
        t = time.perf_counter()
        rc = fn(eng._h, data, len(data), *mid, c.byref(dst), c.byref(dlen))
        dt = time.perf_counter() - t
        assert rc == 0, rc
        out = c.string_at(dst, dlen.value)
        eng.lib.zh_free(dst)
        return dt, out

    tc = tu = 1e9
    for _ in range(args.reps):
        dt, comp = call(eng.lib.zh_compress, src, 1, 2)
        tc = min(tc, dt)
        dt, back = call(eng.lib.zh_uncompress, comp, 0)
        tu = min(tu, dt)
    assert back == src
    print(json.dumps({"workload": "one %d-byte buffer per call, level 1 gzip, host memory in and out" % args.size,
                      "compress_ms": round(tc * 1e3, 3), "uncompress_ms": round(tu * 1e3, 3),
                      "compress_MiBps": round(args.size / tc / 2**20, 1),
                      "uncompress_MiBps": round(args.size / tu / 2**20, 1)}))
    if os.environ.get("ZH_TRACE"):
        call(eng.lib.zh_compress, src, 1, 2)


if __name__ == "__main__":
    main()
