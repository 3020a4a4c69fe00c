#!/usr/bin/env python3
"""Streams that carry no size (dfZlib, dfDeflate: zippy.nim:130-165 -- the reference's inflate grows `dst` as it goes)
through zh_uncompress_batch, next to the same data as gzip members, whose ISIZE sizes the output up front.  The data
compresses 6-7 x (slices of the reference's html_x_4), so every stream outgrows the 4 x guess the host path makes for
an unsized stream: what is timed is the guess's failure, the sizing pass and the second decode (zh_host_batch.hip).

    python tools/bench_unsized.py [--buffers 1024] [--size 1048576] [--reps 3]

Host buffers in, malloc'ed results out: the rates include the PCIe transfers both ways.  Prints one JSON line.
"""
import argparse
import ctypes as c
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def html_buffers(n, size):
    """n buffers of `size` bytes: random-offset slices (16-64 KiB) of html_x_4, seeded (synth.gen_batch "html")."""
    import synth
    return [b.tobytes() for b in synth.gen_batch("html", n, size)]


def call_uncompress(eng, blobs, fmt):
    """Wall time of zh_uncompress_batch itself; results checked by the caller through `keep`."""
    n = len(blobs)
    srcs = (c.c_void_p * n)(*[c.cast(c.c_char_p(k), c.c_void_p) for k in blobs])
    lens = (c.c_size_t * n)(*[len(k) for k in blobs])
    dsts, dlens, sts = (c.c_void_p * n)(), (c.c_size_t * n)(), (c.c_int32 * n)()
    t = time.perf_counter()
    rc = eng.lib.zh_uncompress_batch(eng._h, srcs, lens, n, fmt, dsts, dlens, sts)
    dt = time.perf_counter() - t
    assert rc == 0 and not any(sts), (rc, [s for s in sts if s][:4])
    first = c.string_at(dsts[0], dlens[0])
    last = c.string_at(dsts[n - 1], dlens[n - 1])
    for i in range(n):
        eng.lib.zh_free(dsts[i])
    return dt, first, last


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--buffers", type=int, default=1024)
    ap.add_argument("--size", type=int, default=1 << 20)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch  # noqa: F401  (initialises the HIP runtime the library shares)
    from zippy_amd import api
    eng = api.engine()
    eng.set_gzip_fname_len(0)
    bufs = html_buffers(args.buffers, args.size)
    total = args.buffers * args.size / 2.0**30
    res = {"workload": "%d x %d B of html slices (host buffers), level-1 streams made by this library, "
                       "zh_uncompress_batch" % (args.buffers, args.size)}
    for name, fmt in (("gzip", api.dfGzip), ("zlib", api.dfZlib), ("deflate", api.dfDeflate)):
        blobs, sts = api.compress_batch(bufs, 1, fmt)
        assert all(s == 0 for s in sts)
        ratio = sum(len(b) for b in bufs) / sum(len(z) for z in blobs)
        ts = []
        for _ in range(args.reps + 1):  # (the first call warms the context's device blocks up)
            dt, first, last = call_uncompress(eng, blobs, fmt)
            assert first == bufs[0] and last == bufs[-1]
            ts.append(dt)
        res[name] = {"ms": round(min(ts[1:]) * 1e3, 2), "GiBps": round(total / min(ts[1:]), 3), "ratio": round(ratio, 3),
                     "first_call_ms": round(ts[0] * 1e3, 2)}
    res["zlib_vs_gzip"] = round(res["zlib"]["ms"] / res["gzip"]["ms"], 3)
    res["deflate_vs_gzip"] = round(res["deflate"]["ms"] / res["gzip"]["ms"], 3)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
