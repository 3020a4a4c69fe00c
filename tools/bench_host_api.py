#!/usr/bin/env python3
"""Host-buffer API (what the Nim shim binds) end to end: buffers start and end in host memory, so
the rate includes allocation, H2D, kernels, D2H and the per-output malloc/copy.  DESIGN.md section 5
quotes it next to the device-resident headline.

    python tools/bench_host_api.py [--buffers 1024] [--size 1048576] [--reps 3]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--buffers", type=int, default=1024)
    ap.add_argument("--size", type=int, default=1 << 20)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch  # noqa: F401  (initialises the HIP runtime the library shares)
    import synth
    from zippy_amd import api
    bufs = [b.tobytes() for b in synth.gen_batch("mix", args.buffers, args.size)]
    total = args.buffers * args.size / 2.0**30
    api.engine().set_gzip_fname_len(0)
    outs, sts = api.compress_batch(bufs[:8], 1, api.dfGzip)  # warm-up (context, code objects)
    tc = tu = 1e9
    for _ in range(args.reps):
        t = time.perf_counter()
        outs, sts = api.compress_batch(bufs, 1, api.dfGzip)
        tc = min(tc, time.perf_counter() - t)
        assert all(s == 0 for s in sts)
        t = time.perf_counter()
        back, sts = api.uncompress_batch(outs, api.dfGzip)
        tu = min(tu, time.perf_counter() - t)
        assert all(s == 0 for s in sts)
    assert back == bufs
    cc, cu = c_abi_times(api.engine(), bufs, outs, args.reps)
    c_plain = c_abi_compress_time(api.engine(), bufs, args.reps, False)
    c_pipe = c_abi_compress_time(api.engine(), bufs, args.reps, True)
    ic, iu = c_abi_into_times(api.engine(), bufs, outs, args.reps)
    print(json.dumps({
        "workload": "%d x %d B host buffers through zh_compress_batch / zh_uncompress_batch (level 1, gzip)" %
                    (args.buffers, args.size),
        "c_abi": {"compress_GiBps": round(total / cc, 3), "uncompress_GiBps": round(total / cu, 3),
                  "both_GiBps": round(total / (cc + cu), 3),
                  "compress_one_plan_GiBps": round(total / c_plain, 3),
                  "compress_pipelined_groups_GiBps": round(total / c_pipe, 3),
                  "note": "the C call alone: pageable host buffers in, malloc'ed results out",
                  "into_compress_GiBps": round(total / ic, 3), "into_uncompress_GiBps": round(total / iu, 3),
                  "into_both_GiBps": round(total / (ic + iu), 3),
                  "into_note": "zh_*_batch_into: results into buffers the caller owns and has touched before"},
        "python_mirror": {"compress_GiBps": round(total / tc, 3), "uncompress_GiBps": round(total / tu, 3),
                          "both_GiBps": round(total / (tc + tu), 3),
                          "note": "adds the test mirror's ctypes marshalling (bytes objects in and out)"}}))


def c_abi_times(eng, bufs, blobs, reps):
    """Wall time of zh_compress_batch / zh_uncompress_batch themselves (arguments marshalled
    beforehand, results freed afterwards)."""
    import ctypes as c

    def call(fn, items, *mid):
        n = len(items)
        srcs = (c.c_void_p * n)(*[c.cast(c.c_char_p(k), c.c_void_p) for k in items])
        lens = (c.c_size_t * n)(*[len(k) for k in items])
        dsts, dlens, sts = (c.c_void_p * n)(), (c.c_size_t * n)(), (c.c_int32 * n)()
        t = time.perf_counter()
        rc = fn(eng._h, srcs, lens, n, *mid, dsts, dlens, sts)
        dt = time.perf_counter() - t
        assert rc == 0 and not any(sts)
        for i in range(n):
            eng.lib.zh_free(dsts[i])
        return dt

    cc = min(call(eng.lib.zh_compress_batch, bufs, 1, 2) for _ in range(reps))
    cu = min(call(eng.lib.zh_uncompress_batch, blobs, 0) for _ in range(reps))
    return cc, cu


def c_abi_into_times(eng, bufs, blobs, reps):
    """zh_compress_batch_into / zh_uncompress_batch_into: output buffers allocated (and touched) by the
    caller once, reused by every repetition."""
    import ctypes as c

    def call(fn, items, out_sizes, *mid):
        n = len(items)
        srcs = (c.c_void_p * n)(*[c.cast(c.c_char_p(k), c.c_void_p) for k in items])
        lens = (c.c_size_t * n)(*[len(k) for k in items])
        outs = [bytearray(sz) for sz in out_sizes]  # (zero-filled: the pages exist)
        views = [(c.c_char * len(o)).from_buffer(o) for o in outs]
        caps = (c.c_size_t * n)(*out_sizes)
        best = 1e9
        for _ in range(reps):
            dsts = (c.c_void_p * n)(*[c.addressof(v) for v in views])
            dlens, sts = (c.c_size_t * n)(), (c.c_int32 * n)()
            t = time.perf_counter()
            rc = fn(eng._h, srcs, lens, n, *mid, dsts, caps, dlens, sts)
            best = min(best, time.perf_counter() - t)
            assert rc == 0 and not any(sts)
        return best, outs, list(dlens)

    bound = [eng.compress_bound(len(b)) for b in bufs]
    ic, outs, dl = call(eng.lib.zh_compress_batch_into, bufs, bound, 1, 2)
    assert bytes(outs[0][:dl[0]]) == blobs[0]
    iu, back, _ = call(eng.lib.zh_uncompress_batch_into, blobs, [len(b) for b in bufs], 0)
    assert bytes(back[-1]) == bufs[-1]
    return ic, iu


def c_abi_compress_time(eng, bufs, reps, pipelined):
    """zh_compress_batch with the pipelined-groups rule forced on or off."""
    import ctypes as c
    n = len(bufs)
    srcs = (c.c_void_p * n)(*[c.cast(c.c_char_p(k), c.c_void_p) for k in bufs])
    lens = (c.c_size_t * n)(*[len(k) for k in bufs])
    best = 1e9
    eng.set_host_pipeline(1 if pipelined else 1 << 60, 0)
    try:
        for _ in range(reps):
            dsts, dlens, sts = (c.c_void_p * n)(), (c.c_size_t * n)(), (c.c_int32 * n)()
            t = time.perf_counter()
            rc = eng.lib.zh_compress_batch(eng._h, srcs, lens, n, 1, 2, dsts, dlens, sts)
            best = min(best, time.perf_counter() - t)
            assert rc == 0 and not any(sts)
            for i in range(n):
                eng.lib.zh_free(dsts[i])
    finally:
        eng.set_host_pipeline(0, 0)
    return best


if __name__ == "__main__":
    main()
