#!/usr/bin/env python3
"""One LARGE foreign stream through the index-less path (device-resident, plan API): a 128 MiB G-mix
buffer compressed by system zlib level 6 (one zlib stream, thousands of blocks) and the reference's
libressl-3.4.2.tar.gz fixture.   python tools/bench_one_stream.py [--mib 128]"""
import argparse
import json
import os
import sys
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=128)
    args = ap.parse_args()
    import torch
    import synth
    from zippy_amd import api
    from zippy_amd._binding import Engine
    stream = torch.cuda.current_stream()
    eng = Engine(api.LIB_PATH, stream=stream.cuda_stream)
    out = {}
    cases = []
    data = synth.gen_batch("mix", args.mib, 1 << 20).tobytes()
    cases.append(("zlib6_%dMiB" % args.mib, zlib.compress(data, 6), data, api.dfZlib))
    own, sts = eng.compress_batch([data], 1, api.dfGzip)
    cases.append(("own_level1_%dMiB" % args.mib, own[0], data, api.dfGzip))
    tgz = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tarballs",
                       "libressl-3.4.2.tar.gz")
    if os.path.exists(tgz):
        blob = open(tgz, "rb").read()
        cases.append(("libressl-3.4.2.tar.gz", blob, zlib.decompress(blob, 31), api.dfGzip))
    for name, comp, want, fmt in cases:
        d_c = torch.frombuffer(bytearray(comp + b"\0" * 64), dtype=torch.uint8).cuda()
        d_o = torch.empty(len(want) + 64, dtype=torch.uint8, device="cuda")
        plan = eng.plan_uncompress([0], [len(comp)], [0], [len(want)], fmt)
        plan.set_profiling(True)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t = time.perf_counter()
            plan.run(d_c.data_ptr(), d_o.data_ptr())
            lens, sts = plan.results()
            best = min(best, time.perf_counter() - t)
        assert sts == [0] and lens == [len(want)], (sts, lens)
        assert d_o[:len(want)].cpu().numpy().tobytes() == want
        out[name] = {"compressed_bytes": len(comp), "bytes": len(want), "ms": round(best * 1e3, 2),
                     "GiBps": round(len(want) / 2**30 / best, 3),
                     "kernels_ms": {k: round(v, 2) for k, v in plan.kernel_times() if v > 0.05}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
