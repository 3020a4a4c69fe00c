#!/usr/bin/env python3
"""In-kernel phase timers of the segment-wise decode of ONE large stream (tuning build):
   python -m zippy_amd.build --kprof && python tools/kprof_one_stream.py [--mib 64] [--tgz]"""
import argparse
import ctypes
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FIND = ["staging", "first 13 bits", "code-length codes", "code lengths", "waiting", "#candidates (lane 0)",
        "#full checks (lane 0)", "#waves"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=64)
    ap.add_argument("--tgz", action="store_true")
    args = ap.parse_args()
    import torch
    import synth
    from zippy_amd import api
    from zippy_amd._binding import Engine
    from kprof import show
    eng = Engine(api.LIB_PATH.replace(".so", "_kprof.so"), stream=torch.cuda.current_stream().cuda_stream)
    eng.lib.zh_kprof_read.restype = ctypes.c_int
    eng.lib.zh_kprof_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    if args.tgz:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        comp = open(os.path.join(root, "tests", "golden", "tarballs", "libressl-3.4.2.tar.gz"), "rb").read()
        want, fmt = zlib.decompress(comp, 31), api.dfGzip
    else:
        want = synth.gen_batch("mix", args.mib, 1 << 20).tobytes()
        comp, fmt = zlib.compress(want, 6), api.dfZlib
    d_c = torch.frombuffer(bytearray(comp + b"\0" * 64), dtype=torch.uint8).cuda()
    d_o = torch.empty(len(want) + 64, dtype=torch.uint8, device="cuda")
    plan = eng.plan_uncompress([0], [len(comp)], [0], [len(want)], fmt)
    plan.set_profiling(True)
    for _ in range(2):
        eng.lib.zh_kprof_read(None, 1)
        plan.run(d_c.data_ptr(), d_o.data_ptr())
        lens, sts = plan.results()
    assert sts == [0] and lens == [len(want)]
    slots = (ctypes.c_ulonglong * 64)()
    eng.lib.zh_kprof_read(slots, 0)
    print("kernel ms:", {k: round(v, 3) for k, v in plan.kernel_times() if v > 0.01})
    show("zh_seg_find_kernel", FIND, list(slots[56:64]))
    show("zh_seg_check_kernel", ["set-up", "candidates", "#-", "#candidates", "#passed", "#waves", "#--", "#---"],
         list(slots[16:24]))


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
