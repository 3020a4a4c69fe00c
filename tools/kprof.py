#!/usr/bin/env python3
"""Tuning aid (GPU box only): runs the BestSpeed compress + uncompress plans of bench.py on
the -DZH_KPROF build (python -m zippy_amd.build --kprof) and prints the in-kernel phase
timers (csrc/zh_kprof.h) as cycles per wave.

    python tools/kprof.py [--buffers 1024] [--size 1048576] [--kind mix]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

L1 = ["stage-in", "step: vector part", "step: fast walk", "step: slow walk", "step: table inserts",
      "stats phase", "#steps", "#fast steps", "#events fast", "#events slow", "#match extensions",
      "#waves", "#slow: all lanes miss", "#slow: (unused)", "#slow: shared-slot lane",
      "#slow: plain hit"]
L1P = ["stage-in", "P1 links (one wave)", "P2 first walks (lengths on demand)", "P3 hand-over turns", "P4 output", "#turns", "#waves", "#(unused)"]
EMIT = ["flush + loop", "chunk bitmaps", "bitmaps, source, scan, match fields", "codes", "scan + LDS ORs",
        "last flush", "#waves", "#(unused)"]
HUFF = ["histogram sum", "litlen code", "distance code", "run-length coding", "code-length code", "header bits",
        "tables out + fragment sizes", "#waves"]
INF_O = ["waiting for a round", "working", "#rounds", "#waves", "#long rounds", "#far rounds", "#doubling turns", "#tail matches"]
TOK = ["header + tables", "staging", "sync turns", "scan", "token pass", "#superchunks", "#turns", "#waves"]
INF_D = ["waiting for the output wave", "other work", "#rounds", "#waves", "vector decode", "chain walks", "staging",
         "#(unused)"]


def show(title, names, vals):
    waves = max(1, vals[names.index("#waves")])
    print("== %s (%d waves)" % (title, waves))
    cyc = sum(v for n, v in zip(names, vals) if not n.startswith("#"))
    for n, v in zip(names, vals):
        if n.startswith("#"):
            print("  %-22s %12.1f per wave" % (n, v / waves))
        else:
            print("  %-22s %12.0f cycles per wave  (%5.1f %%)" % (n, v / waves, 100.0 * v / max(cyc, 1)))
    print("  %-22s %12.0f cycles per wave" % ("total", cyc / waves))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--buffers", type=int, default=1024)
    ap.add_argument("--size", type=int, default=1 << 20)
    ap.add_argument("--kind", default="mix")
    ap.add_argument("--inflate", type=int, default=-1)
    ap.add_argument("--l1-parse", type=int, default=-1, help="1: the parallel BestSpeed parse (zh_l1p_match_kernel)")
    ap.add_argument("--lib", default=None, help="another -DZH_KPROF build of the library (python -m zippy_amd.build --kprof --variant ...)")
    ap.add_argument("--foreign", type=int, default=None, help="uncompress gzip members made by system zlib at this level instead")
    args = ap.parse_args()
    import torch
    import synth
    from zippy_amd import api
    from zippy_amd._binding import Engine
    lib_path = args.lib or api.LIB_PATH.replace(".so", "_kprof.so")
    n, size = args.buffers, args.size
    host = synth.gen_batch(args.kind, n, size)
    d_src = torch.from_numpy(host.reshape(-1)).cuda()
    eng = Engine(lib_path, stream=torch.cuda.current_stream().cuda_stream)
    eng.set_gzip_fname_len(0)
    eng.set_inflate_mode(args.inflate)
    eng.set_l1_parse(args.l1_parse)
    eng.lib.zh_kprof_read.restype = ctypes.c_int
    eng.lib.zh_kprof_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    cap = size + size // 8 + 2048
    slot = (cap + 255) & ~255
    d_comp = torch.empty(n * slot, dtype=torch.uint8, device="cuda")
    d_back = torch.empty(n * size, dtype=torch.uint8, device="cuda")
    src_off = [i * size for i in range(n)]
    comp_off = [i * slot for i in range(n)]
    cplan = eng.plan_compress(src_off, [size] * n, comp_off, [cap] * n, 1, api.dfGzip)
    uplan = eng.plan_uncompress(comp_off, [cap] * n, src_off, [size] * n, api.dfGzip)
    uplan.set_src_lens_device(cplan.device_lens())
    cplan.set_profiling(True)
    uplan.set_profiling(True)
    if args.foreign is not None:
        import zlib
        from concurrent.futures import ThreadPoolExecutor
        import numpy as np

        def gz(i):
            c = zlib.compressobj(args.foreign, zlib.DEFLATED, 31)
            return c.compress(host[i].tobytes()) + c.flush()
        with ThreadPoolExecutor(32) as ex:
            blobs = list(ex.map(gz, range(n)))
        stage = np.zeros((n, slot), dtype=np.uint8)
        for i, b in enumerate(blobs):
            stage[i, :len(b)] = np.frombuffer(b, dtype=np.uint8)
        d_comp.copy_(torch.from_numpy(stage.reshape(-1)))
        uplan = eng.plan_uncompress(comp_off, [len(b) for b in blobs], src_off, [size] * n, api.dfGzip)
        uplan.set_profiling(True)
    for it in range(2):
        eng.lib.zh_kprof_read(None, 1)
        if args.foreign is None:
            cplan.run(d_src.data_ptr(), d_comp.data_ptr())
        uplan.run(d_comp.data_ptr(), d_back.data_ptr())
        torch.cuda.synchronize()
    assert torch.equal(d_back, d_src)
    slots = (ctypes.c_ulonglong * 64)()
    eng.lib.zh_kprof_read(slots, 0)
    print("kernel ms:", {k: round(v, 3) for k, v in (cplan.kernel_times() if args.foreign is None else []) + uplan.kernel_times()})
    if args.l1_parse == 1:
        show("zh_l1p_match_kernel (thread 0 of each workgroup, per fragment)", L1P, list(slots[0:8]))
    else:
        show("zh_l1_match_kernel", L1, list(slots[0:16]))
    show("zh_emit_kernel", EMIT, list(slots[32:40]))
    show("zh_huffman_kernel", HUFF, list(slots[40:48]))
    if args.l1_parse != 1 and slots[47]:
        names = ["used symbols", "leaves pushed", "merges (2 pops + a push each)", "depths", "limit: histogram + levels",
                 "limit: quicksort", "limit: lengths by rank", "lengths + canonical codes"]
        print("   the literal / length code (the replay of huffmanCodes), cycles a block:")
        for nm, v in zip(names, slots[56:64]):
            print("     %-32s %9d" % (nm, v // slots[47]))
    show("zh_inflate_tokens_kernel (thread 0 of each stream)", TOK, list(slots[48:56]))
    show("zh_inflate_kernel: output wave", INF_O, list(slots[16:24]))
    show("zh_inflate_kernel: decode wave", INF_D, list(slots[24:32]))


if __name__ == "__main__":
    main()
