#!/usr/bin/env python3
"""BASELINE.json configs[4] on one MI355X: ONE large buffer (default 128 MiB of G-mix; the
reference's tor-list.gold is not in its tree) as independent deflate blocks, compress + indexed
uncompress with the buffer resident in HBM.  Not the headline bench (bench.py); prints one JSON
line with per-kernel times.  Also times the plain (one decoder per stream) decode of the same
stream for comparison when --plain is given (slow: a single stream is serial).

    python tools/bench_c5.py [--mib 128] [--block 32768] [--level 1] [--steps 5] [--plain]
"""
import argparse
import json
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=128)
    ap.add_argument("--block", type=int, default=32768)
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--plain", action="store_true")
    args = ap.parse_args()
    import torch
    import synth
    from zippy_amd import api
    from zippy_amd._binding import Engine

    size = args.mib << 20
    host = synth.gen_batch("mix", args.mib, 1 << 20).reshape(-1)
    d_src = torch.from_numpy(host).cuda()
    stream = torch.cuda.current_stream()
    eng = Engine(api.LIB_PATH, stream=stream.cuda_stream)
    eng.set_gzip_fname_len(0)
    cap = size + size // 8 + 1024 * (size // args.block + 1) + 4096
    d_comp = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    d_back = torch.empty(size, dtype=torch.uint8, device="cuda")
    cplan = eng.plan_compress_blocks([0], [size], [0], [cap], args.level, api.dfGzip, args.block)
    cplan.set_profiling(True)
    cplan.run(d_src.data_ptr(), d_comp.data_ptr())
    (clen,), (cst,) = cplan.results()
    assert cst == 0, cst
    index = cplan.block_index(0)
    uplan = eng.plan_uncompress_indexed(0, clen, 0, size, index, api.dfGzip)
    uplan.set_profiling(True)
    uplan.run(d_comp.data_ptr(), d_back.data_ptr())
    (ulen,), (ust,) = uplan.results()
    assert ust == 0 and ulen == size, (ust, ulen)
    assert torch.equal(d_back, d_src)
    comp_host = d_comp[:clen].cpu().numpy().tobytes()
    assert zlib.decompress(comp_host, 31) == host.tobytes()  # an ordinary gzip member

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tc = tu = 0.0
    kms = {}
    for _ in range(args.steps):
        ev[0].record(stream)
        cplan.run(d_src.data_ptr(), d_comp.data_ptr())
        ev[1].record(stream)
        uplan.run(d_comp.data_ptr(), d_back.data_ptr())
        ev[2].record(stream)
        ev[2].synchronize()
        tc += ev[0].elapsed_time(ev[1])
        tu += ev[1].elapsed_time(ev[2])
        for name, ms in cplan.kernel_times() + uplan.kernel_times():
            kms.setdefault(name, []).append(ms)
    gib = size / 2.0**30
    out = {
        "workload": "1 x %d MiB, %d-byte independent deflate blocks, level %d gzip, resident in HBM" %
                    (args.mib, args.block, args.level),
        "blocks": len(index) - 1,
        "ratio": round(size / clen, 4),
        "compress_GiBps": round(gib * args.steps / (tc * 1e-3), 3),
        "uncompress_indexed_GiBps": round(gib * args.steps / (tu * 1e-3), 3),
        "both_GiBps": round(gib * args.steps / ((tc + tu) * 1e-3), 3),
        "kernels_ms": {k: round(sum(v) / len(v), 4) for k, v in kms.items() if k != "end"},
    }
    if args.plain:
        pplan = eng.plan_uncompress([0], [clen], [0], [size], api.dfGzip)
        pplan.run(d_comp.data_ptr(), d_back.data_ptr())
        ev[0].record(stream)
        pplan.run(d_comp.data_ptr(), d_back.data_ptr())
        ev[1].record(stream)
        ev[1].synchronize()
        (plen,), (pst,) = pplan.results()
        assert pst == 0 and plen == size and torch.equal(d_back, d_src)
        out["uncompress_one_decoder_GiBps"] = round(gib / (ev[0].elapsed_time(ev[1]) * 1e-3), 4)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
