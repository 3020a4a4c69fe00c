#!/usr/bin/env python3
"""Device-resident compress + uncompress rate per synthetic data kind (perf-cliff check):
    python tools/bench_kinds.py [--buffers 1024] [--size 1048576] [--level 1]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--buffers", type=int, default=1024)
    ap.add_argument("--size", type=int, default=1 << 20)
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--kinds", default="mix,rand,zero,runs")
    args = ap.parse_args()
    import torch
    import synth
    from zippy_amd import api
    from zippy_amd._binding import Engine
    n, size = args.buffers, args.size
    stream = torch.cuda.current_stream()
    eng = Engine(api.LIB_PATH, stream=stream.cuda_stream)
    eng.set_gzip_fname_len(0)
    cap = size + size // 8 + 2048
    slot = (cap + 255) & ~255
    d_comp = torch.empty(n * slot, dtype=torch.uint8, device="cuda")
    d_back = torch.empty(n * size, dtype=torch.uint8, device="cuda")
    src_off = [i * size for i in range(n)]
    comp_off = [i * slot for i in range(n)]
    cplan = eng.plan_compress(src_off, [size] * n, comp_off, [cap] * n, args.level, api.dfGzip)
    uplan = eng.plan_uncompress(comp_off, [cap] * n, src_off, [size] * n, api.dfGzip)
    uplan.set_src_lens_device(cplan.device_lens())
    cplan.set_profiling(True)
    uplan.set_profiling(True)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for kind in args.kinds.split(","):
        d_src = torch.from_numpy(synth.gen_batch(kind, n, size).reshape(-1)).cuda()
        for rep in range(2):
            ev[0].record(stream)
            cplan.run(d_src.data_ptr(), d_comp.data_ptr())
            ev[1].record(stream)
            uplan.run(d_comp.data_ptr(), d_back.data_ptr())
            ev[2].record(stream)
            ev[2].synchronize()
        clens, csts = cplan.results()
        ulens, usts = uplan.results()
        assert all(s == 0 for s in csts) and all(s == 0 for s in usts) and torch.equal(d_back, d_src)
        gib = n * size / 2.0**30
        km = {k: round(v, 2) for k, v in cplan.kernel_times() + uplan.kernel_times() if v > 0.5}
        print(json.dumps({"kind": kind, "ratio": round(n * size / sum(clens), 3),
                          "compress_GiBps": round(gib / (ev[0].elapsed_time(ev[1]) * 1e-3), 2),
                          "uncompress_GiBps": round(gib / (ev[1].elapsed_time(ev[2]) * 1e-3), 2), "kernels_ms": km}),
              flush=True)


if __name__ == "__main__":
    main()
