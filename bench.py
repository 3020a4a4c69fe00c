#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X.

  metric   : GiB/s of UNCOMPRESSED data through compress(BestSpeed, gzip) followed
             by uncompress(gzip, CRC verified), value = N_total / (T_compress + T_uncompress)
  workload : 4096 x 1 MiB synthetic "Silesia-mix" buffers per GPU,
             inputs resident in HBM when the timed region starts
  step     : one compress pass + one uncompress pass over the whole batch

    python bench.py [--gpus N] [--steps K] [--warmup W] [--buffers 4096] [--size 1048576]

N > 1 is launched by torch.distributed.run (one process per GPU).  The path has no
exchange step (buffers are independent), so ranks run their own shard
(weak scaling) and only the barrier + max-over-ranks timing go through RCCL.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GIB = float(1 << 30)
HBM_PEAK = 8.0e12  # bytes/s, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW" (spec)


def cpu_baseline(bufs, level, cores):
    """The oracle (C restatement of zippy, oracle/zippy_oracle.c) timed on the host
    cores: compress(level, gzip) + uncompress, one buffer per task."""
    import oracle
    from concurrent.futures import ThreadPoolExecutor
    oracle.lib()

    def comp(b):
        return oracle.compress(b, level, oracle.dfGzip, fname_len=0)

    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(comp, bufs[:cores]))  # warm-up
        t0 = time.perf_counter()
        blobs = list(ex.map(comp, bufs))
        t1 = time.perf_counter()
        outs = list(ex.map(oracle.uncompress, blobs))
        t2 = time.perf_counter()
    assert outs[0] == bufs[0]
    nbytes = sum(len(b) for b in bufs)
    # second yardstick (the reference's own tests/bench.nim does the same): system zlib
    # on the same sample and threads (zlib releases the GIL)
    import zlib
    zl = 1 if level == 1 else 6 if level == -1 else max(0, min(9, level))
    with ThreadPoolExecutor(cores) as ex:
        t3 = time.perf_counter()
        zblobs = list(ex.map(lambda b: zlib.compress(b, zl), bufs))
        t4 = time.perf_counter()
        list(ex.map(zlib.decompress, zblobs))
        t5 = time.perf_counter()
    return {
        "value": nbytes / GIB / (t2 - t0),
        "unit": "GiB/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d x %d B of the same G-mix batch, oracle compress(level %d, gzip)+uncompress, "
                  "%d threads; compress %.3f GiB/s, uncompress %.3f GiB/s" % (
                      len(bufs), len(bufs[0]), level, cores, nbytes / GIB / (t1 - t0),
                      nbytes / GIB / (t2 - t1)),
        "zlib": {"level": zl, "compress_GiBps": round(nbytes / GIB / (t4 - t3), 3),
                 "uncompress_GiBps": round(nbytes / GIB / (t5 - t4), 3),
                 "both_GiBps": round(nbytes / GIB / (t5 - t3), 3),
                 "ratio": round(nbytes / sum(len(z) for z in zblobs), 4)},
    }


def hbm_traffic(kernel, n, size):
    """HBM bytes per launch of `kernel` from the rocprofv3 PMC passes committed under profiles/
    (tools/pmc_traffic.py: separate FETCH_SIZE / WRITE_SIZE runs of this same command, units and
    gfx950 corrections as the MI355X guide prescribes); None when no pass matches this workload."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as fh:
            t = json.load(fh)
        if t.get("buffers") == n and t.get("buffer_bytes") == size:
            return t["kernels"][kernel]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--buffers", type=int, default=4096)
    ap.add_argument("--size", type=int, default=1 << 20)
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (zippy_amd has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    # under torchrun the collective path is used even with one rank (exercises it on a 1-GPU box)
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from zippy_amd import api, synth
    from zippy_amd._binding import Engine

    n, size = args.buffers, args.size
    # ---- synthetic batch (this rank's shard), staged into HBM ----
    t_gen = time.perf_counter()
    host = synth.gen_batch("mix", n, size, first_index=rank * n)
    t_gen = time.perf_counter() - t_gen
    d_src = torch.from_numpy(host.reshape(-1)).cuda()
    stream = torch.cuda.current_stream()
    eng = Engine(api.LIB_PATH, device=local_rank, stream=stream.cuda_stream)
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

    def step():
        cplan.run(d_src.data_ptr(), d_comp.data_ptr())
        uplan.run(d_comp.data_ptr(), d_back.data_ptr())

    for _ in range(max(args.warmup, 1)):  # at least one untimed pass: its results are verified below
        step()
    torch.cuda.synchronize()

    # ---- correctness of what is about to be timed ----
    clens, csts = cplan.results()
    ulens, usts = uplan.results()
    assert all(s == 0 for s in csts), "compress statuses"
    assert all(s == 0 for s in usts), "uncompress statuses (CRC-32 / ISIZE verified on device)"
    assert ulens == [size] * n
    assert torch.equal(d_back, d_src), "round trip mismatch"
    comp_total = sum(clens)

    # ---- timed region ----
    kernel_ms = {}
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    t_comp = t_unc = 0.0
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ev[0].record(stream)
        cplan.run(d_src.data_ptr(), d_comp.data_ptr())
        ev[1].record(stream)
        uplan.run(d_comp.data_ptr(), d_back.data_ptr())
        ev[2].record(stream)
        ev[2].synchronize()
        t_comp += ev[0].elapsed_time(ev[1])
        t_unc += ev[1].elapsed_time(ev[2])
        for name, ms in cplan.kernel_times() + uplan.kernel_times():
            kernel_ms.setdefault(name, []).append(ms)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_uncompressed = n * size * world
    value = total_uncompressed * args.steps / GIB / elapsed
    ms_per_step = elapsed * 1e3 / args.steps

    if rank == 0:
        avg = {k: sum(v) / len(v) for k, v in kernel_ms.items()}
        dom = max((k for k in avg if k.startswith("zh_")), key=lambda k: avg[k])
        # algorithmic bytes per launch of the batch: every uncompressed byte read (compress)
        # or written (uncompress) once, every compressed byte written or read once
        algo_bytes = n * size + comp_total
        achieved = algo_bytes / (avg[dom] * 1e-3)
        out = {
            "metric": "GiB/s uncompressed throughput (compress BestSpeed + uncompress), 4096x1 MiB batch",
            "value": round(value, 3),
            "unit": "GiB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic (G-mix: seeded slices of the reference's own test corpus, SURVEY.md 8d)",
            "config": {
                "workload": "%d x %d B per GPU, compress level %d gzip + uncompress gzip (CRC-32 verified), "
                            "buffers resident in HBM" % (n, size, args.level),
                "buffers_per_gpu": n, "buffer_bytes": size, "level": args.level,
                "sharding": "independent buffers per rank, no data-path collective",
            },
            "compress_GiBps": round(n * size * args.steps / GIB / (t_comp * 1e-3), 3),
            "uncompress_GiBps": round(n * size * args.steps / GIB / (t_unc * 1e-3), 3),
            "ratio": round(n * size / comp_total, 4),
            "kernels_ms": {k: round(v, 4) for k, v in sorted(avg.items(), key=lambda kv: -kv[1])},
            "roofline": {
                "bound": "hbm", "kernel": dom, "achieved": round(achieved / 1e9, 3),
                "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(achieved / HBM_PEAK, 6),
                "traffic": hbm_traffic(dom, n, size),
                "algorithmic_bytes_per_launch": algo_bytes,
                "avg_launch_ms": round(avg[dom], 4),
            },
        }
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            cores = min(os.cpu_count() or 1, 32)
            per_core = max(2, min(24, (16 << 20) // size * 2))
            sample = [host[i].tobytes() for i in range(min(n, cores * per_core))]
            out["cpu_baseline"] = cpu_baseline(sample, args.level, cores)
        out["host_gen_s"] = round(t_gen, 1)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
