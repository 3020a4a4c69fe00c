#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X.

  metric   : GiB/s of UNCOMPRESSED data through compress(BestSpeed, gzip) followed
             by uncompress(gzip, CRC verified), value = N_total / (T_compress + T_uncompress)
  workload : 4096 x 1 MiB synthetic "Silesia-mix" buffers, inputs resident
             in HBM when the timed region starts
  step     : one compress pass + one uncompress pass over the whole batch

    python bench.py [--gpus N] [--steps K] [--warmup W] [--buffers 4096] [--size 1048576]
                    [--scaling strong|weak] [--level 1] [--foreign LEVEL] [--uncompress-only]

One process per GPU.  N > 1 is launched by `python -m torch.distributed.run` (the driver does
that itself; a plain `python bench.py --gpus N` re-executes itself under it) and fails loudly
when the box has fewer than N GPUs.  The path has no exchange step (buffers are independent,
zippy.nim:11-16):
  --scaling strong (default for N > 1): the batch of --buffers buffers is sharded over the ranks
      by contiguous index ranges (zippy_amd/sharding.py shard_range) -- BASELINE.json's
      "4096 x 1 MiB on 8 GPUs";
  --scaling weak: every rank runs --buffers buffers of its own.
The timed region has no data-path collective (barrier + max-over-ranks only).  For N > 1 a
separate, separately reported leg moves the batch from and to rank 0 over RCCL
(scatter_fixed / gather_variable / scatter_variable / gather_fixed): `transfer`.
Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GIB = float(1 << 30)
HBM_PEAK = 8.0e12  # bytes/s, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW" (spec)
HBM_COPY_PEAK = 6.29e12  # bytes/s, the guide's measured copy rate


def source_sha():
    """Fingerprint of the kernel sources: profiles/hbm_traffic.json carries the one it was
    measured with, and its numbers are only quoted for the same sources (the GPU box has no .git)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "zippy_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(d, name), "rb") as fh:
                h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def _stats(times, nbytes):
    avg = sum(times) / len(times)
    sd = math.sqrt(sum((t - avg) ** 2 for t in times) / len(times))
    return {"min_ms": round(min(times) * 1e3, 3), "avg_ms": round(avg * 1e3, 3), "sd_ms": round(sd * 1e3, 3),
            "GiBps_at_min": round(nbytes / GIB / min(times), 4), "GiBps_at_avg": round(nbytes / GIB / avg, 4)}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_cpu_info():
    """What the process may use of the host: the CPUs it may run on, the cgroup's quota and CPU set, the NUMA nodes --
    so that a cpu_baseline that stops scaling can be told from a lease that is smaller than the box."""
    def read(path):
        try:
            with open(path) as fh:
                return fh.read().strip()
        except OSError:
            return None
    info = {"logical_cpus": os.cpu_count(),
            "sched_getaffinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
            "cgroup_cpu_max": read("/sys/fs/cgroup/cpu.max"),
            "cgroup_cpuset_effective": read("/sys/fs/cgroup/cpuset.cpus.effective"),
            "numa_nodes": None, "threads_per_core": None}
    if info["cgroup_cpu_max"] is None:  # cgroup v1
        q, per = read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), read("/sys/fs/cgroup/cpu/cpu.cfs_period_us")
        if q and per:
            info["cgroup_cpu_max"] = "%s %s" % (q, per)
    try:
        info["numa_nodes"] = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()])
    except OSError:
        pass
    sib = read("/sys/devices/system/cpu/cpu0/topology/thread_siblings_list")
    if sib:
        info["threads_per_core"] = len([x for part in sib.split(",") for x in
                                        (range(int(part.split("-")[0]), int(part.split("-")[-1]) + 1))])
    cm = info["cgroup_cpu_max"]
    if cm and cm.split()[0] not in ("max", "-1"):
        try:
            info["cgroup_cpu_quota_cpus"] = round(float(cm.split()[0]) / float(cm.split()[1]), 2)
        except (ValueError, IndexError, ZeroDivisionError):
            pass
    return info


def cpu_baseline(bufs, level, cores, reps=8):
    """The oracle (C restatement of zippy, oracle/zippy_oracle.c) timed on the host cores the way
    the reference times itself (tests/bench.nim:27-28,63-64 with benchy: warm-up, >= 10 repetitions,
    min / avg / sd): compress(level, gzip) and uncompress (CRC verified), one buffer per task, on
    1 thread and on `cores` worker threads -- C threads inside the library, each pinned to its own
    core, the clock around the parallel region only (oracle.batch_mt); system zlib at the matching
    level as the second yardstick (tests/bench.nim:30-34,66-70; Python threads, zlib releases the GIL)."""
    import zlib
    import oracle
    from concurrent.futures import ThreadPoolExecutor
    oracle.lib()
    zl = 1 if level == 1 else 6 if level == -1 else max(0, min(9, level))

    def leg(sample, threads):
        nbytes = sum(len(b) for b in sample)
        out = {}
        _, blobs = oracle.batch_mt(sample, 0, level, oracle.dfGzip, threads, keep=True)  # warm-up
        _, back = oracle.batch_mt(blobs, 1, level, oracle.dfGzip, threads, keep=True)
        assert back == sample
        tc = [oracle.batch_mt(sample, 0, level, oracle.dfGzip, threads)[0] for _ in range(reps)]
        tu = [oracle.batch_mt(blobs, 1, level, oracle.dfGzip, threads)[0] for _ in range(reps)]
        out["oracle"] = {"compress": _stats(tc, nbytes), "uncompress": _stats(tu, nbytes),
                         "both_GiBps_at_avg": round(nbytes / GIB / (sum(tc) / reps + sum(tu) / reps), 4),
                         "both_GiBps_at_min": round(nbytes / GIB / (min(tc) + min(tu)), 4),
                         "ratio": round(nbytes / sum(len(z) for z in blobs), 4)}
        with ThreadPoolExecutor(threads) as ex:
            comp, unc = (lambda b: zlib.compress(b, zl)), zlib.decompress
            blobs = list(ex.map(comp, sample))  # warm-up
            tc, tu = [], []
            for _ in range(3):
                t0 = time.perf_counter()
                blobs = list(ex.map(comp, sample))
                t1 = time.perf_counter()
                list(ex.map(unc, blobs))
                t2 = time.perf_counter()
                tc.append(t1 - t0)
                tu.append(t2 - t1)
            out["zlib"] = {"compress": _stats(tc, nbytes), "uncompress": _stats(tu, nbytes),
                           "both_GiBps_at_avg": round(nbytes / GIB / (sum(tc) / len(tc) + sum(tu) / len(tu)), 4),
                           "ratio": round(nbytes / sum(len(z) for z in blobs), 4)}
        out["threads"] = threads
        out["sample_bytes"] = nbytes
        return out

    one = leg(bufs[:max(1, min(len(bufs), (8 << 20) // max(1, len(bufs[0]))))], 1)
    # What the lease allows, not what the box shows: a cgroup CPU quota (cpu.max) caps the CPU time of ALL threads
    # together -- 256 logical CPUs behind a quota of 16 run 16 threads' worth -- so the thread counts tried are the
    # quota and twice it (threads that wait for memory leave quota to their siblings).  Without a quota: every CPU
    # the process may run on, and -- where there are enough of them to be SMT siblings and for memory bandwidth to
    # matter -- half, a quarter, an eighth as many.  The best of them is the baseline.
    host = host_cpu_info()
    quota = host.get("cgroup_cpu_quota_cpus")
    if quota and quota < cores:
        q = max(1, int(math.ceil(quota)))
        legs = {t: leg(bufs, t) for t in sorted({q, min(cores, 2 * q)})}
    else:
        legs = {cores: leg(bufs, cores)}
        t = cores // 2
        while t >= 32 and len(legs) < 4:  # (256 logical CPUs: 256, 128, 64, 32 threads)
            legs[t] = leg(bufs, t)
            t //= 2
    best = max(legs, key=lambda t: legs[t]["oracle"]["both_GiBps_at_avg"])
    many = legs[best]
    one_v = one["oracle"]["both_GiBps_at_avg"]
    return {
        "host": host,
        "cpu_time_limit": ("cgroup quota: %.4g CPUs of %d logical" % (quota, cores)) if quota and quota < cores else None,
        # the harness's scaling, thread count by thread count: x one thread (compress, uncompress, both)
        "speedup_over_1_thread": {str(t): [round(v["oracle"]["compress"]["GiBps_at_avg"] / one["oracle"]["compress"]["GiBps_at_avg"], 1),
                                           round(v["oracle"]["uncompress"]["GiBps_at_avg"] / one["oracle"]["uncompress"]["GiBps_at_avg"], 1),
                                           round(v["oracle"]["both_GiBps_at_avg"] / one_v, 1)] for t, v in sorted(legs.items())},
        "value": many["oracle"]["both_GiBps_at_avg"],
        "value_at_min": many["oracle"]["both_GiBps_at_min"],
        "unit": "GiB/s",
        "cores": best,
        "cpu": cpu_model(),
        "kind": "port",
        "sample": "%d x %d B of the same G-mix batch (1 thread: %d B), oracle = C restatement of zippy: "
                  "compress(level %d, gzip) + uncompress(CRC verified), one buffer per task on pinned C "
                  "threads (first come, first served; tried with %s threads, the best kept: %d), %d repetitions after a "
                  "warm-up; value = uncompressed bytes / (avg compress + avg uncompress), value_at_min the same with "
                  "the best repetitions" % (
                      len(bufs), len(bufs[0]), one["sample_bytes"], level, " and ".join(str(t) for t in legs), best, reps),
        "value_1_thread": one["oracle"]["both_GiBps_at_avg"],
        "all_cores": many,
        "other_thread_counts": {str(t): v["oracle"] for t, v in legs.items() if t != best},
        "one_thread": one,
        "zlib_level": zl,
    }


def cpu_level_leg(sample, level, threads, reps=3):
    """compress(level, gzip) of `sample` by the oracle on `threads` pinned C threads: GiB/s and ratio (the CPU line of a
    side config; the same harness as cpu_baseline)."""
    import oracle
    nbytes = sum(len(b) for b in sample)
    _, blobs = oracle.batch_mt(sample, 0, level, oracle.dfGzip, threads, keep=True)  # warm-up
    tc = [oracle.batch_mt(sample, 0, level, oracle.dfGzip, threads)[0] for _ in range(reps)]
    return {"compress": _stats(tc, nbytes), "threads": threads, "kind": "port", "ratio": round(nbytes / sum(len(z) for z in blobs), 4),
            "sample": "%d x %d B of the same batch, oracle.compress(level %d, gzip), %d repetitions" % (
                len(sample), len(sample[0]), level, reps)}


def hbm_traffic(workload="headline", field="hbm_bytes_per_launch"):
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/
    (tools/prof/pmc_passes.sh -> tools/pmc_traffic.py: separate FETCH_SIZE / WRITE_SIZE runs of the command
    behind every workload -- the headline step and BASELINE configs 2-5 --, units and gfx950 corrections as the
    MI355X guide prescribes).  Quoted only when the file was taken with THESE kernel sources; otherwise {}
    (-> "traffic": null)."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as fh:
            t = json.load(fh)
        if t.get("source_sha") == source_sha():
            return {k: v[field] for k, v in t["workloads"][workload]["kernels"].items()}
    except (OSError, KeyError, ValueError):
        pass
    return {}


def own_bytes(N, C):
    """Algorithmic bytes a kernel has to move for a shard of N uncompressed / C compressed bytes
    (DESIGN.md 5): the matchers read the source (N), emit reads it again and writes
    the stream (N + C), the serial decoder reads the stream and writes the output (N + C); of the
    split decoder's pair the tokens kernel must read the stream (C) and the writer must write the
    output (N) -- the token records between them are this design's own traffic, not the
    algorithm's, and show up in `traffic` instead; the checksum reads N."""
    return {"zh_l1_match_kernel": N, "zh_chain_search_kernel": N, "zh_chain_walk_kernel": N,
            "zh_l1p_match_kernel": N, "zh_emit_kernel": N + C, "zh_inflate_kernel": N + C,
            "zh_inflate_tokens_kernel": C, "zh_inflate_write_kernel": N, "zh_checksum_pieces_kernel": N}


def collect_kernel_times(kms, plans):
    """One step's entries of zh_plan_kernel_times: a kernel that is launched several times a step (the checksums of
    both passes; ranges of a batch whose scratch is bounded, csrc/zh_plan_run.hip) is one entry -- (ms of all its
    launches, launches)."""
    step = {}
    for pl in plans:
        if pl is None:
            continue
        for name, ms in pl.kernel_times():
            t, c = step.get(name, (0.0, 0))
            step[name] = (t + ms, c + 1)
    for name, tc in step.items():
        kms.setdefault(name, []).append(tc)


def kernel_averages(kms):
    """-> ({kernel: ms a step, all launches together}, {kernel: launches a step})"""
    return ({k: sum(t for t, _ in v) / len(v) for k, v in kms.items()},
            {k: v[-1][1] for k, v in kms.items()})


def time_plans(torch, stream, cplan, uplan, bufs, steps, warmup, verify=None):
    """`warmup` untimed passes (verified by `verify`), then `steps` timed ones with HIP events on
    the launch stream: (compress ms, uncompress ms, per-kernel average ms)."""
    d_src, d_comp, d_back = bufs
    for _ in range(warmup):
        if cplan is not None:
            cplan.run(d_src.data_ptr(), d_comp.data_ptr())
        if uplan is not None:
            uplan.run(d_comp.data_ptr(), d_back.data_ptr())
    torch.cuda.synchronize()
    if verify:
        verify()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tc = tu = 0.0
    kms = {}
    for _ in range(steps):
        ev[0].record(stream)
        if cplan is not None:
            cplan.run(d_src.data_ptr(), d_comp.data_ptr())
        ev[1].record(stream)
        if uplan is not None:
            uplan.run(d_comp.data_ptr(), d_back.data_ptr())
        ev[2].record(stream)
        ev[2].synchronize()
        tc += ev[0].elapsed_time(ev[1])
        tu += ev[1].elapsed_time(ev[2])
        collect_kernel_times(kms, (cplan, uplan))
    if not steps:
        return 0.0, 0.0, {}
    avg, launches = kernel_averages(kms)
    time_plans.launches = launches  # (of the last call: side_configs / pp_fields read it right behind the call)
    return tc / steps, tu / steps, avg


def side_configs(torch, eng, api, synth, stream, host, steps):
    """BASELINE.json configs 2-5 on this GPU, a few steps each, after the headline: one entry per
    config with its throughput (uncompressed GiB/s over the timed passes), ratio, dominant kernel
    and that kernel's fraction of the HBM roof.  Inputs resident in HBM, round trips verified."""
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    out = {}

    def entry(workload, nbytes, comp_bytes, tc, tu, kms, pmc_key=None):
        ms = (tc if tc else 0.0) + (tu if tu else 0.0)
        own = own_bytes(nbytes, comp_bytes)
        launches = dict(getattr(time_plans, "launches", {}))
        dom = max((k for k in kms if k.startswith("zh_")), key=lambda k: kms[k])
        nl = max(1, launches.get(dom, 1))
        e = {"workload": workload, "value": round(nbytes / GIB / (ms * 1e-3), 3), "unit": "GiB/s",
             "ms_per_step": round(ms, 3), "ratio": round(nbytes / comp_bytes, 4),
             "dominant_kernel": dom, "dominant_kernel_ms": round(kms[dom], 4), "dominant_kernel_launches": nl,
             "algorithmic_bytes": own.get(dom, nbytes + comp_bytes),  # (of all its launches together, like its ms)
             "frac": round(own.get(dom, nbytes + comp_bytes) / (kms[dom] * 1e-3) / HBM_PEAK, 6),
             # measured HBM bytes a launch of that kernel (profiles/hbm_traffic.json, this workload's own PMC passes)
             "traffic": (hbm_traffic(pmc_key).get(dom) or 0) * nl or None if pmc_key and full_size else None,
             # (the same as FETCH_SIZE + WRITE_SIZE report it: a 128-byte request tallied at 64)
             "traffic_raw": (hbm_traffic(pmc_key, "hbm_bytes_per_launch_uncorrected").get(dom) or 0) * nl or None
             if pmc_key and full_size else None,
             "kernels_ms": {k: round(v, 4) for k, v in sorted(kms.items(), key=lambda kv: -kv[1]) if k != "end"}}
        if tc and tu:
            e["compress_GiBps"] = round(nbytes / GIB / (tc * 1e-3), 3)
            e["uncompress_GiBps"] = round(nbytes / GIB / (tu * 1e-3), 3)
        return e

    def oracle_sample(cplan, d_comp, data, comp_off, level, k):
        """k of the batch's streams (evenly spread) against oracle.compress, byte for byte: the side configs' own
        evidence of identity, like the headline's parity_sample (checker only, outside every timed region)."""
        import oracle
        clens, _ = cplan.results()
        pick = sorted(set(int(i * len(clens) / k) for i in range(k)))
        same = 0
        for i in pick:
            got = d_comp[comp_off[i]:comp_off[i] + clens[i]].cpu().numpy().tobytes()
            same += got == oracle.compress(data[i].tobytes(), level, oracle.dfGzip, fname_len=0)  # (the engine's FNAME is pinned to 0 letters in main)
        return {"streams": len(pick), "identical": same == len(pick), "against": "oracle.compress(level %d, gzip)" % level}

    def pack_times(cplan, uplan, d_comp, d_back, d_src, n, slot):
        """zh_plan_pack (slots -> streams back to back + device offsets) and zh_plan_unpack (the inverse, into the
        uncompress plan's slots) on this batch: what one GPU adds to a transfer leg, HIP events on the launch stream;
        the unpacked streams are decoded once more and compared."""
        d_packed = torch.empty(n * slot, dtype=torch.uint8, device="cuda")
        d_offs = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
        d_comp2 = torch.zeros_like(d_comp)
        cplan.run(d_src.data_ptr(), d_comp.data_ptr())
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tp = tu = 0.0
        reps = 5
        for it in range(reps + 1):
            ev[0].record(stream)
            cplan.pack(d_comp.data_ptr(), d_packed.data_ptr(), d_packed.numel(), d_offs.data_ptr())
            ev[1].record(stream)
            uplan.unpack(d_packed.data_ptr(), d_offs.data_ptr(), d_comp2.data_ptr())
            ev[2].record(stream)
            ev[2].synchronize()
            if it:  # (the first pass allocates the plan's length array)
                tp += ev[0].elapsed_time(ev[1])
                tu += ev[1].elapsed_time(ev[2])
        uplan.run(d_comp2.data_ptr(), d_back.data_ptr())
        _, usts = uplan.results()
        assert all(x == 0 for x in usts) and torch.equal(d_back, d_src), "pack / unpack round trip"
        uplan.set_src_lens_device(cplan.device_lens())
        packed_bytes = int(d_offs[n].item())
        del d_packed, d_comp2
        return {"pack_ms": round(tp / reps, 4), "unpack_ms": round(tu / reps, 4), "packed_bytes": packed_bytes,
                "slot_bytes": n * slot}

    def pcie_times(d_src, n, size, cap, slot, level, chunks=4):
        """The alternative to the RCCL scatter / gather when the batch lives in HOST memory ("report both"):
        every GPU pulls its own shard over its own PCIe link.  One GPU's share, pinned host buffers: the four legs alone
        (raw in, streams out, streams in, raw out), and both directions as ONE pipeline of `chunks` chunks -- copies on a
        second stream beside the kernels of the chunk before, the streams packed on the device (zh_plan_pack), the host
        only ever waiting for a chunk's byte count.  Never `value`: PCIe-inclusive."""
        k = max(1, min(chunks, n))
        bounds = [n * i // k for i in range(k + 1)]
        h_raw = torch.empty(n * size, dtype=torch.uint8, pin_memory=True)
        h_raw.copy_(d_src)
        h_back = torch.empty(n * size, dtype=torch.uint8, pin_memory=True)
        h_pack = torch.empty(n * slot, dtype=torch.uint8, pin_memory=True)
        h_tot = torch.zeros(k, dtype=torch.int64, pin_memory=True)
        d_in = torch.empty(n * size, dtype=torch.uint8, device="cuda")
        d_comp = torch.empty(n * slot, dtype=torch.uint8, device="cuda")
        d_pack = torch.empty(n * slot, dtype=torch.uint8, device="cuda")
        d_comp2 = torch.empty(n * slot, dtype=torch.uint8, device="cuda")
        d_out = torch.empty(n * size, dtype=torch.uint8, device="cuda")
        d_offs = [torch.zeros(bounds[i + 1] - bounds[i] + 1, dtype=torch.int64, device="cuda") for i in range(k)]
        cps, ups = [], []
        for i in range(k):
            idx = range(bounds[i], bounds[i + 1])
            m = len(idx)
            cps.append(eng.plan_compress([j * size for j in idx], [size] * m, [j * slot for j in idx], [cap] * m, level, api.dfGzip))
            ups.append(eng.plan_uncompress([j * slot for j in idx], [cap] * m, [j * size for j in idx], [size] * m, api.dfGzip))
        copy = torch.cuda.Stream()

        def ev():
            return torch.cuda.Event(enable_timing=True)

        def trip():
            """-> (wall ms of the whole pipeline, packed bytes a chunk)"""
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e_in, e_c, e_t = [ev() for _ in range(k)], [ev() for _ in range(k)], [ev() for _ in range(k)]
            with torch.cuda.stream(copy):  # raw in: all chunks queued, in order
                for i in range(k):
                    lo, hi = bounds[i] * size, bounds[i + 1] * size
                    d_in[lo:hi].copy_(h_raw[lo:hi], non_blocking=True)
                    e_in[i].record(copy)
            for i in range(k):  # compress + pack a chunk as soon as it is in; its byte count follows on the copy stream
                stream.wait_event(e_in[i])
                cps[i].run(d_in.data_ptr(), d_comp.data_ptr())
                cps[i].pack(d_comp.data_ptr(), d_pack.data_ptr() + bounds[i] * slot, (bounds[i + 1] - bounds[i]) * slot,
                            d_offs[i].data_ptr())
                e_c[i].record(stream)
                with torch.cuda.stream(copy):
                    copy.wait_event(e_c[i])
                    h_tot[i:i + 1].copy_(d_offs[i][-1:], non_blocking=True)
                    e_t[i].record(copy)
            tot = []
            for i in range(k):  # streams out, chunk by chunk (the kernels of the chunks behind it are running)
                e_t[i].synchronize()
                tot.append(int(h_tot[i]))
                lo = bounds[i] * slot
                with torch.cuda.stream(copy):
                    h_pack[lo:lo + tot[i]].copy_(d_pack[lo:lo + tot[i]], non_blocking=True)
            copy.synchronize()
            # ... and back: streams in, unpack + uncompress a chunk as soon as it is in, raw out
            e_in2, e_u = [ev() for _ in range(k)], [ev() for _ in range(k)]
            with torch.cuda.stream(copy):
                for i in range(k):
                    lo = bounds[i] * slot
                    d_pack[lo:lo + tot[i]].copy_(h_pack[lo:lo + tot[i]], non_blocking=True)
                    e_in2[i].record(copy)
            for i in range(k):
                stream.wait_event(e_in2[i])
                ups[i].unpack(d_pack.data_ptr() + bounds[i] * slot, d_offs[i].data_ptr(), d_comp2.data_ptr())
                ups[i].run(d_comp2.data_ptr(), d_out.data_ptr())
                e_u[i].record(stream)
                with torch.cuda.stream(copy):
                    copy.wait_event(e_u[i])
                    lo, hi = bounds[i] * size, bounds[i + 1] * size
                    h_back[lo:hi].copy_(d_out[lo:hi], non_blocking=True)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e3, tot
        trip()  # warm-up (plans' first runs, pinned pages touched)
        assert torch.equal(h_back, h_raw), "PCIe pipeline: round trip"
        wall = min(trip()[0] for _ in range(3))
        _, tot = trip()
        for pl in ups:
            _, usts = pl.results()
            assert all(x == 0 for x in usts)
        assert torch.equal(h_back, h_raw), "PCIe pipeline: round trip"
        # the four legs alone, whole share at a time
        e = [ev() for _ in range(5)]
        packed = sum(tot)
        legs = [0.0] * 4
        for _ in range(3):
            e[0].record(stream)
            d_in.copy_(h_raw, non_blocking=True)
            e[1].record(stream)
            h_pack[:packed].copy_(d_pack[:packed], non_blocking=True)
            e[2].record(stream)
            d_pack[:packed].copy_(h_pack[:packed], non_blocking=True)
            e[3].record(stream)
            h_back.copy_(d_out, non_blocking=True)
            e[4].record(stream)
            e[4].synchronize()
            legs = [legs[j] + e[j].elapsed_time(e[j + 1]) / 3.0 for j in range(4)]
        for pl in cps + ups:
            pl.close()
        del d_in, d_comp, d_pack, d_comp2, d_out, h_raw, h_back, h_pack
        torch.cuda.empty_cache()
        return {"h2d_raw_ms": round(legs[0], 3), "d2h_streams_ms": round(legs[1], 3), "h2d_streams_ms": round(legs[2], 3),
                "d2h_raw_ms": round(legs[3], 3), "legs_ms": round(sum(legs), 3), "packed_bytes": packed, "chunks": k,
                "pipelined_trip_ms": round(wall, 3),
                "note": "pinned host memory <-> this GPU, both directions of the share; pipelined = copies on a second "
                        "stream beside the chunks' kernels, compress then uncompress (PCIe-inclusive: never `value`)"}

    def batch(tag, workload, n, size, level, do_c, do_u, foreign=None, nsteps=steps, l1_parse=-1, sample=0, pack=False):
        """sample: that many of the streams against oracle.compress at the same level, byte for byte (outside the
        timed region; the exact parse only); pack: time zh_plan_pack / zh_plan_unpack on the batch's streams."""
        eng.set_l1_parse(l1_parse)
        data = host.reshape(-1)[:n * size].reshape(n, size)
        d_src = torch.from_numpy(data.reshape(-1)).cuda()
        cap = size + size // 8 + 2048
        slot = (cap + 255) & ~255
        d_comp = torch.empty(n * slot, dtype=torch.uint8, device="cuda")
        d_back = torch.empty(n * size, dtype=torch.uint8, device="cuda")
        src_off = [i * size for i in range(n)]
        comp_off = [i * slot for i in range(n)]
        cplan = uplan = None
        comp_lens = None
        if foreign is None:
            cplan = eng.plan_compress(src_off, [size] * n, comp_off, [cap] * n, level, api.dfGzip)
            cplan.set_profiling(True)
        else:
            def gz(i):
                c = zlib.compressobj(foreign, zlib.DEFLATED, 31)
                return c.compress(data[i].tobytes()) + c.flush()
            with ThreadPoolExecutor(min(os.cpu_count() or 1, 32)) as ex:
                blobs = list(ex.map(gz, range(n)))
            comp_lens = [len(b) for b in blobs]
            stage = torch.zeros(n * slot, dtype=torch.uint8)
            sv = stage.numpy()
            for i, b in enumerate(blobs):
                sv[i * slot:i * slot + len(b)] = memoryview(b)
            d_comp.copy_(stage)
            del stage, sv, blobs
        if do_u or foreign is not None:
            uplan = eng.plan_uncompress(comp_off, comp_lens or [cap] * n, src_off, [size] * n, api.dfGzip)
            if comp_lens is None:
                uplan.set_src_lens_device(cplan.device_lens())
            uplan.set_profiling(True)
        state = {}

        def verify():
            if cplan is not None:
                clens, csts = cplan.results()
                assert all(x == 0 for x in csts), tag
                state["C"] = sum(clens)
            else:
                state["C"] = sum(comp_lens)
            if uplan is not None:
                ulens, usts = uplan.results()
                assert all(x == 0 for x in usts) and ulens == [size] * n, tag
                assert torch.equal(d_back, d_src), tag
            else:  # compress only: the streams must still decode (device decoder, CRC-32 checked)
                chk = eng.plan_uncompress(comp_off, [cap] * n, src_off, [size] * n, api.dfGzip)
                chk.set_src_lens_device(cplan.device_lens())
                chk.run(d_comp.data_ptr(), d_back.data_ptr())
                ulens, usts = chk.results()
                assert all(x == 0 for x in usts) and torch.equal(d_back, d_src), tag
                chk.close()

        # warm-up runs both plans that exist; the timed loop only the legs the config names
        time_plans(torch, stream, cplan, uplan, (d_src, d_comp, d_back), 0, 1, verify)
        tc, tu, kms = time_plans(torch, stream, cplan if do_c else None, uplan if do_u else None,
                                 (d_src, d_comp, d_back), nsteps, 0)
        out[tag] = entry(workload, n * size, state["C"], tc if do_c else None, tu if do_u else None, kms,
                         "headline" if tag == "c3_own" else tag)
        if sample and cplan is not None and l1_parse != 1:
            out[tag]["parity_sample"] = oracle_sample(cplan, d_comp, data, comp_off, level, sample)
        if pack and cplan is not None and uplan is not None:
            out[tag].update(pack_times(cplan, uplan, d_comp, d_back, d_src, n, slot))
            cplan.close()
            uplan.close()
            cplan = uplan = None
            del d_comp, d_back
            torch.cuda.empty_cache()
            d_comp = d_back = None
            # the trip as 1 (nothing overlaps: the baseline on equal footing), 2 and 4 chunks: a chunk's kernels are a
            # smaller batch's -- latency chains that do not shrink with it --, so more chunks overlap more and compute slower
            try:  # (a side measurement: whatever goes wrong with pinned memory on some box must not cost the line)
                trips = {c: pcie_times(d_src, n, size, cap, slot, level, chunks=c) for c in (1, 2, 4)}
                best = min(trips, key=lambda c: trips[c]["pipelined_trip_ms"])
                pc = trips[best]
                pc["trip_ms_by_chunks"] = {str(c): t["pipelined_trip_ms"] for c, t in trips.items()}
                pc["value_incl_pcie"] = round(n * size / GIB / ((out[tag]["ms_per_step"] + pc["legs_ms"]) * 1e-3), 3)
                pc["value_pipelined"] = round(n * size / GIB / (pc["pipelined_trip_ms"] * 1e-3), 3)
            except (RuntimeError, MemoryError, AssertionError) as e:
                pc = {"error": repr(e)[:300]}
                torch.cuda.synchronize()
            out[tag]["own_pcie_link"] = pc
        for pl in (cplan, uplan):
            if pl is not None:
                pl.close()
        del d_src, d_comp, d_back
        torch.cuda.empty_cache()
        eng.set_l1_parse(-1)

    nb = host.shape[0]
    full_size = nb == 4096  # (the PMC passes were taken on BASELINE's sizes)
    batch("c2", "c2: 1024x64KiB compress L1", min(1024, nb * 16), 65536, 1, True, False, sample=8)
    batch("c2_parallel_parse", "c2, contract mode", min(1024, nb * 16), 65536, 1, True, False, l1_parse=1)
    batch("c3_own", "c3: %dx1MiB uncompress, own L1 streams" % nb,
          nb, 1 << 20, 1, False, True)
    batch("c3_zlib6", "c3: %dx1MiB uncompress, zlib-6 gzip members" % nb, nb, 1 << 20, 1, False, True, foreign=6)
    share = max(1, nb // 8)
    batch("c4_share", "c4: one GPU's share of eight (%dx1MiB), compress L-1" % share,
          share, 1 << 20, -1, True, False, nsteps=max(2, steps // 2), sample=8)
    # one GPU's share of the HEADLINE step when eight split the batch (strong scaling, SCALE_rNN's N = 8 point as
    # far as one GPU can show it), with what the GPU adds to a transfer leg (pack / unpack)
    batch("share512", "one GPU's share of eight of the headline step (%dx1MiB)" % share, share, 1 << 20, 1, True, True, pack=True)
    batch("share512_parallel_parse", "the same share, contract mode", share, 1 << 20, 1,
          True, True, l1_parse=1)

    # config 5: ONE large buffer as independent 32 KiB deflate blocks (tools/bench_c5.py)
    mib = min(128, nb)
    size = mib << 20
    d_src = torch.from_numpy(host.reshape(-1)[:size]).cuda()
    cap = size + size // 8 + 1024 * (size // 32768 + 1) + 4096
    d_comp = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    d_back = torch.empty(size, dtype=torch.uint8, device="cuda")
    cplan = eng.plan_compress_blocks([0], [size], [0], [cap], 1, api.dfGzip, 32768)
    cplan.set_profiling(True)
    cplan.run(d_src.data_ptr(), d_comp.data_ptr())
    (clen,), (cst,) = cplan.results()
    assert cst == 0, cst
    uplan = eng.plan_uncompress_indexed(0, clen, 0, size, cplan.block_index(0), api.dfGzip)
    uplan.set_profiling(True)

    def verify5():
        (ulen,), (ust,) = uplan.results()
        assert ust == 0 and ulen == size and torch.equal(d_back, d_src), "config 5"
    tc, tu, kms = time_plans(torch, stream, cplan, uplan, (d_src, d_comp, d_back), steps, 1, verify5)
    out["c5"] = entry("c5: 1x%dMiB as 32KiB blocks, compress L1 + indexed uncompress" % mib, size, clen, tc, tu, kms, "c5")
    import oracle  # (checker only, outside the timed region: the first 8 MiB as 32 KiB blocks, bytes and index)
    part = 8 << 20
    if size >= part:
        want, windex = oracle.compress_blocks(host.reshape(-1)[:part].tobytes(), 1, oracle.dfGzip, 32768, fname_len=0)
        pplan = eng.plan_compress_blocks([0], [part], [0], [cap], 1, api.dfGzip, 32768)
        pplan.run(d_src.data_ptr(), d_comp.data_ptr())
        (plen,), (pst,) = pplan.results()
        got = d_comp[:plen].cpu().numpy().tobytes()
        out["c5"]["parity_sample"] = {"bytes": part, "identical": pst == 0 and got == want and
                                      [tuple(e) for e in pplan.block_index(0)] == [tuple(e) for e in windex],
                                      "against": "oracle.compress_blocks(level 1, gzip, 32 KiB): stream and block index"}
        pplan.close()
    cplan.close()
    uplan.close()
    try:
        out["unsized"] = unsized_streams(eng, api, synth, min(256, nb))
    except (RuntimeError, MemoryError, AssertionError) as e:
        out["unsized"] = {"error": repr(e)[:300]}
    return out


def unsized_streams(eng, api, synth, n, size=1 << 20, reps=3):
    """Streams that carry no size (dfZlib: zippy.nim:130-165, the reference grows `dst` as it inflates) through the
    HOST-buffer call zh_uncompress_batch, next to the same data as gzip members (ISIZE sizes the output up front).
    Markup that compresses ~ 6 x: every stream outgrows the 4 x guess, so what is timed is the guess's failure, the
    sizing pass (the tokens kernel's count-only form) and the second decode -- PCIe both ways included, like every
    host-buffer number (tools/bench_unsized.py is the same measurement at 1024 streams)."""
    import ctypes as c
    bufs = [b.tobytes() for b in synth.gen_batch("html", n, size)]

    def call(blobs, fmt):
        k = len(blobs)
        srcs = (c.c_void_p * k)(*[c.cast(c.c_char_p(z), c.c_void_p) for z in blobs])
        lens = (c.c_size_t * k)(*[len(z) for z in blobs])
        dsts, dlens, sts = (c.c_void_p * k)(), (c.c_size_t * k)(), (c.c_int32 * k)()
        t = time.perf_counter()
        rc = eng.lib.zh_uncompress_batch(eng._h, srcs, lens, k, fmt, dsts, dlens, sts)
        dt = time.perf_counter() - t
        ok = rc == 0 and not any(sts) and c.string_at(dsts[0], dlens[0]) == bufs[0] and \
            c.string_at(dsts[k - 1], dlens[k - 1]) == bufs[k - 1]
        for i in range(k):
            eng.lib.zh_free(dsts[i])
        assert ok, "unsized streams: zh_uncompress_batch"
        return dt
    res = {"workload": "%dx%dB html slices, host buffers, zh_uncompress_batch: zlib (no size field) vs gzip" % (n, size)}
    for name, fmt in (("gzip", api.dfGzip), ("zlib", api.dfZlib)):
        blobs, sts = eng.compress_batch(bufs, 1, fmt)
        assert all(x == 0 for x in sts)
        ts = [call(blobs, fmt) for _ in range(reps + 1)][1:]  # (the first call warms the context's device blocks up)
        res[name + "_ms"] = round(min(ts) * 1e3, 3)
        res["ratio"] = round(n * size / sum(len(z) for z in blobs), 3)
    res["value"] = round(n * size / GIB / (res["zlib_ms"] * 1e-3), 3)
    res["unit"] = "GiB/s"
    res["zlib_vs_gzip"] = round(res["zlib_ms"] / res["gzip_ms"], 3)
    return res


def pp_fields(pp, N, comp_rank_exact, roof):
    """The parallel-parse leg's fields of the result line (whole-job numbers; kernel times rank 0's)."""
    total, steps = pp["total"], pp["steps"]
    pk = pp["kernels"]
    return {
        "value_parallel_parse": round(total * steps / GIB / pp["elapsed"], 3),
        "ratio_parallel_parse": round(total / pp["comp_all"], 4),
        "parallel_parse": {
            "what": "same step, BestSpeed matcher = zh_l1p_match_kernel (not the reference's token stream; round "
                    "trip verified on device + zlib sample)",
            "ms_per_step": round(pp["elapsed"] * 1e3 / steps, 3),
            "compress_GiBps": round(total / GIB / (pp["tc"] * 1e-3), 3),
            "uncompress_GiBps": round(total / GIB / (pp["tu"] * 1e-3), 3),
            "size_vs_exact_parse": round(pp["comp_rank"] / comp_rank_exact, 5),
            "parity_sample": pp.get("parity_sample"),
            "kernels_ms": {k: round(v, 4) for k, v in sorted(pk.items(), key=lambda kv: -kv[1]) if k != "end"},
            "kernel_launches": {k: v for k, v in pp.get("launches", {}).items() if v > 1},
            "roofline_matcher": roof(N, pk["zh_l1p_match_kernel"], "zh_l1p_match_kernel",
                                     nl=pp.get("launches", {}).get("zh_l1p_match_kernel", 1)),
        },
    }


def relaunch_under_torchrun(args):
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus:
        raise SystemExit("bench.py --gpus %d: this box has %d GPU(s); refusing to report a %d-GPU number "
                         "from fewer devices" % (args.gpus, have, args.gpus))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--buffers", type=int, default=4096)
    ap.add_argument("--size", type=int, default=1 << 20)
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--scaling", choices=("strong", "weak"), default=None)
    ap.add_argument("--foreign", type=int, default=None, metavar="ZLIB_LEVEL",
                    help="uncompress streams made by system zlib at this level (multi-block, foreign "
                         "Huffman tables) instead of this library's own output; implies --uncompress-only")
    ap.add_argument("--uncompress-only", action="store_true",
                    help="BASELINE config 3: time only the uncompress pass (inflate + CRC-32)")
    ap.add_argument("--compress-only", action="store_true", help="BASELINE configs 2/4: time only the compress pass")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parallel-parse", action="store_true",
                    help="skip the second timing with the opt-in parallel BestSpeed parse")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip BASELINE configs 2-5 (run after the headline at N=1 with the default workload)")
    ap.add_argument("--no-transfer", action="store_true", help="skip the RCCL scatter/gather leg (N > 1)")
    ap.add_argument("--no-parity-sample", action="store_true",
                    help="skip the comparison of 64 of the streams with the oracle's (outside the timed region)")
    args = ap.parse_args()
    if args.foreign is not None:
        args.uncompress_only = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(args)

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d; launch one rank per GPU" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (zippy_amd has no CPU fallback)")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d has no GPU of its own (%d visible)" % (rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    # under torchrun the collective path is used even with one rank (exercises it on a 1-GPU box)
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus

    import synth
    from zippy_amd import api, sharding
    from zippy_amd._binding import Engine

    scaling = args.scaling or ("strong" if world > 1 else "weak")
    size = args.size
    if scaling == "strong":
        lo, hi = sharding.shard_range(args.buffers, rank, world)
        total_buffers = args.buffers
    else:
        lo, hi = rank * args.buffers, (rank + 1) * args.buffers
        total_buffers = args.buffers * world
    n = hi - lo
    # ---- synthetic batch (this rank's shard), staged into HBM ----
    t_gen = time.perf_counter()
    host = synth.gen_batch("mix", n, size, first_index=lo)
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
    if args.foreign is not None:
        # config 3's foreign set: gzip members made by system zlib (multi-block dynamic streams)
        import zlib
        from concurrent.futures import ThreadPoolExecutor

        def gz(i):
            c = zlib.compressobj(args.foreign, zlib.DEFLATED, 31)
            return c.compress(host[i].tobytes()) + c.flush()
        with ThreadPoolExecutor(min(os.cpu_count() or 1, 32)) as ex:  # (zlib releases the GIL)
            blobs = list(ex.map(gz, range(n)))
        comp_lens = [len(b) for b in blobs]
        assert max(comp_lens) <= cap
        stage = np.zeros((n, slot), dtype=np.uint8)
        for i, b in enumerate(blobs):
            stage[i, :len(b)] = np.frombuffer(b, dtype=np.uint8)
        d_comp.copy_(torch.from_numpy(stage.reshape(-1)))
        uplan = eng.plan_uncompress(comp_off, comp_lens, src_off, [size] * n, api.dfGzip)
        del stage, blobs
    else:
        uplan = eng.plan_uncompress(comp_off, [cap] * n, src_off, [size] * n, api.dfGzip)
        uplan.set_src_lens_device(cplan.device_lens())
    cplan.set_profiling(True)
    uplan.set_profiling(True)
    do_c = not args.uncompress_only
    do_u = not args.compress_only

    def step():  # warm-up passes always go both ways: the round trip is verified before timing
        if args.foreign is None:
            cplan.run(d_src.data_ptr(), d_comp.data_ptr())
        uplan.run(d_comp.data_ptr(), d_back.data_ptr())

    for _ in range(max(args.warmup, 1)):  # at least one untimed pass: its results are verified below
        step()
    torch.cuda.synchronize()

    # ---- correctness of what is about to be timed ----
    if args.foreign is None:
        clens, csts = cplan.results()
        assert all(s == 0 for s in csts), "compress statuses"
        comp_total = sum(clens)
    else:
        comp_total = sum(comp_lens)
    ulens, usts = uplan.results()
    assert all(s == 0 for s in usts), "uncompress statuses (CRC-32 / ISIZE verified on device)"
    assert ulens == [size] * n
    assert torch.equal(d_back, d_src), "round trip mismatch"
    # ---- the headline carries its own parity evidence: a sample of the streams about to be timed against the
    # oracle's compress(), byte for byte (the checker, outside the timed region; rank 0's shard) ----
    parity_sample = None
    if args.foreign is None and rank == 0 and not args.no_parity_sample:
        import oracle
        pick = list(range(0, n, max(1, n // 64)))[:64]
        same = 0
        for i in pick:
            z = d_comp[i * slot:i * slot + clens[i]].cpu().numpy().tobytes()
            same += z == oracle.compress(host[i].tobytes(), args.level, oracle.dfGzip, fname_len=0)
        parity_sample = {"n": len(pick), "identical": same == len(pick), "against": "oracle.compress(level %d, gzip)" % args.level,
                         "buffers": "every %d-th of this rank's %d" % (max(1, n // 64), n)}
        assert parity_sample["identical"], "device streams differ from the oracle's (%d of %d equal)" % (same, len(pick))

    # ---- timed region ----
    kernel_ms = {}
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t_comp = t_unc = 0.0
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ev[0].record(stream)
        if do_c:
            cplan.run(d_src.data_ptr(), d_comp.data_ptr())
        ev[1].record(stream)
        if do_u:
            uplan.run(d_comp.data_ptr(), d_back.data_ptr())
        ev[2].record(stream)
        ev[2].synchronize()
        t_comp += ev[0].elapsed_time(ev[1])
        t_unc += ev[1].elapsed_time(ev[2])
        collect_kernel_times(kernel_ms, (cplan if do_c else None, uplan if do_u else None))
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    comp_all = comp_total
    if use_dist:
        t = torch.tensor([elapsed, t_comp, t_unc], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, t_comp, t_unc = (float(x) for x in t.tolist())
        c = torch.tensor([comp_total], device="cuda", dtype=torch.int64)
        dist.all_reduce(c)
        comp_all = int(c.item())

    total_uncompressed = total_buffers * size
    value = total_uncompressed * args.steps / GIB / elapsed
    ms_per_step = elapsed * 1e3 / args.steps

    # ---- the same step with the opt-in PARALLEL BestSpeed parse (zh_set_l1_parse(ctx, 1),
    # csrc/zh_l1p_match.hip): valid streams that are not the reference's bytes, under the north star's
    # encoder contract (round trip exact, size within 2 % of zippy's).  `value` stays the
    # byte-identical parse.  Every rank, timed like the headline (barrier, max over ranks). ----
    pp = None
    if do_c and do_u and args.foreign is None and args.level == 1 and not args.no_parallel_parse:
        import zlib
        eng.set_l1_parse(1)
        pstate = {}

        def verify_pp():
            plens, psts = cplan.results()
            assert all(x == 0 for x in psts), "parallel parse: compress statuses"
            ul, us = uplan.results()
            assert all(x == 0 for x in us) and ul == [size] * n, "parallel parse: uncompress statuses"
            assert torch.equal(d_back, d_src), "parallel parse: round trip mismatch"
            for i in range(0, n, max(1, n // 8)):  # a sample through system zlib as well
                z = d_comp[i * slot:i * slot + plens[i]].cpu().numpy().tobytes()
                assert zlib.decompress(z, 31) == host[i].tobytes(), "parallel parse: zlib disagrees"
            if rank == 0 and not args.no_parity_sample:
                # ... and 64 through the oracle's uncompress() (the reference's decoder restated): the contract
                # this parse is held to is "zippy's own uncompress() gives the input back", not byte identity
                import oracle
                pick = list(range(0, n, max(1, n // 64)))[:64]
                okc = 0
                for i in pick:
                    z = d_comp[i * slot:i * slot + plens[i]].cpu().numpy().tobytes()
                    okc += oracle.uncompress(z, oracle.dfGzip) == host[i].tobytes()
                pstate["parity_sample"] = {"n": len(pick), "round_trips_through_oracle_uncompress": okc == len(pick)}
                assert okc == len(pick), "parallel parse: the oracle's uncompress() disagrees"
            pstate["C"] = sum(plens)
        time_plans(torch, stream, cplan, uplan, (d_src, d_comp, d_back), 0, max(args.warmup, 1), verify_pp)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ptc, ptu, pk = time_plans(torch, stream, cplan, uplan, (d_src, d_comp, d_back), args.steps, 0)
        pp_launches = dict(time_plans.launches)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        p_elapsed = time.perf_counter() - t0
        eng.set_l1_parse(-1)
        p_comp = pstate["C"]
        if use_dist:
            t = torch.tensor([p_elapsed, ptc, ptu], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            p_elapsed, ptc, ptu = (float(x) for x in t.tolist())
            c = torch.tensor([p_comp], device="cuda", dtype=torch.int64)
            dist.all_reduce(c)
            p_comp = int(c.item())
        pp = {"elapsed": p_elapsed, "tc": ptc, "tu": ptu, "kernels": pk, "comp_all": p_comp,
              "comp_rank": pstate["C"], "total": total_uncompressed, "steps": args.steps, "launches": pp_launches,
              "parity_sample": pstate.get("parity_sample")}

    # ---- N > 1: the batch lives on rank 0 and comes home to rank 0 (RCCL over xGMI) ----
    transfer = None
    if use_dist and scaling == "strong" and not args.no_transfer and do_c and do_u:
        try:  # (a side measurement, never run on more than one GPU before the driver does: it must not cost the line)
            transfer = transfer_leg(torch, dist, sharding, synth, rank, world, args.buffers, size, slot, lo, hi,
                                    d_src, d_comp, d_back, cplan, uplan, stream, t_comp / args.steps,
                                    t_unc / args.steps)
        except Exception as e:  # noqa: BLE001
            transfer = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}

    if rank == 0:
        avg, launches = kernel_averages(kernel_ms)  # a kernel's launches of a step together, and how many they are
        dom = max((k for k in avg if k.startswith("zh_")), key=lambda k: avg[k])
        # Algorithmic bytes of this rank's shard: a pass moves every uncompressed byte
        # once and every compressed byte once (N + C); a kernel is charged what IT must move:
        # matcher N (reads the source), emit N + C, inflate C + N, checksum N.
        N, C = n * size, comp_total
        own = own_bytes(N, C)
        traffic = hbm_traffic("headline") if (n, size, args.level) == (4096, 1 << 20, 1) and args.foreign is None else {}
        traffic_raw = (hbm_traffic("headline", "hbm_bytes_per_launch_uncorrected")
                       if (n, size, args.level) == (4096, 1 << 20, 1) and args.foreign is None else {})

        def roof(nbytes, ms, name=None, nl=None):
            """nbytes / ms: of all launches of the kernel in a step together; reported a launch (`nl` of them), like the
            rocprofv3 statistics and the PMC counters under profiles/ are."""
            nl = max(1, nl if nl is not None else launches.get(name, 1))
            a = nbytes / (ms * 1e-3)
            r = {"bound": "hbm", "achieved": round(a / 1e9, 3), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                 "frac": round(a / HBM_PEAK, 6), "frac_of_copy_peak": round(a / HBM_COPY_PEAK, 6),
                 "traffic": traffic.get(name) if name else None,
                 "traffic_raw": traffic_raw.get(name) if name else None,
                 "algorithmic_bytes_per_launch": nbytes // nl, "avg_launch_ms": round(ms / nl, 4), "launches_per_step": nl}
            if name:
                r["kernel"] = name
            return r

        mode = "compress + uncompress" if do_c and do_u else "compress only" if do_c else "uncompress only"
        out = {
            "metric": "GiB/s uncompressed throughput (compress BestSpeed + uncompress), 4096x1 MiB batch",
            "value": round(value, 3),
            "unit": "GiB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": scaling if world > 1 or args.scaling else None,  # one GPU: nothing is scaled
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic (G-mix: seeded slices of the reference's own test corpus, SURVEY.md 8d)",
            "config": {
                "workload": "%d x %d B in total (%d on this rank), %s, level %d gzip%s, CRC-32 verified, "
                            "buffers resident in HBM" % (
                                total_buffers, size, n, mode, args.level,
                                "" if args.foreign is None else "; streams made by system zlib level %d" % args.foreign),
                "buffers_total": total_buffers, "buffers_per_gpu": n, "buffer_bytes": size, "level": args.level,
                "sharding": "contiguous index ranges per rank (shard_range), no data-path collective",
            },
            "compress_GiBps": round(total_uncompressed * args.steps / GIB / (t_comp * 1e-3), 3) if do_c else None,
            "uncompress_GiBps": round(total_uncompressed * args.steps / GIB / (t_unc * 1e-3), 3) if do_u else None,
            "ratio": round(total_uncompressed / comp_all, 4),
            "kernels_ms": {k: round(v, 4) for k, v in sorted(avg.items(), key=lambda kv: -kv[1])},
            "kernel_launches": {k: v for k, v in launches.items() if v > 1},
            "kernels_note": "ms a step, all launches of a kernel together; DESIGN.md 5 says how to read it",
            "roofline": roof(own.get(dom, N + C), avg[dom], dom),
            "roofline_passes": {},
            # (the checksum kernels run in both passes, N bytes each time: a launch reads `b`, all of them `b` x launches)
            "roofline_kernels": {k: roof(b * (launches.get(k, 1) if k.startswith("zh_checksum") else 1), avg[k], k)
                                 for k, b in own.items() if k in avg},
            "source_sha": source_sha(),
            "traffic_note": "L2 memory-side bytes a launch, read requests by size + WRITE_SIZE (profiles/hbm_traffic.json; DESIGN.md 5)",
            "parity_sample": parity_sample,
        }
        if do_c:
            out["roofline_passes"]["compress"] = roof(N + C, t_comp / args.steps, nl=1)
        if do_u:
            out["roofline_passes"]["uncompress"] = roof(N + C, t_unc / args.steps, nl=1)
        if transfer:
            out["transfer"] = transfer
            out["value_incl_transfer"] = transfer.get("value_incl_transfer")
        headline = (world == 1 and do_c and do_u and args.foreign is None and args.level == 1
                    and size == 1 << 20 and n >= 8)
        if pp:
            out.update(pp_fields(pp, N, comp_total, roof))
        if headline and not args.no_configs:
            # the other BASELINE configs, so that the driver's line carries all five
            cplan.close()
            uplan.close()
            del d_comp, d_back, d_src
            torch.cuda.empty_cache()
            out["configs"] = side_configs(torch, eng, api, synth, stream, host, max(2, min(args.steps, 5)))
            # strong-scaling proxy: the share's time against a perfect eighth of the full batch's step
            for tag, full_ms in (("share512", ms_per_step),
                                 ("share512_parallel_parse", out.get("parallel_parse", {}).get("ms_per_step"))):
                e = out["configs"].get(tag)
                if e and full_ms and n == 8 * (n // 8):
                    e["perfect_eighth_ms"] = round(full_ms / 8.0, 3)
                    e["efficiency_vs_perfect_eighth"] = round(full_ms / 8.0 / e["ms_per_step"], 4)
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            # every host core this process may run on (no cap); a repetition is >= 2 core-seconds of
            # work a core pair, i.e. ~ 32 MiB a thread, so that thread start-up and a straggler are
            # a few percent of it: ~ 20 s in all
            try:
                cores = len(os.sched_getaffinity(0))
            except AttributeError:
                cores = os.cpu_count() or 1
            per_core = max(1, (32 << 20) // size)
            sample = [host[i].tobytes() for i in range(min(n, 2048, cores * per_core))]
            out["cpu_baseline"] = cpu_baseline(sample, args.level, cores)
            out["cpu_baseline"]["nproc"] = os.cpu_count()
            if "configs" in out and "c4_share" in out["configs"]:
                # config 4 "ratio vs CPU zippy reported": the oracle at DefaultCompression on the same threads, a
                # bounded sample (level -1 runs at ~ 30 MB/s a thread)
                out["configs"]["c4_share"]["cpu_baseline_level_-1"] = cpu_level_leg(
                    [host[i].tobytes() for i in range(min(n, max(32, out["cpu_baseline"]["cores"])))], -1,
                    out["cpu_baseline"]["cores"])
                # (the device's streams ARE the oracle's at this level -- parity_sample above --, so "ratio vs CPU
                # zippy" is 1 stream for stream; the entry's own ratio is over all of the share's buffers, this one
                # over the sample's)
                out["configs"]["c4_share"]["cpu_baseline_level_-1"]["device_vs_cpu_size"] = (
                    "identical streams" if out["configs"]["c4_share"].get("parity_sample", {}).get("identical") else None)
        out["host_gen_s"] = round(t_gen, 1)
        # the headline-adjacent numbers once more, compact and LAST: a reader who only sees the line's tail sees these
        cf = out.get("configs", {})

        def cv(tag, key="value"):
            return cf.get(tag, {}).get(key)
        out["summary"] = {
            "value": out["value"], "ms": out["ms_per_step"], "c_GiBps": out["compress_GiBps"], "u_GiBps": out["uncompress_GiBps"],
            "value_pp": out.get("value_parallel_parse"), "pp_ms": out.get("parallel_parse", {}).get("ms_per_step"),
            "frac": out["roofline"]["frac"], "dom": out["roofline"].get("kernel"), "dom_ms": out["roofline"]["avg_launch_ms"],
            "c2": cv("c2"), "c2_pp": cv("c2_parallel_parse"), "c3_own": cv("c3_own"), "c3_zlib6": cv("c3_zlib6"),
            "unsized_zlib_vs_gzip": cv("unsized", "zlib_vs_gzip"), "c4_share": cv("c4_share"), "c5": cv("c5"),
            "share512_ms": cv("share512", "ms_per_step"), "share512_eff": cv("share512", "efficiency_vs_perfect_eighth"),
            "share512_pp_ms": cv("share512_parallel_parse", "ms_per_step"),
            "share512_pp_eff": cv("share512_parallel_parse", "efficiency_vs_perfect_eighth"),
            "share512_pcie_pipelined_ms": (cv("share512", "own_pcie_link") or {}).get("pipelined_trip_ms"),
            "cpu_GiBps": out.get("cpu_baseline", {}).get("value"), "cpu_cores": out.get("cpu_baseline", {}).get("cores"),
            "parity": (parity_sample or {}).get("identical"),
        }
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


def transfer_leg(torch, dist, sharding, synth, rank, world, n_total, size, slot, lo, hi, d_src, d_comp, d_back,
                 cplan, uplan, stream, t_comp_ms, t_unc_ms):
    """The whole batch starts and ends on rank 0: raw buffers out (scatter_fixed), compressed
    streams home (gather_variable), compressed streams out again (scatter_variable), raw results
    home (gather_fixed).  Whole buffers only, point-to-point from / to the root over its xGMI
    links; nothing is reduced.  Timed once after a warm-up pass, max over ranks."""
    n = hi - lo
    dev = d_src.device
    whole = None
    if rank == 0:
        whole = torch.from_numpy(synth.gen_batch("mix", n_total, size).reshape(-1)).to(dev)

    def timed(fn):
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return r, float(t.item()) * 1e3

    res = {}
    for it in range(2):  # pass 0 warms RCCL's channels up
        mine, res["scatter_raw_ms"] = timed(lambda: sharding.scatter_fixed(whole, n_total, size, root=0, device=dev))
        assert torch.equal(mine, d_src)
        cplan.run(mine.data_ptr(), d_comp.data_ptr())
        clens, csts = cplan.results()
        assert all(s == 0 for s in csts)

        # the streams alone go over the links, not their worst-case slots: packed back to back on the device
        # (sharding.pack_plan -> zh_plan_pack: two launches, the lengths are the ones the run left on the device; no
        # per-buffer copy anywhere in this leg)
        (packed, lens_t), res["pack_ms"] = timed(lambda: sharding.pack_plan(cplan, d_comp, n))
        assert lens_t.tolist() == clens
        (all_c, all_lens), res["gather_compressed_ms"] = timed(
            lambda: sharding.gather_variable(packed, lens_t, root=0))
        (mine_c, mine_lens), res["scatter_compressed_ms"] = timed(
            lambda: sharding.scatter_variable(all_c, all_lens if rank == 0 else None, n_total, root=0, device=dev))
        assert torch.equal(mine_c, packed) and mine_lens.tolist() == clens
        d_comp.zero_()
        _keep, res["unpack_ms"] = timed(lambda: sharding.unpack_into_plan(uplan, mine_c, mine_lens, d_comp))
        uplan.run(d_comp.data_ptr(), d_back.data_ptr())
        _, usts = uplan.results()
        assert all(s == 0 for s in usts)
        home, res["gather_raw_ms"] = timed(lambda: sharding.gather_fixed(d_back, n_total, size, root=0))
        if rank == 0:
            assert torch.equal(home, whole), "batch did not come home intact"
        del home, all_c, mine_c, packed, mine
    res = {k: round(v, 3) for k, v in res.items()}
    res["scatter_ms"] = round(res["scatter_raw_ms"] + res["scatter_compressed_ms"], 3)
    res["gather_ms"] = round(res["gather_compressed_ms"] + res["gather_raw_ms"], 3)
    moved = sum(res[k] for k in ("scatter_raw_ms", "pack_ms", "gather_compressed_ms", "scatter_compressed_ms",
                                 "unpack_ms", "gather_raw_ms"))
    res["value_incl_transfer"] = round(n_total * size / GIB / ((t_comp_ms + t_unc_ms + moved) * 1e-3), 3)
    res["note"] = ("batch on rank 0 -> shards -> compressed home -> compressed out -> raw home; "
                   "value_incl_transfer = N_total / (T_compress + T_uncompress + all six transfer legs)")
    return res


if __name__ == "__main__":
    main()
