#!/usr/bin/env python3
"""ONE buffer of more than 4 GiB through compress(BestSpeed, gzip) + uncompress on the GPU (device-resident plans): the
size at which every 32-bit position, bit count and ISIZE in the path wraps.  Checked: statuses, the round trip on the
device, the gzip trailer (CRC-32, ISIZE = length mod 2^32, zippy.nim:71-78), the stream through system zlib, and --
byte for byte -- the oracle's compress() of the same buffer (the checker; one host thread, ~ 30 s).
Also reported: whether the stream was DECODED by many workgroups (segment-wise, zh_inflate_seg.hip) -- a chain of
segments that does not hold means one workgroup for the whole stream, the same bytes a hundred times slower.
    python tools/gpu_big_buffer.py [--mib 4100] [--level 1] [--no-oracle]"""
import argparse
import json
import os
import struct
import sys
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(mib, level=1, with_oracle=True, with_zlib=True, kind="mix", foreign=None):
    """kind: synth.gen_batch's (mix, text, rand, zeros, runs ...); foreign: the stream is made by system zlib at that
    level instead (one gzip member; only the decoder is under test then)."""
    import argparse as _a
    args = _a.Namespace(mib=mib, level=level, no_oracle=not with_oracle)
    import numpy as np
    import torch
    import synth
    from zippy_amd import api
    from zippy_amd._binding import Engine
    n = args.mib * (1 << 20) + 12345  # (not a multiple of anything)
    t = time.perf_counter()
    host = np.empty(n, dtype=np.uint8)
    per = 512
    for i in range(0, args.mib, per):  # G-mix, 512 MiB at a time
        k = min(per, args.mib - i)
        host[i << 20:(i + k) << 20] = synth.gen_batch(kind, k, 1 << 20, first_index=i).reshape(-1)
    host[args.mib << 20:] = np.arange(12345, dtype=np.uint32).astype(np.uint8)
    t_gen = time.perf_counter() - t
    stream = torch.cuda.current_stream()
    eng = Engine(api.LIB_PATH, stream=stream.cuda_stream)
    eng.set_gzip_fname_len(0)
    d_src = torch.from_numpy(host).cuda()
    cap = eng.compress_bound(n, api.dfGzip)
    d_comp = torch.zeros(cap + 256, dtype=torch.uint8, device="cuda")
    d_back = torch.empty(n + 256, dtype=torch.uint8, device="cuda")
    t_c = []
    if foreign is not None:
        c = zlib.compressobj(foreign, zlib.DEFLATED, 31)
        zz = c.compress(host) + c.flush()
        clens = [len(zz)]
        d_comp[:len(zz)] = torch.frombuffer(bytearray(zz), dtype=torch.uint8).cuda()
        del zz
    else:
        cplan = eng.plan_compress([0], [n], [0], [cap], args.level, api.dfGzip)
    torch.cuda.synchronize()
    for _ in range(0 if foreign is not None else 2):  # (a plan's first run also pays for its scratch)
        d_comp.zero_()
        torch.cuda.synchronize()
        t = time.perf_counter()
        cplan.run(d_src.data_ptr(), d_comp.data_ptr())
        clens, csts = cplan.results()
        t_c.append(time.perf_counter() - t)
        assert csts == [0], csts
    uplan = eng.plan_uncompress([0], clens, [0], [n], api.dfGzip)
    uplan.set_profiling(True)
    t_u = []
    for _ in range(2):
        torch.cuda.synchronize()
        t = time.perf_counter()
        uplan.run(d_comp.data_ptr(), d_back.data_ptr())
        ulens, usts = uplan.results()
        t_u.append(time.perf_counter() - t)
        assert usts == [0] and ulens == [n], (usts, ulens)
    kernels = {k: round(v, 2) for k, v in uplan.kernel_times() if v > 0.004}
    assert torch.equal(d_back[:n], d_src), "round trip on the device"
    z = d_comp[:clens[0]].cpu().numpy().tobytes()
    crc, isize = struct.unpack("<II", z[-8:])
    want_crc = 0
    for i in range(0, n, 1 << 28):
        want_crc = zlib.crc32(host[i:i + (1 << 28)], want_crc)
    assert crc == want_crc and isize == n % (1 << 32), (crc, want_crc, isize)
    if with_zlib:
        d = zlib.decompressobj(31)
        off = 0
        for i in range(0, len(z), 1 << 26):
            out = d.decompress(z[i:i + (1 << 26)])
            assert out == host[off:off + len(out)].tobytes(), "system zlib disagrees at %d" % off
            off += len(out)
        out = d.flush()
        assert out == host[off:off + len(out)].tobytes() and off + len(out) == n and d.eof
    cut, held = eng.segment_stats()
    res = {"bytes": n, "level": args.level, "compressed_bytes": clens[0], "compress_s": [round(x, 3) for x in t_c],
           "uncompress_s": [round(x, 3) for x in t_u], "uncompress_kernels_ms": kernels, "gen_s": round(t_gen, 1), "zlib_ok": True if with_zlib else None, "trailer_ok": True,  # (None: the zlib check was skipped, not failed)
           "decoded_segment_wise": held == cut and cut >= 2}  # (both runs: by many workgroups, zh_debug_segment_stats)
    res["kind"] = kind
    if foreign is not None:
        res["made_by"] = "system zlib level %d" % foreign
    if not args.no_oracle and foreign is None:
        import oracle
        t = time.perf_counter()
        ref = oracle.compress(host, args.level, oracle.dfGzip, fname_len=0)
        res["oracle_s"] = round(time.perf_counter() - t, 1)
        res["identical_to_oracle"] = ref == z
        assert ref == z, "device stream differs from the oracle's (%d vs %d bytes)" % (len(z), len(ref))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=4100)
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--no-zlib", action="store_true")
    ap.add_argument("--kind", default="mix")
    ap.add_argument("--foreign", type=int, default=None)
    args = ap.parse_args()
    print(json.dumps(run(args.mib, args.level, not args.no_oracle, not args.no_zlib, args.kind, args.foreign)))


if __name__ == "__main__":
    main()
