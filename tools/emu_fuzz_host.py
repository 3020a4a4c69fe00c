#!/usr/bin/env python3
"""CPU-only stress of the host-buffer calls' staging logic (kernel sources under the emulator,
tests/hipemu; 128 KiB staging chunks, 3 host threads, 150 000-byte pipeline groups -- tests/emu.py):
random batch shapes -- empty buffers, buffers of several chunks, many tiny ones -- at random
levels and formats, one plan against pipelined groups against the oracle, then back through
uncompress with damaged members in between.  Usage: python tools/emu_fuzz_host.py [seed] [rounds]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import emu  # noqa: E402
import oracle  # noqa: E402
import synth  # noqa: E402


def main(seed, rounds):
    eng = emu.engine()
    pool = synth.gen_batch("mix", 4, 1 << 20, first_index=seed).tobytes()
    bad = 0
    for r in range(rounds):
        rnd = random.Random(seed * 1000 + r)
        n = rnd.choice((1, 2, 3, 7, 20, 60))
        budget = rnd.choice((3000, 200000, 900000))
        bufs = []
        for _ in range(n):
            kind = rnd.random()
            sz = 0 if kind < 0.15 else rnd.randrange(1, 64) if kind < 0.3 else rnd.randrange(1, max(2, 2 * budget // n))
            o = rnd.randrange(len(pool) - sz)
            bufs.append(pool[o:o + sz])
        level = rnd.choice((1, 1, 1, 0, -2, -1))
        fmt = rnd.choice((oracle.dfGzip, oracle.dfZlib, oracle.dfDeflate))
        try:
            eng.set_host_pipeline(1 << 60, 0)
            one, st1 = eng.compress_batch(bufs, level, fmt)
            eng.set_host_pipeline(1, rnd.choice((70000, 150000, 400000)))
            grp, st2 = eng.compress_batch(bufs, level, fmt)
        finally:
            eng.set_host_pipeline(0, 0)
        want = [oracle.compress(b, level, fmt, fname_len=0) for b in bufs]
        if any(st1) or any(st2) or one != want or grp != want:
            bad += 1
            print("COMPRESS MISMATCH round", r, n, level, fmt)
            continue
        blobs = list(want)
        hurt = set(rnd.sample(range(n), min(n, rnd.choice((0, 0, 1, 3)))))
        for i in hurt:
            b = bytearray(blobs[i])
            if len(b) > 12:
                b[rnd.randrange(10, len(b))] ^= 1 << rnd.randrange(8)
            blobs[i] = bytes(b)
        outs, sts = eng.uncompress_batch(blobs, fmt)
        for i in range(n):
            try:
                ref = oracle.uncompress(blobs[i], fmt)
            except oracle.ZippyError:
                ref = None
            if (outs[i] if sts[i] == 0 else None) != ref:
                bad += 1
                print("UNCOMPRESS MISMATCH round", r, "buffer", i, sts[i], len(blobs[i]))
    print("seed", seed, "rounds", rounds, "bad", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 1,
                       int(sys.argv[2]) if len(sys.argv) > 2 else 30) else 0)
