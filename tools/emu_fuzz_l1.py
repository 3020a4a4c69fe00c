#!/usr/bin/env python3
"""CPU-only stress of the BestSpeed matcher logic (kernel sources under the emulator,
tests/hipemu) against the oracle on inputs chosen to hit the rare walk paths: short
periods (in-step candidates), tiny alphabets (slot collisions), long literal runs
(sparse schedule), runs/zeros, G-mix.  Usage: python tools/emu_fuzz_l1.py [seed]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import emu  # noqa: E402
import oracle  # noqa: E402
import synth  # noqa: E402


def main(seed):
    eng = emu.engine()
    rnd = random.Random(seed)
    bufs = []
    for kind in ("runs", "rand", "zero"):
        bufs += [b.tobytes() for b in synth.gen_batch(kind, 2, 100000, first_index=seed * 7)]
    for per in (1, 2, 3, 4, 5, 7, 8, 13, 31, 32, 33, 63, 64, 65, 100, 257, 300):
        pat = rnd.randbytes(per)
        b = bytearray()
        while len(b) < 70000:
            b += pat * rnd.randrange(1, 40)
            if rnd.random() < 0.3:
                b += rnd.randbytes(rnd.randrange(1, 50))
        bufs.append(bytes(b[:70000]))
    for alpha in (2, 3, 4, 8, 16):
        bufs.append(bytes(rnd.randrange(alpha) for _ in range(80000)))
    # text with random noise islands (literal runs of every length next to matches)
    text = synth.corpus_file("alice29.txt")
    b = bytearray()
    while len(b) < 200000:
        o = rnd.randrange(len(text) - 4000)
        b += text[o:o + rnd.randrange(10, 4000)]
        b += rnd.randbytes(rnd.choice((0, 1, 5, 20, 31, 32, 33, 40, 64, 100, 500)))
    bufs.append(bytes(b))
    bufs += [b.tobytes() for b in synth.gen_batch("mix", 4, 1 << 20, first_index=1000 + seed)]
    # ragged tails around the 15-byte rule
    bufs += [text[100:100 + k] for k in range(0, 80)]
    outs, sts = eng.compress_batch(bufs, 1, oracle.dfDeflate)
    bad = 0
    for i, (src, out, st) in enumerate(zip(bufs, outs, sts)):
        ref = oracle.deflate(src, 1)
        if st != 0 or out != ref:
            bad += 1
            print("MISMATCH", i, len(src), st, len(out or b""), len(ref))
    print("seed", seed, "checked", len(bufs), "bad", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 1) else 0)
