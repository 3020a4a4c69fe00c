#!/usr/bin/env python3
"""GPU-box stress for the chain levels (2..9, -1): the inputs of tools/gpu_fuzz.py plus buffers of several
MiB (more than one deflate block), every level, byte for byte against the oracle.
    python tools/gpu_fuzz_chain.py [first_seed] [n_seeds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle  # noqa: E402
import synth  # noqa: E402
from zippy_amd import api  # noqa: E402
from gpu_fuzz import inputs  # noqa: E402


def main(first, count):
    eng = api.engine()
    eng.set_gzip_fname_len(0)
    bad = 0
    for seed in range(first, first + count):
        bufs, rnd = inputs(seed)
        bufs = [b for b in bufs if len(b) > 64][:24]
        for level in (-1, 2, 3, 4, 5, 6, 7, 8, 9):
            pick = bufs if level in (-1, 5) else rnd.sample(bufs, 8)
            outs, sts = eng.compress_batch(pick, level, oracle.dfDeflate)
            for src, out, st in zip(pick, outs, sts):
                if st != 0 or out != oracle.compress(src, level, oracle.dfDeflate):
                    bad += 1
                    print("LEVEL MISMATCH seed", seed, "level", level, len(src), st)
        # several MiB: windows wrap many times, the last buffer is two deflate blocks
        big = [synth.gen_batch("mix", 3, 1 << 20, first_index=seed * 3)[k].tobytes() for k in range(3)]
        big.append(b"".join(big) + synth.gen_batch("runs", 1, 1500000, first_index=seed)[0].tobytes())
        for level in (-1, rnd.choice((2, 4, 5, 7))):
            outs, sts = eng.compress_batch(big, level, oracle.dfGzip)
            for src, out, st in zip(big, outs, sts):
                if st != 0 or out != oracle.compress(src, level, oracle.dfGzip, fname_len=0):
                    bad += 1
                    print("BIG MISMATCH seed", seed, "level", level, len(src), st)
        print("seed", seed, "ok so far" if not bad else "BAD %d" % bad, flush=True)
    print("gpu_fuzz_chain: seeds %d..%d bad %d" % (first, first + count - 1, bad))
    return bad


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    sys.exit(1 if main(a[0] if a else 1, a[1] if len(a) > 1 else 10) else 0)
