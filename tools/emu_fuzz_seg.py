#!/usr/bin/env python3
"""The segment-wise decoder's handling of found starts that are none, under the CPU emulator (tests/hipemu): streams of
long blocks (this library's, block-parallel form) and of short ones (system zlib), whole and damaged, each with a false
start planted at a random bit (ZH_SEG_FAKE_START) -- bytes and statuses have to be the one-workgroup decoder's
(ZH_SEG=0), whatever the repair rounds make of it.    python tools/emu_fuzz_seg.py [first_seed] [count]"""
import os
import random
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["ZH_SEG_SETUP"] = "0"
import emu  # noqa: E402
import oracle  # noqa: E402
import synth  # noqa: E402


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    eng = emu.engine()
    bad = sound = sound_held = 0
    for seed in range(first, first + count):
        rng = random.Random(seed)
        kind = rng.choice(["text", "mix", "runs"])
        n = rng.randrange(90_000, 260_000)
        src = synth.gen_batch(kind, 1, n, first_index=seed)[0].tobytes()
        if rng.random() < 0.3:  # a stored stretch in the middle
            cut = rng.randrange(n // 4, n // 2)
            src = src[:cut] + synth.gen_batch("rand", 1, rng.randrange(2_000, 40_000), first_index=seed)[0].tobytes() + src[cut:]
        how = rng.randrange(3)
        if how == 0:
            blob, _ = eng.compress_blocks(src, rng.choice([1, 6]), oracle.dfGzip, rng.choice([32768, 65536, 131072]))
        elif how == 1:
            blob = eng.compress_batch([src], rng.choice([1, -1]), oracle.dfGzip)[0][0]
        else:
            c = zlib.compressobj(rng.choice([1, 6, 9]), zlib.DEFLATED, 31)
            blob = c.compress(src) + c.flush()
        damaged = rng.random() < 0.35
        if damaged:  # a flipped bit or a cut
            b = bytearray(blob)
            if rng.random() < 0.5:
                b[rng.randrange(10, len(b) - 8)] ^= 1 << rng.randrange(8)
            else:
                b = b[:rng.randrange(len(b) // 2, len(b) - 1)]
            blob = bytes(b)
        seg_bytes = rng.choice([600, 900, 2048])
        if len(blob) < 4 * seg_bytes:
            continue
        os.environ["ZH_SEG_MIN"] = str(4 * seg_bytes)
        os.environ["ZH_SEG_BYTES"] = str(seg_bytes)
        os.environ.pop("ZH_SEG_FAKE_START", None)
        os.environ["ZH_SEG"] = "0"
        want = eng.uncompress_batch([blob], oracle.dfGzip)
        os.environ["ZH_SEG"] = "1"
        for _ in range(3):
            os.environ["ZH_SEG_FAKE_START"] = str(rng.randrange(64, len(blob) * 8 - 64))
            before = eng.segment_stats()
            got = eng.uncompress_batch([blob], oracle.dfGzip)
            cut, held = (a - b for a, b in zip(eng.segment_stats(), before))
            if not damaged and cut:  # (a sound stream that is left to one workgroup: right bytes, a hundred times slower)
                sound += 1
                sound_held += held == cut
                if held != cut and os.environ.get("ZH_FUZZ_VERBOSE"):
                    print("seed %d: kind %s how %d seg %d fake %s: %d bytes not decoded segment-wise" % (
                        seed, kind, how, seg_bytes, os.environ["ZH_SEG_FAKE_START"], len(blob)), flush=True)
            if got[1] != want[1] or (want[1] == [0] and got[0] != want[0]):
                bad += 1
                print("seed %d: kind %s how %d seg %d fake %s: status %s / %s, bytes equal %s" % (
                    seed, kind, how, seg_bytes, os.environ["ZH_SEG_FAKE_START"], got[1], want[1], got[0] == want[0]), flush=True)
        if seed % 10 == 9:
            print("seed %d ok so far (%d bad)" % (seed, bad), flush=True)
    cut, held = eng.segment_stats()
    print("emu_fuzz_seg: seeds %d..%d bad %d; streams cut %d, chains held %d; sound streams %d, decoded segment-wise %d" % (
        first, first + count - 1, bad, cut, held, sound, sound_held))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
