#!/usr/bin/env python3
"""GPU-box stress (not part of pytest): the same input families as tools/emu_fuzz_l1.py through the
real library, several seeds -- BestSpeed bytes against the oracle, round trips, foreign (zlib
levels 1..9, raw/zlib/gzip) streams, and damaged streams whose accept/reject decision and bytes
must equal the oracle's.   python tools/gpu_fuzz.py [first_seed] [n_seeds]"""
import os
import random
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import synth  # noqa: E402
from zippy_amd import api  # noqa: E402


def inputs(seed):
    rnd = random.Random(seed)
    bufs = []
    for kind in ("runs", "rand", "zero", "mix"):
        bufs += [b.tobytes() for b in synth.gen_batch(kind, 2, 120000, first_index=seed * 7)]
    for per in (1, 2, 3, 5, 8, 13, 32, 33, 64, 65, 257, 300):
        pat = rnd.randbytes(per)
        b = bytearray()
        while len(b) < 70000:
            b += pat * rnd.randrange(1, 40)
            if rnd.random() < 0.3:
                b += rnd.randbytes(rnd.randrange(1, 50))
        bufs.append(bytes(b[:70000]))
    for alpha in (2, 4, 16, 200):
        bufs.append(bytes(rnd.randrange(alpha) for _ in range(60000)))
    text = synth.corpus_file(rnd.choice(["alice29.txt", "html", "urls.10K", "kppkn.gtb", "geo.protodata"]))
    b = bytearray()
    while len(b) < 300000:
        o = rnd.randrange(max(1, len(text) - 4000))
        b += text[o:o + rnd.randrange(10, 4000)]
        b += rnd.randbytes(rnd.choice((0, 1, 5, 20, 31, 32, 33, 40, 64, 100, 500)))
    bufs.append(bytes(b))
    bufs += [text[100:100 + k] for k in range(0, 40)]
    return bufs, rnd


def main(first, count):
    eng = api.engine()
    eng.set_gzip_fname_len(0)
    bad = 0
    for seed in range(first, first + count):
        bufs, rnd = inputs(seed)
        outs, sts = eng.compress_batch(bufs, 1, oracle.dfGzip)
        for i, (src, out, st) in enumerate(zip(bufs, outs, sts)):
            if st != 0 or out != oracle.compress(src, 1, oracle.dfGzip, fname_len=0):
                bad += 1
                print("COMPRESS MISMATCH seed", seed, "case", i, len(src), st)
        back, sts = eng.uncompress_batch(outs)
        bad += sum(1 for s, b, st in zip(bufs, back, sts) if st != 0 or b != s)
        # the opt-in parallel parse: valid streams (oracle + zlib decode them), no larger than 1.02 x the
        # oracle's in total, the same bytes twice
        eng.set_l1_parse(1)
        try:
            po, ps = eng.compress_batch(bufs, 1, oracle.dfGzip)
            po2, _ = eng.compress_batch(bufs, 1, oracle.dfGzip)
        finally:
            eng.set_l1_parse(-1)
        tot_p = tot_o = 0
        for i, (src, out, st, ref) in enumerate(zip(bufs, po, ps, outs)):
            ok = st == 0 and zlib.decompress(out, 31) == src
            if ok:
                try:
                    ok = oracle.uncompress(out, oracle.dfGzip) == src
                except oracle.ZippyError:
                    ok = False
            if not ok:
                bad += 1
                print("PARALLEL PARSE INVALID seed", seed, "case", i, len(src), st)
            tot_p += len(out)
            tot_o += len(ref)
        if po != po2 or tot_p > 1.02 * tot_o:
            bad += 1
            print("PARALLEL PARSE seed", seed, "deterministic", po == po2, "size", tot_p, "oracle", tot_o)
        pb, pst = eng.uncompress_batch(po)
        bad += sum(1 for s_, b_, st in zip(bufs, pb, pst) if st != 0 or b_ != s_)
        # a few buffers at other levels
        for level in (-1, 9, -2, 0):
            pick = [bufs[i] for i in rnd.sample(range(len(bufs)), 4)]
            o2, s2 = eng.compress_batch(pick, level, oracle.dfZlib)
            for src, out, st in zip(pick, o2, s2):
                if st != 0 or out != oracle.compress(src, level, oracle.dfZlib):
                    bad += 1
                    print("LEVEL MISMATCH seed", seed, level, len(src))
        # foreign encoders: every zlib level and container
        streams, golds = [], []
        for src in bufs[:40]:
            lvl = rnd.randrange(1, 10)
            wb = rnd.choice((31, 15, -15))
            c = zlib.compressobj(lvl, zlib.DEFLATED, wb, rnd.choice((1, 8, 9)), rnd.choice((0, 1, 2, 3)))
            streams.append((c.compress(src) + c.flush(), wb))
            golds.append(src)
        for wb, fmt in ((31, oracle.dfGzip), (15, oracle.dfZlib), (-15, oracle.dfDeflate)):
            sel = [(s, g) for (s, w), g in zip(streams, golds) if w == wb]
            if not sel:
                continue
            got, sts = eng.uncompress_batch([s for s, _ in sel], fmt)
            for (s, g), o, st in zip(sel, got, sts):
                if st != 0 or o != g:
                    bad += 1
                    print("FOREIGN MISMATCH seed", seed, wb, len(g), st)
        # damaged streams: same decision (and bytes) as the oracle
        dmg = []
        for s, wb in streams:
            if wb != 31 or len(s) < 40:
                continue
            m = bytearray(s)
            for _ in range(rnd.randrange(1, 4)):
                m[rnd.randrange(12, len(m) - 8)] ^= 1 << rnd.randrange(8)
            dmg.append(bytes(m))
            if rnd.random() < 0.3:
                dmg.append(s[:rnd.randrange(20, len(s))])
        got, sts = eng.uncompress_batch(dmg, oracle.dfGzip)
        for blob, o, st in zip(dmg, got, sts):
            try:
                want, wst = oracle.uncompress(blob, oracle.dfGzip), 0
            except oracle.ZippyError as e:
                want, wst = None, e.status
            if (st == 0) != (wst == 0) or (st == 0 and o != want):
                bad += 1
                print("DAMAGED MISMATCH seed", seed, len(blob), st, wst)
        # block-parallel form: bytes + index against the oracle, indexed decode, wrong index rejected
        big = b"".join(bufs[:12])
        for level, bb in ((1, 32768), (rnd.choice((-1, 6, 9)), rnd.choice((32768, 65536, 131072))), (0, 32768)):
            want, widx = oracle.compress_blocks(big, level, oracle.dfGzip, bb, fname_len=0)
            got, idx = eng.compress_blocks(big, level, oracle.dfGzip, bb)
            if got != want or idx != widx or eng.uncompress_indexed(got, idx, oracle.dfGzip) != big:
                bad += 1
                print("BLOCKS MISMATCH seed", seed, level, bb)
            if len(idx) > 3:
                k = rnd.randrange(1, len(idx) - 1)
                wrong = idx[:k] + [(idx[k][0] + rnd.choice((-3, 1, 5)), idx[k][1])] + idx[k + 1:]
                try:
                    out = eng.uncompress_indexed(got, wrong, oracle.dfGzip)
                    if out != big:
                        bad += 1
                        print("BLOCKS WRONG-INDEX ACCEPTED WITH WRONG BYTES seed", seed)
                except Exception:
                    pass
        # ZIP: create == oracle, readable by zipfile, extract batch == contents
        import io
        import zipfile
        from oracle import zip_oracle
        entries = [("f%03d/%d.bin" % (i % 5, i), bufs[i]) for i in rnd.sample(range(len(bufs)), 25)]
        arc = eng.create_zip(entries, 0x6000, 0x5a21)
        if arc != zip_oracle.create_archive(entries, 0x6000, 0x5a21) or zipfile.ZipFile(io.BytesIO(arc)).testzip():
            bad += 1
            print("ZIP CREATE MISMATCH seed", seed)
        rd = eng.open_zip(arc)
        outs, sts = rd.extract_batch(list(range(len(entries))))
        if any(sts) or outs != [c for _, c in reversed(entries)]:
            bad += 1
            print("ZIP EXTRACT MISMATCH seed", seed)
        rd.close()
        print("seed", seed, "ok so far" if not bad else "BAD %d" % bad, flush=True)
    print("gpu_fuzz: seeds %d..%d bad %d" % (first, first + count - 1, bad))
    return bad


def mutations(count, seed=20260926):
    """tests/fuzz.nim:16-33 at the reference's scale: `count` single-byte mutations of the reference's
    own .gz fixtures plus the same streams truncated at the mutated byte (2 x count blobs), through
    BOTH inflate paths; accept/reject decision and bytes must equal the oracle's."""
    import parity_cases as pc
    eng = api.engine()
    blobs = pc.mutated_fixtures(count, seed=seed)
    want = []
    for b in blobs:
        try:
            want.append(oracle.uncompress(b, oracle.dfDetect))
        except oracle.ZippyError:
            want.append(None)
    bad = 0
    for mode, name in ((0, "split"), (1, "serial")):
        eng.set_inflate_mode(mode)
        accepted = 0
        for lo in range(0, len(blobs), 4000):
            outs, sts = eng.uncompress_batch(blobs[lo:lo + 4000], oracle.dfDetect)
            for k, (o, st) in enumerate(zip(outs, sts)):
                w = want[lo + k]
                accepted += st == 0
                if (st == 0) != (w is not None) or (st == 0 and o != w):
                    bad += 1
                    print("MUTATION MISMATCH", name, lo + k, len(blobs[lo + k]), st)
        print("gpu_fuzz mutations: inflate=%s blobs %d (flip + truncate of %d mutations) accepted %d rejected %d "
              "mismatches so far %d" % (name, len(blobs), count, accepted, len(blobs) - accepted, bad), flush=True)
    eng.set_inflate_mode(-1)
    print("gpu_fuzz mutations: oracle accepted %d of %d; bad %d" % (sum(w is not None for w in want), len(blobs), bad))
    return bad


def seg_mutations(count, seed=20260927):
    """The segment-wise decode of large streams (zh_inflate_seg.hip) under damage: `count` mutations
    (a flipped bit, or the stream cut short) of multi-block foreign streams, eight to a call so that
    every call takes the segment path; accept/reject decision and bytes must equal the oracle's."""
    import parity_cases as pc
    os.environ["ZH_SEG_BYTES"] = "2048"
    os.environ["ZH_SEG_MIN"] = "8192"
    os.environ["ZH_SEG_SETUP"] = "0"
    eng = api.engine()
    rnd = random.Random(seed)
    cases = pc.segmented_streams(1024)
    bad = accepted = 0
    per_fmt = {}
    for _ in range(count):
        blob, fmt, _plain = cases[rnd.randrange(len(cases))]
        if rnd.random() < 0.25:
            m = blob[:rnd.randrange(len(blob) // 2, len(blob))]
        else:
            b = bytearray(blob)
            for _k in range(rnd.choice((1, 1, 1, 2, 3))):
                b[rnd.randrange(2, len(b))] ^= 1 << rnd.randrange(8)
            m = bytes(b)
        per_fmt.setdefault(fmt, []).append(m)
    for fmt, blobs in per_fmt.items():
        for lo in range(0, len(blobs), 8):
            part = blobs[lo:lo + 8]
            outs, sts = eng.uncompress_batch(part, fmt)
            for blob, o, st in zip(part, outs, sts):
                try:
                    w = oracle.uncompress(blob, fmt)
                except oracle.ZippyError:
                    w = None
                accepted += st == 0
                if (st == 0) != (w is not None) or (st == 0 and o != w):
                    bad += 1
                    print("SEGMENT MUTATION MISMATCH fmt", fmt, len(blob), st)
    print("gpu_fuzz segment mutations: %d blobs accepted %d rejected %d bad %d" % (count, accepted, count - accepted, bad))
    # noise and half-noise, default segment geometry: nothing may be accepted wrongly, nothing may hang
    del os.environ["ZH_SEG_BYTES"], os.environ["ZH_SEG_MIN"], os.environ["ZH_SEG_SETUP"]
    big = zlib.compress(synth.gen_batch("mix", 1, 2 << 20, first_index=3)[0].tobytes(), 6)
    noise = []
    for k in range(48):
        n = rnd.randrange(140000, 600000)
        if k % 3 == 0:
            blob = b"\x78\x9c" + rnd.randbytes(n)
        elif k % 3 == 1:
            cut = rnd.randrange(1000, len(big) - 1000)
            blob = big[:cut] + rnd.randbytes(n)
        else:
            b = bytearray(big)
            for _k in range(rnd.randrange(1, 200)):
                b[rnd.randrange(2, len(b))] = rnd.randrange(256)
            blob = bytes(b)
        noise.append(blob)
    for lo in range(0, len(noise), 8):
        part = noise[lo:lo + 8]
        outs, sts = eng.uncompress_batch(part, oracle.dfZlib)
        for blob, o, st in zip(part, outs, sts):
            try:
                w = oracle.uncompress(blob, oracle.dfZlib)
            except oracle.ZippyError:
                w = None
            if (st == 0) != (w is not None) or (st == 0 and o != w):
                bad += 1
                print("SEGMENT NOISE MISMATCH", len(blob), st)
    print("gpu_fuzz segment noise: %d blobs, bad %d" % (len(noise), bad))
    return bad


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--seg-mutations":
        sys.exit(1 if seg_mutations(int(sys.argv[2]) if len(sys.argv) > 2 else 4000) else 0)
    if len(sys.argv) > 1 and sys.argv[1] == "--mutations":
        sys.exit(1 if mutations(int(sys.argv[2]) if len(sys.argv) > 2 else 10000) else 0)
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 4) else 0)
