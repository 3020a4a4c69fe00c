"""Parity cases shared by tests/test_gpu_parity.py (real MI355X through
zippy_amd/libzippy_hip.so) and tests/test_emu_parity.py (the same kernel sources
under the CPU emulator).  `eng` is a zippy_amd._binding.Engine; the oracle
(oracle/) is the checker.  Mirrors the reference's own tests."""
import hashlib
import os
import random
import struct
import zlib

import numpy as np

import oracle
import synth

WBITS = {oracle.dfDeflate: -15, oracle.dfZlib: 15, oracle.dfGzip: 31}
FORMATS = (oracle.dfDeflate, oracle.dfZlib, oracle.dfGzip)


def check_fixtures(eng, max_len=None):
    """tests/test.nim:41-60, tests/test_known_bad.nim:3 -- bit-exact decode."""
    names, blobs = [], []
    for name, meta in synth.manifest()["fixtures"].items():
        if max_len is None or meta["len"] <= max_len:
            names.append(name)
            blobs.append(synth.fixture(name))
    outs, sts = eng.uncompress_batch(blobs)
    for name, out, st in zip(names, outs, sts):
        meta = synth.manifest()["fixtures"][name]
        assert st == 0, (name, st)
        assert len(out) == meta["len"], name
        assert hashlib.sha256(out).hexdigest() == meta["sha256"], name


def check_compress_identical(eng, inputs, levels, formats=FORMATS):
    """Device output == oracle output, byte for byte; an independent decoder
    (zlib) and the oracle's zippy-equivalent decoder both round-trip it."""
    eng.set_gzip_fname_len(0)
    for level in levels:
        for fmt in formats:
            outs, sts = eng.compress_batch(inputs, level, fmt)
            for src, out, st in zip(inputs, outs, sts):
                assert st == 0, (level, fmt, len(src), st)
                ref = oracle.compress(src, level, fmt, fname_len=0)
                assert out == ref, "level %d fmt %d len %d: device %d B vs oracle %d B" % (
                    level, fmt, len(src), len(out), len(ref))
                assert zlib.decompress(out, WBITS[fmt]) == src
                assert oracle.uncompress(out, fmt) == src


def check_parallel_parse(eng, inputs, formats=(oracle.dfGzip,), margin=1.02):
    """The opt-in parallel BestSpeed parse (zh_set_l1_parse(ctx, 1), csrc/zh_l1p_match.hip) under the
    encoder contract of BASELINE.json's north star: every stream decodes to its input through the
    oracle's zippy-equivalent uncompress AND through zlib, the streams of a batch are together no
    larger than `margin` x the oracle's (= zippy's) at level 1, the result does not depend on the
    run, and the other levels still give the oracle's bytes while the switch is on."""
    eng.set_gzip_fname_len(0)
    eng.set_l1_parse(1)
    try:
        for fmt in formats:
            outs, sts = eng.compress_batch(inputs, 1, fmt)
            again, _ = eng.compress_batch(inputs, 1, fmt)
            dev = ref = 0
            for src, out, st in zip(inputs, outs, sts):
                assert st == 0, (fmt, len(src), st)
                assert zlib.decompress(out, WBITS[fmt]) == src, (fmt, len(src))
                assert oracle.uncompress(out, fmt) == src, (fmt, len(src))
                dev += len(out)
                ref += len(oracle.compress(src, 1, fmt, fname_len=0))
            assert outs == again, "parallel parse: two runs, two results"
            assert dev <= margin * ref, "parallel parse: %d B against the oracle's %d B" % (dev, ref)
            back, sts2 = eng.uncompress_batch(outs, oracle.dfDeflate if fmt == oracle.dfDeflate
                                              else oracle.dfDetect)
            assert all(x == 0 for x in sts2) and back == list(inputs)
        small = [b for b in inputs if len(b) <= 70000][:6]
        for level in (-2, 0, -1):
            outs, sts = eng.compress_batch(small, level, oracle.dfGzip)
            for src, out in zip(small, outs):
                assert out == oracle.compress(src, level, oracle.dfGzip, fname_len=0), level
    finally:
        eng.set_l1_parse(-1)
    return dev, ref


def huffman_histograms():
    """Histograms for the code builders: random and skewed ones, Fibonacci frequencies (a tree deeper than any
    limit), one / two / no symbols used, equal frequencies (ties everywhere), the three alphabets' sizes."""
    rnd = random.Random(31)
    out = []
    fib = [1, 1]
    while len(fib) < 40:
        fib.append(fib[-1] + fib[-2])
    for n, minc, limit in ((286, 257, 15), (30, 2, 15), (19, 19, 7)):
        out.append((np.zeros(n, np.uint32), minc, limit))
        for k in (0, 3, n - 1):
            f = np.zeros(n, np.uint32)
            f[k] = 5
            out.append((f, minc, limit))
        f = np.zeros(n, np.uint32)
        f[1] = f[n - 2] = 7
        out.append((f, minc, limit))
        out.append((np.full(n, 3, np.uint32), minc, limit))
        out.append((np.arange(1, n + 1, dtype=np.uint32), minc, limit))
        m = min(n, 34)
        f = np.zeros(n, np.uint32)
        order = list(range(n))
        rnd.shuffle(order)
        for i in range(m):
            f[order[i]] = fib[i]
        out.append((f, minc, limit))
        for _ in range(12):
            used = rnd.randrange(2, n + 1)
            f = np.zeros(n, np.uint32)
            for i in rnd.sample(range(n), used):
                f[i] = max(1, int(rnd.expovariate(1.0 / rnd.choice((2, 50, 5000, 400000)))))
            out.append((f, minc, limit))
    return out


def check_huffman_builders(eng):
    """zh_debug_huffman: the byte-identical builder gives the oracle's huffmanCodes symbol for symbol (codes and
    lengths); contract mode's gives a complete prefix code within the limit whose payload is the optimum's where
    no length had to be cut (= the oracle's cost there) and within 1 % + 16 bits of the oracle's where some had."""
    for f, minc, limit in huffman_histograms():
        want_codes, want_lens = oracle.huffman_codes(f, minc, limit)
        codes, lens = eng.debug_huffman(f, minc, limit, contract=False)
        assert list(lens) == list(want_lens) and list(codes) == list(want_codes), (len(f), limit, "exact builder")
        codes, lens = eng.debug_huffman(f, minc, limit, contract=True)
        assert len(lens) == len(want_lens), (len(f), limit)
        used = [i for i in range(len(f)) if f[i]]
        if len(used) >= 2:
            assert all(1 <= lens[i] <= limit for i in used) and all(lens[i] == 0 for i in range(len(lens)) if i >= len(f) or not f[i])
            assert sum(2.0 ** -int(lens[i]) for i in used) == 1.0, "not a complete prefix code"
            seen = set()
            for i in used:  # canonical, bit-reversed: no code is another's prefix (read first bit first)
                bits = format(int(codes[i]), "0%db" % lens[i])[::-1]
                assert all(bits[:k] not in seen for k in range(1, len(bits) + 1)), "prefix clash"
                seen.add(bits)
            cost = sum(int(f[i]) * int(lens[i]) for i in used)
            ref = sum(int(f[i]) * int(want_lens[i]) for i in used)
            unlimited = _huffman_cost(f)
            if max(int(lens[i]) for i in used) < limit and max(int(want_lens[i]) for i in used) < limit:
                assert cost == unlimited == ref, (cost, unlimited, ref)
            else:
                assert unlimited <= cost <= ref * 1.01 + 16, (cost, ref, unlimited)
        else:
            assert list(lens) == list(want_lens), "special cases as deflate.nim:34-45"


def _huffman_cost(f):
    import heapq
    h = [int(x) for x in f if x]
    heapq.heapify(h)
    cost = 0
    while len(h) > 1:
        a, b = heapq.heappop(h), heapq.heappop(h)
        cost += a + b
        heapq.heappush(h, a + b)
    return cost


def wide_length_count_input():
    """deflate.nim:136-139 counts the symbols of a code length in a uint8, which
    wraps when 256 or more share one length.  3000 bytes at level -2 (Huffman only, no stored
    fallback, dynamic codes because the block is longer than 2048 bytes): one byte value 2745
    times, the other 255 values once each -- with the end-of-block symbol that is 256 symbols of
    frequency 1 under one subtree: 256 codes of length 9."""
    rnd = random.Random(95)
    body = [0x41] * 2745 + [v for v in range(256) if v != 0x41]
    rnd.shuffle(body)
    return bytes(body)


def check_wide_code_length_counts(eng):
    """The case of SURVEY.md 9.5 through device, oracle and zlib: oracle and kernel count in wide
    integers on purpose (the reference's wrap would emit a stream that does not decode)."""
    import heapq
    src = wide_length_count_input()
    # the premise, worked out independently: Huffman code lengths of the literal/length alphabet
    freq = [0] * 286
    for b in src:
        freq[b] += 1
    freq[256] = 1
    heap = [(f, i, None, None) for i, f in enumerate(freq) if f]
    heapq.heapify(heap)
    n = 286
    while len(heap) > 1:
        a, b = heapq.heappop(heap), heapq.heappop(heap)
        heapq.heappush(heap, (a[0] + b[0], n, a, b))
        n += 1
    depth = {}

    def walk(node, d):
        if node[2] is None:
            depth[node[1]] = d
        else:
            walk(node[2], d + 1)
            walk(node[3], d + 1)
    walk(heap[0], 0)
    per_len = {}
    for d in depth.values():
        per_len[d] = per_len.get(d, 0) + 1
    assert max(per_len.values()) >= 256, per_len
    eng.set_gzip_fname_len(0)
    for fmt in (oracle.dfGzip, oracle.dfDeflate):
        out = eng.compress(src, -2, fmt)
        assert out == oracle.compress(src, -2, fmt, fname_len=0)
        assert zlib.decompress(out, WBITS[fmt]) == src
        assert oracle.uncompress(out, fmt) == src
        assert eng.uncompress(out, fmt) == src


def check_roundtrip(eng, inputs, level, fmt=oracle.dfGzip):
    outs, sts = eng.compress_batch(inputs, level, fmt)
    assert all(s == 0 for s in sts)
    back, sts2 = eng.uncompress_batch(outs, oracle.dfDeflate if fmt == oracle.dfDeflate
                                      else oracle.dfDetect)
    assert all(s == 0 for s in sts2), sts2
    for src, b in zip(inputs, back):
        assert b == src


def check_tokens(eng, src, level):
    """Device match list re-expressed as the reference's u16 token stream
     == oracle token stream, block by block."""
    dev = eng.debug_tokens(src, level)
    parts = [oracle.block_tokens(src, level, o, min(len(src) - o, 4194304))[0]
             for o in range(0, max(len(src), 1), 4194304)]
    want = np.concatenate(parts) if parts else np.zeros(0, np.uint16)
    assert np.array_equal(dev, want), (level, len(dev), len(want))


def check_gzip_random_fname(eng, src):
    """zippy.nim:26-42: FNAME of 0..25 letters chosen per call."""
    eng.set_gzip_fname_len(-1)
    seen = set()
    for _ in range(12):
        out = eng.compress(src, 1, oracle.dfGzip)
        assert out[:4] == b"\x1f\x8b\x08\x08"
        k = out.index(b"\x00", 10) - 10
        assert 0 <= k <= 25 and out[10:10 + k] == bytes(range(97, 97 + k))
        seen.add(k)
        assert zlib.decompress(out, 31) == src
        assert oracle.uncompress(out) == src
    eng.set_gzip_fname_len(0)
    assert len(seen) > 1


def check_checksums(eng, blobs):
    for d in blobs:
        assert eng.crc32(d) == zlib.crc32(d)
        assert eng.adler32(d) == zlib.adler32(d)


def check_errors_match_oracle(eng, blobs, data_format=oracle.dfDetect):
    """tests/fuzz.nim / tests/stress.nim contract: a damaged stream either fails
    (any ZippyError) or decodes; device and oracle must agree on which, and on
    the bytes when it decodes."""
    outs, sts = eng.uncompress_batch(blobs, data_format)
    for blob, out, st in zip(blobs, outs, sts):
        try:
            want = oracle.uncompress(blob, data_format)
            ok = True
        except oracle.ZippyError:
            ok = False
        assert (st == 0) == ok, (len(blob), st, ok)
        if ok:
            assert out == want


def mutated_fixtures(count, seed, max_len=70000):
    """tests/fuzz.nim:16-33: flip one byte, then truncate at that position."""
    files = ["randtest1.gz", "randtest2.gz", "randtest3.gz", "rfctest1.gz", "rfctest2.gz",
             "rfctest3.gz", "zerotest1.gz", "zerotest2.gz"]
    files = [f for f in files if synth.manifest()["fixtures"][f]["len"] <= max_len]
    rng = random.Random(seed)
    blobs = []
    for _ in range(count):
        comp = bytearray(synth.fixture(rng.choice(files)))
        pos = rng.randrange(len(comp))
        comp[pos] = rng.randrange(256)
        blobs.append(bytes(comp))
        blobs.append(bytes(comp[:pos]))
    return blobs


def edge_inputs():
    rnd = random.Random(5)
    text = synth.corpus_file("alice29.txt")
    sizes = [0, 1, 2, 4, 5, 14, 15, 16, 17, 255, 256, 2047, 2048, 2049, 32767, 32768, 32769,
             65535, 65536, 65537]
    out = [text[:n] for n in sizes]
    out += [bytes(range(256)), b"\x00" * 70000, rnd.randbytes(70000), b"ab" * 20000]
    return out


def check_error_statuses(eng):
    gz = bytearray(oracle.compress(b"hello world" * 10, 1, fname_len=0))

    def status_of(blob, fmt=oracle.dfDetect):
        _, sts = eng.uncompress_batch([bytes(blob)], fmt)
        return sts[0]

    bad = bytearray(gz)
    bad[3] |= 4
    assert status_of(bad) == 12  # FEXTRA, gzip.nim:40-41
    bad = bytearray(gz)
    bad[-5] ^= 1
    assert status_of(bad) == 8  # CRC, gzip.nim:80-81
    bad = bytearray(gz)
    bad[-1] ^= 1
    assert status_of(bad) != 0  # ISIZE, gzip.nim:83-88
    assert status_of(b"\x00" * 30) == 3  # detect, zippy.nim:125
    assert status_of(b"\x07" + b"\x00" * 16, oracle.dfDeflate) == 17  # BTYPE 3, inflate.nim:288
    # a bad stream must not poison its neighbours
    good = oracle.compress(b"neighbour" * 100, 1, fname_len=0)
    outs, sts = eng.uncompress_batch([good, bytes(bad), good])
    assert sts[0] == 0 and sts[2] == 0 and sts[1] != 0
    assert outs[0] == b"neighbour" * 100 and outs[2] == outs[0]
    # A gzip member whose ISIZE (the host path's output capacity) is far below what its body
    # produces -- one stored block of 65535 bytes behind ISIZE = 1, and a literal/match body
    # behind ISIZE = 3: the decoder stops at the capacity, and the checksum pass must not read
    # past the slot (it did once: out_len overshot the slot by the failing token).  The oracle
    # decodes with a growing buffer and then fails on the size/CRC check; both must reject.
    import struct
    big = bytes(range(256)) * 256
    stored = b"\x01" + struct.pack("<HH", 65535, 0) + big[:65535]
    liar = b"\x1f\x8b\x08\x00" + b"\x00" * 6 + stored + struct.pack("<II", zlib.crc32(big[:65535]), 1)
    assert status_of(liar) != 0
    body = zlib.compress(b"abcdefgh" * 4000, 6)[2:-4]
    liar2 = b"\x1f\x8b\x08\x00" + b"\x00" * 6 + body + struct.pack("<II", 0, 3)
    assert status_of(liar2) != 0
    for blob in (liar, liar2):
        try:
            oracle.uncompress(blob)
            assert False, "the oracle must reject it too"
        except oracle.ZippyError:
            pass
    outs, sts = eng.uncompress_batch([good, liar, liar2, good])
    assert sts[0] == 0 and sts[3] == 0 and sts[1] != 0 and sts[2] != 0 and outs[3] == outs[0]
    import pytest
    from zippy_amd.common import ZippyError
    for level in (10, -3):
        with pytest.raises(ZippyError):
            eng.compress(b"x", level)
    with pytest.raises(ZippyError):
        eng.compress(b"x", 1, oracle.dfDetect)


def check_blocks(eng, src, levels, block_sizes, formats=(oracle.dfGzip,)):
    """Block-parallel form (BASELINE config 5): bytes and index equal the oracle's at the same block
    size, the stream round-trips through the oracle's plain uncompress() and zlib, and the indexed
    decode returns the input."""
    import zlib
    for level in levels:
        for bb in block_sizes:
            for fmt in formats:
                want, want_idx = oracle.compress_blocks(src, level, fmt, bb, fname_len=0)
                got, idx = eng.compress_blocks(src, level, fmt, bb)
                assert got == want, "level %d block %d fmt %d: bytes differ" % (level, bb, fmt)
                assert idx == want_idx, "level %d block %d fmt %d: index differs" % (level, bb, fmt)
                assert oracle.uncompress(got, fmt) == src
                wbits = {oracle.dfGzip: 31, oracle.dfZlib: 15, oracle.dfDeflate: -15}[fmt]
                assert zlib.decompress(got, wbits) == src
                assert eng.uncompress_indexed(got, idx, fmt) == src
                assert eng.uncompress(got, fmt) == src
                if bb == 4194304:
                    assert got == oracle.compress(src, level, fmt, fname_len=0)


def check_blocks_bad_index(eng, src):
    """An index that does not describe the stream fails the call instead of returning wrong bytes."""
    import pytest
    from zippy_amd.common import ZippyError
    got, idx = eng.compress_blocks(src, 1, oracle.dfGzip, 32768)
    assert len(idx) >= 4
    for bad in (
        [idx[0], (idx[1][0] + 1, idx[1][1])] + idx[2:],          # block 1 starts one bit late
        [idx[0], (idx[1][0], idx[1][1] - 7)] + idx[2:],          # block 0 promises 7 bytes less
        idx[:-1] + [(idx[-1][0], idx[-1][1] + 1)],               # total one byte too long
        [idx[0], idx[2]] + idx[2:],                              # repeated entry: block skipped
    ):
        with pytest.raises(ZippyError):
            eng.uncompress_indexed(got, bad, oracle.dfGzip)
    damaged = bytearray(got)
    damaged[len(damaged) // 2] ^= 0x10
    with pytest.raises(ZippyError):
        eng.uncompress_indexed(bytes(damaged), idx, oracle.dfGzip)


# ---- ZIP archives (reference tests: tests/test_ziparchives_read.nim,
# tests/test_ziparchives_write.nim) ----
ZIP_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ziparchives")


def zip_fixture(name):
    with open(os.path.join(ZIP_DIR, name), "rb") as fh:
        return fh.read()


def check_zip_extract(eng, image):
    """Every record of the archive, extracted in one batch, equals the oracle's extractFile."""
    from oracle import zip_oracle
    want = zip_oracle.open_archive(image)
    reader = eng.open_zip(image)
    assert [e["path"].encode("utf-8", "surrogateescape") for e in reader.entries] == list(want.records)
    for e in reader.entries:
        w = want.records[e["path"].encode("utf-8", "surrogateescape")]
        for key in ("is_directory", "header_offset", "crc32", "compressed_size", "uncompressed_size", "unix_mode"):
            assert e[key] == w[key], (e["path"], key)
    files = [i for i, e in enumerate(reader.entries) if not e["is_directory"]]
    outs, sts = reader.extract_batch(files)
    assert all(s == 0 for s in sts)
    for i, out in zip(files, outs):
        assert out == zip_oracle.extract_file(want, reader.entries[i]["path"]), reader.entries[i]["path"]
    dirs = [i for i, e in enumerate(reader.entries) if e["is_directory"]]
    if dirs:
        _, sts = reader.extract_batch(dirs[:3])
        assert all(s == 26 for s in sts)  # "No file record found", ziparchives.nim:89-90
    assert reader.walk_files() == [p.decode("utf-8", "surrogateescape") for p, r in want.records.items()
                                   if not r["is_directory"]]
    reader.close()
    return len(files)


def check_zip_create(eng, entries, dos_time=0x6000, dos_date=0x5a21):
    """createZipArchive: same bytes as the oracle, readable by Python's zipfile, and back."""
    import io
    import zipfile
    from oracle import zip_oracle
    got = eng.create_zip(entries, dos_time, dos_date)
    assert got == zip_oracle.create_archive(entries, dos_time, dos_date)
    zf = zipfile.ZipFile(io.BytesIO(got))
    assert zf.testzip() is None
    pairs = list(entries.items()) if hasattr(entries, "items") else list(entries)
    assert zf.namelist() == [p for p, _ in reversed(pairs)]
    for path, contents in pairs:
        assert zf.read(path) == bytes(contents)
    reader = eng.open_zip(got)
    outs, sts = reader.extract_batch(list(range(len(pairs))))
    assert all(s == 0 for s in sts)
    assert outs == [bytes(c) for _, c in reversed(pairs)]
    reader.close()
    return got


def check_zip_errors(eng):
    import pytest
    from zippy_amd.common import ZippyError
    from oracle import zip_oracle
    either = (ZippyError, oracle.ZippyError)
    good = eng.create_zip([("a.txt", b"hello " * 50), ("dir/b.bin", bytes(range(256)) * 8), ("empty", b"")])
    # damaged payload -> that record fails its CRC (or decode), neighbours survive
    bad = bytearray(good)
    reader = eng.open_zip(good)
    target = reader.entries[1]  # "dir/b.bin"... entries are listed last to first
    reader.close()
    bad[target["header_offset"] + 30 + len(target["path"]) + 20 + 5] ^= 0x40
    reader = eng.open_zip(bytes(bad))
    outs, sts = reader.extract_batch([0, 1, 2])
    assert sts[1] != 0 and sts[0] == 0 and sts[2] == 0
    with pytest.raises(either):
        zip_oracle.extract_file(zip_oracle.open_archive(bytes(bad)), target["path"])
    reader.close()
    for blob in (b"", b"PK\x05\x06", good[:-30], good[:len(good) // 2]):
        with pytest.raises(ZippyError):
            eng.open_zip(blob)
        with pytest.raises(either):
            zip_oracle.open_archive(blob)
    # zip64 fields near INT64_MAX / above it (an overflow-checked reference raises; here the
    # bounds checks must not wrap): a zip64 locator pointing its EOCD64 at 0x7fffffffffffffe0,
    # and a real EOCD64 whose directory offset / size / record count are absurd
    import struct
    eocd = b"PK\x05\x06" + b"\x00" * 18
    for off in (0x7fffffffffffffe0, 0xffffffffffffffff, 1 << 63, 1 << 40):
        blob = b"PK\x06\x07" + struct.pack("<IQI", 0, off, 1) + eocd
        with pytest.raises(ZippyError):
            eng.open_zip(blob)
    for fields in ((1, 1, 0x7ffffffffffffff0, 0), (1, 1, 0, 0x7ffffffffffffff0), (1 << 62, 1 << 62, 10, 0),
                   (0xffffffffffffffff, 0xffffffffffffffff, 46, 0)):
        e64 = b"PK\x06\x06" + struct.pack("<QHHIIQQQQ", 44, 45, 45, 0, 0, *fields)
        blob = e64 + b"PK\x06\x07" + struct.pack("<IQI", 0, 0, 1) + eocd
        with pytest.raises(ZippyError):
            eng.open_zip(blob)
    for entries in ([("", b"x")], [("/abs", b"x")], [("n" * 70000, b"x")]):
        with pytest.raises(ZippyError):
            eng.create_zip(entries)
        with pytest.raises(either):
            zip_oracle.create_archive(entries)
    with pytest.raises(ZippyError):
        eng.open_zip(good).extract_file("nope.txt")


# ---- tarballs (reference test: tests/test_tarballs_read.nim) ----
def tar_fixture():
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tarballs",
                           "libressl-3.4.2.tar.gz"), "rb") as fh:
        return fh.read()


def check_tarball(eng, image):
    from oracle import tar_oracle
    data, want = tar_oracle.open_tarball(image)
    reader = eng.open_tar(image)
    assert reader.data == data
    assert reader.entries == want
    reader.close()
    return len(want)


def make_tar_gz(files, long_name=None):
    """A small .tar.gz with Python's tarfile (GNU format: a long name becomes an 'L' block)."""
    import io
    import tarfile
    buf = io.BytesIO()
    with tarfile.open(fileobj=buf, mode="w:gz", format=tarfile.GNU_FORMAT) as tf:
        for name, contents in files:
            if contents is None:
                info = tarfile.TarInfo(name)
                info.type = tarfile.DIRTYPE
                info.mode = 0o755
                tf.addfile(info)
            else:
                info = tarfile.TarInfo(name)
                info.size = len(contents)
                info.mode = 0o644
                info.mtime = 1234567890
                tf.addfile(info, io.BytesIO(contents))
        if long_name:
            info = tarfile.TarInfo(long_name)
            info.size = 3
            tf.addfile(info, io.BytesIO(b"abc"))
        link = tarfile.TarInfo("link")
        link.type = tarfile.SYMTYPE
        link.linkname = "a.txt"
        tf.addfile(link)
    return buf.getvalue()


def check_tar_errors(eng):
    import gzip
    import pytest
    from zippy_amd.common import ZippyError
    from oracle import tar_oracle
    either = (ZippyError, oracle.ZippyError)
    good = gzip.decompress(make_tar_gz([("a.txt", b"hello")]))
    for bad in (good[:515],                                        # cut inside the first file
                good[:156] + b"Z" + good[157:],                    # vendor type: skipped, not an error
                good[:156] + b"7" + good[157:],                    # unsupported type
                b"../evil".ljust(100, b"\0") + good[100:],         # unsafe path
                good[:124] + b"0000000009\0" + good[135:]):        # "9" is not octal
        expect_ok = bad[156:157] == b"Z"
        for fn in (lambda b: eng.open_tar(b), tar_oracle.open_tarball):
            if expect_ok:
                fn(bad)
            else:
                with pytest.raises(either):
                    fn(bad)
    with pytest.raises(either):
        eng.open_tar(b"\x1f\x8b" + b"\0" * 30)


def check_ragged_staging(eng, scale):
    """Host-buffer calls whose buffers straddle the staging chunks (upload) and whose results
    straddle them again (download): ragged sizes, empty buffers in between, one buffer of
    several chunks.  `scale` stretches the sizes (1 for the emulator's 128 KiB chunks)."""
    import synth
    sizes = [0, 70001, 1, 262144 + 13, 0, 0, 4095, 400000 + 7, 65536, 131072, 3, 0, 99999, 0]
    pool = synth.gen_batch("mix", 8, 1 << 20).tobytes()
    bufs, at = [], 0
    for k, sz in enumerate(sizes):
        sz *= scale
        rep = -(-(sz + 1) // len(pool))
        bufs.append((pool * rep)[at % 4096:at % 4096 + sz])
        at += 977
    for level, fmt in ((1, oracle.dfGzip), (0, oracle.dfZlib), (-2, oracle.dfDeflate)):
        # one plan for the whole batch, then the same batch as pipelined groups: identical results
        try:
            eng.set_host_pipeline(1 << 60, 0)
            outs, sts = eng.compress_batch(bufs, level, fmt)
            eng.set_host_pipeline(1, 150000 * scale)
            outs2, sts2 = eng.compress_batch(bufs, level, fmt)
        finally:
            eng.set_host_pipeline(0, 0)
        assert all(s == 0 for s in sts), sts
        assert sts2 == sts and outs2 == outs, level
        for i in (1, 3, 7, 12) if scale > 1 else range(len(bufs)):
            assert outs[i] == oracle.compress(bufs[i], level, fmt, fname_len=0), (level, i)
        back, sts = eng.uncompress_batch(outs, fmt)
        assert all(s == 0 for s in sts), sts
        assert back == bufs, level
        if fmt == oracle.dfGzip:
            # the same batch as pipelined groups (uncompress_batch_pipelined: gzip members carry their
            # size): identical results; then one member whose ISIZE promises too little -- the group it
            # is in outgrows its slot and the whole batch is sent down the plain path, which sizes and
            # retries: same bytes, that member's status from the ISIZE check like the plain path's
            try:
                eng.set_host_pipeline(1, 150000 * scale)
                back2, sts2 = eng.uncompress_batch(outs, fmt)
                liar = list(outs)
                k = 7
                small = (len(bufs[k]) // 2).to_bytes(4, "little")
                liar[k] = liar[k][:-4] + small
                back3, sts3 = eng.uncompress_batch(liar, fmt)
                bad7 = list(outs)
                bad7[k] = bad7[k][:len(bad7[k]) // 2] + bytes([bad7[k][len(bad7[k]) // 2] ^ 0x55]) + bad7[k][len(bad7[k]) // 2 + 1:]
                back4, sts4 = eng.uncompress_batch(bad7, fmt)
            finally:
                eng.set_host_pipeline(1 << 60, 0)
            assert sts2 == sts and back2 == back, "pipelined uncompress differs"
            plain3, psts3 = eng.uncompress_batch(liar, fmt)
            plain4, psts4 = eng.uncompress_batch(bad7, fmt)
            eng.set_host_pipeline(0, 0)
            assert sts3 == psts3 and back3 == plain3 and sts3[k] != 0, (sts3, psts3)
            assert sts4 == psts4 and back4 == plain4 and sts4[k] != 0
            assert [b for i, b in enumerate(back3) if i != k] == [b for i, b in enumerate(bufs) if i != k]
        if fmt == oracle.dfGzip:  # a damaged member in the middle only fails its own slot
            bad = list(outs)
            bad[7] = bad[7][:len(bad[7]) // 2] + bytes([bad[7][len(bad[7]) // 2] ^ 0x55]) + bad[7][len(bad[7]) // 2 + 1:]
            back, sts = eng.uncompress_batch(bad, fmt)
            assert sts[7] != 0 and back[7] is None
            assert [b for i, b in enumerate(back) if i != 7] == [b for i, b in enumerate(bufs) if i != 7]


def check_batch_into(eng):
    """zh_compress_batch_into / zh_uncompress_batch_into: the results of the ordinary calls, in
    buffers of the caller's; one that is too small only fails its own slot -- with ZH_ERR_DST_TOO_SMALL,
    which is what a binding grows and retries on (include/zippy_hip.h) -- and learns its size.  Once as
    one plan, once as pipelined groups."""
    import synth
    TOO_SMALL = 21  # ZH_ERR_DST_TOO_SMALL
    eng.set_gzip_fname_len(0)
    bufs = [b.tobytes() for b in synth.gen_batch("mix", 5, 70001)] + [b"", b"x" * 300, synth.corpus_file("html")]
    want, sts = eng.compress_batch(bufs, 1, oracle.dfGzip)
    for pipe in ((1 << 60, 0), (1, 150000)):
        try:
            eng.set_host_pipeline(*pipe)
            outs = [bytearray(eng.compress_bound(len(b))) for b in bufs]
            outs[2] = bytearray(100)  # too small
            lens, sts2, filled = eng.compress_batch_into(bufs, outs, 1, oracle.dfGzip)
            for i, b in enumerate(bufs):
                assert lens[i] == len(want[i]), (pipe, i)
                if i == 2:
                    assert sts2[i] == TOO_SMALL and not filled[i], (pipe, sts2[i])
                else:
                    assert sts2[i] == 0 and filled[i] and bytes(outs[i][:lens[i]]) == want[i], (pipe, i)
            for fmt in (oracle.dfGzip, oracle.dfZlib, oracle.dfDeflate):
                blobs, _ = eng.compress_batch(bufs, 1, fmt)
                back = [bytearray(len(b)) for b in bufs]
                back[4] = bytearray(len(bufs[4]) - 1)  # one byte short
                back[7] = bytearray(5)                 # far too short (a sized stream takes the sizing pass)
                blobs = list(blobs)
                blobs[1] = blobs[1][:len(blobs[1]) // 2]  # and a damaged stream
                lens, sts3, filled = eng.uncompress_batch_into(blobs, back, fmt)
                for i, b in enumerate(bufs):
                    if i == 1:
                        assert sts3[i] not in (0, TOO_SMALL), (pipe, fmt, sts3[i])
                    elif i in (4, 7):
                        assert sts3[i] == TOO_SMALL and lens[i] == len(b) and not filled[i], (pipe, fmt, i, sts3[i], lens[i])
                    else:
                        assert sts3[i] == 0 and filled[i] and lens[i] == len(b) and bytes(back[i]) == b, (pipe, fmt, i)
        finally:
            eng.set_host_pipeline(0, 0)


def check_unsized_streams(eng):
    """zlib / raw deflate streams carry no size: the host call guesses 4x, and streams that
    outgrow the guess are sized and decoded again -- next to streams that fit, damaged ones and
    (for dfDetect) gzip members, in one call."""
    import zlib
    import synth
    text = synth.corpus_file("alice29.txt")[:90000]
    rnd = np.random.default_rng(11).integers(0, 256, 140001, dtype=np.uint8).tobytes()
    plain = [b"", b"a", text, b"\x00" * 300000, bytes(range(256)) * 40, b"ab" * 70000, text[:777],
             b"\xff" * 1000000,
             # (round 6: the sizing pass runs on the tokens kernel's count-only form) markup that compresses 6-7 x in many
             # dynamic blocks (system zlib's); three stored blocks and then a long run: stored chains and codes in one sized stream
             rnd + b"\x07" * 2000003,
             synth.corpus_file("html_x_4") + synth.corpus_file("html_x_4")[:190001]]
    for fmt, wb in ((oracle.dfZlib, 15), (oracle.dfDeflate, -15)):
        blobs = []
        for k, b in enumerate(plain):
            if k % 2:
                c = zlib.compressobj(6, zlib.DEFLATED, wb)
                blobs.append(c.compress(b) + c.flush())
            else:
                blobs.append(oracle.compress(b, 1, fmt))
        hurt = bytearray(blobs[3])
        hurt[len(hurt) // 2] ^= 0x40
        blobs.append(bytes(hurt))           # damaged, highly compressible
        blobs.append(blobs[2][:len(blobs[2]) // 2])  # truncated
        outs, sts = eng.uncompress_batch(blobs, fmt)
        for i, blob in enumerate(blobs):
            try:
                want = oracle.uncompress(blob, fmt)
            except oracle.ZippyError:
                want = None
            assert (outs[i] if sts[i] == 0 else None) == want, (fmt, i, sts[i])
            if i < len(plain):
                assert want == plain[i]
    mixed = [oracle.compress(plain[3], 1, oracle.dfZlib), oracle.compress(text, 1, oracle.dfGzip, fname_len=3),
             zlib.compress(plain[7], 9), oracle.compress(plain[5], -1, oracle.dfGzip, fname_len=0)]
    outs, sts = eng.uncompress_batch(mixed)
    assert sts == [0, 0, 0, 0] and outs == [plain[3], text, plain[7], plain[5]]


def check_plan_slots_with_gaps(eng, upload, download, alloc, levels=(1, -2, 0, 6), fills=(0xAB, 0xFF)):
    """Device plan API (include/zippy_hip.h): output slots at odd offsets in memory the caller has filled with a
    pattern -- nothing is cleared beforehand (round 6: the layout kernels zero exactly the words that are OR-ed into,
    csrc/zh_huffman.hip), so a stream's bytes must be the oracle's whatever was there, and NOTHING else may change:
    neither the rest of a slot nor the caller's bytes between slots.  Compressed, fixed, stored (two chunks) and
    empty buffers, the three containers; a misaligned d_dst is refused.  upload(bytes) -> (ptr, keep),
    alloc(n, fill) -> (ptr, keep), download(keep) -> bytes."""
    import pytest
    from zippy_amd.common import ZippyError
    rnd = np.random.default_rng(7).integers(0, 256, 70001, dtype=np.uint8).tobytes()
    srcs = [synth.corpus_file("alice29.txt")[:50000], b"", synth.corpus_file("html")[:33000], b"x" * 70001, rnd, b"ab",
            synth.corpus_file("alice29.txt")[:700]]
    src_off, pos = [], 0
    for s in srcs:
        src_off.append(pos)
        pos += len(s)
    d_src, keep_src = upload(b"".join(srcs) + b"\0" * 16)
    caps = [len(s) + len(s) // 8 + 2048 for s in srcs]
    dst_off, pos = [], 7
    for i, c in enumerate(caps):
        dst_off.append(pos)
        pos += c + (101, 1000, 3, 513, 0, 1, 2)[i]  # (slots 4 / 5: back to back at odd addresses)
    total = pos + 64
    plan = None
    for k, level in enumerate(levels):
        fmt = FORMATS[k % len(FORMATS)]
        fill = fills[k % len(fills)]
        d_dst, keep_dst = alloc(total, fill)
        plan = eng.plan_compress(src_off, [len(s) for s in srcs], dst_off, caps, level, fmt)
        plan.run(d_src, d_dst)
        lens, sts = plan.results()
        assert all(st == 0 for st in sts)
        got = download(keep_dst)
        mine = bytearray(total)
        for i, (s, o, ln) in enumerate(zip(srcs, dst_off, lens)):
            assert got[o:o + ln] == oracle.compress(s, level, fmt, fname_len=0), (level, fmt, i)
            mine[o:o + ln] = b"\1" * ln
        changed = [i for i in range(total) if not mine[i] and got[i] != fill]
        assert not changed, "level %d: bytes outside the streams changed, first at %d" % (level, changed[0])
    with pytest.raises(ZippyError):
        plan.run(d_src, d_dst + 1)


def check_plan_reruns(eng, upload, download, alloc, n, size):
    """A compress plan run three times (zh_l1_match.hip: from the second run on the BestSpeed matcher's waves take the
    cheapest fragments -- by what they cost the run before -- last; more fragments than waves here, so the order is a
    real one): every run's streams are the oracle's, byte for byte."""
    bufs = [b.tobytes() for b in synth.gen_batch("mix", n, size)]
    d_src, keep_src = upload(b"".join(bufs) + b"\0" * 16)
    cap = size + size // 8 + 2048
    slot = (cap + 255) & ~255
    d_dst, keep_dst = alloc(n * slot, 0)
    plan = eng.plan_compress([i * size for i in range(n)], [size] * n, [i * slot for i in range(n)], [cap] * n, 1, oracle.dfGzip)
    want = [oracle.compress(b, 1, oracle.dfGzip, fname_len=0) for b in bufs]
    for run in range(3):
        plan.run(d_src, d_dst)
        lens, sts = plan.results()
        assert all(st == 0 for st in sts), run
        got = download(keep_dst)
        for i, w in enumerate(want):
            assert lens[i] == len(w) and got[i * slot:i * slot + lens[i]] == w, (run, i)


def check_plan_pack(eng, upload, download, alloc):
    """zh_plan_pack / zh_plan_unpack (include/zippy_hip.h; zippy.nim:11-18: whole buffers are all that travels): a
    compress plan's results back to back with n + 1 device offsets == the oracle's streams concatenated (a failed
    buffer counts 0 bytes); the packed streams scattered into an uncompress plan's slots (another layout, odd offsets)
    decode to the inputs; a packed buffer that is too small is not overrun; a stream longer than its slot is cut to
    the slot and fails by itself."""
    import struct
    import pytest
    from zippy_amd.common import ZippyError
    srcs = [synth.corpus_file("alice29.txt")[:70000], b"", synth.corpus_file("html")[:33001], b"x" * 70001,
            synth.corpus_file("kppkn.gtb")[:150003], b"q"]
    n = len(srcs)
    src_off, pos = [], 0
    for s in srcs:
        src_off.append(pos)
        pos += len(s)
    d_src, keep_src = upload(b"".join(srcs) + b"\0" * 16)
    caps = [len(s) + len(s) // 8 + 2048 for s in srcs]
    caps[3] = 40  # too small: this buffer fails (ZH_ERR_DST_TOO_SMALL) and packs as 0 bytes
    dst_off, pos = [], 3
    for i, c in enumerate(caps):
        dst_off.append(pos)
        pos += c + (101, 1000, 3, 513, 77, 9)[i]
    d_dst, keep_dst = alloc(pos + 64, 0xAB)
    cplan = eng.plan_compress(src_off, [len(s) for s in srcs], dst_off, caps, 1, oracle.dfGzip)
    cplan.run(d_src, d_dst)
    want = [oracle.compress(s, 1, oracle.dfGzip, fname_len=0) for s in srcs]
    want[3] = b""
    total = sum(len(w) for w in want)
    d_pack, keep_pack = alloc(total + 64, 0xCD)
    d_offs, keep_offs = alloc(8 * (n + 1), 0xEE)
    cplan.pack(d_dst, d_pack, total + 64, d_offs)   # (no results() in between: the lengths are the device's)
    lens, sts = cplan.results()
    assert [st == 0 for st in sts] == [True, True, True, False, True, True]
    offs = list(struct.unpack("<%dQ" % (n + 1), download(keep_offs)))
    assert offs == [sum(len(w) for w in want[:i]) for i in range(n + 1)]
    packed = download(keep_pack)
    assert packed[:total] == b"".join(want)
    assert packed[total:] == b"\xcd" * 64, "bytes behind the packed streams changed"
    # too small a packed buffer: the offsets say so, nothing behind the capacity is written
    small = total - 1000
    d_pack2, keep_pack2 = alloc(total + 64, 0xCD)
    cplan.pack(d_dst, d_pack2, small, d_offs)
    cplan.results()
    assert struct.unpack("<%dQ" % (n + 1), download(keep_offs))[n] == total > small
    p2 = download(keep_pack2)
    assert p2[:small] == b"".join(want)[:small] and p2[small:] == b"\xcd" * (total + 64 - small)
    # ... and back: into an uncompress plan's source slots (sizes and places of their own), then decoded
    ucaps = [len(w) + (5, 0, 300, 64, 1, 17)[i] for i, w in enumerate(want)]
    ucaps[4] = len(want[4]) - 10  # a slot smaller than its stream: cut, and the stream fails alone
    usrc_off, pos = [], 5
    for c in ucaps:
        usrc_off.append(pos)
        pos += c + 13
    d_usrc, keep_usrc = alloc(pos + 64, 0x11)
    out_caps = [max(len(s), 1) for s in srcs]
    out_off, pos = [], 0
    for c in out_caps:
        out_off.append(pos)
        pos += c + 7
    d_out, keep_out = alloc(pos + 64, 0x22)
    uplan = eng.plan_uncompress(usrc_off, ucaps, out_off, out_caps, oracle.dfGzip)
    uplan.unpack(d_pack, d_offs, d_usrc)
    uplan.run(d_usrc, d_out)
    ulens, usts = uplan.results()
    got = download(keep_out)
    for i, s in enumerate(srcs):
        if i in (3, 4):  # the empty stream of the failed buffer; the cut stream
            assert usts[i] != 0
        else:
            assert usts[i] == 0 and got[out_off[i]:out_off[i] + ulens[i]] == s, i
    staged = download(keep_usrc)
    for i, w in enumerate(want):
        k = min(len(w), ucaps[i])
        assert staged[usrc_off[i]:usrc_off[i] + k] == w[:k]
        assert staged[usrc_off[i] + k:usrc_off[i] + k + 13] == b"\x11" * 13, "bytes outside a source slot changed"
    with pytest.raises(ZippyError):
        cplan.unpack(d_pack, d_offs, d_usrc)  # (a compress plan has no streams to be handed)


def split_inflate_edge_streams():
    """Streams that exercise the corners of the parallel token decode (csrc/zh_inflate_split.hip):
    periodic data (wrong starts never fall in step: the all-starts pass), hundreds of tiny blocks
    with empty stored blocks between them (Z_FULL_FLUSH), fixed-Huffman blocks, long distance codes,
    block ends near subchunk / superchunk borders, stored + compressed blocks mixed."""
    rnd = random.Random(77)
    out = []

    def z(data, level=6, wbits=31, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=0):
        c = zlib.compressobj(level, zlib.DEFLATED, wbits, 9, strategy)
        if not flush_every:
            return c.compress(data) + c.flush()
        parts = []
        for o in range(0, len(data), flush_every):
            parts.append(c.compress(data[o:o + flush_every]))
            parts.append(c.flush(zlib.Z_FULL_FLUSH if (o // flush_every) % 3 else zlib.Z_SYNC_FLUSH))
        parts.append(c.flush())
        return b"".join(parts)

    text = synth.corpus_file("alice29.txt")
    geo = synth.corpus_file("geo.protodata")
    for per in (1, 2, 3, 7, 8, 9, 31, 32, 33, 255, 256, 257, 258, 259, 1000):
        pat = rnd.randbytes(per)
        data = (pat * (300000 // per + 1))[:300000]
        out.append((z(data, rnd.choice((1, 6, 9))), data))
    out.append((z(b"\0" * 3000000, 6), b"\0" * 3000000))
    out.append((z(text, 6, flush_every=100), text))                  # ~1500 blocks
    out.append((z(geo, 9, flush_every=1777), geo))
    out.append((z(text[:70000], 6, strategy=zlib.Z_FIXED), text[:70000]))
    out.append((z(geo, 1, strategy=zlib.Z_HUFFMAN_ONLY), geo))
    out.append((z(text, 1, strategy=zlib.Z_RLE), text))
    big = text + rnd.randbytes(200000) + geo + bytes(100000) + text[::-1] + rnd.randbytes(70000)
    for lvl in (1, 6, 9):
        out.append((z(big, lvl), big))
    out.append((z(big, 6, wbits=15), big))
    out.append((z(big, 6, wbits=-15), big))
    # block ends walked across every bit position of a subchunk border: prefix lengths of one text
    for n in range(16370, 16400):
        out.append((z(text[:n], 6), text[:n]))
    return out


def damaged_header_streams(step=1):
    """Raw deflate streams (no checksum behind them: only the byte comparison catches a wrong byte) whose
    dynamic header is hit bit by bit: every bit of the first 90 bytes flipped, one at a time -- HLIT / HDIST /
    HCLEN, the code-length code's lengths, the run-length coded lengths with their repeat counts.  One input
    has a skewed alphabet (code lengths up to 15: second-level tables), one is text, one uses few symbols
    (long zero runs in the header).  zh_inflate_split.hip reads clean headers with a whole wave and builds the
    tables with the workgroup; anything else goes to the serial reader: both have to agree with the oracle."""
    rnd = random.Random(4242)
    skew = bytes(min(255, int(rnd.expovariate(0.035))) for _ in range(6000))
    few = bytes(rnd.choice(b"ab\x00\xff") for _ in range(2500)) + b"abab" * 50
    text = synth.corpus_file("alice29.txt")[:4000]
    out = []
    for data, strategy in ((skew, zlib.Z_HUFFMAN_ONLY), (text, zlib.Z_DEFAULT_STRATEGY), (few, zlib.Z_DEFAULT_STRATEGY)):
        c = zlib.compressobj(9, zlib.DEFLATED, -15, 9, strategy)
        blob = c.compress(data) + c.flush()
        assert (blob[0] >> 1) & 3 == 2, "expected a dynamic block"
        out.append(blob)
        for bit in range(0, min(len(blob), 90) * 8, step):
            m = bytearray(blob)
            m[bit >> 3] ^= 1 << (bit & 7)
            out.append(bytes(m))
    return out


def check_damaged_headers(eng, step=1):
    check_errors_match_oracle(eng, damaged_header_streams(step), oracle.dfDeflate)


def check_split_inflate_edges(eng):
    """Both inflate paths return the input for every stream above, and the same statuses when the
    output slot is one byte short (device plan API is exercised by check_plan_slots_with_gaps)."""
    cases = split_inflate_edge_streams()
    gz = [c for c in cases if c[0][:2] == b"\x1f\x8b"]
    outs, sts = eng.uncompress_batch([c[0] for c in gz], oracle.dfGzip)
    for (blob, want), got, st in zip(gz, outs, sts):
        assert st == 0 and got == want, (len(blob), len(want), st)
        assert oracle.uncompress(blob, oracle.dfGzip) == want
    zl = [c for c in cases if c[0][:2] != b"\x1f\x8b"]
    for blob, want in zl:
        fmt = oracle.dfZlib if (blob[0] & 0x0f) == 8 and (blob[0] * 256 + blob[1]) % 31 == 0 else oracle.dfDeflate
        got, st = eng.uncompress_batch([blob], fmt)
        assert st == [0] and got[0] == want, (len(blob), fmt, st)


def stored_chain_streams(nblocks=70, small=False):
    """Streams of stored blocks (inflate.nim:252-266), the kind incompressible data makes -- one of 65 535 bytes after
    the other (deflate.nim:186-199) --, for the chain reader of the tokens kernel (64 headers at once) and the
    writer's grouped copy: whole chains of more than 64 blocks, a short last block, an empty one, chains broken by a
    compressed block, by a short block in the middle, and damaged ones (a length that does not match its complement
    in the middle of a chain, a chain that runs past the input).  small: the subset the CPU emulator (a third of a
    megabyte a second on these) runs; the GPU tests run all of them.  -> [(raw deflate, plain or None)]"""
    rnd = random.Random(4711)
    noise = rnd.randbytes(nblocks * 65535 + 12345)

    def stored(data, final, size=65535):
        out = []
        chunks = [data[o:o + size] for o in range(0, len(data), size)] or [b""]
        for k, c in enumerate(chunks):
            out.append(bytes([1 if final and k == len(chunks) - 1 else 0]) + struct.pack("<HH", len(c), len(c) ^ 0xffff) + c)
        return b"".join(out)
    text = synth.corpus_file("alice29.txt")[:50000]
    dyn = zlib.compressobj(6, zlib.DEFLATED, -15)
    dyn_block = dyn.compress(text) + dyn.flush(zlib.Z_FULL_FLUSH)  # (ends byte-aligned with an empty stored block, not final)
    out = []
    out.append((stored(noise, True), noise))                                        # the oracle's own shape at level 0
    if not small:
        out.append((oracle.compress(noise, 0, oracle.dfDeflate), noise))
    out.append((oracle.compress(noise[:200000], 1, oracle.dfDeflate), noise[:200000]))  # incompressible at level 1: stored
    if not small:
        out.append((stored(noise[:65535 * 64], True), noise[:65535 * 64]))          # exactly 64 full blocks, the last final
    out.append((stored(noise[:65535 * 65], False) + stored(b"", True), noise[:65535 * 65]))  # an empty final block behind the chain
    out.append((stored(noise[:65535 * 3], False) + dyn_block + stored(noise[:65535 * (nblocks - 1) + 5], True),
                noise[:65535 * 3] + text + noise[:65535 * (nblocks - 1) + 5]))
    if not small:
        out.append((stored(noise[:65535 * 5], False) + stored(noise[:1000], False, 1000) + stored(noise[:65535 * 65], True),
                    noise[:65535 * 5] + noise[:1000] + noise[:65535 * 65]))
    out.append((stored(noise[:300000], True, 30000), noise[:300000]))               # no full block at all
    good = stored(noise[:65535 * (20 if small else 40)], True)
    bad = bytearray(good)
    bad[17 * 65540 + 3] ^= 0x40                                                     # block 17: NLEN no longer the complement
    out.append((bytes(bad), None))
    out.append((good[:-30000], None))                                               # the last block runs past the input
    bad = bytearray(good)
    bad[9 * 65540] |= 0x06                                                          # block 9: BTYPE 3
    out.append((bytes(bad), None))
    return out


def check_stored_chains(eng, nblocks=70, small=False):
    """The streams above through the device decoder: bytes where they are sound (and through zlib, the referee), the
    oracle's accept / reject decision where they are not; and with an output slot that is too small the status a
    caller grows its buffer on, not another."""
    cases = stored_chain_streams(nblocks, small)
    outs, sts = eng.uncompress_batch([c[0] for c in cases], oracle.dfDeflate)
    for (blob, want), got, st in zip(cases, outs, sts):
        try:
            ref = oracle.uncompress(blob, oracle.dfDeflate)
        except oracle.ZippyError:
            ref = None
        assert ref == want, "the oracle disagrees with the test's expectation"
        assert (st == 0) == (want is not None), (len(blob), st)
        if want is not None:
            assert got == want and zlib.decompress(blob, -15) == want
    one = [cases[1 if small else 0][0]]  # (a batch of one: the wide kernels)
    got, st = eng.uncompress_batch(one, oracle.dfDeflate)
    assert st == [0] and got[0] == cases[1 if small else 0][1]


def check_stored_chain_segmented(eng, monkeypatch, nblocks=70, text_bytes=600000):
    """A chain of full stored blocks in the MIDDLE of a stream that is decoded segment-wise (zh_inflate_seg.hip): the
    segments inside the chain have no block start, the decoder before them reads the chain 64 headers a step and has to
    stop where the next segment's found start is -- the compressed blocks behind the chain."""
    monkeypatch.setenv("ZH_SEG_MIN", "65536")
    monkeypatch.setenv("ZH_SEG_BYTES", "16384")
    monkeypatch.setenv("ZH_SEG_SETUP", "0")
    rnd = random.Random(99)
    noise = rnd.randbytes(nblocks * 65535)
    text = synth.gen_batch("text", 1, text_bytes, first_index=3)[0].tobytes()

    def dyn(data, last):
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        return c.compress(data) + (c.flush() if last else c.flush(zlib.Z_FULL_FLUSH))
    stored = b"".join(bytes([0]) + struct.pack("<HH", 65535, 0) + noise[o:o + 65535] for o in range(0, len(noise), 65535))
    last = bytes([1]) + struct.pack("<HH", 777, 777 ^ 0xffff) + noise[:777]
    for blob, want, holds in ((dyn(text, False) + stored + dyn(text[::-1], True), text + noise + text[::-1], None),
                              (stored + dyn(text, True), noise + text, None),
                              # nothing but stored blocks (incompressible data): chained stored blocks are segment
                              # starts too, so this stream IS decoded by many workgroups
                              (stored + last, noise + noise[:777], True)):
        assert zlib.decompress(blob, -15) == want
        before = eng.segment_stats()
        outs, sts = eng.uncompress_batch([blob], oracle.dfDeflate)
        assert sts == [0] and outs[0] == want
        cut, held = eng.segment_stats()
        assert cut - before[0] == 1, "the stream was not cut into segments"
        if holds:
            assert held - before[1] == 1, "a chain of stored blocks left to one workgroup"
    for k in ("ZH_SEG_MIN", "ZH_SEG_BYTES", "ZH_SEG_SETUP"):
        monkeypatch.delenv(k)


def segmented_streams(scale):
    """Foreign streams for the segment-wise decoder (zh_inflate_seg.hip): many blocks each, the
    kinds of block boundaries it has to cope with.  -> [(blob, format, plain)]"""
    rng = random.Random(77)
    text = synth.gen_batch("text", 1, 96 * scale, first_index=5)[0].tobytes()
    mix = synth.gen_batch("mix", 1, 64 * scale, first_index=6)[0].tobytes()
    runs = synth.gen_batch("runs", 1, 192 * scale, first_index=7)[0].tobytes()
    noise = bytes(rng.getrandbits(8) for _ in range(8 * scale))

    def blocks(plain, level, wbits, flushes, every, strategy=zlib.Z_DEFAULT_STRATEGY):
        c = zlib.compressobj(level, zlib.DEFLATED, wbits, 8, strategy)
        parts = []
        for k, o in enumerate(range(0, len(plain), every)):
            parts.append(c.compress(plain[o:o + every]))
            parts.append(c.flush(flushes[k % len(flushes)]))
        parts.append(c.flush())
        return b"".join(parts)

    out = [(zlib.compress(text, 6), oracle.dfZlib, text)]  # two blocks: one decoder does it all
    # blocks that end anywhere in a byte (Z_BLOCK) and copies that reach back across them
    out.append((blocks(text + mix, 6, -15, [zlib.Z_BLOCK], 2500), oracle.dfDeflate, text + mix))
    out.append((blocks(mix + text, 1, 15, [zlib.Z_BLOCK, zlib.Z_SYNC_FLUSH, zlib.Z_BLOCK, zlib.Z_FULL_FLUSH], 3000),
                oracle.dfZlib, mix + text))
    # fixed-code blocks only: no block start is ever found
    out.append((blocks(mix, 6, -15, [zlib.Z_BLOCK], 3000, zlib.Z_FIXED), oracle.dfDeflate, mix))
    # stored blocks (incompressible stretches) between compressed ones; a zlib stream inside, whose
    # block headers sit byte-aligned in stored blocks of the outer stream
    inner = blocks(text[: 24 * scale], 6, 15, [zlib.Z_BLOCK], 2000)
    plain = text[: 20 * scale] + noise + inner + mix[: 20 * scale] + noise + inner + text[20 * scale: 40 * scale]
    out.append((blocks(plain, 6, 31, [zlib.Z_BLOCK], 4000), oracle.dfGzip, plain))
    # long runs: a few bits per token, back-to-back maximal matches
    plain = b"\0" * (64 * scale) + text[: 8 * scale] + b"ab" * (32 * scale) + runs
    out.append((blocks(plain, 6, 15, [zlib.Z_BLOCK], 20000), oracle.dfZlib, plain))
    return out


def check_segmented(eng, scale, monkeypatch, seg_bytes):
    """Large streams on many workgroups: same bytes and statuses as the one-workgroup decode, for
    whole streams, streams cut short and streams with a flipped bit."""
    monkeypatch.setenv("ZH_SEG_MIN", str(4 * seg_bytes))
    monkeypatch.setenv("ZH_SEG_BYTES", str(seg_bytes))
    monkeypatch.setenv("ZH_SEG_SETUP", "0")  # (the set-up cost that keeps small jobs off this path)
    cases = segmented_streams(scale)
    n_foreign = len(cases)
    # this library's own streams: ONE block (the last one) however long -- every decoder but the
    # first starts inside it; raw deflate has no checksum to catch a wrong byte, only this comparison
    own_src = [synth.gen_batch(kind, 1, 72 * scale, first_index=9)[0].tobytes() for kind in ("text", "mix")]
    for fmt in (oracle.dfDeflate, oracle.dfZlib, oracle.dfGzip):
        for level in (1, 6):
            comp, sts = eng.compress_batch(own_src, level, fmt)
            assert sts == [0, 0]
            cases += [(c, fmt, p) for c, p in zip(comp, own_src) if len(c) >= 4 * seg_bytes]
    n_own = len(cases) - n_foreign
    held_foreign = 0
    for i, (blob, fmt, plain) in enumerate(cases):
        assert len(blob) >= 4 * seg_bytes, len(blob)
        before = eng.segment_stats()
        outs, sts = eng.uncompress_batch([blob], fmt)
        assert sts == [0] and outs[0] == plain, (fmt, len(blob), sts)
        # ... and by MANY workgroups: a chain of segments that does not hold is decoded by one workgroup, with the same
        # bytes -- only the count tells (zh_debug_segment_stats).  This library's own streams always hold; a foreign
        # one with fewer than four block starts (fixed codes, one block) legitimately does not.
        cut, held = (a - b for a, b in zip(eng.segment_stats(), before))
        assert cut >= 1, (i, cut)
        if i >= n_foreign:
            assert held == cut, ("own stream left to one workgroup", i, fmt, len(blob), cut, held)
        else:
            held_foreign += held == cut
    assert held_foreign >= n_foreign - 1, (held_foreign, n_foreign)
    # Bits of the payload that read like a block header and are none (about one a GiB; ZH_SEG_FAKE_START plants one):
    # the decoder before runs past it, the segments behind it still find their sub-starts (from the real header, one
    # found start further back), and the stream is still decoded segment-wise -- such a guess once cost a 1 GiB
    # stream of this library a factor of 100 (tools/gpu_big_buffer.py).
    for blob, fmt, plain in cases[n_foreign:n_foreign + 2]:
        for fake_bit in (len(blob) * 2 + 3, len(blob) * 9 // 2):
            monkeypatch.setenv("ZH_SEG_FAKE_START", str(fake_bit))
            before = eng.segment_stats()
            outs, sts = eng.uncompress_batch([blob], fmt)
            cut, held = (a - b for a, b in zip(eng.segment_stats(), before))
            assert sts == [0] and outs[0] == plain and cut >= 1 and held == cut, (fmt, len(blob), fake_bit, sts, cut, held)
    # ... and such a guess just BEFORE a real block start of the same segment (blocks of many segments: the block-parallel
    # form's, 32 KiB of input each): a segment reports the lowest position that passes for a start, so the guess used to
    # hide the start every segment of the block behind it needs; a segment keeps two candidates now and the first is
    # probed (a superchunk decoded) before anybody relies on it
    if seg_bytes <= 2048:
        src = synth.gen_batch("text", 1, 200 * scale, first_index=3)[0].tobytes()
        blob, index = eng.compress_blocks(src, 1, oracle.dfGzip, 32 * scale)
        for j in (2, 4):
            for back in (40, 300):
                monkeypatch.setenv("ZH_SEG_FAKE_START", str(index[j][0] - back))
                before = eng.segment_stats()
                outs, sts = eng.uncompress_batch([blob], oracle.dfGzip)
                cut, held = (a - b for a, b in zip(eng.segment_stats(), before))
                assert sts == [0] and outs[0] == src and cut >= 1 and held == cut, (j, back, sts, cut, held)
    monkeypatch.delenv("ZH_SEG_FAKE_START")
    # a batch of them at once (same format), small streams (no segments) in between: zlib streams
    zl = [c for c in cases if c[1] == oracle.dfZlib]
    small = [(zlib.compress(c[2][:k], 6), c[1], c[2][:k]) for c, k in zip(zl, (0, 1, 3000))]
    mixed = [x for pair in zip(zl, small) for x in pair]
    outs, sts = eng.uncompress_batch([c[0] for c in mixed], oracle.dfZlib)
    assert sts == [0] * len(mixed) and all(o == c[2] for o, c in zip(outs, mixed))
    # damage: the statuses are those of the ordinary decoder
    rng = random.Random(3)
    blob, fmt, plain = cases[1]
    bad = [blob[: len(blob) * 2 // 3], blob[:-5]]
    for _ in range(10):
        b = bytearray(blob)
        b[rng.randrange(8, len(b) - 8)] ^= 1 << rng.randrange(8)
        bad.append(bytes(b))
    monkeypatch.setenv("ZH_SEG", "0")
    want = [eng.uncompress_batch([b], fmt) for b in bad]
    monkeypatch.setenv("ZH_SEG", "1")
    got = [eng.uncompress_batch([b], fmt) for b in bad]
    for b, w, g_ in zip(bad, want, got):
        assert w[1] == g_[1], (len(b), w[1], g_[1])
        if w[1] == [0]:
            assert w[0] == g_[0]
