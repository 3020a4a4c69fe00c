"""Pins the CPU oracle (oracle/zippy_oracle.c) against everything the reference's
own tests hold for the codec path.  CPU only."""
import hashlib
import heapq
import random
import zlib

import numpy as np
import pytest

import oracle
import synth

# tests/test.nim:16-39
TEST_GOLDS = ["randtest1.gold", "randtest2.gold", "randtest3.gold", "rfctest1.gold",
              "rfctest2.gold", "rfctest3.gold", "zerotest1.gold", "zerotest2.gold", "empty.gold",
              "alice29.txt", "asyoulik.txt", "fireworks.jpg", "geo.protodata", "html", "html_x_4",
              "kppkn.gtb", "lcet10.txt", "paper-100k.pdf", "plrabn12.txt", "urls.10K"]
# tests/test_levels.nim:3-15
LEVEL_GOLDS = ["randtest1.gold", "rfctest1.gold", "zerotest1.gold", "empty.gold", "alice29.txt",
               "asyoulik.txt", "fireworks.jpg", "geo.protodata", "html", "kppkn.gtb",
               "paper-100k.pdf"]
WBITS = {oracle.dfDeflate: -15, oracle.dfZlib: 15, oracle.dfGzip: 31}


def test_known_answer_decode_all_fixtures(manifest):
    """tests/test.nim:41-60 + tests/test_known_bad.nim:3 + the other .gz files
    beside the corpus: 25 fixtures, byte-exact."""
    assert len(manifest["fixtures"]) == 25
    for name, meta in manifest["fixtures"].items():
        out = oracle.uncompress(synth.fixture(name))
        assert len(out) == meta["len"], name
        assert "%08x" % oracle.crc32(out) == meta["crc32"], name
        assert hashlib.sha256(out).hexdigest() == meta["sha256"], name


def test_known_bad_nitter_len():
    assert len(oracle.uncompress(synth.fixture("known_bad_nitter.json.gz"))) == 574


@pytest.mark.parametrize("fmt", [oracle.dfDeflate, oracle.dfZlib, oracle.dfGzip])
def test_roundtrip_three_formats(golds, fmt):
    """tests/test.nim:62-85 (default level, + the 256-byte ramp)."""
    cases = [golds[g] for g in TEST_GOLDS] + [bytes(range(256))]
    for original in cases:
        comp = oracle.compress(original, dataFormat=fmt, fname_len=-1)
        back = oracle.uncompress(comp, oracle.dfDeflate if fmt == oracle.dfDeflate else
                                 oracle.dfDetect)
        assert back == original
        assert zlib.decompress(comp, WBITS[fmt]) == original  # tests/validate.nim direction 1


@pytest.mark.parametrize("level", list(range(-2, 10)))
def test_roundtrip_all_levels(golds, level):
    """tests/test_levels.nim:18-25"""
    for g in LEVEL_GOLDS:
        comp = oracle.compress(golds[g], level, fname_len=-1)
        assert oracle.uncompress(comp) == golds[g]
        assert zlib.decompress(comp, 31) == golds[g]


def test_cross_encoder_zlib_to_oracle(golds):
    """tests/stress2.nim:8-20 and tests/validate.nim direction 2."""
    base = golds["rfctest3.gold"]
    for mult in (1, 2, 7, 33):
        for level in (1, -1):
            data = base * mult
            assert oracle.uncompress(zlib.compress(data, level)) == data
    for g in TEST_GOLDS:
        for wb, fmt in ((-15, oracle.dfDeflate), (15, oracle.dfDetect), (31, oracle.dfDetect)):
            co = zlib.compressobj(6, zlib.DEFLATED, wb)
            comp = co.compress(golds[g]) + co.flush()
            if fmt == oracle.dfDetect and len(comp) <= 6:
                continue
            assert oracle.uncompress(comp, fmt) == golds[g]


def test_hand_derived_kats():
    """SURVEY.md 9.3."""
    assert oracle.deflate(b"", 1).hex() == "010000ffff"
    for level in (0, -1, 2, 9):
        assert oracle.deflate(b"", level).hex() == "010000ffff"
    assert oracle.deflate(b"", -2).hex() == "0300"
    assert oracle.deflate(b"a", 1).hex() == "010100feff61"
    assert oracle.compress(b"", 1, oracle.dfZlib).hex() == "7801010000ffff00000001"
    gz = oracle.compress(b"", 1, oracle.dfGzip, fname_len=3)
    assert gz.hex() == "1f8b0808000000000000" + "616263" + "00" + "010000ffff" + "00" * 8
    ramp = bytes(range(256))
    assert oracle.deflate(ramp, -1) == bytes.fromhex("010001fffe") + ramp


def test_oracle_kat_regression(manifest, golds):
    """The oracle's own raw-deflate output per (file, level) is pinned by hash so
    the restatement cannot drift silently; the HIP encoder targets the same bytes."""
    for name, per_level in manifest["oracle_kat"].items():
        src = golds[name]
        for level, meta in per_level.items():
            body = oracle.deflate(src, int(level))
            assert len(body) == meta["len"], (name, level)
            assert hashlib.sha256(body).hexdigest() == meta["sha256"], (name, level)


def test_stored_thresholds():
    """deflate.nim:274-277: float32(blockLen)*0.98 truncated."""
    rnd = random.Random(1).randbytes(65536)
    assert oracle.deflate(rnd, 1)[:5].hex() == "00ffff0000"  # first stored chunk of 65535
    assert len(oracle.deflate(rnd, 1)) == 65536 + 10


def test_checksums_vs_zlib(golds):
    for g in TEST_GOLDS:
        assert oracle.crc32(golds[g]) == zlib.crc32(golds[g])
        assert oracle.adler32(golds[g]) == zlib.adler32(golds[g])
    for n in (0, 1, 7, 8, 9, 63, 64, 65, 5551, 5552, 5553, 100000):
        d = random.Random(n).randbytes(n)
        assert oracle.crc32(d) == zlib.crc32(d)
        assert oracle.adler32(d) == zlib.adler32(d)


class _N:
    __slots__ = ("freq", "sym", "left", "right")

    def __init__(self, freq, sym=-1, left=None, right=None):
        self.freq, self.sym, self.left, self.right = freq, sym, left, right

    def __lt__(self, other):  # deflate.nim:10-11: compares freq only
        return self.freq < other.freq


def _heapq_code_lengths(freq):
    """deflate.nim:47-75 on top of CPython's heapq, which Nim's std/heapqueue
    is a port of.  Only valid when no length limiting triggers."""
    nodes = [_N(int(f), s) for s, f in enumerate(freq) if f > 0]
    heap = []
    for n in nodes:
        heapq.heappush(heap, n)
    while len(heap) >= 2:
        left = heapq.heappop(heap)
        right = heapq.heappop(heap)
        heapq.heappush(heap, _N(left.freq + right.freq, -1, left, right))
    lens = {}

    def visit(n, level):
        if n.sym == -1:
            visit(n.left, level + 1)
            visit(n.right, level + 1)
        else:
            lens[n.sym] = level
    visit(heap[0], 0)
    return lens


def test_huffman_matches_python_heapq():
    rng = np.random.default_rng(7)
    for trial in range(300):
        n = int(rng.integers(2, 287))
        kind = trial % 4
        if kind == 0:
            freq = rng.integers(0, 50, n)
        elif kind == 1:
            freq = rng.integers(0, 4, n)  # many ties
        elif kind == 2:
            freq = (rng.pareto(1.0, n) * 10).astype(np.int64)
        else:
            freq = rng.integers(1, 3, n)
        freq = np.minimum(freq, 2**31 - 1).astype(np.uint32)
        if (freq > 0).sum() < 2:
            continue
        want = _heapq_code_lengths(freq)
        if max(want.values()) > 15:
            continue
        _, lens = oracle.huffman_codes(freq, 1, 15)
        for s, l in want.items():
            assert lens[s] == l, (trial, s)


def test_huffman_length_limited_is_valid_prefix_code():
    # Fibonacci-like frequencies force depth > 15 (and > 7) -> rebalancing path
    fib = [1, 1]
    while len(fib) < 40:
        fib.append(fib[-1] + fib[-2])
    for n, limit in ((30, 15), (19, 7), (40, 15)):
        freq = np.array(fib[:n], dtype=np.uint32)
        codes, lens = oracle.huffman_codes(freq, 1, limit)
        assert lens.max() <= limit and (lens[:n] > 0).all()
        assert sum(2.0 ** -int(l) for l in lens if l) <= 1.0 + 1e-12
        seen = set()
        for c, l in zip(codes, lens):
            if l:
                key = format(int(c), "0%db" % l)[::-1]  # codes are stored bit-reversed
                assert not any(key.startswith(s) or s.startswith(key) for s in seen)
                seen.add(key)


def test_huffman_special_cases():
    """deflate.nim:34-45"""
    _, lens = oracle.huffman_codes(np.zeros(30, np.uint32), 2, 15)
    assert list(lens) == [1, 1, 0]
    f = np.zeros(30, np.uint32)
    f[5] = 9
    _, lens = oracle.huffman_codes(f, 2, 15)
    assert lens[5] == 1 and lens[0] == 1 and lens.sum() == 2
    f = np.zeros(30, np.uint32)
    f[0] = 9
    _, lens = oracle.huffman_codes(f, 2, 15)
    assert list(lens) == [1, 1, 0]


def _decode_tokens(tokens, src):
    """Replays a token stream against src; returns bytes."""
    out = bytearray()
    i = 0
    while i < len(tokens):
        t = int(tokens[i])
        if t & 0x8000:
            off, length = int(tokens[i + 1]), int(tokens[i + 2])
            for _ in range(length):
                out.append(out[-off])
            i += 3
        else:
            out += src[len(out):len(out) + t]
            i += 1
    return bytes(out)


@pytest.mark.parametrize("level", [1, -1, 2, 9, -2])
def test_token_stream_replays(golds, level):
    for g in ("alice29.txt", "geo.protodata", "zerotest2.gold", "fireworks.jpg"):
        src = golds[g][:300000]
        toks, litlen, dist, numlit = oracle.block_tokens(src, level)
        assert _decode_tokens(toks, src) == src
        assert litlen[256] == 1
        assert litlen[:256].sum() == numlit


def test_snappy_fragment_independence(golds):
    """snappy.nim:150-163: LZ history never crosses a 32 KiB fragment, so the
    token stream of a block is the concatenation of per-fragment streams."""
    src = golds["urls.10K"][:200000]
    whole, *_ = oracle.block_tokens(src, 1)
    parts = [oracle.block_tokens(src[o:o + 32768], 1)[0] for o in range(0, len(src), 32768)]
    assert np.array_equal(whole, np.concatenate(parts))


def test_mutation_and_truncation_raise_only_zippy_error(manifest):
    """tests/fuzz.nim:16-33 restated with a fixed seed."""
    files = ["randtest1.gz", "randtest2.gz", "randtest3.gz", "rfctest1.gz", "rfctest2.gz",
             "rfctest3.gz", "zerotest1.gz", "zerotest2.gz"]
    rng = random.Random(1234)
    for _ in range(1500):
        comp = bytearray(synth.fixture(rng.choice(files)))
        pos = rng.randrange(len(comp))
        comp[pos] = rng.randrange(256)
        for blob in (bytes(comp), bytes(comp[:pos])):
            try:
                oracle.uncompress(blob)
            except oracle.ZippyError:
                pass


def test_stress_runs_roundtrip():
    """tests/stress.nim:10-58 restated (fixed seeds; BestSpeed really passed)."""
    for seed in range(40):
        rng = np.random.default_rng(seed)
        length = int(rng.integers(0, 100001))
        data = synth.gen_runs(rng, length).tobytes()
        shuffled = bytes(rng.permutation(np.frombuffer(data, dtype=np.uint8)))
        for level in (1, -1):
            for d in (data, shuffled):
                comp = oracle.compress(d, level, fname_len=-1)
                assert oracle.uncompress(comp) == d
                assert zlib.decompress(comp, 31) == d


def test_error_categories():
    E = oracle.ZippyError
    with pytest.raises(E):
        oracle.compress(b"x", 10)
    with pytest.raises(E):
        oracle.compress(b"x", -3)
    with pytest.raises(E):
        oracle.compress(b"x", 1, oracle.dfDetect)
    with pytest.raises(E):
        oracle.uncompress(b"\x00" * 30)  # detect fails
    gz = bytearray(oracle.compress(b"hello world" * 10, 1, fname_len=0))
    bad = bytearray(gz)
    bad[3] |= 4  # FEXTRA -> gzip.nim:40-41
    with pytest.raises(E) as ei:
        oracle.uncompress(bytes(bad))
    assert ei.value.status == 12
    bad = bytearray(gz)
    bad[-5] ^= 1  # CRC
    with pytest.raises(E) as ei:
        oracle.uncompress(bytes(bad))
    assert ei.value.status == 8
    bad = bytearray(gz)
    bad[-1] ^= 1  # ISIZE
    with pytest.raises(E) as ei:
        oracle.uncompress(bytes(bad))
    assert ei.value.status == 9
    with pytest.raises(E) as ei:
        oracle.inflate(b"\x07")  # BTYPE 3 -> inflate.nim:288-289
    assert ei.value.status == 17
