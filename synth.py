"""Deterministic synthetic inputs for the batch workloads.

The reference ships no batch corpus, only its single-file test data
(tests/data, mirrored as compressed fixtures in tests/golden/).  The generators
here rebuild that corpus from the committed fixtures with Python's zlib and cut
it into seeded buffers:

  G-mix   "Silesia-mix entropy": random-offset 4-64 KiB slices by class --
          text 40 %, markup/URLs 25 %, structured binary 20 %,
          incompressible 10 %, runs 5 %.
  G-runs  tests/stress.nim:13-27 restated (random byte, run length 0..255).
  G-rand  uniform bytes (forces the stored-block path, deflate.nim:274-277).
  G-zero  all zeros (max-length matches at distance 1).

Test and benchmark data only (bench.py, tests/, tools/): not part of the product package zippy_amd,
which never imports it.  Nothing here touches the oracle or the GPU.
"""
import functools
import json
import os
import zlib

import numpy as np

_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden")

SEED_BASE = 0x5A49505059000000  # "ZIPPY"

_CLASSES = [
    (0.40, ["alice29.txt", "asyoulik.txt", "lcet10.txt", "plrabn12.txt"]),
    (0.25, ["html", "urls.10K"]),
    (0.20, ["geo.protodata", "kppkn.gtb"]),
    (0.10, ["fireworks.jpg", "paper-100k.pdf"]),
    (0.05, None),  # runs
]


@functools.lru_cache(maxsize=None)
def manifest():
    with open(os.path.join(_GOLDEN, "manifest.json")) as fh:
        return json.load(fh)


@functools.lru_cache(maxsize=None)
def fixture(name):
    """Compressed fixture bytes (tests/golden/<name>)."""
    with open(os.path.join(_GOLDEN, name), "rb") as fh:
        return fh.read()


@functools.lru_cache(maxsize=None)
def corpus_file(name):
    """Uncompressed corpus file, recovered from its .gz fixture with zlib and
    verified against the SHA-256 taken from the reference's own copy."""
    import hashlib
    for fx, meta in manifest()["fixtures"].items():
        if meta["gold"] == name and fx.endswith(".gz"):
            data = zlib.decompress(fixture(fx), 47)
            assert hashlib.sha256(data).hexdigest() == meta["sha256"], name
            return data
    raise KeyError(name)


def gen_runs(rng, length):
    """tests/stress.nim:13-27: runs of a random byte, run length uniform 0..255."""
    n_runs = length // 64 + 16
    vals = rng.integers(0, 256, n_runs, dtype=np.uint8)
    lens = rng.integers(0, 256, n_runs)
    out = np.repeat(vals, lens)
    while out.size < length:
        vals = rng.integers(0, 256, n_runs, dtype=np.uint8)
        lens = rng.integers(0, 256, n_runs)
        out = np.concatenate([out, np.repeat(vals, lens)])
    return out[:length]


def gen_mix_buffer(index, size):
    """One G-mix buffer (numpy uint8), seed = SEED_BASE + index."""
    rng = np.random.default_rng(SEED_BASE + index)
    out = np.empty(size, dtype=np.uint8)
    probs = np.array([c[0] for c in _CLASSES])
    pos = 0
    while pos < size:
        cls = int(rng.choice(len(_CLASSES), p=probs))
        want = int(rng.integers(4096, 65537))
        want = min(want, size - pos)
        files = _CLASSES[cls][1]
        if files is None:
            out[pos:pos + want] = gen_runs(rng, want)
        else:
            data = np.frombuffer(corpus_file(files[int(rng.integers(len(files)))]), dtype=np.uint8)
            take = min(want, data.size)
            off = int(rng.integers(0, data.size - take + 1))
            out[pos:pos + take] = data[off:off + take]
            want = take
        pos += want
    return out


def gen_batch(kind, n_buffers, size, first_index=0):
    """Batch as one contiguous uint8 array of shape (n_buffers, size)."""
    out = np.empty((n_buffers, size), dtype=np.uint8)
    for i in range(n_buffers):
        idx = first_index + i
        if kind == "mix":
            out[i] = gen_mix_buffer(idx, size)
        elif kind == "runs":
            out[i] = gen_runs(np.random.default_rng(SEED_BASE + idx), size)
        elif kind == "rand":
            out[i] = np.random.default_rng(SEED_BASE + idx).integers(0, 256, size, dtype=np.uint8)
        elif kind == "zero":
            out[i] = 0
        elif kind == "text":  # only the text class of G-mix (tuning aid: no incompressible stretches)
            rng = np.random.default_rng(SEED_BASE + idx)
            pos = 0
            while pos < size:
                files = _CLASSES[0][1]
                data = np.frombuffer(corpus_file(files[int(rng.integers(len(files)))]), dtype=np.uint8)
                take = min(int(rng.integers(4096, 65537)), size - pos, data.size)
                off = int(rng.integers(0, data.size - take + 1))
                out[i, pos:pos + take] = data[off:off + take]
                pos += take
        elif kind == "html":  # markup only (slices of html_x_4): compresses 6-7 x -- streams that outgrow a 4 x size guess
            rng = np.random.default_rng(SEED_BASE + 0x100000 + idx)
            data = np.frombuffer(corpus_file("html_x_4"), dtype=np.uint8)
            pos = 0
            while pos < size:
                take = min(int(rng.integers(16384, 65537)), size - pos)
                off = int(rng.integers(0, data.size - take + 1))
                out[i, pos:pos + take] = data[off:off + take]
                pos += take
        else:
            raise ValueError(kind)
    return out
