"""the reference runs its tests under sanitizers in CI; the oracle (the C
restatement every parity claim rests on) gets the same treatment here: oracle/zippy_oracle.c built
with -fsanitize=address,undefined, driven through compress at every level and format, the
reference's decode fixtures, damaged streams and the multi-threaded batch entry -- in a child
process, because the sanitizer runtime has to be the first library the process loads."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import random, sys, zlib
sys.path.insert(0, %(root)r)
import oracle
import synth
rnd = random.Random(7)
inputs = [b"", b"a", b"abcd" * 5, bytes(range(256)) * 3, synth.corpus_file("alice29.txt")[:70000],
          synth.gen_batch("mix", 1, 150000)[0].tobytes(), synth.gen_batch("runs", 1, 40000)[0].tobytes(),
          rnd.randbytes(5000), b"\0" * 70000]
for level in range(-2, 10):
    for fmt in (oracle.dfGzip, oracle.dfZlib, oracle.dfDeflate):
        for src in inputs[:6] if level > 3 else inputs:
            z = oracle.compress(src, level, fmt, fname_len=3)
            assert oracle.uncompress(z, fmt) == src
for name, meta in synth.manifest()["fixtures"].items():
    assert len(oracle.uncompress(synth.fixture(name))) == meta["len"]
z = bytearray(oracle.compress(inputs[4], 1, oracle.dfGzip, fname_len=0))
for k in range(300):  # damaged streams: any status is fine, any memory error is not
    b = bytearray(z)
    for _ in range(rnd.randrange(1, 4)):
        b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
    cut = bytes(b[:rnd.randrange(1, len(b))]) if k %% 3 == 0 else bytes(b)
    try:
        oracle.uncompress(cut)
    except oracle.ZippyError:
        pass
data, index = oracle.compress_blocks(inputs[5], 1, oracle.dfGzip, 32768)
assert zlib.decompress(data, 31) == inputs[5] and len(index) >= 2
t, res = oracle.batch_mt(inputs, 0, 1, oracle.dfGzip, 3, keep=True)
t, back = oracle.batch_mt(res, 1, 1, oracle.dfGzip, 3, keep=True)
assert back == inputs
print("sanitized oracle ok")
"""


def test_oracle_under_asan_ubsan():
    lib = os.path.join(ROOT, "oracle", "libzippy_oracle_asan.so")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "libzippy_oracle_asan.so"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no sanitizer runtime in this toolchain: " + r.stderr[-300:])
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan.so not found")
    env = dict(os.environ, LD_PRELOAD=asan, ZIPPY_ORACLE_LIB=lib,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "sanitized oracle ok" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
