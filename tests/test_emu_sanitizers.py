"""The kernel sources and the host side of the library under AddressSanitizer: the g++ / emulator build of
zippy_amd/csrc (tests/hipemu) with -fsanitize=address, a run through every family of kernels -- both BestSpeed
parses and code builders, a chain level, both inflate paths on fixtures, damaged headers and streams, the `_into`
calls, plans whose scratch is forced into ranges, large streams on many workgroups with planted false block
starts -- in a child process (the sanitizer runtime has to be the first
library the process loads).  `__shared__` arrays are plain memory under the emulator, so an index that runs off
one is an error here where the hardware would read its neighbour."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(root)r)
import emu, oracle, synth, parity_cases as pc
eng = emu.engine()
inputs = [synth.corpus_file("alice29.txt")[:70000], synth.corpus_file("geo.protodata")[:40000]] + pc.edge_inputs()[:12]
pc.check_compress_identical(eng, inputs, levels=(1,), formats=(oracle.dfGzip,))
pc.check_compress_identical(eng, inputs[:2], levels=(-1, -2, 0), formats=(oracle.dfDeflate,))
pc.check_huffman_builders(eng)
pc.check_parallel_parse(eng, [b.tobytes() for b in synth.gen_batch("mix", 2, 70000)] + pc.edge_inputs()[:8])
for mode in (0, 1):
    eng.set_inflate_mode(mode)
    pc.check_fixtures(eng, max_len=40000)
    pc.check_damaged_headers(eng, 11)
    pc.check_errors_match_oracle(eng, pc.mutated_fixtures(12, seed=3, max_len=40000))
eng.set_inflate_mode(-1)
pc.check_batch_into(eng)
# large streams on many workgroups: the search queues, the two candidates a segment, sub-starts, the repair round
# (a found start that is none planted in the middle of a block and right before a real block start)
import os, zlib
os.environ.update(ZH_SEG_SETUP="0", ZH_SEG_MIN="2400", ZH_SEG_BYTES="600")
text = synth.gen_batch("text", 1, 150000, first_index=3)[0].tobytes()
blob, index = eng.compress_blocks(text, 1, oracle.dfGzip, 32768)
own = eng.compress_batch([text[:72000]], 1, oracle.dfGzip)[0][0]
for stream, plain, fakes in ((blob, text, (None, index[2][0] - 40, index[3][0] + 3000)), (own, text[:72000], (None, len(own) * 3)),
                            (zlib.compress(text, 6), text, (None, 70001))):
    for fake in fakes:
        if fake is None:
            os.environ.pop("ZH_SEG_FAKE_START", None)
        else:
            os.environ["ZH_SEG_FAKE_START"] = str(fake)
        before = eng.segment_stats()
        outs, sts = eng.uncompress_batch([stream], oracle.dfDetect)
        cut, held = (a - b for a, b in zip(eng.segment_stats(), before))
        assert sts == [0] and outs[0] == plain and cut >= 1 and held == cut, (len(stream), fake, sts, cut, held)
print("sanitized emulator run ok")
"""


def test_emulator_build_under_asan():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan.so not found")
    env = dict(os.environ, LD_PRELOAD=asan, ZH_EMU_VARIANT="asan", ZH_EMU_DEFINES="-fsanitize=address -fno-omit-frame-pointer",
               ZH_SCRATCH_MB="1", ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:detect_stack_use_after_return=0")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "tests": os.path.join(ROOT, "tests")}],
                       capture_output=True, text=True, env=env, timeout=1500)
    if r.returncode != 0 and "emu build failed" in r.stderr and "sanitize" in r.stderr:
        pytest.skip("no sanitizer runtime in this toolchain")
    assert r.returncode == 0, r.stderr[-3000:]
    assert "sanitized emulator run ok" in r.stdout
    assert "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
