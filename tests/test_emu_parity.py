"""CPU-only: the product's kernel sources (zippy_amd/csrc/*.hip) compiled with
g++ against the fiber-based HIP emulator in tests/hipemu, driven through the same
C ABI and checked against the oracle.  This proves kernel *logic* without a GPU;
the hardware parity run is tests/test_gpu_parity.py (-m gpu)."""
import pytest

import emu
import oracle
import parity_cases as pc
import synth


@pytest.fixture(scope="module")
def eng():
    return emu.engine()


@pytest.fixture(params=[0, 1], ids=["split", "serial"])
def inflate_mode(request, eng):
    """Both inflate paths (zh_inflate_split.hip, zh_inflate.hip) against the same expectations."""
    eng.set_inflate_mode(request.param)
    yield request.param
    eng.set_inflate_mode(-1)


def test_emu_fixtures_decode(eng, inflate_mode):
    pc.check_fixtures(eng, max_len=130000)


def test_emu_compress_identical_level1(eng):
    inputs = [synth.corpus_file("alice29.txt")[:100000], synth.corpus_file("geo.protodata")[:70000],
              synth.corpus_file("fireworks.jpg")[:40000]] + pc.edge_inputs()
    pc.check_compress_identical(eng, inputs, levels=(1,))


def test_emu_trailer_behind_the_emission(eng, monkeypatch):
    """Large batches write the trailer with a kernel of its own behind the emission (the checksum runs beside both the
    code builder and the emission: zh_plan_run.hip); ZH_TRAILER_LATE=1 takes the tests' small ones that way too."""
    monkeypatch.setenv("ZH_TRAILER_LATE", "1")
    inputs = [synth.corpus_file("alice29.txt")[:70000], synth.corpus_file("geo.protodata")[:40000], b"", b"abc"]
    pc.check_compress_identical(eng, inputs, levels=(1, -1, 0), formats=(oracle.dfGzip, oracle.dfZlib, oracle.dfDeflate))
    monkeypatch.setenv("ZH_TRAILER_LATE", "0")
    pc.check_compress_identical(eng, inputs[:2], levels=(1,), formats=(oracle.dfGzip,))


def test_emu_uncompress_two_halves(eng, monkeypatch):
    """Large uncompress batches run as two halves on two streams (zh_plan_run.hip: a half's writer beside the other
    half's tokens kernel); ZH_INFLATE_HALVES=4 takes the tests' batches that way: fixtures, own streams, damaged ones."""
    monkeypatch.setenv("ZH_INFLATE_HALVES", "4")
    eng.set_inflate_mode(0)
    try:
        pc.check_fixtures(eng, max_len=130000)
        pc.check_errors_match_oracle(eng, pc.mutated_fixtures(40, seed=7, max_len=40000))
    finally:
        eng.set_inflate_mode(-1)


def test_emu_compress_identical_other_levels(eng):
    inputs = [synth.corpus_file("html")[:40000], synth.corpus_file("alice29.txt")[:3000], b"",
              b"abc", b"\x00" * 5000]
    pc.check_compress_identical(eng, inputs, levels=(-2, 0, -1, 2, 9), formats=(oracle.dfDeflate,))
    pc.check_compress_identical(eng, inputs[:2], levels=(-1,), formats=(oracle.dfGzip, oracle.dfZlib))


def test_emu_tokens(eng):
    pc.check_tokens(eng, synth.corpus_file("urls.10K")[:100000], 1)
    pc.check_tokens(eng, synth.corpus_file("kppkn.gtb")[:50000], -1)
    pc.check_tokens(eng, synth.corpus_file("html")[:20000], -2)


def _chain_inputs():
    # more than a fragment, more than a window, more than one unit of the links' class sort; a second
    # input of two deflate blocks (> 4 MiB: runs and zeros keep the emulator fast)
    mixed = (synth.corpus_file("alice29.txt") + synth.corpus_file("html")[:60000] +
             synth.gen_batch("rand", 1, 20000)[0].tobytes() + synth.corpus_file("alice29.txt")[:50000])
    return mixed, b"\x00" * 2500000 + synth.gen_batch("runs", 1, 1800000)[0].tobytes()


def test_emu_chain_levels_across_windows(eng):
    mixed, two_blocks = _chain_inputs()
    pc.check_compress_identical(eng, [mixed], levels=(-1, 5), formats=(oracle.dfDeflate,))
    pc.check_tokens(eng, mixed, -1)
    pc.check_compress_identical(eng, [two_blocks], levels=(-1,), formats=(oracle.dfGzip,))
    # equal hashes mostly MORE than a window apart (random bytes: ~ 2 positions a slot in 300 KB): `head` entries that
    # have gone stale, whose links lead the reference's walk to positions of OTHER hashes (lz77.nim:88-93), text between
    stale = (synth.gen_batch("rand", 1, 150000)[0].tobytes() + synth.corpus_file("alice29.txt")[:40000] +
             synth.gen_batch("rand", 1, 150000)[0].tobytes()[:110000] + synth.corpus_file("alice29.txt")[:70000])
    pc.check_compress_identical(eng, [stale], levels=(-1, 9), formats=(oracle.dfDeflate,))
    pc.check_tokens(eng, stale, -1)


def test_emu_cross_check_kernels():
    """The test build of the library (-DZH_XCHECK: tests/hipemu/libzippy_hip_emu_xcheck.so here, `python -m
    zippy_amd.build --variant xcheck -DZH_XCHECK` on a GPU box) carries what the product library does not ship: the
    in-order link kernels on request, the every-position search, the one-wave parse (ZH_CHAIN_PREV=serial,
    ZH_CHAIN_SEARCH=dense, ZH_CHAIN_SELECT=serial), the BestSpeed matcher with its table in LDS (ZH_L1_TABLE=lds) and
    the thread-per-candidate block-start check (ZH_SEG_CHECK=serial; switches read once a process).  Each gives the
    oracle's bytes, too."""
    import os
    import subprocess
    import sys
    here, root = os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    chain = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
             "import emu, oracle, parity_cases as pc\n"
             "from test_emu_parity import _chain_inputs\n"
             "pc.check_compress_identical(emu.engine(), [_chain_inputs()[0]], levels=(-1,), formats=(oracle.dfDeflate,))\n"
             % (here, root))
    l1 = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
          "import emu, oracle, parity_cases as pc, synth\n"
          "pc.check_compress_identical(emu.engine(), [synth.corpus_file('alice29.txt')[:100000], synth.corpus_file('html')[:70000]],"
          " levels=(1,), formats=(oracle.dfGzip,))\n" % (here, root))
    seg = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
           "import emu, zlib, synth\n"
           "eng = emu.engine()\n"
           "data = synth.gen_batch('mix', 1, 700000)[0].tobytes()\n"
           "z = zlib.compress(data, 6)\n"
           "outs, sts = eng.uncompress_batch([z])\n"
           "assert sts == [0] and outs[0] == data\n"
           "cut, held = eng.segment_stats()\n"
           "assert cut >= 1 and held == cut, (cut, held)\n" % (here, root))
    base = dict(os.environ, ZH_EMU_VARIANT="xcheck", ZH_EMU_DEFINES="-DZH_XCHECK")
    for code, env in ((chain, {"ZH_CHAIN_PREV": "serial"}), (chain, {"ZH_CHAIN_SEARCH": "dense"}),
                      (chain, {"ZH_CHAIN_SELECT": "serial"}), (l1, {"ZH_L1_TABLE": "lds"}),
                      (seg, {"ZH_SEG_CHECK": "serial", "ZH_SEG_MIN": "65536", "ZH_SEG_BYTES": "8192", "ZH_SEG_SETUP": "0"})):
        r = subprocess.run([sys.executable, "-c", code], env=dict(base, **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (env, r.stderr[-2000:])


def test_product_library_ships_one_implementation():
    """nm -D of the product library (and of the emulator build of the same sources) lists none of the cross-check
    kernels; the -DZH_XCHECK build does."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = ("zh_chain_search_kernel", "zh_chain_select_kernel")
    def syms(path):
        return subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
    import build_emu
    emu_syms = syms(build_emu.build())
    assert not any(n in emu_syms for n in names)
    prod = os.path.join(root, "zippy_amd", "libzippy_hip.so")
    if os.path.exists(prod):
        out = syms(prod)
        assert "zh_plan_run" in out and not any(n in out for n in names), "a cross-check kernel in the product library"
    xc = os.path.join(root, "tests", "hipemu", "libzippy_hip_emu_xcheck.so")
    if os.path.exists(xc):
        assert all(n in syms(xc) for n in names)


def test_emu_lds_order_probe(eng):
    """zh_create asks the device whether the lanes of an LDS atomic are served in ascending order (what the
    class-sorted chain links take their order from): the emulator passes; a device that fails the probe
    (ZH_LDS_ORDER_PROBE=fail pretends one) gets the in-order kernels, says so, and still gives the oracle's bytes."""
    import os
    import subprocess
    import sys
    assert eng.chain_links_parallel()
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import emu, oracle, parity_cases as pc, synth\n"
            "eng = emu.engine()\n"
            "assert not eng.chain_links_parallel()\n"
            "assert b'in-order' in eng.lib.zh_last_error(eng._h)\n"
            "pc.check_compress_identical(eng, [synth.corpus_file('alice29.txt')[:60000]], levels=(-1, 1), formats=(oracle.dfDeflate,))\n"
            % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ZH_LDS_ORDER_PROBE="fail"), capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]


def test_emu_scratch_budget_forced_low():
    """ZH_SCRATCH_MB=1: the chain levels' scratch holds one deflate block at a time and the split decoder's token
    pool a few streams at a time -- device-resident plans run their kernels over ranges of the batch, one after
    the other through the same scratch; same bytes out (a switch read when a plan is made: child process)."""
    import os
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import emu, oracle, parity_cases as pc, synth\n"
            "from test_emu_parity import _chain_inputs\n"
            "eng = emu.engine()\n"
            "mixed, two_blocks = _chain_inputs()\n"
            "pc.check_compress_identical(eng, [two_blocks, mixed[:90000], b'', mixed[:40000]], levels=(-1,), formats=(oracle.dfGzip,))\n"
            "pc.check_tokens(eng, two_blocks, -1)\n"
            "bufs = [b.tobytes() for b in synth.gen_batch('mix', 14, 70001)] + [b'', two_blocks[:300000]]\n"
            "for mode in (0, 1):\n"
            "    eng.set_inflate_mode(mode)\n"
            "    pc.check_roundtrip(eng, bufs, 1)\n"
            "    pc.check_errors_match_oracle(eng, pc.mutated_fixtures(12, seed=5, max_len=40000))\n"
            % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ZH_SCRATCH_MB="1", ZH_PIPE_MIN=str(1 << 60)),
                       capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-3000:]


def test_emu_multi_block_buffer(eng):
    # > 4 MiB: two deflate blocks in one buffer (deflate.nim:228-237); runs/zeros keep it fast
    src = (b"\x00" * 3000000 + synth.gen_batch("runs", 1, 1300000)[0].tobytes())
    pc.check_compress_identical(eng, [src], levels=(1,), formats=(oracle.dfGzip,))
    pc.check_tokens(eng, src, 1)


def test_emu_block_parallel_form(eng):
    # BASELINE config 5 in small: independent 32 KiB / 64 KiB blocks + index, all matcher families
    src = (synth.corpus_file("alice29.txt")[:70000] + synth.gen_batch("rand", 1, 40000)[0].tobytes() +
           b"\x00" * 30000 + synth.corpus_file("html")[:25000])
    pc.check_blocks(eng, src, levels=(1,), block_sizes=(32768, 65536, 4194304),
                    formats=(oracle.dfGzip, oracle.dfDeflate))
    pc.check_blocks(eng, src[:100000], levels=(-1, 0, -2), block_sizes=(32768,), formats=(oracle.dfZlib,))
    pc.check_blocks(eng, b"", levels=(1,), block_sizes=(32768,))
    pc.check_blocks(eng, b"x" * 32768, levels=(1,), block_sizes=(32768,))
    pc.check_blocks_bad_index(eng, src)


def test_emu_zip_archives(eng, inflate_mode):
    # tests/test_ziparchives_read.nim / _write.nim through the batch clients
    assert pc.check_zip_extract(eng, pc.zip_fixture("cat.jpg")) == 3
    image = pc.zip_fixture("Bagnon-10.2.31.zip")
    assert pc.check_zip_extract(eng, image) > 100
    pc.check_zip_create(eng, [("README.txt", b"Hello, World!")])
    pc.check_zip_create(eng, {"a/b.txt": synth.corpus_file("alice29.txt")[:40000], "empty": b"",
                              "caf\u00e9.bin": bytes(range(256)) * 40})
    pc.check_zip_errors(eng)


def test_emu_tarballs(eng, inflate_mode):
    # tests/test_tarballs_read.nim in small (the 20 MB fixture itself runs on the GPU)
    files = [("a.txt", b"hello"), ("d", None), ("d/b.bin", synth.corpus_file("html")[:30000])]
    assert pc.check_tarball(eng, pc.make_tar_gz(files, "d/" + "y" * 140 + ".txt")) == 5
    import gzip
    assert pc.check_tarball(eng, gzip.decompress(pc.make_tar_gz(files))) == 4  # a plain .tar
    pc.check_tar_errors(eng)


def test_emu_roundtrip_and_random_fname(eng, inflate_mode):
    bufs = [b.tobytes() for b in synth.gen_batch("mix", 4, 65536)]
    pc.check_roundtrip(eng, bufs, 1)
    pc.check_gzip_random_fname(eng, bufs[0][:5000])


def test_emu_checksums(eng):
    import random
    pc.check_checksums(eng, [random.Random(n).randbytes(n) for n in (0, 1, 15, 16, 1023, 1024,
                                                                      32768, 32769, 100001)])


def test_emu_damaged_streams(eng, inflate_mode):
    pc.check_errors_match_oracle(eng, pc.mutated_fixtures(60, seed=99, max_len=40000))
    pc.check_error_statuses(eng)


def test_emu_damaged_headers(eng, inflate_mode):
    pc.check_damaged_headers(eng, step=3)


def test_emu_zlib_and_raw_need_sizing_pass(eng, inflate_mode):
    import zlib
    src = synth.corpus_file("alice29.txt")[:60000]
    for wb, fmt in ((15, oracle.dfDetect), (15, oracle.dfZlib), (-15, oracle.dfDeflate)):
        co = zlib.compressobj(6, zlib.DEFLATED, wb)
        blob = co.compress(src) + co.flush()
        assert eng.uncompress(blob, fmt) == src


def test_emu_level1_many_fragments(eng):
    """Wider net for rare parse situations (hash collisions inside one probe step)."""
    bufs = [b.tobytes() for b in synth.gen_batch("mix", 160, 65536, first_index=5000)]
    outs, sts = eng.compress_batch(bufs, 1, oracle.dfDeflate)
    assert all(s == 0 for s in sts)
    for src, out in zip(bufs, outs):
        assert out == oracle.deflate(src, 1)


def test_emu_parallel_parse_contract(eng):
    """zh_l1p_match_kernel under the emulator: valid streams, size within 2 % of the oracle's."""
    inputs = [b.tobytes() for b in synth.gen_batch("mix", 5, 100000)]
    inputs += [b.tobytes() for b in synth.gen_batch("runs", 2, 70001)]
    inputs += pc.edge_inputs()
    pc.check_parallel_parse(eng, inputs, formats=(oracle.dfGzip, oracle.dfDeflate))


def test_emu_huffman_builders(eng):
    pc.check_huffman_builders(eng)


def test_emu_wide_code_length_counts(eng):
    pc.check_wide_code_length_counts(eng)


def test_emu_batch_into(eng):
    pc.check_batch_into(eng)


def test_emu_ragged_staging(eng):
    pc.check_ragged_staging(eng, 1)


def test_emu_unsized_streams(eng, inflate_mode):
    pc.check_unsized_streams(eng)


def test_emu_plan_slots_with_gaps(eng):
    import ctypes
    def upload(b):
        buf = ctypes.create_string_buffer(b, len(b))
        return ctypes.addressof(buf), buf
    def alloc(n, fill):
        buf = ctypes.create_string_buffer(bytes([fill]) * n, n)
        return ctypes.addressof(buf), buf
    pc.check_plan_slots_with_gaps(eng, upload, lambda keep: keep.raw, alloc)


def test_emu_plan_reruns_cheapest_last(eng):
    import ctypes
    def upload(b):
        buf = ctypes.create_string_buffer(b, len(b))
        return ctypes.addressof(buf), buf
    def alloc(n, fill):
        buf = ctypes.create_string_buffer(bytes([fill]) * n, n)
        return ctypes.addressof(buf), buf
    pc.check_plan_reruns(eng, upload, lambda keep: keep.raw, alloc, 9, 300000)  # (90 fragments on the emulator's 64 waves)


def test_emu_plan_pack(eng):
    import ctypes
    def upload(b):
        buf = ctypes.create_string_buffer(b, len(b))
        return ctypes.addressof(buf), buf
    def alloc(n, fill):
        buf = ctypes.create_string_buffer(bytes([fill]) * n, n)
        return ctypes.addressof(buf), buf
    pc.check_plan_pack(eng, upload, lambda keep: keep.raw, alloc)


def test_emu_stored_chains(eng):
    """(the split decoder only, 66-block chains: the serial decoder reads stored blocks the way it always did, and
    the GPU suite runs both modes at 200 blocks)"""
    eng.set_inflate_mode(0)
    try:
        pc.check_stored_chains(eng, 66, small=True)
    finally:
        eng.set_inflate_mode(-1)


def test_emu_stored_chain_segmented(eng, monkeypatch):
    pc.check_stored_chain_segmented(eng, monkeypatch, 66, 200000)


def test_emu_split_inflate_edges(eng, inflate_mode):
    pc.check_split_inflate_edges(eng)


def test_emu_segmented_streams(eng, monkeypatch):
    pc.check_segmented(eng, 1024, monkeypatch, 2048)
    pc.check_segmented(eng, 1024, monkeypatch, 600)  # chains of more than two window groups
