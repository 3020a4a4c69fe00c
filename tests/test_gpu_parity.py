"""Parity proper (-m gpu): hand-written gfx950 kernels through the C ABI
(zippy_amd/libzippy_hip.so) against the oracle, on a real MI355X."""
import hashlib
import random
import zlib

import numpy as np
import pytest

import oracle
import parity_cases as pc
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torch  # torch's bundled HIP runtime has to initialise before libzippy_hip.so's
    torch.cuda.init()
    from zippy_amd import api
    e = api.engine()
    e.set_gzip_fname_len(0)
    return e


@pytest.fixture(params=[0, 1], ids=["split", "serial"])
def inflate_mode(request, eng):
    """Both inflate paths (zh_inflate_split.hip, zh_inflate.hip) against the same expectations."""
    eng.set_inflate_mode(request.param)
    yield request.param
    eng.set_inflate_mode(-1)


def test_gpu_lds_order_probe(eng):
    """zh_create's known-answer launch: this MI355X serves the lanes of a returning LDS atomic in ascending
    order, so the chain levels' links come from the class-sorted kernels (byte identity of those levels is what
    the compress tests below check; a device that failed here would run the in-order kernels instead)."""
    assert eng.chain_links_parallel(), eng.lib.zh_last_error(eng._h)


def test_gpu_cross_check_kernels():
    """The test build (-DZH_XCHECK, libzippy_hip_xcheck.so: built by __graft_entry__.build()) with each cross-check
    switched on in a child process: the in-order links, the every-position search, the one-wave parse, the BestSpeed
    matcher with its table in LDS, the thread-per-candidate block-start check -- the oracle's bytes every time; and
    the product library exports none of those kernels."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from zippy_amd import build
    lib = build.build(variant="xcheck", defines=["-DZH_XCHECK"])
    chain = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
             "import torch; torch.cuda.init()\n"
             "import oracle, synth, parity_cases as pc\n"
             "from zippy_amd import api\n"
             "eng = api.engine(); eng.set_gzip_fname_len(0)\n"
             "bufs = [b.tobytes() for b in synth.gen_batch('mix', 6, 1 << 20)]\n"
             "pc.check_compress_identical(eng, bufs, levels=(-1,), formats=(oracle.dfGzip,))\n"
             % (os.path.join(root, "tests"), root))
    l1 = chain.replace("levels=(-1,)", "levels=(1,)")
    seg = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
           "import torch; torch.cuda.init()\n"
           "import zlib, synth\n"
           "from zippy_amd import api\n"
           "eng = api.engine()\n"
           "data = synth.gen_batch('mix', 1, 24 << 20)[0].tobytes()\n"
           "outs, sts = eng.uncompress_batch([zlib.compress(data, 6)])\n"
           "assert sts == [0] and outs[0] == data\n"
           "cut, held = eng.segment_stats()\n"
           "assert cut == 1 and held == 1, (cut, held)\n" % (os.path.join(root, "tests"), root))
    for code, env in ((chain, {"ZH_CHAIN_PREV": "serial"}), (chain, {"ZH_CHAIN_SEARCH": "dense"}),
                      (chain, {"ZH_CHAIN_SELECT": "serial"}), (l1, {"ZH_L1_TABLE": "lds"}), (seg, {"ZH_SEG_CHECK": "serial"})):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ZIPPY_HIP_LIB=lib, **env), capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, (env, r.stdout[-1000:], r.stderr[-3000:])
    def syms(path):
        return subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
    names = ("zh_chain_search_kernel", "zh_chain_select_kernel")
    assert all(n in syms(lib) for n in names)
    assert not any(n in syms(build.build()) for n in names), "a cross-check kernel in the product library"


def test_gpu_l1_pool_memory_switch():
    """ZH_L1_POOL=uncached / fine (read when a plan is made: child processes): the BestSpeed matcher's table pool as an
    allocation of its own in uncached / fine-grained memory (the measurement switch of DESIGN.md 4.1) -- the oracle's
    bytes either way, more fragments than waves so that tables are reused across fragments."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import torch; torch.cuda.init()\n"
            "import oracle, synth, parity_cases as pc\n"
            "from zippy_amd import api\n"
            "eng = api.engine(); eng.set_gzip_fname_len(0)\n"
            "bufs = [b.tobytes() for b in synth.gen_batch('mix', 200, 1 << 20)]\n"
            "outs, sts = eng.compress_batch(bufs, 1, oracle.dfGzip)\n"
            "assert all(s == 0 for s in sts)\n"
            "for i in range(0, 200, 9): assert outs[i] == oracle.compress(bufs[i], 1, oracle.dfGzip, fname_len=0), i\n"
            % (os.path.join(root, "tests"), root))
    for mode in ("uncached", "fine"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ZH_L1_POOL=mode), capture_output=True, text=True,
                           timeout=900)
        assert r.returncode == 0, (mode, r.stdout[-1000:], r.stderr[-3000:])


def test_gpu_damaged_headers(eng, inflate_mode):
    """Every bit of three dynamic headers flipped in turn (2 160 raw deflate streams): the wave-parallel header
    reader, its fall-back to the serial one and the workgroup-built tables against the oracle."""
    pc.check_damaged_headers(eng)


def test_gpu_scratch_budget_forced_low():
    """ZH_SCRATCH_MB=64 (read when a plan is made: child process): 24 x 1 MiB at DefaultCompression need 288 MiB of
    chain scratch and run as ranges of five blocks; 48 x 1 MiB streams need 200 MiB of token records and decode
    as groups of streams -- through the same scratch, one after the other; same bytes as the oracle's."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import torch; torch.cuda.init()\n"
            "import oracle, parity_cases as pc, synth\n"
            "from zippy_amd import api\n"
            "eng = api.engine(); eng.set_gzip_fname_len(0)\n"
            "bufs = [b.tobytes() for b in synth.gen_batch('mix', 24, 1 << 20)]\n"
            "outs, sts = eng.compress_batch(bufs, -1, oracle.dfGzip)\n"
            "assert all(s == 0 for s in sts)\n"
            "for i in (0, 4, 5, 11, 23):\n"
            "    assert outs[i] == oracle.compress(bufs[i], -1, oracle.dfGzip, fname_len=0), i\n"
            "more = [b.tobytes() for b in synth.gen_batch('mix', 48, 1 << 20)]\n"
            "blobs, sts = eng.compress_batch(more, 1, oracle.dfGzip)\n"
            "back, sts = eng.uncompress_batch(blobs + outs, oracle.dfGzip)\n"
            "assert all(s == 0 for s in sts) and back == more + bufs\n" % (here, os.path.dirname(here)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ZH_SCRATCH_MB="64", ZH_PIPE_MIN=str(1 << 60), ZH_TRACE="1"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "ranges of blocks" in r.stderr and "groups of streams" in r.stderr, r.stderr[-2000:]


def test_gpu_fixtures_decode(eng, inflate_mode):
    pc.check_fixtures(eng)


def test_gpu_compress_identical_all_levels_corpus(eng, golds):
    """tests/test_levels.nim:18-25 and tests/test.nim:62-85, tightened from
    'round-trips' to 'byte-identical to the oracle'."""
    names = ["randtest1.gold", "rfctest1.gold", "zerotest1.gold", "empty.gold", "alice29.txt",
             "asyoulik.txt", "fireworks.jpg", "geo.protodata", "html", "kppkn.gtb",
             "paper-100k.pdf"]
    inputs = [golds[n] for n in names]
    pc.check_compress_identical(eng, inputs, levels=range(-2, 10), formats=(oracle.dfGzip,))
    pc.check_compress_identical(eng, inputs, levels=(1, -1), formats=(oracle.dfDeflate, oracle.dfZlib))


def test_gpu_oracle_kat_hashes(eng, manifest, golds):
    """Committed hashes of the oracle's output (tests/golden/manifest.json)."""
    for name, per_level in manifest["oracle_kat"].items():
        for level in ("1", "-1", "-2", "0", "9"):
            out = eng.compress(golds[name], int(level), oracle.dfDeflate)
            assert len(out) == per_level[level]["len"], (name, level)
            assert hashlib.sha256(out).hexdigest() == per_level[level]["sha256"], (name, level)


def test_gpu_edge_inputs(eng):
    pc.check_compress_identical(eng, pc.edge_inputs(), levels=(1, -1, -2, 0))


def test_gpu_tokens(eng, golds):
    for name in ("urls.10K", "alice29.txt", "geo.protodata", "fireworks.jpg", "zerotest3.gold"):
        pc.check_tokens(eng, golds[name], 1)
    pc.check_tokens(eng, golds["kppkn.gtb"], -1)
    pc.check_tokens(eng, golds["html"], 9)
    pc.check_tokens(eng, golds["html"], -2)


def test_gpu_config2_batch_1024x64k_bestspeed(eng):
    """BASELINE.json configs[1]: 1024 x 64 KiB G-mix, BestSpeed, every buffer
    byte-identical to the oracle and decodable by zlib."""
    bufs = [b.tobytes() for b in synth.gen_batch("mix", 1024, 65536)]
    outs, sts = eng.compress_batch(bufs, 1, oracle.dfGzip)
    assert all(s == 0 for s in sts)
    for src, out in zip(bufs, outs):
        assert out == oracle.compress(src, 1, oracle.dfGzip, fname_len=0)
    for src, out in zip(bufs[::37], outs[::37]):
        assert zlib.decompress(out, 31) == src
    back, sts = eng.uncompress_batch(outs)
    assert all(s == 0 for s in sts) and back == bufs


def test_gpu_config3_uncompress_1mib_streams(eng):
    """BASELINE.json configs[2] at test size: 1 MiB streams pre-gzipped on the host by
    the oracle (level 1) and by system zlib (level 6, multi-block foreign streams)."""
    bufs = [b.tobytes() for b in synth.gen_batch("mix", 48, 1 << 20)]
    blobs = [oracle.compress(b, 1, oracle.dfGzip, fname_len=i % 26) for i, b in enumerate(bufs)]
    blobs += [zlib.compress(b, 6) for b in bufs[:16]]
    co = [zlib.compressobj(6, zlib.DEFLATED, 31) for _ in range(16)]
    blobs += [c.compress(b) + c.flush() for c, b in zip(co, bufs[:16])]
    outs, sts = eng.uncompress_batch(blobs)
    assert all(s == 0 for s in sts), sts
    assert outs == bufs + bufs[:16] + bufs[:16]


def test_gpu_level1_identical_at_scale(eng):
    """Every table slot of the matcher busy at once (512 x 1 MiB = 16384 fragments over 4096
    persistent waves) and every stream slot of inflate: outputs still equal the oracle's, byte
    for byte (a stale hash-table read would only change the parse, not break the round trip)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    bufs = [b.tobytes() for b in synth.gen_batch("mix", 512, 1 << 20, first_index=9000)]
    outs, sts = eng.compress_batch(bufs, 1, oracle.dfGzip)
    assert all(s == 0 for s in sts)
    with ThreadPoolExecutor(min(32, os.cpu_count() or 1)) as ex:
        want = list(ex.map(lambda b: oracle.compress(b, 1, oracle.dfGzip, fname_len=0), bufs))
    assert outs == want
    back, sts = eng.uncompress_batch(outs)
    assert all(s == 0 for s in sts) and back == bufs


def test_gpu_parallel_parse_contract(eng, golds):
    """Opt-in parallel BestSpeed parse (include/zippy_hip.h zh_set_l1_parse): not the oracle's bytes,
    but valid streams the oracle's uncompress and zlib decode to the input, at most 2 % larger in
    total than the oracle's -- on the corpus files, the edge sizes and every data kind."""
    corpus = [golds[n] for n in ("alice29.txt", "asyoulik.txt", "fireworks.jpg", "geo.protodata", "html",
                                 "kppkn.gtb", "paper-100k.pdf", "zerotest1.gold", "randtest1.gold")]
    pc.check_parallel_parse(eng, corpus, formats=(oracle.dfGzip, oracle.dfZlib, oracle.dfDeflate))
    pc.check_parallel_parse(eng, pc.edge_inputs())
    for kind in ("mix", "runs", "text"):
        inputs = [b.tobytes() for b in synth.gen_batch(kind, 24, 1 << 20)]
        dev, ref = pc.check_parallel_parse(eng, inputs)
        print("parallel parse, %s: %d B against the oracle's %d B (%.4f)" % (kind, dev, ref, dev / ref))


def test_gpu_huffman_builders(eng):
    pc.check_huffman_builders(eng)


def test_gpu_wide_code_length_counts(eng):
    """256 symbols of one code length (deflate.nim:136-139 would wrap its uint8 count)."""
    pc.check_wide_code_length_counts(eng)


def test_gpu_batch_into(eng):
    pc.check_batch_into(eng)


def test_gpu_unsized_streams(eng, inflate_mode):
    pc.check_unsized_streams(eng)


def test_gpu_host_api_ragged_staging(eng):
    """~135 MiB through the host-buffer API in 32 MiB staging chunks: buffers and results that
    straddle chunk and thread borders, empty buffers, a damaged member."""
    pc.check_ragged_staging(eng, 128)


def test_gpu_config4_default_compression_ratio(eng):
    """BASELINE.json configs[3] at test size: DefaultCompression, identical to the oracle."""
    bufs = [b.tobytes() for b in synth.gen_batch("mix", 12, 1 << 20)]
    outs, sts = eng.compress_batch(bufs, -1, oracle.dfGzip)
    assert all(s == 0 for s in sts)
    for src, out in zip(bufs, outs):
        assert out == oracle.compress(src, -1, oracle.dfGzip, fname_len=0)


def test_gpu_config4_whole_batch_on_one_gpu(eng):
    """BASELINE.json configs[3] unsharded (4096 x 1 MiB, DefaultCompression, on one GPU):
    2^32 positions, more than one launch of the thread-per-position search may take.  bench.py
    checks every status, every length and (inside uncompress) every CRC-32 of the round trip."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--level", "-1", "--steps", "1",
                        "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] > 0 and 2.4 < line["ratio"] < 2.9


def test_gpu_config5_single_large_buffer(eng):
    """BASELINE.json configs[4] substitute (tor-list.gold is absent): one 40 MiB buffer =
    10 deflate blocks x 128 LZ-independent fragments, compress + uncompress."""
    parts = synth.gen_batch("mix", 40, 1 << 20)
    src = parts.tobytes()
    out = eng.compress(src, 1, oracle.dfGzip)
    assert out == oracle.compress(src, 1, oracle.dfGzip, fname_len=0)
    assert eng.uncompress(out) == src


def test_gpu_config5_block_parallel_100mib(eng):
    """BASELINE.json configs[4]: one 100 MiB+ buffer as independent 32 KiB deflate blocks; the
    stream and its block index equal the oracle's, the indexed decode (one decoder per block)
    and the plain decode both return the input."""
    src = synth.gen_batch("mix", 104, 1 << 20).tobytes()
    want, want_idx = oracle.compress_blocks(src, 1, oracle.dfGzip, 32768, fname_len=0)
    got, idx = eng.compress_blocks(src, 1, oracle.dfGzip, 32768)
    assert got == want and idx == want_idx
    assert len(idx) >= 104 * 32 + 1
    assert eng.uncompress_indexed(got, idx, oracle.dfGzip) == src
    assert zlib.decompress(got, 31) == src


def test_gpu_block_parallel_levels_and_bad_index(eng, golds):
    src = golds["alice29.txt"][:150000] + synth.gen_batch("rand", 1, 70000)[0].tobytes() + golds["html"]
    pc.check_blocks(eng, src, levels=(1, -1, 0, -2, 9), block_sizes=(32768, 131072, 4194304),
                    formats=(oracle.dfGzip,))
    pc.check_blocks(eng, src, levels=(1, 6), block_sizes=(65536,), formats=(oracle.dfZlib, oracle.dfDeflate))
    pc.check_blocks(eng, b"", levels=(1,), block_sizes=(32768,))
    pc.check_blocks_bad_index(eng, src)


def test_gpu_zip_archives(eng, golds, inflate_mode):
    """the reference's archive fixtures through the batch clients, and
    createZipArchive of a few hundred entries in one compress batch."""
    assert pc.check_zip_extract(eng, pc.zip_fixture("cat.jpg")) == 3
    assert pc.check_zip_extract(eng, pc.zip_fixture("Bagnon-10.2.31.zip")) > 100
    pc.check_zip_create(eng, [("README.txt", b"Hello, World!")])
    entries = {}
    rng = random.Random(7)
    names = sorted(golds)
    for i in range(300):
        src = golds[names[i % len(names)]]
        lo = rng.randrange(0, max(1, len(src) - 1))
        entries["d%d/f%03d.bin" % (i % 7, i)] = src[lo:lo + rng.randrange(0, 200000)]
    entries["whole/kppkn.gtb"] = golds["kppkn.gtb"]
    pc.check_zip_create(eng, entries)
    pc.check_zip_errors(eng)


def test_gpu_tarballs(eng, inflate_mode):
    """the reference's tarball fixture (one foreign 3.9 MB gzip member, 20 MB of
    tar) decoded on the device, header walk equal to the oracle's."""
    assert pc.check_tarball(eng, pc.tar_fixture()) > 1000
    assert pc.check_tarball(eng, pc.make_tar_gz([("a.txt", b"hello"), ("d", None)], "d/" + "y" * 140)) == 4
    pc.check_tar_errors(eng)


def test_gpu_property_roundtrip_kinds(eng):
    for kind in ("runs", "rand", "zero", "mix"):
        bufs = [b.tobytes() for b in synth.gen_batch(kind, 32, 200000 + 7)]
        for level in (1, -2, 0):
            pc.check_roundtrip(eng, bufs, level)
        outs, _ = eng.compress_batch(bufs[:4], 1, oracle.dfDeflate)
        for s, o in zip(bufs[:4], outs):
            assert o == oracle.deflate(s, 1)


def test_gpu_stress_runs(eng):
    """tests/stress.nim:10-58 restated with fixed seeds, BestSpeed and default really passed."""
    bufs = []
    for seed in range(64):
        rng = np.random.default_rng(seed)
        data = synth.gen_runs(rng, int(rng.integers(0, 100001)))
        bufs.append(data.tobytes())
        bufs.append(rng.permutation(data).tobytes())
    pc.check_compress_identical(eng, bufs, levels=(1,), formats=(oracle.dfGzip,))
    pc.check_compress_identical(eng, bufs[:24], levels=(-1,), formats=(oracle.dfGzip,))


def test_gpu_cross_encoder_streams(eng, golds):
    """tests/stress2.nim:8-20: zlib-made streams of growing size."""
    base = golds["rfctest3.gold"]
    blobs, want = [], []
    for mult in (1, 2, 5, 17, 40):
        for level in (1, 6, 9):
            blobs.append(zlib.compress(base * mult, level))
            want.append(base * mult)
    outs, sts = eng.uncompress_batch(blobs)
    assert all(s == 0 for s in sts) and outs == want


def test_gpu_damaged_streams_agree_with_oracle(eng, inflate_mode):
    pc.check_errors_match_oracle(eng, pc.mutated_fixtures(400, seed=2024))
    pc.check_error_statuses(eng)


def test_gpu_random_fname_and_checksums(eng, golds):
    pc.check_gzip_random_fname(eng, golds["alice29.txt"])
    pc.check_checksums(eng, [golds["alice29.txt"], golds["urls.10K"], b"", b"a",
                             random.Random(3).randbytes(1 << 20), golds["zerotest3.gold"]])


def test_gpu_device_api_unaligned_offsets(eng, golds):
    """Device-resident API with byte-granular (unaligned) source offsets and slots."""
    import torch
    srcs = [golds["alice29.txt"][:70001], golds["html"][:33333], b"", golds["geo.protodata"][:5]]
    pad = [3, 1, 7, 2]
    blob = bytearray()
    soff = []
    for s, p in zip(srcs, pad):
        blob += b"\xaa" * p
        soff.append(len(blob))
        blob += s
    d_src = torch.frombuffer(bytes(blob) + b"\x00" * 64, dtype=torch.uint8).cuda()
    caps = [eng.compress_bound(len(s)) for s in srcs]
    doff, total = [], 5
    for c in caps:
        doff.append(total)
        total += c + 3
    d_dst = torch.empty(total + 64, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    plan = eng.plan_compress(soff, [len(s) for s in srcs], doff, caps, 1, oracle.dfGzip)
    plan.run(d_src.data_ptr(), d_dst.data_ptr())
    lens, sts = plan.results()
    assert sts == [0] * 4
    host = d_dst.cpu().numpy().tobytes()
    for s, o, n in zip(srcs, doff, lens):
        assert host[o:o + n] == oracle.compress(s, 1, oracle.dfGzip, fname_len=0)
    # and back, straight from the compressed slots with device-side lengths
    ocap = [len(s) for s in srcs]
    ooff, t2 = [], 1
    for c in ocap:
        ooff.append(t2)
        t2 += c + 5
    d_out = torch.zeros(t2 + 64, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    up = eng.plan_uncompress(doff, caps, ooff, ocap, oracle.dfGzip)
    up.set_src_lens_device(plan.device_lens())
    up.run(d_dst.data_ptr(), d_out.data_ptr())
    lens2, sts2 = up.results()
    assert sts2 == [0] * 4 and lens2 == ocap
    host2 = d_out.cpu().numpy().tobytes()
    for s, o in zip(srcs, ooff):
        assert host2[o:o + len(s)] == s
    # slot too small -> per-buffer status, neighbours unaffected, nothing written
    small = eng.plan_compress(soff, [len(s) for s in srcs], doff, [caps[0], 100, caps[2], caps[3]],
                              1, oracle.dfGzip)
    small.run(d_src.data_ptr(), d_dst.data_ptr())
    _, sts3 = small.results()
    assert sts3 == [0, 21, 0, 0]


def test_gpu_plan_slots_with_gaps(eng):
    import torch
    def upload(b):
        t = torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()
        return t.data_ptr(), t
    def alloc(n, fill):
        t = torch.full((n,), fill, dtype=torch.uint8, device="cuda")
        return t.data_ptr(), t
    pc.check_plan_slots_with_gaps(eng, upload, lambda t: t.cpu().numpy().tobytes(), alloc)


def test_gpu_plan_reruns_cheapest_last(eng):
    import torch
    def upload(b):
        t = torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()
        return t.data_ptr(), t
    def alloc(n, fill):
        t = torch.full((n,), fill, dtype=torch.uint8, device="cuda")
        return t.data_ptr(), t
    pc.check_plan_reruns(eng, upload, lambda t: t.cpu().numpy().tobytes(), alloc, 200, 1 << 20)  # (6400 fragments, 5120 waves)


def test_gpu_plan_pack(eng):
    import torch
    def upload(b):
        t = torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()
        return t.data_ptr(), t
    def alloc(n, fill):
        t = torch.full((n,), fill, dtype=torch.uint8, device="cuda")
        return t.data_ptr(), t
    pc.check_plan_pack(eng, upload, lambda t: t.cpu().numpy().tobytes(), alloc)


def test_gpu_stored_chains(eng, inflate_mode):
    pc.check_stored_chains(eng, 200)


def test_gpu_stored_chain_segmented(eng, monkeypatch):
    pc.check_stored_chain_segmented(eng, monkeypatch, 300)


def test_gpu_split_inflate_edges(eng, inflate_mode):
    pc.check_split_inflate_edges(eng)


def test_gpu_segmented_streams(eng, monkeypatch):
    """Large streams on many workgroups (zh_inflate_seg.hip), small segments and the default ones."""
    pc.check_segmented(eng, 1024, monkeypatch, 2048)
    pc.check_segmented(eng, 32 * 1024, monkeypatch, 65536)
    # blocks far longer than a segment: this library's own six-block stream (4 MiB of input a block)
    monkeypatch.delenv("ZH_SEG_MIN")
    monkeypatch.delenv("ZH_SEG_BYTES")
    big = synth.gen_batch("mix", 1, 23 << 20, first_index=11)[0].tobytes()
    comp, sts = eng.compress_batch([big], 1, oracle.dfGzip)
    assert sts == [0]
    outs, sts = eng.uncompress_batch(comp, oracle.dfGzip)
    assert sts == [0] and outs[0] == big


def test_gpu_one_gib_stream_decodes_segment_wise():
    """ONE stream of 1 GiB (this library's own, BestSpeed: blocks of 4 MiB of input, each many segments long): the
    round trip, the gzip trailer -- and that it is DECODED by many workgroups.  This very stream holds bits inside a
    block's payload that pass for a block header; they once cost the segments behind them their sub-starts, the
    decoder before them its token room and the stream the segment-wise decode: 2 s instead of 20 ms, with the right
    bytes.  (tools/gpu_big_buffer.py: also 4 GiB + 12345 bytes against the oracle, byte for byte.)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gpu_big_buffer
    res = gpu_big_buffer.run(1024, level=1, with_oracle=False, with_zlib=False)
    assert res["trailer_ok"] and res["decoded_segment_wise"], res
    assert max(res["uncompress_s"][1:]) < 0.5, res  # (20 ms on an MI355X; one workgroup takes 2 s)
    # 1 GiB of random bytes is 16 K stored blocks: chained stored blocks are segment starts too (zh_seg_find_kernel), so
    # the chain is read and copied by many workgroups (17 ms; one workgroup took 74 ms 64 headers and four blocks' bytes
    # at a time, 193 ms with a header and a block a round); literals only (level -2)
    rand = gpu_big_buffer.run(1024, level=1, with_oracle=False, with_zlib=False, kind="rand")
    assert rand["trailer_ok"] and rand["decoded_segment_wise"] and min(rand["uncompress_s"]) < 0.05, rand
    lits = gpu_big_buffer.run(1024, level=-2, with_oracle=False, with_zlib=False)
    assert lits["trailer_ok"] and min(lits["uncompress_s"]) < 0.2, lits
