"""CPU-only checks of the drop-in boundary: the hipcc-built C-ABI library loads and
exports every symbol include/zippy_hip.h declares; the host mirror keeps the
reference's names.  No compute calls (there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from zippy_amd import build, _binding
    return _binding.load_library(build.build())


def test_header_symbols_all_exported(lib):
    from zippy_amd import _binding
    hdr = open(os.path.join(ROOT, "include", "zippy_hip.h")).read()
    declared = set(re.findall(r"\b(zh_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_binding.EXPORTED_SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_strerror_matches_reference_messages(lib):
    assert lib.zh_strerror(8) == b"Checksum verification failed"  # gzip.nim:81
    assert lib.zh_strerror(13) == b"Invalid buffer, unable to uncompress"  # internal.nim:191-192
    assert lib.zh_strerror(17) == b"Invalid block header"  # inflate.nim:289
    assert lib.zh_compress_bound(0, 2) >= 5 + 10 + 26 + 8


def test_no_gpu_means_loud_failure(lib):
    """The product path must not fall back to a CPU implementation."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from zippy_amd import api
    from zippy_amd.common import ZippyError
    with pytest.raises(ZippyError):
        api.compress(b"abc")


def test_product_package_never_touches_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "zippy_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in text and "zippy_oracle" not in text, f
                assert "hipemu" not in text or f == "_binding.py", f


def test_reference_api_names():
    import zippy_amd
    from zippy_amd import api
    assert (zippy_amd.dfDetect, zippy_amd.dfZlib, zippy_amd.dfGzip, zippy_amd.dfDeflate) == (0, 1, 2, 3)
    assert (zippy_amd.NoCompression, zippy_amd.BestSpeed, zippy_amd.BestCompression,
            zippy_amd.DefaultCompression, zippy_amd.HuffmanOnly) == (0, 1, 9, -1, -2)
    for name in ("compress", "uncompress", "crc32", "adler32"):
        assert callable(getattr(api, name))


def _c_prototypes():
    """name -> number of parameters, from include/zippy_hip.h"""
    hdr = open(os.path.join(ROOT, "include", "zippy_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(zh_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def test_nim_shim_matches_the_header(lib):
    """bindings/nim/hip.nim -- the reference-side binding of INTEGRATION.md as a file; Nim is not in the image, so
    it cannot be compiled here: every `importc` proc must at least name a symbol the header declares and the library
    exports, and take as many parameters as the C prototype (`a, b: cint` counts two)."""
    src = open(os.path.join(ROOT, "bindings", "nim", "hip.nim")).read()
    protos = _c_prototypes()
    procs = re.findall(r"proc\s+(zh_[a-z0-9_]+)\s*\((.*?)\)\s*(?::\s*[A-Za-z_]+\s*)?\{\.importc", src, flags=re.S)
    assert len(procs) >= 20
    for name, params in procs:
        assert name in protos, name
        assert getattr(lib, name) is not None
        n = 0
        for group in params.split(","):  # "level, dataFormat: cint" -> two names before one type
            if group.strip():
                n += 1
        assert n == protos[name], (name, n, protos[name])
    # the shim text in INTEGRATION.md is this file's (so that the two cannot drift apart)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name, _ in procs:
        assert ("proc %s(" % name) in doc, name
