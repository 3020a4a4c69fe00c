"""ctypes binding of the CPU oracle (oracle/zippy_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package zippy_amd.
Mirrors the reference's Nim API (src/zippy.nim:11-16,100-104) on the CPU.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# ZIPPY_ORACLE_LIB: another build of the same source (tests/test_oracle_sanitizers.py: ASan/UBSan)
_LIB_PATH = os.environ.get("ZIPPY_ORACLE_LIB") or os.path.join(_HERE, "libzippy_oracle.so")

dfDetect, dfZlib, dfGzip, dfDeflate = 0, 1, 2, 3
NoCompression, BestSpeed, BestCompression, DefaultCompression, HuffmanOnly = 0, 1, 9, -1, -2


class ZippyError(Exception):
    """src/zippy/common.nim:2"""

    def __init__(self, status, msg):
        super().__init__(msg)
        self.status = status


def build(force=False):
    src = os.path.join(_HERE, "zippy_oracle.c")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, os.path.basename(_LIB_PATH)],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _Buf(ctypes.Structure):
    _fields_ = [("data", ctypes.POINTER(ctypes.c_uint8)), ("len", ctypes.c_size_t),
                ("cap", ctypes.c_size_t)]


class _Meta(ctypes.Structure):
    _fields_ = [("litlen_freq", ctypes.c_uint32 * 286), ("distance_freq", ctypes.c_uint32 * 30),
                ("num_literals", ctypes.c_int64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.zo_strerror.restype = ctypes.c_char_p
        L.zo_strerror.argtypes = [ctypes.c_int]
        L.zo_free.argtypes = [ctypes.c_void_p]
        L.zo_compress.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_int, ctypes.POINTER(_Buf)]
        L.zo_compress_blocks.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_size_t, ctypes.POINTER(_Buf),
                                         ctypes.POINTER(ctypes.POINTER(ctypes.c_uint64)),
                                         ctypes.POINTER(ctypes.c_size_t)]
        L.zo_uncompress.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int,
                                    ctypes.POINTER(_Buf)]
        L.zo_deflate.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int,
                                 ctypes.POINTER(_Buf)]
        L.zo_inflate.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t,
                                 ctypes.POINTER(_Buf)]
        L.zo_crc32.restype = ctypes.c_uint32
        L.zo_crc32.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        L.zo_adler32.restype = ctypes.c_uint32
        L.zo_adler32.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        L.zo_encode_block_tokens.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t,
                                             ctypes.c_int,
                                             ctypes.POINTER(ctypes.POINTER(ctypes.c_uint16)),
                                             ctypes.POINTER(ctypes.c_size_t),
                                             ctypes.POINTER(_Meta)]
        L.zo_huffman_codes.argtypes = [ctypes.POINTER(ctypes.c_uint32), ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int, ctypes.POINTER(ctypes.c_uint16),
                                       ctypes.POINTER(ctypes.c_uint8)]
        L.zo_batch_mt.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_size_t),
                                  ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                  ctypes.POINTER(_Buf), ctypes.POINTER(ctypes.c_double)]
        _lib = L
    return _lib


def batch_mt(bufs, direction, level=BestSpeed, dataFormat=dfGzip, threads=1, keep=False):
    """bench.py's cpu_baseline: compress (direction 0) or uncompress (1) every buffer of `bufs` on
    `threads` pinned worker threads inside the C library.  -> (seconds of the parallel region,
    results or None)."""
    n = len(bufs)
    arr = (ctypes.c_char_p * n)(*bufs)
    lens = (ctypes.c_size_t * n)(*[len(b) for b in bufs])
    outs = (_Buf * n)() if keep else None
    sec = ctypes.c_double(0.0)
    st = lib().zo_batch_mt(arr, lens, n, direction, level, dataFormat, threads, outs, ctypes.byref(sec))
    res = None
    if keep:
        res = [ctypes.string_at(o.data, o.len) if o.len else b"" for o in outs]
        for o in outs:
            if o.data:
                lib().zo_free(o.data)
    if st != 0:
        raise ZippyError(st, lib().zo_strerror(st).decode())
    return sec.value, res


def _take(buf, status):
    try:
        if status != 0:
            raise ZippyError(status, lib().zo_strerror(status).decode())
        return ctypes.string_at(buf.data, buf.len) if buf.len else b""
    finally:
        if buf.data:
            lib().zo_free(buf.data)


def compress(src, level=DefaultCompression, dataFormat=dfGzip, fname_len=0):
    """src/zippy.nim:11-84.  fname_len: gzip FNAME letters (reference: random 0..25)."""
    src = bytes(src)
    buf = _Buf()
    st = lib().zo_compress(src, len(src), level, dataFormat, fname_len, ctypes.byref(buf))
    return _take(buf, st)


def compress_blocks(src, level=DefaultCompression, dataFormat=dfGzip, block_bytes=32768, fname_len=0):
    """Block-parallel variant (BASELINE config 5): compress() with deflate blocks of block_bytes,
    plus the index [(bit_off, out_off), ...] of block starts (last entry closes the list)."""
    src = bytes(src)
    buf = _Buf()
    idx = ctypes.POINTER(ctypes.c_uint64)()
    n = ctypes.c_size_t(0)
    st = lib().zo_compress_blocks(src, len(src), level, dataFormat, fname_len, block_bytes,
                                  ctypes.byref(buf), ctypes.byref(idx), ctypes.byref(n))
    data = _take(buf, st)
    entries = [(idx[2 * i], idx[2 * i + 1]) for i in range(n.value)]
    lib().zo_free(idx)
    return data, entries


def uncompress(src, dataFormat=dfDetect):
    """src/zippy.nim:100-165"""
    src = bytes(src)
    buf = _Buf()
    st = lib().zo_uncompress(src, len(src), dataFormat, ctypes.byref(buf))
    return _take(buf, st)


def deflate(src, level):
    src = bytes(src)
    buf = _Buf()
    st = lib().zo_deflate(src, len(src), level, ctypes.byref(buf))
    return _take(buf, st)


def inflate(src, pos=0):
    src = bytes(src)
    buf = _Buf()
    st = lib().zo_inflate(src, len(src), pos, ctypes.byref(buf))
    return _take(buf, st)


def crc32(src):
    src = bytes(src)
    return lib().zo_crc32(src, len(src))


def adler32(src):
    src = bytes(src)
    return lib().zo_adler32(src, len(src))


def block_tokens(src, level, block_start=0, block_len=None):
    """Token stream (list of uint16) + histograms of one <=4 MiB block."""
    import numpy as np
    src = bytes(src)
    if block_len is None:
        block_len = len(src) - block_start
    toks = ctypes.POINTER(ctypes.c_uint16)()
    n = ctypes.c_size_t()
    meta = _Meta()
    st = lib().zo_encode_block_tokens(src, block_start, block_len, level, ctypes.byref(toks),
                                      ctypes.byref(n), ctypes.byref(meta))
    if st != 0:
        raise ZippyError(st, lib().zo_strerror(st).decode())
    try:
        arr = np.ctypeslib.as_array(toks, shape=(n.value,)).copy() if n.value else np.zeros(
            0, np.uint16)
    finally:
        lib().zo_free(toks)
    return arr, np.array(meta.litlen_freq, dtype=np.uint32), np.array(
        meta.distance_freq, dtype=np.uint32), int(meta.num_literals)


def huffman_codes(freq, min_codes, limit):
    """deflate.nim:13-151 -> (codes, lens) as numpy arrays."""
    import numpy as np
    freq = np.ascontiguousarray(freq, dtype=np.uint32)
    n = len(freq)
    codes = (ctypes.c_uint16 * (n + 2))()
    lens = (ctypes.c_uint8 * (n + 2))()
    k = lib().zo_huffman_codes(freq.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), n, min_codes,
                               limit, codes, lens)
    return np.array(codes[:k], dtype=np.uint16), np.array(lens[:k], dtype=np.uint8)
