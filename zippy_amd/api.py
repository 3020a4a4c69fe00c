"""Host-side mirror of the reference interface for the hot path (src/zippy.nim):

    compress(src, level=DefaultCompression, dataFormat=dfGzip) -> bytes     zippy.nim:11-16,86-98
    uncompress(src, dataFormat=dfDetect) -> bytes                          zippy.nim:100-104,167-177
    crc32(src) / adler32(src)                                              crc.nim:53,74 / adler32.nim:6,65

plus the batch forms the engine is built for.  Same names, argument meaning and
error behaviour (ZippyError) as the reference; every call goes through the C ABI
in include/zippy_hip.h into the gfx950 kernels.  No CPU fallback: importing this
module without the built library, or without a usable GPU, raises.
"""
import os

from ._binding import Engine
from .common import (ZippyError, dfDetect, dfZlib, dfGzip, dfDeflate, NoCompression, BestSpeed,
                     BestCompression, DefaultCompression, HuffmanOnly)

# ZIPPY_HIP_LIB: tuning builds of the same library (tools/); never a different implementation
LIB_PATH = os.environ.get("ZIPPY_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                         "libzippy_hip.so")

_engine = None


def engine():
    global _engine
    if _engine is None:
        _engine = Engine(LIB_PATH)
    return _engine


def compress(src, level=DefaultCompression, dataFormat=dfGzip):
    return engine().compress(src, level, dataFormat)


def uncompress(src, dataFormat=dfDetect):
    return engine().uncompress(src, dataFormat)


def compress_batch(bufs, level=DefaultCompression, dataFormat=dfGzip):
    """n independent compress() calls in one launch sequence -> (outputs, statuses)."""
    return engine().compress_batch(bufs, level, dataFormat)


def uncompress_batch(bufs, dataFormat=dfDetect):
    return engine().uncompress_batch(bufs, dataFormat)


def compress_blocks(src, level=DefaultCompression, dataFormat=dfGzip, block_bytes=32768):
    """compress() with deflate blocks of block_bytes instead of deflate.nim:228's 4 MiB, plus the
    index of block starts; the stream still round-trips through zippy's uncompress()."""
    return engine().compress_blocks(src, level, dataFormat, block_bytes)


def uncompress_indexed(src, index, dataFormat=dfDetect):
    """uncompress() of a stream whose block index is known: one decoder per block."""
    return engine().uncompress_indexed(src, index, dataFormat)


def openZipArchive(image):
    """ziparchives.nim:183 openZipArchive on the bytes of an archive -> reader with walk_files(),
    extract_file(path), extract_batch(indices)."""
    return engine().open_zip(image)


def createZipArchive(entries, dos_time=0, dos_date=0):
    """ziparchives.nim:625-634 createZipArchive(OrderedTable): entries = ordered mapping / pairs."""
    return engine().create_zip(entries, dos_time, dos_date)


def openTarball(image):
    """tarballs.nim:26-124 on the bytes of a .tar.gz / .tar -> reader with .entries, .contents(i)."""
    return engine().open_tar(image)


def crc32(src):
    return engine().crc32(src)


def adler32(src):
    return engine().adler32(src)
