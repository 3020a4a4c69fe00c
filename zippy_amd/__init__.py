"""zippy_amd: MI355X-native batched DEFLATE engine with zippy's API surface.

The host-side mirror of the reference interface lives in zippy_amd.api
(compress / uncompress / crc32 / adler32 with the reference's argument meaning
and error behaviour, src/zippy.nim:11-16,100-104); the compute path is the
C-ABI library declared in include/zippy_hip.h (hand-written gfx950 kernels in
zippy_amd/csrc).  There is no CPU fallback: importing zippy_amd.api without the
built library, or calling it without a GPU, raises.
"""
from .common import (ZippyError, dfDetect, dfZlib, dfGzip, dfDeflate, NoCompression, BestSpeed,
                     BestCompression, DefaultCompression, HuffmanOnly)

__all__ = ["ZippyError", "dfDetect", "dfZlib", "dfGzip", "dfDeflate", "NoCompression",
           "BestSpeed", "BestCompression", "DefaultCompression", "HuffmanOnly"]
