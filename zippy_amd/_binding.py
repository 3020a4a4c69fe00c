"""ctypes binding of the C ABI in include/zippy_hip.h.

`Engine(lib_path)` wraps one zh_ctx.  The product entry point (zippy_amd.api)
always passes zippy_amd/libzippy_hip.so; the test-suite also points this class at
the g++/emulator build of the same sources (tests/hipemu) to check kernel logic
without a GPU.  There is no fallback between the two.
"""
import ctypes
import os

from .common import ZippyError, dfDetect, dfGzip, DefaultCompression

_c = ctypes
_SIGS = {
    "zh_create": (_c.c_int, [_c.c_int, _c.c_void_p, _c.POINTER(_c.c_void_p)]),
    "zh_destroy": (None, [_c.c_void_p]),
    "zh_strerror": (_c.c_char_p, [_c.c_int]),
    "zh_last_error": (_c.c_char_p, [_c.c_void_p]),
    "zh_stream": (_c.c_void_p, [_c.c_void_p]),
    "zh_set_gzip_fname_len": (None, [_c.c_void_p, _c.c_int]),
    "zh_set_host_pipeline": (None, [_c.c_void_p, _c.c_size_t, _c.c_size_t]),
    "zh_chain_links_parallel": (_c.c_int, [_c.c_void_p]),
    "zh_set_inflate_mode": (None, [_c.c_void_p, _c.c_int]),
    "zh_set_l1_parse": (None, [_c.c_void_p, _c.c_int]),
    "zh_compress_bound": (_c.c_size_t, [_c.c_size_t, _c.c_int]),
    "zh_compress_batch": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_size_t),
                                     _c.c_size_t, _c.c_int, _c.c_int, _c.POINTER(_c.c_void_p),
                                     _c.POINTER(_c.c_size_t), _c.POINTER(_c.c_int32)]),
    "zh_compress_batch_into": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_size_t),
                                          _c.c_size_t, _c.c_int, _c.c_int, _c.POINTER(_c.c_void_p),
                                          _c.POINTER(_c.c_size_t), _c.POINTER(_c.c_size_t),
                                          _c.POINTER(_c.c_int32)]),
    "zh_uncompress_batch_into": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_size_t),
                                            _c.c_size_t, _c.c_int, _c.POINTER(_c.c_void_p),
                                            _c.POINTER(_c.c_size_t), _c.POINTER(_c.c_size_t),
                                            _c.POINTER(_c.c_int32)]),
    "zh_uncompress_batch": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_void_p),
                                       _c.POINTER(_c.c_size_t), _c.c_size_t, _c.c_int,
                                       _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_size_t),
                                       _c.POINTER(_c.c_int32)]),
    "zh_device_count": (_c.c_int, []),
    "zh_compress_batch_multi": (_c.c_int, [_c.POINTER(_c.c_void_p), _c.c_size_t, _c.POINTER(_c.c_void_p),
                                           _c.POINTER(_c.c_size_t), _c.c_size_t, _c.c_int, _c.c_int,
                                           _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_size_t),
                                           _c.POINTER(_c.c_int32)]),
    "zh_uncompress_batch_multi": (_c.c_int, [_c.POINTER(_c.c_void_p), _c.c_size_t, _c.POINTER(_c.c_void_p),
                                             _c.POINTER(_c.c_size_t), _c.c_size_t, _c.c_int,
                                             _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_size_t),
                                             _c.POINTER(_c.c_int32)]),
    "zh_compress": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_int, _c.c_int,
                               _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_size_t)]),
    "zh_uncompress": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_int,
                                 _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_size_t)]),
    "zh_crc32": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.POINTER(_c.c_uint32)]),
    "zh_adler32": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.POINTER(_c.c_uint32)]),
    "zh_free": (None, [_c.c_void_p]),
    "zh_device_malloc": (_c.c_int, [_c.c_void_p, _c.c_size_t, _c.POINTER(_c.c_void_p)]),
    "zh_device_free": (None, [_c.c_void_p, _c.c_void_p]),
    "zh_device_upload": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t]),
    "zh_device_download": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t]),
    "zh_plan_compress": (_c.c_int, [_c.c_void_p, _c.c_size_t, _c.POINTER(_c.c_uint64),
                                    _c.POINTER(_c.c_uint64), _c.POINTER(_c.c_uint64),
                                    _c.POINTER(_c.c_uint64), _c.c_int, _c.c_int,
                                    _c.POINTER(_c.c_void_p)]),
    "zh_plan_uncompress": (_c.c_int, [_c.c_void_p, _c.c_size_t, _c.POINTER(_c.c_uint64),
                                      _c.POINTER(_c.c_uint64), _c.POINTER(_c.c_uint64),
                                      _c.POINTER(_c.c_uint64), _c.c_int, _c.POINTER(_c.c_void_p)]),
    "zh_plan_run": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "zh_plan_results": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_uint64), _c.POINTER(_c.c_int32)]),
    "zh_plan_device_lens": (_c.c_void_p, [_c.c_void_p]),
    "zh_plan_device_statuses": (_c.c_void_p, [_c.c_void_p]),
    "zh_plan_set_src_lens_device": (_c.c_int, [_c.c_void_p, _c.c_void_p]),
    "zh_plan_pack": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_uint64, _c.c_void_p]),
    "zh_plan_unpack": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "zh_plan_destroy": (None, [_c.c_void_p]),
    "zh_plan_set_profiling": (None, [_c.c_void_p, _c.c_int]),
    "zh_plan_kernel_times": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_char_p),
                                        _c.POINTER(_c.c_float), _c.c_int]),
    "zh_compress_blocks": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_int, _c.c_int,
                                      _c.c_size_t, _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_size_t),
                                      _c.POINTER(_c.POINTER(_c.c_uint64)), _c.POINTER(_c.c_size_t)]),
    "zh_uncompress_indexed": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_int,
                                         _c.POINTER(_c.c_uint64), _c.c_size_t,
                                         _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_size_t)]),
    "zh_plan_compress_blocks": (_c.c_int, [_c.c_void_p, _c.c_size_t, _c.POINTER(_c.c_uint64),
                                           _c.POINTER(_c.c_uint64), _c.POINTER(_c.c_uint64),
                                           _c.POINTER(_c.c_uint64), _c.c_int, _c.c_int, _c.c_size_t,
                                           _c.POINTER(_c.c_void_p)]),
    "zh_plan_block_index": (_c.c_int, [_c.c_void_p, _c.c_size_t, _c.POINTER(_c.POINTER(_c.c_uint64)),
                                       _c.POINTER(_c.c_size_t)]),
    "zh_plan_uncompress_indexed": (_c.c_int, [_c.c_void_p, _c.c_uint64, _c.c_uint64, _c.c_uint64,
                                              _c.c_uint64, _c.c_int, _c.POINTER(_c.c_uint64),
                                              _c.c_size_t, _c.POINTER(_c.c_void_p)]),
    "zh_crc32_batch": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_size_t), _c.c_size_t,
                                  _c.POINTER(_c.c_uint32)]),
    "zh_compress_batch_crc32": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_size_t),
                                           _c.c_size_t, _c.c_int, _c.c_int, _c.POINTER(_c.c_void_p),
                                           _c.POINTER(_c.c_size_t), _c.POINTER(_c.c_int32),
                                           _c.POINTER(_c.c_uint32)]),
    "zh_uncompress_batch_sized": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_size_t),
                                             _c.c_size_t, _c.c_int, _c.POINTER(_c.c_uint64),
                                             _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_size_t),
                                             _c.POINTER(_c.c_int32), _c.POINTER(_c.c_uint32)]),
    "zh_plan_request_crc32": (_c.c_int, [_c.c_void_p, _c.c_int]),
    "zh_plan_crc32": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_uint32)]),
    "zh_zip_open": (_c.c_int, [_c.c_void_p, _c.c_size_t, _c.POINTER(_c.c_void_p)]),
    "zh_zip_close": (None, [_c.c_void_p]),
    "zh_zip_num_entries": (_c.c_size_t, [_c.c_void_p]),
    "zh_zip_entry_at": (_c.c_int, [_c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "zh_zip_find": (_c.c_int, [_c.c_void_p, _c.c_char_p, _c.c_size_t, _c.POINTER(_c.c_size_t)]),
    "zh_zip_extract_batch": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.POINTER(_c.c_size_t), _c.c_size_t,
                                        _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_size_t),
                                        _c.POINTER(_c.c_int32)]),
    "zh_zip_create": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_char_p), _c.POINTER(_c.c_size_t),
                                 _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_size_t), _c.c_size_t,
                                 _c.c_uint16, _c.c_uint16, _c.POINTER(_c.c_void_p),
                                 _c.POINTER(_c.c_size_t)]),
    "zh_tar_open": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.POINTER(_c.c_void_p)]),
    "zh_tar_close": (None, [_c.c_void_p]),
    "zh_tar_num_entries": (_c.c_size_t, [_c.c_void_p]),
    "zh_tar_entry_at": (_c.c_int, [_c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "zh_tar_data": (_c.c_void_p, [_c.c_void_p, _c.POINTER(_c.c_size_t)]),
    "zh_debug_tokens": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_int,
                                   _c.POINTER(_c.POINTER(_c.c_uint16)), _c.POINTER(_c.c_size_t)]),
    "zh_debug_huffman": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_uint32), _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                    _c.POINTER(_c.c_uint16), _c.POINTER(_c.c_uint8), _c.POINTER(_c.c_int)]),
    "zh_debug_segment_stats": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_uint64), _c.POINTER(_c.c_uint64)]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)  # every entry point include/zippy_hip.h declares


def load_library(path):
    if not os.path.exists(path):
        raise ImportError(
            "%s is missing: build it with `python -m zippy_amd.build` (hipcc, gfx950). "
            "zippy_amd has no CPU fallback." % path)
    lib = ctypes.CDLL(path)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


def _u64(seq):
    arr = (_c.c_uint64 * len(seq))(*[int(x) for x in seq])
    return arr


class Plan:
    """A zh_plan: device-resident batch geometry (include/zippy_hip.h)."""

    def __init__(self, engine, handle, n):
        self.engine, self._h, self.n = engine, handle, n

    def run(self, d_src, d_dst):
        self.engine._check(self.engine.lib.zh_plan_run(self._h, d_src, d_dst))

    def results(self):
        lens = (_c.c_uint64 * self.n)()
        sts = (_c.c_int32 * self.n)()
        self.engine._check(self.engine.lib.zh_plan_results(self._h, lens, sts))
        return list(lens), list(sts)

    def block_index(self, buf=0):
        """[(bit_off, out_off), ...] of buffer `buf` of a compress plan (zh_plan_block_index)."""
        idx = _c.POINTER(_c.c_uint64)()
        n = _c.c_size_t()
        self.engine._check(self.engine.lib.zh_plan_block_index(self._h, buf, _c.byref(idx), _c.byref(n)))
        try:
            return [(idx[2 * i], idx[2 * i + 1]) for i in range(n.value)]
        finally:
            self.engine.lib.zh_free(idx)

    def device_lens(self):
        return self.engine.lib.zh_plan_device_lens(self._h)

    def set_src_lens_device(self, d_lens):
        self.engine._check(self.engine.lib.zh_plan_set_src_lens_device(self._h, d_lens))

    def pack(self, d_slots, d_packed, packed_cap, d_offsets):
        """The plan's results back to back at d_packed, n + 1 device uint64 offsets at d_offsets (zh_plan_pack)."""
        self.engine._check(self.engine.lib.zh_plan_pack(self._h, d_slots, d_packed, packed_cap, d_offsets))

    def unpack(self, d_packed, d_offsets, d_slots):
        """Streams back to back -> an uncompress plan's source slots and device-side lengths (zh_plan_unpack)."""
        self.engine._check(self.engine.lib.zh_plan_unpack(self._h, d_packed, d_offsets, d_slots))

    def set_profiling(self, on=True):
        self.engine.lib.zh_plan_set_profiling(self._h, 1 if on else 0)

    def kernel_times(self):
        names = (_c.c_char_p * 32)()
        ms = (_c.c_float * 32)()
        k = self.engine.lib.zh_plan_kernel_times(self._h, names, ms, 32)
        return [(names[i].decode(), ms[i]) for i in range(k)]

    def close(self):
        if self._h:
            self.engine.lib.zh_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ZipEntry(_c.Structure):
    _fields_ = [("path", _c.c_void_p), ("path_len", _c.c_size_t), ("is_directory", _c.c_int),
                ("header_offset", _c.c_uint64), ("compressed_size", _c.c_uint64),
                ("uncompressed_size", _c.c_uint64), ("crc32", _c.c_uint32), ("unix_mode", _c.c_uint32)]


class ZipReader:
    """zh_zip_reader over a bytes image: the ZipArchiveReader of ziparchives.nim:27-29."""

    def __init__(self, engine, image):
        self.engine = engine
        self._image = bytes(image)  # borrowed by the library until close
        h = _c.c_void_p()
        engine._check(engine.lib.zh_zip_open(self._image, len(self._image), _c.byref(h)))
        self._h = h
        self.entries = []
        for i in range(engine.lib.zh_zip_num_entries(h)):
            e = ZipEntry()
            engine._check(engine.lib.zh_zip_entry_at(h, i, _c.byref(e)))
            self.entries.append({
                "path": _c.string_at(e.path, e.path_len).decode("utf-8", "surrogateescape"),
                "is_directory": bool(e.is_directory), "header_offset": e.header_offset,
                "compressed_size": e.compressed_size, "uncompressed_size": e.uncompressed_size,
                "crc32": e.crc32, "unix_mode": e.unix_mode})

    def walk_files(self):
        return [e["path"] for e in self.entries if not e["is_directory"]]

    def find(self, path):
        raw = path.encode("utf-8", "surrogateescape")
        idx = _c.c_size_t()
        self.engine._check(self.engine.lib.zh_zip_find(self._h, raw, len(raw), _c.byref(idx)))
        return idx.value

    def extract_batch(self, indices):
        """-> (list of bytes | None, statuses) for the records at `indices`, one GPU batch."""
        n = len(indices)
        idx = (_c.c_size_t * n)(*indices)
        dsts = (_c.c_void_p * n)()
        lens = (_c.c_size_t * n)()
        sts = (_c.c_int32 * n)()
        rc = self.engine.lib.zh_zip_extract_batch(self.engine._h, self._h, idx, n, dsts, lens, sts)
        outs = []
        try:
            for i in range(n):
                outs.append(_c.string_at(dsts[i], lens[i]) if dsts[i] and sts[i] == 0 else None)
        finally:
            for i in range(n):
                if dsts[i]:
                    self.engine.lib.zh_free(dsts[i])
        self.engine._check(rc)
        return outs, list(sts)

    def extract_file(self, path):
        outs, sts = self.extract_batch([self.find(path)])
        self.engine._check(sts[0])
        return outs[0]

    def close(self):
        if self._h:
            self.engine.lib.zh_zip_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TarEntry(_c.Structure):
    _fields_ = [("path", _c.c_void_p), ("path_len", _c.c_size_t), ("linkname", _c.c_void_p),
                ("linkname_len", _c.c_size_t), ("typeflag", _c.c_char), ("mode", _c.c_uint32),
                ("mtime", _c.c_int64), ("offset", _c.c_uint64), ("size", _c.c_uint64)]


class TarReader:
    """zh_tar_reader: the entries of a .tar.gz / .tar image (tarballs.nim:61-124)."""

    def __init__(self, engine, image):
        self.engine = engine
        self._image = bytes(image)  # an uncompressed tarball stays borrowed until close
        h = _c.c_void_p()
        engine._check(engine.lib.zh_tar_open(engine._h, self._image, len(self._image), _c.byref(h)))
        self._h = h
        n = _c.c_size_t()
        base = engine.lib.zh_tar_data(h, _c.byref(n))
        self.data = _c.string_at(base, n.value) if n.value else b""
        self.entries = []
        for i in range(engine.lib.zh_tar_num_entries(h)):
            e = TarEntry()
            engine._check(engine.lib.zh_tar_entry_at(h, i, _c.byref(e)))
            self.entries.append({
                "path": _c.string_at(e.path, e.path_len), "linkname": _c.string_at(e.linkname, e.linkname_len),
                "typeflag": e.typeflag, "mode": e.mode, "mtime": e.mtime, "offset": e.offset, "size": e.size})

    def contents(self, i):
        e = self.entries[i]
        return self.data[e["offset"]:e["offset"] + e["size"]]

    def close(self):
        if self._h:
            self.engine.lib.zh_tar_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    def __init__(self, lib_path, device=-1, stream=None):
        self.lib = load_library(lib_path)
        h = _c.c_void_p()
        st = self.lib.zh_create(device, stream, _c.byref(h))
        if st != 0:
            raise ZippyError(st, "zh_create: %s (no usable MI355X? zippy_amd has no CPU fallback)" %
                             self.lib.zh_strerror(st).decode())
        self._h = h

    def close(self):
        if self._h:
            self.lib.zh_destroy(self._h)
            self._h = None

    def _check(self, st):
        if st != 0:
            msg = self.lib.zh_strerror(st).decode()
            if st == 20:
                msg += ": " + self.lib.zh_last_error(self._h).decode()
            raise ZippyError(st, msg)

    def set_gzip_fname_len(self, k):
        self.lib.zh_set_gzip_fname_len(self._h, k)

    def set_inflate_mode(self, mode):
        """0: parallel token decode + writer (default), 1: serial two-wave decoder, -1: default."""
        self.lib.zh_set_inflate_mode(self._h, mode)

    def set_l1_parse(self, mode):
        """BestSpeed match finder: 0 the reference's parse (byte-identical streams, default),
        1 the parallel parse (valid streams of about the same size), -1: default / ZH_L1_PARSE."""
        self.lib.zh_set_l1_parse(self._h, mode)

    def chain_links_parallel(self):
        """True: the device passed zh_create's probe and the chain levels' links are built by the class-sorted
        kernels; False: by the in-order ones (a failed probe, or ZH_CHAIN_PREV=serial)."""
        return bool(self.lib.zh_chain_links_parallel(self._h))

    def set_host_pipeline(self, min_batch_bytes=0, group_bytes=0):
        self.lib.zh_set_host_pipeline(self._h, min_batch_bytes, group_bytes)

    def compress_bound(self, n, data_format=dfGzip):
        return self.lib.zh_compress_bound(n, data_format)

    # ---- host-buffer batch API ----
    def _batch(self, fn, bufs, *mid):
        n = len(bufs)
        keep = [bytes(b) for b in bufs]
        srcs = (_c.c_void_p * n)(*[_c.cast(_c.c_char_p(k), _c.c_void_p) for k in keep])
        lens = (_c.c_size_t * n)(*[len(k) for k in keep])
        dsts = (_c.c_void_p * n)()
        dlens = (_c.c_size_t * n)()
        sts = (_c.c_int32 * n)()
        rc = fn(self._h, srcs, lens, n, *mid, dsts, dlens, sts)
        outs = []
        try:
            for i in range(n):
                outs.append(_c.string_at(dsts[i], dlens[i]) if dsts[i] and sts[i] == 0 else None)
        finally:
            for i in range(n):
                if dsts[i]:
                    self.lib.zh_free(dsts[i])
        self._check(rc)
        return outs, list(sts)

    def compress_batch(self, bufs, level=DefaultCompression, data_format=dfGzip):
        """-> (list of bytes | None, list of statuses)"""
        return self._batch(self.lib.zh_compress_batch, bufs, level, data_format)

    def uncompress_batch(self, bufs, data_format=dfDetect):
        return self._batch(self.lib.zh_uncompress_batch, bufs, data_format)

    def _batch_into(self, fn, bufs, outs, *mid):
        """`outs`: writable buffers (bytearray / ctypes arrays) the library fills.
        -> (lengths, statuses): lengths[i] is the result's size, also when it did not fit."""
        n = len(bufs)
        keep = [bytes(b) if not isinstance(b, bytes) else b for b in bufs]
        srcs = (_c.c_void_p * n)(*[_c.cast(_c.c_char_p(k), _c.c_void_p) for k in keep])
        lens = (_c.c_size_t * n)(*[len(k) for k in keep])
        views = [(_c.c_char * len(o)).from_buffer(o) if len(o) else None for o in outs]
        dsts = (_c.c_void_p * n)(*[_c.addressof(v) if v is not None else None for v in views])
        caps = (_c.c_size_t * n)(*[len(o) for o in outs])
        dlens, sts = (_c.c_size_t * n)(), (_c.c_int32 * n)()
        self._check(fn(self._h, srcs, lens, n, *mid, dsts, caps, dlens, sts))
        filled = [dsts[i] is not None or dlens[i] == 0 for i in range(n)]
        return list(dlens), list(sts), filled

    def compress_batch_into(self, bufs, outs, level=DefaultCompression, data_format=dfGzip):
        return self._batch_into(self.lib.zh_compress_batch_into, bufs, outs, level, data_format)

    def uncompress_batch_into(self, bufs, outs, data_format=dfDetect):
        return self._batch_into(self.lib.zh_uncompress_batch_into, bufs, outs, data_format)

    def _raise_first(self, outs, sts):
        for st in sts:
            if st != 0:
                raise ZippyError(st, self.lib.zh_strerror(st).decode())
        return outs

    def compress(self, src, level=DefaultCompression, data_format=dfGzip):
        try:
            outs, sts = self.compress_batch([src], level, data_format)
        except ZippyError:
            raise
        return self._raise_first(outs, sts)[0]

    def uncompress(self, src, data_format=dfDetect):
        outs, sts = self.uncompress_batch([src], data_format)
        return self._raise_first(outs, sts)[0]

    def segment_stats(self):
        """(streams cut into segments, streams whose chain of segments held) since the context was made."""
        cut, held = _c.c_uint64(), _c.c_uint64()
        self._check(self.lib.zh_debug_segment_stats(self._h, _c.byref(cut), _c.byref(held)))
        return cut.value, held.value

    def debug_huffman(self, freq, min_codes, limit, contract=False):
        """One prefix code from a histogram, by the byte-identical builder or by contract mode's
        (zh_debug_huffman) -> (codes, lens) as numpy arrays of numCodes entries."""
        import numpy as np
        f = np.ascontiguousarray(freq, dtype=np.uint32)
        codes = (_c.c_uint16 * (len(f) + 2))()
        lens = (_c.c_uint8 * (len(f) + 2))()
        n = _c.c_int()
        self._check(self.lib.zh_debug_huffman(self._h, f.ctypes.data_as(_c.POINTER(_c.c_uint32)), len(f), min_codes, limit,
                                              1 if contract else 0, codes, lens, _c.byref(n)))
        return np.array(codes[:n.value], dtype=np.uint16), np.array(lens[:n.value], dtype=np.uint8)

    # ---- ZIP archives (ziparchives.nim) ----
    def open_zip(self, image):
        return ZipReader(self, image)

    def create_zip(self, entries, dos_time=0, dos_date=0):
        """entries: ordered (path, contents) pairs -> archive bytes (createZipArchive)."""
        entries = list(entries.items()) if hasattr(entries, "items") else list(entries)
        n = len(entries)
        names = [p.encode("utf-8", "surrogateescape") if isinstance(p, str) else bytes(p) for p, _ in entries]
        blobs = [bytes(c) for _, c in entries]
        c_names = (_c.c_char_p * n)(*names)
        c_nlens = (_c.c_size_t * n)(*[len(x) for x in names])
        c_blobs = (_c.c_void_p * n)(*[_c.cast(_c.c_char_p(b), _c.c_void_p) for b in blobs])
        c_blens = (_c.c_size_t * n)(*[len(b) for b in blobs])
        dst, dlen = _c.c_void_p(), _c.c_size_t()
        self._check(self.lib.zh_zip_create(self._h, c_names, c_nlens, c_blobs, c_blens, n, dos_time, dos_date,
                                           _c.byref(dst), _c.byref(dlen)))
        try:
            return _c.string_at(dst, dlen.value)
        finally:
            self.lib.zh_free(dst)

    def open_tar(self, image):
        return TarReader(self, image)

    def crc32_batch(self, bufs):
        n = len(bufs)
        keep = [bytes(b) for b in bufs]
        srcs = (_c.c_void_p * n)(*[_c.cast(_c.c_char_p(k), _c.c_void_p) for k in keep])
        lens = (_c.c_size_t * n)(*[len(k) for k in keep])
        out = (_c.c_uint32 * n)()
        self._check(self.lib.zh_crc32_batch(self._h, srcs, lens, n, out))
        return list(out)

    # ---- block-parallel form of one large buffer (BASELINE config 5) ----
    def compress_blocks(self, src, level=DefaultCompression, data_format=dfGzip, block_bytes=32768):
        """-> (compressed bytes, [(bit_off, out_off), ...])"""
        src = bytes(src)
        dst, dlen = _c.c_void_p(), _c.c_size_t()
        idx, n = _c.POINTER(_c.c_uint64)(), _c.c_size_t()
        self._check(self.lib.zh_compress_blocks(self._h, src, len(src), level, data_format, block_bytes,
                                                _c.byref(dst), _c.byref(dlen), _c.byref(idx), _c.byref(n)))
        try:
            return _c.string_at(dst, dlen.value), [(idx[2 * i], idx[2 * i + 1]) for i in range(n.value)]
        finally:
            self.lib.zh_free(dst)
            self.lib.zh_free(idx)

    def uncompress_indexed(self, src, index, data_format=dfDetect):
        src = bytes(src)
        flat = _u64([v for e in index for v in e])
        dst, dlen = _c.c_void_p(), _c.c_size_t()
        self._check(self.lib.zh_uncompress_indexed(self._h, src, len(src), data_format, flat, len(index),
                                                   _c.byref(dst), _c.byref(dlen)))
        try:
            return _c.string_at(dst, dlen.value)
        finally:
            self.lib.zh_free(dst)

    def crc32(self, src):
        src = bytes(src)
        out = _c.c_uint32()
        self._check(self.lib.zh_crc32(self._h, src, len(src), _c.byref(out)))
        return out.value

    def adler32(self, src):
        src = bytes(src)
        out = _c.c_uint32()
        self._check(self.lib.zh_adler32(self._h, src, len(src), _c.byref(out)))
        return out.value

    # ---- device-resident API ----
    def plan_compress(self, src_off, src_len, dst_off, dst_cap, level, data_format):
        h = _c.c_void_p()
        n = len(src_off)
        self._check(self.lib.zh_plan_compress(self._h, n, _u64(src_off), _u64(src_len),
                                              _u64(dst_off), _u64(dst_cap), level, data_format,
                                              _c.byref(h)))
        return Plan(self, h, n)

    def plan_uncompress(self, src_off, src_len, dst_off, dst_cap, data_format=dfDetect):
        h = _c.c_void_p()
        n = len(src_off)
        self._check(self.lib.zh_plan_uncompress(self._h, n, _u64(src_off), _u64(src_len),
                                                _u64(dst_off), _u64(dst_cap), data_format,
                                                _c.byref(h)))
        return Plan(self, h, n)

    def plan_compress_blocks(self, src_off, src_len, dst_off, dst_cap, level, data_format, block_bytes):
        h = _c.c_void_p()
        n = len(src_off)
        self._check(self.lib.zh_plan_compress_blocks(self._h, n, _u64(src_off), _u64(src_len),
                                                     _u64(dst_off), _u64(dst_cap), level, data_format,
                                                     block_bytes, _c.byref(h)))
        return Plan(self, h, n)

    def plan_uncompress_indexed(self, src_off, src_len, dst_off, dst_cap, index, data_format=dfDetect):
        h = _c.c_void_p()
        flat = _u64([v for e in index for v in e])
        self._check(self.lib.zh_plan_uncompress_indexed(self._h, src_off, src_len, dst_off, dst_cap,
                                                        data_format, flat, len(index), _c.byref(h)))
        return Plan(self, h, 1)

    def stream(self):
        return self.lib.zh_stream(self._h)

    # ---- parity introspection ----
    def debug_tokens(self, src, level):
        import numpy as np
        src = bytes(src)
        toks = _c.POINTER(_c.c_uint16)()
        n = _c.c_size_t()
        self._check(self.lib.zh_debug_tokens(self._h, src, len(src), level, _c.byref(toks),
                                             _c.byref(n)))
        try:
            return np.ctypeslib.as_array(toks, shape=(n.value,)).copy() if n.value else np.zeros(
                0, np.uint16)
        finally:
            self.lib.zh_free(toks)
