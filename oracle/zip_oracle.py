"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's ZIP layer
(src/zippy/ziparchives.nim) on top of the codec oracle (oracle/zippy_oracle.c).

    open_archive(image)          openZipArchive   ziparchives.nim:183-372 (memory image instead of a memfile)
    extract_file(reader, path)   extractFile      ziparchives.nim:39-93
    create_archive(entries, ..)  createZipArchive ziparchives.nim:455-634 (OrderedTable form)

Record handling is struct packing, so it is written in Python; every codec call
(compress / uncompress / crc32) goes to the C oracle.  Pinned by the reference's own
archive fixtures (tests/test_ziparchives_read.nim: Bagnon-10.2.31.zip, cat.jpg ->
tests/golden/ziparchives/) with Python's zipfile as the independent referee.
"""
import struct

from . import ZippyError, compress, crc32, uncompress, dfDeflate, BestSpeed

FILE_SIG, CENTRAL_SIG, EOCD_SIG = 0x04034B50, 0x02014B50, 0x06054B50
ZIP64_EOCD_SIG, ZIP64_LOCATOR_SIG, ZIP64_EXTRA = 0x06064B50, 0x07064B50, 1


class ArchiveError(ZippyError):
    def __init__(self, msg):
        ZippyError.__init__(self, -1, msg)


def _u16(b, at):
    return struct.unpack_from("<H", b, at)[0]


def _u32(b, at):
    return struct.unpack_from("<I", b, at)[0]


def _u64(b, at):
    return struct.unpack_from("<Q", b, at)[0]


def _eof():
    raise ArchiveError("Unexpected EOF, invalid archive?")  # internal.nim:197-198


def _validate_utf8(s):
    """Nim std/unicode validateUtf8: -1 if valid, else the index of the first bad byte."""
    i, n = 0, len(s)
    while i < n:
        c = s[i]
        if c <= 127:
            i += 1
        elif c >> 5 == 0b110:
            if c < 0xC2 or not (i + 1 < n and s[i + 1] >> 6 == 2):
                return i
            i += 2
        elif c >> 4 == 0b1110:
            if not (i + 2 < n and s[i + 1] >> 6 == 2 and s[i + 2] >> 6 == 2):
                return i
            i += 3
        elif c >> 3 == 0b11110:
            if not (i + 3 < n and s[i + 1] >> 6 == 2 and s[i + 2] >> 6 == 2 and s[i + 3] >> 6 == 2):
                return i
            i += 4
        else:
            return i
    return -1


def _utf8ify(name):  # ziparchives.nim:108-160 (the table there is code page 437)
    if _validate_utf8(name) == -1:
        return name
    return "".join(chr(c) if c <= 0x7F else bytes([c]).decode("cp437") for c in name).encode("utf-8")


class Reader:
    def __init__(self, image):
        self.image = bytes(image)
        self.records = {}  # utf-8 path (bytes) -> dict, insertion ordered


def open_archive(image):
    r = Reader(image)
    src, size = r.image, len(r.image)
    eocd = size - 22  # :162-173
    while True:
        if eocd < 0:
            _eof()
        if _u32(src, eocd) == EOCD_SIG:
            break
        eocd -= 1
    zip64 = eocd - 20 >= 0 and _u32(src, eocd - 20) == ZIP64_LOCATOR_SIG
    if zip64:  # :208-238
        if _u32(src, eocd - 20 + 4) != 0:
            raise ArchiveError("Unsupported archive, disk number")
        pos = _u64(src, eocd - 20 + 8)
        if _u32(src, eocd - 20 + 16) != 1:
            raise ArchiveError("Unsupported archive, num disks")
        if pos + 64 > size:
            _eof()
        if _u32(src, pos) != ZIP64_EOCD_SIG:
            raise ArchiveError("Invalid central directory file header")
        disk, start_disk = _u32(src, pos + 16), _u32(src, pos + 20)
        on_disk, num, cd_size, cd_start = (_u64(src, pos + 24), _u64(src, pos + 32), _u64(src, pos + 40),
                                           _u64(src, pos + 48))
    else:  # :239-246
        disk, start_disk = _u16(src, eocd + 4), _u16(src, eocd + 6)
        on_disk, num = _u16(src, eocd + 8), _u16(src, eocd + 10)
        cd_size, cd_start = _u32(src, eocd + 12), _u32(src, eocd + 16)
    if disk != 0:
        raise ArchiveError("Unsupported archive, disk number")
    if start_disk != 0:
        raise ArchiveError("Unsupported archive, start disk")
    if on_disk != num:
        raise ArchiveError("Unsupported archive, record number")

    socd = cd_start  # :175-181,257-268
    at, found = eocd, 0
    while at >= 0:
        if _u32(src, at) == CENTRAL_SIG:
            found += 1
            if found == num:
                socd = at
                break
        at -= 1
    off = socd - cd_start
    pos = off + cd_start
    for _ in range(num):  # :275-361
        if pos < 0 or pos + 46 > size:
            _eof()
        if _u32(src, pos) != CENTRAL_SIG:
            raise ArchiveError("Invalid central directory file header")
        flags, method, crc = _u16(src, pos + 8), _u16(src, pos + 10), _u32(src, pos + 16)
        name_len, extra_len, comment_len = _u16(src, pos + 28), _u16(src, pos + 30), _u16(src, pos + 32)
        file_disk, external = _u16(src, pos + 34), _u32(src, pos + 38)
        if method not in (0, 8):
            raise ArchiveError("Unsupported archive, compression method")
        if file_disk != 0:
            raise ArchiveError("Invalid file disk number")
        csize, usize, hoff = _u32(src, pos + 20), _u32(src, pos + 24), _u32(src, pos + 42)
        pos += 46
        if pos + name_len > size:
            _eof()
        raw = src[pos:pos + name_len]
        if raw in r.records:
            raise ArchiveError("Unsupported archive, duplicate entry")
        pos += name_len
        cursor = pos  # :303-341 -- field headers are read at `pos` every time, as the reference does
        while cursor < pos + extra_len:
            if pos + 4 > size:
                _eof()
            fid, flen = _u16(src, pos), _u16(src, pos + 2)
            cursor += 4
            if fid != ZIP64_EXTRA:
                cursor += flen
                continue
            at, fend = cursor, cursor + flen
            if usize == 0xFFFFFFFF:
                if at + 8 > fend or at + 8 > size:
                    _eof()
                usize, at = _u64(src, at), at + 8
            if csize == 0xFFFFFFFF:
                if at + 8 > fend or at + 8 > size:
                    _eof()
                csize, at = _u64(src, at), at + 8
            if hoff == 0xFFFFFFFF:
                if at + 8 > fend or at + 8 > size:
                    _eof()
                hoff, at = _u64(src, at), at + 8
            break
        pos += extra_len + comment_len
        if pos > off + cd_start + cd_size:
            raise ArchiveError("Invalid central directory size")
        path = raw if flags & 0x0800 else _utf8ify(raw)
        is_dir = bool(external & 0x10) or bool(external & (0o040000 << 16)) or path.endswith(b"/")
        r.records[path] = dict(path=path, is_directory=is_dir, header_offset=hoff + off, crc32=crc,
                               compressed_size=csize, uncompressed_size=usize, unix_mode=external >> 16)
    return r


def extract_file(reader, path):  # ziparchives.nim:39-93
    if isinstance(path, str):
        path = path.encode("utf-8", "surrogateescape")
    rec = reader.records.get(path)
    if rec is None:
        raise ArchiveError("No file record found for %r" % path)
    src, size = reader.image, len(reader.image)
    pos = rec["header_offset"]
    if pos + 30 > size:
        _eof()
    if _u32(src, pos) != FILE_SIG:
        raise ArchiveError("Invalid file header")
    method = _u16(src, pos + 8)
    pos += 30 + _u16(src, pos + 26) + _u16(src, pos + 28)
    if pos + rec["compressed_size"] > size:
        _eof()
    if rec["is_directory"]:
        raise ArchiveError("No file record found for %r" % path)
    body = src[pos:pos + rec["compressed_size"]]
    if method == 0:
        out = body
    elif method == 8:
        out = uncompress(body, dfDeflate)
    else:
        raise ArchiveError("Unsupported archive, compression method")
    if crc32(out) != rec["crc32"]:
        raise ArchiveError("Verifying crc32 failed")
    return out


def create_archive(entries, dos_time=0, dos_date=0):  # ziparchives.nim:455-634
    entries = list(entries.items()) if hasattr(entries, "items") else list(entries)
    out = bytearray()
    records = []
    for name, contents in reversed(entries):  # keys.pop() takes the newest key first (:503-505)
        name = name.encode("utf-8", "surrogateescape") if isinstance(name, str) else bytes(name)
        contents = bytes(contents)
        if name == b"":
            raise ArchiveError("Invalid empty file name")
        if name[:1] == b"/":
            raise ArchiveError("File paths must be relative")
        if len(name) > 0xFFFF:
            raise ArchiveError("File name len > uint16.high")
        crc = crc32(contents)
        compressed, method = b"", 0
        if contents:
            compressed, method = compress(contents, BestSpeed, dfDeflate), 8
        records.append((name, len(out), len(contents), len(compressed), method, crc))
        out += struct.pack("<IHHHHHIIIHH", FILE_SIG, 45, 1 << 11, method, dos_time, dos_date, crc,
                           0xFFFFFFFF, 0xFFFFFFFF, len(name), 20)
        out += name
        out += struct.pack("<HHQQ", ZIP64_EXTRA, 16, len(contents), len(compressed))
        out += compressed
    cd_start = len(out)
    for name, hoff, ulen, clen, method, crc in records:
        out += struct.pack("<IHHHHHHIIIHHHHHII", CENTRAL_SIG, 45, 45, 1 << 11, method, dos_time, dos_date, crc,
                           0xFFFFFFFF, 0xFFFFFFFF, len(name), 28, 0, 0, 0, 0, 0xFFFFFFFF)
        out += name
        out += struct.pack("<HHQQQ", ZIP64_EXTRA, 24, ulen, clen, hoff)
    cd_end = len(out)
    out += struct.pack("<IQHHIIQQQQ", ZIP64_EOCD_SIG, 44, 45, 45, 0, 0, len(records), len(records),
                       cd_end - cd_start, cd_start)
    out += struct.pack("<IIQI", ZIP64_LOCATOR_SIG, 0, cd_end, 1)
    out += struct.pack("<IHHHHIIH", EOCD_SIG, 0, 0, 0xFFFF, 0xFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0)
    return bytes(out)
