"""TEST INFRASTRUCTURE ONLY -- CPU restatement of src/zippy/tarballs.nim:5-23,40-124 (the parts
that do not touch the file system): gunzip through the C oracle, then the ustar header walk.
Pinned by the reference's fixture (tests/test_tarballs_read.nim: libressl-3.4.2.tar.gz ->
tests/golden/tarballs/) with Python's tarfile as the independent referee."""
from . import ZippyError, uncompress, dfGzip


class TarError(ZippyError):
    def __init__(self, msg):
        ZippyError.__init__(self, -1, msg)


def _octal(field):  # parseTarOctInt, tarballs.nim:5-23
    start = 0
    while start < len(field) and not (48 <= field[start] <= 57):
        start += 1
    n = 0
    while start + n < len(field) and 48 <= field[start + n] <= 57:
        n += 1
    digits = field[start:start + n]
    if any(d > 55 for d in digits):
        raise TarError("invalid octal digit")
    return int(digits, 8) if n else 0


def _cstr(field):
    return field.split(b"\0", 1)[0]


def _join(head, tail):  # std/os `/`
    if not head:
        return tail
    hs, ts = head.endswith(b"/"), tail.startswith(b"/")
    if hs and ts:
        return head + tail[1:]
    if hs or ts:
        return head + tail
    return head + b"/" + tail


def _safe(path):  # internal.nim:294-302
    if path.startswith(b"/") or path.startswith(b"../") or path.startswith(b"..\\"):
        return False
    return b"/../" not in path and b"\\..\\" not in path


def open_tarball(image):
    """-> (uncompressed tarball, [entry dict, ...]) for files, directories and symlinks."""
    image = bytes(image)
    if len(image) < 2:
        raise TarError("Invalid buffer, unable to uncompress")
    data = uncompress(image, dfGzip) if image[0] == 31 and image[1] == 139 else image
    entries, long_name, pos = [], b"", 0
    while pos < len(data):
        if pos + 512 > len(data):
            raise TarError("Unexpected EOF, invalid archive?")
        h = data[pos:pos + 512]
        name, mode, size, mtime = _cstr(h[0:100]), _octal(h[100:107]), _octal(h[124:135]), _octal(h[136:147])
        typeflag, linkname = h[156:157], _cstr(h[157:257])
        prefix = _cstr(h[345:500]) if _cstr(h[257:263]) == b"ustar" else b""
        pos += 512
        if pos + size > len(data):
            raise TarError("Unexpected EOF, invalid archive?")
        if name or long_name:
            if long_name:
                path, long_name = long_name, b""
            else:
                path = _join(prefix, name)
            if not _safe(path):
                raise TarError("Path not allowed " + repr(path))
            if typeflag in (b"0", b"\0", b"5", b"2"):
                entries.append(dict(path=path, linkname=linkname, typeflag=typeflag, mode=mode, mtime=mtime,
                                    offset=pos, size=size))
            elif typeflag == b"L":
                long_name = data[pos:pos + size]
            elif typeflag in (b"g", b"x") or b"A" <= typeflag <= b"Z":
                pass
            else:
                raise TarError("Unsupported header type " + repr(typeflag))
        pos += (size + 511) & ~511
    return data, entries
