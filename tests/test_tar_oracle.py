"""The tar restatement (oracle/tar_oracle.py) pinned by the reference's tarball fixture
(tests/test_tarballs_read.nim) with Python's tarfile as the independent referee.  CPU only."""
import io
import tarfile

import parity_cases as pc
from oracle import tar_oracle


def _compare(image):
    data, entries = tar_oracle.open_tarball(image)
    ref = tarfile.open(fileobj=io.BytesIO(image), mode="r:*")
    members = [m for m in ref.getmembers() if m.isfile() or m.isdir() or m.issym()]
    # a GNU 'L' block holds "name\0" and the reference keeps all of it (tarballs.nim:114-115); the
    # NUL ends the C string the OS sees, so compare up to it
    assert [e["path"].split(b"\0")[0].decode().rstrip("/") for e in entries] == [m.name.rstrip("/") for m in members]
    assert len(entries) == len(members)
    for e, m in zip(entries, members):
        assert e["mode"] == m.mode and e["mtime"] == int(m.mtime)
        if m.isfile():
            assert data[e["offset"]:e["offset"] + e["size"]] == ref.extractfile(m).read()
        if m.issym():
            assert e["linkname"].decode() == m.linkname
    return len(entries)


def test_libressl_tarball_matches_tarfile():
    assert _compare(pc.tar_fixture()) > 1000


def test_small_tarballs_with_long_names_dirs_and_links():
    long_name = "d/" + "x" * 150 + ".txt"
    image = pc.make_tar_gz([("a.txt", b"hello"), ("d", None), ("d/b.bin", bytes(range(256)) * 9)], long_name)
    assert _compare(image) == 5
    data, entries = tar_oracle.open_tarball(image)
    assert entries[3]["path"] == long_name.encode() + b"\0"
