"""The ZIP restatement (oracle/zip_oracle.py) pinned by the reference's own archive fixtures
(tests/test_ziparchives_read.nim) with Python's zipfile as the independent referee.  CPU only."""
import io
import zipfile

import parity_cases as pc
from oracle import zip_oracle


def test_bagnon_archive_matches_zipfile():
    image = pc.zip_fixture("Bagnon-10.2.31.zip")
    ref = zipfile.ZipFile(io.BytesIO(image))
    r = zip_oracle.open_archive(image)
    assert [p.decode() for p in r.records] == ref.namelist()
    n = 0
    for info in ref.infolist():
        rec = r.records[info.filename.encode()]
        assert rec["is_directory"] == info.is_dir()
        assert rec["crc32"] == info.CRC and rec["uncompressed_size"] == info.file_size
        if not info.is_dir():
            assert zip_oracle.extract_file(r, info.filename) == ref.read(info)
            n += 1
    assert n > 100


def test_archive_appended_to_another_file():
    # tests/test_ziparchives_read.nim:46-55: cat.jpg carries a zip with a.txt, b.txt, c.txt
    image = pc.zip_fixture("cat.jpg")
    r = zip_oracle.open_archive(image)
    assert [p.decode() for p, rec in r.records.items() if not rec["is_directory"]] == ["a.txt", "b.txt", "c.txt"]
    ref = zipfile.ZipFile(io.BytesIO(image))
    for name in ("a.txt", "b.txt", "c.txt"):
        assert zip_oracle.extract_file(r, name) == ref.read(name)


def test_created_archive_is_readable_by_zipfile():
    # tests/test_ziparchives_write.nim:4-7 plus a few more shapes
    entries = [("README.txt", b"Hello, World!"), ("dir/data.bin", bytes(range(256)) * 300), ("empty.txt", b""),
               ("café.txt", b"utf-8 name")]
    blob = zip_oracle.create_archive(entries, 0x6000, 0x5A21)
    zf = zipfile.ZipFile(io.BytesIO(blob))
    assert zf.testzip() is None
    assert zf.namelist() == [p for p, _ in reversed(entries)]
    for path, contents in entries:
        assert zf.read(path) == contents
    back = zip_oracle.open_archive(blob)
    for path, contents in entries:
        assert zip_oracle.extract_file(back, path) == contents


def test_cp437_names_are_converted():
    # a name that is not valid UTF-8 and has no language-encoding flag is read as code page 437
    buf = io.BytesIO()
    with zipfile.ZipFile(buf, "w") as zf:
        zf.writestr(zipfile.ZipInfo("x.txt"), b"abc")
    blob = bytearray(buf.getvalue())
    for at in (blob.find(b"x.txt"), blob.rfind(b"x.txt")):
        blob[at] = 0x82  # e-acute in CP437
    r = zip_oracle.open_archive(bytes(blob))
    assert list(r.records) == ["é.txt".encode()]
