#!/usr/bin/env python3
"""Regenerates tests/golden/ from the reference's own test fixtures.

Run in the build container (needs /root/reference, which does NOT exist on the
GPU box -- hence the committed copies).  What is committed:

  *.gz / *.gzip / *.z   the reference's known-answer DECODE fixtures, byte for
                        byte (tests/test.nim:41-60, tests/test_known_bad.nim:3,
                        tests/bench.nim:4-12).  They are data,
                        not source.
  manifest.json         for every fixture: compressed/uncompressed sizes,
                        CRC-32 and SHA-256 of the expected output ("gold").
                        The gold bytes themselves are NOT committed: tests
                        regenerate them with Python's zlib (an independent
                        referee) and check the SHA-256 recorded here, which was
                        taken from the reference's .gold / corpus files.
  oracle_kat            SHA-256 of the oracle's raw-deflate output per
                        (corpus file, level): a regression pin for the oracle
                        restatement and the byte-exactness target for the HIP
                        encoder.
"""
import hashlib
import json
import os
import shutil
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/tests/data"
sys.path.insert(0, os.path.join(HERE, "..", ".."))

GOLD_OF = {  # fixture -> gold file in the reference (tests/test.nim:3-39)
    "fixed.z": "urls.10K", "empty.gz": "empty.gold", "empty.gzip": "empty.gold",
    "known_bad_nitter.json.gz": None,  # only its length (574) is pinned
}


def main():
    import oracle
    manifest = {"fixtures": {}, "oracle_kat": {}}
    for f in sorted(os.listdir(REF)):
        if not f.endswith((".gz", ".gzip", ".z")):
            continue
        comp = open(os.path.join(REF, f), "rb").read()
        shutil.copyfile(os.path.join(REF, f), os.path.join(HERE, f))
        os.chmod(os.path.join(HERE, f), 0o644)
        if f in GOLD_OF:
            gold_name = GOLD_OF[f]
        else:
            base = f.rsplit(".", 1)[0]  # alice29.txt.gz -> alice29.txt ; rfctest1.gz -> rfctest1
            gold_name = base if os.path.exists(os.path.join(REF, base)) else base + ".gold"
        if gold_name:
            gold = open(os.path.join(REF, gold_name), "rb").read()
        else:
            gold = zlib.decompress(comp, 47)
            assert len(gold) == 574  # tests/test_known_bad.nim:3
        assert zlib.decompress(comp, 47) == gold, f
        manifest["fixtures"][f] = {
            "gold": gold_name, "compressed_len": len(comp), "len": len(gold),
            "crc32": "%08x" % zlib.crc32(gold), "sha256": hashlib.sha256(gold).hexdigest(),
        }
    corpus = ["alice29.txt", "asyoulik.txt", "fireworks.jpg", "geo.protodata", "html", "html_x_4",
              "kppkn.gtb", "lcet10.txt", "paper-100k.pdf", "plrabn12.txt", "urls.10K",
              "rfctest1.gold", "zerotest3.gold", "randtest3.gold"]
    for name in corpus:
        src = open(os.path.join(REF, name), "rb").read()
        manifest["oracle_kat"][name] = {}
        for level in range(-2, 10):
            body = oracle.deflate(src, level)
            assert zlib.decompress(body, -15) == src
            manifest["oracle_kat"][name][str(level)] = {
                "len": len(body), "sha256": hashlib.sha256(body).hexdigest()}
    with open(os.path.join(HERE, "manifest.json"), "w") as fh:
        json.dump(manifest, fh, indent=1, sort_keys=True)
    print("fixtures:", len(manifest["fixtures"]), "kat files:", len(manifest["oracle_kat"]))


if __name__ == "__main__":
    main()
