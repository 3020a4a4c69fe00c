import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu under gpurun)")


@pytest.fixture(scope="session")
def manifest():
    import synth
    return synth.manifest()


@pytest.fixture(scope="session")
def golds(manifest):
    """name -> uncompressed bytes for every corpus/gold file of the reference's
    tests (tests/test.nim:16-39), rebuilt from the committed fixtures."""
    import synth
    out = {}
    for fx, meta in manifest["fixtures"].items():
        if meta["gold"] and fx.endswith(".gz") and meta["gold"] not in out:
            out[meta["gold"]] = synth.corpus_file(meta["gold"])
    out["empty.gold"] = b""
    return out
