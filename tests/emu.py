"""Test helper: the engine built from the SAME kernel sources with g++ against
the fiber emulator in tests/hipemu (no GPU needed).  Test infrastructure only."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "hipemu"))

_engine = None


def engine():
    global _engine
    if _engine is None:
        import build_emu
        from zippy_amd._binding import Engine
        # few table slots, so that the matcher's persistent waves take several fragments each (their
        # tables are reused without clearing) in batches the emulator can afford
        os.environ.setdefault("ZH_L1_SLOTS", "64")
        # small staging chunks and several host threads: the host-buffer calls of every test cross
        # chunk and thread borders inside buffers
        os.environ.setdefault("ZH_PIN_CHUNK", "131072")
        os.environ.setdefault("ZH_HOST_THREADS", "3")
        # and batches of more than one group's worth of input run as pipelined groups
        os.environ.setdefault("ZH_PIPE_MIN", "1")
        os.environ.setdefault("ZH_PIPE_GROUP", "150000")
        # batches of up to three streams decode with the wide (1024-thread) inflate kernels, larger
        # ones with the narrow (256-thread) ones: the tests' batches cover both
        os.environ.setdefault("ZH_INFLATE_WIDE", "3")
        os.environ.setdefault("ZH_INFLATE_MID", "8")  # (... and those of four to eight with the 512-thread ones)
        _engine = Engine(build_emu.build())
        _engine.set_gzip_fname_len(0)
    return _engine
