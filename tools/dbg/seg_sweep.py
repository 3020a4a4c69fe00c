#!/usr/bin/env python3
"""The segment-wise decode's parity cases (tests/parity_cases.py:check_segmented) at many segment lengths,
on the GPU library or, with --emu, under the emulator:  python tools/dbg/seg_sweep.py [--emu] 300 777 2048 ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


class Env:
    def setenv(self, k, v):
        os.environ[k] = v

    def delenv(self, k):
        os.environ.pop(k, None)


def main():
    args = [a for a in sys.argv[1:] if a != "--emu"]
    import parity_cases as pc
    if "--emu" in sys.argv:
        import emu
        eng, scale = emu.engine(), 1024
    else:
        from zippy_amd import api
        eng, scale = api.engine(), 16 * 1024
    for sb in [int(x) for x in args]:
        pc.check_segmented(eng, scale, Env(), sb)
        print("segment bytes", sb, "ok", flush=True)


if __name__ == "__main__":
    main()
