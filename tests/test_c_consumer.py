"""include/zippy_hip.h is a C header: a plain C99 program compiles against it with
-pedantic -Werror and drives the library through it -- the emulator build here, the HIP build
on a GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "c_consumer.c")


def _build_and_run(lib_path, tmp_path):
    exe = os.path.join(str(tmp_path), "c_consumer")
    libdir, libname = os.path.split(lib_path)
    assert libname.startswith("lib") and libname.endswith(".so")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1",
                           "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
                           "-L", libdir, "-l" + libname[3:-3], "-Wl,-rpath," + libdir])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "c_consumer ok" in r.stdout


def test_c_consumer_against_emulator(tmp_path):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
    import build_emu
    _build_and_run(build_emu.build(), tmp_path)


@pytest.mark.gpu
def test_c_consumer_against_hip_library(tmp_path):
    from zippy_amd import build
    _build_and_run(build.build(), tmp_path)
