"""TEST INFRASTRUCTURE ONLY: builds tests/hipemu/libzippy_hip_emu.so -- the very
same zippy_amd/csrc/*.hip sources compiled by g++ against the fiber-based HIP
emulator in this directory, so kernel logic can be exercised without a GPU.
Never loaded by the product package."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "zippy_amd", "csrc")
LIB = os.path.join(HERE, "libzippy_hip_emu.so")
OBJ = os.path.join(HERE, "build")


def build(force=False):
    import sys
    sys.path.insert(0, ROOT)
    from zippy_amd.build import SOURCES, HEADERS
    # ZH_EMU_VARIANT=name ZH_EMU_DEFINES="-DZH_X=1 ...": a build of its own with other compile-time switches
    global LIB, OBJ
    var = os.environ.get("ZH_EMU_VARIANT")
    defines = os.environ.get("ZH_EMU_DEFINES", "").split()
    if var:
        LIB = os.path.join(HERE, "libzippy_hip_emu_%s.so" % var)
        OBJ = os.path.join(HERE, "build_" + var)
    os.makedirs(OBJ, exist_ok=True)
    deps = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(HERE, "hip", "hip_runtime.h")]
    objs = []
    procs = []
    for src in SOURCES + ["emu.cpp"]:
        s = os.path.join(HERE if src == "emu.cpp" else CSRC, src)
        o = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(
                os.path.getmtime(p) for p in [s] + deps):
            cmd = ["g++", "-O1", "-g", "-std=c++17", "-x", "c++", "-fPIC", "-I", HERE,
                   "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unknown-pragmas",
                   "-Wno-attributes"] + defines + ["-c", s, "-o", o]
            procs.append((cmd, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
    for cmd, p in procs:
        _, err = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("emu build failed: %s\n%s" % (" ".join(cmd), err[-8000:]))
    if procs or not os.path.exists(LIB):
        subprocess.check_call(["g++", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build())
