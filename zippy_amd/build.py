"""Builds zippy_amd/libzippy_hip.so (the C-ABI library, include/zippy_hip.h) with
hipcc for gfx950.  hipcc cross-compiles without a GPU, so this runs in the build
container; the resulting .so travels to the GPU box with the repo snapshot.

    python -m zippy_amd.build            # incremental
    python -m zippy_amd.build --force
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libzippy_hip.so")
SOURCES = ["zh_context.hip", "zh_plan_compress.hip", "zh_plan_uncompress.hip", "zh_plan_run.hip", "zh_plan_pack.hip", "zh_host_batch.hip", "zh_host_calls.hip", "zh_checksum.hip", "zh_inflate.hip", "zh_inflate_split.hip", "zh_inflate_seg.hip", "zh_l1_match.hip", "zh_l1p_match.hip",
           "zh_chain_match.hip", "zh_huffman.hip", "zh_emit.hip", "zh_zip.hip", "zh_tar.hip"]
HEADERS = ["zh_common.h", "zh_host.h", "zh_tables.h", "zh_kprof.h", "zh_inflate_tables.h", os.path.join("..", "..", "include", "zippy_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fvisibility=default",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, kprof=False, variant=None, defines=()):
    """kprof=True builds the tuning variant libzippy_hip_kprof.so (-DZH_KPROF: in-kernel
    phase timers, csrc/zh_kprof.h); the product library never carries them.
    variant="name", defines=["-DZH_X=1", ...]: a measurement build libzippy_hip_<name>.so of the same sources
    with other compile-time switches (A/B runs on the GPU box: ZIPPY_HIP_LIB=<that file> python bench.py ...)."""
    tag = ("_kprof" if kprof else "") + ("_" + variant if variant else "")
    objdir = OBJDIR + tag
    lib = LIB.replace(".so", tag + ".so")
    flags = FLAGS + (["-DZH_KPROF", "-fgpu-rdc"] if kprof else []) + list(defines)
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            jobs.append([HIPCC] + flags + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-6000:]))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _newer(lib, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + (["-fgpu-rdc"] if kprof else []) +
            ["-o", lib] + objs)
    return lib


if __name__ == "__main__":
    # python -m zippy_amd.build [--force] [--kprof] [--variant NAME -DZH_X=1 ...]
    var = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else None
    print(build(force="--force" in sys.argv, verbose="--quiet" not in sys.argv, kprof="--kprof" in sys.argv, variant=var,
                defines=[a for a in sys.argv[1:] if a.startswith("-D")]))
