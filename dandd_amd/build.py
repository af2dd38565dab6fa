"""Build libdandd_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m dandd_amd.build [--force] [--save-temps]

The library is a plain C-ABI shared object (include/dandd_hip.h); it is loaded with ctypes
by dandd_amd.engine and never through a torch extension.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libdandd_hip.so")
SOURCES = ["dd_api.hip", "dd_plan.hip", "dd_pack.hip", "dd_sweep.hip", "dd_union.hip", "dd_gram.hip", "dd_pscan.hip", "dd_synth.hip", "dd_exact.hip", "dd_ginflate.hip"]
HEADERS = ["dd_common.h", "dd_io.h", "dd_inflate.h", "dd_kernels.h", "dd_plan.h", os.path.join("..", "..", "include", "dandd_hip.h")]
FLAGS = [
    "-O3",
    "--offload-arch=gfx950",
    "-std=c++17",
    "-fPIC",
    "-shared",
    "-ffp-contract=off",  # Ertl MLE must round like the host/oracle copy
    "-Wall",
    "-Wno-unused-function",
]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, extra=(), verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [hipcc()] + FLAGS + list(extra) + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES] + ["-lz", "-ldl", "-lpthread"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    extra = []
    if "--save-temps" in sys.argv:
        os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
        extra += ["-save-temps=obj"]
    build(force="--force" in sys.argv, extra=extra, verbose=True)
    print(LIB)
