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
SOURCES = ["dd_api.hip", "dd_plan.hip", "dd_pack.hip", "dd_sweep.hip", "dd_union.hip", "dd_gram.hip", "dd_pscan.hip", "dd_synth.hip", "dd_exact.hip", "dd_ginflate.hip", "dd_fastq.hip"]
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


OBJDIR = os.path.join(HERE, "build", "obj")
COMPILE_FLAGS = [f for f in FLAGS if f != "-shared"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


# `dashing`: the reference's command-line contract over the C ABI (csrc/dd_cli.c, plain C99, no Python)
BINDIR = os.path.join(HERE, "bin")
CLI = os.path.join(BINDIR, "dashing")
CLI_SRC = os.path.join(CSRC, "dd_cli.c")
# the same program under GNU parallel's name, in a directory of its own (on PATH only for who wants DandD's k-batches fused:
# what is not `parallel [-j N] '<dashing ... {} ...>' ::: k ...` it hands to the next `parallel` on PATH)
FUSED_PARALLEL = os.path.join(BINDIR, "fused", "parallel")


def build_cli(force=False):
    """gcc, C99, linked against the library next to it (rpath $ORIGIN/../lib); needs build() to have run."""
    inc = os.path.join(HERE, "..", "include")
    deps = [CLI_SRC, os.path.join(inc, "dandd_hip.h"), LIB]
    if (not force and os.path.exists(CLI) and os.path.exists(FUSED_PARALLEL)
            and all(os.path.getmtime(d) <= min(os.path.getmtime(CLI), os.path.getmtime(FUSED_PARALLEL)) for d in deps)):
        return CLI
    os.makedirs(os.path.dirname(FUSED_PARALLEL), exist_ok=True)
    cmd = [os.environ.get("CC", "gcc"), "-std=c99", "-O2", "-Wall", "-Wextra", "-pedantic", "-I" + inc, CLI_SRC, "-o", CLI + ".tmp",
           "-L" + LIBDIR, "-ldandd_hip", "-lz", "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath,$ORIGIN/../../lib", "-Wl,-rpath,/opt/rocm/lib",
           "-Wl,-rpath-link," + LIBDIR]
    subprocess.check_call(cmd)
    os.replace(CLI + ".tmp", CLI)
    import shutil
    shutil.copy2(CLI, FUSED_PARALLEL + ".tmp")          # (a copy, not a link: the tree travels as files)
    os.replace(FUSED_PARALLEL + ".tmp", FUSED_PARALLEL)
    return CLI


def _stale(obj, src):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in [src, os.path.abspath(__file__)] + [os.path.join(CSRC, h) for h in HEADERS])


def build(force=False, extra=(), verbose=False, jobs=None):
    """One object per source (compiled side by side, kept under dandd_amd/build/obj so that a change to one kernel file
    recompiles that file only), linked into the one C-ABI library.  No relocatable device code: no kernel calls across files."""
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    cc = hipcc()
    todo, objs = [], []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJDIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or extra or _stale(obj, src):
            todo.append([cc] + COMPILE_FLAGS + list(extra) + ["-c", src, "-o", obj])
    jobs = jobs or max(1, min(len(todo), (os.cpu_count() or 2) - 1, int(os.environ.get("DD_BUILD_JOBS", "6"))))
    running = []
    for cmd in todo:
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        running.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
        while len([1 for _, p in running if p.poll() is None]) >= jobs:
            running[0][1].wait()
            running = [r for r in running if r[1].poll() is None] + [r for r in running if r[1].poll() not in (None, 0)]
            if any(p.poll() not in (None, 0) for _, p in running):
                break
    bad = [cmd for cmd, p in running if p.wait() != 0]
    if bad:
        raise subprocess.CalledProcessError(1, bad[0])
    link = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp"] + objs + ["-lz", "-ldl", "-lpthread"]
    if verbose:
        print(" ".join(link), file=sys.stderr)
    subprocess.check_call(link, cwd=CSRC)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    extra = []
    if "--save-temps" in sys.argv:
        os.makedirs(OBJDIR, exist_ok=True)
        extra += ["-save-temps=obj"]
    build(force="--force" in sys.argv, extra=extra, verbose=True)
    print(LIB)
    print(build_cli(force="--force" in sys.argv))
