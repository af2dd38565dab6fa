"""CPU sanitizer leg: the oracle's C code under AddressSanitizer + UBSan through its own KAT
tests, and the engine's host-only code (K1 job planner, file loaders) under ASan + UBSan and under ThreadSanitizer.
GPU AddressSanitizer is not available on the pool, so device code is covered by the parity tests only."""
import os
import shutil
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "dandd_amd", "csrc")


def _gcc_lib(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(shutil.which("gcc") is None or _gcc_lib("libasan.so") is None, reason="gcc with libasan needed")
def test_oracle_kats_under_asan_ubsan(tmp_path):
    """oracle/liboracle_asan.so (make asan) behind tests/test_oracle.py: known-answer tests, ragged inputs, k 1..64,
    the exact counter -- any heap overflow / use-after-free / UB in the checker itself fails here."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    env = dict(os.environ, LD_PRELOAD=_gcc_lib("libasan.so"), ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", DD_ORACLE_LIB=os.path.join(ROOT, "oracle", "liboracle_asan.so"))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_oracle.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]


@pytest.mark.parametrize("sanitizer", ["address,undefined", "thread"])
def test_host_code_under_sanitizers(tmp_path, sanitizer):
    if shutil.which("g++") is None:
        pytest.skip("g++ needed")
    exe = str(tmp_path / "sanitize_host")
    cmd = ["g++", "-std=c++17", "-O1", "-g", f"-fsanitize={sanitizer}", "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__",
           "-I/opt/rocm/include", "-I" + CSRC, "-x", "c++", os.path.join(HERE, "native", "sanitize_host.cpp"),
           os.path.join(CSRC, "dd_plan.hip"), "-o", exe, "-L/opt/rocm/lib", "-lamdhip64", "-lz", "-ldl", "-lpthread",
           "-Wl,-rpath,/opt/rocm/lib"]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if b.returncode != 0 and "cannot find" in b.stderr:
        pytest.skip("sanitizer runtime not installed: " + b.stderr[-300:])
    assert b.returncode == 0, b.stderr[-3000:]
    work = tmp_path / "files"
    work.mkdir()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe, str(work)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "sanitize_host: ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    if sanitizer != "thread":  # gzip files through zlib only (no libdeflate on the machine): same answers
        r = subprocess.run([exe, str(work)], env=dict(env, DD_NO_LIBDEFLATE="1"), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "sanitize_host: ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
