"""ctypes wrapper over oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see oracle/dd_oracle.h).  The product (dandd_amd/) never does.
PARITY UNPINNED vs Dashing/KMC: see the header of dd_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(arch=None, out=None):
    """Compile liboracle.so with gcc (make).  arch='native' for timing builds."""
    cmd = ["make", "-s", "-C", _HERE]
    if arch:
        cmd.append(f"ARCH={arch}")
    if out:
        cmd += [f"OUT={out}", "-B"]
    subprocess.check_call(cmd)
    return out or os.path.join(_HERE, "liboracle.so")


def build_cli(arch=None, out=None):
    """Compile orc_cli (oracle/orc_cli.c: the oracle behind `dashing sketch | union | card` command lines) -> its path."""
    out = out or os.path.join(_HERE, "orc_cli")
    cmd = ["make", "-s", "-C", _HERE, "-B", "cli", f"CLI={out}"]
    if arch:
        cmd.append(f"ARCH={arch}")
    subprocess.check_call(cmd)
    return out


def _bind(lib):
    u8p, szp = C.POINTER(C.c_uint8), C.POINTER(C.c_size_t)
    lib.orc_wang64.restype = C.c_uint64
    lib.orc_wang64.argtypes = [C.c_uint64]
    lib.orc_fold128.restype = C.c_uint64
    lib.orc_fold128.argtypes = [C.c_uint64, C.c_uint64]
    lib.orc_idx_rho.restype = None
    lib.orc_idx_rho.argtypes = [C.c_uint64, C.c_int, C.POINTER(C.c_uint32), u8p]
    lib.orc_records.restype = C.c_size_t
    lib.orc_records.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    lib.orc_tokenize.restype = C.c_size_t
    lib.orc_tokenize.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    lib.orc_sketch.restype = C.c_int
    lib.orc_sketch.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.orc_sketch_generic.restype = C.c_int
    lib.orc_sketch_generic.argtypes = lib.orc_sketch.argtypes
    lib.orc_sketch_sweep.restype = C.c_int
    lib.orc_sketch_sweep.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.orc_union.restype = None
    lib.orc_union.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.orc_hist.restype = None
    lib.orc_hist.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    lib.orc_ertl_mle.restype = C.c_double
    lib.orc_ertl_mle.argtypes = [C.c_void_p, C.c_int]
    lib.orc_card.restype = C.c_double
    lib.orc_card.argtypes = [C.c_void_p, C.c_int]
    lib.orc_exact_count.restype = C.c_int
    lib.orc_exact_count.argtypes = [C.POINTER(C.c_void_p), szp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
    lib.orc_synth_size.restype = C.c_size_t
    lib.orc_synth_size.argtypes = [C.c_uint64, C.c_int]
    lib.orc_synth_fasta.restype = C.c_size_t
    lib.orc_synth_fasta.argtypes = [C.c_uint64, C.c_int, C.c_uint64, C.c_int, C.c_void_p]
    lib.orc_synth_realistic_size.restype = C.c_size_t
    lib.orc_synth_realistic_size.argtypes = [C.c_uint64, C.c_uint64]
    lib.orc_synth_realistic_fasta.restype = C.c_size_t
    lib.orc_synth_realistic_fasta.argtypes = [C.c_uint64, C.c_int, C.c_uint64, C.c_void_p]
    lib.orc_splitmix64.restype = C.c_uint64
    lib.orc_splitmix64.argtypes = [C.c_uint64]
    return lib


def lib(path=None):
    global _LIB
    if path:
        return _bind(C.CDLL(path))
    if _LIB is None:
        p = os.environ.get("DD_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")  # DD_ORACLE_LIB: the sanitizer build (tests/test_sanitizers.py)
        if not os.path.exists(p):
            build()
        _LIB = _bind(C.CDLL(p))
    return _LIB


def _buf(b):
    """bytes / bytearray / uint8 ndarray -> contiguous uint8 ndarray (no copy when possible)."""
    if isinstance(b, np.ndarray):
        return np.ascontiguousarray(b, dtype=np.uint8)
    return np.frombuffer(b, dtype=np.uint8)


def wang64(x):
    return int(lib().orc_wang64(C.c_uint64(x & (2**64 - 1))))


def fold128(hi, lo):
    return int(lib().orc_fold128(C.c_uint64(hi), C.c_uint64(lo)))


def idx_rho(h, p):
    idx, rho = C.c_uint32(), C.c_uint8()
    lib().orc_idx_rho(C.c_uint64(h), p, C.byref(idx), C.byref(rho))
    return idx.value, rho.value


def records(fa):
    """Sequence strings of the records of a FASTA / FASTQ buffer, kseq's rules (orc_records) -> list of bytes"""
    a = np.ascontiguousarray(fa, dtype=np.uint8)
    out = np.empty(a.size + 1, dtype=np.uint8)
    n = lib().orc_records(a.ctypes.data, a.size, out.ctypes.data)
    body = out[:n].tobytes()
    return body.split(b"\n")[:-1] if n else []


def tokenize(fa):
    a = _buf(fa)
    n = lib().orc_tokenize(a.ctypes.data, a.size, None)
    out = np.empty(n, dtype=np.uint8)
    lib().orc_tokenize(a.ctypes.data, a.size, out.ctypes.data)
    return out


def sketch(fa, k, p, canonical=True, regs=None):
    a = _buf(fa)
    if regs is None:
        regs = np.zeros(1 << p, dtype=np.uint8)
    rc = lib().orc_sketch(a.ctypes.data, a.size, k, p, int(canonical), regs.ctypes.data)
    if rc:
        raise ValueError(f"orc_sketch rc={rc}")
    return regs


def sketch_generic(fa, k, p, canonical=True):
    a = _buf(fa)
    regs = np.zeros(1 << p, dtype=np.uint8)
    rc = lib().orc_sketch_generic(a.ctypes.data, a.size, k, p, int(canonical), regs.ctypes.data)
    if rc:
        raise ValueError(f"orc_sketch_generic rc={rc}")
    return regs


def sketch_sweep(fa, kmin, kmax, p, canonical=True, regs=None):
    a = _buf(fa)
    if regs is None:
        regs = np.zeros((kmax - kmin + 1, 1 << p), dtype=np.uint8)
    rc = lib().orc_sketch_sweep(a.ctypes.data, a.size, kmin, kmax, p, int(canonical), regs.ctypes.data)
    if rc:
        raise ValueError(f"orc_sketch_sweep rc={rc}")
    return regs


def union(*regs):
    out = np.array(regs[0], dtype=np.uint8, copy=True)
    for r in regs[1:]:
        r = np.ascontiguousarray(r, dtype=np.uint8)
        lib().orc_union(out.ctypes.data, r.ctypes.data, out.size)
    return out


def hist(regs):
    r = np.ascontiguousarray(regs, dtype=np.uint8)
    h = np.zeros(64, dtype=np.uint32)
    lib().orc_hist(r.ctypes.data, r.size, h.ctypes.data)
    return h


def ertl_mle(h, p):
    h = np.ascontiguousarray(h, dtype=np.uint32)
    return float(lib().orc_ertl_mle(h.ctypes.data, p))


def card(regs, p=None):
    r = np.ascontiguousarray(regs, dtype=np.uint8)
    if p is None:
        p = int(r.size).bit_length() - 1
    return float(lib().orc_card(r.ctypes.data, p))


def exact_count(fas, k, canonical=True):
    arrs = [_buf(f) for f in fas]
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    ns = (C.c_size_t * len(arrs))(*[a.size for a in arrs])
    d = C.c_uint64()
    rc = lib().orc_exact_count(ptrs, ns, len(arrs), k, int(canonical), C.byref(d))
    if rc:
        raise ValueError(f"orc_exact_count rc={rc}")
    return d.value


def synth_fasta(seed, gi, nbases, nrec=1):
    n = lib().orc_synth_size(nbases, nrec)
    out = np.empty(n, dtype=np.uint8)
    w = lib().orc_synth_fasta(C.c_uint64(seed), gi, C.c_uint64(nbases), nrec, out.ctypes.data)
    assert w == n, (w, n)
    return out


def synth_realistic(seed, gi, nbases):
    """GC 35 %, 30 % soft-masked repeats, 2 % N, contigs of 2..200 kbp (dd_oracle.c)."""
    n = lib().orc_synth_realistic_size(C.c_uint64(seed), C.c_uint64(nbases))
    out = np.empty(n, dtype=np.uint8)
    w = lib().orc_synth_realistic_fasta(C.c_uint64(seed), gi, C.c_uint64(nbases), out.ctypes.data)
    assert w == n, (w, n)
    return out


def splitmix64(x):
    return int(lib().orc_splitmix64(C.c_uint64(x & (2**64 - 1))))
