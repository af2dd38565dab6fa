"""ctypes binding of libdandd_hip.so (include/dandd_hip.h) -- the only way Python reaches the GPU.

This is the host-side stub a DandD maintainer would drop next to lib/sketch_classes.py in
place of the `subprocess` calls at /root/reference/lib/sketch_classes.py:190,198,221,229,268,274
and /root/reference/lib/huffman_dandd.py:233.

There is NO CPU fallback: `Engine(...)` raises `EngineError` when the shared library is missing
or no gfx950 device is usable.  numpy arrays are host buffers; device buffers are passed as
integer addresses (e.g. `torch.Tensor.data_ptr()`), so no torch type crosses the C ABI.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdandd_hip.so")

KERNEL_PACK, KERNEL_SWEEP, KERNEL_UNION = 0, 1, 2
ABI_VERSION = 4   # include/dandd_hip.h: DD_ABI_VERSION

EXPORTS = [
    "dd_abi_version", "dd_last_error", "dd_create", "dd_destroy", "dd_set_stream", "dd_synchronize",
    "dd_sketch_buffer", "dd_sketch_fasta", "dd_sketch_files", "dd_inflate_files", "dd_last_ingest_stats", "dd_sketch_device", "dd_union", "dd_union_device",
    "dd_card", "dd_card_batch", "dd_card_batch_device", "dd_hist_batch_device", "dd_ertl_mle",
    "dd_progressive", "dd_progressive_device", "dd_pairwise", "dd_pairwise_device",
    "dd_exact_count", "dd_exact_count_device",
    "dd_timing_enable", "dd_timing_read", "dd_timing_reset", "dd_last_sketch_stats", "dd_last_k2_path",
    "dd_synth_size", "dd_synth_fasta_device", "dd_synth_realistic_size", "dd_synth_realistic_device", "dd_plan_sweep",
    "dd_comm_unique_id", "dd_comm_init", "dd_comm_destroy", "dd_comm_info", "dd_allreduce_max_u8", "dd_allgather_u8",
]
COMM_ID_BYTES = 128   # include/dandd_hip.h: DD_COMM_ID_BYTES


class EngineError(RuntimeError):
    pass


_lib = None


def load_library(path=None):
    """dlopen libdandd_hip.so and declare every prototype.  Raises EngineError if absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("DANDD_LIB") or LIB_PATH  # DANDD_LIB: an alternative build (development)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so (same SONAME as
    # /opt/rocm's).  If this library were loaded first it would pull in the system copy, a later
    # `import torch` would add the bundled one, and the second runtime to initialise reports "no
    # ROCm-capable device".  Importing torch first (when it is installed) makes the dynamic loader
    # resolve our NEEDED libamdhip64.so.7 to the copy torch already mapped.
    # A process that will never import torch (the `dandd` CLI) sets DANDD_NO_TORCH=1 and saves the
    # ~1 s import; the library then binds to the system HIP runtime.
    if "torch" not in sys.modules and os.environ.get("DANDD_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    if not os.path.exists(p):
        raise EngineError(
            f"{p} not found: build it with `python -m dandd_amd.build` (hipcc, gfx950). "
            "dandd_amd has no CPU implementation of the sketching path.")
    try:
        lib = C.CDLL(p)
    except OSError as e:  # e.g. libamdhip64 missing
        raise EngineError(f"cannot load {p}: {e}") from e
    vp, sz, i32, u64 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint64
    lib.dd_abi_version.restype = i32
    lib.dd_abi_version.argtypes = []
    if lib.dd_abi_version() != ABI_VERSION:
        raise EngineError(f"{p} has ABI version {lib.dd_abi_version()}, this binding is written against {ABI_VERSION} "
                          "(include/dandd_hip.h: DD_ABI_VERSION); rebuild with `python -m dandd_amd.build --force`")
    lib.dd_last_error.restype = C.c_char_p
    lib.dd_last_error.argtypes = []
    lib.dd_create.restype = vp
    lib.dd_create.argtypes = [i32, i32, i32]
    lib.dd_destroy.restype = None
    lib.dd_destroy.argtypes = [vp]
    lib.dd_set_stream.restype = i32
    lib.dd_set_stream.argtypes = [vp, vp]
    lib.dd_synchronize.restype = i32
    lib.dd_synchronize.argtypes = [vp]
    lib.dd_sketch_buffer.restype = i32
    lib.dd_sketch_buffer.argtypes = [vp, vp, sz, i32, i32, vp]
    lib.dd_sketch_fasta.restype = i32
    lib.dd_sketch_fasta.argtypes = [vp, C.c_char_p, i32, i32, vp]
    lib.dd_sketch_files.restype = i32
    lib.dd_sketch_files.argtypes = [vp, C.POINTER(C.c_char_p), i32, i32, i32, vp, i32]
    lib.dd_comm_unique_id.restype = i32
    lib.dd_comm_unique_id.argtypes = [vp]
    lib.dd_comm_init.restype = i32
    lib.dd_comm_init.argtypes = [vp, i32, i32, vp]
    lib.dd_comm_destroy.restype = i32
    lib.dd_comm_destroy.argtypes = [vp]
    lib.dd_comm_info.restype = i32
    lib.dd_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    lib.dd_allreduce_max_u8.restype = i32
    lib.dd_allreduce_max_u8.argtypes = [vp, vp, C.c_size_t]
    lib.dd_allgather_u8.restype = i32
    lib.dd_allgather_u8.argtypes = [vp, vp, C.c_size_t, vp]
    lib.dd_inflate_files.restype = i32
    lib.dd_inflate_files.argtypes = [vp, C.POINTER(C.c_char_p), i32, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), i32]
    lib.dd_last_ingest_stats.restype = i32
    lib.dd_last_ingest_stats.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(i32), C.POINTER(u64)]
    lib.dd_sketch_device.restype = i32
    lib.dd_sketch_device.argtypes = [vp, C.POINTER(vp), C.POINTER(sz), i32, i32, i32, vp]
    lib.dd_union.restype = i32
    lib.dd_union.argtypes = [vp, C.POINTER(vp), i32, sz, vp]
    lib.dd_union_device.restype = i32
    lib.dd_union_device.argtypes = [vp, C.POINTER(vp), i32, sz, vp]
    lib.dd_card.restype = i32
    lib.dd_card.argtypes = [vp, vp, C.POINTER(C.c_double)]
    lib.dd_card_batch.restype = i32
    lib.dd_card_batch.argtypes = [vp, vp, i32, vp]
    lib.dd_card_batch_device.restype = i32
    lib.dd_card_batch_device.argtypes = [vp, vp, i32, vp]
    lib.dd_hist_batch_device.restype = i32
    lib.dd_hist_batch_device.argtypes = [vp, vp, i32, vp]
    lib.dd_ertl_mle.restype = C.c_double
    lib.dd_ertl_mle.argtypes = [vp, i32]
    lib.dd_progressive.restype = i32
    lib.dd_progressive.argtypes = [vp, vp, i32, i32, vp, i32, vp]
    lib.dd_progressive_device.restype = i32
    lib.dd_progressive_device.argtypes = [vp, vp, i32, i32, vp, i32, vp]
    lib.dd_pairwise.restype = i32
    lib.dd_pairwise.argtypes = [vp, vp, i32, i32, vp]
    lib.dd_pairwise_device.restype = i32
    lib.dd_pairwise_device.argtypes = [vp, vp, i32, i32, vp]
    lib.dd_exact_count.restype = i32
    lib.dd_exact_count.argtypes = [vp, C.POINTER(C.c_char_p), i32, i32, C.POINTER(u64)]
    lib.dd_exact_count_device.restype = i32
    lib.dd_exact_count_device.argtypes = [vp, C.POINTER(vp), C.POINTER(sz), i32, i32, C.POINTER(u64)]
    lib.dd_timing_enable.restype = i32
    lib.dd_timing_enable.argtypes = [vp, i32]
    lib.dd_timing_read.restype = i32
    lib.dd_timing_read.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(i32)]
    lib.dd_timing_reset.restype = i32
    lib.dd_timing_reset.argtypes = [vp]
    lib.dd_last_sketch_stats.restype = i32
    lib.dd_last_sketch_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(i32)]
    lib.dd_last_k2_path.restype = i32
    lib.dd_last_k2_path.argtypes = [vp]
    lib.dd_synth_size.restype = sz
    lib.dd_synth_size.argtypes = [u64, i32]
    lib.dd_synth_fasta_device.restype = i32
    lib.dd_synth_fasta_device.argtypes = [vp, u64, i32, u64, i32, vp]
    lib.dd_synth_realistic_size.restype = sz
    lib.dd_synth_realistic_size.argtypes = [u64, u64]
    lib.dd_synth_realistic_device.restype = i32
    lib.dd_synth_realistic_device.argtypes = [vp, u64, i32, u64, vp]
    lib.dd_plan_sweep.restype = C.c_long
    lib.dd_plan_sweep.argtypes = [i32, C.POINTER(sz), i32, i32, i32, vp, C.c_long]
    if path is None:
        _lib = lib
    return lib


def ertl_mle(hist, log2m):
    """Host-side Ertl ML estimate of one 64-bin histogram (no device needed)."""
    h = np.ascontiguousarray(hist, dtype=np.uint32)
    if h.size != 64:
        raise ValueError("histogram must have 64 bins")
    return float(load_library().dd_ertl_mle(h.ctypes.data, int(log2m)))


def synth_realistic_size(seed, nbases):
    return int(load_library().dd_synth_realistic_size(int(seed), int(nbases)))


def synth_size(nbases, nrec=1):
    return int(load_library().dd_synth_size(int(nbases), int(nrec)))


PLAN_JOB = np.dtype([("kclass", np.int32), ("mode", np.int32), ("lds_bytes", np.int32), ("genome", np.int32),
                     ("kfirst", np.int32), ("nk", np.int32), ("tile_begin", np.uint32), ("tile_end", np.uint32),
                     ("slice", np.int32)])


def plan_sweep(log2m, nbytes, kmin, kmax):
    """The K1 job table dd_sketch_device would use for genomes of `nbytes` bytes (structured array, launch
    order).  Pure host code in the library: works without a GPU."""
    lib = load_library()
    sizes = (C.c_size_t * len(nbytes))(*[int(n) for n in nbytes])
    n = lib.dd_plan_sweep(int(log2m), sizes, len(nbytes), int(kmin), int(kmax), None, 0)
    if n < 0:
        raise EngineError(f"libdandd_hip error {n}: {lib.dd_last_error().decode()}")
    out = np.zeros(n, dtype=PLAN_JOB)
    got = lib.dd_plan_sweep(int(log2m), sizes, len(nbytes), int(kmin), int(kmax), out.ctypes.data, n)
    assert got == n
    return out


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def comm_unique_id():
    """The 128-byte id rank 0 makes for an RCCL communicator (dd_comm_unique_id); every rank passes it to Engine.comm_init."""
    lib = load_library()
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    rc = lib.dd_comm_unique_id(buf)
    if rc != 0:
        raise EngineError(f"libdandd_hip error {rc}: {lib.dd_last_error().decode()}")
    return bytes(buf)


class Engine:
    """One context = one GPU, one HLL size (log2m), canonical or not."""

    def __init__(self, device=0, log2m=14, canonical=True):
        self._lib = load_library()
        self.log2m = int(log2m)
        self.m = 1 << self.log2m
        self.canonical = bool(canonical)
        self.device = int(device)
        self._ctx = self._lib.dd_create(self.device, self.log2m, int(self.canonical))
        if not self._ctx:
            raise EngineError("dd_create failed: " + self._lib.dd_last_error().decode())

    # -- lifetime -------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.dd_destroy(self._ctx)
            self._ctx = None

    # -- plain device memory for callers without torch (the host layer keeps a collection's leaf slab resident) --------------
    _hip = None

    @classmethod
    def _hip_runtime(cls):
        """The HIP runtime libdandd_hip.so is bound to (same SONAME -> the copy already mapped), for hipMalloc / hipMemcpy / hipFree."""
        if cls._hip is None:
            for name in ("libamdhip64.so.7", "libamdhip64.so.6", "libamdhip64.so"):
                try:
                    h = C.CDLL(name)
                    break
                except OSError:
                    h = None
            if h is None:
                raise EngineError("libamdhip64.so not found")
            h.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
            h.hipFree.argtypes = [C.c_void_p]
            h.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            h.hipSetDevice.argtypes = [C.c_int]
            cls._hip = h
        return cls._hip

    def device_alloc(self, nbytes):
        h = self._hip_runtime()
        ptr = C.c_void_p()
        if h.hipSetDevice(self.device) != 0 or h.hipMalloc(C.byref(ptr), int(nbytes)) != 0 or not ptr.value:
            raise EngineError(f"hipMalloc of {nbytes} bytes failed")
        return ptr.value

    def device_free(self, ptr):
        if ptr:
            self._hip_runtime().hipFree(C.c_void_p(int(ptr)))

    def device_upload(self, ptr, host):
        host = _u8(host)
        self.synchronize()
        if self._hip_runtime().hipMemcpy(C.c_void_p(int(ptr)), host.ctypes.data, host.nbytes, 1) != 0:   # hipMemcpyHostToDevice
            raise EngineError("hipMemcpy (host to device) failed")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise EngineError(f"libdandd_hip error {rc}: {self._lib.dd_last_error().decode()}")

    def set_stream(self, hip_stream):
        self._check(self._lib.dd_set_stream(self._ctx, C.c_void_p(int(hip_stream) if hip_stream else 0)))

    def synchronize(self):
        self._check(self._lib.dd_synchronize(self._ctx))

    # -- sketch ---------------------------------------------------------------------------
    def sketch_buffer(self, fasta, kmin, kmax):
        """FASTA bytes (host) -> registers [K][m] uint8."""
        a = _u8(np.frombuffer(fasta, dtype=np.uint8) if not isinstance(fasta, np.ndarray) else fasta)
        regs = np.empty((kmax - kmin + 1, self.m), dtype=np.uint8)
        self._check(self._lib.dd_sketch_buffer(self._ctx, a.ctypes.data, a.size, kmin, kmax, regs.ctypes.data))
        return regs

    def sketch_fasta(self, path, kmin, kmax):
        regs = np.empty((kmax - kmin + 1, self.m), dtype=np.uint8)
        self._check(self._lib.dd_sketch_fasta(self._ctx, os.fsencode(path), kmin, kmax, regs.ctypes.data))
        return regs

    def sketch_files(self, paths, kmin, kmax, nthreads=0):
        """Plain or .gz FASTA files -> registers [n][K][m]; read/inflate overlaps the GPU work."""
        n = len(paths)
        arr = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
        regs = np.empty((n, kmax - kmin + 1, self.m), dtype=np.uint8)
        self._check(self._lib.dd_sketch_files(self._ctx, arr, n, kmin, kmax, regs.ctypes.data, int(nthreads)))
        return regs

    # -- multi-GPU: RCCL behind the C ABI (one context = one process = one GPU) --------------
    def comm_init(self, rank, world, unique_id):
        """Join the communicator `unique_id` names (collective: every rank calls it, each on the GPU it owns)."""
        assert len(unique_id) == COMM_ID_BYTES
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        self._check(self._lib.dd_comm_init(self._ctx, int(rank), int(world), buf))

    def comm_destroy(self):
        self._check(self._lib.dd_comm_destroy(self._ctx))

    def comm_info(self):
        """(rank, world, all-reduces issued, all-gathers issued); world 0 = no communicator."""
        r, w, a, g = C.c_int(), C.c_int(), C.c_ulonglong(), C.c_ulonglong()
        self._check(self._lib.dd_comm_info(self._ctx, C.byref(r), C.byref(w), C.byref(a), C.byref(g)))
        return r.value, w.value, a.value, g.value

    def allreduce_max_u8(self, regs_ptr, n):
        """In-place byte-max all-reduce of n register bytes at device address regs_ptr (ncclUint8 / ncclMax), on the context's stream."""
        self._check(self._lib.dd_allreduce_max_u8(self._ctx, C.c_void_p(int(regs_ptr)), int(n)))

    def allgather_u8(self, send_ptr, n, recv_ptr):
        """n bytes from every rank -> recv[world][n] on every rank (device addresses), on the context's stream."""
        self._check(self._lib.dd_allgather_u8(self._ctx, C.c_void_p(int(send_ptr)), int(n), C.c_void_p(int(recv_ptr))))

    def inflate_files(self, paths, nthreads=0):
        """The bytes the tokenizer reads for every file of a sketch_files pass (dd_inflate_files): for a .gz FASTA file the
        text `zcat` prints, inflated on the device where the device decoder takes the file.  -> list of uint8 arrays."""
        n = len(paths)
        arr = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
        caps = []
        for p in paths:       # room: the trailer's ISIZE for a .gz (a multi-member file's is too small: second try below), else the file
            size = os.path.getsize(p)
            with open(p, "rb") as f:
                head = f.read(2)
                if head == b"\x1f\x8b" and size >= 18:
                    f.seek(size - 4)
                    size = max(int.from_bytes(f.read(4), "little"), 4 * size)
            caps.append(size + 64)
        for _ in range(2):
            bufs = [np.empty(c, dtype=np.uint8) for c in caps]
            outs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
            ccaps = (C.c_size_t * n)(*caps)
            lens = (C.c_size_t * n)()
            rc = self._lib.dd_inflate_files(self._ctx, arr, n, outs, ccaps, lens, int(nthreads))
            if rc == 0 or all(lens[i] <= caps[i] for i in range(n)):
                break
            caps = [max(int(lens[i]), caps[i]) + 64 for i in range(n)]
        self._check(rc)
        return [bufs[i][:lens[i]] for i in range(n)]

    def last_ingest_stats(self):
        """(wall ms, ms waiting for the loader threads, batched launches, FASTA bytes) of the last sketch_files."""
        w, l, b, n = C.c_double(), C.c_double(), C.c_int(), C.c_uint64()
        self._check(self._lib.dd_last_ingest_stats(self._ctx, C.byref(w), C.byref(l), C.byref(b), C.byref(n)))
        return w.value, l.value, b.value, n.value

    def sketch_device(self, fasta_ptrs, nbytes, kmin, kmax, regs_ptr):
        """Batched HBM-resident sketch: device addresses in, regs_ptr[ng][K][m] device address out."""
        n = len(fasta_ptrs)
        ptrs = (C.c_void_p * n)(*[int(x) for x in fasta_ptrs])
        ns = (C.c_size_t * n)(*[int(x) for x in nbytes])
        self._check(self._lib.dd_sketch_device(self._ctx, ptrs, ns, n, kmin, kmax, C.c_void_p(int(regs_ptr))))

    # -- union / card ---------------------------------------------------------------------
    def union(self, sketches):
        arrs = [_u8(s) for s in sketches]
        n = arrs[0].size
        if any(a.size != n for a in arrs):
            raise ValueError("union inputs differ in size")
        out = np.empty_like(arrs[0])
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        self._check(self._lib.dd_union(self._ctx, ptrs, len(arrs), n, out.ctypes.data))
        return out

    def union_device(self, in_ptrs, nbytes, out_ptr):
        ptrs = (C.c_void_p * len(in_ptrs))(*[int(x) for x in in_ptrs])
        self._check(self._lib.dd_union_device(self._ctx, ptrs, len(in_ptrs), int(nbytes), C.c_void_p(int(out_ptr))))

    def card(self, regs):
        r = _u8(regs)
        if r.size != self.m:
            raise ValueError(f"expected {self.m} registers, got {r.size}")
        est = C.c_double()
        self._check(self._lib.dd_card(self._ctx, r.ctypes.data, C.byref(est)))
        return est.value

    def card_batch(self, regs):
        r = _u8(regs).reshape(-1, self.m)
        est = np.empty(r.shape[0], dtype=np.float64)
        self._check(self._lib.dd_card_batch(self._ctx, r.ctypes.data, r.shape[0], est.ctypes.data))
        return est

    def card_batch_device(self, regs_ptr, njobs):
        est = np.empty(njobs, dtype=np.float64)
        self._check(self._lib.dd_card_batch_device(self._ctx, C.c_void_p(int(regs_ptr)), njobs, est.ctypes.data))
        return est

    def hist_batch_device(self, regs_ptr, njobs):
        h = np.empty((njobs, 64), dtype=np.uint32)
        self._check(self._lib.dd_hist_batch_device(self._ctx, C.c_void_p(int(regs_ptr)), njobs, h.ctypes.data))
        return h

    # -- progressive / pairwise -----------------------------------------------------------
    def progressive(self, leaf, orderings):
        """leaf [n][K][m] uint8 (host), orderings [o][n] int -> card [o][n][K] float64."""
        leaf = _u8(leaf)
        n, K = leaf.shape[0], leaf.shape[1]
        ords = np.ascontiguousarray(orderings, dtype=np.int32).reshape(-1, n)
        card = np.empty((ords.shape[0], n, K), dtype=np.float64)
        self._check(self._lib.dd_progressive(self._ctx, leaf.ctypes.data, n, K, ords.ctypes.data,
                                             ords.shape[0], card.ctypes.data))
        return card

    def progressive_device(self, leaf_ptr, n, K, orderings):
        ords = np.ascontiguousarray(orderings, dtype=np.int32).reshape(-1, n)
        card = np.empty((ords.shape[0], n, K), dtype=np.float64)
        self._check(self._lib.dd_progressive_device(self._ctx, C.c_void_p(int(leaf_ptr)), n, K,
                                                    ords.ctypes.data, ords.shape[0], card.ctypes.data))
        return card

    def pairwise(self, leaf):
        leaf = _u8(leaf)
        n, K = leaf.shape[0], leaf.shape[1]
        card = np.empty((n, n, K), dtype=np.float64)
        self._check(self._lib.dd_pairwise(self._ctx, leaf.ctypes.data, n, K, card.ctypes.data))
        return card

    def pairwise_device(self, leaf_ptr, n, K):
        card = np.empty((n, n, K), dtype=np.float64)
        self._check(self._lib.dd_pairwise_device(self._ctx, C.c_void_p(int(leaf_ptr)), n, K, card.ctypes.data))
        return card

    # -- exact distinct k-mer count (KMC stand-in) ----------------------------------------
    def exact_count(self, paths, k):
        """Distinct (canonical per the context) k-mers over all the FASTA files together."""
        arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
        d = C.c_uint64()
        self._check(self._lib.dd_exact_count(self._ctx, arr, len(paths), int(k), C.byref(d)))
        return d.value

    def exact_count_device(self, fasta_ptrs, nbytes, k):
        n = len(fasta_ptrs)
        ptrs = (C.c_void_p * n)(*[int(x) for x in fasta_ptrs])
        ns = (C.c_size_t * n)(*[int(x) for x in nbytes])
        d = C.c_uint64()
        self._check(self._lib.dd_exact_count_device(self._ctx, ptrs, ns, n, int(k), C.byref(d)))
        return d.value

    # -- measurement ----------------------------------------------------------------------
    def timing_enable(self, on=True):
        self._check(self._lib.dd_timing_enable(self._ctx, int(bool(on))))

    def timing_reset(self):
        self._check(self._lib.dd_timing_reset(self._ctx))

    def timing_read(self, which):
        ms, n = C.c_double(), C.c_int()
        self._check(self._lib.dd_timing_read(self._ctx, which, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def last_sketch_stats(self):
        t, u, b = C.c_uint64(), C.c_uint64(), C.c_int()
        self._check(self._lib.dd_last_sketch_stats(self._ctx, C.byref(t), C.byref(u), C.byref(b)))
        return t.value, u.value, b.value

    K2_PATHS = {0: None, 1: "progressive_stream", 2: "progressive_pscan", 3: "pairwise_stream", 4: "pairwise_gram"}

    def last_k2_path(self):
        """Which device form the last progressive / pairwise call took (include/dandd_hip.h: DD_K2_*)."""
        rc = self._lib.dd_last_k2_path(self._ctx)
        if rc < 0:
            self._check(rc)
        return self.K2_PATHS[rc]

    def warmup(self):
        """One tiny sketch + cardinality: HIP loads a kernel module at its first launch, this makes the first launches
        happen now (on whichever thread calls it) instead of inside the first real call."""
        fa = np.frombuffer(b">w\n" + b"ACGTTGCAACGGTCA" * 16 + b"\n", dtype=np.uint8)
        self.card_batch(self.sketch_buffer(fa, 15, 17))

    def synth_realistic_device(self, seed, genome_index, nbases, out_ptr):
        self._check(self._lib.dd_synth_realistic_device(self._ctx, int(seed), int(genome_index), int(nbases), C.c_void_p(int(out_ptr))))

    def synth_fasta_device(self, seed, genome_index, nbases, nrec, out_ptr):
        self._check(self._lib.dd_synth_fasta_device(self._ctx, int(seed), int(genome_index), int(nbases),
                                                    int(nrec), C.c_void_p(int(out_ptr))))
