"""Sketch backends: the three operations DandD obtains from `dashing sketch|union|card`.

The contract is the reference's own process boundary, kept file-based so the
sketch directory layout and every cached file stay where DandD expects them:

    leaf(fasta, ks, out_paths)   <- parallel ... ' dashing sketch -k{} -S R --prefix dir fasta ' ::: ks
                                    (/root/reference/lib/huffman_dandd.py:214-218,
                                     /root/reference/lib/sketch_classes.py:351-366)
    union(in_paths, out_path)    <- dashing union -z -o out in...  (lib/sketch_classes.py:368-373)
    card(path) -> float          <- dashing card --presketched path (lib/sketch_classes.py:306-321)

`HipBackend` is the product: every operation runs on the MI355X through libdandd_hip.so
(dandd_amd.engine).  There is no CPU implementation in this package; tests inject their own
checker backends through the same three methods.
"""
import os
import struct

import numpy as np

MAGIC = b"DDHLL\x01\x00\x00"
_HDR = struct.Struct("<8sBBBB")  # magic, log2m, k, canonical, reserved

# Two containers live under the reference's `.hll` names (lib/sketch_classes.py:47-52,100,110):
#   native  : 12-byte header + 2^log2m register bytes -- what this engine writes by default (no zlib pass
#             over every sketch of every k);
#   dashing : Dashing's own container (gzip or plain: `-z` is optional there), so a sketch directory made by
#             a real DandD + Dashing can be read and, with DANDD_SKETCH_FORMAT=dashing, extended in place.
# Reading detects the container by its first bytes; writing follows the switch.  Dashing's 32-byte header is
# taken from the published sketch library as recalled -- no Dashing binary exists here to check it against
# (DESIGN.md section 6); the register bytes are the part this repo verifies.
_DASH = struct.Struct("<5IId")  # is_calculated, clamp, estimator, joint estimator, nthreads; np; cached value
_ERTL_MLE, _ERTL_JOINT_MLE = 2, 3


def sketch_format():
    fmt = os.environ.get("DANDD_SKETCH_FORMAT", "native").lower()
    if fmt not in ("native", "dashing", "dashing-plain"):
        raise ValueError(f"DANDD_SKETCH_FORMAT={fmt!r}: expected native, dashing or dashing-plain")
    return fmt


def _name_k(path):
    """k from a sketch file NAME (Dashing's container does not hold it): `.w.<k>.spacing.` or `k<k>[nc].hll`."""
    import re
    base = os.path.basename(path)
    m = re.search(r"\.w\.(\d+)\.spacing\.", base) or re.search(r"k(\d+)(?:nc)?\.hll$", base)
    return int(m.group(1)) if m else 0


def write_sketch_file(path, regs, log2m, k, canonical, fmt=None):
    """One HLL sketch on disk, in the container DANDD_SKETCH_FORMAT selects."""
    regs = np.ascontiguousarray(regs, dtype=np.uint8)
    if regs.size != (1 << log2m):
        raise ValueError("register count does not match log2m")
    fmt = fmt or sketch_format()
    tmp = f"{path}.{os.getpid()}.tmp"  # per process: ranks of a multi-GPU run share the sketch directory
    if fmt == "native":
        with open(tmp, "wb", buffering=0) as f:  # (plain write(2) calls; no copy of the registers)
            # a raw write may be short (disk filling, NFS, a signal): loop until every byte is out, or let the
            # OS error surface -- a truncated temp file must never be renamed into the sketch cache
            for piece in (_HDR.pack(MAGIC, log2m, k, 1 if canonical else 0, 0), regs):
                mv = memoryview(piece).cast("B")
                while len(mv):
                    mv = mv[f.write(mv):]
    else:
        head = _DASH.pack(0, 0, _ERTL_MLE, _ERTL_JOINT_MLE, 1, log2m, 0.0)
        if fmt == "dashing":
            import gzip
            with gzip.open(tmp, "wb", compresslevel=6) as f:
                f.write(head)
                f.write(regs.tobytes())
        else:
            with open(tmp, "wb") as f:
                f.write(head)
                f.write(regs.tobytes())
    os.replace(tmp, path)


def read_sketch_file(path):
    """-> (registers, log2m, k, canonical) from either container."""
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:8] == MAGIC:
        magic, log2m, k, canonical, _ = _HDR.unpack_from(raw)
        regs = np.frombuffer(raw, dtype=np.uint8, offset=_HDR.size)
        if regs.size != (1 << log2m):
            raise ValueError(f"{path}: expected {1 << log2m} registers, found {regs.size}")
        return regs, log2m, k, bool(canonical)
    if raw[:2] == b"\x1f\x8b":
        import gzip
        raw = gzip.decompress(raw)
    if len(raw) < _DASH.size:
        raise ValueError(f"{path}: truncated sketch file")
    _calc, _clamp, _est, _jest, _nthr, np_, _value = _DASH.unpack_from(raw)
    regs = np.frombuffer(raw, dtype=np.uint8, offset=_DASH.size)
    if not 4 <= np_ <= 32 or regs.size != (1 << np_):
        raise ValueError(f"{path}: neither a dandd_amd nor a Dashing sketch file")
    # k and the canonical flag are not in Dashing's container: k comes from the file name, and DandD marks
    # non-canonical unions with an `nc` suffix (leaves carry no marker)
    return regs, int(np_), _name_k(path), not os.path.basename(path).endswith("nc.hll")


def convert_main(argv):
    """`python -m dandd_amd.host.backend export <sketch.hll> <out.hll>`  (native -> Dashing's container, gzip)
    `python -m dandd_amd.host.backend import <in.hll> <sketch.hll> K [--no-canon]`  (Dashing's -> native; k and the
    canonical flag are not part of Dashing's container, they live in its file NAME)."""
    if len(argv) >= 3 and argv[0] == "export":
        regs, log2m, k, canon = read_sketch_file(argv[1])
        write_sketch_file(argv[2], regs, log2m, k, canon, fmt="dashing")
        return 0
    if len(argv) >= 4 and argv[0] == "import":
        regs, log2m, _k, _canon = read_sketch_file(argv[1])
        write_sketch_file(argv[2], regs, log2m, int(argv[3]), "--no-canon" not in argv, fmt="native")
        return 0
    print(convert_main.__doc__)
    return 2


class StaleGenome(FileNotFoundError):
    """A member's file is where it was recorded, but is not the file it was: another size."""


class HipExactBackend:
    """`--exact`: the KMC branch of the reference (lib/sketch_classes.py:377-465), which counts
    distinct canonical k-mers exactly.  A "database" here is a small JSON file listing the FASTAs it
    covers -- base name, the directory it was seen in and its size in bytes -- and, once it has been asked
    for, their exact distinct k-mer count, computed on the GPU (sort + distinct, dd_exact_count).  Like a KMC
    database it answers `info` on its own afterwards, and it survives a moved genome directory as long as
    DANDD_GENOMEDIR (or the recorded directory) still holds files of those names AND sizes.  A genome is
    identified the way DandD's catalog identifies it, by base name (lib/species_specifics.py keys `fastahex` by
    basename) -- plus its size, so that two different files of one name in different directories are refused
    instead of silently counted as one, and a cached count is dropped when the file it was made from has been
    replaced.  The reference's own KMC branch recurses forever at this commit; this one works."""

    name = "hip-exact"

    def __init__(self, log2m=20, canonical=True, device=0):
        from ..engine import Engine
        self.canonical = bool(canonical)
        # the register count plays no part in exact counting; the context only needs a valid one
        self.engine = Engine(device=device, log2m=max(4, min(20, int(log2m))), canonical=self.canonical)

    def describe(self, op, **kw):
        args = " ".join(f"{k}={v}" for k, v in kw.items())
        return f"hip-exact:{op} canonical={int(self.canonical)} {args}".strip()

    @staticmethod
    def _write(path, db):
        import json
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "w") as f:
            json.dump(db, f)
        os.replace(tmp, path)

    @staticmethod
    def _read(path):
        import json
        with open(path) as f:
            db = json.load(f)
        if "members" not in db:  # files of rounds 1 and 2: absolute paths, or names + directories, no sizes
            if "names" in db:
                dirs = list(db.get("dirs", []))
                db["members"] = [{"name": n, "dir": next((d for d in dirs if os.path.exists(os.path.join(d, n))), dirs[0] if dirs else ""),
                                  "size": None} for n in db["names"]]
            else:
                db["members"] = [{"name": os.path.basename(p), "dir": os.path.dirname(p), "size": None} for p in db["fastas"]]
            for key in ("names", "dirs", "fastas"):
                db.pop(key, None)
        return db

    @staticmethod
    def _find(member):
        """The file a member stands for: DANDD_GENOMEDIR first (a moved collection), then where it was seen.
        A candidate of another size is not that genome: StaleGenome when one was found, plain FileNotFoundError
        when there is no file of that name at all."""
        tried, replaced = [], False
        for d in [os.environ.get("DANDD_GENOMEDIR"), member.get("dir")]:
            if not d:
                continue
            cand = os.path.join(d, member["name"])
            if os.path.exists(cand):
                if member.get("size") is None or os.path.getsize(cand) == member["size"]:
                    return cand
                tried.append(f"{cand} ({os.path.getsize(cand)} bytes, recorded {member['size']})")
                replaced = True
            else:
                tried.append(cand)
        kind = StaleGenome if replaced else FileNotFoundError
        raise kind(f"{member['name']}: not found as recorded -- tried {tried} (set DANDD_GENOMEDIR to where the genomes are now)")

    def leaf(self, fasta, ks, out_paths):
        full = os.path.abspath(fasta)
        member = {"name": os.path.basename(full), "dir": os.path.dirname(full), "size": os.path.getsize(full)}
        for k, out in zip(ks, out_paths):
            self._write(out, {"k": int(k), "canonical": self.canonical, "members": [member]})

    def union(self, in_paths, out_path):
        parts = [self._read(p) for p in in_paths]
        by_name = {}
        for part in parts:
            for mem in part["members"]:
                seen = by_name.setdefault(mem["name"], mem)
                if seen is not mem and None not in (seen.get("size"), mem.get("size")) and seen["size"] != mem["size"]:
                    raise ValueError(f"two different genomes named {mem['name']}: {os.path.join(seen['dir'], seen['name'])} ({seen['size']} bytes) "
                                     f"and {os.path.join(mem['dir'], mem['name'])} ({mem['size']} bytes); DandD identifies genomes by base name")
        self._write(out_path, {"k": parts[0]["k"], "canonical": self.canonical, "members": [by_name[n] for n in sorted(by_name)]})

    def card(self, path):
        db = self._read(path)
        if db.get("distinct") is not None:
            # a database answers on its own (like a KMC database once its FASTAs are gone) -- unless a genome is
            # where it was and is no longer the file the count was made from: that raises StaleGenome, the caller
            # deletes the database or restores the file
            for m in db["members"]:
                try:
                    self._find(m)
                except StaleGenome:
                    raise
                except FileNotFoundError:
                    pass  # moved away altogether: nothing to compare with
            return float(db["distinct"])
        files = [self._find(m) for m in db["members"]]
        for m, f in zip(db["members"], files):
            if m.get("size") is None:
                m["size"] = os.path.getsize(f)
        db["distinct"] = int(self.engine.exact_count(files, db["k"]))
        self._write(path, db)
        return float(db["distinct"])

    def close(self):
        self.engine.close()


class HipBackend:
    """GPU backend: fused k-sweep leaf sketches, byte-max unions, Ertl-MLE cardinalities."""

    name = "hip"

    def __init__(self, log2m, canonical=True, device=0):
        from collections import OrderedDict
        from ..engine import Engine  # raises EngineError when the library or the GPU is missing
        self.log2m = int(log2m)
        self.canonical = bool(canonical)
        self.engine = Engine(device=device, log2m=self.log2m, canonical=self.canonical)
        # Registers of the sketch files this process wrote or read last, so that a tree does not read back from
        # the sketch directory what it stored a moment ago (at log2m 20 a 10-genome, 37-k tree wrote 370 MiB and
        # read 780 MiB of it again).  The FILES stay the contract (cache hits, other ranks, later runs); an entry is
        # only trusted while its file is still there with the size and modification time it had then.
        self._recent = OrderedDict()  # path -> (registers, k, (file size, mtime))
        self._recent_bytes = 0
        self._recent_limit = int(os.environ.get("DANDD_SKETCH_CACHE_MB", "1024")) << 20
        self._slab_buf = None
        self._dev = None                # (key, device address, capacity) of the leaf slab kept in HBM: _device_slab
        self.resident = False           # set by `dandd serve`: leaf_many leaves the slab it has just made on the device

    def new_command(self):
        """A resident server calls this between commands (deltatree.new_command): the registers kept in memory are keyed by the path
        AS GIVEN, and the next client may mean another file by the same relative path."""
        self._recent.clear()
        self._recent_bytes = 0

    def _remember(self, path, regs, k):
        if self._recent_limit <= 0:
            return
        old = self._recent.pop(path, None)
        if old is not None:
            self._recent_bytes -= old[0].nbytes
        try:
            st = os.stat(path)
        except OSError:
            return
        if regs.base is not None:
            # a row of a [n][K][m] slab: kept as a view it would pin the whole slab for as long as ANY of its rows is
            # cached, and DANDD_SKETCH_CACHE_MB would bound nothing (64 x 31 x 1 MiB: 2 GiB alive behind a 1 GiB limit)
            regs = regs.copy()
        self._recent[path] = (regs, int(k), (st.st_size, st.st_mtime_ns))
        self._recent_bytes += regs.nbytes
        while self._recent_bytes > self._recent_limit and self._recent:
            _, (r, _, _) = self._recent.popitem(last=False)
            self._recent_bytes -= r.nbytes

    def _store(self, path, regs, k):
        write_sketch_file(path, regs, self.log2m, k, self.canonical)
        self._remember(path, regs, k)

    def _load(self, path):
        """-> (registers, log2m, k, canonical) of a sketch file: from memory when this process handled it last."""
        hit = self._recent.get(path)
        if hit is not None:
            try:
                st = os.stat(path)
                same = (st.st_size, st.st_mtime_ns) == hit[2]
            except OSError:
                same = False
            if same:
                self._recent.move_to_end(path)
                return hit[0], self.log2m, hit[1], self.canonical
            self._recent_bytes -= hit[0].nbytes
            del self._recent[path]
        out = read_sketch_file(path)
        if out[1] == self.log2m:
            self._remember(path, out[0], out[2])
        return out

    def describe(self, op, **kw):
        """The string stored where the reference stores its shell command line."""
        args = " ".join(f"{k}={v}" for k, v in kw.items())
        return f"hip:{op} log2m={self.log2m} canonical={int(self.canonical)} {args}".strip()

    def leaf(self, fasta, ks, out_paths):
        ks = [int(k) for k in ks]
        if not ks:
            return
        kmin, kmax = min(ks), max(ks)
        regs = self.engine.sketch_fasta(fasta, kmin, kmax)  # ONE pass over the FASTA for all ks
        for k, out in zip(ks, out_paths):
            self._store(out, regs[k - kmin], k)

    def leaf_many(self, fastas, kmin, kmax, path_of):
        """Sketch MANY FASTAs over [kmin, kmax] through the ingestion pipeline (loader threads read and
        inflate ahead of the GPU) and store every (fasta, k) sketch at path_of(fasta_index, k).
        -> {path: cardinality} of every sketch stored, from ONE batched histogram + estimate launch (the reference
        asks `dashing card` for each file in a process of its own, lib/sketch_classes.py:306-321; the caller puts
        these where it would have cached those answers)."""
        regs = self.engine.sketch_files(list(fastas), kmin, kmax)
        n, K = len(fastas), kmax - kmin + 1
        est = self.engine.card_batch(regs.reshape(n * K, -1)).reshape(n, K) if n else None
        cards, jobs = {}, []
        for i in range(n):
            for k in range(kmin, kmax + 1):
                path = path_of(i, k)
                jobs.append((path, regs[i, k - kmin], k))
                cards[path] = float(est[i, k - kmin])
        # the files by a few threads (write(2) releases the GIL: 2 015 one-MiB sketches of 64 genomes at -r 20 were 0.21 s one
        # after the other), then the memory of what was written -- only the tail that fits it: copying 2 GB of rows to evict
        # half of them again was another 0.22 s of that `tree`
        def put(job):
            write_sketch_file(job[0], job[1], self.log2m, job[2], self.canonical)
        if len(jobs) * regs.shape[-1] >= (64 << 20):
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as pool:
                list(pool.map(put, jobs))
        else:
            for job in jobs:
                put(job)
        keep = max(0, self._recent_limit // max(1, regs.shape[-1]))
        for path, row, k in jobs[len(jobs) - min(len(jobs), keep):]:
            self._remember(path, row, k)
        if self.resident and n > 1:
            self._seed_device_slab([[path_of(i, k) for k in range(kmin, kmax + 1)] for i in range(n)], regs)
        return cards

    def _seed_device_slab(self, leaf_paths, regs):
        """A resident server that has just sketched a collection leaves its leaf slab in HBM (_device_slab's layout and key): the
        `progressive` / `kij` that follow find it there."""
        n, K = len(leaf_paths), len(leaf_paths[0])
        nbytes = (n * K) << self.log2m
        limit = int(os.environ.get("DANDD_DEVICE_CACHE_MB", "16384")) << 20
        if nbytes > limit or not hasattr(self.engine, "device_alloc"):
            return
        order = sorted(range(n), key=lambda i: leaf_paths[i][0])
        key = []
        for i in order:
            for p in leaf_paths[i]:
                st = os.stat(p)
                key.append((os.path.abspath(p), st.st_size, st.st_mtime_ns))
        if self._dev is None or self._dev[2] < nbytes:
            if self._dev is not None:
                self.engine.device_free(self._dev[1])
                self._dev = None
            ptr, cap = self.engine.device_alloc(nbytes), nbytes
        else:
            ptr, cap = self._dev[1], self._dev[2]
        self._dev = None
        per_leaf = K << self.log2m
        for rank, i in enumerate(order):
            self.engine.device_upload(ptr + rank * per_leaf, regs[i])
        self._dev = (tuple(key), ptr, cap)

    def union(self, in_paths, out_path):
        cold = [p for p in in_paths if p not in self._recent]
        if len(cold) >= 8 and len(cold) << self.log2m >= (32 << 20):
            # many inputs that are not in memory (the root of a 64-genome tree at -r 20): read side by side
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as pool:
                got = dict(zip(cold, pool.map(read_sketch_file, cold)))
            parts = [got[p] if p in got else self._load(p) for p in in_paths]
            for p in cold:
                if got[p][1] != self.log2m:
                    raise ValueError(f"{p}: log2m {got[p][1]} does not match the backend's {self.log2m}")
        else:
            parts = [self._load(p) for p in in_paths]
        k = parts[0][2]
        merged = self.engine.union([r for r, _, _, _ in parts])
        self._store(out_path, merged, k)

    def card(self, path):
        regs, log2m, _, _ = self._load(path)
        if log2m != self.log2m:
            raise ValueError(f"{path}: log2m {log2m} does not match the backend's {self.log2m}")
        return float(self.engine.card(regs))

    # ---- whole union schedules in one launch (no reference equivalent: the reference runs one
    # `dashing union` + one `dashing card` process per (set, k)) -----------------------------------
    def _leaf_slab(self, leaf_paths):
        """leaf_paths[n][K] -> uint8 [n][K][m].  Files this process did not handle last are read straight into their rows of
        the slab by a few threads (readinto releases the GIL): 64 x 37 one-MiB sketches were 0.43 s of a 1.3 s `kij` one file
        after the other through read_sketch_file, which also copies every one of them once more."""
        n, K = len(leaf_paths), len(leaf_paths[0])
        m = 1 << self.log2m
        # (the buffer of the last schedule is kept: a resident server's second `kij` does not fault 2 GB of fresh pages in again)
        if self._slab_buf is None or self._slab_buf.size < n * K * m:
            self._slab_buf = None
            self._slab_buf = np.empty(n * K * m, dtype=np.uint8)
        slab = self._slab_buf[: n * K * m].reshape(n, K, m)
        cold = []
        for i, row in enumerate(leaf_paths):
            for kk, p in enumerate(row):
                if p in self._recent:
                    slab[i, kk] = self._load(p)[0]
                else:
                    cold.append((i, kk, p))

        def fill(item):
            i, kk, p = item
            with open(p, "rb", buffering=0) as f:
                head = f.read(_HDR.size)
                if head[:8] == MAGIC and head[8] == self.log2m and f.readinto(memoryview(slab[i, kk])) == m and not f.read(1):
                    return
            slab[i, kk] = read_sketch_file(p)[0]      # (Dashing's container, or a file that is not what its name says: the reader reports it)

        if len(cold) * m >= (64 << 20):
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(16, (os.cpu_count() or 4))) as pool:
                list(pool.map(fill, cold))
        else:
            for item in cold:
                fill(item)
        return slab

    def _device_slab(self, leaf_paths):
        """The leaf slab [n][K][m] in HBM, kept between calls: a resident `dandd serve` answers a second `kij` or `progressive`
        over the same sketch files without reading 2 GB of them and copying them to the device again (64 genomes x 37 k at -r 20:
        0.13 s + 0.06 s of a 0.55 s command).  The copy is trusted only while every file still has the absolute path, size and
        modification time it had when it was loaded; DANDD_DEVICE_CACHE_MB bounds it (default 16384, 0 = off).
        -> (device address, permutation), or (None, None) (too large: the caller goes through host memory)"""
        n, K = len(leaf_paths), len(leaf_paths[0])
        nbytes = (n * K) << self.log2m
        limit = int(os.environ.get("DANDD_DEVICE_CACHE_MB", "16384")) << 20
        if nbytes > limit or not hasattr(self.engine, "device_alloc"):
            return None, None
        # (the slab is kept in the order of its rows' first paths, whatever order the caller lists the leaves in: `progressive`
        # names them in its first ordering's order, `kij` in the tree's -- the same files, one copy)
        order = sorted(range(n), key=lambda i: leaf_paths[i][0])
        perm = np.empty(n, dtype=np.int64)
        perm[order] = np.arange(n)                       # caller's leaf i is row perm[i] of the slab
        rows = [leaf_paths[i] for i in order]
        key = []
        for row in rows:
            for p in row:
                st = os.stat(p)
                key.append((os.path.abspath(p), st.st_size, st.st_mtime_ns))
        key = tuple(key)
        if self._dev is not None and self._dev[0] == key:
            return self._dev[1], perm
        slab = self._leaf_slab(rows)
        if self._dev is None or self._dev[2] < nbytes:
            if self._dev is not None:
                self.engine.device_free(self._dev[1])
                self._dev = None
            ptr, cap = self.engine.device_alloc(nbytes), nbytes
        else:
            ptr, cap = self._dev[1], self._dev[2]
        self._dev = None                       # (not trusted while it is being overwritten)
        self.engine.device_upload(ptr, slab)
        self._dev = (key, ptr, cap)
        return ptr, perm

    def pairwise_cards(self, leaf_paths):
        """|leaf_i U leaf_j| for all pairs and every k column: float64 [n][n][K]"""
        ptr, perm = self._device_slab(leaf_paths)
        if ptr is None:
            return self.engine.pairwise(self._leaf_slab(leaf_paths))
        table = self.engine.pairwise_device(ptr, len(leaf_paths), len(leaf_paths[0]))
        return table[np.ix_(perm, perm)]

    def progressive_cards(self, leaf_paths, orderings):
        """|union of the first j+1 leaves of ordering o| : float64 [o][n][K]"""
        ptr, perm = self._device_slab(leaf_paths)
        if ptr is None:
            return self.engine.progressive(self._leaf_slab(leaf_paths), orderings)
        ords = perm[np.asarray(orderings, dtype=np.int64).reshape(-1, len(leaf_paths))]
        return self.engine.progressive_device(ptr, len(leaf_paths), len(leaf_paths[0]), ords)

    def close(self):
        if self._dev is not None:
            self.engine.device_free(self._dev[1])
            self._dev = None
        self.engine.close()


if __name__ == "__main__":
    import sys
    sys.exit(convert_main(sys.argv[1:]))
