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


def write_sketch_file(path, regs, log2m, k, canonical):
    """One HLL sketch on disk: 12-byte header + 2^log2m register bytes."""
    regs = np.ascontiguousarray(regs, dtype=np.uint8)
    if regs.size != (1 << log2m):
        raise ValueError("register count does not match log2m")
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(_HDR.pack(MAGIC, log2m, k, 1 if canonical else 0, 0))
        f.write(regs.tobytes())
    os.replace(tmp, path)


def read_sketch_file(path):
    with open(path, "rb") as f:
        raw = f.read()
    if len(raw) < _HDR.size:
        raise ValueError(f"{path}: truncated sketch file")
    magic, log2m, k, canonical, _ = _HDR.unpack_from(raw)
    if magic != MAGIC:
        raise ValueError(f"{path}: not a dandd_amd sketch file")
    regs = np.frombuffer(raw, dtype=np.uint8, offset=_HDR.size)
    if regs.size != (1 << log2m):
        raise ValueError(f"{path}: expected {1 << log2m} registers, found {regs.size}")
    return regs, log2m, k, bool(canonical)


class HipExactBackend:
    """`--exact`: the KMC branch of the reference (lib/sketch_classes.py:377-465), which counts
    distinct canonical k-mers exactly.  A "database" here is a small JSON file naming the FASTAs it
    covers; the cardinality is computed on the GPU (sort + distinct, dd_exact_count).  The
    reference's own KMC branch recurses forever at this commit; this one works."""

    name = "hip-exact"

    def __init__(self, log2m=20, canonical=True, device=0):
        from ..engine import Engine
        self.canonical = bool(canonical)
        self.engine = Engine(device=device, log2m=14, canonical=self.canonical)

    def describe(self, op, **kw):
        args = " ".join(f"{k}={v}" for k, v in kw.items())
        return f"hip-exact:{op} canonical={int(self.canonical)} {args}".strip()

    @staticmethod
    def _write(path, k, fastas):
        import json
        tmp = path + ".tmp"
        with open(tmp, "w") as f:
            json.dump({"k": int(k), "fastas": sorted(set(fastas))}, f)
        os.replace(tmp, path)

    @staticmethod
    def _read(path):
        import json
        with open(path) as f:
            return json.load(f)

    def leaf(self, fasta, ks, out_paths):
        for k, out in zip(ks, out_paths):
            self._write(out, k, [os.path.abspath(fasta)])

    def union(self, in_paths, out_path):
        parts = [self._read(p) for p in in_paths]
        self._write(out_path, parts[0]["k"], [f for p in parts for f in p["fastas"]])

    def card(self, path):
        db = self._read(path)
        return float(self.engine.exact_count(db["fastas"], db["k"]))

    def close(self):
        self.engine.close()


class HipBackend:
    """GPU backend: fused k-sweep leaf sketches, byte-max unions, Ertl-MLE cardinalities."""

    name = "hip"

    def __init__(self, log2m, canonical=True, device=0):
        from ..engine import Engine  # raises EngineError when the library or the GPU is missing
        self.log2m = int(log2m)
        self.canonical = bool(canonical)
        self.engine = Engine(device=device, log2m=self.log2m, canonical=self.canonical)

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
            write_sketch_file(out, regs[k - kmin], self.log2m, k, self.canonical)

    def leaf_many(self, fastas, kmin, kmax, path_of):
        """Sketch MANY FASTAs over [kmin, kmax] through the ingestion pipeline (loader threads read and
        inflate ahead of the GPU) and store every (fasta, k) sketch at path_of(fasta_index, k)."""
        regs = self.engine.sketch_files(list(fastas), kmin, kmax)
        for i in range(len(fastas)):
            for k in range(kmin, kmax + 1):
                write_sketch_file(path_of(i, k), regs[i, k - kmin], self.log2m, k, self.canonical)

    def union(self, in_paths, out_path):
        parts = [read_sketch_file(p) for p in in_paths]
        k = parts[0][2]
        merged = self.engine.union([r for r, _, _, _ in parts])
        write_sketch_file(out_path, merged, self.log2m, k, self.canonical)

    def card(self, path):
        regs, log2m, _, _ = read_sketch_file(path)
        if log2m != self.log2m:
            raise ValueError(f"{path}: log2m {log2m} does not match the backend's {self.log2m}")
        return float(self.engine.card(regs))

    # ---- whole union schedules in one launch (no reference equivalent: the reference runs one
    # `dashing union` + one `dashing card` process per (set, k)) -----------------------------------
    def _leaf_slab(self, leaf_paths):
        """leaf_paths[n][K] -> uint8 [n][K][m]"""
        n, K = len(leaf_paths), len(leaf_paths[0])
        slab = np.empty((n, K, 1 << self.log2m), dtype=np.uint8)
        for i, row in enumerate(leaf_paths):
            for kk, p in enumerate(row):
                slab[i, kk] = read_sketch_file(p)[0]
        return slab

    def pairwise_cards(self, leaf_paths):
        """|leaf_i U leaf_j| for all pairs and every k column: float64 [n][n][K]"""
        return self.engine.pairwise(self._leaf_slab(leaf_paths))

    def progressive_cards(self, leaf_paths, orderings):
        """|union of the first j+1 leaves of ordering o| : float64 [o][n][K]"""
        return self.engine.progressive(self._leaf_slab(leaf_paths), orderings)

    def close(self):
        self.engine.close()
