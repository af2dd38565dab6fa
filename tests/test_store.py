"""Sketch naming/caching helpers of the host layer (dandd_amd/host/store.py) against each other: the
fast per-k derivation used in the hot Python loops must produce exactly what the reference-shaped
constructor produces (names, directories, catalog entries)."""
import os

import pytest

from dandd_amd.host import store


def _catalog(tmp_path, tool="dashing"):
    gdir = tmp_path / "genomes"
    gdir.mkdir(parents=True)
    files = []
    for i, body in enumerate([b">a\nACGTACGTAC\n", b">b\nTTGACCAGT\n", b">c\nGGGATTTACCA\n"]):
        f = gdir / f"g{i}.fasta"
        f.write_bytes(body)
        files.append(str(f))
    cat = store.Catalog("t", str(gdir), str(tmp_path / "sk"), 12, tool)
    return cat, files


@pytest.mark.parametrize("tool", ["dashing", "kmc"])
@pytest.mark.parametrize("canon", [True, False])
def test_at_k_equals_constructor(tmp_path, tool, canon):
    exp = {"tool": tool, "registers": 14, "canonicalize": canon, "safety": False}
    for nfiles in (1, 2, 3):
        cat_a, files = _catalog(tmp_path / f"a{nfiles}", tool)
        cat_b, _ = _catalog(tmp_path / f"b{nfiles}", tool)
        cat_b.sketchdir = cat_a.sketchdir  # same root, so full paths are comparable
        subset = files[:nfiles]
        other = [f.replace(f"a{nfiles}", f"b{nfiles}") for f in subset]
        for fa, fb in zip(subset, other):  # leaves first, as in every tree: a union's name needs their digests
            store.SketchPath([fa], 0, cat_a, exp)
            store.SketchPath([fb], 0, cat_b, exp)
        tmpl = store.SketchPath(subset, 0, cat_a, exp)
        for k in (1, 9, 12, 31, 40):
            fast = tmpl.at_k(k, cat_a, exp)
            slow = store.SketchPath(other, k, cat_b, exp)
            for attr in ("files", "ngen", "dir", "base", "relative", "full"):
                assert getattr(fast, attr) == getattr(slow, attr), (attr, k, nfiles)
            assert fast.full == tmpl.with_k(k)
            assert os.path.isdir(fast.dir)
            assert cat_a.sketchinfo[fast.base] == cat_b.sketchinfo[slow.base]


def test_sketch_exists_remembers_only_positive_answers(tmp_path):
    p = str(tmp_path / "x.hll")
    assert not store.sketch_exists(p)
    open(p, "wb").close()
    assert not store.sketch_exists(p)          # empty file: not a sketch
    with open(p, "wb") as f:
        f.write(b"abc")
    assert store.sketch_exists(p)              # appears later: seen
    os.remove(p)
    assert store.sketch_exists(p)              # remembered (sketches are never removed behind the host layer's back)
    store.forget_sketch(p)                     # ... and when the host layer removes one itself it says so
    assert not store.sketch_exists(p)


def test_union_name_is_the_hex_sum_rule(tmp_path):
    exp = {"tool": "dashing", "registers": 14, "canonicalize": True, "safety": False}
    cat, files = _catalog(tmp_path)
    a, b = (store.file_digest(f) for f in files[:2])
    for f in files[:2]:
        store.SketchPath([f], 12, cat, exp)
    sp = store.SketchPath(files[:2], 12, cat, exp)
    assert sp.base == f"{hex(int(a, 16) + int(b, 16))[:15]}_14n2k12"
    assert store.SketchPath(files[:1], 12, cat, exp).base == "g0.fasta.w.12.spacing.14"


def test_dashing_container_round_trip(tmp_path):
    """The (unverified, recalled) Dashing .hll container: what is written reads back, the header is the
    32 bytes the module documents, and the native <-> Dashing conversion keeps the registers."""
    import gzip
    import numpy as np
    from dandd_amd.host import backend, dashing_hll
    rng = np.random.default_rng(3)
    regs = rng.integers(0, 40, size=1 << 12, dtype=np.uint8)
    native, exported, back = (str(tmp_path / n) for n in ("a.hll", "a.dashing.hll", "b.hll"))
    backend.write_sketch_file(native, regs, 12, 21, True)
    assert dashing_hll.main(["export", native, exported]) == 0
    raw = gzip.open(exported, "rb").read()
    assert len(raw) == 32 + (1 << 12) and raw[20:24] == (12).to_bytes(4, "little")
    got, np_, est = dashing_hll.read_dashing_hll(exported)
    assert np_ == 12 and est is None and np.array_equal(got, regs)
    assert dashing_hll.main(["import", exported, back, "21"]) == 0
    r2, log2m, k, canon = backend.read_sketch_file(back)
    assert (log2m, k, canon) == (12, 21, True) and np.array_equal(r2, regs)
