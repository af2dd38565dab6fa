"""Sketch naming/caching helpers of the host layer (dandd_amd/host/store.py) against each other: the
fast per-k derivation used in the hot Python loops must produce exactly what the reference-shaped
constructor produces (names, directories, catalog entries)."""
import os

import numpy as np

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
    from dandd_amd.host import backend
    rng = np.random.default_rng(3)
    regs = rng.integers(0, 40, size=1 << 12, dtype=np.uint8)
    native, exported, back = (str(tmp_path / n) for n in ("a.hll", "a.dashing.hll", "b.hll"))
    backend.write_sketch_file(native, regs, 12, 21, True)
    assert backend.convert_main(["export", native, exported]) == 0
    raw = gzip.open(exported, "rb").read()
    assert len(raw) == 32 + (1 << 12) and raw[20:24] == (12).to_bytes(4, "little")
    got, np_, k_name, _ = backend.read_sketch_file(exported)
    assert np_ == 12 and k_name == 0 and np.array_equal(got, regs)    # k is not in Dashing's container (nor in this name)
    assert backend.convert_main(["import", exported, back, "21"]) == 0
    r2, log2m, k, canon = backend.read_sketch_file(back)
    assert (log2m, k, canon) == (12, 21, True) and np.array_equal(r2, regs)


def test_dashing_container_against_a_byte_level_fixture(tmp_path, monkeypatch):
    """Row f2 (Dashing's `.hll` container) against bytes this package did not write: tests/golden/hll/recalled_p10*.hll
    were built field by field by tests/golden/make_hll_fixture.py (no import of dandd_amd).  The reader takes the plain
    and the gzip form and returns the fixture's registers; the writer reproduces the plain fixture byte for byte and a
    gzip stream that inflates to it.  (The LAYOUT stays a recollection -- POLICIES.md P9; this pins it against drift.)"""
    import gzip
    from dandd_amd.host import backend
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hll")
    want = np.frombuffer(open(os.path.join(gold, "recalled_p10.registers"), "rb").read(), dtype=np.uint8)
    plain = open(os.path.join(gold, "recalled_p10.hll"), "rb").read()
    assert len(plain) == 32 + 1024 and plain[32:] == want.tobytes()
    for name in ("recalled_p10.hll", "recalled_p10.gz.hll"):
        # the file NAME carries k (Dashing's container does not): give the fixture a DandD leaf name
        leaf = tmp_path / f"g0.fasta.w.21.spacing.10.{'gz' if 'gz' in name else 'plain'}.hll"
        leaf.write_bytes(open(os.path.join(gold, name), "rb").read())
        regs, log2m, k, canon = backend.read_sketch_file(str(leaf))
        assert (log2m, canon) == (10, True) and np.array_equal(regs, want)
    out = tmp_path / "written.w.21.spacing.10.hll"
    backend.write_sketch_file(str(out), want, 10, 21, True, fmt="dashing-plain")
    assert out.read_bytes() == plain
    backend.write_sketch_file(str(out), want, 10, 21, True, fmt="dashing")
    assert out.read_bytes()[:2] == b"\x1f\x8b" and gzip.decompress(out.read_bytes()) == plain


def test_exact_databases_identify_genomes_by_name_and_size(tmp_path, monkeypatch):
    """`--exact` databases (the KMC stand-in): a genome is its base name + size.  The same file seen twice is one
    member, two different files of one name are refused, a replaced file invalidates the cached count, a moved
    collection is found through DANDD_GENOMEDIR, and files of rounds 1-2 (names + dirs, no sizes) still load."""
    import json
    from dandd_amd.host.backend import HipExactBackend
    be = object.__new__(HipExactBackend)      # no GPU here: everything below is host logic
    be.canonical = True
    counted = []

    class FakeEngine:
        def exact_count(self, files, k):
            counted.append(list(files))
            return sum(os.path.getsize(f) for f in files)
    be.engine = FakeEngine()
    d1, d2 = tmp_path / "a", tmp_path / "b"
    d1.mkdir(), d2.mkdir()
    (d1 / "g.fa").write_text(">x\nACGT\n")
    (d1 / "h.fa").write_text(">y\nACGTACGT\n")
    (d2 / "g.fa").write_text(">x\nACGTTTTTTT\n")     # another genome under the same name
    monkeypatch.delenv("DANDD_GENOMEDIR", raising=False)
    db = lambda n: str(tmp_path / n)
    be.leaf(str(d1 / "g.fa"), [7], [db("g1")])
    be.leaf(str(d1 / "h.fa"), [7], [db("h1")])
    be.leaf(str(d2 / "g.fa"), [7], [db("g2")])
    be.union([db("g1"), db("h1"), db("g1")], db("u"))
    assert [m["name"] for m in json.load(open(db("u")))["members"]] == ["g.fa", "h.fa"]
    with pytest.raises(ValueError, match="two different genomes named g.fa"):
        be.union([db("g1"), db("g2")], db("bad"))
    assert be.card(db("u")) == float(os.path.getsize(d1 / "g.fa") + os.path.getsize(d1 / "h.fa"))
    assert be.card(db("u")) == be.card(db("u")) and len(counted) == 1          # answered from the database afterwards
    (d1 / "g.fa").write_text(">x\nACGTACGTACGTAAAA\n")                         # replaced: the cached count is stale
    from dandd_amd.host.backend import StaleGenome
    with pytest.raises(StaleGenome):
        be.card(db("u"))
    # a database that holds its count answers on its own once its genomes are gone altogether (round-3 advice: this
    # used to raise, the two cases were told apart by a word both messages contain)
    be.leaf(str(d1 / "h.fa"), [9], [db("h9")])
    want = be.card(db("h9"))
    os.rename(d1 / "h.fa", tmp_path / "h.fa.away")
    n_counted = len(counted)
    assert be.card(db("h9")) == want and len(counted) == n_counted
    os.rename(tmp_path / "h.fa.away", d1 / "h.fa")
    # a moved collection
    moved = tmp_path / "moved"
    moved.mkdir()
    (moved / "h.fa").write_text(">y\nACGTACGT\n")
    os.remove(d1 / "h.fa")
    with pytest.raises(FileNotFoundError):
        be.card(db("h1"))
    monkeypatch.setenv("DANDD_GENOMEDIR", str(moved))
    assert be.card(db("h1")) == float(os.path.getsize(moved / "h.fa"))
    # a round-2 file
    json.dump({"k": 7, "canonical": True, "names": ["h.fa"], "dirs": [str(moved)]}, open(db("old"), "w"))
    assert be.card(db("old")) == float(os.path.getsize(moved / "h.fa"))


def test_rowtable_csv_is_the_csv_modules(tmp_path):
    """RowTable (columns instead of a list of dicts: the 62 496 Jaccard rows of a 64-genome `kij`) is written by hand; the bytes must
    be csv.DictWriter's for the same rows -- floats of every shape, ints, None, strings that need quoting."""
    import csv
    import math
    from dandd_amd.host.deltatree import RowTable, write_listdict_to_csv
    vals = [0.0, -0.0, 1.0, 1e16, 1.5e-7, 123456789.123456789, 0.1 + 0.2, float("inf"), float("nan"), 3, -7, None, "plain", "with,comma",
            'with "quote"', "line\nbreak", "cr\rhere", "", " spaced ", "tab\there"]
    t = RowTable(("b", "a", "fastas", "k"))
    for i, v in enumerate(vals * 20):      # (400 rows of 20 objects: the memoised path; `fastas` goes last as for lists of dicts)
        t.cols["a"].append(v)
        t.cols["b"].append(f"/some/path, with comma/g{i % 3}.fasta" if i % 2 else 1.25 * i)
        t.cols["fastas"].append("x|y")
        t.cols["k"].append(i)
    write_listdict_to_csv(str(tmp_path / "cols.csv"), t)
    rows = list(t)
    assert len(rows) == len(t) == 20 * len(vals) and rows[3] == t[3]
    with open(tmp_path / "dicts.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["a", "b", "k", "fastas"])
        w.writeheader()
        w.writerows(rows)
    assert (tmp_path / "cols.csv").read_bytes() == (tmp_path / "dicts.csv").read_bytes()
    write_listdict_to_csv(str(tmp_path / "list.csv"), rows)            # (the list-of-dicts path: the same file again)
    assert (tmp_path / "list.csv").read_bytes() == (tmp_path / "dicts.csv").read_bytes()
    empty = RowTable(("x", "y"))
    write_listdict_to_csv(str(tmp_path / "empty.csv"), empty)
    assert (tmp_path / "empty.csv").read_bytes() == b"x,y\r\n" and not math.isnan(len(empty))
