"""Pickle interchange with the reference (dandd_amd/host/compat.py).

tests/golden/ref_tree_hll.pickle was written by the REFERENCE's own `dandd tree` (tests/golden/make_golden.py
ran /root/reference/lib/dandd in the build container; the file is data -- pickled objects -- not source).  Its
GLOBALs are huffman_dandd.DeltaSpider, sketch_classes.DashSketchObj, species_specifics.SpeciesSpecifics ...;
`dandd progressive` and `dandd kij` of this package must run from it (/root/reference/lib/dandd_cmd.py:66,108)
and write the rows the reference wrote from the same tree, and the pickles this package writes must carry the
same GLOBAL names so the exchange works in the other direction too."""
import json
import os
import pickle
import shutil

import pytest

import hostcheck

FIXTURE_ROOT = "/tmp/dandd_gold_fixture_hll"  # the absolute paths inside the reference-written pickle
REF_GLOBALS = {("huffman_dandd", "DeltaSpider"), ("huffman_dandd", "DeltaTreeNode"), ("sketch_classes", "SketchFilePath"),
               ("sketch_classes", "DashSketchObj"), ("species_specifics", "SpeciesSpecifics")}


def globals_of(path):
    """(module, name) of every class a pickle refers to: what the unpickler is asked to resolve."""
    from dandd_amd.host import compat
    found = set()

    class Recording(compat._Unpickler):
        def find_class(self, module, name):
            found.add((module, name))
            return super().find_class(module, name)

    with open(path, "rb") as f:
        Recording(f).load()
    return found


@pytest.fixture
def ref_world():
    from dandd_amd.host import deltatree
    shutil.rmtree(FIXTURE_ROOT, ignore_errors=True)
    shutil.copytree(os.path.join(hostcheck.GOLD, "fasta"), os.path.join(FIXTURE_ROOT, "data"))
    sk = os.path.join(FIXTURE_ROOT, "t1", "sketchdb")
    os.makedirs(sk)
    with open(os.path.join(sk, "gold_5_orderings.pickle"), "wb") as f:
        pickle.dump({(0, 1, 2, 3, 4), (4, 2, 0, 3, 1), (1, 3, 4, 0, 2)}, f)
    deltatree.set_backend_factory(lambda r, c: hostcheck.OracleBackend(r, c))
    yield os.path.join(hostcheck.GOLD, "ref_tree_hll.pickle")
    deltatree.set_backend_factory(None)
    shutil.rmtree(FIXTURE_ROOT, ignore_errors=True)


def test_fixture_really_is_a_reference_pickle():
    found = globals_of(os.path.join(hostcheck.GOLD, "ref_tree_hll.pickle"))
    assert REF_GLOBALS <= found, found
    assert not any(mod.startswith("dandd_amd") for mod, _ in found)
    with pytest.raises(ModuleNotFoundError):  # a plain pickle.load needs the reference on sys.path
        with open(os.path.join(hostcheck.GOLD, "ref_tree_hll.pickle"), "rb") as f:
            pickle.load(f)


def test_progressive_and_kij_run_from_a_reference_written_tree(ref_world, tmp_path):
    from dandd_amd.host import cli, compat, deltatree, store
    tree = compat.load_tree(ref_world)
    assert type(tree) is deltatree.DeltaSpider and type(tree.speciesinfo) is store.Catalog
    assert type(tree.root.ksketches[tree.root.bestk]) is deltatree.DashSketchObj
    with open(os.path.join(hostcheck.GOLD, "ref_hll.json")) as f:
        gold = json.load(f)["scenarios"]
    assert float(tree.delta) == float(gold["tree_spider_k10"][0]["delta"])
    o2 = str(tmp_path / "p1")
    os.makedirs(o2)
    cli.main(["progressive", "-d", ref_world, "-o", o2, "--ksweep", "--mink", "8", "--maxk", "14"])
    o3 = str(tmp_path / "k1")
    os.makedirs(o3)
    cli.main(["kij", "-d", ref_world, "-o", o3, "--jaccard", "--mink", "8", "--maxk", "12"])
    got = {"progressive_ksweep_8_14": hostcheck.read_rows(os.path.join(o2, "gold_progu0_5_dashing.csv")),
           "progressive_ksweep_8_14_summary": hostcheck.read_rows(os.path.join(o2, "gold_progu0_5_dashingsummary.csv")),
           "kij": hostcheck.read_rows(os.path.join(o3, "gold_5_dashing.kij.csv")),
           "kij_jaccard_8_12": hostcheck.read_rows(os.path.join(o3, "gold_5_dashing.j.csv"))}
    diffs = hostcheck.compare(got, {k: gold[k] for k in got})
    assert not diffs, "\n".join(diffs[:30])
    # `progressive` saved the tree again: that file carries the reference's names too
    again = os.path.join(o2, "gold_progu0_5_dashing_dtree.pickle")
    assert REF_GLOBALS <= globals_of(again)


def test_pickles_written_here_carry_the_reference_names(tmp_path):
    from dandd_amd.host import cli, compat, deltatree
    deltatree.set_backend_factory(lambda r, c: hostcheck.ExactBackend(r, c))
    try:
        data = str(tmp_path / "data")
        shutil.copytree(os.path.join(hostcheck.GOLD, "fasta"), data)
        out = str(tmp_path / "o")
        cli.main(["tree", "-d", data, "-o", out, "-s", "t", "-k", "10", "-r", "12", "-n", "2"])
        path = os.path.join(out, "t_5_dashing_dtree.pickle")
        found = globals_of(path)
        assert {("huffman_dandd", "DeltaTree"), ("huffman_dandd", "DeltaTreeNode"), ("sketch_classes", "SketchFilePath"),
                ("sketch_classes", "DashSketchObj"), ("species_specifics", "SpeciesSpecifics")} <= found
        assert not any(mod.startswith("dandd_amd") for mod, _ in found), found
        # the names were only swapped in for the dump: the classes are themselves again, and load round-trips
        assert deltatree.DeltaTree.__module__ == "dandd_amd.host.deltatree" and "huffman_dandd" not in __import__("sys").modules
        back = compat.load_tree(path)
        assert back.delta == pytest.approx(compat.load_tree(path).delta) and type(back) is deltatree.DeltaTree
        assert [n.node_title for n in back._dt] == [n.node_title for n in compat.load_tree(path)._dt]
    finally:
        deltatree.set_backend_factory(None)
