"""The host layer (dandd_amd.host: tree / progressive / kij) must write the same CSV rows as the
reference's own Python did when tests/golden/make_golden.py ran it (goldens in tests/golden/ref_*.json).

CPU: checker backends (oracle HLL, exact counter) behind the backend contract -> pins the
orchestration (hill-climb, tree shapes, naming, progressive prefixes, KIJ / Jaccard formulas).
GPU: the real HipBackend -> pins the whole drop-in path against the same goldens.
"""
import json
import os

import pytest

import hostcheck


def _golden(name):
    with open(os.path.join(hostcheck.GOLD, name)) as f:
        return json.load(f)


@pytest.fixture
def host():
    from dandd_amd.host import deltatree
    yield deltatree
    deltatree.set_backend_factory(None)


@pytest.mark.parametrize("which", ["hll", "exact", "hll+schedules"])
def test_cli_rows_match_reference_with_checker_backend(host, tmp_path, which):
    """"hll+schedules": the oracle behind the GPU backend's batch entry points, i.e. every prefetch path of the host
    layer (leaf windows, whole-schedule union cardinalities) on the CPU, against the same reference rows."""
    gold = _golden("ref_exact.json" if which == "exact" else "ref_hll.json")
    factory = {"hll": hostcheck.OracleBackend, "exact": hostcheck.ExactBackend, "hll+schedules": hostcheck.ScheduleBackend}[which]
    host.set_backend_factory(lambda registers, canon: factory(registers, canon))
    got = hostcheck.run_scenarios(str(tmp_path), gold["registers"])
    diffs = hostcheck.compare(got, gold["scenarios"])
    assert not diffs, "\n".join(diffs[:40])


@pytest.mark.gpu
def test_cli_rows_match_reference_on_gpu(host, tmp_path, torch_cuda):
    gold = _golden("ref_hll.json")
    host.set_backend_factory(None)  # the product default: HipBackend on cuda:0
    got = hostcheck.run_scenarios(str(tmp_path), gold["registers"])
    diffs = hostcheck.compare(got, gold["scenarios"])
    assert not diffs, "\n".join(diffs[:40])
    # progressive prefixes and kij pairs went through the batched GPU schedules: their 2-way union
    # sketches were never materialised as files, yet their cardinalities are in the cache
    import glob
    import pickle
    sk = os.path.join(str(tmp_path), "t1", "sketchdb")
    assert glob.glob(os.path.join(sk, "ngen5", "k*", "*.hll"))          # tree root: file-based
    # (k = 8: prefetched for the pairs of `kij` and the prefixes of `progressive`; ks 9..12 are also walked by the
    # --step 2 scenario, whose first prefix IS a pair and goes through the file-based path)
    assert not glob.glob(os.path.join(sk, "ngen2", "k8", "*.hll"))
    with open(os.path.join(sk, "gold_dashing_cardinalities.pickle"), "rb") as f:
        cards = pickle.load(f)
    assert sum(1 for p in cards if os.sep + "ngen2" + os.sep in p) >= 10 * 5


@pytest.mark.gpu
def test_cli_exact_on_gpu_matches_exact_goldens(host, tmp_path, torch_cuda):
    """`tree --exact` (GPU sort+distinct behind the KMC branch) gives the cards/deltas/argmax-k that the
    reference's orchestration produced over an exact counter (ref_exact.json); only names differ."""
    import shutil
    from dandd_amd.host import cli
    gold = _golden("ref_exact.json")["scenarios"]["tree_spider_k10"]
    host.set_backend_factory(None)
    data = os.path.join(str(tmp_path), "data")
    shutil.copytree(os.path.join(hostcheck.GOLD, "fasta"), data)
    out = os.path.join(str(tmp_path), "o")
    cli.main(["tree", "-d", data, "-o", out, "-s", "gold", "-k", "10", "--exact"])
    rows = hostcheck.read_rows(os.path.join(out, "gold_5_kmc_deltas.csv"))
    assert len(rows) == len(gold)
    for r, w in zip(rows, gold):
        for key in ("card", "delta", "k", "ngen", "title", "fastas"):
            assert hostcheck.same_cell(r[key], w[key]), (key, r[key], w[key])
        assert r["sketchloc"].endswith(f"_k{w['k']}") or "n5k" in r["sketchloc"]


def test_second_run_is_served_from_cache(host, tmp_path):
    """All caches warm -> no backend call at all (the reference launches zero subprocesses)."""
    gold = _golden("ref_hll.json")
    calls = []

    class Counting(hostcheck.OracleBackend):
        def leaf(self, *a):
            calls.append("leaf")
            return super().leaf(*a)

        def union(self, *a):
            calls.append("union")
            return super().union(*a)

        def card(self, *a):
            calls.append("card")
            return super().card(*a)

    host.set_backend_factory(lambda r, c: Counting(r, c))
    from dandd_amd.host import cli
    import shutil
    data = os.path.join(str(tmp_path), "data")
    shutil.copytree(os.path.join(hostcheck.GOLD, "fasta"), data)
    out = os.path.join(str(tmp_path), "o")
    args = ["tree", "-d", data, "-o", out, "-s", "gold", "-k", "10", "-r", str(gold["registers"])]
    cli.main(args)
    first = len(calls)
    assert first > 0
    del calls[:]
    cli.main(args)
    assert calls == []
    rows = hostcheck.read_rows(os.path.join(out, "gold_5_dashing_deltas.csv"))
    assert not hostcheck.compare({"tree_spider_k10": rows}, {"tree_spider_k10": gold["scenarios"]["tree_spider_k10"]})


def test_hill_climb_window_extensions_are_batched_over_leaves(host, tmp_path):
    """A backend with the batch entry point (the GPU one has it) gets the leaves' hill-climbs served
    by a few whole-tree batches instead of one call per (leaf, step) -- and the rows do not change."""
    gold = _golden("ref_hll.json")
    log = {"single": 0, "batches": []}

    class Batched(hostcheck.OracleBackend):
        def leaf(self, fasta, ks, outs):
            log["single"] += 1
            return super().leaf(fasta, ks, outs)

        def leaf_many(self, fastas, kmin, kmax, path_of):
            log["batches"].append((len(fastas), kmin, kmax))
            for i, f in enumerate(fastas):
                ks = list(range(kmin, kmax + 1))
                hostcheck.OracleBackend.leaf(self, f, ks, [path_of(i, k) for k in ks])

    from dandd_amd.host import cli
    import shutil

    def run(factory, name):
        host.set_backend_factory(factory)
        data = os.path.join(str(tmp_path), name, "data")
        shutil.copytree(os.path.join(hostcheck.GOLD, "fasta"), data)
        out = os.path.join(str(tmp_path), name, "o")
        # start far below the optimum: every leaf's search leaves the pre-sketched window (kstart +- 3)
        cli.main(["tree", "-d", data, "-o", out, "-s", "gold", "-k", "2", "-r", str(gold["registers"])])
        return hostcheck.read_rows(os.path.join(out, "gold_5_dashing_deltas.csv"))

    class Plain(hostcheck.OracleBackend):  # no batch entry point: one backend call per (leaf, step)
        def leaf(self, fasta, ks, outs):
            log["plain"] = log.get("plain", 0) + 1
            return super().leaf(fasta, ks, outs)

    plain = run(lambda r, c: Plain(r, c), "plain")
    batched = run(lambda r, c: Batched(r, c), "batched")
    strip = lambda rows: [{k: v for k, v in r.items() if k != "command"} for r in rows]  # noqa: E731
    assert not hostcheck.compare({"t": strip(batched)}, {"t": strip(plain)})
    assert len(log["batches"]) >= 2 and all(b[0] >= 2 for b in log["batches"]), log   # window + extensions, many leaves each
    # climbs ride on the batches: at most one straggler call per leaf (a leaf whose search alone goes
    # further), against several per leaf without batching
    assert log["single"] <= 5 and len(log["batches"]) + log["single"] < log["plain"] / 2, log


def test_example_readme_workflow_subset_tree_reuses_leaf_sketches(host, tmp_path):
    """The walk of /root/reference/example/README.md with its own flag spellings (--datadir, --fastas,
    --tag, --nchildren, -k, --dtree, --norder as an abbreviation): a later `tree --fastas subset.txt` over
    the same sketch directory sketches no leaf again -- only the subset's unions are new (README :37)."""
    calls = {"leaf": 0, "union": 0}

    class Counting(hostcheck.OracleBackend):
        def leaf(self, *a):
            calls["leaf"] += 1
            return super().leaf(*a)

        def union(self, *a):
            calls["union"] += 1
            return super().union(*a)

    host.set_backend_factory(lambda r, c: Counting(r, c))
    from dandd_amd.host import cli
    import shutil
    data = os.path.join(str(tmp_path), "fish")
    shutil.copytree(os.path.join(hostcheck.GOLD, "fasta"), data)
    out = os.path.join(str(tmp_path), "out")
    sk = os.path.join(str(tmp_path), "sketchdb")
    cli.main(["tree", "--datadir", data, "--tag", "fish-mito", "-o", out, "-c", sk, "-r", "12"])
    assert calls["leaf"] > 0
    cli.main(["tree", "--datadir", data, "--tag", "fish-mito2", "--nchildren", "2", "-k", "12", "-o", out, "-c", sk, "-r", "12"])
    listing = os.path.join(str(tmp_path), "fish_subset.txt")
    with open(listing, "w") as f:
        for name in sorted(os.listdir(data))[:3]:
            f.write(os.path.join(data, name) + "\n")
    before = dict(calls)
    cli.main(["tree", "--fastas", listing, "--tag", "small_fish", "-k", "12", "-o", out, "-c", sk, "-r", "12"])
    assert calls["leaf"] == before["leaf"], "leaf sketches of the subset were already in the sketch directory"
    assert calls["union"] > before["union"]
    pickle_path = os.path.join(out, "small_fish_3_dashing_dtree.pickle")
    assert os.path.exists(pickle_path)
    cli.main(["progressive", "--dtree", pickle_path, "--norder", "4", "-o", out])
    assert os.path.exists(os.path.join(out, "small_fish_progu4_3_dashing.csv"))
    cli.main(["kij", "--dtree", pickle_path, "-o", out])
    assert os.path.exists(os.path.join(out, "small_fish_3_dashing.kij.csv"))


def test_tree_shapes_for_nchildren(host, tmp_path):
    """Shapes the reference builds for (N, nchildren) -- SURVEY.md section 4.3 probe facts."""
    import shutil
    host.set_backend_factory(lambda r, c: hostcheck.ExactBackend(r, c))
    src = os.path.join(hostcheck.GOLD, "fasta")

    def shape(n, nchildren):
        d = os.path.join(str(tmp_path), f"d{n}_{nchildren}")
        os.makedirs(d)
        for i in range(n):
            shutil.copy(os.path.join(src, f"g{i % 5}.fasta"), os.path.join(d, f"g{i}.fasta"))
            with open(os.path.join(d, f"g{i}.fasta"), "ab") as f:  # make the copies distinct files
                f.write(b">x%d\nACGTTGCA%s\n" % (i, b"A" * i))
        t = host.create_delta_tree(tag="s", genomedir=d, sketchdir=os.path.join(d, "sk"), kstart=8, nchildren=nchildren,
                                   registers=10, ksweep=(8, 8))
        os.makedirs(os.path.join(d, "sk"), exist_ok=True)
        return [[c.node_title for c in n.children] for n in t._dt if n.children]

    os.makedirs(os.path.join(str(tmp_path), "x"), exist_ok=True)
    s52 = shape(5, 2)
    assert s52 == [["g0", "g1"], ["g2", "g3"], ["g4", "g0_g1"], ["g2_g3", "g4_g0_g1"]]
    s73 = shape(7, 3)
    assert s73[0] == ["g0", "g1", "g2"] and s73[-1] == ["g3", "g4", "g5", "g6", "g0_g1_g2"]


def test_dashing_container_sketch_directory_is_consumed_and_extended(host, tmp_path, monkeypatch):
    """a sketch directory whose .hll files are in Dashing's container (as a real DandD + Dashing run
    leaves them, gzip or plain) is read in place -- no leaf is sketched again -- and with
    DANDD_SKETCH_FORMAT=dashing new sketches are written in that container too; rows equal the goldens."""
    import gzip
    import shutil
    gold = _golden("ref_hll.json")
    calls = {"leaf": 0}

    class Counting(hostcheck.OracleBackend):
        def leaf(self, *a):
            calls["leaf"] += 1
            return super().leaf(*a)

    host.set_backend_factory(lambda r, c: Counting(r, c))
    from dandd_amd.host import cli
    data = os.path.join(str(tmp_path), "data")
    shutil.copytree(os.path.join(hostcheck.GOLD, "fasta"), data)
    out = os.path.join(str(tmp_path), "o")
    args = ["tree", "-d", data, "-o", out, "-s", "gold", "-k", "10", "-r", str(gold["registers"])]
    monkeypatch.setenv("DANDD_SKETCH_FORMAT", "dashing")
    cli.main(args)
    assert calls["leaf"] > 0
    sk = os.path.join(out, "sketchdb")
    leaf_files = [os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(sk, "ngen1")) for f in fs if f.endswith(".hll")]
    assert leaf_files and all(open(f, "rb").read(2) == b"\x1f\x8b" for f in leaf_files)
    raw = gzip.open(leaf_files[0], "rb").read()
    assert len(raw) == 32 + (1 << gold["registers"])
    # half of the leaf sketches re-written uncompressed (Dashing without -z), caches dropped: a second run under
    # the default (native) switch must read both flavours and sketch nothing
    for f in leaf_files[::2]:
        plain = gzip.open(f, "rb").read()
        with open(f, "wb") as g:
            g.write(plain)
    monkeypatch.delenv("DANDD_SKETCH_FORMAT")
    for name in os.listdir(sk):
        if name.endswith(".pickle") or name.endswith(".bkp"):
            os.remove(os.path.join(sk, name))
    before = calls["leaf"]
    out2 = os.path.join(str(tmp_path), "o2")
    cli.main(["tree", "-d", data, "-o", out2, "-c", sk, "-s", "gold", "-k", "10", "-r", str(gold["registers"])])
    assert calls["leaf"] == before, "leaf sketches in Dashing's container were sketched again"
    rows = hostcheck.read_rows(os.path.join(out2, "gold_5_dashing_deltas.csv"))
    assert not hostcheck.compare({"tree_spider_k10": rows}, {"tree_spider_k10": gold["scenarios"]["tree_spider_k10"]})


def test_afproject_tuples_and_phylip_export(host, tmp_path):
    """`kij --afproject` (lib/dandd_cmd.py:123-132, lib/huffman_dandd.py:697-718): the pickled tuples are the kij and
    Jaccard rows in the (tool, A, B, k, value, Ak, Bk, ABk) form of the reference's helpers, and the PHYLIP writer
    lays the k = 0 (KIJ) distances out the way helpers/allpairs.py:182-207 does."""
    import pickle
    import shutil
    from dandd_amd.host import cli
    from dandd_amd.host.deltatree import write_phylip
    gold = _golden("ref_hll.json")
    host.set_backend_factory(lambda r, c: hostcheck.OracleBackend(r, c))
    data = os.path.join(str(tmp_path), "data")
    shutil.copytree(os.path.join(hostcheck.GOLD, "fasta"), data)
    out = os.path.join(str(tmp_path), "o")
    cli.main(["tree", "-d", data, "-o", out, "-s", "gold", "-k", "10", "-r", str(gold["registers"])])
    cli.main(["kij", "-d", os.path.join(out, "gold_5_dashing_dtree.pickle"), "-o", out, "--jaccard", "--mink", "8", "--maxk", "12", "--afproject"])
    with open(os.path.join(out, "gold_5_dashing_AFtuples.pickle"), "rb") as f:
        tuples = pickle.load(f)
    kij = {(t[1], t[2]): t for t in tuples if t[3] == 0}
    assert len(kij) == 10 and len(tuples) == 10 + 10 * 5
    for row in gold["scenarios"]["kij"]:
        t = kij[(row["Atitle"], row["Btitle"])]
        assert t[0] == "dashing" and t[4] == float(row["KIJ"]) and (t[5], t[6], t[7]) == (int(row["Ak"]), int(row["Bk"]), int(row["ABk"]))
    for row in gold["scenarios"]["kij_jaccard_8_12"]:
        assert ("dashing", row["Atitle"], row["Btitle"], int(row["kval"]), float(row["jaccard"]), None, None, None) in tuples
    names = write_phylip(tuples, os.path.join(out, "kij.phy"))
    lines = open(os.path.join(out, "kij.phy")).read().splitlines()
    assert names == ["g0", "g1", "g2", "g3", "g4"] and lines[0] == "5" and lines[1] == "g0"
    assert lines[3].split()[0] == "g2" and float(lines[3].split()[2]) == 1 - kij[("g1", "g2")][4]
    assert [len(l.split()) for l in lines[1:]] == [1, 2, 3, 4, 5]
    write_phylip(tuples, os.path.join(out, "j10.phy"), k=10)
    with pytest.raises(ValueError):
        write_phylip(tuples, os.path.join(out, "none.phy"), k=99)


@pytest.mark.gpu
def test_cli_at_the_reference_default_register_count_on_gpu(host, tmp_path, torch_cuda):
    """DandD's default is `-r 20` (/root/reference/lib/dandd_cmd.py:187): the whole drop-in walk -- tree (hill-climb),
    progressive --ksweep, kij --jaccard -- with 2^20 registers (K1 = scatter + sort + replay, 1 MiB per sketch file)
    writes the rows the oracle backend writes behind the same host layer."""
    import shutil
    from dandd_amd.host import cli
    import pickle

    def walk(name, factory):
        host.set_backend_factory(factory)
        work = os.path.join(str(tmp_path), name)
        data = os.path.join(work, "data")
        shutil.copytree(os.path.join(hostcheck.GOLD, "fasta"), data)
        out = os.path.join(work, "o")
        cli.main(["tree", "-d", data, "-o", out, "-s", "gold", "-k", "10"])   # no -r: the default, 20
        tree = os.path.join(out, "gold_5_dashing_dtree.pickle")
        with open(os.path.join(out, "sketchdb", "gold_5_orderings.pickle"), "wb") as f:
            pickle.dump({(0, 1, 2, 3, 4), (4, 2, 0, 3, 1)}, f)
        cli.main(["progressive", "-d", tree, "-o", out, "--ksweep", "--mink", "9", "--maxk", "12"])
        cli.main(["kij", "-d", tree, "-o", out, "--jaccard", "--mink", "9", "--maxk", "11"])
        names = ["gold_5_dashing_deltas.csv", "gold_progu0_5_dashingsummary.csv", "gold_5_dashing.kij.csv", "gold_5_dashing.j.csv"]
        rows = {n: hostcheck.read_rows(os.path.join(out, n)) for n in names}
        assert os.path.getsize([os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(out, "sketchdb", "ngen1")) for f in fs][0]) == 12 + (1 << 20)
        return rows

    gpu = walk("gpu", None)
    cpu = walk("cpu", lambda r, c: hostcheck.OracleBackend(r, c))
    assert all(len(v) > 0 for v in gpu.values())
    diffs = hostcheck.compare(gpu, cpu)
    assert not diffs, "\n".join(diffs[:30])


@pytest.mark.gpu
def test_cfg1_exact_ksweep_on_gpu_matches_exact_goldens(host, tmp_path, torch_cuda):
    """BASELINE config 1 as stated: `dandd tree --exact --ksweep --mink 10 --maxk 20`.  Every node's cardinality at
    every k from the GPU exact counter == what the reference's orchestration recorded over an exact counter
    (ref_exact.json `tree_ksweep_10_20_cards`; the reference's own KMC branch recurses forever, SURVEY section 0, so
    its Dashing-named files stand in: only the names differ)."""
    import pickle
    import re
    import shutil
    from dandd_amd.host import cli
    gold = _golden("ref_exact.json")["scenarios"]["tree_ksweep_10_20_cards"]
    host.set_backend_factory(None)
    data = os.path.join(str(tmp_path), "data")
    shutil.copytree(os.path.join(hostcheck.GOLD, "fasta"), data)
    out = os.path.join(str(tmp_path), "o")
    cli.main(["tree", "-d", data, "-o", out, "-s", "gold", "--exact", "--ksweep", "--mink", "10", "--maxk", "20"])
    with open(os.path.join(out, "sketchdb", "gold_kmc_cardinalities.pickle"), "rb") as f:
        got = {os.path.basename(k): v for k, v in pickle.load(f).items()}

    def key(name):  # (leaf fasta or "root", k) from either naming scheme
        m = re.match(r"(g\d\.fasta)(?:\.w\.(\d+)\.spacing\.\d+\.hll|_k(\d+))$", name)
        if m:
            return m.group(1), int(m.group(2) or m.group(3))
        m = re.match(r"0x[0-9a-f]+_\d+n5k(\d+)(?:\.hll)?$", name)
        assert m, name
        return "root", int(m.group(1))

    want = {key(n): v for n, v in gold.items()}
    have = {key(n): v for n, v in got.items()}
    assert len(want) == 6 * 11 and set(want) <= set(have)
    for k2, v in want.items():
        assert have[k2] == v, (k2, have[k2], v)


@pytest.mark.gpu
def test_progressive_hill_climb_on_gpu(host, tmp_path, torch_cuda):
    """The same walk with the product backend (HipBackend: leaf_many + whole-schedule launches) in hill-climb mode."""
    _schedule_walk(host, tmp_path, [], lambda r, c: hostcheck.OracleBackend(r, c), None)


@pytest.mark.parametrize("sweep", [[], ["--ksweep", "--mink", "4", "--maxk", "14"]])
def test_progressive_and_kij_with_schedule_hooks(host, tmp_path, sweep):
    _schedule_walk(host, tmp_path, sweep, lambda r, c: hostcheck.OracleBackend(r, c), lambda r, c: hostcheck.ScheduleBackend(r, c))


def _schedule_walk(host, tmp_path, sweep, plain_factory, hooked_factory):
    """A backend WITH the schedule entry points (the GPU one; here the oracle dressed up the same way) sends
    `progressive` and `kij` through the prefetch paths: leaf windows sketched ahead, union cardinalities taken from
    whole-schedule tables.  Same rows as the plain three-method backend, and `save` finds every base it recorded --
    the hill-climb `progressive` used to die in save on a k its prefetch window had noted and the search never
    visited (found on the GPU box by scripts/e2e_cli.py --hillclimb)."""
    import shutil
    from dandd_amd.host import cli

    import pickle
    orderings = os.path.join(str(tmp_path), "orderings.pickle")   # fixed: both runs load the same set the same way
    with open(orderings, "wb") as f:
        pickle.dump({(0, 1, 2, 3, 4), (4, 2, 0, 3, 1), (1, 3, 4, 0, 2)}, f)

    def run(factory, name):
        host.set_backend_factory(factory)
        data = os.path.join(str(tmp_path), name, "data")
        shutil.copytree(os.path.join(hostcheck.GOLD, "fasta"), data)
        out = os.path.join(str(tmp_path), name, "o")
        cli.main(["tree", "-d", data, "-o", out, "-s", "gold", "-r", "12"] + sweep)
        pk = os.path.join(out, "gold_5_dashing_dtree.pickle")
        cli.main(["progressive", "-d", pk, "-o", out, "-n", "3", "--orderings", orderings] + sweep)
        cli.main(["kij", "-d", pk, "-o", out] + (["--jaccard"] + sweep if sweep else []))
        rows = {}
        for f in sorted(os.listdir(out)):
            if f.endswith(".csv") and "ordering" not in f:
                rows[f] = hostcheck.read_rows(os.path.join(out, f))
        db = hostcheck.read_rows(os.path.join(out, "gold_progu3_5_dashing_sketchdb.txt"))
        return rows, db

    try:
        plain, db_plain = run(plain_factory, "plain")
        hooked, db_hooked = run(hooked_factory, "hooked")
    finally:
        host.set_backend_factory(None)
    assert sorted(plain) == sorted(hooked) and any("progu3" in f for f in plain)
    for f in plain:
        strip = lambda rows: sorted(tuple(sorted((k, v) for k, v in r.items() if k not in ("sketchloc",))) for r in rows)
        assert strip(plain[f]) == strip(hooked[f]), f
    assert sorted(r["sketchbase"] for r in db_plain) == sorted(r["sketchbase"] for r in db_hooked)


@pytest.mark.gpu
def test_backend_keeps_written_sketches_and_batch_cards(tmp_path, torch_cuda):
    """HipBackend.leaf_many: the cardinalities it returns are the ones `card` gives file by file, the registers of
    a stored sketch are served from memory only while the file on disk is the one that was written, and a cache
    of zero bytes (DANDD_SKETCH_CACHE_MB=0) changes nothing but the reads."""
    import numpy as np
    from dandd_amd.host.backend import HipBackend, read_sketch_file, write_sketch_file
    rng = np.random.default_rng(5)
    fastas = []
    for g in range(3):
        p = os.path.join(str(tmp_path), f"g{g}.fasta")
        with open(p, "wb") as f:
            f.write(b">g\n" + rng.choice(np.frombuffer(b"ACGT", np.uint8), size=20000 + 7000 * g).tobytes() + b"\n")
        fastas.append(p)
    be = HipBackend(12, True)
    path_of = lambda i, k: os.path.join(str(tmp_path), f"s{i}.w.{k}.spacing.12.hll")
    cards = be.leaf_many(fastas, 9, 13, path_of)
    assert len(cards) == 15 and all(c > 0 for c in cards.values())
    for path, c in cards.items():
        regs = read_sketch_file(path)[0]
        assert c == be.engine.card(regs) == be.card(path)
        assert np.array_equal(be._load(path)[0], regs)
    # a file replaced behind the backend's back is read again, a removed one is an error as it always was
    victim = path_of(0, 9)
    other = read_sketch_file(path_of(2, 13))[0]
    os.utime(victim, ns=(1, 1))
    write_sketch_file(victim, other, 12, 9, True)
    assert be.card(victim) == be.engine.card(other) != cards[victim]
    os.remove(victim)
    with pytest.raises(OSError):
        be.card(victim)
    # unions come out the same with and without the memory
    be.union([path_of(0, 11), path_of(1, 11), path_of(2, 11)], os.path.join(str(tmp_path), "u11.hll"))
    be2 = HipBackend(12, True)
    be2._recent_limit = 0
    be2.union([path_of(0, 11), path_of(1, 11), path_of(2, 11)], os.path.join(str(tmp_path), "v11.hll"))
    assert not be2._recent
    assert np.array_equal(read_sketch_file(os.path.join(str(tmp_path), "u11.hll"))[0], read_sketch_file(os.path.join(str(tmp_path), "v11.hll"))[0])
    assert be.card(os.path.join(str(tmp_path), "u11.hll")) == be2.card(os.path.join(str(tmp_path), "v11.hll"))


@pytest.mark.gpu
def test_leaf_slab_stays_in_hbm_between_schedules(tmp_path, torch_cuda, monkeypatch):
    """HipBackend keeps the leaf slab of a schedule on the device: a second `pairwise_cards` / `progressive_cards` over the same
    sketch files reads no file and copies nothing, a file that changed behind its back is loaded again, and DANDD_DEVICE_CACHE_MB=0
    gives the same tables through host memory."""
    import numpy as np
    from dandd_amd.host import backend as B
    rng = np.random.default_rng(11)
    fastas = []
    for g in range(5):
        p = os.path.join(str(tmp_path), f"g{g}.fasta")
        with open(p, "wb") as f:
            f.write(b">g\n" + rng.choice(np.frombuffer(b"ACGT", np.uint8), size=30000 + 5000 * g).tobytes() + b"\n")
        fastas.append(p)
    be = B.HipBackend(12, True)
    path_of = lambda i, k: os.path.join(str(tmp_path), f"s{i}.w.{k}.spacing.12.hll")
    be.leaf_many(fastas, 9, 12, path_of)
    paths = [[path_of(i, k) for k in range(9, 13)] for i in range(5)]
    ords = [[0, 1, 2, 3, 4], [4, 2, 0, 3, 1]]
    pair1, prog1 = be.pairwise_cards(paths), be.progressive_cards(paths, ords)
    reads = []
    real = B.read_sketch_file
    monkeypatch.setattr(B, "read_sketch_file", lambda p: reads.append(p) or real(p))
    be._recent.clear()                                          # (nothing in host memory either: a read would have to go to the files)
    be._recent_bytes = 0
    assert np.array_equal(be.pairwise_cards(paths), pair1) and np.array_equal(be.progressive_cards(paths, ords), prog1)
    assert not reads and be._dev is not None
    # the same files named in another order (progressive lists the leaves in its first ordering's order, kij in the tree's): same copy
    shuffled = [3, 0, 4, 1, 2]
    pair_s = be.pairwise_cards([paths[i] for i in shuffled])
    assert not reads and np.array_equal(pair_s, pair1[np.ix_(shuffled, shuffled)])
    inv = {g: i for i, g in enumerate(shuffled)}
    prog_s = be.progressive_cards([paths[i] for i in shuffled], [[inv[g] for g in o] for o in ords])
    assert not reads and np.array_equal(prog_s, prog1)
    # one file replaced: another size-preserving content, a new mtime
    other = real(path_of(4, 12))[0]
    os.utime(path_of(0, 12), ns=(1, 1))
    B.write_sketch_file(path_of(0, 12), other, 12, 12, True)
    pair2 = be.pairwise_cards(paths)
    assert pair2[0, 0, 3] == pair1[4, 4, 3] and np.array_equal(pair2[1:, 1:], pair1[1:, 1:])
    monkeypatch.setenv("DANDD_DEVICE_CACHE_MB", "0")
    be2 = B.HipBackend(12, True)
    assert np.array_equal(be2.pairwise_cards(paths), pair2) and be2._dev is None
    be.close()
    be2.close()
    # a resident backend (what `dandd serve` makes of it) leaves the slab on the device when it sketches: same tables, no read at all
    monkeypatch.delenv("DANDD_DEVICE_CACHE_MB")
    res = B.HipBackend(12, True)
    res.resident = True
    path_r = lambda i, k: os.path.join(str(tmp_path), f"r{i}.w.{k}.spacing.12.hll")
    res.leaf_many(fastas, 9, 12, path_r)
    res._recent.clear()
    res._recent_bytes = 0
    del reads[:]
    paths_r = [[path_r(i, k) for k in range(9, 13)] for i in range(5)]
    assert res._dev is not None and np.array_equal(res.pairwise_cards(paths_r), pair1) and np.array_equal(res.progressive_cards(paths_r, ords), prog1)
    assert not reads
    res.close()
