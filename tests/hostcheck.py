"""Shared helpers for the host-layer golden replays: checker backends (CPU oracle / exact counter)
that plug into dandd_amd.host.deltatree through the same three-method backend contract as the GPU
backend, and the scenario driver that mirrors tests/golden/make_golden.py."""
import csv
import json
import os
import pickle
import shutil

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


class OracleBackend:
    """dashing sketch/union/card restated by the CPU oracle (test infrastructure)."""
    name = "oracle"

    def __init__(self, log2m, canonical):
        from oracle import dd_oracle
        from dandd_amd.host import backend as B
        self.orc, self.B, self.log2m, self.canonical = dd_oracle, B, int(log2m), bool(canonical)

    def describe(self, op, **kw):
        return f"oracle:{op}"

    def leaf(self, fasta, ks, out_paths):
        fa = np.fromfile(fasta, dtype=np.uint8)
        for k, out in zip(ks, out_paths):
            self.B.write_sketch_file(out, self.orc.sketch(fa, int(k), self.log2m, self.canonical), self.log2m, int(k), self.canonical)

    def union(self, in_paths, out_path):
        parts = [self.B.read_sketch_file(p) for p in in_paths]
        self.B.write_sketch_file(out_path, self.orc.union(*[p[0] for p in parts]), self.log2m, parts[0][2], self.canonical)

    def card(self, path):
        regs, log2m, _, _ = self.B.read_sketch_file(path)
        return self.orc.card(regs, log2m)


class ScheduleBackend(OracleBackend):
    """The oracle backend with the GPU backend's batch entry points (leaf_many, pairwise_cards,
    progressive_cards): the host layer then takes the same prefetch paths it takes on the GPU."""
    name = "oracle+schedules"

    def leaf_many(self, fastas, kmin, kmax, path_of):
        ks = list(range(kmin, kmax + 1))
        for i, f in enumerate(fastas):
            OracleBackend.leaf(self, f, ks, [path_of(i, k) for k in ks])

    def _slab(self, leaf_paths):
        return [[self.B.read_sketch_file(p)[0] for p in row] for row in leaf_paths]

    def pairwise_cards(self, leaf_paths):
        slab = self._slab(leaf_paths)
        n, K = len(slab), len(slab[0])
        out = np.zeros((n, n, K))
        for i in range(n):
            for j in range(n):
                for kk in range(K):
                    out[i, j, kk] = self.orc.card(self.orc.union(slab[i][kk], slab[j][kk]), self.log2m)
        return out

    def progressive_cards(self, leaf_paths, orderings):
        slab = self._slab(leaf_paths)
        n, K = len(slab), len(slab[0])
        out = np.zeros((len(orderings), n, K))
        for o, order in enumerate(orderings):
            for kk in range(K):
                acc = None
                for j, g in enumerate(order):
                    acc = slab[g][kk] if acc is None else self.orc.union(acc, slab[g][kk])
                    out[o, j, kk] = self.orc.card(acc, self.log2m)
        return out


class ExactBackend:
    """KMC stand-in: a 'sketch' is the list of FASTAs, its cardinality the exact distinct count."""
    name = "exact"

    def __init__(self, log2m, canonical):
        from oracle import dd_oracle
        self.orc, self.canonical = dd_oracle, bool(canonical)

    def describe(self, op, **kw):
        return f"exact:{op}"

    def leaf(self, fasta, ks, out_paths):
        for k, out in zip(ks, out_paths):
            with open(out, "w") as f:
                json.dump({"k": int(k), "fastas": [fasta]}, f)

    def union(self, in_paths, out_path):
        parts = [json.load(open(p)) for p in in_paths]
        with open(out_path, "w") as f:
            json.dump({"k": parts[0]["k"], "fastas": sorted({x for p in parts for x in p["fastas"]})}, f)

    def card(self, path):
        s = json.load(open(path))
        return float(self.orc.exact_count([np.fromfile(f, dtype=np.uint8) for f in s["fastas"]], s["k"], self.canonical))


def _norm(key, v):
    if v is None:
        return None
    if key in ("fastas", "files"):
        if v.startswith("["):
            return [os.path.basename(x) for x in eval(v)]
        return [os.path.basename(x) for x in v.split("|")]
    if key in ("A", "B", "sketchloc"):
        return os.path.basename(v) if v else v
    return v


def read_rows(path):
    with open(path, newline="") as f:
        return [{k: _norm(k, v) for k, v in row.items() if k != "command"} for row in csv.DictReader(f)]


def run_scenarios(work, registers):
    """The same CLI walk as make_golden.scenario(), through dandd_amd.host.cli."""
    from dandd_amd.host import cli
    data = os.path.join(work, "data")
    shutil.copytree(os.path.join(GOLD, "fasta"), data)
    out = {}

    def od(name):
        d = os.path.join(work, name)
        os.makedirs(d, exist_ok=True)
        return d

    o = od("t1")
    cli.main(["tree", "-d", data, "-o", o, "-s", "gold", "-k", "10", "-r", str(registers)])
    out["tree_spider_k10"] = read_rows(os.path.join(o, "gold_5_dashing_deltas.csv"))
    tree_pickle = os.path.join(o, "gold_5_dashing_dtree.pickle")
    with open(os.path.join(o, "sketchdb", "gold_5_orderings.pickle"), "wb") as f:
        pickle.dump({(0, 1, 2, 3, 4), (4, 2, 0, 3, 1), (1, 3, 4, 0, 2)}, f)
    o2 = od("p1")
    cli.main(["progressive", "-d", tree_pickle, "-o", o2, "--ksweep", "--mink", "8", "--maxk", "14"])
    out["progressive_ksweep_8_14"] = read_rows(os.path.join(o2, "gold_progu0_5_dashing.csv"))
    out["progressive_ksweep_8_14_summary"] = read_rows(os.path.join(o2, "gold_progu0_5_dashingsummary.csv"))
    o2b = od("p2")
    cli.main(["progressive", "-d", tree_pickle, "-o", o2b])
    out["progressive_hillclimb"] = read_rows(os.path.join(o2b, "gold_progu0_5_dashing.csv"))
    o3 = od("k1")
    cli.main(["kij", "-d", tree_pickle, "-o", o3, "--jaccard", "--mink", "8", "--maxk", "12"])
    out["kij"] = read_rows(os.path.join(o3, "gold_5_dashing.kij.csv"))
    out["kij_jaccard_8_12"] = read_rows(os.path.join(o3, "gold_5_dashing.j.csv"))
    o4 = od("t2")
    cli.main(["tree", "-d", data, "-o", o4, "-s", "gold", "-k", "12", "-r", str(registers), "-n", "2"])
    out["tree_n2_k12"] = read_rows(os.path.join(o4, "gold_5_dashing_deltas.csv"))
    o5 = od("t3")
    cli.main(["tree", "-d", data, "-o", o5, "-s", "gold", "-r", str(registers), "--ksweep", "--mink", "9", "--maxk", "12", "-C"])
    out["tree_ksweep_9_12_nocanon"] = read_rows(os.path.join(o5, "gold_5_dashing_deltas.csv"))
    o6 = od("t4")  # BASELINE config 1 as stated: k-sweep 10..20, every node's cardinality at every k
    cli.main(["tree", "-d", data, "-o", o6, "-s", "gold", "-r", str(registers), "--ksweep", "--mink", "10", "--maxk", "20"])
    out["tree_ksweep_10_20"] = read_rows(os.path.join(o6, "gold_5_dashing_deltas.csv"))
    out["tree_ksweep_10_20_cards"] = card_table(os.path.join(o6, "sketchdb"), "gold")
    o7 = od("t5")
    cli.main(["tree", "-d", data, "-o", o7, "-s", "gold", "-k", "11", "-r", str(registers), "-n", "3"])
    out["tree_n3_k11"] = read_rows(os.path.join(o7, "gold_5_dashing_deltas.csv"))
    o8 = od("p3")
    cli.main(["progressive", "-d", tree_pickle, "-o", o8, "--ksweep", "--mink", "9", "--maxk", "12", "--step", "2"])
    out["progressive_step2_9_12"] = read_rows(os.path.join(o8, "gold_progu0_5_dashing.csv"))
    flist = os.path.join(work, "four.txt")
    with open(flist, "w") as f:
        f.write("\n".join(os.path.join(data, n) for n in ("g3.fasta", "g0.fasta", "g4.fasta", "g1.fasta")) + "\n")
    o9 = od("p4")
    cli.main(["progressive", "-d", tree_pickle, "-o", o9, "-f", flist, "-n", "1", "--ksweep", "--mink", "9", "--maxk", "11"])
    out["progressive_flist_n1_9_11"] = read_rows(os.path.join(o9, "gold_progu1_5_dashing.csv"))
    o10 = od("k2")
    cli.main(["kij", "-d", tree_pickle, "-o", o10, "--jaccard", "--afproject", "--mink", "9", "--maxk", "11"])
    out["kij_af"] = read_rows(os.path.join(o10, "gold_5_dashing.kij.csv"))
    out["kij_af_jaccard_9_11"] = read_rows(os.path.join(o10, "gold_5_dashing.j.csv"))
    with open(os.path.join(o10, "gold_5_dashing_AFtuples.pickle"), "rb") as f:
        out["kij_af_tuples"] = sorted([["" if x is None else str(x) for x in t] for t in pickle.load(f)])
    o11 = od("t6")
    cli.main(["tree", "-d", data, "-f", flist, "-o", o11, "-s", "gold", "-l", "lab", "-c", os.path.join(work, "sk6"), "-k", "9",
              "-r", str(registers), "--fast"])
    out["tree_flist_label_fast"] = read_rows(os.path.join(o11, "gold_lab_4_dashing_deltas.csv"))
    out["tree_flist_label_fast_files"] = sorted(os.listdir(o11))
    return out


def card_table(sketchdir, tag, tool="dashing"):
    with open(os.path.join(sketchdir, f"{tag}_{tool}_cardinalities.pickle"), "rb") as f:
        return {os.path.basename(k): v for k, v in pickle.load(f).items()}


def same_cell(a, b):
    if a == b:
        return True
    try:
        return float(a) == float(b)
    except (TypeError, ValueError):
        return False


def compare(got, want):
    """-> list of human-readable differences between two scenario dicts"""
    diffs = []
    for name, rows in want.items():
        if name.startswith("_"):
            continue
        g = got.get(name)
        if isinstance(rows, dict):  # {sketch basename: cardinality}
            for key, want_v in rows.items():
                if g is None or key not in g or float(g[key]) != float(want_v):
                    diffs.append(f"{name}[{key}]: got {None if g is None else g.get(key)!r}, expected {want_v!r}")
            continue
        if g is None or len(g) != len(rows):
            diffs.append(f"{name}: {None if g is None else len(g)} rows, expected {len(rows)}")
            continue
        if rows and isinstance(rows[0], str):  # a directory listing
            if list(g) != list(rows):
                diffs.append(f"{name}: got {g!r}, expected {rows!r}")
            continue
        if rows and isinstance(rows[0], list):  # tuples (the AFproject pickle), already sorted
            for i, (r, w) in enumerate(zip(g, rows)):
                if len(r) != len(w) or not all(same_cell(a, b) for a, b in zip(r, w)):
                    diffs.append(f"{name}[{i}]: got {r!r}, expected {w!r}")
            continue
        for i, (r, w) in enumerate(zip(g, rows)):
            for key in w:
                if not same_cell(r.get(key), w[key]):
                    diffs.append(f"{name}[{i}].{key}: got {r.get(key)!r}, expected {w[key]!r}")
    return diffs
