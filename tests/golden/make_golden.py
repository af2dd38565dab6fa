#!/usr/bin/env python3
"""Generates the orchestration goldens under tests/golden/ by RUNNING THE REFERENCE's own Python
(/root/reference/lib/dandd, imported/executed in this container only -- it never travels) with the
three external executables it shells out to replaced by shims on PATH:

  dashing   sketch / union / card implemented with this repo's CPU oracle (oracle/dd_oracle.py)
            -- or, with DD_SHIM_BACKEND=exact, an exact distinct-canonical-k-mer counter
            (the KMC stand-in; the reference's own --exact branch recurses forever at this commit,
            /root/reference/lib/sketch_classes.py:289 <-> :413)
  parallel  GNU-parallel's  -j N '<cmd with {}>' ::: a b c  form
            (/root/reference/lib/huffman_dandd.py:217)

What is committed: the input FASTAs (tests/golden/fasta/*.fasta, synthetic, seeded), and the rows
of every CSV the reference wrote (tests/golden/ref_*.json) with paths reduced to basenames.  No
reference source text is stored.  Run from the repo root:  python tests/golden/make_golden.py
"""
import csv
import json
import os
import pickle
import shutil
import stat
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/lib/dandd"
sys.path.insert(0, ROOT)

DASHING_SHIM = r'''#!/usr/bin/env python3
import hashlib, json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
from oracle import dd_oracle as orc
BACKEND = os.environ.get("DD_SHIM_BACKEND", "hll")
LOG = os.environ.get("DD_SHIM_LOG")
a = sys.argv[1:]
def log(**extra):
    # one JSON line per external command the reference issued: its argv as given, what `card` answered, and a digest of the
    # registers `sketch` / `union` left behind (the command trace tests/test_gpu_cli.py replays through the product's `dashing`)
    if LOG:
        with open(LOG, "a") as f:
            f.write(json.dumps(dict(argv=["dashing"] + a, **extra)) + "\n")
def digest(obj):
    return {"regs_sha256": hashlib.sha256(np.array(obj["regs"], dtype=np.uint8).tobytes()).hexdigest()} if "regs" in obj else {}
def load(p):
    with open(p, "rb") as f:
        return json.loads(f.read())
def save(p, obj):
    with open(p, "w") as f:
        json.dump(obj, f)
cmd = a[0]
if cmd == "sketch":
    canon, k, S, prefix, fasta = True, None, None, None, None
    i = 1
    while i < len(a):
        t = a[i]
        if t == "--no-canon": canon = False
        elif t.startswith("-k"): k = int(t[2:])
        elif t == "-S": i += 1; S = int(a[i])
        elif t == "--prefix": i += 1; prefix = a[i]
        elif t: fasta = t
        i += 1
    out = os.path.join(prefix, os.path.basename(fasta) + ".w.%%d.spacing.%%d.hll" %% (k, S))
    obj = {"k": k, "S": S, "canon": canon, "fastas": [fasta]}
    if BACKEND == "hll":
        fa = np.fromfile(fasta, dtype=np.uint8)
        obj["regs"] = orc.sketch(fa, k, S, canon).tolist()
    save(out, obj)
    log(out=out, **digest(obj))
elif cmd == "union":
    assert a[1] == "-z" and a[2] == "-o"
    out, ins = a[3], [load(p) for p in a[4:]]
    obj = dict(ins[0])
    obj["fastas"] = sorted(set(f for s in ins for f in s["fastas"]))
    if BACKEND == "hll":
        obj["regs"] = orc.union(*[np.array(s["regs"], dtype=np.uint8) for s in ins]).tolist()
    save(out, obj)
    log(out=out, **digest(obj))
elif cmd == "card":
    assert a[1] == "--presketched"
    print("#Path\tSize (est.)")
    cards = {}
    for p in a[2:]:
        s = load(p)
        if BACKEND == "hll":
            v = orc.card(np.array(s["regs"], dtype=np.uint8), s["S"])
        else:
            v = float(orc.exact_count([np.fromfile(f, dtype=np.uint8) for f in s["fastas"]], s["k"], s["canon"]))
        print("%%s\t%%r" %% (p, v))
        cards[p] = repr(v)
    log(cards=cards)
else:
    sys.exit("dashing shim: unknown command " + cmd)
'''

PARALLEL_SHIM = r'''#!/usr/bin/env python3
import json, os, subprocess, sys
a = sys.argv[1:]
assert a[0] == "-j", a
cmd = a[2]
sep = a.index(":::")
if os.environ.get("DD_SHIM_LOG"):     # the k-batch call itself (lib/huffman_dandd.py:217), as the reference's shell handed it over
    with open(os.environ["DD_SHIM_LOG"], "a") as f:
        f.write(json.dumps({"parallel_argv": ["parallel"] + a}) + "\n")
for x in a[sep + 1:]:
    subprocess.call(cmd.replace("{}", x), shell=True)
'''


def write_exec(path, text):
    with open(path, "w") as f:
        f.write(text)
    os.chmod(path, os.stat(path).st_mode | stat.S_IEXEC | stat.S_IXGRP | stat.S_IXOTH)


def norm_value(key, v):
    """CSV cell -> JSON value; absolute paths reduced to basenames."""
    if v is None:
        return None
    if key in ("fastas", "files"):
        if v.startswith("["):  # list repr (progressive rows)
            return [os.path.basename(x) for x in eval(v)]
        return [os.path.basename(x) for x in v.split("|")]
    if key in ("A", "B", "sketchloc"):
        return os.path.basename(v) if v else v
    if key == "command":
        return None  # the reference stores its shell line; the engine stores a descriptive string
    return v


def read_csv(path):
    with open(path, newline="") as f:
        return [{k: norm_value(k, v) for k, v in row.items() if k != "command"} for row in csv.DictReader(f)]


def make_fastas(fdir):
    from oracle import dd_oracle as orc
    os.makedirs(fdir, exist_ok=True)
    names = []
    for g in range(5):
        fa = orc.synth_fasta(0xD4ADD, g, 20000, 2)
        name = f"g{g}.fasta"
        with open(os.path.join(fdir, name), "wb") as f:
            f.write(fa.tobytes())
        names.append(name)
    return names


def run_ref(args, env, cwd):
    r = subprocess.run([sys.executable, REF] + args, env=env, cwd=cwd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"reference failed: {args}\n{r.stdout}\n{r.stderr}")
    return r.stdout


FIXTURE_ROOT = "/tmp/dandd_gold_fixture"  # absolute paths end up inside the tree pickle: the test re-creates this place


def card_table(sketchdir, tag, tool="dashing"):
    """{sketch file basename: cardinality} of the reference's cardinality cache (lib/species_specifics.py:78-89)."""
    with open(os.path.join(sketchdir, f"{tag}_{tool}_cardinalities.pickle"), "rb") as f:
        return {os.path.basename(k): v for k, v in sorted(pickle.load(f).items())}


def scenario(backend, fdir, registers):
    """One full walk through the reference CLI; returns every CSV it wrote as rows."""
    work = FIXTURE_ROOT + "_" + backend
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(work)
    try:
        bindir = os.path.join(work, "bin")
        os.makedirs(bindir)
        write_exec(os.path.join(bindir, "dashing"), DASHING_SHIM % {"root": ROOT})
        write_exec(os.path.join(bindir, "parallel"), PARALLEL_SHIM)
        env = dict(os.environ, PATH=bindir + os.pathsep + os.environ["PATH"], DD_SHIM_BACKEND=backend,
                   DD_SHIM_LOG=os.path.join(work, "trace.log"))
        data = os.path.join(work, "data")
        shutil.copytree(fdir, data)
        out = {}

        def outdir(name):
            d = os.path.join(work, name)
            os.makedirs(d, exist_ok=True)
            return d

        # 1. tree, default spider (all leaves under one root), hill-climb from kstart
        o = outdir("t1")
        run_ref(["tree", "-d", data, "-o", o, "-s", "gold", "-k", "10", "-r", str(registers)], env, work)
        out["tree_spider_k10"] = read_csv(os.path.join(o, "gold_5_dashing_deltas.csv"))
        tree_pickle = os.path.join(o, "gold_5_dashing_dtree.pickle")
        # the tree pickle the REFERENCE wrote, as a fixture (data: pickled objects, no source text): the host
        # layer must be able to run `progressive` / `kij` from it (lib/dandd_cmd.py:66,108)
        if backend == "hll":
            shutil.copyfile(tree_pickle, os.path.join(HERE, "ref_tree_hll.pickle"))
        # 2. progressive on that tree with fixed orderings (pre-seeded orderings pickle), ksweep 8..14
        sketchdir = os.path.join(o, "sketchdb")
        orderings = {(0, 1, 2, 3, 4), (4, 2, 0, 3, 1), (1, 3, 4, 0, 2)}
        with open(os.path.join(sketchdir, "gold_5_orderings.pickle"), "wb") as f:
            pickle.dump(orderings, f)
        o2 = outdir("p1")
        run_ref(["progressive", "-d", tree_pickle, "-o", o2, "--ksweep", "--mink", "8", "--maxk", "14"], env, work)
        out["progressive_ksweep_8_14"] = read_csv(os.path.join(o2, "gold_progu0_5_dashing.csv"))
        out["progressive_ksweep_8_14_summary"] = read_csv(os.path.join(o2, "gold_progu0_5_dashingsummary.csv"))
        # 2b. progressive without ksweep (hill-climb per prefix)
        o2b = outdir("p2")
        run_ref(["progressive", "-d", tree_pickle, "-o", o2b], env, work)
        out["progressive_hillclimb"] = read_csv(os.path.join(o2b, "gold_progu0_5_dashing.csv"))
        # 3. kij with jaccard on that tree
        o3 = outdir("k1")
        run_ref(["kij", "-d", tree_pickle, "-o", o3, "--jaccard", "--mink", "8", "--maxk", "12"], env, work)
        out["kij"] = read_csv(os.path.join(o3, "gold_5_dashing.kij.csv"))
        out["kij_jaccard_8_12"] = read_csv(os.path.join(o3, "gold_5_dashing.j.csv"))
        # 4. tree with nchildren=2 (Huffman-like shape) in a fresh sketchdir
        o4 = outdir("t2")
        run_ref(["tree", "-d", data, "-o", o4, "-s", "gold", "-k", "12", "-r", str(registers), "-n", "2"], env, work)
        out["tree_n2_k12"] = read_csv(os.path.join(o4, "gold_5_dashing_deltas.csv"))
        # 5. tree --ksweep (no hill-climb), non-canonical
        o5 = outdir("t3")
        run_ref(["tree", "-d", data, "-o", o5, "-s", "gold", "-r", str(registers), "--ksweep", "--mink", "9",
                 "--maxk", "12", "-C"], env, work)
        out["tree_ksweep_9_12_nocanon"] = read_csv(os.path.join(o5, "gold_5_dashing_deltas.csv"))
        # 6. BASELINE config 1 as stated: `tree --ksweep --mink 10 --maxk 20` (no hill-climb; the deltas rows are
        #    the k=0 placeholders, the result is the cardinality of every node at every k)
        o6 = outdir("t4")
        run_ref(["tree", "-d", data, "-o", o6, "-s", "gold", "-r", str(registers), "--ksweep", "--mink", "10",
                 "--maxk", "20"], env, work)
        out["tree_ksweep_10_20"] = read_csv(os.path.join(o6, "gold_5_dashing_deltas.csv"))
        out["tree_ksweep_10_20_cards"] = card_table(os.path.join(o6, "sketchdb"), "gold")
        # 7. tree with three children per node
        o7 = outdir("t5")
        run_ref(["tree", "-d", data, "-o", o7, "-s", "gold", "-k", "11", "-r", str(registers), "-n", "3"], env, work)
        out["tree_n3_k11"] = read_csv(os.path.join(o7, "gold_5_dashing_deltas.csv"))
        # 8. progressive in steps of two genomes, ksweep 9..12 (same tree, same orderings as 2.)
        o8 = outdir("p3")
        run_ref(["progressive", "-d", tree_pickle, "-o", o8, "--ksweep", "--mink", "9", "--maxk", "12", "--step", "2"], env, work)
        out["progressive_step2_9_12"] = read_csv(os.path.join(o8, "gold_progu0_5_dashing.csv"))
        # 9. progressive over a sub-list of four FASTAs given in a file, in the file's order (-n 1: the one "sorted" ordering)
        flist = os.path.join(work, "four.txt")
        with open(flist, "w") as f:
            f.write("\n".join(os.path.join(data, n) for n in ("g3.fasta", "g0.fasta", "g4.fasta", "g1.fasta")) + "\n")
        o9 = outdir("p4")
        run_ref(["progressive", "-d", tree_pickle, "-o", o9, "-f", flist, "-n", "1", "--ksweep", "--mink", "9", "--maxk", "11"], env, work)
        out["progressive_flist_n1_9_11"] = read_csv(os.path.join(o9, "gold_progu1_5_dashing.csv"))
        # 10. kij with the AFproject tuples (over all five: `kij -f list` dies in the reference itself -- the names are
        #     handed to SubSpider as strings, lib/dandd_cmd.py:113-122 -> lib/huffman_dandd.py:753)
        o10 = outdir("k2")
        run_ref(["kij", "-d", tree_pickle, "-o", o10, "--jaccard", "--afproject", "--mink", "9", "--maxk", "11"], env, work)
        out["kij_af"] = read_csv(os.path.join(o10, "gold_5_dashing.kij.csv"))
        out["kij_af_jaccard_9_11"] = read_csv(os.path.join(o10, "gold_5_dashing.j.csv"))
        with open(os.path.join(o10, "gold_5_dashing_AFtuples.pickle"), "rb") as f:
            out["kij_af_tuples"] = sorted([["" if x is None else str(x) for x in t] for t in pickle.load(f)])
        # 11. tree over a FASTA list, with a label, its own sketch directory and --fast (no pickle, no sketch/DB table)
        o11 = outdir("t6")
        run_ref(["tree", "-d", data, "-f", flist, "-o", o11, "-s", "gold", "-l", "lab", "-c", os.path.join(work, "sk6"), "-k", "9",
                 "-r", str(registers), "--fast"], env, work)
        out["tree_flist_label_fast"] = read_csv(os.path.join(o11, "gold_lab_4_dashing_deltas.csv"))
        out["tree_flist_label_fast_files"] = sorted(os.listdir(o11))
        with open(os.path.join(work, "trace.log")) as f:
            raw = [json.loads(line) for line in f]
        # `parallel` calls are noted where they stand: each covers the next len(values) commands of the trace
        trace, batches = [], []
        for e in raw:
            if "parallel_argv" in e:
                batches.append({"argv": e["parallel_argv"], "first": len(trace), "n": len(e["parallel_argv"]) - 1 - e["parallel_argv"].index(":::")})
            else:
                trace.append(e)
        out["_n_external_commands"] = len(trace)
        if backend == "hll":
            # The command trace itself: every `dashing ...` line the unmodified reference issued (after the `parallel` shim put
            # the k in), in order, with the scenario directory written as @W@; what each `card` printed; a digest of the registers
            # each `sketch` / `union` wrote; and the reference's own cardinality caches at the end (lib/species_specifics.py:78-89).
            def rel(x):
                if isinstance(x, str):
                    return x.replace(work, "@W@")
                if isinstance(x, list):
                    return [rel(v) for v in x]
                if isinstance(x, dict):
                    return {rel(k): rel(v) for k, v in x.items()}
                return x
            caches = {}
            for dirpath, _, files in sorted(os.walk(work)):
                for fn in sorted(files):
                    if fn.endswith("_dashing_cardinalities.pickle"):
                        with open(os.path.join(dirpath, fn), "rb") as f:
                            caches[rel(os.path.join(dirpath, fn))] = {rel(k): repr(v) for k, v in sorted(pickle.load(f).items())}
            with open(os.path.join(HERE, "ref_trace_hll.json"), "w") as f:
                json.dump({"registers": registers, "fastas": sorted(os.listdir(data)), "data": "@W@/data",
                           "commands": rel(trace), "parallel_calls": rel(batches), "cardinality_caches": caches}, f, indent=0, sort_keys=True)
        return out
    finally:
        shutil.rmtree(work, ignore_errors=True)


def main():
    fdir = os.path.join(HERE, "fasta")
    make_fastas(fdir)
    for backend, registers in (("hll", 12), ("exact", 12)):
        res = scenario(backend, fdir, registers)
        path = os.path.join(HERE, f"ref_{backend}.json")
        with open(path, "w") as f:
            json.dump({"backend": backend, "registers": registers, "scenarios": res}, f, indent=1, sort_keys=True)
        print("wrote", path, {k: (len(v) if isinstance(v, list) else v) for k, v in res.items()})


if __name__ == "__main__":
    main()
