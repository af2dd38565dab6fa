"""N>1 path on CPU: world_size-2 gloo run of dandd_amd.dist (sharding plan, MAX all-reduce of the
root slab, card gather) checked against a single-process oracle computation."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_plan_is_balanced_and_deterministic():
    from dandd_amd.dist import shard_by_weight
    w = [50, 10, 40, 30, 20, 60, 5]
    plan = shard_by_weight(w, 3)
    assert sorted(i for p in plan for i in p) == list(range(len(w)))
    loads = [sum(w[i] for i in p) for p in plan]
    assert max(loads) - min(loads) <= max(w)
    assert plan == shard_by_weight(w, 3)
    assert shard_by_weight(w, 1) == [list(range(len(w)))]
    assert shard_by_weight([], 4) == [[], [], [], []]


def test_cfg5_and_cfg4_launch_geometry_on_eight_and_four_gpus():
    """Row (e) stays launch-ready without a node to run it on: the plans `bench.py --config cfg5 --gpus 8` and
    `--config cfg4 --gpus 4` would follow, checked for size on the CPU.  BASELINE cfg 5: 100 x 3 Gbp over 8 GPUs ->
    13 / 12 genomes per rank, every genome on exactly one rank, <= 40 GB of FASTA per rank (288 GB of HBM each).
    BASELINE cfg 4: 30 x 250 Mbp, k 2-32, 4 GPUs, DandD's default log2m 20: the padded slab a rank sends into the
    all-gather of leaf sketches stays <= 2 GB (and the whole exchange far inside one GPU's memory)."""
    sys.path.insert(0, ROOT)
    import bench
    from dandd_amd.dist import allgather_geometry
    plan, fasta_bytes = bench.rank_shards(bench.CONFIGS["cfg5"], 8)
    assert sorted(i for ids in plan for i in ids) == list(range(100))
    assert sorted(len(ids) for ids in plan) == [12] * 4 + [13] * 4
    assert max(fasta_bytes) <= 40e9 and min(fasta_bytes) >= 36e9
    assert plan == bench.rank_shards(bench.CONFIGS["cfg5"], 8)[0]          # every rank computes the same plan
    # a rank's share is what `--config cfg5share` benchmarks on one GPU
    assert bench.CONFIGS["cfg5share"]["genomes"] == max(len(ids) for ids in plan)
    cfg4 = bench.CONFIGS["cfg4"]
    plan4, _ = bench.rank_shards(cfg4, 4)
    assert sorted(len(ids) for ids in plan4) == [7, 7, 8, 8]
    K = cfg4["kmax"] - cfg4["kmin"] + 1
    geo = allgather_geometry(cfg4["genomes"], 4, K, 1 << 20, max(len(ids) for ids in plan4))
    assert geo["rows"] == 9 and geo["sent_bytes"] == 9 * K * (1 << 20) <= 2e9
    assert geo["full_bytes"] == 30 * K * (1 << 20) and geo["peak_bytes"] < 3e9
    assert allgather_geometry(cfg4["genomes"], 4, K, 1 << 14)["sent_bytes"] < 5e6   # log2m 14: 4.6 MB per rank


def test_two_rank_gloo_sweep_matches_single_process(orc, tmp_path):
    kmin, kmax, p = 9, 13, 10
    fastas = []
    for g, nb in enumerate([30000, 8000, 22000, 15000, 4000]):
        path = tmp_path / f"g{g}.fasta"
        path.write_bytes(orc.synth_fasta(0xD4ADD, g, nb, 2).tobytes())
        fastas.append(str(path))
    out = str(tmp_path / "res")
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(HERE, "dist_worker.py"), out, str(kmin), str(kmax), str(p)] + fastas
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = [json.load(open(f"{out}.{rank}")) for rank in range(2)]
    # every genome sketched exactly once across the two ranks
    assert sorted(res[0]["touched"] + res[1]["touched"]) == sorted(os.path.basename(f) for f in fastas)
    assert res[0]["touched"] and res[1]["touched"]
    leaves = [orc.sketch_sweep(np.fromfile(f, dtype=np.uint8), kmin, kmax, p) for f in fastas]
    root = orc.union(*leaves)
    want_leaf = [[orc.card(l[kk], p) for kk in range(kmax - kmin + 1)] for l in leaves]
    want_root = [orc.card(root[kk], p) for kk in range(kmax - kmin + 1)]
    for rr in res:  # both ranks end with the full answer
        assert rr["world"] == 2 and rr["slowest"] == 2.0
        assert rr["leaf_card"] == want_leaf
        assert rr["root_card"] == want_root
        assert rr["root_sha"] == int(root.astype(np.uint64).sum())
        assert rr["gathered_sums"] == [int(l.astype(np.int64).sum()) for l in leaves]   # all-gather: every leaf, in genome order


def test_two_rank_cli_tree_matches_single_process(orc, tmp_path):
    """`torch.distributed.run --nproc-per-node 2 ... cli tree`: the leaf sketches are sharded over the
    ranks and exchanged through the sketch directory; rank 0's outputs equal a single-process run."""
    import csv
    data = tmp_path / "genomes"
    data.mkdir()
    for g, nb in enumerate([30000, 8000, 22000, 15000, 4000, 26000]):
        (data / f"g{g}.fasta").write_bytes(orc.synth_fasta(0xD4ADD, g, nb, 2).tobytes())
    args = ["tree", "-d", str(data), "-s", "dist", "-r", "10", "--ksweep", "--mink", "9", "--maxk", "13"]
    worker = os.path.join(HERE, "cli_dist_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")

    out1, log1 = str(tmp_path / "single"), str(tmp_path / "log1")
    r = subprocess.run([sys.executable, worker, log1] + args + ["-o", out1], env=env, capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]

    out2, log2 = str(tmp_path / "two"), str(tmp_path / "log2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), worker, log2] + args + ["-o", out2]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]

    touched = [json.load(open(f"{log2}.{rank}"))["touched"] for rank in range(2)]
    assert touched[0] and touched[1], touched                      # both ranks sketched something
    assert sorted(touched[0] + touched[1]) == [f"g{g}.fasta" for g in range(6)]  # each leaf exactly once

    def rows(out):
        with open(os.path.join(out, "dist_6_dashing_deltas.csv")) as f:
            return [{k: v for k, v in r.items() if k != "command"} for r in csv.DictReader(f)]
    one, two = rows(out1), rows(out2)
    assert len(one) > 0 and len(one) == len(two)
    for a, b in zip(one, two):
        for key in a:
            va = a[key].replace(out1, "") if isinstance(a[key], str) else a[key]
            vb = b[key].replace(out2, "") if isinstance(b[key], str) else b[key]
            assert va == vb, (key, a[key], b[key])


def test_two_rank_cli_progressive_and_kij_shard_uncached_leaf_sketches(orc, tmp_path):
    """`progressive --ksweep` and `kij` over a k range the tree never sketched: under torch.distributed.run
    EVERY rank takes part (its share of the leaf sketches, by file size), the ranks meet at one barrier, rank 0
    writes the outputs -- equal to a single-process run from an identical tree -- and the launcher exits 0."""
    import csv
    import pickle
    import shutil
    worker = os.path.join(HERE, "cli_dist_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    data = tmp_path / "genomes"
    data.mkdir()
    for g, nb in enumerate([30000, 8000, 22000, 15000, 4000, 26000]):
        (data / f"g{g}.fasta").write_bytes(orc.synth_fasta(0xD4ADD, g, nb, 2).tobytes())
    orderings = str(tmp_path / "orderings.pickle")
    with open(orderings, "wb") as f:
        pickle.dump({(0, 1, 2, 3, 4, 5), (5, 3, 1, 0, 2, 4)}, f)

    def run(cmd_prefix, log, args):
        r = subprocess.run(cmd_prefix + [worker, log] + args, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]

    def rows(path, strip):
        with open(path) as f:
            return [{k: (v.replace(strip, "") if isinstance(v, str) else v) for k, v in r.items() if k != "command"}
                    for r in csv.DictReader(f)]

    results = {}
    for name, nproc in (("one", 1), ("two", 2)):
        out = str(tmp_path / name)
        run([sys.executable], str(tmp_path / f"{name}_tree"), ["tree", "-d", str(data), "-s", "d", "-r", "10", "-k", "10", "-o", out])
        tree = os.path.join(out, "d_6_dashing_dtree.pickle")
        prefix = [sys.executable] if nproc == 1 else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                                                      "--master-addr", "127.0.0.1", "--master-port", str(_free_port())]
        run(prefix, str(tmp_path / f"{name}_prog"), ["progressive", "-d", tree, "-o", out, "-r", orderings, "-n", "2", "--ksweep", "--mink", "20", "--maxk", "23"])
        prefix = prefix[:-1] + [str(_free_port())] if nproc == 2 else prefix
        run(prefix, str(tmp_path / f"{name}_kij"), ["kij", "-d", tree, "-o", out, "--jaccard", "--mink", "26", "--maxk", "28"])
        results[name] = (rows(os.path.join(out, "d_progu2_6_dashingsummary.csv"), out), rows(os.path.join(out, "d_6_dashing.kij.csv"), out),
                         rows(os.path.join(out, "d_6_dashing.j.csv"), out))
    for a, b in zip(results["one"], results["two"]):
        assert len(a) > 0 and a == b
    for step in ("prog", "kij"):
        touched = [json.load(open(str(tmp_path / f"two_{step}.{rank}")))["touched"] for rank in range(2)]
        assert touched[0] and touched[1], (step, touched)                            # both ranks sketched something
        assert sorted(touched[0] + touched[1]) == [f"g{g}.fasta" for g in range(6)]  # each leaf exactly once
