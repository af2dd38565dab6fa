#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X delta-sketching engine.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Default workload (BASELINE.json configs[1], per GPU): 10 synthetic 50-Mbp FASTAs resident in HBM,
HyperLogLog log2m=14, k-sweep 4..40 (K=37).  One STEP = one pass of the whole hot path over that
batch: K0 pack + K1 fused k-sweep sketch of every genome, K2 N-way root union (+ RCCL max
all-reduce of the root when N>1), K2/K3 cardinalities of every leaf and of the root, delta =
max_k card/k on the host.  Metric: Gbp/s = bases sketched over the whole k-sweep / wall time,
aggregated over all ranks.  The step is dandd_amd.dist.sharded_ksweep -- the same function the
world-size-2 tests run.

--config selects the other BASELINE.json workloads (never the driver's default line):
  cfg2       10 x 50 Mbp per GPU, k 4-40 (weak scaling; the default)
  cfg3       64 x 5 Mbp, k 2-32, all-pairs KIJ matrix in the step (genomes sharded over the ranks)
  cfg4       30 x 250 Mbp, k 2-32, 10 orderings (strong scaling: the 30 genomes are sharded over the ranks)
  cfg5       100 x 3 Gbp, k 4-64 (strong scaling; needs >= 4 GPUs for the FASTA bytes to fit)
  cfg4share / cfg5share   one GPU's share of cfg4 / cfg5 on 8 GPUs (8 x 250 Mbp; 13 x 3 Gbp, k 4-64)

Extra objects on the JSON line (DESIGN.md "Measurement"):
  roofline      dominant kernel (K1): algorithmic bytes / its HIP-event time vs 8 TB/s, plus the VALU-issue bound
  cpu_baseline  Dashing itself when a `dashing` binary is on PATH (registers diffed against the oracle), else the
                CPU oracle run the way DandD drives Dashing (one job per (genome, k), each re-parsing the FASTA)
  accuracy_vs_exact   cardinality and delta errors of ALL leaves and the root against the GPU exact counter
  secondary     the same step at log2m 16 (where delta meets the 1 % target) and at DandD's default log2m 20
  ingest        dd_sketch_files on FASTA files (PCIe and file reads included; never `value`)
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0xD4ADD
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_PEAK_LANEOPS = 78.6e12    # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz

CONFIGS = {
    # name: genomes (total or per GPU), Mbp, nrec, kmin, kmax, sharded over ranks?, extra schedule
    "cfg2": dict(genomes=10, mbp=50.0, nrec=5, kmin=4, kmax=40, strong=False, extra=None),
    "cfg3": dict(genomes=64, mbp=5.0, nrec=5, kmin=2, kmax=32, strong=True, extra="pairwise"),
    "cfg4": dict(genomes=30, mbp=250.0, nrec=1, kmin=2, kmax=32, strong=True, extra="progressive"),
    "cfg5": dict(genomes=100, mbp=3000.0, nrec=24, kmin=4, kmax=64, strong=True, extra=None),
    "cfg4share": dict(genomes=8, mbp=250.0, nrec=1, kmin=2, kmax=32, strong=False, extra="progressive"),
    "cfg5share": dict(genomes=13, mbp=3000.0, nrec=24, kmin=4, kmax=64, strong=False, extra=None),
}


def rank_shards(cfg, world):
    """Which genomes of a config each rank holds, and the FASTA bytes that puts into its HBM: the plan main() follows
    (strong scaling: dandd_amd.dist.shard_by_weight over equal sizes; weak: every rank its own cfg['genomes'])."""
    from dandd_amd import dist as ddist
    from dandd_amd.engine import synth_size
    nb = int(cfg["mbp"] * 1e6)
    if cfg["strong"]:
        plan = ddist.shard_by_weight([nb] * cfg["genomes"], world)
    else:
        plan = [list(range(r * cfg["genomes"], (r + 1) * cfg["genomes"])) for r in range(world)]
    per = synth_size(nb, cfg["nrec"])
    return plan, [len(ids) * per for ids in plan]


def usable_cpus():
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (the GPU
    boxes show 256 logical CPUs behind a 16-CPU quota; counting 256 would oversubscribe 16x)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def dashing_baseline(fa, nbases, ks, log2m, jobs, ncpu, exe):
    """a real Dashing, driven the way DandD drives it (`parallel -j 95%` over k,
    /root/reference/lib/huffman_dandd.py:214-218), timed, and its registers diffed against the oracle --
    the only place the "bit-exact vs Dashing" question can be settled (no Dashing exists in the build image)."""
    from concurrent.futures import ThreadPoolExecutor
    from dandd_amd.host.backend import read_sketch_file
    from oracle import dd_oracle as orc
    work = tempfile.mkdtemp(prefix="dd_dashing_")
    try:
        fasta = os.path.join(work, "sample.fasta")
        fa.tofile(fasta)

        def one(k):
            d = os.path.join(work, f"k{k}")
            os.makedirs(d, exist_ok=True)
            subprocess.run([exe, "sketch", f"-k{k}", "-S", str(log2m), "--prefix", d, fasta], check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            return os.path.join(d, f"sample.fasta.w.{k}.spacing.{log2m}.hll")

        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=jobs) as ex:
            outs = list(ex.map(one, ks))
        dt = time.perf_counter() - t0
        same, checked, diagnosis = [], [k for k in ks if k <= 32][:: max(1, len(ks) // 6)], {}
        for k in checked:
            regs = read_sketch_file(outs[ks.index(k)])[0]
            ours = orc.sketch(fa, k, log2m, True)
            same.append(bool(np.array_equal(regs, ours)))
            if not same[-1]:
                diagnosis[str(k)] = diagnose_register_mismatch(regs, ours, log2m, orc)
        out = {"value": nbases / dt / 1e9, "unit": "Gbp/s", "cores": min(jobs, len(ks)), "kind": "dashing",
               "sample": f"1 synthetic genome x {nbases/1e6:g} Mbp, k {ks[0]}-{ks[-1]}, -S {log2m}: one `{exe} sketch` process per k, "
                         f"{jobs} in flight on {ncpu} usable host CPUs, {dt:.1f} s wall",
               "registers_equal_oracle": dict(zip(map(str, checked), same)), "registers_all_equal": all(same)}
        if diagnosis:
            out["mismatch_diagnosis"] = diagnosis   # which RECALL policy of oracle/POLICIES.md the difference points at
        return out
    finally:
        shutil.rmtree(work, ignore_errors=True)


def diagnose_register_mismatch(theirs, ours, log2m, orc):
    """Dashing's registers differ from the oracle's: which assumption of oracle/POLICIES.md does the difference point at?
    Same non-empty registers with other values -> the rho rule (P7); the same multiset of values at other indices -> the
    index bits (P6); statistically the same sketch (cardinalities within 3 sigma) -> the hash or its input (P3, P4, P5);
    a different cardinality -> a different SET of k-mers: the code table or the record rules (P1, P2, P10)."""
    theirs, ours = np.asarray(theirs, dtype=np.uint8), np.asarray(ours, dtype=np.uint8)
    if theirs.shape != ours.shape:
        return {"differing": None, "points_at": "P9 (container: another register count than -S asked for)"}
    sigma = 1.04 / float(np.sqrt(1 << log2m))
    ct, co = float(orc.card(theirs, log2m)), float(orc.card(ours, log2m))
    rel = abs(ct - co) / max(co, 1.0)
    d = {"differing": int((theirs != ours).sum()), "of": int(ours.size),
         "same_support": bool(np.array_equal(theirs > 0, ours > 0)),
         "same_histogram": bool(np.array_equal(np.bincount(theirs, minlength=64), np.bincount(ours, minlength=64))),
         "card_dashing": ct, "card_oracle": co, "card_rel_diff_in_sigma": rel / sigma}
    both = (theirs > 0) & (ours > 0)
    shift = np.unique(theirs[both].astype(int) - ours[both].astype(int)) if both.any() else np.array([0])
    if d["differing"] <= 0.01 * ours.size:
        d["points_at"] = "a few registers only: P4 (the all-T 32-mer), P7's saturation case, or a damaged file"
    elif d["same_histogram"]:
        d["points_at"] = "P6 (which p bits of the hash index the register)"
    elif d["same_support"] and shift.size == 1:
        d["points_at"] = f"P7 (rho: sentinel / offset of the leading-zero count; every register is off by {int(shift[0])})"
    elif rel < 3 * sigma:
        d["points_at"] = "P3 / P5 (canonical form or hash: another but statistically equivalent sketch of the same k-mer set)"
    else:
        d["points_at"] = "P1 / P2 / P10 (another set of k-mers: code table, window reset, record rules)"
    return d


def cpu_baseline(nbases, nrec, kmin, kmax, log2m, ngenomes=4):
    """The CPU side of the same work, the way DandD drives it (/root/reference/helpers/benchmark.sh):
    Dashing itself when it is on PATH, otherwise the oracle as its stand-in, on a bounded sample -- `ngenomes` genomes of
    `nbases / ngenomes` bases as FASTA files in tmpfs:
      stage 1  (benchmark.sh:131-205, lib/huffman_dandd.py:214-218)  one single-threaded job per (genome, k), each RE-READING
               and re-parsing its FASTA file and writing its register file; floor(0.95 x cores) jobs in flight
      stage 2  (benchmark.sh:131-205)  progressive 2-way unions in a seed-42 shuffled order, one job per (step, k): two
               register files read, one written, then a `card` job on the result (file read + estimate)
      stage 3  (benchmark.sh:207-246)  one N-way union per k + its `card` job
      cards    a `card` job per (genome, k) leaf sketch (lib/sketch_classes.py:306-321)
    `value` = bases / (stage 1 + stage 3 + cards): the work of one GPU step (leaf sketches, root union, all cardinalities);
    stage 2 is what `progressive` adds and is reported beside it, as is the stage-1-only rate of earlier rounds."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import dd_oracle as orc
    ncpu = usable_cpus()
    jobs = max(1, int(0.95 * ncpu))
    ks = list(range(kmin, kmax + 1))
    exe = shutil.which("dashing")
    if exe:
        try:
            fa = orc.synth_fasta(SEED, 0, nbases, nrec)
            return dashing_baseline(fa, nbases, [k for k in ks if k <= 32] or ks, log2m, jobs, ncpu, exe)
        except Exception as e:  # a dashing that does not speak the expected CLI: fall back, but say so
            note = f"`{exe}` found but unusable ({type(e).__name__}: {e}); "
    else:
        note = ""
    path = cli = None
    try:  # native-arch build for a fair timing; fall back to the portable build
        path = orc.build(arch="native", out=os.path.join("/tmp", f"liboracle_native_{os.getpid()}.so"))
        lib = orc.lib(path)
    except Exception:
        lib = orc.lib()
    # round 5: every job a PROCESS (oracle/orc_cli.c behind Dashing's command lines), as `parallel` starts `dashing` processes --
    # fork / exec, the FASTA re-read and re-parsed, the register file written, per job; threads of one process only if gcc is missing
    try:
        cli = orc.build_cli(arch="native", out=os.path.join("/tmp", f"orc_cli_{os.getpid()}"))
    except Exception:
        cli = None
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    work = tempfile.mkdtemp(prefix="dd_cpu_", dir=base)
    m = 1 << log2m
    per = nbases // ngenomes
    try:
        fastas = []
        for g in range(ngenomes):
            f = os.path.join(work, f"g{g}.fasta")
            orc.synth_fasta(SEED, g, per, nrec).tofile(f)
            fastas.append(f)
        reg_of = lambda name, k: os.path.join(work, f"{name}.k{k}.hll")

        def sketch_job(gk):   # ctypes releases the GIL for the duration of the C call: real thread parallelism
            g, k = gk
            fa = np.fromfile(fastas[g], dtype=np.uint8)          # every job reads and parses the whole file again
            regs = np.zeros(m, dtype=np.uint8)
            lib.orc_sketch(fa.ctypes.data, fa.size, k, log2m, 1, regs.ctypes.data)
            regs.tofile(reg_of(f"g{g}", k))

        def union_job(args):
            out, ins, k = args
            acc = np.fromfile(reg_of(ins[0], k), dtype=np.uint8)
            for name in ins[1:]:
                r = np.fromfile(reg_of(name, k), dtype=np.uint8)
                lib.orc_union(acc.ctypes.data, r.ctypes.data, acc.size)
            acc.tofile(reg_of(out, k))

        def card_job(args):
            name, k = args
            r = np.fromfile(reg_of(name, k), dtype=np.uint8)
            return lib.orc_card(r.ctypes.data, log2m)

        if cli:
            def sketch_job(gk):          # `dashing sketch -k<K> -S <p> --prefix <dir> <fasta>`; the thread only waits for its child process
                g, k = gk
                subprocess.run([cli, "sketch", f"-k{k}", "-S", str(log2m), "--prefix", work, fastas[g]], check=True)
                os.replace(os.path.join(work, f"g{g}.fasta.w.{k}.spacing.{log2m}.hll"), reg_of(f"g{g}", k))

            def union_job(args):         # `dashing union -o <out> <in...>`
                out, ins, k = args
                subprocess.run([cli, "union", "-o", reg_of(out, k)] + [reg_of(name, k) for name in ins], check=True)

            def card_job(args):          # `dashing card --presketched <path>`, its TSV parsed as lib/sketch_classes.py:318-321 parses it
                name, k = args
                r = subprocess.run([cli, "card", "--presketched", reg_of(name, k)], check=True, capture_output=True, text=True)
                return float(r.stdout.splitlines()[1].split("\t")[-1])

        def timed(fn, items):
            t0 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=jobs) as ex:
                list(ex.map(fn, items))
            return time.perf_counter() - t0

        t1 = timed(sketch_job, [(g, k) for g in range(ngenomes) for k in ks])
        tc = timed(card_job, [(f"g{g}", k) for g in range(ngenomes) for k in ks])
        order = list(range(ngenomes))
        import random
        random.Random(42).shuffle(order)
        t2 = 0.0
        prev = f"g{order[0]}"
        for step, g in enumerate(order[1:], 1):      # a step needs the previous step's union: steps are sequential, ks parallel
            t2 += timed(union_job, [(f"p{step}", [prev, f"g{g}"], k) for k in ks])
            t2 += timed(card_job, [(f"p{step}", k) for k in ks])
            prev = f"p{step}"
        t3 = timed(union_job, [("root", [f"g{g}" for g in range(ngenomes)], k) for k in ks])
        t3 += timed(card_job, [("root", k) for k in ks])
    finally:
        shutil.rmtree(work, ignore_errors=True)
        for tmp in (path, cli):
            if tmp and os.path.exists(tmp):
                os.remove(tmp)
    total = per * ngenomes
    step_s = t1 + tc + t3
    return {
        "value": total / step_s / 1e9,
        "unit": "Gbp/s",
        "cores": min(jobs, len(ks) * ngenomes),
        "kind": "port",
        "stages_s": {"stage1_leaf_sketches": t1, "leaf_cards": tc, "stage2_progressive_unions_and_cards": t2, "stage3_nway_union_and_card": t3},
        "stage1_only_value": total / t1 / 1e9,
        "with_stage2_value": total / (step_s + t2) / 1e9,
        "sample_short": f"{ngenomes} x {per/1e6:g} Mbp FASTA files in tmpfs, k {kmin}-{kmax}, log2m {log2m}: oracle {'PROCESS' if cli else 'thread'} per (genome,k) re-reading its file "
                        f"+ cards + root union, {jobs} in flight, {step_s:.1f} s",
        "sample": note + f"{ngenomes} synthetic genomes x {per/1e6:g} Mbp as FASTA files in {base or 'the temp dir'}, k {kmin}-{kmax}, log2m={log2m}: "
                         f"one single-threaded oracle job per (genome, k) that re-reads and re-parses its file and writes its registers, "
                         f"then `card` jobs per sketch, the N-way root union per k and its `card` (stage 1 + cards + stage 3 = the work of "
                         f"one GPU step = `value`, {step_s:.1f} s wall); stage 2 = seed-42 progressive 2-way unions + cards, reported beside; "
                         f"{jobs} jobs in flight on {ncpu} usable host CPUs ({os.cpu_count()} logical, cgroup quota applied); "
                         + ("every job is a PROCESS (oracle/orc_cli.c: fork / exec, file read and register file per job, as `parallel` runs `dashing`)"
                            if cli else "jobs are threads calling the C oracle, not processes (gcc missing): no fork/exec cost is charged") + "; no `dashing` on PATH",
        "process_per_job": bool(cli),
    }


def load_counters(genomes, mbp, kmin, kmax, p):
    """The committed rocprofv3 passes over this very workload -- profiles/r0[456]_k1_counters_*.json, the newest round first (scripts/profile_k1_counters.sh:
    FETCH_SIZE, WRITE_SIZE, SQ and TCC in separate --pmc passes, round-4 kernels) -- or None when no file matches."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[456]_k1_counters_*.json")), reverse=True):   # the newest round's first
        try:
            with open(path) as f:
                cj = json.load(f)
            w = cj["workload"]
            if (w["genomes"], w["mbp"], w["kmin"], w["kmax"], w["log2m"]) == (genomes, mbp, kmin, kmax, p):
                cj["_file"] = "profiles/" + os.path.basename(path)
                return cj
        except (OSError, KeyError, ValueError):
            pass
    return None


def isa_table():
    """profiles/r0N_isa_classes.json: instruction classes of K1's hot loops, counted by scripts/isa_classes.py in the ISA of
    the shipped build and priced with the measured issue costs (profiles/r01_ubench_issue_costs.txt)."""
    for rnd in ("r06", "r05", "r04"):      # the newest round's table first
        try:
            with open(os.path.join(ROOT, "profiles", f"{rnd}_isa_classes.json")) as f:
                t = json.load(f)
            t["_file"] = f"profiles/{rnd}_isa_classes.json"
            return t
        except (OSError, ValueError):
            pass
    return None


def issue_model(counters, kernel_s_per_step):
    """valu_bound.frac_of_mix: the SIMD time the step's K1 instructions need at their measured issue cost, over the SIMD
    time the step had.  Per kernel: SQ_INSTS_VALU of the committed counter file x the mean ns per instruction of THAT
    kernel's hot loop in the ISA table (its cheap / dear mix; kernels without an entry -- replay, sort, the small-k
    classes -- are priced at the mean of the two classes); summed and divided by 1024 SIMDs x the K1 time of a step."""
    isa = isa_table()
    if not counters or not isa or kernel_s_per_step <= 0:
        return None
    costs = isa["issue_costs"]
    by_prefix = {}
    for name, ent in isa["kernels"].items():
        m = __import__("re").match(r"([a-z_]+)<(\d)", name)
        if m:
            ns = (ent["valu_cheap_per_update"] * costs["cheap_ns"] + ent["valu_dear_per_update"] * costs["dear_ns"]) / ent["valu_per_update"]
            by_prefix[(m.group(1), m.group(2))] = (ns, ent["cheap_fraction"])
    default_ns = 0.5 * (costs["cheap_ns"] + costs["dear_ns"])
    busy_ns, rows = 0.0, {}
    k1 = ("sweep_kernel", "bitmap", "scatter", "sort_chunks", "replay", "bigmap")
    for name, ent in counters["kernels"].items():
        if not name.startswith(k1) or "sq_insts_valu" not in ent:
            continue
        m = __import__("re").match(r"([a-z_]+)<(\d)", name)
        key = (m.group(1), m.group(2)) if m else None
        if key and key[0] == "scatter_first_wg_kernel":
            key = ("scatter_first_bin_kernel", key[1])
        ns, cheap = by_prefix.get(key, (default_ns, None))
        busy_ns += ent["sq_insts_valu"] * ns
        rows[name] = {"valu_wave_instr": ent["sq_insts_valu"], "ns_per_instr": ns, "cheap_fraction_of_hot_loop": cheap}
    frac = busy_ns * 1e-9 / (1024.0 * kernel_s_per_step)
    return {"frac_of_mix": frac, "simd_busy_ms_per_step": busy_ns * 1e-6 / 1024.0, "kernel_ms_per_step": kernel_s_per_step * 1e3,
            "cheap_ns": costs["cheap_ns"], "dear_ns": costs["dear_ns"], "by_kernel": rows,
            "from": isa.get("_file", "profiles/r0N_isa_classes.json") + " (scripts/isa_classes.py over the shipped build's ISA) x " + counters.get("_file", "the counter file") +
                    " (SQ_INSTS_VALU per kernel and step); kernels that overlap on side streams (log2m >= 17) share the step's SIMD time"}


def valu_bound(kmin, kmax, updates_per_s, counters, kernel_s_per_step=0.0):
    """The bound that actually binds K1: VALU issue.  Instructions per (token, k) are SQ_INSTS_VALU of the K1 kernels of
    one step divided by the step's wave-updates, read from the committed counter file (never a table in this script);
    a wave64 instruction occupies a SIMD-32 for 2 cycles at best, so the chip retires at most 256 CU x 4 SIMD x 2.4 GHz / 2
    wave instructions per second (= 78.6 T lane-ops/s): `frac`.  `frac_of_mix` prices the same instructions at the issue
    cost of their class instead (issue_model).  None when no counter file matches the workload."""
    if not counters:
        return None
    ipu = counters["k1_valu_instr_per_update"]
    achieved = updates_per_s * ipu  # lane-instructions per second
    out = {"valu_instr_per_update": ipu, "by_class": counters.get("valu_per_update_by_class"),
           "achieved_lane_instr_per_s": achieved, "peak_lane_instr_per_s": VALU_PEAK_LANEOPS, "frac": achieved / VALU_PEAK_LANEOPS,
           "instr_counts_from": f"{counters.get('_file')} (SQ_INSTS_VALU of every K1 kernel of a step / (bases x K / 64); scripts/profile_r04.sh)"}
    model = issue_model(counters, kernel_s_per_step)
    if model:
        out["frac_of_mix"] = model["frac_of_mix"]
        out["issue_model"] = model
    return out


def k1_roofline(ng, nbytes, nb, K, m, p, kmin, kmax, sweep_ms, sweep_n, steps, genomes, mbp):
    """The `roofline` object of one K1 configuration: algorithmic bytes over the HIP-event time of the step's K1 launches
    against 8 TB/s, the counter-file traffic, and the VALU-issue bound -- for the headline and for every secondary line."""
    counters = load_counters(genomes, mbp, kmin, kmax, p)
    alg_bytes = ng * nbytes + ng * K * m
    s_per_step = sweep_ms / 1e3 / steps
    achieved = alg_bytes / s_per_step / 1e9
    updates_per_s = ng * nb * K / s_per_step
    return {
        "bound": "hbm",
        "kernel": "K1: sweep_kernel launches of one step" + (" (scatter + sort + replay at log2m >= 17; the first epoch binned straight from the hash)" if p >= 17 else ""),
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "traffic": counters["k1_bytes_per_step"]["total"] if counters else None,
        "traffic_from": (f"{counters['_file']} (separate FETCH_SIZE / WRITE_SIZE passes of scripts/profile_r04.sh; fetch "
                         f"{counters['k1_bytes_per_step']['fetch']:.4g} B = 2 x FETCH_SIZE, write {counters['k1_bytes_per_step']['write']:.4g} B)") if counters else None,
        "traffic_note": "K1 bytes per step = 2 x FETCH_SIZE + WRITE_SIZE of separate PMC passes.  log2m <= 16: above the algorithmic bytes "
                        "because each k-group re-reads the 3-bit token stream and every job merges its LDS registers into the slab "
                        "(~230 GB/s, irrelevant to a VALU-bound kernel).  log2m >= 17: the record streams (written by scatter, read "
                        "and rewritten by sort, read by replay) and the register tiles replay loads and stores per epoch",
        "algorithmic_bytes_per_step": alg_bytes,
        "kernel_ms_per_step": sweep_ms / steps,
        "launches_per_step": sweep_n / steps,
        "avg_launch_ms": sweep_ms / max(1, sweep_n),
        "register_updates_per_s": updates_per_s,
        "valu_lane_ops_peak": VALU_PEAK_LANEOPS,
        "valu_bound": valu_bound(kmin, kmax, updates_per_s, counters, s_per_step),
        "note": "integer-VALU bound (hash per (base,k)); see DESIGN.md for ops/update and the VALU fraction",
    }


LINE_LIMIT = 6000      # bytes of the ONE stdout line: a record (helpers/benchmark.sh:23 writes one CSV row per measurement), not a report


def _r(x, sig=5):
    """floats to `sig` significant digits (the sidecar keeps full precision)"""
    if isinstance(x, float):
        return float(f"{x:.{sig}g}") if np.isfinite(x) else None
    return x


def _pick(d, keys):
    return {k: _r(d[k]) for k in keys if isinstance(d, dict) and k in d}


def _roofline_compact(rf):
    if not rf:
        return rf
    out = _pick(rf, ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_step", "kernel_ms_per_step",
                     "launches_per_step", "avg_launch_ms"))
    out["kernel"] = rf.get("kernel", "").split(" (")[0][:80]
    vb = rf.get("valu_bound")
    out["valu_bound"] = _pick(vb, ("valu_instr_per_update", "frac", "frac_of_mix")) if vb else None
    return out


def _step_compact(e):
    """one secondary entry: value, ms, the two roofline fractions and (when measured) the accuracy verdict; no prose"""
    if "error" in e:
        return {"error": str(e["error"])[:120]}
    out = _pick(e, ("value", "ms_per_step", "steps"))
    rf = e.get("roofline")
    if rf:
        out["frac"] = _r(rf["frac"])
        out["traffic"] = _r(rf.get("traffic"))
        out["frac_of_mix"] = _r((rf.get("valu_bound") or {}).get("frac_of_mix"))
    for acc in ("accuracy_vs_exact", "accuracy_subsample"):
        if acc in e:
            out.update(_pick(e[acc], ("delta_rel_err_max_abs", "delta_rel_err", "delta_within_1pct")))
    return out


def _ingest_compact(e):
    if "error" in e:
        return {"error": str(e["error"])[:120]}
    out = _pick(e, ("value", "ms", "best_value", "host_decoder_value", "host_parallel_decoder_value", "serial_decoder_value"))
    for sub in ("small_files", "gzip_files", "bgzf_files", "one_big_gzip_file", "gzip_fastq_files", "multi_member_gzip_files", "big_dir", "big_dir_gzip"):
        if sub in e:
            out[sub] = _ingest_compact(e[sub])
    return out


def compact_line(full, detail_path):
    """The driver's line: the contract's keys and the numbers a reader checks, <= LINE_LIMIT bytes; every `what` / `why` /
    per-kernel table stays in the sidecar (`detail`), which holds the full object."""
    line = {k: _r(full[k]) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                     "vs_baseline", "dtype", "data")}
    cfg = dict(full["config"])
    cfg["workload"] = cfg["workload"].split("; NOTE")[0]
    line["config"] = cfg
    line["roofline"] = _roofline_compact(full["roofline"])
    if full.get("cpu_baseline"):
        cb = full["cpu_baseline"]
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "registers_all_equal"))
        line["cpu_baseline"]["sample"] = cb.get("sample_short") or cb["sample"][:160]
        if "stages_s" in cb:
            line["cpu_baseline"]["stages_s"] = {k: _r(v, 4) for k, v in cb["stages_s"].items()}
    k0 = full.get("roofline_k0")
    line["roofline_k0"] = _pick(k0, ("achieved", "frac", "kernel_ms_per_step")) if k0 else None
    k2 = full.get("roofline_k2")
    line["roofline_k2"] = _pick(k2, ("bound", "path", "achieved", "peak", "unit", "frac", "ms", "traffic")) if k2 else None
    if "accuracy_vs_exact" in full:
        line["accuracy_vs_exact"] = _pick(full["accuracy_vs_exact"], ("sketches", "delta_rel_err_max_abs", "delta_rel_err_root", "delta_within_1pct",
                                                                      "argmax_k_equal", "card_rel_err_rms", "hll_sigma"))
    if "accuracy_subsample" in full:
        line["accuracy_subsample"] = _pick(full["accuracy_subsample"], ("delta_rel_err", "delta_within_1pct", "hll_sigma"))
    if "secondary" in full:
        line["secondary"] = {k: _step_compact(v) for k, v in full["secondary"].items()}
    if "ingest" in full:
        line["ingest"] = _ingest_compact(full["ingest"])
    if "schedule" in full:
        line["schedule"] = full["schedule"]
    line["gpus_active"] = [g[:64] for g in full.get("gpus_active", [])][:8]
    line["collectives"] = full.get("collectives")
    for k in ("delta_genome0", "argmax_k_genome0", "delta_root", "argmax_k_root"):
        line[k] = _r(full.get(k), 8)
    line["detail"] = detail_path
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:      # never print a line the driver cannot read: shed the optional blocks, biggest first
        for k in ("ingest", "secondary", "gpus_active", "collectives"):
            line[k] = {"see": "detail"}
            text = json.dumps(line, separators=(",", ":"))
            if len(text) <= LINE_LIMIT:
                break
    return text


def write_detail(full, path):
    """the full object (every roofline with its per-kernel issue model, every `what` / `why`) beside bench.py; the line names it"""
    try:
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "w") as f:
            json.dump(full, f, indent=1)
        os.replace(tmp, path)
        return path if not path.startswith(ROOT + os.sep) else os.path.relpath(path, ROOT)
    except OSError as e:     # (a read-only tree must not cost the line)
        print(f"bench.py: could not write {path}: {e}", file=sys.stderr)
        return None


class Workload:
    """The genomes of one rank in HBM and the engine-backed callbacks of dandd_amd.dist.sharded_ksweep."""

    def __init__(self, torch, eng, genome_ids, nb, nrec, kmin, kmax, realistic=False):
        from dandd_amd.engine import synth_realistic_size, synth_size
        self.torch, self.eng = torch, eng
        self.kmin, self.kmax, self.K, self.m = kmin, kmax, kmax - kmin + 1, eng.m
        self.ids, self.nb, self.ng = list(genome_ids), nb, len(genome_ids)
        self.nbytes = synth_realistic_size(SEED, nb) if realistic else synth_size(nb, nrec)
        self.fasta = [torch.empty(self.nbytes + 16, dtype=torch.uint8, device="cuda") for _ in self.ids]
        for t, gi in zip(self.fasta, self.ids):
            if realistic:   # GC 35 %, 30 % soft-masked repeats, 2 % N, contigs of 2..200 kbp (dd_synth.hip)
                eng.synth_realistic_device(SEED, gi, nb, t.data_ptr())
            else:
                eng.synth_fasta_device(SEED, gi, nb, nrec, t.data_ptr())
        eng.synchronize()
        self.regs = torch.empty((self.ng + 1, self.K, self.m), dtype=torch.uint8, device="cuda")  # leaves + root
        self.ptrs = [f.data_ptr() for f in self.fasta]
        self.sizes = [self.nbytes] * self.ng
        self.ks = np.arange(kmin, kmax + 1, dtype=np.float64)

    def sketch_into(self, indices, leaves):      # K0 + K1: every local genome in one batched call
        self.eng.sketch_device(self.ptrs, self.sizes, self.kmin, self.kmax, leaves.data_ptr())

    def union_into(self, leaves, root):           # K2 root union of this rank's leaves
        if leaves.shape[0]:
            self.eng.union_device([leaves[g].data_ptr() for g in range(leaves.shape[0])], self.K * self.m, root.data_ptr())
        else:
            root.zero_()

    def card_of(self, regs):                      # K2 + K3
        return self.eng.card_batch_device(regs.data_ptr(), regs.shape[0] * self.K)

    def step(self, ddist):
        _, _, card = ddist.sharded_ksweep(self.sizes, self.K, self.m, self.sketch_into, self.union_into, self.card_of,
                                          regs=self.regs, mine=list(range(self.ng)))
        d = card / self.ks
        return d.max(axis=1), d.argmax(axis=1) + self.kmin, card   # delta, argmax-k per leaf and (last row) root


def accuracy_block(wl, card, what):
    """The "delta rel-err vs KMC --exact" half of the metric, over EVERY leaf and this rank's root: the GPU exact
    counter (sort + distinct of canonical k-mers) at every k, against the HLL cardinalities of the last step."""
    eng, K, ks = wl.eng, wl.K, wl.ks
    exact = np.zeros((wl.ng + 1, K))
    for g in range(wl.ng):
        for kk in range(K):
            exact[g, kk] = eng.exact_count_device([wl.ptrs[g]], [wl.nbytes], wl.kmin + kk)
    for kk in range(K):
        exact[wl.ng, kk] = eng.exact_count_device(wl.ptrs, wl.sizes, wl.kmin + kk)
    rel = (card - exact) / exact
    d_hll, d_ex = (card / ks).max(axis=1), (exact / ks).max(axis=1)
    drel = (d_hll - d_ex) / d_ex
    sigma = 1.04 / float(np.sqrt(wl.m))
    return {
        "what": what,
        "sketches": wl.ng + 1, "ks": K, "hll_sigma": sigma,
        "delta_rel_err_per_sketch": [float(x) for x in drel],
        "delta_rel_err_max_abs": float(np.abs(drel).max()), "delta_rel_err_mean_signed": float(drel.mean()),
        "delta_rel_err_root": float(drel[-1]),
        "argmax_k_equal": int(((card / ks).argmax(axis=1) == (exact / ks).argmax(axis=1)).sum()),
        "card_rel_err_mean_signed": float(rel.mean()), "card_rel_err_rms": float(np.sqrt((rel ** 2).mean())),
        "card_rel_err_max_abs": float(np.abs(rel).max()),
        # saturated small k (every possible k-mer present) is exact in both; the statistic that can show a hash
        # bias is the signed mean over the unsaturated ks, in units of its own standard error
        "card_rel_err_mean_signed_k_ge_14": float(rel[:, ks >= 14].mean()) if (ks >= 14).any() else None,
        "mean_signed_in_sigma_of_mean": float(rel[:, ks >= 14].mean() / (sigma / np.sqrt(max(1, rel[:, ks >= 14].size))))
        if (ks >= 14).any() else None,
        "delta_within_1pct": bool(np.abs(drel).max() <= 0.01),
    }


def subsample_accuracy(eng, wl, card, kb, p):
    """BASELINE cfg 5's accuracy clause ("delta within 1 % of exact on a subsample"): genome 0 of the workload, exact distinct
    k-mers (GPU sort + distinct, in passes) at the HLL's argmax-k and its neighbours and at k = 31 and kmax"""
    kmin, kmax = wl.kmin, wl.kmax
    sub = sorted({k for k in (kb - 1, kb, kb + 1, 31, kmax) if kmin <= k <= kmax})
    ex = {k: float(eng.exact_count_device([wl.ptrs[0]], [wl.nbytes], k)) for k in sub}
    rel = {k: float((card[0][k - kmin] - ex[k]) / ex[k]) for k in sub}
    win = [k for k in sub if abs(k - kb) <= 1]
    d_hll, d_ex = max(card[0][k - kmin] / k for k in win), max(ex[k] / k for k in win)
    return {
        "what": f"genome 0 of the workload ({wl.nb/1e9:g} Gbp), log2m {p}: HLL cardinality against the GPU exact counter at k = {sub}; delta over "
                f"the window argmax-k +- 1 (k = {win})",
        "hll_sigma": 1.04 / float(np.sqrt(wl.m)), "card_rel_err": {str(k): rel[k] for k in sub},
        "delta_rel_err": float((d_hll - d_ex) / d_ex), "delta_within_1pct": bool(abs(d_hll - d_ex) / d_ex <= 0.01)}


def bgzf_bytes(raw, level=6):
    """bgzip's container: gzip members of <= 64 KiB of text, each with its own size in a 'BC' extra field, + the empty EOF block"""
    import zlib
    out = bytearray()
    for a in list(range(0, len(raw), 65280)) + [len(raw)]:
        part = raw[a:a + 65280] if a < len(raw) else b""
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = c.compress(part) + c.flush()
        out += (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + (len(body) + 25).to_bytes(2, "little") + body +
                zlib.crc32(part).to_bytes(4, "little") + len(part).to_bytes(4, "little"))
    return bytes(out)


def ingest_probe(eng, ng, nb, nrec, kmin, kmax, torch, gz=False, reps=10, variants=()):
    """dd_sketch_files over FASTA files (tmpfs when there is one: a warm page cache), third call.
    gz: False plain, True one gzip member per file (gzip -1), "bgzf" bgzip's blocked container (level 6)."""
    import zlib
    from dandd_amd.engine import synth_size
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="dd_ingest_", dir=base)
    try:
        n = synth_size(nb, nrec)
        buf = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
        paths, deferred = [], []
        for g in range(ng):
            eng.synth_fasta_device(SEED, g, nb, nrec, buf.data_ptr())
            eng.synchronize()
            p = os.path.join(d, f"g{g:03d}.fasta" + (".gz" if gz else ""))
            if gz == "fastq":     # the genome cut into 150-base reads, four-line FASTQ, gzip -6 (one member): resolved on the device (dd_fastq.hip)
                raw = buf[:n].cpu().numpy()
                seq = raw[np.isin(raw, np.frombuffer(b"ACGTacgtN", np.uint8))]
                nreads = seq.size // 150
                rec = np.empty((nreads, 3 + 150 + 3 + 150 + 1), dtype=np.uint8)
                rec[:, :3] = np.frombuffer(b"@r\n", np.uint8)
                rec[:, 3:153] = seq[:nreads * 150].reshape(nreads, 150)
                rec[:, 153:156] = np.frombuffer(b"\n+\n", np.uint8)
                rec[:, 156:306] = np.frombuffer(b"FFFFF:FFFF,FFFFFFFF:F", np.uint8)[np.arange(150) % 21]
                rec[:, 306] = 10
                deferred.append((p, [rec.tobytes()]))
            elif gz == "members":   # `cat a.fa.gz b.fa.gz`: two gzip -6 members per file, each decoded as a stream of its own on the device
                raw = buf[:n].cpu().numpy().tobytes()
                deferred.append((p, [raw[:n // 2], raw[n // 2:]]))
            elif gz == "bgzf":
                with open(p, "wb") as f:
                    f.write(bgzf_bytes(buf[:n].cpu().numpy().tobytes()))
            elif gz == "gzip6":   # one gzip member, level 6 (what plain `gzip` writes), compressed below by a thread per file
                deferred.append((p, [buf[:n].cpu().numpy().tobytes()]))
            elif gz:  # one gzip member, level 1 (what `gzip -1` writes)
                co = zlib.compressobj(1, zlib.DEFLATED, 31)
                with open(p, "wb") as f:
                    f.write(co.compress(buf[:n].cpu().numpy().tobytes()) + co.flush())
            else:
                buf[:n].cpu().numpy().tofile(p)
            paths.append(p)
        if deferred:     # (zlib releases the GIL: 8 x 250 MB at level 6 take ~20 s this way instead of minutes)
            from concurrent.futures import ThreadPoolExecutor

            def squeeze(job):          # one gzip -6 member per part
                path, parts = job
                with open(path, "wb") as f:
                    for raw in parts:
                        co = zlib.compressobj(6, zlib.DEFLATED, 31)
                        for a in range(0, len(raw), 1 << 24):
                            f.write(co.compress(raw[a:a + (1 << 24)]))
                        f.write(co.flush())
            with ThreadPoolExecutor(max_workers=min(len(deferred), usable_cpus())) as pool:
                list(pool.map(squeeze, deferred))
            deferred.clear()

        def timed(n):
            times = []
            for _ in range(n):  # the context's host buffers are pinned as they are reused: steady from the 4th call on
                t0 = time.perf_counter()
                eng.sketch_files(paths, kmin, kmax, 0)
                times.append(time.perf_counter() - t0)
            steady = sorted(times[3:] if len(times) > 4 else times[1:])
            return steady[len(steady) // 2], steady[0]

        med, best = timed(reps)
        _, wait, batches, nbytes = eng.last_ingest_stats()
        # the same files through other decoders (environment switches of dd_sketch_files), e.g. the host's beside the device's
        other = {}
        for key, env, n in variants:
            saved = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                other[key] = ng * nb / timed(n)[0] / 1e9
            except Exception as e:
                other[key] = f"{type(e).__name__}: {e}"
            finally:
                for k, v in saved.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
        return {**other, "value": ng * nb / med / 1e9, "unit": "Gbp/s", "ms": med * 1e3, "best_value": ng * nb / best / 1e9, "best_ms": best * 1e3,
                "launches": batches, "fasta_MB": nbytes / 1e6,
                "what": f"dd_sketch_files: {ng} x {nb/1e6:g} Mbp {'BGZF (bgzip -6; blocks inflated on the GPU, dd_ginflate.hip)' if gz == 'bgzf' else 'gzip -6 (one member per file; inflated on the GPU in pieces, dd_ginflate.hip)' if gz == 'gzip6' else 'gzip -1 (one member per file; inflated on the GPU in pieces, dd_ginflate.hip, unless DD_NO_GPU_GUNZIP)' if gz else 'plain'} FASTA files in {base or 'the temp dir'} (warm page cache) -> "
                        f"pinned host buffers -> H2D on a copy stream overlapped with K0/K1 -> registers back to the host; "
                        + ("[four-line FASTQ, 150-base reads, gzip -6: inflated AND resolved on the device, dd_fastq.hip] " if gz == "fastq" else
                           "[two gzip -6 members per file] " if gz == "members" else "") +
                        f"k {kmin}-{kmax}; `value` = MEDIAN of calls {'4-' + str(reps) if reps > 4 else '2-' + str(reps)} on one context, best beside it "
                        f"(PCIe-inclusive: reported beside the headline `value`, never as it)"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def spawn_ranks(n, argv, backend_env=None):
    """`python3 bench.py --gpus N` without a launcher: start the N ranks ourselves, as FRESH child processes of a
    parent that has not imported torch or touched HIP (a process that has initialised the GPU must never exec or be
    replaced on this pool), through the same `python -m torch.distributed.run` line the driver would use.  Rank 0's
    JSON line is relayed on stdout, everything else the ranks print goes to stderr; exit code = the launcher's."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("OMP_NUM_THREADS", "1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this pool
    env["DD_BENCH_SPAWNED"] = "1"
    env.update(backend_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for l in r.stdout.splitlines():
        if l.startswith("{") and '"metric"' in l:
            line = l
        else:
            print(l, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    elif r.returncode == 0:
        print("bench.py: the ranks exited 0 without a result line", file=sys.stderr)
        return 1
    return r.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg2")
    ap.add_argument("--genomes", type=int, default=None)
    ap.add_argument("--mbp", type=float, default=None)
    ap.add_argument("--kmin", type=int, default=None)
    ap.add_argument("--kmax", type=int, default=None)
    ap.add_argument("--log2m", type=int, default=14)
    ap.add_argument("--nrec", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-accuracy", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-ingest", action="store_true")
    ap.add_argument("--cpu-sample-mbp", type=float, default=256.0)
    ap.add_argument("--share-mbp", type=float, default=None, help="size of the genomes of secondary.cfg5share_* (default 3000: the real share)")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"),
                    help="sidecar with the FULL result object (the stdout line is its compact form and names this file)")
    ap.add_argument("--abi-comm", action="store_true",
                    help="data-path collectives through the C ABI (dd_allreduce_max_u8 / dd_allgather_u8: RCCL called by "
                         "libdandd_hip.so) instead of torch.distributed, which then only carries the communicator's id")
    ap.add_argument("--force-dist", action="store_true",
                    help="run through the launcher and a process group even at --gpus 1 (world size 1): the RCCL "
                         "all-reduce / all-gather of the N>1 path execute in librccl on a one-GPU box")
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    for key in ("genomes", "mbp", "kmin", "kmax", "nrec"):
        if getattr(args, key) is not None:
            cfg[key] = getattr(args, key)
    headline = args.config == "cfg2"

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ   # under torch.distributed.run (ours or the driver's)
    if launched and args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not launched and (args.gpus > 1 or args.force_dist):
        # no launcher around us: be the launcher (nothing in this process has touched torch or HIP yet)
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    use_group = world > 1 or args.force_dist

    # CPU baseline first (rank 0, N=1 only), before this process touches the GPU
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(int(args.cpu_sample_mbp * 1e6), cfg["nrec"], cfg["kmin"], cfg["kmax"], args.log2m)

    import torch
    import torch.distributed as dist
    from dandd_amd import dist as ddist
    from dandd_amd.engine import Engine, KERNEL_PACK, KERNEL_SWEEP, KERNEL_UNION

    # Functional test of the N>1 path on a box with fewer GPUs than ranks (never a measurement):
    # DD_BENCH_BACKEND=gloo DD_BENCH_SHARE_DEVICE=1 puts every rank on cuda:0 and reduces through gloo.
    backend = os.environ.get("DD_BENCH_BACKEND", "nccl")
    if os.environ.get("DD_BENCH_SHARE_DEVICE"):
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{local_rank} but this node shows {torch.cuda.device_count()} GPU(s) "
                         f"(--gpus {args.gpus})")
    torch.cuda.set_device(local_rank)
    if use_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    nb = int(cfg["mbp"] * 1e6)
    kmin, kmax, p = cfg["kmin"], cfg["kmax"], args.log2m
    K, m = kmax - kmin + 1, 1 << p
    # strong: the job's genomes sharded over the ranks (equal sizes: round robin by weight); weak: every rank brings its own
    ids = rank_shards(cfg, world)[0][rank]
    total_genomes = cfg["genomes"] if cfg["strong"] else cfg["genomes"] * world
    eng = Engine(device=local_rank, log2m=p, canonical=True)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    wl = Workload(torch, eng, ids, nb, cfg["nrec"], kmin, kmax)
    ng, nbytes = wl.ng, wl.nbytes
    if args.abi_comm and use_group:
        ddist.use_abi_comm(eng)

    # progressive / pairwise schedules run over the JOB's leaves: with several ranks the leaf slabs are all-gathered
    # first and the orderings split over the ranks (cfg 4 is a 4-GPU config)
    n_sched = total_genomes if (cfg["strong"] and world > 1) else ng
    gather = bool(cfg["extra"]) and (n_sched != ng or args.force_dist)   # leaf slabs go through the all-gather
    orderings = None
    if cfg["extra"] == "progressive":
        # committed fixtures for the 8-genome share and for cfg 4 itself (30 genomes); other sizes: seeded permutations
        fixture = {8: "cfg4_orderings.json", 30: "cfg4_orderings_n30.json"}.get(n_sched)
        if fixture:
            with open(os.path.join(ROOT, "tests", "golden", fixture)) as f:
                orderings = [o for o in json.load(f)["orderings"]]
        else:
            rng = np.random.default_rng(42)
            orderings = [list(map(int, rng.permutation(n_sched))) for _ in range(10)]
        orderings = orderings[rank::world] if n_sched != ng else orderings

    def step():
        out = wl.step(ddist)
        slab, n = wl.regs, ng
        if gather:
            slab, n = ddist.allgather_leaves(wl.regs[:ng], ids if cfg["strong"] else list(range(ng)), n_sched), n_sched
        wl.sched = (slab, n)      # (what the schedule ran on: the K2 roofline pass below reuses it, no second collective)
        if cfg["extra"] == "pairwise" and n:
            wl.pair = eng.pairwise_device(slab.data_ptr(), n, K)                 # all pairs x all k
        elif cfg["extra"] == "progressive" and n and orderings:
            wl.prog = eng.progressive_device(slab.data_ptr(), n, K, orderings)   # this rank's orderings, every prefix, all k
        return out

    def fence():
        torch.cuda.synchronize()
        if use_group:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    eng.timing_enable(True)
    eng.timing_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        delta, bestk, card = step()
    fence()
    dt = time.perf_counter() - t0
    sweep_ms, sweep_n = eng.timing_read(KERNEL_SWEEP)
    pack_ms, pack_n = eng.timing_read(KERNEL_PACK)
    union_ms, union_n = eng.timing_read(KERNEL_UNION)
    eng.timing_enable(False)
    dt = ddist.max_over_ranks(dt, device="cuda")
    # which GPU every rank really ran on (rank 0 prints them: one entry per rank, all different on a real node)
    gpus_active = ddist.gather_strings(f"cuda:{local_rank} {torch.cuda.get_device_name(local_rank)} "
                                       f"uuid={getattr(torch.cuda.get_device_properties(local_rank), 'uuid', '?')}")

    extras = {}
    if cfg["extra"] == "progressive" and getattr(wl, "prog", None) is not None and n_sched == total_genomes:
        # every ordering's last prefix is the union of all the job's genomes: its cardinalities must be the root's
        ok = bool(np.array_equal(wl.prog[:, -1, :], np.broadcast_to(np.asarray(card[ng]), wl.prog[:, -1, :].shape)))
        extras["schedule"] = {"kind": "progressive", "genomes": n_sched, "orderings_this_rank": len(orderings),
                              "last_prefix_equals_root": ok}
        if not ok:
            raise SystemExit("bench.py: progressive schedule's full union differs from the root sketch")
    if rank == 0 and world == 1 and headline:
        if not args.no_accuracy:
            extras["accuracy_vs_exact"] = accuracy_block(
                wl, card, f"log2m {p}: all {ng} leaves + their root, k {kmin}-{kmax}; exact = GPU sort+distinct of canonical "
                          "k-mers (dd_exact_count_device, the KMC --exact stand-in)")
        if not args.no_secondary:
            sec = {}

            def timed_steps(p2, gids, nb2, nrec2, k0, k1, why, warm=2, reps=3, realistic=False, counters_key=None, accuracy=None):
                """the same step (dist.sharded_ksweep on an engine of its own) on another register count / genome set"""
                e2 = Engine(device=local_rank, log2m=p2, canonical=True)
                try:
                    e2.set_stream(torch.cuda.current_stream().cuda_stream)
                    w2 = Workload(torch, e2, gids, nb2, nrec2, k0, k1, realistic=realistic)
                    for _ in range(warm):
                        w2.step(ddist)
                    torch.cuda.synchronize()
                    e2.timing_enable(True)
                    e2.timing_reset()
                    t1 = time.perf_counter()
                    for _ in range(reps):
                        _, bk2, card2 = w2.step(ddist)
                    torch.cuda.synchronize()
                    d2 = (time.perf_counter() - t1) / reps
                    sw2 = e2.timing_read(KERNEL_SWEEP)
                    e2.timing_enable(False)
                    entry = {"why": why, "value": len(gids) * nb2 / d2 / 1e9, "unit": "Gbp/s", "ms_per_step": d2 * 1e3, "steps": reps}
                    if counters_key:
                        entry["roofline"] = k1_roofline(len(gids), w2.nbytes, nb2, k1 - k0 + 1, 1 << p2, p2, k0, k1, sw2[0], sw2[1], reps, *counters_key)
                    if accuracy == "all":
                        entry["accuracy_vs_exact"] = accuracy_block(w2, card2, f"log2m {p2}, same genomes and k range")
                    elif accuracy == "subsample":
                        entry["accuracy_subsample"] = subsample_accuracy(e2, w2, card2, int(bk2[0]), p2)
                    del w2
                    return entry
                finally:
                    e2.close()
                    torch.cuda.empty_cache()

            def guarded(name, *a, **kw):      # (a secondary figure must never cost the headline line)
                try:
                    sec[name] = timed_steps(*a, **kw)
                except Exception as e:
                    sec[name] = {"error": f"{type(e).__name__}: {e}"}

            for p2, why in ((16, "the register count at which delta meets the 1 % target"),
                            (20, "DandD's default -r 20 (/root/reference/lib/dandd_cmd.py:187): registers in HBM, scatter + sort + replay")):
                if p2 != p:
                    guarded(f"log2m{p2}", p2, ids, nb, cfg["nrec"], kmin, kmax, why, counters_key=(cfg["genomes"], cfg["mbp"]),
                            accuracy="all" if p2 == 16 and not args.no_accuracy else None)
            # ... and what a bacterial collection at DandD's defaults is: many small genomes at -r 20 (4.8 tokens per
            # register: nearly every update becomes a record; a regime of its own)
            guarded("log2m20_64x5Mbp", 20, list(range(64)), 5_000_000, cfg["nrec"], kmin, kmax, "many small genomes at DandD's default -r 20",
                    counters_key=(64, 5.0))
            # ... and input that is not i.i.d. uniform (the best case of the k <= 9 "set complete" exit and of every
            # spread assumption): GC 35 %, 30 % repeats, 2 % N, contigs of 2..200 kbp; same sizes, log2m 14 and 20
            for p4 in (14, 20):
                guarded(f"realistic_log2m{p4}", p4, ids, nb, cfg["nrec"], kmin, kmax,
                        "GC 35 %, 20 % interspersed + 10 % tandem repeats (soft-masked), 2 % N, contigs of 2-200 kbp (dd_synth.hip); the same 10 x 50 Mbp step",
                        realistic=True)
            # ... and THE NORTH-STAR WORKLOAD: one GPU's share of BASELINE cfg 5 (100 x 3 Gbp over 8 GPUs = 13 genomes, k 4-64, K = 61),
            # at log2m 16 (where delta is within 1 % of exact) and at DandD's default log2m 20; generated on the device, 39.5 GB of
            # FASTA resident in HBM; delta of genome 0 against the GPU exact counter at the argmax-k window
            share = CONFIGS["cfg5share"]
            share_nb = int((args.share_mbp if args.share_mbp is not None else share["mbp"]) * 1e6)
            for p5 in (16, 20):
                guarded(f"cfg5share_log2m{p5}", p5, list(range(share["genomes"])), share_nb, share["nrec"], share["kmin"], share["kmax"],
                        f"BASELINE cfg 5's per-GPU share (13 x {share_nb/1e9:g} Gbp, k 4-64) at log2m {p5}", warm=1, reps=2,
                        counters_key=(share["genomes"], share_nb / 1e6), accuracy=None if args.no_accuracy else "subsample")
            extras["secondary"] = sec
        if not args.no_ingest:
            extras["ingest"] = ingest_probe(eng, ng, nb, cfg["nrec"], kmin, kmax, torch)
            # the cfg 3 shape of the same path: many small files coalesced into a few launches
            extras["ingest"]["small_files"] = ingest_probe(eng, 64, 5_000_000, cfg["nrec"], kmin, kmax, torch)
            # ... and as most genome directories really are: .gz, ONE gzip member per file.  Round 4: inflated on the GPU too
            # (dd_ginflate.hip: block starts found by trial, pieces decoded without their history, placeholders resolved along a
            # chain of windows); DD_NO_GPU_GUNZIP=1 beside it = the host decoder (libdeflate or zlib, one thread per file)
            host = lambda key, sw, n=5: (key, {sw: "1"}, n)
            extras["ingest"]["gzip_files"] = ingest_probe(eng, ng, nb, cfg["nrec"], kmin, kmax, torch, gz=True, variants=[host("host_decoder_value", "DD_NO_GPU_GUNZIP")])
            # ... and bgzip'd (htslib's blocked gzip): independent <= 64 KiB members, inflated on the GPU (dd_ginflate.hip) -- the
            # compressed bytes cross PCIe, the host only walks the block sizes; DD_NO_GPU_INFLATE=1 beside it = the host decoder
            # ... and (round 5) what round 4 still sent to the host decoder: four-line FASTQ in .gz (inflated AND resolved on the
            # device, dd_fastq.hip), and files of several gzip members (`cat a.fa.gz b.fa.gz`).  (Round 6: all `ng` files like the other
            # probes -- rounds 4 and 5 took four, because zlib -6 in one Python thread is slow, and a call over four files is two
            # batches of two: mostly the pipeline filling; the files are now compressed by a thread each.)
            for key, mode, sw, n_files in (("bgzf_files", "bgzf", "DD_NO_GPU_INFLATE", ng), ("gzip_fastq_files", "fastq", "DD_NO_GPU_FASTQ", ng),
                                           ("multi_member_gzip_files", "members", "DD_NO_GPU_GUNZIP", ng)):
                try:
                    extras["ingest"][key] = ingest_probe(eng, n_files, nb, cfg["nrec"], kmin, kmax, torch, gz=mode, reps=8 if key != "bgzf_files" else 10,
                                                         variants=[host("host_decoder_value", sw, 4)])
                except Exception as e:
                    extras["ingest"][key] = {"error": f"{type(e).__name__}: {e}"}
            # ... and at STEADY STATE (round 6): 8 x 250 Mbp = 2 GB of text, plain and as plain `gzip` writes it (level 6, one member) --
            # the ten-file probes above are 20-40 ms calls, a third of which is the pipeline filling and draining
            if args.config == "cfg2" and args.genomes is None and args.mbp is None:
                try:
                    extras["ingest"]["big_dir"] = ingest_probe(eng, 8, 250_000_000, 1, kmin, kmax, torch, reps=6)
                    extras["ingest"]["big_dir_gzip"] = ingest_probe(eng, 8, 250_000_000, 1, kmin, kmax, torch, gz="gzip6", reps=6)
                except Exception as e:
                    extras["ingest"].setdefault("big_dir", {"error": f"{type(e).__name__}: {e}"})
                    extras["ingest"].setdefault("big_dir_gzip", {"error": f"{type(e).__name__}: {e}"})
            # ... and ONE large .gz (a whole assembly as NCBI ships it: a single gzip member): its deflate stream is cut at
            # block boundaries and the pieces are decoded in parallel without their history -- on the GPU (dd_ginflate.hip), by
            # the host's loader threads (dd_inflate.h), or serially by one thread (libdeflate)
            try:
                big_nb = max(40_000_000, 8 * nb)     # 400 Mbp for the headline workload
                extras["ingest"]["one_big_gzip_file"] = ingest_probe(
                    eng, 1, big_nb, 24, kmin, kmax, torch, gz=True, reps=4,
                    variants=[("host_parallel_decoder_value", {"DD_NO_GPU_GUNZIP": "1"}, 4), ("serial_decoder_value", {"DD_NO_GPU_GUNZIP": "1", "DD_NO_PARALLEL_GZIP": "1"}, 3)])
            except Exception as e:
                extras["ingest"]["one_big_gzip_file"] = {"error": f"{type(e).__name__}: {e}"}

    # (HBM-side traffic of K1 per step: PMC counters cannot be read from inside this process, so the number comes from the
    # committed rocprofv3 passes over this very workload -- k1_roofline / load_counters -- and is only reported when this
    # run's workload is a profiled one)
    # the K2 schedule of the config (all pairs / progressive) timed on its own, against its own roofline
    k2 = None
    if rank == 0 and cfg["extra"] and ng:
        slab, n = wl.sched        # rank 0 alone from here on: no collective may be called
        run = (lambda: eng.pairwise_device(slab.data_ptr(), n, K)) if cfg["extra"] == "pairwise" else \
              (lambda: eng.progressive_device(slab.data_ptr(), n, K, orderings))
        run()
        eng.timing_enable(True)
        eng.timing_reset()
        for _ in range(3):
            run()
        k2_ms = eng.timing_read(KERNEL_UNION)[0] / 3
        eng.timing_enable(False)
        path = eng.last_k2_path()     # which device form ran (dd_last_k2_path): never assumed from log2m or n here
        hist_bytes = 256
        lo = slab[:n].amin(dim=(0, 2)).cpu().numpy().astype(int)
        hi = slab[:n].amax(dim=(0, 2)).cpu().numpy().astype(int)
        thresholds = [int(b - a) for a, b in zip(lo, hi)]        # per k: the register values between which a threshold says something

        def k2_traffic(tag):
            for rnd in ("r04", "r03"):
                try:
                    with open(os.path.join(ROOT, "profiles", f"{rnd}_k2_counters_{tag}_p{p}.json")) as f:
                        kj = json.load(f)
                    if (kj["workload"]["genomes"], kj["workload"]["K"], kj["workload"]["log2m"]) == (n, K, p) and kj["workload"].get("path", path) == path:
                        return kj["bytes_per_launch"]["total"], f"profiles/{rnd}_k2_counters_{tag}_p{p}.json"
                except (OSError, KeyError, ValueError):
                    pass
            return None, None

        if cfg["extra"] == "pairwise" and path == "pairwise_gram":
            ns = (n + 63) // 64
            blocks = ns * 3 + ns * (ns - 1) // 2 * 4                      # 32 x 32 blocks per (k, threshold, 32 registers)
            if n > 64:    # 128-row diagonal units (10 blocks) hold the pairs (2u, 2u + 1)
                blocks = (ns + 1) // 2 * 10 + (ns * (ns - 1) // 2 - ns // 2) * 4
            mfma = sum(thresholds) * (m // 32) * blocks
            tops = 2.0 * 32 * 32 * 32 * mfma / (k2_ms / 1e3) / 1e12
            k2 = {"bound": "mfma", "path": path,
                  "kernel": "gram_kernel + range + finish + mle (dd_gram.hip: all pairs as int8 Gram matrices, v_mfma_i32_32x32x32_i8)",
                  "achieved": tops, "peak": 5000.0, "unit": "TOP/s", "frac": tops / 5000.0,
                  "peak_note": "int8 dense = 2 x the bf16 dense peak of MI355X_MICROARCH.md (2.5 PF) at 2.4 GHz; the chip holds ~2.06 GHz under this load",
                  "ms": k2_ms, "pairs": n * (n + 1) // 2, "thresholds_per_k": thresholds,
                  "useful_fraction_of_blocks": (n * (n + 1) / 2) / (blocks * 1024.0),
                  "compulsory_bytes": n * K * m + n * (n + 1) // 2 * K * hist_bytes}
            k2["traffic"], k2["traffic_from"] = k2_traffic("cfg3")
        elif cfg["extra"] == "progressive" and path == "progressive_pscan":
            # bit-plane scan: every (ordering, threshold) chain reads its 32 bytes of plane per prefix and 256 registers
            # from LDS -- that read stream is what the kernel is made of: priced against the LDS
            lds_bytes = float(len(orderings)) * n * sum(thresholds) * (m // 8)
            lds_peak = 128.0 * 256 * 2.4e9 / 1e9                         # 128 B per clock and CU, 256 CUs, 2.4 GHz: GB/s
            gbs = lds_bytes / (k2_ms / 1e3) / 1e9
            k2 = {"bound": "lds", "path": path,
                  "kernel": "pscan_kernel + register range + finish + mle (dd_pscan.hip: running AND of threshold bit planes, one popcount per prefix)",
                  "achieved": gbs, "peak": lds_peak, "unit": "GB/s", "frac": gbs / lds_peak, "ms": k2_ms,
                  "lds_bytes_read": lds_bytes, "thresholds_per_k": thresholds,
                  "algorithmic_bytes": len(orderings) * n * K * hist_bytes + n * K * m,
                  "note": "achieved = plane bytes the chains read from LDS (orderings x prefixes x thresholds x m / 8) over the launch time; peak = "
                          "ds_read_b128 rate of MI355X_MICROARCH.md (128 B/clk/CU).  HBM sees the leaf slab once (n x K x m) plus the histograms",
                  "pmc": "profiles/r03_k2_pscan_pmc_p20.txt (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 39-42 %: the gap to the roof)"}
            k2["traffic"], k2["traffic_from"] = k2_traffic("cfg4share" if n == 8 else "cfg4")
        else:
            stream_pairs = cfg["extra"] == "pairwise"
            units = n * (n + 1) // 2 if stream_pairs else len(orderings) * n
            nbytes_alg = units * K * m * (2 if stream_pairs else 1) + units * K * hist_bytes
            gbs = nbytes_alg / (k2_ms / 1e3) / 1e9
            k2 = {"bound": "hbm", "path": path,
                  "kernel": "pairwise_kernel (streaming: byte-max of two rows, one LDS histogram per pair)" if stream_pairs else
                            "progressive_kernel (running byte-max along every ordering, one LDS histogram per prefix)",
                  "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "ms": k2_ms,
                  "algorithmic_bytes": nbytes_alg,
                  "note": "bytes = the rows read per (pair | ordering, prefix) and k + the histograms written; the kernel is bound by one LDS atomic "
                          "per register per unit, not by HBM (DESIGN.md, K2)"}
            k2["traffic"], k2["traffic_from"] = k2_traffic(("cfg3" if stream_pairs else ("cfg4share" if n == 8 else "cfg4")))

    # BASELINE cfg 5's accuracy clause ("delta within 1 % of exact on a subsample"): genome 0 of the share
    if rank == 0 and args.config in ("cfg5", "cfg5share") and not args.no_accuracy and ng:
        extras["accuracy_subsample"] = subsample_accuracy(eng, wl, card, int(bestk[0]), p)

    if rank == 0:
        steps = args.steps
        total_bases = total_genomes * nb * steps
        # algorithmic bytes of one step on one GPU: FASTA read once + registers written once
        out = {
            "metric": "Gbp/s sketched over k-sweep",
            "value": total_bases / dt / 1e9,
            "unit": "Gbp/s",
            "n_gpus": world,
            "gpus_active": gpus_active,
            "collectives": dict(ddist.STATS, backend=(("rccl via the C ABI (dd_comm_*), id over " + dist.get_backend()) if (use_group and args.abi_comm)
                                                      else dist.get_backend() if use_group else None),
                                launcher=("bench.py self-spawn" if os.environ.get("DD_BENCH_SPAWNED") else
                                          "torch.distributed.run" if launched else None)),
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if cfg["strong"] else "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.config}: {ng} x {cfg['mbp']:g} Mbp synthetic FASTA on this GPU ({total_genomes} over {world} GPU(s)) resident in HBM, "
                            f"HLL log2m={p}, k-sweep {kmin}-{kmax} (K={K}), leaf sketches + root union + all cardinalities + delta"
                            + (f" + {cfg['extra']} schedule" if cfg["extra"] else "")
                            + ("; NOTE the metric's accuracy half (delta within 1 % of exact) is NOT met at log2m 14 (HLL sigma 0.81 %; "
                               "accuracy_vs_exact) and IS met at log2m 16 (secondary.log2m16, the same step)" if headline and p == 14 else ""),
                "genomes_per_gpu": ng, "bases_per_genome": nb, "kmin": kmin, "kmax": kmax, "log2m": p,
                "parallelism": f"genomes sharded over {world} GPU(s)" + ("; RCCL max all-reduce of the root" if use_group else ""),
            },
            "roofline": k1_roofline(ng, nbytes, nb, K, m, p, kmin, kmax, sweep_ms, sweep_n, steps, cfg["genomes"], cfg["mbp"]),
            "other_kernels_ms_per_step": {"pack_K0": pack_ms / steps, "union_hist_K2": union_ms / steps},
            "roofline_k2": k2,
            # the HBM-bound kernel of the path: FASTA bytes read once + 3 bits per base written
            # (algorithmic; K0 actually reads the FASTA twice)
            "roofline_k0": {"bound": "hbm", "kernel": "pack_stats + pack_scan + pack_write (K0)",
                            "achieved": (ng * nbytes + ng * nb * 0.375) / (pack_ms / 1e3 / steps) / 1e9 if pack_ms > 0 else None,
                            "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": (ng * nbytes + ng * nb * 0.375) / (pack_ms / 1e3 / steps) / 1e9 / HBM_PEAK_GBS if pack_ms > 0 else None,
                            "kernel_ms_per_step": pack_ms / steps},
            "delta_genome0": float(delta[0]) if ng else None, "argmax_k_genome0": int(bestk[0]) if ng else None,
            "delta_root": float(delta[ng]), "argmax_k_root": int(bestk[ng]),
        }
        out.update(extras)
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(compact_line(out, write_detail(out, os.path.abspath(args.detail))), flush=True)
    if use_group:
        dist.barrier()
        ddist.drop_abi_comm()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
