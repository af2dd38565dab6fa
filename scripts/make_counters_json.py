"""profiles/r0N_k1_counters_*.json from the rocprofv3 passes of scripts/profile_r03.sh / profile_r04.sh over
`scripts/quick_bench.py NG NB KMIN KMAX <P> [NREC]` (= one bench.py step per iteration, 3 iterations; default workload
10 x 50 Mbp, k 4-40 = cfg 2).

FETCH_SIZE / WRITE_SIZE are in KiB and come from separate passes.  On gfx950 FETCH_SIZE reports half of the bytes of wide
(16 B per lane) coalesced reads (MI355X_MICROARCH.md, HBM): `fetch_bytes` below is 2 x FETCH_SIZE, the raw figure is kept
beside it.  usage: make_counters_json.py <dir with counters_p<P>_*.csv> <P> <out.json> [NG NB KMIN KMAX [TAG]]
(TAG: the files are counters_<TAG>_*.csv instead of counters_p<P>_*.csv)"""
import collections
import csv
import json
import os
import re
import sys

d, P, dest = sys.argv[1], int(sys.argv[2]), sys.argv[3]
STEPS = 3                      # quick_bench iterations
NG, NB, KMIN, KMAX = 10, 50_000_000, 4, 40
if len(sys.argv) >= 8:
    NG, NB, KMIN, KMAX = int(sys.argv[4]), int(float(sys.argv[5])), int(sys.argv[6]), int(sys.argv[7])
TAG = sys.argv[8] if len(sys.argv) >= 9 else f"p{P}"


def short(name):
    name = re.sub(r"\(anonymous namespace\)::|dd::|void ", "", name)
    m = re.match(r"([A-Za-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def load(tag):
    f = os.path.join(d, f"counters_{TAG}_{tag}.csv")
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    if not os.path.exists(f):
        return acc, disp, dur
    seen = set()
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
        if (r["Dispatch_Id"]) not in seen:
            seen.add(r["Dispatch_Id"])
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    return acc, disp, dur


fetch, fdisp, _ = load("FETCH_SIZE")
write, _, _ = load("WRITE_SIZE")
sq, sdisp, sdur = load("SQ_INSTS_VALU")
tcc, _, _ = load("TCC_HIT_sum")
K1 = ("sweep_kernel", "bitmap_kernel", "bitmap_finish_kernel", "scatter_kernel", "scatter_first", "sort_chunks_kernel", "replay_kernel", "bigmap_kernel",
      "bigmap_finish_kernel", "bucket_", "cursor", "filter")
out = {"workload": {"genomes": NG, "mbp": NB / 1e6, "kmin": KMIN, "kmax": KMAX, "log2m": P},
       "made_by": "scripts/profile_k1_counters.sh (profile_r04.sh / profile_r03.sh in rounds 4 / 3) + scripts/make_counters_json.py (rocprofv3 --pmc, one counter set per pass)",
       "note": f"per STEP (one dd_sketch_device call over {NG} x {NB / 1e6:g} Mbp); fetch_bytes = 2 x FETCH_SIZE (gfx950 wide-read correction), "
               "fetch_raw_bytes as counted; synth/pack/union kernels listed but not summed into k1",
       "kernels": {}}
k1_fetch = k1_write = k1_valu = 0.0
for k in sorted(set(fetch) | set(write) | set(sq)):
    if k.startswith("synth") or k.startswith("__amd"):
        continue
    f = fetch[k].get("FETCH_SIZE", 0.0) * 1024.0 / STEPS
    w = write[k].get("WRITE_SIZE", 0.0) * 1024.0 / STEPS
    ent = {"launches_per_step": len(fdisp.get(k, sdisp.get(k, []))) / STEPS, "fetch_raw_bytes": f, "fetch_bytes": 2 * f, "write_bytes": w}
    if k in sq:
        ent.update({c.lower(): v / STEPS for c, v in sq[k].items()})
        ent["ms_per_step_in_pmc_run"] = sdur[k] / STEPS
    if k in tcc:
        ent.update({c: v / STEPS for c, v in tcc[k].items()})
    out["kernels"][k] = ent
    if any(k.startswith(x) for x in K1):
        k1_fetch += 2 * f
        k1_write += w
        k1_valu += sq[k].get("SQ_INSTS_VALU", 0.0) / STEPS
out["k1_bytes_per_step"] = {"fetch": k1_fetch, "write": k1_write, "total": k1_fetch + k1_write}
updates = NG * NB * (KMAX - KMIN + 1)
out["k1_valu_wave_instr_per_step"] = k1_valu
out["k1_valu_instr_per_update"] = k1_valu / (updates / 64.0)   # SQ_INSTS_VALU counts wave instructions; 64 updates per wave-step
# per k class where the kernels are per class (log2m <= 16: sweep_kernel<KC, ...>): k ranges of the classes
cls = {"bitmap_kernel": (KMIN, min(KMAX, 9)), "sweep_kernel<0": (max(KMIN, 10), min(KMAX, 16)), "sweep_kernel<1": (max(KMIN, 17), min(KMAX, 32)),
       "sweep_kernel<3": (max(KMIN, 33), min(KMAX, 48)), "sweep_kernel<2": (max(KMIN, 49), min(KMAX, 64))}
cls = {k: v for k, v in cls.items() if v[0] <= v[1]}
per = {}
for k, ent in out["kernels"].items():
    for pre, (lo, hi) in cls.items():
        if k.startswith(pre) and "sq_insts_valu" in ent:
            nk = hi - lo + 1
            per.setdefault(pre, {"k_lo": lo, "k_hi": hi, "valu_per_update": 0.0, "ms_per_step_in_pmc_run": 0.0})
            per[pre]["valu_per_update"] += ent["sq_insts_valu"] / (NG * NB * nk / 64.0)
            per[pre]["ms_per_step_in_pmc_run"] += ent.get("ms_per_step_in_pmc_run", 0.0)
out["valu_per_update_by_class"] = per
json.dump(out, open(dest, "w"), indent=1)
print(json.dumps({"log2m": P, "k1_bytes_per_step": out["k1_bytes_per_step"], "k1_valu_instr_per_update": out["k1_valu_instr_per_update"],
                  "by_class": per}))
