"""profiles/r03_k1_counters_p<P>.json from the rocprofv3 passes of scripts/profile_r03.sh over
`scripts/quick_bench.py 10 50e6 4 40 <P>` (= one bench.py cfg 2 step per iteration, 3 iterations).

FETCH_SIZE / WRITE_SIZE are in KiB and come from separate passes.  On gfx950 FETCH_SIZE reports half of the bytes of wide
(16 B per lane) coalesced reads (MI355X_MICROARCH.md, HBM): `fetch_bytes` below is 2 x FETCH_SIZE, the raw figure is kept
beside it.  usage: make_counters_json.py <dir with counters_p<P>_*.csv> <P> <out.json>"""
import collections
import csv
import json
import os
import re
import sys

d, P, dest = sys.argv[1], int(sys.argv[2]), sys.argv[3]
STEPS = 3                      # quick_bench iterations
NG, NB, KMIN, KMAX = 10, 50_000_000, 4, 40


def short(name):
    name = re.sub(r"\(anonymous namespace\)::|dd::|void ", "", name)
    m = re.match(r"([A-Za-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def load(tag):
    f = os.path.join(d, f"counters_p{P}_{tag}.csv")
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
K1 = ("sweep_kernel", "bitmap_kernel", "bitmap_finish_kernel", "scatter_kernel", "sort_chunks_kernel", "replay_kernel", "bigmap_kernel",
      "bigmap_finish_kernel", "bucket_", "cursor", "filter")
out = {"workload": {"genomes": NG, "mbp": NB / 1e6, "kmin": KMIN, "kmax": KMAX, "log2m": P},
       "made_by": "scripts/profile_r03.sh + scripts/make_counters_json.py (rocprofv3 --pmc, one counter set per pass)",
       "note": "per STEP (one dd_sketch_device call over 10 x 50 Mbp); fetch_bytes = 2 x FETCH_SIZE (gfx950 wide-read correction), "
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
cls = {"bitmap_kernel": (4, 9), "sweep_kernel<0": (10, 16), "sweep_kernel<1": (17, 32), "sweep_kernel<3": (33, 40)}
per = {}
for k, ent in out["kernels"].items():
    for pre, (lo, hi) in cls.items():
        if k.startswith(pre) and "sq_insts_valu" in ent:
            nk = hi - lo + 1
            per.setdefault(pre, {"k_lo": lo, "k_hi": hi, "valu_per_update": 0.0})
            per[pre]["valu_per_update"] += ent["sq_insts_valu"] / (NG * NB * nk / 64.0)
out["valu_per_update_by_class"] = per
json.dump(out, open(dest, "w"), indent=1)
print(json.dumps({"log2m": P, "k1_bytes_per_step": out["k1_bytes_per_step"], "k1_valu_instr_per_update": out["k1_valu_instr_per_update"],
                  "by_class": per}))
