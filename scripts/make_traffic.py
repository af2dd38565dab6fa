"""profiles/traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE collected separately,
as MI355X_MICROARCH.md prescribes) over `scripts/quick_bench.py 10 50e6` (= bench.py's default
workload).  FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide
(16 B/lane) coalesced reads, which is what K1 issues, so the fetch side is doubled."""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fetch_dir, write_dir = sys.argv[1], sys.argv[2]


def per_kernel(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "")
        key = re.split(r"[(]", name)[0].split("::")[-1]
        acc[key].append(float(r["Counter_Value"]) * 1024.0)
    return acc


fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
out = {"workload": {"genomes": 10, "mbp": 50.0, "kmin": 4, "kmax": 40, "log2m": 14},
       "note": "bytes per launch, mean over launches; fetch = 2 x FETCH_SIZE (gfx950 wide-read correction)",
       "kernels": {}}
steps = 3  # quick_bench runs 3 iterations
sweep_fetch = sweep_write = 0.0
for k in sorted(set(fetch) | set(write)):
    f = sum(fetch.get(k, [0])) / max(1, len(fetch.get(k, [0])))
    w = sum(write.get(k, [0])) / max(1, len(write.get(k, [0])))
    out["kernels"][k] = {"fetch_bytes": 2 * f, "write_bytes": w, "launches_seen": len(fetch.get(k, []))}
    if k.startswith("sweep_kernel") or k.startswith("bitmap_"):
        sweep_fetch += 2 * sum(fetch.get(k, [0])) / steps
        sweep_write += sum(write.get(k, [0])) / steps
out["k1_bytes_per_step"] = {"fetch": sweep_fetch, "write": sweep_write, "total": sweep_fetch + sweep_write}
dest = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "traffic.json")
with open(dest, "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps(out["k1_bytes_per_step"]))
