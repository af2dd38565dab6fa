#!/bin/bash
# scripts/ubench_lds_read.hip: time per pattern, then SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per pattern.  pmc_lds_read.sh OUTNAME
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-lds_read}; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result scripts/ubench_lds_read.hip -o /tmp/ubench_lds_read || exit 1
/tmp/ubench_lds_read | tee $OUT/times.txt
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/raw -o x -- /tmp/ubench_lds_read > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/raw/**/*counter_collection.csv", recursive=True)[0]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    key = (r["Kernel_Name"], r["Dispatch_Id"])
    acc.setdefault(key, {})[r["Counter_Name"]] = acc.get(key, {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
seen = {}
for (name, disp), c in acc.items():
    seen[name] = c          # the second (long) launch of every mode overwrites the warm-up
with open("$OUT/counters.txt", "w") as out:
    for name, c in seen.items():
        line = f'{name[:40]:40s} ' + "  ".join(f"{k} {v:.3e}" for k, v in sorted(c.items())) + f'   conflict/active {c.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, c.get("SQ_LDS_IDX_ACTIVE", 0)):.3f}'
        print(line); out.write(line + "\n")
PY
rm -rf $OUT/raw
