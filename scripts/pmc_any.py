"""Sum rocprofv3 --pmc counters per kernel over a whole run (development aid).
usage: pmc_any.py <counter_collection.csv> [name filter]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::|dd::|void ", "", r["Kernel_Name"])
    name = re.split(r"\(", name)[0][:48]
    if flt and flt not in name:
        continue
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[name].add(r["Dispatch_Id"])
for name in sorted(agg):
    print(f"{name}  ({len(calls[name])} dispatches)")
    for k, v in sorted(agg[name].items()):
        print(f"    {k:28s} {v:.5g}")
