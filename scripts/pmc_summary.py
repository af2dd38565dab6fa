"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (development aid)."""
import csv, collections, re, sys
f = sys.argv[1]
nb = float(sys.argv[2]) if len(sys.argv) > 2 else 500e6
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name']
    key = re.sub(r'\(anonymous namespace\)::', '', name)
    key = re.split(r'[<(]', key.replace('void ', ''))[0].split('::')[-1][:40]
    if 'sweep_kernel<' in name:
        key = 'sweep' + name.split('sweep_kernel<')[1][0]
    agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
    dur[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
upd = {'sweep0': nb * 7, 'sweep1': nb * 16, 'sweep3': nb * 8, 'sweep2': nb * 8, 'bitmap_kernel': nb * 6}
for kc in sorted(agg):
    d = {k: sum(v) / len(v) for k, v in agg[kc].items()}
    print(kc, "avg ms (pmc run) %.3f" % (sum(dur[kc]) / len(dur[kc])))
    for k, v in d.items():
        extra = f"  per wave-update {v / (upd[kc] / 64):.2f}" if kc in upd else ""
        print(f"  {k:24s} {v:.4g}{extra}")
