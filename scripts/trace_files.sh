cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/trace_files; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/raw -o t -- python3 scripts/files_probe.py 10 50e6 > $OUT/probe.txt 2>&1
python3 - <<PY
import csv, glob
kt = list(csv.DictReader(open(glob.glob("$OUT/raw/**/*kernel_trace.csv", recursive=True)[0])))
mc = list(csv.DictReader(open(glob.glob("$OUT/raw/**/*memory_copy_trace.csv", recursive=True)[0])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]) for r in kt]
ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Name", "")) ) for r in mc]
ev.sort()
# last call = last 'pack_stats' group: print the final ~70 events with times relative
t_end = ev[-1][1]
sel = [e for e in ev if e[0] > t_end - 45e6]
t0 = sel[0][0]
for a, b, n in sel:
    if (b - a) > 50e3: print(f"{(a-t0)/1e6:8.2f} -> {(b-t0)/1e6:8.2f}  {(b-a)/1e6:7.3f} ms  {n}")
PY
rm -rf $OUT/raw
