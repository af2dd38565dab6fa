"""Where a fresh process spends its time before and inside its first dd_sketch_files calls (scripts/prof_bringup.sh).
  python scripts/init_probe.py LOG2M [DIR_WITH_FASTAS]      (DANDD_NO_TORCH=1 for the one-shot CLI's conditions)"""
import os, sys, time, glob
t0 = time.time()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
t1 = time.time()
from dandd_amd.engine import Engine
t2 = time.time()
eng = Engine(0, int(sys.argv[1]) if len(sys.argv) > 1 else 14, True)
t3 = time.time()
fa = np.frombuffer(b">w\n" + b"ACGTTGCAACGGTCA" * 16 + b"\n", dtype=np.uint8)
r = eng.sketch_buffer(fa, 15, 17)
t4 = time.time()
eng.card_batch(r)
t5 = time.time()
r = eng.sketch_buffer(fa, 4, 40)
t6 = time.time()
print(f"numpy import {1e3*(t1-t0):.0f} ms, engine import {1e3*(t2-t1):.0f}, Engine() {1e3*(t3-t2):.0f}, first sketch_buffer k15-17 {1e3*(t4-t3):.0f}, first card {1e3*(t5-t4):.0f}, sketch k4-40 {1e3*(t6-t5):.0f}")
files = sorted(glob.glob(sys.argv[2] + "/*.fasta")) if len(sys.argv) > 2 else []
if files:
    for it in range(3):
        t = time.time()
        eng.sketch_files(files, 4, 40)
        print(f"sketch_files call {it}: {1e3*(time.time()-t):.0f} ms")
print(f"total in-process {1e3*(time.time()-t0):.0f} ms", flush=True)
if os.environ.get("PROBE_FAST_EXIT"):
    os._exit(0)
