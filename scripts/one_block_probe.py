import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np, torch, bench
from dandd_amd.engine import Engine
from oracle import dd_oracle as orc
eng = Engine(0, 14, True)
raw = orc.synth_fasta(0xD4ADD, 0, 3_000_000, 4).tobytes()
d = tempfile.mkdtemp()
for name, n in (("one", 65280), ("eight", 8 * 65280), ("big", 40 * 65280)):
    p = os.path.join(d, name + ".fa.gz")
    open(p, "wb").write(bench.bgzf_bytes(raw[:n]))
    for _ in range(3):
        eng.sketch_files([p], 20, 21)
print("done")
