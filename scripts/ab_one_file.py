import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
from dandd_amd.engine import Engine
from oracle import dd_oracle as orc
d = "/dev/shm/ab_one"; os.makedirs(d, exist_ok=True)
for p in (14, 20):
    eng = Engine(0, p, True)
    for mbp in (0.02, 5, 50, 250):
        f = os.path.join(d, f"g{mbp}.fa"); orc.synth_fasta(5, 0, int(mbp * 1e6), 3).tofile(f)
        for name, fn in (("sketch_fasta", lambda: eng.sketch_fasta(f, 4, 40)), ("sketch_files", lambda: eng.sketch_files([f], 4, 40))):
            fn(); fn()
            ts = []
            for _ in range(7):
                t = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t)
            print(f"log2m {p} {mbp:>6} Mbp {name:13s} median {1e3 * sorted(ts)[3]:8.2f} ms  best {1e3 * min(ts):8.2f}")
        a, b = eng.sketch_fasta(f, 4, 40), eng.sketch_files([f], 4, 40)
        assert np.array_equal(np.asarray(a).reshape(-1), np.asarray(b).reshape(-1))
