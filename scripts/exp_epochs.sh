run() { P=$1; shift; echo "== log2m $P $*"; env "$@" python scripts/quick_bench.py 10 50e6 4 40 $P | sed -n 3p; }
# (quick_bench enables timing spans: side streams are off there; use wall of a plain loop instead)
t() { P=$1; shift; env "$@" python - <<PY
import sys, time, torch
sys.path.insert(0, ".")
from dandd_amd.engine import Engine, synth_size
eng = Engine(0, $P, True); nb = 50_000_000; n = synth_size(nb, 5)
bufs = []
for g in range(10):
    b = torch.empty(n + 16, dtype=torch.uint8, device="cuda"); eng.synth_fasta_device(0xD4ADD, g, nb, 5, b.data_ptr()); bufs.append(b)
regs = torch.empty((10, 37, 1 << $P), dtype=torch.uint8, device="cuda"); eng.synchronize()
ts=[]
for it in range(7):
    t0 = time.time(); eng.sketch_device([b.data_ptr() for b in bufs], [n] * 10, 4, 40, regs.data_ptr()); eng.synchronize(); ts.append(time.time() - t0)
dt=min(ts[2:]); print(f"log2m $P $*: {dt*1e3:.2f} ms  {0.5/dt:.2f} Gbp/s")
PY
}
t 20 X=1
t 20 DD_BUCKET_E0=16
t 20 DD_BUCKET_E0=48
t 20 DD_BUCKET_E0=64
t 20 DD_BUCKET_GB=32
t 20 DD_BUCKET_EMAX=128
t 20 DD_BUCKET_FBITS=8
t 18 X=1
t 18 DD_BUCKET_E0=8
t 18 DD_BUCKET_E0=32
t 19 X=1
t 19 DD_BUCKET_E0=32
