# where does a fresh `dandd tree` process spend its first dd_sketch_files call?
D=$(mktemp -d /tmp/ddcli_XXXX); mkdir -p $D/g
python - <<PY
import torch, os
from dandd_amd.engine import Engine, synth_size
eng = Engine(0, 14, True); n = synth_size(50_000_000, 5)
buf = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
for g in range(10):
    eng.synth_fasta_device(0xD4ADD, g, 50_000_000, 5, buf.data_ptr()); eng.synchronize()
    buf[:n].cpu().numpy().tofile("$D/g/g%03d.fasta" % g)
PY
for i in 1 2; do
/usr/bin/time -f "wall %e s" env DD_TRACE_FILES=1 python -m dandd_amd.host.cli tree -d $D/g -o $D/o$i -s e -r 14 --ksweep --mink 4 --maxk 40 2>&1 | grep -v "^\[dd_sketch_files\] t=\|saved to\|amdgpu"
done
python - <<PY
import time, os, sys
os.environ["DANDD_NO_TORCH"]="1"
t0=time.time()
from dandd_amd.engine import Engine
import numpy as np
t1=time.time(); eng=Engine(0,14,True); t2=time.time()
paths=sorted("$D/g/"+f for f in os.listdir("$D/g"))
r=eng.sketch_files(paths,4,32); t3=time.time()
r=eng.sketch_files(paths,4,32); t4=time.time()
print(f"import {t1-t0:.3f} engine {t2-t1:.3f} first sketch_files {t3-t2:.3f} second {t4-t3:.3f}")
PY
rm -rf $D
