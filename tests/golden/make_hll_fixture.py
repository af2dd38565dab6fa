"""Byte-level fixture of Dashing's `.hll` container AS RECALLED (oracle/POLICIES.md P9), written without
importing anything from dandd_amd: every header field is spelled out with int.to_bytes / struct here, so that
dandd_amd/host/backend.py's reader and writer are pinned against bytes they did not produce.  The layout itself is
unverified (no Dashing binary or source in this image); what the fixture guarantees is that it cannot drift unnoticed.

    python tests/golden/make_hll_fixture.py      ->  tests/golden/hll/recalled_p10.hll (plain), recalled_p10.gz.hll (gzip)

Layout (little endian), 32 bytes then 2^np register bytes:
    u32 is_calculated   0: the cached estimate below is not valid
    u32 clamp           0
    u32 estimator       2 = ERTL_MLE  (dashing's default `card` estimator)
    u32 joint_estimator 3 = ERTL_JOINT_MLE
    u32 nthreads        1
    u32 np              log2 of the register count
    f64 value           cached estimate (0.0)
"""
import gzip
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))
NP = 10
regs = bytes((i * 2654435761 >> 7) % 23 if i % 5 else 0 for i in range(1 << NP))   # a fixed, uneven register pattern
head = b"".join(v.to_bytes(4, "little") for v in (0, 0, 2, 3, 1, NP)) + struct.pack("<d", 0.0)
assert len(head) == 32
os.makedirs(os.path.join(HERE, "hll"), exist_ok=True)
with open(os.path.join(HERE, "hll", "recalled_p10.hll"), "wb") as f:
    f.write(head + regs)
with open(os.path.join(HERE, "hll", "recalled_p10.gz.hll"), "wb") as f:
    f.write(gzip.compress(head + regs, compresslevel=6, mtime=0))
with open(os.path.join(HERE, "hll", "recalled_p10.registers"), "wb") as f:
    f.write(regs)
print("wrote", 32 + len(regs), "bytes")
