"""Which kind of BGZF file does the device decoder refuse?  (DD_INFLATE_STRICT=1: no host fallback)"""
import os, sys, zlib, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dandd_amd.engine import Engine, EngineError
from oracle import dd_oracle as orc
sys.path.insert(0, "tests")
os.environ["DD_INFLATE_STRICT"] = "1"
SEED = 0xD4ADD
def bgzf(raw, level=1, strategy=0, block=65280):
    out = bytearray()
    for a in list(range(0, len(raw), block)) + [len(raw)]:
        part = raw[a:a + block] if a < len(raw) else b""
        c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        body = c.compress(part) + c.flush()
        out += (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + (len(body) + 25).to_bytes(2, "little") + body +
                zlib.crc32(part).to_bytes(4, "little") + len(part).to_bytes(4, "little"))
    return bytes(out)
eng = Engine(0, 14, True)
uniform, real = orc.synth_fasta(SEED, 0, 3_000_000, 4).tobytes(), orc.synth_realistic(SEED, 1, 2_000_000).tobytes()
lowent = (b">x\n" + b"ACGT" * 20 + b"\n") * 20000
d = tempfile.mkdtemp()
for name, raw, kw in (("l1", uniform, dict(level=1)), ("l6", uniform, dict(level=6)), ("l9", real, dict(level=9)),
                      ("fixed", uniform[:400_000], dict(level=6, strategy=zlib.Z_FIXED)), ("stored", uniform[:700_000], dict(level=0)),
                      ("huff", uniform[:500_000], dict(level=6, strategy=zlib.Z_HUFFMAN_ONLY)), ("rle", real[:500_000], dict(level=6, strategy=zlib.Z_RLE)),
                      ("tiny", uniform[:3000], dict(level=6, block=1)), ("full", uniform, dict(level=6, block=65536)), ("lowent", lowent, dict(level=9))):
    p = os.path.join(d, name + ".fa.gz")
    data = bgzf(raw, **kw)
    open(p, "wb").write(data)
    try:
        g = eng.sketch_files([p], 19, 21)[0]
        print(name, "ok", np.array_equal(g, eng.sketch_buffer(np.frombuffer(raw, np.uint8), 19, 21)))
    except EngineError as e:
        print(name, "REFUSED", str(e)[-80:])
        # find the block: one-block files
        off = 0; bi = 0; pos = 0
        block = kw.get("block", 65280)
        while off < len(data):
            size = int.from_bytes(data[off + 16:off + 18], "little") + 1
            one = data[off:off + size] + bgzf(b"")
            q = os.path.join(d, "one.fa.gz")
            open(q, "wb").write(b"" + one)
            part = raw[pos:pos + block]
            try:
                eng.sketch_files([q], 19, 21)
            except EngineError:
                print("   block", bi, "text bytes", len(part), "compressed", size, "first bytes", part[:40])
                np.save(f"gpurun_out/refused_{name}_{bi}.npy", np.frombuffer(part, np.uint8))
                break
            off += size; bi += 1; pos += block
