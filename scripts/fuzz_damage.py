"""Damaged gzip files through dd_sketch_files (device decoders first, host decoder behind them): N files with random bit
flips, overwritten or zeroed stretches, truncations and swapped blocks in BGZF and single-member containers.  The process
must survive (a device memory fault would end it), and every call must raise exactly when zlib's gzread -- the reader of the
reference's stack -- fails on the file, else give the registers of the text gzread hands out.      python scripts/fuzz_damage.py [N] [SEED]"""
import gzip, os, sys, tempfile, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from dandd_amd.engine import Engine, EngineError
from oracle import dd_oracle as orc

def bgzf(raw, level=6, block=65280):
    out = bytearray()
    for a in list(range(0, len(raw), block)) + [len(raw)]:
        part = raw[a:a + block] if a < len(raw) else b""
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = c.compress(part) + c.flush()
        out += (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + (len(body) + 25).to_bytes(2, "little") + body +
                zlib.crc32(part).to_bytes(4, "little") + len(part).to_bytes(4, "little"))
    return bytes(out)

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
os.environ["DD_GUNZIP_MIN_KB"] = "16"
raws = [orc.synth_fasta(0xD4ADD, g, 400_000 + 150_000 * g, 3).tobytes() for g in range(3)] + [orc.synth_realistic(0xD4ADD, 7, 600_000).tobytes()]
goods = []
for raw in raws:
    for level in (1, 6):
        goods.append(("bgzf", bgzf(raw, level)))
        co = zlib.compressobj(level, zlib.DEFLATED, 31)
        goods.append(("member", co.compress(raw) + co.flush()))
# ... and text that inflates 50-1000 x (runs of N, one repeated line): its pieces do not fit their ranges and take the ARENA, which is
# sized by the trailer's ISIZE -- with damage kind 6 (an ISIZE lowered to anywhere in [n / 2, the real length)) the pieces' lengths add
# up to more than the arena holds (advisor, round 4: inflate_kernel<2> wrote past it; the other texts inflate ~4 x and never get there)
lowent = (b">x\n" + b"ACGT" * 20 + b"\n") * 60000 + raws[0][:200_000] + b"N" * 3_000_000 + b"\n" + raws[1][:150_000]
for level in (6, 9):
    co = zlib.compressobj(level, zlib.DEFLATED, 31)
    goods.append(("compressible member", co.compress(lowent) + co.flush()))
d = tempfile.mkdtemp()
eng, refusals = Engine(0, 14, True), 0
counts = {"refused": 0, "read": 0}
t0 = time.time()
for it in range(n_cfg):
    kind, good = goods[int(rng.integers(len(goods)))]
    bad = bytearray(good)
    how = int(rng.integers(6))
    lo = 18 if kind == "bgzf" else 10
    if kind == "compressible member" and rng.integers(3):
        how = 6
        real = int.from_bytes(bad[-4:], "little")
        bad[-4:] = int(rng.integers(len(bad) // 2, real)).to_bytes(4, "little")
    elif how == 0:
        bad[lo + int(rng.integers(len(bad) - lo))] ^= 1 << int(rng.integers(8))
    elif how == 1:
        for _ in range(int(rng.integers(1, 6))):
            bad[lo + int(rng.integers(len(bad) - lo))] = int(rng.integers(256))
    elif how == 2:      # a zeroed stretch
        a = lo + int(rng.integers(len(bad) - lo - 64))
        n0 = int(rng.integers(1, 64))
        bad[a:a + n0] = b"\0" * n0
    elif how == 3:      # truncated (the container's last bytes gone)
        bad = bad[:len(bad) - int(rng.integers(1, 4000))]
    elif how == 4:      # two stretches swapped
        a, b2 = sorted(int(x) for x in rng.integers(lo, len(bad) - 300, size=2))
        if b2 - a > 200:
            bad[a:a + 100], bad[b2:b2 + 100] = bad[b2:b2 + 100], bad[a:a + 100]
    else:               # the trailer's CRC or ISIZE
        bad[len(bad) - 1 - int(rng.integers(8))] ^= 1 << int(rng.integers(8))
    p = os.path.join(d, "bad.fa.gz")
    open(p, "wb").write(bytes(bad))
    # The reference reads through zlib's gzread (klib's kseq, and so Dashing; the host loader too).  gzread is stricter than
    # nothing and laxer than Python's gzip module: a data error or a bad header of a later member fails the read; a stream
    # that ENDS before its end-of-stream marker (a truncated file, or damage that makes a member swallow the rest of the
    # file) hands out what was decoded and says end of file; bytes that are no gzip header behind a complete member are
    # ignored; a file that does not start with the gzip magic is read as it is.
    try:
        text, buf, first = b"", bytes(bad), True
        while buf:
            if buf[:2] != b"\x1f\x8b":
                if first:
                    text = buf
                break
            dobj = zlib.decompressobj(31)
            text += dobj.decompress(buf)
            if not dobj.eof:
                break
            buf, first = dobj.unused_data, False
        want = eng.sketch_buffer(np.frombuffer(text, np.uint8), 19, 21) if len(text) else None
        empty = not len(text)
    except zlib.error:
        want, empty = None, False
    try:
        got = eng.sketch_files([p], 19, 21)[0]
    except EngineError:
        got = None
    if empty:
        continue
    if (got is None) != (want is None) or (want is not None and not np.array_equal(got, want)):
        print(f"DISAGREEMENT at {it}: {kind}, damage {how}: engine {'raised' if got is None else 'read'}, zlib {'raised' if want is None else 'read'}")
        open("gpurun_out/fuzz_damage_fail.gz", "wb").write(bytes(bad))
        sys.exit(1)
    counts["refused" if got is None else "read"] += 1
    if got is None:
        refusals += 1
        if refusals >= 2:       # (three refusals keep a context on the host decoder: a fresh one meets the device decoder again)
            eng.close()
            eng, refusals = Engine(0, 14, True), 0
print(f"{n_cfg} damaged files: {counts['refused']} refused like zlib, {counts['read']} read like zlib, in {time.time() - t0:.1f} s")
