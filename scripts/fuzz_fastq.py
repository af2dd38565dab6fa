"""Randomized check of the device's FASTQ rules (dd_fastq.hip) against the oracle's kseq reading (oracle/dd_oracle.c: orc_records).
Texts that are mostly four-line FASTQ, with the perturbations real files and broken files have -- a record missing a line, a blank
line, a sequence over two lines, a quality text of another length, CRLF, no final newline, '+' lines with and without the name,
'>' headers, '@' / '+' / '>' as first quality character, empty reads, junk in front -- compressed as one gzip member, several
members, or BGZF, go through dd_sketch_files twice: strict (the device must either take the text or refuse it) and, if refused,
with the host behind it.  Whatever path a text takes, the registers must be the oracle's for the plain bytes: a text the device
ACCEPTS although kseq reads it differently shows up as a mismatch.   python scripts/fuzz_fastq.py [N] [SEED]"""
import os, sys, time, tempfile, zlib
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


n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
os.environ["DD_GUNZIP_MIN_KB"] = "1"
d = tempfile.mkdtemp()
eng = Engine(0, 14, True)
BASES, QUAL = np.frombuffer(b"ACGTACGTACGTNacgt", np.uint8), np.frombuffer(b"IIIIFFFF#5:,ACGT@>+", np.uint8)
taken = {"device": 0, "host": 0}
t0 = time.time()
for it in range(n_cfg):
    eol = b"\r\n" if rng.integers(0, 6) == 0 else b"\n"
    nrec = int(rng.choice([1, 3, 50, 400, 3000]))
    p_bad = float(rng.choice([0.0, 0.0, 0.0, 0.002, 0.05]))        # most texts are clean, some have one odd record, some many
    fixed = int(rng.choice([0, 36, 100, 151]))
    recs = []
    if rng.integers(0, 12) == 0:
        recs.append(bytes(rng.choice(BASES, size=int(rng.integers(1, 60)))) + eol)          # junk in front of the first header
    for r in range(nrec):
        L = fixed if fixed and rng.integers(0, 20) else int(rng.integers(0, 300))
        s, q = bytes(rng.choice(BASES, size=L)), bytes(rng.choice(QUAL, size=L))
        head = (b"@" if rng.integers(0, 40) else b">") + b"r%d len=%d" % (r, L)
        plus = b"+" + (b"r%d" % r if rng.integers(0, 3) == 0 else b"")
        lines = [head, s, plus, q]
        if rng.random() < p_bad:
            kind = int(rng.integers(0, 8))
            if kind == 0 and L > 2: lines = [head, s[:L // 2], s[L // 2:], plus, q]                     # sequence over two lines
            elif kind == 1 and L > 2: lines = [head, s, plus, q[:L // 2], q[L // 2:]]                   # quality over two lines
            elif kind == 2: lines = [head, s, plus, q + b"I"]                                             # quality too long
            elif kind == 3 and L: lines = [head, s, plus, q[:-1]]                                         # ... too short
            elif kind == 4: lines = [head, s, plus]                                                       # no quality line
            elif kind == 5: lines = [head, s, b"", plus, q]                                               # a blank line
            elif kind == 6: lines = [head, s]                                                             # a FASTA record in between
            else: lines = [head, s, plus, q, b""]                                                         # a blank line behind the record
        recs.append(eol.join(lines) + eol)
    text = b"".join(recs)
    if rng.integers(0, 5) == 0 and text.endswith(eol):
        text = text[:-len(eol)]                                                                           # no final newline
    want = orc.sketch_sweep(np.frombuffer(text, np.uint8), 19, 21, 14)
    cont = int(rng.integers(0, 3))
    if cont == 0:
        co = zlib.compressobj(int(rng.choice([1, 6, 9])), zlib.DEFLATED, 31)
        data = co.compress(text) + co.flush()
    elif cont == 1:
        data = bgzf(text, int(rng.choice([1, 6])), int(rng.choice([4096, 65280])))
    else:      # two members, cut anywhere (inside a record too)
        cut = int(rng.integers(0, len(text) + 1))
        data = b""
        for part in (text[:cut], text[cut:]):
            co = zlib.compressobj(6, zlib.DEFLATED, 31)
            data += co.compress(part) + co.flush()
    path = os.path.join(d, "reads.fq.gz")
    open(path, "wb").write(data)
    os.environ["DD_INFLATE_STRICT"] = "1"
    try:
        got = eng.sketch_files([path], 19, 21)[0]
        taken["device"] += 1
    except EngineError:
        os.environ.pop("DD_INFLATE_STRICT")
        got = eng.sketch_files([path], 19, 21)[0]
        taken["host"] += 1
    os.environ.pop("DD_INFLATE_STRICT", None)
    if not np.array_equal(got, want):
        print(f"MISMATCH draw {it}: container {cont}, {nrec} records, eol {eol!r}, p_bad {p_bad}: {int((got != want).sum())} registers differ")
        open("gpurun_out/fuzz_fastq_fail.txt", "wb").write(text)
        sys.exit(1)
print(f"{n_cfg} random FASTQ-like texts ({taken['device']} taken by the device's rules, {taken['host']} sent on to the host's kseq state machine): "
      f"registers equal the oracle's reading in {time.time() - t0:.1f} s")
