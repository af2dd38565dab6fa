"""Generates tests/golden/kat_arith.json from tests/pyref.py ONLY (pure Python, arbitrary-precision
ints; nothing from oracle/ or dandd_amd/ is imported), so the committed vectors pin the C oracle and,
through it, the HIP kernels.  Run from the repo root:  python tests/golden/make_kat.py"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import pyref  # noqa: E402

rnd = random.Random(20261002)
kat = {"wang64": [], "idx_rho": [], "fold128": [], "sketch": [], "mle": [], "records": []}
# record rules (kseq's, as recalled: POLICIES.md P10): input -> the sequence string of every record
for text in ["", ">h\n", ">h", "ACGT", "ACGTNNACGT\n>late\nGGCC\n", "junk junk >h1 comment\nACGT\nTTGA\n\n>h2\nCC\n",
             ">a\r\nACGTAC\r\nGTACGG\r\n", ">a\nAC\rGT\nAA\r\r\nC\n", ">a\nACGT>ACGT\n@b also a header\nTTTT\n",
             "@r1\nACGTACGT\n+\nIIIIIIII\n@r2\nGGCCA\n+r2\nII@>I\n", "@r1\nACGT\nACGT\n+\nIIII\nII@I\n@r2\nTT\n+\n>I\n",
             "@r1\nACGT\n+\nII\n", ">x\nACGT\n+\nACGT\nGGGG>y\nCC\n", "@q\nAC\n+\n@@\n@q2\nGG\n+\n>>\n>f\nTTA\n",
             "\n\n>x\n\nACG\n\nT\n", ">only\n\n\n", "@", "+\nACGT\n>z\nAC\n",
             # kseq_read's -2 (round 5): a FASTQ record cut off inside its '+' line, or whose quality text is not exactly as long
             # as its sequence, ends the reading -- that record and everything behind it are dropped; one quality line is read
             # even for an empty sequence
             "@r1\nACGT\n+", "@r1\nACGT\n+\n", "@r1\nACGT\n+\nIIIII\n@r2\nAA\n+\nII\n",
             "@ok\nAC\n+\nII\n@bad\nACGT\n+\nII\n@after\nGG\n+\nII\n", "@e\n+\n\n@r2\nAA\n+\nII\n", "@e\n+\n@r2\nAA\n+\nII\n",
             "@e\n+", "@e\n+\n", ">fa\nACGT\n@fq\nGG\n+\nI\n>fa2\nTT\n", "@r\nAC\nGT\n+\nII\nI\n@r2\nCC\n+\nII\n"]:
    kat["records"].append([text, [r.decode("latin-1") for r in pyref.records(text.encode("latin-1"))]])
for x in [0, 1, 2, 0xDEADBEEF, 2**32, 2**63, 2**64 - 1] + [rnd.getrandbits(64) for _ in range(40)]:
    kat["wang64"].append([hex(x), hex(pyref.wang64(x))])
for _ in range(40):
    h = rnd.getrandbits(64) >> rnd.choice([0, 0, 8, 20, 40, 50])
    for p in (10, 14, 20):
        i, r = pyref.idx_rho(h, p)
        kat["idx_rho"].append([hex(h), p, i, r])
for _ in range(10):
    hi, lo = rnd.getrandbits(rnd.choice([2, 16, 64])), rnd.getrandbits(64)
    kat["fold128"].append([hex(hi), hex(lo), hex(pyref.fold128(hi, lo))])


def rand_fasta(n, nrec):
    out = []
    for r in range(nrec):
        out.append(f">rec{r} some description")
        s = "".join(rnd.choice("ACGTACGTACGTacgtN") for _ in range(n // nrec))
        out += [s[i:i + 60] for i in range(0, len(s), 60)]
    return "\n".join(out) + "\n"


for k, p, canon in [(4, 8, True), (11, 10, True), (16, 10, False), (21, 12, True), (32, 10, True), (33, 10, True), (47, 8, False), (64, 8, True)]:
    fa = rand_fasta(1500, 3)
    regs = pyref.sketch(fa.encode(), k, p, canon)
    kat["sketch"].append({"fasta": fa, "k": k, "p": p, "canonical": canon,
                          "nonzero": {str(i): v for i, v in enumerate(regs) if v}})
for p in (8, 14):
    m = 1 << p
    for n in (3, m // 3, 4 * m, 300 * m):
        regs = [0] * m
        for _ in range(min(n, 30000)):
            i, r = pyref.idx_rho(rnd.getrandbits(64), p)
            regs[i] = max(regs[i], r)
        hist = [0] * 64
        for v in regs:
            hist[v] += 1
        kat["mle"].append({"p": p, "hist": hist, "estimate": pyref.ertl_mle(hist, p)})
with open(os.path.join(HERE, "kat_arith.json"), "w") as f:
    json.dump(kat, f, indent=0)
print("wrote kat_arith.json", {k: len(v) for k, v in kat.items()})
