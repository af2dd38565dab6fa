"""CPU tests pinning the oracle: known-answer vectors, an independent pure-Python restatement,
the exact counter, and the algebraic properties the sketch must have.  No GPU."""
import json
import os

import numpy as np
import pytest

import pyref

HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 0xD4ADD


def test_wang_kats(orc):
    # SURVEY.md Appendix A.2 known answers (computed there independently of this repo)
    assert orc.wang64(0) == 0x77CFA1EEF01BCA90
    assert orc.wang64(1) == 0x5BCA7C69B794F8CE
    assert orc.wang64(0xDEADBEEF) == 0x386F2A5F36B257CB
    rng = np.random.default_rng(1)
    for x in [int(v) for v in rng.integers(0, 2**63, size=200)] + [2**64 - 1, 2**63, 12345]:
        assert orc.wang64(x) == pyref.wang64(x)


def test_idx_rho_kats(orc):
    h = orc.wang64(12345)
    assert orc.idx_rho(h, 14) == (6802, 4)
    assert orc.idx_rho(h, 20) == (435332, 1)
    for p in (4, 10, 14, 20):
        assert orc.idx_rho(0, p) == (0, 64 - p + 1)          # all-zero tail saturates at q+1
        assert orc.idx_rho(2**64 - 1, p) == ((1 << p) - 1, 1)
        for x in range(50):
            h = pyref.wang64(x * 7919)
            assert orc.idx_rho(h, p) == pyref.idx_rho(h, p)


def test_golden_kat_file(orc):
    with open(os.path.join(HERE, "golden", "kat_arith.json")) as f:
        kat = json.load(f)
    for x, h in kat["wang64"]:
        assert orc.wang64(int(x, 16)) == int(h, 16)
    for h, p, idx, rho in kat["idx_rho"]:
        assert orc.idx_rho(int(h, 16), p) == (idx, rho)
    for hi, lo, x in kat["fold128"]:
        assert orc.fold128(int(hi, 16), int(lo, 16)) == int(x, 16)
    for case in kat["sketch"]:
        fa = case["fasta"].encode()
        regs = orc.sketch(np.frombuffer(fa, dtype=np.uint8), case["k"], case["p"], case["canonical"])
        nz = {int(i): int(regs[i]) for i in np.nonzero(regs)[0]}
        assert nz == {int(k): v for k, v in case["nonzero"].items()}
    for case in kat["mle"]:
        assert orc.ertl_mle(np.array(case["hist"], dtype=np.uint32), case["p"]) == case["estimate"]
    for text, want in kat["records"]:      # kseq's record rules (POLICIES.md P10): FASTA, FASTQ, junk in front, CRLF
        got = orc.records(np.frombuffer(text.encode("latin-1"), dtype=np.uint8))
        assert [r.decode("latin-1") for r in got] == want, repr(text)


RAGGED = [
    b"", b">h\n", b">h", b"ACGT", b"ACGTNNACGTTTGA\n", b">a\nACGTACGTAC\nGGTTAACC\n>b desc\nTTGACCAGT\n",
    b">a\r\nACGTAC\r\nGTACGG\r\n", b">a\nacgtnACGT>ACGT\n\n\nAC\n", b"\n\n>x\n\nACGTAGCTAGCAT\n", b">a\nA\n>b\nC\n>c\nG\n",
    # kseq's record rules: text in front of the first header, '@' headers, FASTQ (also multi-line, also with '@' / '>' in the
    # quality), a '\r' inside a line, a '+' line in a FASTA file
    b"junk >h1 c\nACGTTGCA\nAC\n", b"ACGTACGTAGCTAGCTAGCATCG\n", b"@r1\nACGTTGCAAC\n+\nIIIIIIIIII\n@r2\nGGCATGCAT\n+\nII@>IIIII\n",
    b"@r1\nACGTTG\nCAACGT\n+\nIIIIII\nI@IIII\n>fa\nTTGACCA\n", b">a\nACGT\rTTGA\nACGTAC\r\r\nGT\n", b">x\nACGTTGCA\n+\nACGTTGCA\nGG>y\nCCATGG\n",
]


@pytest.mark.parametrize("fa", RAGGED)
def test_tokenizer_and_sketch_match_python(orc, fa):
    assert list(orc.tokenize(np.frombuffer(fa, dtype=np.uint8))) == pyref.tokenize(fa)
    for k in (1, 3, 8):
        for canon in (True, False):
            got = orc.sketch(np.frombuffer(fa, dtype=np.uint8), k, 6, canon)
            assert list(got) == pyref.sketch(fa, k, 6, canon)
            assert orc.exact_count([np.frombuffer(fa, dtype=np.uint8)], k, canon) == pyref.exact_count([fa], k, canon)


@pytest.mark.parametrize("k", [1, 2, 15, 16, 17, 31, 32, 33, 48, 63, 64])
def test_sketch_matches_python_across_k(orc, k):
    fa = orc.synth_fasta(SEED, 3, 3000, 2)
    for canon in (True, False):
        assert list(orc.sketch(fa, k, 8, canon)) == pyref.sketch(fa.tobytes(), k, 8, canon)
    assert orc.exact_count([fa], k) == pyref.exact_count([fa.tobytes()], k)
    assert np.array_equal(orc.sketch(fa, k, 8), orc.sketch_generic(fa, k, 8))


def test_mle_matches_python_and_is_accurate(orc):
    rng = np.random.default_rng(2)
    for p in (6, 10, 14):
        m = 1 << p
        for n in (0, 1, 10, m // 2, 5 * m, 200 * m):
            regs = np.zeros(m, dtype=np.uint8)
            if n:
                hs = [pyref.wang64(int(v)) for v in rng.integers(0, 2**62, size=min(n, 20000))]
                for h in hs:
                    i, r = pyref.idx_rho(h, p)
                    regs[i] = max(regs[i], r)
            hist = orc.hist(regs)
            assert int(hist.sum()) == m
            est = orc.ertl_mle(hist, p)
            assert est == pyref.ertl_mle([int(v) for v in hist], p)
            nn = min(n, 20000)
            if nn >= 100:
                assert abs(est - nn) / nn < 5 * 1.04 / np.sqrt(m) + 0.01
    sat = np.zeros(64, dtype=np.uint32)
    sat[64 - 14 + 1] = 1 << 14
    assert np.isinf(orc.ertl_mle(sat, 14))


def test_union_is_sketch_of_concatenation(orc):
    a = orc.synth_fasta(SEED, 0, 40000, 2)
    b = orc.synth_fasta(SEED, 1, 30000, 1)
    both = np.concatenate([a, b])
    for k in (5, 21, 40):
        ra, rb = orc.sketch(a, k, 12), orc.sketch(b, k, 12)
        assert np.array_equal(orc.union(ra, rb), orc.sketch(both, k, 12))
        assert np.array_equal(orc.union(ra, rb), np.maximum(ra, rb))
        assert np.array_equal(orc.union(ra, ra), ra)  # idempotent


def test_estimate_within_hll_error_of_exact(orc):
    fa = orc.synth_fasta(SEED, 0, 400000, 4)
    sigma = 1.04 / np.sqrt(1 << 14)
    regs = orc.sketch_sweep(fa, 8, 36, 14)
    errs = []
    for k in range(8, 37, 4):
        exact = orc.exact_count([fa], k)
        est = orc.card(regs[k - 8])
        errs.append((est - exact) / exact)
    assert max(abs(e) for e in errs) < 4 * sigma, errs


def test_canonical_is_strand_symmetric(orc):
    fa = orc.synth_fasta(SEED, 2, 20000, 1)
    body = b"".join(fa.tobytes().split(b"\n")[1:])
    comp = bytes.maketrans(b"ACGTacgtNn", b"TGCAtgcaNn")
    rc = b">rc\n" + body.translate(comp)[::-1] + b"\n"
    for k in (7, 20, 33):
        assert np.array_equal(orc.sketch(fa, k, 10), orc.sketch(np.frombuffer(rc, dtype=np.uint8), k, 10))


def test_synth_fasta_shape(orc):
    fa = orc.synth_fasta(SEED, 5, 1000000, 4).tobytes()
    lines = fa.split(b"\n")
    assert fa.endswith(b"\n") and fa.count(b">") == 4
    assert all(len(l) <= 80 for l in lines)
    body = b"".join(l for l in lines if not l.startswith(b">"))
    assert len(body) == 1000000
    frac_n = body.upper().count(b"N") / len(body)
    frac_lower = sum(1 for c in body if c >= 97) / len(body)
    assert 0.0 < frac_n < 0.005 and 0.05 < frac_lower < 0.2
    # genome 5 differs from genome 6 by roughly 2 x 1 % substitutions
    other = b"".join(l for l in orc.synth_fasta(SEED, 6, 1000000, 4).tobytes().split(b"\n") if not l.startswith(b">"))
    diff = sum(1 for x, y in zip(body.upper(), other.upper()) if x != y) / len(body)
    assert 0.01 < diff < 0.04
