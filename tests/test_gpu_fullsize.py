"""GPU parity at BASELINE.json's full sizes (configs 2-5, per-GPU shares) and of the benchmarked call itself.

The oracle costs ~0.8 s per (50 Mbp, k), so full-size runs are pinned three ways: (i) the oracle on a
few sampled (genome, k) rows, (ii) the batched call against single-genome calls, (iii) size-independent
properties (whole == max of parts, running max == flat union, symmetry, determinism, HLL vs the exact
counter).  Genomes are generated on the device (byte-identical to oracle.synth_fasta, which
test_gpu_parity pins) and copied back only for the rows the oracle checks.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0xD4ADD
HERE = os.path.dirname(os.path.abspath(__file__))


def _device_genomes(torch, eng, specs):
    """specs: [(genome_index, nbases, nrec)] -> (list of uint8 device tensors, list of byte sizes)"""
    from dandd_amd.engine import synth_size
    bufs, sizes = [], []
    for gi, nb, nrec in specs:
        n = synth_size(nb, nrec)
        t = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
        eng.synth_fasta_device(SEED, gi, nb, nrec, t.data_ptr())
        bufs.append(t)
        sizes.append(n)
    eng.synchronize()
    return bufs, sizes


def _record_starts(torch, buf, size, step=1 << 28):
    """Offsets of the '>' bytes (searched piecewise: one boolean mask of 3 GB is too much for nonzero)."""
    found = []
    for a in range(0, size, step):
        b = min(size, a + step)
        found.append(torch.nonzero(buf[a:b] == ord(">")).flatten().cpu().numpy() + a)
    return np.concatenate(found)


def _host(t, n):
    return t[:n].cpu().numpy()


def test_cfg2_benchmarked_call(engine_factory, torch_cuda, orc):
    """bench.py's step: 10 x 50 Mbp, k 4..40, log2m 14 in ONE dd_sketch_device call.  Rows of three
    sampled genomes == the single-genome call on the same bytes, and == the oracle for two ks each."""
    torch = torch_cuda
    eng = engine_factory(14, True)
    ng, nb, kmin, kmax = 10, 50_000_000, 4, 40
    K = kmax - kmin + 1
    bufs, sizes = _device_genomes(torch, eng, [(g, nb, 5) for g in range(ng)])
    regs = torch.empty((ng, K, eng.m), dtype=torch.uint8, device="cuda")
    eng.sketch_device([b.data_ptr() for b in bufs], sizes, kmin, kmax, regs.data_ptr())
    eng.synchronize()
    got = regs.cpu().numpy()
    # a second call (job tables reused) is byte-identical
    eng.sketch_device([b.data_ptr() for b in bufs], sizes, kmin, kmax, regs.data_ptr())
    eng.synchronize()
    assert np.array_equal(regs.cpu().numpy(), got)
    for g, ks in ((0, (9, 24)), (4, (16, 33)), (9, (10, 40))):
        fa = _host(bufs[g], sizes[g])
        assert np.array_equal(eng.sketch_buffer(fa, kmin, kmax), got[g]), f"genome {g}: batched != single call"
        for k in ks:
            assert np.array_equal(got[g, k - kmin], orc.sketch(fa, k, 14, True)), f"genome {g}, k={k} differs from the oracle"
    # the step's other half: root union and cardinalities
    root = torch.empty((K, eng.m), dtype=torch.uint8, device="cuda")
    eng.union_device([regs[g].data_ptr() for g in range(ng)], K * eng.m, root.data_ptr())
    eng.synchronize()
    assert np.array_equal(root.cpu().numpy(), got.max(axis=0))
    card = eng.card_batch_device(regs.data_ptr(), ng * K).reshape(ng, K)
    assert card[4, 16 - kmin] == orc.card(got[4, 16 - kmin], 14)


@pytest.mark.parametrize("p", [18, 20])
def test_cfg2_genome_registers_in_hbm(engine_factory, torch_cuda, orc, p):
    """DandD's default register count (-r 20, /root/reference/lib/dandd_cmd.py:187) and log2m 18 on the
    50 Mbp genome: bit-exact vs the oracle for one k of the bitmap class, the 64-bit class and the 96-bit
    class, at the size where bucket overflow, contention and stale bounds actually occur."""
    torch = torch_cuda
    eng = engine_factory(p, True)
    bufs, sizes = _device_genomes(torch, eng, [(0, 50_000_000, 5), (1, 3_000_000, 2)])
    kmin, kmax = 9, 40
    K = kmax - kmin + 1
    regs = torch.empty((2, K, eng.m), dtype=torch.uint8, device="cuda")
    eng.sketch_device([b.data_ptr() for b in bufs], sizes, kmin, kmax, regs.data_ptr())
    eng.synchronize()
    fa = _host(bufs[0], sizes[0])
    for k in (9, 21, 40):
        got = regs[0, k - kmin].cpu().numpy()
        want = orc.sketch(fa, k, p, True)
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, f"log2m {p}, k={k}: {bad.size} registers differ, first idx {bad[0]}: got {got[bad[0]]} want {want[bad[0]]}"
    small = _host(bufs[1], sizes[1])
    assert np.array_equal(regs[1, 17 - kmin].cpu().numpy(), orc.sketch(small, 17, p, True))
    # determinism of the whole slab
    again = torch.empty_like(regs)
    eng.sketch_device([b.data_ptr() for b in bufs], sizes, kmin, kmax, again.data_ptr())
    eng.synchronize()
    assert torch.equal(regs, again)


def test_cfg3_all_pairs_kij(engine_factory, torch_cuda, orc):
    """BASELINE cfg 3: 64 x 5 Mbp, the reference's default k range 2..32 (/root/reference/lib/
    dandd_cmd.py:149-150), all-pairs union cardinalities in one dd_pairwise_device launch."""
    torch = torch_cuda
    eng = engine_factory(14, True)
    n, nb, kmin, kmax = 64, 5_000_000, 2, 32
    K = kmax - kmin + 1
    bufs, sizes = _device_genomes(torch, eng, [(g, nb, 5) for g in range(n)])
    regs = torch.empty((n, K, eng.m), dtype=torch.uint8, device="cuda")
    eng.sketch_device([b.data_ptr() for b in bufs], sizes, kmin, kmax, regs.data_ptr())
    eng.synchronize()
    pair = eng.pairwise_device(regs.data_ptr(), n, K)
    leaf_card = eng.card_batch_device(regs.data_ptr(), n * K).reshape(n, K)
    assert np.array_equal(pair[np.arange(n), np.arange(n)], leaf_card)          # diagonal == own cardinality
    assert np.array_equal(pair, pair.transpose(1, 0, 2))                        # symmetric
    rng = np.random.default_rng(7)
    cache = {}

    def oracle_row(g, k):
        if (g, k) not in cache:
            cache[(g, k)] = orc.sketch(_host(bufs[g], sizes[g]), k, 14, True)
        return cache[(g, k)]

    for _ in range(8):
        i, j = sorted(rng.choice(n, size=2, replace=False).tolist())
        k = int(rng.integers(kmin, kmax + 1))
        a, b = oracle_row(i, k), oracle_row(j, k)
        assert np.array_equal(regs[i, k - kmin].cpu().numpy(), a)
        assert pair[i, j, k - kmin] == orc.card(np.maximum(a, b), 14), (i, j, k)
    # KIJ from the matrix is what the formula gives on oracle numbers for one pair
    ks = np.arange(kmin, kmax + 1)
    i, j = 3, 41
    da, db, dab = (leaf_card[i] / ks).max(), (leaf_card[j] / ks).max(), (pair[i, j] / ks).max()
    kbest = int(ks[(pair[i, j] / ks).argmax()])
    want_ab = orc.card(np.maximum(oracle_row(i, kbest), oracle_row(j, kbest)), 14) / kbest
    assert dab == want_ab
    assert np.isfinite((da + db - dab) / dab)


def test_cfg3_cli_kij_jaccard_rows(torch_cuda, orc, tmp_path):
    """`dandd kij --jaccard` on 8 of the cfg 3 genomes as files: the GPU backend writes the same rows as
    the oracle backend behind the same host layer (/root/reference/lib/huffman_dandd.py:666-695,772-815)."""
    import hostcheck
    from dandd_amd.host import cli, deltatree
    data = tmp_path / "data"
    data.mkdir()
    for g in range(8):
        (data / f"b{g}.fasta").write_bytes(orc.synth_fasta(SEED, g, 5_000_000, 5).tobytes())

    def walk(name, factory):
        deltatree.set_backend_factory(factory)
        out = str(tmp_path / name)
        cli.main(["tree", "-d", str(data), "-o", out, "-s", "c3", "-k", "12", "-r", "14"])
        cli.main(["kij", "-d", os.path.join(out, "c3_8_dashing_dtree.pickle"), "-o", out, "--jaccard", "--mink", "11", "--maxk", "14"])
        return (hostcheck.read_rows(os.path.join(out, "c3_8_dashing.kij.csv")),
                hostcheck.read_rows(os.path.join(out, "c3_8_dashing.j.csv")))

    try:
        gpu = walk("gpu", None)
        cpu = walk("cpu", lambda r, c: hostcheck.OracleBackend(r, c))
    finally:
        deltatree.set_backend_factory(None)
    assert len(gpu[0]) == 28 and len(gpu[1]) == 28 * 4
    assert not hostcheck.compare({"kij": gpu[0], "j": gpu[1]}, {"kij": cpu[0], "j": cpu[1]})


def test_cfg4_share_progressive(engine_factory, torch_cuda, orc):
    """BASELINE cfg 4, one GPU's share: 8 x 250 Mbp (chromosome scale), the committed 10 orderings.
    dd_progressive_device == running byte-max over each ordering (the flat prefix unions of
    /root/reference/lib/huffman_dandd.py:644-663), leaf rows == oracle for two (genome, k)."""
    torch = torch_cuda
    eng = engine_factory(14, True)
    with open(os.path.join(HERE, "golden", "cfg4_orderings.json")) as f:
        ords = json.load(f)["orderings"]
    n, nb, kmin, kmax = 8, 250_000_000, 2, 32
    K = kmax - kmin + 1
    bufs, sizes = _device_genomes(torch, eng, [(100 + g, nb, 1) for g in range(n)])
    regs = torch.empty((n, K, eng.m), dtype=torch.uint8, device="cuda")
    eng.sketch_device([b.data_ptr() for b in bufs], sizes, kmin, kmax, regs.data_ptr())
    eng.synchronize()
    for g, k in ((2, 20), (7, 31)):
        assert np.array_equal(regs[g, k - kmin].cpu().numpy(), orc.sketch(_host(bufs[g], sizes[g]), k, 14, True)), (g, k)
    prog = eng.progressive_device(regs.data_ptr(), n, K, ords)
    run = torch.empty((len(ords), n, K, eng.m), dtype=torch.uint8, device="cuda")
    for o, order in enumerate(ords):
        acc = torch.zeros((K, eng.m), dtype=torch.uint8, device="cuda")
        for j, g in enumerate(order):
            acc = torch.maximum(acc, regs[g])
            run[o, j] = acc
    torch.cuda.synchronize()
    want = eng.card_batch_device(run.data_ptr(), len(ords) * n * K).reshape(len(ords), n, K)
    assert np.array_equal(prog, want)
    # the library's own N-way union agrees with the running max for one prefix
    out = torch.empty((K, eng.m), dtype=torch.uint8, device="cuda")
    eng.union_device([regs[g].data_ptr() for g in ords[3][:5]], K * eng.m, out.data_ptr())
    eng.synchronize()
    assert torch.equal(out, run[3, 4])
    # every ordering ends in the same root, and delta grows along an ordering (helpers/clean_abba.py:41)
    assert all(np.array_equal(prog[o, n - 1], prog[0, n - 1]) for o in range(len(ords)))
    ks = np.arange(kmin, kmax + 1)
    delta = (prog / ks).max(axis=2)
    assert np.all(np.diff(delta, axis=1) > 0)


def test_cfg5_share_whole_genome(engine_factory, torch_cuda, orc):
    """BASELINE cfg 5, one genome of one GPU's share: 3 Gbp, 24 records, k 4..64 (K = 61).  whole ==
    max(two halves split at a record boundary), deterministic, HLL within 4 sigma of the GPU exact counter
    at two ks; the oracle itself on a 1 Gbp genome for one k of the 128-bit class."""
    torch = torch_cuda
    eng = engine_factory(14, True)
    kmin, kmax = 4, 64
    K = kmax - kmin + 1
    (buf,), (size,) = _device_genomes(torch, eng, [(200, 3_000_000_000, 24)])
    # record starts: '>' bytes
    starts = _record_starts(torch, buf, size)
    assert starts.size == 24
    cut = int(starts[12])  # split at a record boundary; the second half gets its own (aligned) buffer
    second = buf[cut:size].clone()
    regs = torch.empty((3, K, eng.m), dtype=torch.uint8, device="cuda")
    eng.sketch_device([buf.data_ptr()], [size], kmin, kmax, regs[0].data_ptr())
    eng.sketch_device([buf.data_ptr(), second.data_ptr()], [cut, size - cut], kmin, kmax, regs[1].data_ptr())
    eng.synchronize()
    assert torch.equal(regs[0], torch.maximum(regs[1], regs[2]))
    again = torch.empty((K, eng.m), dtype=torch.uint8, device="cuda")
    eng.sketch_device([buf.data_ptr()], [size], kmin, kmax, again.data_ptr())
    eng.synchronize()
    assert torch.equal(again, regs[0])
    card = eng.card_batch_device(regs[0].data_ptr(), K)
    del second
    torch.cuda.empty_cache()
    for k in (21, 55):
        exact = eng.exact_count_device([buf.data_ptr()], [size], k)
        assert abs(card[k - kmin] - exact) / exact < 4 * 1.04 / np.sqrt(eng.m), (k, card[k - kmin], exact)
    del buf
    torch.cuda.empty_cache()
    (g1,), (s1,) = _device_genomes(torch, eng, [(201, 1_000_000_000, 24)])
    one = torch.empty((1, eng.m), dtype=torch.uint8, device="cuda")
    eng.sketch_device([g1.data_ptr()], [s1], 61, 61, one.data_ptr())
    eng.synchronize()
    assert np.array_equal(one[0].cpu().numpy(), orc.sketch(_host(g1, s1), 61, 14, True))


def test_exact_count_of_a_union_larger_than_the_key_budget(engine_factory, torch_cuda):
    """The `--exact` yardstick at scale (/root/reference/lib/sketch_classes.py:453-465 unions any number of KMC
    databases): 4 x 3 Gbp = 12 GB of FASTA, 12.2 G k-mers = 195 GB of keys at once -- counted in passes under the
    default 24 GiB budget.  The same number comes out under a different budget (different partition), the union
    with a repeated input does not change it, and the HLL estimate of the union agrees within 4 sigma."""
    torch = torch_cuda
    eng = engine_factory(14, True)
    bufs, sizes = _device_genomes(torch, eng, [(300 + g, 3_000_000_000, 24) for g in range(4)])
    ptrs = [b.data_ptr() for b in bufs]
    k = 31
    total = eng.exact_count_device(ptrs, sizes, k)
    passes = eng.last_sketch_stats()[2]
    assert passes >= 8
    os.environ["DD_EXACT_MB"] = str(40 * 1024)
    try:
        assert eng.exact_count_device(ptrs, sizes, k) == total
        assert eng.last_sketch_stats()[2] < passes
        one = eng.exact_count_device(ptrs[:1], sizes[:1], k)
        assert eng.exact_count_device([ptrs[0], ptrs[0]], [sizes[0], sizes[0]], k) == one   # a repeated input adds nothing
    finally:
        del os.environ["DD_EXACT_MB"]
    assert one < total < 4 * one + 1
    regs = torch.empty((5, 1, eng.m), dtype=torch.uint8, device="cuda")
    eng.sketch_device(ptrs, sizes, k, k, regs.data_ptr())
    eng.union_device([regs[g].data_ptr() for g in range(4)], eng.m, regs[4].data_ptr())
    est = eng.card_batch_device(regs[4].data_ptr(), 1)[0]
    assert abs(est - total) / total < 4 * 1.04 / np.sqrt(eng.m), (est, total)


# ------------------------------------------------------------------------------------------------------------
# DandD's DEFAULT register count (-r 20, /root/reference/lib/dandd_cmd.py:187) at the sizes of configs 4 and 5:
# the scatter + sort + replay path with u32 record cursors, capacity clamps and the compare-and-swap overflow
# (dd_sweep.hip) at the sizes where those can actually break.
# ------------------------------------------------------------------------------------------------------------
def _halves(torch, buf, size, nrec):
    starts = _record_starts(torch, buf, size)
    assert starts.size == nrec
    cut = int(starts[nrec // 2])
    return cut, buf[cut:size].clone()      # the second half gets its own (aligned) buffer


def test_default_registers_whole_genome_3gbp(engine_factory, torch_cuda, orc):
    """3 Gbp x log2m 20 x k 21..64 (64-, 96- and 128-bit window classes in one call): whole == max(two halves split at
    a record boundary), deterministic, HLL within 4 sigma of the GPU exact counter at k 21 and 55; and the oracle
    itself on a 1 Gbp genome at log2m 20 for k 31 (64-bit class) and k 61 (128-bit class)."""
    torch = torch_cuda
    eng = engine_factory(20, True)
    kmin, kmax = 21, 64
    K = kmax - kmin + 1
    (buf,), (size,) = _device_genomes(torch, eng, [(200, 3_000_000_000, 24)])
    cut, second = _halves(torch, buf, size, 24)
    regs = torch.empty((3, K, eng.m), dtype=torch.uint8, device="cuda")
    eng.sketch_device([buf.data_ptr()], [size], kmin, kmax, regs[0].data_ptr())
    eng.sketch_device([buf.data_ptr(), second.data_ptr()], [cut, size - cut], kmin, kmax, regs[1].data_ptr())
    eng.synchronize()
    whole, parts = regs[0], torch.maximum(regs[1], regs[2])
    for k in (21, 40, 55, 64):     # named so that a failure says which class broke
        assert torch.equal(whole[k - kmin], parts[k - kmin]), f"k={k}: whole != max(halves)"
    assert torch.equal(whole, parts)
    again = torch.empty((K, eng.m), dtype=torch.uint8, device="cuda")
    eng.sketch_device([buf.data_ptr()], [size], kmin, kmax, again.data_ptr())
    eng.synchronize()
    assert torch.equal(again, whole)
    card = eng.card_batch_device(whole.data_ptr(), K)
    del second, again
    torch.cuda.empty_cache()
    for k in (21, 55):
        exact = eng.exact_count_device([buf.data_ptr()], [size], k)
        assert abs(card[k - kmin] - exact) / exact < 4 * 1.04 / np.sqrt(eng.m), (k, card[k - kmin], exact)
    del buf, regs
    torch.cuda.empty_cache()
    (g1,), (s1,) = _device_genomes(torch, eng, [(201, 1_000_000_000, 24)])
    fa = _host(g1, s1)
    for k in (31, 61):
        one = torch.empty((1, eng.m), dtype=torch.uint8, device="cuda")
        eng.sketch_device([g1.data_ptr()], [s1], k, k, one.data_ptr())
        eng.synchronize()
        got, want = one[0].cpu().numpy(), orc.sketch(fa, k, 20, True)
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, f"1 Gbp, log2m 20, k={k}: {bad.size} registers differ, first idx {bad[0]}: got {got[bad[0]]} want {want[bad[0]]}"


@pytest.mark.parametrize("p", [18, 20])
def test_many_small_genomes_at_default_registers(engine_factory, torch_cuda, orc, p):
    """A bacterial collection at DandD's defaults (cfg 3's genomes, -r 20): 64 x 5 Mbp in ONE call.  At log2m 20 the only
    epoch is the unfiltered first one (4.8 tokens per register): every update is a record, binned straight into its index
    tile's region of the row's stream (16 bins of 4480 per tile of tokens), the classes run back to back.  Rows of five
    genomes against the oracle for one k of the 32-, 64- and 96-bit classes; the whole slab twice."""
    torch = torch_cuda
    eng = engine_factory(p, True)
    ng, nb, kmin, kmax = 64, 5_000_000, 10, 40
    K = kmax - kmin + 1
    bufs, sizes = _device_genomes(torch, eng, [(g, nb, 5) for g in range(ng)])
    regs = torch.empty((ng, K, eng.m), dtype=torch.uint8, device="cuda")
    eng.sketch_device([b.data_ptr() for b in bufs], sizes, kmin, kmax, regs.data_ptr())
    eng.synchronize()
    for g, k in ((0, 12), (7, 16), (21, 17), (40, 31), (63, 40)):
        got = regs[g, k - kmin].cpu().numpy()
        want = orc.sketch(_host(bufs[g], sizes[g]), k, p, True)
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, f"log2m {p}, genome {g}, k={k}: {bad.size} registers differ, first idx {bad[0]}: got {got[bad[0]]} want {want[bad[0]]}"
    again = torch.empty_like(regs)
    eng.sketch_device([b.data_ptr() for b in bufs], sizes, kmin, kmax, again.data_ptr())
    eng.synchronize()
    assert torch.equal(regs, again)


@pytest.mark.parametrize("p", [15, 16, 17])
def test_cfg2_genome_between_the_register_regimes(engine_factory, torch_cuda, orc, p):
    """The register counts between the two regimes that had full-size parity (14: several ks per LDS group; 18, 20:
    scatter + replay), on the 50 Mbp BASELINE genome: log2m 15 (two ks per 80 KiB group), 16 -- where the metric's
    accuracy clause is met: ONE k per group, the window push unshared -- and 17, the first size whose registers live in
    HBM (two 64 KiB index tiles per row, first epoch binned).  One k of every kernel class (small-k sets, 32-, 64-,
    96- and 128-bit windows) against the oracle, bit for bit, plus determinism of the whole slab."""
    torch = torch_cuda
    eng = engine_factory(p, True)
    bufs, sizes = _device_genomes(torch, eng, [(0, 50_000_000, 5)])
    kmin, kmax = 8, 52
    K = kmax - kmin + 1
    regs = torch.empty((1, K, eng.m), dtype=torch.uint8, device="cuda")
    eng.sketch_device([bufs[0].data_ptr()], sizes, kmin, kmax, regs.data_ptr())
    eng.synchronize()
    fa = _host(bufs[0], sizes[0])
    for k in (8, 13, 27, 41, 52):
        got = regs[0, k - kmin].cpu().numpy()
        want = orc.sketch(fa, k, p, True)
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, f"log2m {p}, k={k}: {bad.size} registers differ, first idx {bad[0]}: got {got[bad[0]]} want {want[bad[0]]}"
    again = torch.empty_like(regs)
    eng.sketch_device([bufs[0].data_ptr()], sizes, kmin, kmax, again.data_ptr())
    eng.synchronize()
    assert torch.equal(regs, again)


@pytest.mark.parametrize("p", [14, 16, 20])
def test_cfg5_share_as_benchmarked(engine_factory, torch_cuda, p):
    """BASELINE cfg 5, one GPU's share exactly as `bench.py --config cfg5share` runs it: 13 x 3 Gbp (39.5 GB of
    FASTA) in ONE dd_sketch_device call, k 4..64 (K = 61).  Rows of the first and the last genome == their own
    single-genome calls (which test_default_registers_whole_genome_3gbp / test_cfg5_share_whole_genome pin through
    halves, the exact counter and the oracle); the call is deterministic; no row is empty."""
    torch = torch_cuda
    eng = engine_factory(p, True)
    ng, nb, kmin, kmax = 13, 3_000_000_000, 4, 64
    K = kmax - kmin + 1
    bufs, sizes = _device_genomes(torch, eng, [(g, nb, 24) for g in range(ng)])
    regs = torch.empty((ng, K, eng.m), dtype=torch.uint8, device="cuda")
    eng.sketch_device([b.data_ptr() for b in bufs], sizes, kmin, kmax, regs.data_ptr())
    eng.synchronize()
    one = torch.empty((K, eng.m), dtype=torch.uint8, device="cuda")
    for g in (0, 12):
        eng.sketch_device([bufs[g].data_ptr()], [sizes[g]], kmin, kmax, one.data_ptr())
        eng.synchronize()
        for k in range(kmin, kmax + 1):
            assert torch.equal(one[k - kmin], regs[g, k - kmin]), f"log2m {p}: genome {g}, k={k}: batched row != single-genome call"
    assert bool((regs.view(ng * K, eng.m).max(dim=1).values > 0).all())
    again = torch.empty_like(regs)
    eng.sketch_device([b.data_ptr() for b in bufs], sizes, kmin, kmax, again.data_ptr())
    eng.synchronize()
    assert torch.equal(again, regs)
    # genomes differ (1 % substitutions): a row copied from a neighbour would pass the checks above for g = 0, 12 only
    card = eng.card_batch_device(regs.data_ptr(), ng * K).reshape(ng, K)
    assert len({float(c) for c in card[:, 31 - kmin]}) == ng


@pytest.mark.parametrize("p", [14, 20])
def test_cfg4_progressive_at_its_true_size(engine_factory, torch_cuda, orc, p):
    """BASELINE cfg 4 whole: n = 30 genomes of 250 Mbp (7.6 GB: one MI355X holds the 4-GPU job), 10 committed
    orderings, k 2..32.  dd_progressive_device == running torch.maximum along every ordering + dd_card_batch_device
    (the flat prefix unions of /root/reference/lib/huffman_dandd.py:644-663), at log2m 14 and at DandD's default 20."""
    torch = torch_cuda
    eng = engine_factory(p, True)
    with open(os.path.join(HERE, "golden", "cfg4_orderings_n30.json")) as f:
        ords = json.load(f)["orderings"]
    n, nb, kmin, kmax = 30, 250_000_000, 2, 32
    assert len(ords) == 10 and all(sorted(o) == list(range(n)) for o in ords)
    K = kmax - kmin + 1
    bufs, sizes = _device_genomes(torch, eng, [(100 + g, nb, 1) for g in range(n)])
    regs = torch.empty((n, K, eng.m), dtype=torch.uint8, device="cuda")
    eng.sketch_device([b.data_ptr() for b in bufs], sizes, kmin, kmax, regs.data_ptr())
    eng.synchronize()
    g, k = 17, 25
    assert np.array_equal(regs[g, k - kmin].cpu().numpy(), orc.sketch(_host(bufs[g], sizes[g]), k, p, True)), (g, k)
    del bufs
    torch.cuda.empty_cache()
    prog = eng.progressive_device(regs.data_ptr(), n, K, ords)
    assert prog.shape == (10, n, K)
    for o, order in enumerate(ords):       # one ordering at a time: 30 x 31 MiB of running unions at log2m 20
        run = torch.empty((n, K, eng.m), dtype=torch.uint8, device="cuda")
        acc = torch.zeros((K, eng.m), dtype=torch.uint8, device="cuda")
        for j, gi in enumerate(order):
            acc = torch.maximum(acc, regs[gi])
            run[j] = acc
        torch.cuda.synchronize()
        want = eng.card_batch_device(run.data_ptr(), n * K).reshape(n, K)
        assert np.array_equal(prog[o], want), f"log2m {p}: ordering {o} differs from the running maximum"
        del run
    assert all(np.array_equal(prog[o, n - 1], prog[0, n - 1]) for o in range(10))


@pytest.mark.parametrize("p", [18, 20])
def test_record_stream_overflow_at_full_size(engine_factory, torch_cuda, orc, monkeypatch, p):
    """DD_BUCKET_CAP=1: one 1024-record chunk per row, so nearly every record of the 50 Mbp genome takes the
    compare-and-swap fallback of the scatter kernels (dd_sweep.hip) -- thousands of waves contending for the same
    registers, which the 330 kbp knob test cannot produce.  Registers must still be the oracle's."""
    torch = torch_cuda
    monkeypatch.setenv("DD_BUCKET_CAP", "1")
    eng = engine_factory(p, True)
    bufs, sizes = _device_genomes(torch, eng, [(0, 50_000_000, 5)])
    kmin, kmax = 12, 34
    K = kmax - kmin + 1
    regs = torch.empty((1, K, eng.m), dtype=torch.uint8, device="cuda")
    eng.sketch_device([bufs[0].data_ptr()], sizes, kmin, kmax, regs.data_ptr())
    eng.synchronize()
    monkeypatch.delenv("DD_BUCKET_CAP")
    ref = torch.empty_like(regs)
    eng.sketch_device([bufs[0].data_ptr()], sizes, kmin, kmax, ref.data_ptr())     # the roomy record streams
    eng.synchronize()
    assert torch.equal(regs, ref)
    fa = _host(bufs[0], sizes[0])
    for k in (13, 33):
        assert np.array_equal(regs[0, k - kmin].cpu().numpy(), orc.sketch(fa, k, p, True)), (p, k)
