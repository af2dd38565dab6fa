"""The K1 job tables (dandd_amd/csrc/dd_plan.hip) checked on the CPU through dd_plan_sweep: every
(genome, k, token tile) of a sketch call must be covered by exactly one workgroup job, with the window
class, register mode and LDS footprint the kernels assume.  A hole here would not crash anything -- the
k-mers of the missing tiles would silently never reach the sketch."""
import os

import numpy as np
import pytest

os.environ.setdefault("DANDD_NO_TORCH", "1")
from dandd_amd.engine import plan_sweep  # noqa: E402

TILE = 1024 * 64  # tokens per tile; tokens <= FASTA bytes
LDS_MAX = 160 * 1024
KCLASS_RANGE = {-2: (10, 11), -1: (1, 9), 0: (1, 16), 1: (16, 32), 3: (33, 48), 2: (49, 64)}

SIZES = {
    "cfg2": [50_600_000] * 10,
    "ragged": [5_000_000, 0, 1, 63, 64, 65_536, 65_537, 1_000_000, 12_345_678],
    "single_big": [3_100_000_000],
    "many_small": [4_000_000 + 1000 * i for i in range(64)],
}


def check(log2m, sizes, kmin, kmax):
    jobs = plan_sweep(log2m, sizes, kmin, kmax)
    m = 1 << log2m
    ntiles = [(n + TILE - 1) // TILE for n in sizes]
    cover = [np.zeros((kmax - kmin + 1, nt), dtype=np.int32) for nt in ntiles]
    extra_slices = {}   # big-bitmap class: (genome, k, slice >= 1) -> coverage of the genome's tiles
    for j in jobs:
        lo, hi = KCLASS_RANGE[int(j["kclass"])]
        if j["tile_end"] <= j["tile_begin"]:
            continue  # idle filler of the XCD-affine order
        k0, nk, g = int(j["kfirst"]), int(j["nk"]), int(j["genome"])
        assert nk >= 1 and lo <= k0 and k0 + nk - 1 <= hi, j
        assert kmin <= k0 and k0 + nk - 1 <= kmax
        assert 0 <= g < len(sizes) and j["tile_end"] <= ntiles[g]
        assert 0 <= j["lds_bytes"] <= LDS_MAX
        mode = int(j["mode"])
        if j["kclass"] == -1:
            assert mode == 0
        elif j["kclass"] == -2:
            # log2m >= 19, bucket mode: k = 10 (and 11 at log2m 20) recorded exactly, one 2^20-bit slice of the
            # index space per job; every slice of a k must see every tile once
            assert log2m >= 19 and k0 <= (11 if log2m >= 20 else 10) and nk == 1 and mode == 0
            assert sum(sizes) // len(sizes) >= (1 << max(0, log2m - 17)) * 4 ** k0 or os.environ.get("DD_BIGMAP_ANY_SIZE")
            assert j["lds_bytes"] == 128 * 1024 and 0 <= j["slice"] < (2 if k0 == 11 else 1)
            if j["slice"] > 0:
                extra_slices.setdefault((g, k0, int(j["slice"])), np.zeros(ntiles[g], dtype=np.int32))[j["tile_begin"]:j["tile_end"]] += 1
                continue
        elif log2m >= 17:
            logg = max(1, log2m - 17)                               # a 64 KiB filter of 4-bit entries
            assert mode == 5 and nk == 1                           # scatter + replay, one k per job
            assert j["lds_bytes"] == (m >> logg) // 2 + 2 * 16 * 128 * 4    # the filter + two record queues per wave
        else:
            assert mode == 0 and nk * m <= j["lds_bytes"]          # the group's registers fit the LDS asked for
        cover[g][k0 - kmin:k0 - kmin + nk, j["tile_begin"]:j["tile_end"]] += 1
    for g, c in enumerate(cover):
        assert c.size == 0 or (c.min() == 1 and c.max() == 1), f"genome {g}: tiles covered {c.min()}..{c.max()} times"
    big = jobs[jobs["kclass"] == -2]
    for g, k in {(int(j["genome"]), int(j["kfirst"])) for j in big if j["tile_end"] > j["tile_begin"]}:
        for sl in range(1, 2 if k == 11 else 1):
            c = extra_slices.get((g, k, sl))
            assert c is not None and c.min() == 1 and c.max() == 1, f"genome {g} k {k} slice {sl}"
    if log2m >= 19:
        # ... for genomes with several times more tokens than the set can have members (the finish kernel hashes the
        # whole set once per 128 KiB index tile): on average >= tiles x 4^k bytes
        last, avg, tiles = (11 if log2m >= 20 else 10), sum(sizes) // max(1, len(sizes)), 1 << max(0, log2m - 17)
        while last >= 10 and avg < tiles * 4 ** last and not os.environ.get("DD_BIGMAP_ANY_SIZE"):
            last -= 1
        want = {k for k in range(max(kmin, 10), min(kmax, last) + 1)} if any(ntiles) else set()
        assert {int(k) for k in big["kfirst"]} == want
    return jobs


@pytest.mark.parametrize("name", sorted(SIZES))
@pytest.mark.parametrize("log2m", [10, 14, 16, 17, 18, 20])
def test_every_tile_of_every_k_is_covered_once(name, log2m):
    if name == "single_big" and log2m not in (14, 20):
        pytest.skip("one large case per register mode is enough")
    check(log2m, SIZES[name], 4, 40)


@pytest.mark.parametrize("krange", [(1, 64), (1, 1), (9, 10), (16, 17), (32, 33), (48, 49), (64, 64), (10, 20), (10, 11), (11, 12)])
def test_k_ranges_and_class_boundaries(krange):
    for log2m in (12, 14, 19, 20):
        check(log2m, SIZES["ragged"], *krange)


@pytest.mark.parametrize("sizes", [[0], [0, 0, 0], [0, 1, 0]])
def test_calls_with_only_empty_genomes(sizes):
    """A 0-byte FASTA at any register count (bucket mode has no epoch then): no job with tiles, nothing out of range."""
    for log2m in (14, 17, 18, 19, 20):
        jobs = check(log2m, sizes, 4, 40)
        real = jobs[jobs["tile_end"] > jobs["tile_begin"]]
        assert len(real) == (0 if not any(sizes) else len(real)) and all(int(j["genome"]) == 1 for j in real)


def test_launch_shape_of_the_headline_config():
    jobs = check(14, SIZES["cfg2"], 4, 40)
    by_class = {int(c): jobs[jobs["kclass"] == c] for c in np.unique(jobs["kclass"])}
    assert sorted(by_class) == [-1, 0, 1, 3]
    assert set(by_class[-1]["kfirst"]) == {4} and set(by_class[-1]["nk"]) == {6}
    assert by_class[1]["lds_bytes"].max() <= 80 * 1024     # two workgroups per CU
    assert by_class[0]["lds_bytes"].max() <= 48 * 1024     # three for the 32-bit class
    assert all(len(v) >= 2048 for v in by_class.values())  # >> 256 CUs x 2 resident workgroups


def test_bucket_mode_epochs(monkeypatch):
    """log2m >= 17: jobs come epoch by epoch (launch order), epochs are the same tile ranges for every row and
    double in length."""
    for env in ({}, {"DD_BUCKET_E0": "1", "DD_BUCKET_EMAX": "3"}, {"DD_BUCKET_E0": "2"}, {"DD_BUCKET_GB": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for log2m in (17, 18, 19, 20):
            jobs = check(log2m, SIZES["ragged"], 8, 35)
            real = jobs[(jobs["tile_end"] > jobs["tile_begin"]) & (jobs["kclass"] >= 0)]
            if "DD_BUCKET_GB" not in env:
                e0 = int(env.get("DD_BUCKET_E0", max(8, 4 * (1 << log2m) // TILE)))
                ntiles = [(n + TILE - 1) // TILE for n in SIZES["ragged"]]
                if "DD_BUCKET_E0" not in env and max(ntiles) <= e0 + e0 // 4:
                    e0 = max(e0, max(ntiles))   # genomes barely longer than the first epoch: one epoch
                emax = max(e0, int(env.get("DD_BUCKET_EMAX", 256)))
                edges = [0, e0]
                while edges[-1] < 1000:
                    edges.append(edges[-1] + min(emax, edges[-1]))
                for kc in np.unique(real["kclass"]):
                    sel = real[real["kclass"] == kc]
                    # launch order never goes back to an earlier epoch, and no job straddles an epoch edge
                    e_lo = np.searchsorted(edges, sel["tile_begin"].astype(np.int64), side="right")
                    e_hi = np.searchsorted(edges, sel["tile_end"].astype(np.int64) - 1, side="right")
                    assert np.all(np.diff(e_lo) >= 0) and np.array_equal(e_lo, e_hi), (env, log2m, kc)
        for k in env:
            monkeypatch.delenv(k)


def test_knobs_change_the_plan_not_the_coverage(monkeypatch):
    for env in ({"DD_BUCKET_E0": "2"}, {"DD_BUCKET_EMAX": "2", "DD_BUCKET_E0": "1"}, {"DD_BUCKET_CAP": "8"}, {"DD_BUCKET_GB": "1"}, {"DD_BIGMAP_ANY_SIZE": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for log2m in (14, 18):
            jobs = plan_sweep(log2m, SIZES["ragged"], 2, 40)
            ntiles = [(n + TILE - 1) // TILE for n in SIZES["ragged"]]
            cover = [np.zeros((39, nt), dtype=np.int32) for nt in ntiles]
            for j in jobs:
                if j["tile_end"] > j["tile_begin"]:
                    cover[j["genome"]][j["kfirst"] - 2:j["kfirst"] - 2 + j["nk"], j["tile_begin"]:j["tile_end"]] += 1
            assert all(c.size == 0 or (c.min() == 1 and c.max() == 1) for c in cover), env
        for k in env:
            monkeypatch.delenv(k)


def test_exact_set_class_is_planned_by_genome_size(monkeypatch):
    """log2m 20: k = 10, 11 go to the exact-set class for 50 Mbp genomes, only k = 10 for 10 Mbp ones, neither for
    5 Mbp ones (the set's finish kernel would hash more k-mers than the genome has tokens); DD_BIGMAP_ANY_SIZE forces it."""
    def big_ks(sizes, log2m=20):
        jobs = check(log2m, sizes, 4, 40)
        return sorted({int(k) for k in jobs[jobs["kclass"] == -2]["kfirst"]})
    assert big_ks([50_600_000] * 10) == [10, 11]
    assert big_ks([10_000_000] * 10) == [10]
    assert big_ks([5_000_000] * 64) == []
    assert big_ks([50_600_000] * 10, 19) == [10] and big_ks([3_000_000] * 4, 19) == []
    monkeypatch.setenv("DD_BIGMAP_ANY_SIZE", "1")
    assert big_ks([5_000_000] * 64) == [10, 11] and big_ks([70_000, 0, 5], 19) == [10]
