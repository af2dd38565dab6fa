// dd_plan.hip -- K1 job tables (host code only; see dd_plan.h).
//
// What the reference leaves to `parallel -j 95%` (one process per k, /root/reference/lib/
// huffman_dandd.py:214-218) is decided here: how the (genome x k x token-tile) space of one sketch call
// is cut into workgroup jobs, in which order they are handed out, and with how much LDS.
#include "dd_plan.h"

#include <stdlib.h>

#include <algorithm>

namespace dd {

PlanKnobs PlanKnobs::from_env() {
    PlanKnobs k;
    k.bigmap_any_size = getenv("DD_BIGMAP_ANY_SIZE") != nullptr;
    if (const char* e = getenv("DD_BUCKET_E0")) k.bucket_e0_tiles = (size_t)std::max(1, atoi(e));
    if (const char* e = getenv("DD_BUCKET_EMAX")) k.bucket_emax_tiles = (size_t)std::max(1, atoi(e));
    if (const char* e = getenv("DD_BUCKET_CAP")) k.bucket_cap_chunks = (size_t)std::max(1, atoi(e));
    if (const char* e = getenv("DD_BUCKET_GB")) k.bucket_budget = (size_t)std::max(1, atoi(e)) << 30;
    return k;
}

namespace {

constexpr int kThreads = 1024;
constexpr size_t kTileTokens = (size_t)kThreads * kSegTokens;

size_t tiles_of(size_t nbytes) { return (nbytes + kTileTokens - 1) / kTileTokens; }  // tokens <= bytes

SweepJob make_job(int genome, int kfirst, int nk, int kmin, size_t t0, size_t t1) {
    SweepJob j;
    j.genome = genome;
    j.kfirst = kfirst;
    j.nk = nk;
    j.krow = kfirst - kmin;
    j.tile_begin = (unsigned)t0;
    j.tile_end = (unsigned)t1;
    j.slice = 0;
    return j;
}

}  // namespace

bool plan_bigmap_range(int log2m, int kmin, int kmax, const PlanKnobs& knobs, const size_t* nbytes, int ngenomes, int* ka, int* kb) {
    int last = log2m >= kBucketFromLog2m ? bigmap_last_k(log2m) : 0;
    // The finish kernel hashes a k's whole set (up to 4^k / 2 k-mers) once per 128 KiB index tile of the row; going
    // through the record stream hashes every token once.  Small genomes (64 x 5 Mbp at log2m 20: 7 ms of finish
    // kernel for 640 M tokens' worth of rows) keep the hashed path, whose unfiltered first epoch covers most of
    // them anyway.
    size_t total = 0;
    for (int g = 0; g < ngenomes; ++g) total += nbytes[g];
    const size_t avg = ngenomes > 0 ? total / (size_t)ngenomes : 0;
    const size_t tiles = (size_t)1 << std::max(0, log2m - 17);
    while (!knobs.bigmap_any_size && last >= kBigmapMinK && avg < tiles * ((size_t)1 << (2 * last))) --last;
    const int a = std::max(kmin, kBigmapMinK), b = std::min(kmax, last);
    if (a > b) return false;
    if (ka) *ka = a;
    if (kb) *kb = b;
    return true;
}

std::vector<SweepClass> plan_sweep(int log2m, int canonical, const size_t* nbytes, int ngenomes, int kmin,
                                   int kmax, const PlanKnobs& knobs) {
    std::vector<SweepClass> classes;
    if (ngenomes <= 0 || kmin < 1 || kmax > 64 || kmin > kmax) return classes;
    const int p = log2m;
    const size_t m = (size_t)1 << p;
    // log2m <= 16: the registers of a job's k-group live in LDS.  80 KiB per workgroup = two 1024-thread workgroups (8 waves
    // per SIMD) per CU: measured 1.35x faster than one 160 KiB workgroup (4 waves per SIMD cannot cover the LDS latency of
    // the dependent hash -> read -> compare chain).  log2m >= 17: bucket mode -- the registers stay in HBM and are reached
    // through record streams (17: 22.8 Gbp/s that way, 18.7 with one 128 KiB row per workgroup in LDS).
    const bool bucket_mode = p >= kBucketFromLog2m;
    const size_t lds_budget = std::max<size_t>(80 * 1024, m);
    const int slots = (int)std::min<size_t>(64, lds_budget / m);
    const bool use_bitmaps = kmin <= kBitmapMaxK;
    int big_ka = 0, big_kb = 0;
    const bool use_big = plan_bigmap_range(p, kmin, kmax, knobs, nbytes, ngenomes, &big_ka, &big_kb);

    size_t total_tiles = 0, max_tiles = 0;
    for (int g = 0; g < ngenomes; ++g) {
        total_tiles += tiles_of(nbytes[g]);
        max_tiles = std::max(max_tiles, tiles_of(nbytes[g]));
    }

    // Bucket mode (scatter + replay, dd_kernels.h): epochs are ranges of token tiles, the same for every row.
    // The first holds two tokens per register (nothing can be filtered before the registers have been
    // seen), each later one is as long as everything before it -- the filter's bounds rise by about
    // one per doubling -- up to the length whose worst case (every update survives) fits the HBM budget.
    // a 64 KiB filter of 4-bit entries: one per 2^(p-17) registers (16 bytes at least)
    int bucket_logg = std::max(1, p - 17);
    while (bucket_logg > 0 && (m >> bucket_logg) / 2 < 16) --bucket_logg;
    const int nb_log2 = std::max(0, p - 16);  // index tiles of 64 KiB
    std::vector<size_t> epoch_edge;           // epoch e covers tiles [epoch_edge[e], epoch_edge[e+1])
    size_t epoch_longest = 0, bucket_row_tokens = 0;
    if (bucket_mode) {
        int first_hashed = use_bitmaps ? std::max(kmin, kBitmapMaxK + 1) : kmin;
        if (use_big) first_hashed = std::max(first_hashed, big_kb + 1);
        const size_t nrows = (size_t)ngenomes * (size_t)std::max(0, kmax - first_hashed + 1);
        // (the first epoch runs unfiltered and cheaply -- every register is zero, every update a record --, so it is
        // made four tokens per register long, 8 tiles at least: measured best at log2m 18, 19 and 20 with the
        // dense record stream; two per register before that)
        size_t e0 = knobs.bucket_e0_tiles ? knobs.bucket_e0_tiles : std::max<size_t>(8, 4 * m / kTileTokens);
        // (genomes barely longer than that -- 5 Mbp at log2m 20 -- are not given a second, filtered epoch for their
        // last few tiles: 37.7 -> 33.9 ms for 64 x 5 Mbp)
        if (!knobs.bucket_e0_tiles && max_tiles <= e0 + e0 / 4) e0 = std::max<size_t>(e0, max_tiles);
        size_t emax = knobs.bucket_emax_tiles;
        bucket_row_tokens = knobs.bucket_budget / (std::max<size_t>(1, nrows) * 9 / 2);  // 4 B per record + slack
        // (so many rows that the budget cannot hold four tokens per register of each: a shorter first epoch rather
        // than one whose records overflow into the compare-and-swap path)
        if (!knobs.bucket_e0_tiles) e0 = std::max<size_t>(1, std::min(e0, bucket_row_tokens / kTileTokens));
        if (!emax) emax = std::min<size_t>(256, bucket_row_tokens / kTileTokens);
        emax = std::max(emax, e0);
        epoch_edge.push_back(0);
        size_t len = e0;
        while (epoch_edge.back() < max_tiles) {
            epoch_longest = std::max(epoch_longest, len);
            epoch_edge.push_back(epoch_edge.back() + len);
            len = std::min(emax, epoch_edge.back());
        }
    }

    const int lo0 = use_big ? big_kb + 1 : (use_bitmaps ? kBitmapMaxK + 1 : 1);
    const struct { int kc, ka, kb; } class_tab[6] = {
        {kBitmapClass, 1, use_bitmaps ? kBitmapMaxK : 0}, {kBigmapClass, big_ka, use_big ? big_kb : 0},
        {0, lo0, 16}, {1, 17, 32}, {3, 33, 48}, {2, 49, 64}};
    for (const auto& ct : class_tab) {
        const int kc = ct.kc;
        const int ka = std::max(kmin, ct.ka), kb = std::min(kmax, ct.kb);
        if (ka > kb) continue;
        const int nks = kb - ka + 1;
        SweepClass sc;
        sc.kclass = kc;
        sc.kfirst = ka;
        sc.klast = kb;
        sc.plan.log2m = p;
        sc.plan.canonical = canonical;
        sc.plan.threads = kThreads;
        int max_nk = 0;

        if (kc == kBigmapClass) {
            // one (genome, k, slice) pass over all the genome's tiles, cut into ~1024 jobs over the class (a job loads
            // and merges its 128 KiB slice whatever its length: not below 8 tiles); tile-range major, so the passes
            // that read the same tokens run at the same time
            size_t passes = 0;
            for (int k = ka; k <= kb; ++k) passes += (size_t)bigmap_slices(k, canonical != 0);
            const size_t tpj = std::max<size_t>(8, (total_tiles * passes + 1023) / 1024);
            for (size_t t0 = 0; t0 < max_tiles; t0 += tpj)
                for (int g = 0; g < ngenomes; ++g) {
                    const size_t ntiles = tiles_of(nbytes[g]);
                    if (t0 >= ntiles) continue;
                    for (int k = ka; k <= kb; ++k)
                        for (int sl = 0; sl < bigmap_slices(k, canonical != 0); ++sl) {
                            SweepJob j = make_job(g, k, 1, kmin, t0, std::min(ntiles, t0 + tpj));
                            j.slice = sl;
                            sc.jobs.push_back(j);
                        }
                }
            max_nk = 1;
        } else if (bucket_mode && kc != kBitmapClass) {
            // one k per job; row r of the class goes to XCD r % 8 in every epoch (its filter and the token
            // tiles its ks share stay in that XCD's L2); job 8*i + x is the i-th job of XCD x
            const size_t nepochs = epoch_edge.size() - 1;
            size_t max_jobs_row_epoch = 1;
            const SweepJob idle = make_job(0, ka, 1, kmin, 0, 0);  // empty tile range: the workgroup exits at once
            for (size_t e = 0; e < nepochs; ++e) {
                sc.epoch_begin.push_back(sc.jobs.size());
                const size_t t_lo = epoch_edge[e], t_hi = epoch_edge[e + 1];
                size_t tile_rows = 0;
                for (int g = 0; g < ngenomes; ++g) {
                    const size_t nt = tiles_of(nbytes[g]);
                    if (nt > t_lo) tile_rows += (std::min(nt, t_hi) - t_lo) * (size_t)nks;
                }
                if (!tile_rows) continue;
                // ~16 jobs per resident workgroup slot (2048 .. 16384 scatter jobs per launch measured: 8192 is best at
                // log2m 18 and 20); a job reloads its row's filter, so not below 2 tiles once the epoch is long enough
                const size_t slots = 8192;
                const size_t tpj = std::max<size_t>(std::min<size_t>(2, t_hi - t_lo), (tile_rows + slots - 1) / slots);
                std::vector<std::vector<SweepJob>> per_xcd(8);
                int row = 0;
                for (int g = 0; g < ngenomes; ++g) {
                    const size_t nt = std::min(tiles_of(nbytes[g]), t_hi);
                    for (int q = 0; q < nks; ++q, ++row) {
                        size_t nj = 0;
                        for (size_t t0 = t_lo; t0 < nt; t0 += tpj, ++nj)
                            per_xcd[row % 8].push_back(make_job(g, ka + q, 1, kmin, t0, std::min(nt, t0 + tpj)));
                        max_jobs_row_epoch = std::max(max_jobs_row_epoch, nj);
                    }
                }
                size_t longest = 0;
                for (auto& v : per_xcd) longest = std::max(longest, v.size());
                for (size_t i = 0; i < longest; ++i)
                    for (int x = 0; x < 8; ++x) sc.jobs.push_back(i < per_xcd[x].size() ? per_xcd[x][i] : idle);
            }
            sc.epoch_begin.push_back(sc.jobs.size());
            max_nk = 1;
            // Capacity of a row's stream.  Only the first epoch turns every token into a record; a later epoch that
            // starts after s tokens and is L long leaves about m L / s of them (m or fewer with the doubling schedule),
            // so four records per register cover it several times over -- and what should still not fit goes to the
            // registers by compare-and-swap, exactly (dd_sweep.hip).  (Sized for the longest epoch's every token the areas
            // of a 10 x 50 Mbp call at log2m 20 were 20 GB; 5.4 GB measure the same.)  A call with so many rows that even
            // the first epoch's worst case exceeds the budget gets what the budget allows.
            // (no epoch at all when every genome of the call is empty: epoch_edge is {0} then)
            const size_t first_tokens = nepochs ? (epoch_edge[1] - epoch_edge[0]) * kTileTokens : 0;
            const size_t per_row = std::min(std::max(first_tokens, 4 * m), bucket_row_tokens);
            // (the first epoch's binned tiles take 70 chunks of stream per tile of tokens, not 64: dd_sweep.hip)
            sc.plan.cap_chunks = (unsigned)(knobs.bucket_cap_chunks ? knobs.bucket_cap_chunks
                                                : (per_row + per_row / 8) / 1024 + max_jobs_row_epoch * (kThreads / 64) + 16);
            sc.plan.logg = bucket_logg;
            sc.plan.nb_log2 = nb_log2;
            sc.plan.nepochs = (int)nepochs;
        } else {
            // The 32-bit class needs few enough VGPRs for 12 waves per SIMD, and measures ~7 % faster with
            // three 48 KiB workgroups per CU than with two of 80 KiB; the wider classes do not.
            int slots_c = slots;
            if (kc == 0) slots_c = (int)std::max<size_t>(1, std::min<size_t>(slots, (48 * 1024) / m));
            const int ngroups = kc == kBitmapClass ? 1 : (nks + slots_c - 1) / slots_c;
            // ~jobs_per_cu jobs per CU over the whole class so the dispatcher can balance the tail
            const size_t target_jobs = 256 * 32;
            size_t tiles_per_job = std::max<size_t>(1, (total_tiles * ngroups + target_jobs - 1) / target_jobs);
            // A job loads and merges its k-group's registers whatever its length: small calls (one batch of the
            // ingestion pipeline: 2 x 50 Mbp, 26 x 5 Mbp) keep four tiles per job as long as that still leaves
            // four jobs per workgroup slot (measured 5-7 % on such calls; large calls are unaffected).
            for (size_t want = 4; want > tiles_per_job; want >>= 1)
                if (total_tiles * ngroups >= want * 2048) {
                    tiles_per_job = want;
                    break;
                }
            // Tile-major order: workgroups that run concurrently work on different (genome, k-group)
            // slabs, so each slab has been warmed by its earlier tiles when its later jobs start.
            // Jobs are handed out in table order; the last quarter of the tiles goes out in jobs a
            // quarter the size, so the launch does not end waiting on a few full-size stragglers.
            const size_t taper_from = max_tiles - max_tiles / 4;
            const size_t full_tiles_per_job = tiles_per_job;
            sc.jobs.reserve((total_tiles / tiles_per_job + (size_t)ngenomes) * ngroups * 2 + 64);
            for (size_t t0 = 0; t0 < max_tiles; t0 += tiles_per_job) {
                if (t0 >= taper_from) tiles_per_job = std::max<size_t>(1, full_tiles_per_job / 4);
                for (int g = 0; g < ngenomes; ++g) {
                    const size_t ntiles = tiles_of(nbytes[g]);
                    if (t0 >= ntiles) continue;
                    int kcur = ka;
                    for (int q = 0; q < ngroups; ++q) {
                        const int nk = nks / ngroups + (q < nks % ngroups ? 1 : 0);
                        sc.jobs.push_back(make_job(g, kcur, nk, kmin, t0, std::min(ntiles, t0 + tiles_per_job)));
                        max_nk = std::max(max_nk, nk);
                        kcur += nk;
                    }
                }
            }
            // The k-groups of one (genome, tile range) read the same token bytes.  Workgroups are dealt
            // round-robin over the 8 XCDs in blockIdx order, so within every block of 8 units x ngroups
            // jobs emit group-major: the groups of unit i then sit at indices i, i+8, i+16, ... = one
            // XCD, back to back, and the re-reads hit that XCD's L2 instead of HBM (speed only).
            if (ngroups > 1) {
                const size_t nunits = sc.jobs.size() / ngroups;
                std::vector<SweepJob> re;
                re.reserve(sc.jobs.size());
                for (size_t u0 = 0; u0 < nunits; u0 += 8) {
                    const size_t nu = std::min<size_t>(8, nunits - u0);
                    for (int q = 0; q < ngroups; ++q)
                        for (size_t u = 0; u < nu; ++u) re.push_back(sc.jobs[(u0 + u) * ngroups + q]);
                }
                sc.jobs.swap(re);
            }
        }
        if (sc.jobs.empty()) continue;
        if (kc == kBitmapClass) {
            sc.plan.mode = 0;
            sc.plan.lds_bytes = (bitmap_offset(kb) + bitmap_words(kb) - bitmap_offset(ka)) * 4;
        } else if (kc == kBigmapClass) {
            sc.plan.mode = 0;
            sc.plan.lds_bytes = kBigmapSliceWords * 4;
        } else if (bucket_mode) {
            sc.plan.mode = kBucketMode;  // the 4-bit filter, then two 128-entry record queues per wave (dd_sweep.hip: scatter_kernel)
            sc.plan.lds_bytes = (int)((m >> bucket_logg) / 2) + (kThreads / 64) * 128 * 4 * 2;
        } else {
            sc.plan.mode = 0;
            sc.plan.lds_bytes = (int)((size_t)max_nk * m);
        }
        classes.push_back(std::move(sc));
    }
    // one record-area geometry for all rows of the call
    unsigned cap = 0;
    for (const SweepClass& sc : classes)
        if (sc.plan.mode == kBucketMode) cap = std::max(cap, sc.plan.cap_chunks);
    for (SweepClass& sc : classes)
        if (sc.plan.mode == kBucketMode) sc.plan.cap_chunks = cap;
    return classes;
}

}  // namespace dd
