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
    if (const char* e = getenv("DD_LDS_KB")) {
        k.lds_budget = std::min<size_t>((size_t)sweep_max_lds_bytes(), (size_t)std::max(1, atoi(e)) * 1024);
        k.lds_budget_forced = true;
    }
    if (const char* e = getenv("DD_JOBS_PER_CU")) k.jobs_per_cu = (size_t)std::max(1, atoi(e));
    if (const char* e = getenv("DD_JOBS_PER_ROW")) k.jobs_per_row = (size_t)std::max(1, atoi(e));
    if (const char* e = getenv("DD_GLOBAL_FROM_P")) k.global_from_p = std::max(16, std::min(18, atoi(e)));
    k.use_bitmaps = !getenv("DD_NO_BITMAP");
    k.use_bigmaps = k.use_bitmaps && !getenv("DD_NO_BIGMAP");
    k.bigmap_any_size = getenv("DD_BIGMAP_ANY_SIZE") != nullptr;
    k.filter = !getenv("DD_NO_FILTER");
    k.xcd_affinity = !getenv("DD_NO_XCD_AFFINITY");
    k.taper = !getenv("DD_NO_TAPER");
    k.buckets = !getenv("DD_NO_BUCKETS");
    if (const char* e = getenv("DD_BUCKET_E0")) k.bucket_e0_tiles = (size_t)std::max(1, atoi(e));
    if (const char* e = getenv("DD_BUCKET_EMAX")) k.bucket_emax_tiles = (size_t)std::max(1, atoi(e));
    if (const char* e = getenv("DD_BUCKET_CAP")) k.bucket_cap_chunks = (size_t)std::max(1, atoi(e));
    if (const char* e = getenv("DD_BUCKET_LOGG")) k.bucket_logg = std::max(0, std::min(8, atoi(e))) + 1;  // stored + 1: 0 = not set
    if (const char* e = getenv("DD_BUCKET_FBITS")) k.bucket_fbits = atoi(e) == 4 ? 4 : 8;
    if (const char* e = getenv("DD_BUCKET_PROBE")) k.bucket_probe = atoi(e) ? 1 : 0;
    if (const char* e = getenv("DD_BUCKET_SLOTS")) k.bucket_slots = (size_t)std::max(64, atoi(e));
    if (const char* e = getenv("DD_ROW_GROUP_MB")) k.row_group_mb = (size_t)std::max(0, atoi(e));
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
    const size_t m = (size_t)1 << log2m;
    const bool global_regs = m > (size_t)sweep_max_lds_bytes() || log2m >= knobs.global_from_p;
    const bool bucket_mode = global_regs && knobs.buckets && knobs.filter;
    int last = (bucket_mode && knobs.use_bitmaps && knobs.use_bigmaps) ? bigmap_last_k(log2m) : 0;
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
    // Registers stay in HBM when one array does not fit LDS (log2m >= 18), behind a 64 KiB LDS filter
    // (one byte per 4 / 8 / 16 registers) unless that is switched off.
    const bool global_regs = m > (size_t)sweep_max_lds_bytes() || p >= knobs.global_from_p;
    const int filter_logg = (global_regs && knobs.filter) ? std::max(2, p - 16) : 0;
    // 80 KiB per workgroup = two 1024-thread workgroups (8 waves per SIMD) per CU: measured 1.35x faster
    // than one 160 KiB workgroup (4 waves per SIMD cannot cover the LDS latency of the dependent
    // hash -> read -> compare chain); a single array larger than that takes what it needs.
    const size_t lds_budget = std::max(knobs.lds_budget, m);
    const int slots = global_regs ? 64 : (int)std::min<size_t>(64, lds_budget / m);
    const bool use_bitmaps = knobs.use_bitmaps && kmin <= kBitmapMaxK;
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
    const bool bucket_mode = global_regs && knobs.buckets && knobs.filter;
    // ks per FILTERED scatter job (the first epoch's jobs always hold one).  1; DD_BUCKET_NK=2 for A/B runs only: two
    // were measured slower both at one workgroup per CU and, with 16 KiB filters, at two (profiles/r03_bucket_path.txt)
    const int bucket_nk_knob = getenv("DD_BUCKET_NK") && atoi(getenv("DD_BUCKET_NK")) == 2 ? 2 : 1;
    const int bucket_probe = knobs.bucket_probe >= 0 ? knobs.bucket_probe : 1;
    const int bucket_fbits = knobs.bucket_fbits ? knobs.bucket_fbits : 4;  // measured: 4-bit entries win at log2m 18, 19 and 20
    const int bucket_nk = (bucket_probe && bucket_fbits == 4) ? bucket_nk_knob : 1;  // (the only two-k kernels built)
    // a 64 KiB filter: 2^(p-16) registers per byte-wide entry, half as many per 4-bit entry
    int bucket_logg = knobs.bucket_logg ? knobs.bucket_logg - 1 : std::max(1, p - 16 - (bucket_fbits == 4 ? 1 : 0));
    // (a knob that asks for more filter than a workgroup can hold gets the finest one that fits; 16 bytes at least)
    while (bucket_nk * ((m >> bucket_logg) * bucket_fbits / 8 + (kThreads / 64) * 1024) > (size_t)sweep_max_lds_bytes()) ++bucket_logg;
    while (bucket_logg > 0 && (m >> bucket_logg) * bucket_fbits / 8 < 16) --bucket_logg;
    // index tiles of 64 KiB (DD_BUCKET_TILE_LOG2=17: 128 KiB, one replay workgroup per CU, segments twice as long -- A/B)
    const int tile_log2 = getenv("DD_BUCKET_TILE_LOG2") ? std::max(16, std::min(17, atoi(getenv("DD_BUCKET_TILE_LOG2")))) : 16;
    const int nb_log2 = std::max(0, p - tile_log2);
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
            // (row groups: single-epoch calls with the binned first epoch only -- many small genomes, the regime whose
            // record traffic is the bound; a row's area is what cap_chunks below comes to: ~(1 + 1/8) x 4 B per token)
            int group_rows_e0 = 0;
            if (knobs.row_group_mb && nepochs == 1 && !getenv("DD_BUCKET_NO_FIRST")) {
                const size_t row_bytes = std::max<size_t>(1, (max_tiles * kTileTokens * 9 / 8) * 4);
                group_rows_e0 = (int)std::max<size_t>(8, (knobs.row_group_mb << 20) / row_bytes / 8 * 8);
            }
            size_t max_jobs_row_epoch = 1;
            const SweepJob idle = make_job(0, ka, 1, kmin, 0, 0);
            for (size_t e = 0; e < nepochs; ++e) {
                sc.epoch_begin.push_back(sc.jobs.size());
                const size_t t_lo = epoch_edge[e], t_hi = epoch_edge[e + 1];
                size_t tile_rows = 0;
                for (int g = 0; g < ngenomes; ++g) {
                    const size_t nt = tiles_of(nbytes[g]);
                    if (nt > t_lo) tile_rows += (std::min(nt, t_hi) - t_lo) * (size_t)nks;
                }
                if (!tile_rows) continue;
                // ~16 jobs per resident workgroup slot; a job reloads its rows' filters, so not below 2 tiles
                // once the epoch is long enough to allow it
                const size_t slots = knobs.bucket_slots;  // (2048 .. 16384 measured: 8192 is best at log2m 18 and 20)
                const int nk_e = (e == 0 && !getenv("DD_BUCKET_NO_FIRST")) ? 1 : bucket_nk;
                size_t tpj = std::max<size_t>(std::min<size_t>(2, t_hi - t_lo), (tile_rows / nk_e + slots - 1) / slots);
                // (row groups: a group is one round of workgroups; one tile per job fills the chip best -- DD_ROW_GROUP_TPJ for A/B)
                if (group_rows_e0 && e == 0) tpj = getenv("DD_ROW_GROUP_TPJ") ? (size_t)std::max(1, atoi(getenv("DD_ROW_GROUP_TPJ"))) : 1;
                std::vector<std::vector<SweepJob>> per_xcd(8);
                int row = 0;
                for (int g = 0; g < ngenomes; ++g) {
                    const size_t nt = std::min(tiles_of(nbytes[g]), t_hi);
                    for (int q = 0; q < nks; q += nk_e, ++row) {  // one or two consecutive ks per job
                        const int nkj = std::min(nk_e, nks - q);
                        size_t nj = 0;
                        for (size_t t0 = t_lo; t0 < nt; t0 += tpj, ++nj)
                            per_xcd[knobs.xcd_affinity ? row % 8 : 0].push_back(make_job(g, ka + q, nkj, kmin, t0, std::min(nt, t0 + tpj)));
                        max_jobs_row_epoch = std::max(max_jobs_row_epoch, nj);
                    }
                }
                if (group_rows_e0 && e == 0) {
                    // row groups: the same order, cut at every group_rows-th row; each group's XCD lists are padded to
                    // one length of their own, so that a group is a contiguous range of the table
                    sc.group_rows = group_rows_e0;
                    std::vector<std::vector<SweepJob>> gx(8);
                    int grow = 0;
                    auto flush = [&]() {
                        size_t longest = 0;
                        for (auto& v : gx) longest = std::max(longest, v.size());
                        sc.group_begin.push_back(sc.jobs.size());
                        for (size_t i = 0; i < longest; ++i)
                            for (int x = 0; x < 8; ++x) sc.jobs.push_back(i < gx[x].size() ? gx[x][i] : idle);
                        for (auto& v : gx) v.clear();
                    };
                    for (int g = 0; g < ngenomes; ++g) {
                        const size_t nt = std::min(tiles_of(nbytes[g]), t_hi);
                        for (int q = 0; q < nks; ++q, ++grow) {
                            if (grow && grow % group_rows_e0 == 0) flush();
                            for (size_t t0 = t_lo; t0 < nt; t0 += tpj)
                                gx[knobs.xcd_affinity ? grow % 8 : 0].push_back(make_job(g, ka + q, 1, kmin, t0, std::min(nt, t0 + tpj)));
                        }
                    }
                    flush();
                    sc.group_begin.push_back(sc.jobs.size());
                } else if (!knobs.xcd_affinity) {
                    sc.jobs.insert(sc.jobs.end(), per_xcd[0].begin(), per_xcd[0].end());
                } else {
                    size_t longest = 0;
                    for (auto& v : per_xcd) longest = std::max(longest, v.size());
                    for (size_t i = 0; i < longest; ++i)
                        for (int x = 0; x < 8; ++x) sc.jobs.push_back(i < per_xcd[x].size() ? per_xcd[x][i] : idle);
                }
            }
            sc.epoch_begin.push_back(sc.jobs.size());
            max_nk = bucket_nk;
            // Every token of the longest epoch may leave a record (nothing is filtered while the registers
            // are still empty), plus the partly filled chunk every wave of every job leaves.  Records beyond
            // the capacity are not lost: they go straight to the row by compare-and-swap (dd_sweep.hip).
            // (a call with so many rows that even the first epoch's worst case exceeds the budget gets what
            // the budget allows; the overflow path keeps it exact)
            // Capacity of a row's stream.  Only the first epoch turns every token into a record; a later epoch that
            // starts after s tokens and is L long leaves about m L / s of them (m or fewer with the doubling schedule),
            // so four records per register cover it several times over -- and what should still not fit goes to the
            // registers by compare-and-swap, exactly.  (Sized for the longest epoch's every token the areas of a
            // 10 x 50 Mbp call at log2m 20 were 20 GB; 5.4 GB measure the same 26.8 ms.)
            // (no epoch at all when every genome of the call is empty: epoch_edge is {0} then)
            const size_t first_tokens = nepochs ? (epoch_edge[1] - epoch_edge[0]) * kTileTokens : 0;
            const size_t per_row = std::min(std::max(first_tokens, 4 * m), bucket_row_tokens);
            // (the first epoch's binned tiles take 70 chunks of stream per tile of tokens, not 64: dd_sweep.hip)
            sc.plan.cap_chunks = (unsigned)(knobs.bucket_cap_chunks ? knobs.bucket_cap_chunks
                                                : (per_row + per_row / 8) / 1024 + max_jobs_row_epoch * (kThreads / 64) + 16);
            sc.plan.logg = bucket_logg;
            sc.plan.fbits = bucket_fbits;
            sc.plan.nk_job = bucket_nk;
            sc.plan.probe = bucket_probe;
            sc.plan.nb_log2 = nb_log2;
            sc.plan.nepochs = (int)nepochs;
        } else if (global_regs && kc != kBitmapClass && (filter_logg || knobs.xcd_affinity)) {
            // Registers in HBM.  The arrays a workgroup touches should sit in ITS XCD's 4 MiB L2: k-groups
            // are cut to <= 3 MiB of arrays (one k with the filter), each (genome, k-group) row is given to
            // one XCD, and because workgroups are dealt round-robin over the 8 XCDs in blockIdx order, job
            // 8*i + x is the i-th job of XCD x.  Placement is a speed assumption only: every register update
            // is an agent-scope atomic, correct wherever the workgroup lands.
            const int g_l2 = filter_logg ? 1 : (int)std::max<size_t>(1, ((size_t)3 << 20) / m);
            const int ngr = (nks + g_l2 - 1) / g_l2;
            std::vector<std::vector<SweepJob>> per_xcd(8);
            int row = 0;
            for (int g = 0; g < ngenomes; ++g) {
                const size_t ntiles = tiles_of(nbytes[g]);
                size_t jobs_per_row = 128;  // >= 64 resident workgroups share a row
                if (filter_logg) {
                    // a filtered job learns its filter as it goes (bounds rise only where it probes), so
                    // jobs are long: >= 48 tiles (3 M tokens) each, ~4096 jobs over the launch, >= 4 per row
                    const size_t nrows = (size_t)ngenomes * ngr;
                    jobs_per_row = std::max<size_t>(4, std::min<size_t>(4096 / std::max<size_t>(1, nrows), ntiles / 48));
                    if (knobs.jobs_per_row) jobs_per_row = knobs.jobs_per_row;
                }
                const size_t tpj = std::max<size_t>(1, ntiles / jobs_per_row);
                int kcur = ka;
                for (int q = 0; q < ngr; ++q, ++row) {
                    const int nk = nks / ngr + (q < nks % ngr ? 1 : 0);
                    for (size_t t0 = 0; t0 < ntiles; t0 += tpj)
                        per_xcd[row % 8].push_back(make_job(g, kcur, nk, kmin, t0, std::min(ntiles, t0 + tpj)));
                    max_nk = std::max(max_nk, nk);
                    kcur += nk;
                }
            }
            size_t longest = 0;
            for (auto& v : per_xcd) longest = std::max(longest, v.size());
            const SweepJob idle = make_job(0, ka, 1, kmin, 0, 0);  // empty tile range: the workgroup exits at once
            for (size_t i = 0; i < longest; ++i)
                for (int x = 0; x < 8; ++x) sc.jobs.push_back(i < per_xcd[x].size() ? per_xcd[x][i] : idle);
        } else {
            // The 32-bit class needs few enough VGPRs for 12 waves per SIMD, and measures ~7 % faster with
            // three 48 KiB workgroups per CU than with two of 80 KiB; the wider classes do not.
            int slots_c = slots;
            if (kc == 0 && !global_regs && !knobs.lds_budget_forced)
                slots_c = (int)std::max<size_t>(1, std::min<size_t>(slots, (48 * 1024) / m));
            const int ngroups = kc == kBitmapClass ? 1 : (nks + slots_c - 1) / slots_c;
            // ~jobs_per_cu jobs per CU over the whole class so the dispatcher can balance the tail
            const size_t target_jobs = 256 * knobs.jobs_per_cu;
            size_t tiles_per_job = std::max<size_t>(1, (total_tiles * ngroups + target_jobs - 1) / target_jobs);
            // A job loads and merges its k-group's registers whatever its length: small calls (one batch of the
            // ingestion pipeline: 2 x 50 Mbp, 26 x 5 Mbp) keep four tiles per job as long as that still leaves
            // four jobs per workgroup slot (measured 5-7 % on such calls; large calls are unaffected).
            if (!knobs.lds_budget_forced)
                for (size_t want = 4; want > tiles_per_job; want >>= 1)
                    if (total_tiles * ngroups >= want * 2048) {
                        tiles_per_job = want;
                        break;
                    }
            // Tile-major order: workgroups that run concurrently work on different (genome, k-group)
            // slabs, so each slab has been warmed by its earlier tiles when its later jobs start.
            // Jobs are handed out in table order; the last quarter of the tiles goes out in jobs a
            // quarter the size, so the launch does not end waiting on a few full-size stragglers.
            const size_t taper_from = knobs.taper ? max_tiles - max_tiles / 4 : max_tiles;
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
            if (ngroups > 1 && knobs.xcd_affinity) {
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
            sc.plan.mode = kBucketMode;  // the filter, then a 128-entry record queue per wave
            sc.plan.lds_bytes = bucket_nk * ((int)((m >> bucket_logg) * bucket_fbits / 8) + (kThreads / 64) * 128 * 4 * (bucket_probe ? 2 : 1));
        } else if (filter_logg) {
            sc.plan.mode = filter_logg;  // the filter, then a 128-entry candidate queue per wave (dd_sweep.hip)
            sc.plan.lds_bytes = (int)(m >> filter_logg) + (kThreads / 64) * 128 * 4;
        } else if (global_regs) {
            sc.plan.mode = 1;
            sc.plan.lds_bytes = 0;
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
