// dd_plan.h -- the K1 job tables: which workgroup sketches which (genome, k-group, tile range).
// Pure host code, no HIP calls: dd_sketch_device (dd_api.hip) uploads the tables, tests inspect them
// through dd_plan_sweep (include/dandd_hip.h) without a GPU.
#pragma once
#include <stddef.h>

#include <vector>

#include "dd_kernels.h"

namespace dd {

constexpr int kBitmapClass = -1;  // SweepClass::kclass of the small-k presence-bitmap class
constexpr int kBigmapClass = -2;  // ... of the k = 10 (, 11) big-bitmap class of log2m >= 19 (dd_kernels.h)

struct SweepClass {
    int kclass;      // kBitmapClass, or the window class of sweep_kernel: 0 (k <= 16), 1 (<= 32), 3 (33..48), 2 (49..64)
    int kfirst, klast;
    SweepPlan plan;  // launch shape of the class (mode, LDS bytes, threads)
    std::vector<SweepJob> jobs;
    // bucket mode (plan.mode == kBucketMode): the jobs of epoch e are jobs[epoch_begin[e] .. epoch_begin[e+1]);
    // one scatter launch per (class, epoch), one replay launch per epoch over the rows of all classes
    std::vector<size_t> epoch_begin;
    // row groups (knobs.row_group_mb; single-epoch calls only): the class's rows -- numbered genome * nks + (k - kfirst) --
    // in groups of group_rows; the scatter jobs of group i are jobs[group_begin[i] .. group_begin[i+1]), and the group's
    // replay follows them at once, while its records are still in the 256 MiB memory-side cache.  Empty = one launch per epoch.
    std::vector<size_t> group_begin;
    int group_rows = 0;
};

// development knobs, read from the environment by from_env() (README.md lists them)
struct PlanKnobs {
    size_t lds_budget = 80 * 1024;  // per workgroup: two 1024-thread workgroups (8 waves/SIMD) per CU
    bool lds_budget_forced = false;
    size_t jobs_per_cu = 32;
    size_t jobs_per_row = 0;        // filtered mode; 0 = heuristic
    int global_from_p = 17;         // registers stay in HBM from this log2m on (17: 22.8 Gbp/s through scatter + replay, 18.7 with one 128 KiB row per workgroup in LDS)
    bool use_bitmaps = true, use_bigmaps = true, filter = true, xcd_affinity = true, taper = true;
    bool bigmap_any_size = false;   // tests: the exact-set class whatever the genomes' sizes
    // registers in HBM, two-phase: scatter (idx, rho) records into per-(row, index tile) buckets, replay
    // each bucket into an LDS-resident tile (no global atomics); off = the filtered compare-and-swap path
    bool buckets = true;
    size_t bucket_e0_tiles = 0;     // tiles in the first epoch; 0 = four tokens per register (4 m / 65536), at least 8
    size_t bucket_emax_tiles = 0;   // longest epoch; 0 = what the budget below allows, at most 256 tiles
    size_t bucket_cap_chunks = 0;   // 64-record chunks per bucket; 0 = every token of the longest epoch fits
    int bucket_logg = 0;            // registers per filter entry (log2) PLUS ONE; 0 = default (a 64 KiB filter)
    int bucket_fbits = 0;           // bits per filter entry (8 or 4); 0 = 4
    int bucket_probe = -1;          // second-level filter against the row itself: 1 / 0; -1 = default
    size_t bucket_budget = (size_t)16 << 30;  // HBM for the record areas of one call
    size_t bucket_slots = 8192;     // scatter jobs per (class, epoch) aimed at
    size_t row_group_mb = 0;        // > 0: scatter -> replay per group of rows whose record areas sum to <= this (DD_ROW_GROUP_MB)
    static PlanKnobs from_env();
    bool operator==(const PlanKnobs& o) const {
        return lds_budget == o.lds_budget && lds_budget_forced == o.lds_budget_forced && jobs_per_cu == o.jobs_per_cu &&
               jobs_per_row == o.jobs_per_row && global_from_p == o.global_from_p && use_bitmaps == o.use_bitmaps && use_bigmaps == o.use_bigmaps && bigmap_any_size == o.bigmap_any_size &&
               filter == o.filter && xcd_affinity == o.xcd_affinity && taper == o.taper && buckets == o.buckets &&
               bucket_e0_tiles == o.bucket_e0_tiles && bucket_emax_tiles == o.bucket_emax_tiles &&
               bucket_cap_chunks == o.bucket_cap_chunks && bucket_logg == o.bucket_logg && bucket_fbits == o.bucket_fbits && bucket_probe == o.bucket_probe &&
               bucket_budget == o.bucket_budget && bucket_slots == o.bucket_slots && row_group_mb == o.row_group_mb;
    }
};

// the ks of a call that go to the big-bitmap class (false: none)
// (`nbytes`: the call's genomes -- an exact set only pays when a genome has several times more tokens than the
// set can have members, because every index tile's workgroup hashes the whole set afterwards)
bool plan_bigmap_range(int log2m, int kmin, int kmax, const PlanKnobs& knobs, const size_t* nbytes, int ngenomes, int* ka, int* kb);
std::vector<SweepClass> plan_sweep(int log2m, int canonical, const size_t* nbytes, int ngenomes, int kmin,
                                   int kmax, const PlanKnobs& knobs);

}  // namespace dd
