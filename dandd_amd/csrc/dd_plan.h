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
};

// What the environment may change about a plan (README.md lists every knob): capacities, and the switches the tests use to
// reach, on small inputs, the paths large inputs take (several epochs, a full record stream, the exact-set class).
struct PlanKnobs {
    size_t bucket_e0_tiles = 0;     // DD_BUCKET_E0: tiles in the first epoch; 0 = four tokens per register (4 m / 65536), at least 8
    size_t bucket_emax_tiles = 0;   // DD_BUCKET_EMAX: longest epoch; 0 = what the budget below allows, at most 256 tiles
    size_t bucket_cap_chunks = 0;   // DD_BUCKET_CAP: 1024-record chunks per row's stream; 0 = every token of the first epoch fits
    size_t bucket_budget = (size_t)16 << 30;  // DD_BUCKET_GB: HBM for the record areas of one call
    bool bigmap_any_size = false;   // DD_BIGMAP_ANY_SIZE (tests): the exact-set class whatever the genomes' sizes
    static PlanKnobs from_env();
    bool operator==(const PlanKnobs& o) const {
        return bucket_e0_tiles == o.bucket_e0_tiles && bucket_emax_tiles == o.bucket_emax_tiles && bucket_cap_chunks == o.bucket_cap_chunks &&
               bucket_budget == o.bucket_budget && bigmap_any_size == o.bigmap_any_size;
    }
};

// the ks of a call that go to the big-bitmap class (false: none)
// (`nbytes`: the call's genomes -- an exact set only pays when a genome has several times more tokens than the
// set can have members, because every index tile's workgroup hashes the whole set afterwards)
bool plan_bigmap_range(int log2m, int kmin, int kmax, const PlanKnobs& knobs, const size_t* nbytes, int ngenomes, int* ka, int* kb);
std::vector<SweepClass> plan_sweep(int log2m, int canonical, const size_t* nbytes, int ngenomes, int kmin,
                                   int kmax, const PlanKnobs& knobs);

}  // namespace dd
