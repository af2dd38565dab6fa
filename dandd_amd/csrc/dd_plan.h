// dd_plan.h -- the K1 job tables: which workgroup sketches which (genome, k-group, tile range).
// Pure host code, no HIP calls: dd_sketch_device (dd_api.hip) uploads the tables, tests inspect them
// through dd_plan_sweep (include/dandd_hip.h) without a GPU.
#pragma once
#include <stddef.h>

#include <vector>

#include "dd_kernels.h"

namespace dd {

constexpr int kBitmapClass = -1;  // SweepClass::kclass of the small-k presence-bitmap class

struct SweepClass {
    int kclass;      // kBitmapClass, or the window class of sweep_kernel: 0 (k <= 16), 1 (<= 32), 3 (33..48), 2 (49..64)
    int kfirst, klast;
    SweepPlan plan;  // launch shape of the class (mode, LDS bytes, threads)
    std::vector<SweepJob> jobs;
};

// development knobs, read from the environment by from_env() (README.md lists them)
struct PlanKnobs {
    size_t lds_budget = 80 * 1024;  // per workgroup: two 1024-thread workgroups (8 waves/SIMD) per CU
    bool lds_budget_forced = false;
    size_t jobs_per_cu = 32;
    size_t jobs_per_row = 0;        // filtered mode; 0 = heuristic
    int global_from_p = 18;         // registers stay in HBM from this log2m on
    bool use_bitmaps = true, filter = true, xcd_affinity = true, taper = true;
    static PlanKnobs from_env();
    bool operator==(const PlanKnobs& o) const {
        return lds_budget == o.lds_budget && lds_budget_forced == o.lds_budget_forced && jobs_per_cu == o.jobs_per_cu &&
               jobs_per_row == o.jobs_per_row && global_from_p == o.global_from_p && use_bitmaps == o.use_bitmaps &&
               filter == o.filter && xcd_affinity == o.xcd_affinity && taper == o.taper;
    }
};

std::vector<SweepClass> plan_sweep(int log2m, int canonical, const size_t* nbytes, int ngenomes, int kmin,
                                   int kmax, const PlanKnobs& knobs);

}  // namespace dd
