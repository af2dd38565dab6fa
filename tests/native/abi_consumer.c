/* A plain C99 consumer of include/dandd_hip.h: what a cgo / JNI / Rust-FFI binding sees.  Built by
 * tests/test_abi.py with -std=c99 -pedantic -Werror and linked against libdandd_hip.so; exercises the entry
 * points that need no GPU and the error path of the ones that do. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dandd_hip.h"

int main(void) {
    size_t sizes[2] = {5000000u, 70000u};
    long n, got;
    dd_plan_job *jobs;
    dd_ctx *ctx;
    if (dd_abi_version() != DD_ABI_VERSION) return 10;
    n = dd_plan_sweep(14, sizes, 2, 4, 40, NULL, 0);
    if (n <= 0) return 11;
    jobs = (dd_plan_job *)malloc((size_t)n * sizeof *jobs);
    if (!jobs) return 12;
    got = dd_plan_sweep(14, sizes, 2, 4, 40, jobs, n);
    if (got != n || jobs[0].tile_end <= jobs[0].tile_begin || jobs[0].kfirst < 4) return 13;
    free(jobs);
    if (dd_plan_sweep(3, sizes, 2, 4, 40, NULL, 0) >= 0) return 14; /* log2m below the supported range */
    if (!dd_last_error() || !strlen(dd_last_error())) return 15;
    if (dd_synth_size(1000, 2) == 0) return 16;
    ctx = dd_create(63, 14, 1); /* no such device anywhere: must fail loudly, never fall back */
    if (ctx) {
        dd_destroy(ctx);
        return 17;
    }
    if (!strlen(dd_last_error())) return 18;
    if (dd_last_k2_path(NULL) >= 0) return 19; /* ABI 3: a query on no context is an error code, not a path */
    { /* ABI 4: the multi-GPU entries exist and refuse a null context with an error code */
        int world = -1;
        if (dd_comm_info(NULL, NULL, &world, NULL, NULL) >= 0 || dd_allreduce_max_u8(NULL, NULL, 0) >= 0 || dd_allgather_u8(NULL, NULL, 0, NULL) >= 0 ||
            dd_comm_init(NULL, 0, 1, NULL) >= 0 || dd_comm_unique_id(NULL) >= 0)
            return 20;
        if (DD_COMM_ID_BYTES != 128) return 21;
    }
    printf("abi_consumer: ok (%ld jobs)\n", n);
    return 0;
}
