// ubench_k1.hip -- where do K1's cycles go?  (development aid; includes the product kernel source so
// the very same device functions are timed.)
// Build:  hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -Idandd_amd/csrc scripts/ubench_k1.hip -o build/ubench_k1
// Every variant runs the K1 inner loop on register-resident pseudo-random tokens (no HBM), two
// 1024-thread workgroups per CU (80 KiB LDS each), LDS registers pre-filled so that the
// "register rises" path is never (fill 255) or realistically (fill 0, long run) taken.
// Reported: SIMD cycles per wave-update (one (token, k) update of one wave).
#include "../dandd_amd/csrc/dd_sweep.hip"
#include <stdio.h>
#include <stdlib.h>

using namespace dd;

// MODE 0: full fast path (sweep_token, pair-interleaved)      MODE 1: hash only, no LDS access
// MODE 2: hash + probe (idx, lz) but no LDS read / branch     MODE 3: single (not paired) updates
template <int KC, int MODE>
__global__ __launch_bounds__(1024) void k1_model(uint32_t* out, int nk, int kfirst, int p, int iters,
                                                 uint32_t seed, uint32_t fill) {
    uint32_t* z = reinterpret_cast<uint32_t*>(g_lds);
    for (uint32_t i = threadIdx.x; i < ((uint32_t)nk << p) / 4; i += blockDim.x) z[i] = fill;
    __syncthreads();
    Windows<KC> win;
    uint32_t s = seed ^ ((blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u);
    uint64_t acc = 0;
    auto lds_slot = [](int j) { return RegsLds{(uint32_t)j}; };
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const uint32_t cw = s ^ (s >> 15);
#pragma unroll 1
        for (int i = 0; i < 16; ++i) {
            const uint32_t c = (cw >> (2 * i)) & 3u;
            win.push(c);
            if (MODE == 0) {
                sweep_token<KC, true, false>(win, 64, kfirst, nk, p, lds_slot);
            } else if (MODE == 5) {
                // experiment: wave-level pre-filter on the hash alone -- skip the LDS probe of a pair when no
                // lane's rho can exceed the smallest register of its array (here: smallest = 8)
                const uint32_t thr = (1u << (32 - 8)) - 1u;
#pragma unroll 1
                for (int j = 0; j < nk; j += 2) {
                    const uint64_t h0 = win.template hash<true>(kfirst + j), h1 = win.template hash<true>(kfirst + j + 1);
                    const uint32_t w0 = __builtin_amdgcn_alignbit((uint32_t)(h0 >> 32), (uint32_t)h0, 32 - p);
                    const uint32_t w1 = __builtin_amdgcn_alignbit((uint32_t)(h1 >> 32), (uint32_t)h1, 32 - p);
                    if (__any((w0 <= thr) | (w1 <= thr))) hll_update2(lds_slot(j), h0, lds_slot(j + 1), h1, p);
                }
            } else if (MODE == 4) {
                // experiment: k loop fully unrolled (nk == 4), so masks/shifts are loop-invariant scalars
                hll_update2(lds_slot(0), win.template hash<true>(kfirst), lds_slot(1), win.template hash<true>(kfirst + 1), p);
                hll_update2(lds_slot(2), win.template hash<true>(kfirst + 2), lds_slot(3), win.template hash<true>(kfirst + 3), p);
            } else if (MODE == 3) {
#pragma unroll 1
                for (int j = 0; j < nk; ++j) hll_update(lds_slot(j), win.template hash<true>(kfirst + j), p);
            } else {
#pragma unroll 1
                for (int j = 0; j < nk; ++j) {
                    const uint64_t h = win.template hash<true>(kfirst + j);
                    if (MODE == 1) acc ^= h;
                    else {
                        const Probe q = probe(h, p);
                        acc += q.hi ^ q.lz;
                    }
                }
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)acc ^ (uint32_t)(acc >> 32);
}

template <int KC, int MODE>
static double run(uint32_t* out, int nk, int kfirst, int p, int iters, uint32_t fill, int lds_kb, double ghz) {
    auto kern = k1_model<KC, MODE>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int wgs_per_cu = (160 / lds_kb);
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), (size_t)lds_kb * 1024, 0, out, nk, kfirst, p, iters / 8 + 1, 1u, fill);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), (size_t)lds_kb * 1024, 0, out, nk, kfirst, p, iters, 1u, fill);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: (16 * wgs_per_cu / 4) waves, each iters*16*nk updates
    const double waves_per_simd = 16.0 * wgs_per_cu / 4.0;
    const double upd = waves_per_simd * (double)iters * 16.0 * nk;
    return (double)ms * 1e-3 * ghz * 1e9 / upd;
}

int main(int argc, char** argv) {
    const double ghz = 2.34;
    const int p = 14;
    int iters = argc > 1 ? atoi(argv[1]) : 2000;
    uint32_t* out;
    if (hipMalloc(&out, 1024 * 1024 * sizeof(uint32_t)) != hipSuccess) return 1;
    printf("K1 inner-loop model, p=%d, cycles per wave-update per SIMD (2.34 GHz), canonical\n", p);
    printf("%-44s %8s %8s\n", "variant", "80KiB", "160KiB");
#define ROW(NAME, KC, MODE, NK, KF, FILL)                                                       \
    printf("%-44s %8.1f %8.1f\n", NAME, run<KC, MODE>(out, NK, KF, p, iters, FILL, 80, ghz),    \
           run<KC, MODE>(out, NK, KF, p, iters, FILL, 160, ghz));
    ROW("class0 k13..16 full, never raises", 0, 0, 4, 13, 0xFFFFFFFFu)
    ROW("class0 k13..16 full, from cold", 0, 0, 4, 13, 0u)
    ROW("class0 k13..16 unpaired, never raises", 0, 3, 4, 13, 0xFFFFFFFFu)
    ROW("class0 k13..16 hash+probe only", 0, 2, 4, 13, 0u)
    ROW("class0 k13..16 hash only", 0, 1, 4, 13, 0u)
    ROW("class1 k21..24 full, never raises", 1, 0, 4, 21, 0xFFFFFFFFu)
    ROW("class1 k21..24 full, k loop unrolled", 1, 4, 4, 21, 0xFFFFFFFFu)
    ROW("class1 k21..24 hash pre-filter (min reg 8)", 1, 5, 4, 21, 0xFFFFFFFFu)
    ROW("class0 k13..16 hash pre-filter (min reg 8)", 0, 5, 4, 13, 0xFFFFFFFFu)
    ROW("class0 k13..16 full, k loop unrolled", 0, 4, 4, 13, 0xFFFFFFFFu)
    ROW("class3 k33..36 full, k loop unrolled", 3, 4, 4, 33, 0xFFFFFFFFu)
    ROW("class1 k21..24 full, from cold", 1, 0, 4, 21, 0u)
    ROW("class1 k21..24 unpaired, never raises", 1, 3, 4, 21, 0xFFFFFFFFu)
    ROW("class1 k21..24 hash+probe only", 1, 2, 4, 21, 0u)
    ROW("class1 k21..24 hash only", 1, 1, 4, 21, 0u)
    ROW("class2 k49..52 full, from cold", 2, 0, 4, 49, 0u)
    ROW("class2 k49..52 hash only", 2, 1, 4, 49, 0u)
    ROW("class3 k33..36 full, never raises", 3, 0, 4, 33, 0xFFFFFFFFu)
    ROW("class3 k33..36 hash only", 3, 1, 4, 33, 0u)
    ROW("class3 k45..48 full, never raises", 3, 0, 4, 45, 0xFFFFFFFFu)
    ROW("class2 k49..52 full, never raises", 2, 0, 4, 49, 0xFFFFFFFFu)
    ROW("class2 k61..64 full, never raises", 2, 0, 4, 61, 0xFFFFFFFFu)
    hipFree(out);
    return 0;
}
