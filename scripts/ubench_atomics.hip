// ubench_atomics.hip -- throughput of scattered memory-side atomics on gfx950 (development aid for the
// log2m >= 18 K1 path).  Build: hipcc -O3 --offload-arch=gfx950 scripts/ubench_atomics.hip -o build/ubench_atomics
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int OP>
__global__ __launch_bounds__(1024) void k(uint32_t* buf, uint64_t mask, int iters, uint32_t* sink) {
    uint64_t s = (blockIdx.x * 1024ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const uint64_t idx = (s >> 20) & mask;
        const uint32_t v = (uint32_t)(s >> 58);
        if (OP == 0) atomicMax(&buf[idx], v);                       // no return
        else if (OP == 1) acc += atomicMax(&buf[idx], v);           // with return
        else if (OP == 2) acc += atomicCAS(&buf[idx], 0u, v);       // CAS with return
        else if (OP == 3) atomicOr(&buf[idx], 1u << (v & 31));      // OR no return
        else if (OP == 4) acc += __hip_atomic_load(&buf[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1 load
        else acc += buf[idx];                                       // plain load
    }
    if (acc == 0xDEADBEEF) sink[0] = acc;
}

template <int OP>
void run(const char* name, uint32_t* buf, uint64_t words, uint32_t* sink) {
    const int iters = 256, grid = 2048;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(1024), 0, 0, buf, words - 1, 16, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(1024), 0, 0, buf, words - 1, iters, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s %8.1f MiB target: %7.2f G ops/s\n", name, words * 4.0 / (1 << 20), (double)grid * 1024 * iters / ms / 1e6);
}

int main() {
    uint32_t *buf, *sink;
    const uint64_t maxw = 1ull << 28;  // 1 GiB
    if (hipMalloc(&buf, maxw * 4) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    hipMemset(buf, 0, maxw * 4);
    for (uint64_t words : {1ull << 18, 1ull << 22, 1ull << 26, 1ull << 28}) {
        run<0>("atomicMax no return", buf, words, sink);
        run<1>("atomicMax with return", buf, words, sink);
        run<2>("atomicCAS with return", buf, words, sink);
        run<3>("atomicOr no return", buf, words, sink);
        run<4>("agent-scope load (sc1)", buf, words, sink);
        run<5>("plain load", buf, words, sink);
    }
    return 0;
}
