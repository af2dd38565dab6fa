// LDS atomic cost against the number of active lanes (for the delta form of the progressive histogram):
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_lds_atomic.hip -o /tmp/ubench_lds_atomic && /tmp/ubench_lds_atomic
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

template <int KEEP>   // lanes with (lane % KEEP) == 0 take part
__global__ __launch_bounds__(1024) void k(const uint32_t* __restrict__ vals, int iters, uint32_t* out) {
    __shared__ uint32_t h[64][32];
    for (int i = threadIdx.x; i < 64 * 32; i += blockDim.x) (&h[0][0])[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, copy = threadIdx.x & 31;
    uint32_t x = vals[blockIdx.x * blockDim.x + threadIdx.x];
    if (lane % KEEP == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int b = 0; b < 4; ++b) atomicAdd(&h[(x >> (8 * b)) & 63][copy], 1u);
            x = x * 1664525u + 1013904223u;
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) atomicAdd(&out[threadIdx.x], h[threadIdx.x][0]);
}

template <int KEEP>
static void run(const uint32_t* vals, uint32_t* out) {
    const int iters = 2048, blocks = 512;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k<KEEP>, dim3(blocks), dim3(1024), 0, 0, vals, 16, out);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<KEEP>, dim3(blocks), dim3(1024), 0, 0, vals, iters, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double wave_instr = (double)blocks * 16 * iters * 4;   // atomic wave-instructions issued
    // 512 blocks over 256 CUs: 2 per CU, 32 waves per CU
    printf("1 lane in %2d active: %.3f ms, %.2f G wave-atomics/s, %.1f cycles per wave-atomic per CU (2.4 GHz), %.2f T lane-atomics/s\n", KEEP, ms,
           wave_instr / ms / 1e6, ms * 1e-3 * 2.4e9 / (wave_instr / 256), wave_instr * (64.0 / KEEP) / ms / 1e9);
}

int main() {
    const size_t n = 512 * 1024;
    std::vector<uint32_t> v(n);
    uint32_t s = 12345;
    for (auto& x : v) x = (s = s * 1664525u + 1013904223u) >> 3;
    uint32_t *d, *out;
    hipMalloc(&d, n * 4);
    hipMalloc(&out, 256);
    hipMemcpy(d, v.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(out, 0, 256);
    run<1>(d, out);
    run<2>(d, out);
    run<4>(d, out);
    run<8>(d, out);
    run<16>(d, out);
    run<64>(d, out);
    return 0;
}
