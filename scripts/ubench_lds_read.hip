// What do LDS reads of 16 bytes per lane cost, and what does SQ_LDS_BANK_CONFLICT count for them?  (K2 pscan_kernel reads
// its bit planes with ds_read_b128 and shows 39-42 % "conflict cycles" whatever the chain-to-lane mapping: VERDICT r03 #6.)
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_lds_read.hip -o /tmp/ubench_lds_read && /tmp/ubench_lds_read
//   rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -- /tmp/ubench_lds_read
// MODE: 0 b128, lane i at 16 i (a wave reads 1 KiB in a row)      1 b128, lanes 144 bytes apart (pscan: rows of D + 4 words)
//       2 b128, pscan's addresses: chains of 28 thresholds per ordering, every ordering on its own leaf row
//       3 b64, lane i at 8 i      4 b32, lane i at 4 i      5 b128, lane i at 16 (i ^ (i >> 3 & 7))-style rotation inside groups of 8
//       6 two b64 halves of mode 0's 16 bytes (ds_read2_b64 would be the same)  7 b128 all lanes the same address (broadcast)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512) void k(int iters, uint32_t* out) {
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i * 2654435761u;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t base;
    if (MODE == 0 || MODE == 6) base = 16u * lane;
    else if (MODE == 1) base = 144u * lane;
    else if (MODE == 2) {
        const uint32_t c = threadIdx.x, o = c / 28u, t = c % 28u;       // 512 lanes: 18 orderings' worth of chains
        const uint32_t g = (o * 7u + 3u) % 11u;                           // the ordering's current leaf
        base = (g * 28u + t) * 144u;
    } else if (MODE == 3) base = 8u * lane;
    else if (MODE == 4) base = 4u * lane;
    else if (MODE == 5) base = 16u * ((lane & ~7u) | ((lane + (lane >> 3)) & 7u));
    else base = 0;
    base += wave * 64u;   // (waves start on different rows; 64 KiB image)
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t a = (base + 1024u * (uint32_t)(it & 7)) & 0xFFF0u;
        if (MODE == 3) {
            const u32x2 v = *((const __attribute__((address_space(3))) u32x2*)(uintptr_t)(a & ~7u));
            acc += v.x ^ v.y;
        } else if (MODE == 4) {
            acc += *((const __attribute__((address_space(3))) uint32_t*)(uintptr_t)a);
        } else if (MODE == 6) {
            const u32x2 v = *((const __attribute__((address_space(3))) u32x2*)(uintptr_t)a);
            const u32x2 w = *((const __attribute__((address_space(3))) u32x2*)(uintptr_t)(a + 8u));
            acc += v.x ^ v.y ^ w.x ^ w.y;
        } else {
            const u32x4 v = *((const __attribute__((address_space(3))) u32x4*)(uintptr_t)a);
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345u) out[0] = acc;
}

template <int MODE>
static void run(uint32_t* out, const char* what, int bytes) {
    const int iters = 4096, blocks = 512;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 65536, 0, 16, out);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 65536, 0, iters, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double reads = (double)blocks * 8 * iters * (MODE == 6 ? 2 : 1);   // wave-wide read instructions
    printf("mode %d %-58s %.3f ms  %.1f cycles per wave-read per CU (2.4 GHz)  %.0f B/clk/CU\n", MODE, what, ms, ms * 1e-3 * 2.4e9 / (reads / 256),
           (double)blocks * 8 * iters * 64 * bytes / 256 / (ms * 1e-3 * 2.4e9));
}

int main() {
    uint32_t* out;
    hipMalloc(&out, 256);
    run<0>(out, "b128, lanes 16 B apart", 16);
    run<1>(out, "b128, lanes 144 B apart (rows of 32 + 4 words)", 16);
    run<2>(out, "b128, pscan chains (28 thresholds x 18 orderings)", 16);
    run<3>(out, "b64, lanes 8 B apart", 8);
    run<4>(out, "b32, lanes 4 B apart", 4);
    run<5>(out, "b128, rotated inside groups of 8 lanes", 16);
    run<6>(out, "2 x b64 = the 16 bytes of mode 0", 16);
    run<7>(out, "b128, every lane the same address", 16);
    hipDeviceSynchronize();
    return 0;
}
