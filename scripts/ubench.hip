// ubench.hip -- gfx950 VALU / LDS instruction issue-cost microbenchmark (development aid for K1).
// Build:  hipcc -O2 --offload-arch=gfx950 scripts/ubench.hip -o gpurun_out/ubench
// Each kernel runs ITERS iterations of 16 independent copies of one instruction (or short
// sequence) per wave; reported: ns per wave-instruction per SIMD at 1, 2 and 4 waves per SIMD,
// and the cost relative to v_xor_b32 at the same occupancy.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ITERS 32768

#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

// 32-bit ops: a[i] = op(a[i], b)
#define K32(NAME, ASM)                                                                     \
    __global__ void NAME(uint32_t* out, uint32_t seed) {                                   \
        uint32_t a[16], b = seed ^ threadIdx.x, c = seed * 3 + 1;                          \
        for (int i = 0; i < 16; ++i) a[i] = seed + i * 77 + threadIdx.x;                   \
        for (int it = 0; it < ITERS; ++it) {                                               \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); \
        }                                                                                  \
        uint32_t r = 0;                                                                    \
        for (int i = 0; i < 16; ++i) r ^= a[i];                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                    \
    }

// 64-bit ops: a[i] (pair) = op(a[i], b)
#define K64(NAME, ASM)                                                                     \
    __global__ void NAME(uint32_t* out, uint32_t seed) {                                   \
        uint64_t a[16], b = ((uint64_t)seed << 20) ^ threadIdx.x;                          \
        uint32_t c = seed | 1;                                                             \
        for (int i = 0; i < 16; ++i) a[i] = (uint64_t)seed * (i + 3) + threadIdx.x;        \
        for (int it = 0; it < ITERS; ++it) {                                               \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); \
        }                                                                                  \
        uint64_t r = 0;                                                                    \
        for (int i = 0; i < 16; ++i) r ^= a[i];                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(r ^ (r >> 32));            \
    }

K32(k_xor, "v_xor_b32 %0, %0, %1")
K32(k_add, "v_add_u32 %0, %0, %1")
K32(k_lshl, "v_lshlrev_b32 %0, 3, %0")
K32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 11")
K32(k_lshl_or, "v_lshl_or_b32 %0, %0, 2, %1")
K32(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
K32(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
K32(k_xad, "v_xad_u32 %0, %0, %1, %2")
K32(k_add3, "v_add3_u32 %0, %0, %1, %2")
K32(k_bfe, "v_bfe_u32 %0, %0, 3, 9")
K32(k_bfi, "v_bfi_b32 %0, %1, %0, %2")
K32(k_perm, "v_perm_b32 %0, %0, %1, %2")
K32(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
K32(k_mul_hi, "v_mul_hi_u32 %0, %0, %1")
K32(k_mul_u24, "v_mul_u32_u24 %0, %0, %1")
K32(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %2")
K32(k_ffbh, "v_ffbh_u32 %0, %0")
K32(k_min, "v_min_u32 %0, %0, %1")
K32(k_min3, "v_min3_u32 %0, %0, %1, %2")
K32(k_not, "v_not_b32 %0, %0")
K32(k_addco_pair, "v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %2, vcc")
K32(k_cmp_cnd, "v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc")
K32(k_sub_co_pair, "v_sub_co_u32 %0, vcc, %0, %1\n\tv_subb_co_u32 %0, vcc, %0, %2, vcc")
K32(k_and, "v_and_b32 %0, %0, %1")
K32(k_or, "v_or_b32 %0, %0, %1")
K32(k_mov, "v_mov_b32 %0, %1")
K32(k_sub, "v_sub_u32 %0, %0, %1")
K32(k_lshr, "v_lshrrev_b32 %0, 3, %0")
K32(k_cnd, "v_cndmask_b32 %0, %0, %1, vcc")
K32(k_cmp, "v_cmp_lt_u32 vcc, %0, %1")
K32(k_fma, "v_fma_f32 %0, %0, %1, %2")
K32(k_xor_dep, "v_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %2")
K64(k_lshl64, "v_lshlrev_b64 %0, 3, %0")
K64(k_lshr64, "v_lshrrev_b64 %0, 24, %0")
K64(k_lshladd64, "v_lshl_add_u64 %0, %0, 3, %1")
K64(k_lshladd64_0, "v_lshl_add_u64 %0, %0, 0, %1")
K64(k_mad64, "v_mad_u64_u32 %0, vcc, %2, %2, %0")
K64(k_cmp64, "v_cmp_lt_u64 vcc, %0, %1")

// LDS: random byte read / u32 max / cmpswap against a 64 KiB table
#define KLDS(NAME, BODY)                                                                     \
    __global__ void NAME(uint32_t* out, uint32_t seed) {                                     \
        __shared__ uint32_t tab[16384];                                                      \
        for (int i = threadIdx.x; i < 16384; i += blockDim.x) tab[i] = i * 2654435761u;      \
        __syncthreads();                                                                     \
        uint32_t x = seed + threadIdx.x * 2654435761u + blockIdx.x, acc = 0;                 \
        for (int it = 0; it < ITERS; ++it) {                                                 \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                 \
                x = x * 1664525u + 1013904223u;                                              \
                BODY                                                                         \
            }                                                                                \
        }                                                                                    \
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc + tab[threadIdx.x];                 \
    }
KLDS(k_lds_lcg_only, acc ^= x;)
KLDS(k_lds_read_u8, acc += reinterpret_cast<volatile uint8_t*>(tab)[x >> 16];)
KLDS(k_lds_read_b32, acc += reinterpret_cast<volatile uint32_t*>(tab)[x >> 18];)
KLDS(k_lds_max_u32, atomicMax(&tab[x >> 18], x & 63u);)
KLDS(k_lds_max_u32_rtn, acc += atomicMax(&tab[x >> 18], x & 63u);)
KLDS(k_lds_cas, acc += atomicCAS(&tab[x >> 18], x, x + 1);)

__global__ void k_clock(unsigned long long* out) {
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    uint32_t a = threadIdx.x;
    for (int it = 0; it < 4000000; ++it) asm volatile("v_xor_b32 %0, %0, %0\n\tv_add_u32 %0, 1, %0" : "+v"(a));
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = a; }
}

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Entry {
    const char* name;
    kern_t k;
    int instr_per_rep;  // instructions counted per asm statement
};

int main(int argc, char** argv) {
    Entry tests[] = {
        {"v_xor_b32", k_xor, 1}, {"v_add_u32", k_add, 1}, {"v_lshlrev_b32", k_lshl, 1},
        {"v_alignbit_b32", k_alignbit, 1}, {"v_lshl_or_b32", k_lshl_or, 1}, {"v_and_or_b32", k_and_or, 1},
        {"v_lshl_add_u32", k_lshl_add, 1}, {"v_xad_u32", k_xad, 1}, {"v_add3_u32", k_add3, 1},
        {"v_bfe_u32", k_bfe, 1}, {"v_bfi_b32", k_bfi, 1}, {"v_perm_b32", k_perm, 1},
        {"v_mul_lo_u32", k_mul_lo, 1}, {"v_mul_hi_u32", k_mul_hi, 1}, {"v_mul_u32_u24", k_mul_u24, 1},
        {"v_mad_u32_u24", k_mad_u24, 1}, {"v_ffbh_u32", k_ffbh, 1}, {"v_min_u32", k_min, 1},
        {"v_min3_u32", k_min3, 1}, {"v_not_b32", k_not, 1},
        {"add_co+addc (pair)", k_addco_pair, 1}, {"sub_co+subb (pair)", k_sub_co_pair, 1},
        {"cmp_lt_u32+cndmask (pair)", k_cmp_cnd, 1},
        {"v_and_b32", k_and, 1}, {"v_or_b32", k_or, 1}, {"v_mov_b32", k_mov, 1}, {"v_sub_u32", k_sub, 1},
        {"v_lshrrev_b32", k_lshr, 1}, {"v_cndmask_b32", k_cnd, 1}, {"v_cmp_lt_u32", k_cmp, 1}, {"v_fma_f32", k_fma, 1},
        {"2x v_xor dependent", k_xor_dep, 1},
        {"v_lshlrev_b64", k_lshl64, 1}, {"v_lshrrev_b64", k_lshr64, 1}, {"v_lshl_add_u64 sh3", k_lshladd64, 1},
        {"v_lshl_add_u64 sh0", k_lshladd64_0, 1}, {"v_mad_u64_u32", k_mad64, 1},
        {"v_cmp_lt_u64", k_cmp64, 1},
        {"lcg only (2 valu)", k_lds_lcg_only, 1}, {"lcg + ds_read_u8 rand", k_lds_read_u8, 1},
        {"lcg + ds_read_b32 rand", k_lds_read_b32, 1}, {"lcg + ds_max_u32", k_lds_max_u32, 1},
        {"lcg + ds_max_rtn_u32", k_lds_max_u32_rtn, 1}, {"lcg + ds_cmpst_rtn", k_lds_cas, 1},
    };
    const int nt = sizeof(tests) / sizeof(tests[0]);
    uint32_t* out;
    if (hipMalloc(&out, 256 * 16 * 1024 * sizeof(uint32_t)) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    {
        unsigned long long* cl; hipMalloc(&cl, 64);
        for (int rep = 0; rep < 3; ++rep) {
            k_clock<<<256 * 4, 256>>>(cl);
            unsigned long long h[3]; hipMemcpy(h, cl, 24, hipMemcpyDeviceToHost);
            printf("clock probe: %llu shader cycles in %llu x 10 ns -> %.3f GHz\n", h[0], h[1], (double)h[0] / ((double)h[1] * 10.0));
        }
    }
    printf("%-28s %10s %10s %10s   (ns per wave-instr per SIMD; x = relative to v_xor_b32)\n", "instr", "1w/SIMD", "2w/SIMD", "4w/SIMD");
    double base[3] = {0, 0, 0};
    for (int t = 0; t < nt; ++t) {
        double ns[3];
        for (int wi = 0; wi < 3; ++wi) {
            const int wps = 1 << wi;                 // waves per SIMD
            dim3 block(256 * wps), grid(256);        // one block per CU: 4*wps waves
            tests[t].k<<<grid, block>>>(out, 12345u);  // warm
            hipDeviceSynchronize();
            hipEventRecord(e0);
            tests[t].k<<<grid, block>>>(out, 12345u);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            // per SIMD: wps waves x ITERS x 16 statements
            ns[wi] = (double)ms * 1e6 / ((double)wps * ITERS * 16);
        }
        if (t == 0) memcpy(base, ns, sizeof ns);
        printf("%-28s %7.3f ns %7.3f ns %7.3f ns   x%.2f x%.2f x%.2f\n", tests[t].name, ns[0], ns[1], ns[2],
               ns[0] / base[0], ns[1] / base[1], ns[2] / base[2]);
    }
    hipFree(out);
    return 0;
}
