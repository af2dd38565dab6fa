// replay_probe.hip -- what a replay that is NOT "read the byte, compare, compare-and-swap the word" would cost (DESIGN.md section 8,
// "what comes next" (3)).  The shipped replay_kernel<true> applies every record of a bin to byte registers in LDS: one ds_read_b32,
// a compare, and for the records that raise their register a 32-bit CAS (+ retries when a neighbour's byte moved).  Candidates:
//   V 2: one 32-bit word per register, 16 384 registers per 64 KiB tile (the scatter would need 64 bins instead of 16), every
//        record ONE ds_max_u32 that returns nothing -- no read, no compare, no retry;
//   V 3: the same with 32 768 registers per tile (128 KiB of LDS: one workgroup per CU, 32 bins).
// against V 1 = the shipped inner loop (RegsLds, cas_raise: this file INCLUDES dandd_amd/csrc/dd_sweep.hip), V 4 = V 1 without
// the CAS (read + compare only: the floor of any byte-register form) and V 0 = the record loads alone.  Records are synthetic:
// uniform indices inside the tile, rho geometric (half of them from 2 up: the rho = 1 bits of half the registers), ~4 records per
// register as for 64 x 5 Mbp at log2m 20; a workgroup loads its tile's bytes, applies its records in pieces of 512 per wave (two
// 16-byte loads per lane, the next piece in flight: the product's shape) and stores the tile.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dandd_amd/csrc scripts/replay_probe.hip -o scripts/build/replay_probe
//   run  : scripts/build/replay_probe [records_in_millions=2400]
// Not part of the product; nothing here is linked into libdandd_hip.so.
#ifndef REPLAY_PROBE_NO_MAIN   // (scripts/scatter_trace.hip includes this file for its overlap experiment: the kernels only)
#include "../dandd_amd/csrc/dd_sweep.hip"
#endif

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <initializer_list>

namespace dd {
namespace {

__global__ void fill_records(uint32_t* __restrict__ recs, size_t n, uint32_t tile_mask, unsigned long long seed) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long x = seed + i * 0x9E3779B97F4A7C15ull;
    x ^= x >> 30, x *= 0xBF58476D1CE4E5B9ull, x ^= x >> 27, x *= 0x94D049BB133111EBull, x ^= x >> 31;
    const uint32_t idx = (uint32_t)x & tile_mask;
    const uint32_t geo = (uint32_t)__builtin_ctzll((x >> 24) | (1ull << 39));   // 0, 1, 2, ... with p = 1/2, 1/4, ...
    const uint32_t rho = ((idx & 1u) ? 2u : 1u) + geo;                          // (odd registers: their rho = 1 updates were bits)
    recs[i] = (rho << 24) | idx;
}

// V 0 loads only | V 1 shipped (bytes, read + compare + CAS) | V 2 words, ds_max_u32, 16 Ki registers | V 3 words, 32 Ki registers
// V 4 bytes, read + compare only
template <int V>
__global__ __launch_bounds__(1024) void replay_probe(const uint32_t* __restrict__ recs, uint32_t pieces_per_wave, uint8_t* __restrict__ regs,
                                                     uint32_t* __restrict__ sink) {
    lds_starts_at_zero();
    constexpr bool WORDS = V == 2 || V == 3;
    constexpr uint32_t tile = V == 2 ? 16384u : (V == 3 ? 32768u : 65536u);      // registers per workgroup
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    uint8_t* const tile_g = regs + (size_t)blockIdx.x * tile;
    // the tile's bytes in
    if (!WORDS) {
        uint4* l4 = reinterpret_cast<uint4*>(g_lds);
        for (uint32_t i = threadIdx.x; i < (tile >> 4); i += blockDim.x) l4[i] = gload16(tile_g + (size_t)i * 16);
    } else {
        for (uint32_t i = threadIdx.x; i < (tile >> 2); i += blockDim.x) {
            const uint32_t w = gload4(reinterpret_cast<const uint32_t*>(tile_g) + i);
            *reinterpret_cast<uint4*>(g_lds + 16u * i) = make_uint4(w & 0xFFu, (w >> 8) & 0xFFu, (w >> 16) & 0xFFu, w >> 24);
        }
    }
    __syncthreads();
    uint32_t acc = 0;
    constexpr int U = 4;
    auto apply_u = [&](const uint32_t (&e)[U]) {
        if (V == 0) {
            acc ^= e[0] ^ e[1] ^ e[2] ^ e[3];
        } else if (WORDS) {
#pragma unroll
            for (int i = 0; i < U; ++i)
                atomicMax(&lds32(4u * (e[i] & (tile - 1u))), e[i] >> 24);
        } else {
            uint32_t wd[U];
            uint32_t retry = 0;
#pragma unroll
            for (int i = 0; i < U; ++i) wd[i] = RegsLds::load32(e[i] & (tile - 1u));
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const uint32_t a = e[i] & (tile - 1u), rho = e[i] >> 24, sh = RegsLds::shift(a), cur = (wd[i] >> sh) & 0xFFu;
                if (V == 4) {
                    acc += rho > cur;
                    continue;
                }
                if (rho > cur) {
                    const uint32_t prev = RegsLds::cas32(a, wd[i], wd[i] + ((rho - cur) << sh));
                    if (prev != wd[i]) retry |= 1u << i;
                    wd[i] = prev;
                }
            }
            if (V == 1 && __any(retry != 0u)) {
#pragma unroll
                for (int i = 0; i < U; ++i)
                    if ((retry >> i) & 1u) (void)cas_raise<RegsLds>(e[i] & (tile - 1u), wd[i], e[i] >> 24);
            }
        }
    };
    const uint32_t* const mine = recs + ((size_t)blockIdx.x * 16u * pieces_per_wave) * 512u;
    auto piece = [&](uint32_t j, uint4& a, uint4& b) {   // piece j of this wave: 512 records, 8 per lane
        const uint32_t* base = mine + ((size_t)j * 16u + wave) * 512u;
        a = gload16(base + 4u * lane);
        b = gload16(base + 256u + 4u * lane);
    };
    uint4 ca, cb, na = make_uint4(0, 0, 0, 0), nb = na;
    piece(0, ca, cb);
    for (uint32_t j = 0; j < pieces_per_wave; ++j) {
        if (j + 1u < pieces_per_wave) piece(j + 1u, na, nb);
        const uint32_t ea[U] = {ca.x, ca.y, ca.z, ca.w}, eb[U] = {cb.x, cb.y, cb.z, cb.w};
        apply_u(ea);
        apply_u(eb);
        ca = na, cb = nb;
    }
    __syncthreads();
    // the tile's bytes out
    if (!WORDS) {
        const uint4* l4 = reinterpret_cast<const uint4*>(g_lds);
        for (uint32_t i = threadIdx.x; i < (tile >> 4); i += blockDim.x) gstore16(tile_g + (size_t)i * 16, l4[i]);
    } else {
        for (uint32_t i = threadIdx.x; i < (tile >> 2); i += blockDim.x) {
            const uint4 w = *reinterpret_cast<const uint4*>(g_lds + 16u * i);
            gstore4(tile_g + (size_t)i * 4, w.x | (w.y << 8) | (w.z << 16) | (w.w << 24));
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// how fast can ANY kernel read 9.6 GB?  (the ceiling the replay's record loads are held against)
// SHAPE 0: grid-stride, 16 bytes per lane per step, 4 steps in flight | 1: each workgroup its own contiguous MiB, 4 loads in flight
// per lane | 2: as 1 with 256-thread workgroups
template <int SHAPE>
__global__ __launch_bounds__(1024) void read_probe(const uint4* __restrict__ src, size_t n16, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    if (SHAPE == 0) {
        const size_t stride = (size_t)gridDim.x * blockDim.x;
        size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        for (; i + 3 * stride < n16; i += 4 * stride) {
            const uint4 a = gload16(src + i), b = gload16(src + i + stride), c = gload16(src + i + 2 * stride), d = gload16(src + i + 3 * stride);
            acc ^= a.x ^ a.w ^ b.x ^ b.w ^ c.x ^ c.w ^ d.x ^ d.w;
        }
    } else {
        const size_t per = (1u << 20) / 16;   // one MiB per workgroup
        const uint4* mine = src + (size_t)blockIdx.x * per;
        for (uint32_t i = threadIdx.x; i + 3u * blockDim.x < per; i += 4u * blockDim.x) {
            const uint4 a = gload16(mine + i), b = gload16(mine + i + blockDim.x), c = gload16(mine + i + 2u * blockDim.x), d = gload16(mine + i + 3u * blockDim.x);
            acc ^= a.x ^ a.w ^ b.x ^ b.w ^ c.x ^ c.w ^ d.x ^ d.w;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

}  // namespace
}  // namespace dd

#ifndef REPLAY_PROBE_NO_MAIN
#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

template <int V>
static int run(const char* what, uint32_t* recs, size_t nrec, uint8_t* regs, size_t regs_bytes, uint32_t* sink, unsigned long long* checksum) {
    constexpr uint32_t tile = V == 2 ? 16384u : (V == 3 ? 32768u : 65536u);
    constexpr size_t lds = V == 3 ? 131072 : 65536;
    const uint32_t recs_per_wg = tile * 4u;                            // ~4 records per register
    const uint32_t pieces_per_wave = recs_per_wg / (16u * 512u);
    const unsigned blocks = (unsigned)(nrec / recs_per_wg);
    if ((size_t)blocks * tile > regs_bytes) return 1;
    // records regenerated per variant: indices inside ITS tile
    hipLaunchKernelGGL(dd::fill_records, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, 0, recs, nrec, tile - 1u, 0xD4ADDull);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(dd::replay_probe<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemset(regs, 0, (size_t)blocks * tile));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(dd::replay_probe<V>, dim3(blocks), dim3(1024), lds, 0, recs, pieces_per_wave, regs, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    // a checksum of the registers: V 1, V 2 and V 3 must each equal the max over their records (checked among themselves by the
    // sum of all register bytes, which depends on the tile size only through the records' indices -- printed, compared by eye)
    unsigned long long sum = 0;
    {
        const size_t n = (size_t)blocks * tile;
        uint8_t* h = (uint8_t*)malloc(n);
        CK(hipMemcpy(h, regs, n, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) sum += h[i];
        free(h);
    }
    *checksum = sum;
    if (V == 1 || V == 2 || V == 3) {   // the first and the last workgroup's tile against the max over its records, on the host
        for (unsigned blk : {0u, blocks - 1u}) {
            uint32_t* hr = (uint32_t*)malloc((size_t)recs_per_wg * 4);
            uint8_t *hg = (uint8_t*)malloc(tile), *want = (uint8_t*)calloc(tile, 1);
            CK(hipMemcpy(hr, recs + (size_t)blk * recs_per_wg, (size_t)recs_per_wg * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hg, regs + (size_t)blk * tile, tile, hipMemcpyDeviceToHost));
            for (uint32_t i = 0; i < recs_per_wg; ++i) {
                const uint32_t a = hr[i] & (tile - 1u), rho = hr[i] >> 24;
                if (rho > want[a]) want[a] = (uint8_t)rho;
            }
            if (memcmp(hg, want, tile) != 0) {
                printf("V%d: workgroup %u's registers are NOT the max over its records\n", V, blk);
                return 1;
            }
            free(hr), free(hg), free(want);
        }
    }
    const double records = (double)blocks * recs_per_wg;
    printf("V%d %-78s %7.2f ms  %6.3f CU-cycles per record (256 CUs, 2.4 GHz)  %5.0f G records/s  register sum %llu\n", V, what, best,
           best * 1e-3 * 256 * 2.4e9 / records, records / best / 1e6, sum);
    return 0;
}

template <int SHAPE>
static int run_read(const char* what, const uint32_t* recs, size_t nrec, uint32_t* sink, unsigned threads, unsigned blocks_or_0) {
    const size_t n16 = nrec / 4;
    const unsigned blocks = blocks_or_0 ? blocks_or_0 : (unsigned)(n16 / ((1u << 20) / 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(dd::read_probe<SHAPE>, dim3(blocks), dim3(threads), 0, 0, reinterpret_cast<const uint4*>(recs), n16, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    printf("read %-86s %7.2f ms  %5.2f TB/s\n", what, best, nrec * 4.0 / best / 1e9);
    return 0;
}

int main(int argc, char** argv) {
    const size_t nrec = (size_t)(argc > 1 ? atof(argv[1]) : 2400.0) * 1000000ull / 262144ull * 262144ull;
    uint32_t *recs, *sink;
    uint8_t* regs;
    const size_t regs_bytes = nrec / 4 + (1 << 20);
    CK(hipMalloc(&recs, nrec * 4));
    CK(hipMalloc(&regs, regs_bytes));
    CK(hipMalloc(&sink, 256));
    printf("%.2f G records (%.1f GB), ~4 per register\n", nrec / 1e9, nrec * 4 / 1e9);
    unsigned long long s[5];
    int rc = 0;
    rc |= run<0>("record loads only (two 16-byte loads per lane per piece, next piece in flight)", recs, nrec, regs, regs_bytes, sink, &s[0]);
    rc |= run<4>("bytes, 65 536 per tile: ds_read_b32 + compare, no CAS (floor of the byte form)", recs, nrec, regs, regs_bytes, sink, &s[4]);
    rc |= run<1>("bytes, 65 536 per tile: read, compare, CAS, retries (the SHIPPED inner loop)", recs, nrec, regs, regs_bytes, sink, &s[1]);
    rc |= run<2>("words, 16 384 per tile (64 KiB, two workgroups per CU): one ds_max_u32, no return", recs, nrec, regs, regs_bytes, sink, &s[2]);
    rc |= run<3>("words, 32 768 per tile (128 KiB, ONE workgroup per CU): one ds_max_u32, no return", recs, nrec, regs, regs_bytes, sink, &s[3]);
    printf("-- the ceiling: plain reads of the same %.1f GB --\n", nrec * 4 / 1e9);
    rc |= run_read<0>("grid-stride, 2048 workgroups of 1024, four 16-byte loads in flight per lane", recs, nrec, sink, 1024, 2048);
    rc |= run_read<0>("grid-stride, 8192 workgroups of 256", recs, nrec, sink, 256, 8192);
    rc |= run_read<1>("one contiguous MiB per workgroup of 1024, four loads in flight per lane", recs, nrec, sink, 1024, 0);
    rc |= run_read<1>("one contiguous MiB per workgroup of 256", recs, nrec, sink, 256, 0);
    return rc;
}
#endif
