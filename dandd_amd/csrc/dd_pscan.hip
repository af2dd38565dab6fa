// dd_pscan.hip -- K2 progressive unions as a bit-plane AND-scan.
//
// Replaces, for `dandd progressive` (DeltaTree.sketch_ordering / progressive_union,
// /root/reference/lib/huffman_dandd.py:624-663), the Σj `dashing union` + `dashing card` processes per ordering and k.
// What is needed per (ordering o, prefix j, k) is the 64-bin histogram of U_j = max(leaf[o_0], .., leaf[o_j]).
//
// The streaming kernel (dd_union.hip: progressive_kernel) keeps U_j in registers and pays one LDS atomic per register
// per prefix -- 4.7 cycles per wave-wide ds_add_u32 whatever the bytes are (scripts/ubench_lds_atomic.hip): a 1.4 ms
// floor for 10 orderings x 30 prefixes x 37 k of 2^20 registers, 3.15 ms measured (1.8 ms through this file).  Here the
// cumulative histogram is
// counted instead: U_j[r] <= v  <=>  every leaf of the prefix has leaf[r] <= v, so with the bit planes
//         B_g,v = { r : leaf_g[r] <= v }          (one bit per register)
// F_o,j(v) = popcount(B_o0,v & B_o1,v & .. & B_oj,v): a running AND along the ordering and one popcount per prefix --
// 32 registers per instruction instead of one -- and hist_j(v) = F_j(v) - F_j(v-1).  Only thresholds between the
// smallest and the largest register of the k column are needed (gram_range_kernel of dd_gram.hip).
//
// One workgroup per (k, register range); per tile of 32 D registers:
//   convert   every thread takes (leaf g, 32 registers): the 32 bytes are bit-sliced into six planes (two
//             instructions per byte-dword and bit), and each needed threshold's plane is eq(vmin) | .. | eq(v), five
//             ANDs of planes or their complements per threshold; planes go to LDS as [g][threshold][d] (rows of D + 4
//             words: a lane's 16-byte reads and its neighbours' then fall on different banks; rows of D words with their
//             16-byte chunks rotated by the threshold measured 67 % conflict cycles against 39 %)
//   scan      one LANE per chain (ordering, threshold): for each of the D plane words P = ~0, then for every prefix
//             P &= plane[o_j][d][t], count_j += popcount(P) -- the counts stay in the lane's registers across all
//             tiles of the range, so nothing is reduced until the very end.
// The conversion is shared by every ordering and threshold of the workgroup, which is what makes it affordable.
// Exact integers throughout; tests/test_gpu_parity.py checks every cardinality against the streaming kernel's.
#include "dd_common.h"
#include "dd_kernels.h"

#include <stdlib.h>

#include <algorithm>

namespace dd {
namespace {

constexpr int PS_THREADS = 512;

// 16 bytes at an absolute LDS byte address (ds_read_b128 with `words` as its immediate offset)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
DD_D uint4 lds_read16(uint32_t byte_addr, int words) {
    const u32x4 v = *((const __attribute__((address_space(3))) u32x4*)(uintptr_t)byte_addr + words / 4);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// the six bit planes of 32 registers (8 dwords of 4 bytes): bit i + 8 q of plane b = bit b of byte q of dword i
DD_D void bit_slice(const uint32_t (&w)[8], uint32_t (&pl)[6]) {
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t s = i >= b ? w[i] << (i - b) : w[i] >> (b - i);
            acc |= s & (0x01010101u << i);
        }
        pl[b] = acc;
    }
}

// NMAX: prefixes held in registers (n <= NMAX); UPT: (leaf, 32 registers) units a thread converts per tile
// At most 128 registers (the next prefix's reads in flight, the ordering's leaves as bytes, the running store pointer):
// two workgroups share a CU and hide each other's barriers and LDS latencies, and the host picks tiles small enough
// for two workgroups' planes.  (One workgroup of 237 registers per CU with the next eight reads in flight during the
// fold of the current eight measured 2.43 ms where the d-major two-workgroup form took 2.11 and this one takes 1.82.)
// DG: 16-byte groups of plane words a lane folds per prefix before it moves to the next prefix (one row address per DG
// reads).
// (Converter waves running a tile ahead of the scan on a second set of planes were built and measured no better --
// profiles/r04_k2_pscan_lds.txt; the kernel converts, then scans, a tile at a time.)
template <int NMAX, int UPT, int DG>
__global__ __launch_bounds__(PS_THREADS, 4) void pscan_kernel(const uint8_t* __restrict__ leaf, int n, int K, int p,
                                                           const int32_t* __restrict__ ord, int no, const uint32_t* __restrict__ rng,
                                                           int RR, int tiles_per_range, int D, int chain_pitch,
                                                           uint32_t* __restrict__ part) {
    extern __shared__ uint32_t lds[];            // planes [g][t][DP], then the orderings [no][n] as bytes
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)lds != 0u) __builtin_trap();  // (the scan reads at absolute LDS addresses)
    const int k = blockIdx.x / RR, rr = blockIdx.x % RR;
    const int vmin = (int)rng[2 * k], vmax = (int)rng[2 * k + 1];
    const int T = vmax - vmin;                   // thresholds vmin .. vmax-1
    if (T <= 0) return;                          // every register of the column equal: F is 0 below it, m from it on
    const int DP = D + 4;                        // words per (leaf, threshold) row: 16 consecutive rows start on 16 different banks
    uint32_t* planes = lds;
    const uint32_t set_words = (uint32_t)n * (uint32_t)T * (uint32_t)DP;      // one set of planes
    uint8_t* ord_s = reinterpret_cast<uint8_t*>(lds + (size_t)set_words);
    for (int i = threadIdx.x; i < no * n; i += PS_THREADS) ord_s[i] = (uint8_t)ord[i];
    __syncthreads();
    // this lane's chain
    const int c = threadIdx.x;
    const bool chain = c < no * T;
    const int o = chain ? c / T : 0, t = chain ? c % T : 0;
    uint32_t cnt[NMAX], gp[NMAX / 4];            // the ordering's leaves as bytes, four to a register
    const uint32_t tDP = (uint32_t)(t * DP), rowDP = (uint32_t)(T * DP);
#pragma unroll
    for (int j = 0; j < NMAX / 4; ++j) {
        gp[j] = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) gp[j] |= (uint32_t)ord_s[o * n + (4 * j + i < n ? 4 * j + i : n - 1)] << (8 * i);
    }
#pragma unroll
    for (int j = 0; j < NMAX; ++j) cnt[j] = 0;
    const int units = n * D;                     // (leaf, 32 registers) pairs of a tile
    const size_t tile_regs = (size_t)32 * D;
    // ranges of tiles_total / RR tiles, the remainder spread over the first ranges
    const int tiles_total = tiles_per_range;     // (argument reused: all tiles of a row)
    const int tile0 = (int)(((long long)tiles_total * rr) / RR), ntiles = (int)(((long long)tiles_total * (rr + 1)) / RR) - tile0;
    const size_t reg0 = (size_t)tile0 * tile_regs;
    // every thread converts: thread x takes units x, x + PS_THREADS, ..
    constexpr int nconv = PS_THREADS;
    const int cx = (int)threadIdx.x;
    uint4 cur[UPT][2];
    auto load1 = [&](int q, int tile) {   // unit q of this thread, of `tile`
        // (threads past the last unit load the last unit again: with every load unconditional the compiler can count
        // them, and waits for the current tile's bytes with the next tile's loads still in flight)
        int u = cx + q * nconv;
        u = u < units ? u : units - 1;
        const int g = u / D, d = u % D;
        const uint8_t* src = leaf + (((size_t)g * K + k) << p) + reg0 + (size_t)tile * tile_regs + (size_t)d * 32;
        cur[q][0] = gload16(src);
        cur[q][1] = gload16(src + 16);
    };
    // ---- convert: bytes -> threshold planes of `tile`
    auto convert = [&](int tile) {
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
            const int u = cx + q * nconv;
            if (u < units) {
                const uint32_t w[8] = {cur[q][0].x, cur[q][0].y, cur[q][0].z, cur[q][0].w, cur[q][1].x, cur[q][1].y, cur[q][1].z, cur[q][1].w};
                uint32_t x[6], nx[6];
                bit_slice(w, x);
                load1(q, tile + 1 < ntiles ? tile + 1 : tile);   // (the bytes are sliced: their registers take those of the tile this thread converts next)
#pragma unroll
                for (int b = 0; b < 6; ++b) nx[b] = ~x[b];
                uint32_t* dst = planes + (size_t)(u / D) * T * DP + (u % D);   // [g][.][d]
                uint32_t le = 0;
#pragma unroll
                for (int v = 0; v < 64; ++v) {
                    if (v >= vmin && v < vmax) {             // (wave-uniform; the bits of v are compile-time constants)
                        uint32_t eq = (v & 1) ? x[0] : nx[0];
#pragma unroll
                        for (int b = 1; b < 6; ++b) eq &= ((v >> b) & 1) ? x[b] : nx[b];
                        le |= eq;
                        *dst = le;
                        dst += DP;           // (a running pointer: 64 precomputed row addresses would be hoisted out of the tile loop)
                    }
                }
            }
        }
    };
#pragma unroll
    for (int q = 0; q < UPT; ++q) load1(q, 0);
    for (int tile = 0; tile < ntiles; ++tile) {
        convert(tile);
        __syncthreads();
        // ---- scan: running AND along the ordering, one popcount per prefix
        // Prefix-major over chunks of DG 16-byte groups: a prefix's row address is computed once per chunk and its DG
        // reads carry their word offsets as immediates (the d-major form paid the byte extraction, a 64-bit multiply-add
        // and a move per READ: 14 instructions per (prefix, four words) where this takes 9 + 3 / DG); the next prefix's
        // reads are issued before the current one's words are folded; every count is one chain of v_bcnt accumulations.
        if (chain) {
            for (int d0 = 0; d0 < D; d0 += 4 * DG) {
                uint32_t P[4 * DG];
#pragma unroll
                for (int i = 0; i < 4 * DG; ++i) P[i] = ~0u;
                // (rows are read at their absolute LDS byte address -- the kernel's only LDS is the dynamic array, which
                // then starts at 0, checked at the top -- so a row's address is ONE 24-bit multiply-add, full rate, where
                // pointer arithmetic took a quarter-rate v_mul_lo_u32, a shift and an add)
                const uint32_t at0 = 4u * (tDP + (uint32_t)d0), row_bytes = 4u * rowDP;
                auto row = [&](int j) {
                    uint32_t g = (gp[j >> 2] >> (8 * (j & 3))) & 0xffu;
                    asm volatile("" : "+v"(g));   // (keeps the 32 row offsets from being hoisted into 32 registers again)
                    return __umul24(g, row_bytes) + at0;
                };
                constexpr bool PF = true;
                uint4 nxt[DG];
                if (PF) {
                    const uint32_t r0 = row(0);
#pragma unroll
                    for (int q = 0; q < DG; ++q) nxt[q] = lds_read16(r0, 4 * q);
                }
#pragma unroll
                for (int j = 0; j < NMAX; ++j) {
                    uint4 xs[DG];
                    if (PF) {
#pragma unroll
                        for (int q = 0; q < DG; ++q) xs[q] = nxt[q];
                        if (j + 1 < NMAX) {
                            const uint32_t r1 = row(j + 1);
#pragma unroll
                            for (int q = 0; q < DG; ++q) nxt[q] = lds_read16(r1, 4 * q);
                        }
                    } else {
                        const uint32_t r0 = row(j);
#pragma unroll
                        for (int q = 0; q < DG; ++q) xs[q] = lds_read16(r0, 4 * q);
                    }
                    uint32_t cj = cnt[j];
#pragma unroll
                    for (int q = 0; q < DG; ++q) {
                        P[4 * q] &= xs[q].x, P[4 * q + 1] &= xs[q].y, P[4 * q + 2] &= xs[q].z, P[4 * q + 3] &= xs[q].w;
                        // (v_bcnt_u32_b32 adds its count to an accumulator operand; written as C the compiler turns the
                        // chain into a tree of counts-from-zero and v_add3s, half an instruction more per word)
#pragma unroll
                        for (int i = 0; i < 4; ++i) asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(cj) : "v"(P[4 * q + i]));
                    }
                    cnt[j] = cj;
                    __builtin_amdgcn_sched_barrier(0);   // (left alone the scheduler gathers the reads of many prefixes and spills)
                }
            }
        }
        __syncthreads();
    }
    if (chain) {
        uint32_t* out = part + ((size_t)k * RR + rr) * (size_t)n * chain_pitch + c;
#pragma unroll
        for (int j = 0; j < NMAX; ++j)
            if (j < n) gstore4(out + (size_t)j * chain_pitch, cnt[j]);
    }
}

// one wave per (ordering, prefix, k): lane v sums threshold v's partial counts over the register ranges and the wave
// differences F into hist[((o * n) + j) * K + k][64]
__global__ __launch_bounds__(256) void pscan_finish_kernel(const uint32_t* __restrict__ part, int n, int K, int p, int no,
                                                           const uint32_t* __restrict__ rng, int RR, int chain_pitch,
                                                           uint32_t* __restrict__ hist) {
    const int lane = threadIdx.x & 63;
    const size_t job = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (job >= (size_t)no * n * K) return;
    const int k = (int)(job % (size_t)K);
    const int j = (int)((job / (size_t)K) % (size_t)n);
    const int o = (int)(job / (size_t)K / (size_t)n);
    const int vmin = (int)rng[2 * k], vmax = (int)rng[2 * k + 1];
    const int T = vmax - vmin;
    const uint32_t m = 1u << p;
    uint32_t F = lane < vmin ? 0u : m;
    if (lane >= vmin && lane < vmax) {
        F = 0;
        const uint32_t* src = part + ((size_t)k * RR * n + j) * (size_t)chain_pitch + (size_t)(o * T + (lane - vmin));
        for (int rr = 0; rr < RR; ++rr) F += gload4(src + (size_t)rr * n * chain_pitch);
    }
    const uint32_t prev = __shfl_up(F, 1);
    gstore4(hist + (((size_t)o * n + j) * K + k) * 64 + lane, lane ? F - prev : F);
}

}  // namespace

// plane words per tile row: the largest power of two for which the planes of a tile (n leaves x D words x T thresholds)
// fit 128 KiB of LDS and its (leaf, word) units fit four per thread
static int pscan_words(int n, int T, size_t lds_limit) {
    int D = 32;
    while (D > 4 && ((size_t)n * (D + 4) * T * 4 > lds_limit || n * D > PS_THREADS)) D >>= 1;   // one (leaf, word) unit per thread
    return D;
}

// n <= 32: a lane keeps one count per prefix in registers; longer orderings take the streaming kernel.  log2m >= 18:
// below that the streaming kernel is the faster one (10 orderings x 30 prefixes x 37 k, same box: 0.10 / 0.22 / 0.34 ms
// against 0.17 / 0.21 / 0.35 at log2m 15 / 16 / 17; 0.70 / 1.57 / 3.15 against 0.61 / 1.07 / 2.11 at 18 / 19 / 20).
bool pscan_usable(int n, int no, int p) { return p >= 18 && n >= 2 && n <= 32 && no >= 1; }

// scratch: the range pairs of every k, then the partial counts [k][range][prefix][chain]
size_t pscan_scratch_bytes(int n, int K, int p, int no) {
    const size_t pitch = PS_THREADS;
    (void)no;
    return (((size_t)K * 2 * sizeof(uint32_t) + 255) & ~(size_t)255) + (size_t)K * 64 * (size_t)n * pitch * sizeof(uint32_t);
}

// rng_host: the (min, max) pairs gram_range_kernel left at the start of `scratch`, read back by the caller (K pairs).
// Orderings are taken as many at a time as have their chains in the 512 lanes of a workgroup.  hist_dev is written
// in full.  Returns false when a column's thresholds do not fit (the caller falls back to the streaming kernel).
bool launch_progressive_pscan(const uint8_t* leaf_dev, int n, int K, int p, const int32_t* ord_dev, int norder, const uint32_t* rng_host,
                              void* scratch, uint32_t* hist_dev, hipStream_t st) {
    int Tmax = 1;
    for (int k = 0; k < K; ++k) Tmax = std::max(Tmax, (int)rng_host[2 * k + 1] - (int)rng_host[2 * k]);
    // as many orderings per launch as have their chains in one workgroup's lanes, shared out evenly over the launches
    const int fit = PS_THREADS / Tmax;
    if (fit < 1) return false;
    const int launches = (norder + fit - 1) / fit;
    const int group = (norder + launches - 1) / launches;
    const int D = pscan_words(n, Tmax, (size_t)76 << 10);
    const size_t lds_bytes = (size_t)n * (D + 4) * Tmax * 4 + (((size_t)group * n + 15) & ~(size_t)15);
    if (lds_bytes > ((size_t)158 << 10)) return false;
    const size_t m = (size_t)1 << p;
    const int tiles = (int)(m / ((size_t)32 * D));
    // register ranges: four rounds of two workgroups per CU over all k (just under, never just over), at least 4 tiles each
    int RR = std::max(1, std::min(64, (8 * 256) / K));
    while (RR > 1 && tiles / RR < 4) --RR;
    const int tiles_per_range = tiles;    // (the kernel cuts [0, tiles) into RR ranges itself)
    uint8_t* base = static_cast<uint8_t*>(scratch);
    const uint32_t* rng = reinterpret_cast<const uint32_t*>(base);
    uint32_t* part = reinterpret_cast<uint32_t*>(base + (((size_t)K * 2 * sizeof(uint32_t) + 255) & ~(size_t)255));
    const int pitch = PS_THREADS;
    for (int o0 = 0; o0 < norder; o0 += group) {
        const int no = std::min(group, norder - o0);
#define DD_PSCAN_LAUNCH(NMAX, UPT, DG)                                                                                                         \
    do {                                                                                                                                      \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pscan_kernel<NMAX, UPT, DG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
        hipLaunchKernelGGL((pscan_kernel<NMAX, UPT, DG>), dim3((unsigned)(K * RR)), dim3(PS_THREADS), lds_bytes, st, leaf_dev, n, K, p,        \
                           ord_dev + (size_t)o0 * n, no, rng, RR, tiles_per_range, D, pitch, part);                                           \
    } while (0)
        // (D = 4, 8, 16 or 32 plane words per row: one or two 16-byte groups per chunk of the scan; four groups -- 16
        // running ANDs beside the 32 counts -- spill)
        if (D >= 8) DD_PSCAN_LAUNCH(32, 1, 2);
        else DD_PSCAN_LAUNCH(32, 1, 1);
#undef DD_PSCAN_LAUNCH
        const size_t jobs = (size_t)no * n * K;
        hipLaunchKernelGGL(pscan_finish_kernel, dim3((unsigned)((jobs + 3) / 4)), dim3(256), 0, st, part, n, K, p, no, rng, RR, pitch,
                           hist_dev + (size_t)o0 * n * K * 64);
    }
    return true;
}

}  // namespace dd
