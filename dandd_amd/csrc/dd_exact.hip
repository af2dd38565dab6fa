// dd_exact.hip -- exact number of distinct (canonical) k-mers on the GPU: the KMC stand-in.
//
// Replaces  kmc -ci1 -cs2 -k<K> [-b] -fm <fasta> <db> <tmp>  +  kmc_tools complex (union)  +
// kmc_tools info | grep 'total k-mers'   (/root/reference/lib/sketch_classes.py:395,444-448,453-465):
// every k-mer occurrence of the token stream is materialised (8 or 16 bytes), radix-sorted
// (rocPRIM device radix sort over the 2k significant bits) and the distinct values are counted.
// This is the accuracy yardstick of the benchmark ("delta rel-err vs KMC --exact") and the
// engine's answer to `dandd tree --exact`; it is HBM-bound on the sort passes.
#include <string.h>

#include <algorithm>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "dd_common.h"
#include "dd_kernels.h"

namespace dd {
namespace {

// Partition of the k-mer space for inputs whose k-mers do not fit the HBM budget at once: 4096 bins by a mix
// of the k-mer itself, so equal k-mers always share a bin and the distinct count is the sum over bins.
constexpr int kExactBinBits = 12;
DD_D uint32_t exact_bin(uint64_t lo, uint64_t hi) { return (uint32_t)(splitmix64(lo ^ (hi * 0x9E3779B97F4A7C15ull)) >> (64 - kExactBinBits)); }

// one thread per 64-token segment.
//   MODE 0: the k-mer ending at token t goes to out[base + t] (everything at once; unwritten slots keep a sentinel)
//   MODE 1: only counts k-mers per bin into hist[4096] (LDS histogram per workgroup, one flush)
//   MODE 2: k-mers whose bin is in [bin_lo, bin_hi) are appended densely (wave-aggregated atomic on counters[3])
template <bool CANON, bool WIDE, int MODE>
__global__ __launch_bounds__(256) void kmer_extract_kernel(const ExactGenome* __restrict__ tab, int k,
                                                          uint64_t* __restrict__ out_lo,
                                                          uint64_t* __restrict__ out_hi,
                                                          unsigned long long* __restrict__ counters,
                                                          unsigned long long* __restrict__ hist, uint32_t bin_lo,
                                                          uint32_t bin_hi) {
    __shared__ uint32_t lhist[MODE == 1 ? (1 << kExactBinBits) : 1];
    if (MODE == 1) {
        for (int i = threadIdx.x; i < (1 << kExactBinBits); i += blockDim.x) lhist[i] = 0;
        __syncthreads();
    }
    const ExactGenome g = tab[blockIdx.y];
    const unsigned long long ntok = *g.ntok;
    const unsigned long long seg = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned valid = 0, allt = 0;
    // (MODE 2 keeps every lane in the token loop: the append is a wave-level operation)
    const bool live = seg * kSegTokens < ntok;
    if (MODE == 2 ? __any(live) : live) {
        const uint4* codes4 = reinterpret_cast<const uint4*>(g.codes);
        const uint2* bad2 = reinterpret_cast<const uint2*>(g.bad);
        // 128-bit windows serve every k <= 64; the sort, not this kernel, is the cost
        uint64_t fh = 0, fl = 0, rh = 0, rl = 0;
        int run = 0;
        const uint64_t mlo = (k >= 32) ? ~0ull : ((1ull << (2 * k)) - 1ull);
        const uint64_t mhi = (k <= 32) ? 0ull : ((k == 64) ? ~0ull : ((1ull << (2 * k - 64)) - 1ull));
        const int s = 128 - 2 * k;  // right shift that aligns the reverse-complement window
        for (int part = (seg > 0 ? 0 : 1); part < 2; ++part) {
            const unsigned long long sidx = seg - 1 + part;
            const uint4 c4 = live ? codes4[sidx] : make_uint4(0, 0, 0, 0);
            const uint2 b2 = live ? bad2[sidx] : make_uint2(~0u, ~0u);
            const uint32_t cw[4] = {c4.x, c4.y, c4.z, c4.w};
            const uint64_t bw = ((uint64_t)b2.y << 32) | b2.x;
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll 1
            for (int i = 0; i < 16; ++i) {
                const int t = w * 16 + i;
                const uint32_t c = (cw[w] >> (2 * i)) & 3u;
                run = ((bw >> t) & 1ull) ? 0 : run + 1;
                fh = (fh << 2) | (fl >> 62);
                fl = (fl << 2) | c;
                rl = (rl >> 2) | (rh << 62);
                rh = (rh >> 2) | ((uint64_t)(3u - c) << 62);
                const bool have = part != 0 && run >= k;
                if (MODE != 2 && !have) continue;
                uint64_t ah = fh & mhi, al = fl & mlo;
                if (CANON) {
                    uint64_t bh, bl;
                    if (s >= 64) {
                        bh = 0;
                        bl = rh >> (s - 64);
                    } else if (s == 0) {
                        bh = rh;
                        bl = rl;
                    } else {
                        bh = rh >> s;
                        bl = (rl >> s) | (rh << (64 - s));
                    }
                    if (bh < ah || (bh == ah && bl < al)) {
                        ah = bh;
                        al = bl;
                    }
                }
                if (MODE == 0) {
                    const unsigned long long pos = g.base + sidx * kSegTokens + (unsigned)t;
                    out_lo[pos] = al;
                    if (WIDE) out_hi[pos] = ah;
                    ++valid;
                    if (al == mlo && ah == mhi) allt = 1;
                } else if (MODE == 1) {
                    atomicAdd(&lhist[exact_bin(al, WIDE ? ah : 0ull)], 1u);
                } else {
                    const uint32_t bin = exact_bin(al, WIDE ? ah : 0ull);
                    const bool take = have && bin >= bin_lo && bin < bin_hi;
                    const unsigned long long mask = __ballot(take);
                    if (mask) {
                        const uint32_t lane = threadIdx.x & 63u;
                        unsigned long long basepos = 0;
                        if (lane == (uint32_t)__builtin_ctzll(mask)) basepos = atomicAdd(&counters[3], (unsigned long long)__builtin_popcountll(mask));
                        basepos = __shfl(basepos, __builtin_ctzll(mask));
                        if (take) {
                            const unsigned long long pos = basepos + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
                            out_lo[pos] = al;
                            if (WIDE) out_hi[pos] = ah;
                        }
                    }
                }
            }
        }
    }
    if (MODE == 1) {
        __syncthreads();
        for (int i = threadIdx.x; i < (1 << kExactBinBits); i += blockDim.x)
            if (lhist[i]) atomicAdd(&hist[i], (unsigned long long)lhist[i]);
        return;
    }
    if (MODE == 2) return;
    // wave-level reduction of the two statistics, one atomic per wave
    for (int d = 32; d > 0; d >>= 1) {
        valid += __shfl_down(valid, d);
        allt |= __shfl_down(allt, d);
    }
    if ((threadIdx.x & 63) == 0) {
        if (valid) atomicAdd(&counters[0], (unsigned long long)valid);
        if (allt) atomicOr(&counters[1], 1ull);
    }
}

template <bool WIDE>
__global__ __launch_bounds__(256) void count_unique_kernel(const uint64_t* __restrict__ lo,
                                                          const uint64_t* __restrict__ hi, size_t n,
                                                          uint64_t mlo, uint64_t mhi,
                                                          unsigned long long* __restrict__ counters) {
    unsigned cnt = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        bool fresh = (i == 0);
        if (!fresh) {
            fresh = ((lo[i] ^ lo[i - 1]) & mlo) != 0;
            if (WIDE) fresh |= ((hi[i] ^ hi[i - 1]) & mhi) != 0;
        }
        cnt += fresh;
    }
    for (int d = 32; d > 0; d >>= 1) cnt += __shfl_down(cnt, d);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&counters[2], (unsigned long long)cnt);
}

}  // namespace

void launch_kmer_extract(const ExactGenome* tab_dev, int ng, size_t max_segments, int k, int canonical,
                         uint64_t* lo, uint64_t* hi, unsigned long long* counters, hipStream_t st, int mode,
                         unsigned long long* hist, uint32_t bin_lo, uint32_t bin_hi) {
    if (ng <= 0 || !max_segments) return;
    const dim3 grid((unsigned)((max_segments + 255) / 256), (unsigned)ng), block(256);
    const bool wide = k > 32;
#define DD_EX(CN, WD, MD) \
    hipLaunchKernelGGL((kmer_extract_kernel<CN, WD, MD>), grid, block, 0, st, tab_dev, k, lo, hi, counters, hist, bin_lo, bin_hi)
#define DD_EX_MODE(CN, WD)             \
    do {                               \
        if (mode == 0) DD_EX(CN, WD, 0); \
        else if (mode == 1) DD_EX(CN, WD, 1); \
        else DD_EX(CN, WD, 2);         \
    } while (0)
    if (canonical) {
        if (wide) DD_EX_MODE(true, true); else DD_EX_MODE(true, false);
    } else {
        if (wide) DD_EX_MODE(false, true); else DD_EX_MODE(false, false);
    }
#undef DD_EX_MODE
#undef DD_EX
}

size_t exact_sort_temp_bytes(size_t n, int k) {
    size_t a = 0, b = 0;
    uint64_t* nul = nullptr;
    (void)rocprim::radix_sort_keys(nullptr, a, nul, nul, n, 0, 64);
    if (k > 32) (void)rocprim::radix_sort_pairs(nullptr, b, nul, nul, nul, nul, n, 0, 64);
    return std::max(a, b) + 256;
}

// Sorts (lo[, hi]) using (lo_alt[, hi_alt]) as the other half of the double buffer and adds the
// number of distinct values (compared on their 2k significant bits) to counters[2].
hipError_t launch_exact_sort_count(uint64_t* lo, uint64_t* hi, uint64_t* lo_alt, uint64_t* hi_alt, size_t n,
                                   int k, void* temp, size_t temp_bytes, unsigned long long* counters,
                                   hipStream_t st) {
    if (!n) return hipSuccess;
    const uint64_t mlo = (k >= 32) ? ~0ull : ((1ull << (2 * k)) - 1ull);
    const uint64_t mhi = (k <= 32) ? 0ull : ((k == 64) ? ~0ull : ((1ull << (2 * k - 64)) - 1ull));
    hipError_t e;
    const uint64_t *slo, *shi = nullptr;
    if (k <= 32) {
        e = rocprim::radix_sort_keys(temp, temp_bytes, lo, lo_alt, n, 0, (unsigned)(2 * k), st);
        if (e != hipSuccess) return e;
        slo = lo_alt;
    } else {
        // stable LSD over the 128-bit key: low word first, then the high word's 2k-64 bits
        e = rocprim::radix_sort_pairs(temp, temp_bytes, lo, lo_alt, hi, hi_alt, n, 0, 64, st);
        if (e != hipSuccess) return e;
        e = rocprim::radix_sort_pairs(temp, temp_bytes, hi_alt, hi, lo_alt, lo, n, 0, (unsigned)(2 * k - 64), st);
        if (e != hipSuccess) return e;
        slo = lo;
        shi = hi;
    }
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (k <= 32)
        hipLaunchKernelGGL(count_unique_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, slo, shi, n, mlo, mhi, counters);
    else
        hipLaunchKernelGGL(count_unique_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, slo, shi, n, mlo, mhi, counters);
    return hipGetLastError();
}

}  // namespace dd
