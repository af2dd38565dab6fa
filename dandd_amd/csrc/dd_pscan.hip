// dd_pscan.hip -- K2 progressive unions as a bit-plane AND-scan.
//
// Replaces, for `dandd progressive` (DeltaTree.sketch_ordering / progressive_union,
// /root/reference/lib/huffman_dandd.py:624-663), the Σj `dashing union` + `dashing card` processes per ordering and k.
// What is needed per (ordering o, prefix j, k) is the 64-bin histogram of U_j = max(leaf[o_0], .., leaf[o_j]).
//
// The streaming kernel (dd_union.hip: progressive_kernel) keeps U_j in registers and pays one LDS atomic per register
// per prefix -- 4.7 cycles per wave-wide ds_add_u32 whatever the bytes are (scripts/ubench_lds_atomic.hip): a 1.4 ms
// floor for 10 orderings x 30 prefixes x 37 k of 2^20 registers, 3.1 ms measured.  Here the cumulative histogram is
// counted instead: U_j[r] <= v  <=>  every leaf of the prefix has leaf[r] <= v, so with the bit planes
//         B_g,v = { r : leaf_g[r] <= v }          (one bit per register)
// F_o,j(v) = popcount(B_o0,v & B_o1,v & .. & B_oj,v): a running AND along the ordering and one popcount per prefix --
// 32 registers per instruction instead of one -- and hist_j(v) = F_j(v) - F_j(v-1).  Only thresholds between the
// smallest and the largest register of the k column are needed (gram_range_kernel of dd_gram.hip).
//
// One workgroup per (k, register range); per tile of 32 D registers:
//   convert   every thread takes (leaf g, 32 registers): the 32 bytes are bit-sliced into six planes (two
//             instructions per byte-dword and bit), and each needed threshold's plane is eq(vmin) | .. | eq(v), five
//             ANDs of planes or their complements per threshold; planes go to LDS as [g][d][threshold]
//   scan      one LANE per chain (ordering, threshold): for each of the D plane words P = ~0, then for every prefix
//             P &= plane[o_j][d][t], count_j += popcount(P) -- the counts stay in the lane's registers across all
//             tiles of the range, so nothing is reduced until the very end.
// The conversion is shared by every ordering and threshold of the workgroup, which is what makes it affordable.
// Exact integers throughout; tests/test_gpu_parity.py checks every cardinality against the streaming kernel's.
#include "dd_common.h"
#include "dd_kernels.h"

#include <algorithm>

namespace dd {
namespace {

constexpr int PS_THREADS = 512;

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

// NMAX: prefixes held in registers (n <= NMAX)
template <int NMAX>
__global__ __launch_bounds__(PS_THREADS) void pscan_kernel(const uint8_t* __restrict__ leaf, int n, int K, int p,
                                                           const int32_t* __restrict__ ord, int no, const uint32_t* __restrict__ rng,
                                                           int RR, int tiles_per_range, int D, int chain_pitch,
                                                           uint32_t* __restrict__ part) {
    extern __shared__ uint32_t lds[];            // planes [g][d][T], then the orderings [no][n] as bytes
    const int k = blockIdx.x / RR, rr = blockIdx.x % RR;
    const int vmin = (int)rng[2 * k], vmax = (int)rng[2 * k + 1];
    const int T = vmax - vmin;                   // thresholds vmin .. vmax-1
    if (T <= 0) return;                          // every register of the column equal: F is 0 below it, m from it on
    uint32_t* planes = lds;
    uint8_t* ord_s = reinterpret_cast<uint8_t*>(lds + (size_t)n * D * T);
    for (int i = threadIdx.x; i < no * n; i += PS_THREADS) ord_s[i] = (uint8_t)ord[i];
    __syncthreads();
    // this lane's chain
    const int c = threadIdx.x;
    const bool chain = c < no * T;
    const int o = chain ? c / T : 0, t = chain ? c % T : 0;
    uint32_t base[NMAX], cnt[NMAX];
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {
        base[j] = j < n ? (uint32_t)((int)ord_s[o * n + j] * D * T + t) : 0u;
        cnt[j] = 0;
    }
    const int units = n * D;                     // (leaf, 32 registers) pairs of a tile
    const size_t tile_regs = (size_t)32 * D;
    const size_t reg0 = ((size_t)rr * tiles_per_range) * tile_regs;
    constexpr int UPT = 4;                       // units per thread at most (units <= UPT * PS_THREADS: the host's choice of D)
    uint4 cur[UPT][2], nxt[UPT][2];
    auto load = [&](uint4 (&dst)[UPT][2], int tile) {
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
            const int u = (int)threadIdx.x + q * PS_THREADS;
            if (u < units) {
                const int g = u / D, d = u % D;
                const uint8_t* src = leaf + (((size_t)g * K + k) << p) + reg0 + (size_t)tile * tile_regs + (size_t)d * 32;
                dst[q][0] = gload16(src);
                dst[q][1] = gload16(src + 16);
            }
        }
    };
    load(cur, 0);
    for (int tile = 0; tile < tiles_per_range; ++tile) {
        if (tile + 1 < tiles_per_range) load(nxt, tile + 1);
        // ---- convert: bytes -> threshold planes
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
            const int u = (int)threadIdx.x + q * PS_THREADS;
            if (u < units) {
                const uint32_t w[8] = {cur[q][0].x, cur[q][0].y, cur[q][0].z, cur[q][0].w, cur[q][1].x, cur[q][1].y, cur[q][1].z, cur[q][1].w};
                uint32_t x[6], nx[6];
                bit_slice(w, x);
#pragma unroll
                for (int b = 0; b < 6; ++b) nx[b] = ~x[b];
                uint32_t* dst = planes + (size_t)u * T;      // [g][d][.]: u = g * D + d
                uint32_t le = 0;
#pragma unroll
                for (int v = 0; v < 64; ++v) {
                    if (v >= vmin && v < vmax) {             // (wave-uniform; the bits of v are compile-time constants)
                        uint32_t eq = (v & 1) ? x[0] : nx[0];
#pragma unroll
                        for (int b = 1; b < 6; ++b) eq &= ((v >> b) & 1) ? x[b] : nx[b];
                        le |= eq;
                        dst[v - vmin] = le;
                    }
                }
            }
        }
        __syncthreads();
        // ---- scan: running AND along the ordering, one popcount per prefix
        if (chain) {
            for (int d = 0; d < D; ++d) {
                uint32_t P = ~0u;
#pragma unroll
                for (int j = 0; j < NMAX; ++j) {
                    if (j < n) {
                        P &= planes[base[j] + (uint32_t)(d * T)];
                        cnt[j] += (uint32_t)__popc(P);
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
            cur[q][0] = nxt[q][0];
            cur[q][1] = nxt[q][1];
        }
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
static int pscan_words(int n, int T) {
    int D = 32;
    while (D > 1 && ((size_t)n * D * T * 4 > ((size_t)128 << 10) || n * D > 4 * PS_THREADS)) D >>= 1;
    return D;
}

bool pscan_usable(int n, int no, int p) { return p >= 12 && n >= 2 && n <= 64 && no >= 1; }

// scratch: the range pairs of every k, then the partial counts [k][range][prefix][chain]
size_t pscan_scratch_bytes(int n, int K, int p, int no) {
    const size_t pitch = ((size_t)std::min(no, 8) * (size_t)(64 - p + 1) + 63) / 64 * 64;
    return (((size_t)K * 2 * sizeof(uint32_t) + 255) & ~(size_t)255) + (size_t)K * 64 * (size_t)n * pitch * sizeof(uint32_t);
}

// rng_host: the (min, max) pairs gram_range_kernel left at the start of `scratch`, read back by the caller (K pairs).
// Orderings are taken eight at a time (8 x 45 thresholds at most = 360 chains <= 512 lanes).  hist_dev is written
// in full.  Returns false when a column's thresholds do not fit (the caller falls back to the streaming kernel).
bool launch_progressive_pscan(const uint8_t* leaf_dev, int n, int K, int p, const int32_t* ord_dev, int norder, const uint32_t* rng_host,
                              void* scratch, uint32_t* hist_dev, hipStream_t st) {
    int Tmax = 1;
    for (int k = 0; k < K; ++k) Tmax = std::max(Tmax, (int)rng_host[2 * k + 1] - (int)rng_host[2 * k]);
    const int group = 8;
    if (Tmax * std::min(norder, group) > PS_THREADS) return false;
    const int D = pscan_words(n, Tmax);
    const size_t lds_bytes = (size_t)n * D * Tmax * 4 + (((size_t)group * n + 15) & ~(size_t)15);
    if (lds_bytes > ((size_t)150 << 10)) return false;
    const size_t m = (size_t)1 << p;
    const int tiles = (int)(m / ((size_t)32 * D));
    // register ranges: ~4 workgroups per CU over all k, at least 8 tiles each
    int RR = 1;
    while (RR < 64 && K * RR < 1024 && tiles / (RR * 2) >= 8) RR *= 2;
    const int tiles_per_range = tiles / RR;
    uint8_t* base = static_cast<uint8_t*>(scratch);
    const uint32_t* rng = reinterpret_cast<const uint32_t*>(base);
    uint32_t* part = reinterpret_cast<uint32_t*>(base + (((size_t)K * 2 * sizeof(uint32_t) + 255) & ~(size_t)255));
    const int pitch = (int)(((size_t)std::min(norder, group) * (size_t)(64 - p + 1) + 63) / 64 * 64);
    for (int o0 = 0; o0 < norder; o0 += group) {
        const int no = std::min(group, norder - o0);
        if (n <= 32) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pscan_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            hipLaunchKernelGGL(pscan_kernel<32>, dim3((unsigned)(K * RR)), dim3(PS_THREADS), lds_bytes, st, leaf_dev, n, K, p, ord_dev + (size_t)o0 * n, no,
                               rng, RR, tiles_per_range, D, pitch, part);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pscan_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            hipLaunchKernelGGL(pscan_kernel<64>, dim3((unsigned)(K * RR)), dim3(PS_THREADS), lds_bytes, st, leaf_dev, n, K, p, ord_dev + (size_t)o0 * n, no,
                               rng, RR, tiles_per_range, D, pitch, part);
        }
        const size_t jobs = (size_t)no * n * K;
        hipLaunchKernelGGL(pscan_finish_kernel, dim3((unsigned)((jobs + 3) / 4)), dim3(256), 0, st, part, n, K, p, no, rng, RR, pitch,
                           hist_dev + (size_t)o0 * n * K * 64);
    }
    return true;
}

}  // namespace dd
