// dd_gram.hip -- K2 all-pairs: union-cardinality histograms of every pair of sketches as Gram matrices.
//
// Replaces, for `dandd kij` (DeltaTree.pairwise_spiders, /root/reference/lib/huffman_dandd.py:666-695), the
// 2016 x K `dashing union` + `dashing card` process pairs: what is needed per (pair, k) is the 64-bin histogram of
// max(a, b) over the m registers of the two sketches.
//
// The histogram of a byte-max is a contraction in disguise.  max(a_r, b_r) <= v  <=>  a_r <= v and b_r <= v, so the
// cumulative histogram of the pair (i, j) at threshold v is
//         F_ij(v) = sum_r [a_ir <= v] * [a_jr <= v]
// -- for one (k, v) the whole n x n table F is the Gram matrix X X^T of the n x m 0/1 matrix X_v[i][r] = [a_ir <= v],
// and hist_ij(v) = F_ij(v) - F_ij(v-1).  The per-byte LDS atomic of the streaming kernel (dd_union.hip: one atomic
// per register per PAIR, 67.6 G of them for 64 sketches of 1 MiB x 31 k, and every row re-streamed for every
// partner: 32x the slab in traffic) becomes one int8 matrix-core instruction per 32 x 32 pairs x 32 registers:
// v_mfma_i32_32x32x32_i8 on operands thresholded in registers, two VALU instructions per operand dword
//         t = (0x80 + v) * 0x01010101 - x          bit 7 of every byte: x_byte <= v  (no borrow: every byte stays >= 1)
//         t &= 0x80808080                          operand byte = -128 or 0
// so a hit contributes (-128) * (-128) = 2^14 to the int32 accumulator and a wave may sum 2^16 registers before the
// count is taken out (acc >> 14).  Counts are exact integers; histograms are bit-identical to the streaming kernel's.
// Only thresholds between the smallest and the largest register of a k column are computed (gram_range_kernel):
// F is 0 below and m above.
//
// Register bytes must be HLL registers (<= 63), as everywhere in this library.
//
// Work unit = one WAVE: (super-block pair of 64 rows, k, register range, <= TS thresholds).  A k column's thresholds
// are shared out evenly over ceil(T / 4 TS) workgroups and over the four waves of each, which stream the same rows:
// 64-byte pieces of every row, brought into an LDS ring by LDS-DMA (global_load_lds_dwordx4), seven stages in flight
// -- the 192 accumulator registers leave no room for a register prefetch that deep, and the rows come from HBM or the
// Infinity Cache.  Two waves per SIMD.  Partial counts of the register ranges are stored (plain, coalesced, in the
// accumulator layout) and summed by gram_finish_kernel, which also turns F into the histogram layout dd_union.hip's
// kernels write: hist[((i * n) + j) * K + kk][64], i <= j.
//
// Measured (MI355X, 64 sketches of 2^20 registers x 31 k, ~30 thresholds per column; profiles/r03_k2_gram.txt):
// 2.2 ms for the Gram kernel (matrix pipes 64 % busy at the 2.06 GHz the chip holds under this load: the rest is VALU
// issue -- six thresholding instructions per MFMA share the SIMD's issue port with it), 0.38 ms for the range pass
// (the slab read once at 5.6 TB/s), 0.11 ms for the finish; 14.0 ms for the streaming kernel.
#include "dd_common.h"
#include "dd_kernels.h"

#include <algorithm>

namespace dd {
namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kGramRange = 1 << 16;  // registers per wave: 2^14 per hit x 2^16 hits stays below 2^31

DD_D uint32_t bmin4(uint32_t a, uint32_t b) {
    const uint32_t t = (a | 0x80808080u) - b;
    const uint32_t m = ((t >> 7) & 0x01010101u) * 0xFFu;  // 0xFF where a >= b
    return (b & m) | (a & ~m);
}

// smallest and largest register of every k column over all n sketches: rng[2k] = min, rng[2k+1] = max
// (rng starts as {63, 0} pairs).  One workgroup per (row, 64 KiB piece); HBM-bound, the slab is read once.
__global__ __launch_bounds__(256) void gram_range_kernel(const uint8_t* __restrict__ leaf, int K, int p,
                                                         int pieces, uint32_t* __restrict__ rng) {
    const size_t row = blockIdx.x / pieces;
    const int piece = blockIdx.x % pieces;
    const size_t m16 = ((size_t)1 << p) >> 4;
    const size_t per = (m16 + pieces - 1) / pieces;
    const size_t lo = (size_t)piece * per, hi = lo + per < m16 ? lo + per : m16;
    const uint8_t* src = leaf + (row << p);
    uint32_t mx = 0, mn = 0x3f3f3f3fu;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const uint4 v = gload16(src + (i << 4));
        const uint32_t a = v.x & 0x3f3f3f3fu, b = v.y & 0x3f3f3f3fu, c = v.z & 0x3f3f3f3fu, d = v.w & 0x3f3f3f3fu;
        mx = bmax4(bmax4(mx, a), bmax4(b, bmax4(c, d)));
        mn = bmin4(bmin4(mn, a), bmin4(b, bmin4(c, d)));
    }
    uint32_t hi8 = 0, lo8 = 63;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const uint32_t x = (mx >> (8 * b)) & 0xff, y = (mn >> (8 * b)) & 0xff;
        hi8 = x > hi8 ? x : hi8;
        lo8 = y < lo8 ? y : lo8;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const uint32_t x = __shfl_xor(hi8, s), y = __shfl_xor(lo8, s);
        hi8 = x > hi8 ? x : hi8;
        lo8 = y < lo8 ? y : lo8;
    }
    __shared__ uint32_t s_hi[4], s_lo[4];
    if ((threadIdx.x & 63) == 0) s_hi[threadIdx.x >> 6] = hi8, s_lo[threadIdx.x >> 6] = lo8;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            hi8 = s_hi[w] > hi8 ? s_hi[w] : hi8;
            lo8 = s_lo[w] < lo8 ? s_lo[w] : lo8;
        }
        // 62 addresses for ~30 000 workgroups: only a workgroup that would move a bound touches it (a stale read
        // costs a needless atomic, never a wrong bound)
        const int k = (int)(row % (size_t)K);
        if (lo8 < gload4_fresh(&rng[2 * k])) atomicMin(&rng[2 * k], lo8);
        if (hi8 > gload4_fresh(&rng[2 * k + 1])) atomicMax(&rng[2 * k + 1], hi8);
    }
}

// super-block pair number sp -> (P, Q): the diagonal pairs are (sp, sp); the others count P < Q row by row
template <bool DIAG>
DD_D int2 gram_pair(int sp, int ns) {
    if (DIAG) return make_int2(sp, sp);
    int P = 0;
    while (sp >= ns - 1 - P) {
        sp -= ns - 1 - P;
        ++P;
    }
    return make_int2(P, P + 1 + sp);
}

__global__ void gram_range_init_kernel(uint32_t* __restrict__ rng, int K) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) rng[2 * k] = 63, rng[2 * k + 1] = 0;
}

DD_D v4i threshold16(const uint4& x, uint32_t thr) {
    v4i t;
    t.x = (int)((thr - x.x) & 0x80808080u);
    t.y = (int)((thr - x.y) & 0x80808080u);
    t.z = (int)((thr - x.z) & 0x80808080u);
    t.w = (int)((thr - x.w) & 0x80808080u);
    return t;
}
// DIAG: both operands are the 64 rows of super-block P: blocks (0,0), (0,1), (1,1) of its 2 x 2 halves.
// !DIAG: rows of P against rows of Q > P: blocks (0,0), (0,1), (1,0), (1,1).
template <bool DIAG>
struct GramShape {
    static constexpr int TS = DIAG ? 4 : 3;   // thresholds per wave at most: TS x NB x 16 = 192 accumulator registers, two waves per SIMD
    static constexpr int NB = DIAG ? 3 : 4;   // 32 x 32 blocks per wave
};

// LDS ring: a stage is 64 bytes of every row of the unit (NH x 32 rows), filled by LDS-DMA (global_load_lds_dwordx4:
// no staging registers, so kRingDepth stages stay in flight per workgroup -- the rows are streamed from HBM / the
// Infinity Cache with ~1 us of latency to cover and the accumulators leave no registers for a deep prefetch).  A DMA
// writes 1 KiB = lane x 16 B contiguously, its SOURCE address is per lane: lane l of the DMA for 16-row group g
// fetches bytes [16 (l / 16), +16) of row 16 g + l % 16, so the image of a stage is [group][16-byte chunk][row % 16]
// and the 16 lanes that ds_read_b128 services together (rows distinct mod 16, one chunk) hit 64 distinct banks.
constexpr int kRingSlots = 8, kRingDepth = kRingSlots - 1;

DD_D void glds16(const uint8_t* src, uint8_t* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const DD_GLOBAL void*)src, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int N>
DD_D void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// part[(((unit_sp * K + k) * RR + rr) * slots + slot) * NB * 1024 + block * 1024 + reg * 64 + lane], slot = threshold - vmin_k
// The streaming loop of one wave with NT thresholds (NT = 0: a wave that only feeds the ring), fully unrolled over
// its thresholds: NT x NB x 16 accumulator registers.
template <bool DIAG, int NT>
DD_D void gram_body(const uint8_t* const* src, uint8_t* ring, uint32_t ring_lds, int wave, int lane, int nst, uint32_t thr0,
                    uint32_t* __restrict__ out) {
    constexpr int NB = GramShape<DIAG>::NB;
    constexpr int NH = DIAG ? 2 : 4;         // 32-row operand sets of a stage
    constexpr int G = NH / 2;                // DMAs per wave and stage: NH * 2 groups of 16 rows over 4 waves
    constexpr int STAGE = NH * 32 * 64;      // bytes
    constexpr int NA = NT ? NT : 1;
    auto issue = [&](int s) {
        uint8_t* dst = ring + (s % kRingSlots) * STAGE;
#pragma unroll
        for (int g = 0; g < G; ++g) glds16(src[g] + (size_t)s * 64, dst + (wave + 4 * g) * 1024);
    };
    v16i acc[NA][NB];
#pragma unroll
    for (int t = 0; t < NA; ++t)
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][q][e] = 0;
    // operand reads: lane (r = lane % 32, half = lane / 32) takes chunk 2 u + half of rows r (+ 32 h) for k-step u
    const int r = lane & 31, half = lane >> 5;
    const int rd0 = 16 * ((r >> 4) * 64 + half * 16 + (r & 15));   // operand set 0, step 0; set h: + h * 2048; step 1: + 512

    for (int s = 0; s < kRingDepth && s < nst; ++s) issue(s);
    for (int i = 0; i < nst; ++i) {
        if (nst - i >= kRingDepth)
            wait_vm<(kRingDepth - 1) * G>();   // this wave's DMAs of stage i have landed ...
        else
            wait_vm<0>();
        __builtin_amdgcn_s_barrier();          // ... and so have the other waves'; everyone is done with stage i - 1
        if (i + kRingDepth < nst) issue(i + kRingDepth);
        if (NT == 0) continue;
        // (read by hand: the compiler cannot tell which DMA a ds_read of the ring depends on and would drain them all
        // -- s_waitcnt vmcnt(0) -- in front of every read; the counted wait and the barrier above are the ordering)
        const uint32_t st = ring_lds + (uint32_t)((i % kRingSlots) * STAGE + rd0);
        dd_u32x4 raw[2][NH];
        if (DIAG)
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:512\n\tds_read_b128 %2, %4 offset:2048\n\t"
                         "ds_read_b128 %3, %4 offset:2560\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(raw[0][0]), "=&v"(raw[1][0]), "=&v"(raw[0][1]), "=&v"(raw[1][1])
                         : "v"(st)
                         : "memory");
        else
            asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:512\n\tds_read_b128 %2, %8 offset:2048\n\t"
                         "ds_read_b128 %3, %8 offset:2560\n\tds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:4608\n\t"
                         "ds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:6656\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(raw[0][0]), "=&v"(raw[1][0]), "=&v"(raw[0][1]), "=&v"(raw[1][1]), "=&v"(raw[0][2 % NH]),
                           "=&v"(raw[1][2 % NH]), "=&v"(raw[0][3 % NH]), "=&v"(raw[1][3 % NH])
                         : "v"(st)
                         : "memory");
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            uint4 x[NH];
#pragma unroll
            for (int h = 0; h < NH; ++h) x[h] = make_uint4(raw[u][h].x, raw[u][h].y, raw[u][h].z, raw[u][h].w);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const uint32_t thr = thr0 + (uint32_t)t * 0x01010101u;
                if (DIAG) {
                    const v4i a0 = threshold16(x[0], thr), a1 = threshold16(x[1], thr);
                    acc[t][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, a0, acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, a1, acc[t][1], 0, 0, 0);
                    acc[t][2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, a1, acc[t][2], 0, 0, 0);
                } else {
                    const v4i a0 = threshold16(x[0], thr), a1 = threshold16(x[1], thr);
                    const v4i b0 = threshold16(x[2], thr), b1 = threshold16(x[3], thr);
                    acc[t][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b1, acc[t][1], 0, 0, 0);
                    acc[t][2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b0, acc[t][2], 0, 0, 0);
                    acc[t][3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, acc[t][3], 0, 0, 0);
                }
            }
        }
    }
    // counts out: accumulator = 2^14 x hits
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) gstore4(out + ((size_t)t * NB + q) * 1024 + e * 64, (uint32_t)acc[t][q][e] >> 14);
}

// part[(((unit_sp * K + k) * RR + rr) * slots + slot) * NB * 1024 + block * 1024 + reg * 64 + lane], slot = threshold - vmin_k
template <bool DIAG, int TS>
__global__ __launch_bounds__(256, 2) void gram_kernel(const uint8_t* __restrict__ leaf, int n, int K, int p,
                                                      const uint32_t* __restrict__ rng, int sp_base, int ns,
                                                      int RR, int len, int slots, uint32_t* __restrict__ part) {
    constexpr int NB = GramShape<DIAG>::NB;
    constexpr int NH = DIAG ? 2 : 4;
    constexpr int G = NH / 2;
    constexpr int STAGE = NH * 32 * 64;
    __shared__ __attribute__((aligned(1024))) uint8_t ring[kRingSlots * STAGE];
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)ring;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int quads = (slots + 4 * TS - 1) / (4 * TS);
    // blockIdx -> (threshold quad, register range, k, super-block pair): quad fastest, so the workgroups that
    // stream the same bytes are launched next to each other
    int b = blockIdx.x;
    const int quad = b % quads;
    b /= quads;
    const int rr = b % RR;
    b /= RR;
    const int k = b % K;
    const int sp = b / K;
    // thresholds vmin .. vmax-1 carry information (F = 0 below vmin, m from vmax on): T of them, 4 TS per workgroup,
    // shared out evenly over its four waves (one wave per SIMD: the matrix pipes of a CU finish together)
    const int vmin = (int)rng[2 * k], vmax = (int)rng[2 * k + 1];
    const int T = vmax - vmin;
    const int nq = (T + 4 * TS - 1) / (4 * TS);       // workgroups that share the column's thresholds, evenly
    if (quad >= nq) return;
    const int q0 = quad * (T / nq) + (quad < T % nq ? quad : T % nq);
    const int mine_all = T / nq + (quad < T % nq ? 1 : 0);
    const int per = (mine_all + 3) >> 2;
    const int slot0 = q0 + wave * per;                                   // first threshold of this wave, relative to vmin
    const int left = mine_all - wave * per;
    const int nthr = __builtin_amdgcn_readfirstlane(left < per ? (left > 0 ? left : 0) : per);
    const int2 PQ = gram_pair<DIAG>(sp_base + sp, ns);
    // this lane's DMA sources: rows 16 (wave + 4 g) + lane % 16 of the stage, chunk lane / 16
    const uint8_t* src[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int srow = 16 * (wave + 4 * g) + (lane & 15);          // row of the stage: operand set srow / 32
        int row = ((srow < 64 ? PQ.x : PQ.y) << 6) + (srow & 63);
        row = row < n ? row : n - 1;  // rows beyond n: any valid row, their counts are never read
        src[g] = leaf + (((size_t)row * K + k) << p) + (size_t)rr * (size_t)len + (size_t)(lane >> 4) * 16;
    }
    const uint32_t thr0 = (uint32_t)(0x80 + vmin + slot0) * 0x01010101u;
    uint32_t* out = part + ((((size_t)sp * K + k) * RR + rr) * slots + slot0) * (size_t)(NB * 1024) + lane;
    const int nst = len >> 6;
    switch (nthr) {
#define DD_GRAM_CASE(NT) \
    case NT: gram_body<DIAG, (NT <= TS ? NT : 0)>(src, ring, ring_lds, wave, lane, nst, thr0, out); break;   // (NT > TS never occurs)
        DD_GRAM_CASE(1) DD_GRAM_CASE(2) DD_GRAM_CASE(3) DD_GRAM_CASE(4)
#undef DD_GRAM_CASE
        default: gram_body<DIAG, 0>(src, ring, ring_lds, wave, lane, nst, thr0, out); break;
    }
}

// One workgroup per (super-block pair, k, 32 x 32 block, accumulator register): the 64 entries one accumulator register
// holds across the wave's lanes.  Wave w sums the partial counts of thresholds w, w + 4, ... over the register ranges
// (256-byte reads in the layout the Gram kernel stored), the sums are transposed through LDS, and every entry's
// cumulative counts are differenced into its histogram row hist[((i * n) + j) * K + k][64] (256-byte writes).
__global__ __launch_bounds__(256) void gram_finish_kernel(const uint32_t* __restrict__ part, int n, int K, int p,
                                                          const uint32_t* __restrict__ rng, int diag, int sp_base, int ns,
                                                          int RR, int slots, uint32_t* __restrict__ hist) {
    __shared__ uint32_t F[64][65];   // [entry][threshold]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int NB = diag ? 3 : 4;
    int b = blockIdx.x;
    const int reg = b & 15;
    b >>= 4;
    const int block = b % NB;
    b /= NB;
    const int k = b % K;
    const int sp = b / K;
    const int vmin = (int)rng[2 * k], vmax = (int)rng[2 * k + 1];
    const uint32_t m = 1u << p;
    const uint32_t* src = part + ((size_t)sp * K + k) * RR * slots * (size_t)(NB * 1024) + (size_t)block * 1024 + reg * 64 + lane;
    for (int v = wave; v < 64; v += 4) {
        uint32_t f = v < vmin ? 0u : m;
        if (v >= vmin && v < vmax) {
            f = 0;
            for (int rr = 0; rr < RR; ++rr) f += gload4(src + ((size_t)rr * slots + (size_t)(v - vmin)) * (size_t)(NB * 1024));
        }
        F[lane][v] = f;
    }
    __syncthreads();
    const int2 PQ = diag ? gram_pair<true>(sp_base + sp, ns) : gram_pair<false>(sp_base + sp, ns);
    const int bi = diag ? (block == 2) : (block >> 1), bj = diag ? (block >= 1) : (block & 1);
    for (int e = wave; e < 64; e += 4) {
        // entry e of accumulator register `reg`: row (reg & 3) + 8 (reg >> 2) + 4 (e >> 5), column e & 31 of the block
        const int i = (PQ.x << 6) + bi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (e >> 5);
        const int j = (PQ.y << 6) + bj * 32 + (e & 31);
        if (i >= n || j >= n || i > j) continue;
        const uint32_t f = F[e][lane], prev = lane ? F[e][lane - 1] : 0u;
        gstore4(hist + (((size_t)i * n + j) * K + k) * 64 + lane, f - prev);
    }
}

}  // namespace

// rng_dev[2k], rng_dev[2k+1] = smallest / largest register of k column k over the n sketches of the slab
void launch_register_range(const uint8_t* leaf_dev, int n, int K, int p, uint32_t* rng_dev, hipStream_t st) {
    const size_t m = (size_t)1 << p;
    hipLaunchKernelGGL(gram_range_init_kernel, dim3((unsigned)((K + 63) / 64)), dim3(64), 0, st, rng_dev, K);
    const int pieces = (int)((m + 65535) / 65536);
    hipLaunchKernelGGL(gram_range_kernel, dim3((unsigned)((size_t)n * K * pieces)), dim3(256), 0, st, leaf_dev, K, p, pieces, rng_dev);
}

bool gram_usable(int n, int p) { return p >= 12 && n >= 2; }

// thresholds a k column can need: registers live in 0 .. 64 - p + 1, thresholds vmin .. vmax - 1
static int gram_slots(int p) { return 64 - p + 1; }

// registers per wave: the whole row up to 2^16 (the accumulator's headroom), less when that leaves too few
// workgroups to fill the chip -- two per CU at least, not below 1024 registers
static int gram_len(int nsp, int K, int p) {
    size_t len = std::min<size_t>((size_t)1 << p, (size_t)kGramRange);
    while (len > 1024 && (size_t)nsp * K * (((size_t)1 << p) / len) < 512) len >>= 1;
    return (int)len;
}

size_t gram_scratch_bytes(int n, int K, int p, int* sp_per_launch) {
    const int ns = (n + 63) / 64;
    const size_t total_sp = (size_t)ns * (ns + 1) / 2;
    const size_t m = (size_t)1 << p;
    const size_t RR = m / (size_t)gram_len(ns, K, p);   // (the diagonal launch has the fewest units: its split is the finest)
    // a launch covers as many super-block pairs as fit ~1 GiB of partial counts (at least one)
    const size_t per_sp = (size_t)K * RR * (size_t)gram_slots(p) * 4096 * sizeof(uint32_t);
    size_t fit = ((size_t)1 << 30) / per_sp;
    fit = fit < 1 ? 1 : fit > total_sp ? total_sp : fit;
    if (sp_per_launch) *sp_per_launch = (int)fit;
    return fit * per_sp + (((size_t)K * 2 * sizeof(uint32_t) + 255) & ~(size_t)255);
}

// hist_dev must have been zeroed by the caller (the lower triangle stays zero).  scratch: gram_scratch_bytes().
void launch_pairwise_gram(const uint8_t* leaf_dev, int n, int K, int p, uint32_t* hist_dev, void* scratch,
                          hipStream_t st) {
    const int ns = (n + 63) / 64;
    const size_t m = (size_t)1 << p;
    int sp_fit = 1;
    (void)gram_scratch_bytes(n, K, p, &sp_fit);
    uint8_t* base = static_cast<uint8_t*>(scratch);
    uint32_t* rng = reinterpret_cast<uint32_t*>(base);
    uint32_t* part = reinterpret_cast<uint32_t*>(base + (((size_t)K * 2 * sizeof(uint32_t) + 255) & ~(size_t)255));

    launch_register_range(leaf_dev, n, K, p, rng, st);
    // the diagonal super-block pairs (P, P), then the pairs P < Q; sp_fit of them per launch
    const int slots = gram_slots(p);
    const int len = gram_len(ns, K, p);
    const int RR = (int)(m / (size_t)len);
    for (int diag = 1; diag >= 0; --diag) {
        const int total = diag ? ns : ns * (ns - 1) / 2;
        const int ts = diag ? GramShape<true>::TS : GramShape<false>::TS;
        const int quads = (slots + 4 * ts - 1) / (4 * ts);
        for (int a = 0; a < total; a += sp_fit) {
            const int cnt = std::min(sp_fit, total - a);
            const unsigned grid = (unsigned)((size_t)cnt * K * RR * quads);
            if (diag)
                hipLaunchKernelGGL((gram_kernel<true, GramShape<true>::TS>), dim3(grid), dim3(256), 0, st, leaf_dev, n, K, p, rng, a, ns, RR, len, slots, part);
            else
                hipLaunchKernelGGL((gram_kernel<false, GramShape<false>::TS>), dim3(grid), dim3(256), 0, st, leaf_dev, n, K, p, rng, a, ns, RR, len, slots, part);
            hipLaunchKernelGGL(gram_finish_kernel, dim3((unsigned)((size_t)cnt * K * (diag ? 3 : 4) * 16)), dim3(256), 0, st, part, n, K, p, rng,
                               diag, a, ns, RR, slots, hist_dev);
        }
    }
}

}  // namespace dd
