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
// Work unit = one WAVE: (super-block pair of 64 rows, k, register range of <= 65 536, group of TS thresholds).
// The four waves of a workgroup hold four threshold groups of the same rows and registers, so the row bytes they
// stream (64-byte pieces of 64 rows) are fetched once into the CU's L1.  Partial counts of the register ranges are
// stored (plain, coalesced, in the accumulator layout) and summed by gram_finish_kernel, which also turns F into
// the histogram layout dd_union.hip's kernels write: hist[((i * n) + j) * K + kk][64], i <= j.
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
    if ((threadIdx.x & 63) == 0) {
        const int k = (int)(row % (size_t)K);
        atomicMin(&rng[2 * k], lo8);
        atomicMax(&rng[2 * k + 1], hi8);
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
DD_D uint4 mask63(uint4 x) {
    x.x &= 0x3f3f3f3fu, x.y &= 0x3f3f3f3fu, x.z &= 0x3f3f3f3fu, x.w &= 0x3f3f3f3fu;
    return x;
}

// DIAG: both operands are the 64 rows of super-block P: blocks (0,0), (0,1), (1,1) of its 2 x 2 halves.
// !DIAG: rows of P against rows of Q > P: blocks (0,0), (0,1), (1,0), (1,1).
template <bool DIAG>
struct GramShape {
    static constexpr int TS = DIAG ? 4 : 3;   // thresholds per wave: TS x NB x 16 accumulator registers
    static constexpr int NB = DIAG ? 3 : 4;   // 32 x 32 blocks per wave
};

// part[(((unit_sp * K + k) * RR + rr) * slots + slot) * NB * 1024 + block * 1024 + reg * 64 + lane], slot = threshold - vmin_k
template <bool DIAG>
__global__ __launch_bounds__(256, DIAG ? 2 : 1) void gram_kernel(const uint8_t* __restrict__ leaf, int n, int K, int p,
                                                      const uint32_t* __restrict__ rng, int sp_base, int ns,
                                                      int RR, int slots, uint32_t* __restrict__ part) {
    constexpr int TS = GramShape<DIAG>::TS, NB = GramShape<DIAG>::NB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int groups = slots / TS, quads = (groups + 3) >> 2;
    // blockIdx -> (threshold-group quad, register range, k, super-block pair): quad fastest, so the workgroups that
    // stream the same bytes are launched next to each other
    int b = blockIdx.x;
    const int quad = b % quads;
    b /= quads;
    const int rr = b % RR;
    b /= RR;
    const int k = b % K;
    const int sp = b / K;
    const int tg = quad * 4 + wave;
    const int vmin = (int)rng[2 * k], vmax = (int)rng[2 * k + 1];
    const int slot0 = tg * TS;
    // thresholds vmin .. vmax-1 carry information (F = 0 below vmin, m from vmax on)
    if (tg >= groups || vmin + slot0 > vmax - 1) return;
    const int2 PQ = gram_pair<DIAG>(sp_base + sp, ns);
    const int r = lane & 31, half = lane >> 5;
    const size_t m = (size_t)1 << p;
    const size_t len = m < (size_t)kGramRange ? m : (size_t)kGramRange;
    const size_t off0 = (size_t)rr * len + (size_t)half * 32;
    const uint8_t* rows[DIAG ? 2 : 4];
#pragma unroll
    for (int h = 0; h < (DIAG ? 2 : 4); ++h) {
        int row = ((h < 2 ? PQ.x : PQ.y) << 6) + ((h & 1) << 5) + r;
        row = row < n ? row : n - 1;  // rows beyond n: any valid row, their counts are never read
        rows[h] = leaf + (((size_t)row * K + k) << p) + off0;
    }
    uint32_t thr[TS];
#pragma unroll
    for (int t = 0; t < TS; ++t) thr[t] = (uint32_t)(0x80 + vmin + slot0 + t) * 0x01010101u;
    v16i acc[TS][NB];
#pragma unroll
    for (int t = 0; t < TS; ++t)
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][q][e] = 0;

    // 64 registers per iteration: this lane's 32 bytes (two 16-byte k-slices) of each of its rows
    uint4 cur[DIAG ? 2 : 4][2], nxt[DIAG ? 2 : 4][2];
#pragma unroll
    for (int h = 0; h < (DIAG ? 2 : 4); ++h) {
        cur[h][0] = gload16(rows[h]);
        cur[h][1] = gload16(rows[h] + 16);
    }
    for (size_t o = 0; o < len; o += 64) {
        const size_t on = o + 64 < len ? o + 64 : o;  // (the last iteration re-reads its own bytes)
#pragma unroll
        for (int h = 0; h < (DIAG ? 2 : 4); ++h) {
            nxt[h][0] = gload16(rows[h] + on);
            nxt[h][1] = gload16(rows[h] + on + 16);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            uint4 x[DIAG ? 2 : 4];
#pragma unroll
            for (int h = 0; h < (DIAG ? 2 : 4); ++h) x[h] = mask63(cur[h][u]);
#pragma unroll
            for (int t = 0; t < TS; ++t) {
                if (DIAG) {
                    const v4i a0 = threshold16(x[0], thr[t]), a1 = threshold16(x[1], thr[t]);
                    acc[t][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, a0, acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, a1, acc[t][1], 0, 0, 0);
                    acc[t][2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, a1, acc[t][2], 0, 0, 0);
                } else {
                    const v4i a0 = threshold16(x[0], thr[t]), a1 = threshold16(x[1], thr[t]);
                    const v4i b0 = threshold16(x[2], thr[t]), b1 = threshold16(x[3], thr[t]);
                    acc[t][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b1, acc[t][1], 0, 0, 0);
                    acc[t][2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b0, acc[t][2], 0, 0, 0);
                    acc[t][3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, acc[t][3], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < (DIAG ? 2 : 4); ++h) {
            cur[h][0] = nxt[h][0];
            cur[h][1] = nxt[h][1];
        }
    }
    // counts out: accumulator = 2^14 x hits
    uint32_t* out = part + ((((size_t)sp * K + k) * RR + rr) * slots + slot0) * (size_t)(NB * 1024) + lane;
#pragma unroll
    for (int t = 0; t < TS; ++t) {
        if (vmin + slot0 + t > vmax - 1) break;
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) gstore4(out + ((size_t)t * NB + q) * 1024 + e * 64, (uint32_t)acc[t][q][e] >> 14);
    }
}

// One wave per (pair i <= j of the super-block pairs [sp_begin, sp_begin + sp_count), k): lane v sums the partial
// counts of threshold v over the register ranges, the wave differences F into hist[((i * n) + j) * K + k][64].
__global__ __launch_bounds__(256) void gram_finish_kernel(const uint32_t* __restrict__ part, int n, int K, int p,
                                                          const uint32_t* __restrict__ rng, int diag, int sp_base, int ns,
                                                          int sp_count, int RR, int slots, uint32_t* __restrict__ hist) {
    const int lane = threadIdx.x & 63;
    const size_t job = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // (sp, il, jl, k), k fastest
    const int k = (int)(job % (size_t)K);
    size_t rest = job / (size_t)K;
    const int jl = (int)(rest & 63);
    rest >>= 6;
    const int il = (int)(rest & 63);
    const int sp = (int)(rest >> 6);
    if (sp >= sp_count) return;
    const int2 PQ = diag ? gram_pair<true>(sp_base + sp, ns) : gram_pair<false>(sp_base + sp, ns);
    const int i = (PQ.x << 6) + il, j = (PQ.y << 6) + jl;
    if (i >= n || j >= n || i > j) return;
    const int bi = il >> 5, bj = jl >> 5, rr_ = il & 31, cc = jl & 31;
    const int NB = diag ? 3 : 4;
    const int block = diag ? bi + bj : bi * 2 + bj;
    const int reg = (rr_ & 3) + 4 * (rr_ >> 3), ln = cc + 32 * ((rr_ >> 2) & 1);
    const int vmin = (int)rng[2 * k], vmax = (int)rng[2 * k + 1];
    const uint32_t m = 1u << p;
    uint32_t F;
    if (lane < vmin)
        F = 0;
    else if (lane >= vmax)
        F = m;
    else {
        F = 0;
        const uint32_t* src = part + (((size_t)sp * K + k) * RR * slots + (size_t)(lane - vmin)) * (size_t)(NB * 1024) +
                              (size_t)block * 1024 + reg * 64 + ln;
        for (int rr = 0; rr < RR; ++rr) F += gload4(src + (size_t)rr * slots * (size_t)(NB * 1024));
    }
    const uint32_t prev = __shfl_up(F, 1);
    gstore4(hist + (((size_t)i * n + j) * K + k) * 64 + lane, lane ? F - prev : F);
}

}  // namespace

bool gram_usable(int n, int p) { return p >= 12 && n >= 2; }

// thresholds a k column can need: registers live in 0 .. 64 - p + 1
static int gram_slots(int p, int ts) {
    const int t = 64 - p + 1;  // thresholds 0 .. 64 - p
    return (t + ts - 1) / ts * ts;
}

size_t gram_scratch_bytes(int n, int K, int p, int* sp_per_launch) {
    const int ns = (n + 63) / 64;
    const size_t m = (size_t)1 << p;
    const size_t RR = (m + kGramRange - 1) / kGramRange;
    // a launch covers as many super-block pairs as fit ~1 GiB of partial counts (at least one)
    const size_t per_sp = (size_t)K * RR * (size_t)gram_slots(p, 3) * 4096 * sizeof(uint32_t);
    const size_t total_sp = (size_t)ns * (ns + 1) / 2;
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
    const int RR = (int)((m + kGramRange - 1) / kGramRange);
    int sp_fit = 1;
    (void)gram_scratch_bytes(n, K, p, &sp_fit);
    uint8_t* base = static_cast<uint8_t*>(scratch);
    uint32_t* rng = reinterpret_cast<uint32_t*>(base);
    uint32_t* part = reinterpret_cast<uint32_t*>(base + (((size_t)K * 2 * sizeof(uint32_t) + 255) & ~(size_t)255));

    hipLaunchKernelGGL(gram_range_init_kernel, dim3((unsigned)((K + 63) / 64)), dim3(64), 0, st, rng, K);
    const int pieces = (int)((m + 65535) / 65536);
    hipLaunchKernelGGL(gram_range_kernel, dim3((unsigned)((size_t)n * K * pieces)), dim3(256), 0, st, leaf_dev, K, p, pieces, rng);
    // the diagonal super-block pairs (P, P), then the pairs P < Q; sp_fit of them per launch
    for (int diag = 1; diag >= 0; --diag) {
        const int total = diag ? ns : ns * (ns - 1) / 2;
        const int ts = diag ? GramShape<true>::TS : GramShape<false>::TS;
        const int slots = gram_slots(p, ts);
        const int quads = (slots / ts + 3) / 4;
        for (int a = 0; a < total; a += sp_fit) {
            const int cnt = std::min(sp_fit, total - a);
            const unsigned grid = (unsigned)((size_t)cnt * K * RR * quads);
            if (diag)
                hipLaunchKernelGGL(gram_kernel<true>, dim3(grid), dim3(256), 0, st, leaf_dev, n, K, p, rng, a, ns, RR, slots, part);
            else
                hipLaunchKernelGGL(gram_kernel<false>, dim3(grid), dim3(256), 0, st, leaf_dev, n, K, p, rng, a, ns, RR, slots, part);
            const size_t jobs = (size_t)cnt * 64 * 64 * K;
            hipLaunchKernelGGL(gram_finish_kernel, dim3((unsigned)((jobs + 3) / 4)), dim3(256), 0, st, part, n, K, p, rng, diag, a, ns, cnt,
                               RR, slots, hist_dev);
        }
    }
}

}  // namespace dd
