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
// Work unit = one WAVE: (unit of rows, k, register range, <= TS thresholds).  A k column's thresholds are shared out over
// ceil(T / (WAVES TS)) workgroups and over the waves of each, which stream the same rows: 64-byte pieces of every row, brought
// into an LDS ring by LDS-DMA (global_load_lds_dwordx4), seven stages in flight -- the 160-192 accumulator registers leave no
// room for a register prefetch that deep, and the rows come from HBM or the Infinity Cache.  Two waves per SIMD.  Partial counts
// of the register ranges are stored (plain, coalesced, in the accumulator layout) and summed by gram_finish_kernel, which also
// turns F into the histogram layout dd_union.hip's kernels write: hist[((i * n) + j) * K + kk][64], i <= j.
//
// Round 3 (MI355X, 64 sketches of 2^20 registers x 31 k, ~30 thresholds per column; profiles/r03_k2_gram.txt): 2.2 ms for the
// Gram kernel, 0.38 ms for the range pass (the slab read once at 5.6 TB/s), 0.11 ms for the finish; 14.0 ms for the streaming
// kernel.  Round 5 (profiles/r05_k2_gram.txt):
//   * taking the barrier, the LDS reads, the DMAs or the thresholding OUT of the kernel, one at a time, moves its time by < 7 %:
//     at n = 64 it runs at the pace of its matrix instructions at the clock the chip holds under them (~2.0 GHz);
//   * n > 64 was a different matter: every column had ids for the widest column a sketch can have and the unused ones left at
//     once.  Workgroups go round the XCDs, and round the shader engines inside one, in id order, so a PERIODIC pattern of empty ids
//     starves some of them for the whole launch: the off-diagonal kernel (3 of 4 ids used) ran on six of the eight XCDs.  The
//     ids are compact now (gram_kernel): n = 128 at log2m 20 8.9 -> 7.7 ms, n = 256 (K 8) 9.0 -> 7.3 ms;
//   * 128-row diagonal units in workgroups of eight waves (kDiag2): n = 128 7.7 -> 7.4 ms, 0.52-0.56 of the int8 dense peak
//     at the nominal 2.4 GHz for the whole call (range pass, finish and estimator included).
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

// A wave's unit of rows, three shapes:
//   kOff    rows of 64-row super-block P against rows of Q > P: blocks (0,0), (0,1), (1,0), (1,1) of 32 x 32 pairs
//   kDiag   the 64 rows of super-block P against themselves (n <= 64): blocks (0,0), (0,1), (1,1) of its 2 x 2 halves
//   kDiag2  (round 5, n > 64) the 128 rows of super-blocks 2u, 2u + 1 against themselves: the ten blocks i <= j of its 4 x 4
//           quarters.  A thresholded operand serves five matrix instructions instead of three (kDiag) or two (kOff): 3.2
//           thresholding instructions per MFMA instead of 5.3 / 8 on the issue port the two share.
enum { kOff = 0, kDiag = 1, kDiag2 = 2 };

// unit number sp -> (P, Q): kDiag (sp, sp); kDiag2 (2 sp, 2 sp + 1); kOff counts P < Q row by row -- with `paired`, without the
// pairs (2 u, 2 u + 1), which are inside the kDiag2 units
template <int SHAPE>
DD_D int2 gram_pair(int sp, int ns, int paired = 0) {
    if (SHAPE == kDiag) return make_int2(sp, sp);
    if (SHAPE == kDiag2) return make_int2(2 * sp, 2 * sp + 1);
    int P = 0;
    for (;;) {
        const int gone = (paired && !(P & 1) && P + 1 < ns) ? 1 : 0;
        const int cnt = ns - 1 - P - gone;
        if (sp < cnt) return make_int2(P, P + 1 + gone + sp);
        sp -= cnt;
        ++P;
    }
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
template <int SHAPE>
struct GramShape {
    static constexpr int TS = SHAPE == kDiag ? 4 : SHAPE == kOff ? 3 : 1;   // thresholds per wave at most: TS x NB x 16 = 192 (160) accumulator registers, two waves per SIMD
    static constexpr int NB = SHAPE == kDiag ? 3 : SHAPE == kOff ? 4 : 10;  // 32 x 32 blocks per wave
    static constexpr int NH = SHAPE == kDiag ? 2 : 4;                       // 32-row operand sets of a stage
    // waves of a workgroup: they stream the same rows and have their own thresholds.  kDiag2's one threshold per wave is too
    // little work per byte streamed for four (the LDS-DMA issue and the L2 -> LDS traffic set its pace: 20 matrix instructions per
    // two DMAs of a wave): eight waves, one workgroup per CU, one DMA per wave and stage
    static constexpr int WAVES = SHAPE == kDiag2 ? 8 : 4;
};
// block q of a kDiag2 unit -> (quarter i, quarter j), i <= j
DD_HD int diag2_bi(int q) { return q < 4 ? 0 : q < 7 ? 1 : q < 9 ? 2 : 3; }
DD_HD int diag2_bj(int q) { return q < 4 ? q : q < 7 ? q - 3 : q < 9 ? q - 5 : 3; }

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
template <int SHAPE, int NT>
DD_D void gram_body(const uint8_t* const* src, uint8_t* ring, uint32_t ring_lds, int wave, int lane, int nst, uint32_t thr0,
                    uint32_t* __restrict__ out) {
    constexpr int NB = GramShape<SHAPE>::NB;
    constexpr int NH = GramShape<SHAPE>::NH; // 32-row operand sets of a stage
    constexpr bool DIAG = SHAPE == kDiag;
    constexpr int WAVES = GramShape<SHAPE>::WAVES;
    constexpr int G = NH * 2 / WAVES;        // DMAs per wave and stage: NH * 2 groups of 16 rows over the waves
    constexpr int STAGE = NH * 32 * 64;      // bytes
    constexpr int NA = NT ? NT : 1;
    auto issue = [&](int s) {
        uint8_t* dst = ring + (s % kRingSlots) * STAGE;
#pragma unroll
        for (int g = 0; g < G; ++g) glds16(src[g] + (size_t)s * 64, dst + (wave + WAVES * g) * 1024);
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
                } else if (SHAPE == kDiag2) {
                    v4i a[4];
#pragma unroll
                    for (int h = 0; h < 4; ++h) a[h] = threshold16(x[h % NH], thr);
#pragma unroll
                    for (int q = 0; q < NB; ++q)
                        acc[t][q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[diag2_bi(q)], a[diag2_bj(q)], acc[t][q], 0, 0, 0);
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
template <int SHAPE, int TS>
__global__ __launch_bounds__(GramShape<SHAPE>::WAVES * 64, 8 / GramShape<SHAPE>::WAVES) void gram_kernel(const uint8_t* __restrict__ leaf, int n, int K, int p,
                                                      const uint32_t* __restrict__ rng, int sp_base, int ns,
                                                      int RR, int len, int slots, int units, int units_sp, int paired, uint32_t* __restrict__ part) {
    constexpr int NB = GramShape<SHAPE>::NB;
    constexpr int NH = GramShape<SHAPE>::NH;
    constexpr int WAVES = GramShape<SHAPE>::WAVES;
    constexpr int G = NH * 2 / WAVES;
    constexpr int STAGE = NH * 32 * 64;
    __shared__ __attribute__((aligned(1024))) uint8_t ring[kRingSlots * STAGE];
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)ring;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // blockIdx -> (unit, k, register range, workgroup of the column): a column's thresholds vmin .. vmax-1 (F = 0 below vmin, m from
    // vmax on) -- T of them -- are shared out over nq = ceil(T / (WAVES TS)) workgroups and the waves of each (one or two waves per
    // SIMD: the matrix pipes of a CU finish together).  The ids are COMPACT: the grid is sized for the widest column a sketch can
    // have (`quads` workgroups), the ids beyond the workgroups there really are leave at once, and they are the LAST ids.  Round 4
    // gave every column `quads` ids and let the unused ones leave: workgroups go round the XCDs (and the shader engines inside one)
    // in id order, a periodic pattern of empty ids leaves some of them with less work for the whole launch -- the off-diagonal
    // kernel (3 of 4 ids used) ran on six of the eight XCDs (profiles/r05_k2_gram.txt).
    int item = blockIdx.x;
    if (item >= units) return;
    int k = -1, nq = 0, before_k = 0, vmin = 0, T = 0, sp = 0;
    {
        // workgroups of one unit: RR x sum_k nq_k; columns in chunks of 64 (one wave computes the running sums)
        int per_unit = 0;
        for (int c0 = 0; c0 < K; c0 += 64) {
            const int kk = c0 + lane;
            int mine = 0;
            if (kk < K) mine = ((int)rng[2 * kk + 1] - (int)rng[2 * kk] + WAVES * TS - 1) / (WAVES * TS);
            int incl = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl_up(incl, d);
                if (lane >= d) incl += up;
            }
            per_unit += __shfl(incl, 63);
        }
        per_unit *= RR;
        if (per_unit == 0) return;                     // (every column constant: nothing to count)
        sp = item / per_unit;
        if (sp >= units_sp) return;                    // (uniform: past the last workgroup there is)
        int r = item - sp * per_unit, base = 0;
        for (int c0 = 0; c0 < K && k < 0; c0 += 64) {
            const int kk = c0 + lane;
            int mine = 0;
            if (kk < K) mine = ((int)rng[2 * kk + 1] - (int)rng[2 * kk] + WAVES * TS - 1) / (WAVES * TS);
            int incl = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl_up(incl, d);
                if (lane >= d) incl += up;
            }
            // the column whose range of workgroups holds r: first lane with base + RR * incl > r
            const unsigned long long hit = __ballot((base + incl) * RR > r);
            if (hit) {
                const int l = __builtin_ctzll(hit);
                k = c0 + l;
                nq = __shfl(mine, l);
                before_k = (base + __shfl(incl, l) - nq) * RR;
            }
            base += __shfl(incl, 63);
        }
        r -= before_k;
        vmin = (int)rng[2 * k];
        T = (int)rng[2 * k + 1] - vmin;
        item = r;                                      // = rr * nq + quad
    }
    const int rr = item / nq, quad = item - rr * nq;
    // TS > 1: the column's thresholds evenly over its workgroups (every wave of every workgroup the same count, +-1).  TS == 1 (a
    // wave has a threshold or has none): full workgroups first, the rest in the last one -- a workgroup costs its busiest SIMD's
    // waves, and 8 + 8 + 8 + 8 + 1 is cheaper than 7 + 7 + 7 + 6 + 6
    const int q0 = TS == 1 ? quad * WAVES : quad * (T / nq) + (quad < T % nq ? quad : T % nq);
    const int mine_all = TS == 1 ? (T - q0 < WAVES ? T - q0 : WAVES) : T / nq + (quad < T % nq ? 1 : 0);
    const int per = (mine_all + WAVES - 1) / WAVES;
    const int slot0 = q0 + wave * per;                                   // first threshold of this wave, relative to vmin
    const int left = mine_all - wave * per;
    const int nthr = __builtin_amdgcn_readfirstlane(left < per ? (left > 0 ? left : 0) : per);
    const int2 PQ = gram_pair<SHAPE>(sp_base + sp, ns, paired);
    // this lane's DMA sources: rows 16 (wave + 4 g) + lane % 16 of the stage, chunk lane / 16
    const uint8_t* src[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int srow = 16 * (wave + WAVES * g) + (lane & 15);          // row of the stage: operand set srow / 32
        int row = ((srow < 64 ? PQ.x : PQ.y) << 6) + (srow & 63);
        row = row < n ? row : n - 1;  // rows beyond n: any valid row, their counts are never read
        src[g] = leaf + (((size_t)row * K + k) << p) + (size_t)rr * (size_t)len + (size_t)(lane >> 4) * 16;
    }
    const uint32_t thr0 = (uint32_t)(0x80 + vmin + slot0) * 0x01010101u;
    uint32_t* out = part + ((((size_t)sp * K + k) * RR + rr) * slots + slot0) * (size_t)(NB * 1024) + lane;
    const int nst = len >> 6;
    switch (nthr) {
#define DD_GRAM_CASE(NT) \
    case NT: gram_body<SHAPE, (NT <= TS ? NT : 0)>(src, ring, ring_lds, wave, lane, nst, thr0, out); break;   // (NT > TS never occurs)
        DD_GRAM_CASE(1) DD_GRAM_CASE(2) DD_GRAM_CASE(3) DD_GRAM_CASE(4)
#undef DD_GRAM_CASE
        default: gram_body<SHAPE, 0>(src, ring, ring_lds, wave, lane, nst, thr0, out); break;
    }
}

// One workgroup per (unit, k, 32 x 32 block, accumulator register): the 64 entries one accumulator register
// holds across the wave's lanes.  Wave w sums the partial counts of thresholds w, w + 4, ... over the register ranges
// (256-byte reads in the layout the Gram kernel stored), the sums are transposed through LDS, and every entry's
// cumulative counts are differenced into its histogram row hist[((i * n) + j) * K + k][64] (256-byte writes).
__global__ __launch_bounds__(256) void gram_finish_kernel(const uint32_t* __restrict__ part, int n, int K, int p,
                                                          const uint32_t* __restrict__ rng, int shape, int sp_base, int ns,
                                                          int RR, int slots, int paired, uint32_t* __restrict__ hist) {
    __shared__ uint32_t F[64][65];   // [entry][threshold]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int NB = shape == kDiag ? 3 : shape == kOff ? 4 : 10;
    int b = blockIdx.x;
    const int reg = b & 15;
    b >>= 4;
    const int block = b % NB;
    b /= NB;
    const int k = b % K;
    const int sp = b / K;
    const int2 PQ = shape == kDiag ? gram_pair<kDiag>(sp_base + sp, ns) : shape == kOff ? gram_pair<kOff>(sp_base + sp, ns, paired) : gram_pair<kDiag2>(sp_base + sp, ns);
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
    const int bi = shape == kDiag ? (block == 2) : shape == kOff ? (block >> 1) : diag2_bi(block);
    const int bj = shape == kDiag ? (block >= 1) : shape == kOff ? (block & 1) : diag2_bj(block);
    const int jbase = (shape == kOff ? PQ.y : PQ.x) << 6;   // (kDiag2: quarters 0..3 of the 128 rows that start at super-block PQ.x)
    for (int e = wave; e < 64; e += 4) {
        // entry e of accumulator register `reg`: row (reg & 3) + 8 (reg >> 2) + 4 (e >> 5), column e & 31 of the block
        const int i = (PQ.x << 6) + bi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (e >> 5);
        const int j = jbase + bj * 32 + (e & 31);
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

// n > 64: the diagonal is covered by 128-row units (kDiag2) and the off-diagonal launch skips the 64 x 64 pairs inside them
static bool gram_diag2(int n) { return n > 64; }

size_t gram_scratch_bytes(int n, int K, int p, int* sp_per_launch) {
    const int ns = (n + 63) / 64;
    const bool d2 = gram_diag2(n);
    const size_t total_sp = (size_t)ns * (ns + 1) / 2;
    const size_t m = (size_t)1 << p;
    const size_t RR = m / (size_t)gram_len(d2 ? 4 * ((ns + 1) / 2) : ns, K, p);   // (the diagonal launch has the fewest units: its split is the finest)
    // a launch covers as many units as fit ~1 GiB of partial counts (at least one)
    const size_t per_sp = (size_t)K * RR * (size_t)gram_slots(p) * (d2 ? 10 : 4) * 1024 * sizeof(uint32_t);
    size_t fit = ((size_t)1 << 30) / per_sp;
    fit = fit < 1 ? 1 : fit > total_sp ? total_sp : fit;
    if (sp_per_launch) *sp_per_launch = (int)fit;
    return fit * per_sp + (((size_t)K * 2 * sizeof(uint32_t) + 255) & ~(size_t)255);
}

// hist_dev must have been zeroed by the caller (the lower triangle stays zero).  scratch: gram_scratch_bytes().
void launch_pairwise_gram(const uint8_t* leaf_dev, int n, int K, int p, uint32_t* hist_dev, void* scratch,
                          hipStream_t st) {
    const int ns = (n + 63) / 64;
    const bool d2 = gram_diag2(n);
    const size_t m = (size_t)1 << p;
    int sp_fit = 1;
    (void)gram_scratch_bytes(n, K, p, &sp_fit);
    uint8_t* base = static_cast<uint8_t*>(scratch);
    uint32_t* rng = reinterpret_cast<uint32_t*>(base);
    uint32_t* part = reinterpret_cast<uint32_t*>(base + (((size_t)K * 2 * sizeof(uint32_t) + 255) & ~(size_t)255));

    launch_register_range(leaf_dev, n, K, p, rng, st);
    // the diagonal units, then the pairs P < Q; sp_fit of them per launch
    const int slots = gram_slots(p);
    const int len = gram_len(d2 ? 4 * ((ns + 1) / 2) : ns, K, p);
    const int RR = (int)(m / (size_t)len);
    for (int diag = 1; diag >= 0; --diag) {
        const int shape = diag ? (d2 ? kDiag2 : kDiag) : kOff;
        const int total = shape == kDiag ? ns : shape == kDiag2 ? (ns + 1) / 2 : ns * (ns - 1) / 2 - (d2 ? ns / 2 : 0);
        const int ts = shape == kDiag ? GramShape<kDiag>::TS : shape == kOff ? GramShape<kOff>::TS : GramShape<kDiag2>::TS;
        const int nb = shape == kDiag ? GramShape<kDiag>::NB : shape == kOff ? GramShape<kOff>::NB : GramShape<kDiag2>::NB;
        const int wv = shape == kDiag2 ? GramShape<kDiag2>::WAVES : 4;
        const int quads = (slots + wv * ts - 1) / (wv * ts);
        const int paired_off = d2 ? 1 : 0;
        for (int a = 0; a < total; a += sp_fit) {
            const int cnt = std::min(sp_fit, total - a);
            const int units = (int)((size_t)cnt * K * RR * quads);
            const unsigned grid = (unsigned)units;
#define DD_GRAM_LAUNCH(S) hipLaunchKernelGGL((gram_kernel<S, GramShape<S>::TS>), dim3(grid), dim3(GramShape<S>::WAVES * 64), 0, st, leaf_dev, n, K, p, rng, a, ns, RR, len, slots, units, cnt, (S == kOff ? paired_off : 0), part)
            if (shape == kDiag) DD_GRAM_LAUNCH(kDiag); else if (shape == kDiag2) DD_GRAM_LAUNCH(kDiag2); else DD_GRAM_LAUNCH(kOff);
#undef DD_GRAM_LAUNCH
            hipLaunchKernelGGL(gram_finish_kernel, dim3((unsigned)((size_t)cnt * K * nb * 16)), dim3(256), 0, st, part, n, K, p, rng,
                               shape, a, ns, RR, slots, d2 ? 1 : 0, hist_dev);
        }
    }
}

}  // namespace dd
