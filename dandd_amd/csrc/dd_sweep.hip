// dd_sweep.hip -- K1: fused k-sweep HyperLogLog sketch over the 2-bit token stream.
//
// Replaces  parallel -j 95% ' dashing sketch -k{} -S <p> --prefix <dir> <fasta> ' ::: kmin..kmax
// (/root/reference/lib/huffman_dandd.py:214-218, /root/reference/lib/sketch_classes.py:351-366):
// instead of one process per k, each re-reading and re-parsing the FASTA, one launch walks the
// token stream once per k-group, with the group's register arrays resident in LDS.
//
// Work decomposition
//   job (one workgroup) = (genome, k-group, range of tiles); tile = blockDim.x segments of 64
//   tokens; a thread owns one segment per tile: it loads the segment's 16 B of codes + 8 B of
//   BREAK bits and the previous segment's (the halo that primes the rolling windows), then for
//   each token updates one shared forward / reverse-complement window and, for every k of the
//   group, masks/shifts the k-mer out of the windows, canonicalises, hashes (Wang 64), and
//   raises LDS register  reg[k][h >> (64-p)]  to  rho(h).
//   The LDS registers are byte-max-merged into the genome's [K][m] slab in HBM at job end.
//
// Bound: integer VALU issue (~50 instructions per (token, k)); HBM traffic is 3 bits per token
// per k-group.  No MFMA: this is hashing, not a contraction.
#include "dd_common.h"
#include "dd_kernels.h"

namespace dd {
namespace {

DD_D uint32_t ffbh(uint32_t x) {  // leading zeros; 0xFFFFFFFF for x == 0
    uint32_t r;
    asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// Exact byte-max into an LDS byte through a 32-bit compare-and-swap on its word.
DD_D void lds_byte_max(uint8_t* regs, uint32_t byte_addr, uint32_t rho) {
    uint32_t* w = reinterpret_cast<uint32_t*>(regs + (byte_addr & ~3u));
    const uint32_t sh = (byte_addr & 3u) * 8u;
    uint32_t old = *reinterpret_cast<volatile uint32_t*>(w);
    while (true) {
        uint32_t cur = (old >> sh) & 0xFFu;
        if (rho <= cur) break;
        uint32_t nw = (old & ~(0xFFu << sh)) | (rho << sh);
        uint32_t prev = atomicCAS(w, old, nw);
        if (prev == old) break;
        old = prev;
    }
}

// reg[idx] = max(reg[idx], rho(h)); the common case (no change) costs one LDS byte read.
DD_D void hll_update(uint8_t* regs, uint64_t h, int p) {
    const uint32_t idx = (uint32_t)(h >> (64 - p));
    const uint64_t hs = h << p;
    const uint32_t hiw = (uint32_t)(hs >> 32);
    const uint32_t lz = ffbh(hiw);  // rho - 1 when hiw != 0
    const uint32_t cur = *reinterpret_cast<volatile uint8_t*>(regs + idx);
    if (lz >= cur) {  // rho > cur  (lz = 0xFFFFFFFF when hiw == 0: resolved here)
        uint32_t rho = lz + 1;
        if (hiw == 0) rho = 33u + (uint32_t)__builtin_clz((uint32_t)hs | (1u << (p - 1)));
        if (rho > cur) lds_byte_max(regs, idx, rho);
    }
}

// Rolling windows, one set per thread, shared by every k of the group.
//   KC 0: k <= 16 (32-bit windows)   KC 1: k <= 32 (64-bit)   KC 2: k <= 64 (128-bit)
template <int KC>
struct Windows;

template <>
struct Windows<0> {
    uint32_t fw = 0, rc = 0;
    DD_D void push(uint32_t c) {
        fw = (fw << 2) | c;
        rc = (rc >> 2) | ((3u - c) << 30);
    }
    template <bool CANON>
    DD_D uint64_t kmer_hash_input(int k) const {
        const uint32_t f = (k == 16) ? fw : (fw & ((1u << (2 * k)) - 1u));
        if (!CANON) return f;
        const uint32_t r = rc >> (32 - 2 * k);
        return f < r ? f : r;
    }
};

template <>
struct Windows<1> {
    uint64_t fw = 0, rc = 0;
    DD_D void push(uint32_t c) {
        fw = (fw << 2) | c;
        rc = (rc >> 2) | ((uint64_t)(3u - c) << 62);
    }
    template <bool CANON>
    DD_D uint64_t kmer_hash_input(int k) const {
        const uint64_t f = (k == 32) ? fw : (fw & ((1ull << (2 * k)) - 1ull));
        if (!CANON) return f;
        const uint64_t r = rc >> (64 - 2 * k);
        return f < r ? f : r;
    }
};

template <>
struct Windows<2> {
    uint64_t fh = 0, fl = 0, rh = 0, rl = 0;
    DD_D void push(uint32_t c) {
        fh = (fh << 2) | (fl >> 62);
        fl = (fl << 2) | c;
        rl = (rl >> 2) | (rh << 62);
        rh = (rh >> 2) | ((uint64_t)(3u - c) << 62);
    }
    template <bool CANON>
    DD_D uint64_t kmer_hash_input(int k) const {  // 33 <= k <= 64
        const int hb = 2 * k - 64;                 // bits of the k-mer in the high word, 2..64
        const uint64_t ah = (hb == 64) ? fh : (fh & ((1ull << hb) - 1ull));
        const uint64_t al = fl;
        if (!CANON) return fold128(ah, al);
        const int s = 128 - 2 * k;  // 0..62
        const uint64_t bh = s ? (rh >> s) : rh;
        const uint64_t bl = s ? ((rl >> s) | (rh << (64 - s))) : rl;
        const bool f_lt = (ah < bh) || (ah == bh && al < bl);
        return fold128(f_lt ? ah : bh, f_lt ? al : bl);
    }
};

// GLOBAL = false: registers of the group live in LDS (2^p * nk bytes <= 160 KiB).
// GLOBAL = true : registers are updated in place in the genome's HBM slab (p >= 18, where
//                 one array no longer fits LDS); same arithmetic, L2-resident read-compare
//                 and a 32-bit CAS only when a register rises.
template <int KC, bool CANON, bool GLOBAL>
__global__ __launch_bounds__(1024) void sweep_kernel(const SweepGenome* __restrict__ genomes,
                                                    const SweepJob* __restrict__ jobs, int p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const SweepJob job = jobs[blockIdx.x];
    const SweepGenome g = genomes[job.genome];
    const int nk = job.nk, kfirst = job.kfirst;
    const uint32_t m = 1u << p;
    const unsigned long long ntok = *g.ntok;

    uint8_t* const regs0 = GLOBAL ? g.regs + ((size_t)job.krow << p) : lds;
    // zero the group's registers
    if (!GLOBAL) {
        uint4* z = reinterpret_cast<uint4*>(lds);
        const uint32_t n16 = (uint32_t)nk * (m >> 4);
        for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x) z[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();

    const int prime = kfirst + nk - 2;  // (largest k of the group) - 1 halo tokens prime the windows
    const uint4* codes4 = reinterpret_cast<const uint4*>(g.codes);
    const uint2* bad2 = reinterpret_cast<const uint2*>(g.bad);

    for (unsigned tile = job.tile_begin; tile < job.tile_end; ++tile) {
        const unsigned long long seg = (unsigned long long)tile * blockDim.x + threadIdx.x;
        if (seg * kSegTokens >= ntok) continue;
        Windows<KC> win;
        int run = 0;
        if (seg > 0) {
            const uint4 hc = codes4[seg - 1];
            const uint2 hb = bad2[seg - 1];
            const uint32_t cw[4] = {hc.x, hc.y, hc.z, hc.w};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const uint32_t bw = ((w & 2) ? hb.y : hb.x) >> ((w & 1) * 16);
#pragma unroll 1
                for (int i = 0; i < 16; ++i) {
                    if (w * 16 + i < kSegTokens - prime) continue;
                    const uint32_t c = (cw[w] >> (2 * i)) & 3u;
                    run = ((bw >> i) & 1u) ? 0 : run + 1;
                    win.push(c);
                }
            }
        }
        const uint4 sc = codes4[seg];
        const uint2 sb = bad2[seg];
        const uint32_t cw[4] = {sc.x, sc.y, sc.z, sc.w};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t bw = ((w & 2) ? sb.y : sb.x) >> ((w & 1) * 16);
#pragma unroll 1
            for (int i = 0; i < 16; ++i) {
                const uint32_t c = (cw[w] >> (2 * i)) & 3u;
                run = ((bw >> i) & 1u) ? 0 : run + 1;
                win.push(c);
#pragma unroll 1
                for (int j = 0; j < nk; ++j) {
                    const int k = kfirst + j;
                    if (run >= k) {
                        const uint64_t x = win.template kmer_hash_input<CANON>(k);
                        hll_update(regs0 + ((size_t)j << p), wang64(x), p);
                    }
                }
            }
        }
    }
    __syncthreads();

    // merge the group's registers into the genome's slab (rows krow .. krow+nk-1 are contiguous)
    if (!GLOBAL) {
        const uint4* l4 = reinterpret_cast<const uint4*>(lds);
        uint32_t* gw = reinterpret_cast<uint32_t*>(g.regs + ((size_t)job.krow << p));
        const uint32_t n16 = (uint32_t)nk * (m >> 4);
        for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x) {
            const uint4 lv = l4[i];
            const uint4 gv = reinterpret_cast<const uint4*>(gw)[i];
            const uint32_t l[4] = {lv.x, lv.y, lv.z, lv.w};
            const uint32_t o[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t old = o[q];
                uint32_t mx = bmax4(old, l[q]);
                while (mx != old) {
                    uint32_t prev = atomicCAS(&gw[4 * i + q], old, mx);
                    if (prev == old) break;
                    old = prev;
                    mx = bmax4(old, l[q]);
                }
            }
        }
    }
}

template <int KC, bool CANON, bool GLOBAL>
void launch_one(const SweepGenome* genomes, const SweepJob* jobs, int njobs, const SweepPlan& plan,
                hipStream_t st) {
    auto kern = sweep_kernel<KC, CANON, GLOBAL>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, sweep_max_lds_bytes());
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)njobs), dim3((unsigned)plan.threads),
                       (size_t)plan.lds_bytes, st, genomes, jobs, plan.log2m);
}

}  // namespace

int sweep_max_lds_bytes() { return 160 * 1024; }

void launch_sweep(const SweepGenome* genomes, const SweepJob* jobs, int njobs, int kclass,
                  const SweepPlan& plan, hipStream_t st) {
    if (njobs <= 0) return;
#define DD_DISPATCH(KC, CN, GL) launch_one<KC, CN, GL>(genomes, jobs, njobs, plan, st)
#define DD_DISPATCH_KC(CN, GL)                \
    do {                                      \
        if (kclass == 0) DD_DISPATCH(0, CN, GL);      \
        else if (kclass == 1) DD_DISPATCH(1, CN, GL); \
        else DD_DISPATCH(2, CN, GL);                  \
    } while (0)
    const bool gl = plan.lds_bytes == 0;
    if (plan.canonical) {
        if (gl) DD_DISPATCH_KC(true, true); else DD_DISPATCH_KC(true, false);
    } else {
        if (gl) DD_DISPATCH_KC(false, true); else DD_DISPATCH_KC(false, false);
    }
}

}  // namespace dd
