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
// Bound: integer VALU issue.  Measured issue costs on gfx950 (scripts/ubench.hip, 4 waves/SIMD,
// 2.34 GHz): v_xor/and/or/not/mov/add/sub/lshrrev_b32 ~2.5 cycles per wave64 instruction;
// everything else used here (v_lshlrev_b32, v_alignbit, v_mul_lo_u32, v_ffbh, v_cmp, v_cndmask and
// every 64-bit op: v_lshrrev_b64, v_lshl_add_u64, v_mad_u64_u32, v_cmp_lt_u64) ~4.2 cycles.  A 64-bit
// instruction therefore costs the same as one 32-bit shift, so the hash below is written in 64-bit
// instructions and 32-bit work is steered to the cheap class.  HBM traffic is 3 bits per token
// per k-group.  No MFMA: this is hashing, not a contraction.
#include "dd_common.h"
#include "dd_kernels.h"

namespace dd {
namespace {

extern __shared__ __attribute__((aligned(16))) uint8_t g_lds[];

DD_D uint32_t ffbh(uint32_t x) {  // leading zeros; 0xFFFFFFFF for x == 0
    uint32_t r;
    asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
DD_D uint32_t mul_lo(uint32_t a, uint32_t c) {  // opaque to the optimiser: stays one v_mul_lo_u32
    uint32_t r;
    asm("v_mul_lo_u32 %0, %1, %2" : "=v"(r) : "v"(a), "s"(c));
    return r;
}
template <int SH>
DD_D uint64_t lshl_add64(uint64_t a, uint64_t b) {  // (a << SH) + b, SH in 0..4, one instruction
    uint64_t r;
    asm("v_lshl_add_u64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "n"(SH), "v"(b));
    return r;
}
// x * C + addend (mod 2^64), C a 32-bit constant: v_mad_u64_u32 + v_mul_lo_u32 + v_add_u32
template <bool HI_ZERO>
DD_D uint64_t mul64_c32(uint64_t x, uint32_t C, uint64_t addend) {
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    const uint64_t pr = (uint64_t)lo * C + addend;
    if (HI_ZERO) return pr;
    const uint32_t ph = (uint32_t)(pr >> 32) + mul_lo(hi, C);
    return ((uint64_t)ph << 32) | (uint32_t)pr;
}

// Thomas Wang 64-bit mix, identical to dd::wang64 (asserted on every lane by the GPU parity tests),
// arranged for the gfx950 issue costs above: 18 instructions.
template <bool HI_ZERO>
DD_D uint64_t wang64_fast(uint64_t x) {
    x = mul64_c32<HI_ZERO>(x, 0x1FFFFFu, ~0ull);  // ~x + (x << 21) = x * (2^21 - 1) - 1
    x ^= x >> 24;
    x = mul64_c32<false>(x, 265u, 0ull);          // x + (x << 3) + (x << 8)
    x ^= x >> 14;
    x = lshl_add64<2>(lshl_add64<2>(x, x), x);    // x + (x << 2) + (x << 4) = ((5x) << 2) + x
    x ^= x >> 28;
    return lshl_add64<0>(x << 31, x);             // x + (x << 31)
}

// ---- register stores -------------------------------------------------------------------------
// LDS: byte registers, 32-bit compare-and-swap on the containing word when a register must rise.
struct RegsLds {
    uint32_t base;  // byte offset of the slot in g_lds
    DD_D uint32_t load8(uint32_t i) const { return g_lds[base + i]; }
    DD_D uint32_t load32(uint32_t i) const {
        return reinterpret_cast<const uint32_t*>(g_lds)[(base + i) >> 2];  // plain ds_read_b32
    }
    DD_D uint32_t cas32(uint32_t i, uint32_t expect, uint32_t desired) const {
        return atomicCAS(reinterpret_cast<uint32_t*>(g_lds + base + i), expect, desired);
    }
};
// HBM/L2: same protocol on the genome's slab (p >= 18: one array no longer fits LDS).
struct RegsGlobal {
    uint8_t* base;
    DD_D uint32_t load8(uint32_t i) const {
        return __hip_atomic_load(base + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    DD_D uint32_t load32(uint32_t i) const {
        return __hip_atomic_load(static_cast<uint32_t*>(__builtin_assume_aligned(base + i, 4)),
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    DD_D uint32_t cas32(uint32_t i, uint32_t expect, uint32_t desired) const {
        return atomicCAS(reinterpret_cast<uint32_t*>(base + i), expect, desired);
    }
};

// Exact byte-max into register idx.
template <typename R>
DD_D void reg_raise(const R& regs, uint32_t idx, uint32_t rho) {
    const uint32_t wi = idx & ~3u, sh = (idx & 3u) * 8u;
    uint32_t old = regs.load32(wi);
    while (true) {
        const uint32_t cur = (old >> sh) & 0xFFu;
        if (rho <= cur) break;
        const uint32_t nw = (old & ~(0xFFu << sh)) | (rho << sh);
        const uint32_t prev = regs.cas32(wi, old, nw);
        if (prev == old) break;
        old = prev;
    }
}

// idx = h >> (64-p) and lz = rho(h) - 1 (0xFFFFFFFF when the top 32 bits of h << p are all zero)
struct Probe {
    uint32_t idx, lz, hiw, lo;
};
DD_D Probe probe(uint64_t h, int p) {
    const uint32_t hi = (uint32_t)(h >> 32), lo = (uint32_t)h;
    Probe r;
    r.idx = hi >> (32 - p);
    r.hiw = __builtin_amdgcn_alignbit(hi, lo, 32 - p);  // bits 63..32 of (h << p)
    r.lz = ffbh(r.hiw);
    r.lo = lo;
    return r;
}
// the rare path: the register (last seen as cur) may have to rise
template <typename R>
DD_D void raise_checked(const R& regs, const Probe& q, uint32_t cur, int p) {
    uint32_t rho = q.lz + 1;
    if (q.hiw == 0) rho = 33u + (uint32_t)__builtin_clz((q.lo << p) | (1u << (p - 1)));
    if (rho > cur) reg_raise(regs, q.idx, rho);
}
// reg[h >> (64-p)] = max(., rho(h)); the common case (no change) is one byte read + compare.
template <typename R>
DD_D void hll_update(const R& regs, uint64_t h, int p) {
    const Probe q = probe(h, p);
    const uint32_t cur = regs.load8(q.idx);
    if (q.lz >= cur) raise_checked(regs, q, cur, p);  // rho > cur, or hiw == 0 (resolved there)
}
// two independent updates interleaved: both hash chains and both LDS reads are in flight
// together, one wave-level branch covers the common no-change case of both
template <typename R>
DD_D void hll_update2(const R& r0, uint64_t h0, const R& r1, uint64_t h1, int p) {
    const Probe a = probe(h0, p), b = probe(h1, p);
    const uint32_t c0 = r0.load8(a.idx), c1 = r1.load8(b.idx);
    if ((a.lz >= c0) | (b.lz >= c1)) {
        if (a.lz >= c0) raise_checked(r0, a, c0, p);
        if (b.lz >= c1) raise_checked(r1, b, c1, p);
    }
}

// ---- rolling windows, one set per thread, shared by every k of the group ------------------------
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
    DD_D uint64_t hash(int k) const {
        const uint32_t f = (k == 16) ? fw : (fw & ((1u << (2 * k)) - 1u));
        if (!CANON) return wang64_fast<true>(f);
        const uint32_t r = rc >> (32 - 2 * k);
        return wang64_fast<true>(f < r ? f : r);
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
    DD_D uint64_t hash(int k) const {
        const uint64_t f = (k == 32) ? fw : (fw & ((1ull << (2 * k)) - 1ull));
        if (!CANON) return wang64_fast<false>(f);
        const uint64_t r = rc >> (64 - 2 * k);
        return wang64_fast<false>(f < r ? f : r);
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
    DD_D uint64_t hash(int k) const {  // 33 <= k <= 64
        const int hb = 2 * k - 64;    // bits of the k-mer in the high word, 2..64
        const uint64_t ah = (hb == 64) ? fh : (fh & ((1ull << hb) - 1ull));
        const uint64_t al = fl;
        if (!CANON) return wang64_fast<false>(fold128(ah, al));
        const int s = 128 - 2 * k;  // 0..62
        const uint64_t bh = s ? (rh >> s) : rh;
        const uint64_t bl = s ? ((rl >> s) | (rh << (64 - s))) : rl;
        const bool f_lt = (ah < bh) || (ah == bh && al < bl);
        return wang64_fast<false>(fold128(f_lt ? ah : bh, f_lt ? al : bl));
    }
};

// every k of the group for the token just pushed
template <int KC, bool CANON, bool CHECK, typename MakeRegs>
DD_D void sweep_token(const Windows<KC>& win, int run, int kfirst, int nk, int p, const MakeRegs& slot) {
    // ks ascend, so a lane whose run is too short for k is also too short for every later k
    int j = 0;
#pragma unroll 1
    for (; j + 1 < nk; j += 2) {
        const int k = kfirst + j;
        if (CHECK && run < k + 1) break;
        hll_update2(slot(j), win.template hash<CANON>(k), slot(j + 1), win.template hash<CANON>(k + 1), p);
    }
    if (j < nk && (!CHECK || run >= kfirst + j)) hll_update(slot(j), win.template hash<CANON>(kfirst + j), p);
}

// GLOBAL = false: registers of the group live in LDS (2^p * nk bytes <= 160 KiB).
// GLOBAL = true : registers are updated in place in the genome's HBM slab.
template <int KC, bool CANON, bool GLOBAL>
__global__ __launch_bounds__(1024) void sweep_kernel(const SweepGenome* __restrict__ genomes,
                                                    const SweepJob* __restrict__ jobs, int p) {
    const SweepJob job = jobs[blockIdx.x];
    const SweepGenome g = genomes[job.genome];
    const int nk = job.nk, kfirst = job.kfirst;
    const uint32_t m = 1u << p;
    const unsigned long long ntok = *g.ntok;
    uint8_t* const slab = g.regs + ((size_t)job.krow << p);

    if (!GLOBAL) {
        // Warm start: begin from whatever earlier jobs have already merged into the slab.  Any
        // (possibly stale) snapshot is a valid lower bound of the final registers, and a warm
        // array makes the "register rises" path rare: after T tokens have been absorbed only
        // ~m/T of the updates still raise a register.
        uint4* z = reinterpret_cast<uint4*>(g_lds);
        const uint4* src = reinterpret_cast<const uint4*>(slab);
        const uint32_t n16 = (uint32_t)nk * (m >> 4);
        for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x) z[i] = src[i];
    }
    __syncthreads();

    const int kmaxg = kfirst + nk - 1;
    const int prime = kmaxg - 1;  // halo tokens that prime the windows
    const uint4* codes4 = reinterpret_cast<const uint4*>(g.codes);
    const uint2* bad2 = reinterpret_cast<const uint2*>(g.bad);
    auto lds_slot = [p](int j) { return RegsLds{(uint32_t)j << p}; };
    auto glb_slot = [slab, p](int j) { return RegsGlobal{slab + ((size_t)j << p)}; };

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
                // wave-uniform fast path: no lane of the wave is within kmaxg tokens of a BREAK
                if (__all(run >= kmaxg)) {
                    if (GLOBAL) sweep_token<KC, CANON, false>(win, run, kfirst, nk, p, glb_slot);
                    else sweep_token<KC, CANON, false>(win, run, kfirst, nk, p, lds_slot);
                } else {
                    if (GLOBAL) sweep_token<KC, CANON, true>(win, run, kfirst, nk, p, glb_slot);
                    else sweep_token<KC, CANON, true>(win, run, kfirst, nk, p, lds_slot);
                }
            }
        }
    }
    __syncthreads();

    // merge the group's registers into the genome's slab (rows krow .. krow+nk-1 are contiguous)
    if (!GLOBAL) {
        const uint4* l4 = reinterpret_cast<const uint4*>(g_lds);
        uint32_t* gw = reinterpret_cast<uint32_t*>(slab);
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
#define DD_DISPATCH_KC(CN, GL)                        \
    do {                                              \
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
