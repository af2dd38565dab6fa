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

#include <atomic>

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
// kernels that address LDS absolutely (RegsLds, scatter_update's filter read) call this first
DD_D void lds_starts_at_zero() {
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)g_lds != 0u) __builtin_trap();
}

// LDS: byte registers, 32-bit compare-and-swap on the containing word when a register must rise.
struct RegsLds {
    uint32_t slot;  // the slot's registers start at byte slot << p of g_lds
    using Addr = uint32_t;
    // address of register  hi >> (32-p)  (the top p bits of the hash): one v_alignbit of slot:hi
    DD_D Addr at(uint32_t hi, int p) const { return __builtin_amdgcn_alignbit(slot, hi, 32 - p); }
    DD_D static uint32_t shift(Addr a) { return (a & 3u) * 8u; }
    // Registers are read at their ABSOLUTE LDS address: every kernel that uses this struct has no static LDS, so the
    // dynamic array g_lds starts at 0 (lds_starts_at_zero() at the top of each checks it), and indexing through the
    // g_lds symbol would cost a `v_add_u32 v, 0, v` of its link-time address on every read -- 1.5 of the 31.5 VALU
    // instructions of a k 17..32 update at log2m <= 16.
    DD_D uint32_t bound(Addr a) const { return *(const __attribute__((address_space(3))) uint8_t*)(uintptr_t)a; }  // the register itself
    DD_D static uint32_t load32(Addr a) { return *(const __attribute__((address_space(3))) uint32_t*)(uintptr_t)(a & ~3u); }
    DD_D static uint32_t cas32(Addr a, uint32_t expect, uint32_t desired) {
        return atomicCAS(reinterpret_cast<uint32_t*>(g_lds + (a & ~3u)), expect, desired);
    }
};
// HBM/L2: same protocol on the genome's slab (p >= 18: one array no longer fits LDS).
struct RegsGlobal {
    uint8_t* base;  // 16-byte aligned
    using Addr = uint8_t*;
    DD_D Addr at(uint32_t hi, int p) const { return base + (hi >> (32 - p)); }
    DD_D static uint32_t shift(Addr a) { return ((uint32_t)(uintptr_t)a & 3u) * 8u; }
    DD_D static uint8_t* word(Addr a) {
        return static_cast<uint8_t*>(__builtin_assume_aligned(a - ((uintptr_t)a & 3u), 4));
    }
    DD_D uint32_t bound(Addr a) const { return gload1_fresh(a); }
    DD_D static uint32_t load32(Addr a) { return gload4_fresh(word(a)); }
    DD_D static uint32_t cas32(Addr a, uint32_t expect, uint32_t desired) { return gcas32(word(a), expect, desired); }
};

// 16 bytes of global memory as other agents' atomics left them (two relaxed agent-scope 8-byte
// loads: they bypass this XCD's non-coherent L2).
DD_D uint4 load16_fresh(const uint8_t* p) {
    const uint8_t* q = static_cast<const uint8_t*>(__builtin_assume_aligned(p, 16));
    const unsigned long long a = gload8_fresh(q), b = gload8_fresh(q + 8);
    return make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
}
DD_D uint32_t load4_fresh(const uint32_t* p) { return gload4_fresh(p); }

// hi = top word of h (its top p bits index the register) and lz = rho(h) - 1 (0xFFFFFFFF when the
// top 32 bits of h << p are all zero)
struct Probe {
    uint32_t hi, lz, hiw, lo;
};
DD_D Probe probe(uint64_t h, int p) {
    const uint32_t hi = (uint32_t)(h >> 32), lo = (uint32_t)h;
    Probe r;
    r.hi = hi;
    r.hiw = __builtin_amdgcn_alignbit(hi, lo, 32 - p);  // bits 63..32 of (h << p)
    r.lz = ffbh(r.hiw);
    r.lo = lo;
    return r;
}
// rho(h) from a probe: lz + 1, in its long form only when the 32 bits after the index are all zero
// (p = 2^-32: behind a wave-level branch)
DD_D uint32_t rho_of(const Probe& q, int p) {
    uint32_t rho = q.lz + 1;  // 0 where hiw == 0
    if (__builtin_expect(__any(q.hiw == 0), 0)) {
        if (q.hiw == 0) rho = 33u + (uint32_t)__builtin_clz((q.lo << p) | (1u << (p - 1)));
    }
    return rho;
}
// Exact byte-max of rho into the register at `a` through a 32-bit CAS on the containing word,
// starting from the word value `old`; returns the register's value afterwards.  The byte is raised by
// ADDING (rho - cur) << shift: no carry can leave the byte.
template <typename R>
DD_D uint32_t cas_raise(typename R::Addr a, uint32_t old, uint32_t rho) {
    const uint32_t sh = R::shift(a);
    while (true) {
        const uint32_t cur = (old >> sh) & 0xFFu;
        if (rho <= cur) return cur;
        const uint32_t prev = R::cas32(a, old, old + ((rho - cur) << sh));
        if (prev == old) return rho;
        old = prev;
    }
}
// The rare path: the register at `a` was seen below rho.  Every instruction here is paid by the
// whole wave for (typically) one lane, so it is kept short.
template <typename R>
DD_D void raise(const R&, typename R::Addr a, const Probe& q, int p) {
    (void)cas_raise<R>(a, R::load32(a), rho_of(q, p));
}

// (log2m >= 17: the registers stay in HBM and are reached through record streams -- scatter + sort + replay, below)
constexpr uint32_t kQueueEntries = 128;  // per wave and queue (scatter_kernel); a push adds <= 64 to < 64 waiting
DD_D uint32_t min4(uint32_t w) {  // smallest byte
    const uint32_t a = w & 0xFFu, b = (w >> 8) & 0xFFu, c = (w >> 16) & 0xFFu, d = w >> 24;
    const uint32_t ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}
// reg[h >> (64-p)] = max(., rho(h)); the common case (no change) is one byte read + compare.
template <typename R>
DD_D void hll_update(const R& regs, uint64_t h, int p) {
    const Probe q = probe(h, p);
    const typename R::Addr a = regs.at(q.hi, p);
    if (q.lz >= regs.bound(a)) raise(regs, a, q, p);  // rho > bound, or hiw == 0 (resolved there)
}
// two independent updates interleaved: both hash chains and both LDS reads are in flight
// together, one wave-level branch covers the common no-change case of both
template <typename R>
DD_D void hll_update2(const R& r0, uint64_t h0, const R& r1, uint64_t h1, int p) {
    const Probe qa = probe(h0, p), qb = probe(h1, p);
    const typename R::Addr a = r0.at(qa.hi, p), b = r1.at(qb.hi, p);
    const uint32_t c0 = r0.bound(a), c1 = r1.bound(b);
    if ((qa.lz >= c0) | (qb.lz >= c1)) {
        if (qa.lz >= c0) raise(r0, a, qa, p);
        if (qb.lz >= c1) raise(r1, b, qb, p);
    }
}

// Reverse the order of the 16 2-bit fields of a code word: the token stream stores token j at bits
// [2j, 2j+1] (oldest lowest), the forward window wants the newest token lowest.
DD_D uint32_t pairrev32(uint32_t x) {
    x = __builtin_bitreverse32(x);  // v_bfrev_b32: fields reversed, but so are the two bits inside each
    return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
}
DD_D uint64_t pack64(uint32_t hi, uint32_t lo) { return ((uint64_t)hi << 32) | lo; }

// ---- rolling windows, one set per thread, shared by every k of the group ------------------------
// prime(hc): the state after pushing the 64 tokens of the previous segment (hc = its four code
// words), computed with a handful of bit operations instead of 64 pushes.  The reverse-complement
// window holds tokens in stream order, newest on top, so it is simply the complement of the words.
//   KC 0: k <= 16 (32-bit windows)   KC 1: 16 <= k <= 32 (64-bit)   KC 3: 33 <= k <= 48 (96-bit: 64 + 32)
//   KC 2: 49 <= k <= 64 (128-bit: Windows<7>, four 32-bit words)
template <int KC>
struct Windows;

template <>
struct Windows<0> {
    uint32_t fw = 0, rc = 0;
    DD_D void prime(const uint4& hc) {
        fw = pairrev32(hc.w);
        rc = ~hc.w;
    }
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
    DD_D void prime(const uint4& hc) {
        fw = pack64(pairrev32(hc.z), pairrev32(hc.w));
        rc = ~pack64(hc.w, hc.z);
    }
    DD_D void push(uint32_t c) {
        fw = (fw << 2) | c;
        rc = (rc >> 2) | ((uint64_t)(3u - c) << 62);
    }
    template <bool CANON>
    DD_D uint64_t hash(int k) const {
        // 16 <= k <= 32: the low word of the window belongs to the k-mer whole, only the high word is masked
        const uint32_t mhi = (k == 32) ? ~0u : ((1u << (2 * k - 32)) - 1u);
        const uint64_t f = pack64((uint32_t)(fw >> 32) & mhi, (uint32_t)fw);
        if (!CANON) return wang64_fast<false>(f);
        const uint64_t r = rc >> (64 - 2 * k);
        return wang64_fast<false>(f < r ? f : r);
    }
};

// The 64-bit class again, as 32-bit halves: a push is two funnel shifts and two shift-or instructions, and the
// compiler has no 64-bit value to keep a copy of (with u64 members it spent three more instructions per push).
// Used by the scatter kernels, where the push is paid per (token, k); sweep_kernel shares one push between the
// ks of a group and keeps Windows<1> (measured equal there).
template <>
struct Windows<5> {
    uint32_t fl = 0, fh = 0, rl = 0, rh = 0;
    DD_D void prime(const uint4& hc) {
        fh = pairrev32(hc.z);
        fl = pairrev32(hc.w);
        rh = ~hc.w;
        rl = ~hc.z;
    }
    DD_D void push(uint32_t c) {
        fh = __builtin_amdgcn_alignbit(fh, fl, 30);  // (fw << 2) high word
        fl = (fl << 2) | c;
        rl = __builtin_amdgcn_alignbit(rh, rl, 2);   // (rc >> 2) low word
        rh = (rh >> 2) | ((3u - c) << 30);
    }
    template <bool CANON>
    DD_D uint64_t hash(int k) const {
        const uint32_t mhi = (k == 32) ? ~0u : ((1u << (2 * k - 32)) - 1u);
        const uint64_t f = pack64(fh & mhi, fl);
        if (!CANON) return wang64_fast<false>(f);
        const uint64_t r = pack64(rh, rl) >> (64 - 2 * k);
        return wang64_fast<false>(f < r ? f : r);
    }
};

// 33 <= k <= 48: the k-mer is 66..96 bits, so the high part fits one 32-bit register and every
// step on it (mask, funnel shift of the reverse complement, compare, select, fold multiply) is a
// 32-bit instruction instead of a 64-bit pair.
template <>
struct Windows<3> {
    uint64_t fl = 0, rt = 0;  // forward: low 64 bits;  reverse complement, top-aligned: bits 95..32
    uint32_t fh = 0, rb = 0;  // forward: bits 95..64;  reverse complement: bits 31..0
    DD_D void prime(const uint4& hc) {
        fl = pack64(pairrev32(hc.z), pairrev32(hc.w));
        fh = pairrev32(hc.y);
        rt = ~pack64(hc.w, hc.z);
        rb = ~hc.y;
    }
    DD_D void push(uint32_t c) {
        fh = __builtin_amdgcn_alignbit(fh, (uint32_t)(fl >> 32), 30);  // (fh << 2) | (fl >> 62)
        fl = (fl << 2) | c;
        rb = __builtin_amdgcn_alignbit((uint32_t)rt, rb, 2);           // (rb >> 2) | (rt << 30)
        rt = (rt >> 2) | ((uint64_t)(3u - c) << 62);
    }
    template <bool CANON>
    DD_D uint64_t hash(int k) const {
        const int hb = 2 * k - 64;  // 2..32 bits of the k-mer above bit 63
        const uint32_t ah = (hb == 32) ? fh : (fh & ((1u << hb) - 1u));
        uint32_t hi = ah;
        uint64_t lo = fl;
        if (CANON) {
            const uint32_t s = 96u - 2u * (uint32_t)k;  // 0..30
            const uint32_t r3 = (uint32_t)(rt >> 32), r2 = (uint32_t)rt;
            const uint32_t bh = r3 >> s;
            const uint64_t bl = ((uint64_t)__builtin_amdgcn_alignbit(r3, r2, s) << 32) |
                                __builtin_amdgcn_alignbit(r2, rb, s);
            const bool f_lt = (ah < bh) | ((ah == bh) & (fl < bl));  // bitwise: no exec-mask short circuit
            hi = f_lt ? ah : bh;
            lo = f_lt ? fl : bl;
        }
        // fold128(hi, lo) with hi < 2^32: hi * G = hi * G_lo + ((hi * G_hi) << 32)   (mod 2^64)
        const uint64_t hg = (uint64_t)hi * 0x7F4A7C15u + ((uint64_t)(hi * 0x9E3779B9u) << 32);
        return wang64_fast<false>(lo ^ hg);
    }
};

// The 96-bit class as three 32-bit words per window (see Windows<5>): used by the scatter kernels.
template <>
struct Windows<6> {
    uint32_t f0 = 0, f1 = 0, f2 = 0;  // forward window, low .. high word
    uint32_t r0 = 0, r1 = 0, r2 = 0;  // reverse complement, top-aligned in 96 bits: r2 holds bits 95..64
    DD_D void prime(const uint4& hc) {
        f0 = pairrev32(hc.w);
        f1 = pairrev32(hc.z);
        f2 = pairrev32(hc.y);
        r2 = ~hc.w;
        r1 = ~hc.z;
        r0 = ~hc.y;
    }
    DD_D void push(uint32_t c) {
        f2 = __builtin_amdgcn_alignbit(f2, f1, 30);
        f1 = __builtin_amdgcn_alignbit(f1, f0, 30);
        f0 = (f0 << 2) | c;
        r0 = __builtin_amdgcn_alignbit(r1, r0, 2);
        r1 = __builtin_amdgcn_alignbit(r2, r1, 2);
        r2 = (r2 >> 2) | ((3u - c) << 30);
    }
    template <bool CANON>
    DD_D uint64_t hash(int k) const {
        const int hb = 2 * k - 64;  // 2..32 bits of the k-mer above bit 63
        const uint32_t ah = (hb == 32) ? f2 : (f2 & ((1u << hb) - 1u));
        const uint64_t fl = pack64(f1, f0);
        uint32_t hi = ah;
        uint64_t lo = fl;
        if (CANON) {
            const uint32_t s = 96u - 2u * (uint32_t)k;  // 0..30
            const uint32_t bh = r2 >> s;
            const uint64_t bl = pack64(__builtin_amdgcn_alignbit(r2, r1, s), __builtin_amdgcn_alignbit(r1, r0, s));
            const bool f_lt = (ah < bh) | ((ah == bh) & (fl < bl));  // bitwise: no exec-mask short circuit
            hi = f_lt ? ah : bh;
            lo = f_lt ? fl : bl;
        }
        const uint64_t hg = (uint64_t)hi * 0x7F4A7C15u + ((uint64_t)(hi * 0x9E3779B9u) << 32);
        return wang64_fast<false>(lo ^ hg);
    }
};

// The 128-bit class as four 32-bit words per window (see Windows<5>): used by the scatter kernels, whose push is paid per
// (token, k) -- with u64 members the compiler spent 16 instructions on a push (two 64-bit copies, shift / or pairs,
// and the funnel shifts that take the words apart again at the hash), this form 12.
template <>
struct Windows<7> {
    uint32_t f0 = 0, f1 = 0, f2 = 0, f3 = 0;  // forward window, low .. high word
    uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;  // reverse complement, top-aligned in 128 bits
    DD_D void prime(const uint4& hc) {
        f0 = pairrev32(hc.w);
        f1 = pairrev32(hc.z);
        f2 = pairrev32(hc.y);
        f3 = pairrev32(hc.x);
        r3 = ~hc.w;
        r2 = ~hc.z;
        r1 = ~hc.y;
        r0 = ~hc.x;
    }
    DD_D void push(uint32_t c) {
        f3 = __builtin_amdgcn_alignbit(f3, f2, 30);
        f2 = __builtin_amdgcn_alignbit(f2, f1, 30);
        f1 = __builtin_amdgcn_alignbit(f1, f0, 30);
        f0 = (f0 << 2) | c;
        r0 = __builtin_amdgcn_alignbit(r1, r0, 2);
        r1 = __builtin_amdgcn_alignbit(r2, r1, 2);
        r2 = __builtin_amdgcn_alignbit(r3, r2, 2);
        r3 = (r3 >> 2) | ((3u - c) << 30);
    }
    template <bool CANON>
    DD_D uint64_t hash(int k) const {  // 49 <= k <= 64
        const int hb = 2 * k - 96;  // bits of the k-mer in the top word, 2..32
        const uint32_t mh = (hb == 32) ? ~0u : ((1u << hb) - 1u);
        const uint64_t ah = pack64(f3 & mh, f2);
        const uint64_t al = pack64(f1, f0);
        if (!CANON) return wang64_fast<false>(fold128(ah, al));
        const uint32_t s = 128u - 2u * (uint32_t)k;  // 0..30
        const uint64_t bh = pack64(r3 >> s, __builtin_amdgcn_alignbit(r3, r2, s));
        const uint64_t bl = pack64(__builtin_amdgcn_alignbit(r2, r1, s), __builtin_amdgcn_alignbit(r1, r0, s));
        const bool f_lt = (ah < bh) | ((ah == bh) & (al < bl));  // bitwise: no exec-mask short circuit
        return wang64_fast<false>(fold128(f_lt ? ah : bh, f_lt ? al : bl));
    }
};

// every k of the group for the token just pushed
template <int KC, bool CANON, bool CHECK, typename Win, typename MakeRegs>
DD_D void sweep_token(const Win& win, int run, int kfirst, int nk, int p, const MakeRegs& slot) {
    // The loop counter stays wave-uniform in both variants (k-dependent masks and shifts are then
    // scalar); in the CHECK variant lanes whose run is too short for k are simply predicated off.
    int j = 0;
#pragma unroll 1
    for (; j + 1 < nk; j += 2) {
        const int k = kfirst + j;
        if (!CHECK || run >= k + 1)
            hll_update2(slot(j), win.template hash<CANON>(k), slot(j + 1), win.template hash<CANON>(k + 1), p);
        else if (run >= k)
            hll_update(slot(j), win.template hash<CANON>(k), p);
    }
    if (j < nk && (!CHECK || run >= kfirst + j)) hll_update(slot(j), win.template hash<CANON>(kfirst + j), p);
}

// The registers of the job's k-group live in LDS (2^p * nk bytes <= 160 KiB): log2m <= 16.
template <int KC, bool CANON>
__global__ __launch_bounds__(1024) void sweep_kernel(const SweepGenome* __restrict__ genomes,
                                                    const SweepJob* __restrict__ jobs, int p) {
    lds_starts_at_zero();
    const SweepJob job = jobs[blockIdx.x];
    const SweepGenome g = genomes[job.genome];
    const int nk = job.nk, kfirst = job.kfirst;
    const uint32_t m = 1u << p;
    const unsigned long long ntok = gload8u(g.ntok);
    uint8_t* const slab = g.regs + ((size_t)job.krow << p);

    // A thread's input for one tile: its segment (sc, sb) and the previous one (hc, hb: the halo
    // that primes the windows; segment 0 starts behind a BREAK).  The loads of tile t+1 are issued
    // before tile t is processed, and those of the first tile before the warm start below.
    struct TileIn {
        uint4 hc, sc;
        uint2 hb, sb;
        bool live;  // the segment lies inside the token stream
    };
    auto fetch = [&](unsigned tile, TileIn& t) {
        const unsigned long long seg = (unsigned long long)tile * blockDim.x + threadIdx.x;
        t.live = tile < job.tile_end && seg * kSegTokens < ntok;
        t.hc = make_uint4(0, 0, 0, 0);
        t.hb = make_uint2(~0u, ~0u);
        t.sc = make_uint4(0, 0, 0, 0);  // a segment outside the stream is all BREAKs
        t.sb = make_uint2(~0u, ~0u);
        if (!t.live) return;
        if (seg > 0) {
            t.hc = gload16(g.codes + (seg - 1) * 4);
            t.hb = gload8(g.bad + (seg - 1) * 2);
        }
        t.sc = gload16(g.codes + seg * 4);
        t.sb = gload8(g.bad + seg * 2);
    };
    TileIn next;
    fetch(job.tile_begin, next);

    {
        // Warm start: begin from whatever earlier jobs have already merged into the slab.  Any
        // (possibly stale) snapshot is a valid lower bound of the final registers, and a warm
        // array makes the "register rises" path rare: after T tokens have been absorbed only
        // ~m/T of the updates still raise a register.
        // The snapshot is read with agent-scope loads: other XCDs merge into the slab with
        // memory-side atomics, which a plain load served by THIS XCD's L2 would not see (it kept
        // returning the zeroed lines: every flush then CAS-ed every word and jobs started cold).
        uint4* z = reinterpret_cast<uint4*>(g_lds);
        const uint32_t n16 = (uint32_t)nk * (m >> 4);
        for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x) z[i] = load16_fresh(slab + (size_t)i * 16);
    }
    __syncthreads();

    const int kmaxg = kfirst + nk - 1;
    auto lds_slot = [](int j) { return RegsLds{(uint32_t)j}; };

    for (unsigned tile = job.tile_begin; tile < job.tile_end; ++tile) {
        const TileIn cur = next;
        fetch(tile + 1, next);
        if (!cur.live) continue;
        const uint4 hc = cur.hc, sc = cur.sc;
        const uint2 hb = cur.hb, sb = cur.sb;
        const uint32_t cw[4] = {sc.x, sc.y, sc.z, sc.w};
        // (the 128-bit class as 32-bit words, Windows<7>: 1.6 % at log2m 16, 0.6 % at 14 on k 49..64; the 64- and 96-bit
        // classes measure equal to slightly slower in that form here, where one push serves several ks, and keep u64 members)
        Windows<KC == 2 ? 7 : KC> win;
        win.prime(hc);
        if (__all((hb.x | hb.y | sb.x | sb.y) == 0u)) {
            // No BREAK within 128 tokens of any lane of the wave (the common case away from
            // record boundaries and N runs): every window of every k <= 64 is valid.
#pragma unroll
            for (int w = 0; w < 4; ++w) {
#pragma unroll 1
                for (int i = 0; i < 16; ++i) {
                    win.push((cw[w] >> (2 * i)) & 3u);
                    sweep_token<KC, CANON, false>(win, 0, kfirst, nk, p, lds_slot);
                }
            }
            continue;
        }
        // run = clean tokens ending at the current one; enters as the clean tail of the halo
        int run = hb.y ? __builtin_clz(hb.y) : 32 + (hb.x ? __builtin_clz(hb.x) : 32);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t bw = ((w & 2) ? sb.y : sb.x) >> ((w & 1) * 16);
#pragma unroll 1
            for (int i = 0; i < 16; ++i) {
                const uint32_t c = (cw[w] >> (2 * i)) & 3u;
                run = ((bw >> i) & 1u) ? 0 : run + 1;
                win.push(c);
                // wave-uniform fast path: no lane of the wave is within kmaxg tokens of a BREAK
                if (__all(run >= kmaxg)) sweep_token<KC, CANON, false>(win, run, kfirst, nk, p, lds_slot);
                else sweep_token<KC, CANON, true>(win, run, kfirst, nk, p, lds_slot);
            }
        }
    }
    __syncthreads();

    // merge the group's registers into the genome's slab (rows krow .. krow+nk-1 are contiguous)
    const uint4* l4 = reinterpret_cast<const uint4*>(g_lds);
    uint32_t* gw = reinterpret_cast<uint32_t*>(slab);
    const uint32_t n16 = (uint32_t)nk * (m >> 4);
    for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x) {
        const uint4 lv = l4[i];
        const uint4 gv = load16_fresh(slab + (size_t)i * 16);
        const uint32_t l[4] = {lv.x, lv.y, lv.z, lv.w};
        const uint32_t o[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t old = o[q];
            uint32_t mx = bmax4(old, l[q]);
            while (mx != old) {
                uint32_t prev = gcas32(&gw[4 * i + q], old, mx);
                if (prev == old) break;
                old = prev;
                mx = bmax4(old, l[q]);
            }
        }
    }
}

// ---- small-k class (k <= kBitmapMaxK): presence bitmaps instead of hashing every occurrence -----
// The sketch depends only on the SET of canonical k-mers, and for k <= 9 that set has at most
// 4^9 members, so per (token, k) this kernel only does: mask/shift, canonical min, one LDS word
// read and a bit test (an LDS atomic OR the first time a k-mer is seen).  bitmap_finish_kernel then
// hashes each recorded k-mer exactly once.  Registers are bit-identical to hashing every occurrence.
__constant__ int c_bitmap_off[kBitmapMaxK + 2] = {0, bitmap_offset(1), bitmap_offset(2), bitmap_offset(3),
                                                 bitmap_offset(4), bitmap_offset(5), bitmap_offset(6),
                                                 bitmap_offset(7), bitmap_offset(8), bitmap_offset(9),
                                                 kBitmapWords};
static_assert(kBitmapMaxK == 9 && bitmap_offset(9) + bitmap_words(9) == kBitmapWords &&
              kBitmapWords + kBitmapMaxK < kBitmapStride, "bitmap layout");

// amdgpu_num_sgpr: above 80 SGPRs only 7 waves per SIMD are admitted, i.e. ONE 1024-thread
// workgroup per CU instead of two (MI355X_MICROARCH.md, residency) -- measured 2x on this kernel.
template <bool CANON>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(72))) void bitmap_kernel(const SweepGenome* __restrict__ genomes,
                                                     const SweepJob* __restrict__ jobs) {
    const SweepJob job = jobs[blockIdx.x];
    const SweepGenome g = genomes[job.genome];
    const int kfirst = job.kfirst, klast = job.kfirst + job.nk - 1;
    const unsigned long long ntok = gload8u(g.ntok);
    const int w0 = c_bitmap_off[kfirst], w1 = c_bitmap_off[klast + 1];
    // the LDS image holds words w0..w1-1 of the genome's bitmap block, addressed by their global index
    uint32_t* const bits = reinterpret_cast<uint32_t*>(g_lds) - w0;
    uint32_t kmask = __builtin_amdgcn_readfirstlane(((2u << klast) - 1u) & ~((1u << kfirst) - 1u));  // bit k set: k in the job
    // A k whose set is COMPLETE -- every possible (canonical) k-mer already recorded, which small k
    // reach within the first few hundred thousand tokens of any genome -- can gain nothing from more
    // tokens.  The job that first sees a complete set (below, and again after its merge) raises a flag
    // behind the genome's bitmaps; a job that finds all its ks flagged returns before loading anything.
    uint32_t* const complete = g.bitmap + kBitmapWords;  // one word per k (the block's slack, zeroed per call)
    // warm start from what earlier jobs recorded (any snapshot is a subset of the final set), counting
    // the k-mers each k already has
    __shared__ uint32_t have[kBitmapMaxK + 2];
    __shared__ uint32_t flagged;
    // ONE thread reads the flags for the workgroup: another workgroup may set a flag at any moment, and waves
    // that read it at different times would disagree on kmask -- some would return while others go on to read
    // LDS words the returned waves were meant to load.
    if (threadIdx.x == 0) {
        uint32_t f = 0;
        for (int k = kfirst; k <= klast; ++k)
            if (load4_fresh(complete + k)) f |= 1u << k;
        flagged = f;
    }
    if (threadIdx.x <= kBitmapMaxK) have[threadIdx.x] = 0;
    __syncthreads();
    kmask = __builtin_amdgcn_readfirstlane(kmask & ~flagged);
    if (kmask == 0u) return;
    for (int i = w0 + (int)threadIdx.x; i < w1; i += blockDim.x) {
        const uint32_t v = load4_fresh(&g.bitmap[i]);
        bits[i] = v;
        if (v) {
            int k = kfirst;
            for (int j = kfirst + 1; j <= klast; ++j) k = (i >= c_bitmap_off[j]) ? j : k;
            atomicAdd(&have[k], (uint32_t)__builtin_popcount(v));
        }
    }
    __syncthreads();
    auto all_of = [](int k) { return CANON ? (1u << (2 * k - 1)) + ((k & 1) ? 0u : (1u << (k - 1))) : (1u << (2 * k)); };
    for (int k = kfirst; k <= klast; ++k) {
        if (have[k] == all_of(k)) {
            kmask &= ~(1u << k);
            if (threadIdx.x == 0) gstore4(complete + k, 1u);
        }
    }
    kmask = __builtin_amdgcn_readfirstlane(kmask);
    if (kmask == 0u) return;

    const int prime = klast - 1;
    for (unsigned tile = job.tile_begin; tile < job.tile_end; ++tile) {
        const unsigned long long seg = (unsigned long long)tile * blockDim.x + threadIdx.x;
        if (seg * kSegTokens >= ntok) continue;
        uint32_t fw = 0, rc = 0;
        int run = 0;
        if (seg > 0) {
            const uint4 hc = gload16(g.codes + (seg - 1) * 4);
            const uint2 hb = gload8(g.bad + (seg - 1) * 2);
            const uint32_t cw = hc.w, bw = hb.y >> 16;  // last 16 tokens of the halo (prime <= 8)
#pragma unroll 1
            for (int i = 16 - prime; i < 16; ++i) {
                const uint32_t c = (cw >> (2 * i)) & 3u;
                run = ((bw >> i) & 1u) ? 0 : run + 1;
                fw = (fw << 2) | c;
                rc = (rc >> 2) | ((3u - c) << 30);
            }
        }
        const uint4 sc = gload16(g.codes + seg * 4);
        const uint2 sb = gload8(g.bad + seg * 2);
        const uint32_t cws[4] = {sc.x, sc.y, sc.z, sc.w};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t bw = ((w & 2) ? sb.y : sb.x) >> ((w & 1) * 16);
#pragma unroll 1
            for (int i = 0; i < 16; ++i) {
                const uint32_t c = (cws[w] >> (2 * i)) & 3u;
                run = ((bw >> i) & 1u) ? 0 : run + 1;
                fw = (fw << 2) | c;
                rc = (rc >> 2) | ((3u - c) << 30);
                // fully unrolled over k so masks, shifts and bitmap offsets are immediates and the
                // LDS word reads of all ks are in flight together; `need` collects the (rare) lanes
                // that saw a new k-mer
                uint32_t xs[kBitmapMaxK + 1], seen[kBitmapMaxK + 1];
                uint32_t all_seen = 1u;  // bit 0 stays set while every k-mer of this token is known
#pragma unroll
                for (int k = 1; k <= kBitmapMaxK; ++k) {
                    if (!((kmask >> k) & 1u)) continue;  // wave-uniform (one scalar bit test, no live SGPR pair per k)
                    uint32_t x = fw & ((1u << (2 * k)) - 1u);
                    if (CANON) {
                        const uint32_t r = rc >> (32 - 2 * k);
                        x = x < r ? x : r;
                    }
                    xs[k] = x;
                    seen[k] = bits[bitmap_offset(k) + (x >> 5)];
                    all_seen &= seen[k] >> (x & 31u);
                }
                // Validity (run >= k) is only consulted on the rare path: an invalid window near a
                // BREAK can at worst send its lane there for nothing.
                if (!(all_seen & 1u)) {
#pragma unroll
                    for (int k = 1; k <= kBitmapMaxK; ++k) {
                        if (!((kmask >> k) & 1u) || run < k) continue;
                        if (!((seen[k] >> (xs[k] & 31u)) & 1u))
                            atomicOr(&bits[bitmap_offset(k) + (xs[k] >> 5)], 1u << (xs[k] & 31u));
                    }
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x <= kBitmapMaxK) have[threadIdx.x] = 0;
    __syncthreads();
    for (int i = w0 + (int)threadIdx.x; i < w1; i += blockDim.x) {
        const uint32_t mine = bits[i], theirs = load4_fresh(&g.bitmap[i]);
        if (mine & ~theirs) gor32(&g.bitmap[i], mine);
        int k = kfirst;
        for (int j = kfirst + 1; j <= klast; ++j) k = (i >= c_bitmap_off[j]) ? j : k;
        if (mine | theirs) atomicAdd(&have[k], (uint32_t)__builtin_popcount(mine | theirs));
    }
    __syncthreads();
    if (threadIdx.x >= (unsigned)kfirst && threadIdx.x <= (unsigned)klast && have[threadIdx.x] == all_of((int)threadIdx.x))
        gstore4(complete + threadIdx.x, 1u);
}

// grid = (ks, genomes, index tiles): a workgroup builds one 64 KiB tile of the row in LDS (the whole row when it is
// smaller) from ALL the k-mers of the set -- hashing a k-mer 16 times at log2m 20 costs nothing next to what one
// workgroup per row doing global compare-and-swaps cost there (2.3 ms for the k = 9 rows alone).
template <bool CANON_UNUSED>
__global__ __launch_bounds__(1024) void bitmap_finish_kernel(const SweepGenome* __restrict__ genomes,
                                                            int kfirst, int kmin, int p, int tile_log2) {
    lds_starts_at_zero();
    const SweepGenome g = genomes[blockIdx.y];
    const int k = kfirst + (int)blockIdx.x;
    const uint32_t tile = 1u << tile_log2, b = blockIdx.z;
    const uint32_t* bm = g.bitmap + c_bitmap_off[k];
    const int nw = c_bitmap_off[k + 1] - c_bitmap_off[k];
    if (gridDim.z > 1 && nw <= 512) {
        // k <= 7 in a row of several tiles: at most 8256 k-mers for a row of 2^17 .. 2^20 registers that the call
        // zeroed when it started -- ONE workgroup raises the few registers in place instead of 16 writing tiles of
        // zeros (64 genomes at log2m 20: 4096 of the launch's 6144 workgroups)
        if (b != 0) return;
        uint8_t* const row = g.regs + ((size_t)(k - kmin) << p);
        for (int w = threadIdx.x; w < nw; w += blockDim.x) {
            uint32_t v = gload4(bm + w);
            while (v) {
                const uint32_t bit = (uint32_t)__builtin_ctz(v);
                v &= v - 1;
                hll_update(RegsGlobal{row}, wang64_fast<true>(((uint32_t)w << 5) | bit), p);
            }
        }
        return;
    }
    uint8_t* const out = g.regs + ((size_t)(k - kmin) << p) + (size_t)b * tile;
    uint4* z = reinterpret_cast<uint4*>(g_lds);
    for (uint32_t i = threadIdx.x; i < (tile >> 4); i += blockDim.x) z[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    for (int w = threadIdx.x; w < nw; w += blockDim.x) {
        uint32_t v = gload4(bm + w);
        while (v) {
            const uint32_t bit = (uint32_t)__builtin_ctz(v);
            v &= v - 1;
            const Probe q = probe(wang64_fast<true>(((uint32_t)w << 5) | bit), p);
            const uint32_t idx = q.hi >> (32 - p);
            if ((idx >> tile_log2) != b) continue;
            const uint32_t a = idx & (tile - 1u), rho = rho_of(q, p);
            const uint32_t wd = RegsLds::load32(a);
            if (rho > ((wd >> RegsLds::shift(a)) & 0xFFu)) (void)cas_raise<RegsLds>(a, wd, rho);
        }
    }
    __syncthreads();
    const uint4* l4 = reinterpret_cast<const uint4*>(g_lds);
    for (uint32_t i = threadIdx.x; i < (tile >> 4); i += blockDim.x) gstore16(out + (size_t)i * 16, l4[i]);
}


// ---- big-bitmap class (dd_kernels.h): k = 10 (, 11) at log2m >= 19 ---------------------------------------
// index of the k-mer whose forward / reverse-complement values are fw / rc (both 2k bits)
template <bool CANON>
DD_D uint32_t bigmap_index(uint32_t fw, uint32_t rc, int k) {
    if (!CANON) return fw;
    if (k & 1) {
        const uint32_t y = ((fw >> k) & 1u) ? rc : fw;  // the strand whose middle base is A or C
        return ((y >> (k + 1)) << k) | (y & ((1u << k) - 1u));
    }
    return fw < rc ? fw : rc;
}
// ... and back: the value that is hashed
template <bool CANON>
DD_D uint32_t bigmap_kmer(uint32_t idx, int k) {
    if (!CANON || !(k & 1)) return idx;
    const uint32_t y = ((idx >> k) << (k + 1)) | (idx & ((1u << k) - 1u));
    uint32_t r = __brev(y);                                        // bases reversed, the two bits of each swapped
    r = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
    r = (~r) >> (32 - 2 * k);
    return y < r ? y : r;
}

// One workgroup per CU (128 KiB of LDS): the slice as earlier jobs left it, this job's tiles, merge.
template <bool CANON>
__global__ __launch_bounds__(1024) void bigmap_kernel(const SweepGenome* __restrict__ genomes, const SweepJob* __restrict__ jobs) {
    const SweepJob job = jobs[blockIdx.x];
    const SweepGenome g = genomes[job.genome];
    const int k = job.kfirst;
    const uint32_t slice = (uint32_t)job.slice;
    const unsigned long long ntok = gload8u(g.ntok);
    uint32_t* const home = g.bigmap + bigmap_offset_words(k, CANON) + (size_t)slice * kBigmapSliceWords;
    uint32_t* const bits = reinterpret_cast<uint32_t*>(g_lds);
    for (int i = threadIdx.x; i < kBigmapSliceWords; i += blockDim.x) bits[i] = load4_fresh(home + i);
    __syncthreads();
    const uint32_t mask = (1u << (2 * k)) - 1u;
    const int top = 2 * k - 2, prime = k - 1;
    for (unsigned tile = job.tile_begin; tile < job.tile_end; ++tile) {
        const unsigned long long seg = (unsigned long long)tile * blockDim.x + threadIdx.x;
        if (seg * kSegTokens >= ntok) continue;
        uint32_t fw = 0, rc = 0;
        int run = 0;
        if (seg > 0) {
            const uint4 hc = gload16(g.codes + (seg - 1) * 4);
            const uint2 hb = gload8(g.bad + (seg - 1) * 2);
            const uint32_t cw = hc.w, bw = hb.y >> 16;  // last 16 tokens of the halo (prime <= 10)
#pragma unroll 1
            for (int i = 16 - prime; i < 16; ++i) {
                const uint32_t c = (cw >> (2 * i)) & 3u;
                run = ((bw >> i) & 1u) ? 0 : run + 1;
                fw = ((fw << 2) | c) & mask;
                rc = (rc >> 2) | ((3u - c) << top);
            }
        }
        const uint4 sc = gload16(g.codes + seg * 4);
        const uint2 sb = gload8(g.bad + seg * 2);
        const uint32_t cws[4] = {sc.x, sc.y, sc.z, sc.w};
        auto record = [&](uint32_t idx, bool ok) {
            if ((idx >> 20) == slice && ok) {
                const uint32_t at = (idx & 0xFFFFFu) >> 5, bit = 1u << (idx & 31u);
                if (!(bits[at] & bit)) atomicOr(&bits[at], bit);
            }
        };
        if (__all((sb.x | sb.y) == 0u && run >= prime)) {
            // the usual case: no BREAK anywhere in the wave's 64 x 64 tokens, every window valid
#pragma unroll
            for (int w = 0; w < 4; ++w) {
#pragma unroll 4
                for (int i = 0; i < 16; ++i) {
                    const uint32_t c = (cws[w] >> (2 * i)) & 3u;
                    fw = ((fw << 2) | c) & mask;
                    rc = (rc >> 2) | ((3u - c) << top);
                    record(bigmap_index<CANON>(fw, rc, k), true);
                }
            }
            continue;
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t bw = ((w & 2) ? sb.y : sb.x) >> ((w & 1) * 16);
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {
                const uint32_t c = (cws[w] >> (2 * i)) & 3u;
                run = ((bw >> i) & 1u) ? 0 : run + 1;
                fw = ((fw << 2) | c) & mask;
                rc = (rc >> 2) | ((3u - c) << top);
                record(bigmap_index<CANON>(fw, rc, k), run >= k);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kBigmapSliceWords; i += blockDim.x) {
        const uint32_t mine = bits[i];
        if (mine && (mine & ~load4_fresh(home + i))) gor32(home + i, mine);
    }
}

// grid = (ks, genomes, index tiles), as bitmap_finish_kernel
template <bool CANON>
__global__ __launch_bounds__(1024) void bigmap_finish_kernel(const SweepGenome* __restrict__ genomes,
                                                            int kfirst, int kmin, int p, int tile_log2) {
    lds_starts_at_zero();
    const SweepGenome g = genomes[blockIdx.y];
    const int k = kfirst + (int)blockIdx.x;
    const uint32_t tile = 1u << tile_log2, b = blockIdx.z;
    uint8_t* const out = g.regs + ((size_t)(k - kmin) << p) + (size_t)b * tile;
    uint4* z = reinterpret_cast<uint4*>(g_lds);
    for (uint32_t i = threadIdx.x; i < (tile >> 4); i += blockDim.x) z[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const uint32_t* bm = g.bigmap + bigmap_offset_words(k, CANON);
    const uint32_t nw = (uint32_t)bigmap_slices(k, CANON) * kBigmapSliceWords;
    for (uint32_t w = threadIdx.x; w < nw; w += blockDim.x) {
        uint32_t v = gload4(bm + w);
        while (v) {
            const uint32_t bit = (uint32_t)__builtin_ctz(v);
            v &= v - 1;
            const Probe q = probe(wang64_fast<true>(bigmap_kmer<CANON>((w << 5) | bit, k)), p);
            const uint32_t idx = q.hi >> (32 - p);
            if ((idx >> tile_log2) != b) continue;
            const uint32_t a = idx & (tile - 1u), rho = rho_of(q, p);
            const uint32_t wd = RegsLds::load32(a);
            if (rho > ((wd >> RegsLds::shift(a)) & 0xFFu)) (void)cas_raise<RegsLds>(a, wd, rho);
        }
    }
    __syncthreads();
    const uint4* l4 = reinterpret_cast<const uint4*>(g_lds);
    for (uint32_t i = threadIdx.x; i < (tile >> 4); i += blockDim.x) gstore16(out + (size_t)i * 16, l4[i]);
}

// ---- log2m >= 17, bucket mode: scatter + sort + replay (dd_kernels.h) -------------------------------
// The compare-and-swap path above is bound by the device's scattered-atomic rate (27 G/s measured, any
// atomic, any footprint: profiles/r01_ubench_atomics.txt).  Two earlier forms of this path were measured
// (profiles/r02_bucket_path.txt): records stored one by one to per-index-tile chunks ran into the same
// wall (a 4-byte store that is not part of a whole line leaves the L2 as a fabric write of its own);
// records staged per (wave, index tile) in LDS and flushed as 128-byte lines made the stores cheap but
// cost 30 VALU + 30 SALU per wave-update for the staging -- the kernel is issue-bound, so that doubled it.
// Hence: scatter does NO partitioning.  A wave appends its surviving records to one LDS queue (ballot +
// mbcnt + one ds_write) and, whenever 64 wait, stores them as one 256-byte block to the ROW's record
// stream; the chunks of the stream are sorted by index tile afterwards (sort_chunks_kernel, or the first epoch's
// scatter itself), and the replay workgroups of a row (one per 64 KiB index tile) read only their own segments.
// The rows a sort / replay / reset launch covers: rows k0 .. k0+nks-1 of every genome (one k class), numbered
// densely; table index = genome * K + k0 + local % nks.
struct RowSet {
    int K, k0, nks, nrows;  // nrows = genomes * nks, or the rows of one row group
    int row0;               // ... which starts at this row of the class
    DD_D int index(uint32_t local) const {
        local += (uint32_t)row0;
        return (int)(local / (uint32_t)nks) * K + k0 + (int)(local % (uint32_t)nks);
    }
};
constexpr uint32_t kChunkRecords = 1024;         // 4 KiB; one global atomic hands out one chunk of the row's stream

struct Scatter {
    uint32_t queue;        // byte offset in g_lds of this wave's two record queues (2 x kQueueEntries x 4 B)
    uint32_t* area;        // the row's record stream, chunk c at area + c * kChunkRecords
    uint32_t* cursor;      // records reserved so far (may run past the capacity: readers clamp)
    uint8_t* regs;         // the row itself: what candidates are probed against, and where records go when the stream is full
    uint32_t cap_chunks;
    int fshift;            // hash high word >> fshift = index of the register group's filter entry (32 - p + logg)
    int ishift;            // hash high word >> ishift = register index (32 - p)
    uint32_t himask;       // the index bits of the hash high word
};
constexpr uint32_t kScatterUnit = 256;  // records a wave reserves at a time (a multiple of 64)
DD_D uint32_t& lds32(uint32_t off) { return *reinterpret_cast<uint32_t*>(g_lds + off); }
DD_D uint32_t gadd32(void* p, uint32_t v) {
    return __hip_atomic_fetch_add((DD_GLOBAL uint32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 64 records (one per lane; null records have rho 0) leave for the row's stream.  The stream is DENSE: the row's
// cursor counts records, a wave reserves kScatterUnit of them with one atomic add (`cur` / `left` = its reservation), so
// the 1024-record chunks the sort and the replay work on are all full whatever the job sizes were.  (Round 2's first
// form gave every wave of every job a chunk of its own: 4.4 M chunks per log2m 20 step for 1.7 G records, i.e. 38 % full,
// and the replay's per-tile segments 30 records long.)  Whole waves, uniform state.
DD_D void scatter_block(const Scatter& s, uint32_t rec, uint32_t& cur, uint32_t& left) {
    const uint32_t lane = threadIdx.x & 63u;
    if (left == 0u) {
        uint32_t c = 0;
        if (lane == 0) c = gadd32(s.cursor, kScatterUnit);
        cur = __builtin_amdgcn_readfirstlane(c);
        left = kScatterUnit / 64u;
    }
    const uint32_t pos = cur;
    cur += 64u;
    --left;
    if (pos + 64u > s.cap_chunks * kChunkRecords) {
        // the stream is full: the records go to their registers directly (exact, slow, rare)
        if (rec >> 24) {
            uint8_t* a = s.regs + (rec & 0xFFFFFFu);
            (void)cas_raise<RegsGlobal>(a, RegsGlobal::load32(a), rec >> 24);
        }
        return;
    }
    gstore4(s.area + pos + lane, rec);
}
// Second-level filter: 64 queued candidates are checked against the ROW ITSELF -- one byte load per
// lane from the registers as the last replay left them (the row of the jobs an XCD is running stays in that
// XCD's L2: job order, dd_plan.hip) -- and only those that really exceed their register move on to a second
// queue and, 64 at a time, to the stream.  The group-minimum filter lets ~25 % of the updates through at log2m
// 20; about 10 % really raise a register.  Exact either way: a register only rises, so its last stored value is
// a lower bound.
// Candidates wait in the first queue as the hash word's index bits with rho - 1 in the low byte (one v_and_or when
// they are queued -- that code runs on nearly every update of the wave; 0xFF = no candidate); what survives the
// probe is put into record form, idx | rho << 24, here, once per 64 candidates.
DD_D void scatter_probe(const Scatter& s, uint32_t cand, uint32_t& waiting2, uint32_t& cur, uint32_t& left) {
    const uint32_t rm1 = cand & 0xFFu, idx = cand >> s.ishift;
    bool live = rm1 != 0xFFu;
    if (live) live = rm1 >= (uint32_t)*(const DD_GLOBAL uint8_t*)(s.regs + idx);  // rho > register
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(live);
    if (mask) {
        if (live) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            lds32(s.queue + kQueueEntries * 4u + 4u * (waiting2 + rank)) = idx | ((rm1 + 1u) << 24);
        }
        waiting2 += (uint32_t)__builtin_popcountll(mask);
        if (waiting2 >= 64u) {
            waiting2 -= 64u;
            scatter_block(s, lds32(s.queue + kQueueEntries * 4u + 4u * (waiting2 + (threadIdx.x & 63u))), cur, left);
        }
    }
}
// One update.  Reached by whole waves (`valid`: the lane has a k-mer); `waiting`, `waiting2`, `cur` are wave-uniform.
// The filter holds 4-bit bounds (saturating at 15), two register groups per byte: twice the resolution of byte entries
// in the same 64 KiB of LDS for three more instructions per update (measured better at log2m 18, 19 and 20).
DD_D void scatter_update(const Scatter& s, uint32_t& waiting, uint32_t& waiting2, uint32_t& cur, uint32_t& left, uint64_t h, int p, bool valid) {
    const Probe q = probe(h, p);
    // entry e = hi >> fshift sits in nibble e & 1 of byte e >> 1: address and nibble shift straight from hi (three
    // instructions instead of five; fshift >= 32 - 20 + 1).  The byte is read at its absolute LDS address: this
    // kernel has no static LDS, so the dynamic array starts at 0 (checked when the job starts), and going through
    // the g_lds symbol costs a v_add of its link-time address, 0, on every update.
    const uint32_t at = q.hi >> (s.fshift + 1);
    const uint32_t bound = __builtin_amdgcn_ubfe((uint32_t)*(const __attribute__((address_space(3))) uint8_t*)(uintptr_t)at, (q.hi >> (s.fshift - 2)) & 4u, 4u);
    const bool cand = valid && q.lz >= bound;  // rho > bound (or hiw == 0: rho >= 33)
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(cand);
    if (mask) {
        if (cand) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            lds32(s.queue + 4u * (waiting + rank)) = (q.hi & s.himask) | (rho_of(q, p) - 1u);  // (scatter_probe's form)
        }
        waiting += (uint32_t)__builtin_popcountll(mask);
        if (waiting >= 64u) {
            waiting -= 64u;
            scatter_probe(s, lds32(s.queue + 4u * (waiting + (threadIdx.x & 63u))), waiting2, cur, left);
        }
    }
}

// The filtered epochs' scatter: one k per job.  (Two consecutive ks per job -- shared token loads and window push, 7 of the
// ~50 VALU instructions of an update -- measured SLOWER on MI355X both with two 64 KiB filters = one workgroup per CU and
// with two 16 KiB filters at two workgroups per CU: profiles/r02_bucket_path.txt, profiles/r03_bucket_path.txt.)
template <int KC, bool CANON>
__global__ __launch_bounds__(1024) void scatter_kernel(const SweepGenome* __restrict__ genomes,
                                                      const SweepJob* __restrict__ jobs, int p, ScatterParams sp) {
    const SweepJob job = jobs[blockIdx.x];
    if (job.tile_begin >= job.tile_end) return;  // filler of the XCD-affine order
    lds_starts_at_zero();  // scatter_update reads the filter at absolute LDS addresses
    const SweepGenome g = genomes[job.genome];
    const int k = job.kfirst;
    const uint32_t m = 1u << p;
    const unsigned long long ntok = gload8u(g.ntok);
    const uint32_t nflt = (m >> sp.logg) >> 1;  // bytes of the filter

    struct TileIn {
        uint4 hc, sc;
        uint2 hb, sb;
        bool live;
    };
    auto fetch = [&](unsigned tile, TileIn& t) {
        const unsigned long long seg = (unsigned long long)tile * blockDim.x + threadIdx.x;
        t.live = tile < job.tile_end && seg * kSegTokens < ntok;
        t.hc = make_uint4(0, 0, 0, 0);
        t.hb = make_uint2(~0u, ~0u);
        t.sc = make_uint4(0, 0, 0, 0);
        t.sb = make_uint2(~0u, ~0u);
        if (!t.live) return;
        if (seg > 0) {
            t.hc = gload16(g.codes + (seg - 1) * 4);
            t.hb = gload8(g.bad + (seg - 1) * 2);
        }
        t.sc = gload16(g.codes + seg * 4);
        t.sb = gload8(g.bad + seg * 2);
    };
    TileIn next;
    fetch(job.tile_begin, next);

    // the row's filter as the previous epoch's replay left it (plain loads: written by an earlier kernel) at LDS offset 0,
    // then the per-wave queues
    const BucketRow row = sp.rows[(size_t)job.genome * sp.K + job.krow];
    {
        uint4* f4 = reinterpret_cast<uint4*>(g_lds);
        for (uint32_t i = threadIdx.x; i < (nflt >> 4); i += blockDim.x) f4[i] = gload16(row.filter + (size_t)i * 16);
    }
    Scatter s;
    s.queue = nflt + (threadIdx.x >> 6) * (kQueueEntries * 4u * 2u);
    s.area = row.area;
    s.cursor = row.cursor;
    s.regs = row.regs;
    s.cap_chunks = sp.cap_chunks;
    s.fshift = 32 - p + sp.logg;
    s.ishift = 32 - p;
    s.himask = ~((1u << (32 - p)) - 1u);
    uint32_t waiting = 0, waiting2 = 0, cur = 0, left = 0;
    __syncthreads();

    for (unsigned tile = job.tile_begin; tile < job.tile_end; ++tile) {
        const TileIn in = next;
        fetch(tile + 1, next);
        // Lanes beyond the stream stay in the loop as all-BREAK segments while any lane of their wave has
        // tokens: the queue counters and the stream offsets must stay wave-uniform.
        if (!__any(in.live)) continue;
        const uint4 hc = in.hc, sc = in.sc;
        const uint2 hb = in.hb, sb = in.sb;
        const uint32_t cw[4] = {sc.x, sc.y, sc.z, sc.w};
        Windows<KC == 1 ? 5 : (KC == 3 ? 6 : (KC == 2 ? 7 : KC))> win;
        win.prime(hc);
        if (__all((hb.x | hb.y | sb.x | sb.y) == 0u)) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
#pragma unroll 1
                for (int i = 0; i < 16; ++i) {
                    win.push((cw[w] >> (2 * i)) & 3u);
                    scatter_update(s, waiting, waiting2, cur, left, win.template hash<CANON>(k), p, true);
                }
            }
            continue;
        }
        int run = hb.y ? __builtin_clz(hb.y) : 32 + (hb.x ? __builtin_clz(hb.x) : 32);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t bw = ((w & 2) ? sb.y : sb.x) >> ((w & 1) * 16);
#pragma unroll 1
            for (int i = 0; i < 16; ++i) {
                const uint32_t c = (cw[w] >> (2 * i)) & 3u;
                run = ((bw >> i) & 1u) ? 0 : run + 1;
                win.push(c);
                scatter_update(s, waiting, waiting2, cur, left, win.template hash<CANON>(k), p, run >= k);
            }
        }
    }
    // what still waits leaves as a block padded with null records, and what is left of the wave's last reservation is
    // filled with null blocks (the stream has no holes: sort and replay read all of it)
    const uint32_t lane = threadIdx.x & 63u;
    if (waiting) scatter_probe(s, lane < waiting ? lds32(s.queue + 4u * lane) : 0xFFu, waiting2, cur, left);
    if (waiting2) scatter_block(s, lane < waiting2 ? lds32(s.queue + kQueueEntries * 4u + 4u * lane) : 0u, cur, left);
    while (left) scatter_block(s, 0u, cur, left);
}

// a wave-uniform value that arrived through a vector load (a table entry): moved to scalar registers
DD_D uint64_t uniform64(uint64_t v) {
    // (the builtin returns int: without the casts the low half would be sign-extended over the high one)
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)), lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
    return ((uint64_t)hi << 32) | lo;
}
template <typename T>
DD_D T* uniform_ptr(T* q) { return reinterpret_cast<T*>(uniform64(reinterpret_cast<uint64_t>(q))); }

// ---- first epoch, BINNED tiles of tokens (round 4, what runs) ----------------------------------------------------
// Sorting 16 384 records per workgroup still costs two LDS atomics per record (count, then place) plus the pass over
// the collection area: 2.4 of the 4.5 ms of a class-0 launch over 64 x 5 Mbp at log2m 20, against 2.1 ms of hashing
// (timing-only builds, profiles/r04_bucket_path.txt).  A record's index tile is the top bits of a hash, so the 65 536
// records a workgroup makes of one tile of tokens spread over the 16 bins (index tile x copy) as evenly as coin flips
// do: 4096 per bin, sigma 62.  So every bin of a chunk gets a FIXED region of kBinCap = 4480 records (+ 6 sigma) in
// the row's stream, a record's slot is ONE returning LDS atomic on the workgroup's counter of its bin, and the record
// goes straight from the hash to its slot -- no collection area, no counting pass, no placement pass, one workgroup
// barrier per 64 updates (the counters of odd and even tiles alternate; wave 0 saves and clears a tile's counters
// behind the barrier while the others already fill the next tile's).  A bin that should ever overflow sends the
// record to its register by compare-and-swap (exact; ~3e-10 per bin).  The replay reads a bin's records -- 16 KiB
// in one piece -- with 16-byte loads.  Stream space: 70 instead of 64 chunks of 1024 records per tile of tokens.
constexpr uint32_t kBinCap = 4480;                        // records per (chunk, bin): a multiple of 64
constexpr uint32_t kBinChunkRecords = 16u * kBinCap;      // 71 680 = 70 x 1024: stream space of one tile of tokens
constexpr uint32_t kBinPosSlot = 128u;                    // LDS: counters [2][16] at 0, the job's position behind them
constexpr uint32_t kBinLdsBytes = 256u;
constexpr int kOnesLog2Max = 19;                          // registers of a row whose rho = 1 updates are bits in LDS (below)

// Updates of rho = 1 (round 5): HALF of all updates have rho = 1, and all a register can learn from them is that it
// is not empty.  They leave no record: the workgroup keeps one bit per register of its row in LDS (m / 8 bytes behind the
// counters, 64 KiB at most), sets it with a ds_or and ORs the words into the row's bitmap in HBM
// when its job ends (BucketRow::ones; 32 K atomics per job against the ~330 K four-byte stores they stand for); the replay
// raises a register that is still 0 behind a set bit to 1 when it writes the tile back.  Exact: max(rho) over a register's
// updates is 1 iff there is an update and none has rho >= 2.
template <int KC, bool CANON>
__global__ __launch_bounds__(1024) void scatter_first_bin_kernel(
    const SweepGenome* __restrict__ genomes, const SweepJob* __restrict__ jobs, int p, ScatterParams sp) {
    const SweepJob job = jobs[blockIdx.x];
    if (job.tile_begin >= job.tile_end) return;  // filler of the XCD-affine order
    lds_starts_at_zero();
    const SweepGenome g = genomes[job.genome];
    const int k = job.kfirst;
    const unsigned long long ntok = uniform64(gload8u(g.ntok));
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;

    struct TileIn {
        uint4 hc, sc;
        uint2 hb, sb;
        bool live;
    };
    auto fetch = [&](unsigned tile, TileIn& t) {
        const unsigned long long seg = (unsigned long long)tile * blockDim.x + threadIdx.x;
        t.live = tile < job.tile_end && seg * kSegTokens < ntok;
        t.hc = make_uint4(0, 0, 0, 0);
        t.hb = make_uint2(~0u, ~0u);
        t.sc = make_uint4(0, 0, 0, 0);
        t.sb = make_uint2(~0u, ~0u);
        if (!t.live) return;
        if (seg > 0) {
            t.hc = gload16(g.codes + (seg - 1) * 4);
            t.hb = gload8(g.bad + (seg - 1) * 2);
        }
        t.sc = gload16(g.codes + seg * 4);
        t.sb = gload8(g.bad + seg * 2);
    };
    TileIn next;
    fetch(job.tile_begin, next);

    const BucketRow row = sp.rows[(size_t)job.genome * sp.K + job.krow];
    uint32_t* const area = uniform_ptr(row.area);
    uint16_t* const counts = uniform_ptr(row.seg);  // [chunk][16]: records in each bin
    uint8_t* const regs = uniform_ptr(row.regs);
    const uint32_t cap_records = sp.cap_chunks * kChunkRecords;
    const int cshift = 4 - sp.nb_log2, tile_sh = 32 - sp.nb_log2;
    if (threadIdx.x < 32u) lds32(4u * threadIdx.x) = 0;
    if (threadIdx.x == 0) lds32(kBinPosSlot) = gadd32(row.cursor, (job.tile_end - job.tile_begin) * kBinChunkRecords);
    // (at most 2^19 bits = 64 KiB, so that two workgroups still share a CU: at log2m 20 only the updates of the lower half of the
    // row's registers are bits, the others stay records -- one workgroup per CU costs this kernel 6 %, profiles/r05_bucket_path.txt)
    const uint32_t ones_regs = 1u << (p < kOnesLog2Max ? p : kOnesLog2Max), ones_words = ones_regs >> 5;
    for (uint32_t w = threadIdx.x; w < ones_words; w += blockDim.x) lds32(kBinLdsBytes + 4u * w) = 0;
    __syncthreads();
    const uint32_t pos0 = __builtin_amdgcn_readfirstlane(lds32(kBinPosSlot));
    const uint32_t copy = lane & ((1u << cshift) - 1u);

    for (unsigned tile = job.tile_begin; tile < job.tile_end; ++tile) {
        const TileIn in = next;
        fetch(tile + 1, next);
        const uint32_t t = tile - job.tile_begin;
        const uint32_t ctr = (t & 1u) * 64u;
        const uint32_t cpos = pos0 + t * kBinChunkRecords;
        const bool room = cpos + kBinChunkRecords <= cap_records;  // else: the stream is full, records go to the registers (exact, slow, rare)
        if (__any(in.live)) {
            const uint4 hc = in.hc, sc = in.sc;
            const uint2 hb = in.hb, sb = in.sb;
            const uint32_t cw[4] = {sc.x, sc.y, sc.z, sc.w};
            uint8_t* const chunk = reinterpret_cast<uint8_t*>(area + cpos);   // (wave-uniform: the store below is base + 32-bit offset)
                    Windows<KC == 1 ? 5 : (KC == 3 ? 6 : (KC == 2 ? 7 : KC))> win;
            win.prime(hc);
            // (deferring a record's store until the next update's atomic is out, so that the slot's LDS round trip overlaps a
            // hash, and unrolling the token loop by two were both measured: no difference -- the loop is not waiting there)
            auto update = [&](bool valid) {
                const Probe q = probe(win.template hash<CANON>(k), p);
                if (!valid) return;
                const uint32_t rho = rho_of(q, p);
                if (rho == 1u && (q.hi >> (32 - p)) < ones_regs) {
                    const uint32_t idx = q.hi >> (32 - p);
                    atomicOr(&lds32(kBinLdsBytes + ((idx >> 5) << 2)), 1u << (idx & 31u));
                    return;
                }
                const uint32_t rec = (q.hi >> (32 - p)) | (rho << 24);
                const uint32_t bin0 = (q.hi >> tile_sh) << cshift;  // + copy = the bin
                uint32_t slot = kBinCap;
                if (room) slot = atomicAdd(&lds32(ctr + ((bin0 | copy) << 2)), 1u);
                if (__builtin_expect(slot < kBinCap, 1)) {
                    // (round 5: the slot's address as the chunk's uniform base + a 32-bit byte offset -- one v_mad_u32_u24 and a shift
                    // in front of a store with an SGPR base instead of a multiply and two 64-bit adds; A/B on one box: 23.3-23.6 against
                    // 23.4-23.8 ms for 64 x 5 Mbp at log2m 20, 25.0-25.2 against 24.9-25.0 for 10 x 50 Mbp -- within the noise: the kernel is
                    // not waiting for its VALU, profiles/r05_bucket_path.txt)
                    gstore4(chunk + (__umul24(bin0 | copy, kBinCap) + slot) * 4u, rec);
                } else {
                    uint8_t* a = regs + (rec & 0xFFFFFFu);
                    (void)cas_raise<RegsGlobal>(a, RegsGlobal::load32(a), rec >> 24);
                }
            };
            if (__all((hb.x | hb.y | sb.x | sb.y) == 0u)) {
#pragma unroll
                for (int w = 0; w < 4; ++w) {
#pragma unroll 1
                    for (int i = 0; i < 16; ++i) {
                        win.push((cw[w] >> (2 * i)) & 3u);
                        update(true);
                    }
                }
            } else {
                int run = hb.y ? __builtin_clz(hb.y) : 32 + (hb.x ? __builtin_clz(hb.x) : 32);
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const uint32_t bw = ((w & 2) ? sb.y : sb.x) >> ((w & 1) * 16);
#pragma unroll 1
                    for (int i = 0; i < 16; ++i) {
                        run = ((bw >> i) & 1u) ? 0 : run + 1;
                        win.push((cw[w] >> (2 * i)) & 3u);
                        update(run >= k);
                    }
                }
            }
        }
        __syncthreads();  // the tile's records are placed and counted; the other parity's counters are clear
        if (wave == 0u && lane < 16u) {
            const uint32_t c = lds32(ctr + 4u * lane);
            lds32(ctr + 4u * lane) = 0;
            if (room) ((DD_GLOBAL uint16_t*)counts)[(size_t)(cpos / kBinChunkRecords) * 16u + lane] = (uint16_t)(c < kBinCap ? c : kBinCap);
        }
    }
    // (behind the last tile's barrier: every ds_or of the job is in)
    uint32_t* const ones = uniform_ptr(row.ones);
    for (uint32_t w = threadIdx.x; w < ones_words; w += blockDim.x) {
        const uint32_t v = lds32(kBinLdsBytes + 4u * w);
        if (v) atomicOr(ones + w, v);
    }
}

// Between scatter and replay when a row has more than one index tile (log2m >= 17): every chunk of every
// stream is sorted by index tile in place (one wave per chunk: LDS counting sort), null records dropped,
// and the start of each tile's segment is noted in seg[chunk][tile].  A replay workgroup then reads only
// its own segments; without this every one of the 8 workgroups of a log2m 20 row (128 KiB tiles then)
// inspected every record (measured: 42 of 72 ms).  HBM-bound: each record is read and written once more.
__global__ __launch_bounds__(256) void sort_chunks_kernel(const BucketRow* __restrict__ rows, RowSet rs, int p, int nb_log2,
                                                         uint32_t cap_chunks, int wgs_per_row) {
    __shared__ uint32_t sorted[4][kChunkRecords];
    __shared__ uint32_t hist[4][16];
    const BucketRow row = rows[rs.index(blockIdx.x / (uint32_t)wgs_per_row)];
    if (!row.area) return;
    const uint32_t handed = gload4(row.cursor);
    const uint32_t nrec = handed < cap_chunks * kChunkRecords ? handed : cap_chunks * kChunkRecords;  // reservations are multiples of 64, the capacity of 1024
    const uint32_t nchunks = (nrec + kChunkRecords - 1u) / kChunkRecords;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, nb = 1u << nb_log2;
    const int tshift = p - nb_log2;
    for (uint32_t c = (blockIdx.x % wgs_per_row) * 4u + wave; c < nchunks; c += (uint32_t)wgs_per_row * 4u) {
        const uint32_t f = nrec - c * kChunkRecords < kChunkRecords ? nrec - c * kChunkRecords : kChunkRecords;
        uint32_t e[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t pos = (uint32_t)i * 256u + lane * 4u;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (pos < f) v = gload16(row.area + (size_t)c * kChunkRecords + pos);
            e[4 * i] = v.x, e[4 * i + 1] = v.y, e[4 * i + 2] = v.z, e[4 * i + 3] = v.w;
        }
        if (lane < 16) hist[wave][lane] = 0;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (e[i] >> 24) atomicAdd(&hist[wave][(e[i] & 0xFFFFFFu) >> tshift], 1u);
        __builtin_amdgcn_wave_barrier();
        // exclusive prefix over the (at most 16) tiles: lanes 0..15
        const uint32_t mine = lane < nb ? hist[wave][lane] : 0u;
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d);
            if (lane >= (uint32_t)d) incl += up;
        }
        const uint32_t total = __shfl(incl, (int)nb - 1);
        __builtin_amdgcn_wave_barrier();
        if (lane < nb) {
            hist[wave][lane] = incl - mine;
            ((DD_GLOBAL uint16_t*)row.seg)[(size_t)c * 16u + lane] = (uint16_t)(incl - mine);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (e[i] >> 24) sorted[wave][atomicAdd(&hist[wave][(e[i] & 0xFFFFFFu) >> tshift], 1u)] = e[i];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t pos = (uint32_t)i * 256u + lane * 4u;
            if (pos < total) {
                const uint4 v = *reinterpret_cast<const uint4*>(&sorted[wave][pos]);  // past `total`: stale, never read
                gstore16(row.area + (size_t)c * kChunkRecords + pos, v);
            }
        }
        if (lane == 0) gstore4(row.fill + c, total);
        __builtin_amdgcn_wave_barrier();
    }
}

// One workgroup per (row, index tile); LDS: the tile (64 KiB, or m bytes if smaller); two workgroups per
// CU.  BINS = false (filtered epochs): a wave takes every 16th chunk of the row's stream, U at a time, and reads only the
// segment of its own tile that sort_chunks_kernel left; segment headers, records and the LDS work of three consecutive steps
// overlap.  BINS = true (the first epoch's binned tiles, scatter_first_bin_kernel): chunk C = 16 bins of kBinCap records'
// room, counts in seg[C][16]; unit u = (chunk, copy of this tile's bin); the 512-record pieces of a unit go round the 16
// waves, so every wave has a 2 KiB piece in flight while it applies the previous one; the rho = 1 updates, which left a bit
// instead of a record, are applied when the tile is written back.
template <bool BINS>
__global__ __launch_bounds__(1024) void replay_kernel(const BucketRow* __restrict__ rows, RowSet rs, int p, int logg,
                                                     int nb_log2, uint32_t cap_chunks) {
    lds_starts_at_zero();
    const uint32_t nb = 1u << nb_log2;
    const uint32_t within = blockIdx.x >> 3, xcd = blockIdx.x & 7u;
    const uint32_t r = (within >> nb_log2) * 8u + xcd, b = within & (nb - 1u);
    if (r >= (uint32_t)rs.nrows) return;
    const BucketRow row = rows[rs.index(r)];
    if (!row.area) return;
    const uint32_t handed = gload4(row.cursor);
    if (handed == 0u) return;  // nothing was recorded for this row in this epoch: registers and filter stand
    const uint32_t nrec = handed < cap_chunks * kChunkRecords ? handed : cap_chunks * kChunkRecords;
    const uint32_t nchunks = (nrec + kChunkRecords - 1u) / kChunkRecords;
    const uint32_t tile = 1u << (p - nb_log2);
    uint8_t* const tile_g = row.regs + (size_t)b * tile;
    uint4* l4 = reinterpret_cast<uint4*>(g_lds);
    for (uint32_t i = threadIdx.x; i < (tile >> 4); i += blockDim.x) l4[i] = gload16(tile_g + (size_t)i * 16);
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    auto apply = [&](uint32_t e) {  // null records (rho 0) fall through
        const uint32_t rho = e >> 24, a = e & (tile - 1u);
        const uint32_t w = RegsLds::load32(a);
        if (rho > ((w >> RegsLds::shift(a)) & 0xFFu)) (void)cas_raise<RegsLds>(a, w, rho);
    };
    constexpr int U = 4;
    const DD_GLOBAL uint16_t* seg = (const DD_GLOBAL uint16_t*)row.seg;
    // U records of a lane in three sweeps -- all register words read, all first compare-and-swaps issued, then the
    // (rare) retries -- instead of read / compare / CAS record by record: the LDS round trips of one lane's records
    // overlap (an LDS atomic orders every later LDS access of the wave behind it, so the record-by-record form ran
    // them back to back; while the registers are still filling, half the records raise one).  A word changed in
    // between -- by a neighbour, or by this lane's previous record -- fails its CAS and is retried from the value
    // that came back.
    auto apply_u = [&](const uint32_t (&e)[U]) {
        uint32_t wd[U];
        uint32_t retry = 0;  // bit i: record i's first CAS found another value than the one read
#pragma unroll
        for (int i = 0; i < U; ++i) wd[i] = RegsLds::load32(e[i] & (tile - 1u));
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const uint32_t a = e[i] & (tile - 1u), rho = e[i] >> 24, sh = RegsLds::shift(a), cur = (wd[i] >> sh) & 0xFFu;
            if (rho > cur) {
                const uint32_t prev = RegsLds::cas32(a, wd[i], wd[i] + ((rho - cur) << sh));
                if (prev != wd[i]) retry |= 1u << i;
                wd[i] = prev;
            }
        }
        if (__any(retry != 0u)) {
#pragma unroll
            for (int i = 0; i < U; ++i)
                if ((retry >> i) & 1u) (void)cas_raise<RegsLds>(e[i] & (tile - 1u), wd[i], e[i] >> 24);
        }
    };
    if (BINS) {
        const int cshift = 4 - nb_log2;
        const uint32_t nunits = (nrec / kBinChunkRecords) << cshift;
        const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        struct Piece {
            uint4 a, b;
            uint32_t off, cnt;  // wave-uniform: the piece's first record within its bin, the bin's records
        };
        auto header = [&](uint32_t u) -> uint32_t {  // records in unit u's bin
            return u < nunits ? (uint32_t)seg[(size_t)(u >> cshift) * 16u + ((b << cshift) | (u & ((1u << cshift) - 1u)))] : 0u;
        };
        auto issue = [&](uint32_t u, uint32_t cnt, Piece& P) {
            P.off = ((wave - u) & 15u) * 512u;
            P.cnt = __builtin_amdgcn_readfirstlane(cnt);
            P.a = P.b = make_uint4(0, 0, 0, 0);
            if (P.off >= P.cnt) return;
            const uint32_t* base = row.area + (size_t)(u >> cshift) * kBinChunkRecords + ((b << cshift) | (u & ((1u << cshift) - 1u))) * kBinCap + P.off;
            if (P.off + 4u * lane < P.cnt) P.a = gload16(base + 4u * lane);  // (a quad may straddle the bin's last record: still inside its region)
            if (P.off + 256u + 4u * lane < P.cnt) P.b = gload16(base + 256u + 4u * lane);
        };
        uint32_t c0 = header(0), c1 = header(1), c2 = header(2);
        Piece cur, nxt;
        issue(0, c0, cur);
        for (uint32_t u = 0; u < nunits; ++u) {
            issue(u + 1u, c1, nxt);
            c1 = c2;
            c2 = header(u + 3u);
            if (cur.off < cur.cnt) {
                uint32_t ea[U] = {cur.a.x, cur.a.y, cur.a.z, cur.a.w}, eb[U] = {cur.b.x, cur.b.y, cur.b.z, cur.b.w};
                if (cur.off + 512u > cur.cnt) {  // the bin's last piece: what lies behind its last record is nulled
                    const uint32_t d = cur.off + 4u * lane;
#pragma unroll
                    for (int j = 0; j < U; ++j) {
                        ea[j] = d + (uint32_t)j < cur.cnt ? ea[j] : 0u;
                        eb[j] = d + 256u + (uint32_t)j < cur.cnt ? eb[j] : 0u;
                    }
                }
                apply_u(ea);
                apply_u(eb);
            }
            cur = nxt;
        }
    } else {
    struct Head {
        uint32_t st[U], en[U];
    };
    struct Recs {
        uint32_t r0[U], r1[U];
    };
    auto heads = [&](uint32_t c, Head& h) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t cc = c + 16u * u;
            h.st[u] = h.en[u] = 0;
            if (cc < nchunks) {
                if (nb > 1u) {
                    h.st[u] = seg[(size_t)cc * 16u + b];
                    h.en[u] = b + 1u < nb ? (uint32_t)seg[(size_t)cc * 16u + b + 1u] : gload4(row.fill + cc);
                } else {  // unsorted single-tile rows: the raw stream, null records included
                    h.en[u] = nrec - cc * kChunkRecords < kChunkRecords ? nrec - cc * kChunkRecords : kChunkRecords;
                }
            }
        }
    };
    auto records = [&](uint32_t c, const Head& h, Recs& v) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t* base = row.area + (size_t)(c + 16u * u) * kChunkRecords;
            const uint32_t i0 = h.st[u] + lane, i1 = i0 + 64u;
            v.r0[u] = i0 < h.en[u] ? gload4(base + i0) : 0u;
            v.r1[u] = i1 < h.en[u] ? gload4(base + i1) : 0u;
        }
    };
    const uint32_t step = 16u * U, c_first = threadIdx.x >> 6;
    Head h1, h2;
    Recs v1;
    heads(c_first, h1);
    heads(c_first + step, h2);
    records(c_first, h1, v1);
    for (uint32_t c = c_first; c < nchunks; c += step) {
        const Head h0 = h1;
        const Recs v0 = v1;
        h1 = h2;
        heads(c + 2u * step, h2);      // headers two steps ahead
        records(c + step, h1, v1);     // records one step ahead
        // The 2U records of the step, U at a time (apply_u; all 2U together need 75+ VGPRs, and above 64 only one
        // 1024-thread workgroup fits a CU)
        apply_u(v0.r0);
        apply_u(v0.r1);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t* base = row.area + (size_t)(c + 16u * u) * kChunkRecords;
            for (uint32_t i = h0.st[u] + 128u + lane; i < h0.en[u]; i += 64u) apply(gload4(base + i));  // longer than twice the expected size
        }
    }
    }
    __syncthreads();
    if (BINS) {
        // the updates with rho = 1 left no record, only a bit (scatter_first_bin_kernel): a register still 0 behind
        // a set bit becomes 1.  Thread i holds registers 16 i .. 16 i + 15 of the tile = halfword i of the tile's bits.
        const DD_GLOBAL uint16_t* bits = (const DD_GLOBAL uint16_t*)row.ones + (((size_t)b * tile) >> 4);
        for (uint32_t i = threadIdx.x; i < (tile >> 4); i += blockDim.x) {
            const uint32_t h = bits[i];
            if (!h) continue;
            uint4 v = l4[i];
            auto raise = [](uint32_t w, uint32_t nib) {
                const uint32_t set = ((nib & 0xFu) * 0x00204081u) & 0x01010101u;                 // bit j of the nibble -> byte j
                const uint32_t zero = (~(w + 0x7F7F7F7Fu) & 0x80808080u) >> 7;                    // 1 in every byte that is 0 (bytes < 128)
                return w | (set & zero);
            };
            v.x = raise(v.x, h), v.y = raise(v.y, h >> 4), v.z = raise(v.z, h >> 8), v.w = raise(v.w, h >> 12);
            l4[i] = v;
        }
        __syncthreads();
    }
    for (uint32_t i = threadIdx.x; i < (tile >> 4); i += blockDim.x) gstore16(tile_g + (size_t)i * 16, l4[i]);
    // the tile's part of the filter: minimum of every group of 2^logg registers
    const uint32_t G = 1u << logg;
    auto group_min = [&](uint32_t f) {
        uint32_t lo = 0xFFu;
        if (G >= 4u) {
            for (uint32_t w = 0; w < G; w += 4) {
                const uint32_t mn = min4(*reinterpret_cast<const uint32_t*>(g_lds + f * G + w));
                lo = mn < lo ? mn : lo;
            }
        } else {
            for (uint32_t w = 0; w < G; ++w) lo = g_lds[f * G + w] < lo ? g_lds[f * G + w] : lo;
        }
        return lo;
    };
    // (4-bit entries, saturating at 15, two register groups per byte: measured better than byte entries at log2m 18, 19, 20)
    const uint32_t ngroups = tile >> logg;
    uint8_t* const flt = row.filter + ((((size_t)b * tile) >> logg) >> 1);
    for (uint32_t f = threadIdx.x; f < (ngroups >> 1); f += blockDim.x) {
        const uint32_t a = group_min(2u * f), c = group_min(2u * f + 1u);
        flt[f] = (uint8_t)((a < 15u ? a : 15u) | ((c < 15u ? c : 15u) << 4));
    }
}

// the stream cursors of all rows back to zero for the next epoch (replay's workgroups of a row cannot do
// it themselves: its sibling tiles may still be reading the cursor)
__global__ void reset_cursors_kernel(const BucketRow* __restrict__ rows, RowSet rs) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rs.nrows && rows[rs.index((uint32_t)r)].area) gstore4(rows[rs.index((uint32_t)r)].cursor, 0u);
}

// Dynamic LDS above 64 KiB must be allowed per kernel AND per device (a process may hold contexts on
// several GPUs); remembered in one bit per device id.
void allow_full_lds(const void* kern, std::atomic<unsigned long long>& done, int static_lds_bytes = 0) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_relaxed) & bit)) {
        // dynamic + the kernel's static LDS must stay within the CU's 160 KiB or the call is refused
        if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - static_lds_bytes) != hipSuccess)
            (void)hipGetLastError();  // not sticky: a launch that needs the room will report it
        done.fetch_or(bit, std::memory_order_relaxed);
    }
}

template <int KC, bool CANON>
void launch_one(const SweepGenome* genomes, const SweepJob* jobs, int njobs, const SweepPlan& plan,
                hipStream_t st) {
    auto kern = sweep_kernel<KC, CANON>;
    static std::atomic<unsigned long long> attr_done{0};  // one bit per device: the attribute is per device
    allow_full_lds(reinterpret_cast<const void*>(kern), attr_done);
    hipLaunchKernelGGL(kern, dim3((unsigned)njobs), dim3((unsigned)plan.threads),
                       (size_t)plan.lds_bytes, st, genomes, jobs, plan.log2m);
}

}  // namespace

int sweep_max_lds_bytes() { return 160 * 1024; }

void launch_bitmap(const SweepGenome* genomes, const SweepJob* jobs, int njobs, int kfirst, int klast,
                   int canonical, hipStream_t st) {
    if (njobs <= 0 || kfirst < 1 || klast > kBitmapMaxK || klast < kfirst) return;
    const size_t lds = (size_t)(bitmap_offset(klast) + bitmap_words(klast) - bitmap_offset(kfirst)) * 4;
    static std::atomic<unsigned long long> attr_done[2] = {{0}, {0}};
    if (canonical) {
        allow_full_lds(reinterpret_cast<const void*>(bitmap_kernel<true>), attr_done[0], 256);
        hipLaunchKernelGGL(bitmap_kernel<true>, dim3((unsigned)njobs), dim3(1024), lds, st, genomes, jobs);
    } else {
        allow_full_lds(reinterpret_cast<const void*>(bitmap_kernel<false>), attr_done[1], 256);
        hipLaunchKernelGGL(bitmap_kernel<false>, dim3((unsigned)njobs), dim3(1024), lds, st, genomes, jobs);
    }
}

void launch_bitmap_finish(const SweepGenome* genomes, int ngenomes, int kfirst, int klast, int kmin, int log2m,
                          hipStream_t st) {
    if (ngenomes <= 0 || klast < kfirst) return;
    const int tile_log2 = std::min(log2m, 16);
    auto kern = bitmap_finish_kernel<true>;
    hipLaunchKernelGGL(kern, dim3((unsigned)(klast - kfirst + 1), (unsigned)ngenomes, 1u << (log2m - tile_log2)), dim3(1024),
                       (size_t)1 << tile_log2, st, genomes, kfirst, kmin, log2m, tile_log2);
}

void launch_bigmap(const SweepGenome* genomes, const SweepJob* jobs, int njobs, int canonical, hipStream_t st) {
    if (njobs <= 0) return;
    static std::atomic<unsigned long long> attr_done[2] = {{0}, {0}};
    if (canonical) {
        allow_full_lds(reinterpret_cast<const void*>(bigmap_kernel<true>), attr_done[0]);
        hipLaunchKernelGGL(bigmap_kernel<true>, dim3((unsigned)njobs), dim3(1024), (size_t)kBigmapSliceWords * 4, st, genomes, jobs);
    } else {
        allow_full_lds(reinterpret_cast<const void*>(bigmap_kernel<false>), attr_done[1]);
        hipLaunchKernelGGL(bigmap_kernel<false>, dim3((unsigned)njobs), dim3(1024), (size_t)kBigmapSliceWords * 4, st, genomes, jobs);
    }
}

void launch_bigmap_finish(const SweepGenome* genomes, int ngenomes, int kfirst, int klast, int kmin, int log2m,
                          int canonical, hipStream_t st) {
    if (ngenomes <= 0 || klast < kfirst) return;
    // 128 KiB tiles, one workgroup per CU: every workgroup hashes the row's whole set (up to 2 M k-mers), so fewer,
    // larger tiles are less work
    const int tile_log2 = std::min(log2m, 17);
    const dim3 grid((unsigned)(klast - kfirst + 1), (unsigned)ngenomes, 1u << (log2m - tile_log2));
    static std::atomic<unsigned long long> attr_done[2] = {{0}, {0}};
    allow_full_lds(reinterpret_cast<const void*>(canonical ? bigmap_finish_kernel<true> : bigmap_finish_kernel<false>), attr_done[canonical ? 0 : 1]);
    if (canonical)
        hipLaunchKernelGGL(bigmap_finish_kernel<true>, grid, dim3(1024), (size_t)1 << tile_log2, st, genomes, kfirst, kmin, log2m, tile_log2);
    else
        hipLaunchKernelGGL(bigmap_finish_kernel<false>, grid, dim3(1024), (size_t)1 << tile_log2, st, genomes, kfirst, kmin, log2m, tile_log2);
}

// One scatter launch of a k class (log2m >= 17).  first_epoch: every register of the call is still zero -- the unfiltered,
// binned form (scatter_first_bin_kernel); later epochs: the filtered form (scatter_kernel).
void launch_scatter(const SweepGenome* genomes, const SweepJob* jobs, int njobs, int kclass, const SweepPlan& plan,
                    const ScatterParams& sp, hipStream_t st, bool first_epoch) {
    if (njobs <= 0) return;
#define DD_SCATTER(KC, CN)                                                                                                      \
    do {                                                                                                                        \
        static std::atomic<unsigned long long> attr_done[2] = {{0}, {0}};                                                       \
        if (first_epoch) {                                                                                                      \
            auto kern = scatter_first_bin_kernel<KC, CN>;                                                                       \
            allow_full_lds(reinterpret_cast<const void*>(kern), attr_done[0]);                                                  \
            hipLaunchKernelGGL(kern, dim3((unsigned)njobs), dim3(1024),                                                         \
                               (size_t)kBinLdsBytes + (((size_t)1 << std::min(plan.log2m, kOnesLog2Max)) >> 3), st, genomes, jobs, plan.log2m, sp); \
        } else {                                                                                                                \
            auto kern = scatter_kernel<KC, CN>;                                                                                 \
            allow_full_lds(reinterpret_cast<const void*>(kern), attr_done[1]);                                                  \
            hipLaunchKernelGGL(kern, dim3((unsigned)njobs), dim3((unsigned)plan.threads), (size_t)plan.lds_bytes, st, genomes, jobs, plan.log2m, sp); \
        }                                                                                                                       \
    } while (0)
#define DD_SCATTER_KC(CN)                        \
    do {                                         \
        if (kclass == 0) DD_SCATTER(0, CN);      \
        else if (kclass == 1) DD_SCATTER(1, CN); \
        else if (kclass == 3) DD_SCATTER(3, CN); \
        else DD_SCATTER(2, CN);                  \
    } while (0)
    if (plan.canonical) DD_SCATTER_KC(true);
    else DD_SCATTER_KC(false);
#undef DD_SCATTER_KC
#undef DD_SCATTER
}

// first_epoch: the records are the binned tiles scatter_first_bin_kernel left (+ the rows' rho = 1 bits); else the filtered
// scatter's dense stream, whose chunks are sorted by index tile first
void launch_replay(const BucketRow* rows, int ngenomes, int K, int k0, int nks, const SweepPlan& plan, hipStream_t st, bool first_epoch) {
    const RowSet rs{K, k0, nks, ngenomes * nks, 0};
    if (rs.nrows <= 0) return;
    const size_t tile = (size_t)1 << (plan.log2m - plan.nb_log2);
    const unsigned blocks = (unsigned)((rs.nrows + 7) / 8) * 8u << plan.nb_log2;
    if (!first_epoch) {
        const int wgs_per_row = 32;
        hipLaunchKernelGGL(sort_chunks_kernel, dim3((unsigned)rs.nrows * wgs_per_row), dim3(256), 0, st, rows, rs, plan.log2m,
                           plan.nb_log2, plan.cap_chunks, wgs_per_row);
    }
    static std::atomic<unsigned long long> attr_done[2] = {{0}, {0}};
    if (first_epoch) {
        allow_full_lds(reinterpret_cast<const void*>(replay_kernel<true>), attr_done[1]);
        hipLaunchKernelGGL(replay_kernel<true>, dim3(blocks), dim3(1024), tile, st, rows, rs, plan.log2m, plan.logg, plan.nb_log2, plan.cap_chunks);
    } else {
        allow_full_lds(reinterpret_cast<const void*>(replay_kernel<false>), attr_done[0]);
        hipLaunchKernelGGL(replay_kernel<false>, dim3(blocks), dim3(1024), tile, st, rows, rs, plan.log2m, plan.logg, plan.nb_log2, plan.cap_chunks);
    }
    hipLaunchKernelGGL(reset_cursors_kernel, dim3((unsigned)(rs.nrows + 255) / 256), dim3(256), 0, st, rows, rs);
}

void launch_sweep(const SweepGenome* genomes, const SweepJob* jobs, int njobs, int kclass,
                  const SweepPlan& plan, hipStream_t st) {
    if (njobs <= 0) return;
#define DD_DISPATCH_KC(CN)                                                          \
    do {                                                                            \
        if (kclass == 0) launch_one<0, CN>(genomes, jobs, njobs, plan, st);         \
        else if (kclass == 1) launch_one<1, CN>(genomes, jobs, njobs, plan, st);    \
        else if (kclass == 3) launch_one<3, CN>(genomes, jobs, njobs, plan, st);    \
        else launch_one<2, CN>(genomes, jobs, njobs, plan, st);                     \
    } while (0)
    if (plan.canonical) DD_DISPATCH_KC(true);
    else DD_DISPATCH_KC(false);
#undef DD_DISPATCH_KC
}

}  // namespace dd
