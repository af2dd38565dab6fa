// dd_ginflate.hip -- gzip inflated on the GPU, straight into the FASTA buffer K0 reads: BGZF blocks, and ordinary .gz files.
//
// Real genome directories hold .fa.gz (/root/reference/lib/species_specifics.py:93) and every `dashing sketch` job of
// the reference inflates its input again (lib/huffman_dandd.py:214-218: one process per k).  On the host ten 50 Mbp
// .gz files at once are bound by the CPUs' aggregate inflate rate (3.8-5.6 Gbp/s on 16 cores, profiles/r03_ingest_gzip.txt)
// while the kernels behind them run at 20-36 Gbp/s.  A BGZF file (bgzip, htslib) is a sequence of independent gzip
// members of <= 64 KiB of text each, every one saying its own compressed size: they can be decoded anywhere, in any order.
//
// One wave per block.  A deflate stream is a serial thing -- every Huffman code starts where the previous one ended --,
// but the wave does not walk it one symbol at a time: LANE i DECODES THE SYMBOL THAT WOULD START AT BIT i of a 64-bit
// window (literal/length code, extra bits, distance code, extra bits: two table gathers from LDS), a walk follows the
// chain of symbols that really are there (round 4: scalar, one v_readlane each; round 5: by all lanes at once, pointer
// jumping -- parallel_walk), and the lanes of the 64-byte output batch look up the symbol they belong to and note where
// their byte comes from -- a literal, or an earlier position of the text, which may lie inside the batch itself (resolved
// by pointer jumping when the batch leaves, round 5).  A batch costs ONE load and ONE contiguous store.  Block headers,
// code tables (built code by code, the replicas of a code spread over the lanes), codes longer than the tables' 10 bits
// and copies that do not fit what is left of a batch go one symbol at a time.
// The block's TEXT lives where it is going -- the FASTA buffer in HBM --, not in LDS: a wave needs 9.25 KiB of LDS (its
// code tables), seventeen waves share a CU, and the serial chain of one block hides behind sixteen others.  What bounds a
// launch is the CU's scalar issue slot: instructions per symbol (profiles/r04_bgzf.txt: 10 ms per block -> 3.1).
// A copy reads what earlier batches of the same wave stored: a batch waits for the stores before it (issued a batch ago:
// free) and reads past the vector L1 (sc1).
// Anything that is not a valid block -- bad code lengths, a distance before the block's start, a length that does not
// match the member's ISIZE, a text whose CRC-32 is not the member's (the wave reads its text back: text_crc) -- raises the
// launch's error count and the caller runs the call again with the host decoder (dd_inflate.h), which words the error.
//
// ORDINARY .gz files (ONE gzip member: what `gzip` and the sequence archives write) take the second half of this file
// (launch_gunzip_members): find_starts_kernel finds deflate block starts by trial, one per 16-128 KiB range of the
// compressed file; the same decoder (inflate_kernel<3>) decodes every piece between two starts WITHOUT the 32 KiB in
// front of it, into 16-bit symbols -- a byte, or "position p of that unknown window"; piece_maps_kernel /
// group_windows_kernel compose the pieces' window-to-window maps in two levels; translate_kernel turns symbols into text
// in the buffer K0 reads; chunk_crc_kernel checks it against the member's CRC-32 (the host combines the chunks).
// Ten 50 Mbp .gz: 12 Gbp/s through dd_sketch_files against 6 with the host decoder; one 3 Gbp .gz: 14.9 against 4.6
// (profiles/r04_gunzip.txt).  Round 5 (profiles/r05_gunzip.txt): several members per file, four-line FASTQ (dd_fastq.hip),
// gzip -1 from 6.7 to 10.7-12.9 Gbp/s, one 400 Mbp member from 8.0 to 11.
#include "dd_common.h"
#include "dd_kernels.h"

#include <atomic>

namespace dd {
namespace {

extern __shared__ __attribute__((aligned(16))) uint8_t g_lds[];

// (round 5, measured and put back: 9-bit tables -- 2 KiB each, 5.25 KiB per wave instead of 9.25, 24-28 waves per CU instead of
// 17 -- gave ten gzip -1 files 8.0 -> 8.3 Gbp/s and took one 400 Mbp gzip -6 file from 10.3 to 9.4 (codes of 10 bits go through
// decode_slow); BGZF and gzip -6 directories unchanged: profiles/r05_gunzip.txt)
constexpr int FAST = 10;
constexpr uint32_t kTableBytes = 4u << FAST;
constexpr uint32_t kLitInfo = 0;                  // u32[1 << FAST]: literal / length code table (FAST-bit lookup)
constexpr uint32_t kDistInfo = kLitInfo + kTableBytes;  // u32[1 << FAST]: distance code table
constexpr uint32_t kLitCount = kDistInfo + kTableBytes; // u16[16] + u16[288]: codes longer than FAST bits, puff-style
constexpr uint32_t kLitSymbol = kLitCount + 32u;
constexpr uint32_t kDistCount = kLitSymbol + 576u;
constexpr uint32_t kDistSymbol = kDistCount + 32u;
constexpr uint32_t kLens = kDistSymbol + 64u;     // u8[320]: code lengths while a table is built
constexpr uint32_t kClInfo = kLens + 320u;        // u16[128]: code-length code table (7-bit lookup)
constexpr uint32_t kInflateLds = (kClInfo + 256u + 15u) & ~15u;   // 9.25 KiB: seventeen one-wave workgroups per CU
static_assert(kTableBytes >= 1024u, "text_crc keeps its 256-entry table in the literal table's place");

__constant__ uint16_t c_len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t c_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t c_dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t c_dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t c_cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

DD_D uint32_t& l32(uint32_t off) { return *reinterpret_cast<uint32_t*>(g_lds + off); }
DD_D uint16_t& l16(uint32_t off) { return *reinterpret_cast<uint16_t*>(g_lds + off); }
DD_D uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane(v); }
DD_D uint64_t uni64(uint64_t v) { return ((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v); }
DD_D uint32_t gload1(const uint8_t* p) { return *(const DD_GLOBAL uint8_t*)p; }
DD_D uint32_t lane_value(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }

// The wave's bit reader: every lane holds the same state.  The block's compressed words come through the lanes
// themselves: lane j keeps word (window + j) of the input, a refill of the bit buffer is ONE v_readlane, and the next
// window of 256 bytes is asked for (one coalesced load) when the current one is entered, a whole window ahead of need.
// (Words taken from HBM as they were needed cost a memory round trip per 32 bits of input: 6.5 ms per block; a ring in
// LDS costs eight instructions per word and 2 KiB per wave; profiles/r04_bgzf.txt.)
struct WBits {
    const uint32_t* w;     // the input as 4-byte aligned words
    uint32_t wi;           // next word to put into `ahead`
    uint32_t nwords;       // words that belong to the block (beyond: zeros)
    uint32_t cur, nxt;     // this lane's word of the window that holds word wi, and of the one after it
    uint64_t buf;
    int cnt;
    uint32_t ahead;        // W[wi - 1], already taken from the window
    DD_D uint32_t fetch(uint32_t first) const {   // this lane's word of the 64 that start at `first`
        const uint32_t i = first + (threadIdx.x & 63u);
        return i < nwords ? gload4(w + i) : 0u;
    }
    DD_D uint32_t word() {   // W[wi++]
        const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)(wi & 63u));
        ++wi;
        if ((wi & 63u) == 0u) {
            cur = nxt;
            nxt = fetch(wi + 64u);
        }
        return v;
    }
    DD_D void seek(uint32_t q) {   // the next word() is W[q]
        wi = q;
        cur = fetch(q & ~63u);
        nxt = fetch((q & ~63u) + 64u);
    }
    DD_D void start_at(uint32_t q, uint32_t r) {   // the reader stands at bit r (< 32) of W[q]
        seek(q);
        buf = word();
        ahead = word();
        buf >>= r;
        cnt = 32 - (int)r;
        refill();
    }
    DD_D void init(const uint8_t* p, uint32_t nbytes) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(p);
        const uint32_t skip = (uint32_t)(a & 3u);
        w = reinterpret_cast<const uint32_t*>(a - skip);
        nwords = (skip + nbytes + 3u) / 4u;
        start_at(0, 8u * skip);
    }
    DD_D uint64_t bit_pos() const { return (uint64_t)(wi - 1u) * 32u - (uint64_t)cnt; }   // bits of W consumed (`ahead` is read but not in the buffer)
    DD_D void refill() {   // from >= 0 valid bits to >= 32
        buf |= (uint64_t)ahead << cnt;
        cnt += 32;
        ahead = word();
    }
    DD_D void need() { if (cnt <= 32) refill(); }   // more than 32 valid bits afterwards
    DD_D uint32_t peek(int k) const { return (uint32_t)buf & ((1u << k) - 1u); }   // k <= 16
    DD_D void drop(int k) { buf >>= k; cnt -= k; }
    DD_D uint32_t take(int k) {
        if (cnt < k) refill();
        const uint32_t v = peek(k);
        drop(k);
        return v;
    }
    // bytes of the input consumed so far, counting a partly used byte as consumed
    DD_D uint32_t bytes_used(const uint8_t* p) const {
        const uintptr_t base = reinterpret_cast<uintptr_t>(w);
        const uint64_t bits = (uint64_t)(wi - 1) * 32ull - (uint64_t)cnt;   // (`ahead` is read but not in the buffer)
        return (uint32_t)((bits + 7ull) / 8ull - (reinterpret_cast<uintptr_t>(p) - base));
    }
};

// A canonical Huffman code from the lengths at g_lds[kLens + first .. + n): info table (FAST-bit lookup) at `info`, the
// puff-style count / symbol arrays at `cnt_off` / `sym_off` for longer codes.  kind: 0 literal/length tree, 1 distance tree.
// Wave-uniform; returns false when the lengths are not a usable code.
__device__ __noinline__ bool build_table(uint32_t first, int n, int kind, uint32_t info, uint32_t cnt_off, uint32_t sym_off) {
    const uint32_t lane = threadIdx.x & 63u;
    // count[l]: lanes 0..15 hold one length each
    uint32_t mine = 0;
    if (lane < 16u)
        for (int i = 0; i < n; ++i) mine += (g_lds[kLens + first + i] == lane) ? 1u : 0u;
    if (lane < 16u) l16(cnt_off + 2u * lane) = (uint16_t)mine;
    for (uint32_t i = lane; i < (1u << FAST); i += 64u) l32(info + 4u * i) = 0;
    __builtin_amdgcn_wave_barrier();
    int left = 1, nonzero = 0;
    // next canonical code and next index into symbol[] of each length: lane l keeps length l's pair
    uint32_t my_code = 0, my_off = 0;
    uint32_t c = 0, o = 0;
    for (uint32_t l = 1; l <= 15u; ++l) {
        const uint32_t cl = uni(l16(cnt_off + 2u * l));
        left = (left << 1) - (int)cl;
        if (left < 0) return false;
        nonzero += (int)cl;
        if (lane == l) my_code = c, my_off = o;
        c = (c + cl) << 1;
        o += cl;
    }
    if (nonzero == 0) return kind == 1;   // (a block of literals only may come with no distance code at all: RFC 1951, 3.2.7)
    if (left > 0 && !(kind == 1 && nonzero == 1)) return false;   // incomplete: only a one-code distance tree may be
    // every symbol in turn (uniform), its table replicas spread over the lanes
    for (int i = 0; i < n; ++i) {
        const uint32_t l = uni((uint32_t)g_lds[kLens + first + i]);
        if (!l) continue;
        const uint32_t cd = lane_value(my_code, l), at = lane_value(my_off, l);
        if (lane == l) ++my_code, ++my_off;
        if (lane == 0) l16(sym_off + 2u * at) = (uint16_t)i;
        if (l > (uint32_t)FAST) continue;
        uint32_t v;
        if (kind == 0) {
            if (i < 256) v = l | (1u << 4) | ((uint32_t)i << 11);
            else if (i == 256) v = l | (2u << 4);
            else if (i <= 285) v = l | (3u << 4) | ((uint32_t)c_len_extra[i - 257] << 7) | ((uint32_t)c_len_base[i - 257] << 11);
            else v = 0;   // 286, 287: never valid in a stream
        } else {
            v = i <= 29 ? (l | (3u << 4) | ((uint32_t)c_dist_extra[i] << 7) | ((uint32_t)c_dist_base[i] << 11)) : 0u;
        }
        const uint32_t rev = __builtin_bitreverse32(cd) >> (32u - l);
        for (uint32_t f = rev + (lane << l); f < (1u << FAST); f += 64u << l) l32(info + 4u * f) = v;
    }
    __builtin_amdgcn_wave_barrier();
    return true;
}

// a code longer than FAST bits (or an invalid one), from the low bits of `bits`: walk the lengths, one bit at a time.
// -> symbol << 4 | code length, or ~0u when there is no such code.  (The bit reader stays in the caller's registers:
// handing it over by reference put it, and with it every shift of the hot loop, into scratch memory.)
__device__ __noinline__ uint32_t decode_slow(uint64_t bits, uint32_t cnt_off, uint32_t sym_off) {
    int code = 0, first = 0, index = 0;
    for (int l = 1; l <= 15; ++l) {
        code |= (int)(bits & 1ull);
        bits >>= 1;
        const int c = (int)uni(l16(cnt_off + 2u * (uint32_t)l));
        if (code - c < first) return (uni(l16(sym_off + 2u * (uint32_t)(index + (code - first)))) << 4) | (uint32_t)l;
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    return ~0u;
}

// ---- CRC-32 of the inflated text (the member's trailer carries it) ----
// x^(2^n) mod P for n = 0..31 in zlib's reflected notation (bit 31 = x^0), P = 0xedb88320: each entry is the square of
// the one before (multmodp below); generated by squaring 0x40000000 (= x^1).
__constant__ uint32_t c_x2n[32] = {0x40000000u, 0x20000000u, 0x08000000u, 0x00800000u, 0x00008000u, 0xedb88320u, 0xb1e6b092u, 0xa06a2517u, 0xed627daeu, 0x88d14467u, 0xd7bbfe6au, 0xec447f11u, 0x8e7ea170u, 0x6427800eu, 0x4d47bae0u, 0x09fe548fu, 0x83852d0fu, 0x30362f1au, 0x7b5a9cc3u, 0x31fec169u, 0x9fec022au, 0x6c8dedc4u, 0x15d6874du, 0x5fde7a4eu, 0xbad90e37u, 0x2e4e5eefu, 0x4eaba214u, 0xa8a472c0u, 0x429a969eu, 0x148d302au, 0xc40ba6d0u, 0xc4e22c3cu};

// a(x) * b(x) mod P.  `a` is wave-uniform (the loop's exit is), b is per lane.
DD_D uint32_t multmodp(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (uint32_t m = 0x80000000u;; m >>= 1) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1u)) == 0u) break;
        }
        b = (b >> 1) ^ ((b & 1u) ? 0xedb88320u : 0u);
    }
    return p;
}

// The CRC-32 of text[0, n), by the whole wave: lane i takes the i-th 1/64 of the text (the FIRST lane's part is the short
// one, so that every right-hand operand of a combination has a length that depends on the level only), byte-wise with a
// 256-entry table in LDS -- the Huffman tables' place, the block is decoded --, then six levels of
//   crc(A || B) = crc(A) * x^(8 |B|) mod P  ^  crc(B)          (zlib's crc32_combine).
// ~1 % of a block's instructions.  The text is read back past the vector L1; the caller has waited for its stores.
__device__ __noinline__ uint32_t text_crc(const uint8_t* text, uint32_t n) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t i = lane; i < 256u; i += 64u) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? 0xedb88320u : 0u);
        l32(kLitInfo + 4u * i) = c;
    }
    __builtin_amdgcn_wave_barrier();
    auto byte_in = [](uint32_t crc, uint32_t v) { return l32(kLitInfo + 4u * ((crc ^ v) & 255u)) ^ (crc >> 8); };
    uint32_t lo = 0, hi = n, len = 0;
    if (n >= 8192u) {   // (shorter: a file's last block; every lane does all of it)
        len = (n + 63u) / 64u;
        const uint32_t pad = 64u * len - n;   // < 64 <= len
        lo = lane ? lane * len - pad : 0u;
        hi = (lane + 1u) * len - pad;
    }
    uint32_t crc = ~0u, p = lo;
    for (; p < hi && ((reinterpret_cast<uintptr_t>(text) + p) & 3u); ++p) crc = byte_in(crc, gload1_fresh(text + p));
    for (; p + 4u <= hi; p += 4u) {
        const uint32_t v = gload4_fresh(text + p);
        crc = byte_in(crc, v);
        crc = byte_in(crc, v >> 8);
        crc = byte_in(crc, v >> 16);
        crc = byte_in(crc, v >> 24);
    }
    for (; p < hi; ++p) crc = byte_in(crc, gload1_fresh(text + p));
    crc = ~crc;
    if (n >= 8192u) {
        uint32_t c = 0x80000000u;   // x^(8 len): x^0, times x^(2^(k + 3)) for every bit k of len
        for (uint32_t k = 0, m = len; m; m >>= 1, ++k)
            if (m & 1u) c = uni(multmodp(uni(c_x2n[(k + 3u) & 31u]), c));
        for (int j = 0; j < 6; ++j) {
            const uint32_t right = (uint32_t)__shfl_down((int)crc, 1u << j);
            crc = multmodp(c, crc) ^ right;
            c = uni(multmodp(c, c));
        }
    }
    return uni(crc);
}

}  // namespace

// The header of a dynamic-Huffman block (the reader stands behind BTYPE): code-length code, the two trees' code lengths,
// both symbol tables into LDS.  Wave-uniform; false: not a valid header.
DD_D bool dynamic_tables(WBits& b) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t hlit = b.take(5) + 257u, hdist = b.take(5) + 1u, hclen = b.take(4) + 4u;
    if (hlit > 286u || hdist > 30u) return false;
    // the code-length code: 19 lengths of 3 bits, a 7-bit table
    if (lane < 19u) g_lds[kLens + lane] = 0;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = 0; i < hclen; ++i) {
        const uint32_t v = b.take(3);
        if (lane == 0) g_lds[kLens + c_cl_order[i]] = (uint8_t)v;
    }
    __builtin_amdgcn_wave_barrier();
    {
        int left = 1;
        uint32_t mine = 0;   // lane l: how many of the 19 have length l, then its next code
        if (lane < 8u)
            for (int i = 0; i < 19; ++i) mine += ((uint32_t)g_lds[kLens + i] == lane) ? 1u : 0u;
        const uint32_t zeros = lane_value(mine, 0);
        uint32_t c = 0, my_code = 0;
        for (uint32_t l = 1; l <= 7u; ++l) {
            const uint32_t cl = lane_value(mine, l);
            left = (left << 1) - (int)cl;
            if (lane == l) my_code = c;
            c = (c + cl) << 1;
        }
        if (left != 0 && !(zeros == 18u && left > 0)) return false;   // (one code of one bit is tolerated, as zlib does)
        for (uint32_t i = lane; i < 128u; i += 64u) l16(kClInfo + 2u * i) = 0;
        __builtin_amdgcn_wave_barrier();
        for (int i = 0; i < 19; ++i) {
            const uint32_t l = uni((uint32_t)g_lds[kLens + i]);
            if (!l) continue;
            const uint32_t cd = lane_value(my_code, l);
            if (lane == l) ++my_code;
            const uint32_t rev = __builtin_bitreverse32(cd) >> (32u - l);
            for (uint32_t f = rev + (lane << l); f < 128u; f += 64u << l) l16(kClInfo + 2u * f) = (uint16_t)(l | ((uint32_t)i << 4));
        }
        __builtin_amdgcn_wave_barrier();
    }
    // the literal/length and distance code lengths, run-length coded
    uint32_t i = 0, prev = 0;
    while (i < hlit + hdist) {
        b.need();
        const uint32_t e = uni(l16(kClInfo + 2u * b.peek(7)));
        if (!e) return false;
        b.drop((int)(e & 15u));
        const uint32_t s = e >> 4;
        uint32_t rep = 1, val = s;
        if (s == 16u) {
            if (!i) return false;
            val = prev;
            rep = 3u + b.take(2);
        } else if (s == 17u) {
            val = 0;
            rep = 3u + b.take(3);
        } else if (s == 18u) {
            val = 0;
            rep = 11u + b.take(7);
        }
        if (i + rep > hlit + hdist) return false;
        // (lengths of the two trees go to their own places: literal/length at 0.., distance at 288..)
        for (uint32_t r = lane; r < rep; r += 64u) {
            const uint32_t sym = i + r;
            g_lds[kLens + (sym < hlit ? sym : 288u + (sym - hlit))] = (uint8_t)val;
        }
        i += rep;
        prev = val;
    }
    __builtin_amdgcn_wave_barrier();
    if (uni((uint32_t)g_lds[kLens + 256u]) == 0u) return false;   // no end-of-block code
    if (!uni(build_table(0, (int)hlit, 0, kLitInfo, kLitCount, kLitSymbol)) || !uni(build_table(288, (int)hdist, 1, kDistInfo, kDistCount, kDistSymbol))) return false;
    return true;
}

// the piece a wave of the raw-deflate modes works on: entry `idx` of the batch's piece table
struct PieceRef {
    int file;
    uint64_t start, end;   // bit positions (a 3 Gbp assembly's .gz has 7 x 10^9); end = ~0: up to the stream's final block
    uint32_t j, ranges;    // the finder range it starts in, and how many ranges it runs over
};
DD_D bool piece_of(const RawFile* files, int nfiles, const uint64_t* starts, uint32_t idx, PieceRef& r) {
    int f = 0;
    while (f + 1 < nfiles && idx >= uni(files[f + 1].piece0)) ++f;
    const uint32_t j = idx - uni(files[f].piece0), ng = uni(files[f].nguess);
    r.file = f;
    r.start = uni64(starts[idx]);
    r.end = ~0ull;
    r.j = j;
    r.ranges = ng - j;
    if (j >= ng || r.start == ~0ull) return false;
    for (uint32_t k = j + 1; k < ng; ++k) {
        const uint64_t e = uni64(starts[idx - j + k]);
        if (e != ~0ull) {
            r.end = e;
            r.ranges = k - j;
            break;
        }
    }
    return true;
}

// ---- the walk over a window's symbols, all lanes at once (round 5) ------------------------------------------------------
// The scalar walk below follows the chain of symbols one v_readlane and ~20 scalar instructions at a time.  That is the right
// shape for gzip -6 text -- three or four matches per 64-bit window -- and the wrong one for gzip -1 and for anything else
// that is mostly LITERALS: DNA's four letters get 2-bit codes, a window holds up to 32 symbols, and the walk, not the decoding,
// is what a launch spends its instructions on (ten gzip -1 files: 32 ms of inflate kernel against 18 for gzip -6 of the same text).
// The chain is a linked list -- lane i's successor is lane i + len_i -- and "which lanes does the list from lane 0 visit" is
// pointer jumping: every lane keeps the set R of lanes its first 2^k steps visit and the lane J it stands on then; six rounds of
// R |= R[J], J = J[J] (three ds_bpermute each) and lane 0 holds the whole chain, however many symbols it has.  A wave-wide prefix
// sum of the symbols' output lengths (DPP) places them in the batch; the first one that does not fit ends the batch.
// ~70 instructions per window whatever it holds, against ~22 per symbol.  MEASURED (profiles/r05_gunzip.txt): no difference --
// ten gzip -1 files 7.98 / 7.90 / 7.87 Gbp/s with the scalar walk / this one behind windows of five symbols or more / this one
// always, BGZF 16.6-17.1 all three: gzip -1 DNA is not literal runs but SHORT MATCHES (2.3 bits per base: ~3 symbols per window,
// where the two walks cost the same), and swapping 66 scalar instructions for 70 vector ones changes nothing because the kernel
// is bound by neither unit's issue rate but by its waves' serial chains.  -- That was with a third of a gzip -1 member's matches
// kept OUT of the walk (their source might lie inside the batch).  Since flush() resolves those (same round) a window's walkable
// symbols doubled, and this walk became the default: inflate_kernel<3> 9.16 -> 7.73 ms per batch of five gzip -1 members, 9.18 ->
// 8.23 at gzip -6 (rocprofv3).  The scalar walk and the mixed mode are gone from the build (round 6); git history has them.
template <int CTRL, int ROW_MASK>
DD_D uint32_t dpp_add(uint32_t x) {
    return x + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xf, false);
}
DD_D uint32_t wave_inclusive_sum(uint32_t x) {
    x = dpp_add<0x111, 0xf>(x);   // row_shr:1
    x = dpp_add<0x112, 0xf>(x);   // row_shr:2
    x = dpp_add<0x114, 0xf>(x);   // row_shr:4
    x = dpp_add<0x118, 0xf>(x);   // row_shr:8   -> inclusive sums inside every row of 16
    x = dpp_add<0x142, 0xa>(x);   // row_bcast:15 into rows 1 and 3
    x = dpp_add<0x143, 0xc>(x);   // row_bcast:31 into rows 2 and 3
    return x;
}
DD_D uint32_t bperm(uint32_t lane_index, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(lane_index << 2), (int)v); }
// pk: what lane i's symbol is (0: not for the walk; else bits 0..5 its length in bits, 6..14 the bytes it makes).  Out: the
// lanes whose symbols join the batch, each one's slot (osv), the bytes they make, where the walk stands and what it found there
// (pks: 1 = the window is used up, 0 = a symbol the lanes could not finish, else the pk of a symbol the batch has no room for).
DD_D void parallel_walk(uint32_t pk, uint32_t lane, uint32_t used, unsigned long long& mark, uint32_t& pos, uint32_t& outacc, uint32_t& pks, uint32_t& osv) {
    const uint32_t room = 64u - used;
    const bool valid = pk != 0u;
    const uint32_t nxt = lane + (pk & 63u);
    uint32_t J = (valid && nxt < 64u) ? nxt : lane;    // (a lane the walk stops at points at itself)
    uint32_t rlo = lane < 32u ? 1u << lane : 0u, rhi = lane < 32u ? 0u : 1u << (lane - 32u);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const uint32_t jn = bperm(J, J), a = bperm(J, rlo), b = bperm(J, rhi);
        rlo |= a, rhi |= b;
        J = jn;
    }
    const unsigned long long chain = ((unsigned long long)uni(rhi) << 32) | uni(rlo);   // (lane 0's: the walk starts at the window's first bit)
    const unsigned long long vm = __ballot(valid);
    const uint32_t t = 63u - (uint32_t)__builtin_clzll(chain);   // where the chain ends: a symbol that leaves the window, or one the lanes could not finish
    const bool t_valid = (vm >> t) & 1ull;
    const unsigned long long all = chain & vm;
    const uint32_t ol = ((all >> lane) & 1ull) ? pk >> 6 : 0u;
    const uint32_t incl = wave_inclusive_sum(ol), before = incl - ol;
    const unsigned long long bad = __ballot(ol != 0u && incl > room);
    if (bad) {
        const uint32_t fb = (uint32_t)__builtin_ctzll(bad);
        mark = all & ((1ull << fb) - 1ull);
        pos = fb;
        pks = lane_value(pk, fb);
        outacc = lane_value(before, fb);
    } else {
        mark = all;
        pos = t_valid ? t + (lane_value(pk, t) & 63u) : t;
        pks = t_valid ? 1u : 0u;
        outacc = lane_value(incl, 63u);
    }
    if ((mark >> lane) & 1ull) osv = used + before;
}

// MODE 0: grid = BGZF blocks (jobs), one wave each; the text goes out as bytes, the member's ISIZE and CRC-32 are checked.
// MODE 1 / 2: grid = pieces of single-member gzip files (files / starts): raw deflate data from a block start found by
// find_starts_kernel up to the next one, decoded WITHOUT its 32 KiB of history -- a copy that reaches in front of the
// piece yields placeholders 0x8000 | position in that unknown window (pugz's idea; dd_inflate.h does the same on the
// host).  MODE 3, the one pass nearly every piece needs: 16-bit symbols into the piece's own ranges of the file's symbol
// area (5 x the compressed bytes + 32 Ki symbols per range: DNA inflates 3-4 x), lens[idx] = its text; a piece that
// does not fit (runs of N, repeats) is marked in over[idx] instead and gets MODE 1 -- only count its text -- and, once
// the offsets are known, MODE 2 -- write its symbols into the file's arena at abase[idx].  (Counting EVERY piece first
// and writing dense symbols second cost two full passes: 21 of the 47 ms of ten 50 Mbp files.)
// `errors`: blocks / pieces that could not be decoded (the caller falls back to the host).
template <int MODE>
__global__ __launch_bounds__(64) void inflate_kernel(const InflateJob* __restrict__ jobs, const RawFile* __restrict__ files, int nfiles,
                                                     const uint64_t* __restrict__ starts, uint32_t* __restrict__ lens, uint32_t* __restrict__ over,
                                                     const uint32_t* __restrict__ abase, uint32_t* __restrict__ errors) {
    constexpr bool RAW = MODE != 0;
    constexpr bool WRITES = MODE == 2 || MODE == 3;   // (16-bit symbols)
    bool too_long = false;                             // MODE 3: the piece does not fit its ranges
    constexpr uint32_t kLit = RAW ? 0x40000000u : 0x80000000u;   // a batch lane's source: a literal (else an offset in the text; RAW: negative = in front of the piece)
    const uint32_t lane = threadIdx.x & 63u;
    bool ok = true;
    const uint8_t* in;
    uint32_t n, out_len;
    uint64_t piece_end = ~0ull;
    uint8_t* out = nullptr;
    uint16_t* sym = nullptr;
    PieceRef pr{0, 0, ~0ull, 0, 0};
    if (!RAW) {
        const InflateJob job = jobs[blockIdx.x];
        in = job.in, n = job.in_len, out = job.out, out_len = job.out_len;
    } else {
        if (!piece_of(files, nfiles, starts, blockIdx.x, pr)) {
            if (MODE == 3 && lane == 0) lens[blockIdx.x] = 0, over[blockIdx.x] = 0;
            return;
        }
        if (MODE != 3 && uni(over[blockIdx.x]) == 0u) return;   // (the piece fitted its ranges)
        // (MODE 2 writes into the arena at offsets piece_offsets_kernel made from the counted lengths: a batch that already
        // holds an error -- lengths that do not add up to the member's ISIZE, an arena total beyond it: a damaged or wrapped
        // trailer, `cat a.gz b.gz` -- goes to the host decoder anyway and must not be written)
        if (MODE == 2 && uni(__hip_atomic_load(errors, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0u) return;
        const RawFile rf = files[pr.file];
        in = rf.in, n = rf.in_len, piece_end = pr.end;
        if (MODE == 3) {
            const uint64_t cap = (uint64_t)pr.ranges * rf.range_syms;
            out_len = cap < rf.isize ? (uint32_t)cap : rf.isize;
            sym = rf.sym + (size_t)pr.j * rf.range_syms;
        } else {
            out_len = MODE == 1 ? rf.isize : uni(lens[blockIdx.x]);
            if (MODE == 2) {
                const uint32_t ab = uni(abase[blockIdx.x]);
                if ((uint64_t)ab + out_len > rf.isize) {   // the arena holds ISIZE symbols (piece_offsets_kernel has raised the error already)
                    if (lane == 0) atomicAdd(errors, 1u);
                    return;
                }
                sym = rf.arena + ab;
            }
        }
    }
    uint32_t at = 0;
    // The text goes out 64 bytes at a time: every lane owns one byte of the batch [bstart, bstart + used) and knows where it
    // comes from -- a literal, or an earlier position of the text --, so a batch of ~8 symbols costs ONE load and ONE
    // (contiguous) store, and the serial chain pays a round trip to L2 per batch instead of per symbol.
    uint32_t bstart = 0, used = 0;   // wave-uniform
    uint32_t from = 0;               // this lane's byte: kLit | literal, or its source offset in the text
    auto flush = [&]() {
        if (used) {
            // A lane whose byte comes from INSIDE the batch (a match that starts fewer bytes back than the batch is long: every
            // third match of a gzip -1 member of DNA, round 5) takes over its source lane's source, all lanes at once, until
            // none points into the batch any more: pointer jumping, at most six rounds of one ds_bpermute.  (Round 4 kept such
            // matches out of the lanes' walk and decoded them one at a time.)
            if (MODE != 1) {
                for (;;) {
                    const bool inside = lane < used && !(from & kLit) && (int)from >= (int)bstart;
                    if (!__any(inside)) break;
                    const uint32_t theirs = bperm(inside ? from - bstart : lane, from);
                    if (inside) from = theirs;
                }
            }
            if (MODE == 0) {
                uint32_t v = from & 0xFFu;
                // (what the batch copies was stored by earlier batches of this wave: they have landed before it is read --
                // a batch ago they were issued, the wait is free -- and the bytes are read past the vector L1)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane < used && !(from >> 31)) v = gload1_fresh(out + from);
                if (lane < used) out[bstart + lane] = (uint8_t)v;
            } else if (WRITES) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane < used) {
                    const int f = (int)from;
                    uint32_t v;
                    if (f < 0) v = 0x8000u | (uint32_t)(f + 32768);   // (>= -32768: checked where the copy was met)
                    else if (from & kLit) v = from & 0xFFu;
                    else v = __hip_atomic_load((const DD_GLOBAL uint16_t*)(sym + f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *(DD_GLOBAL uint16_t*)(sym + bstart + lane) = (uint16_t)v;
                }
            }
            bstart += used;
            used = 0;
        }
    };
    // gzip member header: 10 fixed bytes, FEXTRA (BGZF's 'BC' field lives there), then -- not in BGZF, but legal -- name, comment, CRC16
    uint32_t hdr = 0;
    if (!RAW) {
        if (n < 28u || uni(gload1(in)) != 0x1fu || uni(gload1(in + 1)) != 0x8bu || uni(gload1(in + 2)) != 8u) ok = false;
        if (ok) {
            const uint32_t flg = uni(gload1(in + 3));
            hdr = 10;
            if (flg & 4u) hdr += 2u + (uni(gload1(in + 10)) | (uni(gload1(in + 11)) << 8));
            if (flg & (8u | 16u | 2u | 0xE0u)) ok = false;   // (name / comment / header CRC: bgzip writes none; the host decoder takes such files)
            if (hdr + 8u > n) ok = false;
        }
    }
    WBits b;
    bool final_seen = false;
    if (ok) {
        if (!RAW) b.init(in + hdr, n - hdr);
        else {
            b.w = reinterpret_cast<const uint32_t*>(in);   // (the file's bytes start on a 256-byte boundary)
            b.nwords = (n + 3u) / 4u;
            b.start_at((uint32_t)(pr.start >> 5), (uint32_t)pr.start & 31u);
        }
        for (;;) {
            const uint32_t bfinal = b.take(1), btype = b.take(2);
            if (btype == 3u) { ok = false; break; }
            if (btype == 0u) {
                b.drop(b.cnt & 7);
                const uint32_t len = b.take(16), nlen = b.take(16);
                if ((len ^ nlen) != 0xffffu) { ok = false; break; }
                if (at + len > out_len) { ok = false, too_long = true; break; }
                // stored bytes: straight from the input (the reader stands on a byte boundary; its word base stays)
                const uint8_t* const base = reinterpret_cast<const uint8_t*>(b.w);
                const uint32_t data = (uint32_t)(b.bit_pos() >> 3);   // byte offset of the data from the reader's base
                if ((uint32_t)(base - in) + data + len + 8u > n) { ok = false; break; }
                flush();
                if (MODE == 0)
                    for (uint32_t i = lane; i < len; i += 64u) out[at + i] = (uint8_t)gload1(base + data + i);
                if (WRITES)
                    for (uint32_t i = lane; i < len; i += 64u) *(DD_GLOBAL uint16_t*)(sym + at + i) = (uint16_t)gload1(base + data + i);
                bstart = at + len;
                at += len;
                b.start_at((data + len) >> 2, 8u * ((data + len) & 3u));
            } else {
                if (btype == 1u) {
                    for (uint32_t i = lane; i < 320u; i += 64u) g_lds[kLens + i] = (uint8_t)(i < 144u ? 8 : i < 256u ? 9 : i < 280u ? 7 : i < 288u ? 8 : 5);
                    __builtin_amdgcn_wave_barrier();
                    // (the fixed distance code has 32 codes of 5 bits -- 30 and 31 may not occur, their table entries stay
                    // empty; with 30 the code is incomplete and every fixed block was refused: DD_INFLATE_STRICT found it)
                    if (!uni(build_table(0, 288, 0, kLitInfo, kLitCount, kLitSymbol)) || !uni(build_table(288, 32, 1, kDistInfo, kDistCount, kDistSymbol))) { ok = false; break; }
                } else {
                    if (!dynamic_tables(b)) { ok = false; break; }
                }
                // THE BLOCK'S SYMBOLS, a window of 64 bit positions at a time.
                // A wave issues one instruction in ~4-8 cycles and the waves of a CU share ONE scalar issue slot: a launch
                // of thousands of blocks costs what its scalar instructions cost (the lockstep form -- one table lookup,
                // one set of field extractions, one copy per symbol, all scalar -- ran ~100 scalar instructions per symbol:
                // 5.7 ms per block alone, 13.7 ms for 3 880 blocks; profiles/r04_bgzf.txt).  So the lanes decode: lane i
                // decodes the WHOLE symbol that would start at bit i of the window -- literal/length code, extra bits,
                // distance code, extra bits: two table gathers --; a scalar walk from the symbol that really starts the
                // window follows the chain (one v_readlane and a dozen scalar instructions per symbol) and gives every
                // symbol on it its place in the output batch; then the lanes of the batch look their symbol up and note
                // where their byte comes from.  Symbols the lanes cannot finish alone leave the walk to the one-symbol
                // path below: codes longer than the tables' 10 bits, the end of the block, a match whose source lies before
                // the text's start, one that does not fit a batch.  (Round 4 also sent every match there whose source MIGHT
                // be inside the batch -- distance < length + 64 --: 2 % of the matches of a gzip -6 member of DNA, but 31 % at
                // gzip -1, whose matcher takes the most recent occurrence.  Now flush() resolves sources inside the batch by
                // pointer jumping and they stay in the walk: inflate_kernel<3> over gzip -1 members 12-14 -> ~9 ms per batch,
                // ten 50 Mbp files 8.6 -> 10.9 Gbp/s on one box, 8.6 -> 9.3 on another.)
                uint32_t q, r, s0, s1, s2, s3, s4;   // the window: bit r of word W[q] = s0; s0..s4 = W[q .. q + 4]
                {
                    const uint64_t P = b.bit_pos();
                    q = (uint32_t)(P >> 5);
                    r = (uint32_t)P & 31u;
                    b.seek(q);
                    s0 = b.word(), s1 = b.word(), s2 = b.word(), s3 = b.word(), s4 = b.word();
                }
                uint32_t osv = 0;   // a symbol's lane: the batch slot of its first byte
                for (;;) {
                    // lane i: the 64 bits from bit r + i on
                    const uint32_t t = lane + r, kq = t >> 5, sh = t & 31u;
                    const uint32_t wa = kq == 0u ? s0 : (kq == 1u ? s1 : s2), wb = kq == 0u ? s1 : (kq == 1u ? s2 : s3), wc = kq == 0u ? s2 : (kq == 1u ? s3 : s4);
                    const uint32_t x_lo = __builtin_amdgcn_alignbit(wb, wa, sh), x_hi = __builtin_amdgcn_alignbit(wc, wb, sh);
                    const uint32_t e1 = l32(kLitInfo + 4u * (x_lo & ((1u << FAST) - 1u)));
                    const uint32_t l1 = e1 & 15u, ex1 = (e1 >> 7) & 15u, kd = (e1 >> 4) & 7u;
                    const uint32_t v1 = (e1 >> 11) + ((x_lo >> l1) & ((1u << ex1) - 1u));   // the literal, or the match's length
                    const uint32_t t1 = l1 + ex1;
                    const uint32_t y = (uint32_t)((((uint64_t)x_hi << 32) | x_lo) >> t1);
                    const uint32_t e2 = l32(kDistInfo + 4u * (y & ((1u << FAST) - 1u)));
                    const uint32_t l2 = e2 & 15u, ex2 = (e2 >> 7) & 15u;
                    const uint32_t dv = (e2 >> 11) + ((y >> l2) & ((1u << ex2) - 1u));
                    // bits 0..5: the symbol's length in bits; 6..14: bytes it makes; 0 = not for the walk
                    uint32_t pk = 0;
                    if (kd == 1u) pk = l1 | (1u << 6);
                    else if (kd == 3u && e2 != 0u && dv <= at + (RAW ? 32768u : 0u)) pk = (t1 + l2 + ex2) | (v1 << 6);
                    unsigned long long mark = 0, starts = 0;
                    uint32_t pos = 0, outacc = 0, pks = 0;
                    parallel_walk(pk, lane, used, mark, pos, outacc, pks, osv);
                    // which slots of the batch start a symbol: the symbols' lanes say so in LDS, the slots' lanes read it back
                    g_lds[kLens + 512u + lane] = 0;
                    __builtin_amdgcn_wave_barrier();
                    if ((mark >> lane) & 1ull) g_lds[kLens + 512u + osv] = 1;
                    __builtin_amdgcn_wave_barrier();
                    starts = __ballot(g_lds[kLens + 512u + lane] != 0);
                    if (outacc) {
                        if (at + outacc > out_len) { ok = false, too_long = true; break; }
                        // the symbols' lanes say where their bytes come from; the batch's lanes find their symbol by counting
                        // (the counting pass of the raw mode needs none of it)
                        if (MODE != 1 && ((mark >> lane) & 1ull)) {
                            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mark >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mark, 0u));
                            const uint32_t src = kd == 1u ? (kLit | v1) : bstart + osv - dv;   // (RAW: may wrap below zero = in front of the piece)
                            *reinterpret_cast<uint2*>(g_lds + kLens + 8u * rank) = make_uint2(src, osv);
                        }
                        __builtin_amdgcn_wave_barrier();
                        if (MODE != 1 && lane - used < outacc) {
                            const uint32_t ord = __builtin_amdgcn_mbcnt_hi((uint32_t)(starts >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)starts, 0u)) +
                                                 (uint32_t)((starts >> lane) & 1ull) - 1u;
                            const uint2 sy = *reinterpret_cast<const uint2*>(g_lds + kLens + 8u * ord);
                            from = sy.x + (lane - sy.y);
                        }
                        __builtin_amdgcn_wave_barrier();
                        used += outacc;
                        at += outacc;
                        if (used == 64u) flush();
                    }
                    if (pos < 64u && pks != 1u) {
                        // ONE SYMBOL, step by step, from the 64 bits at `pos` (a symbol takes at most 15 + 5 + 15 + 13)
                        if (pks != 0u) {   // a match for the walk that does not fit what is left of the batch
                            if (used) {
                                flush();
                                goto advance;
                            }
                        }
                        uint64_t bits = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)x_hi, (int)pos) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)x_lo, (int)pos);
                        uint32_t kind, val, ex;
                        const uint32_t li = uni(l32(kLitInfo + 4u * ((uint32_t)bits & ((1u << FAST) - 1u))));
                        if (li) {
                            bits >>= li & 15u, pos += li & 15u;
                            kind = (li >> 4) & 7u;
                            ex = (li >> 7) & 15u;
                            val = li >> 11;
                        } else {   // a code longer than the table's 10 bits
                            const uint32_t rs = uni(decode_slow(bits, kLitCount, kLitSymbol));
                            const uint32_t sy = rs >> 4;
                            if (rs == ~0u || sy > 285u) { ok = false; break; }
                            bits >>= rs & 15u, pos += rs & 15u;
                            kind = sy < 256u ? 1u : (sy == 256u ? 2u : 3u);
                            ex = sy > 256u ? uni((uint32_t)c_len_extra[sy - 257u]) : 0u;
                            val = sy < 256u ? sy : (sy > 256u ? uni((uint32_t)c_len_base[sy - 257u]) : 0u);
                        }
                        if (kind == 1u) {
                            if (at >= out_len) { ok = false, too_long = true; break; }
                            if (lane == used) from = kLit | val;
                            ++at;
                            if (++used == 64u) flush();
                        } else if (kind != 3u) {   // end of block (kind 0: a code the stream may not use)
                            if (kind != 2u) ok = false;
                            flush();
                            r += pos;
                            break;
                        } else {
                            const uint32_t len = val + ((uint32_t)bits & ((1u << ex) - 1u));
                            bits >>= ex, pos += ex;
                            const uint32_t di = uni(l32(kDistInfo + 4u * ((uint32_t)bits & ((1u << FAST) - 1u))));
                            uint32_t dist, dex;
                            if (di) {
                                bits >>= di & 15u, pos += di & 15u;
                                dex = (di >> 7) & 15u;
                                dist = di >> 11;
                            } else {
                                const uint32_t rs = uni(decode_slow(bits, kDistCount, kDistSymbol));
                                const uint32_t ds = rs >> 4;
                                if (rs == ~0u || ds > 29u) { ok = false; break; }
                                bits >>= rs & 15u, pos += rs & 15u;
                                dex = uni((uint32_t)c_dist_extra[ds]);
                                dist = uni((uint32_t)c_dist_base[ds]);
                            }
                            dist += (uint32_t)bits & ((1u << dex) - 1u);
                            pos += dex;
                            if (dist > at + (RAW ? 32768u : 0u)) { ok = false; break; }
                            if (at + len > out_len) { ok = false, too_long = true; break; }
                            // the copy: its bytes join the batch (several batches when it is long).  It reads the dist-byte pattern
                            // in front of it; should that reach into the batch itself, the batch leaves first.
                            const uint32_t pat = at - dist;   // (RAW: may be "negative")
                            at += len;
                            auto place = [&](auto src_of) {
                                uint32_t done = 0;
                                do {
                                    const uint32_t space = 64u - used, left = len - done;
                                    const uint32_t take = left < space ? left : space;
                                    const uint32_t o = done + lane - used;   // this lane's offset inside the match (if it is one of the `take`)
                                    if (MODE != 1 && lane - used < take) from = pat + src_of(o);
                                    used += take;
                                    done += take;
                                    if (used == 64u) flush();
                                } while (done < len);
                            };
                            if (dist >= len) place([](uint32_t o) { return o; });
                            else if (dist == 1u) place([](uint32_t) { return 0u; });   // (a run of N, of one base)
                            else place([&](uint32_t o) { return o % dist; });          // the pattern repeats inside the match
                        }
                    }
                advance:
                    r += pos;
                    while (r >= 32u) {
                        s0 = s1, s1 = s2, s2 = s3, s3 = s4;
                        s4 = b.word();
                        ++q;
                        r -= 32u;
                    }
                }
                if (ok) {   // the reader takes over again where the symbols ended (r may have run past s0)
                    q += r >> 5;
                    b.start_at(q, r & 31u);
                }
                if (!ok) break;
            }
            if (bfinal) {
                final_seen = true;
                break;
            }
            if (RAW) {   // the piece ends where the next one starts -- exactly there, or the starts are not block starts
                const uint64_t P = b.bit_pos();
                if (P == piece_end) break;
                if (P > piece_end) { ok = false; break; }
            }
        }
    }
    if (RAW) {
        if (ok && final_seen != (piece_end == ~0ull)) ok = false;
        if (ok && final_seen && b.bytes_used(in) + 8u != n) ok = false;   // ONE member: CRC-32 and ISIZE right behind the final block
        if (MODE == 3 && !ok && too_long && out_len < files[pr.file].isize) {   // not an error: the piece is counted, then written to the arena
            if (lane == 0) lens[blockIdx.x] = 0, over[blockIdx.x] = 1;
            return;
        }
        if (MODE == 3 && lane == 0) over[blockIdx.x] = 0;
        if ((MODE == 1 || MODE == 3) && lane == 0) lens[blockIdx.x] = ok ? at : 0u;
        if (MODE == 2 && ok && at != out_len) ok = false;
        // (a piece that alone is longer than the member's ISIZE: the trailer's matter -- kSizeMismatch --, not the decoder's)
        if (!ok && lane == 0) {   // (class bits are OR-ed, counts added: 65 536 pieces x 2^16 would add up to zero)
            if (too_long && out_len >= files[pr.file].isize) atomicOr(errors, kSizeMismatch);
            else atomicAdd(errors, 1u);
        }
        return;
    }
    // the member's trailer: CRC-32, ISIZE
    if (ok) {
        const uint32_t used = b.bytes_used(in + hdr);
        if (hdr + used + 8u > n) ok = false;
        else {
            const uint8_t* t = in + hdr + used;
            auto le32 = [&](const uint8_t* q) { return uni(gload1(q)) | (uni(gload1(q + 1)) << 8) | (uni(gload1(q + 2)) << 16) | (uni(gload1(q + 3)) << 24); };
            const uint32_t crc = le32(t), isize = le32(t + 4);
            if (isize != at || at != out_len) ok = false;
            else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the text's last batch has landed
                if (text_crc(out, at) != crc) ok = false;
            }
        }
    }
    if (!ok) {
        if (lane == 0) atomicAdd(errors, 1u);
        return;
    }
}

// ---- single-member gzip files: where do deflate blocks start? ---------------------------------------------------
// A wave per guess: file f's guess j covers the bit positions [first_bit + j G, first_bit + (j + 1) G) and reports the
// FIRST position in it that heads a valid dynamic-Huffman block (guess 0 reports first_bit itself).  Lane i tests the
// position base + i: block type 2, HLIT <= 29, HDIST <= 29, the code-length code complete (or a single code) -- 22 % pass
// the first, ~1 % of those the second --; survivors queue up in LDS and are put to the full test 64 at a time, a
// candidate per lane: its code lengths decoded with a bit reader of the lane's own, the literal/length code complete
// with an end-of-block code, the distance code complete or a single code.  What passes that is a block start or a
// one-in-10^9 impostor; an impostor makes a piece end somewhere else than the next one starts and the call goes to the
// host decoder.  Stored and fixed blocks are not looked for (they are decoded as parts of pieces).
constexpr uint32_t kFindTable = kLitInfo;           // u8[64][128]: every lane's code-length code (7-bit lookup): 8 KiB from the symbol tables' place on
constexpr uint32_t kFindQueue = kInflateLds > 8192u ? kInflateLds : 8192u;   // u32[128]: candidate bit positions waiting for the full test (behind the lanes' tables
                                                                              // and behind everything a header parse writes)
constexpr uint32_t kFindLds = kFindQueue + 512u;
constexpr uint8_t k_cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};   // (c_cl_order, for unrolled loops)

__global__ __launch_bounds__(64) void find_starts_kernel(const RawFile* __restrict__ files, int nfiles, uint64_t* __restrict__ starts) {
    const uint32_t lane = threadIdx.x & 63u;
    int f = 0;
    while (f + 1 < nfiles && blockIdx.x >= uni(files[f + 1].piece0)) ++f;
    const RawFile rf = files[f];
    const uint32_t j = blockIdx.x - rf.piece0;
    if (j >= rf.nguess) return;
    if (j == 0) {
        if (lane == 0) starts[blockIdx.x] = rf.first_bit;
        return;
    }
    const uint64_t total_bits = ((uint64_t)rf.in_len - 8u) * 8u;   // (the trailer is no place for a block)
    const uint64_t lo = (uint64_t)rf.first_bit + (uint64_t)j * rf.guess_bits;
    uint64_t found = ~0ull;
    if (lo + 64u < total_bits) {
        const uint64_t hi = lo + rf.guess_bits < total_bits ? lo + rf.guess_bits : total_bits;
        const uint32_t* const W = reinterpret_cast<const uint32_t*>(rf.in);
        const uint32_t nwords = (rf.in_len + 3u) / 4u;
        auto word_at = [&](uint32_t i) { return i < nwords ? gload4(W + i) : 0u; };
        // the full test of up to 64 queued candidates, one per lane; -> the smallest that passes, or ~0u
        auto full_test = [&](uint32_t nq) -> uint64_t {   // (the queue holds positions as offsets from `lo`)
            const uint64_t cand = lo + (lane < nq ? l32(kFindQueue + 4u * lane) : 0u);
            bool live = lane < nq;
            // the lane's bit reader
            uint32_t wi = (uint32_t)(cand >> 5);
            uint64_t buf = ((uint64_t)word_at(wi + 1u) << 32 | word_at(wi)) >> ((uint32_t)cand & 31u);
            int cnt = 64 - (int)((uint32_t)cand & 31u);
            wi += 2u;
            auto need = [&](int k) {
                if (cnt < k) {
                    buf |= (uint64_t)word_at(wi) << cnt;
                    cnt += 32;
                    ++wi;
                }
            };
            auto take = [&](int k) {
                need(k);
                const uint32_t v = (uint32_t)buf & ((1u << k) - 1u);
                buf >>= k, cnt -= k;
                return v;
            };
            (void)take(3);
            const uint32_t hlit = take(5) + 257u, hdist = take(5) + 1u, hclen = take(4) + 4u;
            // its code-length code: lengths, canonical codes, a 128-entry table of its own in LDS
            uint32_t cl[19];
#pragma unroll
            for (int i = 0; i < 19; ++i) cl[i] = 0;
            uint32_t count[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 19; ++i) {
                const uint32_t v = (uint32_t)i < hclen ? take(3) : 0u;
#pragma unroll
                for (int s2 = 0; s2 < 19; ++s2)
                    if (k_cl_order[i] == s2) cl[s2] = v;
            }
#pragma unroll
            for (int s2 = 0; s2 < 19; ++s2)
#pragma unroll
                for (int l = 1; l < 8; ++l) count[l] += cl[s2] == (uint32_t)l ? 1u : 0u;
            uint32_t next[8], c = 0;
#pragma unroll
            for (int l = 1; l < 8; ++l) {
                next[l] = c;
                c = (c + count[l]) << 1;
            }
            uint8_t* const tab = g_lds + kFindTable + 128u * lane;
            for (int i = 0; i < 128; i += 4) *reinterpret_cast<uint32_t*>(tab + i) = 0;
#pragma unroll
            for (int s2 = 0; s2 < 19; ++s2) {
                const uint32_t l = cl[s2];
                if (live && l) {
                    uint32_t code = 0;
#pragma unroll
                    for (int q = 1; q < 8; ++q)
                        if (l == (uint32_t)q) code = next[q]++;
                    const uint32_t rev = __builtin_bitreverse32(code) >> (32u - l);
                    for (uint32_t e = rev; e < 128u; e += 1u << l) tab[e] = (uint8_t)(l | ((uint32_t)s2 << 3));
                }
            }
            // the literal/length and distance code lengths, run-length coded: Kraft sums in units of 2^-15
            const uint32_t totalsym = hlit + hdist;
            uint32_t i = 0, prev = 0, kraft_ll = 0, kraft_d = 0, nz_d = 0, eob = 0;
            while (__any(live && i < totalsym)) {
                if (live && i < totalsym) {
                    need(14);
                    const uint32_t e = tab[(uint32_t)buf & 127u];
                    if (!e) live = false;
                    else {
                        buf >>= e & 7u, cnt -= (int)(e & 7u);
                        const uint32_t sy = e >> 3;
                        uint32_t rep = 1, val = sy;
                        if (sy == 16u) {
                            if (!i) live = false;
                            val = prev;
                            rep = 3u + ((uint32_t)buf & 3u);
                            buf >>= 2, cnt -= 2;
                        } else if (sy == 17u) {
                            val = 0;
                            rep = 3u + ((uint32_t)buf & 7u);
                            buf >>= 3, cnt -= 3;
                        } else if (sy == 18u) {
                            val = 0;
                            rep = 11u + ((uint32_t)buf & 127u);
                            buf >>= 7, cnt -= 7;
                        }
                        if (i + rep > totalsym) live = false;
                        if (live && val) {
                            const uint32_t n_ll = i < hlit ? (hlit - i < rep ? hlit - i : rep) : 0u, n_d = rep - n_ll;
                            kraft_ll += n_ll << (15u - val);
                            kraft_d += n_d << (15u - val);
                            nz_d += n_d;
                            if (i <= 256u && 256u < i + rep) eob = val;
                        }
                        i += rep;
                        prev = val;
                    }
                }
            }
            const bool pass = live && eob != 0u && kraft_ll == (1u << 15) && (kraft_d == (1u << 15) || nz_d <= 1u) && hlit <= 286u && hdist <= 30u;
            // What passes is a block start or, once in ~10^9 positions, an impostor (8 x 50 Mbp of gzip -6 held one).  The
            // last word has a trial decoding, wave-uniform, with the decoder's own tables: the header parsed again, then
            // the first symbols -- every code valid, every literal a byte of text (9 .. 126: FASTA has no others).
            unsigned long long m = __ballot(pass);
            while (m) {
                const int first = __builtin_ctzll(m);
                const uint64_t c0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(cand >> 32), first) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)cand, first);
                m &= m - 1ull;
                WBits v;
                v.w = W, v.nwords = nwords;
                v.start_at((uint32_t)(c0 >> 5), (uint32_t)c0 & 31u);
                (void)v.take(3);
                bool good = dynamic_tables(v);
                for (int k = 0; good && k < 24; ++k) {
                    v.need();
                    const uint32_t li = uni(l32(kLitInfo + 4u * v.peek(FAST)));
                    uint32_t kind, ex;
                    if (li) {
                        v.drop((int)(li & 15u));
                        kind = (li >> 4) & 7u, ex = (li >> 7) & 15u;
                        if (kind == 1u && ((li >> 11) < 9u || (li >> 11) > 126u)) good = false;
                    } else {
                        const uint32_t r = uni(decode_slow(v.buf, kLitCount, kLitSymbol));
                        const uint32_t sy = r >> 4;
                        if (r == ~0u || sy > 285u) { good = false; break; }
                        v.drop((int)(r & 15u));
                        kind = sy < 256u ? 1u : (sy == 256u ? 2u : 3u);
                        ex = sy > 256u ? uni((uint32_t)c_len_extra[sy - 257u]) : 0u;
                        if (kind == 1u && (sy < 9u || sy > 126u)) good = false;
                    }
                    if (kind == 2u) break;
                    if (kind == 3u) {
                        v.drop((int)ex);
                        v.need();
                        const uint32_t di = uni(l32(kDistInfo + 4u * v.peek(FAST)));
                        if (di) v.drop((int)((di & 15u) + ((di >> 7) & 15u)));
                        else {
                            const uint32_t r = uni(decode_slow(v.buf, kDistCount, kDistSymbol));
                            if (r == ~0u || (r >> 4) > 29u) { good = false; break; }
                            v.drop((int)(r & 15u));
                            v.need();
                            v.drop((int)uni((uint32_t)c_dist_extra[r >> 4]));
                        }
                    } else if (kind != 1u) good = false;
                }
                if (good) return c0;
            }
            return ~0ull;
        };
        // Kraft sum and count of the code-length code's non-zero lengths, three 3-bit lengths at a time: a 512-entry table at the
        // front of LDS (round 5; the unrolled 19-length sum was ~95 of the scan's ~110 VALU instructions per 64 positions, and the
        // scan is what this kernel's 5 ms per batch were made of).  full_test() overwrites it with its own tables: rebuilt after.
        auto build_lut = [&]() {
            for (uint32_t v = lane; v < 512u; v += 64u) {
                uint32_t kr = 0, nzv = 0;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const uint32_t l = (v >> (3 * q)) & 7u;
                    kr += l ? (128u >> l) : 0u;
                    nzv += l ? 1u : 0u;
                }
                l32(kLitInfo + 4u * v) = kr | (nzv << 16);
            }
            __builtin_amdgcn_wave_barrier();
        };
        build_lut();
        uint32_t nq = 0;
        // The stream comes in through ONE load per 29 steps: lane l holds word cbase + l of a 64-word chunk, a step takes the six
        // words its 64 positions span out of it with v_readlane (the step's first word is wave-uniform) and every lane picks its
        // three by the word its position starts in.  (Four loads per lane and step, each behind a bounds check and a 64-bit
        // address, were what the scan waited for: round 5, profiles/r05_gunzip.txt.)
        uint32_t cbase = (uint32_t)(lo >> 5), cw = word_at(cbase + lane);
        const uint32_t r0 = (uint32_t)lo & 31u;   // (base = lo + 64 n: its bit inside its word never changes)
        const uint32_t tt = lane + r0, kq = tt >> 5, sh = tt & 31u;
        for (uint64_t base = lo; base < hi && found == ~0ull; base += 64u) {
            // the 96 bits from position base + lane on
            const uint64_t pos = base + lane;
            uint32_t qrel = (uint32_t)(base >> 5) - cbase;
            if (qrel + 5u > 63u) {
                cbase += qrel;
                cw = word_at(cbase + lane);
                qrel = 0;
            }
            const uint32_t s0 = lane_value(cw, qrel), s1 = lane_value(cw, qrel + 1u), s2 = lane_value(cw, qrel + 2u), s3 = lane_value(cw, qrel + 3u),
                           s4 = lane_value(cw, qrel + 4u), s5 = lane_value(cw, qrel + 5u);
            const uint32_t w0 = kq == 0u ? s0 : (kq == 1u ? s1 : s2), w1 = kq == 0u ? s1 : (kq == 1u ? s2 : s3), w2 = kq == 0u ? s2 : (kq == 1u ? s3 : s4),
                           w3 = kq == 0u ? s3 : (kq == 1u ? s4 : s5);
            const uint32_t x0 = __builtin_amdgcn_alignbit(w1, w0, sh), x1 = __builtin_amdgcn_alignbit(w2, w1, sh), x2 = __builtin_amdgcn_alignbit(w3, w2, sh);
            const uint32_t hclen = ((x0 >> 13) & 15u) + 4u;
            bool cand = pos < hi && ((x0 >> 1) & 3u) == 2u && ((x0 >> 3) & 31u) <= 29u && ((x0 >> 8) & 31u) <= 29u;
            // Kraft sum of the code-length code (3-bit lengths from bit 17 on, the first hclen of them) in units of 2^-7
            const uint32_t f_lo = __builtin_amdgcn_alignbit(x1, x0, 17), f_hi = __builtin_amdgcn_alignbit(x2, x1, 17);
            const uint32_t nbits = 3u * hclen;   // 12 .. 57
            const uint32_t fa = f_lo & (nbits >= 32u ? ~0u : (1u << (nbits & 31u)) - 1u), fb = f_hi & (nbits > 32u ? (1u << ((nbits - 32u) & 31u)) - 1u : 0u);
            auto lut = [&](uint32_t nine) { return l32(kLitInfo + 4u * (nine & 511u)); };
            const uint32_t sum = lut(fa) + lut(fa >> 9) + lut(fa >> 18) + lut((fa >> 27) | (fb << 5)) + lut(fb >> 4) + lut(fb >> 13) + lut(fb >> 22);
            const uint32_t kraft = sum & 0xFFFFu, nz = sum >> 16;
            cand = cand && (kraft == 128u || nz == 1u);
            const unsigned long long m = __ballot(cand);
            if (m) {
                if (cand) l32(kFindQueue + 4u * (nq + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)))) = (uint32_t)(pos - lo);
                nq += (uint32_t)__builtin_popcountll(m);
                __builtin_amdgcn_wave_barrier();
                if (nq >= 64u) {
                    found = full_test(64u);
                    __builtin_amdgcn_wave_barrier();
                    build_lut();
                    // the rest of the queue moves to the front
                    const uint32_t restv = lane < nq - 64u ? l32(kFindQueue + 4u * (64u + lane)) : 0u;
                    __builtin_amdgcn_wave_barrier();
                    if (lane < nq - 64u) l32(kFindQueue + 4u * lane) = restv;
                    nq -= 64u;
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        if (found == ~0ull && nq) found = full_test(nq);
    }
    if (lane == 0) starts[blockIdx.x] = found;
}

// text offsets of the pieces (one wave per file; <= a few thousand pieces): offs[i] = sum of the lens before i; the sum
// must be the member's ISIZE
// ... and abase[i] = the same sum over the pieces that go to the arena
__global__ __launch_bounds__(64) void piece_offsets_kernel(const RawFile* __restrict__ files, const uint32_t* __restrict__ lens, const uint32_t* __restrict__ over,
                                                           uint32_t* __restrict__ offs, uint32_t* __restrict__ abase, uint32_t* __restrict__ errors) {
    const RawFile rf = files[blockIdx.x];
    const uint32_t lane = threadIdx.x & 63u;
    // (64-bit sums: a piece's length is bounded by ISIZE, their SUM is not -- a trailer whose ISIZE is smaller than the text
    // (damage, two members, a text beyond 4 GiB whose ISIZE is the length mod 2^32) must not wrap back into "equal")
    uint64_t run = 0, arun = 0;
    for (uint32_t b0 = 0; b0 < rf.nguess; b0 += 64u) {
        const uint32_t i = b0 + lane, mine = i < rf.nguess ? lens[rf.piece0 + i] : 0u, amine = (i < rf.nguess && over[rf.piece0 + i]) ? mine : 0u;
        uint64_t incl = mine, aincl = amine;
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t up = (uint64_t)__shfl_up((long long)incl, d), aup = (uint64_t)__shfl_up((long long)aincl, d);
            if ((int)lane >= d) incl += up, aincl += aup;
        }
        // (offsets beyond ISIZE are clamped: nothing reads them once the error below is up, and nothing may index with a wrapped one)
        const uint64_t o = run + incl - mine, a = arun + aincl - amine;
        if (i < rf.nguess) offs[rf.piece0 + i] = (uint32_t)(o < rf.isize ? o : rf.isize), abase[rf.piece0 + i] = (uint32_t)(a < rf.isize ? a : rf.isize);
        run += (uint64_t)__shfl((long long)incl, 63);
        arun += (uint64_t)__shfl((long long)aincl, 63);
    }
    // kSizeMismatch: the pieces decoded, but not to the text the trailer announces (the host tells this from a refused block)
    if ((run != (uint64_t)rf.isize || arun > (uint64_t)rf.isize) && lane == 0) atomicOr(errors, kSizeMismatch);
}

// where a piece's symbols are: its ranges of the symbol area, or the arena
DD_D const uint16_t* piece_symbols(const RawFile& rf, uint32_t i, const uint32_t* over, const uint32_t* abase) {
    return over[rf.piece0 + i] ? rf.arena + abase[rf.piece0 + i] : rf.sym + (size_t)i * rf.range_syms;
}

// What stands in the 32 KiB in front of every piece?  Piece i turns the window in front of it into the window behind it:
// every position of the new window is a byte of the piece or a position of the old window -- a MAP of 32 768 16-bit
// entries, and maps compose.  Walking a file's pieces one after the other with the window in LDS cost ~7 us a piece on ONE
// CU (52 of the 241 ms of a 3 Gbp assembly's 7 500 pieces) while the chip waited.  Two levels instead:
//   piece_maps_kernel     a workgroup per GROUP of 32 ranges, all groups of all files side by side: starting from the
//                         identity, compose the group's pieces; the map in front of each piece (relative to the group's
//                         start) is stored, and the group's whole map at the end
//   group_windows_kernel  a workgroup per file walks its GROUPS (a 32nd of the steps): the window at each group's start
//   translate_kernel      a placeholder goes through its piece's map and, if that still points in front of the group, through
//                         the group's window
// A thread owns the window positions t, t + 1024, ..: a wave's symbol loads are 128 contiguous bytes, and a step's symbols
// are asked for a step ahead (the chain waits for LDS and a barrier per piece, not for HBM).
DD_D uint16_t* piece_map(const RawFile& rf, uint32_t i) { return reinterpret_cast<uint16_t*>(rf.windows) + (size_t)i * 32768u; }
DD_D uint16_t* group_map(const RawFile& rf, uint32_t g) { return reinterpret_cast<uint16_t*>(rf.windows) + ((size_t)rf.nguess + g) * 32768u; }
DD_D uint8_t* group_window(const RawFile& rf, uint32_t g) { return rf.windows + ((size_t)rf.nguess + rf.ngroups) * 65536u + (size_t)g * 32768u; }

__global__ __launch_bounds__(1024) void piece_maps_kernel(const RawFile* __restrict__ files, int nfiles, const uint32_t* __restrict__ lens,
                                                          const uint32_t* __restrict__ over, const uint32_t* __restrict__ abase, const uint32_t* __restrict__ errors) {
    extern __shared__ __attribute__((aligned(16))) uint8_t win[];   // u16 [2][32768]
    if (*errors) return;   // (a refused batch: lengths and offsets may not fit each other; the call goes to the host anyway)
    int f = 0;
    while (f + 1 < nfiles && blockIdx.x >= files[f + 1].group0) ++f;
    const RawFile rf = files[f];
    const uint32_t g = blockIdx.x - rf.group0;
    if (g >= rf.ngroups) return;
    const uint32_t first = g * kPieceGroup, last = first + kPieceGroup < rf.nguess ? first + kPieceGroup : rf.nguess;
    uint16_t* maps = reinterpret_cast<uint16_t*>(win);
    const uint32_t t0 = threadIdx.x;
#pragma unroll
    for (int q = 0; q < 32; ++q) maps[t0 + 1024u * q] = (uint16_t)(0x8000u | (t0 + 1024u * q));   // the identity
    __syncthreads();
    auto next_piece = [&](uint32_t i) {   // first piece with text at or behind i
        while (i < last && lens[rf.piece0 + i] == 0u) ++i;
        return i;
    };
    uint16_t cur_s[32], nxt_s[32];
    auto fetch = [&](uint32_t i, uint16_t (&dst)[32]) {
        if (i >= last) return;
        const uint32_t L = lens[rf.piece0 + i];
        const uint16_t* const s = piece_symbols(rf, i, over, abase);
        const int p0 = (int)L - 32768 + (int)t0;
#pragma unroll
        for (int q = 0; q < 32; ++q) dst[q] = (p0 + 1024 * q >= 0) ? s[p0 + 1024 * q] : (uint16_t)0;
    };
    uint32_t i = next_piece(first), cur = 0;
    fetch(i, cur_s);
    while (i < last) {
        const uint32_t L = lens[rf.piece0 + i], inext = next_piece(i + 1u);
        fetch(inext, nxt_s);
        uint16_t* const before = piece_map(rf, i);
        const uint16_t* const w = maps + cur * 32768u;
        uint16_t* const nw = maps + (cur ^ 1u) * 32768u;
#pragma unroll
        for (int q = 0; q < 16; ++q) reinterpret_cast<uint32_t*>(before)[t0 + 1024u * q] = reinterpret_cast<const uint32_t*>(w)[t0 + 1024u * q];
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const uint32_t pos = t0 + 1024u * q;
            const int p = (int)L - 32768 + (int)pos;
            const uint32_t sy = cur_s[q];
            nw[pos] = (uint16_t)(p >= 0 ? ((sy & 0x8000u) ? (uint32_t)w[sy & 0x7fffu] : sy) : (uint32_t)w[pos + L]);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 32; ++q) cur_s[q] = nxt_s[q];
        cur ^= 1u;
        i = inext;
    }
    uint16_t* const gm = group_map(rf, g);
    const uint16_t* const w = maps + cur * 32768u;
#pragma unroll
    for (int q = 0; q < 16; ++q) reinterpret_cast<uint32_t*>(gm)[t0 + 1024u * q] = reinterpret_cast<const uint32_t*>(w)[t0 + 1024u * q];
}

__global__ __launch_bounds__(1024) void group_windows_kernel(const RawFile* __restrict__ files, const uint32_t* __restrict__ errors) {
    extern __shared__ __attribute__((aligned(16))) uint8_t win[];   // u8 [2][32768]
    if (*errors) return;
    const RawFile rf = files[blockIdx.x];
    const uint32_t t0 = threadIdx.x;
    for (uint32_t t = t0; t < 32768u / 4u; t += 1024u) reinterpret_cast<uint32_t*>(win)[t] = 0;   // nothing stands in front of the stream
    __syncthreads();
    uint16_t cur_m[32], nxt_m[32];
    auto fetch = [&](uint32_t g, uint16_t (&dst)[32]) {
        if (g >= rf.ngroups) return;
        const uint16_t* const m = group_map(rf, g);
#pragma unroll
        for (int q = 0; q < 32; ++q) dst[q] = m[t0 + 1024u * q];
    };
    uint32_t cur = 0;
    fetch(0, cur_m);
    for (uint32_t g = 0; g < rf.ngroups; ++g) {
        fetch(g + 1u, nxt_m);
        uint8_t* const at_start = group_window(rf, g);
        const uint8_t* const w = win + cur * 32768u;
        uint8_t* const nw = win + (cur ^ 1u) * 32768u;
#pragma unroll
        for (int q = 0; q < 8; ++q) reinterpret_cast<uint32_t*>(at_start)[t0 + 1024u * q] = reinterpret_cast<const uint32_t*>(w)[t0 + 1024u * q];
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const uint32_t v = cur_m[q];
            nw[t0 + 1024u * q] = (uint8_t)((v & 0x8000u) ? (uint32_t)w[v & 0x7fffu] : v);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 32; ++q) cur_m[q] = nxt_m[q];
        cur ^= 1u;
    }
}

// symbols -> text: one workgroup per 64 KiB of a file's text; a placeholder is looked up in the map in front of its piece and, if
// that points in front of the piece's group, in the window at the group's start
__global__ __launch_bounds__(256) void translate_kernel(const RawFile* __restrict__ files, int nfiles, const uint32_t* __restrict__ chunk0,
                                                        const uint32_t* __restrict__ lens, const uint32_t* __restrict__ offs, const uint32_t* __restrict__ over,
                                                        const uint32_t* __restrict__ abase, const uint32_t* __restrict__ errors) {
    if (*errors) return;
    int f = 0;
    while (f + 1 < nfiles && blockIdx.x >= chunk0[f + 1]) ++f;
    const RawFile rf = files[f];
    const uint32_t c = blockIdx.x - chunk0[f], begin = c * 65536u, end = begin + 65536u < rf.isize ? begin + 65536u : rf.isize;
    if (begin >= rf.isize) return;
    // the piece that holds `begin`: the last one with offs <= begin and a text of its own (binary search, then a few steps)
    uint32_t lo = 0, hi = rf.nguess;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) / 2u;
        if (offs[rf.piece0 + mid] <= begin) lo = mid;
        else hi = mid;
    }
    // (round 5: the piece's offset, end and symbols stay in registers until a position leaves the piece -- the first form
    // re-read lens / offs / over / abase in front of every symbol, three dependent loads on a chain of five -- and four positions
    // go per step, their loads side by side: 2.42 -> 1.13 ms for 400 MB of text; the kernel waits for memory latency, not bandwidth)
    uint32_t pi = lo, off = offs[rf.piece0 + pi], pend = off + lens[rf.piece0 + pi];
    const uint16_t* syms = piece_symbols(rf, pi, over, abase);
    auto settle = [&](uint32_t p) {   // the piece that holds position p (pieces without a text of their own are stepped over)
        while (pi + 1u < rf.nguess && p >= pend) {
            ++pi;
            off = offs[rf.piece0 + pi];
            pend = off + lens[rf.piece0 + pi];
            syms = piece_symbols(rf, pi, over, abase);
        }
    };
    auto resolve = [&](uint32_t sy, uint32_t piece) {
        if (sy & 0x8000u) {
            sy = piece_map(rf, piece)[sy & 0x7fffu];
            if (sy & 0x8000u) sy = group_window(rf, piece / kPieceGroup)[sy & 0x7fffu];
        }
        return sy;
    };
    uint32_t p = begin + threadIdx.x;
    for (; p + 768u < end; p += 1024u) {
        settle(p);
        if (p + 768u < pend) {   // all four in this piece: four independent loads
            uint32_t sy[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) sy[q] = syms[p + 256u * q - off];
#pragma unroll
            for (int q = 0; q < 4; ++q) rf.text[p + 256u * q] = (uint8_t)resolve(sy[q], pi);
        } else {
#pragma unroll 1
            for (int q = 0; q < 4; ++q) {
                settle(p + 256u * q);
                rf.text[p + 256u * q] = (uint8_t)resolve(syms[p + 256u * q - off], pi);
            }
        }
    }
    for (; p < end; p += 256u) {
        settle(p);
        rf.text[p] = (uint8_t)resolve(syms[p - off], pi);
    }
}

// CRC-32 of every 64 KiB of the texts (one wave each): the host combines them (zlib's crc32_combine) and compares with the trailer's
__global__ __launch_bounds__(64) void chunk_crc_kernel(const RawFile* __restrict__ files, int nfiles, const uint32_t* __restrict__ chunk0, uint32_t* __restrict__ crcs,
                                                       const uint32_t* __restrict__ errors) {
    if (*errors) return;
    int f = 0;
    while (f + 1 < nfiles && blockIdx.x >= uni(chunk0[f + 1])) ++f;
    const RawFile rf = files[f];
    const uint32_t c = blockIdx.x - uni(chunk0[f]), begin = c * 65536u;
    if (begin >= rf.isize) return;
    const uint32_t n = rf.isize - begin < 65536u ? rf.isize - begin : 65536u;
    const uint32_t crc = text_crc(rf.text + begin, n);
    if ((threadIdx.x & 63u) == 0u) crcs[blockIdx.x] = crc;
}

size_t inflate_lds_bytes() { return kInflateLds; }

static void inflate_attributes() {
    static std::atomic<unsigned long long> done{0};   // one bit per device: the attributes are per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_relaxed) & bit) return;
    for (const void* k : {reinterpret_cast<const void*>(inflate_kernel<0>), reinterpret_cast<const void*>(inflate_kernel<1>), reinterpret_cast<const void*>(inflate_kernel<2>), reinterpret_cast<const void*>(inflate_kernel<3>),
                          reinterpret_cast<const void*>(find_starts_kernel), reinterpret_cast<const void*>(chunk_crc_kernel)})
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFindLds) != hipSuccess) (void)hipGetLastError();
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(piece_maps_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 131072) != hipSuccess) (void)hipGetLastError();
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(group_windows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 65536) != hipSuccess) (void)hipGetLastError();
    done.fetch_or(bit, std::memory_order_relaxed);
}

void launch_inflate_bgzf(const InflateJob* jobs_dev, int njobs, uint32_t* errors_dev, hipStream_t st) {
    if (njobs <= 0) return;
    inflate_attributes();
    hipLaunchKernelGGL(inflate_kernel<0>, dim3((unsigned)njobs), dim3(64), kInflateLds, st, jobs_dev, nullptr, 0, nullptr, nullptr, nullptr, nullptr, errors_dev);
}

// Single-member gzip files on the device: block starts -> piece lengths -> offsets -> symbols -> windows -> text -> CRCs.
// npieces = sum of the files' nguess; nchunks = sum of their 64 KiB text chunks (chunk0_dev: first chunk of each file, nfiles + 1 entries).
void launch_gunzip_members(const RawFile* files_dev, int nfiles, int npieces, int ngroups, int nchunks, uint64_t* starts, uint32_t* tables_dev, size_t stride,
                           const uint32_t* chunk0_dev, uint32_t* crcs_dev, uint32_t* errors_dev, hipStream_t st) {
    if (nfiles <= 0 || npieces <= 0) return;
    inflate_attributes();
    uint32_t *lens = tables_dev, *offs = tables_dev + stride, *over = tables_dev + 2 * stride, *abase = tables_dev + 3 * stride;
    const dim3 grid((unsigned)npieces), wave(64);
    hipLaunchKernelGGL(find_starts_kernel, grid, wave, kFindLds, st, files_dev, nfiles, starts);
    hipLaunchKernelGGL(inflate_kernel<3>, grid, wave, kInflateLds, st, nullptr, files_dev, nfiles, starts, lens, over, nullptr, errors_dev);
    hipLaunchKernelGGL(inflate_kernel<1>, grid, wave, kInflateLds, st, nullptr, files_dev, nfiles, starts, lens, over, nullptr, errors_dev);   // (the pieces marked in `over` only)
    hipLaunchKernelGGL(piece_offsets_kernel, dim3((unsigned)nfiles), wave, 0, st, files_dev, lens, over, offs, abase, errors_dev);
    hipLaunchKernelGGL(inflate_kernel<2>, grid, wave, kInflateLds, st, nullptr, files_dev, nfiles, starts, lens, over, abase, errors_dev);
    hipLaunchKernelGGL(piece_maps_kernel, dim3((unsigned)ngroups), dim3(1024), 131072, st, files_dev, nfiles, lens, over, abase, errors_dev);
    hipLaunchKernelGGL(group_windows_kernel, dim3((unsigned)nfiles), dim3(1024), 65536, st, files_dev, errors_dev);
    if (nchunks > 0) {
        hipLaunchKernelGGL(translate_kernel, dim3((unsigned)nchunks), dim3(256), 0, st, files_dev, nfiles, chunk0_dev, lens, offs, over, abase, errors_dev);
        hipLaunchKernelGGL(chunk_crc_kernel, dim3((unsigned)nchunks), wave, kInflateLds, st, files_dev, nfiles, chunk0_dev, crcs_dev, errors_dev);
    }
}

}  // namespace dd
