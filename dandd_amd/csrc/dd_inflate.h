// dd_inflate.h -- parallel decompression of ONE gzip member, host code only (real genome
// directories hold .fa.gz / .fna.gz, /root/reference/lib/species_specifics.py:93, and a deflate stream is serial: one
// 3 Gbp assembly through libdeflate is ~5 s on one core while the GPU needs 0.1 s for it).
//
// Two cases:
//  * BGZF (bgzip: every <= 64 KiB block is a gzip member that says its own compressed size in a 'BC' extra field):
//    blocks are independent, the loader threads inflate them straight into place.
//  * a plain gzip member (gzip, pigz, NCBI's .fna.gz): the stream is cut into pieces at deflate block boundaries found
//    by trial decoding (a dynamic-Huffman block header followed by a whole block of text bytes and another valid
//    header does not occur by chance), every piece is decoded by its own thread WITHOUT its 32 KiB history -- a copy
//    that reaches back before the piece yields 16-bit placeholders naming the history position -- and once the piece
//    before it is known the placeholders are replaced (the two-pass scheme of pugz, Kerbiriou & Chikhi 2019, written
//    from its published description).  A piece must end exactly where the next one was found to start, the member's
//    ISIZE and CRC-32 are checked at the end; anything unexpected returns "not applicable" and the caller's serial
//    path (libdeflate / zlib, dd_io.h) decides and words the error.
// Literals are required to be text (tab, CR, LF, 0x20..0x7E) only while LOOKING for a block start; decoding itself is
// plain inflate.
#pragma once
#include <stdint.h>
#include <string.h>
#include <sys/mman.h>
#include <zlib.h>

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

namespace dd {
namespace inflate_detail {

struct Bits {
    const uint8_t* p;
    size_t n;        // bytes
    size_t pos;      // next byte to load
    uint64_t buf = 0;
    int cnt = 0;     // valid bits in buf
    bool over = false;
    Bits(const uint8_t* d, size_t len, uint64_t bitpos) : p(d), n(len), pos((size_t)(bitpos >> 3)) {
        refill();
        const int skip = (int)(bitpos & 7);
        buf >>= skip;
        cnt -= skip;
    }
    inline void refill() {
        if (pos + 8 <= n) {
            // eight bytes at once: the bits above `cnt` that do not make a whole byte are ORed in again, identically, by
            // the next refill (the stream position only advances by whole bytes)
            uint64_t w;
            memcpy(&w, p + pos, 8);
            buf |= w << cnt;
            const int bytes = (63 - cnt) >> 3;
            pos += (size_t)bytes;
            cnt += bytes * 8;
            return;
        }
        while (cnt <= 56) {
            if (pos < n) buf |= (uint64_t)p[pos] << cnt;
            else if (pos > n + 8) { over = true; }
            ++pos;   // (bytes beyond the end read as zero; `over` trips once decoding runs well past it)
            cnt += 8;
        }
    }
    inline uint32_t peek(int k) const { return (uint32_t)(buf & ((1ull << k) - 1)); }
    inline void drop(int k) { buf >>= k; cnt -= k; }
    inline uint32_t take(int k) {
        if (cnt < k) refill();
        const uint32_t v = peek(k);
        drop(k);
        return v;
    }
    uint64_t bitpos() const { return (uint64_t)pos * 8 - (uint64_t)cnt; }
};

// canonical Huffman code: fast table for codes of <= FAST bits, puff-style count/symbol walk for the rest
struct Huff {
    static constexpr int FAST = 10;
    uint16_t fast[1 << FAST];      // (symbol << 4) | length, 0 = longer code or invalid
    uint16_t count[16], symbol[320];
    // literal/length codes only: two symbols at once where 11 bits hold two literal codes (DNA text under gzip -1 is
    // ~2.2 bits per base, nearly all of them literals).  Entry: bits 0..4 = bits consumed, bits 5..6 = literals held
    // (0: not a literal or a longer code -> decode() decides), bits 8..15 / 16..23 the literals.
    static constexpr int PAIR = 11;
    uint32_t pair[1 << PAIR];
    // one lookup per symbol for everything else a FAST-bit code can be: bits 0..3 code length, bits 4..6 kind (0 = longer
    // code / invalid, 1 literal, 2 end of block, 3 length or distance), bits 7..10 extra bits, bits 11.. base (or the literal)
    uint32_t info[1 << FAST];
    void build_info(const uint16_t* base, const uint8_t* extra, int first, int nsym) {   // symbols first .. first+nsym-1 carry (base, extra)
        for (uint32_t idx = 0; idx < (1u << FAST); ++idx) {
            const uint16_t e = fast[idx];
            const int l = e & 15, sym = e >> 4;
            uint32_t v = 0;
            if (e) {
                if (first && sym < 256) v = (uint32_t)l | (1u << 4) | ((uint32_t)sym << 11);
                else if (first && sym == 256) v = (uint32_t)l | (2u << 4);
                else if (sym >= first && sym < first + nsym) v = (uint32_t)l | (3u << 4) | ((uint32_t)extra[sym - first] << 7) | ((uint32_t)base[sym - first] << 11);
            }
            info[idx] = v;
        }
    }
    int maxlen = 0;
    void build_pairs() {
        for (uint32_t idx = 0; idx < (1u << PAIR); ++idx) {
            const uint16_t e1 = fast[idx & ((1u << FAST) - 1)];
            const int l1 = e1 & 15, s1 = e1 >> 4;
            uint32_t v = 0;
            if (e1 && s1 < 256) {
                v = (uint32_t)l1 | (1u << 5) | ((uint32_t)s1 << 8);
                const uint16_t e2 = fast[(idx >> l1) & ((1u << FAST) - 1)];
                const int l2 = e2 & 15, s2 = e2 >> 4;
                if (e2 && s2 < 256 && l1 + l2 <= PAIR) v = (uint32_t)(l1 + l2) | (2u << 5) | ((uint32_t)s1 << 8) | ((uint32_t)s2 << 16);
            }
            pair[idx] = v;
        }
    }
    // returns false when the lengths do not form a usable code (over-subscribed; incomplete unless `allow_incomplete`)
    bool build(const uint8_t* len, int n, bool allow_incomplete) {
        memset(count, 0, sizeof count);
        for (int i = 0; i < n; ++i) ++count[len[i]];
        if (count[0] == n) return false;
        int left = 1;
        maxlen = 0;
        for (int l = 1; l <= 15; ++l) {
            left <<= 1;
            left -= count[l];
            if (left < 0) return false;
            if (count[l]) maxlen = l;
        }
        if (left > 0 && !(allow_incomplete && n - count[0] == 1)) return false;
        uint16_t offs[16];
        offs[1] = 0;
        for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
        for (int i = 0; i < n; ++i)
            if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
        memset(fast, 0, sizeof fast);
        // canonical codes, bit-reversed into the LSB-first table
        uint32_t code = 0;
        int idx = 0;
        for (int l = 1; l <= 15; ++l) {
            for (int c = 0; c < count[l]; ++c, ++code, ++idx) {
                if (l > FAST) continue;
                uint32_t rev = 0;
                for (int b = 0; b < l; ++b) rev |= ((code >> b) & 1u) << (l - 1 - b);
                for (uint32_t f = rev; f < (1u << FAST); f += 1u << l) fast[f] = (uint16_t)((symbol[idx] << 4) | l);
            }
            code <<= 1;
        }
        return true;
    }
    inline int decode(Bits& b) const {
        if (b.cnt < 15) b.refill();
        const uint16_t e = fast[b.peek(FAST)];
        if (e) {
            b.drop(e & 15);
            return e >> 4;
        }
        int code = 0, first = 0, index = 0;
        uint64_t bits = b.buf;
        for (int l = 1; l <= 15; ++l) {
            code |= (int)(bits & 1);
            bits >>= 1;
            const int c = count[l];
            if (code - c < first) {
                b.drop(l);
                return symbol[index + (code - first)];
            }
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
};

static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

struct BlockCodes {
    Huff lit, dist;
    bool read_dynamic(Bits& b) {
        const int hlit = (int)b.take(5) + 257, hdist = (int)b.take(5) + 1, hclen = (int)b.take(4) + 4;
        if (hlit > 286 || hdist > 30) return false;
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t cl[19] = {0};
        for (int i = 0; i < hclen; ++i) cl[order[i]] = (uint8_t)b.take(3);
        Huff clc;
        if (!clc.build(cl, 19, false)) return false;
        uint8_t lens[320];
        int i = 0;
        while (i < hlit + hdist) {
            const int s = clc.decode(b);
            if (s < 0 || b.over) return false;
            if (s < 16) {
                lens[i++] = (uint8_t)s;
            } else {
                int rep, val = 0;
                if (s == 16) {
                    if (!i) return false;
                    val = lens[i - 1];
                    rep = 3 + (int)b.take(2);
                } else if (s == 17) {
                    rep = 3 + (int)b.take(3);
                } else {
                    rep = 11 + (int)b.take(7);
                }
                if (i + rep > hlit + hdist) return false;
                while (rep--) lens[i++] = (uint8_t)val;
            }
        }
        if (!lens[256]) return false;   // no end-of-block code
        return lit.build(lens, hlit, false) && dist.build(lens + hlit, hdist, true);
    }
    void set_fixed() {
        uint8_t l[288];
        for (int i = 0; i < 144; ++i) l[i] = 8;
        for (int i = 144; i < 256; ++i) l[i] = 9;
        for (int i = 256; i < 280; ++i) l[i] = 7;
        for (int i = 280; i < 288; ++i) l[i] = 8;
        lit.build(l, 288, false);
        uint8_t d[30];
        for (int i = 0; i < 30; ++i) d[i] = 5;
        dist.build(d, 30, true);
    }
};

inline bool is_text(int c) { return (c >= 0x20 && c <= 0x7e) || c == '\n' || c == '\r' || c == '\t'; }

// 16-bit symbols of one piece; reused from piece to piece by the thread that owns it (fresh memory costs a page fault
// per 2048 symbols: a third of the decoding time)
struct SymBuf {
    uint16_t* p = nullptr;
    size_t len = 0, cap = 0;
    SymBuf() = default;
    SymBuf(const SymBuf&) = delete;
    SymBuf& operator=(const SymBuf&) = delete;
    ~SymBuf() { free(p); }
    bool room(size_t extra) {   // make sure `extra` more symbols fit
        if (len + extra <= cap) return true;
        // 2 MiB-aligned, huge pages asked for: a piece's symbols are ~100 MB of fresh memory per thread
        const size_t kHuge = (size_t)2 << 20;
        size_t want = std::max(cap * 2, len + extra + ((size_t)1 << 20));
        want = (want * sizeof(uint16_t) + kHuge - 1) / kHuge * kHuge;
        void* q = nullptr;
        if (posix_memalign(&q, kHuge, want) != 0 || !q) return false;
        (void)madvise(q, want, MADV_HUGEPAGE);
        if (len) memcpy(q, p, len * sizeof(uint16_t));
        free(p);
        p = static_cast<uint16_t*>(q);
        cap = want / sizeof(uint16_t);
        return true;
    }
};

// Decode deflate blocks from bit `start` into 16-bit symbols (0..255 bytes, 256 + w = byte w of the unknown 32 KiB
// window that precedes the piece; `known_start`: there is no such window, a reference before the piece is an error).
// Stops after the final block, or at a block boundary at or beyond bit `stop_at` (0 = never), or -- trial mode,
// `trial_blocks` > 0, out == nullptr -- after that many blocks, demanding text literals.
// Returns 0 ok, -1 bad data, -2 out of memory.
struct PieceResult {
    uint64_t end_bit = 0;
    bool final_seen = false;
};
inline int decode_piece(const uint8_t* in, size_t n, uint64_t start, uint64_t stop_at, bool known_start, int trial_blocks,
                        SymBuf* out, PieceResult* res, size_t max_symbols = ~(size_t)0) {
    Bits b(in, n, start);
    BlockCodes codes;
    size_t produced = 0;      // symbols (trial mode keeps none)
    int blocks = 0;
    if (out) out->len = 0;
    for (;;) {
        if (b.over) return -1;
        const uint32_t bfinal = b.take(1), btype = b.take(2);
        if (btype == 3) return -1;
        if (btype == 0) {
            b.drop(b.cnt & 7);   // to the byte boundary
            if (b.cnt < 32) b.refill();
            const uint32_t len = b.take(16), nlen = b.take(16);
            if ((len ^ nlen) != 0xffffu) return -1;
            if (trial_blocks && len == 0) return -1;   // (an empty stored block is too weak a signature to start from)
            if (out && (out->len > max_symbols || !out->room(len))) return out->len > max_symbols ? -1 : -2;
            for (uint32_t i = 0; i < len; ++i) {
                const int c = (int)b.take(8);
                if (b.over) return -1;
                if (trial_blocks) {
                    if (!is_text(c)) return -1;
                } else {
                    out->p[out->len++] = (uint16_t)c;
                }
                ++produced;
            }
        } else {
            if (btype == 1) {
                if (trial_blocks) return -1;   // (fixed codes validate nothing: never a starting point)
                codes.set_fixed();
            } else if (!codes.read_dynamic(b)) {
                return -1;
            }
            if (trial_blocks) {
                for (;;) {
                    const int s = codes.lit.decode(b);
                    if (s < 0 || b.over) return -1;
                    if (s < 256) {
                        if (!is_text(s)) return -1;
                        ++produced;
                        continue;
                    }
                    if (s == 256) break;
                    if (s > 285) return -1;
                    if (b.cnt < 32) b.refill();
                    produced += (size_t)(kLenBase[s - 257] + (int)b.take(kLenExtra[s - 257]));
                    const int ds = codes.dist.decode(b);
                    if (ds < 0 || ds > 29) return -1;
                    if (b.cnt < 16) b.refill();
                    b.drop(kDistExtra[ds]);
                }
            } else {
                codes.lit.build_pairs();
                codes.lit.build_info(kLenBase, kLenExtra, 257, 29);
                codes.dist.build_info(kDistBase, kDistExtra, 0, 30);
                uint16_t* o = out->p;
                size_t at = out->len, lim = out->cap;
                const long long floor = known_start ? 0 : -32768;
                for (;;) {
                    if (at + 260 > lim) {   // one check per step covers two literals or the longest copy
                        if (at > max_symbols || b.over) return -1;   // (a damaged or truncated stream can decode to anything: bounded, then refused)
                        out->len = at;
                        if (!out->room(65536)) return -2;
                        o = out->p;
                        lim = out->cap;
                    }
                    if (b.cnt < 32) b.refill();
                    const uint32_t pe = codes.lit.pair[b.peek(Huff::PAIR)];
                    if (pe) {
                        o[at] = (uint16_t)((pe >> 8) & 0xff);
                        o[at + 1] = (uint16_t)(pe >> 16);
                        at += (pe >> 5) & 3;
                        b.drop((int)(pe & 31));
                        continue;
                    }
                    // (after the refill above >= 32 bits are in hand: a length code with its extra bits takes <= 20 of them)
                    int len;
                    const uint32_t li = codes.lit.info[b.peek(Huff::FAST)];
                    if (li) {
                        b.drop((int)(li & 15));
                        const uint32_t kind = (li >> 4) & 7;
                        if (kind == 1) {
                            o[at++] = (uint16_t)(li >> 11);
                            continue;
                        }
                        if (kind == 2) break;
                        const int ex = (int)((li >> 7) & 15);
                        len = (int)(li >> 11) + (int)b.peek(ex);
                        b.drop(ex);
                    } else {
                        const int s = codes.lit.decode(b);
                        if (s < 256) {
                            if (s < 0) return -1;
                            o[at++] = (uint16_t)s;
                            continue;
                        }
                        if (s == 256) break;
                        if (s > 285) return -1;
                        if (b.cnt < 32) b.refill();
                        len = kLenBase[s - 257] + (int)b.take(kLenExtra[s - 257]);
                    }
                    if (b.cnt < 32) b.refill();
                    int dist;
                    const uint32_t di = codes.dist.info[b.peek(Huff::FAST)];
                    if (di) {
                        b.drop((int)(di & 15));
                        const int ex = (int)((di >> 7) & 15);
                        dist = (int)(di >> 11) + (int)b.peek(ex);
                        b.drop(ex);
                    } else {
                        const int ds = codes.dist.decode(b);
                        if (ds < 0 || ds > 29) return -1;
                        if (b.cnt < 16) b.refill();
                        dist = kDistBase[ds] + (int)b.take(kDistExtra[ds]);
                    }
                    const long long from = (long long)at - dist;
                    if (from < floor) return -1;
                    if (from >= 0) {
                        const uint16_t* src = o + from;
                        uint16_t* dst = o + at;
                        if (dist >= len) memcpy(dst, src, (size_t)len * 2);
                        else
                            for (int i = 0; i < len; ++i) dst[i] = src[i];
                    } else {
                        for (int i = 0; i < len; ++i) {
                            const long long sp = from + i;
                            o[at + (size_t)i] = sp >= 0 ? o[sp] : (uint16_t)(256 + 32768 + sp);
                        }
                    }
                    at += (size_t)len;
                }
                if (b.over) return -1;
                produced = at;
                out->len = at;
            }
        }
        ++blocks;
        if (bfinal) {
            res->final_seen = true;
            break;
        }
        if (trial_blocks && blocks >= trial_blocks) break;
        if (stop_at && b.bitpos() >= stop_at) break;
    }
    res->end_bit = b.bitpos();
    if (trial_blocks && produced < 1024 && !res->final_seen) return -1;
    return 0;
}

// first bit position >= from (and < limit) at which two text blocks in a row decode cleanly and the next header is valid
inline uint64_t find_block_start(const uint8_t* in, size_t n, uint64_t from, uint64_t limit) {
    for (uint64_t bit = from; bit < limit; ++bit) {
        // cheap rejects before a full trial: BTYPE must be 0 or 2 (bits 1..2 of the header)
        const uint32_t hdr = (uint32_t)((in[bit >> 3] | ((uint32_t)(((bit >> 3) + 1 < n) ? in[(bit >> 3) + 1] : 0) << 8)) >> (bit & 7));
        const uint32_t btype = (hdr >> 1) & 3;
        if (btype != 2 && btype != 0) continue;
        if (hdr & 1) continue;    // a final block is never a useful starting point
        PieceResult r;
        if (decode_piece(in, n, bit, 0, false, 3, nullptr, &r) == 0) return bit;
    }
    return ~0ull;
}

inline size_t gzip_header_len(const uint8_t* in, size_t n) {   // 0 = not a gzip member / truncated
    if (n < 18 || in[0] != 0x1f || in[1] != 0x8b || in[2] != 8) return 0;
    const int flg = in[3];
    size_t p = 10;
    if (flg & 4) {
        if (p + 2 > n) return 0;
        p += 2 + ((size_t)in[p] | ((size_t)in[p + 1] << 8));
    }
    if (flg & 8) {
        while (p < n && in[p]) ++p;
        ++p;
    }
    if (flg & 16) {
        while (p < n && in[p]) ++p;
        ++p;
    }
    if (flg & 2) p += 2;
    return p < n ? p : 0;
}

// BGZF: the size of the block that starts at `in` (its 'BC' subfield), 0 if it is not one
inline size_t bgzf_block_size(const uint8_t* in, size_t n) {
    if (n < 18 || in[0] != 0x1f || in[1] != 0x8b || in[2] != 8 || !(in[3] & 4)) return 0;
    const size_t xlen = (size_t)in[10] | ((size_t)in[11] << 8);
    if (12 + xlen > n) return 0;
    for (size_t q = 12; q + 4 <= 12 + xlen;) {
        const size_t slen = (size_t)in[q + 2] | ((size_t)in[q + 3] << 8);
        if (in[q] == 'B' && in[q + 1] == 'C' && slen == 2 && q + 6 <= 12 + xlen) {
            const size_t bsize = ((size_t)in[q + 4] | ((size_t)in[q + 5] << 8)) + 1;
            return bsize >= 26 && bsize <= n ? bsize : 0;
        }
        q += 4 + slen;
    }
    return 0;
}

template <typename F>
inline void run_threads(int nthreads, F&& body) {
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads; ++t) pool.emplace_back([&body, t] { body(t); });
    body(0);
    for (auto& th : pool) th.join();
}

}  // namespace inflate_detail

// Every member of a BGZF file inflated in parallel straight into `out` (which reserve() grows).  `inflate_member`
// decodes one whole gzip member: (in, n, dst, cap, &used, &made) -> true on success.  Returns 1 done, 0 not BGZF /
// anything unexpected (caller falls back), -1 out of memory.
template <typename Buf, typename MemberFn>
inline int gunzip_bgzf_parallel(const uint8_t* in, size_t n, Buf& out, int nthreads, MemberFn&& inflate_member) {
    using namespace inflate_detail;
    if (!bgzf_block_size(in, n)) return 0;
    struct Blk {
        size_t in_off, in_len, out_off, out_len;
    };
    std::vector<Blk> blks;
    size_t p = 0, total = 0;
    while (p < n) {
        const size_t bs = bgzf_block_size(in + p, n - p);
        if (!bs) {
            // trailing zeros are tolerated (as gzread tolerates them); anything else: not a clean BGZF file
            for (size_t q = p; q < n; ++q)
                if (in[q]) return 0;
            break;
        }
        const size_t isize = (size_t)in[p + bs - 4] | ((size_t)in[p + bs - 3] << 8) | ((size_t)in[p + bs - 2] << 16) | ((size_t)in[p + bs - 1] << 24);
        if (isize > 65536) return 0;
        blks.push_back(Blk{p, bs, total, isize});
        total += isize;
        p += bs;
    }
    if (!out.reserve(std::max(out.cap, total + 64))) return -1;
    std::atomic<size_t> next{0};
    std::atomic<int> bad{0};
    run_threads(std::max(1, nthreads), [&](int) {
        for (;;) {
            const size_t i0 = next.fetch_add(64);
            if (i0 >= blks.size() || bad.load()) return;
            for (size_t i = i0; i < std::min(blks.size(), i0 + 64); ++i) {
                size_t used = 0, made = 0;
                // (a member must also END where its BSIZE says: bytes left over inside a block are not a clean BGZF file)
                if (!inflate_member(in + blks[i].in_off, blks[i].in_len, out.p + blks[i].out_off, blks[i].out_len, &used, &made) ||
                    made != blks[i].out_len || used != blks[i].in_len) {
                    bad.store(1);
                    return;
                }
            }
        }
    });
    if (bad.load()) return 0;
    out.len = total;
    return 1;
}

// One plain gzip member (the whole of `in`) decoded by `nthreads` threads, in rounds of `nthreads` pieces so that the
// 16-bit symbol buffers (2 bytes per output byte) stay bounded and are reused.  Returns 1 done, 0 not applicable (not
// gzip, too small to be worth it, several members, no block start found, a piece that does not end where the next
// begins, wrong ISIZE / CRC: the serial path decides), -1 out of memory.
typedef uint32_t (*Crc32Fn)(uint32_t, const void*, size_t);   // libdeflate_crc32's signature; nullptr = zlib's crc32
template <typename Buf>
inline int gunzip_member_parallel(const uint8_t* in, size_t n, Buf& out, int nthreads, size_t min_piece = (size_t)4 << 20, Crc32Fn fast_crc = nullptr) {
    using namespace inflate_detail;
    const size_t hdr = gzip_header_len(in, n);
    if (!hdr || n < hdr + 8 + 2 * min_piece || nthreads < 2) return 0;
    const bool trace = getenv("DD_TRACE_FILES") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    const size_t body_end = n - 8;   // if the file is ONE member, its trailer sits here (checked at the end)
    // pieces of 4 .. 16 MiB of compressed bytes, a multiple of the thread count of them when the member is large enough
    // (a round of pieces ends at a barrier: 4 MiB pieces -- 16 rounds for a 3 Gbp genome -- measured 1.5x slower)
    size_t pieces = std::min<size_t>((body_end - hdr) / min_piece, std::max<size_t>((size_t)nthreads, (body_end - hdr) / (4 * min_piece)));
    if (pieces > (size_t)nthreads) pieces = pieces / (size_t)nthreads * (size_t)nthreads;
    if (pieces < 2) return 0;
    // piece starts: the true start, then block boundaries found near the even cuts
    std::vector<uint64_t> start(pieces, ~0ull);
    start[0] = (uint64_t)hdr * 8;
    {
        std::atomic<size_t> next{1};
        run_threads((int)std::min<size_t>((size_t)nthreads, pieces), [&](int) {
            for (;;) {
                const size_t c = next.fetch_add(1);
                if (c >= pieces) return;
                const uint64_t from = ((uint64_t)hdr + (uint64_t)(body_end - hdr) * (uint64_t)c / (uint64_t)pieces) * 8;
                start[c] = find_block_start(in, body_end, from, std::min<uint64_t>(from + ((uint64_t)1 << 23), (uint64_t)body_end * 8));
            }
        });
    }
    // pieces whose start was not found merge into their predecessor
    std::vector<uint64_t> st;
    for (uint64_t s : start)
        if (s != ~0ull && (st.empty() || s > st.back())) st.push_back(s);
    if (st.size() < 2) return 0;
    const size_t np = st.size();
    const double t_found = now();
    const uint32_t want_crc = (uint32_t)in[n - 8] | ((uint32_t)in[n - 7] << 8) | ((uint32_t)in[n - 6] << 16) | ((uint32_t)in[n - 5] << 24);
    const uint32_t want_size = (uint32_t)in[n - 4] | ((uint32_t)in[n - 3] << 8) | ((uint32_t)in[n - 2] << 16) | ((uint32_t)in[n - 1] << 24);
    // ISIZE is the size modulo 2^32: right for anything below 4 GiB, a floor otherwise (the buffer then grows)
    if (!out.reserve(std::max(out.cap, std::max<size_t>((size_t)want_size, 2 * n) + 64))) return -1;
    const int T = (int)std::min<size_t>((size_t)nthreads, np);
    std::vector<SymBuf> sym((size_t)T);
    std::vector<PieceResult> res((size_t)T);
    std::vector<uint8_t> win_prev, win_next(32768);   // the 32 KiB before the first piece of the round
    std::vector<std::vector<uint8_t>> win((size_t)T);
    std::vector<uint32_t> crcs((size_t)T);
    size_t total = 0;
    uint32_t crc_all = (uint32_t)crc32(0L, Z_NULL, 0);
    double t_decode = 0, t_resolve = 0;
    bool final_seen = false;
    uint64_t end_bit = 0;
    for (size_t c0 = 0; c0 < np; c0 += (size_t)T) {
        const size_t cnt = std::min<size_t>((size_t)T, np - c0);
        std::atomic<int> status{0};
        const double ta = now();
        run_threads((int)cnt, [&](int t) {
            const size_t c = c0 + (size_t)t;
            const size_t packed = (size_t)((c + 1 < np ? st[c + 1] : (uint64_t)body_end * 8) - st[c]) / 8;
            sym[(size_t)t].len = 0;
            if (!sym[(size_t)t].room(packed * 4 + 65536)) {
                status.store(-2);
                return;
            }
            // (text inflates 3-5x; a piece that claims more than 32x, or more than the whole member's ISIZE, is either
            // damaged or so repetitive that the serial decoder -- which needs the output buffer only, not two bytes of
            // symbol per output byte and thread -- is the better tool: "not applicable")
            size_t max_symbols = packed * 32 + ((size_t)1 << 20);
            if ((size_t)want_size >= n) max_symbols = std::min(max_symbols, (size_t)want_size + 1);  // (ISIZE is modulo 2^32: one smaller than the file has wrapped)
            const int rc = decode_piece(in, body_end, st[c], c + 1 < np ? st[c + 1] : 0, c == 0, 0, &sym[(size_t)t], &res[(size_t)t],
                                        max_symbols);
            if (rc) status.store(rc);
        });
        t_decode += now() - ta;
        // (a symbol buffer that could not grow is "not applicable" too: the serial path decides, with the output buffer alone)
        if (status.load()) return 0;
        // a piece ends exactly where the next was found to start; only the very last sees the final block
        std::vector<size_t> off(cnt + 1, total);
        for (size_t t = 0; t < cnt; ++t) {
            const size_t c = c0 + t;
            if (c + 1 < np ? (res[t].end_bit != st[c + 1] || res[t].final_seen) : !res[t].final_seen) return 0;
            if (c + 1 < np && sym[t].len < 32768) return 0;
            off[t + 1] = off[t] + sym[t].len;
        }
        final_seen = res[cnt - 1].final_seen;
        end_bit = res[cnt - 1].end_bit;
        if (off[cnt] + 64 > out.cap) {
            out.len = total;   // (reserve() carries `len` bytes over)
            if (!out.reserve(std::max(out.cap * 2, off[cnt] + 64))) return -1;
        }
        // windows: the 32 KiB before every piece of the round, front to back (each needs the one before it)
        for (size_t t = 0; t < cnt; ++t) {
            win[t] = t ? win_next : win_prev;
            if (sym[t].len < 32768) break;      // (only the last piece of the member may be that short)
            const uint16_t* s = sym[t].p + sym[t].len - 32768;
            const bool first = c0 + t == 0;
            for (int i = 0; i < 32768; ++i) {
                const uint16_t v = s[i];
                if (v < 256) win_next[(size_t)i] = (uint8_t)v;
                else if (first) return 0;       // the first piece has no history to refer to
                else win_next[(size_t)i] = win[t][v - 256];
            }
        }
        win_prev = win_next;
        const double tb = now();
        std::atomic<int> bad{0};
        run_threads((int)cnt, [&](int t) {
            uint8_t* dst = out.p + off[(size_t)t];
            const uint16_t* s = sym[(size_t)t].p;
            const size_t m = sym[(size_t)t].len;
            if (c0 + (size_t)t == 0) {
                for (size_t i = 0; i < m; ++i) {
                    if (s[i] >= 256) {
                        bad.store(1);
                        return;
                    }
                    dst[i] = (uint8_t)s[i];
                }
            } else {
                // one table for bytes and placeholders alike: no branch per symbol
                std::vector<uint8_t> lut(256 + 32768);
                for (int v = 0; v < 256; ++v) lut[(size_t)v] = (uint8_t)v;
                memcpy(lut.data() + 256, win[(size_t)t].data(), 32768);
                const uint8_t* L = lut.data();
                for (size_t i = 0; i < m; ++i) dst[i] = L[s[i]];
            }
            uint32_t k;
            if (fast_crc) {
                k = fast_crc(0, dst, m);
            } else {
                k = (uint32_t)crc32(0L, Z_NULL, 0);
                for (size_t a = 0; a < m; a += (size_t)1 << 30) k = (uint32_t)crc32(k, dst + a, (uInt)std::min<size_t>(m - a, (size_t)1 << 30));
            }
            crcs[(size_t)t] = k;
        });
        if (bad.load()) return 0;
        for (size_t t = 0; t < cnt; ++t) crc_all = (uint32_t)crc32_combine(crc_all, crcs[t], (z_off_t)(off[t + 1] - off[t]));
        total = off[cnt];
        t_resolve += now() - tb;
    }
    // the member must end where the file's last 8 bytes begin (a second member, or junk, is the serial path's business)
    if (!final_seen || (end_bit + 7) / 8 != (uint64_t)body_end) return 0;
    if ((uint32_t)total != want_size || crc_all != want_crc) return 0;
    out.len = total;
    if (trace)
        fprintf(stderr, "[gunzip_member_parallel] %zu -> %zu bytes, %zu pieces, %d threads: block starts %.3f s, decode %.3f s, resolve + crc %.3f s, all %.3f s\n",
                n, total, np, nthreads, t_found - t_begin, t_decode, t_resolve, now() - t_begin);
    return 1;
}

}  // namespace dd
