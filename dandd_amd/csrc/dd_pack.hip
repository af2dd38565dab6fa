// dd_pack.hip -- K0: FASTA bytes in HBM -> 2-bit token stream (+ 1-bit BREAK mask).
//
// First half of what one `dashing sketch` process does before hashing (kseq record parsing +
// bonsai's 2-bit encoder; command built at /root/reference/lib/sketch_classes.py:351-366):
// headers and newlines are dropped, A/C/G/T (either case) become 0..3, every other byte and
// every record boundary becomes a BREAK that resets the k-mer windows.  The oracle's
// statement of the same rules is oracle/dd_oracle.c:orc_records + orc_tokenize.
// Record rules (kseq's FASTA rules as recalled, oracle/POLICIES.md P10): everything before the FIRST '>' or '@' of the
// buffer (anywhere, not only at a line start) is skipped -- pack_first finds it, the bytes in front of it read as
// newlines --; a line that starts with '>' or '@' is a header line; a '\r' right in front of a line end (or of the end of
// the buffer) is dropped, any other '\r' is an ambiguous byte.  FASTQ records ('+' lines, quality text) are resolved on
// the host before the bytes get here (dd_io.h: fastq_to_fasta; dd_sketch_device takes FASTA).
//
// The only state that crosses a byte is "am I inside a header line" (1 bit).  A span of bytes
// acts on that bit as IDENTITY (no newline and no line start in it), or as a CONSTANT (it contains a
// newline, or starts at a line start: its end state is then decided locally).  Spans compose like
// that at every level: 64-byte thread spans inside a wave (two ballots and a find-first-set),
// waves inside a 16 KiB chunk (LDS), chunks inside a genome (pack_scan).  Three launches, batched
// over all genomes of a call (gridDim.y = genome):
//   pack_stats : per chunk -> its IDENTITY/CONSTANT summary and its token count under both
//                possible incoming states
//   pack_scan  : one workgroup per genome: incoming state and exclusive token offset of every chunk
//   pack_write : per chunk -> each thread assembles its tokens in registers (32-bit code pieces,
//                16-bit BREAK pieces), ORs them into the chunk's LDS image at their final bit
//                positions, the image is copied out; only the first and last word of a chunk can be
//                shared with a neighbour and go out through atomicOr (pack_scan zeroed them)
// HBM-bound by design (1 B read twice + 0.375 B written per base); see DESIGN.md for the measured rate.
#include "dd_common.h"
#include "dd_kernels.h"

namespace dd {
namespace {

constexpr int T = kPackThreads;        // 256 threads = 4 waves
constexpr int SUB = kPackBytesPerThread / 16;  // 16-byte pieces per thread
constexpr int NW = T / 64;

enum : int { KIND_ID = 0, KIND_C0 = 1, KIND_C1 = 2 };  // action of a span on the "in header" bit

DD_D uint32_t base_code(uint32_t c) {
    uint32_t x = c | 0x20u;
    return x == 'a' ? 0u : x == 'c' ? 1u : x == 'g' ? 2u : x == 't' ? 3u : 4u;
}

DD_D uint64_t pack_u64(uint32_t hi, uint32_t lo) { return ((uint64_t)hi << 32) | lo; }

struct Bytes16 {
    uint32_t w[4];
    DD_D uint32_t at(int i) const { return (w[i >> 2] >> ((i & 3) * 8)) & 0xFFu; }
};

// 16 bytes; bytes at or beyond n, and bytes in front of the buffer's first header character (`first`), read as '\n':
// outside a header line a newline emits nothing, inside one it closes the line -- a header that the end of the buffer
// cuts short still yields its record's BREAK -- and what follows it is a line start.
DD_D Bytes16 load16(const uint8_t* fa, size_t n, size_t pos, size_t first) {
    Bytes16 b;
    if (pos + 16 <= n && pos >= first) {
        const uint4 v = gload16(fa + pos);
        b.w[0] = v.x; b.w[1] = v.y; b.w[2] = v.z; b.w[3] = v.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t w = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                size_t p = pos + 4 * j + i;
                uint32_t c = (p < n && p >= first) ? fa[p] : (uint32_t)'\n';
                w |= c << (8 * i);
            }
            b.w[j] = w;
        }
    }
    return b;
}

// Line state carried through a thread's bytes (same rules as orc_tokenize), plus the two tallies
// that make one pass enough: tokens emitted before the first newline (they exist only if the span was
// entered OUTSIDE a header) and tokens emitted from the first newline on (independent of how it was entered).
struct LineState {
    bool hdr;      // inside a header line
    bool ls;       // next byte is the first of a line
    bool seen_nl;  // a newline has been consumed
    int pre, rest;
};

// Tokens of 16 bytes.  PACK: also assemble them (codes: token j at bits 2j..2j+1, bad: bit j).
// (`after`: the byte behind the 16 -- a '\r' is dropped only in front of a line end)
template <bool PACK>
DD_D int scan16(const Bytes16& b, uint32_t after, LineState& s, uint32_t& codes, uint32_t& bad) {
    int cnt = 0;
    if (PACK) {
        codes = 0;
        bad = 0;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t c = b.at(i);
        if (s.ls) s.hdr = (c == '>' || c == '@');
        if (c == '\n') {
            if (s.hdr) {  // one BREAK per header line
                if (PACK) bad |= 1u << cnt;
                ++cnt;
                if (!PACK) ++s.rest;
            }
            s.hdr = false;
            s.ls = true;
            s.seen_nl = true;
        } else {
            s.ls = false;
            const bool cr_at_eol = c == '\r' && (i < 15 ? b.at(i < 15 ? i + 1 : 15) : after) == '\n';
            if (!(s.hdr || cr_at_eol)) {
                if (PACK) {
                    const uint32_t code = base_code(c);
                    codes |= (code & 3u) << (2 * cnt);
                    bad |= (code >> 2) << cnt;
                } else {
                    if (s.seen_nl) ++s.rest; else ++s.pre;
                }
                ++cnt;
            }
        }
    }
    return cnt;
}

// Bit-parallel view of a thread's 64 bytes (SWAR over the 16 words, no per-byte loop).
struct SpanBits {
    uint32_t codes[4];  // 2 bits per byte: A/a 0, C/c 1, G/g 2, T/t 3 (meaningless where bad)
    uint32_t bad[2];    // 1 bit per byte: not one of A C G T a c g t
    uint32_t nl[2];     // 1 bit per byte: '\n'
    bool plain;         // every byte is '\n' or lies in 0x40..0x7F: no '>', no '\r', nothing that
                        // needs the line machine -- each byte is a base, a BREAK, or a dropped newline
};
DD_D SpanBits span_bits(const Bytes16 (&b)[SUB]) {
    SpanBits r;
    uint32_t other = 0;
    r.bad[0] = r.bad[1] = r.nl[0] = r.nl[1] = 0;
#pragma unroll
    for (int j = 0; j < SUB; ++j) {
        uint32_t cj = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t w = b[j].w[i];
            const uint32_t x = w | 0x20202020u;
            const uint32_t c2 = ((x >> 1) ^ (x >> 2)) & 0x03030303u;        // a c g t -> 0 1 2 3
            const uint32_t expect = __builtin_amdgcn_perm(0u, 0x74676361u, c2);  // "acgt"[c2] per byte
            const uint32_t d = x ^ expect;                                   // zero byte: A/C/G/T
            const uint32_t nz = (((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) & 0x80808080u;
            const uint32_t v = w ^ 0x0A0A0A0Au;                              // zero byte: newline
            const uint32_t nlm = ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u;
            other |= ((((~w) & 0x40404040u) << 1) | (w & 0x80808080u)) & ~nlm;  // outside 0x40..0x7F
            // gather the four 2-bit codes / four flag bits of the word with one multiply each
            cj |= ((c2 * 0x01041040u) >> 24) << (8 * i);
            const int word = 4 * j + i;  // bytes 4*word .. 4*word+3 of the span
            r.bad[word >> 3] |= ((nz * 0x00204081u) >> 28) << (4 * (word & 7));
            r.nl[word >> 3] |= ((nlm * 0x00204081u) >> 28) << (4 * (word & 7));
        }
        r.codes[j] = cj;
    }
    r.plain = other == 0;
    return r;
}

// What a thread's 64 bytes look like from outside: the span's action on the header bit and its
// token count under either incoming bit.  The two cases differ only before the first newline:
// entered inside a header, the bytes up to that newline emit nothing and the newline itself emits the
// record BREAK.  Plain spans (almost all of a genome) get this from the newline mask; the others
// run ONE pass of the byte machine (entered as "not in a header").
struct ThreadSpan {
    Bytes16 b[SUB];
    uint32_t after;   // the byte behind the span ('\n' at the end of the buffer)
    SpanBits bits;
    bool line_start;  // first byte is the first of a line (the incoming bit is then irrelevant)
    bool plain;       // bits.plain and the whole span lies inside the file
    int kind;         // action on the header bit
    int t0, t1;       // tokens if entered outside / inside a header line
};

DD_D ThreadSpan load_span(const uint8_t* fa, size_t n, size_t pos, size_t first) {
    ThreadSpan t;
#pragma unroll
    for (int j = 0; j < SUB; ++j) t.b[j] = load16(fa, n, pos + 16 * j, first);
    t.line_start = (pos <= first) || (pos - 1 < n ? *(const DD_GLOBAL uint8_t*)(fa + pos - 1) == '\n' : false);
    t.bits = span_bits(t.b);
    t.plain = t.bits.plain && pos + kPackBytesPerThread <= n;
    // a byte that is not A/C/G/T at a line start may be a header's '@' ('>' and '+' are outside the plain range anyway):
    // such a span takes the byte machine (lines that start inside an N run do too: a few per genome)
    if (t.plain) {
        const uint64_t starts = (pack_u64(t.bits.nl[1], t.bits.nl[0]) << 1) | (t.line_start ? 1ull : 0ull);
        if (pack_u64(t.bits.bad[1], t.bits.bad[0]) & starts) t.plain = false;
    }
    t.after = '\n';
    if (!t.plain && pos + kPackBytesPerThread < n) t.after = *(const DD_GLOBAL uint8_t*)(fa + pos + kPackBytesPerThread);
    if (pos >= n) {  // tail of the last chunk
        t.kind = KIND_ID;
        t.t0 = t.t1 = 0;
    } else if (t.plain) {
        const int nnl = __builtin_popcount(t.bits.nl[0]) + __builtin_popcount(t.bits.nl[1]);
        t.kind = (nnl || t.line_start) ? KIND_C0 : KIND_ID;  // no '>' in the span: it ends outside a header
        t.t0 = kPackBytesPerThread - nnl;
        if (t.line_start) t.t1 = t.t0;
        else if (nnl == 0) t.t1 = 0;
        else {
            const int q1 = t.bits.nl[0] ? __builtin_ctz(t.bits.nl[0]) : 32 + __builtin_ctz(t.bits.nl[1]);
            t.t1 = (kPackBytesPerThread - 1 - q1) - (nnl - 1) + 1;  // bytes after the first newline, less newlines, + BREAK
        }
    } else {
        LineState s{false, t.line_start, false, 0, 0};
        uint32_t d0, d1;
#pragma unroll
        for (int j = 0; j < SUB; ++j) (void)scan16<false>(t.b[j], j + 1 < SUB ? t.b[j + 1 < SUB ? j + 1 : j].at(0) : t.after, s, d0, d1);
        t.kind = (s.seen_nl || t.line_start) ? (s.hdr ? KIND_C1 : KIND_C0) : KIND_ID;
        t.t0 = s.pre + s.rest;
        t.t1 = t.line_start ? t.t0 : s.rest + (s.seen_nl ? 1 : 0);
    }
    return t;
}

// Header bit entering each thread of the workgroup, given the bit entering the chunk, and the
// chunk's own summary.  determined = false for threads whose incoming bit is the chunk's.
struct Incoming {
    bool hdr;
    bool determined;
    int chunk_kind;
};
DD_D Incoming propagate(int kind, bool chunk_in, int* sm /*[NW]*/) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long cm = __ballot(kind != KIND_ID), vm = __ballot(kind == KIND_C1);
    const unsigned long long below = cm & ((1ull << lane) - 1ull);
    Incoming r;
    bool from_wave = below != 0;
    bool val = false;
    if (from_wave) val = (vm >> (63 - __builtin_clzll(below))) & 1ull;
    if (lane == 0) sm[wv] = cm ? (((vm >> (63 - __builtin_clzll(cm))) & 1ull) ? KIND_C1 : KIND_C0) : KIND_ID;
    __syncthreads();
    bool wave_in = chunk_in, wave_det = false;
    int ck = KIND_ID;
    for (int w = 0; w < NW; ++w) {
        const int k = sm[w];
        if (k != KIND_ID) {
            ck = k;
            if (w < wv) {
                wave_in = (k == KIND_C1);
                wave_det = true;
            }
        }
    }
    r.hdr = from_wave ? val : wave_in;
    r.determined = from_wave || wave_det;
    r.chunk_kind = ck;
    __syncthreads();
    return r;
}

// inclusive sum over the workgroup (32-bit); *total = workgroup sum
DD_D int block_incl_sum(int v, int* sm /*[NW]*/, int* total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) sm[wv] = inc;
    __syncthreads();
    int pre = 0, tot = 0;
    for (int w = 0; w < NW; ++w) {
        if (w < wv) pre += sm[w];
        tot += sm[w];
    }
    *total = tot;
    __syncthreads();
    return pre + inc;
}

// scratch layout of one genome: four arrays of (nchunks + 1) int64
struct Scratch {
    long long *kind_in, *cnt0, *cnt1, *slot, *first;
    DD_D explicit Scratch(const PackGenome& g)
        : kind_in(g.scratch), cnt0(g.scratch + (g.nchunks + 1)), cnt1(g.scratch + 2 * (g.nchunks + 1)),
          slot(g.scratch + 3 * (g.nchunks + 1)), first(g.scratch + 4 * (g.nchunks + 1)) {}
};

// ---------------------------------------------------------------------------------------
// grid = genomes: position of the buffer's first '>' or '@' (n if there is none: kseq then finds no record and the
// sketch stays empty).  A well-formed file has it at byte 0 and the loop ends in its first round.
__global__ __launch_bounds__(1024) void pack_first(const PackGenome* __restrict__ tab) {
    __shared__ unsigned long long found;
    const PackGenome G = tab[blockIdx.x];
    const Scratch S(G);
    if (threadIdx.x == 0) found = ~0ull;
    __syncthreads();
    for (size_t base = 0; base < G.n; base += 1024u * 16u) {
        const size_t pos = base + (size_t)threadIdx.x * 16u;
        if (pos < G.n) {
            const Bytes16 b = load16(G.fa, G.n, pos, 0);
#pragma unroll
            for (int i = 15; i >= 0; --i) {
                const uint32_t c = b.at(i);
                if ((c == '>' || c == '@') && pos + i < G.n) atomicMin(&found, (unsigned long long)(pos + i));
            }
        }
        __syncthreads();
        if (found != ~0ull) break;
        __syncthreads();
    }
    if (threadIdx.x == 0) *S.first = (long long)(found == ~0ull ? G.n : found);
}

// ---------------------------------------------------------------------------------------
// grid = (max chunks over the batch, genomes)
__global__ __launch_bounds__(T) void pack_stats(const PackGenome* __restrict__ tab) {
    __shared__ int sm[NW];
    const PackGenome G = tab[blockIdx.y];
    const size_t c = blockIdx.x;
    if (c >= G.nchunks) return;
    const size_t pos = c * (size_t)kPackChunk + (size_t)threadIdx.x * kPackBytesPerThread;
    const ThreadSpan t = load_span(G.fa, G.n, pos, (size_t)gload8u(Scratch(G).first));
    const Incoming in = propagate(t.kind, false, sm);
    int t0, t1;  // tokens if the CHUNK is entered outside / inside a header
    if (in.determined || t.line_start) {
        t0 = t1 = in.hdr ? t.t1 : t.t0;
    } else {
        t0 = t.t0;
        t1 = t.t1;
    }
    int s0, s1;
    block_incl_sum(t0, sm, &s0);
    block_incl_sum(t1, sm, &s1);
    if (threadIdx.x == 0) {
        const Scratch S(G);
        S.kind_in[c] = in.chunk_kind;
        S.cnt0[c] = s0;
        S.cnt1[c] = s1;
    }
}

// one workgroup of 1024 threads per genome: incoming header bit and token offset of every chunk
__global__ __launch_bounds__(1024) void pack_scan(const PackGenome* __restrict__ tab) {
    __shared__ int smk[16];
    __shared__ long long sms[16];
    const PackGenome G = tab[blockIdx.x];
    const size_t nchunks = G.nchunks;
    const Scratch S(G);
    const TokenStream out = G.out;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    bool carry_hdr = false;  // header bit entering the next block of 1024 chunks
    long long carry_sum = 0;
    for (size_t blk = 0; blk < nchunks; blk += 1024) {
        const size_t c = blk + threadIdx.x;
        const bool live = c < nchunks;
        const int kind = live ? (int)S.kind_in[c] : KIND_ID;
        // incoming header bit of chunk c = value of the nearest constant chunk before it
        const unsigned long long cm = __ballot(kind != KIND_ID), vm = __ballot(kind == KIND_C1);
        const unsigned long long below = cm & ((1ull << lane) - 1ull);
        if (lane == 0) smk[wv] = cm ? (((vm >> (63 - __builtin_clzll(cm))) & 1ull) ? KIND_C1 : KIND_C0) : KIND_ID;
        __syncthreads();
        bool hdr_in = carry_hdr, next_carry = carry_hdr;
        for (int w = 0; w < 16; ++w) {
            const int k = smk[w];
            if (k != KIND_ID) {
                next_carry = (k == KIND_C1);
                if (w < wv) hdr_in = (k == KIND_C1);
            }
        }
        if (below) hdr_in = (vm >> (63 - __builtin_clzll(below))) & 1ull;
        const long long cnt = live ? (hdr_in ? S.cnt1[c] : S.cnt0[c]) : 0;
        // exclusive 64-bit sum
        long long inc = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const long long o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        if (lane == 63) sms[wv] = inc;
        __syncthreads();
        long long pre = 0, tot = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wv) pre += sms[w];
            tot += sms[w];
        }
        if (live) {
            const long long s = carry_sum + pre + inc - cnt;
            S.kind_in[c] = hdr_in ? 1 : 0;
            S.slot[c] = s;
            out.codes[s >> 4] = 0;
            out.bad[s >> 5] = 0;
        }
        carry_hdr = next_carry;
        carry_sum += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const long long total = carry_sum;
        S.slot[nchunks] = total;
        *out.ntok = (unsigned long long)total;
        const long long pad_end = (total + kSegTokens - 1) / kSegTokens * kSegTokens;
        // the word holding `total` is a partial boundary word of the last chunk: zero it,
        // then mark every padding token as BREAK
        out.codes[total >> 4] = 0;
        out.bad[total >> 5] = 0;
        for (long long w = total >> 4; w < (pad_end >> 4); ++w) out.codes[w] = 0;
        for (long long w = total >> 5; w < (pad_end >> 5); ++w)
            out.bad[w] = (w == (total >> 5)) ? (~0u << (total & 31)) : ~0u;
    }
}

__global__ __launch_bounds__(T) void pack_write(const PackGenome* __restrict__ tab) {
    __shared__ int sm[NW];
    // LDS image of the chunk's words, swizzled: a thread's 64 tokens cover ~4 consecutive code words
    // (~2 BREAK words), so in every LDS instruction lane i touches word ~4i+j; storing word w at
    // (w mod 4) * ROW + w / 4 makes those addresses consecutive across lanes instead of 8-way bank conflicts
    constexpr int ROW = kPackChunk / 64 + 4;  // 260
    __shared__ uint32_t lcodes[4 * ROW];  // <= 16384 tokens -> <= 1025 code words (+1 spill)
    __shared__ uint32_t lbad[2 * ROW];
    auto cidx = [](int w) { return (w & 3) * ROW + (w >> 2); };
    auto bidx = [](int w) { return (w & 1) * ROW + (w >> 1); };
    const PackGenome G = tab[blockIdx.y];
    const size_t c = blockIdx.x;
    if (c >= G.nchunks) return;
    const Scratch SC(G);
    const TokenStream out = G.out;
    const size_t pos = c * (size_t)kPackChunk + (size_t)threadIdx.x * kPackBytesPerThread;
    const ThreadSpan t = load_span(G.fa, G.n, pos, (size_t)gload8u(SC.first));
    const Incoming in = propagate(t.kind, SC.kind_in[c] != 0, sm);
    const int cnt = in.hdr ? t.t1 : t.t0;
    int total;
    const int inc = block_incl_sum(cnt, sm, &total);
    if (total == 0) return;
    const long long S = SC.slot[c], E = S + total;  // global token range of this chunk
    const long long cw0 = S >> 4, bw0 = S >> 5;
    const int ncw = (int)(((E - 1) >> 4) - cw0) + 1, nbw = (int)(((E - 1) >> 5) - bw0) + 1;
    for (int i = threadIdx.x; i < 4 * ROW; i += T) lcodes[i] = 0;
    for (int i = threadIdx.x; i < 2 * ROW; i += T) lbad[i] = 0;
    __syncthreads();
    if (cnt) {
        long long tk = S + (inc - cnt);  // global index of this thread's next token
        if (t.plain && !in.hdr) {
            // Every byte but the newlines is a token: squeeze the newline fields out of the 128-bit
            // code string and the 64-bit BREAK string (highest first, so lower positions stay put),
            // then OR the dense strings into the image at the thread's bit offset.
            uint64_t clo = pack_u64(t.bits.codes[1], t.bits.codes[0]), chi = pack_u64(t.bits.codes[3], t.bits.codes[2]);
            uint64_t bad = pack_u64(t.bits.bad[1], t.bits.bad[0]);
            uint64_t nl = pack_u64(t.bits.nl[1], t.bits.nl[0]);
            while (nl) {
                const int q = 63 - __builtin_clzll(nl);
                nl &= ~(1ull << q);
                const uint64_t keep = (1ull << q) - 1ull;
                bad = (bad & keep) | ((bad >> 1) & ~keep);
                if (q < 32) {
                    const uint64_t k2 = (1ull << (2 * q)) - 1ull;
                    clo = (clo & k2) | (((clo >> 2) | (chi << 62)) & ~k2);
                    chi >>= 2;
                } else {
                    const uint64_t k2 = (1ull << (2 * (q - 32))) - 1ull;
                    chi = (chi & k2) | ((chi >> 2) & ~k2);
                }
            }
            // (the shifts fill the vacated top fields with zeros: exactly cnt fields remain)
            const int wi = (int)((tk >> 4) - cw0), sh = (int)(tk & 15) * 2;
            const uint32_t c[5] = {(uint32_t)clo, (uint32_t)(clo >> 32), (uint32_t)chi, (uint32_t)(chi >> 32), 0u};
            uint32_t prev = 0;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const uint32_t v = (uint32_t)((((uint64_t)c[i] << 32) | prev) >> (32 - sh));  // funnel shift left by sh
                prev = c[i];
                if (v) atomicOr(&lcodes[cidx(wi + i)], v);
            }
            if (bad) {
                const int bi = (int)((tk >> 5) - bw0), bs = (int)(tk & 31);
                const uint32_t bb[3] = {(uint32_t)bad, (uint32_t)(bad >> 32), 0u};
                uint32_t pb = 0;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const uint32_t v = (uint32_t)((((uint64_t)bb[i] << 32) | pb) >> (32 - bs));
                    pb = bb[i];
                    if (v) atomicOr(&lbad[bidx(bi + i)], v);
                }
            }
        } else {
            LineState s{in.hdr, t.line_start, false, 0, 0};
#pragma unroll
            for (int j = 0; j < SUB; ++j) {
                uint32_t pc, pb;
                const int n16 = scan16<true>(t.b[j], j + 1 < SUB ? t.b[j + 1 < SUB ? j + 1 : j].at(0) : t.after, s, pc, pb);
                if (n16) {
                    const int wi = (int)((tk >> 4) - cw0), sh = (int)(tk & 15) * 2;
                    if (pc << sh) atomicOr(&lcodes[cidx(wi)], pc << sh);
                    if (sh && (pc >> (32 - sh))) atomicOr(&lcodes[cidx(wi + 1)], pc >> (32 - sh));
                    if (pb) {
                        const int bi = (int)((tk >> 5) - bw0), bs = (int)(tk & 31);
                        atomicOr(&lbad[bidx(bi)], pb << bs);
                        if (bs > 16 && (pb >> (32 - bs))) atomicOr(&lbad[bidx(bi + 1)], pb >> (32 - bs));
                    }
                    tk += n16;
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ncw; i += T) {
        const long long w = cw0 + i;
        const uint32_t v = lcodes[cidx(i)];
        if ((w << 4) >= S && (w << 4) + 16 <= E)
            out.codes[w] = v;
        else if (v)
            atomicOr(&out.codes[w], v);
    }
    for (int i = threadIdx.x; i < nbw; i += T) {
        const long long w = bw0 + i;
        const uint32_t v = lbad[bidx(i)];
        if ((w << 5) >= S && (w << 5) + 32 <= E)
            out.bad[w] = v;
        else if (v)
            atomicOr(&out.bad[w], v);
    }
}

}  // namespace

void launch_pack_batch(const PackGenome* tab_dev, int ngenomes, size_t max_chunks, hipStream_t st) {
    if (ngenomes <= 0) return;
    const dim3 grid((unsigned)max_chunks, (unsigned)ngenomes);
    hipLaunchKernelGGL(pack_first, dim3((unsigned)ngenomes), dim3(1024), 0, st, tab_dev);
    if (max_chunks) hipLaunchKernelGGL(pack_stats, grid, dim3(T), 0, st, tab_dev);
    hipLaunchKernelGGL(pack_scan, dim3((unsigned)ngenomes), dim3(1024), 0, st, tab_dev);
    if (max_chunks) hipLaunchKernelGGL(pack_write, grid, dim3(T), 0, st, tab_dev);
}

}  // namespace dd
