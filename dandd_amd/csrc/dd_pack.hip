// dd_pack.hip -- K0: FASTA bytes in HBM -> 2-bit token stream (+ 1-bit BREAK mask).
//
// First half of what one `dashing sketch` process does before hashing (kseq record parsing +
// bonsai's 2-bit encoder; command built at /root/reference/lib/sketch_classes.py:351-366):
// headers and newlines are dropped, A/C/G/T (either case) become 0..3, every other byte and
// every record boundary becomes a BREAK that resets the k-mer windows.  The oracle's
// statement of the same rules is oracle/dd_oracle.c:orc_tokenize.
//
// HBM-bound: reads each FASTA byte twice (stats pass + write pass; the second pass of a
// <=256 MiB genome is served by the Infinity Cache) and writes 3 bits per token.  Three launches:
//   pack_stats  : per 4 KiB chunk -> last newline position, token counts under both
//                 possible incoming line states (inside a header line / inside sequence)
//   pack_scan   : one workgroup; running max of newline positions + exclusive sum of token
//                 counts over chunks (decides each chunk's incoming state by looking at the
//                 byte after the last newline before it)
//   pack_write  : per chunk -> tokens staged in LDS, packed 16/32 per word, partial boundary
//                 words merged with atomicOr (pack_scan zeroed them)
#include "dd_common.h"
#include "dd_kernels.h"

namespace dd {
namespace {

constexpr int T = kPackThreads;

DD_D uint32_t base_code(uint32_t c) {
    uint32_t x = c | 0x20u;
    return x == 'a' ? 0u : x == 'c' ? 1u : x == 'g' ? 2u : x == 't' ? 3u : 4u;
}

struct Bytes16 {
    uint32_t w[4];
    DD_D uint32_t at(int i) const { return (w[i >> 2] >> ((i & 3) * 8)) & 0xFFu; }
};

// 16 bytes of this thread; bytes at or beyond n read as '\r' (emits nothing, changes nothing)
DD_D Bytes16 load16(const uint8_t* fa, size_t n, size_t pos) {
    Bytes16 b;
    if (pos + 16 <= n) {
        uint4 v = *reinterpret_cast<const uint4*>(fa + pos);
        b.w[0] = v.x; b.w[1] = v.y; b.w[2] = v.z; b.w[3] = v.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t w = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                size_t p = pos + 4 * j + i;
                uint32_t c = p < n ? fa[p] : (uint32_t)'\r';
                w |= c << (8 * i);
            }
            b.w[j] = w;
        }
    }
    return b;
}

// Line state machine over the thread's 16 bytes (same rules as orc_tokenize).
//   prev_nl : the byte before this thread's first byte is '\n' (or there is none)
//   hdr_in  : this thread starts inside a header line (ignored when prev_nl)
template <bool EMIT>
DD_D int scan16(const Bytes16& b, bool prev_nl, bool hdr_in, uint8_t* out) {
    int cnt = 0;
    bool hdr = hdr_in, ls = prev_nl;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        uint32_t c = b.at(i);
        if (ls) hdr = (c == '>');
        if (c == '\n') {
            if (hdr) {
                if (EMIT) out[cnt] = 4;
                ++cnt;
            }
            hdr = false;
            ls = true;
        } else {
            ls = false;
            if (!(hdr || c == '\r')) {
                if (EMIT) out[cnt] = (uint8_t)base_code(c);
                ++cnt;
            }
        }
    }
    return cnt;
}

// Same state machine, but the thread's tokens are assembled in registers: 2-bit codes in `codes`
// (token j at bits 2j..2j+1) and BREAK flags in `bad` (bit j).  Returns the token count (<= 16).
DD_D int scan16_pack(const Bytes16& b, bool prev_nl, bool hdr_in, uint32_t& codes, uint32_t& bad) {
    int cnt = 0;
    bool hdr = hdr_in, ls = prev_nl;
    codes = 0;
    bad = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        uint32_t c = b.at(i);
        if (ls) hdr = (c == '>');
        if (c == '\n') {
            if (hdr) {
                bad |= 1u << cnt;
                ++cnt;
            }
            hdr = false;
            ls = true;
        } else {
            ls = false;
            if (!(hdr || c == '\r')) {
                const uint32_t code = base_code(c);
                codes |= (code & 3u) << (2 * cnt);
                bad |= (code >> 2) << cnt;
                ++cnt;
            }
        }
    }
    return cnt;
}

DD_D long long last_newline(const Bytes16& b, size_t pos) {
    long long r = -1;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (b.at(i) == '\n') r = (long long)(pos + i);
    return r;
}

// ---- workgroup scans over T=256 threads (4 waves) ------------------------------------
template <typename V, typename Op>
DD_D V wave_incl(V v, Op op) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        V o = __shfl_up(v, d);
        if (lane >= d) v = op(v, o);
    }
    return v;
}

struct MaxLL { DD_D long long operator()(long long a, long long b) const { return a > b ? a : b; } };
struct SumLL { DD_D long long operator()(long long a, long long b) const { return a + b; } };

// inclusive scan across the block; *total receives the block-wide reduction.
// sm must hold blockDim.x/64 elements.  ident is the identity of op.
template <typename Op>
DD_D long long block_incl(long long v, Op op, long long ident, long long* sm, long long* total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    long long inc = wave_incl(v, op);
    __syncthreads();
    if (lane == 63) sm[wv] = inc;
    __syncthreads();
    long long pre = ident, tot = ident;
    for (int i = 0; i < nw; ++i) {
        long long x = sm[i];
        if (i < wv) pre = op(pre, x);
        tot = op(tot, x);
    }
    *total = tot;
    return op(pre, inc);
}

// Is position `pos` (not at a line start) inside a header line, given N = position of the last
// '\n' before pos (or -1)?  The line starts at N+1; it is a header iff it starts with '>'.
DD_D bool line_is_header(const uint8_t* fa, long long N) { return fa[N + 1] == '>'; }

// ---------------------------------------------------------------------------------------
// scratch layout of one genome: four arrays of (nchunks + 1) int64
struct Scratch {
    long long *lastnl, *cntH, *cntS, *slot;
    DD_D explicit Scratch(const PackGenome& g)
        : lastnl(g.scratch), cntH(g.scratch + (g.nchunks + 1)), cntS(g.scratch + 2 * (g.nchunks + 1)),
          slot(g.scratch + 3 * (g.nchunks + 1)) {}
};

// grid = (max chunks over the batch, genomes)
__global__ __launch_bounds__(T) void pack_stats(const PackGenome* __restrict__ tab) {
    __shared__ long long sm[T / 64];
    const PackGenome G = tab[blockIdx.y];
    const size_t c = blockIdx.x;
    if (c >= G.nchunks) return;
    const uint8_t* __restrict__ fa = G.fa;
    const size_t n = G.n;
    const Scratch S(G);
    long long* lastnl = S.lastnl;
    long long* cntH = S.cntH;
    long long* cntS = S.cntS;
    const size_t pos = c * (size_t)kPackChunk + (size_t)threadIdx.x * 16;
    Bytes16 b = load16(fa, n, pos);
    const bool prev_nl = (pos == 0) || (pos - 1 < n ? fa[pos - 1] == '\n' : false);
    long long ln = last_newline(b, pos);
    long long tot;
    long long inc = block_incl(ln, MaxLL(), -1, sm, &tot);
    long long before = __shfl_up(inc, 1);  // exclusive: previous thread's inclusive value
    if ((threadIdx.x & 63) == 0) before = -1;
    // cross-wave part of the exclusive value
    {
        long long pre = -1;
        for (int i = 0; i < (int)(threadIdx.x >> 6); ++i) pre = pre > sm[i] ? pre : sm[i];
        before = before > pre ? before : pre;
    }
    int tH, tS;
    if (pos >= n) {  // tail of the last chunk: nothing to tokenise
        tH = tS = 0;
    } else if (prev_nl) {
        tH = tS = scan16<false>(b, true, false, nullptr);
    } else if (before >= 0) {  // a newline earlier in this chunk decides the state
        bool h = line_is_header(fa, before);
        tH = tS = scan16<false>(b, false, h, nullptr);
    } else {  // depends on the chunk's incoming state
        tH = scan16<false>(b, false, true, nullptr);
        tS = scan16<false>(b, false, false, nullptr);
    }
    long long sH, sS;
    block_incl((long long)tH, SumLL(), 0, sm, &sH);
    block_incl((long long)tS, SumLL(), 0, sm, &sS);
    if (threadIdx.x == 0) {
        lastnl[c] = tot;
        cntH[c] = sH;
        cntS[c] = sS;
    }
}

// one workgroup of 1024 threads per genome
__global__ __launch_bounds__(1024) void pack_scan(const PackGenome* __restrict__ tab) {
    __shared__ long long sm[16];
    const PackGenome G = tab[blockIdx.x];
    const uint8_t* __restrict__ fa = G.fa;
    const size_t nchunks = G.nchunks;
    const Scratch SC(G);
    long long* lastnl_Nin = SC.lastnl;
    const long long* cntH = SC.cntH;
    const long long* cntS = SC.cntS;
    long long* slot_base = SC.slot;
    const TokenStream out = G.out;
    long long carryN = -1, carryS = 0;
    for (size_t blk = 0; blk < nchunks; blk += 1024) {
        const size_t c = blk + threadIdx.x;
        const bool live = c < nchunks;
        long long ln = live ? lastnl_Nin[c] : -1;
        long long totN;
        long long incN = block_incl(ln, MaxLL(), -1, sm, &totN);
        // exclusive = max over threads before me (and carry)
        __syncthreads();
        __shared__ long long tmp[1024];
        tmp[threadIdx.x] = incN;
        __syncthreads();
        long long Nin = threadIdx.x ? tmp[threadIdx.x - 1] : -1;
        Nin = Nin > carryN ? Nin : carryN;
        long long cnt = 0;
        if (live) {
            const size_t pos = c * (size_t)kPackChunk;
            bool hdr = false;
            if (pos > 0 && fa[pos - 1] != '\n') hdr = line_is_header(fa, Nin);
            cnt = hdr ? cntH[c] : cntS[c];
            lastnl_Nin[c] = Nin;
        }
        long long totS;
        long long incS = block_incl(cnt, SumLL(), 0, sm, &totS);
        if (live) {
            long long S = carryS + incS - cnt;
            slot_base[c] = S;
            out.codes[S >> 4] = 0;
            out.bad[S >> 5] = 0;
        }
        carryN = carryN > totN ? carryN : totN;
        carryS += totS;
        __syncthreads();
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long total = carryS;
        slot_base[nchunks] = total;
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
    __shared__ long long sm[T / 64];
    __shared__ uint32_t lcodes[kPackChunk / 16 + 2];  // <= 4096 tokens -> <= 257 code words (+1 spill)
    __shared__ uint32_t lbad[kPackChunk / 32 + 2];
    const PackGenome G = tab[blockIdx.y];
    const size_t c = blockIdx.x;
    if (c >= G.nchunks) return;
    const uint8_t* __restrict__ fa = G.fa;
    const size_t n = G.n;
    const Scratch SC(G);
    const long long* Nin = SC.lastnl;
    const long long* slot_base = SC.slot;
    const TokenStream out = G.out;
    const size_t pos = c * (size_t)kPackChunk + (size_t)threadIdx.x * 16;
    Bytes16 b = load16(fa, n, pos);
    const bool prev_nl = (pos == 0) || (pos - 1 < n ? fa[pos - 1] == '\n' : false);
    long long ln = last_newline(b, pos);
    long long tot;
    long long inc = block_incl(ln, MaxLL(), -1, sm, &tot);
    long long before = __shfl_up(inc, 1);
    if ((threadIdx.x & 63) == 0) before = -1;
    {
        long long pre = -1;
        for (int i = 0; i < (int)(threadIdx.x >> 6); ++i) pre = pre > sm[i] ? pre : sm[i];
        before = before > pre ? before : pre;
    }
    if (before < 0) before = Nin[c];
    bool h = false;
    if (!prev_nl && pos < n) h = line_is_header(fa, before);
    uint32_t my_codes, my_bad;
    const int cnt = scan16_pack(b, prev_nl, h, my_codes, my_bad);
    long long total;
    const long long incS = block_incl((long long)cnt, SumLL(), 0, sm, &total);
    const long long S = slot_base[c], E = S + total;  // global token range of this chunk
    if (total == 0) return;

    // The chunk's words are built in LDS at their final bit positions: a thread ORs its <=16
    // tokens (one 32-bit code piece, one 16-bit BREAK piece, each possibly straddling two words),
    // then the words are copied out; only the chunk's first and last word can be shared with a
    // neighbouring chunk and go out through atomicOr (pack_scan zeroed them).
    const long long cw0 = S >> 4, bw0 = S >> 5;
    const int ncw = (int)(((E - 1) >> 4) - cw0) + 1, nbw = (int)(((E - 1) >> 5) - bw0) + 1;
    for (int i = threadIdx.x; i < ncw + 1; i += T) lcodes[i] = 0;
    for (int i = threadIdx.x; i < nbw + 1; i += T) lbad[i] = 0;
    __syncthreads();
    if (cnt) {
        const long long t0 = S + (incS - cnt);  // global index of this thread's first token
        const int wi = (int)((t0 >> 4) - cw0), sh = (int)(t0 & 15) * 2;
        atomicOr(&lcodes[wi], my_codes << sh);
        if (sh && (my_codes >> (32 - sh))) atomicOr(&lcodes[wi + 1], my_codes >> (32 - sh));
        const int bi = (int)((t0 >> 5) - bw0), bs = (int)(t0 & 31);
        if (my_bad) {
            atomicOr(&lbad[bi], my_bad << bs);
            if (bs > 16 && (my_bad >> (32 - bs))) atomicOr(&lbad[bi + 1], my_bad >> (32 - bs));
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ncw; i += T) {
        const long long w = cw0 + i;
        const uint32_t v = lcodes[i];
        if ((w << 4) >= S && (w << 4) + 16 <= E)
            out.codes[w] = v;
        else if (v)
            atomicOr(&out.codes[w], v);
    }
    for (int i = threadIdx.x; i < nbw; i += T) {
        const long long w = bw0 + i;
        const uint32_t v = lbad[i];
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
    if (max_chunks) hipLaunchKernelGGL(pack_stats, grid, dim3(T), 0, st, tab_dev);
    hipLaunchKernelGGL(pack_scan, dim3((unsigned)ngenomes), dim3(1024), 0, st, tab_dev);
    if (max_chunks) hipLaunchKernelGGL(pack_write, grid, dim3(T), 0, st, tab_dev);
}

}  // namespace dd
