/*
 * dd_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see dd_oracle.h).
 * PARITY UNPINNED against Dashing/KMC (neither is present; the reference has
 * no golden vectors) -- pinned by tests/golden KATs and the exact counter.
 *
 * Plain C99 + unsigned __int128, scalar, single-threaded: one pass per
 * (FASTA, k), exactly the granularity at which DandD launches `dashing sketch`
 * (/root/reference/lib/huffman_dandd.py:214-218).
 */
#include "dd_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ============================================================== POLICY BLOCK
 * Every assumption about what `dashing` does that could not be checked against a Dashing binary or its source
 * (RECALL) is ONE definition in this block; oracle/POLICIES.md lists them with the product
 * lines that implement the same rule and the known-answer vectors that would change.  A mismatch found the day a
 * binary is available (bench.py diffs its registers against this oracle, bench.py:dashing_baseline) is a one-line
 * change here plus the matching line of the kernels. */

/* P1  2-bit code table: A/a 0, C/c 1, G/g 2, T/t 3; every other byte is ambiguous and resets the k-mer window */
static const int8_t *code_lut(void) {
    static int8_t lut[256];
    static int init = 0;
    if (!init) {
        memset(lut, 4, sizeof lut);
        lut['A'] = lut['a'] = 0;
        lut['C'] = lut['c'] = 1;
        lut['G'] = lut['g'] = 2;
        lut['T'] = lut['t'] = 3;
        init = 1;
    }
    return lut;
}
/* P3  canonical form: the smaller of the k-mer and its reverse complement as unsigned integers, no XOR mask;
 * P4  the all-T 32-mer (every bit set: bonsai's "ambiguous" marker) is an ordinary k-mer */
static inline uint64_t pol_canonical64(uint64_t fw, uint64_t rc) { return rc < fw ? rc : fw; }
static inline u128 pol_canonical128(u128 fw, u128 rc) { return rc < fw ? rc : fw; }
/* P5  hash of a k-mer (k <= 32): Thomas Wang's 64-bit mix == sketch::hash::WangHash.
 * KATs: wang(0)=0x77cfa1eef01bca90 wang(1)=0x5bca7c69b794f8ce */
uint64_t orc_wang64(uint64_t key) {
    key = (~key) + (key << 21);
    key = key ^ (key >> 24);
    key = (key + (key << 3)) + (key << 8); /* * 265 */
    key = key ^ (key >> 14);
    key = (key + (key << 2)) + (key << 4); /* * 21 */
    key = key ^ (key >> 28);
    key = key + (key << 31);
    return key;
}
/* P5b k in 33..64 has no Dashing analogue (/root/reference/lib/huffman_dandd.py:109): the 128-bit canonical k-mer is
 * folded to 64 bits before the same hash.  Engine-defined, not a RECALL. */
uint64_t orc_fold128(uint64_t hi, uint64_t lo) { return lo ^ (hi * 0x9E3779B97F4A7C15ull); }
static inline uint64_t pol_hash(uint64_t hi, uint64_t lo, int k) { return orc_wang64(k <= 32 ? lo : orc_fold128(hi, lo)); }
/* P6  register index = the TOP p bits of the hash;  P7  rho = leading zeros of the remaining 64 - p bits, counted
 * with a sentinel bit behind them, + 1: in [1, 64 - p + 1]   (sketch::hll::hllbase_t::add) */
static inline uint32_t pol_index(uint64_t h, int p) { return (uint32_t)(h >> (64 - p)); }
static inline uint8_t pol_rho(uint64_t h, int p) { return (uint8_t)(__builtin_clzll(((h << 1) | 1) << (p - 1)) + 1); }
/* P8  Ertl's ML estimator stops at a relative step of 1e-2 / sqrt(m) (dashing's default ERTL_MLE) */
static inline double pol_mle_relerr(uint64_t m) { return 1e-2 / sqrt((double)m); }
/* P10 record rules: kseq's (below, orc_records) */
/* ======================================================== end of POLICY BLOCK */

void orc_idx_rho(uint64_t h, int p, uint32_t *idx, uint8_t *rho) {
    *idx = pol_index(h, p);
    *rho = pol_rho(h, p);
}

/* ----------------------------------------------------------------- records */

/* P10  The sequence text of a FASTA / FASTQ buffer as klib's kseq.h reads it (Dashing parses its inputs with kseq;
 * RECALL of kseq_read):
 *   - everything before the first '>' or '@' (anywhere, not only at a line start) is skipped;
 *   - the rest of that line is the header (name + comment);
 *   - the following lines are sequence until a line STARTS with '>', '@' or '+' (or the input ends); empty lines are
 *     skipped; the line terminator is dropped, and so is ONE '\r' in front of it;
 *   - a line starting with '>' or '@' starts the next record;
 *   - a line starting with '+' makes the record a FASTQ record: the rest of that line is skipped, then whole lines are
 *     read as quality until they hold at least as many characters as the sequence (one line at least); the reader then
 *     looks for the next '>' or '@' ANYWHERE in what follows (kseq's last_char = 0);
 *   - a FASTQ record whose '+' line is cut off by the end of the input, or whose quality text is not exactly as long as
 *     its sequence, is an error (kseq_read returns -2): that record and everything behind it are not read.
 * Output: the sequence bytes of every record verbatim (a '\r' inside a line, digits, anything), each record CLOSED by
 * one '\n' -- a byte that cannot occur in sequence text -- so that the encoder sees one BREAK per record.  `out` needs
 * n + 1 bytes at most (NULL: count only).  Returns the number of bytes written. */
size_t orc_records(const uint8_t *fa, size_t n, uint8_t *out) {
    size_t i = 0, o = 0;
    int have_header = 0; /* kseq's last_char: the header character of the next record has been consumed */
    for (;;) {
        if (!have_header) {
            while (i < n && fa[i] != '>' && fa[i] != '@') ++i;
            if (i >= n) break;
            ++i; /* the header character */
        }
        have_header = 0;
        while (i < n && fa[i] != '\n') ++i; /* name and comment */
        if (i < n) ++i;
        size_t seq_len = 0;
        const size_t o_rec = o; /* where this record's text starts: a truncated FASTQ record is taken back */
        int c = -1; /* the character that ended the sequence part; -1: end of input */
        while (i < n) {
            c = fa[i];
            if (c == '>' || c == '+' || c == '@') break;
            if (c == '\n') { /* empty line */
                ++i;
                c = -1;
                continue;
            }
            size_t e = i;
            while (e < n && fa[e] != '\n') ++e;
            size_t len = e - i;
            /* kseq drops one trailing '\r' of the accumulated sequence (if it is longer than one character) */
            if (len && fa[e - 1] == '\r' && seq_len + len > 1) --len;
            if (out) memcpy(out + o, fa + i, len);
            o += len;
            seq_len += len;
            i = e < n ? e + 1 : e;
            c = -1;
        }
        if (out) out[o] = '\n';
        ++o;
        if (i >= n) break;
        if (c == '>' || c == '@') {
            ++i;
            have_header = 1;
            continue;
        }
        /* c == '+': FASTQ.  Skip the rest of the '+' line, then quality lines until they cover the sequence.
         * kseq_read returns -2 -- and the reader's `while (kseq_read(ks) >= 0)` loop drops THIS record and everything
         * behind it -- when the input ends inside the '+' line ("no quality string") or when the quality text is not
         * exactly as long as the sequence; its loop tests the length AFTER reading a line, so one quality line is read
         * even for an empty sequence. */
        while (i < n && fa[i] != '\n') ++i;
        if (i >= n) return o_rec;
        ++i;
        size_t qual_len = 0;
        do {
            if (i >= n) break; /* ks_getuntil2 at the end of the input: nothing read */
            size_t e = i;
            while (e < n && fa[e] != '\n') ++e;
            size_t len = e - i;
            if (len && fa[e - 1] == '\r' && qual_len + len > 1) --len;
            qual_len += len;
            i = e < n ? e + 1 : e;
        } while (qual_len < seq_len);
        if (qual_len != seq_len) return o_rec;
    }
    return o;
}

/* ------------------------------------------------------------- tokenizer */

/* Token values 0..3 = A,C,G,T; 4 = BREAK: an ambiguous byte of a record's sequence, or the end of a record (one per
 * record, so windows never span records). */
size_t orc_tokenize(const uint8_t *fa, size_t n, uint8_t *out) {
    const int8_t *lut = code_lut();
    uint8_t *seq = (uint8_t *)malloc(n + 1);
    if (!seq) abort();
    const size_t m = orc_records(fa, n, seq);
    if (out)
        for (size_t i = 0; i < m; ++i) out[i] = (uint8_t)lut[seq[i]];  /* ('\n', the record mark, is ambiguous like any other) */
    free(seq);
    return m;
}

/* ---------------------------------------------------------------- sketch */

/* k-mer stream of bonsai's unspaced/unwindowed encoder (A.1) over kseq's records: rolling forward window, rolling
 * reverse-complement window, canonical form by policy P3. */
typedef void (*kmer_fn)(void *ctx, uint64_t hi, uint64_t lo);

static void for_each_kmer(const uint8_t *fa, size_t n, int k, int canonical, kmer_fn fn, void *ctx) {
    const int8_t *lut = code_lut();
    const u128 mask = (k == 64) ? ~(u128)0 : (((u128)1 << (2 * k)) - 1);
    uint8_t *seq = (uint8_t *)malloc(n + 1);
    if (!seq) abort();
    const size_t m = orc_records(fa, n, seq);
    u128 fw = 0, rc = 0;
    int run = 0;
    for (size_t i = 0; i < m; ++i) {
        int c = lut[seq[i]];
        if (c > 3) {
            run = 0;
            continue;
        }
        fw = ((fw << 2) | (u128)c) & mask;
        rc = (rc >> 2) | ((u128)(3 - c) << (2 * (k - 1)));
        if (++run >= k) {
            u128 x = canonical ? pol_canonical128(fw, rc) : fw;
            fn(ctx, (uint64_t)(x >> 64), (uint64_t)x);
        }
    }
    free(seq);
}

struct hll_ctx {
    uint8_t *regs;
    int p, k;
};

static void hll_add(void *vctx, uint64_t hi, uint64_t lo) {
    struct hll_ctx *c = (struct hll_ctx *)vctx;
    const uint64_t h = pol_hash(hi, lo, c->k);
    const uint32_t idx = pol_index(h, c->p);
    const uint8_t rho = pol_rho(h, c->p);
    if (rho > c->regs[idx]) c->regs[idx] = rho;
}

/* Same stream as for_each_kmer, k <= 32, 64-bit windows, everything inlined: this is the loop a tuned single-threaded
 * `dashing sketch` job spends its time in (kseq copies a record's sequence out, the encoder runs over the copy), so it is
 * also what bench.py times as the CPU baseline. */
static void sketch64(const uint8_t *fa, size_t n, int k, int p, int canonical, uint8_t *regs) {
    const int8_t *lut = code_lut();
    const uint64_t mask = (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1);
    const int rsh = 2 * (k - 1);
    uint8_t *seq = (uint8_t *)malloc(n + 1);
    if (!seq) abort();
    const size_t m = orc_records(fa, n, seq);
    uint64_t fw = 0, rc = 0;
    int run = 0;
    for (size_t i = 0; i < m; ++i) {
        int c = lut[seq[i]];
        if (c > 3) {
            run = 0;
            continue;
        }
        fw = ((fw << 2) | (uint64_t)c) & mask;
        rc = (rc >> 2) | ((uint64_t)(3 - c) << rsh);
        if (++run >= k) {
            const uint64_t h = pol_hash(0, canonical ? pol_canonical64(fw, rc) : fw, k);
            const uint32_t idx = pol_index(h, p);
            const uint8_t rho = pol_rho(h, p);
            if (rho > regs[idx]) regs[idx] = rho;
        }
    }
    free(seq);
}

int orc_sketch(const uint8_t *fa, size_t n, int k, int p, int canonical, uint8_t *regs) {
    if (k < 1 || k > 64 || p < 4 || p > 24) return -1;
    if (k <= 32) {
        sketch64(fa, n, k, p, canonical, regs);
        return 0;
    }
    struct hll_ctx c = {regs, p, k};
    for_each_kmer(fa, n, k, canonical, hll_add, &c);
    return 0;
}

/* always the generic 128-bit stream (used by tests to cross-check sketch64) */
int orc_sketch_generic(const uint8_t *fa, size_t n, int k, int p, int canonical, uint8_t *regs) {
    if (k < 1 || k > 64 || p < 4 || p > 24) return -1;
    struct hll_ctx c = {regs, p, k};
    for_each_kmer(fa, n, k, canonical, hll_add, &c);
    return 0;
}

int orc_sketch_sweep(const uint8_t *fa, size_t n, int kmin, int kmax, int p, int canonical,
                     uint8_t *regs) {
    if (kmin < 1 || kmax > 64 || kmin > kmax) return -1;
    for (int k = kmin; k <= kmax; ++k) {
        int rc = orc_sketch(fa, n, k, p, canonical, regs + ((size_t)(k - kmin) << p));
        if (rc) return rc;
    }
    return 0;
}

/* ----------------------------------------------------------- union / card */

void orc_union(uint8_t *dst, const uint8_t *src, size_t m) {
    for (size_t i = 0; i < m; ++i)
        if (src[i] > dst[i]) dst[i] = src[i];
}

void orc_hist(const uint8_t *regs, size_t m, uint32_t hist[64]) {
    memset(hist, 0, 64 * sizeof(uint32_t));
    for (size_t i = 0; i < m; ++i) hist[regs[i] & 63]++;
}

/* Ertl 2017, "New cardinality estimation algorithms for HyperLogLog sketches",
 * Algorithm 8 (ML estimator, secant iteration), relative tolerance 1e-2/sqrt(m)
 * -- dashing's default `ERTL_MLE`.  IEEE double throughout;
 * build with -ffp-contract=off so the product's copy agrees bit for bit. */
double orc_ertl_mle(const uint32_t c[64], int p) {
    const int q = 64 - p;
    const uint64_t m = 1ull << p;
    if (c[q + 1] == m) return INFINITY;
    int kmin, kmax;
    for (kmin = 0; c[kmin] == 0; ++kmin) {}
    int kminp = kmin > 1 ? kmin : 1;
    for (kmax = q + 1; kmax && c[kmax] == 0; --kmax) {}
    int kmaxp = kmax < q ? kmax : q;
    double z = 0.0;
    for (int k = kmaxp; k >= kminp; --k) z = 0.5 * z + (double)c[k];
    z = ldexp(z, -kminp);
    double cprime = (double)c[q + 1];
    if (q >= 1) cprime += (double)c[kmaxp];
    double a = z + (double)c[0];
    double mprime = (double)(m - c[0]);
    double b = z + ldexp((double)c[q + 1], -q);
    double x = (b <= 1.5 * a) ? mprime / (0.5 * b + a) : (mprime / b) * log1p(b / a);
    double dx = x, gprev = 0.0;
    const double relerr = pol_mle_relerr(m);
    while (dx > x * relerr) {
        int kappam1;
        frexp(x, &kappam1);
        int sh = (kmaxp + 1 > kappam1 + 2) ? kmaxp + 1 : kappam1 + 2;
        double xp = ldexp(x, -sh);
        double xp2 = xp * xp;
        double h = xp - xp2 / 3.0 + (xp2 * xp2) * (1.0 / 45.0 - xp2 / 472.5);
        for (int k = kappam1; k >= kmaxp; --k) {
            double hp = 1.0 - h;
            h = (xp + h * hp) / (xp + hp);
            xp += xp;
        }
        double g = cprime * h;
        for (int k = kmaxp - 1; k >= kminp; --k) {
            double hp = 1.0 - h;
            h = (xp + h * hp) / (xp + hp);
            xp += xp;
            g += (double)c[k] * h;
        }
        g += x * a;
        if (gprev < g && g <= mprime)
            dx *= (g - mprime) / (gprev - g);
        else
            dx = 0.0;
        x += dx;
        gprev = g;
    }
    return x * (double)m;
}

double orc_card(const uint8_t *regs, int p) {
    uint32_t hist[64];
    orc_hist(regs, (size_t)1 << p, hist);
    return orc_ertl_mle(hist, p);
}

/* ----------------------------------------------------------- exact count */

struct vec128 {
    u128 *v;
    size_t n, cap;
};

static void vec_push(void *vctx, uint64_t hi, uint64_t lo) {
    struct vec128 *s = (struct vec128 *)vctx;
    if (s->n == s->cap) {
        s->cap = s->cap ? s->cap * 2 : (1u << 16);
        s->v = (u128 *)realloc(s->v, s->cap * sizeof(u128));
        if (!s->v) abort();
    }
    s->v[s->n++] = ((u128)hi << 64) | lo;
}

static int cmp128(const void *a, const void *b) {
    u128 x = *(const u128 *)a, y = *(const u128 *)b;
    return x < y ? -1 : (x > y);
}

/* `kmc -ci1 -cs2 -k K [-b]` + `kmc_tools complex (+)` + `kmc_tools info` (A.5):
 * number of distinct (canonical) k-mers over all inputs. */
int orc_exact_count(const uint8_t *const *fas, const size_t *ns, int nbuf, int k, int canonical,
                    uint64_t *distinct) {
    if (k < 1 || k > 64) return -1;
    struct vec128 s = {0, 0, 0};
    for (int i = 0; i < nbuf; ++i) for_each_kmer(fas[i], ns[i], k, canonical, vec_push, &s);
    if (s.n) qsort(s.v, s.n, sizeof(u128), cmp128);  /* (an input without a single k-mer leaves s.v NULL: found by the UBSan leg) */
    uint64_t d = 0;
    for (size_t i = 0; i < s.n; ++i)
        if (i == 0 || s.v[i] != s.v[i - 1]) ++d;
    free(s.v);
    *distinct = d;
    return 0;
}

/* -------------------------------------------------------- synthetic FASTA */

uint64_t orc_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

#define SYN_HDR 16
#define SYN_LINE 80

static uint64_t rec_len(uint64_t nbases, int nrec, int r) {
    uint64_t per = nbases / (uint64_t)nrec;
    return (r == nrec - 1) ? nbases - per * (uint64_t)(nrec - 1) : per;
}

size_t orc_synth_size(uint64_t nbases, int nrec) {
    size_t tot = 0;
    for (int r = 0; r < nrec; ++r) {
        uint64_t L = rec_len(nbases, nrec, r);
        tot += SYN_HDR + L + (L + SYN_LINE - 1) / SYN_LINE;
    }
    return tot;
}

static uint8_t synth_base(uint64_t seed, uint64_t seed_g, uint64_t pos) {
    static const char up[4] = {'A', 'C', 'G', 'T'};
    uint32_t b = (uint32_t)(orc_splitmix64(seed ^ pos) & 3);
    uint64_t r = orc_splitmix64(seed_g ^ pos);
    if (r % 100 == 0) b = (b + 1 + (uint32_t)((r >> 32) % 3)) & 3;
    uint8_t ch = (uint8_t)up[b];
    if (orc_splitmix64(seed_g ^ 0x4E4E4E4E00000000ull ^ (pos / 100)) % 1000 == 0) ch = 'N';
    if (orc_splitmix64(seed_g ^ 0x6C6C6C6C00000000ull ^ (pos / 500)) % 10 == 0) ch |= 0x20;
    return ch;
}

size_t orc_synth_fasta(uint64_t seed, int gi, uint64_t nbases, int nrec, uint8_t *out) {
    static const char hexd[] = "0123456789abcdef";
    const uint64_t seed_g = orc_splitmix64(seed + (uint64_t)gi + 1);
    size_t o = 0;
    uint64_t pos = 0;
    for (int r = 0; r < nrec; ++r) {
        uint64_t L = rec_len(nbases, nrec, r);
        /* ">gGGGG.rRRRR   \n" */
        uint8_t *h = out + o;
        h[0] = '>';
        h[1] = 'g';
        for (int d = 0; d < 4; ++d) h[2 + d] = (uint8_t)hexd[(gi >> (12 - 4 * d)) & 15];
        h[6] = '.';
        h[7] = 'r';
        for (int d = 0; d < 4; ++d) h[8 + d] = (uint8_t)hexd[(r >> (12 - 4 * d)) & 15];
        h[12] = h[13] = h[14] = ' ';
        h[15] = '\n';
        o += SYN_HDR;
        for (uint64_t j = 0; j < L; ++j) {
            out[o++] = synth_base(seed, seed_g, pos++);
            if (j % SYN_LINE == SYN_LINE - 1 || j == L - 1) out[o++] = '\n';
        }
    }
    return o;
}

/* ---- "realistic" synthetic genomes: the i.i.d. uniform generator above is the BEST case of
 * the k <= 9 "set complete" early exit and of every hash-spread assumption.  This one has what assemblies have:
 * GC 35 %, 20 % interspersed repeats (copies of 64 elements of 300..6000 bases), 10 % tandem repeats (units of 2..60
 * bases, 512-base blocks), soft-masked (lowercase) repeats, 2 % N in 100-base runs, contigs of 2..200 kbp, and 1 % per-base
 * divergence between the genomes of one seed.  Counter-based like the other: base g of genome gi depends on (seed, gi, g)
 * only.  dandd_amd/csrc/dd_synth.hip generates the same bytes on the device (tests/test_gpu_parity.py pins the two). */
static uint64_t real_contig_len(uint64_t seed, uint64_t idx) {
    const uint64_t h = orc_splitmix64(seed ^ 0xC047160000000000ull ^ idx);
    const uint64_t base = 2000ull << ((h >> 40) % 7);
    const uint64_t len = base + h % base;
    return len < 200000 ? len : 200000;
}

static uint8_t real_base(uint64_t seed, uint64_t seed_g, uint64_t g) {
    static const char up[4] = {'A', 'C', 'G', 'T'};
    const uint64_t blk = g >> 9;
    const uint64_t hb = orc_splitmix64(seed ^ 0x5EED5EED00000000ull ^ blk);
    const unsigned kind = (unsigned)(hb % 100);
    uint64_t r;
    if (kind < 20) {
        const uint64_t e = (hb >> 8) & 63;
        const uint64_t elen = 300 + orc_splitmix64(seed ^ 0xE1E100000000ull ^ e) % 5700;
        const uint64_t off = ((hb >> 16) % elen + (g & 511)) % elen;
        r = orc_splitmix64(seed ^ ((0xABCD0000ull + e) << 32) ^ off);
    } else if (kind < 30) {
        const uint64_t u = 2 + (hb >> 8) % 59;
        r = orc_splitmix64(seed ^ 0x7A7A000000000000ull ^ (blk << 8) ^ ((g & 511) % u));
    } else {
        r = orc_splitmix64(seed ^ g);
    }
    const unsigned hi = (unsigned)((r >> 32) & 1);
    unsigned b = (r % 100) < 35 ? 1 + hi : 3 * hi;
    const uint64_t rg = orc_splitmix64(seed_g ^ g);
    if (rg % 100 == 0) b = (b + 1 + (unsigned)((rg >> 32) % 3)) & 3;
    uint8_t ch = (uint8_t)up[b];
    if (orc_splitmix64(seed_g ^ 0x4E4E4E4E00000000ull ^ (g / 100)) % 50 == 0) ch = 'N';
    if (kind < 30) ch |= 0x20;
    return ch;
}

size_t orc_synth_realistic_size(uint64_t seed, uint64_t nbases) {
    size_t tot = 0;
    uint64_t done = 0;
    for (uint64_t c = 0; done < nbases; ++c) {
        uint64_t L = real_contig_len(seed, c);
        if (L > nbases - done) L = nbases - done;
        tot += SYN_HDR + L + (L + SYN_LINE - 1) / SYN_LINE;
        done += L;
    }
    return tot;
}

size_t orc_synth_realistic_fasta(uint64_t seed, int gi, uint64_t nbases, uint8_t *out) {
    static const char hexd[] = "0123456789abcdef";
    const uint64_t seed_g = orc_splitmix64(seed + (uint64_t)gi + 1);
    size_t o = 0;
    uint64_t pos = 0;
    for (uint64_t c = 0; pos < nbases; ++c) {
        uint64_t L = real_contig_len(seed, c);
        if (L > nbases - pos) L = nbases - pos;
        uint8_t *h = out + o;
        h[0] = '>';
        h[1] = 'g';
        for (int d = 0; d < 4; ++d) h[2 + d] = (uint8_t)hexd[(gi >> (12 - 4 * d)) & 15];
        h[6] = '.';
        h[7] = 'r';
        for (int d = 0; d < 4; ++d) h[8 + d] = (uint8_t)hexd[(c >> (12 - 4 * d)) & 15];
        h[12] = h[13] = h[14] = ' ';
        h[15] = '\n';
        o += SYN_HDR;
        for (uint64_t j = 0; j < L; ++j) {
            out[o++] = real_base(seed, seed_g, pos++);
            if (j % SYN_LINE == SYN_LINE - 1 || j == L - 1) out[o++] = '\n';
        }
    }
    return o;
}
