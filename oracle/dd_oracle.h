/*
 * dd_oracle.h -- CPU ORACLE for the DandD delta-sketching hot path.
 *
 * *** TEST INFRASTRUCTURE ONLY. ***  Nothing under dandd_amd/ (the product) may
 * include, link, import or execute this.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, and only as the checker / the timed
 * CPU baseline -- never as the thing shipped.
 *
 * What it restates: the arithmetic that jessicabonnie/dandd obtains by shelling
 * out to `dashing sketch|union|card` and `kmc`/`kmc_tools`
 *   call sites: /root/reference/lib/sketch_classes.py:312,358-365,370-372 (dashing)
 *               /root/reference/lib/sketch_classes.py:395,444-448,453-465 (kmc)
 *               /root/reference/lib/huffman_dandd.py:217                  (GNU parallel over k)
 * The algorithm itself lives in third-party code that is ABSENT from
 * /root/reference: dnbaker/dashing (v1.x, with submodules dnbaker/bonsai and
 * dnbaker/sketch; version UNPINNED -- reference README.md:17 says "latest binary
 * release") and KMC3 (bioconda, unpinned).  This file restates their PUBLISHED
 * algorithms: 2-bit canonical k-mers (bonsai Encoder, unspaced/unwindowed path),
 * Thomas Wang's 64-bit integer mix (sketch::hash::WangHash), the HyperLogLog
 * register rule of sketch::hll::hllbase_t::add, byte-max union, and Ertl's
 * maximum-likelihood estimator (Ertl 2017, Algorithm 8 -- dashing's default
 * ERTL_MLE).  No line of those projects is available here.
 *
 * PARITY UNPINNED: the reference has no tests, no golden .hll files and no
 * expected cardinalities; no dashing binary exists in
 * this image.  The oracle is pinned only by (i) known-answer vectors computed
 * independently in Python (tests/golden/kat_*.json), (ii) the exact counter
 * below (HLL estimate within 3 sigma of the exact distinct count), and (iii)
 * orchestration goldens produced by importing the reference's Python with
 * this oracle behind `dashing`/`parallel` shims (tests/golden/make_golden.py).
 * "Bit-exact vs Dashing" therefore means "bit-exact vs this oracle"; the
 * Dashing claim itself is untested.
 */
#ifndef DD_ORACLE_H
#define DD_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- hash / register rule ---- */
uint64_t orc_wang64(uint64_t key);
/* fold of a 128-bit canonical k-mer (k in 33..64) to the 64-bit hash input.
 * EXTENSION: Dashing stops at k=32 (/root/reference/lib/huffman_dandd.py:109);
 * this repo defines x = lo ^ (hi * 0x9E3779B97F4A7C15) and hashes x with wang64. */
uint64_t orc_fold128(uint64_t hi, uint64_t lo);
/* idx = h >> (64-p); rho = clz64(((h<<1)|1) << (p-1)) + 1  in [1, 64-p+1] */
void orc_idx_rho(uint64_t h, int p, uint32_t *idx, uint8_t *rho);

/* ---- FASTA / FASTQ -> records -> token stream (A.1; policy P10 of POLICIES.md) ----
 * orc_records: the sequence text of every record as klib's kseq.h reads it (everything before the first '>' or '@' is
 * skipped; a line starting with '>' / '@' is a header, one starting with '+' opens FASTQ quality lines; one '\r' in
 * front of a line end is dropped), each record closed by one '\n'.  out: n + 1 bytes, or NULL to count only.
 * orc_tokenize: token values 0..3 = A,C,G,T (case-insensitive); 4 = BREAK (any other sequence byte, and one at the end
 * of every record so windows never span records).  Returns the number of tokens; out may be NULL to count only. */
size_t orc_records(const uint8_t *fa, size_t n, uint8_t *out);
size_t orc_tokenize(const uint8_t *fa, size_t n, uint8_t *out);

/* ---- one `dashing sketch -k K -S p [--no-canon]` job: max-merges into regs[2^p] ---- */
int orc_sketch(const uint8_t *fa, size_t n, int k, int p, int canonical, uint8_t *regs);
/* same result through the generic 128-bit window path (cross-check of the k<=32 fast path) */
int orc_sketch_generic(const uint8_t *fa, size_t n, int k, int p, int canonical, uint8_t *regs);
/* whole k-sweep, one independent pass per k (how DandD drives Dashing):
 * regs is [kmax-kmin+1][2^p], max-merged into. */
int orc_sketch_sweep(const uint8_t *fa, size_t n, int kmin, int kmax, int p, int canonical,
                     uint8_t *regs);

/* ---- `dashing union`: dst[i] = max(dst[i], src[i]) ---- */
void orc_union(uint8_t *dst, const uint8_t *src, size_t m);
/* ---- `dashing card`: 64-bin histogram + Ertl MLE (A.4) ---- */
void orc_hist(const uint8_t *regs, size_t m, uint32_t hist[64]);
double orc_ertl_mle(const uint32_t hist[64], int p);
double orc_card(const uint8_t *regs, int p);

/* ---- KMC stand-in (A.5): exact number of distinct (canonical) k-mers over
 * nbuf FASTA buffers, k in 1..64.  Memory: 16 B per k-mer occurrence. ---- */
int orc_exact_count(const uint8_t *const *fas, const size_t *ns, int nbuf, int k, int canonical,
                    uint64_t *distinct);

/* ---- synthetic FASTA, replayable byte-for-byte on the GPU ----
 * genome gi of a family: ancestor = splitmix64(seed ^ pos) & 3, 1 % substitutions,
 * 0.1 % of positions inside aligned 100-base N runs, 10 % inside aligned 500-base
 * lowercase runs, nrec records with 16-byte headers, 80-column lines. */
size_t orc_synth_size(uint64_t nbases, int nrec);
size_t orc_synth_fasta(uint64_t seed, int gi, uint64_t nbases, int nrec, uint8_t *out);
uint64_t orc_splitmix64(uint64_t x);
/* "realistic" mode: GC 35 %, 30 % repeats (soft-masked), 2 % N, contigs of 2..200 kbp (dd_oracle.c) */
size_t orc_synth_realistic_size(uint64_t seed, uint64_t nbases);
size_t orc_synth_realistic_fasta(uint64_t seed, int gi, uint64_t nbases, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
