/*
 * dandd_hip.h -- C ABI of libdandd_hip.so: the MI355X (gfx950) delta-sketching engine
 * that replaces DandD's shell-outs to `dashing sketch|union|card` and GNU `parallel`.
 *
 * Boundary being replaced (reference = jessicabonnie/dandd, paths under /root/reference):
 *   the seven subprocess call sites lib/sketch_classes.py:190,198 (leaf sketch),
 *   :221,229 (union), :268,274 (card) and lib/huffman_dandd.py:233 (the
 *   `parallel -j 95% '<cmd {}>' ::: k...` k-batch).  The reference has no FFI; its
 *   "plugin API" is a CLI + filesystem + stdout contract.  Each
 *   entry point below names the command line(s) it stands in for.  INTEGRATION.md
 *   shows the ctypes stub a DandD maintainer would add to lib/sketch_classes.py.
 *
 * Conventions: every function returns 0 on success or a negative DD_E* code; the
 * message is available from dd_last_error() (thread-local).  The caller owns every
 * buffer.  `dd_ctx` is opaque, bound to one GPU, and not thread-safe (one context
 * per thread/device).  There is NO CPU fallback: dd_create fails loudly when no
 * gfx950 device is usable.  Pointers named *_dev are device (HBM) addresses on the
 * context's GPU; all others are host addresses.  No torch types cross this boundary.
 *
 * Register layout: one byte per HyperLogLog register, m = 2^log2m registers per
 * sketch, `[K][m]` row-major for a k-sweep kmin..kmax (K = kmax-kmin+1).
 */
#ifndef DANDD_HIP_H
#define DANDD_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DD_ABI_VERSION 4

#define DD_OK 0
#define DD_EINVAL (-1)   /* bad argument */
#define DD_ENODEV (-2)   /* no usable gfx950 device / HIP runtime error at create */
#define DD_EHIP (-3)     /* HIP runtime error during a call */
#define DD_EIO (-4)      /* file could not be read */
#define DD_ENOMEM (-5)

typedef struct dd_ctx dd_ctx;

int dd_abi_version(void);
const char *dd_last_error(void);

/* Replaces the per-process configuration of `dashing sketch -S <log2m> [--no-canon]`
 * (lib/sketch_classes.py:358-365; canon_command :20-29).  log2m in 4..20, k in 1..64
 * (Dashing itself stops at k=32, lib/huffman_dandd.py:109; 33..64 is this engine's
 * documented extension).  Returns NULL on failure. */
dd_ctx *dd_create(int device, int log2m, int canonical);
void dd_destroy(dd_ctx *);
/* Run all of this context's work on the given hipStream_t (NULL = default stream). */
int dd_set_stream(dd_ctx *, void *hip_stream);
int dd_synchronize(dd_ctx *);

/* ---- leaf sketch over a whole k-sweep ------------------------------------------
 * Replaces   parallel -j 95% ' dashing sketch -k{} -S <R> --prefix <dir> <fasta> ' ::: kmin..kmax
 * (lib/huffman_dandd.py:214-218 + lib/sketch_classes.py:351-366): ONE pass over the
 * FASTA bytes instead of one process per k.  regs[K][m] is overwritten. */
int dd_sketch_buffer(dd_ctx *, const uint8_t *fasta, size_t nbytes, int kmin, int kmax,
                     uint8_t *regs);
/* path may be plain or gzip-compressed (DandD's inputs are .fa/.fasta/.fna[.gz],
 * lib/species_specifics.py:93).  Files under 4 MiB: one read, .gz inflated on the host with zlib;
 * larger ones take dd_sketch_files' pipeline (below) as a directory of one. */
int dd_sketch_fasta(dd_ctx *, const char *path, int kmin, int kmax, uint8_t *regs);
/* Ingestion pipeline for a whole directory of genomes: `nthreads` loader threads (0 = auto) read
 * and inflate into pinned host buffers ahead of the GPU (bounded pool); a copy stream moves batch b+1
 * to the device and batch b-1's registers back while batch b is sketched; consecutive small files
 * share one launch (~128 MB per batch).  regs[nfiles][K][m] on the host.  Replaces the reference's
 * sequential per-genome loop (lib/huffman_dandd.py:402-407), each iteration of which re-inflates the
 * file once per k.  gzip files -- BGZF (bgzip) and ordinary single-member files of 1 MiB .. 1 GiB --
 * are copied compressed and inflated on the device, their CRC-32 and ISIZE checked; a block the device refuses sends the call through the host decoder, which
 * reports what is wrong (DD_NO_GPU_INFLATE=1: host decoder from the start).  FASTQ is accepted (kseq's
 * record rules: oracle/POLICIES.md P10). */
int dd_sketch_files(dd_ctx *, const char *const *paths, int nfiles, int kmin, int kmax,
                    uint8_t *regs, int nthreads);
/* statistics of the last dd_sketch_files call: wall time, time the GPU-driving thread waited for the loader
 * threads, number of batched launches, FASTA bytes sent to the device */
int dd_last_ingest_stats(dd_ctx *, double *wall_ms, double *loader_wait_ms, int *batches, uint64_t *bytes);
/* The text dd_sketch_files works on, for verification: the same pipeline (loaders, H2D, the device decoders of BGZF and
 * single-member .gz files, the host decoder and kseq's FASTQ rewrite where those apply), but every file's bytes AS THE
 * TOKENIZER IS ABOUT TO READ THEM are copied back into out[i] (caps[i] bytes of room; lens[i] = the text's length, also
 * when the buffer was too small: DD_EINVAL then).  For a .gz FASTA file that is exactly what `zcat` prints -- the check
 * that stands in for  zcat <fasta.gz> | cmp - <what dashing's gzread saw>  (lib/species_specifics.py:93: inputs are .gz). */
int dd_inflate_files(dd_ctx *, const char *const *paths, int nfiles, uint8_t *const *out, const size_t *caps,
                     size_t *lens, int nthreads);
/* Batched, HBM-resident form: ngenomes FASTA byte buffers already on the device,
 * regs_dev[ngenomes][K][m] on the device.  Asynchronous on the context's stream. */
int dd_sketch_device(dd_ctx *, const uint8_t *const *fasta_dev, const size_t *nbytes,
                     int ngenomes, int kmin, int kmax, uint8_t *regs_dev);

/* ---- union -----------------------------------------------------------------------
 * Replaces   dashing union -z -o <out> <in1> ... <inN>   (lib/sketch_classes.py:368-373):
 * out[i] = max_j in[j][i], len bytes (any multiple of m, e.g. a whole [K][m] slab). */
int dd_union(dd_ctx *, const uint8_t *const *in, int n, size_t len, uint8_t *out);
int dd_union_device(dd_ctx *, const uint8_t *const *in_dev, int n, size_t len, uint8_t *out_dev);

/* ---- cardinality -----------------------------------------------------------------
 * Replaces   dashing card --presketched <path...>   (lib/sketch_classes.py:306-321):
 * 64-bin register histogram + Ertl maximum-likelihood estimate, one double per sketch. */
int dd_card(dd_ctx *, const uint8_t *regs, double *est);
int dd_card_batch(dd_ctx *, const uint8_t *regs /*[njobs][m]*/, int njobs, double *est);
int dd_card_batch_device(dd_ctx *, const uint8_t *regs_dev, int njobs, double *est /*host*/);
/* histogram only (device kernel), hist[njobs][64] on the host */
int dd_hist_batch_device(dd_ctx *, const uint8_t *regs_dev, int njobs, uint32_t *hist);
/* Ertl MLE of one 64-bin histogram (host arithmetic, IEEE double, no device needed) */
double dd_ertl_mle(const uint32_t hist[64], int log2m);

/* ---- progressive unions ----------------------------------------------------------
 * Replaces the flat prefix unions of DeltaTree.sketch_ordering
 * (lib/huffman_dandd.py:644-663): for ordering o and prefix length j,
 * card[o][j-1][kk] = |union of leaf[ord[o][0..j-1]]| at k = kmin+kk, computed as a
 * running byte-max (max is associative, so it equals the flat union bit for bit) -- or, from
 * log2m 18 on and for n <= 32, as a running AND of threshold bit planes with a popcount per prefix
 * (dd_pscan.hip): the same integers.  Register bytes must be <= 63. */
int dd_progressive(dd_ctx *, const uint8_t *leaf /*[n][K][m]*/, int n, int K,
                   const int32_t *orderings /*[norder][n]*/, int norder,
                   double *card /*[norder][n][K]*/);
int dd_progressive_device(dd_ctx *, const uint8_t *leaf_dev, int n, int K,
                          const int32_t *orderings, int norder, double *card);

/* ---- all-pairs unions ------------------------------------------------------------
 * Replaces the 2-way unions of DeltaTree.pairwise_spiders (lib/huffman_dandd.py:666-695):
 * card[i][j][kk] for i<j is |leaf_i U leaf_j|; card[i][i][kk] is |leaf_i|; the lower
 * triangle mirrors the upper.  From log2m 12 on the histograms behind the estimates are counted as int8
 * Gram matrices on the matrix cores (dd_gram.hip; F_ij(v) = sum_r [a_ir <= v][a_jr <= v]): the same
 * integers as the byte-max + histogram kernel.  Register bytes must be <= 63. */
int dd_pairwise(dd_ctx *, const uint8_t *leaf /*[n][K][m]*/, int n, int K,
                double *card /*[n][n][K]*/);
int dd_pairwise_device(dd_ctx *, const uint8_t *leaf_dev, int n, int K, double *card);

/* ---- exact distinct k-mer count (the KMC stand-in) --------------------------------------
 * Replaces   kmc -ci1 -cs2 -k<K> [-b] -fm <fasta> <db> <tmp>   +   kmc_tools complex (set union)
 * +   kmc_tools info <db> | grep 'total k-mers'   (lib/sketch_classes.py:395,444-448,453-465):
 * number of distinct (canonical, per the context) k-mers over ALL n inputs together, k in 1..64.
 * Uses 16 (k<=32) or 32 (k>32) bytes of HBM per input byte up to a budget of 24 GiB (DD_EXACT_MB overrides);
 * larger inputs are counted in passes over disjoint parts of the k-mer space, so a union of any number of
 * genomes that fits HBM as FASTA bytes can be counted. */
int dd_exact_count_device(dd_ctx *, const uint8_t *const *fasta_dev, const size_t *nbytes, int n,
                          int k, uint64_t *distinct);
int dd_exact_count(dd_ctx *, const char *const *paths, int n, int k, uint64_t *distinct);

/* ---- measurement hooks (bench.py) -------------------------------------------------
 * When enabled, every launch of kernel `which` is bracketed by HIP events on the
 * context's stream.  dd_timing_read synchronises the stream and returns the summed
 * device time and the number of launches since the last reset. */
#define DD_KERNEL_PACK 0
#define DD_KERNEL_SWEEP 1
#define DD_KERNEL_UNION 2
#define DD_KERNEL_COUNT 3
int dd_timing_enable(dd_ctx *, int on);
int dd_timing_read(dd_ctx *, int which, double *total_ms, int *launches);
int dd_timing_reset(dd_ctx *);
/* per-call statistics of the last dd_sketch_* call: tokens (bases + breaks) packed,
 * register updates issued (tokens x K upper bound), number of sweep workgroups */
int dd_last_sketch_stats(dd_ctx *, uint64_t *tokens, uint64_t *updates, int *sweep_blocks);
/* which kernels the last dd_progressive* / dd_pairwise* call of the context ran (ABI 3): the union schedules of
 * lib/huffman_dandd.py:644-695 have two device forms each, picked by register count and set sizes -- a benchmark
 * line must name the one that ran, not the one its author expected */
#define DD_K2_NONE 0
#define DD_K2_PROGRESSIVE_STREAM 1 /* progressive_kernel: running byte-max, one LDS histogram per prefix  */
#define DD_K2_PROGRESSIVE_PSCAN 2  /* pscan_kernel: running AND of threshold bit planes (log2m >= 18, n <= 32) */
#define DD_K2_PAIRWISE_STREAM 3    /* pairwise_kernel: one LDS atomic per register per pair */
#define DD_K2_PAIRWISE_GRAM 4      /* gram_kernel: int8 Gram matrices on the matrix cores (log2m >= 12) */
int dd_last_k2_path(dd_ctx *);

/* ---- synthetic FASTA on the device (bench / tests) ----------
 * Byte-identical to oracle/dd_oracle.c:orc_synth_fasta for the same arguments. */
size_t dd_synth_size(uint64_t nbases, int nrec);
int dd_synth_fasta_device(dd_ctx *, uint64_t seed, int genome_index, uint64_t nbases, int nrec,
                          uint8_t *out_dev);
/* "realistic" mode: GC 35 %, 30 % soft-masked repeats (interspersed + tandem), 2 % N, contigs of 2..200 kbp; byte-identical
 * to oracle/dd_oracle.c:orc_synth_realistic_fasta.  The contig structure (and so the size) depends on the seed. */
size_t dd_synth_realistic_size(uint64_t seed, uint64_t nbases);
int dd_synth_realistic_device(dd_ctx *, uint64_t seed, int genome_index, uint64_t nbases, uint8_t *out_dev);

/* ---- the K1 job table of a sketch call, without running it (tests; needs no GPU) --------
 * What stands in for `parallel -j 95%`'s process-per-k scheduling (lib/huffman_dandd.py:214-218):
 * how dd_sketch_device would cut (genome x k x 65536-token tile) into workgroup jobs for genomes of
 * these sizes.  Writes at most `cap` jobs in launch order and returns how many there are (or a
 * negative DD_E* code).  kclass: -1 small-k bitmap class (k <= 9), -2 the exact k-mer sets of k = 10 (, 11)
 * at log2m >= 19 (one job per slice of the k-mer index space: `slice`), else the window class of the launch
 * (0: k <= 16, 1: <= 32, 3: 33..48, 2: 49..64); mode: 0 registers in LDS (log2m <= 16); 5 (log2m >= 17) registers in HBM
 * through scatter + chunk sort + replay, jobs listed epoch by epoch. */
typedef struct {
    int kclass, mode, lds_bytes;
    int genome, kfirst, nk;
    unsigned tile_begin, tile_end;
    int slice; /* big-bitmap class (kclass -2, log2m >= 19): the slice of k's index space the job records */
} dd_plan_job;
long dd_plan_sweep(int log2m, const size_t *nbytes, int ngenomes, int kmin, int kmax, dd_plan_job *out,
                   long cap);

/* ---- multi-GPU: one context = one process = one GPU; RCCL over xGMI -----------------------------------------
 * The reference's only parallelism is GNU parallel over k on one host (lib/huffman_dandd.py:217).  Here (genome x k)
 * sketch jobs are split over the GPUs of a node by the CALLER (every rank sketches its own genomes: no exchange), and
 * what crosses xGMI is
 *   dd_allreduce_max_u8   the root union: every rank's [K][m] slab of byte-max-merged registers, in place
 *                         (ncclAllReduce, ncclUint8, ncclMax) -- stands in for the N-way  dashing union -z -o <root> <all leaves>
 *                         (lib/sketch_classes.py:368-373) over leaves that live on different GPUs;
 *   dd_allgather_u8       every rank's leaf slabs, for the schedules that need all leaves on every rank
 *                         (dd_progressive*, dd_pairwise*): recv_dev[world][n].
 * Both run on the context's stream (dd_set_stream) and return once enqueued.  Rank 0 makes the 128-byte id with
 * dd_comm_unique_id and hands it to the other ranks by any means (a file, MPI, a socket); every rank then calls
 * dd_comm_init -- collectively -- on a context of the GPU it owns.  librccl.so is opened on first use (DD_RCCL_LIB names
 * another copy); a process that never calls these needs no RCCL. */
#define DD_COMM_ID_BYTES 128
int dd_comm_unique_id(uint8_t *id /* [DD_COMM_ID_BYTES] */);
int dd_comm_init(dd_ctx *, int rank, int world, const uint8_t *id);
int dd_comm_destroy(dd_ctx *);
/* world = 0: the context belongs to no communicator; the counters are the collectives issued since dd_comm_init */
int dd_comm_info(dd_ctx *, int *rank, int *world, unsigned long long *allreduces, unsigned long long *allgathers);
int dd_allreduce_max_u8(dd_ctx *, uint8_t *regs_dev, size_t n);
int dd_allgather_u8(dd_ctx *, const uint8_t *send_dev, size_t n, uint8_t *recv_dev /* [world][n] */);

#ifdef __cplusplus
}
#endif
#endif
