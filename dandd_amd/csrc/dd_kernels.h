// dd_kernels.h -- host-callable launchers of the gfx950 kernels behind the C ABI.
#pragma once
#include <vector>
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace dd {

// ---------------------------------------------------------------------------------------
// Token stream of one genome in HBM (output of K0, input of K1).
//   codes : 2 bits per token, token j of word w at bits [2j, 2j+1]   (16 tokens / u32);
//           unspecified for BREAK tokens (no window that contains one is ever used)
//   bad   : 1 bit per token, 1 = BREAK (non-ACGT byte or record boundary) (32 tokens / u32)
//   ntok  : device scalar, number of tokens; the stream is padded with BREAKs to a
//           multiple of 64 tokens (one K1 thread segment).
// ---------------------------------------------------------------------------------------
struct TokenStream {
    uint32_t* codes;
    uint32_t* bad;
    unsigned long long* ntok;
};

constexpr int kPackThreads = 256;
constexpr int kPackBytesPerThread = 64;
constexpr int kPackChunk = kPackThreads * kPackBytesPerThread;  // 16 KiB of FASTA per K0 workgroup
constexpr int kSegTokens = 64;                 // tokens per K1 thread segment

// scratch for one K0 run over n bytes: 4 x int64 per chunk, + one for the position of the first header character
inline size_t pack_chunks(size_t n) { return (n + kPackChunk - 1) / kPackChunk; }
inline size_t pack_scratch_bytes(size_t n) { return ((pack_chunks(n) + 1) * 4 + 2) * sizeof(long long); }
// capacity (in u32 words) of the code / bad arrays for an n-byte FASTA (tokens <= n)
inline size_t codes_words(size_t n) { return ((n + 63) / 64 + 1) * 4; }
inline size_t bad_words(size_t n) { return ((n + 63) / 64 + 1) * 2; }

struct PackGenome {             // one per genome, device-resident table
    const uint8_t* fa;          // FASTA bytes, 16-byte aligned
    size_t n;                   // bytes
    size_t nchunks;             // pack_chunks(n)
    long long* scratch;         // pack_scratch_bytes(n)
    TokenStream out;
};
// gridDim.y = genome: every genome of the batch is packed by the same three launches
void launch_pack_batch(const PackGenome* tab_dev, int ngenomes, size_t max_chunks, hipStream_t st);

// ---------------------------------------------------------------------------------------
// K1: fused k-sweep sketch.
// ---------------------------------------------------------------------------------------
struct SweepGenome {            // one per genome, device-resident table
    const uint32_t* codes;
    const uint32_t* bad;
    const unsigned long long* ntok;
    uint8_t* regs;              // [K][m] for this genome
    uint32_t* bitmap;           // presence bitmaps of the canonical k-mers, k = 1..kBitmapMaxK (or null)
    uint32_t* bigmap;           // presence bitmaps of k = 10 (, 11) at log2m >= 19 (or null): bigmap_offset_words
};

// Small-k path: for k <= kBitmapMaxK there are at most 4^k <= 262144 distinct k-mers, so K1 only
// records WHICH k-mers occur (one LDS bit each) and hashes every distinct one once afterwards.
constexpr int kBitmapMaxK = 9;           // (k = 10 was tried as a 128 KiB class of its own: its set completes too late to pay)
constexpr int kBitmapWords = 10924;      // sum over k = 1..9 of max(1, 4^k / 32)
constexpr int kBitmapStride = 11008;     // words per genome (256-byte multiple); the slack holds one "complete" flag per k
// first word of k's bitmap inside a genome's block
inline constexpr int bitmap_offset(int k) {
    int off = 0;
    for (int j = 1; j < k; ++j) off += (j < 3) ? 1 : (1 << (2 * j - 5));
    return off;
}
inline constexpr int bitmap_words(int k) { return (k < 3) ? 1 : (1 << (2 * k - 5)); }
// Second small-k class, bucket mode at log2m >= 19 only: k = 10 (and k = 11 at log2m 20).  There a row has fewer
// distinct k-mers than (about twice) its registers, a register group almost always holds an EMPTY register, the
// group-minimum filter lets everything through and every update pays a probe of the row: 2.4 ms per k against
// 1.0 for the other rows (10 x 50 Mbp).  The k-mer set itself is smaller than the row, so it is recorded
// exactly, in 128 KiB slices of LDS (one workgroup per CU), and each distinct k-mer is hashed afterwards.
// Index of a k-mer: the canonical value (2k bits); for odd k in canonical mode the strand whose MIDDLE base is
// A or C with that base's high bit dropped (2k - 1 bits: one of a k-mer and its reverse complement always
// qualifies), which halves the slices of k = 11.
constexpr int kBigmapMinK = 10, kBigmapMaxK = 11;
constexpr int kBigmapSliceWords = 1 << 15;  // 2^20 bits
inline constexpr int bigmap_last_k(int log2m) { return log2m >= 20 ? 11 : (log2m >= 19 ? 10 : 0); }
inline constexpr int bigmap_bits(int k, bool canon) { return (canon && (k & 1)) ? 2 * k - 1 : 2 * k; }
inline constexpr int bigmap_slices(int k, bool canon) { return 1 << (bigmap_bits(k, canon) - 20); }
inline constexpr size_t bigmap_offset_words(int k, bool canon) {  // k's first word in a genome's block
    size_t off = 0;
    for (int j = kBigmapMinK; j < k; ++j) off += (size_t)bigmap_slices(j, canon) * kBigmapSliceWords;
    return off;
}
struct SweepJob {               // one per workgroup, device-resident table
    int genome;
    int kfirst;                 // first k of the group (consecutive ks)
    int nk;                     // number of ks in the group (<= slots that fit LDS)
    int krow;                   // row of kfirst in the genome's [K][m] slab
    unsigned tile_begin, tile_end;  // tiles of (threads x 64) tokens
    int slice;                  // big-bitmap jobs: which 2^20-bit slice of k's index space the workgroup records
};
constexpr int kBucketMode = 5;
constexpr int kBucketFromLog2m = 17;  // registers stay in HBM (scatter + replay) from this log2m on
struct SweepPlan {
    int log2m;
    int canonical;
    int threads;                // workgroup size
    int lds_bytes;              // dynamic LDS per workgroup
    int mode;                   // 0: registers in LDS (log2m <= 16); 5 (kBucketMode): in HBM, scatter to buckets + replay (below)
    // kBucketMode only
    int logg = 0;               // one 4-bit filter entry (bounds saturate at 15, two entries per byte) per 2^logg registers
    int nb_log2 = 0;            // 2^nb_log2 index tiles of 64 KiB per row (replay)
    unsigned cap_chunks = 0;    // 1024-record chunks per row and epoch
    int nepochs = 0;
};
void launch_sweep(const SweepGenome* genomes_dev, const SweepJob* jobs_dev, int njobs, int kclass,
                  const SweepPlan& plan, hipStream_t st);

// ---------------------------------------------------------------------------------------
// K1 for log2m >= 17 (one register array does not fit LDS twice per CU, or at all; DandD's default is -r 20,
// /root/reference/lib/dandd_cmd.py:187): three kernels per EPOCH (a range of token tiles).
//   scatter: hash every k-mer of the epoch; an update whose rho cannot exceed the filter's lower bound
//            for its register group is dropped, the others are queued per wave in LDS, checked 64 at a time
//            against the row itself, and what survives leaves as 4-byte records (idx | rho << 24), one 256-byte
//            block at a time, for the ROW's record stream -- a dense stream: waves reserve 256 records per atomic
//            add on the row's cursor.  The first epoch (registers all zero) has no filter and no queues: a record goes
//            straight from the hash into the bin of its index tile (16 fixed regions of 4480 records per tile of tokens,
//            counts in seg[chunk][16]), except the updates of rho = 1 -- half of them --, which set a bit of BucketRow::ones.
//   sort   : every 1024-record chunk is sorted by index tile in place (HBM-bound streaming pass; not needed
//            after the first epoch).
//   replay : one workgroup per (row, 64 KiB index tile): tile into LDS, apply the tile's segment of every
//            chunk with LDS operations, store the tile back with plain 16-byte stores and refresh the
//            tile's filter.
// The filter of epoch e is exact knowledge of the registers after epoch e-1, so what scatter drops can
// never matter; the first epoch is four tokens per register long, each later one as long as all before it.
// ---------------------------------------------------------------------------------------
struct BucketRow {              // one per (genome, k) row of the call: table[genome * K + (k - kmin)]
    uint8_t* regs;              // the row's m registers in the caller's slab
    uint32_t* area;             // record stream: chunk c at area + c * 1024; null = row not bucketed
    uint32_t* cursor;           // records reserved this epoch, in blocks of 64 (may run past the capacity: overflow)
    uint32_t* fill;             // [cap_chunks] records of each chunk after the sort dropped the null ones
    uint16_t* seg;              // [cap_chunks][16] where each index tile's records start inside a sorted chunk
    uint8_t* filter;            // [m >> logg] lower bound per register group
    uint32_t* ones;             // [m / 32] first epoch: bit r = register r saw an update with rho = 1
};
struct ScatterParams {
    const BucketRow* rows;
    int K;                      // rows per genome
    int logg;
    unsigned cap_chunks;
    int nb_log2;                // index tiles per row (log2)
};
void launch_scatter(const SweepGenome* genomes_dev, const SweepJob* jobs_dev, int njobs, int kclass,
                    const SweepPlan& plan, const ScatterParams& sp, hipStream_t st, bool first_epoch);
// (sort +) replay + cursor reset of rows k0 .. k0+nks-1 (indices into a genome's K rows) of every genome; first_epoch: the
// records are the binned tiles the first epoch's scatter left (no sort pass)
void launch_replay(const BucketRow* rows_dev, int ngenomes, int K, int k0, int nks, const SweepPlan& plan, hipStream_t st, bool first_epoch);
int sweep_max_lds_bytes();
// small-k class: jobs carry ks <= kBitmapMaxK; records k-mer presence in genome.bitmap
// (kfirst..klast: the ks of the class; the LDS image covers exactly their bitmaps)
void launch_bitmap(const SweepGenome* genomes_dev, const SweepJob* jobs_dev, int njobs, int kfirst, int klast,
                   int canonical, hipStream_t st);
// one workgroup per (genome, k in [kfirst, klast]): hash every recorded k-mer once into slab row k-kmin
// big-bitmap class (log2m >= 19): jobs carry one k and one slice each; finish hashes the recorded sets into rows
// kfirst..klast, one workgroup per (k, genome, 64 KiB index tile)
void launch_bigmap(const SweepGenome* genomes_dev, const SweepJob* jobs_dev, int njobs, int canonical, hipStream_t st);
void launch_bigmap_finish(const SweepGenome* genomes_dev, int ngenomes, int kfirst, int klast, int kmin, int log2m,
                          int canonical, hipStream_t st);
void launch_bitmap_finish(const SweepGenome* genomes_dev, int ngenomes, int kfirst, int klast, int kmin,
                          int log2m, hipStream_t st);

// ---------------------------------------------------------------------------------------
// K2: byte-max union + 64-bin histograms.
// ---------------------------------------------------------------------------------------
void launch_union(const uint8_t* const* in_dev, int n, size_t len, uint8_t* out_dev, hipStream_t st);
void launch_hist(const uint8_t* regs_dev, int njobs, int log2m, uint32_t* hist_dev, hipStream_t st);
// running max along each ordering; hist[(o*n + j)*K + kk][64]
void launch_progressive(const uint8_t* leaf_dev, int n, int K, int log2m, const int32_t* ord_dev,
                        int norder, uint32_t* hist_dev, hipStream_t st);
// hist[((i*n)+j)*K + kk][64] for i <= j
void launch_pairwise(const uint8_t* leaf_dev, int n, int K, int log2m, uint32_t* hist_dev,
                     hipStream_t st);
// the same histograms through int8 Gram matrices on the matrix cores (dd_gram.hip); hist_dev zeroed by the caller
bool gram_usable(int n, int log2m);
size_t gram_scratch_bytes(int n, int K, int log2m, int* sp_per_launch);
void launch_pairwise_gram(const uint8_t* leaf_dev, int n, int K, int log2m, uint32_t* hist_dev, void* scratch,
                          hipStream_t st);
void launch_register_range(const uint8_t* leaf_dev, int n, int K, int log2m, uint32_t* rng_dev, hipStream_t st);
// progressive unions as a bit-plane AND-scan (dd_pscan.hip); the streaming launch_progressive stays as the fallback
bool pscan_usable(int n, int norder, int log2m);
size_t pscan_scratch_bytes(int n, int K, int log2m, int norder);
bool launch_progressive_pscan(const uint8_t* leaf_dev, int n, int K, int log2m, const int32_t* ord_dev, int norder, const uint32_t* rng_host,
                              void* scratch, uint32_t* hist_dev, hipStream_t st);
void launch_mle(const uint32_t* hist_dev, size_t njobs, int log2m, double* est_dev, hipStream_t st);

// ---------------------------------------------------------------------------------------
// exact distinct k-mer count (KMC stand-in): extract -> radix sort -> count distinct
// ---------------------------------------------------------------------------------------
struct ExactGenome {
    const uint32_t* codes;
    const uint32_t* bad;
    const unsigned long long* ntok;
    unsigned long long base;    // first slot of this genome in the k-mer arrays
};
// mode 0: every k-mer to its token's slot; counters[0] += valid k-mers written, counters[1] |= 1 if a valid
//         k-mer is all-ones (T^k, non-canonical)
// mode 1: hist[4096] += k-mers per bin of the k-mer space (kExactBins bins by a mix of the k-mer)
// mode 2: k-mers of bins [bin_lo, bin_hi) appended densely from slot counters[3] on (counters[3] += their number)
constexpr int kExactBins = 4096;
void launch_kmer_extract(const ExactGenome* tab_dev, int ng, size_t max_segments, int k, int canonical,
                         uint64_t* lo, uint64_t* hi, unsigned long long* counters, hipStream_t st, int mode = 0,
                         unsigned long long* hist = nullptr, uint32_t bin_lo = 0, uint32_t bin_hi = 0);
size_t exact_sort_temp_bytes(size_t n, int k);
// counters[2] += number of distinct values among the n slots (unwritten slots hold all-ones)
hipError_t launch_exact_sort_count(uint64_t* lo, uint64_t* hi, uint64_t* lo_alt, uint64_t* hi_alt, size_t n,
                                   int k, void* temp, size_t temp_bytes, unsigned long long* counters,
                                   hipStream_t st);

// ---------------------------------------------------------------------------------------
// BGZF blocks inflated on the device (dd_ginflate.hip): one wave per block, text straight into the FASTA buffer
// ---------------------------------------------------------------------------------------
struct InflateJob {
    const uint8_t* in;          // the block: a whole gzip member (header, deflate data, CRC-32, ISIZE)
    uint32_t in_len;
    uint32_t out_len;           // its ISIZE (<= 65536)
    uint8_t* out;               // where its text goes
};
// one single-member gzip file of a batch, inflated on the device in pieces (dd_ginflate.hip: launch_gunzip_members)
struct RawFile {
    const uint8_t* in;          // the whole file on the device, 256-byte aligned
    uint32_t in_len;            // ... up to the end of THIS member (a file of several members: one RawFile each, same `in`)
    uint64_t first_bit;         // where its deflate data starts (behind the member's gzip header), counted from `in`
    uint32_t guess_bits;        // the block-start finder looks at one range of this many bits per piece
    uint32_t nguess;            // ranges = entries of this file in starts / lens / offs
    uint32_t piece0;            // its first entry there
    uint32_t isize;             // bytes of text (the member's ISIZE)
    uint16_t* sym;              // [nguess][range_syms] symbols (a byte, or 0x8000 | position in the 32 KiB in front of the piece):
                                //   a piece that starts in range j and runs over r ranges owns r x range_syms of them
    uint32_t range_syms;
    uint16_t* arena;            // [isize]: the symbols of pieces too long for their ranges (counted first, then written here)
    uint8_t* windows;           // what stands in the 32 KiB in front of every piece, in two levels (dd_ginflate.hip: piece_maps_kernel):
                                //   u16 [nguess][32768] maps relative to the piece's group start, u16 [ngroups][32768] the groups' own maps,
                                //   u8 [ngroups][32768] the windows at the groups' starts
    uint32_t group0, ngroups;   // groups of kPieceGroup ranges; group0 = this file's first group in the batch
    uint8_t* text;              // [isize] where the text goes
};
// one device-inflated text of a batch and what kseq's record rules ask of it (dd_fastq.hip)
struct TextJob {
    uint8_t* text;
    uint32_t n;
    uint32_t fastq;             // classed by its first bytes: 1 = four-line FASTQ expected, 0 = FASTA (no line may start with '+')
    uint32_t block0;            // its first 4 KiB block among the batch's
    uint32_t* blk_count;        // [blocks]: newlines per block, then their exclusive scan (FASTQ only)
    uint32_t* nl;               // [nl_cap]: the positions of its newlines
    uint32_t nl_cap;
    uint32_t* nl_total;
};
// OR-ed into *errors_dev when a text is not what its first bytes said (not a decoder refusal).  OR-ed, not added: one thread per
// offending RECORD raises it, and 2304 records x 2^24 is 9 x 2^32 = 0 -- a text with exactly that many odd records once came out
// "accepted" (scripts/fuzz_fastq.py, seed 510 draw 206)
constexpr uint32_t kNotFourLine = 0x1000000u;
void launch_text_rules(const TextJob* jobs_dev, int njobs, uint32_t nblocks, bool any_fastq, uint32_t* errors_dev, hipStream_t st);
size_t inflate_lds_bytes();
// starts_dev: npieces u64 (bit positions); tables_dev: four arrays of npieces u32 (lens, offs, over, abase), `stride` words apart
constexpr uint32_t kPieceGroup = 32;
constexpr uint32_t kSizeMismatch = 0x10000u;   // OR-ed into *errors_dev when a file's pieces do not add up to its ISIZE (decoder refusals ADD 1 each)
inline size_t gunzip_window_bytes(size_t nguess) {   // the `windows` area of a file with that many ranges
    const size_t ngroups = (nguess + kPieceGroup - 1) / kPieceGroup;
    return nguess * 65536 + ngroups * (65536 + 32768);
}
void launch_gunzip_members(const RawFile* files_dev, int nfiles, int npieces, int ngroups, int nchunks, uint64_t* starts_dev, uint32_t* tables_dev, size_t stride,
                           const uint32_t* chunk0_dev, uint32_t* crcs_dev, uint32_t* errors_dev, hipStream_t st);
// *errors_dev += blocks that did not decode (the caller falls back to the host decoder)
void launch_inflate_bgzf(const InflateJob* jobs_dev, int njobs, uint32_t* errors_dev, hipStream_t st);

// synthetic FASTA
void launch_synth(uint64_t seed, int gi, uint64_t nbases, int nrec, uint8_t* out_dev, hipStream_t st);
size_t synth_size(uint64_t nbases, int nrec);
// "realistic" mode (dd_synth.hip): contig table on the host, bytes on the device
std::vector<uint64_t> synth_realistic_table(uint64_t seed, uint64_t nbases);
void launch_synth_realistic(uint64_t seed, int gi, const uint64_t* tab_dev, uint32_t ncontigs, uint64_t total, uint8_t* out_dev,
                            hipStream_t st);

}  // namespace dd
