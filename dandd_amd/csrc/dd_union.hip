// dd_union.hip -- K2 (byte-max union + 64-bin register histograms) and K3 (Ertl MLE on device).
//
// Replaces  dashing union -z -o <out> <in...>   (/root/reference/lib/sketch_classes.py:368-373)
// and the histogram half of  dashing card --presketched <path>  (:306-321), batched over the
// union schedules DandD generates:
//   * N-way union                      (DeltaTree root / any tree node)
//   * running max along an ordering     (DeltaTree.sketch_ordering, lib/huffman_dandd.py:644-663;
//                                        equals the flat prefix unions because max is associative)
//   * all pairs                         (DeltaTree.pairwise_spiders, lib/huffman_dandd.py:666-695)
// HBM/L2-bound: 16-byte loads, SWAR byte max, LDS histograms privatised 32 ways, bin-major (one bank per copy).
#include "dd_common.h"
#include "dd_kernels.h"

namespace dd {
namespace {

constexpr int HCOPIES = 32;  // privatised LDS histograms per workgroup

// Histogram image: h[bin][copy], copy = lane % 32.  ds_add_u32 is serviced in two groups of 32 lanes, each over 32
// banks of 4 bytes: with the copy as the fastest index every lane of a group adds into ITS OWN bank whatever bins the
// bytes name -- no bank conflict is possible (lanes l and l + 32 share a copy but not a group).  The copy-major
// image this replaces (h[copy][65]) put (copy + bin) % 32 on the bank: ~3.5 lanes of a group collided on average.
#ifdef DD_HIST_COPY_MAJOR   // (A/B builds only)
typedef uint32_t HistImage[HCOPIES][65];
#define DD_HIST_AT(h, bin, copy) (h)[copy][bin]
#else
typedef uint32_t HistImage[64][HCOPIES];
#define DD_HIST_AT(h, bin, copy) (h)[bin][copy]
#endif

DD_D uint4 bmax16(uint4 a, uint4 b) {
    return make_uint4(bmax4(a.x, b.x), bmax4(a.y, b.y), bmax4(a.z, b.z), bmax4(a.w, b.w));
}

__global__ __launch_bounds__(256) void union_kernel(const uint8_t* const* __restrict__ in, int n,
                                                    size_t len16, uint8_t* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len16;
         i += (size_t)gridDim.x * blockDim.x) {
        uint4 acc = reinterpret_cast<const uint4*>(in[0])[i];
        for (int j = 1; j < n; ++j) acc = bmax16(acc, reinterpret_cast<const uint4*>(in[j])[i]);
        reinterpret_cast<uint4*>(out)[i] = acc;
    }
}

// add the 16 register bytes of v to the workgroup's privatised histograms
DD_D void hist_add16(HistImage h, uint4 v) {
    const int copy = threadIdx.x & (HCOPIES - 1);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int b = 0; b < 4; ++b) atomicAdd(&DD_HIST_AT(h, (w[q] >> (8 * b)) & 63u, copy), 1u);
    }
}

DD_D void hist_zero(HistImage h) {
    for (int i = threadIdx.x; i < (int)(sizeof(HistImage) / 4); i += blockDim.x) (&h[0][0])[i] = 0;
}

// fold the privatised copies and add them to a global 64-bin histogram
DD_D void hist_flush(HistImage h, uint32_t* __restrict__ gh, bool exclusive) {
    if (threadIdx.x < 64) {
        uint32_t s = 0;
#pragma unroll
        for (int c = 0; c < HCOPIES; ++c) s += DD_HIST_AT(h, threadIdx.x, (c + threadIdx.x) & (HCOPIES - 1));  // (rotated: thread t starts at bank t)
        if (exclusive)
            gh[threadIdx.x] = s;
        else if (s)
            atomicAdd(&gh[threadIdx.x], s);
    }
}

// PC = 16-byte pieces per thread: a workgroup's tile is PC x 16 KiB of a row.  Rows of 64 KiB and more use PC = 4:
// four times fewer barriers, histogram folds and (rows of several tiles add their partial histograms with global
// atomics) global atomics per byte -- 264 M of those for the pairs of 64 sketches of 1 MiB before.
// one workgroup per (sketch, tile of its registers)
template <int PC>
__global__ __launch_bounds__(1024) void hist_kernel(const uint8_t* __restrict__ regs, int p,
                                                    int tiles, uint32_t* __restrict__ hist) {
    __shared__ HistImage h;
    const int job = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const size_t m16 = ((size_t)1 << p) >> 4;
    const size_t piece = (size_t)tile * blockDim.x * PC + threadIdx.x;
    hist_zero(h);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PC; ++q)
        if (piece + (size_t)q * blockDim.x < m16)
            hist_add16(h, reinterpret_cast<const uint4*>(regs + ((size_t)job << p))[piece + (size_t)q * blockDim.x]);
    __syncthreads();
    hist_flush(h, hist + (size_t)job * 64, tiles == 1);
}

// one workgroup per (ordering, k, tile): running max over the ordering, one histogram per prefix
template <int PC>
__global__ __launch_bounds__(1024) void progressive_kernel(const uint8_t* __restrict__ leaf, int n,
                                                           int K, int p, int tiles,
                                                           const int32_t* __restrict__ ord,
                                                           uint32_t* __restrict__ hist) {
    __shared__ HistImage h;
    const int tile = blockIdx.x % tiles;
    const int kk = (blockIdx.x / tiles) % K;
    const int o = blockIdx.x / tiles / K;
    const size_t m16 = ((size_t)1 << p) >> 4;
    const size_t piece = (size_t)tile * blockDim.x * PC + threadIdx.x;
    uint4 run[PC];
#pragma unroll
    for (int q = 0; q < PC; ++q) run[q] = make_uint4(0, 0, 0, 0);
    for (int j = 0; j < n; ++j) {
        const int gi = ord[(size_t)o * n + j];
        hist_zero(h);
        __syncthreads();
        const uint8_t* src = leaf + (((size_t)gi * K + kk) << p);
#pragma unroll
        for (int q = 0; q < PC; ++q)
            if (piece + (size_t)q * blockDim.x < m16) {
                run[q] = bmax16(run[q], reinterpret_cast<const uint4*>(src)[piece + (size_t)q * blockDim.x]);
                hist_add16(h, run[q]);
            }
        __syncthreads();
        hist_flush(h, hist + (((size_t)o * n + j) * K + kk) * 64, tiles == 1);
        __syncthreads();
    }
}

// one workgroup per (i, k, tile): row i of the pair matrix, j = i..n-1
template <int PC>
__global__ __launch_bounds__(1024) void pairwise_kernel(const uint8_t* __restrict__ leaf, int n,
                                                        int K, int p, int tiles,
                                                        uint32_t* __restrict__ hist) {
    __shared__ HistImage h;
    const int tile = blockIdx.x % tiles;
    const int kk = (blockIdx.x / tiles) % K;
    const int i = blockIdx.x / tiles / K;
    const size_t m16 = ((size_t)1 << p) >> 4;
    const size_t piece = (size_t)tile * blockDim.x * PC + threadIdx.x;
    uint4 a[PC];
#pragma unroll
    for (int q = 0; q < PC; ++q) {
        a[q] = make_uint4(0, 0, 0, 0);
        if (piece + (size_t)q * blockDim.x < m16)
            a[q] = reinterpret_cast<const uint4*>(leaf + (((size_t)i * K + kk) << p))[piece + (size_t)q * blockDim.x];
    }
    for (int j = i; j < n; ++j) {
        hist_zero(h);
        __syncthreads();
        const uint8_t* src = leaf + (((size_t)j * K + kk) << p);
#pragma unroll
        for (int q = 0; q < PC; ++q)
            if (piece + (size_t)q * blockDim.x < m16)
                hist_add16(h, bmax16(a[q], reinterpret_cast<const uint4*>(src)[piece + (size_t)q * blockDim.x]));
        __syncthreads();
        hist_flush(h, hist + (((size_t)i * n + j) * K + kk) * 64, tiles == 1);
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void mle_kernel(const uint32_t* __restrict__ hist, size_t njobs,
                                                  int p, double relerr, double* __restrict__ est) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= njobs) return;
    uint32_t c[64];
    uint64_t tot = 0;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        c[i] = hist[j * 64 + i];
        tot += c[i];
    }
    // a histogram that does not count exactly m registers is an unused slot (e.g. the lower
    // triangle of the pair matrix): report 0 instead of walking off the array
    est[j] = (tot == (1ull << p)) ? ertl_mle(c, p, relerr) : 0.0;
}

// ---- synthetic FASTA (byte-identical to oracle/dd_oracle.c:orc_synth_fasta) ------------
constexpr uint64_t SYN_HDR = 16, SYN_LINE = 80;

DD_HD uint64_t syn_rec_bytes(uint64_t L) { return SYN_HDR + L + (L + SYN_LINE - 1) / SYN_LINE; }

DD_D uint8_t syn_base(uint64_t seed, uint64_t seed_g, uint64_t pos) {
    uint32_t b = (uint32_t)(splitmix64(seed ^ pos) & 3);
    const uint64_t r = splitmix64(seed_g ^ pos);
    if (r % 100 == 0) b = (b + 1 + (uint32_t)((r >> 32) % 3)) & 3;
    uint8_t ch = (uint8_t)("ACGT"[b]);
    if (splitmix64(seed_g ^ 0x4E4E4E4E00000000ull ^ (pos / 100)) % 1000 == 0) ch = 'N';
    if (splitmix64(seed_g ^ 0x6C6C6C6C00000000ull ^ (pos / 500)) % 10 == 0) ch |= 0x20;
    return ch;
}

__global__ __launch_bounds__(256) void synth_kernel(uint64_t seed, uint64_t seed_g, int gi,
                                                    uint64_t nbases, int nrec, uint64_t total,
                                                    uint8_t* __restrict__ out) {
    const uint64_t per = nbases / (uint64_t)nrec;
    const uint64_t recsz = syn_rec_bytes(per);
    for (uint64_t off = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; off < total;
         off += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t r = recsz ? off / recsz : 0;
        if (r > (uint64_t)(nrec - 1)) r = nrec - 1;
        const uint64_t o = off - r * recsz;
        const uint64_t L = (r == (uint64_t)(nrec - 1)) ? nbases - per * (uint64_t)(nrec - 1) : per;
        uint8_t ch;
        if (o < SYN_HDR) {
            const char hexd[] = "0123456789abcdef";
            if (o == 0) ch = '>';
            else if (o == 1) ch = 'g';
            else if (o < 6) ch = hexd[(gi >> (12 - 4 * (int)(o - 2))) & 15];
            else if (o == 6) ch = '.';
            else if (o == 7) ch = 'r';
            else if (o < 12) ch = hexd[((int)r >> (12 - 4 * (int)(o - 8))) & 15];
            else if (o < 15) ch = ' ';
            else ch = '\n';
        } else {
            const uint64_t q = o - SYN_HDR;
            const uint64_t line = q / (SYN_LINE + 1), col = q % (SYN_LINE + 1);
            const uint64_t j = line * SYN_LINE + col;
            if (col == SYN_LINE || j >= L) ch = '\n';
            else ch = syn_base(seed, seed_g, r * per + j);
        }
        out[off] = ch;
    }
}

inline int pieces_for(int p) { return p >= 16 ? 4 : 1; }  // 16-byte pieces per thread (the kernels' PC)
inline int tiles_for(int p) {
    const size_t m16 = ((size_t)1 << p) >> 4, per_tile = (size_t)1024 * pieces_for(p);
    return (int)((m16 + per_tile - 1) / per_tile);
}
inline int threads_for(int p) {
    const size_t m16 = ((size_t)1 << p) >> 4;
    size_t t = m16 < 1024 ? m16 : 1024;
    if (t < 64) t = 64;
    return (int)t;
}

}  // namespace

void launch_union(const uint8_t* const* in_dev, int n, size_t len, uint8_t* out_dev, hipStream_t st) {
    const size_t len16 = len >> 4;
    if (!len16) return;
    size_t blocks = (len16 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(union_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in_dev, n, len16, out_dev);
}

void launch_hist(const uint8_t* regs_dev, int njobs, int p, uint32_t* hist_dev, hipStream_t st) {
    if (njobs <= 0) return;
    const int tiles = tiles_for(p);
    if (tiles > 1) (void)hipMemsetAsync(hist_dev, 0, (size_t)njobs * 64 * sizeof(uint32_t), st);
    if (pieces_for(p) == 4)
        hipLaunchKernelGGL(hist_kernel<4>, dim3((unsigned)(njobs * tiles)), dim3(threads_for(p)), 0, st, regs_dev, p, tiles, hist_dev);
    else
        hipLaunchKernelGGL(hist_kernel<1>, dim3((unsigned)(njobs * tiles)), dim3(threads_for(p)), 0, st, regs_dev, p, tiles, hist_dev);
}

void launch_progressive(const uint8_t* leaf_dev, int n, int K, int p, const int32_t* ord_dev,
                        int norder, uint32_t* hist_dev, hipStream_t st) {
    if (n <= 0 || K <= 0 || norder <= 0) return;
    const int tiles = tiles_for(p);
    if (tiles > 1) (void)hipMemsetAsync(hist_dev, 0, (size_t)norder * n * K * 64 * sizeof(uint32_t), st);
    if (pieces_for(p) == 4)
        hipLaunchKernelGGL(progressive_kernel<4>, dim3((unsigned)(norder * K * tiles)), dim3(threads_for(p)), 0, st, leaf_dev, n, K, p, tiles, ord_dev, hist_dev);
    else
        hipLaunchKernelGGL(progressive_kernel<1>, dim3((unsigned)(norder * K * tiles)), dim3(threads_for(p)), 0, st, leaf_dev, n, K, p, tiles, ord_dev, hist_dev);
}

void launch_pairwise(const uint8_t* leaf_dev, int n, int K, int p, uint32_t* hist_dev, hipStream_t st) {
    if (n <= 0 || K <= 0) return;
    const int tiles = tiles_for(p);
    (void)hipMemsetAsync(hist_dev, 0, (size_t)n * n * K * 64 * sizeof(uint32_t), st);
    if (pieces_for(p) == 4)
        hipLaunchKernelGGL(pairwise_kernel<4>, dim3((unsigned)(n * K * tiles)), dim3(threads_for(p)), 0, st, leaf_dev, n, K, p, tiles, hist_dev);
    else
        hipLaunchKernelGGL(pairwise_kernel<1>, dim3((unsigned)(n * K * tiles)), dim3(threads_for(p)), 0, st, leaf_dev, n, K, p, tiles, hist_dev);
}

void launch_mle(const uint32_t* hist_dev, size_t njobs, int p, double* est_dev, hipStream_t st) {
    if (!njobs) return;
    hipLaunchKernelGGL(mle_kernel, dim3((unsigned)((njobs + 255) / 256)), dim3(256), 0, st, hist_dev,
                       njobs, p, mle_relerr(p), est_dev);
}

size_t synth_size(uint64_t nbases, int nrec) {
    const uint64_t per = nbases / (uint64_t)nrec;
    const uint64_t last = nbases - per * (uint64_t)(nrec - 1);
    return (size_t)(syn_rec_bytes(per) * (uint64_t)(nrec - 1) + syn_rec_bytes(last));
}

void launch_synth(uint64_t seed, int gi, uint64_t nbases, int nrec, uint8_t* out_dev, hipStream_t st) {
    const uint64_t total = synth_size(nbases, nrec);
    if (!total) return;
    const uint64_t seed_g = splitmix64(seed + (uint64_t)gi + 1);
    uint64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(synth_kernel, dim3((unsigned)blocks), dim3(256), 0, st, seed, seed_g, gi, nbases,
                       nrec, total, out_dev);
}

}  // namespace dd
