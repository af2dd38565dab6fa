// dd_fastq.hip -- FASTQ among the texts the device has inflated (dd_ginflate.hip), resolved where they are: in HBM.
//
// `dashing sketch` reads its inputs through klib's kseq.h, FASTA and FASTQ alike (/root/reference/lib/sketch_classes.py:358-365
// hands it whatever the species directory holds; oracle/POLICIES.md P10).  K0 knows kseq's FASTA rules; a FASTQ record's
// quality text is a length-counted field, which the host resolves with kseq's own state machine (dd_io.h: fastq_to_fasta).
// Round 4 therefore sent every .gz whose text starts with '@' to the HOST decoder, inflate and all.  But nearly every FASTQ
// file there is has the plain four-line form -- '@' header, ONE sequence line, '+' line, ONE quality line of the same length --
// and for that form kseq's reading is a statement about LINES:
//     line 4r     starts with '@'                        -> header
//     line 4r + 1 does not start with '>', '@' or '+'    -> the record's sequence
//     line 4r + 2 starts with '+'                        -> skipped to its end
//     line 4r + 3 is as long as line 4r + 1              -> the quality text: consumed whole, whatever it holds
// (lengths after kseq's "one '\r' in front of the line end goes").  So: the positions of the text's newlines are compacted into
// an array (count per 4 KiB, scan, write), one thread per record checks the four conditions, and the first byte of the '+' line
// and of the quality line is overwritten with '>' -- a header line to K0, skipped to its end.  Nothing is copied, nothing moves.
// A text that is not of this form -- multi-line FASTQ, a truncated record, FASTA behind '@' headers, anything kseq would read
// differently -- raises kNotFourLine and the call is run again through the host, whose kseq state machine has the last word.
// The other direction is checked too (advisor, round 4): a text classed as FASTA by its first 256 bytes is scanned for a line
// that starts with '+' (dd_io.h: has_plus_line over the WHOLE text, as the host path does); if there is one, kNotFourLine.
#include "dd_common.h"
#include "dd_kernels.h"

namespace dd {
namespace {

constexpr uint32_t kTextBlock = 4096;   // bytes of text per workgroup of 256 threads: 16 per thread

DD_D uint32_t exact_newline_mask(const uint4& v) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t m = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int b = 0; b < 4; ++b) m |= (((w[j] >> (8 * b)) & 0xFFu) == 0x0Au ? 1u : 0u) << (4 * j + b);
    return m;
}
DD_D uint4 load_text16(const TextJob& t, uint32_t off) {   // 16 bytes at `off` (a multiple of 16); past the text: zeros
    if (off + 16u <= t.n) return gload16(t.text + off);
    uint32_t w[4] = {0, 0, 0, 0};
    for (uint32_t i = off; i < t.n; ++i) w[(i - off) >> 2] |= (uint32_t)t.text[i] << (8u * ((i - off) & 3u));
    return make_uint4(w[0], w[1], w[2], w[3]);
}
DD_D const TextJob& job_of(const TextJob* jobs, int njobs, uint32_t block, uint32_t& local) {
    int f = 0;
    while (f + 1 < njobs && block >= jobs[f + 1].block0) ++f;
    local = block - jobs[f].block0;
    return jobs[f];
}

// FASTA-classed texts: a line that starts with '+' means the first 256 bytes lied.  FASTQ-classed texts: newlines per block.
__global__ __launch_bounds__(256) void text_scan_kernel(const TextJob* __restrict__ jobs, int njobs, uint32_t* __restrict__ errors) {
    uint32_t local;
    const TextJob& t = job_of(jobs, njobs, blockIdx.x, local);
    const uint32_t off = local * kTextBlock + threadIdx.x * 16u;
    uint32_t nl = 0;
    bool plus = false;
    if (off < t.n) {
        const uint4 v = load_text16(t, off);
        nl = exact_newline_mask(v);
        if (!t.fastq) {
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            uint32_t prev = off ? (uint32_t)t.text[off - 1] : 0x0Au;   // (the text's first byte counts as a line start)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const uint32_t c = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
                plus |= off + (uint32_t)i < t.n && c == '+' && prev == 0x0Au;
                prev = c;
            }
        }
    }
    if (!t.fastq) {
        if (__any(plus) && (threadIdx.x & 63u) == 0u) atomicOr(errors, kNotFourLine);
        return;
    }
    __shared__ uint32_t part[4];
    uint32_t c = (uint32_t)__popc(nl);
    for (int d = 32; d; d >>= 1) c += (uint32_t)__shfl_down((int)c, d);
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) t.blk_count[local] = part[0] + part[1] + part[2] + part[3];
}

// one workgroup per FASTQ text: exclusive scan of its blocks' newline counts (in place), the total to *nl_total
__global__ __launch_bounds__(1024) void text_offsets_kernel(const TextJob* __restrict__ jobs, int njobs) {
    const TextJob& t = jobs[blockIdx.x];
    if (!t.fastq) return;
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    const uint32_t nblocks = (t.n + kTextBlock - 1u) / kTextBlock, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < nblocks; b0 += 1024u) {
        const uint32_t i = b0 + threadIdx.x, mine = i < nblocks ? t.blk_count[i] : 0u;
        uint32_t incl = mine;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
            if ((int)lane >= d) incl += up;
        }
        if (lane == 63u) wsum[wave] = incl;
        __syncthreads();
        uint32_t before = carry_s;
        for (uint32_t w = 0; w < wave; ++w) before += wsum[w];
        if (i < nblocks) t.blk_count[i] = before + incl - mine;
        __syncthreads();
        if (threadIdx.x == 1023u) carry_s = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *t.nl_total = carry_s;
}

// the positions of a FASTQ text's newlines, in order: nl[offset of the block + rank inside it]
__global__ __launch_bounds__(256) void text_newlines_kernel(const TextJob* __restrict__ jobs, int njobs, uint32_t* __restrict__ errors) {
    uint32_t local;
    const TextJob& t = job_of(jobs, njobs, blockIdx.x, local);
    if (!t.fastq) return;
    const uint32_t off = local * kTextBlock + threadIdx.x * 16u;
    const uint32_t nl = off < t.n ? exact_newline_mask(load_text16(t, off)) : 0u;
    __shared__ uint32_t wsum[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, mine = (uint32_t)__popc(nl);
    uint32_t incl = mine;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
        if ((int)lane >= d) incl += up;
    }
    if (lane == 63u) wsum[wave] = incl;
    __syncthreads();
    uint32_t at = t.blk_count[local] + incl - mine;
    for (uint32_t w = 0; w < wave; ++w) at += wsum[w];
    for (uint32_t m = nl; m; m &= m - 1u, ++at) {
        if (at < t.nl_cap) t.nl[at] = off + (uint32_t)__builtin_ctz(m);
        else if (at == t.nl_cap) atomicOr(errors, kNotFourLine);   // (more lines than a FASTQ text of this size has: lines of < 8 bytes on average)
    }
}

// one thread per four lines
__global__ __launch_bounds__(256) void fastq_records_kernel(const TextJob* __restrict__ jobs, int njobs, uint32_t* __restrict__ errors) {
    uint32_t local;
    const TextJob& t = job_of(jobs, njobs, blockIdx.x, local);   // (block0 / blocks in units of kTextBlock bytes of text: >= one thread per 16 bytes, a record has more)
    if (!t.fastq) return;
    const uint32_t count = *t.nl_total;
    if (count > t.nl_cap) return;                                  // (reported by text_newlines_kernel)
    const uint32_t nlines = count + ((t.n && t.text[t.n - 1] != 0x0Au) ? 1u : 0u);
    const uint32_t r = local * 256u + threadIdx.x;
    if (r == 0u && (nlines == 0u || (nlines & 3u) != 0u)) atomicOr(errors, kNotFourLine);
    if (4u * r + 3u >= nlines) return;
    auto start = [&](uint32_t i) { return i ? t.nl[i - 1] + 1u : 0u; };
    auto end = [&](uint32_t i) { return i < count ? t.nl[i] : t.n; };
    auto len = [&](uint32_t s, uint32_t e) {   // kseq: one '\r' in front of the line end goes (accumulated length > 1)
        const uint32_t l = e - s;
        return (l > 1u && t.text[e - 1] == '\r') ? l - 1u : l;
    };
    const uint32_t h = start(4u * r), s = start(4u * r + 1u), se = end(4u * r + 1u), p = start(4u * r + 2u), pe = end(4u * r + 2u), q = start(4u * r + 3u), qe = end(4u * r + 3u);
    bool ok = end(4u * r) > h && t.text[h] == '@';
    ok = ok && pe > p && t.text[p] == '+';
    if (se > s) {
        const uint8_t c = t.text[s];
        ok = ok && c != '>' && c != '@' && c != '+';
    }
    ok = ok && len(s, se) == len(q, qe);
    if (!ok) {
        atomicOr(errors, kNotFourLine);
        return;
    }
    t.text[p] = '>';
    if (qe > q) t.text[q] = '>';
}

}  // namespace

// texts of one batch: FASTA-classed ones are scanned for a '+' line, FASTQ-classed ones get their '+' and quality lines turned into
// header lines; *errors_dev |= kNotFourLine when a text is not what it was classed as (the caller goes to the host)
void launch_text_rules(const TextJob* jobs_dev, int njobs, uint32_t nblocks, bool any_fastq, uint32_t* errors_dev, hipStream_t st) {
    if (njobs <= 0 || !nblocks) return;
    hipLaunchKernelGGL(text_scan_kernel, dim3(nblocks), dim3(256), 0, st, jobs_dev, njobs, errors_dev);
    if (!any_fastq) return;
    hipLaunchKernelGGL(text_offsets_kernel, dim3((unsigned)njobs), dim3(1024), 0, st, jobs_dev, njobs);
    hipLaunchKernelGGL(text_newlines_kernel, dim3(nblocks), dim3(256), 0, st, jobs_dev, njobs, errors_dev);
    // (a record is at least four lines = four bytes: one thread per 16 bytes of text is more than enough; blocks of 256 records
    // are counted in the same units as the text's 4 KiB blocks)
    hipLaunchKernelGGL(fastq_records_kernel, dim3(nblocks), dim3(256), 0, st, jobs_dev, njobs, errors_dev);
}

}  // namespace dd
