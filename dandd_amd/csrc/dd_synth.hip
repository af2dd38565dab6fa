// dd_synth.hip -- "realistic" synthetic genomes on the device (bench / tests only; BASELINE.md section 4 defines the
// uniform generator, this is its hard-case sibling): GC 35 %, 20 % interspersed repeats (copies of 64 elements of
// 300..6000 bases), 10 % tandem repeats (units of 2..60 bases), repeats soft-masked, 2 % N in 100-base runs, contigs
// of 2..200 kbp, 1 % divergence between the genomes of one seed.  Byte-identical to
// oracle/dd_oracle.c:orc_synth_realistic_fasta (tests/test_gpu_parity.py); counter-based, one thread per output byte.
#include "dd_common.h"
#include "dd_kernels.h"

#include <vector>

namespace dd {
namespace {

constexpr uint64_t RS_HDR = 16, RS_LINE = 80;

DD_HD uint64_t real_contig_len(uint64_t seed, uint64_t idx) {
    const uint64_t h = splitmix64(seed ^ 0xC047160000000000ull ^ idx);
    const uint64_t base = 2000ull << ((h >> 40) % 7);
    const uint64_t len = base + h % base;
    return len < 200000 ? len : 200000;
}

DD_D uint8_t real_base(uint64_t seed, uint64_t seed_g, uint64_t g) {
    const uint64_t blk = g >> 9;
    const uint64_t hb = splitmix64(seed ^ 0x5EED5EED00000000ull ^ blk);
    const uint32_t kind = (uint32_t)(hb % 100);
    uint64_t r;
    if (kind < 20) {          // a copy of one of 64 repeat elements, entered at a block-specific offset
        const uint64_t e = (hb >> 8) & 63;
        const uint64_t elen = 300 + splitmix64(seed ^ 0xE1E100000000ull ^ e) % 5700;
        const uint64_t off = ((hb >> 16) % elen + (g & 511)) % elen;
        r = splitmix64(seed ^ ((0xABCD0000ull + e) << 32) ^ off);
    } else if (kind < 30) {   // a tandem repeat: the block repeats a unit of 2..60 bases
        const uint64_t u = 2 + (hb >> 8) % 59;
        r = splitmix64(seed ^ 0x7A7A000000000000ull ^ (blk << 8) ^ ((g & 511) % u));
    } else {
        r = splitmix64(seed ^ g);
    }
    const uint32_t hi = (uint32_t)((r >> 32) & 1);
    uint32_t b = (r % 100) < 35 ? 1 + hi : 3 * hi;   // GC 35 %
    const uint64_t rg = splitmix64(seed_g ^ g);
    if (rg % 100 == 0) b = (b + 1 + (uint32_t)((rg >> 32) % 3)) & 3;
    uint8_t ch = (uint8_t)("ACGT"[b]);
    if (splitmix64(seed_g ^ 0x4E4E4E4E00000000ull ^ (g / 100)) % 50 == 0) ch = 'N';
    if (kind < 30) ch |= 0x20;
    return ch;
}

// tab[2c] = first byte of contig c in the file, tab[2c + 1] = its first base; tab[2 ncontigs] = total bytes, [.. + 1] = nbases
__global__ __launch_bounds__(256) void synth_realistic_kernel(uint64_t seed, uint64_t seed_g, int gi, const uint64_t* __restrict__ tab,
                                                              uint32_t ncontigs, uint64_t total, uint8_t* __restrict__ out) {
    for (uint64_t off = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; off < total; off += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t lo = 0, hi = ncontigs;   // last contig whose first byte is <= off
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (tab[2 * (size_t)mid] <= off) lo = mid;
            else hi = mid;
        }
        const uint64_t o = off - tab[2 * (size_t)lo];
        const uint64_t L = tab[2 * (size_t)lo + 3] - tab[2 * (size_t)lo + 1];
        uint8_t ch;
        if (o < RS_HDR) {
            const char hexd[] = "0123456789abcdef";
            if (o == 0) ch = '>';
            else if (o == 1) ch = 'g';
            else if (o < 6) ch = hexd[(gi >> (12 - 4 * (int)(o - 2))) & 15];
            else if (o == 6) ch = '.';
            else if (o == 7) ch = 'r';
            else if (o < 12) ch = hexd[(lo >> (12 - 4 * (int)(o - 8))) & 15];
            else if (o < 15) ch = ' ';
            else ch = '\n';
        } else {
            const uint64_t q = o - RS_HDR;
            const uint64_t line = q / (RS_LINE + 1), col = q % (RS_LINE + 1);
            const uint64_t j = line * RS_LINE + col;
            if (col == RS_LINE || j >= L) ch = '\n';
            else ch = real_base(seed, seed_g, tab[2 * (size_t)lo + 1] + j);
        }
        out[off] = ch;
    }
}

}  // namespace

// (byte start, base start) of every contig + the end pair
std::vector<uint64_t> synth_realistic_table(uint64_t seed, uint64_t nbases) {
    std::vector<uint64_t> tab;
    uint64_t bytes = 0, done = 0;
    for (uint64_t c = 0; done < nbases; ++c) {
        uint64_t L = real_contig_len(seed, c);
        if (L > nbases - done) L = nbases - done;
        tab.push_back(bytes);
        tab.push_back(done);
        bytes += RS_HDR + L + (L + RS_LINE - 1) / RS_LINE;
        done += L;
    }
    tab.push_back(bytes);
    tab.push_back(done);
    return tab;
}

void launch_synth_realistic(uint64_t seed, int gi, const uint64_t* tab_dev, uint32_t ncontigs, uint64_t total, uint8_t* out_dev,
                            hipStream_t st) {
    if (!total) return;
    const uint64_t seed_g = splitmix64(seed + (uint64_t)gi + 1);
    uint64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(synth_realistic_kernel, dim3((unsigned)blocks), dim3(256), 0, st, seed, seed_g, gi, tab_dev, ncontigs, total, out_dev);
}

}  // namespace dd
