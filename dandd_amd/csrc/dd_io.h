// dd_io.h -- host-side file ingestion of libdandd_hip: whole FASTA files (plain or gzip, as DandD's species
// directories hold them, /root/reference/lib/species_specifics.py:93) into reusable host buffers.  Host code only
// (no kernels): tests/native/sanitize_host.cpp builds it with AddressSanitizer / ThreadSanitizer on the CPU.
#pragma once
#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <algorithm>
#include <string>
#include <vector>

#include "dd_inflate.h"

namespace dd {

// Host bytes of one file.  Pageable flavour: malloc storage in 2 MiB-aligned blocks the kernel may back
// with huge pages (512x fewer page faults for a 250 MB .. 3 GB buffer).  Pinned flavour (the ingestion
// pipeline's pool): hipHostMalloc storage, so the H2D copy of a loaded file is a true asynchronous DMA at
// PCIe speed; pinning is slow, which is why those buffers live in the context and are reused from file
// to file and from call to call.  Contents are carried over when a buffer grows.
struct FileBuf {
    uint8_t* p = nullptr;
    size_t len = 0, cap = 0;
    bool pinned = false;
    FileBuf() = default;
    FileBuf(const FileBuf&) = delete;
    FileBuf& operator=(const FileBuf&) = delete;
    ~FileBuf() { release(); }
    void release() {
        if (p) {
            if (pinned) (void)hipHostFree(p);
            else free(p);
        }
        p = nullptr;
        len = cap = 0;
    }
    bool reserve(size_t n) {
        if (n <= cap) return true;
        const size_t want = (n + kHuge - 1) / kHuge * kHuge;
        void* q = nullptr;
        if (pinned) {
            if (hipHostMalloc(&q, want, hipHostMallocDefault) != hipSuccess || !q) {
                (void)hipGetLastError();
                return false;
            }
        } else {
            if (posix_memalign(&q, kHuge, want) != 0 || !q) return false;
            (void)madvise(q, want, MADV_HUGEPAGE);
        }
        if (len) memcpy(q, p, len);
        if (p) {
            if (pinned) (void)hipHostFree(p);
            else free(p);
        }
        p = static_cast<uint8_t*>(q);
        cap = want;
        return true;
    }
    static constexpr size_t kHuge = (size_t)2 << 20;
    const uint8_t* data() const { return p; }
    size_t size() const { return len; }
};

// libdeflate, when the machine has it (this image ships libdeflate.so.0 as somebody's dependency, without
// headers): whole-buffer gzip decoding 2-3x faster than zlib's streaming inflate, which is what bounds a
// directory of .fa.gz genomes (zlib: ~190 MB/s of FASTA per loader thread).  Looked up once with dlopen, the four
// entry points declared here as its stable C ABI has them; absent or failing, zlib does the work as before.
struct Deflate {
    void* (*alloc)() = nullptr;
    void (*release)(void*) = nullptr;
    int (*gunzip_ex)(void*, const void*, size_t, void*, size_t, size_t*, size_t*) = nullptr;  // 0 ok, 1 bad data, 3 no room
    uint32_t (*crc32)(uint32_t, const void*, size_t) = nullptr;   // carry-less-multiply CRC-32: ~10x zlib 1.2's
    static const Deflate& get() {
        static const Deflate d = [] {
            Deflate r;
            if (getenv("DD_NO_LIBDEFLATE")) return r;
            void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
            if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
            if (!h) return r;
            r.alloc = reinterpret_cast<void* (*)()>(dlsym(h, "libdeflate_alloc_decompressor"));
            r.release = reinterpret_cast<void (*)(void*)>(dlsym(h, "libdeflate_free_decompressor"));
            r.gunzip_ex = reinterpret_cast<int (*)(void*, const void*, size_t, void*, size_t, size_t*, size_t*)>(
                dlsym(h, "libdeflate_gzip_decompress_ex"));
            r.crc32 = reinterpret_cast<uint32_t (*)(uint32_t, const void*, size_t)>(dlsym(h, "libdeflate_crc32"));
            if (!r.alloc || !r.release || !r.gunzip_ex) r = Deflate();
            return r;
        }();
        return d;
    }
};

// one whole gzip member with zlib (the machine has no libdeflate): false on any error
inline bool zlib_gunzip_member(const uint8_t* in, size_t n, uint8_t* dst, size_t cap, size_t* used, size_t* made) {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 15 + 16) != Z_OK) return false;
    zs.next_in = const_cast<Bytef*>(in);
    zs.next_out = dst;
    size_t in_left = n, out_left = cap;
    int rc = Z_OK;
    while (rc == Z_OK) {
        zs.avail_in = (uInt)std::min<size_t>(in_left, (size_t)1 << 30);
        zs.avail_out = (uInt)std::min<size_t>(out_left, (size_t)1 << 30);
        const uInt ai = zs.avail_in, ao = zs.avail_out;
        rc = inflate(&zs, Z_NO_FLUSH);
        in_left -= ai - zs.avail_in;
        out_left -= ao - zs.avail_out;
        if (rc == Z_OK && ai == zs.avail_in && ao == zs.avail_out) break;   // no progress: out of input or room
    }
    inflateEnd(&zs);
    if (rc != Z_STREAM_END) return false;
    *used = n - in_left;
    *made = cap - out_left;
    return true;
}

// A gzip file (any number of members) decoded in memory: BGZF blocks and large single members in parallel over `par`
// threads (dd_inflate.h), everything else member by member through libdeflate (zlib when the machine has none).
// Returns 1 done, 0 not applicable (not gzip, anything unexpected: the caller's zlib streaming path then decides and
// words the error), -1 out of memory.
inline int read_gzip_whole(const char* path, FileBuf& out, int par = 1) {
    const Deflate& lib = Deflate::get();
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return 0;
    struct stat sb;
    if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 18) {
        close(fd);
        return 0;
    }
    const size_t n = (size_t)sb.st_size;
    uint8_t magic[2] = {0, 0};
    if (pread(fd, magic, 2, 0) != 2 || magic[0] != 0x1f || magic[1] != 0x8b) {  // a plain file: nothing to do here
        close(fd);
        return 0;
    }
    // the compressed bytes are only ever read: map the file (page cache / tmpfs pages, no copy); read() as a fallback
    bool mapped = true;
    uint8_t* in = static_cast<uint8_t*>(mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0));   // (not MAP_POPULATE: one thread would fill the page table that the decoding threads fill in parallel)
    size_t have = n;
    if (in == MAP_FAILED) {
        mapped = false;
        in = static_cast<uint8_t*>(malloc(n));
        if (!in) {
            close(fd);
            return 0;
        }
        have = 0;
        while (have < n) {
            const ssize_t got = pread(fd, in + have, n - have, (off_t)have);
            if (got <= 0) break;
            have += (size_t)got;
        }
    }
    close(fd);
    auto drop_input = [&] {
        if (mapped) munmap(in, n);
        else free(in);
    };
    int rc = 0;
    void* d = nullptr;
    if (have == n && in[0] == 0x1f && in[1] == 0x8b) {
        out.len = 0;
        if (par > 1 && !getenv("DD_NO_PARALLEL_GZIP")) {
            // bgzip: independent blocks; a big plain member: pieces decoded without their history, then resolved
            rc = gunzip_bgzf_parallel(in, n, out, par, [&lib](const uint8_t* src, size_t len, uint8_t* dst, size_t cap, size_t* used, size_t* made) {
                if (lib.gunzip_ex) {
                    void* dd = lib.alloc();
                    if (!dd) return false;
                    const int r = lib.gunzip_ex(dd, src, len, dst, cap, used, made);
                    lib.release(dd);
                    return r == 0;
                }
                return zlib_gunzip_member(src, len, dst, cap, used, made);
            });
            if (rc == 0 && par >= 3) rc = gunzip_member_parallel(in, n, out, par, (size_t)4 << 20, lib.crc32);
            if (rc != 0) {
                drop_input();
                if (rc != 1) out.len = 0;
                return rc;
            }
        }
        if (lib.gunzip_ex) d = lib.alloc();
    }
    if (have == n && in[0] == 0x1f && in[1] == 0x8b && (d || !lib.gunzip_ex)) {
        // ISIZE of the last member: the whole size of a one-member file, a lower bound otherwise
        const size_t isize = (size_t)in[n - 4] | ((size_t)in[n - 3] << 8) | ((size_t)in[n - 2] << 16) | ((size_t)in[n - 1] << 24);
        out.len = 0;
        rc = out.reserve(std::max(out.cap, std::max(isize + 64, 3 * n))) ? 1 : -1;
        size_t pos = 0;
        while (rc == 1 && pos < n) {
            if (n - pos < 18 || in[pos] != 0x1f || in[pos + 1] != 0x8b) {
                // zeros / garbage after the last member are ignored, as gzread ignores them; anything else is zlib's call
                if (!out.len) rc = 0;
                break;
            }
            size_t used = 0, made = 0;
            if (d) {
                const int r = lib.gunzip_ex(d, in + pos, n - pos, out.p + out.len, out.cap - out.len, &used, &made);
                if (r == 0) {
                    out.len += made;
                    pos += used;
                } else if (r == 3) {
                    if (!out.reserve(out.cap * 2)) rc = -1;
                } else {
                    rc = 0;  // bad data: let zlib find and word it
                }
            } else {
                rc = 0;      // no libdeflate: the streaming zlib path below reads the file as before
            }
        }
        if (d) lib.release(d);
    }
    drop_input();
    if (rc != 1) out.len = 0;
    return rc;
}

inline bool read_file_bytes(const char* path, FileBuf& out, std::string& err, int par);

// ---- FASTQ ---------------------------------------------------------------------------------------------------------
// `dashing sketch` reads its inputs through klib's kseq.h, which takes FASTA and FASTQ alike
// (/root/reference/lib/sketch_classes.py:358-365 hands it whatever the species directory holds).  K0's line machine
// knows kseq's FASTA rules; a FASTQ record's quality text is a length-counted field that a parallel tokenizer cannot
// delimit, so buffers that hold a '+' line are rewritten on the host -- by kseq's own state machine, as recalled
// (oracle/POLICIES.md P10) -- into the FASTA K0 reads: one ">" line and one sequence line per record.
inline bool has_plus_line(const uint8_t* p, size_t n) {
    if (n && p[0] == '+') return true;
    for (const uint8_t* q = p; n && (q = static_cast<const uint8_t*>(memchr(q, '+', (size_t)(p + n - q)))) != nullptr; ++q)
        if (q > p && q[-1] == '\n') return true;
    return false;
}
// the same question for one piece [off, off + len) of a buffer that is being read in pieces by several threads: a '+'
// at the piece's first byte is NOT looked at (the byte before it may not be there yet: plus_at_piece_start)
inline bool piece_has_plus_line(const uint8_t* p, size_t off, size_t len) {
    if (len && off == 0 && p[0] == '+') return true;
    const uint8_t* end = p + off + len;
    for (const uint8_t* q = p + off + 1; q < end && (q = static_cast<const uint8_t*>(memchr(q, '+', (size_t)(end - q)))) != nullptr; ++q)
        if (q[-1] == '\n') return true;
    return false;
}
inline bool plus_at_piece_start(const uint8_t* p, size_t off) { return off > 0 && p[off] == '+' && p[off - 1] == '\n'; }
// src[0..n) -> dst (at least n + 2 bytes; may be src itself: the output never runs ahead of the input by more than two
// bytes, which memmove tolerates); returns the bytes written
inline size_t fastq_to_fasta(const uint8_t* src, size_t n, uint8_t* dst) {
    size_t i = 0, o = 0;
    bool have_header = false;  // kseq's last_char: the next record's header character has been consumed
    for (;;) {
        if (!have_header) {
            while (i < n && src[i] != '>' && src[i] != '@') ++i;
            if (i >= n) break;
            ++i;
        }
        have_header = false;
        while (i < n && src[i] != '\n') ++i;
        const bool header_closed = i < n;
        if (i < n) ++i;
        // (in place: everything read so far is at least as long as what goes out, except for a header cut short by the
        // end of the buffer -- hence the two bytes of slack)
        const size_t o_rec = o;   // (a truncated FASTQ record is taken back: below)
        dst[o++] = '>';
        if (header_closed || i < n) dst[o++] = '\n';
        size_t seq_len = 0;
        int c = -1;
        while (i < n) {
            c = src[i];
            if (c == '>' || c == '+' || c == '@') break;
            if (c == '\n') {
                ++i;
                c = -1;
                continue;
            }
            size_t e = i;
            while (e < n && src[e] != '\n') ++e;
            size_t len = e - i;
            if (len && src[e - 1] == '\r' && seq_len + len > 1) --len;  // one '\r' in front of the line end goes
            memmove(dst + o, src + i, len);
            // (sequence lines are joined: a '>' or '@' that was in the middle of a line must not end up at a line start)
            o += len;
            seq_len += len;
            i = e < n ? e + 1 : e;
            c = -1;
        }
        if (seq_len) dst[o++] = '\n';
        if (i >= n) break;
        if (c == '>' || c == '@') {
            ++i;
            have_header = true;
            continue;
        }
        // kseq_read's -2: the input ends inside the '+' line, or the quality text is not exactly as long as the sequence --
        // `while (kseq_read(ks) >= 0)` then drops this record and everything behind it; one quality line is read even for
        // an empty sequence (kseq tests the length after the read).  oracle/POLICIES.md P10, oracle/dd_oracle.c: orc_records.
        while (i < n && src[i] != '\n') ++i;  // the rest of the '+' line
        if (i >= n) return o_rec;
        ++i;
        size_t qual_len = 0;
        do {
            if (i >= n) break;
            size_t e = i;
            while (e < n && src[e] != '\n') ++e;
            size_t len = e - i;
            if (len && src[e - 1] == '\r' && qual_len + len > 1) --len;
            qual_len += len;
            i = e < n ? e + 1 : e;
        } while (qual_len < seq_len);
        if (qual_len != seq_len) return o_rec;
    }
    return o;
}
// a file's bytes as K0 wants them; false: out of memory
inline bool normalize_records(FileBuf& fb, int known_plus = -1) {   // known_plus: 0 / 1 = the caller has looked already
    if (known_plus == 0 || (known_plus < 0 && !has_plus_line(fb.p, fb.len))) return true;
    if (!fb.reserve(fb.len + 16)) return false;
    fb.len = fastq_to_fasta(fb.p, fb.len, fb.p);
    return true;
}

// Whole FASTA (or FASTQ) file into memory; gzip (any number of members) or plain, decided by zlib itself.
inline bool read_fasta_file(const char* path, FileBuf& out, std::string& err, int par = 1) {
    if (!read_file_bytes(path, out, err, par)) return false;
    if (!normalize_records(out)) {
        err = std::string("out of host memory reading ") + path;
        return false;
    }
    return true;
}
// CRC-32 of a concatenation from the CRCs of its parts (zlib's crc32_combine; this image's zlib 1.2.11 has no
// crc32_combine_gen, and its crc32_combine squares a 32 x 32 matrix per call: ~3 us x 7 700 chunks held a batch back 20 ms).
// Reflected polynomial arithmetic as in zlib >= 1.2.12: a(x) b(x) mod P, P = 0xedb88320, bit 31 = x^0.
inline uint32_t crc_multmodp(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (uint32_t m = 0x80000000u;; m >>= 1) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1u)) == 0u) break;
        }
        b = (b >> 1) ^ ((b & 1u) ? 0xedb88320u : 0u);
    }
    return p;
}
inline uint32_t crc_x8n(uint32_t nbytes) {   // x^(8 nbytes) mod P
    uint32_t p = 0x80000000u, sq = 0x40000000u;   // x^0, x^1
    for (int i = 0; i < 3; ++i) sq = crc_multmodp(sq, sq);   // x^8
    for (; nbytes; nbytes >>= 1, sq = crc_multmodp(sq, sq))
        if (nbytes & 1u) p = crc_multmodp(sq, p);
    return p;
}

// The gzip members of a file for the device decoder (dd_ginflate.hip): where each one's deflate data starts, where it ends,
// its trailer's CRC-32 and ISIZE.  `gzip` and the sequence archives write ONE member; `cat a.fa.gz b.fa.gz` (and pigz -i,
// and a gzip run that appended) several: a member says nothing about its length, so the next one is looked for by its header
// -- 1f 8b 08, a flag byte without reserved bits, an XFL and an OS byte of the values RFC 1952 knows, optional fields that
// fit -- and the eight bytes in front of it are taken for the trailer of the one before.  A header look-alike inside
// compressed data (~5 per 10^12 bytes) makes a member that does not end at its final block: the device refuses it and the
// host reads the file.  false: not a file that path takes (not gzip, a header it does not know, an ISIZE that cannot be that
// member's, small or very many members, a '+' line in the first bytes of a text that does not start with '@'): the host
// decoder reads it.  Whether every member really ends where the next header stands only the decoding shows.
struct GzMember {
    uint64_t first_bit = 0;   // of the FILE: the member's deflate data starts there
    size_t end = 0;           // byte offset in the file behind the member's trailer
    uint32_t isize = 0, crc = 0;
    bool fastq = false;       // (first member) the text starts with '@': four-line FASTQ is expected and checked on the device (dd_fastq.hip)
};
inline size_t gzip_header_bytes(const uint8_t* p, size_t n) {   // 0: not a member header this path knows
    if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xE0)) return 0;
    size_t h = 10;
    if (p[3] & 4) {
        if (h + 2 > n) return 0;
        h += 2 + ((size_t)p[h] | ((size_t)p[h + 1] << 8));
    }
    for (int bit : {8, 16})
        if (p[3] & bit) {
            while (h < n && p[h]) ++h;
            ++h;
        }
    if (p[3] & 2) h += 2;
    return h + 8 <= n ? h : 0;
}
// the positions q in [lo, hi) with p[q .. q + 2] == 1f 8b 08 (hi + 2 <= the buffer's length), appended in order.  16 positions
// per step (SSE2: three compares, ~8 GB/s; memchr on one byte stops every 256 bytes, glibc's memmem runs at ~1.5 GB/s -- a
// 116 MB member was held back 60 ms), because every .gz pays this scan for the rare file that has a second member.
inline void gzip_magic_scan(const uint8_t* p, size_t lo, size_t hi, std::vector<size_t>& out) {
    size_t q = lo;
#if defined(__SSE2__)
    const __m128i a = _mm_set1_epi8(0x1f), b = _mm_set1_epi8((char)0x8b), c = _mm_set1_epi8(0x08);
    for (; q + 16 <= hi; q += 16) {
        const __m128i x = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + q)), y = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + q + 1)),
                      z = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + q + 2));
        unsigned m = (unsigned)_mm_movemask_epi8(_mm_and_si128(_mm_and_si128(_mm_cmpeq_epi8(x, a), _mm_cmpeq_epi8(y, b)), _mm_cmpeq_epi8(z, c)));
        for (; m; m &= m - 1) out.push_back(q + (size_t)__builtin_ctz(m));
    }
#endif
    for (; q < hi; ++q)
        if (p[q] == 0x1f && p[q + 1] == 0x8b && p[q + 2] == 0x08) out.push_back(q);
}
// `magic`: the sorted positions of 1f 8b 08 in the file when the caller has scanned it already (in pieces, by several threads)
inline bool gzip_members_parse(const uint8_t* p, size_t n, std::vector<GzMember>& ms, const std::vector<size_t>* magic = nullptr) {
    ms.clear();
    if (n < 64) return false;
    std::vector<size_t> scanned;
    if (!magic) {
        gzip_magic_scan(p, 0, n - 17, scanned);
        magic = &scanned;
    }
    size_t mi = 0;
    uint64_t text_total = 0;
    for (size_t start = 0; start < n;) {
        const size_t h = gzip_header_bytes(p + start, n - start);
        if (!h) return false;
        size_t next = n;   // where the next member's header stands
        for (; mi < magic->size(); ++mi) {
            const size_t q = (*magic)[mi];
            if (q < start + h + 10 || q + 18 > n) continue;
            const uint8_t* r = p + q;
            if (!(r[3] & 0xE0) && (r[8] == 0 || r[8] == 2 || r[8] == 4) && (r[9] <= 13 || r[9] == 255) &&
                gzip_header_bytes(r, n - q)) {
                next = q;
                break;
            }
        }
        GzMember gm;
        const uint8_t* t = p + next - 8;
        const size_t len = next - start;
        gm.first_bit = 8ull * (start + h);
        gm.end = next;
        gm.crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        gm.isize = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
        if (gm.isize < len / 2 || (size_t)gm.isize > (size_t)1032 * len) return false;   // (not this member's length: a header look-alike, or damage)
        // (ISIZE is the text's length mod 2^32: a large member whose ISIZE says "inflates less than 2 x" is far more likely text
        // beyond 4 GiB -- FASTA never compresses that badly -- and the device path's offsets are 32-bit)
        if (len >= ((size_t)256 << 20) && (size_t)gm.isize < 2 * len) return false;
        text_total += gm.isize;
        ms.push_back(gm);
        start = next;
        // (a file of many small members -- a blocked format this path does not know -- is not worth a piece table per member)
        if (ms.size() > 64 || (next < n && len < ((size_t)64 << 10))) return false;
    }
    if (ms.empty() || text_total >= ((uint64_t)1 << 32) - 65536) return false;
    // FASTQ: look at the first bytes of text
    uint8_t first[256];
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 15 + 16) != Z_OK) return false;
    zs.next_in = const_cast<uint8_t*>(p);
    zs.avail_in = (uInt)std::min<size_t>(n, 1 << 16);
    zs.next_out = first;
    zs.avail_out = sizeof first;
    const int zr = inflate(&zs, Z_SYNC_FLUSH);
    const size_t made = sizeof first - zs.avail_out;
    inflateEnd(&zs);
    if ((zr != Z_OK && zr != Z_STREAM_END) || !made) return false;
    // a text that starts with '@' is taken for four-line FASTQ -- the device checks every record and resolves it in place
    // (dd_fastq.hip; anything else comes back to the host's kseq state machine) --, unless DD_NO_GPU_FASTQ sends it to the host
    // from the start; a '+' line among the first bytes of a text that does not start with '@' is for the host as well
    ms[0].fastq = first[0] == '@';
    if (ms[0].fastq ? getenv("DD_NO_GPU_FASTQ") != nullptr : has_plus_line(first, made)) return false;
    return true;
}
// ONE member (the tests' and the sanitizer harness's entry)
inline bool gzip_member_parse(const uint8_t* p, size_t n, GzMember& gm) {
    std::vector<GzMember> ms;
    if (!gzip_members_parse(p, n, ms) || ms.size() != 1) return false;
    gm = ms[0];
    return true;
}

inline bool read_file_bytes(const char* path, FileBuf& out, std::string& err, int par) {
    const int fast = read_gzip_whole(path, out, par);
    if (fast == 1) return true;
    if (fast < 0) {
        err = std::string("out of host memory reading ") + path;
        return false;
    }
    gzFile f = gzopen(path, "rb");
    if (!f) {
        err = std::string("cannot open ") + path;
        return false;
    }
    gzbuffer(f, 1u << 20);
    // size hint: a plain file is read in one piece; a compressed one usually inflates ~4x
    size_t hint = 1u << 22;
    struct stat sb;
    if (stat(path, &sb) == 0 && sb.st_size > 0) hint = (size_t)sb.st_size + 1;
    out.len = 0;
    if (!out.reserve(std::max(out.cap, hint))) {
        err = std::string("out of host memory reading ") + path;
        gzclose(f);
        return false;
    }
    for (;;) {
        if (out.len == out.cap && !out.reserve(out.cap * 2)) {
            err = std::string("out of host memory reading ") + path;
            gzclose(f);
            return false;
        }
        const unsigned want = (unsigned)std::min<size_t>(out.cap - out.len, 1u << 30);
        const int got = gzread(f, out.p + out.len, want);
        if (got < 0) {
            int code = 0;
            err = std::string("read error on ") + path + ": " + gzerror(f, &code);
            gzclose(f);
            return false;
        }
        if (got == 0) break;
        out.len += (size_t)got;
    }
    gzclose(f);
    return true;
}


}  // namespace dd
