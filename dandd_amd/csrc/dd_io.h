// dd_io.h -- host-side file ingestion of libdandd_hip: whole FASTA files (plain or gzip, as DandD's species
// directories hold them, /root/reference/lib/species_specifics.py:93) into reusable host buffers.  Host code only
// (no kernels): tests/native/sanitize_host.cpp builds it with AddressSanitizer / ThreadSanitizer on the CPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <zlib.h>

#include <algorithm>
#include <string>

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

// Whole FASTA file into memory; gzip (any number of members) or plain, decided by zlib itself.
inline bool read_fasta_file(const char* path, FileBuf& out, std::string& err) {
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
