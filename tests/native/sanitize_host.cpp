// Host-only sanitizer build of the engine's CPU-side code ("race detection / sanitizers"):
//   * dd_plan.hip  -- the K1 job tables (pure host code): every (genome, k, tile) covered exactly once, for every
//                     register mode, under AddressSanitizer + UBSan;
//   * dd_io.h      -- the loader used by the ingestion pipeline: many threads reading plain / gzip / multi-member
//                     files into pooled, reused, growing buffers (the access pattern of dd_sketch_files), under
//                     ASan + UBSan and again under ThreadSanitizer.
// No device code is compiled and no HIP call is made (pinned buffers are not used here); built and run by
// tests/test_sanitizers.py with g++ -x c++.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <random>
#include <vector>

#include "dd_io.h"
#include "dd_plan.h"

namespace dd {
int sweep_max_lds_bytes() { return 160 * 1024; }  // defined next to the kernels in the real library
}

static int failures = 0;
#define CHECK(cond, ...)                      \
    do {                                      \
        if (!(cond)) {                        \
            ++failures;                       \
            fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
            fprintf(stderr, __VA_ARGS__);     \
            fprintf(stderr, "\n");            \
        }                                     \
    } while (0)

static void check_plan(int log2m, const std::vector<size_t>& sizes, int kmin, int kmax, const dd::PlanKnobs& knobs) {
    const std::vector<dd::SweepClass> classes = dd::plan_sweep(log2m, 1, sizes.data(), (int)sizes.size(), kmin, kmax, knobs);
    const size_t tile = 1024 * 64;
    const int K = kmax - kmin + 1;
    std::vector<std::vector<int>> cover(sizes.size());
    for (size_t g = 0; g < sizes.size(); ++g) cover[g].assign(((sizes[g] + tile - 1) / tile) * K, 0);
    for (const dd::SweepClass& sc : classes) {
        CHECK(sc.plan.lds_bytes >= 0 && sc.plan.lds_bytes <= 160 * 1024, "lds %d", sc.plan.lds_bytes);
        if (sc.plan.mode == dd::kBucketMode) {
            CHECK((int)sc.epoch_begin.size() == sc.plan.nepochs + 1 && sc.epoch_begin.back() == sc.jobs.size(), "epoch table");
            CHECK(sc.plan.cap_chunks > 0, "capacity");
        }
        for (const dd::SweepJob& j : sc.jobs) {
            if (j.tile_end <= j.tile_begin) continue;
            if (sc.kclass == dd::kBigmapClass && j.slice > 0) continue;  // further slices of a k's index space re-read the same tiles
            CHECK(j.genome >= 0 && j.genome < (int)sizes.size(), "genome %d", j.genome);
            const size_t nt = (sizes[j.genome] + tile - 1) / tile;
            CHECK(j.tile_end <= nt && j.kfirst >= kmin && j.kfirst + j.nk - 1 <= kmax && j.krow == j.kfirst - kmin, "job range");
            for (int kk = 0; kk < j.nk; ++kk)
                for (unsigned t = j.tile_begin; t < j.tile_end; ++t) ++cover[j.genome][(size_t)t * K + (j.kfirst - kmin + kk)];
        }
    }
    for (size_t g = 0; g < sizes.size(); ++g)
        for (int v : cover[g]) {
            if (v != 1) {
                CHECK(false, "log2m %d genome %zu: a (tile, k) is covered %d times", log2m, g, v);
                return;
            }
        }
}

static std::string write_file(const std::string& dir, int i, const std::string& body, int flavour) {
    const std::string path = dir + "/f" + std::to_string(i) + (flavour ? ".fa.gz" : ".fa");
    if (flavour == 0) {
        FILE* f = fopen(path.c_str(), "wb");
        fwrite(body.data(), 1, body.size(), f);
        fclose(f);
    } else {
        // flavour 1: one gzip member; flavour 2: two members back to back
        FILE* f = fopen(path.c_str(), "wb");
        const size_t cut = flavour == 2 ? body.size() / 3 : body.size();
        for (int part = 0; part < (flavour == 2 ? 2 : 1); ++part) {
            const size_t a = part ? cut : 0, b = part ? body.size() : cut;
            const std::string tmp = path + ".part";
            gzFile g = gzopen(tmp.c_str(), "wb1");
            gzwrite(g, body.data() + a, (unsigned)(b - a));
            gzclose(g);
            FILE* t = fopen(tmp.c_str(), "rb");
            char buf[65536];
            size_t n;
            while ((n = fread(buf, 1, sizeof buf, t)) > 0) fwrite(buf, 1, n, f);
            fclose(t);
            remove(tmp.c_str());
        }
        fclose(f);
    }
    return path;
}

static void check_loaders(const std::string& dir) {
    // the access pattern of dd_sketch_files: loaders take files in order, a bounded pool of reused buffers,
    // the consumer releases them in order
    const int nfiles = 40, nthreads = 6, window = nthreads + 2;
    std::vector<std::string> paths, bodies;
    for (int i = 0; i < nfiles; ++i) {
        std::string body = ">r" + std::to_string(i) + "\n";
        unsigned x = 12345u + i;
        const size_t n = 1000 + (size_t)i * 37 * 1000 % 700000;
        for (size_t j = 0; j < n; ++j) {
            x = x * 1664525u + 1013904223u;
            body += "ACGTN"[(x >> 24) % 5];
            if (j % 80 == 79) body += '\n';
        }
        bodies.push_back(body);
        paths.push_back(write_file(dir, i, body, i % 3));
    }
    std::vector<dd::FileBuf> pool(window);
    std::vector<int> free_bufs;
    for (int b = 0; b < window; ++b) free_bufs.push_back(b);
    struct Slot {
        int buf = -1;
        bool ok = false, done = false;
    };
    std::vector<Slot> slots(nfiles);
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<int> next{0};
    int consumed = 0;
    auto loader = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= nfiles) return;
            int b;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return i < consumed + window && !free_bufs.empty(); });
                b = free_bufs.back();
                free_bufs.pop_back();
            }
            std::string err;
            const bool ok = dd::read_fasta_file(paths[i].c_str(), pool[b], err);
            {
                std::lock_guard<std::mutex> lk(mu);
                slots[i].buf = b;
                slots[i].ok = ok;
                slots[i].done = true;
            }
            cv.notify_all();
        }
    };
    std::vector<std::thread> threads;
    for (int t = 0; t < nthreads; ++t) threads.emplace_back(loader);
    for (int i = 0; i < nfiles; ++i) {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return slots[i].done; });
        }
        const dd::FileBuf& fb = pool[slots[i].buf];
        CHECK(slots[i].ok && fb.size() == bodies[i].size() && memcmp(fb.data(), bodies[i].data(), fb.size()) == 0, "file %d read back wrong", i);
        {
            std::lock_guard<std::mutex> lk(mu);
            free_bufs.push_back(slots[i].buf);
            consumed = i + 1;
        }
        cv.notify_all();
    }
    for (auto& t : threads) t.join();
    dd::FileBuf fb;
    std::string err;
    CHECK(!dd::read_fasta_file((dir + "/missing.fa").c_str(), fb, err) && !err.empty(), "a missing file must fail");
}

static void check_gzip_edges(const std::string& dir) {
    // what zlib's gzread tolerates or refuses, the libdeflate path must tolerate or refuse the same way
    std::string body = ">e\n";
    for (int j = 0; j < 300000; ++j) body += "ACGT"[(j * 7 + j / 13) & 3];
    const std::string one = write_file(dir, 900, body, 1);
    auto slurp = [](const std::string& p) {
        std::string s;
        FILE* f = fopen(p.c_str(), "rb");
        char buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
        fclose(f);
        return s;
    };
    auto spit = [](const std::string& p, const std::string& s) {
        FILE* f = fopen(p.c_str(), "wb");
        fwrite(s.data(), 1, s.size(), f);
        fclose(f);
    };
    const std::string gz = slurp(one);
    dd::FileBuf fb;
    std::string err;
    spit(dir + "/pad.fa.gz", gz + std::string(512, '\0'));                    // zero padding after the member
    CHECK(dd::read_fasta_file((dir + "/pad.fa.gz").c_str(), fb, err) && fb.size() == body.size() && memcmp(fb.data(), body.data(), body.size()) == 0,
          "padded gzip: %s", err.c_str());
    spit(dir + "/cut.fa.gz", gz.substr(0, gz.size() / 2));                    // truncated: an error, not silence
    err.clear();
    const bool cut_ok = dd::read_fasta_file((dir + "/cut.fa.gz").c_str(), fb, err);
    CHECK(!cut_ok || fb.size() < body.size(), "truncated gzip read back whole");
    const std::string empty = write_file(dir, 901, std::string(), 1);         // an empty member
    CHECK(dd::read_fasta_file(empty.c_str(), fb, err) && fb.size() == 0, "empty gzip member");
    spit(dir + "/tiny.fa", ">t\nAC\n");                                       // shorter than any gzip file
    CHECK(dd::read_fasta_file((dir + "/tiny.fa").c_str(), fb, err) && fb.size() == 6, "tiny plain file");
    std::string flip = gz;                                                    // a damaged stream
    flip[flip.size() / 2] ^= 0x55;
    spit(dir + "/bad.fa.gz", flip);
    err.clear();
    const bool bad_ok = dd::read_fasta_file((dir + "/bad.fa.gz").c_str(), fb, err);
    CHECK(!bad_ok || fb.size() != body.size() || memcmp(fb.data(), body.data(), body.size()) != 0, "damaged gzip read back as if intact");
}

// ---- dd_io.h: FASTQ records rewritten as FASTA, in place ----------------------------------------------------------------
// random text over the characters the record rules care about: the in-place form (exactly as large a buffer as the input
// plus the two bytes of slack the function asks for, so ASan sees any overrun) must equal the out-of-place form, never
// grow by more than two bytes and leave no line that starts with '+'
static void fastq_rewrite_cases() {
    std::mt19937 rng(20261004);
    const char alphabet[] = "ACGTacgtN>@+\n\n\n\r I#";
    for (int it = 0; it < 20000; ++it) {
        const size_t n = rng() % 96;
        std::string in(n, 'A');
        for (auto& c : in) c = alphabet[rng() % (sizeof alphabet - 1)];
        std::vector<uint8_t> out(n + 2), inplace(n + 2);
        const size_t m = dd::fastq_to_fasta(reinterpret_cast<const uint8_t*>(in.data()), n, out.data());
        memcpy(inplace.data(), in.data(), n);
        const size_t m2 = dd::fastq_to_fasta(inplace.data(), n, inplace.data());
        CHECK(m <= n + 2 && m == m2 && memcmp(out.data(), inplace.data(), m) == 0, "FASTQ rewrite: in place differs from out of place");
        CHECK(!dd::has_plus_line(out.data(), m), "FASTQ rewrite: a '+' line survived");
        // (a second pass may still shorten it: a '\r' that survived at the end of a record's sequence now stands in front of
        // a line end -- where it is an ambiguous byte behind the record's last k-mer either way)
        std::vector<uint8_t> again(m + 2);
        CHECK(dd::fastq_to_fasta(out.data(), m, again.data()) <= m, "FASTQ rewrite: a second pass grew the buffer");
        // the loaders' form of has_plus_line: every piece looked through on its own (exactly sized copies: ASan sees a read
        // outside a piece's bytes other than the one byte in front of it), the pieces' first bytes afterwards
        bool piecewise = false;
        std::vector<size_t> cuts{0};
        while (cuts.back() < n) cuts.push_back(std::min(n, cuts.back() + 1 + rng() % 40));
        std::vector<uint8_t> whole(in.begin(), in.end());
        for (size_t c = 0; c + 1 < cuts.size(); ++c) {
            piecewise |= dd::piece_has_plus_line(whole.data(), cuts[c], cuts[c + 1] - cuts[c]);
            piecewise |= dd::plus_at_piece_start(whole.data(), cuts[c]);
        }
        CHECK(piecewise == dd::has_plus_line(whole.data(), n), "FASTQ detection: piece by piece differs from the whole buffer");
    }
}

// ---- dd_io.h: what the device decoder's host side does with untrusted bytes ----------------------------------------------
// gzip_member_parse on real members with every optional header field, on the same members cut short and with random bytes
// in their first 64 (exactly sized heap copies: ASan sees any read past the end); the chunk-wise CRC-32 combination against
// zlib's crc32 of the whole
static std::string gz_member(const std::string& body, int level, int strategy, const std::string& extra);
static void device_gunzip_host_side() {
    std::mt19937 rng(20261005);
    std::string body;
    for (int i = 0; i < 40000; ++i) body += "ACGT"[rng() % 4];
    body = ">seq one\n" + body + "\n";
    const std::string plain = gz_member(body, 5, Z_DEFAULT_STRATEGY, std::string());   // (level 5: the helper writes no name field)
    {
        std::vector<uint8_t> v(plain.begin(), plain.end());
        dd::GzMember gm;
        CHECK(dd::gzip_member_parse(v.data(), v.size(), gm) && gm.isize == body.size() && gm.first_bit == 80 && gm.crc == (uint32_t)crc32(0, (const Bytef*)body.data(), (uInt)body.size()),
              "gzip_member_parse: a plain member");
    }
    // FEXTRA + FNAME + FCOMMENT + FHCRC in front of the same deflate data
    for (int flags = 0; flags < 16; ++flags) {
        std::string h = plain.substr(0, 10);
        h[3] = (char)((flags & 1 ? 4 : 0) | (flags & 2 ? 8 : 0) | (flags & 4 ? 16 : 0) | (flags & 8 ? 2 : 0));
        if (flags & 1) h += std::string("\x05\x00" "AB" "\x01\x00" "x", 7);   // XLEN 5: one subfield 'AB' of one byte
        if (flags & 2) h += std::string("genome.fa") + '\0';
        if (flags & 4) h += std::string("a comment") + '\0';
        if (flags & 8) {   // FHCRC: the low half of the CRC-32 of the header so far (zlib checks it)
            const uint32_t hc = (uint32_t)crc32(0, (const Bytef*)h.data(), (uInt)h.size());
            h += (char)(hc & 0xff);
            h += (char)((hc >> 8) & 0xff);
        }
        const std::string file = h + plain.substr(10);
        std::vector<uint8_t> v(file.begin(), file.end());
        dd::GzMember gm;
        CHECK(dd::gzip_member_parse(v.data(), v.size(), gm) && gm.first_bit == 8 * h.size() && gm.isize == body.size(), "gzip_member_parse: optional header fields");
        for (int it = 0; it < 300; ++it) {     // cut short / damaged: any answer, no read outside the copy
            const size_t cut = 1 + rng() % std::min<size_t>(v.size(), 200);
            std::vector<uint8_t> w(v.begin(), v.begin() + (rng() % 3 ? (ptrdiff_t)cut : (ptrdiff_t)v.size()));
            for (int k = 0; k < 4 && !w.empty(); ++k) w[rng() % std::min<size_t>(w.size(), 64)] = (uint8_t)rng();
            dd::GzMember g2;
            (void)dd::gzip_member_parse(w.data(), w.size(), g2);
        }
    }
    {   // a text that starts with '@' is classed as FASTQ (checked and resolved on the device, dd_fastq.hip); a '+' line among the
        // first bytes of a text that does not start with '@' goes to the host
        const std::string fq = gz_member("@r1\nACGT\n+\nIIII\n" + body, 5, Z_DEFAULT_STRATEGY, std::string());
        std::vector<uint8_t> v(fq.begin(), fq.end());
        dd::GzMember gm;
        CHECK(dd::gzip_member_parse(v.data(), v.size(), gm) && gm.fastq, "gzip_member_parse: a text that starts with '@' is FASTQ for the device");
        const std::string odd = gz_member(">x\nACGT\n+\nIIII\n" + body, 5, Z_DEFAULT_STRATEGY, std::string());
        std::vector<uint8_t> w(odd.begin(), odd.end());
        CHECK(!dd::gzip_member_parse(w.data(), w.size(), gm), "gzip_member_parse: a '+' line behind a '>' header must go to the host");
        const std::string fa = gz_member(">x\nACGT\n" + body, 5, Z_DEFAULT_STRATEGY, std::string());
        std::vector<uint8_t> u(fa.begin(), fa.end());
        CHECK(dd::gzip_member_parse(u.data(), u.size(), gm) && !gm.fastq, "gzip_member_parse: FASTA");
    }
    {   // round 5: files of SEVERAL members.  gzip_magic_scan == a byte-wise search on buffers with the pattern at every alignment and
        // across the 16-byte steps; gzip_members_parse finds the members of `cat a.gz b.gz c.gz` (texts of >= 64 KiB each), with
        // the scan done by the caller in pieces too; damaged and cut-short copies: any answer, no read outside the copy
        for (int it = 0; it < 300; ++it) {
            const size_t n = 3 + rng() % 400;
            std::vector<uint8_t> v(n);
            for (auto& c : v) c = (uint8_t)(rng() % 5 == 0 ? 0x1f : rng());
            for (int k = 0; k < 6; ++k) {
                const size_t q = rng() % (n - 2);
                v[q] = 0x1f, v[q + 1] = 0x8b, v[q + 2] = 0x08;
            }
            std::vector<size_t> want, got;
            for (size_t q = 0; q + 2 < n; ++q)
                if (v[q] == 0x1f && v[q + 1] == 0x8b && v[q + 2] == 0x08) want.push_back(q);
            const size_t lo = rng() % (n - 2), hi = lo + rng() % (n - 2 - lo + 1);
            dd::gzip_magic_scan(v.data(), lo, hi, got);
            std::vector<size_t> sub;
            for (size_t q : want)
                if (q >= lo && q < hi) sub.push_back(q);
            CHECK(got == sub, "gzip_magic_scan: differs from the byte-wise search");
        }
        std::string bodies[3];
        for (int b = 0; b < 3; ++b) {
            for (int i = 0; i < 300000 + 60000 * b; ++i) bodies[b] += "ACGT"[rng() % 4];   // (>= 64 KiB compressed each: smaller members are the host's)
            bodies[b] = ">seq " + std::to_string(b) + "\n" + bodies[b] + "\n";
        }
        const std::string m0 = gz_member(bodies[0], 6, Z_DEFAULT_STRATEGY, std::string()), m1 = gz_member(bodies[1], 1, Z_DEFAULT_STRATEGY, std::string()),
                          m2 = gz_member(bodies[2], 9, Z_DEFAULT_STRATEGY, std::string());
        const std::string file = m0 + m1 + m2;
        std::vector<uint8_t> v(file.begin(), file.end());
        std::vector<dd::GzMember> ms;
        CHECK(dd::gzip_members_parse(v.data(), v.size(), ms) && ms.size() == 3, "gzip_members_parse: three members");
        if (ms.size() == 3) {
            CHECK(ms[0].end == m0.size() && ms[1].end == m0.size() + m1.size() && ms[2].end == file.size(), "gzip_members_parse: member ends");
            CHECK(ms[1].first_bit == 8 * (m0.size() + 10) && ms[2].first_bit == 8 * (m0.size() + m1.size() + 10), "gzip_members_parse: first bits are the file's");
            for (int b = 0; b < 3; ++b)
                CHECK(ms[b].isize == bodies[b].size() && ms[b].crc == (uint32_t)crc32(0, (const Bytef*)bodies[b].data(), (uInt)bodies[b].size()), "gzip_members_parse: trailers");
        }
        // the caller's own scan, in pieces with the straddling positions checked separately (what the loader threads do)
        std::vector<size_t> magic;
        const size_t piece = 70001;
        for (size_t off = 0; off < v.size(); off += piece) {
            const size_t len = std::min(piece, v.size() - off);
            if (len > 2) dd::gzip_magic_scan(v.data(), off, off + len - 2, magic);
            if (off >= 2)
                for (size_t q = off - 2; q < off && q + 2 < v.size(); ++q)
                    if (v[q] == 0x1f && v[q + 1] == 0x8b && v[q + 2] == 0x08) magic.push_back(q);
        }
        std::sort(magic.begin(), magic.end());
        magic.erase(std::unique(magic.begin(), magic.end()), magic.end());
        std::vector<dd::GzMember> ms2;
        CHECK(dd::gzip_members_parse(v.data(), v.size(), ms2, &magic) && ms2.size() == 3 && ms2[1].end == ms[1].end, "gzip_members_parse: with the caller's scan");
        dd::GzMember one;
        CHECK(!dd::gzip_member_parse(v.data(), v.size(), one), "gzip_member_parse: three members are not ONE");
        for (int it = 0; it < 300; ++it) {
            std::vector<uint8_t> w(v.begin(), v.begin() + (ptrdiff_t)(rng() % 3 ? 1 + rng() % v.size() : v.size()));
            for (int k = 0; k < 6; ++k) w[rng() % w.size()] = (uint8_t)rng();
            if (rng() % 2) {   // a header look-alike somewhere
                const size_t q = rng() % w.size();
                const uint8_t fake[10] = {0x1f, 0x8b, 0x08, 0, 0, 0, 0, 0, 0, 3};
                for (size_t k = 0; k < 10 && q + k < w.size(); ++k) w[q + k] = fake[k];
            }
            std::vector<dd::GzMember> m3;
            (void)dd::gzip_members_parse(w.data(), w.size(), m3);
            for (const dd::GzMember& g : m3) CHECK(g.end <= w.size() && g.first_bit / 8 < g.end, "gzip_members_parse: a member outside the buffer");
        }
        // many small members: not for the device
        std::string small;
        for (int b = 0; b < 5; ++b) small += gz_member(bodies[0].substr(0, 20000), 6, Z_DEFAULT_STRATEGY, std::string());
        std::vector<uint8_t> sv(small.begin(), small.end());
        CHECK(!dd::gzip_members_parse(sv.data(), sv.size(), ms), "gzip_members_parse: small members must go to the host");
    }
    for (int it = 0; it < 200; ++it) {      // CRC-32 of 64 KiB chunks combined == CRC-32 of the whole
        const size_t n = 1 + rng() % 300000;
        std::vector<uint8_t> d(n);
        for (auto& c : d) c = (uint8_t)rng();
        const uint32_t full = dd::crc_x8n(65536u);
        uint32_t crc = 0;
        for (size_t a = 0; a < n; a += 65536) {
            const uint32_t len = (uint32_t)std::min<size_t>(65536, n - a), part = (uint32_t)crc32(0, d.data() + a, len);
            crc = a ? (dd::crc_multmodp(len == 65536u ? full : dd::crc_x8n(len), crc) ^ part) : part;
        }
        CHECK(crc == (uint32_t)crc32(0, d.data(), (uInt)n), "CRC-32 combination differs from zlib's crc32");
    }
}

// ---- dd_inflate.h: a gzip member decoded in pieces without their history, BGZF blocks in parallel -------------------
struct TestBuf {   // the part of FileBuf the decoders use
    uint8_t* p = nullptr;
    size_t len = 0, cap = 0;
    ~TestBuf() { free(p); }
    bool reserve(size_t n) {
        if (n <= cap) return true;
        uint8_t* q = static_cast<uint8_t*>(realloc(p, n));
        if (!q) return false;
        p = q;
        cap = n;
        return true;
    }
};

static std::string gz_member(const std::string& body, int level, int strategy = Z_DEFAULT_STRATEGY, const std::string& extra = std::string());
static std::string gz_member(const std::string& body, int level, int strategy, const std::string& extra) {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, level, Z_DEFLATED, 15 + 16, 8, strategy);
    gz_header hd;
    memset(&hd, 0, sizeof hd);
    std::string name = "x.fa";
    if (!extra.empty()) {
        hd.extra = reinterpret_cast<Bytef*>(const_cast<char*>(extra.data()));
        hd.extra_len = (uInt)extra.size();
        deflateSetHeader(&zs, &hd);
    } else if (level == 6) {   // one flavour with a file name in the header
        hd.name = reinterpret_cast<Bytef*>(const_cast<char*>(name.c_str()));
        deflateSetHeader(&zs, &hd);
    }
    std::string out(deflateBound(&zs, (uLong)body.size()) + 64 + extra.size(), '\0');
    zs.next_in = reinterpret_cast<Bytef*>(const_cast<char*>(body.data()));
    zs.avail_in = (uInt)body.size();
    zs.next_out = reinterpret_cast<Bytef*>(&out[0]);
    zs.avail_out = (uInt)out.size();
    deflate(&zs, Z_FINISH);
    out.resize(out.size() - zs.avail_out);
    deflateEnd(&zs);
    return out;
}

static std::string genome_text(size_t nbases, unsigned seed, bool repeats) {
    std::string body;
    body.reserve(nbases + nbases / 60 + 64);
    unsigned x = seed;
    size_t col = 0;
    for (size_t j = 0; j < nbases; ++j) {
        if (j % 200000 == 0) {
            if (col) body += '\n';
            body += ">contig" + std::to_string(j) + " synthetic\n";
            col = 0;
        }
        x = x * 1664525u + 1013904223u;
        char c = "ACGT"[(x >> 24) & 3];
        if (repeats && (j / 3000) % 3 == 0) c = "ACGTTTGACA"[j % 10];          // tandem repeats: long matches, far references
        if ((x >> 12) % 977 == 0) c = 'N';
        if ((j / 5000) % 7 == 0) c = (char)(c | 0x20);
        body += c;
        if (++col == 60) body += '\n', col = 0;
    }
    body += '\n';
    return body;
}

static void check_parallel_inflate() {
    const std::string big = genome_text(6000000, 7u, true), rnd = genome_text(3000000, 99u, false);
    int accepted = 0;
    for (int level : {1, 6, 9})
        for (const std::string* body : {&big, &rnd})
            for (int threads : {2, 5}) {
                const std::string gz = gz_member(*body, level);
                TestBuf out;
                // 128 KiB pieces: ~10-20 pieces per member, every one of them decoded without its history
                const int rc = dd::gunzip_member_parallel(reinterpret_cast<const uint8_t*>(gz.data()), gz.size(), out, threads, (size_t)128 << 10);
                CHECK(rc == 1 || rc == 0, "parallel gunzip rc %d", rc);
                if (rc == 1) {
                    ++accepted;
                    CHECK(out.len == body->size() && memcmp(out.p, body->data(), body->size()) == 0, "level %d threads %d: parallel gunzip differs from the input",
                          level, threads);
                }
            }
    CHECK(accepted >= 10, "the parallel decoder declined %d of 12 ordinary members", 12 - accepted);
    {   // fixed-Huffman and stored blocks (Z_FIXED; level 0): legal streams the block finder must not start in, decoded all the same
        for (int flavour = 0; flavour < 2; ++flavour) {
            const std::string gz = flavour ? gz_member(rnd, 0) : gz_member(rnd, 6, Z_FIXED);
            TestBuf out;
            const int rc = dd::gunzip_member_parallel(reinterpret_cast<const uint8_t*>(gz.data()), gz.size(), out, 4, (size_t)128 << 10);
            CHECK(rc == 0 || (rc == 1 && out.len == rnd.size() && memcmp(out.p, rnd.data(), rnd.size()) == 0), "flavour %d: rc %d", flavour, rc);
        }
    }
    {   // damage, truncation, two members, binary content: never a wrong answer, never a crash -- "not applicable" at worst
        const std::string gz = gz_member(big, 1);
        for (int kind = 0; kind < 6; ++kind) {
            std::string bad = gz;
            if (kind == 0) bad[bad.size() / 3] ^= 0x10;
            else if (kind == 1) bad.resize(bad.size() * 2 / 3);
            else if (kind == 2) bad += gz;                       // two members: the serial path's business
            else if (kind == 3) bad[bad.size() - 6] ^= 0x01;     // wrong CRC
            else if (kind == 4) bad[bad.size() - 2] ^= 0x01;     // wrong ISIZE
            else bad += std::string(100, '\0');
            TestBuf out;
            const int rc = dd::gunzip_member_parallel(reinterpret_cast<const uint8_t*>(bad.data()), bad.size(), out, 4, (size_t)128 << 10);
            CHECK(rc == 0 || (rc == 1 && out.len == big.size() && memcmp(out.p, big.data(), big.size()) == 0), "damage kind %d accepted as rc %d with other bytes", kind, rc);
            CHECK(!(rc == 1 && kind <= 4), "damage kind %d went unnoticed", kind);
        }
        std::string binary(4000000, '\0');
        unsigned x = 5;
        for (char& c : binary) x = x * 1664525u + 1013904223u, c = (char)(x >> 24);
        const std::string bz = gz_member(binary, 1);
        TestBuf out;
        const int rc = dd::gunzip_member_parallel(reinterpret_cast<const uint8_t*>(bz.data()), bz.size(), out, 4, (size_t)128 << 10);
        CHECK(rc == 0 || (rc == 1 && out.len == binary.size() && memcmp(out.p, binary.data(), binary.size()) == 0), "binary member: rc %d", rc);
    }
    {   // BGZF: 64 KiB blocks, each a member with a 'BC' extra field that says its compressed size, + the empty EOF block
        std::string file;
        const size_t blk = 65280;
        for (size_t a = 0; a <= big.size(); a += blk) {
            const std::string part = a < big.size() ? big.substr(a, blk) : std::string();
            std::string extra("BC\x02\x00\x00\x00", 6);
            std::string m = gz_member(part, 5, Z_DEFAULT_STRATEGY, extra);
            const size_t bsize = m.size() - 1;
            m[16] = (char)(bsize & 0xff);       // BSIZE sits at offset 16 of a BGZF block header
            m[17] = (char)(bsize >> 8);
            file += m;
        }
        auto member = [](const uint8_t* src, size_t len, uint8_t* dst, size_t cap, size_t* used, size_t* made) {
            return dd::zlib_gunzip_member(src, len, dst, cap, used, made);
        };
        TestBuf out;
        int rc = dd::gunzip_bgzf_parallel(reinterpret_cast<const uint8_t*>(file.data()), file.size(), out, 5, member);
        CHECK(rc == 1 && out.len == big.size() && memcmp(out.p, big.data(), big.size()) == 0, "BGZF: rc %d", rc);
        std::string cut = file.substr(0, file.size() - 40);       // a block cut short
        TestBuf out2;
        rc = dd::gunzip_bgzf_parallel(reinterpret_cast<const uint8_t*>(cut.data()), cut.size(), out2, 3, member);
        CHECK(rc == 0, "truncated BGZF accepted (rc %d)", rc);
        std::string flip = file;
        flip[file.size() / 2] ^= 0x40;
        TestBuf out3;
        rc = dd::gunzip_bgzf_parallel(reinterpret_cast<const uint8_t*>(flip.data()), flip.size(), out3, 3, member);
        CHECK(rc == 0 || (rc == 1 && out3.len == big.size() && memcmp(out3.p, big.data(), big.size()) == 0), "damaged BGZF: rc %d with other bytes", rc);
        const std::string plain = gz_member(rnd, 6);
        TestBuf out4;
        CHECK(dd::gunzip_bgzf_parallel(reinterpret_cast<const uint8_t*>(plain.data()), plain.size(), out4, 3, member) == 0, "a plain member taken for BGZF");
    }
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    dd::PlanKnobs base;
    const std::vector<size_t> ragged = {5000000, 0, 1, 63, 64, 65536, 65537, 1000000, 12345678};
    for (int log2m : {10, 14, 17, 18, 20})
        for (auto kr : {std::pair<int, int>{4, 40}, {1, 64}, {9, 10}, {32, 33}}) check_plan(log2m, ragged, kr.first, kr.second, base);
    check_plan(14, {3100000000ull}, 4, 64, base);
    // calls whose every genome is empty (a 0-byte FASTA at DandD's default -r 20): no epoch exists in bucket mode
    for (int log2m : {14, 17, 18, 19, 20})
        for (const std::vector<size_t>& sizes : {std::vector<size_t>{0}, std::vector<size_t>{0, 0, 0}}) {
            check_plan(log2m, sizes, 4, 40, base);
            for (const dd::SweepClass& sc : dd::plan_sweep(log2m, 1, sizes.data(), (int)sizes.size(), 4, 40, base))
                CHECK(sc.jobs.empty() || sc.plan.cap_chunks < (1u << 20), "log2m %d: empty input asks for %u chunks per row", log2m, sc.plan.cap_chunks);
        }
    check_plan(20, std::vector<size_t>(13, 3040000000ull), 4, 64, base);
    dd::PlanKnobs k2 = base;
    k2.bucket_e0_tiles = 1;
    k2.bucket_emax_tiles = 2;
    check_plan(18, ragged, 8, 35, k2);
    k2 = base;
    k2.bucket_cap_chunks = 3;                      // (a stream so short that records overflow into the registers)
    k2.bigmap_any_size = true;
    check_plan(19, ragged, 8, 35, k2);
    k2 = base;
    k2.bucket_budget = (size_t)1 << 30;            // (a budget that cuts the first epoch short)
    check_plan(20, ragged, 2, 20, k2);
    check_loaders(argv[1]);
    check_gzip_edges(argv[1]);
    check_parallel_inflate();
    fastq_rewrite_cases();
    device_gunzip_host_side();
    if (failures) fprintf(stderr, "%d failure(s)\n", failures);
    else printf("sanitize_host: ok\n");
    return failures ? 1 : 0;
}
