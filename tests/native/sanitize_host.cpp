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
    k2.buckets = false;
    check_plan(19, ragged, 8, 35, k2);
    k2 = base;
    k2.bucket_e0_tiles = 1;
    k2.bucket_emax_tiles = 2;
    k2.xcd_affinity = false;
    check_plan(18, ragged, 8, 35, k2);
    k2 = base;
    k2.filter = false;
    k2.use_bitmaps = false;
    check_plan(18, ragged, 2, 20, k2);
    check_loaders(argv[1]);
    check_gzip_edges(argv[1]);
    if (failures) fprintf(stderr, "%d failure(s)\n", failures);
    else printf("sanitize_host: ok\n");
    return failures ? 1 : 0;
}
