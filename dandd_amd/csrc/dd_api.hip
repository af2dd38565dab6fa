// dd_api.hip -- the C ABI of libdandd_hip.so (declared in include/dandd_hip.h).
//
// Host-side orchestration only: workspace management in HBM, job tables for the sweep,
// stream/event plumbing.  Every entry point names the DandD command line it replaces in
// include/dandd_hip.h.  There is no CPU fallback anywhere in this file.
#include "../../include/dandd_hip.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <dlfcn.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <chrono>
#include <thread>
#include <vector>

#include "dd_common.h"
#include "dd_io.h"
#include "dd_kernels.h"
#include "dd_plan.h"

using dd::FileBuf;
using dd::read_fasta_file;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define DD_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(DD_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                        __LINE__);                                                            \
    } while (0)

struct DevBuf {  // grow-only device allocation
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t n) {
        if (n <= cap) return DD_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 8 + 256;
        if (hipMalloc(&p, want) != hipSuccess) {
            p = nullptr;
            return fail(DD_ENOMEM, "hipMalloc(%zu) failed", want);
        }
        cap = want;
        return DD_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct HostBuf {  // grow-only pinned host staging
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t n) {
        if (n <= cap) return DD_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 8 + 256;
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) {
            p = nullptr;
            return fail(DD_ENOMEM, "hipHostMalloc(%zu) failed", want);
        }
        cap = want;
        return DD_OK;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct TimedSpan {
    hipEvent_t a, b;
};

}  // namespace

struct dd_ctx {
    int device = 0, p = 14, canonical = 1;
    hipStream_t stream = nullptr;
    bool timing = false;
    std::vector<TimedSpan> spans[DD_KERNEL_COUNT];
    std::vector<hipEvent_t> pool;
    // workspaces
    DevBuf tokens, scratch, tables, fasta, regs, ptrs, hist, est, ord, bitmaps, bigmaps, exact, buckets, gram, synth;
    HostBuf stage, stage_jobs, stage_rows;  // genome/pack tables and K1 job tables are uploaded in two steps
    // the job tables of the last few sketch calls: a call over genomes of the same sizes and the same k range (a
    // pipeline sketching batches of a few recurring shapes, a benchmark loop) reuses them, on the host and in HBM
    struct PlanEntry {
        bool valid = false;
        int kmin = 0, kmax = 0;
        std::vector<size_t> sizes;
        dd::PlanKnobs knobs;
        std::vector<dd::SweepClass> classes;
        std::vector<size_t> job_off;
        DevBuf jobtab;
        unsigned long long last_use = 0;
    };
    PlanEntry plans[8];
    unsigned long long plan_clock = 0;
    hipEvent_t stage_free = nullptr;  // signalled when the last upload from `stage` completed
    // a second set of staging buffers: dd_sketch_device alternates, so that a call can be issued while the uploads of
    // the call before it are still queued behind work of other streams (the ingestion pipeline issues batch b + 1
    // while batch b waits for its files to be copied or inflated)
    HostBuf stage_alt, stage_jobs_alt, stage_rows_alt;
    hipEvent_t stage_free_alt = nullptr;
    // ingestion pipeline (dd_sketch_files): pinned host buffers for the loader threads, a copy stream, two
    // device buffer sets (FASTA bytes in, register slabs out) and two pinned bounce buffers for the results
    std::vector<FileBuf*> file_pool;
    hipStream_t copy_stream_b = nullptr;  // device-inflated batches alternate between two: a launch of the inflate kernel is as long as ONE block takes, two in flight hide each other
    hipStream_t copy_stream = nullptr, out_stream = nullptr;  // H2D and D2H on streams of their own: an in-order stream would park batch b+1's upload behind batch b's results
    DevBuf pipe_fasta[2], pipe_regs[2];
    HostBuf pipe_out[2];
    // BGZF files inflated on the device (dd_ginflate.hip): compressed bytes, block table and error count of a batch
    DevBuf pipe_gz[2], pipe_jobs[2], pipe_err[2];
    HostBuf pipe_jobs_host[2], pipe_err_host[2];
    // single-member gzip files inflated on the device: symbols, windows, the piece tables (RawFile[], starts, lens, offs,
    // chunk0, crcs) and their host copies
    DevBuf pipe_sym[2], pipe_win[2], pipe_raw[2];
    HostBuf pipe_raw_host[2], pipe_crc_host[2];
    // kseq's record rules over device-inflated texts (dd_fastq.hip): the batch's TextJob table, newline counts and positions
    DevBuf pipe_txt[2];
    HostBuf pipe_txt_host[2];
    bool no_gpu_inflate = false;   // this context inflates on the host (set for the retry of a call, for good after three)
    int inflate_refusals = 0;      // calls in which the device decoder refused a block
    bool inflate_retry = false;    // ... and the call that met it is run again
    // dd_inflate_files: the text of every file of the running dd_sketch_files pass, as K0 is about to read it, goes here
    struct TextSink {
        uint8_t* const* out;
        const size_t* caps;
        size_t* lens;
        bool short_buffer;
    };
    TextSink* text_sink = nullptr;
    bool inflate_retry_counts = false;   // ... and counts towards the three strikes (a size mismatch or a lack of device memory does not:
                                         //     the decoder did its work, the FILE -- damaged trailer, two members, text beyond 4 GiB -- is not for it)
    hipEvent_t pipe_h2d[2] = {nullptr, nullptr}, pipe_done[2] = {nullptr, nullptr}, pipe_d2h[2] = {nullptr, nullptr};
    hipStream_t side[8] = {};  // k classes of a small call run side by side
    hipEvent_t side_done[8] = {}, side_go = nullptr;
    int ingest_calls = 0;
    // HBM the record streams of one log2m >= 17 call may take: a sixth of the device (48 GiB of 288), 16 GiB at least
    size_t bucket_budget = (size_t)16 << 30;
    double ingest_ms[4] = {0, 0, 0, 0};  // last dd_sketch_files call: wall, waiting for loaders, batches, bytes (as a double)
    // stats of the last sketch call
    uint64_t st_tokens = 0, st_updates = 0;
    int st_blocks = 0;
    int k2_path = 0;  // DD_K2_*: what the last progressive / pairwise call ran
    // multi-GPU (dd_comm_*): this context's rank in an RCCL communicator, one context = one process = one GPU
    void* comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    unsigned long long comm_calls[2] = {0, 0};   // all-reduces, all-gathers issued
};

namespace {

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        (void)hipGetDevice(&prev);
        if (prev != dev) (void)hipSetDevice(dev);
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

hipEvent_t get_event(dd_ctx* c) {
    if (!c->pool.empty()) {
        hipEvent_t e = c->pool.back();
        c->pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct Span {  // brackets a launch (or a whole phase) on the context's stream with events when timing is on
    dd_ctx* c;
    int which;
    bool on;
    TimedSpan s{};
    Span(dd_ctx* c_, int which_, bool on_ = true) : c(c_), which(which_), on(on_ && c_->timing) {
        if (on) {
            s.a = get_event(c);
            s.b = get_event(c);
            (void)hipEventRecord(s.a, c->stream);
        }
    }
    ~Span() {
        if (on) {
            (void)hipEventRecord(s.b, c->stream);
            c->spans[which].push_back(s);
        }
    }
};

// upload a host table through the pinned staging buffer (async on the stream)
int upload(dd_ctx* c, HostBuf& stage, void* dst_dev, const void* src, size_t bytes, size_t stage_off) {
    if (!bytes) return DD_OK;
    memcpy(static_cast<char*>(stage.p) + stage_off, src, bytes);
    DD_HIP(hipMemcpyAsync(dst_dev, static_cast<char*>(stage.p) + stage_off, bytes,
                          hipMemcpyHostToDevice, c->stream));
    return DD_OK;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// CPUs this process may really use: the affinity mask capped by the cgroup quota (a container that shows
// 256 logical CPUs behind a 16-CPU quota must not get 256 loader threads)
int usable_cpus() {
    int n = (int)std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::max(1, CPU_COUNT(&set));
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32];
        long period = 0;
        if (fscanf(f, "%31s %ld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0)
            n = std::min(n, std::max(1, (int)(atol(quota) / period)));
        fclose(f);
    }
    return n;
}

int check_ctx(dd_ctx* c) {
    if (!c) return fail(DD_EINVAL, "null context");
    return DD_OK;
}

// histograms already on the device -> estimates on the host (device MLE, bit-identical to the
// host MLE: same IEEE operations, no contraction; asserted by tests/test_gpu_parity.py)
int estimates_from_hist(dd_ctx* c, const uint32_t* hist_dev, size_t njobs, double* est_host) {
    int rc = c->est.reserve(njobs * sizeof(double));
    if (rc) return rc;
    dd::launch_mle(hist_dev, njobs, c->p, static_cast<double*>(c->est.p), c->stream);
    DD_HIP(hipGetLastError());
    DD_HIP(hipMemcpyAsync(est_host, c->est.p, njobs * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    DD_HIP(hipStreamSynchronize(c->stream));
    return DD_OK;
}

}  // namespace

extern "C" {

int dd_abi_version(void) { return DD_ABI_VERSION; }

const char* dd_last_error(void) { return g_err.c_str(); }

dd_ctx* dd_create(int device, int log2m, int canonical) {
    if (log2m < 4 || log2m > 20) {
        fail(DD_EINVAL, "log2m=%d outside 4..20", log2m);
        return nullptr;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        fail(DD_ENODEV, "no HIP device visible (%s): libdandd_hip has no CPU path",
             e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return nullptr;
    }
    if (device < 0 || device >= ndev) {
        fail(DD_ENODEV, "device %d not in 0..%d", device, ndev - 1);
        return nullptr;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
        fail(DD_ENODEV, "hipGetDeviceProperties(%d) failed", device);
        return nullptr;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fail(DD_ENODEV, "device %d is %s; this library is built for gfx950 only", device,
             prop.gcnArchName);
        return nullptr;
    }
    dd_ctx* c = new dd_ctx();
    c->device = device;
    c->p = log2m;
    c->canonical = canonical ? 1 : 0;
    c->bucket_budget = std::min<size_t>((size_t)48 << 30, std::max<size_t>((size_t)16 << 30, prop.totalGlobalMem / 6));
    DeviceGuard g(device);
    if (hipEventCreateWithFlags(&c->stage_free, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->stage_free_alt, hipEventDisableTiming) != hipSuccess) {
        fail(DD_ENODEV, "hipEventCreate failed");
        delete c;
        return nullptr;
    }
    return c;
}

void dd_destroy(dd_ctx* c) {
    if (!c) return;
    DeviceGuard g(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)dd_comm_destroy(c);
    for (auto& v : c->spans)
        for (auto& s : v) {
            (void)hipEventDestroy(s.a);
            (void)hipEventDestroy(s.b);
        }
    for (auto e : c->pool) (void)hipEventDestroy(e);
    if (c->stage_free) (void)hipEventDestroy(c->stage_free);
    if (c->stage_free_alt) (void)hipEventDestroy(c->stage_free_alt);
    for (auto& pe : c->plans) pe.jobtab.release();
    for (DevBuf* b : {&c->tokens, &c->scratch, &c->tables, &c->fasta, &c->regs, &c->ptrs, &c->hist,
                      &c->est, &c->ord, &c->bitmaps, &c->bigmaps, &c->exact, &c->buckets, &c->gram, &c->synth})
        b->release();
    c->stage.release();
    c->stage_jobs.release();
    c->stage_rows.release();
    c->stage_alt.release();
    c->stage_jobs_alt.release();
    c->stage_rows_alt.release();
    for (FileBuf* fb : c->file_pool) delete fb;
    for (int i = 0; i < 2; ++i) {
        c->pipe_fasta[i].release();
        c->pipe_regs[i].release();
        c->pipe_out[i].release();
        c->pipe_gz[i].release();
        c->pipe_sym[i].release();
        c->pipe_win[i].release();
        c->pipe_raw[i].release();
        c->pipe_raw_host[i].release();
        c->pipe_crc_host[i].release();
        c->pipe_txt[i].release();
        c->pipe_txt_host[i].release();
        c->pipe_jobs[i].release();
        c->pipe_err[i].release();
        c->pipe_jobs_host[i].release();
        c->pipe_err_host[i].release();
        for (hipEvent_t e : {c->pipe_h2d[i], c->pipe_done[i], c->pipe_d2h[i]})
            if (e) (void)hipEventDestroy(e);
    }
    for (int i = 0; i < 8; ++i)
        if (c->side[i]) {
            (void)hipStreamDestroy(c->side[i]);
            (void)hipEventDestroy(c->side_done[i]);
        }
    if (c->side_go) {
        (void)hipEventDestroy(c->side_go);
    }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->copy_stream_b) (void)hipStreamDestroy(c->copy_stream_b);
    if (c->out_stream) (void)hipStreamDestroy(c->out_stream);
    delete c;
}

int dd_set_stream(dd_ctx* c, void* hip_stream) {
    if (check_ctx(c)) return DD_EINVAL;
    hipStream_t next = static_cast<hipStream_t>(hip_stream);
    if (next != c->stream) {
        // Work queued on the old stream still uses the context's tables and workspaces (the cached K1 job
        // tables were uploaded there); nothing orders a new stream behind it, so it is drained first.
        DeviceGuard g(c->device);
        // The old handle is not touched: a caller may hand over a new stream because it already destroyed the old
        // one, and synchronising a destroyed hipStream_t is undefined.  Draining the device covers the old stream
        // whether it still exists or not (destroying a stream lets its queued work finish); switching streams is rare.
        DD_HIP(hipDeviceSynchronize());
        c->stream = next;
    }
    return DD_OK;
}

int dd_synchronize(dd_ctx* c) {
    if (check_ctx(c)) return DD_EINVAL;
    DeviceGuard g(c->device);
    DD_HIP(hipStreamSynchronize(c->stream));
    return DD_OK;
}

// ------------------------------------------------------------------------------ sketch
// The side streams the k classes of a call run on: `n` of them (at most 8), made when first asked for -- a stream
// costs 2 ms to create and as much again to destroy, which a one-shot process pays in full.
static int ensure_side_streams(dd_ctx* c, int n) {
    if (!c->side_go) DD_HIP(hipEventCreateWithFlags(&c->side_go, hipEventDisableTiming));
    for (int i = 0; i < std::min(n, 8); ++i) {
        if (c->side[i]) continue;
        DD_HIP(hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking));
        DD_HIP(hipEventCreateWithFlags(&c->side_done[i], hipEventDisableTiming));
    }
    return DD_OK;
}

int dd_sketch_device(dd_ctx* c, const uint8_t* const* fasta_dev, const size_t* nbytes, int ngenomes,
                     int kmin, int kmax, uint8_t* regs_dev) {
    if (check_ctx(c)) return DD_EINVAL;
    if (ngenomes < 0 || !regs_dev || (ngenomes && (!fasta_dev || !nbytes)))
        return fail(DD_EINVAL, "null argument");
    if (kmin < 1 || kmax > 64 || kmin > kmax) return fail(DD_EINVAL, "k range %d..%d outside 1..64", kmin, kmax);
    for (int g = 0; g < ngenomes; ++g) {
        if (nbytes[g] && !fasta_dev[g]) return fail(DD_EINVAL, "genome %d: null buffer", g);
        if (reinterpret_cast<uintptr_t>(fasta_dev[g]) & 15)
            return fail(DD_EINVAL, "genome %d: device buffer must be 16-byte aligned", g);
    }
    DeviceGuard guard(c->device);
    const int p = c->p, K = kmax - kmin + 1;
    const size_t m = (size_t)1 << p;
    hipStream_t st = c->stream;

    DD_HIP(hipMemsetAsync(regs_dev, 0, (size_t)ngenomes * K * m, st));
    if (!ngenomes) return DD_OK;

    // ---- workspace: token streams of all genomes + one K0 scratch --------------------
    std::vector<size_t> off_codes(ngenomes), off_bad(ngenomes), off_ntok(ngenomes);
    size_t tot = 0, max_n = 0;
    for (int g = 0; g < ngenomes; ++g) {
        off_codes[g] = tot;
        tot += align_up(dd::codes_words(nbytes[g]) * 4, 256);
        off_bad[g] = tot;
        tot += align_up(dd::bad_words(nbytes[g]) * 4, 256);
        off_ntok[g] = tot;
        tot += 256;
        max_n = std::max(max_n, nbytes[g]);
    }
    std::vector<size_t> off_scratch(ngenomes);
    size_t scratch_tot = 0;
    for (int g = 0; g < ngenomes; ++g) {
        off_scratch[g] = scratch_tot;
        scratch_tot += align_up(dd::pack_scratch_bytes(nbytes[g]), 256);
    }
    int rc;
    if ((rc = c->tokens.reserve(tot))) return rc;
    if ((rc = c->scratch.reserve(scratch_tot))) return rc;
    char* tb = static_cast<char*>(c->tokens.p);
    char* sb = static_cast<char*>(c->scratch.p);

    // presence bitmaps for the small-k class (k <= 9), zeroed per call
    const bool use_bitmaps = kmin <= dd::kBitmapMaxK;
    uint32_t* bitmap_base = nullptr;
    if (use_bitmaps) {
        const size_t bbytes = (size_t)ngenomes * dd::kBitmapStride * sizeof(uint32_t);
        if ((rc = c->bitmaps.reserve(bbytes))) return rc;
        bitmap_base = static_cast<uint32_t*>(c->bitmaps.p);
        DD_HIP(hipMemsetAsync(bitmap_base, 0, bbytes, st));
    }

    // ... and for k = 10 (, 11) at log2m >= 19 (dd_kernels.h)
    uint32_t* bigmap_base = nullptr;
    size_t bigmap_stride = 0;
    {
        int ka = 0, kb = 0;
        if (dd::plan_bigmap_range(p, kmin, kmax, dd::PlanKnobs::from_env(), nbytes, ngenomes, &ka, &kb)) {
            bigmap_stride = dd::bigmap_offset_words(kb + 1, c->canonical != 0);
            const size_t bbytes = (size_t)ngenomes * bigmap_stride * sizeof(uint32_t);
            if ((rc = c->bigmaps.reserve(bbytes))) return rc;
            bigmap_base = static_cast<uint32_t*>(c->bigmaps.p);
            DD_HIP(hipMemsetAsync(bigmap_base, 0, bbytes, st));
        }
    }

    // ---- K0 / K1 genome tables -----------------------------------------------------------
    std::vector<dd::SweepGenome> gtab(ngenomes);
    std::vector<dd::PackGenome> ptab(ngenomes);
    uint64_t tokens_ub = 0;
    size_t max_chunks = 0;
    for (int g = 0; g < ngenomes; ++g) {
        dd::TokenStream ts{reinterpret_cast<uint32_t*>(tb + off_codes[g]),
                           reinterpret_cast<uint32_t*>(tb + off_bad[g]),
                           reinterpret_cast<unsigned long long*>(tb + off_ntok[g])};
        ptab[g] = dd::PackGenome{fasta_dev[g], nbytes[g], dd::pack_chunks(nbytes[g]),
                                 reinterpret_cast<long long*>(sb + off_scratch[g]), ts};
        max_chunks = std::max(max_chunks, ptab[g].nchunks);
        gtab[g] = dd::SweepGenome{ts.codes, ts.bad, ts.ntok, regs_dev + (size_t)g * K * m,
                                  bitmap_base ? bitmap_base + (size_t)g * dd::kBitmapStride : nullptr,
                                  bigmap_base ? bigmap_base + (size_t)g * bigmap_stride : nullptr};
        tokens_ub += nbytes[g];
    }
    (void)max_n;

    // ---- genome tables up, K0 launched: the K1 job tables are planned on the host meanwhile -------
    const size_t pack_off = align_up(sizeof(dd::SweepGenome) * ngenomes, 256);
    const size_t gtab_bytes = pack_off + align_up(sizeof(dd::PackGenome) * ngenomes, 256);
    if ((rc = c->tables.reserve(gtab_bytes))) return rc;
    // the staging buffers may still be feeding the uploads of the call before last (they alternate: dd_ctx)
    std::swap(c->stage, c->stage_alt);
    std::swap(c->stage_jobs, c->stage_jobs_alt);
    std::swap(c->stage_rows, c->stage_rows_alt);
    std::swap(c->stage_free, c->stage_free_alt);
    DD_HIP(hipEventSynchronize(c->stage_free));
    if ((rc = c->stage.reserve(gtab_bytes))) return rc;
    char* tdev = static_cast<char*>(c->tables.p);
    if ((rc = upload(c, c->stage, tdev, gtab.data(), sizeof(dd::SweepGenome) * ngenomes, 0))) return rc;
    if ((rc = upload(c, c->stage, tdev + pack_off, ptab.data(), sizeof(dd::PackGenome) * ngenomes, pack_off))) return rc;
    {
        Span sp(c, DD_KERNEL_PACK);  // K0: pack every genome of the batch (three launches)
        dd::launch_pack_batch(reinterpret_cast<const dd::PackGenome*>(tdev + pack_off), ngenomes,
                              max_chunks, st);
    }
    DD_HIP(hipGetLastError());

    // ---- K1 job tables (dd_plan.hip), built while K0 runs -----------------------------------
    dd::PlanKnobs knobs = dd::PlanKnobs::from_env();
    // (longer epochs = fewer launches and sharper filters per record: +4 % on 13 x 3 Gbp at log2m 20 with 48 GiB)
    if (!getenv("DD_BUCKET_GB")) knobs.bucket_budget = c->bucket_budget;
    dd_ctx::PlanEntry* hit = nullptr;
    dd_ctx::PlanEntry* oldest = &c->plans[0];
    for (auto& pe : c->plans) {
        if (pe.valid && pe.kmin == kmin && pe.kmax == kmax && pe.knobs == knobs && pe.sizes.size() == (size_t)ngenomes &&
            std::equal(pe.sizes.begin(), pe.sizes.end(), nbytes))
            hit = &pe;
        if (pe.last_use < oldest->last_use) oldest = &pe;
    }
    auto& pc = hit ? *hit : *oldest;
    pc.last_use = ++c->plan_clock;
    if (!hit) {
        // (the entry being replaced may still be read by kernels of an earlier call: its device table is only ever
        // written by copies on this same stream, and a table that must grow is freed by hipFree, which waits)
        pc.valid = false;
        pc.classes = dd::plan_sweep(p, c->canonical, nbytes, ngenomes, kmin, kmax, knobs);
        size_t job_bytes = 0;
        pc.job_off.assign(pc.classes.size(), 0);
        for (size_t i = 0; i < pc.classes.size(); ++i) {
            pc.job_off[i] = job_bytes;
            job_bytes += align_up(sizeof(dd::SweepJob) * pc.classes[i].jobs.size(), 256);
        }
        if ((rc = pc.jobtab.reserve(job_bytes))) return rc;
        if ((rc = c->stage_jobs.reserve(job_bytes))) return rc;
        for (size_t i = 0; i < pc.classes.size(); ++i)
            if ((rc = upload(c, c->stage_jobs, static_cast<char*>(pc.jobtab.p) + pc.job_off[i], pc.classes[i].jobs.data(),
                             sizeof(dd::SweepJob) * pc.classes[i].jobs.size(), pc.job_off[i])))
                return rc;
        pc.kmin = kmin;
        pc.kmax = kmax;
        pc.knobs = knobs;
        pc.sizes.assign(nbytes, nbytes + ngenomes);
        pc.valid = true;
    }
    const std::vector<dd::SweepClass>& classes = pc.classes;
    const std::vector<size_t>& job_off = pc.job_off;
    char* jdev = static_cast<char*>(pc.jobtab.p);
    DD_HIP(hipEventRecord(c->stage_free, st));

    // ---- bucket mode (log2m >= 17): row table, cursors, filters and record areas ----------------
    const dd::SweepPlan* bplan = nullptr;
    for (const dd::SweepClass& sc : classes)
        if (sc.plan.mode == dd::kBucketMode) bplan = &sc.plan;
    const dd::BucketRow* rows_dev = nullptr;
    const int nrows = ngenomes * K;
    if (bplan) {
        const size_t flt_bytes = align_up((m >> bplan->logg) / 2, 16), area_bytes = (size_t)bplan->cap_chunks * 4096;  // 4-bit filter entries; 1024 records per chunk
        const size_t fill_bytes = align_up((size_t)bplan->cap_chunks * 4, 256) + align_up((size_t)bplan->cap_chunks * 32, 256);  // fill + seg
        int first_hashed = K, hashed_per_genome = 0;  // rows of a genome that belong to a bucket class
        for (const dd::SweepClass& sc : classes)
            if (sc.plan.mode == dd::kBucketMode) {
                first_hashed = std::min(first_hashed, sc.kfirst - kmin);
                hashed_per_genome += sc.klast - sc.kfirst + 1;
            }
        const size_t nhashed = (size_t)ngenomes * hashed_per_genome;
        const size_t tab_bytes = align_up(sizeof(dd::BucketRow) * nrows, 256);
        // one cursor per row, each in a 256-byte slot of its own: every block of a row is reserved by an atomic add on it,
        // and neighbouring rows are written from other XCDs
        const size_t cur_stride = 256;
        const size_t cur_bytes = align_up((size_t)nrows * cur_stride, 256);
        const size_t flt_tot = align_up(nhashed * flt_bytes, 256);
        // the first epoch's updates of rho = 1: one bit per register instead of a record each (dd_sweep.hip,
        // scatter_first_bin_kernel); the bits start at zero with the cursors and filters
        const size_t ones_bytes = m / 8, ones_tot = align_up(nhashed * ones_bytes, 256);
        if ((rc = c->buckets.reserve(tab_bytes + cur_bytes + flt_tot + ones_tot + nhashed * (fill_bytes + area_bytes)))) return rc;
        if ((rc = c->stage_rows.reserve(tab_bytes))) return rc;
        char* bb = static_cast<char*>(c->buckets.p);
        char* fills = bb + tab_bytes + cur_bytes + flt_tot + ones_tot;
        char* areas = fills + nhashed * fill_bytes;
        std::vector<dd::BucketRow> rtab(nrows);
        size_t h = 0;
        for (int g = 0; g < ngenomes; ++g)
            for (int kk = 0; kk < K; ++kk) {
                dd::BucketRow& r = rtab[(size_t)g * K + kk];
                r.regs = regs_dev + ((size_t)g * K + kk) * m;
                r.cursor = reinterpret_cast<uint32_t*>(bb + tab_bytes + ((size_t)g * K + kk) * cur_stride);
                const bool hashed = kk >= first_hashed && kk < first_hashed + hashed_per_genome;
                r.filter = hashed ? reinterpret_cast<uint8_t*>(bb + tab_bytes + cur_bytes + h * flt_bytes) : nullptr;
                r.ones = hashed ? reinterpret_cast<uint32_t*>(bb + tab_bytes + cur_bytes + flt_tot + h * ones_bytes) : nullptr;
                r.fill = hashed ? reinterpret_cast<uint32_t*>(fills + h * fill_bytes) : nullptr;
                r.seg = hashed ? reinterpret_cast<uint16_t*>(fills + h * fill_bytes + align_up((size_t)bplan->cap_chunks * 4, 256)) : nullptr;
                r.area = hashed ? reinterpret_cast<uint32_t*>(areas + h * area_bytes) : nullptr;
                h += hashed ? 1 : 0;
            }
        // cursors, filters and bits start at zero: nothing handed out, every register's lower bound is 0
        DD_HIP(hipMemsetAsync(bb + tab_bytes, 0, cur_bytes + flt_tot + ones_tot, st));
        if ((rc = upload(c, c->stage_rows, bb, rtab.data(), sizeof(dd::BucketRow) * nrows, 0))) return rc;
        rows_dev = reinterpret_cast<const dd::BucketRow*>(bb);
        DD_HIP(hipEventRecord(c->stage_free, st));
    }

    // ---- K1 launches -------------------------------------------------------------------
    auto launch_lds_class = [&](const dd::SweepClass& sc, size_t i, hipStream_t ks) {
        const dd::SweepGenome* gt = reinterpret_cast<const dd::SweepGenome*>(tdev);
        const dd::SweepJob* jt = reinterpret_cast<const dd::SweepJob*>(jdev + job_off[i]);
        if (sc.kclass == dd::kBitmapClass) {
            dd::launch_bitmap(gt, jt, (int)sc.jobs.size(), sc.kfirst, sc.klast, c->canonical, ks);
            dd::launch_bitmap_finish(gt, ngenomes, sc.kfirst, sc.klast, kmin, p, ks);
        } else if (sc.kclass == dd::kBigmapClass) {
            dd::launch_bigmap(gt, jt, (int)sc.jobs.size(), c->canonical, ks);
            dd::launch_bigmap_finish(gt, ngenomes, sc.kfirst, sc.klast, kmin, p, c->canonical, ks);
        } else {
            dd::launch_sweep(gt, jt, (int)sc.jobs.size(), sc.kclass, sc.plan, ks);
        }
    };
    // The k classes are independent.  On a big call they are launched back to back (running them side by side
    // was measured neutral to slightly slower: they compete for the same VALUs).  On a SMALL call -- one batch of
    // the ingestion pipeline, a single genome -- every class is only a few rounds of workgroups long and ends
    // with a tail of idle CUs: there the classes go to side streams so that one's tail overlaps another's body.
    int blocks = 0;
    size_t lds_jobs = 0;
    int lds_classes = 0;
    for (const dd::SweepClass& sc : classes)
        if (sc.plan.mode != dd::kBucketMode) lds_jobs += sc.jobs.size(), ++lds_classes;
    const bool side = lds_classes > 1 && lds_jobs < 12000;
    if (side && (rc = ensure_side_streams(c, lds_classes))) return rc;
    // log2m >= 17, see below.  A call whose only epoch is the unfiltered first one (many small genomes: 64 x 5 Mbp at
    // log2m 20) runs its classes one after the other instead: its scatter (returning LDS atomics, 4-byte stores) and
    // its replay (HBM reads at 5 TB/s) each have the chip to themselves then -- 24.4 -> 22.9 ms with round 4's kernels
    // (profiles/r04_bucket_path.txt); calls with filtered epochs keep the side streams (26.8 against 24.9 ms without).
    const bool side_b = bplan && bplan->nepochs > 1;
    // (launches that run side by side are timed as ONE span on the caller's stream: per-launch spans would overlap)
    std::unique_ptr<Span> phase((side || side_b) ? new Span(c, DD_KERNEL_SWEEP) : nullptr);
    if (side) DD_HIP(hipEventRecord(c->side_go, st));
    int lane_no = 0;
    for (size_t i = 0; i < classes.size(); ++i) {
        const dd::SweepClass& sc = classes[i];
        if (sc.plan.mode == dd::kBucketMode || side_b) continue;
        hipStream_t ks = st;
        if (side) {
            ks = c->side[lane_no & 7];
            DD_HIP(hipStreamWaitEvent(ks, c->side_go, 0));
        }
        Span sp(c, DD_KERNEL_SWEEP, !side);
        launch_lds_class(sc, i, ks);
        if (side) {
            DD_HIP(hipEventRecord(c->side_done[lane_no & 7], ks));
            DD_HIP(hipStreamWaitEvent(st, c->side_done[lane_no & 7], 0));
            ++lane_no;
        }
        blocks += (int)sc.jobs.size();
    }
    if (bplan) {
        // Every k class is a pipeline of its own -- scatter(e), (sort(e),) replay(e), scatter(e+1) ... over its own rows --
        // so, when there are filtered epochs, each gets a side stream: the tails of one class's launches are filled by the
        // others' work.  (Starting the pipelines one first-epoch scatter apart, and streams of different priorities, were
        // measured and lost: profiles/r03_bucket_path.txt, r04_bucket_path.txt.)
        const dd::ScatterParams sp{rows_dev, K, bplan->logg, bplan->cap_chunks, bplan->nb_log2};
        if (side_b && (rc = ensure_side_streams(c, (int)classes.size()))) return rc;
        if (side_b) DD_HIP(hipEventRecord(c->side_go, st));
        int lane_b = 0;
        for (size_t i = 0; i < classes.size(); ++i) {
            const dd::SweepClass& sc = classes[i];
            hipStream_t ks = st;
            if (side_b) {
                ks = c->side[lane_b & 7];
                DD_HIP(hipStreamWaitEvent(ks, c->side_go, 0));
            }
            if (sc.plan.mode != dd::kBucketMode) {
                if (!side_b) continue;  // (already launched above)
                launch_lds_class(sc, i, ks);   // the small-k classes (their rows are not bucketed) run beside the pipelines
                blocks += (int)sc.jobs.size();
            }
            for (int e = 0; sc.plan.mode == dd::kBucketMode && e < bplan->nepochs; ++e) {
                const size_t j0 = sc.epoch_begin[e], j1 = sc.epoch_begin[e + 1];
                if (j1 == j0) continue;
                Span span(c, DD_KERNEL_SWEEP, !side_b);
                dd::launch_scatter(reinterpret_cast<const dd::SweepGenome*>(tdev),
                                   reinterpret_cast<const dd::SweepJob*>(jdev + job_off[i]) + j0, (int)(j1 - j0),
                                   sc.kclass, sc.plan, sp, ks, e == 0);
                dd::launch_replay(rows_dev, ngenomes, K, sc.kfirst - kmin, sc.klast - sc.kfirst + 1, *bplan, ks, e == 0);
                blocks += (int)(j1 - j0);
            }
            if (side_b) {
                DD_HIP(hipEventRecord(c->side_done[lane_b & 7], ks));
                DD_HIP(hipStreamWaitEvent(st, c->side_done[lane_b & 7], 0));
                ++lane_b;
            }
        }
    }
    phase.reset();  // (closes the span: every side stream has been joined into the caller's stream above)
    DD_HIP(hipGetLastError());
    c->st_tokens = tokens_ub;
    c->st_updates = tokens_ub * (uint64_t)K;
    c->st_blocks = blocks;
    return DD_OK;
}

int dd_sketch_buffer(dd_ctx* c, const uint8_t* fasta, size_t nbytes, int kmin, int kmax, uint8_t* regs) {
    if (check_ctx(c)) return DD_EINVAL;
    if (!regs || (nbytes && !fasta)) return fail(DD_EINVAL, "null argument");
    if (kmin < 1 || kmax > 64 || kmin > kmax) return fail(DD_EINVAL, "k range %d..%d outside 1..64", kmin, kmax);
    DeviceGuard guard(c->device);
    const size_t K = (size_t)(kmax - kmin + 1), m = (size_t)1 << c->p;
    int rc;
    // (FASTQ in a host buffer: resolved into the FASTA K0 reads, as the file paths do -- dd_io.h)
    FileBuf fq;
    if (dd::has_plus_line(fasta, nbytes)) {
        if (!fq.reserve(nbytes + 16)) return fail(DD_ENOMEM, "out of host memory");
        fq.len = nbytes = dd::fastq_to_fasta(fasta, nbytes, fq.p);
        fasta = fq.p;
    }
    if ((rc = c->fasta.reserve(nbytes + 16))) return rc;
    if ((rc = c->regs.reserve(K * m))) return rc;
    if (nbytes) DD_HIP(hipMemcpyAsync(c->fasta.p, fasta, nbytes, hipMemcpyHostToDevice, c->stream));
    const uint8_t* ptrs[1] = {static_cast<const uint8_t*>(c->fasta.p)};
    const size_t ns[1] = {nbytes};
    if ((rc = dd_sketch_device(c, ptrs, ns, 1, kmin, kmax, static_cast<uint8_t*>(c->regs.p)))) return rc;
    DD_HIP(hipMemcpyAsync(regs, c->regs.p, K * m, hipMemcpyDeviceToHost, c->stream));
    DD_HIP(hipStreamSynchronize(c->stream));
    return DD_OK;
}

int dd_sketch_fasta(dd_ctx* c, const char* path, int kmin, int kmax, uint8_t* regs) {
    if (check_ctx(c)) return DD_EINVAL;
    if (!path) return fail(DD_EINVAL, "null path");
    // A file of some size takes the ingestion pipeline of dd_sketch_files (loader threads reading slices into pinned buffers, the
    // copy under way while they read, .gz inflated on the device): one plain 50 Mbp file 12.7 -> 4.2 ms, 250 Mbp 60 -> 18 ms at
    // log2m 14 (scripts/ab_one_file.py); below 4 MiB this path's one read + one copy is the shorter one (20 kbp: 0.4 against 0.9 ms).
    {
        struct stat sb;
        if (stat(path, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size >= ((off_t)4 << 20)) {
            const char* one[1] = {path};
            return dd_sketch_files(c, one, 1, kmin, kmax, regs, 0);
        }
    }
    FileBuf buf;
    std::string err;
    // (one file: a .gz is inflated by every CPU this process may use -- BGZF blocks, or pieces of a plain member)
    if (!read_fasta_file(path, buf, err, usable_cpus())) return fail(DD_EIO, "%s", err.c_str());
    return dd_sketch_buffer(c, buf.data(), buf.size(), kmin, kmax, regs);
}

// Many FASTA files (plain or .gz, as DandD's species directories hold them,
// /root/reference/lib/species_specifics.py:93); regs is [nfiles][K][m] on the host.  A pipeline:
//   loader threads   read + inflate into pinned host buffers of the context's pool, ahead of the GPU,
//                    bounded by the pool (a directory of whole genomes cannot exhaust host memory);
//   copy stream      H2D of batch b+1 while the compute stream sketches batch b, D2H of batch b-1's
//                    register slabs into a pinned bounce buffer (event-chained, no per-file sync);
//   compute stream   ONE dd_sketch_device launch per batch -- consecutive small files are coalesced until a
//                    batch holds ~128 MB, so a directory of 5 Mbp genomes fills the chip instead of
//                    launching 77 workgroups per file.
// The reference's loop is one genome at a time, each re-read and re-inflated once per k
// (lib/huffman_dandd.py:402-407).
// bounce buffer -> the caller's (pageable) array: one thread moves ~10 GB/s, and a log2m 20 batch is 37 MB per file
static void parallel_copy(uint8_t* dst, const uint8_t* src, size_t n, int nthreads) {
    const size_t kPer = (size_t)8 << 20;
    const int parts = (int)std::min<size_t>((size_t)std::max(1, std::min(nthreads, 8)), (n + kPer - 1) / kPer);
    if (parts <= 1) {
        memcpy(dst, src, n);
        return;
    }
    std::vector<std::thread> th;
    const size_t step = ((n / parts) + 4095) & ~(size_t)4095;
    for (int t = 1; t < parts; ++t) {
        const size_t off = step * t;
        if (off >= n) break;
        th.emplace_back([=] { memcpy(dst + off, src + off, std::min(step, n - off)); });
    }
    memcpy(dst, src, std::min(step, n));
    for (auto& t : th) t.join();
}

static int sketch_files_impl(dd_ctx* c, const char* const* paths, int nfiles, int kmin, int kmax, uint8_t* regs, int nthreads);

int dd_sketch_files(dd_ctx* c, const char* const* paths, int nfiles, int kmin, int kmax, uint8_t* regs,
                    int nthreads) {
    if (check_ctx(c)) return DD_EINVAL;
    c->inflate_retry = false;
    c->inflate_retry_counts = false;
    int rc = sketch_files_impl(c, paths, nfiles, kmin, kmax, regs, nthreads);
    // (DD_INFLATE_STRICT=1: no second try -- the tests and scripts/fuzz_inflate.py set it so that a decoder bug cannot hide
    // behind the fallback)
    if (rc != DD_OK && c->inflate_retry && !getenv("DD_INFLATE_STRICT")) {
        // a BGZF block the device decoder would not take: the whole call again with every .gz inflated on the host, whose
        // decoder either reads the file or says what is wrong with it
        // (only this call -- one damaged file must not cost a long-lived context its device path --, unless it keeps
        // happening: three refusals and the context stays on the host)
        c->inflate_retry = false;
        const bool was = c->no_gpu_inflate;
        c->no_gpu_inflate = true;
        rc = sketch_files_impl(c, paths, nfiles, kmin, kmax, regs, nthreads);
        c->no_gpu_inflate = was || (c->inflate_retry_counts && ++c->inflate_refusals >= 3);
    }
    return rc;
}

// One BGZF file's blocks, found on the host (a walk over the 'BC' size fields: ~800 per 50 Mbp file); the blocks
// themselves are inflated on the device.  false: not a BGZF file the device path takes (the host decoder reads it).
struct BgzfBlock {
    size_t in_off;
    uint32_t in_len, out_len;
    size_t out_off;
};
static bool bgzf_parse(const uint8_t* data, size_t n, std::vector<BgzfBlock>& blks, size_t& out_size, bool& fastq);
static bool bgzf_for_device(const char* path, FileBuf& fb, std::vector<BgzfBlock>& blks, size_t& out_size, bool& fastq) {
    using namespace dd::inflate_detail;
    struct stat sb;
    if (stat(path, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 28 || (size_t)sb.st_size > ((size_t)3 << 30)) return false;
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    uint8_t head[64];
    const size_t got = fread(head, 1, sizeof head, f);
    if (!bgzf_block_size(head, std::max<size_t>(got, 65536))) {   // (only the header is needed here; the size is checked in the walk)
        fclose(f);
        return false;
    }
    const size_t n = (size_t)sb.st_size;
    fb.len = 0;
    bool ok = fb.reserve(n + 16) && fseeko(f, 0, SEEK_SET) == 0 && fread(fb.p, 1, n, f) == n;
    fclose(f);
    if (!ok || !bgzf_parse(fb.p, n, blks, out_size, fastq)) return false;
    fb.len = n;
    return true;
}
// the same for a file whose bytes are in memory already (large files are read in pieces by several loaders)
static bool bgzf_parse(const uint8_t* data, size_t n, std::vector<BgzfBlock>& blks, size_t& out_size, bool& fastq) {
    using namespace dd::inflate_detail;
    blks.clear();
    size_t p = 0, total = 0;
    while (p < n) {
        const size_t bs = bgzf_block_size(data + p, n - p);
        if (!bs) {
            for (size_t q = p; q < n; ++q)
                if (data[q]) return false;   // (trailing zeros are tolerated, as gzread tolerates them)
            break;
        }
        const uint8_t* t = data + p + bs - 4;
        const size_t isize = (size_t)t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
        if (isize > 65536) return false;
        blks.push_back(BgzfBlock{p, (uint32_t)bs, (uint32_t)isize, total});   // (empty members too -- the EOF block --: their CRC-32 and ISIZE are checked like any other's)
        total += isize;
        p += bs;
    }
    if (blks.empty()) return false;
    // (the text rules of dd_fastq.hip and TextJob hold offsets in 32 bits, as the gzip path's do: a text of 4 GiB or more is
    // for the host decoder -- the same bound as gzip_members_parse's)
    if (total >= ((uint64_t)1 << 32) - 65536) return false;
    // FASTQ (reads, not assemblies) starts with '@': look at the first block's text
    {
        uint8_t first[256];
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, 15 + 16) != Z_OK) return false;
        zs.next_in = const_cast<uint8_t*>(data) + blks[0].in_off;
        zs.avail_in = blks[0].in_len;
        zs.next_out = first;
        zs.avail_out = sizeof first;
        const int zr = inflate(&zs, Z_SYNC_FLUSH);
        const size_t made = sizeof first - zs.avail_out;
        inflateEnd(&zs);
        if ((zr != Z_OK && zr != Z_STREAM_END) || !made) return false;
        // (round 5: a text that starts with '@' stays on the device as four-line FASTQ, checked record by record there: dd_fastq.hip)
        fastq = first[0] == '@';
        if (fastq ? getenv("DD_NO_GPU_FASTQ") != nullptr : dd::has_plus_line(first, made)) return false;
    }
    out_size = total;
    return true;
}

// One single-member gzip file for the device path (dd_ginflate.hip: launch_gunzip_members): the raw bytes into `fb`, where
// the deflate data starts, the trailer's CRC-32 and ISIZE.  false: not a file that path takes (small, huge, not gzip,
// FASTQ): the host decoder reads it.  (Whether the file is ONE member only the decoding shows: the device refuses a
// stream whose final block is not followed by exactly the 8 trailer bytes.)
static bool gzip_member_size_ok(size_t n) {
    const size_t min_bytes = (size_t)(getenv("DD_GUNZIP_MIN_KB") ? std::max(1, atoi(getenv("DD_GUNZIP_MIN_KB"))) : 1024) << 10;
    // (below 1 GiB: ISIZE is the text's length mod 2^32 and DNA inflates 3.5-4 x, so a larger member's text may lie beyond
    // 4 GiB, which the device path's 32-bit offsets cannot hold -- a 3 Gbp assembly's .gz is ~0.98 GB; larger files take the
    // host's parallel decoder.  gzip_member_parse looks at the ratio as well, piece_offsets_kernel sums in 64 bits.)
    return n >= min_bytes && n < ((size_t)1 << 30);
}
using dd::GzMember;
using dd::gzip_members_parse;
static bool gzip_member_for_device(const char* path, FileBuf& fb, std::vector<GzMember>& gms) {
    struct stat sb;
    if (stat(path, &sb) != 0 || !S_ISREG(sb.st_mode) || !gzip_member_size_ok((size_t)sb.st_size)) return false;
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    const size_t n = (size_t)sb.st_size;
    fb.len = 0;
    const bool ok = fb.reserve(n + 16) && fread(fb.p, 1, n, f) == n;
    fclose(f);
    if (!ok || !gzip_members_parse(fb.p, n, gms)) return false;
    fb.len = n;
    return true;
}
static int sketch_files_impl(dd_ctx* c, const char* const* paths, int nfiles, int kmin, int kmax, uint8_t* regs, int nthreads) {
    if (nfiles < 0 || (nfiles && (!paths || !regs))) return fail(DD_EINVAL, "null argument");
    if (kmin < 1 || kmax > 64 || kmin > kmax) return fail(DD_EINVAL, "k range %d..%d outside 1..64", kmin, kmax);
    for (int i = 0; i < nfiles; ++i)
        if (!paths[i]) return fail(DD_EINVAL, "null path at index %d", i);
    if (!nfiles) return DD_OK;
    DeviceGuard guard(c->device);
    if (nthreads <= 0) nthreads = std::min(16, usable_cpus());
    const size_t slab = (size_t)(kmax - kmin + 1) << c->p;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    const bool trace = getenv("DD_TRACE_FILES") != nullptr;

    // batch size in files: ~128 MB of FASTA per launch, judged by what is on disk (a .gz inflates ~4x)
    size_t disk_bytes = 0;
    bool any_gz = false;
    for (int i = 0; i < nfiles; ++i) {
        struct stat sb;
        if (stat(paths[i], &sb) == 0 && sb.st_size > 0) {
            const size_t n = (size_t)sb.st_size, L = strlen(paths[i]);
            const bool gz = L > 3 && strcmp(paths[i] + L - 3, ".gz") == 0;
            any_gz |= gz;
            disk_bytes += gz ? 4 * n : n;
        }
    }
    const bool gpu_inflate = !c->no_gpu_inflate && !getenv("DD_NO_GPU_INFLATE");
    // (DD_INFLATE_STRICT=2, tests: a context whose device decoder has been switched off by three refusals says so instead of
    // quietly decoding on the host)
    if (any_gz && c->no_gpu_inflate && getenv("DD_INFLATE_STRICT") && atoi(getenv("DD_INFLATE_STRICT")) >= 2)
        return fail(DD_EIO, "the device decoder is switched off on this context (three refused calls)");
    const size_t avg = std::max<size_t>(1, disk_bytes / (size_t)nfiles);
    // (log2m >= 17: the scatter/sort/replay path runs epoch by epoch over all rows of a launch and wants many rows)
    // (Batches that grow -- 64, 128, 256 MB -- were measured against fixed 128 MB ones once the job tables of several
    // batch shapes could be kept: 20.6-22.9 ms against 19.1 for 10 x 50 Mbp.  Fixed it is.)
    // (a context's FIRST call at log2m >= 17 keeps to 128 MB: the record areas and the pinned register staging are
    // allocated for a batch's rows, and hipMalloc + hipHostMalloc of a 512 MB batch's 5.5 GB + 0.4 GB cost a one-shot
    // `dandd tree -r 20` 0.23 s against 0.06 s; a long-lived context grows them on its second call)
    // (BGZF files inflated on the device: a launch of the inflate kernel takes as long as ONE block does -- 3 ms, the
    // serial walk of a deflate stream by one wave -- whether it holds 1 block or the 4 000 the chip keeps in flight, so
    // those calls batch ~320 MB of text -- five 50 Mbp files, one round of blocks --, in two batches at least (the
    // second's inflate runs under the first's sweep), and a batch waits up to 3 ms for its files instead of leaving
    // with the first one loaded: 64 x 5 Mbp went out as 2 + 8 + 8 + 46 files, four launches one behind the other)
    const size_t kBatchBytes = (size_t)(getenv("DD_BATCH_MB") ? std::max(1, atoi(getenv("DD_BATCH_MB")))
                                        : (c->p >= 17 ? (c->ingest_calls == 0 ? 128 : 512) : (any_gz && gpu_inflate ? 320 : 128))) << 20;
    // (at most 256 files per launch: the loaders' window is two batches of host buffers of 2 MiB at least; with 64,
    // a thousand 100 kbp plasmids took 23 launches of ~3 ms each)
    const size_t kMaxBatchFiles = 256;
    int batch_files = (int)std::max<size_t>(1, std::min<size_t>(kMaxBatchFiles, kBatchBytes / avg));
    const bool full_batches = any_gz && gpu_inflate;
    if (full_batches && nfiles >= 2) {
        // two batches at least (the second's inflate runs under the first's sweep), and EQUAL ones; and rather two batches a
        // quarter larger than three: there are two sets of buffers, so a third batch is issued only when the first has retired
        // -- ten gzip -1 files went out as 4 + 4 at t = 4 ms and 2 at t = 40 ms, whose find + inflate + sweep then ran alone
        // for the call's last 22 of 62 ms (DD_TRACE_FILES; profiles/r05_gunzip.txt)
        int nb = (nfiles + batch_files - 1) / batch_files;
        if (nb == 3 && nfiles * 2 <= batch_files * 5) nb = 2;
        nb = std::max(nb, 2);
        batch_files = (nfiles + nb - 1) / nb;
    }
    // loaders may run two batches ahead of the GPU
    const int window = std::max(nthreads + 2, 2 * batch_files + nthreads);
    // Pinning host memory costs ~0.4 ms per MB: a buffer starts pageable (a one-shot `dandd tree` process never
    // pays that) and is re-made pinned when a LATER call takes it again -- a long-lived context (a pipeline, a
    // benchmark loop) has a fully pinned pool from its third call on.
    while ((int)c->file_pool.size() < window) c->file_pool.push_back(new FileBuf());
    const bool promote = ++c->ingest_calls >= 2;
    if (!c->copy_stream) {
        DD_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        DD_HIP(hipStreamCreateWithFlags(&c->copy_stream_b, hipStreamNonBlocking));
        DD_HIP(hipStreamCreateWithFlags(&c->out_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            DD_HIP(hipEventCreateWithFlags(&c->pipe_h2d[i], hipEventDisableTiming));
            DD_HIP(hipEventCreateWithFlags(&c->pipe_done[i], hipEventDisableTiming));
            DD_HIP(hipEventCreateWithFlags(&c->pipe_d2h[i], hipEventDisableTiming));
        }
    }

    std::vector<int> free_bufs;
    for (int b = 0; b < window; ++b) free_bufs.push_back(b);
    struct Slot {
        int buf = -1;
        std::string err;
        bool ok = true, done = false;
        bool claimed = false, ready = false;  // a loader took the file's buffer / the buffer can be written to
        int pieces_left = 0;
        bool plus = false;                    // a piece of the file holds a line that starts with '+': FASTQ (dd_io.h)
        size_t plain_size = 0;                // > 0: not gzip, read in pieces by several loaders
        bool dev_inflate = false;             // BGZF: the buffer holds the COMPRESSED file, the device inflates its blocks
        size_t out_size = 0;                  // ... into this many bytes of text
        std::vector<BgzfBlock> blks;
        bool gz_raw = false;                  // a large .gz read as it is, in pieces by several loaders (plain_size = its size): meant for the device
        bool dev_gunzip = false;              // gzip members (usually ONE): the buffer holds the compressed file, the device inflates it in pieces
        std::vector<GzMember> gms;
        std::vector<size_t> magic;            // a .gz read in pieces: where 1f 8b 08 stands (member headers?), found piece by piece by the loaders
        bool fastq = false;                   // a device-inflated text that starts with '@': four-line FASTQ, checked and resolved on the device
    };
    std::vector<Slot> slots(nfiles);
    const bool gpu_gunzip = gpu_inflate && !getenv("DD_NO_GPU_GUNZIP");
    const size_t raw_pieces_from = (size_t)(getenv("DD_GUNZIP_PIECES_MB") ? std::max(1, atoi(getenv("DD_GUNZIP_PIECES_MB"))) : 32) << 20;
    // Work items in file order.  A plain file is cut into 8 MiB pieces that different loaders pread into the
    // file's pinned buffer -- the first file of a directory is then in memory after one piece-time instead of
    // one file-time, which is what the GPU waits for at the start; a gzip file is one item (zlib is serial).
    struct Item {
        int file;
        size_t off, len;  // len 0: the whole file through zlib
    };
    std::vector<Item> items;
    const size_t kPiece = (size_t)8 << 20;
    const size_t kFirstPiece = (size_t)2 << 20;
    for (int i = 0; i < nfiles; ++i) {
        struct stat sb;
        unsigned char magic[18] = {0};
        bool plain = false;
        if (stat(paths[i], &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
            if (FILE* f = fopen(paths[i], "rb")) {
                const size_t got = fread(magic, 1, sizeof magic, f);
                plain = got >= 2 && !(magic[0] == 0x1f && magic[1] == 0x8b);
                // a large gzip file -- BGZF, or one that may be ONE member --: its compressed bytes are read
                // like a plain file's, by several loaders (one fread of a 700 MB file held the device path back 150 ms)
                const bool bc = (magic[3] & 4) && magic[12] == 'B' && magic[13] == 'C';   // BGZF's extra field
                if (!plain && got == sizeof magic && magic[2] == 8 && (size_t)sb.st_size >= raw_pieces_from && (size_t)sb.st_size < ((size_t)3 << 30) &&
                    (bc ? gpu_inflate : (gpu_gunzip && gzip_member_size_ok((size_t)sb.st_size))))
                    plain = slots[i].gz_raw = true;
                fclose(f);
            }
        }
        if (plain) {
            slots[i].plain_size = (size_t)sb.st_size;
            // (the first files in finer pieces still: every loader works on file 0 until it is complete, and the GPU
            // sits idle until then)
            const size_t piece = i < 2 ? kFirstPiece : kPiece;
            for (size_t off = 0; off < slots[i].plain_size; off += piece) {
                items.push_back(Item{i, off, std::min(piece, slots[i].plain_size - off)});
                ++slots[i].pieces_left;
            }
        } else {
            items.push_back(Item{i, 0, 0});
            slots[i].pieces_left = 1;
        }
    }
    // A gzip file is one item, but not one thread's worth of work: with fewer .gz files than loaders every one of them
    // is inflated by its share of the CPUs (BGZF blocks / pieces of a plain member, dd_inflate.h); a directory of many
    // .gz files keeps one (libdeflate) thread per file, which is the faster decoder per core.
    int ngz = 0;
    for (int i = 0; i < nfiles; ++i) ngz += slots[i].plain_size == 0;
    const int gz_par = ngz ? std::max(1, nthreads / ngz) : 1;
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<size_t> next{0};
    int consumed = 0;  // files whose host buffer went back to the pool (guarded by mu)
    const int device = c->device;
    auto loader = [&]() {
        (void)hipSetDevice(device);  // pinned allocations belong to the context's device
        for (;;) {
            const size_t w = next.fetch_add(1);
            if (w >= items.size()) return;
            const Item it = items[w];
            Slot& sl = slots[it.file];
            bool mine = false;  // this loader sets the file's buffer up
            {
                // Only files consumed .. consumed+window-1 may hold a buffer: they are consumed in
                // order, so a later file must never take the buffer an earlier one is waiting for.
                std::unique_lock<std::mutex> lk(mu);
                if (!sl.claimed) {
                    sl.claimed = mine = true;
                    cv.wait(lk, [&] { return it.file < consumed + window && !free_bufs.empty(); });
                    sl.buf = free_bufs.back();
                    free_bufs.pop_back();
                } else {
                    cv.wait(lk, [&] { return sl.ready; });
                }
            }
            FileBuf& fb = *c->file_pool[sl.buf];
            std::string err;
            bool ok = true;
            if (mine) {
                if (promote && !fb.pinned && fb.p) {
                    fb.release();
                    fb.pinned = true;
                }
                fb.len = 0;
                if (sl.plain_size) {
                    ok = fb.reserve(sl.plain_size + 16);
                    if (ok) fb.len = sl.plain_size;
                    else err = std::string("out of pinned host memory reading ") + paths[it.file];
                }
                {
                    std::lock_guard<std::mutex> lk(mu);
                    sl.ready = true;
                    if (!ok) sl.ok = false, sl.err = err;
                }
                cv.notify_all();
            }
            if (it.len == 0) {
                if (gpu_inflate && bgzf_for_device(paths[it.file], fb, sl.blks, sl.out_size, sl.fastq)) sl.dev_inflate = true;
                else if (gpu_gunzip && gzip_member_for_device(paths[it.file], fb, sl.gms)) sl.dev_gunzip = true;
                else ok = read_fasta_file(paths[it.file], fb, err, gz_par);
            } else if (ok && fb.cap >= sl.plain_size) {
                FILE* f = fopen(paths[it.file], "rb");
                ok = f && fseeko(f, (off_t)it.off, SEEK_SET) == 0 && fread(fb.p + it.off, 1, it.len, f) == it.len;
                if (!ok) err = std::string("read error on ") + paths[it.file];
                if (f) fclose(f);
            }
            if (sl.dev_gunzip && !sl.out_size) {   // (the members' texts stand one behind the other in the file's text buffer)
                for (const GzMember& gm : sl.gms) sl.out_size += gm.isize;
                sl.fastq = sl.gms[0].fastq;
            }
            // (every loader looks through the piece it has just read -- the bytes are still in its cache -- instead of one
            // of them through the whole file at the end: that pass held every file back 2-3 ms)
            const bool plus_here = it.len && ok && !sl.gz_raw && dd::piece_has_plus_line(fb.p, it.off, it.len);
            std::vector<size_t> magic_here;   // (a .gz read in pieces: every loader scans what it has just read for member headers)
            if (it.len && ok && sl.gz_raw && it.len > 2) dd::gzip_magic_scan(fb.p, it.off, it.off + it.len - 2, magic_here);
            bool last, plus;
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!ok && sl.ok) sl.ok = false, sl.err = err;
                sl.plus |= plus_here;
                sl.magic.insert(sl.magic.end(), magic_here.begin(), magic_here.end());
                plus = sl.plus;
                last = --sl.pieces_left == 0;
            }
            if (last) {
                // a plain file read in pieces is whole now: FASTQ records are resolved before K0 sees the bytes (dd_io.h;
                // read_fasta_file has done the same for the files that came through zlib)
                if (it.len && ok && sl.gz_raw) {
                    // the compressed file is whole: one member for the device, or (FASTQ, an odd header) the host decoder after all
                    if (bgzf_parse(fb.p, sl.plain_size, sl.blks, sl.out_size, sl.fastq)) sl.dev_inflate = true;
                    else if (gpu_gunzip) {
                        // (the positions that straddle two pieces, then all of them in order)
                        for (const Item& o : items)
                            if (o.file == it.file && o.off >= 2)
                                for (size_t q = o.off - 2; q < o.off && q + 2 < sl.plain_size; ++q)
                                    if (fb.p[q] == 0x1f && fb.p[q + 1] == 0x8b && fb.p[q + 2] == 0x08) sl.magic.push_back(q);
                        std::sort(sl.magic.begin(), sl.magic.end());
                        sl.magic.erase(std::unique(sl.magic.begin(), sl.magic.end()), sl.magic.end());
                        if (gzip_members_parse(fb.p, sl.plain_size, sl.gms, &sl.magic)) sl.dev_gunzip = true;
                    }
                    if (!sl.dev_inflate && !sl.dev_gunzip) ok = read_fasta_file(paths[it.file], fb, err, gz_par);
                    if (sl.dev_gunzip) {
                        sl.out_size = 0;
                        for (const GzMember& gm : sl.gms) sl.out_size += gm.isize;
                        sl.fastq = sl.gms[0].fastq;
                    }
                } else if (it.len && ok) {
                    for (const Item& o : items)
                        if (o.file == it.file && !plus) plus = dd::plus_at_piece_start(fb.p, o.off);
                    if (!dd::normalize_records(fb, plus ? 1 : 0)) ok = false, err = std::string("out of host memory reading ") + paths[it.file];
                }
                std::lock_guard<std::mutex> lk(mu);
                if (!ok && sl.ok) sl.ok = false, sl.err = err;
                sl.done = true;
            }
            cv.notify_all();
        }
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; ++t) pool.emplace_back(loader);

    int rc = DD_OK;
    std::string first_err;
    double t_wait = 0;
    int nbatches = 0;
    size_t total_bytes = 0;
    struct InFlight {
        int first = 0, count = 0;   // files of the batch
        bool active = false;
        bool inflated = false;      // some of its files were inflated on the device: the error count is looked at
        struct Member {             // a single-member gzip file inflated on the device: its text's CRC-32 is put together here
            uint32_t chunk0, nchunks, isize, crc;
        };
        std::vector<Member> members;
    };
    InFlight fly[2];
    // hand a finished batch's results to the caller and its host buffers back to the pool
    auto retire = [&](int set) -> int {
        InFlight& f = fly[set];
        if (!f.active) return DD_OK;
        f.active = false;
        const bool arrived = hipEventSynchronize(c->pipe_d2h[set]) == hipSuccess;
        bool refused = arrived && f.inflated && *static_cast<const uint32_t*>(c->pipe_err_host[set].p) != 0;
        if (arrived && !refused) {
            const uint32_t* crcs = static_cast<const uint32_t*>(c->pipe_crc_host[set].p);
            for (const InFlight::Member& m : f.members) {
                uint32_t crc = 0;
                const uint32_t full = dd::crc_x8n(65536u);
                for (uint32_t k = 0; k < m.nchunks; ++k) {
                    const uint32_t len = std::min<uint32_t>(65536u, m.isize - k * 65536u);
                    crc = k ? (dd::crc_multmodp(len == 65536u ? full : dd::crc_x8n(len), crc) ^ crcs[m.chunk0 + k]) : crcs[m.chunk0];
                }
                if (crc != m.crc) {
                    refused = true;
                    *static_cast<uint32_t*>(c->pipe_err_host[set].p) = 1;
                }
            }
        }
        if (arrived && !refused)
            parallel_copy(regs + (size_t)f.first * slab, static_cast<const uint8_t*>(c->pipe_out[set].p), (size_t)f.count * slab, nthreads);
        else
            (void)hipStreamSynchronize(c->copy_stream), (void)hipStreamSynchronize(c->copy_stream_b);  // nothing may still read the host buffers that go back below
        {
            std::lock_guard<std::mutex> lk(mu);
            for (int i = f.first; i < f.first + f.count; ++i) free_bufs.push_back(slots[i].buf);
            consumed = f.first + f.count;
        }
        cv.notify_all();
        // (a failed batch has still given its buffers back: the loaders must never wait for ever)
        if (refused) {
            c->inflate_retry = true;
            // (a CRC that does not match, a block the decoder would not take: counted; pieces that decode but do not add up
            // to the trailer's ISIZE: the file's matter, not the decoder's)
            const uint32_t ecount = *static_cast<const uint32_t*>(c->pipe_err_host[set].p);
            if ((ecount & (dd::kSizeMismatch - 1u)) != 0u || ecount < dd::kSizeMismatch) c->inflate_retry_counts = true;
            return fail(DD_EIO, "ingestion pipeline: %u block(s) / piece(s) / file(s) refused by the device decoder", *static_cast<const uint32_t*>(c->pipe_err_host[set].p));
        }
        return arrived ? DD_OK : fail(DD_EHIP, "ingestion pipeline: D2H failed");
    };
    auto release_unsent = [&](int first, int count) {  // error path: the loaders must never wait for ever
        std::lock_guard<std::mutex> lk(mu);
        for (int i = first; i < first + count; ++i)
            if (slots[i].buf >= 0) free_bufs.push_back(slots[i].buf);
        consumed = first + count;
    };

    int i = 0;
    while (i < nfiles) {
        // the next batch: consecutive files, as many as are wanted and already loaded (at least one)
        const int set = nbatches & 1;
        const double ta = now();
        int count = 0;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return slots[i].done; });
            if (full_batches)
                cv.wait_for(lk, std::chrono::milliseconds(3), [&] {
                    for (int j = i; j < std::min(nfiles, i + batch_files); ++j)
                        if (!slots[j].done) return false;
                    return true;
                });
            // as many as a batch wants and are already loaded (at least one): the GPU is never kept waiting for a
            // full batch; equal batches also let dd_sketch_device reuse its job tables and the buffers below
            while (count < batch_files && i + count < nfiles && slots[i + count].done) ++count;
            // (device-inflated batches: no small batch at the end -- a launch of the inflate kernel over two files' pieces takes as
            // long as one over five, with a quarter of the chip: ten gzip -1 files went out as 4 + 4 + 2 and the last two cost
            // 22 of the call's 60 ms.  What would be left is fewer than half a batch: it joins this one, waited for.)
            if (full_batches && i + count < nfiles && nfiles - (i + count) < (batch_files + 1) / 2) {
                cv.wait(lk, [&] {
                    for (int j = i + count; j < nfiles; ++j)
                        if (!slots[j].done) return false;
                    return true;
                });
                count = nfiles - i;
            }
            // (a directory of small files: batch sizes come from a short list -- powers of two, the full batch, the
            // tail -- so that the job tables of every shape are in the plan cache from the second call on; planning
            // a shape never seen costs ~2 ms of host time with the GPU waiting)
            if (!full_batches && count < batch_files && i + count < nfiles)
                while (count & (count - 1)) count &= count - 1;
        }
        t_wait += now() - ta;
        if (rc == DD_OK) {
            for (int j = i; j < i + count; ++j)
                if (!slots[j].ok) {
                    rc = DD_EIO;
                    first_err = slots[j].err;
                    break;
                }
        }
        if (rc != DD_OK) {  // drain: give every buffer back as its file arrives
            release_unsent(i, count);
            cv.notify_all();
            i += count;
            continue;
        }
        // this buffer set was used by batch nbatches-2: finish that one first
        const double tr = now();
        if ((rc = retire(set)) != DD_OK) {
            first_err = g_err;
            continue;
        }
        const double ti = now();
        std::vector<size_t> sizes(count), offs(count);
        size_t tot = 0;
        size_t gz_tot = 0, njobs = 0;
        std::vector<size_t> gz_off(count, 0);
        // (single-member gzip files: the finder looks at one range of `guess_bits` per piece; 32 KiB of compressed data are ~2
        // deflate blocks of gzip -6 DNA, so a third of every range is scanned before its first block start turns up)
        // (files of 48 MB and more take 64 KiB ranges: half as many links in the chain of windows, which one workgroup per
        // file walks at ~7 us a piece -- 1 x 400 Mbp: 5.4 -> 6.9 Gbp/s, 3 x 300 Mbp: 7.2 -> 8.1)
        auto guess_bits_of = [&](size_t file_bytes) {
            // (round 5, with the windows composed in two levels: 16 KiB up to 400 MB of compressed file (was 32, and 64 from 48 MB: one
            // 400 Mbp member 7.5-7.7 -> 8.3 Gbp/s at gzip -1, 9.7-10.1 -> 10.3 at gzip -6) -- ten 50 Mbp gzip -1 files 7.2 -> 7.9-8.3
            // Gbp/s with 16 / 8 KiB, gzip -6 11.1 -> 11.5 / 11.3, 64 x 5 Mbp 9.1 -> 9.2 / 9.6; profiles/r05_gunzip.txt)
            return (size_t)(getenv("DD_GUNZIP_GUESS_KB") ? std::max(4, atoi(getenv("DD_GUNZIP_GUESS_KB"))) : (file_bytes >= ((size_t)400 << 20) ? 128 : 16)) << 13;
        };
        // a range's symbols: 5 x its compressed bytes (DNA inflates 3-4 x) + 32 Ki; a piece that needs more takes the arena -- and a
        // second and third pass of the decoder over it (count, then write).  Round 6: the factor follows the MEMBER's own ratio
        // (twice ISIZE / compressed length: a piece runs from the first block start of its range to the first of the next, up to
        // two ranges' worth of bits) when that is larger -- four-line FASTQ whose quality text compresses well inflates 6 x, most
        // pieces overflowed, and inflate_kernel<1> + <2> cost a batch 12.5 ms beside the 9.9 of <3> (profiles/r06_ingest.txt);
        // capped at 64 x: beyond that (runs of N) the arena is the right place
        auto range_syms_of = [&](size_t guess_bits, size_t isize, size_t clen) {
            const double ratio = clen ? 2.0 * (double)isize / (double)clen : 0.0;
            const double f = std::min(64.0, std::max(5.0, ratio));
            return (size_t)(f * (double)(guess_bits / 8)) + 32768;
        };
        size_t nmem = 0, npieces = 0, nchunks = 0, sym_tot = 0, win_tot = 0, ngroups = 0;
        for (int j = 0; j < count; ++j) {
            const Slot& sj = slots[i + j];
            sizes[j] = (sj.dev_inflate || sj.dev_gunzip) ? sj.out_size : c->file_pool[sj.buf]->size();
            offs[j] = tot;
            tot += align_up(sizes[j] + 16, 256);
            if (sj.dev_inflate || sj.dev_gunzip) {
                gz_off[j] = gz_tot;
                gz_tot += align_up(c->file_pool[sj.buf]->size() + 16, 256);
                njobs += sj.blks.size();
            }
            if (sj.dev_gunzip)
                for (const GzMember& gm : sj.gms) {   // every member a "file" of the decoder's tables
                    const size_t guess_bits = guess_bits_of(gm.end - gm.first_bit / 8), range_syms = range_syms_of(guess_bits, gm.isize, gm.end - gm.first_bit / 8);
                    const size_t bits = gm.end * 8 - gm.first_bit;
                    const size_t ng = (bits + guess_bits - 1) / guess_bits;
                    ++nmem;
                    npieces += ng;
                    nchunks += (gm.isize + 65535u) / 65536u;
                    sym_tot += align_up(ng * range_syms * 2 + 256, 256) + align_up((size_t)gm.isize * 2 + 256, 256);   // the ranges' symbols, the arena
                    win_tot += align_up(dd::gunzip_window_bytes(ng), 256);
                    ngroups += (ng + dd::kPieceGroup - 1) / dd::kPieceGroup;
                }
        }
        // the piece tables of the batch's single-member gzip files, one block of device memory: RawFile[nmem],
        // starts (u64) / lens / offs / over / abase [npieces], chunk0 [nmem + 1], crcs [nchunks]
        const size_t raw_files = align_up(nmem * sizeof(dd::RawFile), 256), raw_u32 = align_up(npieces * 4, 256), raw_chunk0 = align_up((nmem + 1) * 4, 256);
        const size_t raw_bytes = raw_files + 6 * raw_u32 + raw_chunk0 + align_up(nchunks * 4, 256);   // (starts are 64-bit: two of the six)
        // (a 3 Gbp assembly's .gz takes 16 GB of symbol area per buffer set: a long-lived context gives that back when a later
        // batch needs an eighth of it or less -- not at the end of every call: hipFree + hipMalloc of 16 GB per call cost an
        // occasional 2 s.  The set's previous batch has been retired above: nothing reads the buffer any more.)
        if (c->pipe_sym[set].cap > ((size_t)4 << 30) && sym_tot <= c->pipe_sym[set].cap / 8) {
            c->pipe_sym[set].release();
            c->pipe_win[set].release();
        }
        if (nmem && ((rc = c->pipe_gz[set].reserve(gz_tot + 16)) != DD_OK || (rc = c->pipe_sym[set].reserve(sym_tot)) != DD_OK ||
                     (rc = c->pipe_win[set].reserve(win_tot)) != DD_OK || (rc = c->pipe_raw[set].reserve(raw_bytes)) != DD_OK ||
                     (rc = c->pipe_raw_host[set].reserve(raw_files + raw_chunk0)) != DD_OK || (rc = c->pipe_crc_host[set].reserve(nchunks * 4 + 256)) != DD_OK ||
                     (rc = c->pipe_err[set].reserve(256)) != DD_OK || (rc = c->pipe_err_host[set].reserve(256)) != DD_OK)) {
            // (10 x the compressed bytes + 2 x the text of symbol area, 64 KiB of windows per piece: a device that cannot
            // give that can still sketch the file -- the call runs again with the host decoder; not a strike)
            if (rc == DD_ENOMEM) c->inflate_retry = true;
            first_err = g_err;
            release_unsent(i, count);
            cv.notify_all();
            i += count;
            continue;
        }
        total_bytes += tot;
        if (njobs && ((rc = c->pipe_gz[set].reserve(gz_tot + 16)) != DD_OK || (rc = c->pipe_jobs[set].reserve(njobs * sizeof(dd::InflateJob))) != DD_OK ||
                      (rc = c->pipe_jobs_host[set].reserve(njobs * sizeof(dd::InflateJob))) != DD_OK || (rc = c->pipe_err[set].reserve(256)) != DD_OK ||
                      (rc = c->pipe_err_host[set].reserve(256)) != DD_OK)) {
            first_err = g_err;
            release_unsent(i, count);
            cv.notify_all();
            i += count;
            continue;
        }
        // (growing a device buffer frees the old one: the compute stream may still read it for the batch before
        // last only if that batch has not been retired -- it has, above)
        if ((rc = c->pipe_fasta[set].reserve(tot + 16)) != DD_OK || (rc = c->pipe_regs[set].reserve((size_t)count * slab)) != DD_OK ||
            (rc = c->pipe_out[set].reserve((size_t)count * slab)) != DD_OK) {
            first_err = g_err;
            release_unsent(i, count);
            cv.notify_all();
            i += count;
            continue;
        }
        hipError_t e = hipSuccess;
        hipStream_t cs = ((njobs || nmem) && set) ? c->copy_stream_b : c->copy_stream;
        std::vector<const uint8_t*> ptrs(count);
        dd::InflateJob* jobs_host = njobs ? static_cast<dd::InflateJob*>(c->pipe_jobs_host[set].p) : nullptr;
        size_t nj = 0;
        dd::RawFile* raw_host = nmem ? static_cast<dd::RawFile*>(c->pipe_raw_host[set].p) : nullptr;
        uint32_t* chunk0_host = nmem ? reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(c->pipe_raw_host[set].p) + raw_files) : nullptr;
        size_t mi = 0, piece_at = 0, chunk_at = 0, sym_at = 0, win_at = 0, group_at = 0;
        std::vector<InFlight::Member> members;
        for (int j = 0; j < count && e == hipSuccess; ++j) {
            ptrs[j] = static_cast<const uint8_t*>(c->pipe_fasta[set].p) + offs[j];
            const Slot& sj = slots[i + j];
            const FileBuf& fbj = *c->file_pool[sj.buf];
            if (sj.dev_inflate) {
                // the COMPRESSED file goes over PCIe (a quarter of the text); its blocks are inflated into ptrs[j] below
                uint8_t* gz = static_cast<uint8_t*>(c->pipe_gz[set].p) + gz_off[j];
                e = hipMemcpyAsync(gz, fbj.data(), fbj.size(), hipMemcpyHostToDevice, cs);
                for (const BgzfBlock& b : sj.blks)
                    jobs_host[nj++] = dd::InflateJob{gz + b.in_off, b.in_len, b.out_len, const_cast<uint8_t*>(ptrs[j]) + b.out_off};
            } else if (sj.dev_gunzip) {
                uint8_t* gz = static_cast<uint8_t*>(c->pipe_gz[set].p) + gz_off[j];
                e = hipMemcpyAsync(gz, fbj.data(), fbj.size(), hipMemcpyHostToDevice, cs);
                size_t text_at = 0;   // the members' texts one behind the other
                for (const GzMember& gm : sj.gms) {
                    dd::RawFile& rf = raw_host[mi];
                    const size_t guess_bits = guess_bits_of(gm.end - gm.first_bit / 8), range_syms = range_syms_of(guess_bits, gm.isize, gm.end - gm.first_bit / 8);
                    const size_t ng = (gm.end * 8 - gm.first_bit + guess_bits - 1) / guess_bits;
                    rf.in = gz;                       // (positions are the FILE's: its bytes start on a 256-byte boundary, a member's need not)
                    rf.in_len = (uint32_t)gm.end;     // ... and the member ends here: CRC-32 and ISIZE right behind its final block
                    rf.first_bit = gm.first_bit;
                    rf.guess_bits = (uint32_t)guess_bits;
                    rf.nguess = (uint32_t)ng;
                    rf.piece0 = (uint32_t)piece_at;
                    rf.isize = gm.isize;
                    rf.sym = reinterpret_cast<uint16_t*>(static_cast<uint8_t*>(c->pipe_sym[set].p) + sym_at);
                    rf.range_syms = (uint32_t)range_syms;
                    rf.arena = reinterpret_cast<uint16_t*>(static_cast<uint8_t*>(c->pipe_sym[set].p) + sym_at + align_up(ng * range_syms * 2 + 256, 256));
                    rf.windows = static_cast<uint8_t*>(c->pipe_win[set].p) + win_at;
                    rf.group0 = (uint32_t)group_at;
                    rf.ngroups = (uint32_t)((ng + dd::kPieceGroup - 1) / dd::kPieceGroup);
                    rf.text = const_cast<uint8_t*>(ptrs[j]) + text_at;
                    chunk0_host[mi] = (uint32_t)chunk_at;
                    members.push_back(InFlight::Member{(uint32_t)chunk_at, (gm.isize + 65535u) / 65536u, gm.isize, gm.crc});
                    piece_at += ng;
                    chunk_at += (gm.isize + 65535u) / 65536u;
                    sym_at += align_up(ng * range_syms * 2 + 256, 256) + align_up((size_t)gm.isize * 2 + 256, 256);
                    win_at += align_up(dd::gunzip_window_bytes(ng), 256);
                    group_at += (ng + dd::kPieceGroup - 1) / dd::kPieceGroup;
                    text_at += gm.isize;
                    ++mi;
                }
            } else if (sizes[j]) {
                e = hipMemcpyAsync(const_cast<uint8_t*>(ptrs[j]), fbj.data(), sizes[j], hipMemcpyHostToDevice, cs);
            }
        }
        if ((njobs || nmem) && e == hipSuccess) e = hipMemsetAsync(c->pipe_err[set].p, 0, 4, cs);
        // (round 5, measured and dropped: the second batch's decoders BEHIND the first's -- an event between the two copy streams --
        // instead of side by side: ten gzip -1 files 55.0 -> 60.4 ms, gzip -6 43.2 -> 47.5, 64 x 5 Mbp 33.9 -> 38.8: a lone
        // inflate launch cannot fill the chip, its time is its longest piece's, and two launches hide each other's tails)
        if (nmem && e == hipSuccess) {
            // block starts -> piece lengths -> offsets -> symbols -> windows -> text -> CRC-32 of every 64 KiB (dd_ginflate.hip)
            chunk0_host[nmem] = (uint32_t)chunk_at;
            uint8_t* rb = static_cast<uint8_t*>(c->pipe_raw[set].p);
            e = hipMemcpyAsync(rb, raw_host, nmem * sizeof(dd::RawFile), hipMemcpyHostToDevice, cs);
            if (e == hipSuccess) e = hipMemcpyAsync(rb + raw_files + 6 * raw_u32, chunk0_host, (nmem + 1) * 4, hipMemcpyHostToDevice, cs);
            if (e == hipSuccess) {
                uint32_t* crcs_dev = reinterpret_cast<uint32_t*>(rb + raw_files + 6 * raw_u32 + raw_chunk0);
                dd::launch_gunzip_members(reinterpret_cast<const dd::RawFile*>(rb), (int)nmem, (int)npieces, (int)ngroups, (int)nchunks, reinterpret_cast<uint64_t*>(rb + raw_files),
                                          reinterpret_cast<uint32_t*>(rb + raw_files + 2 * raw_u32), raw_u32 / 4,
                                          reinterpret_cast<const uint32_t*>(rb + raw_files + 6 * raw_u32), crcs_dev, static_cast<uint32_t*>(c->pipe_err[set].p), cs);
                e = hipGetLastError();
                if (e == hipSuccess) e = hipMemcpyAsync(c->pipe_crc_host[set].p, crcs_dev, nchunks * 4, hipMemcpyDeviceToHost, cs);
            }
        }
        if (njobs && e == hipSuccess) {
            e = hipMemcpyAsync(c->pipe_jobs[set].p, jobs_host, njobs * sizeof(dd::InflateJob), hipMemcpyHostToDevice, cs);
            if (e == hipSuccess) {
                dd::launch_inflate_bgzf(static_cast<const dd::InflateJob*>(c->pipe_jobs[set].p), (int)njobs, static_cast<uint32_t*>(c->pipe_err[set].p), cs);
                e = hipGetLastError();
            }
        }
        if ((njobs || nmem) && e == hipSuccess) {
            // kseq's record rules over the texts the device has just inflated (dd_fastq.hip): no line of a FASTA-classed text may
            // start with '+'; a FASTQ-classed text must be four-line FASTQ, and its '+' and quality lines become header lines
            std::vector<dd::TextJob> tj;
            size_t blocks = 0, words = 0;
            bool any_fastq = false;
            for (int j = 0; j < count; ++j) {
                const Slot& sj = slots[i + j];
                if (!(sj.dev_inflate || sj.dev_gunzip) || !sizes[j]) continue;
                dd::TextJob t{};
                t.text = const_cast<uint8_t*>(ptrs[j]);
                if (sizes[j] >= ((uint64_t)1 << 32)) {   // (bgzf_parse / gzip_members_parse refuse such files: never reached)
                    rc = fail(DD_EINVAL, "a device-inflated text of %zu bytes does not fit the text rules' 32-bit offsets", (size_t)sizes[j]);
                    first_err = g_err;
                    e = hipErrorInvalidValue;
                    break;
                }
                t.n = (uint32_t)sizes[j];
                t.fastq = sj.fastq ? 1u : 0u;
                t.block0 = (uint32_t)blocks;
                const size_t nb4k = (sizes[j] + 4095) / 4096;
                blocks += nb4k;
                if (sj.fastq) {
                    any_fastq = true;
                    t.nl_cap = (uint32_t)(sizes[j] / 8 + 16);
                    // (offsets in words from the table's end; turned into pointers below)
                    t.blk_count = reinterpret_cast<uint32_t*>(words);
                    t.nl = reinterpret_cast<uint32_t*>(words + nb4k);
                    t.nl_total = reinterpret_cast<uint32_t*>(words + nb4k + t.nl_cap);
                    words += nb4k + t.nl_cap + 4;
                }
                tj.push_back(t);
            }
            if (!tj.empty() && e == hipSuccess) {
                const size_t tab = align_up(tj.size() * sizeof(dd::TextJob), 256);
                if ((rc = c->pipe_txt[set].reserve(tab + words * 4 + 256)) != DD_OK || (rc = c->pipe_txt_host[set].reserve(tab)) != DD_OK) {
                    first_err = g_err;
                    e = hipErrorOutOfMemory;
                } else {
                    uint32_t* base = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(c->pipe_txt[set].p) + tab);
                    for (dd::TextJob& t : tj)
                        if (t.fastq) {
                            t.blk_count = base + reinterpret_cast<size_t>(t.blk_count);
                            t.nl = base + reinterpret_cast<size_t>(t.nl);
                            t.nl_total = base + reinterpret_cast<size_t>(t.nl_total);
                        }
                    memcpy(c->pipe_txt_host[set].p, tj.data(), tj.size() * sizeof(dd::TextJob));
                    e = hipMemcpyAsync(c->pipe_txt[set].p, c->pipe_txt_host[set].p, tj.size() * sizeof(dd::TextJob), hipMemcpyHostToDevice, cs);
                    if (e == hipSuccess) {
                        dd::launch_text_rules(static_cast<const dd::TextJob*>(c->pipe_txt[set].p), (int)tj.size(), (uint32_t)blocks, any_fastq,
                                              static_cast<uint32_t*>(c->pipe_err[set].p), cs);
                        e = hipGetLastError();
                    }
                }
            }
        }
        if ((njobs || nmem) && e == hipSuccess) e = hipMemcpyAsync(c->pipe_err_host[set].p, c->pipe_err[set].p, 4, hipMemcpyDeviceToHost, cs);
        if (c->text_sink && e == hipSuccess) {
            // dd_inflate_files: every file's text as it stands in the buffer K0 reads -- inflated by the kernels above where
            // the device decoder took the file -- back to the caller, on the stream that made it
            for (int j = 0; j < count && e == hipSuccess; ++j) {
                c->text_sink->lens[i + j] = sizes[j];
                if (sizes[j] > c->text_sink->caps[i + j]) c->text_sink->short_buffer = true;
                else if (sizes[j]) e = hipMemcpyAsync(c->text_sink->out[i + j], ptrs[j], sizes[j], hipMemcpyDeviceToHost, cs);
            }
        }
        if (e == hipSuccess) e = hipEventRecord(c->pipe_h2d[set], cs);
        if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->pipe_h2d[set], 0);
        if (e == hipSuccess) {
            rc = dd_sketch_device(c, ptrs.data(), sizes.data(), count, kmin, kmax, static_cast<uint8_t*>(c->pipe_regs[set].p));
            if (rc != DD_OK) first_err = g_err;
        }
        if (e == hipSuccess && rc == DD_OK) e = hipEventRecord(c->pipe_done[set], c->stream);
        if (e == hipSuccess && rc == DD_OK) e = hipStreamWaitEvent(c->out_stream, c->pipe_done[set], 0);
        if (e == hipSuccess && rc == DD_OK)
            e = hipMemcpyAsync(c->pipe_out[set].p, c->pipe_regs[set].p, (size_t)count * slab, hipMemcpyDeviceToHost, c->out_stream);
        if (e == hipSuccess && rc == DD_OK) e = hipEventRecord(c->pipe_d2h[set], c->out_stream);
        if (e != hipSuccess && rc == DD_OK) {
            rc = DD_EHIP;
            first_err = std::string("ingestion pipeline: ") + hipGetErrorString(e);
        }
        if (rc != DD_OK) {
            (void)hipStreamSynchronize(c->copy_stream), (void)hipStreamSynchronize(c->copy_stream_b);  // nothing may still read the host buffers
            (void)hipStreamSynchronize(c->stream);
            (void)hipStreamSynchronize(c->out_stream);
            release_unsent(i, count);
            cv.notify_all();
        } else {
            fly[set].first = i;
            fly[set].count = count;
            fly[set].active = true;
            fly[set].inflated = njobs != 0 || nmem != 0;
            fly[set].members = std::move(members);
            ++nbatches;
        }
        if (trace)
            fprintf(stderr, "[dd_sketch_files] t=%.2f batch %d: files %d..%d (%.1f MB): waited %.2f ms for loaders, %.2f ms retiring, %.2f ms issuing\n",
                    now() - t_begin, nbatches - 1, i, i + count - 1, tot / 1e6, tr - ta, ti - tr, now() - ti);
        i += count;
    }
    // batches retire in order: the older of the two first
    for (int k2 = 0; k2 < 2; ++k2) {
        const int set = (nbatches + k2) & 1;
        if (rc == DD_OK) {
            if ((rc = retire(set)) != DD_OK) first_err = g_err;
        } else if (fly[set].active) {
            (void)hipStreamSynchronize(c->copy_stream), (void)hipStreamSynchronize(c->copy_stream_b);
            release_unsent(fly[set].first, fly[set].count);
            fly[set].active = false;
            cv.notify_all();
        }
    }
    for (auto& t : pool) t.join();
    c->ingest_ms[0] = now() - t_begin;
    c->ingest_ms[1] = t_wait;
    c->ingest_ms[2] = nbatches;
    c->ingest_ms[3] = (double)total_bytes;
    if (trace)
        fprintf(stderr, "[dd_sketch_files] %d files, %d batches of <= %d files, %.1f ms (%.1f ms waiting for loaders), %.1f MB\n",
                nfiles, nbatches, batch_files, c->ingest_ms[0], t_wait, total_bytes / 1e6);
    if (rc != DD_OK) return fail(rc, "%s", first_err.c_str());
    return DD_OK;
}

int dd_last_ingest_stats(dd_ctx* c, double* wall_ms, double* loader_wait_ms, int* batches, uint64_t* bytes) {
    if (check_ctx(c)) return DD_EINVAL;
    if (wall_ms) *wall_ms = c->ingest_ms[0];
    if (loader_wait_ms) *loader_wait_ms = c->ingest_ms[1];
    if (batches) *batches = (int)c->ingest_ms[2];
    if (bytes) *bytes = (uint64_t)c->ingest_ms[3];
    return DD_OK;
}

// The ingestion pipeline's text, for checking the device decoders byte by byte (tests/test_gpu_parity.py, scripts/fuzz_inflate.py):
// one dd_sketch_files pass (k = 21 only) whose batches also copy every file's text -- as K0 is about to read it -- to the caller.
int dd_inflate_files(dd_ctx* c, const char* const* paths, int nfiles, uint8_t* const* out, const size_t* caps, size_t* lens, int nthreads) {
    if (check_ctx(c)) return DD_EINVAL;
    if (nfiles < 0 || (nfiles && (!paths || !out || !caps || !lens))) return fail(DD_EINVAL, "null argument");
    for (int i = 0; i < nfiles; ++i) {
        if (!out[i] && caps[i]) return fail(DD_EINVAL, "null buffer at index %d", i);
        lens[i] = 0;
    }
    std::vector<uint8_t> regs((size_t)nfiles << c->p);
    dd_ctx::TextSink sink{out, caps, lens, false};
    c->text_sink = &sink;
    const int rc = dd_sketch_files(c, paths, nfiles, 21, 21, regs.data(), nthreads);
    c->text_sink = nullptr;
    if (rc != DD_OK) return rc;
    if (sink.short_buffer) return fail(DD_EINVAL, "a buffer is smaller than its file's text (the sizes needed are in lens[])");
    return DD_OK;
}

// ------------------------------------------------------------------------- exact count
int dd_exact_count_device(dd_ctx* c, const uint8_t* const* fasta_dev, const size_t* nbytes, int n, int k,
                          uint64_t* distinct) {
    if (check_ctx(c)) return DD_EINVAL;
    if (n < 0 || !distinct || (n && (!fasta_dev || !nbytes))) return fail(DD_EINVAL, "null argument");
    if (k < 1 || k > 64) return fail(DD_EINVAL, "k=%d outside 1..64", k);
    for (int g = 0; g < n; ++g) {
        if (nbytes[g] && !fasta_dev[g]) return fail(DD_EINVAL, "input %d: null buffer", g);
        if (reinterpret_cast<uintptr_t>(fasta_dev[g]) & 15)
            return fail(DD_EINVAL, "input %d: device buffer must be 16-byte aligned", g);
    }
    *distinct = 0;
    if (!n) return DD_OK;
    DeviceGuard guard(c->device);
    hipStream_t st = c->stream;
    int rc;

    // token streams (K0), laid out like dd_sketch_device does
    std::vector<size_t> off_codes(n), off_bad(n), off_ntok(n), off_scratch(n);
    size_t tot = 0, scratch_tot = 0, slots = 0, max_segments = 0, max_chunks = 0;
    std::vector<unsigned long long> base(n);
    for (int g = 0; g < n; ++g) {
        off_codes[g] = tot;
        tot += align_up(dd::codes_words(nbytes[g]) * 4, 256);
        off_bad[g] = tot;
        tot += align_up(dd::bad_words(nbytes[g]) * 4, 256);
        off_ntok[g] = tot;
        tot += 256;
        off_scratch[g] = scratch_tot;
        scratch_tot += align_up(dd::pack_scratch_bytes(nbytes[g]), 256);
        base[g] = slots;
        const size_t segs = (nbytes[g] + dd::kSegTokens - 1) / dd::kSegTokens;
        slots += segs * dd::kSegTokens;
        max_segments = std::max(max_segments, segs);
        max_chunks = std::max(max_chunks, dd::pack_chunks(nbytes[g]));
    }
    if (!slots) return DD_OK;
    const bool wide = k > 32;
    const size_t arrays = wide ? 4 : 2;
    // HBM for the k-mer arrays (keys + the sort's other half): everything at once when that fits the budget,
    // else in passes over disjoint parts of the k-mer space (below).  KMC unions arbitrarily many databases
    // (/root/reference/lib/sketch_classes.py:453-465); so must this.
    size_t budget = (size_t)24 << 30;
    if (const char* e = getenv("DD_EXACT_MB")) budget = (size_t)std::max(1, atoi(e)) << 20;
    const bool single = arrays * slots * sizeof(uint64_t) <= budget;
    const size_t cap = single ? slots : std::max<size_t>(budget / (arrays * sizeof(uint64_t)), 4096);  // k-mers per pass
    if ((rc = c->tokens.reserve(tot))) return rc;
    if ((rc = c->scratch.reserve(scratch_tot))) return rc;
    char* tb = static_cast<char*>(c->tokens.p);
    char* sb = static_cast<char*>(c->scratch.p);

    std::vector<dd::PackGenome> ptab(n);
    std::vector<dd::ExactGenome> etab(n);
    for (int g = 0; g < n; ++g) {
        dd::TokenStream ts{reinterpret_cast<uint32_t*>(tb + off_codes[g]), reinterpret_cast<uint32_t*>(tb + off_bad[g]),
                           reinterpret_cast<unsigned long long*>(tb + off_ntok[g])};
        ptab[g] = dd::PackGenome{fasta_dev[g], nbytes[g], dd::pack_chunks(nbytes[g]),
                                 reinterpret_cast<long long*>(sb + off_scratch[g]), ts};
        etab[g] = dd::ExactGenome{ts.codes, ts.bad, ts.ntok, base[g]};
    }
    const size_t pbytes = align_up(sizeof(dd::PackGenome) * n, 256), ebytes = align_up(sizeof(dd::ExactGenome) * n, 256);
    if ((rc = c->tables.reserve(pbytes + ebytes))) return rc;
    DD_HIP(hipEventSynchronize(c->stage_free));
    if ((rc = c->stage.reserve(pbytes + ebytes))) return rc;
    char* tdev = static_cast<char*>(c->tables.p);
    if ((rc = upload(c, c->stage, tdev, ptab.data(), sizeof(dd::PackGenome) * n, 0))) return rc;
    if ((rc = upload(c, c->stage, tdev + pbytes, etab.data(), sizeof(dd::ExactGenome) * n, pbytes))) return rc;
    DD_HIP(hipEventRecord(c->stage_free, st));
    {
        Span sp(c, DD_KERNEL_PACK);
        dd::launch_pack_batch(reinterpret_cast<const dd::PackGenome*>(tdev), n, max_chunks, st);
    }
    const dd::ExactGenome* etab_dev = reinterpret_cast<const dd::ExactGenome*>(tdev + pbytes);

    // layout of the k-mer workspace for `cap` keys: counters (256 B) | histogram (32 KiB) | lo | lo_alt [| hi | hi_alt] | temp
    const size_t hist_bytes = (size_t)dd::kExactBins * sizeof(unsigned long long);
    auto carve = [&](size_t keys, size_t temp_bytes, unsigned long long*& counters, unsigned long long*& hist, uint64_t*& lo,
                     uint64_t*& lo_alt, uint64_t*& hi, uint64_t*& hi_alt, void*& temp) -> int {
        const size_t stride = align_up(keys * sizeof(uint64_t), 256);
        int r = c->exact.reserve(256 + hist_bytes + arrays * stride + temp_bytes + 256);
        if (r) return r;
        char* eb = static_cast<char*>(c->exact.p);
        counters = reinterpret_cast<unsigned long long*>(eb);
        hist = reinterpret_cast<unsigned long long*>(eb + 256);
        char* kb = eb + 256 + hist_bytes;
        lo = reinterpret_cast<uint64_t*>(kb);
        lo_alt = reinterpret_cast<uint64_t*>(kb + stride);
        hi = wide ? reinterpret_cast<uint64_t*>(kb + 2 * stride) : nullptr;
        hi_alt = wide ? reinterpret_cast<uint64_t*>(kb + 3 * stride) : nullptr;
        temp = kb + arrays * stride;
        return DD_OK;
    };
    unsigned long long *counters = nullptr, *hist = nullptr;
    uint64_t *lo = nullptr, *lo_alt = nullptr, *hi = nullptr, *hi_alt = nullptr;
    void* temp = nullptr;

    if (single) {
        const size_t key_bytes = slots * sizeof(uint64_t);
        const size_t temp_bytes = dd::exact_sort_temp_bytes(slots, k);
        if ((rc = carve(slots, temp_bytes, counters, hist, lo, lo_alt, hi, hi_alt, temp))) return rc;
        DD_HIP(hipMemsetAsync(counters, 0, 256, st));
        DD_HIP(hipMemsetAsync(lo, 0xFF, key_bytes, st));  // unwritten slots read as the all-ones sentinel
        if (wide) DD_HIP(hipMemsetAsync(hi, 0xFF, key_bytes, st));
        dd::launch_kmer_extract(etab_dev, n, max_segments, k, c->canonical, lo, hi, counters, st);
        DD_HIP(hipGetLastError());
        DD_HIP(dd::launch_exact_sort_count(lo, hi, lo_alt, hi_alt, slots, k, temp, temp_bytes, counters, st));
        unsigned long long h[3] = {0, 0, 0};
        DD_HIP(hipMemcpyAsync(h, counters, sizeof h, hipMemcpyDeviceToHost, st));
        DD_HIP(hipStreamSynchronize(st));
        // the all-ones group holds the sentinels of unwritten slots and/or genuine T^k k-mers
        const bool sentinel_present = h[0] < (unsigned long long)slots, all_t = h[1] != 0;
        *distinct = h[2] - ((sentinel_present || all_t) ? 1 : 0) + (all_t ? 1 : 0);
        return DD_OK;
    }

    // ---- more k-mers than the budget holds: passes over disjoint parts of the k-mer space ------------
    // The k-mer space is cut into 4096 bins by a mix of the k-mer itself (equal k-mers share a bin), a
    // counting pass sizes the bins, consecutive bins are grouped into passes of at most `cap` k-mers, and every
    // pass extracts (densely), sorts and counts only its own bins: distinct = sum over passes.
    size_t temp_bytes = dd::exact_sort_temp_bytes(cap, k);
    if ((rc = carve(cap, temp_bytes, counters, hist, lo, lo_alt, hi, hi_alt, temp))) return rc;
    DD_HIP(hipMemsetAsync(counters, 0, 256 + hist_bytes, st));
    dd::launch_kmer_extract(etab_dev, n, max_segments, k, c->canonical, lo, hi, counters, st, 1, hist, 0, 0);
    DD_HIP(hipGetLastError());
    std::vector<unsigned long long> bins(dd::kExactBins);
    DD_HIP(hipMemcpyAsync(bins.data(), hist, hist_bytes, hipMemcpyDeviceToHost, st));
    DD_HIP(hipStreamSynchronize(st));
    const unsigned long long biggest = *std::max_element(bins.begin(), bins.end());
    size_t pass_cap = cap;
    if (biggest > pass_cap) {
        // one bin alone is over the budget (one k-mer repeated billions of times lands in one bin): the arrays
        // grow to hold it if the device has the room, otherwise this input cannot be counted here
        pass_cap = (size_t)biggest;
        temp_bytes = dd::exact_sort_temp_bytes(pass_cap, k);
        if ((rc = carve(pass_cap, temp_bytes, counters, hist, lo, lo_alt, hi, hi_alt, temp)))
            return fail(DD_ENOMEM, "exact count: one part of the k-mer space holds %llu k-mers, more than fits in HBM", biggest);
    }
    unsigned long long total = 0;
    int npass = 0;
    for (uint32_t b0 = 0; b0 < (uint32_t)dd::kExactBins;) {
        unsigned long long in_pass = 0;
        uint32_t b1 = b0;
        while (b1 < (uint32_t)dd::kExactBins && in_pass + bins[b1] <= pass_cap) in_pass += bins[b1++];
        if (in_pass) {
            DD_HIP(hipMemsetAsync(counters, 0, 256, st));
            dd::launch_kmer_extract(etab_dev, n, max_segments, k, c->canonical, lo, hi, counters, st, 2, hist, b0, b1);
            DD_HIP(hipGetLastError());
            // (every slot below in_pass is written: no sentinel, T^k is an ordinary value here)
            DD_HIP(dd::launch_exact_sort_count(lo, hi, lo_alt, hi_alt, (size_t)in_pass, k, temp, temp_bytes, counters, st));
            unsigned long long h[4] = {0, 0, 0, 0};
            DD_HIP(hipMemcpyAsync(h, counters, sizeof h, hipMemcpyDeviceToHost, st));
            DD_HIP(hipStreamSynchronize(st));
            if (h[3] != in_pass) return fail(DD_EHIP, "exact count: pass over bins %u..%u appended %llu k-mers, %llu expected", b0, b1, h[3], in_pass);
            total += h[2];
            ++npass;
        }
        b0 = b1;
    }
    c->st_blocks = npass;  // (visible through dd_last_sketch_stats: how many passes the last exact count took)
    *distinct = total;
    return DD_OK;
}

int dd_exact_count(dd_ctx* c, const char* const* paths, int n, int k, uint64_t* distinct) {
    if (check_ctx(c)) return DD_EINVAL;
    if (n < 0 || !distinct || (n && !paths)) return fail(DD_EINVAL, "null argument");
    DeviceGuard guard(c->device);
    std::vector<size_t> sizes(n), offs(n);
    std::vector<FileBuf> bufs(n);
    size_t tot = 0;
    for (int i = 0; i < n; ++i) {
        std::string err;
        if (!paths[i] || !read_fasta_file(paths[i], bufs[i], err, usable_cpus())) return fail(DD_EIO, "%s", err.c_str());
        sizes[i] = bufs[i].size();
        offs[i] = tot;
        tot += align_up(sizes[i] + 16, 256);
    }
    int rc;
    if ((rc = c->fasta.reserve(tot + 16))) return rc;
    std::vector<const uint8_t*> ptrs(n);
    for (int i = 0; i < n; ++i) {
        ptrs[i] = static_cast<const uint8_t*>(c->fasta.p) + offs[i];
        if (sizes[i])
            DD_HIP(hipMemcpyAsync(const_cast<uint8_t*>(ptrs[i]), bufs[i].data(), sizes[i], hipMemcpyHostToDevice, c->stream));
    }
    DD_HIP(hipStreamSynchronize(c->stream));  // host buffers are pageable; release them before the sort
    std::vector<FileBuf>().swap(bufs);
    return dd_exact_count_device(c, ptrs.data(), sizes.data(), n, k, distinct);
}

// ------------------------------------------------------------------------------- union
int dd_union_device(dd_ctx* c, const uint8_t* const* in_dev, int n, size_t len, uint8_t* out_dev) {
    if (check_ctx(c)) return DD_EINVAL;
    if (n < 1 || !in_dev || !out_dev) return fail(DD_EINVAL, "bad argument");
    if (len % 16) return fail(DD_EINVAL, "len must be a multiple of 16");
    DeviceGuard guard(c->device);
    int rc;
    if ((rc = c->ptrs.reserve(sizeof(void*) * n))) return rc;
    DD_HIP(hipEventSynchronize(c->stage_free));
    if ((rc = c->stage.reserve(sizeof(void*) * n))) return rc;
    if ((rc = upload(c, c->stage, c->ptrs.p, in_dev, sizeof(void*) * n, 0))) return rc;
    DD_HIP(hipEventRecord(c->stage_free, c->stream));
    {
        Span sp(c, DD_KERNEL_UNION);
        dd::launch_union(static_cast<const uint8_t* const*>(c->ptrs.p), n, len, out_dev, c->stream);
    }
    DD_HIP(hipGetLastError());
    return DD_OK;
}

int dd_union(dd_ctx* c, const uint8_t* const* in, int n, size_t len, uint8_t* out) {
    if (check_ctx(c)) return DD_EINVAL;
    if (n < 1 || !in || !out) return fail(DD_EINVAL, "bad argument");
    if (len % 16) return fail(DD_EINVAL, "len must be a multiple of 16");
    DeviceGuard guard(c->device);
    int rc;
    if ((rc = c->regs.reserve((size_t)(n + 1) * len))) return rc;
    uint8_t* base = static_cast<uint8_t*>(c->regs.p);
    std::vector<const uint8_t*> ptrs(n);
    for (int i = 0; i < n; ++i) {
        DD_HIP(hipMemcpyAsync(base + (size_t)i * len, in[i], len, hipMemcpyHostToDevice, c->stream));
        ptrs[i] = base + (size_t)i * len;
    }
    if ((rc = dd_union_device(c, ptrs.data(), n, len, base + (size_t)n * len))) return rc;
    DD_HIP(hipMemcpyAsync(out, base + (size_t)n * len, len, hipMemcpyDeviceToHost, c->stream));
    DD_HIP(hipStreamSynchronize(c->stream));
    return DD_OK;
}

// -------------------------------------------------------------------------------- card
double dd_ertl_mle(const uint32_t hist[64], int log2m) {
    return dd::ertl_mle(hist, log2m, dd::mle_relerr(log2m));
}

int dd_hist_batch_device(dd_ctx* c, const uint8_t* regs_dev, int njobs, uint32_t* hist) {
    if (check_ctx(c)) return DD_EINVAL;
    if (njobs < 0 || (njobs && (!regs_dev || !hist))) return fail(DD_EINVAL, "bad argument");
    if (!njobs) return DD_OK;
    DeviceGuard guard(c->device);
    int rc;
    if ((rc = c->hist.reserve((size_t)njobs * 64 * sizeof(uint32_t)))) return rc;
    {
        Span sp(c, DD_KERNEL_UNION);
        dd::launch_hist(regs_dev, njobs, c->p, static_cast<uint32_t*>(c->hist.p), c->stream);
    }
    DD_HIP(hipGetLastError());
    DD_HIP(hipMemcpyAsync(hist, c->hist.p, (size_t)njobs * 64 * sizeof(uint32_t), hipMemcpyDeviceToHost,
                          c->stream));
    DD_HIP(hipStreamSynchronize(c->stream));
    return DD_OK;
}

int dd_card_batch_device(dd_ctx* c, const uint8_t* regs_dev, int njobs, double* est) {
    if (check_ctx(c)) return DD_EINVAL;
    if (njobs < 0 || (njobs && (!regs_dev || !est))) return fail(DD_EINVAL, "bad argument");
    if (!njobs) return DD_OK;
    DeviceGuard guard(c->device);
    int rc;
    if ((rc = c->hist.reserve((size_t)njobs * 64 * sizeof(uint32_t)))) return rc;
    {
        Span sp(c, DD_KERNEL_UNION);
        dd::launch_hist(regs_dev, njobs, c->p, static_cast<uint32_t*>(c->hist.p), c->stream);
    }
    DD_HIP(hipGetLastError());
    return estimates_from_hist(c, static_cast<const uint32_t*>(c->hist.p), (size_t)njobs, est);
}

int dd_card_batch(dd_ctx* c, const uint8_t* regs, int njobs, double* est) {
    if (check_ctx(c)) return DD_EINVAL;
    if (njobs < 0 || (njobs && (!regs || !est))) return fail(DD_EINVAL, "bad argument");
    if (!njobs) return DD_OK;
    DeviceGuard guard(c->device);
    const size_t bytes = (size_t)njobs << c->p;
    int rc;
    if ((rc = c->regs.reserve(bytes))) return rc;
    DD_HIP(hipMemcpyAsync(c->regs.p, regs, bytes, hipMemcpyHostToDevice, c->stream));
    return dd_card_batch_device(c, static_cast<const uint8_t*>(c->regs.p), njobs, est);
}

int dd_card(dd_ctx* c, const uint8_t* regs, double* est) { return dd_card_batch(c, regs, 1, est); }

// ------------------------------------------------------------------------- progressive
int dd_progressive_device(dd_ctx* c, const uint8_t* leaf_dev, int n, int K, const int32_t* orderings,
                          int norder, double* card) {
    if (check_ctx(c)) return DD_EINVAL;
    if (n < 1 || K < 1 || norder < 1 || !leaf_dev || !orderings || !card)
        return fail(DD_EINVAL, "bad argument");
    for (size_t i = 0; i < (size_t)norder * n; ++i)
        if (orderings[i] < 0 || orderings[i] >= n) return fail(DD_EINVAL, "ordering entry %d outside 0..%d", orderings[i], n - 1);
    DeviceGuard guard(c->device);
    const size_t njobs = (size_t)norder * n * K;
    int rc;
    if ((rc = c->hist.reserve(njobs * 64 * sizeof(uint32_t)))) return rc;
    if ((rc = c->ord.reserve(sizeof(int32_t) * norder * n))) return rc;
    DD_HIP(hipEventSynchronize(c->stage_free));
    if ((rc = c->stage.reserve(sizeof(int32_t) * norder * n))) return rc;
    if ((rc = upload(c, c->stage, c->ord.p, orderings, sizeof(int32_t) * norder * n, 0))) return rc;
    DD_HIP(hipEventRecord(c->stage_free, c->stream));
    {
        Span sp(c, DD_KERNEL_UNION);
        // bit-plane AND-scan (dd_pscan.hip) where it applies; DD_PROGRESSIVE_STREAM=1 keeps the streaming kernel of
        // dd_union.hip (one LDS atomic per register per prefix) for A/B runs and for the equality test
        bool done = false;
        if (dd::pscan_usable(n, norder, c->p) && !getenv("DD_PROGRESSIVE_STREAM")) {
            if ((rc = c->gram.reserve(dd::pscan_scratch_bytes(n, K, c->p, norder)))) return rc;
            dd::launch_register_range(leaf_dev, n, K, c->p, static_cast<uint32_t*>(c->gram.p), c->stream);
            std::vector<uint32_t> rng((size_t)K * 2);   // (which thresholds exist decides the tile size: 296 bytes back to the host)
            DD_HIP(hipMemcpyAsync(rng.data(), c->gram.p, rng.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            DD_HIP(hipStreamSynchronize(c->stream));
            done = dd::launch_progressive_pscan(leaf_dev, n, K, c->p, static_cast<const int32_t*>(c->ord.p), norder, rng.data(), c->gram.p,
                                                static_cast<uint32_t*>(c->hist.p), c->stream);
        }
        c->k2_path = done ? DD_K2_PROGRESSIVE_PSCAN : DD_K2_PROGRESSIVE_STREAM;
        if (!done)
            dd::launch_progressive(leaf_dev, n, K, c->p, static_cast<const int32_t*>(c->ord.p), norder,
                                   static_cast<uint32_t*>(c->hist.p), c->stream);
    }
    DD_HIP(hipGetLastError());
    return estimates_from_hist(c, static_cast<const uint32_t*>(c->hist.p), njobs, card);
}

int dd_progressive(dd_ctx* c, const uint8_t* leaf, int n, int K, const int32_t* orderings, int norder,
                   double* card) {
    if (check_ctx(c)) return DD_EINVAL;
    if (n < 1 || K < 1 || !leaf) return fail(DD_EINVAL, "bad argument");
    DeviceGuard guard(c->device);
    const size_t bytes = ((size_t)n * K) << c->p;
    int rc;
    if ((rc = c->regs.reserve(bytes))) return rc;
    DD_HIP(hipMemcpyAsync(c->regs.p, leaf, bytes, hipMemcpyHostToDevice, c->stream));
    return dd_progressive_device(c, static_cast<const uint8_t*>(c->regs.p), n, K, orderings, norder, card);
}

// ---------------------------------------------------------------------------- pairwise
int dd_pairwise_device(dd_ctx* c, const uint8_t* leaf_dev, int n, int K, double* card) {
    if (check_ctx(c)) return DD_EINVAL;
    if (n < 1 || K < 1 || !leaf_dev || !card) return fail(DD_EINVAL, "bad argument");
    DeviceGuard guard(c->device);
    const size_t njobs = (size_t)n * n * K;
    int rc;
    if ((rc = c->hist.reserve(njobs * 64 * sizeof(uint32_t)))) return rc;
    // all pairs as int8 Gram matrices on the matrix cores (dd_gram.hip); DD_PAIRWISE_STREAM=1 keeps the streaming
    // kernel of dd_union.hip (one LDS atomic per register per pair) for A/B runs and for the equality test
    const bool gram = dd::gram_usable(n, c->p) && !getenv("DD_PAIRWISE_STREAM");
    if (gram && (rc = c->gram.reserve(dd::gram_scratch_bytes(n, K, c->p, nullptr)))) return rc;
    c->k2_path = gram ? DD_K2_PAIRWISE_GRAM : DD_K2_PAIRWISE_STREAM;
    {
        Span sp(c, DD_KERNEL_UNION);
        if (gram) {
            DD_HIP(hipMemsetAsync(c->hist.p, 0, njobs * 64 * sizeof(uint32_t), c->stream));
            dd::launch_pairwise_gram(leaf_dev, n, K, c->p, static_cast<uint32_t*>(c->hist.p), c->gram.p, c->stream);
        } else {
            dd::launch_pairwise(leaf_dev, n, K, c->p, static_cast<uint32_t*>(c->hist.p), c->stream);
        }
    }
    DD_HIP(hipGetLastError());
    // lower triangle histograms are all-zero: give them the mirrored estimate afterwards
    std::vector<double> tmp(njobs);
    if ((rc = c->est.reserve(njobs * sizeof(double)))) return rc;
    // only the upper triangle (i <= j) holds real histograms; estimate everything on the device
    // would waste work on empty ones, so fill empties with m in bin 0 -> estimate 0 cheaply
    dd::launch_mle(static_cast<const uint32_t*>(c->hist.p), njobs, c->p, static_cast<double*>(c->est.p),
                   c->stream);
    DD_HIP(hipGetLastError());
    DD_HIP(hipMemcpyAsync(tmp.data(), c->est.p, njobs * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    DD_HIP(hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const int a = i <= j ? i : j, b = i <= j ? j : i;
            memcpy(card + ((size_t)i * n + j) * K, tmp.data() + ((size_t)a * n + b) * K, sizeof(double) * K);
        }
    return DD_OK;
}

int dd_pairwise(dd_ctx* c, const uint8_t* leaf, int n, int K, double* card) {
    if (check_ctx(c)) return DD_EINVAL;
    if (n < 1 || K < 1 || !leaf) return fail(DD_EINVAL, "bad argument");
    DeviceGuard guard(c->device);
    const size_t bytes = ((size_t)n * K) << c->p;
    int rc;
    if ((rc = c->regs.reserve(bytes))) return rc;
    DD_HIP(hipMemcpyAsync(c->regs.p, leaf, bytes, hipMemcpyHostToDevice, c->stream));
    return dd_pairwise_device(c, static_cast<const uint8_t*>(c->regs.p), n, K, card);
}

// ------------------------------------------------------------------------- measurement
int dd_timing_enable(dd_ctx* c, int on) {
    if (check_ctx(c)) return DD_EINVAL;
    c->timing = on != 0;
    return DD_OK;
}

int dd_timing_reset(dd_ctx* c) {
    if (check_ctx(c)) return DD_EINVAL;
    DeviceGuard guard(c->device);
    DD_HIP(hipStreamSynchronize(c->stream));
    for (auto& v : c->spans) {
        for (auto& s : v) {
            c->pool.push_back(s.a);
            c->pool.push_back(s.b);
        }
        v.clear();
    }
    return DD_OK;
}

int dd_timing_read(dd_ctx* c, int which, double* total_ms, int* launches) {
    if (check_ctx(c)) return DD_EINVAL;
    if (which < 0 || which >= DD_KERNEL_COUNT) return fail(DD_EINVAL, "bad kernel id %d", which);
    DeviceGuard guard(c->device);
    DD_HIP(hipStreamSynchronize(c->stream));
    double tot = 0;
    for (auto& s : c->spans[which]) {
        float ms = 0;
        DD_HIP(hipEventElapsedTime(&ms, s.a, s.b));
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = (int)c->spans[which].size();
    return DD_OK;
}

int dd_last_sketch_stats(dd_ctx* c, uint64_t* tokens, uint64_t* updates, int* sweep_blocks) {
    if (check_ctx(c)) return DD_EINVAL;
    if (tokens) *tokens = c->st_tokens;
    if (updates) *updates = c->st_updates;
    if (sweep_blocks) *sweep_blocks = c->st_blocks;
    return DD_OK;
}

int dd_last_k2_path(dd_ctx* c) {
    if (check_ctx(c)) return DD_EINVAL;
    return c->k2_path;
}

// --------------------------------------------------------------------------- synthetic
long dd_plan_sweep(int log2m, const size_t* nbytes, int ngenomes, int kmin, int kmax, dd_plan_job* out,
                   long cap) {
    if (log2m < 4 || log2m > 20) return fail(DD_EINVAL, "log2m %d outside 4..20", log2m);
    if (ngenomes < 0 || (ngenomes && !nbytes) || cap < 0 || (cap && !out)) return fail(DD_EINVAL, "null argument");
    if (kmin < 1 || kmax > 64 || kmin > kmax) return fail(DD_EINVAL, "k range %d..%d outside 1..64", kmin, kmax);
    const std::vector<dd::SweepClass> classes =
        dd::plan_sweep(log2m, 1, nbytes, ngenomes, kmin, kmax, dd::PlanKnobs::from_env());
    long n = 0;
    for (const dd::SweepClass& sc : classes) {
        for (const dd::SweepJob& j : sc.jobs) {
            if (n < cap)
                out[n] = dd_plan_job{sc.kclass, sc.plan.mode, sc.plan.lds_bytes, j.genome, j.kfirst, j.nk,
                                     j.tile_begin, j.tile_end, j.slice};
            ++n;
        }
    }
    return n;
}

size_t dd_synth_size(uint64_t nbases, int nrec) {
    if (nrec < 1) return 0;
    return dd::synth_size(nbases, nrec);
}

int dd_synth_fasta_device(dd_ctx* c, uint64_t seed, int genome_index, uint64_t nbases, int nrec,
                          uint8_t* out_dev) {
    if (check_ctx(c)) return DD_EINVAL;
    if (nrec < 1 || nrec > 65535 || genome_index < 0 || genome_index > 65535 || !out_dev)
        return fail(DD_EINVAL, "bad argument");
    DeviceGuard guard(c->device);
    dd::launch_synth(seed, genome_index, nbases, nrec, out_dev, c->stream);
    DD_HIP(hipGetLastError());
    return DD_OK;
}

size_t dd_synth_realistic_size(uint64_t seed, uint64_t nbases) {
    if (!nbases) return 0;
    const std::vector<uint64_t> tab = dd::synth_realistic_table(seed, nbases);
    return (size_t)tab[tab.size() - 2];
}

int dd_synth_realistic_device(dd_ctx* c, uint64_t seed, int genome_index, uint64_t nbases, uint8_t* out_dev) {
    if (check_ctx(c)) return DD_EINVAL;
    if (genome_index < 0 || genome_index > 65535 || !out_dev) return fail(DD_EINVAL, "bad argument");
    if (!nbases) return DD_OK;
    DeviceGuard guard(c->device);
    const std::vector<uint64_t> tab = dd::synth_realistic_table(seed, nbases);
    int rc;
    if ((rc = c->synth.reserve(tab.size() * sizeof(uint64_t)))) return rc;
    DD_HIP(hipMemcpyAsync(c->synth.p, tab.data(), tab.size() * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    DD_HIP(hipStreamSynchronize(c->stream));   // (`tab` is pageable host memory about to go out of scope)
    dd::launch_synth_realistic(seed, genome_index, static_cast<const uint64_t*>(c->synth.p), (uint32_t)(tab.size() / 2 - 1),
                               tab[tab.size() - 2], out_dev, c->stream);
    DD_HIP(hipGetLastError());
    return DD_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------- multi-GPU: RCCL over xGMI behind the C ABI
// (genome x k) jobs shard over the GPUs of a node with no data-path exchange; what crosses xGMI is the root --
// every rank's [K][m] slab of byte-max-merged registers through ncclAllReduce(ncclUint8, ncclMax) -- and, for the schedules that
// need every leaf (progressive, kij), one ncclAllGather of the ranks' leaf slabs.  The reference's only parallelism is
// `parallel -j 95%` over k on one host (/root/reference/lib/huffman_dandd.py:217).  librccl is opened at the first dd_comm_*
// call (the copy already mapped into the process if there is one -- PyTorch-ROCm brings its own), never linked: a single-GPU
// user of this library needs no RCCL.
#include <rccl/rccl.h>
namespace {
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};
RcclApi* rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // DD_RCCL_LIB names THE copy to use (nothing else is tried when it is set)
        const char* named = getenv("DD_RCCL_LIB");
        const char* names[] = {named, named ? nullptr : "librccl.so.1", named ? nullptr : "librccl.so", named ? nullptr : "/opt/rocm/lib/librccl.so.1"};
        for (int pass = 0; pass < 2 && !api.lib; ++pass)      // pass 0: a copy that is already mapped (RTLD_NOLOAD)
            for (const char* n : names)
                if (n && !api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
        if (!api.lib) {
            const char* e = dlerror();                        // one call: dlerror() clears its state when read
            api.why = std::string("librccl.so not found (") + (e ? e : "?") + "); set DD_RCCL_LIB";
            return;
        }
        auto sym = [&](const char* n) {
            void* f = dlsym(api.lib, n);
            if (!f && api.why.empty()) api.why = std::string("librccl: no symbol ") + n;
            return f;
        };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return &api;
}
int rccl_ready(RcclApi*& api) {
    api = rccl();
    if (!api->why.empty()) return fail(DD_ENODEV, "RCCL: %s", api->why.c_str());
    return DD_OK;
}
#define DD_RCCL(api, expr)                                                                                     \
    do {                                                                                                       \
        const ncclResult_t r_ = (expr);                                                                        \
        if (r_ != ncclSuccess) return fail(DD_EHIP, "RCCL: %s failed: %s", #expr, (api)->GetErrorString(r_)); \
    } while (0)
}  // namespace

static_assert(DD_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "dandd_hip.h: DD_COMM_ID_BYTES is ncclUniqueId's size");

int dd_comm_unique_id(uint8_t* id) {
    if (!id) return fail(DD_EINVAL, "null argument");
    RcclApi* api;
    int rc;
    if ((rc = rccl_ready(api))) return rc;
    ncclUniqueId u;
    DD_RCCL(api, api->GetUniqueId(&u));
    memcpy(id, u.internal, DD_COMM_ID_BYTES);
    return DD_OK;
}

int dd_comm_init(dd_ctx* c, int rank, int world, const uint8_t* id) {
    if (check_ctx(c)) return DD_EINVAL;
    if (!id || world < 1 || rank < 0 || rank >= world) return fail(DD_EINVAL, "rank %d of %d", rank, world);
    if (c->comm) return fail(DD_EINVAL, "this context already belongs to a communicator (dd_comm_destroy first)");
    RcclApi* api;
    int rc;
    if ((rc = rccl_ready(api))) return rc;
    DeviceGuard guard(c->device);   // ncclCommInitRank binds the communicator to the CURRENT device: the context's
    ncclUniqueId u;
    memcpy(u.internal, id, DD_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    DD_RCCL(api, api->CommInitRank(&comm, world, u, rank));
    c->comm = comm;
    c->comm_rank = rank;
    c->comm_world = world;
    c->comm_calls[0] = c->comm_calls[1] = 0;
    return DD_OK;
}

int dd_comm_destroy(dd_ctx* c) {
    if (!c) return DD_EINVAL;
    if (!c->comm) return DD_OK;
    RcclApi* api = rccl();
    DeviceGuard guard(c->device);
    (void)hipStreamSynchronize(c->stream);
    const ncclResult_t r = api->CommDestroy ? api->CommDestroy(static_cast<ncclComm_t>(c->comm)) : ncclSuccess;
    c->comm = nullptr;
    c->comm_rank = 0;
    c->comm_world = 1;
    return r == ncclSuccess ? DD_OK : fail(DD_EHIP, "RCCL: ncclCommDestroy failed");
}

int dd_comm_info(dd_ctx* c, int* rank, int* world, unsigned long long* allreduces, unsigned long long* allgathers) {
    if (check_ctx(c)) return DD_EINVAL;
    if (rank) *rank = c->comm_rank;
    if (world) *world = c->comm ? c->comm_world : 0;   // 0: no communicator
    if (allreduces) *allreduces = c->comm_calls[0];
    if (allgathers) *allgathers = c->comm_calls[1];
    return DD_OK;
}

int dd_allreduce_max_u8(dd_ctx* c, uint8_t* regs_dev, size_t n) {
    if (check_ctx(c)) return DD_EINVAL;
    if (!c->comm) return fail(DD_EINVAL, "no communicator on this context (dd_comm_init)");
    if (n && !regs_dev) return fail(DD_EINVAL, "null argument");
    if (!n) return DD_OK;
    RcclApi* api = rccl();
    DeviceGuard guard(c->device);
    DD_RCCL(api, api->AllReduce(regs_dev, regs_dev, n, ncclUint8, ncclMax, static_cast<ncclComm_t>(c->comm), c->stream));
    ++c->comm_calls[0];
    return DD_OK;
}

int dd_allgather_u8(dd_ctx* c, const uint8_t* send_dev, size_t n, uint8_t* recv_dev) {
    if (check_ctx(c)) return DD_EINVAL;
    if (!c->comm) return fail(DD_EINVAL, "no communicator on this context (dd_comm_init)");
    if (n && (!send_dev || !recv_dev)) return fail(DD_EINVAL, "null argument");
    if (!n) return DD_OK;
    RcclApi* api = rccl();
    DeviceGuard guard(c->device);
    DD_RCCL(api, api->AllGather(send_dev, recv_dev, n, ncclUint8, static_cast<ncclComm_t>(c->comm), c->stream));
    ++c->comm_calls[1];
    return DD_OK;
}
