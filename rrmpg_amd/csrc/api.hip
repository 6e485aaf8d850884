// api.hip -- error plumbing, device queries and the host-pointer entry points
// (rr_<model>_simulate) of the C-ABI declared in include/rrhip.h.
//
// The host-pointer family is the drop-in for the reference's seam, where
// Model.simulate() hands numpy arrays to run_* and gets numpy arrays back
// (reference: rrmpg/models/hbvedu.py:190-214).  It owns no state between
// calls: it uploads the (small, shared) forcing and the parameter block,
// sweeps the parameter-set axis in column blocks sized to the free HBM, runs
// the *_simulate_dev entry point on each block and copies each [T][nc] device
// slab into the caller's [T][N] array with a pitched copy.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

static thread_local char g_err[512] = "";

void rr_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *rr_last_error(void) { return g_err; }
extern "C" int rr_version(void) { return 100; }

// ---- options (rrhip.h RR_OPT_*) -----------------------------------------------
// Three layers, consulted in this order by rr_option():
//   1. the rr_call_options of the host-pointer call in progress on this thread
//      (rr_<model>_simulate_opt; inherited by the call's shard threads),
//   2. the calling thread's standing options (rr_thread_options: the way a
//      caller of the *_simulate_dev family pins a kernel variant),
//   3. the process-wide measurement / test values (rr_debug_set_option).
// Layers 1 and 2 are thread local: concurrent callers never see each other's
// choices.
static std::atomic<int64_t> g_options[RR_OPT_COUNT_] = {
    {0}, {-1} /* HBV variant: heuristic */, {0}, {0}, {0}, {0}, {0}, {0},
    {-1} /* HBV tiles: by sweep size */, {-1} /* records: by sweep size */};
static thread_local const rr_call_options *tl_call = nullptr;
static thread_local rr_call_options tl_standing;
static thread_local bool tl_standing_on = false;

int64_t rr_option(int option)
{
    if (tl_call && tl_call->value[option] != RR_OPT_UNSET)
        return tl_call->value[option];
    if (tl_standing_on && tl_standing.value[option] != RR_OPT_UNSET)
        return tl_standing.value[option];
    return g_options[option].load(std::memory_order_relaxed);
}

static bool option_value_ok(int option, int64_t value)
{
    switch (option) {
    case RR_OPT_HBV_VARIANT: return value == -1 || value == 0 || value == 3;
    case RR_OPT_GR4J_FORCE_LDS: return value == 0 || value == 1;
    case RR_OPT_FUSED_VARIANT: return value >= 0 && value <= 3;
    case RR_OPT_GR4J_VARIANT: return value == 0 || value == 1;
    case RR_OPT_MAX_BLOCK_COLS: return value >= 0;
    case RR_OPT_GATHER_THREADS: return value >= 0 && value <= 256;
    case RR_OPT_HOST_SHARDS: return value >= -1 && value <= 1024;
    case RR_OPT_TIME_TILES: return value >= -1 && value <= 64 && value != 1;
    case RR_OPT_WARM_RECORDS: return value >= -1 && value <= 1;
    default: return false;
    }
}

static int check_call_options(const char *who, const rr_call_options *opt)
{
    if (!opt) return RR_OK;
    if (opt->struct_bytes != sizeof(rr_call_options)) {
        rr_set_error("%s: rr_call_options.struct_bytes is %zu, this library's "
                     "struct has %zu (rr_call_options_init sets it)", who,
                     (size_t)opt->struct_bytes, sizeof(rr_call_options));
        return RR_E_PARAM;
    }
    for (int o = 1; o < RR_OPT_COUNT_; ++o)
        if (opt->value[o] != RR_OPT_UNSET && !option_value_ok(o, opt->value[o])) {
            rr_set_error("%s: option %d does not take %lld", who, o,
                         (long long)opt->value[o]);
            return RR_E_PARAM;
        }
    return RR_OK;
}

// The options of one host-pointer call, for its duration on this thread (and,
// through host_fan_out, on the call's shard threads).
struct CallOptionsScope {
    const rr_call_options *prev;
    explicit CallOptionsScope(const rr_call_options *opt) : prev(tl_call)
    {
        if (opt) tl_call = opt;
    }
    ~CallOptionsScope() { tl_call = prev; }
};

extern "C" void rr_call_options_init(rr_call_options *opt)
{
    if (!opt) return;
    opt->struct_bytes = sizeof(rr_call_options);
    for (int64_t &v : opt->value) v = RR_OPT_UNSET;
}

extern "C" int rr_call_options_set(rr_call_options *opt, int option,
                                   int64_t value)
{
    if (!opt || opt->struct_bytes != sizeof(rr_call_options)) {
        rr_set_error("rr_call_options_set: options not initialised "
                     "(rr_call_options_init)");
        return RR_E_PARAM;
    }
    if (option < 1 || option >= RR_OPT_COUNT_ ||
        (value != RR_OPT_UNSET && !option_value_ok(option, value))) {
        rr_set_error("rr_call_options_set: option %d does not take %lld",
                     option, (long long)value);
        return RR_E_PARAM;
    }
    opt->value[option] = value;
    return RR_OK;
}

extern "C" int rr_thread_options(const rr_call_options *opt)
{
    if (!opt) {
        tl_standing_on = false;
        return RR_OK;
    }
    const int rc = check_call_options("rr_thread_options", opt);
    if (rc != RR_OK) return rc;
    tl_standing = *opt;
    tl_standing_on = true;
    return RR_OK;
}

extern "C" int rr_debug_set_option(int option, int64_t value)
{
    if (!option_value_ok(option, value)) {
        rr_set_error("rr_debug_set_option: option %d does not take %lld",
                     option, (long long)value);
        return RR_E_PARAM;
    }
    g_options[option].store(value, std::memory_order_relaxed);
    return RR_OK;
}

extern "C" int64_t rr_debug_get_option(int option)
{
    if (option < 1 || option >= RR_OPT_COUNT_) return INT64_MIN;
    return g_options[option].load(std::memory_order_relaxed);
}

extern "C" int rr_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int rr_simd_count()
{
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        (void)hipGetLastError();
        return 1024;
    }
    int v = cached[dev].load(std::memory_order_relaxed);
    if (v > 0) return v;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount,
                              dev) != hipSuccess || cus < 1) {
        (void)hipGetLastError();
        cus = 256;
    }
    v = 4 * cus;
    cached[dev].store(v, std::memory_order_relaxed);
    return v;
}

int rr_check_common(const char *who, int64_t T, int64_t N, int64_t ld,
                    const void *params, const void *qobs, const void *sse)
{
    g_err[0] = 0;
    if (T < 0 || N < 0) {
        rr_set_error("%s: negative size (T=%lld, N=%lld)", who, (long long)T,
                     (long long)N);
        return RR_E_SIZE;
    }
    if (T > 2000000000) {
        // (the kernels count days in 32 bits)
        rr_set_error("%s: T = %lld exceeds 2e9 timesteps", who, (long long)T);
        return RR_E_SIZE;
    }
    if (ld < N) {
        rr_set_error("%s: ld=%lld < N=%lld", who, (long long)ld, (long long)N);
        return RR_E_SIZE;
    }
    if (N > 0 && !params) {
        rr_set_error("%s: params is NULL", who);
        return RR_E_NULL;
    }
    if ((qobs == nullptr) != (sse == nullptr)) {
        rr_set_error("%s: qobs and sse must be given together", who);
        return RR_E_NULL;
    }
    return RR_OK;
}

int rr_check_outputs(const char *who, const void *qsim, bool any_storage)
{
    if (any_storage && !qsim) {
        rr_set_error("%s: storage outputs come with qsim (as the reference's "
                     "return_storage returns them), not alone", who);
        return RR_E_NULL;
    }
    return RR_OK;
}

// --------------------------------------------------------------------------
// host-pointer family
// --------------------------------------------------------------------------
namespace {

// ---- per-device context kept between calls ---------------------------------
// Model.fit() calls the host entry points thousands of times with the SAME
// forcing arrays and one parameter set (reference: rrmpg/models/hbvedu.py:305,
// 310-346).  Allocating and freeing eight device buffers and re-uploading the
// forcing on every call cost more than the sweep itself, so each device keeps
//   * two streams (compute / copy) and two events,
//   * its small device buffers (inputs, parameters, workspace, score vector,
//     output slabs up to CACHE_MAX_BUF bytes each) -- grow only,
//   * for every input the length and a 64-bit hash of the bytes last
//     uploaded: an input whose bytes have not changed is not uploaded again,
//   * a small pinned bounce buffer for parameter blocks and score vectors.
// Multi-gigabyte output slabs are released when the call returns.  The
// context is guarded by a mutex (the reference's seam is single threaded;
// concurrent callers of one device are serialised).  rr_release_cached_memory
// gives everything back.
constexpr size_t CACHE_MAX_BUF = (size_t)64 << 20;
constexpr size_t PINNED_BYTES = (size_t)4 << 20;
constexpr int MAX_INPUTS = 10, MAX_OUTS = 8, MAX_DEVICES = 64;

struct Slot {
    void *p = nullptr;
    size_t cap = 0;
    size_t bytes = 0;      // content (inputs only)
    uint64_t hash = 0, hash2 = 0;
    bool valid = false;
    template <class T> T *as() const { return (T *)p; }
};

// Pinned staging ring of the large gathers (see Gatherer below)
constexpr int RING_SLOTS = 16;
constexpr size_t RING_SLOT_BYTES = (size_t)16 << 20;

struct HostCtx {
    std::mutex mu;
    bool ready = false;
    hipStream_t compute = nullptr, copy = nullptr, copy2 = nullptr;
    hipEvent_t done[2] = {nullptr, nullptr}, copied[2] = {nullptr, nullptr},
               copied2[2] = {nullptr, nullptr};
    Slot in[MAX_INPUTS], par, ws[2], sse[2], slab[2][MAX_OUTS];
    void *pinned = nullptr;
    void *ring = nullptr;                    // RING_SLOTS * RING_SLOT_BYTES
    hipEvent_t ring_ev[RING_SLOTS] = {};
};
HostCtx g_ctx[MAX_DEVICES];

int slot_reserve(Slot &s, size_t bytes)
{
    if (bytes == 0) bytes = 8;
    if (s.cap >= bytes) return RR_OK;
    if (s.p) (void)hipFree(s.p);
    s.p = nullptr; s.cap = 0; s.valid = false;
    hipError_t e = hipMalloc(&s.p, bytes);
    if (e != hipSuccess) {
        s.p = nullptr;
        (void)hipGetLastError();
        rr_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return RR_E_HIP;
    }
    s.cap = bytes;
    return RR_OK;
}

void slot_free(Slot &s)
{
    if (s.p) (void)hipFree(s.p);
    s = Slot();
}

// 128 bits of content hash (two independent multiplicative streams over the
// same words): an input is taken as unchanged only if its length and both
// halves agree with what was uploaded last.
struct Hash128 {
    uint64_t a, b;
    bool operator==(const Hash128 &o) const { return a == o.a && b == o.b; }
};
Hash128 hash_bytes(const void *src, size_t bytes)
{
    const unsigned char *p = (const unsigned char *)src;
    uint64_t h = 0x9E3779B97F4A7C15ull ^ bytes;
    uint64_t g = 0xC2B2AE3D27D4EB4Full + bytes;
    size_t k = 0;
    for (; k + 8 <= bytes; k += 8) {
        uint64_t w;
        memcpy(&w, p + k, 8);
        h = (h ^ w) * 0xD6E8FEB86659FD93ull;
        h ^= h >> 32;
        g = (g + w) * 0xFF51AFD7ED558CCDull;
        g ^= g >> 29;
    }
    for (; k < bytes; ++k) {
        h = (h ^ p[k]) * 0x100000001B3ull;
        g = (g + p[k]) * 0xC4CEB9FE1A85EC53ull;
    }
    return {h, g};
}

// One host-pointer call: holds the device's context for its duration.
struct HostCall {
    HostCtx *c = nullptr;
    std::unique_lock<std::mutex> lock;
    int next_input = 0;

    int open(const char *who)
    {
        if (rr_device_count() < 1) {
            rr_set_error("no HIP device visible: librrhip has no CPU path");
            return RR_E_NODEVICE;
        }
        int dev = 0;
        RR_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= MAX_DEVICES) {
            rr_set_error("%s: device index %d out of range", who, dev);
            return RR_E_NODEVICE;
        }
        c = &g_ctx[dev];
        lock = std::unique_lock<std::mutex>(c->mu);
        if (!c->ready) {
            RR_HIP(hipStreamCreateWithFlags(&c->compute, hipStreamNonBlocking));
            RR_HIP(hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking));
            RR_HIP(hipStreamCreateWithFlags(&c->copy2, hipStreamNonBlocking));
            for (int k = 0; k < 2; ++k) {
                RR_HIP(hipEventCreateWithFlags(&c->done[k],
                                               hipEventDisableTiming));
                RR_HIP(hipEventCreateWithFlags(&c->copied[k],
                                               hipEventDisableTiming));
                RR_HIP(hipEventCreateWithFlags(&c->copied2[k],
                                               hipEventDisableTiming));
            }
            RR_HIP(hipHostMalloc(&c->pinned, PINNED_BYTES, hipHostMallocDefault));
            c->ready = true;
        }
        return RR_OK;
    }

    // device copy of a read-only input; uploaded only if its bytes changed
    int input(const void *src, size_t bytes, const void **dev)
    {
        Slot &s = c->in[next_input++];
        const Hash128 h = hash_bytes(src, bytes);
        if (!(s.valid && s.bytes == bytes && s.hash == h.a &&
              s.hash2 == h.b)) {
            int rc = slot_reserve(s, bytes);
            if (rc != RR_OK) return rc;
            s.valid = false;
            if (bytes)
                RR_HIP(hipMemcpyAsync(s.p, src, bytes, hipMemcpyHostToDevice,
                                      c->compute));
            s.bytes = bytes; s.hash = h.a; s.hash2 = h.b; s.valid = true;
        }
        *dev = s.p;
        return RR_OK;
    }

    // the parameter block (changes from call to call: always uploaded; small
    // blocks go through the pinned bounce buffer so the copy is one DMA)
    int params(const void *src, size_t bytes, const double **dev)
    {
        int rc = slot_reserve(c->par, bytes);
        if (rc != RR_OK) return rc;
        if (bytes <= PINNED_BYTES / 2) {
            memcpy(c->pinned, src, bytes);
            src = c->pinned;
        }
        RR_HIP(hipMemcpyAsync(c->par.p, src, bytes, hipMemcpyHostToDevice,
                              c->compute));
        *dev = c->par.as<double>();
        return RR_OK;
    }

    ~HostCall()
    {
        if (!c) return;
        // buffers above the cache limit do not outlive the call
        auto trim = [](Slot &s) { if (s.cap > CACHE_MAX_BUF) slot_free(s); };
        for (Slot &s : c->in) trim(s);
        trim(c->par);
        for (int k = 0; k < 2; ++k) {
            trim(c->ws[k]); trim(c->sse[k]);
            for (Slot &s : c->slab[k]) trim(s);
        }
    }
};

// One host output array [T][rows_per_t][N] mirrored by a device slab
// [T][rows_per_t][nc].
struct OutSpec {
    double *host;          // may be NULL (not requested)
    int64_t rows_per_t;    // 1, or L for the [T][L][N] storages
};

// Column-block width: two slabs of every requested output (double buffered:
// block k+1 is computed while block k crosses PCIe) within half of the free
// device memory; large results are cut into at least 8 blocks so that page
// faulting, DMA and compute overlap.
template <class WsBytes>
int64_t pick_block(int64_t T, int64_t N, const std::vector<OutSpec> &outs,
                   WsBytes ws_bytes)
{
    size_t per_col = 0;
    for (const OutSpec &o : outs)
        if (o.host) per_col += (size_t)T * (size_t)o.rows_per_t * 8;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)8 << 30;
    const size_t budget = free_b / 2;
    // the workspace grows with the block too (the GR4J family's
    // unit-hydrograph scratch: nc * (6 ceil(x4) + 2) * 8 B per slab): a block
    // has to fit with both of its workspaces
    auto shrink_to_budget = [&](int64_t nc) {
        while (nc > 64 && 2 * (ws_bytes(nc) + per_col * (size_t)nc) > budget)
            nc = (nc / 2 + 63) / 64 * 64;
        return nc;
    };
    if (per_col == 0) return shrink_to_budget(N);
    int64_t nc = (int64_t)(budget / (2 * per_col));
    if ((size_t)N * per_col > ((size_t)1 << 30)) {
        const int64_t eighth = rr_ceil_div(N, 8);
        if (eighth < nc) nc = eighth;
    }
    // whole 4-KiB pages per row segment where the budget allows, whole waves
    // otherwise; never below one column
    if (nc >= 512) nc = (nc / 512) * 512;
    else if (nc >= 64) nc = (nc / 64) * 64;
    else if (nc < 1) nc = 1;
    // test hook: force a small column block to exercise the pitched gather
    const int64_t cap = rr_option(RR_OPT_MAX_BLOCK_COLS);
    if (cap > 0 && cap < nc) nc = cap;
    if (nc > N) nc = N;
    return shrink_to_budget(nc);
}

// ---- large results: pinned staging ring + host copy threads -----------------
// A D2H copy straight into the caller's (pageable, freshly allocated) numpy
// array runs at ~40 GB/s at best and ~13 GB/s while its pages are still being
// faulted in; into pinned memory the same link gives ~57 GB/s
// (profiles/README.md).  So a large gather goes through a ring of pinned
// slots: the calling thread cuts every slab into chunks of whole rows (dense
// on the device: one plain async copy each) and keeps the DMA queue full,
// while a pool of host threads waits for each chunk's event and scatters its
// rows into the caller's [T][N] array -- first touch of the pages included,
// so nothing is written twice.
struct Gatherer {
    struct Job { int slot; double *host; int64_t row0, rows, nc; };
    HostCtx &c;
    int device;
    int64_t N;             // row pitch of the caller's arrays, in doubles
    std::mutex m;
    std::condition_variable cv_job, cv_free;
    std::deque<Job> jobs;
    std::vector<int> free_slots;
    bool finished = false;
    int in_flight = 0;
    unsigned chunk_no = 0;
    std::vector<std::thread> workers;

    Gatherer(HostCtx &ctx, int dev, int64_t n) : c(ctx), device(dev), N(n) {}

    // false: the pinned ring cannot be had (the caller then takes the pitched
    // copies, which need no staging memory)
    bool start()
    {
        if (!c.ring) {
            if (hipHostMalloc(&c.ring, RING_SLOTS * RING_SLOT_BYTES,
                              hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                c.ring = nullptr;
                return false;
            }
            for (int k = 0; k < RING_SLOTS; ++k)
                if (hipEventCreateWithFlags(&c.ring_ev[k],
                                            hipEventDisableTiming) != hipSuccess) {
                    (void)hipGetLastError();
                    for (int j = 0; j < k; ++j) {
                        (void)hipEventDestroy(c.ring_ev[j]);
                        c.ring_ev[j] = nullptr;
                    }
                    (void)hipHostFree(c.ring);
                    c.ring = nullptr;
                    return false;
                }
        }
        for (int k = 0; k < RING_SLOTS; ++k) free_slots.push_back(k);
        unsigned nt = std::thread::hardware_concurrency();
        if (nt > 12) nt = 12;
        if (nt < 2) nt = 2;
        if (rr_option(RR_OPT_GATHER_THREADS) > 0)
            nt = (unsigned)rr_option(RR_OPT_GATHER_THREADS);
        for (unsigned k = 0; k < nt; ++k)
            workers.emplace_back([this]() { work(); });
        return true;
    }

    void work()
    {
        (void)hipSetDevice(device);
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_job.wait(lk, [&] { return !jobs.empty() || finished; });
                if (jobs.empty()) return;
                j = jobs.front();
                jobs.pop_front();
            }
            (void)hipEventSynchronize(c.ring_ev[j.slot]);
            const char *src = (const char *)c.ring + (size_t)j.slot * RING_SLOT_BYTES;
            const size_t seg = (size_t)j.nc * 8;
            for (int64_t r = 0; r < j.rows; ++r)
                memcpy(j.host + (j.row0 + r) * N, src + (size_t)r * seg, seg);
            {
                std::lock_guard<std::mutex> lk(m);
                free_slots.push_back(j.slot);
                --in_flight;
            }
            cv_free.notify_all();
        }
    }

    // dev: dense [rows][nc] on the device; host: column i0 of row 0 of the
    // caller's [rows][N] array
    int gather(const double *dev, double *host, int64_t rows, int64_t nc)
    {
        const size_t seg = (size_t)nc * 8;
        int64_t per = (int64_t)(RING_SLOT_BYTES / seg);
        if (per < 1) {
            rr_set_error("gather: a row segment of %lld columns exceeds the "
                         "staging slot", (long long)nc);
            return RR_E_SIZE;
        }
        for (int64_t r0 = 0; r0 < rows; r0 += per) {
            const int64_t n = (rows - r0 < per) ? (rows - r0) : per;
            int slot;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_free.wait(lk, [&] { return !free_slots.empty(); });
                slot = free_slots.back();
                free_slots.pop_back();
                ++in_flight;
            }
            // (chunks alternate between two copy streams: two DMA engines)
            hipStream_t st = (chunk_no++ & 1) ? c.copy2 : c.copy;
            hipError_t e = hipMemcpyAsync(
                (char *)c.ring + (size_t)slot * RING_SLOT_BYTES, dev + r0 * nc,
                (size_t)n * seg, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipEventRecord(c.ring_ev[slot], st);
            if (e != hipSuccess) {
                // hand the slot back: finish() waits for in_flight == 0
                {
                    std::lock_guard<std::mutex> lk(m);
                    free_slots.push_back(slot);
                    --in_flight;
                }
                cv_free.notify_all();
                rr_set_error("gather: staging copy failed: %s",
                             hipGetErrorString(e));
                return RR_E_HIP;
            }
            {
                std::lock_guard<std::mutex> lk(m);
                jobs.push_back({slot, host, r0, n, nc});
            }
            cv_job.notify_one();
        }
        return RR_OK;
    }

    void finish()
    {
        {
            std::unique_lock<std::mutex> lk(m);
            cv_free.wait(lk, [&] { return in_flight == 0; });
            finished = true;
        }
        cv_job.notify_all();
        for (std::thread &t : workers) t.join();
        workers.clear();
    }
    ~Gatherer() { if (!workers.empty()) finish(); }
};

// Sweeps the N parameter sets in column blocks.  launch(i0, nc, slabs, d_sse,
// workspace, stream) enqueues block [i0, i0 + nc) on `stream`; while block
// k + 1 is computed, block k's slabs cross PCIe into the caller's [T][N]
// arrays (through the staging ring when the result is large, with pitched
// copies otherwise).  gr4j_family: the deferred x4 check of
// rr_gr4j_plan_status is made for every block.
// ldh: row pitch (in doubles) of the caller's arrays -- N, or more when this
// sweep fills a column block of a wider array (host_fan_out below).
// ws_bytes_of(nc): workspace a block of nc columns needs; the launch gets the
// size of the workspace it is handed.
template <class WsBytes, class Launch>
int sweep_blocks_run(HostCall &call, int64_t T, int64_t N, int64_t ldh,
                     const std::vector<OutSpec> &outs, double *sse_host,
                     WsBytes ws_bytes_of, bool gr4j_family, Launch launch)
{
    HostCtx &c = *call.c;
    const int64_t nc_max = pick_block(T, N, outs, ws_bytes_of);
    const size_t ws_bytes = ws_bytes_of(nc_max);
    const int64_t nb = rr_ceil_div(N, nc_max);
    const int nslab = nb > 1 ? 2 : 1;
    size_t out_bytes = 0;
    double *ptrs[2][MAX_OUTS] = {};
    for (int b = 0; b < nslab; ++b) {
        for (size_t k = 0; k < outs.size(); ++k)
            if (outs[k].host) {
                int rc = slot_reserve(c.slab[b][k],
                                      (size_t)T * outs[k].rows_per_t * nc_max * 8);
                if (rc != RR_OK) return rc;
                ptrs[b][k] = c.slab[b][k].as<double>();
                if (b == 0) out_bytes += (size_t)T * outs[k].rows_per_t * N * 8;
            }
        int rc = slot_reserve(c.ws[b], ws_bytes);
        if (rc != RR_OK) return rc;
        if (sse_host && (rc = slot_reserve(c.sse[b], (size_t)nc_max * 8)) != RR_OK)
            return rc;
    }
    auto block = [&](int64_t k, int64_t &i0, int64_t &nc) {
        i0 = k * nc_max;
        nc = (N - i0 < nc_max) ? (N - i0) : nc_max;
    };
    auto enqueue = [&](int64_t k) {
        int64_t i0, nc;
        block(k, i0, nc);
        const int b = (int)(k & (nslab - 1));
        // the slab is free once block k - 2 has left it
        if (k >= 2) {
            RR_HIP(hipStreamWaitEvent(c.compute, c.copied[b], 0));
            RR_HIP(hipStreamWaitEvent(c.compute, c.copied2[b], 0));
        }
        int rc = launch(i0, nc, ptrs[b], sse_host ? c.sse[b].as<double>() : nullptr,
                        c.ws[b].p, ws_bytes, (void *)c.compute);
        if (rc != RR_OK) return rc;
        RR_HIP(hipEventRecord(c.done[b], c.compute));
        return RR_OK;
    };

    int dev = 0;
    RR_HIP(hipGetDevice(&dev));
    Gatherer ring(c, dev, ldh);
    bool staged = out_bytes >= ((size_t)64 << 20) &&
                  (size_t)nc_max * 8 <= RING_SLOT_BYTES;
    int rc = enqueue(0);
    if (rc != RR_OK) return rc;
    if (staged && !ring.start()) staged = false;
    for (int64_t k = 0; k < nb; ++k) {
        int64_t i0, nc;
        block(k, i0, nc);
        const int b = (int)(k & (nslab - 1));
        // block k + 1 computes while block k is gathered
        if (k + 1 < nb && (rc = enqueue(k + 1)) != RR_OK) return rc;
        RR_HIP(hipStreamWaitEvent(c.copy, c.done[b], 0));
        RR_HIP(hipStreamWaitEvent(c.copy2, c.done[b], 0));
        if (gr4j_family &&
            (rc = rr_gr4j_plan_status(c.ws[b].p, (void *)c.copy)) != RR_OK)
            return rc;
        for (size_t o = 0; o < outs.size(); ++o) {
            if (!outs[o].host) continue;
            const int64_t rows = T * outs[o].rows_per_t;
            if (staged) {
                if ((rc = ring.gather(ptrs[b][o], outs[o].host + i0, rows,
                                      nc)) != RR_OK)
                    return rc;
            } else {
                RR_HIP(hipMemcpy2DAsync(outs[o].host + i0, (size_t)ldh * 8,
                                        ptrs[b][o], (size_t)nc * 8,
                                        (size_t)nc * 8, (size_t)rows,
                                        hipMemcpyDeviceToHost, c.copy));
            }
        }
        RR_HIP(hipEventRecord(c.copied[b], c.copy));
        RR_HIP(hipEventRecord(c.copied2[b], c.copy2));
        if (sse_host) {
            const size_t bytes = (size_t)nc * 8;
            if (bytes <= PINNED_BYTES / 2) {    // one DMA into pinned memory
                void *bounce = (char *)c.pinned + PINNED_BYTES / 2;
                RR_HIP(hipMemcpyAsync(bounce, c.sse[b].p, bytes,
                                      hipMemcpyDeviceToHost, c.copy));
                RR_HIP(hipStreamSynchronize(c.copy));
                memcpy(sse_host + i0, bounce, bytes);
            } else {
                RR_HIP(hipMemcpyAsync(sse_host + i0, c.sse[b].p, bytes,
                                      hipMemcpyDeviceToHost, c.copy));
                RR_HIP(hipStreamSynchronize(c.copy));
            }
        }
    }
    if (staged) ring.finish();
    RR_HIP(hipStreamSynchronize(c.copy));
    RR_HIP(hipStreamSynchronize(c.copy2));
    return RR_OK;
}

template <class WsBytes, class Launch>
int sweep_blocks(HostCall &call, int64_t T, int64_t N, int64_t ldh,
                 const std::vector<OutSpec> &outs, double *sse_host,
                 WsBytes ws_bytes_of, bool gr4j_family, Launch launch)
{
    const int rc = sweep_blocks_run(call, T, N, ldh, outs, sse_host,
                                    ws_bytes_of, gr4j_family, launch);
    if (rc != RR_OK) {
        // an early exit may leave kernels and copies in flight that still
        // use the slabs and the pinned bounce buffer the next call reuses:
        // drain them (the error message of the failing step is kept)
        HostCtx &c = *call.c;
        char keep[sizeof(g_err)];
        memcpy(keep, g_err, sizeof(keep));
        (void)hipStreamSynchronize(c.compute);
        (void)hipStreamSynchronize(c.copy);
        (void)hipStreamSynchronize(c.copy2);
        (void)hipGetLastError();
        memcpy(g_err, keep, sizeof(keep));
    }
    return rc;
}

// ---- the parameter-set axis over several GPUs, inside one call ----------------
// RR_OPT_HOST_SHARDS = S > 1 (or -1: one shard per visible device) cuts the N
// sets of a host-pointer call into S contiguous shards (the partition of
// rrmpg_amd/sharding.py: lengths differ by at most one, the longer ones
// first); shard j runs on device (current + j) % device_count from a thread
// of its own, with that device's context, and fills ITS columns of the
// caller's [T][N] arrays (row pitch N).  No collective: the sets are
// independent, the forcing is uploaded to every device (<= 1.4 MB), and the
// scores land in the caller's vector.  body(first, n) runs sets
// [first, first + n).  More shards than devices is allowed (they share a
// device, serialised by its context): the path can be exercised on one GPU.
// Largest x4 of a host parameter block (NaN ignored: the device scan reports
// it) -- sizes the unit-hydrograph scratch of the GR4J-family sweeps.
static double host_max_x4(const double *params, int64_t n, int stride, int idx)
{
    double m = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        const double v = params[i * stride + idx];
        if (v > m) m = v;
    }
    return m;
}
#define RR_HOST_X4_LIMIT 1e5
static int check_host_x4(const char *who, double max_x4)
{
    if (max_x4 > RR_HOST_X4_LIMIT) {
        rr_set_error("%s: x4 up to %g: unit hydrographs longer than %g days "
                     "are not supported", who, max_x4, RR_HOST_X4_LIMIT);
        return RR_E_PARAM;
    }
    return RR_OK;
}

static inline double *rr_col(double *a, int64_t first)
{
    return a ? a + first : nullptr;
}

template <class Body>
int host_fan_out(int64_t N, Body body)
{
    int64_t shards = rr_option(RR_OPT_HOST_SHARDS);
    const int ndev = rr_device_count();
    if (shards < 0) shards = ndev;
    if (shards > N) shards = N;
    if (shards <= 1 || ndev < 1) return body((int64_t)0, N);
    int cur = 0;
    RR_HIP(hipGetDevice(&cur));
    std::vector<int> rcs((size_t)shards, RR_OK);
    std::vector<std::string> msgs((size_t)shards);
    std::vector<std::thread> th;
    const int64_t base = N / shards, extra = N % shards;
    // (the shard threads see the options the calling thread sees)
    const rr_call_options *const call_opt = tl_call;
    const bool standing_on = tl_standing_on;
    const rr_call_options standing = tl_standing;
    for (int64_t j = 0; j < shards; ++j) {
        const int64_t first = j * base + (j < extra ? j : extra);
        const int64_t n = base + (j < extra ? 1 : 0);
        th.emplace_back([&, j, first, n]() {
            tl_call = call_opt;
            tl_standing = standing;
            tl_standing_on = standing_on;
            if (hipSetDevice((cur + (int)j) % ndev) != hipSuccess) {
                (void)hipGetLastError();
                rcs[(size_t)j] = RR_E_HIP;
                msgs[(size_t)j] = "hipSetDevice failed in a shard thread";
                return;
            }
            rcs[(size_t)j] = body(first, n);
            if (rcs[(size_t)j] != RR_OK) msgs[(size_t)j] = g_err;
        });
    }
    for (std::thread &t : th) t.join();
    for (int64_t j = 0; j < shards; ++j)
        if (rcs[(size_t)j] != RR_OK) {
            rr_set_error("%s", msgs[(size_t)j].c_str());
            return rcs[(size_t)j];
        }
    return RR_OK;
}

}  // namespace

extern "C" int rr_release_cached_memory(void)
{
    int dev = 0;
    if (rr_device_count() < 1) return RR_OK;
    RR_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEVICES) return RR_OK;
    HostCtx &c = g_ctx[dev];
    std::lock_guard<std::mutex> g(c.mu);
    for (Slot &s : c.in) slot_free(s);
    slot_free(c.par);
    for (int k = 0; k < 2; ++k) {
        slot_free(c.ws[k]); slot_free(c.sse[k]);
        for (Slot &s : c.slab[k]) slot_free(s);
    }
    if (c.ring) {                       // the 256-MiB pinned staging ring
        (void)hipHostFree(c.ring);
        c.ring = nullptr;
        for (int k = 0; k < RING_SLOTS; ++k) {
            if (c.ring_ev[k]) (void)hipEventDestroy(c.ring_ev[k]);
            c.ring_ev[k] = nullptr;
        }
    }
    return RR_OK;
}

extern "C" int rr_abc_simulate_opt(const double *prec, int64_t T,
                               double initial_state, const double *params,
                               int64_t N, double *qsim, double *storage,
                               const double *qobs, double *sse,
    const rr_call_options *opt)
{
    {
        const int orc = check_call_options("rr_abc_simulate_opt", opt);
        if (orc != RR_OK) return orc;
    }
    const CallOptionsScope options_scope(opt);

    int rc = rr_check_common("rr_abc_simulate", T, N, N, params, qobs, sse);
    if (rc != RR_OK) return rc;
    if ((rc = rr_check_outputs("rr_abc_simulate", qsim,
                               storage != nullptr)) != RR_OK)
        return rc;
    if (T == 0 || N == 0) return RR_OK;
    if (!prec) { rr_set_error("rr_abc_simulate: prec is NULL"); return RR_E_NULL; }
    return host_fan_out(N, [&](int64_t first, int64_t n) -> int {
        int rc;
        HostCall call;
        if ((rc = call.open("rr_abc_simulate")) != RR_OK) return rc;
        const void *d_prec, *d_qobs = nullptr;
        const double *d_par;
        if ((rc = call.input(prec, (size_t)T * 8, &d_prec)) != RR_OK) return rc;
        if ((rc = call.input(qobs, qobs ? (size_t)T * 8 : 0, &d_qobs)) != RR_OK)
            return rc;
        if ((rc = call.params(params + first * 3, (size_t)n * 3 * 8, &d_par)) != RR_OK) return rc;
        auto wsb = [&](int64_t nc) { return rr_abc_workspace_bytes(T, nc); };
        std::vector<OutSpec> outs = {{rr_col(qsim, first), 1}, {rr_col(storage, first), 1}};
        return sweep_blocks(call, T, n, N, outs, rr_col(sse, first), wsb, false,
            [&](int64_t i0, int64_t nc, double **o, double *d_sse, void *ws,
                size_t ws_size, void *st) {
                return rr_abc_simulate_dev(
                    (const double *)d_prec, T, initial_state, d_par + i0 * 3, nc,
                    o[0], o[1], nc, qobs ? (const double *)d_qobs : nullptr,
                    qobs ? d_sse : nullptr, ws, ws_size, st);
            });
    });
}

extern "C" int rr_abc_simulate(const double *prec, int64_t T,
                               double initial_state, const double *params,
                               int64_t N, double *qsim, double *storage,
                               const double *qobs, double *sse)
{
    return rr_abc_simulate_opt(prec, T, initial_state, params, N, qsim, storage, qobs, sse, nullptr);
}

extern "C" int rr_hbvedu_simulate_opt(
    const double *temp, const double *prec, const int8_t *month,
    const double *PE_m, const double *T_m, int64_t T, double snow_init,
    double soil_init, double s1_init, double s2_init, const double *params,
    int64_t N, double *qsim, double *snow, double *soil, double *s1,
    double *s2, const double *qobs, double *sse,
    const rr_call_options *opt)
{
    {
        const int orc = check_call_options("rr_hbvedu_simulate_opt", opt);
        if (orc != RR_OK) return orc;
    }
    const CallOptionsScope options_scope(opt);

    int rc = rr_check_common("rr_hbvedu_simulate", T, N, N, params, qobs, sse);
    if (rc != RR_OK) return rc;
    if ((rc = rr_check_outputs("rr_hbvedu_simulate", qsim,
                               snow || soil || s1 || s2)) != RR_OK)
        return rc;
    if (T == 0 || N == 0) return RR_OK;
    if (!temp || !prec || !month || !PE_m || !T_m) {
        rr_set_error("rr_hbvedu_simulate: NULL forcing pointer");
        return RR_E_NULL;
    }
    return host_fan_out(N, [&](int64_t first, int64_t n) -> int {
        int rc;
        HostCall call;
        if ((rc = call.open("rr_hbvedu_simulate")) != RR_OK) return rc;
        const void *d_temp, *d_prec, *d_month, *d_pe, *d_tm, *d_qobs = nullptr;
        const double *d_par;
        if ((rc = call.input(temp, (size_t)T * 8, &d_temp)) != RR_OK) return rc;
        if ((rc = call.input(prec, (size_t)T * 8, &d_prec)) != RR_OK) return rc;
        if ((rc = call.input(month, (size_t)T, &d_month)) != RR_OK) return rc;
        if ((rc = call.input(PE_m, 12 * 8, &d_pe)) != RR_OK) return rc;
        if ((rc = call.input(T_m, 12 * 8, &d_tm)) != RR_OK) return rc;
        if ((rc = call.input(qobs, qobs ? (size_t)T * 8 : 0, &d_qobs)) != RR_OK)
            return rc;
        if ((rc = call.params(params + first * 11, (size_t)n * 11 * 8, &d_par)) != RR_OK)
            return rc;
        auto wsb = [&](int64_t nc) { return rr_hbvedu_workspace_bytes(T, nc); };
        std::vector<OutSpec> outs = {{rr_col(qsim, first), 1}, {rr_col(snow, first), 1}, {rr_col(soil, first), 1}, {rr_col(s1, first), 1},
                                     {rr_col(s2, first), 1}};
        return sweep_blocks(call, T, n, N, outs, rr_col(sse, first), wsb, false,
            [&](int64_t i0, int64_t nc, double **o, double *d_sse, void *ws,
                size_t ws_size, void *st) {
                return rr_hbvedu_simulate_dev(
                    (const double *)d_temp, (const double *)d_prec,
                    (const int8_t *)d_month, (const double *)d_pe,
                    (const double *)d_tm, T, snow_init, soil_init, s1_init,
                    s2_init, d_par + i0 * 11, nc, o[0], o[1], o[2], o[3], o[4],
                    nc, qobs ? (const double *)d_qobs : nullptr,
                    qobs ? d_sse : nullptr, ws, ws_size, st);
            });
    });
}

extern "C" int rr_hbvedu_simulate(
    const double *temp, const double *prec, const int8_t *month,
    const double *PE_m, const double *T_m, int64_t T, double snow_init,
    double soil_init, double s1_init, double s2_init, const double *params,
    int64_t N, double *qsim, double *snow, double *soil, double *s1,
    double *s2, const double *qobs, double *sse)
{
    return rr_hbvedu_simulate_opt(temp, prec, month, PE_m, T_m, T, snow_init, soil_init, s1_init, s2_init, params, N, qsim, snow, soil, s1, s2, qobs, sse, nullptr);
}

extern "C" int rr_gr4j_simulate_opt(const double *prec, const double *etp,
                                int64_t T, double s_init, double r_init,
                                const double *params, int64_t N, double *qsim,
                                double *s_store, double *r_store,
                                const double *qobs, double *sse,
    const rr_call_options *opt)
{
    {
        const int orc = check_call_options("rr_gr4j_simulate_opt", opt);
        if (orc != RR_OK) return orc;
    }
    const CallOptionsScope options_scope(opt);

    int rc = rr_check_common("rr_gr4j_simulate", T, N, N, params, qobs, sse);
    if (rc != RR_OK) return rc;
    if ((rc = rr_check_outputs("rr_gr4j_simulate", qsim,
                               s_store || r_store)) != RR_OK)
        return rc;
    if (T == 0 || N == 0) return RR_OK;
    if (!prec || !etp) {
        rr_set_error("rr_gr4j_simulate: NULL forcing pointer");
        return RR_E_NULL;
    }
    return host_fan_out(N, [&](int64_t first, int64_t n) -> int {
        int rc;
        HostCall call;
        if ((rc = call.open("rr_gr4j_simulate")) != RR_OK) return rc;
        const void *d_prec, *d_etp, *d_qobs = nullptr;
        const double *d_par;
        if ((rc = call.input(prec, (size_t)T * 8, &d_prec)) != RR_OK) return rc;
        if ((rc = call.input(etp, (size_t)T * 8, &d_etp)) != RR_OK) return rc;
        if ((rc = call.input(qobs, qobs ? (size_t)T * 8 : 0, &d_qobs)) != RR_OK)
            return rc;
        if ((rc = call.params(params + first * 4, (size_t)n * 4 * 8, &d_par)) != RR_OK) return rc;
        const double max_x4 = host_max_x4(params + first * 4, n, 4, 3);
        if ((rc = check_host_x4("rr_gr4j_simulate", max_x4)) != RR_OK)
            return rc;
        auto wsb = [&](int64_t nc) {
            return rr_gr4j_workspace_bytes_x4(T, nc, max_x4);
        };
        std::vector<OutSpec> outs = {{rr_col(qsim, first), 1}, {rr_col(s_store, first), 1}, {rr_col(r_store, first), 1}};
        return sweep_blocks(call, T, n, N, outs, rr_col(sse, first), wsb, true,
            [&](int64_t i0, int64_t nc, double **o, double *d_sse, void *ws,
                size_t ws_size, void *st) {
                return rr_gr4j_simulate_dev(
                    (const double *)d_prec, (const double *)d_etp, T, s_init,
                    r_init, d_par + i0 * 4, nc, o[0], o[1], o[2], nc,
                    qobs ? (const double *)d_qobs : nullptr,
                    qobs ? d_sse : nullptr, ws, ws_size, st);
            });
    });
}

extern "C" int rr_gr4j_simulate(const double *prec, const double *etp,
                                int64_t T, double s_init, double r_init,
                                const double *params, int64_t N, double *qsim,
                                double *s_store, double *r_store,
                                const double *qobs, double *sse)
{
    return rr_gr4j_simulate_opt(prec, etp, T, s_init, r_init, params, N, qsim, s_store, r_store, qobs, sse, nullptr);
}

extern "C" int rr_cemaneige_simulate_opt(
    const double *prec, const double *mean_temp, const double *frac_solid_prec,
    int64_t T, int64_t L, double snow_pack_init, double thermal_state_init,
    const double *params, int64_t N, double *outflow, double *G, double *eTG,
    const double *qobs, double *sse,
    const rr_call_options *opt)
{
    {
        const int orc = check_call_options("rr_cemaneige_simulate_opt", opt);
        if (orc != RR_OK) return orc;
    }
    const CallOptionsScope options_scope(opt);

    int rc = rr_check_common("rr_cemaneige_simulate", T, N, N, params, qobs,
                             sse);
    if (rc != RR_OK) return rc;
    if ((rc = rr_check_outputs("rr_cemaneige_simulate", outflow,
                               G || eTG)) != RR_OK)
        return rc;
    if (T == 0 || N == 0) return RR_OK;
    if (L < 1) { rr_set_error("rr_cemaneige_simulate: L < 1"); return RR_E_PARAM; }
    if (!prec || !mean_temp || !frac_solid_prec) {
        rr_set_error("rr_cemaneige_simulate: NULL forcing pointer");
        return RR_E_NULL;
    }
    return host_fan_out(N, [&](int64_t first, int64_t n) -> int {
        int rc;
        HostCall call;
        if ((rc = call.open("rr_cemaneige_simulate")) != RR_OK) return rc;
        const void *d_prec, *d_temp, *d_frac, *d_qobs = nullptr;
        const double *d_par;
        const size_t tl = (size_t)T * (size_t)L * 8;
        if ((rc = call.input(prec, tl, &d_prec)) != RR_OK) return rc;
        if ((rc = call.input(mean_temp, tl, &d_temp)) != RR_OK) return rc;
        if ((rc = call.input(frac_solid_prec, tl, &d_frac)) != RR_OK) return rc;
        if ((rc = call.input(qobs, qobs ? (size_t)T * 8 : 0, &d_qobs)) != RR_OK)
            return rc;
        if ((rc = call.params(params + first * 2, (size_t)n * 2 * 8, &d_par)) != RR_OK) return rc;
        auto wsb = [&](int64_t nc) {
            return rr_cemaneige_workspace_bytes(T, L, nc);
        };
        std::vector<OutSpec> outs = {{rr_col(outflow, first), 1}, {rr_col(G, first), L}, {rr_col(eTG, first), L}};
        return sweep_blocks(call, T, n, N, outs, rr_col(sse, first), wsb, false,
            [&](int64_t i0, int64_t nc, double **o, double *d_sse, void *ws,
                size_t ws_size, void *st) {
                return rr_cemaneige_simulate_dev(
                    (const double *)d_prec, (const double *)d_temp,
                    (const double *)d_frac, T, L, snow_pack_init,
                    thermal_state_init, d_par + i0 * 2, nc, o[0], o[1], o[2], nc,
                    qobs ? (const double *)d_qobs : nullptr,
                    qobs ? d_sse : nullptr, ws, ws_size, st);
            });
    });
}

extern "C" int rr_cemaneige_simulate(
    const double *prec, const double *mean_temp, const double *frac_solid_prec,
    int64_t T, int64_t L, double snow_pack_init, double thermal_state_init,
    const double *params, int64_t N, double *outflow, double *G, double *eTG,
    const double *qobs, double *sse)
{
    return rr_cemaneige_simulate_opt(prec, mean_temp, frac_solid_prec, T, L, snow_pack_init, thermal_state_init, params, N, outflow, G, eTG, qobs, sse, nullptr);
}

extern "C" int rr_cemaneigegr4j_simulate_opt(
    const double *prec, const double *mean_temp, const double *etp,
    const double *frac_solid_prec, int64_t T, int64_t L,
    double snow_pack_init, double thermal_state_init, double s_init,
    double r_init, const double *params, int64_t N, double *qsim, double *G,
    double *eTG, double *s_store, double *r_store, const double *qobs,
    double *sse,
    const rr_call_options *opt)
{
    {
        const int orc = check_call_options("rr_cemaneigegr4j_simulate_opt", opt);
        if (orc != RR_OK) return orc;
    }
    const CallOptionsScope options_scope(opt);

    int rc = rr_check_common("rr_cemaneigegr4j_simulate", T, N, N, params,
                             qobs, sse);
    if (rc != RR_OK) return rc;
    if ((rc = rr_check_outputs("rr_cemaneigegr4j_simulate", qsim,
                               G || eTG || s_store || r_store)) != RR_OK)
        return rc;
    if (T == 0 || N == 0) return RR_OK;
    if (L < 1) { rr_set_error("rr_cemaneigegr4j_simulate: L < 1"); return RR_E_PARAM; }
    if (!prec || !mean_temp || !etp || !frac_solid_prec) {
        rr_set_error("rr_cemaneigegr4j_simulate: NULL forcing pointer");
        return RR_E_NULL;
    }
    return host_fan_out(N, [&](int64_t first, int64_t n) -> int {
        int rc;
        HostCall call;
        if ((rc = call.open("rr_cemaneigegr4j_simulate")) != RR_OK) return rc;
        const void *d_prec, *d_temp, *d_etp, *d_frac, *d_qobs = nullptr;
        const double *d_par;
        const size_t tl = (size_t)T * (size_t)L * 8;
        if ((rc = call.input(prec, tl, &d_prec)) != RR_OK) return rc;
        if ((rc = call.input(mean_temp, tl, &d_temp)) != RR_OK) return rc;
        if ((rc = call.input(etp, (size_t)T * 8, &d_etp)) != RR_OK) return rc;
        if ((rc = call.input(frac_solid_prec, tl, &d_frac)) != RR_OK) return rc;
        if ((rc = call.input(qobs, qobs ? (size_t)T * 8 : 0, &d_qobs)) != RR_OK)
            return rc;
        if ((rc = call.params(params + first * 6, (size_t)n * 6 * 8, &d_par)) != RR_OK) return rc;
        const double max_x4 = host_max_x4(params + first * 6, n, 6, 5);
        if ((rc = check_host_x4("rr_cemaneigegr4j_simulate", max_x4)) != RR_OK)
            return rc;
        auto wsb = [&](int64_t nc) {
            return rr_cemaneigegr4j_workspace_bytes_x4(T, L, nc, max_x4);
        };
        std::vector<OutSpec> outs = {{rr_col(qsim, first), 1}, {rr_col(G, first), L}, {rr_col(eTG, first), L}, {rr_col(s_store, first), 1},
                                     {rr_col(r_store, first), 1}};
        return sweep_blocks(call, T, n, N, outs, rr_col(sse, first), wsb, true,
            [&](int64_t i0, int64_t nc, double **o, double *d_sse, void *ws,
                size_t ws_size, void *st) {
                return rr_cemaneigegr4j_simulate_dev(
                    (const double *)d_prec, (const double *)d_temp,
                    (const double *)d_etp, (const double *)d_frac, T, L,
                    snow_pack_init, thermal_state_init, s_init, r_init,
                    d_par + i0 * 6, nc, o[0], o[1], o[2], o[3], o[4], nc,
                    qobs ? (const double *)d_qobs : nullptr,
                    qobs ? d_sse : nullptr, ws, ws_size, st);
            });
    });
}

extern "C" int rr_cemaneigegr4j_simulate(
    const double *prec, const double *mean_temp, const double *etp,
    const double *frac_solid_prec, int64_t T, int64_t L,
    double snow_pack_init, double thermal_state_init, double s_init,
    double r_init, const double *params, int64_t N, double *qsim, double *G,
    double *eTG, double *s_store, double *r_store, const double *qobs,
    double *sse)
{
    return rr_cemaneigegr4j_simulate_opt(prec, mean_temp, etp, frac_solid_prec, T, L, snow_pack_init, thermal_state_init, s_init, r_init, params, N, qsim, G, eTG, s_store, r_store, qobs, sse, nullptr);
}

// ---- device self-test hook (not part of include/rrhip.h) -------------------
// out[i] = div_by_invariant_m(a[i], b[i]) and ref[i] = a[i] / b[i], host arrays.
__global__ void dbg_div_kernel(const double *a, const double *b, double *out,
                               double *ref, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const double av = (i < n) ? a[i] : 1.0, bv = (i < n) ? b[i] : 1.0;
    const InvDivisor d = make_inv_divisor(bv);
    const double q = div_by_invariant_m(av, inv_div_numerator_mask0(av), d,
                                        RR_LANES(d.ok));
    if (i < n) {
        out[i] = q;
        ref[i] = av / bv;
    }
}

extern "C" int rrdbg_divide_by_invariant(const double *a, const double *b,
                                         double *out, double *ref, int64_t n)
{
    if (rr_device_count() < 1) {
        rr_set_error("no HIP device visible: librrhip has no CPU path");
        return RR_E_NODEVICE;
    }
    double *d[4] = {nullptr, nullptr, nullptr, nullptr};
    const size_t bytes = (size_t)(n > 0 ? n : 1) * 8;
    int rc = RR_OK;
    for (int k = 0; k < 4 && rc == RR_OK; ++k)
        if (hipMalloc((void **)&d[k], bytes) != hipSuccess) {
            rr_set_error("rrdbg_divide_by_invariant: hipMalloc failed");
            rc = RR_E_HIP;
        }
    if (rc == RR_OK &&
        (hipMemcpy(d[0], a, (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess ||
         hipMemcpy(d[1], b, (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess))
        rc = RR_E_HIP;
    if (rc == RR_OK) {
        hipLaunchKernelGGL(dbg_div_kernel, dim3((unsigned)rr_ceil_div(n, 256)),
                           dim3(256), 0, nullptr, d[0], d[1], d[2], d[3], n);
        if (hipMemcpy(out, d[2], (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(ref, d[3], (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess)
            rc = RR_E_HIP;
    }
    for (int k = 0; k < 4; ++k)
        if (d[k]) (void)hipFree(d[k]);
    if (rc == RR_E_HIP) rr_set_error("rrdbg_divide_by_invariant: HIP call failed");
    return rc;
}

// ---- device test hook (not part of include/rrhip.h) --------------------------
// A kernel that does nothing but hold the GPU: `blocks` workgroups of 256
// threads spin until `microseconds` of wall time have passed since each
// started.  tests/test_gpu_async.py runs two time-tiled sweeps beside it: the
// tiles' tickets (common.h) must get every sweep through whatever else shares
// the device and in whatever order its workgroups start.
__global__ void dbg_spin_kernel(unsigned long long ticks)
{
    const unsigned long long t0 = __builtin_readcyclecounter();
    (void)t0;
    const unsigned long long r0 = wall_clock64();
    while (wall_clock64() - r0 < ticks) __builtin_amdgcn_s_sleep(32);
}

extern "C" int rrdbg_spin_dev(int blocks, int64_t microseconds, void *stream)
{
    if (blocks < 1 || microseconds < 0 || microseconds > 5000000) {
        rr_set_error("rrdbg_spin_dev: bad arguments");
        return RR_E_PARAM;
    }
    // wall_clock64 counts at 100 MHz on gfx9
    hipLaunchKernelGGL(dbg_spin_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream,
                       (unsigned long long)microseconds * 100ull);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// ---- next tier: hysteresis / ice-melt couplings (host pointers) ------------
namespace {

enum SnowVariant { HYST = 1, ICE = 2 };

int snow_gr4j_host(const char *who, int variant, const double *prec,
                   const double *mean_temp, const double *etp,
                   const double *frac_ice, const double *frac_solid_prec,
                   int64_t T, int64_t L, double snow_pack_init,
                   double thermal_state_init, double sca_init, double s_init,
                   double r_init, const double *params, int64_t N,
                   double *qsim, double *G, double *eTG, double *s_store,
                   double *r_store, double *sca, double *icemelt,
                   double *snowmelt, const double *qobs, double *sse,
                   const rr_call_options *opt)
{
    {
        const int orc = check_call_options(who, opt);
        if (orc != RR_OK) return orc;
    }
    const CallOptionsScope options_scope(opt);
    const bool hyst = variant & HYST, ice = variant & ICE;
    const int npar = 6 + (hyst ? 2 : 0) + (ice ? 1 : 0);
    int rc = rr_check_common(who, T, N, N, params, qobs, sse);
    if (rc != RR_OK) return rc;
    if ((rc = rr_check_outputs(who, qsim,
                               G || eTG || s_store || r_store || sca || icemelt ||
                                   snowmelt)) != RR_OK)
        return rc;
    if (T == 0 || N == 0) return RR_OK;
    if (L < 1) { rr_set_error("%s: L < 1", who); return RR_E_PARAM; }
    if (!prec || !mean_temp || !etp || !frac_solid_prec || (ice && !frac_ice)) {
        rr_set_error("%s: NULL forcing pointer", who);
        return RR_E_NULL;
    }
    return host_fan_out(N, [&](int64_t first, int64_t n) -> int {
        int rc;
        HostCall call;
        if ((rc = call.open(who)) != RR_OK) return rc;
        const void *d_prec, *d_temp, *d_etp, *d_frac, *d_fice = nullptr,
                   *d_qobs = nullptr;
        const double *d_par;
        const size_t tl = (size_t)T * (size_t)L * 8;
        if ((rc = call.input(prec, tl, &d_prec)) != RR_OK) return rc;
        if ((rc = call.input(mean_temp, tl, &d_temp)) != RR_OK) return rc;
        if ((rc = call.input(etp, (size_t)T * 8, &d_etp)) != RR_OK) return rc;
        if ((rc = call.input(frac_solid_prec, tl, &d_frac)) != RR_OK) return rc;
        if ((rc = call.input(frac_ice, ice ? (size_t)L * 8 : 0, &d_fice)) != RR_OK)
            return rc;
        if ((rc = call.input(qobs, qobs ? (size_t)T * 8 : 0, &d_qobs)) != RR_OK)
            return rc;
        if ((rc = call.params(params + first * npar, (size_t)n * npar * 8, &d_par)) != RR_OK)
            return rc;
        const double max_x4 = host_max_x4(params + first * npar, n, npar,
                                          (hyst ? 4 : 2) + 3);
        if ((rc = check_host_x4(who, max_x4)) != RR_OK) return rc;
        auto wsb = [&](int64_t nc) {
            return rr_snowgr4j_workspace_bytes_x4(T, L, nc, max_x4);
        };
        std::vector<OutSpec> outs = {{rr_col(qsim, first), 1}, {rr_col(G, first), L}, {rr_col(eTG, first), L}, {rr_col(s_store, first), 1},
                                     {rr_col(r_store, first), 1}, {rr_col(sca, first), L}, {rr_col(icemelt, first), 1},
                                     {rr_col(snowmelt, first), 1}};
        return sweep_blocks(call, T, n, N, outs, rr_col(sse, first), wsb, true,
            [&](int64_t i0, int64_t nc, double **o, double *d_sse, void *ws,
                size_t ws_size, void *st) {
                const double *p = d_par + i0 * npar;
                const double *qo = qobs ? (const double *)d_qobs : nullptr;
                double *so = qobs ? d_sse : nullptr;
                const double *pr = (const double *)d_prec,
                             *tm = (const double *)d_temp,
                             *et = (const double *)d_etp,
                             *fr = (const double *)d_frac,
                             *fi = (const double *)d_fice;
                if (hyst && ice)
                    return rr_cemaneigehystgr4jice_simulate_dev(
                        pr, tm, et, fi, fr, T, L, snow_pack_init,
                        thermal_state_init, sca_init, s_init, r_init, p, nc, o[0],
                        o[1], o[2], o[3], o[4], o[5], o[6], o[7], nc, qo, so, ws,
                        ws_size, st);
                if (hyst)
                    return rr_cemaneigehystgr4j_simulate_dev(
                        pr, tm, et, fr, T, L, snow_pack_init, thermal_state_init,
                        sca_init, s_init, r_init, p, nc, o[0], o[1], o[2], o[3],
                        o[4], o[5], nc, qo, so, ws, ws_size, st);
                return rr_cemaneigegr4jice_simulate_dev(
                    pr, tm, et, fi, fr, T, L, snow_pack_init, thermal_state_init,
                    s_init, r_init, p, nc, o[0], o[1], o[2], o[3], o[4], o[6], nc,
                    qo, so, ws, ws_size, st);
            });
    });
}

}  // namespace

extern "C" int rr_cemaneigehystgr4j_simulate_opt(
    const double *prec, const double *mean_temp, const double *etp,
    const double *frac_solid_prec, int64_t T, int64_t L,
    double snow_pack_init, double thermal_state_init, double sca_init,
    double s_init, double r_init, const double *params, int64_t N,
    double *qsim, double *G, double *eTG, double *s_store, double *r_store,
    double *sca, const double *qobs, double *sse,
    const rr_call_options *opt)
{

    return snow_gr4j_host("rr_cemaneigehystgr4j_simulate", HYST, prec,
                          mean_temp, etp, nullptr, frac_solid_prec, T, L,
                          snow_pack_init, thermal_state_init, sca_init, s_init,
                          r_init, params, N, qsim, G, eTG, s_store, r_store,
                          sca, nullptr, nullptr, qobs, sse, opt);
}

extern "C" int rr_cemaneigehystgr4j_simulate(
    const double *prec, const double *mean_temp, const double *etp,
    const double *frac_solid_prec, int64_t T, int64_t L,
    double snow_pack_init, double thermal_state_init, double sca_init,
    double s_init, double r_init, const double *params, int64_t N,
    double *qsim, double *G, double *eTG, double *s_store, double *r_store,
    double *sca, const double *qobs, double *sse)
{
    return rr_cemaneigehystgr4j_simulate_opt(prec, mean_temp, etp, frac_solid_prec, T, L, snow_pack_init, thermal_state_init, sca_init, s_init, r_init, params, N, qsim, G, eTG, s_store, r_store, sca, qobs, sse, nullptr);
}

extern "C" int rr_cemaneigegr4jice_simulate_opt(
    const double *prec, const double *mean_temp, const double *etp,
    const double *frac_ice, const double *frac_solid_prec, int64_t T,
    int64_t L, double snow_pack_init, double thermal_state_init,
    double s_init, double r_init, const double *params, int64_t N,
    double *qsim, double *G, double *eTG, double *s_store, double *r_store,
    double *icemelt, const double *qobs, double *sse,
    const rr_call_options *opt)
{

    return snow_gr4j_host("rr_cemaneigegr4jice_simulate", ICE, prec, mean_temp,
                          etp, frac_ice, frac_solid_prec, T, L, snow_pack_init,
                          thermal_state_init, 0.0, s_init, r_init, params, N,
                          qsim, G, eTG, s_store, r_store, nullptr, icemelt,
                          nullptr, qobs, sse, opt);
}

extern "C" int rr_cemaneigegr4jice_simulate(
    const double *prec, const double *mean_temp, const double *etp,
    const double *frac_ice, const double *frac_solid_prec, int64_t T,
    int64_t L, double snow_pack_init, double thermal_state_init,
    double s_init, double r_init, const double *params, int64_t N,
    double *qsim, double *G, double *eTG, double *s_store, double *r_store,
    double *icemelt, const double *qobs, double *sse)
{
    return rr_cemaneigegr4jice_simulate_opt(prec, mean_temp, etp, frac_ice, frac_solid_prec, T, L, snow_pack_init, thermal_state_init, s_init, r_init, params, N, qsim, G, eTG, s_store, r_store, icemelt, qobs, sse, nullptr);
}

extern "C" int rr_cemaneigehystgr4jice_simulate_opt(
    const double *prec, const double *mean_temp, const double *etp,
    const double *frac_ice, const double *frac_solid_prec, int64_t T,
    int64_t L, double snow_pack_init, double thermal_state_init,
    double sca_init, double s_init, double r_init, const double *params,
    int64_t N, double *qsim, double *G, double *eTG, double *s_store,
    double *r_store, double *sca, double *icemelt, double *snowmelt,
    const double *qobs, double *sse,
    const rr_call_options *opt)
{

    return snow_gr4j_host("rr_cemaneigehystgr4jice_simulate", HYST | ICE, prec,
                          mean_temp, etp, frac_ice, frac_solid_prec, T, L,
                          snow_pack_init, thermal_state_init, sca_init, s_init,
                          r_init, params, N, qsim, G, eTG, s_store, r_store,
                          sca, icemelt, snowmelt, qobs, sse, opt);
}

extern "C" int rr_cemaneigehystgr4jice_simulate(
    const double *prec, const double *mean_temp, const double *etp,
    const double *frac_ice, const double *frac_solid_prec, int64_t T,
    int64_t L, double snow_pack_init, double thermal_state_init,
    double sca_init, double s_init, double r_init, const double *params,
    int64_t N, double *qsim, double *G, double *eTG, double *s_store,
    double *r_store, double *sca, double *icemelt, double *snowmelt,
    const double *qobs, double *sse)
{
    return rr_cemaneigehystgr4jice_simulate_opt(prec, mean_temp, etp, frac_ice, frac_solid_prec, T, L, snow_pack_init, thermal_state_init, sca_init, s_init, r_init, params, N, qsim, G, eTG, s_store, r_store, sca, icemelt, snowmelt, qobs, sse, nullptr);
}

