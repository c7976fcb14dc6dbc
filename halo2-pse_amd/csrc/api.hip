// api.hip -- the extern "C" surface of libhalo2hip.so (include/halo2hip.h), the device
// context, workspace buffers and HIP-event stage timers.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <shared_mutex>
#include <thread>

#include "../../include/halo2hip.h"
#include "../../include/halo2hip_debug.h"
#include "engine.h"

// the optional lazy pin cache (further down, inside the extern "C" block: declared with that linkage here too, g++ insists)
extern "C" {
static void lazy_forget(const void* key);
static void lazy_reset();
static void lazy_pin_consider(const uint64_t* bases_xy, size_t n);
}

namespace h2 {

int gen_scalars_device(uint64_t seed, uint64_t start, size_t n, Fe* d_out, hipStream_t s);
int gen_points_device(uint64_t seed, uint64_t start, size_t n, Affine* d_out, hipStream_t s);
void msm_set_window(uint32_t c);
void msm_set_max_chunk(size_t m);
void msm_set_stream(uint32_t chunks, double ratio, size_t min_n);
size_t msm_debug_ladder(size_t n, uint32_t chunks, double ratio, bool with_bases, size_t* out, size_t cap);
void msm_set_heavy_div(size_t d);
void msm_set_bin_entries(size_t d);
void msm_set_split_records(bool on);
void msm_set_bucket_order(int local);
void msm_set_quad_tail(bool on);
void msm_set_split_buckets(bool on);
void msm_set_plane_tail(bool on);
void ecfft_set_quad(bool on);
void ecfft_set_lazy(bool on);
void msm_set_fuse_limits(size_t entries, size_t max_n);
void msm_set_rowcol(uint64_t lanes, uint32_t flavour);
void ntt_set_smax(uint32_t v);
void ntt_set_two_pass(uint32_t lo, uint32_t hi);
void ntt_set_full_twiddle_budget(uint64_t bytes);
void ntt_set_batch_bytes(uint64_t bytes);
void ntt_set_two_pass_log_j(int v);
void ntt_set_full_max_log_m(uint32_t v);
void ntt_set_fold_tables(bool on);
void ntt_set_two_pass_batch_wgs(uint64_t v);
void msm_set_reserved_cus(uint32_t k);
uint32_t msm_get_reserved_cus();
uint32_t msm_get_window(size_t n);

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int DevBuf::ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) {
        H2_CHECK(hipDeviceSynchronize());  // a previous call's kernels may still read the old block
        H2_CHECK(hipFree(p));
        p = nullptr;
        cap = 0;
    }
    size_t want = bytes + bytes / 8 + 4096;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        p = nullptr;
        set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        return H2HIP_ENOMEM;
    }
    cap = want;
    return 0;
}

int HostBuf::ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) {
        H2_CHECK(hipDeviceSynchronize());
        H2_CHECK(hipHostFree(p));
        p = nullptr;
        cap = 0;
    }
    size_t want = bytes * 2 + 4096;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e != hipSuccess) {
        p = nullptr;
        set_error("hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        return H2HIP_ENOMEM;
    }
    cap = want;
    return 0;
}

void HostBuf::release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
}

void DevBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

// ---- devices -----------------------------------------------------------------------------------------------------
// One Ctx per device of h2hip_init's list.  g_ctx is the primary (device_ids[0]); its mutex serialises the entry
// points.  The other devices only ever run MSM shards (best_multiexp splits its pairs the same way over rayon
// threads, arithmetic.rs:137-153), each from its own worker thread.
// Locking (round 3; one process-wide mutex before): g_engine_mu is held shared by every entry point and exclusively by
// h2hip_init / h2hip_shutdown, so the device list cannot change under a running call.  Each device context has its own
// mutex: a `_device` entry point locks only the context that owns its pointers, so calls on different GPUs -- and concurrent
// best_fft callers spread over devices (plonk/permutation/keygen.rs:216-233) -- overlap.  Entry points that touch every
// device (host-pointer MSMs over several GPUs, pin / unpin, pinned_info) lock all contexts in list order; the lazy-pin
// bookkeeping and the configuration are only touched under the first context's mutex.
static std::shared_mutex g_engine_mu;
static Ctx g_ctx;
static std::vector<Ctx*> g_devs;  // g_devs[0] == &g_ctx once ready
Ctx* ctx() { return &g_ctx; }
int n_devices() { return (int)g_devs.size(); }
Ctx* ctx_at(int i) { return g_devs[(size_t)i]; }

int Ctx::stage_h2d(void* d_dst, const void* h_src, size_t bytes, hipStream_t s) {
    const size_t ring = (size_t)4 << 20;
    if (bytes > ring / 2) {  // too large for the ring: a synchronous copy
        H2_CHECK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, s));
        H2_CHECK(hipStreamSynchronize(s));
        return 0;
    }
    if (!stage.p) {
        int rc = stage.ensure(ring);
        if (rc) return rc;
        stage_off = 0;
    }
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (stage_off + need > ring) {  // wrap: copies queued from the old contents must have left the ring
        H2_CHECK(hipDeviceSynchronize());
        stage_off = 0;
    }
    char* h = (char*)stage.p + stage_off;
    memcpy(h, h_src, bytes);
    stage_off += need;
    H2_CHECK(hipMemcpyAsync(d_dst, h, bytes, hipMemcpyHostToDevice, s));
    return 0;
}

int Ctx::timer_begin(const char* name, hipStream_t s) {
    if (!profiling) return -1;
    StageTimer t;
    t.name = name;
    if (hipEventCreate(&t.e0) != hipSuccess || hipEventCreate(&t.e1) != hipSuccess) return -1;
    (void)hipEventRecord(t.e0, s);
    t.pending = true;
    timers.push_back(t);
    return (int)timers.size() - 1;
}

int Ctx::timer_kernel(const char* name, hipEvent_t* e0, hipEvent_t* e1) {
    if (!profiling_kernel || profiling) return -1;
    StageTimer t;
    t.name = name;
    if (hipEventCreate(&t.e0) != hipSuccess || hipEventCreate(&t.e1) != hipSuccess) return -1;
    t.pending = true;
    timers.push_back(t);
    *e0 = t.e0;
    *e1 = t.e1;
    return (int)timers.size() - 1;
}

void Ctx::timer_end(int id, hipStream_t s) {
    if (id < 0 || id >= (int)timers.size()) return;
    (void)hipEventRecord(timers[id].e1, s);
}

// fold finished event pairs into per-name totals (kept in entries with e0 == nullptr)
void Ctx::timers_collect() {
    std::vector<StageTimer> totals;
    for (auto& t : timers) {
        double ms = t.total_ms;
        uint64_t cnt = t.count;
        if (t.pending) {
            float f = 0.f;
            (void)hipEventSynchronize(t.e1);
            if (hipEventElapsedTime(&f, t.e0, t.e1) == hipSuccess) {
                ms = f;
                cnt = 1;
            }
            (void)hipEventDestroy(t.e0);
            (void)hipEventDestroy(t.e1);
        }
        bool found = false;
        for (auto& u : totals)
            if (u.name == t.name) {
                u.total_ms += ms;
                u.count += cnt;
                found = true;
            }
        if (!found) {
            StageTimer u;
            u.name = t.name;
            u.total_ms = ms;
            u.count = cnt;
            totals.push_back(u);
        }
    }
    timers.swap(totals);
}

int Ctx::ws_acquire(hipStream_t s) {
    if (ws_used && ws_last_stream != s) H2_CHECK(hipStreamWaitEvent(s, ws_event, 0));
    return 0;
}

int Ctx::ws_release(hipStream_t s) {
    if (!ws_event) H2_CHECK(hipEventCreateWithFlags(&ws_event, hipEventDisableTiming));
    H2_CHECK(hipEventRecord(ws_event, s));
    ws_last_stream = s;
    ws_used = true;
    return 0;
}

// aux1 (stage A) and aux2 (stage C) run on a reserved slice of the CUs, aux_b (stage B) on the rest.  The slice
// takes K/16 of every 16 consecutive CU ids, so it is spread over all XCDs whatever the id -> XCD mapping is.
int Ctx::ensure_aux(size_t n_events) {
    const uint32_t K = msm_get_reserved_cus();
    if (aux1 && aux_reserved != K) {
        H2_CHECK(hipDeviceSynchronize());
        (void)hipStreamDestroy(aux1);
        (void)hipStreamDestroy(aux2);
        (void)hipStreamDestroy(aux_b);
        aux1 = aux2 = aux_b = nullptr;
    }
    if (!aux1) {
        const int n_cu = sm_count;
        const uint32_t words = (uint32_t)((n_cu + 31) / 32);
        std::vector<uint32_t> mask_ac(words, 0), mask_b(words, 0);
        const uint32_t per16 = (K * 16 + (uint32_t)n_cu - 1) / (uint32_t)n_cu;  // reserved ids out of every 16
        for (int i = 0; i < n_cu; i++) {
            bool reserved = K > 0 && (uint32_t)(i % 16) >= 16 - per16;
            (reserved ? mask_ac : mask_b)[i / 32] |= 1u << (i % 32);
        }
        if (K == 0 || per16 >= 16) {  // no partition: every stream sees every CU
            for (auto& w : mask_ac) w = 0xffffffffu;
            for (auto& w : mask_b) w = 0xffffffffu;
        }
        H2_CHECK(hipExtStreamCreateWithCUMask(&aux1, words, mask_ac.data()));
        H2_CHECK(hipExtStreamCreateWithCUMask(&aux2, words, mask_ac.data()));
        H2_CHECK(hipExtStreamCreateWithCUMask(&aux_b, words, mask_b.data()));
        aux_reserved = K;
    }
    while (aux_events.size() < n_events) {
        hipEvent_t e;
        H2_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        aux_events.push_back(e);
    }
    return 0;
}

// ---- configuration (environment, read once at init) -----------------------------------------------------
struct Config {
    size_t multi_gpu_min_n = (size_t)1 << 18;  // HALO2_HIP_MULTI_GPU_MIN_N: smaller MSMs stay on the primary device
    size_t msm_min_n = (size_t)1 << 10;        // HALO2_HIP_MSM_MIN_N: below this the Rust shim keeps the CPU body
    uint32_t ntt_min_log_n = 10;               // HALO2_HIP_NTT_MIN_LOGN: likewise for best_fft
    bool gather_rccl = false;                  // HALO2_HIP_GATHER=rccl: the devices' set sums meet through ncclAllGather (default: host fold)
    bool fixed_base = true;                    // HALO2_HIP_FIXED_BASE=0: h2hip_bases_pin keeps the points only
    size_t table_max_bytes = (size_t)160 << 30;  // HALO2_HIP_TABLE_MAX_GB: largest window table built at pin time
    bool allow_dup = false;                    // HALO2_HIP_ALLOW_DUPLICATE_DEVICES=1: rehearsal on a one-GPU box
    bool roctx = false;                        // HALO2_HIP_ROCTX=1: roctx range around every entry point
    uint32_t lazy_pin_after = 0;               // HALO2_HIP_LAZY_PIN=k: a host bases array seen k times unpinned is pinned by the library
    size_t column_cache_bytes = (size_t)64 << 30;  // HALO2_HIP_COLUMN_CACHE_GB: HBM the pinned proving-key columns may take (least recently used dropped first)
};
static Config g_cfg;

static bool env_u64(const char* name, uint64_t* out) {
    const char* v = getenv(name);
    if (!v || !*v) return false;
    char* end = nullptr;
    unsigned long long x = strtoull(v, &end, 0);
    if (end == v) return false;
    *out = x;
    return true;
}

static void read_config() {
    Config c;
    uint64_t v;
    if (env_u64("HALO2_HIP_MULTI_GPU_MIN_N", &v)) c.multi_gpu_min_n = (size_t)v;
    if (env_u64("HALO2_HIP_MSM_MIN_N", &v)) c.msm_min_n = (size_t)v;
    if (env_u64("HALO2_HIP_NTT_MIN_LOGN", &v)) c.ntt_min_log_n = (uint32_t)v;
    if (env_u64("HALO2_HIP_FIXED_BASE", &v)) c.fixed_base = v != 0;
    if (env_u64("HALO2_HIP_TABLE_MAX_GB", &v)) c.table_max_bytes = (size_t)v << 30;
    if (env_u64("HALO2_HIP_ALLOW_DUPLICATE_DEVICES", &v)) c.allow_dup = v != 0;
    if (env_u64("HALO2_HIP_ROCTX", &v)) c.roctx = v != 0;
    if (env_u64("HALO2_HIP_LAZY_PIN", &v)) c.lazy_pin_after = (uint32_t)v;
    if (env_u64("HALO2_HIP_COLUMN_CACHE_GB", &v)) c.column_cache_bytes = (size_t)v << 30;
    if (env_u64("HALO2_HIP_NTT_TWIDDLE_MB", &v)) ntt_set_full_twiddle_budget(v << 20);
    if (env_u64("HALO2_HIP_EVALH_CODEGEN", &v)) evalh_debug_set_codegen((int)v, 0);
    if (env_u64("HALO2_HIP_MSM_WINDOW", &v) && v >= 2 && v <= 24) msm_set_window((uint32_t)v);
    {  // HALO2_HIP_STREAM=0: host-slice MSMs upload whole arrays ahead of the run (no copier thread); HALO2_HIP_STREAM_MIN_N: threshold
        uint64_t on = 1, min_n = 0;
        const bool has_on = env_u64("HALO2_HIP_STREAM", &on), has_min = env_u64("HALO2_HIP_STREAM_MIN_N", &min_n);
        if (has_on || has_min) msm_set_stream(has_on && on == 0 ? 1 : 0, 0.0, has_min ? (size_t)min_n : 0);
    }
    const char* g = getenv("HALO2_HIP_GATHER");
    if (g && !strcmp(g, "rccl")) c.gather_rccl = true;
    g_cfg = c;
}

// ---- roctx ranges (optional; the library is dlopen'ed so that nothing links against the tracer) ------------------------
static int (*g_roctx_push)(const char*) = nullptr;
static int (*g_roctx_pop)() = nullptr;
static void roctx_load() {
    if (g_roctx_push) return;
    void* h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    g_roctx_push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
    g_roctx_pop = (int (*)())dlsym(h, "roctxRangePop");
    if (!g_roctx_push || !g_roctx_pop) g_roctx_push = nullptr, g_roctx_pop = nullptr;
}

// ---- RCCL (dlopen'ed: a single-GPU consumer never needs it) ---------------------------------------------------------------
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::vector<ncclComm_t> comms;  // one per device of the engine, empty when the host gathers
};
static Rccl g_rccl;

static bool rccl_load() {
    Rccl& r = g_rccl;
    if (r.lib) return true;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);  // the copy torch loaded, if any (same SONAME)
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return false;
    r.CommInitAll = (decltype(r.CommInitAll))dlsym(h, "ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
    r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.CommInitAll || !r.CommDestroy || !r.AllGather || !r.GroupStart || !r.GroupEnd || !r.GetErrorString) return false;
    r.lib = h;
    return true;
}

// one communicator per engine device (ncclCommInitAll); needs distinct devices.  0 on success.
static int rccl_comms_create() {
    if (!g_rccl.comms.empty()) return 0;
    std::vector<int> ids;
    for (Ctx* x : g_devs) ids.push_back(x->device);
    for (size_t i = 0; i < ids.size(); i++)
        for (size_t j = 0; j < i; j++)
            if (ids[i] == ids[j]) {
                set_error("RCCL needs distinct devices (device %d is listed twice); partials are gathered on the host", ids[i]);
                return H2HIP_EINVAL;
            }
    if (!rccl_load()) {
        set_error("librccl.so.1 not found; partials are gathered on the host");
        return H2HIP_EDEVICE;
    }
    g_rccl.comms.assign(ids.size(), nullptr);
    ncclResult_t r = g_rccl.CommInitAll(g_rccl.comms.data(), (int)ids.size(), ids.data());
    (void)hipSetDevice(ids[0]);
    if (r != ncclSuccess) {
        set_error("ncclCommInitAll failed (%s); partials are gathered on the host", g_rccl.GetErrorString(r));
        g_rccl.comms.clear();
        return H2HIP_EDEVICE;
    }
    return 0;
}

// ---- worker threads: one per secondary device, alive from init to shutdown -------------------------------------------------
struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> task;
    bool has_task = false, done = false, quit = false;
    int rc = 0;
    std::string err;
};
static std::vector<Worker*> g_workers;  // g_workers[i] serves g_devs[i]; [0] is unused (the caller's thread runs device 0)

static void worker_main(Worker* w, int device) {
    (void)hipSetDevice(device);
    std::unique_lock<std::mutex> lk(w->m);
    for (;;) {
        w->cv.wait(lk, [&] { return w->has_task || w->quit; });
        if (w->quit) return;
        std::function<int()> t = std::move(w->task);
        w->has_task = false;
        lk.unlock();
        g_err[0] = 0;
        int rc = t();
        lk.lock();
        w->rc = rc;
        w->err = g_err;
        w->done = true;
        w->cv.notify_all();
    }
}

static void worker_post(int i, std::function<int()> f) {
    Worker* w = g_workers[(size_t)i];
    std::lock_guard<std::mutex> lk(w->m);
    w->task = std::move(f);
    w->has_task = true;
    w->done = false;
    w->cv.notify_all();
}

static int worker_wait(int i) {
    Worker* w = g_workers[(size_t)i];
    std::unique_lock<std::mutex> lk(w->m);
    w->cv.wait(lk, [&] { return w->done; });
    if (w->rc) set_error("device %d: %s", g_devs[(size_t)i]->device, w->err.c_str());
    return w->rc;
}

// Run f(i) for every engine device i < n_use: device 0 on the calling thread, the others on their workers.
static int on_devices(int n_use, const std::function<int(int)>& f) {
    for (int i = 1; i < n_use; i++) worker_post(i, [&f, i] { return f(i); });
    int rc = 0;
    if (hipSetDevice(g_devs[0]->device) != hipSuccess) {
        set_error("hipSetDevice(%d) failed", g_devs[0]->device);
        rc = H2HIP_EDEVICE;
    } else {
        rc = f(0);
    }
    char first_err[sizeof(g_err)];
    memcpy(first_err, g_err, sizeof(g_err));
    for (int i = 1; i < n_use; i++) {
        int r = worker_wait(i);
        if (r && !rc) {
            rc = r;
            memcpy(first_err, g_err, sizeof(g_err));
        }
    }
    if (rc) memcpy(g_err, first_err, sizeof(g_err));
    return rc;
}

// ---- copier threads (engine.h) ----------------------------------------------------------------------------------------------
struct Copier {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;       // work for the thread
    std::condition_variable done_cv;  // progress for a blocked copier_wait
    std::deque<CopyJob> q;
    hipStream_t stream = nullptr;
    int device = 0;
    bool quit = false;
    size_t popped = 0;  // jobs taken off the queue this session (under m)
    std::atomic<size_t> done{0};
    std::atomic<int> err{0};
    std::atomic<int> waiters{0};
};

static void copier_main(Copier* cp) {
    (void)hipSetDevice(cp->device);
    std::unique_lock<std::mutex> lk(cp->m);
    for (;;) {
        cp->cv.wait(lk, [&] { return !cp->q.empty() || cp->quit; });
        if (cp->quit) return;
        const CopyJob j = cp->q.front();
        cp->q.pop_front();
        cp->popped++;
        hipStream_t st = cp->stream;
        lk.unlock();
        if (!cp->err.load(std::memory_order_relaxed)) {
            hipError_t e = j.gate ? hipStreamWaitEvent(st, j.gate, 0) : hipSuccess;
            if (e == hipSuccess) e = hipMemcpyAsync(j.dst, j.src, j.bytes, j.d2h ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice, st);
            if (e == hipSuccess && j.ev) e = hipEventRecord(j.ev, st);
            if (e != hipSuccess) cp->err.store((int)e);
        }
        cp->done.fetch_add(1);
        lk.lock();
        if (cp->waiters.load()) cp->done_cv.notify_all();
    }
}

bool copier_ready(Ctx* c, bool down) {
    Copier*& cp = c->copier[down ? 1 : 0];
    if (cp) return true;
    try {
        cp = new Copier();
        cp->device = c->device;
        cp->th = std::thread(copier_main, cp);
    } catch (...) {  // no memory / no thread: nothing may unwind across the C ABI; the caller takes its unstreamed path
        if (cp) delete cp;
        cp = nullptr;
        set_error("copier thread for device %d could not be started", c->device);
        return false;
    }
    return true;
}

int copier_begin(Ctx* c, bool down, hipStream_t stream) {
    if (!copier_ready(c, down)) return H2HIP_ENOMEM;
    Copier* cp = c->copier[down ? 1 : 0];
    std::lock_guard<std::mutex> lk(cp->m);
    if (!cp->q.empty()) {  // a session is over only when its jobs are (copier_abort on every error path)
        set_error("copier: previous session still has jobs queued");
        return H2HIP_EDEVICE;
    }
    cp->stream = stream;
    cp->popped = 0;
    cp->done.store(0);
    cp->err.store(0);
    return 0;
}

int copier_push(Ctx* c, bool down, std::vector<CopyJob>& jobs) {
    Copier* cp = c->copier[down ? 1 : 0];
    try {
        std::lock_guard<std::mutex> lk(cp->m);
        for (const CopyJob& j : jobs) cp->q.push_back(j);
        cp->cv.notify_all();
    } catch (...) {
        set_error("copier: out of memory");
        return H2HIP_ENOMEM;
    }
    jobs.clear();
    return 0;
}

static void copier_block_until(Copier* cp, size_t n_jobs) {
    // a job is tens to hundreds of microseconds away: spin briefly (the enqueueing thread wants to react within a launch latency),
    // then sleep on the condition variable instead of holding a host core through a 10-ms upload
    for (uint32_t spins = 0; spins < 4000; spins++)
        if (cp->done.load(std::memory_order_acquire) >= n_jobs) return;
    cp->waiters.fetch_add(1);
    {
        std::unique_lock<std::mutex> lk(cp->m);
        cp->done_cv.wait(lk, [&] { return cp->done.load() >= n_jobs; });
    }
    cp->waiters.fetch_sub(1);
}

int copier_wait(Ctx* c, bool down, size_t n_jobs) {
    Copier* cp = c->copier[down ? 1 : 0];
    copier_block_until(cp, n_jobs);
    if (int e = cp->err.load()) {
        set_error("%s copy of a streamed piece failed: %s", down ? "device-to-host" : "host-to-device", hipGetErrorString((hipError_t)e));
        return H2HIP_EDEVICE;
    }
    return 0;
}

void copier_abort(Ctx* c, bool down, size_t n_jobs) {
    Copier* cp = c->copier[down ? 1 : 0];
    if (!cp || cp->done.load(std::memory_order_acquire) >= n_jobs) return;
    {
        std::lock_guard<std::mutex> lk(cp->m);
        if (cp->q.empty() && cp->done.load() == cp->popped) return;  // idle already
    }
    if (!cp->err.load()) cp->err.store((int)hipErrorUnknown);  // the jobs not yet started are skipped (and still counted)
    for (;;) {  // idle = nothing queued and every popped job counted: the thread no longer touches the caller's memory or events
        {
            std::lock_guard<std::mutex> lk(cp->m);
            if (cp->q.empty() && cp->done.load() == cp->popped) return;
        }
        std::this_thread::yield();
    }
}

void copier_stop(Ctx* c) {
    for (int k = 0; k < 2; k++) {
        Copier* cp = c->copier[k];
        if (!cp) continue;
        {
            std::lock_guard<std::mutex> lk(cp->m);
            cp->quit = true;
            cp->cv.notify_all();
        }
        cp->th.join();
        delete cp;
        c->copier[k] = nullptr;
    }
}

static bool unpin_everywhere(const void* key);
static bool pinned_validate(const uint64_t* bases_xy, size_t n);

static int init_one(Ctx* c, int dev) {
    H2_CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    H2_CHECK(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library carries gfx950 code objects only", dev, prop.gcnArchName);
        return H2HIP_EDEVICE;
    }
    c->device = dev;
    c->sm_count = prop.multiProcessorCount;
    H2_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->ready = true;
    return 0;
}

static void release_ctx(Ctx* c);

static std::atomic<int> g_init_failed{0};  // an attempt found no usable GPU; lazy initialisation does not try again (ensure_init)

static int do_init(const int* device_ids, int n_ids) {
    Ctx* c = ctx();
    std::unique_lock<std::shared_mutex> lk(g_engine_mu);
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        set_error("no usable HIP device (%s); libhalo2hip has no CPU fallback", e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        g_init_failed.store(1, std::memory_order_release);
        return H2HIP_EDEVICE;
    }
    g_init_failed.store(0, std::memory_order_release);
    if (!c->ready) read_config();  // a running engine keeps the configuration it started with: entry points read it under the shared lock
    // the device list: the caller's, else HALO2_HIP_DEVICES ("0,1,2,3"), else the calling thread's current device
    std::vector<int> ids;
    if (device_ids && n_ids > 0) {
        ids.assign(device_ids, device_ids + n_ids);
    } else if (const char* v = getenv("HALO2_HIP_DEVICES")) {
        for (const char* q = v; *q;) {
            char* end = nullptr;
            long d = strtol(q, &end, 10);
            if (end == q) break;
            ids.push_back((int)d);
            q = *end == ',' ? end + 1 : end;
            if (*end && *end != ',') break;
        }
    }
    if (ids.empty()) {
        int dev = 0;
        H2_CHECK(hipGetDevice(&dev));
        ids.push_back(dev);
    }
    if (c->ready) {  // idempotent for the same list; a different one needs h2hip_shutdown first
        bool same = ids.size() == g_devs.size();
        for (size_t i = 0; same && i < ids.size(); i++) same = g_devs[i]->device == ids[i];
        if (same || !(device_ids && n_ids > 0)) return 0;
        set_error("h2hip_init: already initialised with a different device list; call h2hip_shutdown first");
        return H2HIP_EINVAL;
    }
    for (size_t i = 0; i < ids.size(); i++) {
        if (ids[i] < 0 || ids[i] >= count) {
            set_error("device id %d out of range (0..%d)", ids[i], count - 1);
            return H2HIP_EINVAL;
        }
        for (size_t j = 0; j < i; j++)
            if (ids[j] == ids[i] && !g_cfg.allow_dup) {
                set_error("device id %d listed twice (HALO2_HIP_ALLOW_DUPLICATE_DEVICES=1 allows it for rehearsals)", ids[i]);
                return H2HIP_EINVAL;
            }
    }
    if (g_cfg.roctx) roctx_load();
    g_devs.clear();
    g_devs.push_back(c);
    int rc = init_one(c, ids[0]);
    for (size_t i = 1; !rc && i < ids.size(); i++) {
        Ctx* x = new Ctx();
        g_devs.push_back(x);
        rc = init_one(x, ids[i]);
    }
    if (rc) {
        for (size_t i = 0; i < g_devs.size(); i++) {
            release_ctx(g_devs[i]);
            if (i) delete g_devs[i];
        }
        g_devs.clear();
        return rc;
    }
    (void)hipSetDevice(ids[0]);
    // secondary devices: worker threads, and RCCL communicators for the gather of the partials (distinct devices only)
    g_workers.assign(ids.size(), nullptr);
    for (size_t i = 1; i < ids.size(); i++) {
        g_workers[i] = new Worker();
        g_workers[i]->th = std::thread(worker_main, g_workers[i], ids[i]);
    }
    if (ids.size() > 1 && g_cfg.gather_rccl) (void)rccl_comms_create();  // not fatal: the host gathers the partials instead
    return 0;
}

int ensure_init() {
    if (g_init_failed.load(std::memory_order_acquire)) {  // no usable GPU, found out once: the CPU-fallback path takes no lock and makes no HIP call
        set_error("no usable HIP device (cached from the first attempt; h2hip_init or h2hip_shutdown retries); libhalo2hip has no CPU fallback");
        return H2HIP_EDEVICE;
    }
    {
        std::shared_lock<std::shared_mutex> lk(g_engine_mu);
        if (ctx()->ready) return 0;
    }
    return do_init(nullptr, 0);  // takes the engine lock exclusively; a second thread arriving here finds the engine ready
}

static inline Fe fe_from_u64x4(const uint64_t v[4]) {
    Fe o;
    memcpy(o.l, v, 32);
    return o;
}

static inline void xyzz_to_out(const XYZZ& r, uint64_t out_xyz[12]) {
    Jac j = xyzz_to_jac(r);
    memcpy(out_xyz, &j, 96);
}

static int check_fr(const uint64_t v[4], const char* what) {
    Fe f = fe_from_u64x4(v);
    if (!fe_is_canonical<FrP>(f)) {
        set_error("%s is not a reduced Fr element", what);
        return H2HIP_EINVAL;
    }
    return 0;
}

struct Entry {
    Ctx* c;
    std::shared_lock<std::shared_mutex> engine;
    std::vector<std::unique_lock<std::recursive_mutex>> held;  // the contexts this call owns, in list order
    int rc;
    bool ranged = false;
    // d_ptr: a device pointer of the call, or nullptr; with several devices the call runs on the device that owns it.
    // all_devices: the call reads or changes every device's state (pinned caches, multi-device MSM).
    explicit Entry(const char* name = nullptr, const void* d_ptr = nullptr, bool all_devices = false) : c(ctx()), rc(0) {
        for (;;) {
            rc = ensure_init();
            if (rc) return;
            engine = std::shared_lock<std::shared_mutex>(g_engine_mu);
            if (ctx()->ready) break;
            engine.unlock();  // shut down by another thread in between: initialise again
        }
        if (d_ptr && g_devs.size() > 1) {
            hipPointerAttribute_t at;
            if (hipPointerGetAttributes(&at, d_ptr) == hipSuccess) {
                Ctx* owner = nullptr;
                for (Ctx* x : g_devs)
                    if (x->device == at.device) {
                        owner = x;
                        break;
                    }
                if (!owner) {
                    set_error("device pointer belongs to device %d, which is not in h2hip_init's list", at.device);
                    rc = H2HIP_EINVAL;
                    return;
                }
                c = owner;
            } else {
                (void)hipGetLastError();
            }
        }
        if (all_devices) {
            for (Ctx* x : g_devs) held.emplace_back(x->mu);
        } else {
            held.emplace_back(c->mu);
        }
        if (hipSetDevice(c->device) != hipSuccess) {
            set_error("hipSetDevice(%d) failed", c->device);
            rc = H2HIP_EDEVICE;
            return;
        }
        if (name && g_roctx_push) {
            g_roctx_push(name);
            ranged = true;
        }
    }
    ~Entry() {
        if (ranged) g_roctx_pop();
    }
};

static int ntt_host(uint64_t* a, const Fe& omega, uint32_t log_n, const NttScale* sc, const uint64_t* src, size_t src_elems) {
    Entry en("h2hip_ntt_host");
    if (en.rc) return en.rc;
    Ctx* c = en.c;
    size_t bytes = sizeof(Fe) << log_n;
    int rc = c->ntt_io.ensure(bytes);
    if (rc) return rc;
    const void* from = src ? (const void*)src : (const void*)a;
    size_t in_bytes = src ? src_elems * sizeof(Fe) : bytes;
    H2_CHECK(hipMemcpyAsync(c->ntt_io.p, from, in_bytes, hipMemcpyHostToDevice, c->stream));
    rc = ntt_device(c, (Fe*)c->ntt_io.p, omega, log_n, sc, c->stream);
    if (rc) return rc;
    H2_CHECK(hipMemcpyAsync(a, c->ntt_io.p, bytes, hipMemcpyDeviceToHost, c->stream));
    H2_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

// ---- host-pointer batched transforms: a three-stage pipeline over the columns ----------------------------------------------------
// create_proof converts its columns back to back from `Vec<F>`s (plonk/prover.rs:487 lagrange_to_coeff per advice column,
// plonk/evaluation.rs:306-323 coeff_to_extended per advice / instance column).  One call each means upload -> transform ->
// download in series, 11.6 x the device-resident transform at 2^22 (round 3).  Here column i + 1 crosses PCIe upwards and column
// i - 1 downwards (full duplex, one copier thread per direction: pageable hipMemcpyAsync blocks its caller) while column i is
// transformed on the engine's stream: per column max(upload, transform, download) instead of their sum.
static size_t g_ntt_host_batch_bytes = (size_t)4 << 30;  // device memory one pipelined run may hold; larger batches are cut into runs
static size_t g_ntt_host_group_bytes = (size_t)2 << 20;  // columns smaller than this are grouped per pipeline step

struct PipeDrain {  // every exit: the copiers idle, their streams drained -- nothing touches the caller's columns after the call
    Ctx* c;
    hipStream_t su, sd;
    ~PipeDrain() {
        copier_abort(c, false, SIZE_MAX);
        copier_abort(c, true, SIZE_MAX);
        (void)hipStreamSynchronize(su);
        (void)hipStreamSynchronize(sd);
    }
};

// m columns: in[i] (in_elems elements) -> out[i] (2^log_n elements); in[i] == out[i] is fine (every column is read before it is written)
static int ntt_host_pipeline(Ctx* c, const uint64_t* const* in, size_t in_elems, uint64_t* const* out, size_t m, const Fe& omega, uint32_t log_n,
                             const NttScale* sc) {
    const size_t col = sizeof(Fe) << log_n, in_bytes = in_elems * sizeof(Fe);
    size_t gc = g_ntt_host_group_bytes / col;
    if (gc < 1) gc = 1;
    const size_t G = (m + gc - 1) / gc;
    int rc;
    if ((rc = c->ntt_io.ensure(m * col))) return rc;
    if ((rc = c->ensure_aux(2 * G + 2))) return rc;
    hipStream_t s = c->stream, su = c->aux2, sd = c->aux1;
    hipEvent_t* ev = c->aux_events.data();  // [2g] group g uploaded, [2g + 1] group g transformed
    char* base = (char*)c->ntt_io.p;
    H2_CHECK(hipStreamSynchronize(s));
    if ((rc = copier_begin(c, false, su))) return rc;
    if ((rc = copier_begin(c, true, sd))) return rc;
    PipeDrain drain{c, su, sd};
    std::vector<CopyJob> jobs;
    std::vector<Fe*> cols(m);
    for (size_t i = 0; i < m; i++) {
        cols[i] = (Fe*)(base + i * col);
        jobs.push_back(CopyJob{cols[i], in[i], in_bytes, ((i + 1) % gc == 0 || i + 1 == m) ? ev[2 * (i / gc)] : nullptr, nullptr, false});
    }
    if ((rc = copier_push(c, false, jobs))) return rc;
    for (size_t g = 0; g < G; g++) {
        const size_t i0 = g * gc, cnt = m - i0 < gc ? m - i0 : gc;
        if ((rc = copier_wait(c, false, i0 + cnt))) return rc;
        H2_CHECK(hipStreamWaitEvent(s, ev[2 * g], 0));
        rc = cnt == 1 ? ntt_device(c, cols[i0], omega, log_n, sc, s) : ntt_device_batch(c, cols.data() + i0, nullptr, cnt, omega, log_n, sc, s);
        if (rc) return rc;
        H2_CHECK(hipEventRecord(ev[2 * g + 1], s));
        for (size_t i = i0; i < i0 + cnt; i++) jobs.push_back(CopyJob{out[i], cols[i], col, nullptr, i == i0 ? ev[2 * g + 1] : nullptr, true});
        if ((rc = copier_push(c, true, jobs))) return rc;
    }
    if ((rc = copier_wait(c, true, m))) return rc;
    H2_CHECK(hipStreamSynchronize(sd));
    H2_CHECK(hipStreamSynchronize(s));
    return 0;
}

static int ntt_host_batch_on(Ctx* c, const uint64_t* const* in, size_t in_elems, uint64_t* const* out, size_t count, const Fe& omega, uint32_t log_n,
                             const NttScale* sc) {
    const size_t col = sizeof(Fe) << log_n;
    size_t per = g_ntt_host_batch_bytes / col;
    if (per < 2) per = 2;
    const bool threads = count > 1 && copier_ready(c, false) && copier_ready(c, true);
    for (size_t i0 = 0; i0 < count; i0 += per) {
        const size_t m = count - i0 < per ? count - i0 : per;
        if (m > 1 && threads) {
            int rc = ntt_host_pipeline(c, in + i0, in_elems, out + i0, m, omega, log_n, sc);
            if (rc) return rc;
            continue;
        }
        for (size_t i = i0; i < i0 + m; i++) {  // a lone column (or no copier threads): upload, transform, download in order
            int rc = c->ntt_io.ensure(col);
            if (rc) return rc;
            H2_CHECK(hipMemcpyAsync(c->ntt_io.p, in[i], in_elems * sizeof(Fe), hipMemcpyHostToDevice, c->stream));
            if ((rc = ntt_device(c, (Fe*)c->ntt_io.p, omega, log_n, sc, c->stream))) return rc;
            H2_CHECK(hipMemcpyAsync(out[i], c->ntt_io.p, col, hipMemcpyDeviceToHost, c->stream));
            H2_CHECK(hipStreamSynchronize(c->stream));
        }
    }
    return 0;
}

// NTT replicas (second bullet): with several devices the columns are dealt in contiguous shares, every device
// runs the pipeline on its share over its own PCIe link; no column crosses xGMI.
static int ntt_host_batch(const char* name, const uint64_t* const* in, size_t in_elems, uint64_t* const* out, size_t count, const Fe& omega,
                          uint32_t log_n, const NttScale* sc) {
    if (!count) return 0;
    bool multi;
    {
        Entry en(name);
        if (en.rc) return en.rc;
        multi = g_devs.size() > 1 && count > 1;
        if (!multi) return ntt_host_batch_on(en.c, in, in_elems, out, count, omega, log_n, sc);
    }
    Entry en(name, nullptr, true);
    if (en.rc) return en.rc;
    const size_t nd = g_devs.size() < count ? g_devs.size() : count;
    return on_devices((int)nd, [&](int d) -> int {
        const size_t lo = count * (size_t)d / nd, hi = count * ((size_t)d + 1) / nd;
        Ctx* x = g_devs[(size_t)d];
        H2_CHECK(hipSetDevice(x->device));
        return ntt_host_batch_on(x, in + lo, in_elems, out + lo, hi - lo, omega, log_n, sc);
    });
}

static void make_zeta_scale(NttScale* sc, bool into_coset, const uint64_t g_coset[4], const uint64_t g_coset_inv[4], const Fe* divisor) {
    // distribute_powers_zeta (poly/domain.rs:335-351): a[i] *= [1, c0, c1][i % 3],
    // (c0, c1) = (g_coset, g_coset_inv) into the coset, swapped on the way out
    Fe c0 = fe_from_u64x4(into_coset ? g_coset : g_coset_inv);
    Fe c1 = fe_from_u64x4(into_coset ? g_coset_inv : g_coset);
    if (into_coset) {
        sc->in_scale = true;
        sc->in3[0] = fe_one<FrP>();
        sc->in3[1] = c0;
        sc->in3[2] = c1;
    } else {
        sc->out_scale = true;
        sc->out3[0] = *divisor;
        sc->out3[1] = fe_mul<FrP>(*divisor, c0);
        sc->out3[2] = fe_mul<FrP>(*divisor, c1);
    }
}


static void release_ctx(Ctx* c) {
    copier_stop(c);
    if (c->device >= 0) (void)hipSetDevice(c->device);
    if (c->ready) (void)hipDeviceSynchronize();
    c->timers_collect();
    c->timers.clear();
    ntt_twiddles_free(c);
    evalh_modules_free(c);
    for (auto& kv : c->pinned) {
        (void)hipFree(kv.second.d);
        if (kv.second.d_sample) (void)hipFree(kv.second.d_sample);
    }
    c->pinned.clear();
    for (auto& kv : c->pinned_cols) (void)hipFree(kv.second.d);
    c->pinned_cols.clear();
    c->pinned_cols_bytes = 0;
    c->pin_flag.release();
    c->ntt_ws.release();
    c->ntt_io.release();
    for (int k = 0; k < 3; k++) {
        c->msm_scalars[k].release();
        c->msm_slot[k].release();
    }
    c->msm_bases.release();
    c->host_ws.release();
    c->host_planes.release();
    c->stage.release();
    c->stage_off = 0;
    c->misc.release();
    c->evalh_ws.release();
    c->evalh_slots.release();
    c->ecfft_ws.release();
    c->ntt_ptrs.release();
    c->gather.release();
    c->gen_table.release();
    if (c->stream) (void)hipStreamDestroy(c->stream);
    c->stream = nullptr;
    for (auto e : c->aux_events) (void)hipEventDestroy(e);
    c->aux_events.clear();
    if (c->aux1) (void)hipStreamDestroy(c->aux1);
    if (c->aux2) (void)hipStreamDestroy(c->aux2);
    if (c->aux_b) (void)hipStreamDestroy(c->aux_b);
    c->aux1 = c->aux2 = c->aux_b = nullptr;
    if (c->ws_event) (void)hipEventDestroy(c->ws_event);
    c->ws_event = nullptr;
    c->ws_used = false;
    c->ready = false;
}

// drop `key` from every device's pinned cache; true when it was there
static bool unpin_everywhere(const void* key) {
    bool found = false;
    for (Ctx* x : g_devs) {
        auto it = x->pinned.find(key);
        if (it == x->pinned.end()) continue;
        found = true;
        (void)hipSetDevice(x->device);
        (void)hipDeviceSynchronize();
        (void)hipFree(it->second.d);
        if (it->second.d_sample) (void)hipFree(it->second.d_sample);
        x->pinned.erase(it);
    }
    if (!g_devs.empty()) (void)hipSetDevice(g_devs[0]->device);
    return found;
}

// A host-keyed pinned entry serves this call only if n fits and the caller's array still carries the sampled points:
// a mismatch means the allocation was freed and reused -- the entry is dropped on every device.  A prefix so short that
// fewer than four samples fall inside it (n <= total >> 12) is not vouched for by the fingerprint (point 0 of every KZG
// `g` is the generator): such a call bypasses the entry (returns false, entry kept) and uploads its few points.
static bool pinned_validate(const uint64_t* bases_xy, size_t n) {
    Ctx* c = ctx();
    auto it = c->pinned.find((const void*)bases_xy);
    if (it == c->pinned.end() || it->second.device_key) return true;
    size_t total = 0;
    for (Ctx* x : g_devs) {
        auto jt = x->pinned.find((const void*)bases_xy);
        if (jt != x->pinned.end() && jt->second.hi > total) total = jt->second.hi;
    }
    bool ok = n <= total;
    if (ok) {
        for (uint32_t k = 0; ok && k < H2_PIN_SAMPLES; k++) {
            const size_t i = pin_sample_index(total, k);
            if (i < n) ok = memcmp(it->second.sample + 64 * k, bases_xy + 8 * i, 64) == 0;
        }
    }
    if (!ok) {
        ::lazy_forget((const void*)bases_xy);
        (void)unpin_everywhere((const void*)bases_xy);
        return true;  // nothing left to bypass
    }
    return n == total || n > pin_sample_index(total, 3);
}

// ---- device-pointer keys (h2hip_bases_pin_device) ---------------------------------------------------------------------------
// The key is a device address; torch's caching allocator hands the same address out again readily, so a caller that freed a
// pinned buffer without unpinning it would get the OLD table's commitments.  Guard: the entry keeps H2_PIN_SAMPLES points of
// the array in device memory; the first workgroup of the MSM's first kernel (msm_l1_count_kernel) compares them with the caller's buffer and writes its
// verdict to a pinned host word, which is read when the MSM's own result has arrived (the call waits for that anyway: no extra
// synchronisation).  A mismatch drops the entry and the MSM is run again in the plain form over the caller's points.
__global__ void pin_sample_kernel(const Affine* __restrict__ src, size_t total, Affine* __restrict__ out) {
    if (threadIdx.x < H2_PIN_SAMPLES) out[threadIdx.x] = src[pin_sample_index(total, threadIdx.x)];
}

// window table of a device-pointer key on context c, if n fits; *entry gets the cache entry
static const MsmTable* pinned_table(Ctx* c, const void* key, size_t n, MsmTable* out, const PinnedBases** entry) {
    auto it = c->pinned.find(key);
    if (it == c->pinned.end() || !it->second.device_key || n > it->second.n || !it->second.c) return nullptr;
    out->table = (const Affine*)it->second.d;
    out->stride = it->second.n;
    out->c = it->second.c;
    out->W = it->second.W;
    *entry = &it->second;
    return out;
}

// `count` MSMs over device-resident scalars and bases on context c: the fixed-base form when d_bases is a pinned key whose
// fingerprint still matches, else the plain form.  Synchronous (the sums come back through the host).
static int msm_device_keyed(Ctx* c, const Fe* const* d_scalars, const Affine* d_bases, size_t n, size_t count, XYZZ* out, hipStream_t s) {
    MsmTable tab;
    const PinnedBases* pb = nullptr;
    const MsmTable* t = (n && count) ? pinned_table(c, d_bases, n, &tab, &pb) : nullptr;
    if (t && n != pb->n && n <= pin_sample_index(pb->n, 3)) t = nullptr;  // too short a prefix for the fingerprint to vouch for: plain form
    volatile uint32_t* flag = nullptr;
    if (t && pb->d_sample) {
        int rc = c->pin_flag.ensure(64);
        if (rc) return rc;
        flag = (volatile uint32_t*)c->pin_flag.p;
        *flag = 0;
        c->pin_chk.bases = d_bases;
        c->pin_chk.samples = (const Affine*)pb->d_sample;
        c->pin_chk.flag = (uint32_t*)c->pin_flag.p;
        c->pin_chk.n = n;
        c->pin_chk.total = pb->n;
    }
    int rc = msm_batch_device(c, d_scalars, false, d_bases, n, count, out, s, t);
    c->pin_chk = Ctx::PinCheck();
    if (rc || !flag || !*flag) return rc;
    // stale: another array lives at the pinned address now
    H2_CHECK(hipDeviceSynchronize());
    auto it = c->pinned.find((const void*)d_bases);
    if (it != c->pinned.end()) {
        (void)hipFree(it->second.d);
        if (it->second.d_sample) (void)hipFree(it->second.d_sample);
        c->pinned.erase(it);
    }
    return msm_batch_device(c, d_scalars, false, d_bases, n, count, out, s, nullptr);
}

// (pinned proving-key columns: see h2hip_columns_pin further down)
static void column_sample(const uint64_t* h_col, size_t elems, uint8_t* out) {
    for (uint32_t k = 0; k < H2_COL_SAMPLES; k++) memcpy(out + 32 * k, h_col + 4 * pin_sample_index(elems, k), 32);
}

const Fe* pinned_column_lookup(Ctx* c, const uint64_t* h_col, size_t elems) {
    auto it = c->pinned_cols.find(h_col);
    if (it == c->pinned_cols.end()) return nullptr;
    PinnedColumn& pc = it->second;
    uint8_t now[H2_COL_SAMPLES * 32];
    if (pc.elems == elems) column_sample(h_col, elems, now);
    if (pc.elems != elems || memcmp(now, pc.sample, sizeof(now)) != 0) {  // another array lives at this address now: forget the copy
        (void)hipDeviceSynchronize();
        (void)hipFree(pc.d);
        c->pinned_cols_bytes -= pc.elems * sizeof(Fe);
        c->pinned_cols.erase(it);
        return nullptr;
    }
    pc.last_use = ++c->pinned_cols_tick;
    return (const Fe*)pc.d;
}

}  // namespace h2

using namespace h2;

extern "C" {

int h2hip_init(const int* device_ids, int n_devices) { return do_init(device_ids, n_devices); }

void h2hip_shutdown(void) {
    Ctx* c = ctx();
    std::unique_lock<std::shared_mutex> lk(g_engine_mu);  // waits for every running entry point
    g_init_failed.store(0, std::memory_order_release);     // the next call looks for a GPU again
    if (!c->ready) return;
    for (size_t i = 1; i < g_workers.size(); i++) {
        Worker* w = g_workers[i];
        {
            std::lock_guard<std::mutex> wl(w->m);
            w->quit = true;
            w->cv.notify_all();
        }
        w->th.join();
        delete w;
    }
    g_workers.clear();
    for (ncclComm_t cm : g_rccl.comms)
        if (cm) (void)g_rccl.CommDestroy(cm);
    g_rccl.comms.clear();
    for (size_t i = g_devs.size(); i-- > 0;) {
        release_ctx(g_devs[i]);
        if (i) delete g_devs[i];
    }
    g_devs.clear();
    lazy_reset();
    evalh_rtc_shutdown();
}

const char* h2hip_last_error(void) { return g_err; }
const char* h2hip_version(void) { return "halo2hip 0.2 (gfx950)"; }

int h2hip_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return -1;
    return count;
}

int h2hip_num_devices(void) {
    std::shared_lock<std::shared_mutex> lk(g_engine_mu);
    return ctx()->ready ? (int)g_devs.size() : 0;
}

// The shim asks these on every best_multiexp / best_fft, from rayon workers: the configuration is read under the engine lock
// (h2hip_init may be rewriting it), and without a usable GPU they answer "never" from the cached failure -- no lock, no HIP call.
size_t h2hip_msm_min_n(void) {
    if (ensure_init()) return SIZE_MAX;
    std::shared_lock<std::shared_mutex> lk(g_engine_mu);
    return g_cfg.msm_min_n;
}

uint32_t h2hip_ntt_min_log_n(void) {
    if (ensure_init()) return UINT32_MAX;
    std::shared_lock<std::shared_mutex> lk(g_engine_mu);
    return g_cfg.ntt_min_log_n;
}

int h2hip_msm_bn254_device(const void* d_scalars, const void* d_bases_xy, size_t n, uint64_t out_xyz[12], void* stream) {
    if (!out_xyz || (n && (!d_scalars || !d_bases_xy))) {
        set_error("msm: null argument");
        return H2HIP_EINVAL;
    }
    Entry en("h2hip_msm_bn254_device", d_scalars);
    if (en.rc) return en.rc;
    hipStream_t s = (hipStream_t)stream;
    XYZZ r;
    const Fe* sc = (const Fe*)d_scalars;
    int rc = msm_device_keyed(en.c, &sc, (const Affine*)d_bases_xy, n, 1, &r, s);
    if (rc) return rc;
    xyzz_to_out(r, out_xyz);
    return 0;
}

// `count` MSMs of n pairs with host-resident scalars on device context c: the bases come from c's pinned copy of
// key[lo .. lo + n) when there is one (with its window table), else they are uploaded.
static int msm_shard_host(Ctx* c, const uint64_t* const* scalars, const uint64_t* bases_xy, size_t lo, size_t n, size_t count, XYZZ* out,
                          bool use_cache = true) {
    std::vector<const Fe*> sc(count);
    for (size_t j = 0; j < count; j++) sc[j] = (const Fe*)scalars[j] + lo;
    MsmTable tab;
    const Affine *d_bases = nullptr, *h_bases = nullptr;
    const MsmTable* t = nullptr;
    auto it = c->pinned.find((const void*)bases_xy);
    if (use_cache && it != c->pinned.end() && it->second.lo <= lo && lo + n <= it->second.hi) {
        const PinnedBases& pb = it->second;
        d_bases = (const Affine*)pb.d + (lo - pb.lo);
        if (pb.c) {
            tab.table = d_bases;
            tab.stride = pb.n;
            tab.c = pb.c;
            tab.W = pb.W;
            t = &tab;
        }
    } else {
        h_bases = (const Affine*)bases_xy + lo;  // unpinned: the points cross PCIe inside the run, chunk by chunk when it streams
    }
    return msm_batch_device(c, sc.data(), true, d_bases, n, count, out, c->stream, t, h_bases);
}

// The partials of the devices meet on the host: every device's run has already returned its sums through pinned memory (the
// call is synchronous), so the left fold of arithmetic.rs:153 is n_use - 1 additions on numbers the calling thread holds.
static void fold_host(int n_use, size_t count, const std::vector<std::vector<XYZZ>>& parts, uint64_t* out_xyz) {
    for (size_t j = 0; j < count; j++) {
        XYZZ acc = xyzz_identity();
        for (int d = 0; d < n_use; d++) xyzz_add(acc, parts[(size_t)d][j]);
        xyzz_to_out(acc, out_xyz + 12 * j);
    }
}

// HALO2_HIP_GATHER=rccl (fixed-base form only: one set sum per MSM): the sums stage C left in every device's `gather` buffer
// (Ctx::gather_want) are all-gathered device to device over xGMI -- 128 B per (device, MSM), ncclUint8, one group call --
// copied down once from the first device and folded on the host.  all[d * count + j] = sum of MSM j on device d.
// Any failure leaves `all` untouched; the caller then folds the host copies instead.
static int gather_rccl(size_t count, std::vector<XYZZ>* all) {
    const int nd = (int)g_devs.size();
    const size_t bytes = count * sizeof(XYZZ);
    if (g_rccl.comms.size() != (size_t)nd || bytes > H2_GATHER_OWN) return H2HIP_EINVAL;
    ncclResult_t nr = g_rccl.GroupStart();
    for (int d = 0; nr == ncclSuccess && d < nd; d++) {
        Ctx* x = g_devs[(size_t)d];
        nr = g_rccl.AllGather(x->gather.p, (char*)x->gather.p + H2_GATHER_OWN, bytes, ncclUint8, g_rccl.comms[(size_t)d], x->stream);
    }
    ncclResult_t ne = g_rccl.GroupEnd();
    if (nr == ncclSuccess) nr = ne;
    if (nr != ncclSuccess) {
        set_error("ncclAllGather failed: %s", g_rccl.GetErrorString(nr));
        return H2HIP_EDEVICE;
    }
    std::vector<XYZZ> got((size_t)nd * count);
    Ctx* c0 = g_devs[0];
    H2_CHECK(hipSetDevice(c0->device));
    H2_CHECK(hipMemcpyAsync(got.data(), (char*)c0->gather.p + H2_GATHER_OWN, bytes * (size_t)nd, hipMemcpyDeviceToHost, c0->stream));
    H2_CHECK(hipStreamSynchronize(c0->stream));
    for (int d = 1; d < nd; d++) {  // every rank's collective has drained before the buffers are reused
        H2_CHECK(hipSetDevice(g_devs[(size_t)d]->device));
        H2_CHECK(hipStreamSynchronize(g_devs[(size_t)d]->stream));
    }
    H2_CHECK(hipSetDevice(c0->device));
    all->swap(got);
    return 0;
}

static int msm_host_common(const uint64_t* const* scalars, const uint64_t* bases_xy, size_t n, size_t count, uint64_t* out_xyz) {
    Entry en("h2hip_msm_bn254", nullptr, true);
    if (en.rc) return en.rc;
    Ctx* c = en.c;
    if (n == 0 || count == 0) {
        for (size_t j = 0; j < count; j++) xyzz_to_out(xyzz_identity(), out_xyz + 12 * j);
        return 0;
    }
    // a pinned entry whose fingerprint no longer matches the caller's array is stale (the allocation was freed and
    // reused): drop it on every device and fall back to uploading
    const bool use_cache = pinned_validate(bases_xy, n);
    lazy_pin_consider(bases_xy, n);
    const int nd = (int)g_devs.size();
    if (nd == 1 || n < g_cfg.multi_gpu_min_n || !use_cache) {
        std::vector<XYZZ> r(count);
        int rc = msm_shard_host(c, scalars, bases_xy, 0, n, count, r.data(), use_cache);
        if (rc) return rc;
        for (size_t j = 0; j < count; j++) xyzz_to_out(r[j], out_xyz + 12 * j);
        return 0;
    }
    // Several devices: contiguous ranges of pairs per device, as best_multiexp chunks them over threads
    // (arithmetic.rs:137-152).  Pinned bases fix the partition (each device holds its range and its window table).
    std::vector<size_t> lo((size_t)nd), hi((size_t)nd);
    auto it0 = c->pinned.find((const void*)bases_xy);
    if (it0 != c->pinned.end()) {
        for (int d = 0; d < nd; d++) {
            auto it = g_devs[(size_t)d]->pinned.find((const void*)bases_xy);
            size_t l = it != g_devs[(size_t)d]->pinned.end() ? it->second.lo : n, h = it != g_devs[(size_t)d]->pinned.end() ? it->second.hi : n;
            lo[(size_t)d] = l < n ? l : n;
            hi[(size_t)d] = h < n ? h : n;
        }
    } else {
        const size_t per = n / (size_t)nd;
        for (int d = 0; d < nd; d++) {
            lo[(size_t)d] = (size_t)d * per;
            hi[(size_t)d] = d == nd - 1 ? n : (size_t)(d + 1) * per;
        }
    }
    std::vector<std::vector<XYZZ>> parts((size_t)nd, std::vector<XYZZ>(count, xyzz_identity()));
    // RCCL gather only when asked for, communicators exist and every device runs the fixed-base form over its pinned share
    bool use_rccl = g_cfg.gather_rccl && g_rccl.comms.size() == (size_t)nd && count * sizeof(XYZZ) <= H2_GATHER_OWN;
    for (int d = 0; use_rccl && d < nd; d++) {
        auto it = g_devs[(size_t)d]->pinned.find((const void*)bases_xy);
        const bool idle = hi[(size_t)d] == lo[(size_t)d];
        if (!idle && (it == g_devs[(size_t)d]->pinned.end() || !it->second.c)) use_rccl = false;
    }
    int rc = on_devices(nd, [&](int d) -> int {
        Ctx* x = g_devs[(size_t)d];
        const size_t m = hi[(size_t)d] - lo[(size_t)d];
        x->gather_want = false;
        x->gather_off = 0;
        if (use_rccl) {
            int r = x->gather.ensure(H2_GATHER_OWN * (size_t)(nd + 1));
            if (r) return r;
            H2_CHECK(hipMemsetAsync(x->gather.p, 0, count * sizeof(XYZZ), x->stream));  // an idle device contributes identities
            x->gather_want = true;
        }
        int r = m ? msm_shard_host(x, scalars, bases_xy, lo[(size_t)d], m, count, parts[(size_t)d].data()) : 0;
        x->gather_want = false;
        return r;
    });
    if (rc) return rc;
    if (use_rccl) {
        for (int d = 0; d < nd; d++)
            if (hi[(size_t)d] != lo[(size_t)d] && g_devs[(size_t)d]->gather_off != count * sizeof(XYZZ)) use_rccl = false;  // not one run per device
    }
    if (use_rccl) {
        std::vector<XYZZ> all;
        if (gather_rccl(count, &all) == 0) {
            for (int d = 0; d < nd; d++)
                for (size_t j = 0; j < count; j++) parts[(size_t)d][j] = all[(size_t)d * count + j];
        }  // else: the host copies are complete and correct -- fold those
    }
    fold_host(nd, count, parts, out_xyz);
    return 0;
}

int h2hip_msm_bn254(const uint64_t* scalars, const uint64_t* bases_xy, size_t n, uint64_t out_xyz[12]) {
    if (!out_xyz || (n && (!scalars || !bases_xy))) {
        set_error("msm: null argument");
        return H2HIP_EINVAL;
    }
    return msm_host_common(&scalars, bases_xy, n, 1, out_xyz);
}

int h2hip_msm_bn254_batch(const uint64_t* const* scalars, const uint64_t* bases_xy, size_t n, size_t count, uint64_t* out_xyz) {
    if ((count && (!out_xyz || !scalars)) || (n && count && !bases_xy)) {
        set_error("msm_batch: null argument");
        return H2HIP_EINVAL;
    }
    for (size_t j = 0; j < count; j++)
        if (n && !scalars[j]) {
            set_error("msm_batch: scalars[%zu] is null", j);
            return H2HIP_EINVAL;
        }
    return msm_host_common(scalars, bases_xy, n, count, out_xyz);
}

int h2hip_msm_bn254_batch_device(const void* const* d_scalars, const void* d_bases_xy, size_t n, size_t count, uint64_t* out_xyz, void* stream) {
    if ((count && (!out_xyz || !d_scalars)) || (n && count && !d_bases_xy)) {
        set_error("msm_batch: null argument");
        return H2HIP_EINVAL;
    }
    for (size_t j = 0; j < count; j++)
        if (n && !d_scalars[j]) {
            set_error("msm_batch: d_scalars[%zu] is null", j);
            return H2HIP_EINVAL;
        }
    Entry en("h2hip_msm_bn254_batch_device", count ? d_scalars[0] : nullptr);
    if (en.rc) return en.rc;
    std::vector<XYZZ> r(count);
    int rc = msm_device_keyed(en.c, (const Fe* const*)d_scalars, (const Affine*)d_bases_xy, n, count, r.data(), (hipStream_t)stream);
    if (rc) return rc;
    for (size_t j = 0; j < count; j++) xyzz_to_out(r[j], out_xyz + 12 * j);
    return 0;
}

// Pin points [lo, hi) of the caller's array on device context c (worker thread of that device, or the caller's for
// device 0): device copy + window table.  d_src != nullptr: the points already live on this device.
static int pin_on_device(Ctx* c, const void* key, const uint64_t* h_points, const void* d_src, size_t lo, size_t hi, size_t n_total,
                         const uint8_t* sample, bool device_key) {
    (void)n_total;
    auto it = c->pinned.find(key);
    if (it != c->pinned.end()) {
        H2_CHECK(hipDeviceSynchronize());
        (void)hipFree(it->second.d);
        if (it->second.d_sample) (void)hipFree(it->second.d_sample);
        c->pinned.erase(it);
    }
    if (hi <= lo) return 0;
    const size_t n = hi - lo;
    PinnedBases pb;
    pb.n = n;
    pb.lo = lo;
    pb.hi = hi;
    pb.device_key = device_key;
    if (sample) memcpy(pb.sample, sample, sizeof(pb.sample));
    uint32_t cw = 0, W = 1;
    if (g_cfg.fixed_base && n <= ((size_t)1 << 26)) {
        cw = msm_table_window(n);
        W = (255 + cw - 1) / cw;
        if ((size_t)W * n * sizeof(Affine) > g_cfg.table_max_bytes) cw = 0, W = 1;
    }
    hipError_t e = hipMalloc(&pb.d, (size_t)W * n * sizeof(Affine));
    if (e != hipSuccess && cw) {  // no room for the table: keep the points only
        (void)hipGetLastError();
        cw = 0;
        W = 1;
        e = hipMalloc(&pb.d, n * sizeof(Affine));
    }
    if (e != hipSuccess) {
        set_error("bases_pin: hipMalloc(%zu) failed: %s", (size_t)W * n * sizeof(Affine), hipGetErrorString(e));
        return H2HIP_ENOMEM;
    }
    int rc = 0;
    if (d_src) {
        if (!cw) H2_CHECK(hipMemcpyAsync(pb.d, (const Affine*)d_src + lo, n * sizeof(Affine), hipMemcpyDeviceToDevice, c->stream));
    } else {
        H2_CHECK(hipMemcpyAsync(pb.d, (const Affine*)h_points + lo, n * sizeof(Affine), hipMemcpyHostToDevice, c->stream));
    }
    if (cw) rc = msm_table_build(c, d_src ? (const Affine*)d_src + lo : (const Affine*)pb.d, n, cw, (Affine*)pb.d, c->stream);
    if (!rc && device_key) {  // the fingerprint of a device key stays on the device
        if (hipMalloc(&pb.d_sample, H2_PIN_SAMPLES * sizeof(Affine)) != hipSuccess) {
            (void)hipGetLastError();
            pb.d_sample = nullptr;
            set_error("bases_pin: hipMalloc of the fingerprint failed");
            rc = H2HIP_ENOMEM;
        } else {
            hipLaunchKernelGGL(pin_sample_kernel, dim3(1), dim3(64), 0, c->stream, (const Affine*)d_src + lo, n, (Affine*)pb.d_sample);
        }
    }
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) {
        set_error("bases_pin: table build failed");
        rc = H2HIP_EDEVICE;
    }
    if (rc) {
        (void)hipFree(pb.d);
        if (pb.d_sample) (void)hipFree(pb.d_sample);
        return rc;
    }
    pb.c = cw;
    pb.W = cw ? W : 0;
    c->pinned[key] = pb;
    return 0;
}

static void pin_sample(const uint64_t* bases_xy, size_t n, uint8_t* out) {
    for (uint32_t k = 0; k < H2_PIN_SAMPLES; k++) memcpy(out + 64 * k, bases_xy + 8 * pin_sample_index(n, k), 64);
}

// pin a host array on every device (its contiguous share each): the body of h2hip_bases_pin, also reached by the lazy cache
static int pin_host_everywhere(const uint64_t* bases_xy, size_t n) {
    uint8_t sample[H2_PIN_SAMPLES * 64];
    pin_sample(bases_xy, n, sample);
    const int nd = (int)g_devs.size();
    const int use = n >= g_cfg.multi_gpu_min_n ? nd : 1;
    const size_t per = n / (size_t)use;
    int rc = on_devices(nd, [&](int d) -> int {
        size_t lo = d < use ? (size_t)d * per : n, hi = d < use ? (d == use - 1 ? n : (size_t)(d + 1) * per) : n;
        return pin_on_device(g_devs[(size_t)d], (const void*)bases_xy, bases_xy, nullptr, lo, hi, n, sample, false);
    });
    if (rc) (void)unpin_everywhere((const void*)bases_xy);
    return rc;
}

int h2hip_bases_pin(const uint64_t* bases_xy, size_t n) {
    if (!bases_xy || !n) {
        set_error("bases_pin: null/empty");
        return H2HIP_EINVAL;
    }
    Entry en("h2hip_bases_pin", nullptr, true);
    if (en.rc) return en.rc;
    lazy_forget((const void*)bases_xy);  // an explicit pin is the caller's to unpin
    return pin_host_everywhere(bases_xy, n);
}

// ---- optional lazy cache (HALO2_HIP_LAZY_PIN=k, off by default) ----------------------------------------------------------
// The unpatched drop-in calls best_multiexp(coeffs, &params.g[..]) with no handle in the signature and uploads 64 B per
// point every time.  With the knob set, a host array that arrives unpinned for the k-th time with the same address and
// the same sampled points is pinned by the library itself (window table included).  From then on it is an ordinary
// pinned entry: every lookup re-checks the 16 sampled points and a mismatch drops it, so a freed and reused
// allocation costs an upload, not a wrong commitment; what the samples cannot see is a caller that rewrites part of a
// live array in place -- ParamsKZG never does (downsize truncates g and builds a new g_lagrange, kzg/commitment.rs:267-275),
// which is why the knob is opt-in.  At most H2_LAZY_MAX arrays are held this way, the least recently used goes first.
#define H2_LAZY_MAX 4
struct LazySeen {
    const void* key;
    size_t n;
    uint32_t count;
    uint8_t sample[H2_PIN_SAMPLES * 64];
};
static std::vector<LazySeen> g_lazy_seen;        // candidates, at most 8, oldest first
static std::vector<const void*> g_lazy_pinned;   // arrays this cache pinned, least recently used first

static void lazy_reset() {
    g_lazy_seen.clear();
    g_lazy_pinned.clear();
}

static void lazy_forget(const void* key) {
    for (size_t i = 0; i < g_lazy_seen.size(); i++)
        if (g_lazy_seen[i].key == key) {
            g_lazy_seen.erase(g_lazy_seen.begin() + (long)i);
            break;
        }
    for (size_t i = 0; i < g_lazy_pinned.size(); i++)
        if (g_lazy_pinned[i] == key) {
            g_lazy_pinned.erase(g_lazy_pinned.begin() + (long)i);
            break;
        }
}

// called by the host-pointer MSMs after pinned_validate; pins `bases_xy` when it has been seen often enough
static void lazy_pin_consider(const uint64_t* bases_xy, size_t n) {
    if (!g_cfg.lazy_pin_after || n < 1024) return;
    const void* key = (const void*)bases_xy;
    if (ctx()->pinned.count(key)) {  // in use: refresh its place in the LRU order if it is ours
        for (size_t i = 0; i + 1 < g_lazy_pinned.size(); i++)
            if (g_lazy_pinned[i] == key) {
                g_lazy_pinned.erase(g_lazy_pinned.begin() + (long)i);
                g_lazy_pinned.push_back(key);
                break;
            }
        return;
    }
    for (size_t i = 0; i < g_lazy_pinned.size(); i++)  // ours once, dropped since by a failed validation
        if (g_lazy_pinned[i] == key) {
            g_lazy_pinned.erase(g_lazy_pinned.begin() + (long)i);
            break;
        }
    uint8_t sample[H2_PIN_SAMPLES * 64];
    pin_sample(bases_xy, n, sample);
    LazySeen* e = nullptr;
    for (auto& x : g_lazy_seen)
        if (x.key == key) e = &x;
    if (!e) {
        if (g_lazy_seen.size() >= 8) g_lazy_seen.erase(g_lazy_seen.begin());
        g_lazy_seen.push_back(LazySeen());
        e = &g_lazy_seen.back();
        e->key = key;
        e->count = 0;
    }
    if (e->count && (e->n != n || memcmp(e->sample, sample, sizeof(sample)) != 0)) e->count = 0;  // another array at this address
    e->n = n;
    memcpy(e->sample, sample, sizeof(sample));
    if (++e->count < g_cfg.lazy_pin_after) return;
    lazy_forget(key);
    while (g_lazy_pinned.size() >= H2_LAZY_MAX) {
        (void)unpin_everywhere(g_lazy_pinned.front());
        g_lazy_pinned.erase(g_lazy_pinned.begin());
    }
    if (pin_host_everywhere(bases_xy, n) == 0) g_lazy_pinned.push_back(key);  // a failed pin (no room for the table) just leaves the upload path
}

uint32_t h2hip_lazy_pin_after(void) {
    if (ensure_init()) return 0;
    std::shared_lock<std::shared_mutex> lk(g_engine_mu);
    return g_cfg.lazy_pin_after;
}

// test hook: the knob without the environment
int h2hip_debug_set_lazy_pin(uint32_t after) {
    Entry en("h2hip_debug_set_lazy_pin");
    if (en.rc) return en.rc;
    g_cfg.lazy_pin_after = after;
    return 0;
}

int h2hip_bases_pin_device(const void* d_bases_xy, size_t n, void* stream) {
    if (!d_bases_xy || !n) {
        set_error("bases_pin_device: null/empty");
        return H2HIP_EINVAL;
    }
    Entry en("h2hip_bases_pin_device", d_bases_xy);
    if (en.rc) return en.rc;
    H2_CHECK(hipStreamSynchronize((hipStream_t)stream));  // the points may have been produced on the caller's stream
    return pin_on_device(en.c, d_bases_xy, nullptr, d_bases_xy, 0, n, n, nullptr, true);
}

int h2hip_bases_unpin(const void* bases_xy) {
    {
        std::shared_lock<std::shared_mutex> eng(g_engine_mu);
        if (!ctx()->ready) {  // nothing can be pinned; do not start the engine for this (interpreter teardown after h2hip_shutdown)
            set_error("bases_unpin: pointer was not pinned (engine not initialised)");
            return H2HIP_EINVAL;
        }
    }
    Entry en("h2hip_bases_unpin", nullptr, true);
    if (en.rc) return en.rc;
    lazy_forget(bases_xy);
    if (!unpin_everywhere(bases_xy)) {
        set_error("bases_unpin: pointer was not pinned");
        return H2HIP_EINVAL;
    }
    return 0;
}

// test hook: push `count` Jacobian partials per engine device through the library's RCCL gather (communicators are
// created on demand, also for a single device) and fold them; out = fold over the devices of partials_xyz[j].
int h2hip_debug_rccl_gather_selftest(const uint64_t* partials_xyz, size_t count, uint64_t* out_xyz) {
    if (!partials_xyz || !out_xyz || !count) {
        set_error("rccl_gather_selftest: null argument");
        return H2HIP_EINVAL;
    }
    Entry en("h2hip_debug_rccl_gather_selftest", nullptr, true);
    if (en.rc) return en.rc;
    int rc = rccl_comms_create();
    if (rc) return rc;
    const int nd = (int)g_devs.size();
    if (count * sizeof(XYZZ) > H2_GATHER_OWN) {
        set_error("rccl_gather_selftest: at most %zu partials", H2_GATHER_OWN / sizeof(XYZZ));
        return H2HIP_EINVAL;
    }
    std::vector<XYZZ> mine(count);
    for (size_t j = 0; j < count; j++) {
        Jac jc;
        memcpy(&jc, partials_xyz + 12 * j, 96);
        mine[j] = jac_to_xyzz(jc);
    }
    rc = on_devices(nd, [&](int d) -> int {  // stands in for stage C of a run: the device's sums land in its gather buffer
        Ctx* x = g_devs[(size_t)d];
        int r = x->gather.ensure(H2_GATHER_OWN * (size_t)(nd + 1));
        if (r) return r;
        H2_CHECK(hipMemcpyAsync(x->gather.p, mine.data(), count * sizeof(XYZZ), hipMemcpyHostToDevice, x->stream));
        H2_CHECK(hipStreamSynchronize(x->stream));
        return 0;
    });
    if (rc) return rc;
    std::vector<XYZZ> all;
    if ((rc = gather_rccl(count, &all))) return rc;
    std::vector<std::vector<XYZZ>> parts((size_t)nd, std::vector<XYZZ>(count));
    for (int d = 0; d < nd; d++)
        for (size_t j = 0; j < count; j++) parts[(size_t)d][j] = all[(size_t)d * count + j];
    fold_host(nd, count, parts, out_xyz);
    return 0;
}

int h2hip_bases_pinned_info(const void* bases_xy, size_t* n_points, uint32_t* window_bits, uint32_t* windows, size_t* device_bytes) {
    Entry en(nullptr, nullptr, true);
    if (en.rc) return en.rc;
    size_t n = 0, bytes = 0;
    uint32_t c = 0, W = 0;
    bool found = false;
    for (Ctx* x : g_devs) {
        auto it = x->pinned.find(bases_xy);
        if (it == x->pinned.end()) continue;
        found = true;
        n += it->second.n;
        bytes += (size_t)(it->second.c ? it->second.W : 1) * it->second.n * sizeof(Affine);
        if (it->second.c) c = it->second.c, W = it->second.W;
    }
    if (!found) {
        set_error("bases_pinned_info: pointer is not pinned");
        return H2HIP_EINVAL;
    }
    if (n_points) *n_points = n;
    if (window_bits) *window_bits = c;
    if (windows) *windows = W;
    if (device_bytes) *device_bytes = bytes;
    return 0;
}

int h2hip_g1_fold(const uint64_t* partials_xyz, size_t k, uint64_t out_xyz[12]) {
    if (!out_xyz || (k && !partials_xyz)) {
        set_error("g1_fold: null argument");
        return H2HIP_EINVAL;
    }
    XYZZ acc = xyzz_identity();
    for (size_t i = 0; i < k; i++) {
        Jac j;
        memcpy(&j, partials_xyz + 12 * i, 96);
        xyzz_add(acc, jac_to_xyzz(j));
    }
    xyzz_to_out(acc, out_xyz);
    return 0;
}

int h2hip_g1_batch_normalize(const uint64_t* xyz, size_t k, uint64_t* xy) {
    if (k && (!xyz || !xy)) {
        set_error("g1_batch_normalize: null argument");
        return H2HIP_EINVAL;
    }
    // Montgomery's trick on the z coordinates (identity points are skipped)
    std::vector<Fe> prefix(k);
    Fe acc = fe_one<Q>();
    for (size_t i = 0; i < k; i++) {
        Jac j;
        memcpy(&j, xyz + 12 * i, 96);
        prefix[i] = acc;
        if (!fe_is_zero(j.z)) acc = fe_mul<Q>(acc, j.z);
    }
    Fe inv = fe_inv<Q>(acc);
    for (size_t i = k; i-- > 0;) {
        Jac j;
        memcpy(&j, xyz + 12 * i, 96);
        Affine a;
        if (fe_is_zero(j.z)) {
            a.x = fe_zero<Q>();
            a.y = fe_zero<Q>();
        } else {
            Fe zi = fe_mul<Q>(inv, prefix[i]);
            inv = fe_mul<Q>(inv, j.z);
            Fe zi2 = fe_sqr<Q>(zi);
            a.x = fe_mul<Q>(j.x, zi2);
            a.y = fe_mul<Q>(j.y, fe_mul<Q>(zi2, zi));
        }
        memcpy(xy + 8 * i, &a, 64);
    }
    return 0;
}

int h2hip_device_alloc(size_t bytes, void** d_ptr) {
    if (!d_ptr) {
        set_error("device_alloc: null argument");
        return H2HIP_EINVAL;
    }
    Entry en("h2hip_device_alloc");
    if (en.rc) return en.rc;
    hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 1);
    if (e != hipSuccess) {
        *d_ptr = nullptr;
        set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return H2HIP_ENOMEM;
    }
    return 0;
}

int h2hip_device_free(void* d_ptr) {
    Entry en("h2hip_device_free", d_ptr);
    if (en.rc) return en.rc;
    H2_CHECK(hipDeviceSynchronize());
    H2_CHECK(hipFree(d_ptr));
    return 0;
}

int h2hip_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes, void* stream) {
    if (bytes && (!d_dst || !h_src)) {
        set_error("memcpy_h2d: null argument");
        return H2HIP_EINVAL;
    }
    Entry en("h2hip_memcpy_h2d", d_dst);
    if (en.rc) return en.rc;
    H2_CHECK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    H2_CHECK(hipStreamSynchronize((hipStream_t)stream));  // the source is a borrowed host slice
    return 0;
}

int h2hip_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes, void* stream) {
    if (bytes && (!h_dst || !d_src)) {
        set_error("memcpy_d2h: null argument");
        return H2HIP_EINVAL;
    }
    Entry en("h2hip_memcpy_d2h", d_src);
    if (en.rc) return en.rc;
    H2_CHECK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    H2_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

int h2hip_memset_zero(void* d_dst, size_t bytes, void* stream) {
    if (bytes && !d_dst) {
        set_error("memset_zero: null argument");
        return H2HIP_EINVAL;
    }
    Entry en("h2hip_memset_zero", d_dst);
    if (en.rc) return en.rc;
    H2_CHECK(hipMemsetAsync(d_dst, 0, bytes, (hipStream_t)stream));
    return 0;
}

int h2hip_stream_synchronize(void* stream) {
    Entry en("h2hip_stream_synchronize");
    if (en.rc) return en.rc;
    H2_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

int h2hip_g1_to_affine(const uint64_t xyz[12], uint64_t xy[8]) {
    if (!xyz || !xy) {
        set_error("g1_to_affine: null argument");
        return H2HIP_EINVAL;
    }
    Jac j;
    memcpy(&j, xyz, 96);
    Affine a = xyzz_to_affine(jac_to_xyzz(j));
    memcpy(xy, &a, 64);
    return 0;
}

int h2hip_ntt_bn254_fr_device(void* d_a, const uint64_t omega[4], uint32_t log_n, void* stream) {
    if (!d_a || !omega) {
        set_error("ntt: null argument");
        return H2HIP_EINVAL;
    }
    if (check_fr(omega, "omega")) return H2HIP_EINVAL;
    Entry en("h2hip_ntt_bn254_fr_device", d_a);
    if (en.rc) return en.rc;
    hipStream_t s = (hipStream_t)stream;
    return ntt_device(en.c, (Fe*)d_a, fe_from_u64x4(omega), log_n, nullptr, s);
}

int h2hip_ntt_bn254_fr(uint64_t* a, const uint64_t omega[4], uint32_t log_n) {
    if (!a || !omega || log_n > 28) {
        set_error("ntt: bad argument");
        return H2HIP_EINVAL;
    }
    if (check_fr(omega, "omega")) return H2HIP_EINVAL;
    return ntt_host(a, fe_from_u64x4(omega), log_n, nullptr, nullptr, 0);
}

int h2hip_ifft_bn254_fr_device(void* d_a, const uint64_t omega_inv[4], uint32_t log_n, const uint64_t divisor[4], void* stream) {
    if (!d_a || !omega_inv || !divisor) {
        set_error("ifft: null argument");
        return H2HIP_EINVAL;
    }
    if (check_fr(omega_inv, "omega_inv") || check_fr(divisor, "divisor")) return H2HIP_EINVAL;
    Entry en("h2hip_ifft_bn254_fr_device", d_a);
    if (en.rc) return en.rc;
    hipStream_t s = (hipStream_t)stream;
    NttScale sc;
    sc.out_scale = true;
    sc.out3[0] = sc.out3[1] = sc.out3[2] = fe_from_u64x4(divisor);
    return ntt_device(en.c, (Fe*)d_a, fe_from_u64x4(omega_inv), log_n, &sc, s);
}

int h2hip_ifft_bn254_fr(uint64_t* a, const uint64_t omega_inv[4], uint32_t log_n, const uint64_t divisor[4]) {
    if (!a || !omega_inv || !divisor || log_n > 28) {
        set_error("ifft: bad argument");
        return H2HIP_EINVAL;
    }
    if (check_fr(omega_inv, "omega_inv") || check_fr(divisor, "divisor")) return H2HIP_EINVAL;
    NttScale sc;
    sc.out_scale = true;
    sc.out3[0] = sc.out3[1] = sc.out3[2] = fe_from_u64x4(divisor);
    return ntt_host(a, fe_from_u64x4(omega_inv), log_n, &sc, nullptr, 0);
}

int h2hip_coeff_to_extended_bn254_fr_device(void* d_a, uint32_t k, uint32_t extended_k, const uint64_t extended_omega[4],
                                            const uint64_t g_coset[4], const uint64_t g_coset_inv[4], void* stream) {
    if (!d_a || !extended_omega || !g_coset || !g_coset_inv || k > extended_k || extended_k > 28) {
        set_error("coeff_to_extended: bad argument");
        return H2HIP_EINVAL;
    }
    if (check_fr(extended_omega, "extended_omega") || check_fr(g_coset, "g_coset") || check_fr(g_coset_inv, "g_coset_inv")) return H2HIP_EINVAL;
    Entry en("h2hip_coeff_to_extended_bn254_fr_device", d_a);
    if (en.rc) return en.rc;
    hipStream_t s = (hipStream_t)stream;
    NttScale sc;
    make_zeta_scale(&sc, true, g_coset, g_coset_inv, nullptr);
    sc.in_len = 1ull << k;
    return ntt_device(en.c, (Fe*)d_a, fe_from_u64x4(extended_omega), extended_k, &sc, s);
}

int h2hip_coeff_to_extended_bn254_fr(const uint64_t* a, uint32_t k, uint64_t* out, uint32_t extended_k,
                                     const uint64_t extended_omega[4], const uint64_t g_coset[4], const uint64_t g_coset_inv[4]) {
    if (!a || !out || !extended_omega || !g_coset || !g_coset_inv || k > extended_k || extended_k > 28) {
        set_error("coeff_to_extended: bad argument");
        return H2HIP_EINVAL;
    }
    if (check_fr(extended_omega, "extended_omega") || check_fr(g_coset, "g_coset") || check_fr(g_coset_inv, "g_coset_inv")) return H2HIP_EINVAL;
    NttScale sc;
    make_zeta_scale(&sc, true, g_coset, g_coset_inv, nullptr);
    sc.in_len = 1ull << k;
    return ntt_host(out, fe_from_u64x4(extended_omega), extended_k, &sc, a, (size_t)1 << k);
}

int h2hip_extended_to_coeff_bn254_fr_device(void* d_a, uint32_t extended_k, const uint64_t extended_omega_inv[4],
                                            const uint64_t extended_ifft_divisor[4], const uint64_t g_coset[4],
                                            const uint64_t g_coset_inv[4], void* stream) {
    if (!d_a || !extended_omega_inv || !extended_ifft_divisor || !g_coset || !g_coset_inv || extended_k > 28) {
        set_error("extended_to_coeff: bad argument");
        return H2HIP_EINVAL;
    }
    if (check_fr(extended_omega_inv, "extended_omega_inv") || check_fr(extended_ifft_divisor, "extended_ifft_divisor") ||
        check_fr(g_coset, "g_coset") || check_fr(g_coset_inv, "g_coset_inv"))
        return H2HIP_EINVAL;
    Entry en("h2hip_extended_to_coeff_bn254_fr_device", d_a);
    if (en.rc) return en.rc;
    hipStream_t s = (hipStream_t)stream;
    NttScale sc;
    Fe div = fe_from_u64x4(extended_ifft_divisor);
    make_zeta_scale(&sc, false, g_coset, g_coset_inv, &div);
    return ntt_device(en.c, (Fe*)d_a, fe_from_u64x4(extended_omega_inv), extended_k, &sc, s);
}

int h2hip_extended_to_coeff_bn254_fr(uint64_t* a, uint32_t extended_k, const uint64_t extended_omega_inv[4],
                                     const uint64_t extended_ifft_divisor[4], const uint64_t g_coset[4], const uint64_t g_coset_inv[4]) {
    if (!a || !extended_omega_inv || !extended_ifft_divisor || !g_coset || !g_coset_inv || extended_k > 28) {
        set_error("extended_to_coeff: bad argument");
        return H2HIP_EINVAL;
    }
    if (check_fr(extended_omega_inv, "extended_omega_inv") || check_fr(extended_ifft_divisor, "extended_ifft_divisor") ||
        check_fr(g_coset, "g_coset") || check_fr(g_coset_inv, "g_coset_inv"))
        return H2HIP_EINVAL;
    NttScale sc;
    Fe div = fe_from_u64x4(extended_ifft_divisor);
    make_zeta_scale(&sc, false, g_coset, g_coset_inv, &div);
    return ntt_host(a, fe_from_u64x4(extended_omega_inv), extended_k, &sc, nullptr, 0);
}

int h2hip_divide_by_vanishing_poly_bn254_fr_device(void* d_a, uint32_t extended_k, const uint64_t* t_evaluations, uint32_t t_len, void* stream) {
    if (!d_a || !t_evaluations || extended_k > 28 || t_len == 0) {
        set_error("divide_by_vanishing_poly: bad argument");
        return H2HIP_EINVAL;
    }
    for (uint32_t i = 0; i < t_len; i++)
        if (check_fr(t_evaluations + 4 * i, "t_evaluations[i]")) return H2HIP_EINVAL;
    Entry en("h2hip_divide_by_vanishing_poly_bn254_fr_device", d_a);
    if (en.rc) return en.rc;
    return scale_periodic_device(en.c, (Fe*)d_a, 1ull << extended_k, t_evaluations, t_len, (hipStream_t)stream);
}

int h2hip_divide_by_vanishing_poly_bn254_fr(uint64_t* a, uint32_t extended_k, const uint64_t* t_evaluations, uint32_t t_len) {
    if (!a || !t_evaluations || extended_k > 28 || t_len == 0) {
        set_error("divide_by_vanishing_poly: bad argument");
        return H2HIP_EINVAL;
    }
    for (uint32_t i = 0; i < t_len; i++)
        if (check_fr(t_evaluations + 4 * i, "t_evaluations[i]")) return H2HIP_EINVAL;
    Entry en("h2hip_divide_by_vanishing_poly_bn254_fr");
    if (en.rc) return en.rc;
    Ctx* c = en.c;
    size_t bytes = sizeof(Fe) << extended_k;
    int rc = c->ntt_io.ensure(bytes);
    if (rc) return rc;
    H2_CHECK(hipMemcpyAsync(c->ntt_io.p, a, bytes, hipMemcpyHostToDevice, c->stream));
    rc = scale_periodic_device(c, (Fe*)c->ntt_io.p, 1ull << extended_k, t_evaluations, t_len, c->stream);
    if (rc) return rc;
    H2_CHECK(hipMemcpyAsync(a, c->ntt_io.p, bytes, hipMemcpyDeviceToHost, c->stream));
    H2_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

// ---- batched device-resident transforms: `count` columns of the same size, one launch per NTT pass
static int batch_args_ok(void* const* d_a, size_t count, uint32_t log_n, const char* what) {
    if (log_n > 28 || (count && !d_a)) {
        set_error("%s: bad argument", what);
        return 0;
    }
    for (size_t i = 0; i < count; i++)
        if (!d_a[i]) {
            set_error("%s: null column %zu", what, i);
            return 0;
        }
    return 1;
}

// SURVEY.md 8(e), second bullet: NTTs are single-GPU, but independent transforms (the coset NTTs of
// plonk/evaluation.rs:306-323) can run on several GPUs as whole columns.  A batch whose columns all live on one engine device
// runs there, asynchronously on the caller's stream, under that device's lock only.  A batch whose columns live on SEVERAL
// engine devices is split by owner: each device transforms its own columns as one batch on its engine stream, the groups run
// concurrently (one host thread per device), and the call returns when all of them are done (the caller's stream is
// synchronised first).  No column crosses xGMI: the caller decides the placement.  With the same device listed twice
// (HALO2_HIP_ALLOW_DUPLICATE_DEVICES, rehearsal) the columns of that device are dealt round-robin to its contexts.
// owner[i] = index (in the engine's device list) of the context that transforms column i; *mixed: more than one context is involved.
// Called with the engine lock held (inside an Entry).
static int batch_owners(const char* name, void* const* d_a, size_t count, std::vector<int>* owner, bool* mixed) {
    const int nd = (int)g_devs.size();
    owner->assign(count, 0);
    *mixed = false;
    if (nd <= 1) return 0;
    std::vector<int> next_dup((size_t)nd, 0);
    for (size_t i = 0; i < count; i++) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, d_a[i]) != hipSuccess) {
            (void)hipGetLastError();
            set_error("%s: column %zu is not a device pointer", name, i);
            return H2HIP_EINVAL;
        }
        std::vector<int> cand;
        for (int d = 0; d < nd; d++)
            if (g_devs[(size_t)d]->device == at.device) cand.push_back(d);
        if (cand.empty()) {
            set_error("%s: column %zu lives on device %d, which is not in h2hip_init's list", name, i, at.device);
            return H2HIP_EINVAL;
        }
        (*owner)[i] = cand[(size_t)(next_dup[(size_t)cand[0]]++) % cand.size()];
        if ((*owner)[i] != (*owner)[0]) *mixed = true;
    }
    return 0;
}

static int batch_over_devices(const char* name, void* const* d_a, size_t count, void* stream, uint32_t log_n, const Fe& omega, const NttScale* sc) {
    if (!count) return 0;
    std::vector<int> owner;
    bool mixed = false;
    {
        Entry en(name, d_a[0]);
        if (en.rc) return en.rc;
        int rc = batch_owners(name, d_a, count, &owner, &mixed);
        if (rc) return rc;
        if (!mixed) return ntt_device_batch(en.c, (Fe* const*)d_a, nullptr, count, omega, log_n, sc, (hipStream_t)stream);
    }
    Entry en(name, nullptr, true);  // several devices: all their locks, in list order
    if (en.rc) return en.rc;
    int rc = batch_owners(name, d_a, count, &owner, &mixed);  // again, now that the device list cannot change under the call
    if (rc) return rc;
    const int nd = (int)g_devs.size();
    H2_CHECK(hipStreamSynchronize((hipStream_t)stream));  // the columns may have been produced on the caller's stream
    std::vector<std::vector<Fe*>> cols((size_t)nd);
    for (size_t i = 0; i < count; i++) cols[(size_t)owner[i]].push_back((Fe*)d_a[i]);
    return on_devices(nd, [&](int d) -> int {
        Ctx* x = g_devs[(size_t)d];
        if (cols[(size_t)d].empty()) return 0;
        H2_CHECK(hipSetDevice(x->device));
        int r = ntt_device_batch(x, cols[(size_t)d].data(), nullptr, cols[(size_t)d].size(), omega, log_n, sc, x->stream);
        if (r) return r;
        H2_CHECK(hipStreamSynchronize(x->stream));
        return 0;
    });
}

int h2hip_ntt_bn254_fr_batch_device(void* const* d_a, size_t count, const uint64_t omega[4], uint32_t log_n, void* stream) {
    if (!omega || !batch_args_ok(d_a, count, log_n, "ntt_batch")) return H2HIP_EINVAL;
    if (check_fr(omega, "omega")) return H2HIP_EINVAL;
    return batch_over_devices("h2hip_ntt_bn254_fr_batch_device", d_a, count, stream, log_n, fe_from_u64x4(omega), nullptr);
}

int h2hip_ifft_bn254_fr_batch_device(void* const* d_a, size_t count, const uint64_t omega_inv[4], uint32_t log_n, const uint64_t divisor[4],
                                     void* stream) {
    if (!omega_inv || !divisor || !batch_args_ok(d_a, count, log_n, "ifft_batch")) return H2HIP_EINVAL;
    if (check_fr(omega_inv, "omega_inv") || check_fr(divisor, "divisor")) return H2HIP_EINVAL;
    NttScale sc;
    sc.out_scale = true;
    sc.out3[0] = sc.out3[1] = sc.out3[2] = fe_from_u64x4(divisor);
    return batch_over_devices("h2hip_ifft_bn254_fr_batch_device", d_a, count, stream, log_n, fe_from_u64x4(omega_inv), &sc);
}

int h2hip_coeff_to_extended_bn254_fr_batch_device(void* const* d_a, size_t count, uint32_t k, uint32_t extended_k, const uint64_t extended_omega[4],
                                                  const uint64_t g_coset[4], const uint64_t g_coset_inv[4], void* stream) {
    if (!extended_omega || !g_coset || !g_coset_inv || k > extended_k || !batch_args_ok(d_a, count, extended_k, "coeff_to_extended_batch"))
        return H2HIP_EINVAL;
    if (check_fr(extended_omega, "extended_omega") || check_fr(g_coset, "g_coset") || check_fr(g_coset_inv, "g_coset_inv")) return H2HIP_EINVAL;
    NttScale sc;
    make_zeta_scale(&sc, true, g_coset, g_coset_inv, nullptr);
    sc.in_len = 1ull << k;
    return batch_over_devices("h2hip_coeff_to_extended_bn254_fr_batch_device", d_a, count, stream, extended_k, fe_from_u64x4(extended_omega), &sc);
}

// ---- the same conversions on host columns, pipelined (ntt_host_batch)
static int host_batch_args_ok(const uint64_t* const* in, uint64_t* const* out, size_t count, uint32_t log_n, const char* what) {
    if (log_n > 28 || (count && (!in || !out))) {
        set_error("%s: bad argument", what);
        return 0;
    }
    for (size_t i = 0; i < count; i++)
        if (!in[i] || !out[i]) {
            set_error("%s: null column %zu", what, i);
            return 0;
        }
    return 1;
}

int h2hip_ntt_bn254_fr_batch(uint64_t* const* a, size_t count, const uint64_t omega[4], uint32_t log_n) {
    if (!omega || !host_batch_args_ok(a, a, count, log_n, "ntt_batch")) return H2HIP_EINVAL;
    if (check_fr(omega, "omega")) return H2HIP_EINVAL;
    return ntt_host_batch("h2hip_ntt_bn254_fr_batch", a, (size_t)1 << log_n, a, count, fe_from_u64x4(omega), log_n, nullptr);
}

int h2hip_ifft_bn254_fr_batch(uint64_t* const* a, size_t count, const uint64_t omega_inv[4], uint32_t log_n, const uint64_t divisor[4]) {
    if (!omega_inv || !divisor || !host_batch_args_ok(a, a, count, log_n, "ifft_batch")) return H2HIP_EINVAL;
    if (check_fr(omega_inv, "omega_inv") || check_fr(divisor, "divisor")) return H2HIP_EINVAL;
    NttScale sc;
    sc.out_scale = true;
    sc.out3[0] = sc.out3[1] = sc.out3[2] = fe_from_u64x4(divisor);
    return ntt_host_batch("h2hip_ifft_bn254_fr_batch", a, (size_t)1 << log_n, a, count, fe_from_u64x4(omega_inv), log_n, &sc);
}

int h2hip_coeff_to_extended_bn254_fr_batch(const uint64_t* const* a, uint32_t k, uint64_t* const* out, size_t count, uint32_t extended_k,
                                           const uint64_t extended_omega[4], const uint64_t g_coset[4], const uint64_t g_coset_inv[4]) {
    if (!extended_omega || !g_coset || !g_coset_inv || k > extended_k || !host_batch_args_ok(a, out, count, extended_k, "coeff_to_extended_batch"))
        return H2HIP_EINVAL;
    if (check_fr(extended_omega, "extended_omega") || check_fr(g_coset, "g_coset") || check_fr(g_coset_inv, "g_coset_inv")) return H2HIP_EINVAL;
    NttScale sc;
    make_zeta_scale(&sc, true, g_coset, g_coset_inv, nullptr);
    sc.in_len = 1ull << k;
    return ntt_host_batch("h2hip_coeff_to_extended_bn254_fr_batch", a, (size_t)1 << k, out, count, fe_from_u64x4(extended_omega), extended_k, &sc);
}

int h2hip_extended_to_coeff_bn254_fr_batch(uint64_t* const* a, size_t count, uint32_t extended_k, const uint64_t extended_omega_inv[4],
                                           const uint64_t extended_ifft_divisor[4], const uint64_t g_coset[4], const uint64_t g_coset_inv[4]) {
    if (!extended_omega_inv || !extended_ifft_divisor || !g_coset || !g_coset_inv || !host_batch_args_ok(a, a, count, extended_k, "extended_to_coeff_batch"))
        return H2HIP_EINVAL;
    if (check_fr(extended_omega_inv, "extended_omega_inv") || check_fr(extended_ifft_divisor, "extended_ifft_divisor") ||
        check_fr(g_coset, "g_coset") || check_fr(g_coset_inv, "g_coset_inv"))
        return H2HIP_EINVAL;
    NttScale sc;
    Fe div = fe_from_u64x4(extended_ifft_divisor);
    make_zeta_scale(&sc, false, g_coset, g_coset_inv, &div);
    return ntt_host_batch("h2hip_extended_to_coeff_bn254_fr_batch", a, (size_t)1 << extended_k, a, count, fe_from_u64x4(extended_omega_inv), extended_k, &sc);
}

int h2hip_debug_set_ntt_host_batch(uint64_t run_bytes, uint64_t group_bytes) {
    g_ntt_host_batch_bytes = run_bytes ? (size_t)run_bytes : (size_t)4 << 30;
    g_ntt_host_group_bytes = group_bytes ? (size_t)group_bytes : (size_t)2 << 20;
    return 0;
}

// ---- pinned proving-key columns (round 4) ------------------------------------------------------------------------------------------
// The host-pointer evaluate_h (patch 0003's path) uploads every column it is given, and most of them never change between proofs:
// pk.fixed_cosets, pk.l0 / l_last / l_active_row and pk.permutation.cosets are functions of the proving key -- 22 of the 27 full-size
// columns of the k = 18 bench system, 2.8 GB of 3.5 GB per call.  h2hip_columns_pin keeps such columns in HBM keyed by their host
// pointer; evaluate_h_host looks every full-size host column up and skips the upload on a hit.  Guarded like the pinned bases: a
// lookup compares H2_COL_SAMPLES sampled elements of the caller's memory (the first, and a geometric ladder to the last) with what was
// seen at pin time, so a freed and reused allocation costs an upload, never a stale column.  What the samples cannot see is a caller
// rewriting a pinned column in place; a ProvingKey never does.
extern "C" int h2hip_columns_pin(const uint64_t* const* cols, size_t count, size_t elems) {
    if ((count && !cols) || elems == 0 || elems > ((size_t)1 << 28)) {
        set_error("columns_pin: bad argument");
        return H2HIP_EINVAL;
    }
    for (size_t i = 0; i < count; i++)
        if (!cols[i]) {
            set_error("columns_pin: null column %zu", i);
            return H2HIP_EINVAL;
        }
    Entry en("h2hip_columns_pin");
    if (en.rc) return en.rc;
    Ctx* c = en.c;  // evaluate_h runs on the engine's first device
    const size_t bytes = elems * sizeof(Fe);
    const size_t budget = g_cfg.column_cache_bytes;  // (the Entry above holds the engine lock shared: the configuration cannot change under it)
    for (size_t i = 0; i < count; i++) {
        if (pinned_column_lookup(c, cols[i], elems)) continue;  // already there (and still the same array)
        while (!c->pinned_cols.empty() && c->pinned_cols_bytes + bytes > budget) {  // least recently used first
            auto victim = c->pinned_cols.begin();
            for (auto it = c->pinned_cols.begin(); it != c->pinned_cols.end(); ++it)
                if (it->second.last_use < victim->second.last_use) victim = it;
            H2_CHECK(hipDeviceSynchronize());
            (void)hipFree(victim->second.d);
            c->pinned_cols_bytes -= victim->second.elems * sizeof(Fe);
            c->pinned_cols.erase(victim);
        }
        if (bytes > budget) continue;  // does not fit at all: the call stays correct, this column is uploaded per call
        PinnedColumn pc;
        if (hipMalloc(&pc.d, bytes) != hipSuccess) {
            (void)hipGetLastError();
            set_error("columns_pin: out of device memory after %zu of %zu columns (the rest is uploaded per call)", i, count);
            return H2HIP_ENOMEM;
        }
        hipError_t e = hipMemcpyAsync(pc.d, cols[i], bytes, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            (void)hipFree(pc.d);
            set_error("columns_pin: upload failed: %s", hipGetErrorString(e));
            return H2HIP_EDEVICE;
        }
        pc.elems = elems;
        pc.last_use = ++c->pinned_cols_tick;
        column_sample(cols[i], elems, pc.sample);
        c->pinned_cols[cols[i]] = pc;
        c->pinned_cols_bytes += bytes;
    }
    return 0;
}

extern "C" int h2hip_columns_unpin(const uint64_t* const* cols, size_t count) {
    if (count && !cols) return H2HIP_EINVAL;
    {
        std::shared_lock<std::shared_mutex> lk(g_engine_mu);
        if (!ctx()->ready) return 0;  // nothing is pinned in an engine that is not running; the call never starts it
    }
    Entry en("h2hip_columns_unpin");
    if (en.rc) return en.rc;
    Ctx* c = en.c;
    bool synced = false;
    for (size_t i = 0; i < count; i++) {
        auto it = c->pinned_cols.find(cols[i]);
        if (it == c->pinned_cols.end()) continue;
        if (!synced) {
            H2_CHECK(hipDeviceSynchronize());  // a queued evaluate_h may still read the copy
            synced = true;
        }
        (void)hipFree(it->second.d);
        c->pinned_cols_bytes -= it->second.elems * sizeof(Fe);
        c->pinned_cols.erase(it);
    }
    return 0;
}

extern "C" int h2hip_columns_pinned_info(size_t* n_columns, size_t* device_bytes) {
    Entry en("h2hip_columns_pinned_info");
    if (en.rc) return en.rc;
    if (n_columns) *n_columns = en.c->pinned_cols.size();
    if (device_bytes) *device_bytes = en.c->pinned_cols_bytes;
    return 0;
}

int h2hip_g_to_lagrange_bn254_device(const void* d_g_xy, uint32_t k, void* d_g_lagrange_xy, void* stream) {
    if (!d_g_xy || !d_g_lagrange_xy || k > 28) {
        set_error("g_to_lagrange: bad argument");
        return H2HIP_EINVAL;
    }
    Entry en("h2hip_g_to_lagrange_bn254_device", d_g_xy);
    if (en.rc) return en.rc;
    return g_to_lagrange_device(en.c, (const Affine*)d_g_xy, k, (Affine*)d_g_lagrange_xy, (hipStream_t)stream);
}

int h2hip_g_to_lagrange_bn254(const uint64_t* g_xy, uint32_t k, uint64_t* g_lagrange_xy) {
    if (!g_xy || !g_lagrange_xy || k > 28) {
        set_error("g_to_lagrange: bad argument");
        return H2HIP_EINVAL;
    }
    Entry en("h2hip_g_to_lagrange_bn254");
    if (en.rc) return en.rc;
    Ctx* c = en.c;
    const size_t bytes = sizeof(Affine) << k;
    int rc = c->ntt_io.ensure(2 * bytes);
    if (rc) return rc;
    Affine* d_in = (Affine*)c->ntt_io.p;
    Affine* d_out = (Affine*)((char*)c->ntt_io.p + bytes);
    if ((rc = c->ws_acquire(c->stream))) return rc;
    H2_CHECK(hipMemcpyAsync(d_in, g_xy, bytes, hipMemcpyHostToDevice, c->stream));
    if ((rc = g_to_lagrange_device(c, d_in, k, d_out, c->stream))) return rc;
    H2_CHECK(hipMemcpyAsync(g_lagrange_xy, d_out, bytes, hipMemcpyDeviceToHost, c->stream));
    H2_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int h2hip_fft_bn254_g1_device(void* d_a_xyz, const uint64_t omega[4], uint32_t log_n, void* stream) {
    if (!d_a_xyz || !omega || log_n > 28) {
        set_error("fft_g1: bad argument");
        return H2HIP_EINVAL;
    }
    if (check_fr(omega, "omega")) return H2HIP_EINVAL;
    Entry en("h2hip_fft_bn254_g1_device", d_a_xyz);
    if (en.rc) return en.rc;
    return fft_g1_device(en.c, (Jac*)d_a_xyz, fe_from_u64x4(omega), log_n, (hipStream_t)stream);
}

int h2hip_fft_bn254_g1(uint64_t* a_xyz, const uint64_t omega[4], uint32_t log_n) {
    if (!a_xyz || !omega || log_n > 28) {
        set_error("fft_g1: bad argument");
        return H2HIP_EINVAL;
    }
    if (check_fr(omega, "omega")) return H2HIP_EINVAL;
    Entry en("h2hip_fft_bn254_g1");
    if (en.rc) return en.rc;
    Ctx* c = en.c;
    const size_t bytes = sizeof(Jac) << log_n;
    int rc = c->ntt_io.ensure(bytes);
    if (rc) return rc;
    H2_CHECK(hipMemcpyAsync(c->ntt_io.p, a_xyz, bytes, hipMemcpyHostToDevice, c->stream));
    if ((rc = fft_g1_device(c, (Jac*)c->ntt_io.p, fe_from_u64x4(omega), log_n, c->stream))) return rc;
    H2_CHECK(hipMemcpyAsync(a_xyz, c->ntt_io.p, bytes, hipMemcpyDeviceToHost, c->stream));
    H2_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int h2hip_kzg_setup_bn254_device(uint32_t k, const uint64_t secret[4], void* d_g_xy, void* d_g_lagrange_xy, void* stream) {
    if (!secret || !d_g_xy || !d_g_lagrange_xy) {
        set_error("kzg_setup: null argument");
        return H2HIP_EINVAL;
    }
    if (check_fr(secret, "secret")) return H2HIP_EINVAL;
    Entry en("h2hip_kzg_setup_bn254_device", d_g_xy);
    if (en.rc) return en.rc;
    return kzg_setup_device(en.c, k, fe_from_u64x4(secret), (Affine*)d_g_xy, (Affine*)d_g_lagrange_xy, (hipStream_t)stream);
}

int h2hip_kzg_setup_bn254(uint32_t k, const uint64_t secret[4], uint64_t* g_xy, uint64_t* g_lagrange_xy) {
    if (!secret || !g_xy || !g_lagrange_xy || k > 28) {
        set_error("kzg_setup: bad argument");
        return H2HIP_EINVAL;
    }
    if (check_fr(secret, "secret")) return H2HIP_EINVAL;
    Entry en("h2hip_kzg_setup_bn254");
    if (en.rc) return en.rc;
    Ctx* c = en.c;
    const size_t bytes = sizeof(Affine) << k;
    int rc = c->ntt_io.ensure(2 * bytes);
    if (rc) return rc;
    Affine* d_g = (Affine*)c->ntt_io.p;
    Affine* d_gl = (Affine*)((char*)c->ntt_io.p + bytes);
    if ((rc = kzg_setup_device(c, k, fe_from_u64x4(secret), d_g, d_gl, c->stream))) return rc;
    H2_CHECK(hipMemcpyAsync(g_xy, d_g, bytes, hipMemcpyDeviceToHost, c->stream));
    H2_CHECK(hipMemcpyAsync(g_lagrange_xy, d_gl, bytes, hipMemcpyDeviceToHost, c->stream));
    H2_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int h2hip_evaluate_h_bn254(const h2hip_evalh_desc* desc, uint64_t* values) {
    if (!desc || !values) {
        set_error("evaluate_h: null argument");
        return H2HIP_EINVAL;
    }
    if (evaluate_h_validate(desc, values)) return H2HIP_EINVAL;
    Entry en("h2hip_evaluate_h_bn254");
    if (en.rc) return en.rc;
    return evaluate_h_host(en.c, desc, values, false, en.c->stream);
}

int h2hip_evaluate_h_bn254_device(const h2hip_evalh_desc* desc, void* d_values, void* stream) {
    if (!desc || !d_values) {
        set_error("evaluate_h: null argument");
        return H2HIP_EINVAL;
    }
    if (evaluate_h_validate(desc, d_values)) return H2HIP_EINVAL;
    Entry en("h2hip_evaluate_h_bn254_device", d_values);
    if (en.rc) return en.rc;
    return evaluate_h_host(en.c, desc, (uint64_t*)d_values, true, (hipStream_t)stream);
}

int h2hip_gen_scalars_device(uint64_t seed, uint64_t start, size_t n, void* d_out, void* stream) {
    if (n && !d_out) {
        set_error("gen_scalars: null output");
        return H2HIP_EINVAL;
    }
    Entry en("h2hip_gen_scalars_device", d_out);
    if (en.rc) return en.rc;
    return gen_scalars_device(seed, start, n, (Fe*)d_out, (hipStream_t)stream);
}

int h2hip_gen_points_device(uint64_t seed, uint64_t start, size_t n, void* d_out, void* stream) {
    if (n && !d_out) {
        set_error("gen_points: null output");
        return H2HIP_EINVAL;
    }
    Entry en("h2hip_gen_points_device", d_out);
    if (en.rc) return en.rc;
    return gen_points_device(seed, start, n, (Affine*)d_out, (hipStream_t)stream);
}

int h2hip_set_msm_window(uint32_t c) {
    if (c != 0 && (c < 2 || c > 24)) {
        set_error("msm window must be 0 (auto) or 2..24");
        return H2HIP_EINVAL;
    }
    msm_set_window(c);
    return 0;
}

uint32_t h2hip_get_msm_window(size_t n) { return msm_get_window(n); }
uint32_t h2hip_get_msm_window_fixed_base(size_t n) { return msm_table_window(n); }

// test hook: split inputs above m pairs into consecutive chunks (default 2^26, the 31-bit pair-index limit)
int h2hip_debug_set_msm_max_chunk(size_t m) {
    msm_set_max_chunk(m);
    return 0;
}

// test / tuning hook: a host-resident MSM of at least min_n pairs streams in `chunks` pieces whose sizes grow by 1 / ratio
// (ratio_permille / 1000 = upload time over compute time per pair); chunks = 1 turns streaming off; zeros restore the defaults
int h2hip_debug_set_msm_stream(uint32_t chunks, uint32_t ratio_permille, size_t min_n) {
    msm_set_stream(chunks, ratio_permille / 1000.0, min_n);
    return 0;
}

// test hook, needs no GPU: the chunk sizes a streamed host-slice MSM of n pairs is cut into (chunks / ratio_permille 0 = the
// defaults in force; with_bases: the points cross PCIe too); returns the number of chunks, sizes[0 .. min(that, cap))
size_t h2hip_debug_msm_stream_ladder(size_t n, uint32_t chunks, uint32_t ratio_permille, int with_bases, size_t* sizes, size_t cap) {
    if (!sizes && cap) return 0;
    return msm_debug_ladder(n, chunks, ratio_permille / 1000.0, with_bases != 0, sizes, cap);
}

// tuning hook: buckets above (entries of the MSM) / d go to the chunked path (default 32768; 0 restores it)
int h2hip_debug_set_msm_heavy_div(size_t d) {
    msm_set_heavy_div(d);
    return 0;
}

// tuning hook: g_to_lagrange layers with one quad of lanes per butterfly up to k = 14 (1, default) or one lane (0)
int h2hip_debug_set_g2l_quad(int on) {
    ecfft_set_quad(on != 0);
    return 0;
}

// tuning hook: a fused run holds at most `entries` entries (default 2^26) and fusing applies up to `max_n` pairs per MSM (default 2^18); 0 = default
int h2hip_debug_set_msm_fuse_limits(size_t entries, size_t max_n) {
    msm_set_fuse_limits(entries, max_n);
    return 0;
}

// tuning hook: lane budget of the first row/column pass (default 65536 = one wave per SIMD) and its multiplier flavour
int h2hip_debug_set_msm_rowcol(uint64_t lanes, int use_asm) {
    msm_set_rowcol(lanes, (uint32_t)use_asm);
    return 0;
}

// tuning hook: several lanes per bucket in runs with few buckets (1, default) or always one (0)
int h2hip_debug_set_msm_split_buckets(int on) {
    msm_set_split_buckets(on != 0);
    return 0;
}

// tuning hook: the reduction tail with one quad of lanes per group operation (1, default) or one lane (0)
int h2hip_debug_set_msm_plane_tail(int on) {
    Entry en;
    if (en.rc) return en.rc;
    msm_set_plane_tail(on != 0);
    return 0;
}

int h2hip_debug_set_msm_quad_tail(int on) {
    msm_set_quad_tail(on != 0);
    return 0;
}

// tuning hook: 1 = order the buckets by size inside each sort bin only (no global pass); 0 = global order (default)
int h2hip_debug_set_msm_bucket_order(int local) {
    msm_set_bucket_order(local);
    return 0;
}

// tuning hook: level-1 records of runs with multi-tile bins as two arrays (1, default) or as (entry, bucket id) pairs (0)
int h2hip_debug_set_msm_split_records(int on) {
    msm_set_split_records(on != 0);
    return 0;
}

// tuning hook: target number of entries per coarse bin of the MSM's two-level sort (default 8192; 0 restores it)
int h2hip_debug_set_msm_bin_entries(size_t d) {
    msm_set_bin_entries(d);
    return 0;
}

// test hook: programs needing more slots than v use the global-workspace form of the evaluate_h kernels (default 256)
int h2hip_debug_set_evalh_max_local_slots(uint32_t v) {
    evalh_debug_set_max_local_slots(v);
    return 0;
}

// test hook: HBM one group of lookup cosets may take in evaluate_h (0 = default, 2 GB); a small value forces one lookup per group
int h2hip_debug_set_evalh_lookup_group_bytes(uint64_t v) {
    evalh_debug_set_lookup_group_bytes(v);
    return 0;
}

// test / tuning hook, needs no GPU: compile a graph as evaluate_h would and report the program's size
int h2hip_debug_evalh_program_muls(const h2hip_graph* g, uint32_t* n_mul) { return evalh_debug_program_muls(g, n_mul) ? H2HIP_EINVAL : 0; }

int h2hip_debug_set_evalh_codegen(int mode, uint32_t max_ops) {
    Entry en;
    if (en.rc) return en.rc;
    evalh_debug_set_codegen(mode, max_ops);
    return 0;
}

int h2hip_debug_evalh_codegen_source(const h2hip_graph* g, char* buf, size_t cap, size_t* len, int compile, double* seconds, size_t* code_bytes) {
    return evalh_debug_codegen_source(g, buf, cap, len, compile, seconds, code_bytes);
}

int h2hip_debug_evalh_codegen_stats(uint64_t out[5]) {
    if (!out) return H2HIP_EINVAL;
    evalh_debug_codegen_stats(out);
    return 0;
}

int h2hip_debug_evalh_compile_stats(const h2hip_graph* g, uint32_t* n_ops, uint32_t* n_slots) {
    return evalh_debug_compile_stats(g, n_ops, n_slots);
}

// test / tuning hook: batches of small MSMs run fused (default) or pipelined over streams
int h2hip_debug_set_msm_fuse_small(int on) {
    msm_set_fuse_small(on != 0);
    return 0;
}

int h2hip_debug_set_ntt_smax(uint32_t v) {
    ntt_set_smax(v);
    return 0;
}

// tuning hook: sizes 2^lo..2^hi (within 18..22) take the two-pass plan; hi < lo turns it off
// tuning hook: HBM (bytes per device) the two-pass plan's full inter-pass twiddle tables may take; 0 = two-level table only
int h2hip_debug_set_ntt_twiddle_budget(uint64_t bytes) {
    ntt_set_full_twiddle_budget(bytes);
    return 0;
}

// tuning hook: bytes of columns + workspace one launch of a batched transform spans (0 = default)
int h2hip_debug_set_ntt_batch_bytes(uint64_t bytes) {
    ntt_set_batch_bytes(bytes);
    return 0;
}

// tuning hook: pairs of workgroups per pass from which batched columns of 2^17 / 2^18 points take the two-pass plan (0 = default 512)
int h2hip_debug_set_ntt_two_pass_batch_wgs(uint64_t v) {
    ntt_set_two_pass_batch_wgs(v);
    return 0;
}

// tuning hook: the largest strided pass (log2 of its M) that reads its inter-pass twiddles from a table (0 = default 24)
int h2hip_debug_set_ntt_full_max_log_m(uint32_t v) {
    ntt_set_full_max_log_m(v);
    return 0;
}

// A/B hook: 0 = the inverse transform's 1/n is multiplied in by the last pass even where the first pass reads a table (rounds 1-3)
int h2hip_debug_set_ntt_fold_tables(int on) {
    ntt_set_fold_tables(on != 0);
    return 0;
}

// tuning hook: log2 columns per workgroup of the two-pass kernels (-1 = default)
int h2hip_debug_set_ntt_two_pass_log_j(int v) {
    ntt_set_two_pass_log_j(v);
    return 0;
}

int h2hip_debug_set_ntt_two_pass(uint32_t lo, uint32_t hi) {
    ntt_set_two_pass(lo, hi);
    return 0;
}

// undocumented tuning knob (not in the public header): CUs reserved for the sort / reduce stages of a batch
int h2hip_debug_set_reserved_cus(uint32_t k) {
    msm_set_reserved_cus(k);
    return 0;
}

int h2hip_profile_enable(int on) {
    Ctx* c = ctx();
    std::shared_lock<std::shared_mutex> eng(g_engine_mu);
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    c->profiling = on == 1;
    c->profiling_kernel = on == 2;
    return 0;
}

int h2hip_profile_reset(void) {
    Ctx* c = ctx();
    std::shared_lock<std::shared_mutex> eng(g_engine_mu);
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    c->timers_collect();
    c->timers.clear();
    return 0;
}

int h2hip_profile_get(const char* stage, double* total_ms, uint64_t* count) {
    if (!stage || !total_ms || !count) {
        set_error("profile_get: null argument");
        return H2HIP_EINVAL;
    }
    Ctx* c = ctx();
    std::shared_lock<std::shared_mutex> eng(g_engine_mu);
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    c->timers_collect();
    *total_ms = 0.0;
    *count = 0;
    for (auto& t : c->timers)
        if (t.name == stage) {
            *total_ms = t.total_ms;
            *count = t.count;
        }
    return 0;
}

}  // extern "C"
