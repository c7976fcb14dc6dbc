// engine.h -- internal (not installed) declarations shared by the translation units of
// libhalo2hip.so: device context, workspace arena, per-stage HIP-event timers, error plumbing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/halo2hip.h"
#include "ecu.h"

namespace h2 {

void set_error(const char* fmt, ...);

#define H2_CHECK(expr)                                                                             \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            h2::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));    \
            return 2;                                                                              \
        }                                                                                          \
    } while (0)

// grow-only device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes);
    void release();
};

// grow-only pinned host buffer
struct HostBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes);
    void release();
};

struct TwiddleKey {
    uint32_t omega[8];
    uint32_t log_n;
    bool operator<(const TwiddleKey& o) const {
        for (int i = 0; i < 8; i++)
            if (omega[i] != o.omega[i]) return omega[i] < o.omega[i];
        return log_n < o.log_n;
    }
};

struct TwiddleTable {
    Fu* lo = nullptr;  // omega^i, i < 2^lo_bits                                  (I-form limbs, fieldu.h)
    Fu* hi = nullptr;  // omega^(i << lo_bits), i < 2^(log_n - lo_bits) (at least 1 entry)
    uint32_t lo_bits = 0;
    // the tile DFT's own twiddles w_R^i, i < R / 2, for R = 2^s (ntt.hip get_stage_twiddles): one table per radix a plan has used on
    // this domain.  Keyed by s, not by plan: a lone transform and a batch of the same domain take different plans (three passes /
    // two), and until round 4 every switch freed and rebuilt the other plan's tables behind a device-wide synchronisation.
    Fu* stage_of[13] = {};
    // the inter-pass twiddles of a pass as one table, when the budget allows (ntt.hip get_full_twiddles); tag = layout << 31 |
    // scaled << 30 | log_m << 8 | s (layout 1: the two-pass plan's pass 1; scaled: every entry times full_scale[k], the 1/n of the inverse
    // transform that uses this domain -- its last pass then closes with the direct reduction instead of a multiplication).  Up to
    // H2_TW_FULL tables per domain, none is ever replaced.
#define H2_TW_FULL 12
    Fu* full[H2_TW_FULL] = {};
    uint32_t full_tag[H2_TW_FULL] = {};
    Fe full_scale[H2_TW_FULL] = {};
    // `lo` times one constant (the 1/n of the inverse transform that uses this domain): a first pass that combines its inter-pass
    // twiddles from the two-level table then scales for free, and the last pass closes with the direct reduction (ntt.hip ntt_run)
    Fu* lo_scaled = nullptr;
    Fe lo_scale;
};

// Fixed-base window table of a pinned base array (msm.hip): row j (of `stride` points) = 2^(c j) * P, j < W
struct MsmTable {
    const Affine* table = nullptr;
    size_t stride = 0;
    uint32_t c = 0, W = 0;
};

#define H2_GATHER_OWN ((size_t)64 << 10)  // bytes of Ctx::gather a device's own set sums may take; the gathered ones follow
#define H2_PIN_SAMPLES 16
// index of fingerprint sample k of an array of `total` points
static inline __host__ __device__ size_t pin_sample_index(size_t total, uint32_t k) { return k == 0 ? 0 : (total - 1) >> (H2_PIN_SAMPLES - 1 - k); }
// One entry of the pinned-bases cache, keyed by the caller's pointer (host or device).
struct PinnedBases {
    void* d = nullptr;         // device copy: W x n points with the window table, or n points without one
    size_t n = 0;              // points per row
    uint32_t c = 0, W = 0;     // window width of the table; 0: no table (rows = 1)
    bool device_key = false;   // the key is a device pointer (h2hip_bases_pin_device)
    // fingerprint of the caller's array at pin time: H2_PIN_SAMPLES points (pin_sample_index: the first one and a geometric
    // ladder up to the last, so that every prefix length still sees several), compared byte for byte on every lookup -- a
    // freed-and-reused allocation at the same address must not hit the stale copy.  Host keys: compared on the host
    // (pinned_validate); device keys: d_sample holds the same bytes in device memory and the first workgroup of the MSM's first
    // kernel compares them with the caller's buffer (Ctx::pin_chk -> msm_l1_count_kernel), its verdict is read when the MSM's own result arrives.
    uint8_t sample[H2_PIN_SAMPLES * 64];
    void* d_sample = nullptr;
    size_t lo = 0, hi = 0;     // multi-GPU: this device holds points [lo, hi) of the caller's array (n = hi - lo)
};

struct Ctx;

// A constant column of a proving key (pk.fixed_cosets, pk.l0 / l_last / l_active_row, pk.permutation.cosets) kept in HBM across
// evaluate_h calls, keyed by the host pointer and guarded by a fingerprint as the pinned bases are (api.hip, h2hip_columns_pin)
#define H2_COL_SAMPLES 16
struct PinnedColumn {
    void* d = nullptr;
    size_t elems = 0;
    uint64_t last_use = 0;
    uint8_t sample[H2_COL_SAMPLES * 32];
};

struct StageTimer {
    std::string name;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    double total_ms = 0.0;
    uint64_t count = 0;
    bool pending = false;
};

// ---- copier threads (api.hip): hipMemcpyAsync to or from the caller's pageable memory moves at PCIe speed but returns only
// when its piece has crossed, so the thread that enqueues kernels must not sit in it.  Each device context has two helper
// threads, one per direction (PCIe is full duplex: the uploads of column i + 1 and the downloads of column i - 1 overlap).
// A copier walks a job queue in order; per job: wait for `gate` on its stream (an event the enqueueing thread has already
// recorded), copy, record `ev`, bump `done`.  Users: the streamed host-slice MSM (msm.hip) and the host-pointer batched
// transforms (api.hip, ntt_host_batch).
struct CopyJob {
    void* dst;
    const void* src;
    size_t bytes;
    hipEvent_t ev;    // recorded on the copy stream after this job; nullptr: none
    hipEvent_t gate;  // the copy stream waits for it before this job; nullptr: none
    bool d2h;
};
struct Copier;
// begin a session on c's upload (down = false) or download (down = true) copier -- started on first use --: `done` and the
// error state are reset, later jobs go to `stream`.  H2HIP_ENOMEM / H2HIP_EDEVICE when the thread cannot be started.
bool copier_ready(Ctx* c, bool down);  // the thread exists (started now if need be); false: it cannot be started, take the unstreamed path
int copier_begin(Ctx* c, bool down, hipStream_t stream);
// append jobs to the session (the vector is left empty)
int copier_push(Ctx* c, bool down, std::vector<CopyJob>& jobs);
// block until the first n_jobs jobs of the session have been issued (for pageable memory: have crossed); non-zero: a copy failed
int copier_wait(Ctx* c, bool down, size_t n_jobs);
// error-path drain: the jobs not yet started are skipped, the call returns once the copier no longer touches the caller's memory
void copier_abort(Ctx* c, bool down, size_t n_jobs);

struct Ctx {
    int device = -1;
    bool ready = false;
    hipStream_t stream = nullptr;  // the engine's own stream (host-pointer entry points)
    std::recursive_mutex mu;       // serialises entry points: re-entrant callers (rayon workers) are safe
    DevBuf ntt_ws, ntt_io, msm_scalars[3], msm_bases, msm_slot[3], misc, evalh_ws, evalh_slots, ecfft_ws, ntt_ptrs, gather, gen_table;
    HostBuf host_ws;               // pinned host memory for the window sums coming back
    HostBuf host_planes;           // ... and for the bit-plane sums of a run whose tail the host finishes (msm.hip msm_planes_finish)
    HostBuf pin_flag;              // one word the device-key fingerprint check writes its verdict to
    // a fingerprint check the next MSM run's first kernel carries out (msm_l1_count_kernel's first workgroup: no launch of its own);
    // set by msm_device_keyed around its msm_batch_device call, taken by the first msm_stage_a
    struct PinCheck {
        const Affine* bases = nullptr;
        const Affine* samples = nullptr;
        uint32_t* flag = nullptr;
        size_t n = 0, total = 0;
    } pin_chk;
    // Small host tables the kernels read (pointer lists, constants) go through this pinned ring, so that the
    // asynchronous copy never reads a caller's stack or a std::vector that is gone by the time the DMA runs.
    HostBuf stage;
    size_t stage_off = 0;
    int stage_h2d(void* d_dst, const void* h_src, size_t bytes, hipStream_t s);
    std::map<TwiddleKey, TwiddleTable> twiddles;
    size_t tw_full_bytes = 0;  // HBM held by the two-pass plan's full inter-pass twiddle tables
    std::map<const void*, PinnedBases> pinned;
    std::map<const void*, PinnedColumn> pinned_cols;  // h2hip_columns_pin: the host-pointer evaluate_h finds a proving key's constant columns here
    size_t pinned_cols_bytes = 0;
    uint64_t pinned_cols_tick = 0;
    // profiling
    bool profiling = false;        // every stage bracketed by a pair of events (each record costs the stream ~10 us)
    bool profiling_kernel = false; // only the dominant kernel, timed through its own dispatch (hipExtLaunchKernelGGL): no gap
    std::vector<StageTimer> timers;
    int timer_begin(const char* name, hipStream_t s);
    void timer_end(int id, hipStream_t s);
    int timer_kernel(const char* name, hipEvent_t* e0, hipEvent_t* e1);  // events for hipExtLaunchKernelGGL; -1: not profiling
    void timers_collect();
    int sm_count = 256;
    // The workspaces above are shared by every call.  Calls are serialised on the host by `mu`, but
    // device entry points return while their kernels are still queued; a later call on a different
    // stream first waits (on the device) for the previous user's last kernel.
    hipEvent_t ws_event = nullptr;
    hipStream_t ws_last_stream = nullptr;
    bool ws_used = false;
    int ws_acquire(hipStream_t s);
    int ws_release(hipStream_t s);
    // internal streams + events for the MSM's window-group pipeline
    hipStream_t aux1 = nullptr, aux2 = nullptr, aux_b = nullptr;
    uint32_t aux_reserved = 0xffffffffu;  // CU reservation the aux streams were created with
    std::vector<hipEvent_t> aux_events;
    int ensure_aux(size_t n_events);
    // evalh.hip: the per-circuit gates kernels loaded on this device, by source hash -> (hipModule_t, hipFunction_t); a pair of nulls
    // marks a code object that would not load here
    std::map<uint64_t, std::pair<void*, void*>> evalh_mods;
    Copier* copier[2] = {nullptr, nullptr};  // [0] uploads, [1] downloads; created on first use, joined by copier_stop (release_ctx)
    // multi-device engine, HALO2_HIP_GATHER=rccl: stage C also leaves the run's set sums in `gather` (device memory), from where
    // ncclAllGather takes them; gather_off counts the bytes left there by this call (SIZE_MAX / 2 and up: not one run, unusable)
    bool gather_want = false;
    size_t gather_off = 0;
};

void copier_stop(Ctx* c);

// Between ws_acquire and ws_release a call owns the shared workspaces.  An error return in between must not leave kernels
// queued on the internal streams unaccounted for: the guard drains them and still records the release, so the next
// call (on whatever stream) waits for everything this one started.
struct WsGuard {
    Ctx* c;
    hipStream_t s;
    bool done = false;
    WsGuard(Ctx* c_, hipStream_t s_) : c(c_), s(s_) {}
    int release() {
        done = true;
        return c->ws_release(s);
    }
    ~WsGuard() {
        if (done) return;
        if (c->aux1) (void)hipStreamSynchronize(c->aux1);
        if (c->aux2) (void)hipStreamSynchronize(c->aux2);
        if (c->aux_b) (void)hipStreamSynchronize(c->aux_b);
        (void)hipStreamSynchronize(s);
        (void)c->ws_release(s);
    }
};

Ctx* ctx();             // the primary device's context (device_ids[0] of h2hip_init)
int n_devices();        // devices the engine was initialised with (>= 1 once ready)
Ctx* ctx_at(int i);     // context of the i-th device of h2hip_init's list
int ensure_init();      // lazily h2hip_init(NULL, 0), or the HALO2_HIP_DEVICES list

// ntt.hip
struct NttScale {
    // optional fused pointwise steps (poly/domain.rs:246-247, :294, :355-360)
    bool in_scale = false;    // a[i] *= in3[i % 3] on the first-pass load
    Fe in3[3];
    uint64_t in_len = 0;      // elements at index >= in_len are read as zero (resize(.., zero), domain.rs:247)
    bool out_scale = false;   // a[i] *= out3[i % 3] on the final-pass store
    Fe out3[3];
};
// d_src (optional): the first pass reads its input there instead of d_data (which is then output only); with
// sc->in_len set only d_src[0 .. in_len) is read
void ntt_twiddles_free(Ctx* c);
int ntt_device(Ctx* c, Fe* d_data, const Fe& omega, uint32_t log_n, const NttScale* sc, hipStream_t s, const Fe* d_src = nullptr);

int ntt_power_table(Ctx* c, const Fe& omega, uint32_t log_n, hipStream_t s, const Fu** lo, const Fu** hi, uint32_t* lo_bits);
int ntt_device_batch(Ctx* c, Fe* const* h_datas, const Fe* const* h_srcs, size_t count, const Fe& omega, uint32_t log_n, const NttScale* sc,
                     hipStream_t s);

int scale_periodic_device(Ctx* c, Fe* d_a, uint64_t n, const uint64_t* h_t, uint32_t t_len, hipStream_t s);

// ecfft.hip
int g_to_lagrange_device(Ctx* c, const Affine* d_g, uint32_t k, Affine* d_out, hipStream_t s);
int ec_normalize_device(const XYZZ* d_in, Affine* d_out, uint64_t n, hipStream_t s);  // batched XYZZ -> affine
int fft_g1_device(Ctx* c, Jac* d_a, const Fe& omega, uint32_t log_n, hipStream_t s);   // best_fft::<G1>, in place on Jacobian points

// setup.hip
int kzg_setup_device(Ctx* c, uint32_t k, const Fe& s, Affine* d_g, Affine* d_gl, hipStream_t stream);

// api.hip: the device copy of a pinned host column whose fingerprint still matches the caller's memory, or nullptr
const Fe* pinned_column_lookup(Ctx* c, const uint64_t* h_col, size_t elems);

// evalh.hip
void evalh_debug_set_max_local_slots(uint32_t v);
void evalh_debug_set_lookup_group_bytes(uint64_t v);
int evalh_debug_compile_stats(const h2hip_graph* g, uint32_t* n_ops, uint32_t* n_slots);
int evalh_debug_program_muls(const h2hip_graph* g, uint32_t* n_mul);
void evalh_debug_set_codegen(int mode, uint32_t max_ops);  // 0 off, 1 background compile (default), 2 compile inline; max_ops 0 = default
void evalh_debug_codegen_stats(uint64_t out[5]);            // compiled, failed, generated-kernel launches, interpreter launches, disk-cache hits
int evalh_debug_codegen_source(const h2hip_graph* g, char* buf, size_t cap, size_t* len, int compile, double* seconds, size_t* code_bytes);
void evalh_modules_free(Ctx* c);                            // unload this device's generated kernels (release_ctx)
void evalh_rtc_shutdown();                                  // join the compile threads (h2hip_shutdown)
int evaluate_h_validate(const h2hip_evalh_desc* d, const void* values);
int evaluate_h_host(Ctx* c, const h2hip_evalh_desc* d, uint64_t* values, bool dev, hipStream_t s);

// msm.hip
void msm_set_fuse_small(bool on);
// tab != nullptr: fixed-base form over tab's window table (d_bases unused)
int msm_device(Ctx* c, const Fe* d_scalars, const Affine* d_bases, size_t n, XYZZ* h_out, hipStream_t s, const MsmTable* tab = nullptr);
int msm_batch_device(Ctx* c, const Fe* const* scalars, bool scalars_on_host, const Affine* d_bases, size_t n, size_t count, XYZZ* h_out,
                     hipStream_t s, const MsmTable* tab = nullptr, const Affine* h_bases = nullptr);
uint32_t msm_table_window(size_t n);  // window width a table for n pinned points is built with
int msm_table_build(Ctx* c, const Affine* d_points, size_t n, uint32_t cw, Affine* d_table, hipStream_t s);

}  // namespace h2
