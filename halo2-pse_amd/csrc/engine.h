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
#include "ecu.cuh"

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
    Fu* lo = nullptr;  // omega^i, i < 2^lo_bits                                  (I-form limbs, fieldu.cuh)
    Fu* hi = nullptr;  // omega^(i << lo_bits), i < 2^(log_n - lo_bits) (at least 1 entry)
    uint32_t lo_bits = 0;
};

struct PinnedBases {
    void* d = nullptr;
    size_t n = 0;
};

struct StageTimer {
    std::string name;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    double total_ms = 0.0;
    uint64_t count = 0;
    bool pending = false;
};

struct Ctx {
    int device = -1;
    bool ready = false;
    hipStream_t stream = nullptr;  // the engine's own stream (host-pointer entry points)
    std::recursive_mutex mu;       // serialises entry points: re-entrant callers (rayon workers) are safe
    DevBuf ntt_ws, ntt_io, msm_scalars[3], msm_bases, msm_slot[3], misc, evalh_ws, evalh_slots, ecfft_ws, ntt_ptrs;
    HostBuf host_ws;               // pinned host memory for the window sums coming back
    std::map<TwiddleKey, TwiddleTable> twiddles;
    std::map<const void*, PinnedBases> pinned;
    // profiling
    bool profiling = false;
    std::vector<StageTimer> timers;
    int timer_begin(const char* name, hipStream_t s);
    void timer_end(int id, hipStream_t s);
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
};

Ctx* ctx();             // the process-wide context (one process per GPU)
int ensure_init();      // lazily h2hip_init(NULL, 0)

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
int ntt_device(Ctx* c, Fe* d_data, const Fe& omega, uint32_t log_n, const NttScale* sc, hipStream_t s, const Fe* d_src = nullptr);

int ntt_device_batch(Ctx* c, Fe* const* h_datas, const Fe* const* h_srcs, size_t count, const Fe& omega, uint32_t log_n, const NttScale* sc,
                     hipStream_t s);

int scale_periodic_device(Ctx* c, Fe* d_a, uint64_t n, const uint64_t* h_t, uint32_t t_len, hipStream_t s);

// ecfft.hip
int g_to_lagrange_device(Ctx* c, const Affine* d_g, uint32_t k, Affine* d_out, hipStream_t s);

// evalh.hip
void evalh_debug_set_max_local_slots(uint32_t v);
int evalh_debug_compile_stats(const h2hip_graph* g, uint32_t* n_ops, uint32_t* n_slots);
int evaluate_h_validate(const h2hip_evalh_desc* d, const void* values);
int evaluate_h_host(Ctx* c, const h2hip_evalh_desc* d, uint64_t* values, bool dev, hipStream_t s);

// msm.hip
void msm_set_fuse_small(bool on);
int msm_device(Ctx* c, const Fe* d_scalars, const Affine* d_bases, size_t n, XYZZ* h_out, hipStream_t s);
int msm_batch_device(Ctx* c, const Fe* const* scalars, bool scalars_on_host, const Affine* d_bases, size_t n, size_t count, XYZZ* h_out,
                     hipStream_t s);

}  // namespace h2
