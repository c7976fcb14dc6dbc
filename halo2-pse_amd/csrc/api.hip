// api.hip -- the extern "C" surface of libhalo2hip.so (include/halo2hip.h), the device
// context, workspace buffers and HIP-event stage timers.
#include <stdarg.h>
#include <string.h>

#include "../../include/halo2hip.h"
#include "engine.h"

namespace h2 {

int gen_scalars_device(uint64_t seed, uint64_t start, size_t n, Fe* d_out, hipStream_t s);
int gen_points_device(uint64_t seed, uint64_t start, size_t n, Affine* d_out, hipStream_t s);
void msm_set_window(uint32_t c);
void msm_set_groups(uint32_t g);
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

void DevBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

static Ctx g_ctx;
Ctx* ctx() { return &g_ctx; }

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

int Ctx::ensure_aux(size_t n_events) {
    if (!aux1) H2_CHECK(hipStreamCreateWithFlags(&aux1, hipStreamNonBlocking));
    if (!aux2) H2_CHECK(hipStreamCreateWithFlags(&aux2, hipStreamNonBlocking));
    while (aux_events.size() < n_events) {
        hipEvent_t e;
        H2_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        aux_events.push_back(e);
    }
    return 0;
}

static int do_init(const int* device_ids, int n_devices) {
    Ctx* c = ctx();
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (c->ready) return 0;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error("no usable HIP device (%s); libhalo2hip has no CPU fallback", e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return H2HIP_EDEVICE;
    }
    int dev = 0;
    if (device_ids && n_devices > 0) {
        dev = device_ids[0];
        if (dev < 0 || dev >= count) {
            set_error("device id %d out of range (0..%d)", dev, count - 1);
            return H2HIP_EINVAL;
        }
        H2_CHECK(hipSetDevice(dev));
    } else {
        H2_CHECK(hipGetDevice(&dev));
    }
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

int ensure_init() {
    if (ctx()->ready) return 0;
    return do_init(nullptr, 0);
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
    std::unique_lock<std::recursive_mutex> lk;
    int rc;
    Entry() : c(ctx()), rc(0) {
        rc = ensure_init();
        if (!rc) {
            lk = std::unique_lock<std::recursive_mutex>(c->mu);
            if (hipSetDevice(c->device) != hipSuccess) {
                set_error("hipSetDevice(%d) failed", c->device);
                rc = H2HIP_EDEVICE;
            }
        }
    }
};

static int ntt_host(uint64_t* a, const Fe& omega, uint32_t log_n, const NttScale* sc, const uint64_t* src, size_t src_elems) {
    Entry en;
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

}  // namespace h2

using namespace h2;

extern "C" {

int h2hip_init(const int* device_ids, int n_devices) { return do_init(device_ids, n_devices); }

void h2hip_shutdown(void) {
    Ctx* c = ctx();
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (!c->ready) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    c->timers_collect();
    c->timers.clear();
    for (auto& kv : c->twiddles) {
        (void)hipFree(kv.second.lo);
        (void)hipFree(kv.second.hi);
    }
    c->twiddles.clear();
    for (auto& kv : c->pinned) (void)hipFree(kv.second.d);
    c->pinned.clear();
    c->ntt_ws.release();
    c->ntt_io.release();
    c->msm_scalars.release();
    c->msm_bases.release();
    c->msm_ws.release();
    c->misc.release();
    (void)hipStreamDestroy(c->stream);
    c->stream = nullptr;
    for (auto e : c->aux_events) (void)hipEventDestroy(e);
    c->aux_events.clear();
    if (c->aux1) (void)hipStreamDestroy(c->aux1);
    if (c->aux2) (void)hipStreamDestroy(c->aux2);
    c->aux1 = c->aux2 = nullptr;
    if (c->ws_event) (void)hipEventDestroy(c->ws_event);
    c->ws_event = nullptr;
    c->ws_used = false;
    c->ready = false;
}

const char* h2hip_last_error(void) { return g_err; }
const char* h2hip_version(void) { return "halo2hip 0.1 (gfx950)"; }

int h2hip_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return -1;
    return count;
}

int h2hip_msm_bn254_device(const void* d_scalars, const void* d_bases_xy, size_t n, uint64_t out_xyz[12], void* stream) {
    if (!out_xyz || (n && (!d_scalars || !d_bases_xy))) {
        set_error("msm: null argument");
        return H2HIP_EINVAL;
    }
    Entry en;
    if (en.rc) return en.rc;
    hipStream_t s = (hipStream_t)stream;
    XYZZ r;
    int rc = msm_device(en.c, (const Fe*)d_scalars, (const Affine*)d_bases_xy, n, &r, s);
    if (rc) return rc;
    xyzz_to_out(r, out_xyz);
    return 0;
}

int h2hip_msm_bn254(const uint64_t* scalars, const uint64_t* bases_xy, size_t n, uint64_t out_xyz[12]) {
    if (!out_xyz || (n && (!scalars || !bases_xy))) {
        set_error("msm: null argument");
        return H2HIP_EINVAL;
    }
    Entry en;
    if (en.rc) return en.rc;
    Ctx* c = en.c;
    if (n == 0) {
        xyzz_to_out(xyzz_identity(), out_xyz);
        return 0;
    }
    int rc = c->msm_scalars.ensure(n * sizeof(Fe));
    if (rc) return rc;
    H2_CHECK(hipMemcpyAsync(c->msm_scalars.p, scalars, n * sizeof(Fe), hipMemcpyHostToDevice, c->stream));
    const Affine* d_bases = nullptr;
    auto it = c->pinned.find((const void*)bases_xy);
    if (it != c->pinned.end() && it->second.n >= n) {
        d_bases = (const Affine*)it->second.d;
    } else {
        rc = c->msm_bases.ensure(n * sizeof(Affine));
        if (rc) return rc;
        H2_CHECK(hipMemcpyAsync(c->msm_bases.p, bases_xy, n * sizeof(Affine), hipMemcpyHostToDevice, c->stream));
        d_bases = (const Affine*)c->msm_bases.p;
    }
    XYZZ r;
    rc = msm_device(c, (const Fe*)c->msm_scalars.p, d_bases, n, &r, c->stream);
    if (rc) return rc;
    xyzz_to_out(r, out_xyz);
    return 0;
}

int h2hip_bases_pin(const uint64_t* bases_xy, size_t n) {
    if (!bases_xy || !n) {
        set_error("bases_pin: null/empty");
        return H2HIP_EINVAL;
    }
    Entry en;
    if (en.rc) return en.rc;
    Ctx* c = en.c;
    auto it = c->pinned.find((const void*)bases_xy);
    if (it != c->pinned.end()) {
        H2_CHECK(hipDeviceSynchronize());
        (void)hipFree(it->second.d);
        c->pinned.erase(it);
    }
    PinnedBases pb;
    pb.n = n;
    hipError_t e = hipMalloc(&pb.d, n * sizeof(Affine));
    if (e != hipSuccess) {
        set_error("bases_pin: hipMalloc(%zu) failed: %s", n * sizeof(Affine), hipGetErrorString(e));
        return H2HIP_ENOMEM;
    }
    H2_CHECK(hipMemcpy(pb.d, bases_xy, n * sizeof(Affine), hipMemcpyHostToDevice));
    c->pinned[(const void*)bases_xy] = pb;
    return 0;
}

int h2hip_bases_unpin(const uint64_t* bases_xy) {
    Entry en;
    if (en.rc) return en.rc;
    Ctx* c = en.c;
    auto it = c->pinned.find((const void*)bases_xy);
    if (it == c->pinned.end()) {
        set_error("bases_unpin: pointer was not pinned");
        return H2HIP_EINVAL;
    }
    H2_CHECK(hipDeviceSynchronize());
    (void)hipFree(it->second.d);
    c->pinned.erase(it);
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
    Entry en;
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
    Entry en;
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
    Entry en;
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
    Entry en;
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

int h2hip_gen_scalars_device(uint64_t seed, uint64_t start, size_t n, void* d_out, void* stream) {
    if (n && !d_out) {
        set_error("gen_scalars: null output");
        return H2HIP_EINVAL;
    }
    Entry en;
    if (en.rc) return en.rc;
    return gen_scalars_device(seed, start, n, (Fe*)d_out, (hipStream_t)stream);
}

int h2hip_gen_points_device(uint64_t seed, uint64_t start, size_t n, void* d_out, void* stream) {
    if (n && !d_out) {
        set_error("gen_points: null output");
        return H2HIP_EINVAL;
    }
    Entry en;
    if (en.rc) return en.rc;
    return gen_points_device(seed, start, n, (Affine*)d_out, (hipStream_t)stream);
}

int h2hip_set_msm_window(uint32_t c) {
    if (c != 0 && (c < 2 || c > 22)) {
        set_error("msm window must be 0 (auto) or 2..22");
        return H2HIP_EINVAL;
    }
    msm_set_window(c);
    return 0;
}

uint32_t h2hip_get_msm_window(size_t n) { return msm_get_window(n); }

int h2hip_set_msm_groups(uint32_t g) {
    if (g > 16) {
        set_error("msm groups must be 0 (auto) or 1..16");
        return H2HIP_EINVAL;
    }
    msm_set_groups(g);
    return 0;
}

int h2hip_profile_enable(int on) {
    Ctx* c = ctx();
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    c->profiling = on != 0;
    return 0;
}

int h2hip_profile_reset(void) {
    Ctx* c = ctx();
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
