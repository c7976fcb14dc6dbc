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
void msm_set_max_chunk(size_t m);
void ntt_set_smax(uint32_t v);
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
    for (int k = 0; k < 3; k++) {
        c->msm_scalars[k].release();
        c->msm_slot[k].release();
    }
    c->msm_bases.release();
    c->host_ws.release();
    c->misc.release();
    c->evalh_ws.release();
    c->evalh_slots.release();
    c->ecfft_ws.release();
    c->ntt_ptrs.release();
    (void)hipStreamDestroy(c->stream);
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

static int msm_host_common(const uint64_t* const* scalars, const uint64_t* bases_xy, size_t n, size_t count, uint64_t* out_xyz) {
    Entry en;
    if (en.rc) return en.rc;
    Ctx* c = en.c;
    if (n == 0 || count == 0) {
        for (size_t j = 0; j < count; j++) xyzz_to_out(xyzz_identity(), out_xyz + 12 * j);
        return 0;
    }
    const Affine* d_bases = nullptr;
    auto it = c->pinned.find((const void*)bases_xy);
    if (it != c->pinned.end() && it->second.n >= n) {
        d_bases = (const Affine*)it->second.d;
    } else {
        int rc = c->msm_bases.ensure(n * sizeof(Affine));
        if (rc) return rc;
        H2_CHECK(hipMemcpyAsync(c->msm_bases.p, bases_xy, n * sizeof(Affine), hipMemcpyHostToDevice, c->stream));
        d_bases = (const Affine*)c->msm_bases.p;
    }
    std::vector<XYZZ> r(count);
    int rc = msm_batch_device(c, (const Fe* const*)scalars, true, d_bases, n, count, r.data(), c->stream);
    if (rc) return rc;
    for (size_t j = 0; j < count; j++) xyzz_to_out(r[j], out_xyz + 12 * j);
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
    Entry en;
    if (en.rc) return en.rc;
    std::vector<XYZZ> r(count);
    int rc = msm_batch_device(en.c, (const Fe* const*)d_scalars, false, (const Affine*)d_bases_xy, n, count, r.data(), (hipStream_t)stream);
    if (rc) return rc;
    for (size_t j = 0; j < count; j++) xyzz_to_out(r[j], out_xyz + 12 * j);
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
    Entry en;
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
    Entry en;
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
    Entry en;
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
    Entry en;
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
    Entry en;
    if (en.rc) return en.rc;
    H2_CHECK(hipMemsetAsync(d_dst, 0, bytes, (hipStream_t)stream));
    return 0;
}

int h2hip_stream_synchronize(void* stream) {
    Entry en;
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

int h2hip_divide_by_vanishing_poly_bn254_fr_device(void* d_a, uint32_t extended_k, const uint64_t* t_evaluations, uint32_t t_len, void* stream) {
    if (!d_a || !t_evaluations || extended_k > 28 || t_len == 0) {
        set_error("divide_by_vanishing_poly: bad argument");
        return H2HIP_EINVAL;
    }
    for (uint32_t i = 0; i < t_len; i++)
        if (check_fr(t_evaluations + 4 * i, "t_evaluations[i]")) return H2HIP_EINVAL;
    Entry en;
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
    Entry en;
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

int h2hip_ntt_bn254_fr_batch_device(void* const* d_a, size_t count, const uint64_t omega[4], uint32_t log_n, void* stream) {
    if (!omega || !batch_args_ok(d_a, count, log_n, "ntt_batch")) return H2HIP_EINVAL;
    if (check_fr(omega, "omega")) return H2HIP_EINVAL;
    Entry en;
    if (en.rc) return en.rc;
    return ntt_device_batch(en.c, (Fe* const*)d_a, nullptr, count, fe_from_u64x4(omega), log_n, nullptr, (hipStream_t)stream);
}

int h2hip_ifft_bn254_fr_batch_device(void* const* d_a, size_t count, const uint64_t omega_inv[4], uint32_t log_n, const uint64_t divisor[4],
                                     void* stream) {
    if (!omega_inv || !divisor || !batch_args_ok(d_a, count, log_n, "ifft_batch")) return H2HIP_EINVAL;
    if (check_fr(omega_inv, "omega_inv") || check_fr(divisor, "divisor")) return H2HIP_EINVAL;
    Entry en;
    if (en.rc) return en.rc;
    NttScale sc;
    sc.out_scale = true;
    sc.out3[0] = sc.out3[1] = sc.out3[2] = fe_from_u64x4(divisor);
    return ntt_device_batch(en.c, (Fe* const*)d_a, nullptr, count, fe_from_u64x4(omega_inv), log_n, &sc, (hipStream_t)stream);
}

int h2hip_coeff_to_extended_bn254_fr_batch_device(void* const* d_a, size_t count, uint32_t k, uint32_t extended_k, const uint64_t extended_omega[4],
                                                  const uint64_t g_coset[4], const uint64_t g_coset_inv[4], void* stream) {
    if (!extended_omega || !g_coset || !g_coset_inv || k > extended_k || !batch_args_ok(d_a, count, extended_k, "coeff_to_extended_batch"))
        return H2HIP_EINVAL;
    if (check_fr(extended_omega, "extended_omega") || check_fr(g_coset, "g_coset") || check_fr(g_coset_inv, "g_coset_inv")) return H2HIP_EINVAL;
    Entry en;
    if (en.rc) return en.rc;
    NttScale sc;
    make_zeta_scale(&sc, true, g_coset, g_coset_inv, nullptr);
    sc.in_len = 1ull << k;
    return ntt_device_batch(en.c, (Fe* const*)d_a, nullptr, count, fe_from_u64x4(extended_omega), extended_k, &sc, (hipStream_t)stream);
}

int h2hip_g_to_lagrange_bn254_device(const void* d_g_xy, uint32_t k, void* d_g_lagrange_xy, void* stream) {
    if (!d_g_xy || !d_g_lagrange_xy || k > 28) {
        set_error("g_to_lagrange: bad argument");
        return H2HIP_EINVAL;
    }
    Entry en;
    if (en.rc) return en.rc;
    return g_to_lagrange_device(en.c, (const Affine*)d_g_xy, k, (Affine*)d_g_lagrange_xy, (hipStream_t)stream);
}

int h2hip_g_to_lagrange_bn254(const uint64_t* g_xy, uint32_t k, uint64_t* g_lagrange_xy) {
    if (!g_xy || !g_lagrange_xy || k > 28) {
        set_error("g_to_lagrange: bad argument");
        return H2HIP_EINVAL;
    }
    Entry en;
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

int h2hip_evaluate_h_bn254(const h2hip_evalh_desc* desc, uint64_t* values) {
    if (!desc || !values) {
        set_error("evaluate_h: null argument");
        return H2HIP_EINVAL;
    }
    if (evaluate_h_validate(desc, values)) return H2HIP_EINVAL;
    Entry en;
    if (en.rc) return en.rc;
    return evaluate_h_host(en.c, desc, values, false, en.c->stream);
}

int h2hip_evaluate_h_bn254_device(const h2hip_evalh_desc* desc, void* d_values, void* stream) {
    if (!desc || !d_values) {
        set_error("evaluate_h: null argument");
        return H2HIP_EINVAL;
    }
    if (evaluate_h_validate(desc, d_values)) return H2HIP_EINVAL;
    Entry en;
    if (en.rc) return en.rc;
    return evaluate_h_host(en.c, desc, (uint64_t*)d_values, true, (hipStream_t)stream);
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

// test hook: split inputs above m pairs into consecutive chunks (default 2^26, the 31-bit pair-index limit)
int h2hip_debug_set_msm_max_chunk(size_t m) {
    msm_set_max_chunk(m);
    return 0;
}

// test hook: programs needing more slots than v use the global-workspace form of the evaluate_h kernels (default 256)
int h2hip_debug_set_evalh_max_local_slots(uint32_t v) {
    evalh_debug_set_max_local_slots(v);
    return 0;
}

// test / tuning hook, needs no GPU: compile a graph as evaluate_h would and report the program's size
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

// undocumented tuning knob (not in the public header): CUs reserved for the sort / reduce stages of a batch
int h2hip_debug_set_reserved_cus(uint32_t k) {
    msm_set_reserved_cus(k);
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
