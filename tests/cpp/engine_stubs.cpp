// Stand-ins for the kernels' host drivers (ntt.hip, msm.hip, evalh.hip, ecfft.hip, setup.hip, gen.hip) so that csrc/api.hip links
// into the ThreadSanitizer test without any device code (tests/cpp/test_engine_tsan.cpp).  They keep what the host logic relies on:
// the transform touches every word of its column (adds one: the test sees data go up, through and down intact), the MSM reads its
// scalars and writes its sums, and a host-slice MSM drives the upload copier the way msm.hip's msm_stream_host does.
#include <string.h>

#include "engine.h"

namespace h2 {

int gen_scalars_device(uint64_t, uint64_t, size_t, Fe*, hipStream_t) { return 0; }
int gen_points_device(uint64_t, uint64_t, size_t, Affine*, hipStream_t) { return 0; }
void msm_set_window(uint32_t) {}
void msm_set_max_chunk(size_t) {}
void msm_set_stream(uint32_t, double, size_t) {}
size_t msm_debug_ladder(size_t, uint32_t, double, bool, size_t*, size_t) { return 0; }
void msm_set_heavy_div(size_t) {}
void msm_set_bin_entries(size_t) {}
void msm_set_split_records(bool) {}
void msm_set_bucket_order(int) {}
void msm_set_quad_tail(bool) {}
void msm_set_split_buckets(bool) {}
void msm_set_plane_tail(bool) {}
void ecfft_set_quad(bool) {}
void ecfft_set_lazy(bool) {}
void msm_set_fuse_limits(size_t, size_t) {}
void msm_set_rowcol(uint64_t, uint32_t) {}
void ntt_set_smax(uint32_t) {}
void ntt_set_two_pass(uint32_t, uint32_t) {}
void ntt_set_full_twiddle_budget(uint64_t) {}
void ntt_set_batch_bytes(uint64_t) {}
void ntt_set_two_pass_log_j(int) {}
void ntt_set_full_max_log_m(uint32_t) {}
void ntt_set_fold_tables(bool) {}
void ntt_set_two_pass_batch_wgs(uint64_t) {}
void msm_set_reserved_cus(uint32_t) {}
uint32_t msm_get_reserved_cus() { return 0; }
uint32_t msm_get_window(size_t) { return 13; }
uint32_t msm_table_window(size_t) { return 13; }
void msm_set_fuse_small(bool) {}
void ntt_twiddles_free(Ctx*) {}
void evalh_debug_set_max_local_slots(uint32_t) {}
void evalh_debug_set_lookup_group_bytes(uint64_t) {}
int evalh_debug_compile_stats(const h2hip_graph*, uint32_t*, uint32_t*) { return 0; }
int evalh_debug_program_muls(const h2hip_graph*, uint32_t*) { return 0; }
void evalh_debug_set_codegen(int, uint32_t) {}
void evalh_debug_codegen_stats(uint64_t out[5]) { memset(out, 0, 5 * sizeof(uint64_t)); }
int evalh_debug_codegen_source(const h2hip_graph*, char*, size_t, size_t*, int, double*, size_t*) { return 0; }
void evalh_modules_free(Ctx*) {}
void evalh_rtc_shutdown() {}
int evaluate_h_validate(const h2hip_evalh_desc*, const void*) { return 0; }
int evaluate_h_host(Ctx*, const h2hip_evalh_desc*, uint64_t*, bool, hipStream_t) { return 0; }
int g_to_lagrange_device(Ctx*, const Affine*, uint32_t, Affine*, hipStream_t) { return 0; }
int fft_g1_device(Ctx*, Jac*, const Fe&, uint32_t, hipStream_t) { return 0; }
int kzg_setup_device(Ctx*, uint32_t, const Fe&, Affine*, Affine*, hipStream_t) { return 0; }
int scale_periodic_device(Ctx*, Fe*, uint64_t, const uint64_t*, uint32_t, hipStream_t) { return 0; }

int msm_table_build(Ctx* c, const Affine* d_points, size_t n, uint32_t, Affine* d_table, hipStream_t s) {
    if (d_table != d_points) memcpy(d_table, d_points, n * sizeof(Affine));
    int rc = c->ws_acquire(s);
    if (rc) return rc;
    WsGuard guard(c, s);
    return guard.release();
}

int ntt_device(Ctx* c, Fe* d_data, const Fe&, uint32_t log_n, const NttScale* sc, hipStream_t s, const Fe*) {
    int rc = c->ws_acquire(s);
    if (rc) return rc;
    WsGuard guard(c, s);
    rc = c->ntt_ws.ensure(sizeof(Fe) << log_n);  // the shared workspace, as ntt_run takes it
    if (rc) return rc;
    const uint64_t in_len = sc && sc->in_len ? sc->in_len : (1ull << log_n);
    uint64_t* w = (uint64_t*)d_data;
    for (uint64_t i = 0; i < (4ull << log_n); i++) w[i] = (i < 4 * in_len ? w[i] : 0) + 1;
    memcpy(c->ntt_ws.p, d_data, sizeof(Fe) << log_n);
    return guard.release();
}

int ntt_device_batch(Ctx* c, Fe* const* h_datas, const Fe* const*, size_t count, const Fe& omega, uint32_t log_n, const NttScale* sc, hipStream_t s) {
    for (size_t i = 0; i < count; i++) {
        int rc = ntt_device(c, h_datas[i], omega, log_n, sc, s, nullptr);
        if (rc) return rc;
    }
    return 0;
}

int msm_batch_device(Ctx* c, const Fe* const* scalars, bool scalars_on_host, const Affine*, size_t n, size_t count, XYZZ* h_out, hipStream_t s,
                     const MsmTable*, const Affine*) {
    int rc = c->msm_scalars[0].ensure(n * count * sizeof(Fe) + 64);
    if (rc) return rc;
    if ((rc = c->ensure_aux(count + 2))) return rc;
    if ((rc = c->ws_acquire(s))) return rc;
    WsGuard guard(c, s);
    Fe* d = (Fe*)c->msm_scalars[0].p;
    if (scalars_on_host && n && copier_ready(c, false)) {  // the pattern of msm_stream_host / msm_fused_groups_host
        std::vector<CopyJob> jobs;
        for (size_t j = 0; j < count; j++) jobs.push_back(CopyJob{d + j * n, scalars[j], n * sizeof(Fe), c->aux_events[1 + j], nullptr, false});
        if ((rc = copier_begin(c, false, c->aux2))) return rc;
        struct Drain {
            Ctx* c;
            size_t k;
            ~Drain() { copier_abort(c, false, k); }
        } drain{c, count};
        if ((rc = copier_push(c, false, jobs))) return rc;
        for (size_t j = 0; j < count; j++)
            if ((rc = copier_wait(c, false, j + 1))) return rc;
    } else {
        for (size_t j = 0; j < count; j++)
            if (n) memcpy(d + j * n, scalars[j], n * sizeof(Fe));
    }
    for (size_t j = 0; j < count; j++) {
        h_out[j] = xyzz_identity();
        if (n) h_out[j].x.l[0] = d[j * n + n - 1].l[0];  // depends on the uploaded scalars
    }
    return guard.release();
}

}  // namespace h2

extern "C" void* h2stub_device_alloc(int device, size_t bytes) {
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(device);
    void* p = nullptr;
    (void)hipMalloc(&p, bytes);
    (void)hipSetDevice(prev);
    return p;
}
