// setup.hip -- ParamsKZG::setup (halo2_proofs/src/poly/kzg/commitment.rs:61-129) with the secret supplied by the
// caller: g[i] = [s^i] G1 (:71-87) and g_lagrange[i] = [l_i(s)] G1 with l_i(s) = (s^n - 1) / n * w^i / (s - w^i)
// (:89-104), 2 * 2^k scalar multiplications of ONE point -- the G1 generator (1, 2).
//
// A fixed base needs no ladder: with the table T[j][d] = d * 2^(8 j) * G (j < 32, 0 < d < 256; 512 KB of affine
// points, built once per process and L2-resident), [e] G = sum_j T[j][byte j of e] is at most 32 mixed additions
// (~300 field multiplications against ~3 200 for double-and-add).  One lane per output point; the scalars s^i and
// l_i(s) are formed on the device (one Fermat inversion per lane for 1 / (s - w^i)); the XYZZ results go through the
// batched normalisation of ecfft.hip, so the affine output is canonical and is compared limb for limb.
#include <string.h>

#include <vector>

#include "engine.h"
#include "host64.h"

namespace h2 {

#define SETUP_WIN 32   // 8-bit windows of a 256-bit scalar
#define SETUP_DIG 256

// lane i: e_g[i] = s^i, e_gl[i] = mult * w^i / (s - w^i), both as canonical integers
__global__ void __launch_bounds__(256) kzg_setup_scalars_kernel(Fe s, Fe mult, Fe root, uint64_t n, Fe* __restrict__ e_g, Fe* __restrict__ e_gl) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    e_g[i] = fe_to_canonical<FrP>(fe_pow_u64<FrP>(s, i));                            // :76-81
    const Fe root_pow = fe_pow_u64<FrP>(root, i);                                    // :99
    const Fe d = fu_mul_canon<FrU>(fu_inv<FrU>(fu_from_ext(fe_sub<FrP>(s, root_pow))), fu_one_e<FrU>());  // (s - root_pow).invert(), :100
    e_gl[i] = fe_to_canonical<FrP>(fe_mul<FrP>(fe_mul<FrP>(mult, root_pow), d));     // :100
}

// out[i] = [e[i]] G by table lookups: one mixed addition per non-zero byte of e[i]
__global__ void __launch_bounds__(256) kzg_setup_mul_kernel(const Fe* __restrict__ e, const Affine* __restrict__ table, uint64_t n,
                                                            XYZZ* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fe sc = e[i];
    XYZZu acc = xyzzu_identity();
#pragma unroll
    for (int limb = 0; limb < 8; limb++) {
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const uint32_t d = (sc.l[limb] >> (8 * b)) & 0xffu;
            if (d) xyzzu_add_affine<FqU>(acc, table[(limb * 4 + b) * SETUP_DIG + d], false);
        }
    }
    out[i] = xyzzu_to_ext(acc);
}

// T[j][d] = d * 2^(8 j) * G as XYZZ, on the host (8 k group operations, a few milliseconds once per process)
static void build_generator_table(std::vector<XYZZ>& t) {
    t.assign((size_t)SETUP_WIN * SETUP_DIG, xyzz_identity());
    Affine g;
    g.x = fe_from_u64<FqP>(1);
    g.y = fe_from_u64<FqP>(2);
    h64::P base = h64::from_xyzz(xyzz_from_affine(g));
    for (int j = 0; j < SETUP_WIN; j++) {
        h64::P cur = base;
        for (int d = 1; d < SETUP_DIG; d++) {
            t[(size_t)j * SETUP_DIG + d] = h64::to_xyzz(cur);
            h64::padd(cur, base);
        }
        base = cur;  // 256 * 2^(8 j) * G = 2^(8 (j + 1)) * G
    }
}

// d_g, d_gl: 2^k affine points each (device).  s: the secret, Montgomery form.  Queued on `stream`; does not wait.
int kzg_setup_device(Ctx* c, uint32_t k, const Fe& s, Affine* d_g, Affine* d_gl, hipStream_t stream) {
    if (k > FrP::S) {
        set_error("kzg_setup: assertion failed: k <= Fr::S (%u > %u)", k, FrP::S);  // :64
        return 1;
    }
    const uint64_t n = 1ull << k;
    // :90-97: root = ROOT_OF_UNITY_INV^-1 squared S - k times; multiplier = (s^n - 1) / n
    Fe root;
    memcpy(root.l, FrP::ROOT_OF_UNITY, sizeof(root.l));  // ROOT_OF_UNITY_INV.invert()
    for (uint32_t i = k; i < FrP::S; i++) root = fe_sqr<FrP>(root);
    const Fe n_inv = fe_inv<FrP>(fe_from_u64<FrP>(n));
    const Fe s_n = fe_pow_u64<FrP>(s, n);
    if (fe_eq(s_n, fe_one<FrP>())) {
        set_error("kzg_setup: s is a 2^k-th root of unity ((s - root_pow).invert().unwrap() panics in the reference)");
        return 1;
    }
    const Fe mult = fe_mul<FrP>(fe_sub<FrP>(s_n, fe_one<FrP>()), n_inv);
    int rc = c->ws_acquire(stream);
    if (rc) return rc;
    WsGuard guard(c, stream);
    // the generator's table: XYZZ from the host, normalised to affine on the device, kept for the life of the context
    const size_t tab_points = (size_t)SETUP_WIN * SETUP_DIG;
    if (!c->gen_table.p) {
        std::vector<XYZZ> host_tab;
        build_generator_table(host_tab);
        if ((rc = c->gen_table.ensure(tab_points * (sizeof(Affine) + sizeof(XYZZ))))) return rc;
        XYZZ* d_x = (XYZZ*)((char*)c->gen_table.p + tab_points * sizeof(Affine));
        H2_CHECK(hipMemcpyAsync(d_x, host_tab.data(), tab_points * sizeof(XYZZ), hipMemcpyHostToDevice, stream));
        H2_CHECK(hipStreamSynchronize(stream));  // host_tab lives on this frame
        if ((rc = ec_normalize_device(d_x, (Affine*)c->gen_table.p, tab_points, stream))) return rc;
    }
    const Affine* table = (const Affine*)c->gen_table.p;
    // workspace: two scalar arrays and one XYZZ array of n elements
    if ((rc = c->ecfft_ws.ensure(n * (2 * sizeof(Fe) + sizeof(XYZZ)) + 512))) return rc;
    Fe* e_g = (Fe*)c->ecfft_ws.p;
    Fe* e_gl = e_g + n;
    XYZZ* tmp = (XYZZ*)(((uintptr_t)(e_gl + n) + 255) & ~(uintptr_t)255);
    const uint32_t grid = (uint32_t)((n + 255) / 256);
    int tid = c->timer_begin("kzg_setup", stream);
    hipLaunchKernelGGL(kzg_setup_scalars_kernel, dim3(grid), dim3(256), 0, stream, s, mult, root, n, e_g, e_gl);
    H2_CHECK(hipGetLastError());
    hipLaunchKernelGGL(kzg_setup_mul_kernel, dim3(grid), dim3(256), 0, stream, (const Fe*)e_g, table, n, tmp);
    H2_CHECK(hipGetLastError());
    if ((rc = ec_normalize_device(tmp, d_g, n, stream))) return rc;  // batch_normalize, :83-87
    hipLaunchKernelGGL(kzg_setup_mul_kernel, dim3(grid), dim3(256), 0, stream, (const Fe*)e_gl, table, n, tmp);
    H2_CHECK(hipGetLastError());
    if ((rc = ec_normalize_device(tmp, d_gl, n, stream))) return rc;  // :106-116
    c->timer_end(tid, stream);
    return guard.release();
}

}  // namespace h2
