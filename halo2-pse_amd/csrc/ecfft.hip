// ecfft.hip -- g_to_lagrange (halo2_proofs/src/arithmetic.rs:277-301): best_fft with G = bn256::G1 over the SRS points,
// then the 1/n scale and batch_normalize.  Called by ParamsKZG::downsize (poly/kzg/commitment.rs:267-275).
//
// The butterfly of a curve-point FFT is t = [w] b; (a, b) <- (a + t, a - t): one 254-bit scalar multiplication per
// butterfly, k/2 * n of them -- VALU work, a few hundred bytes per butterfly.  One lane per butterfly; layers follow
// the reference's iterative form (arithmetic.rs:202-230: bit-reversal, then chunks 2, 4, ... n with
// twiddles[i * twiddle_chunk]).  Between layers the points are brought back to affine by a batched inversion so that
// every addition inside the scalar ladder is a mixed addition (9.2 vs 14 multiplications):
//   ecfft_layer_kernel      t = [w] b by a GLV ladder (glv.h: w = w1 + w2 * LAMBDA, 2 + 2-bit joint windows) in the
//                           unsaturated XYZZ arithmetic of ecu.h (the MSM's), the two closing additions in canonical ec.h arithmetic; layer 0 applies the
//                           bit-reversal on its loads; butterflies with w = 1 skip the ladder (arithmetic.rs:255-260)
//   ec_normalize_kernel     XYZZ -> affine, Montgomery's trick over 8 points per lane
//   ec_scale_kernel         [n_inv] p for every point (wave-uniform scalar: no divergence), arithmetic.rs:286-290
// Only the group elements are defined by the reference (its Jacobian coordinates depend on the butterfly order); the
// affine output is canonical, so it is compared limb for limb.
#include <string.h>

#include "ecq.h"
#include "engine.h"
#include "glv.h"

namespace h2 {

struct Scalar256 {
    uint32_t w[8];  // canonical integer, little-endian
};

// acc = [k] p for an affine p (E-form) and a GLV-decomposed scalar k = k1 + k2 * LAMBDA: a joint ladder over |k1| and
// |k2| (< 2^130) with p1 = +-p and p2 = +-phi(p) = +-(BETA * x, y), fixed 2 + 2-bit windows, MSB first.  The table
// a * p1 + b * p2 (a, b < 4; 15 entries) lives in per-lane scratch as XYZZ points (the digits differ per lane); per window
// 2 doublings and one general addition: 130 doublings + 65 additions against 252 + 63 for 4-bit windows over the full
// scalar (and ~254 + 254 for a wave of lanes with unrelated scalars under plain double-and-add).
__device__ XYZZu ec_mul_affine_glv(const Affine& p, const GlvScalar& k) {
    if (affine_is_identity(p)) return xyzzu_identity();
    const uint32_t BETA_I[9] = {0x0a337995u, 0x158d1d23u, 0x189c9b98u, 0x12fa4e45u, 0x185faadcu, 0x0176f16du, 0x0eed93bau, 0x14291140u, 0x000c0afeu};
    const Fu x1 = fu_from_ext(p.x);
    Fu y1 = fu_from_ext(p.y), y2 = y1;
    const Fu x2 = fu_mul<QU>(x1, fu_const<QU>(BETA_I));  // I-form of BETA * x, in (-0.2 q, 1.2 q)
    if (k.neg1) y1 = fu_neg(y1);
    if (k.neg2) y2 = fu_neg(y2);
    XYZZu tab[16];  // tab[a + 4 b] = a * p1 + b * p2
    tab[0] = xyzzu_identity();
    tab[1] = xyzzu_identity();
    xyzzu_add_mixed<QU>(tab[1], x1, y1);
    tab[2] = xyzzu_double_affine<QU>(x1, y1);
    tab[3] = tab[2];
    xyzzu_add_mixed<QU>(tab[3], x1, y1);
    tab[4] = xyzzu_identity();
    xyzzu_add_mixed<QU>(tab[4], x2, y2);
    tab[8] = xyzzu_double_affine<QU>(x2, y2);
    tab[12] = tab[8];
    xyzzu_add_mixed<QU>(tab[12], x2, y2);
    for (int b = 1; b < 4; b++)
        for (int a = 1; a < 4; a++) {
            tab[a + 4 * b] = tab[a + 4 * (b - 1)];
            xyzzu_add_mixed<QU>(tab[a + 4 * b], x2, y2);
        }
    XYZZu acc = xyzzu_identity();
    for (int i = 64; i >= 0; i--) {
        acc = xyzzu_double(xyzzu_double(acc));
        const uint32_t d1 = (k.k1[i >> 4] >> ((i & 15) * 2)) & 3, d2 = (k.k2[i >> 4] >> ((i & 15) * 2)) & 3;
        const uint32_t d = d1 | (d2 << 2);
        if (d) xyzzu_add(acc, tab[d]);
    }
    return acc;
}

// the same for a wave-uniform scalar (the final [1/n]): plain double-and-add with mixed additions, no divergence
__device__ XYZZu ec_mul_affine_uniform(const Affine& p, const Scalar256& e) {
    XYZZu acc = xyzzu_identity();
    if (affine_is_identity(p)) return acc;
    const Fu px = fu_from_ext(p.x), py = fu_from_ext(p.y);
    for (int i = 253; i >= 0; i--) {
        acc = xyzzu_double(acc);
        if ((e.w[i >> 5] >> (i & 31)) & 1) xyzzu_add_mixed<QU>(acc, px, py);
    }
    return acc;
}

struct EcfftLayer {
    const Affine* in;
    XYZZ* out;
    const GlvScalar* tw;  // omega_inv^i, i < n / 2, decomposed
    uint32_t log_n, s;    // layer s: half = 2^s
};

__global__ void __launch_bounds__(256) ecfft_layer_kernel(EcfftLayer L) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n = 1ull << L.log_n;
    if (tid >= n / 2) return;
    const uint64_t half = 1ull << L.s;
    const uint64_t i = tid & (half - 1), blk = tid >> L.s;
    const uint64_t ia = (blk << (L.s + 1)) + i, ib = ia + half;
    uint64_t la = ia, lb = ib;
    if (L.s == 0) {  // best_fft's swap loop (arithmetic.rs:186-191), folded into the first layer's loads
        la = __brevll(ia) >> (64 - L.log_n);
        lb = __brevll(ib) >> (64 - L.log_n);
    }
    const Affine a = L.in[la], b = L.in[lb];
    XYZZ t;
    if (i == 0) {
        t = xyzz_from_affine(b);  // twiddle one
    } else {
        const GlvScalar w = L.tw[i << (L.log_n - 1 - L.s)];  // twiddles[i * twiddle_chunk], twiddle_chunk = n / 2^(s+1)
        t = xyzzu_to_ext(ec_mul_affine_glv(b, w));
    }
    XYZZ hi = t, lo = t;
    lo.y = fe_neg<FqP>(t.y);
    xyzz_add_mixed(hi, a);  // a + t
    xyzz_add_mixed(lo, a);  // a - t
    L.out[ia] = hi;
    L.out[ib] = lo;
}

// The same ladder with one QUAD of lanes per butterfly (ecq.h): up to k = 14 a layer is at most 2^13 butterflies -- a
// fraction of the chip's 65 536 SIMD lanes -- and the 130 doublings + 65 additions of a ladder are a pure latency chain.  The four
// lanes of a quad hold the same point and digits; the 15-entry table is built by every lane for itself (mixed additions).
__device__ XYZZu ec_mul_affine_glv_q(const Affine& p, const GlvScalar& k, uint32_t role) {
    if (affine_is_identity(p)) return xyzzu_identity();
    const uint32_t BETA_I[9] = {0x0a337995u, 0x158d1d23u, 0x189c9b98u, 0x12fa4e45u, 0x185faadcu, 0x0176f16du, 0x0eed93bau, 0x14291140u, 0x000c0afeu};
    const Fu x1 = fu_from_ext(p.x);
    Fu y1 = fu_from_ext(p.y), y2 = y1;
    const Fu x2 = fu_mul<QU>(x1, fu_const<QU>(BETA_I));
    if (k.neg1) y1 = fu_neg(y1);
    if (k.neg2) y2 = fu_neg(y2);
    XYZZu tab[16];
    tab[0] = xyzzu_identity();
    tab[1] = xyzzu_identity();
    xyzzu_add_mixed<QU>(tab[1], x1, y1);
    tab[2] = xyzzu_double_affine<QU>(x1, y1);
    tab[3] = tab[2];
    xyzzu_add_mixed<QU>(tab[3], x1, y1);
    tab[4] = xyzzu_identity();
    xyzzu_add_mixed<QU>(tab[4], x2, y2);
    tab[8] = xyzzu_double_affine<QU>(x2, y2);
    tab[12] = tab[8];
    xyzzu_add_mixed<QU>(tab[12], x2, y2);
    for (int b = 1; b < 4; b++)
        for (int a = 1; a < 4; a++) {
            tab[a + 4 * b] = tab[a + 4 * (b - 1)];
            xyzzu_add_mixed<QU>(tab[a + 4 * b], x2, y2);
        }
    XYZZu acc = xyzzu_identity();
    for (int i = 64; i >= 0; i--) {
        acc = xyzzu_double_q(xyzzu_double_q(acc, role), role);
        const uint32_t d1 = (k.k1[i >> 4] >> ((i & 15) * 2)) & 3, d2 = (k.k2[i >> 4] >> ((i & 15) * 2)) & 3;
        const uint32_t d = d1 | (d2 << 2);
        if (d) xyzzu_add_q(acc, tab[d], role);
    }
    return acc;
}

__global__ void __launch_bounds__(256) ecfft_layer_quad_kernel(EcfftLayer L) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t tid = gid >> 2;
    const uint32_t role = (uint32_t)gid & 3;
    const uint64_t n = 1ull << L.log_n;
    if (tid >= n / 2) return;
    const uint64_t half = 1ull << L.s;
    const uint64_t i = tid & (half - 1), blk = tid >> L.s;
    const uint64_t ia = (blk << (L.s + 1)) + i, ib = ia + half;
    uint64_t la = ia, lb = ib;
    if (L.s == 0) {
        la = __brevll(ia) >> (64 - L.log_n);
        lb = __brevll(ib) >> (64 - L.log_n);
    }
    const Affine a = L.in[la], b = L.in[lb];
    XYZZ t;
    if (i == 0) {
        t = xyzz_from_affine(b);
    } else {
        const GlvScalar w = L.tw[i << (L.log_n - 1 - L.s)];
        t = xyzzu_to_ext(ec_mul_affine_glv_q(b, w, role));
    }
    if (role != 0) return;  // the two closing additions are canonical arithmetic: one lane
    XYZZ hi = t, lo = t;
    lo.y = fe_neg<FqP>(t.y);
    xyzz_add_mixed(hi, a);
    xyzz_add_mixed(lo, a);
    L.out[ia] = hi;
    L.out[ib] = lo;
}

// Small transforms (k <= 14; round 4): the points stay XYZZ between the layers.  A layer of <= 2^13 butterflies is a latency chain on a
// fraction of the chip, and the batched normalisation that followed every layer -- one Fermat inversion per lane, ~380 dependent
// multiplications = 0.17 ms, plus a launch -- was a fifth of the call at k = 12 for the sake of mixed additions in a table of 15 entries.
// Here the ladder's base point is XYZZ: phi(p) = (BETA * X, Y, ZZ, ZZZ), the table is built with general additions (13 x 4.8
// multiplications more than with mixed ones, of ~4 000 per ladder), the butterflies run in place (each owns its two elements) and only the
// final layer's result is normalised.
__device__ XYZZu ec_mul_xyzz_glv_q(const XYZZu& p, const GlvScalar& k, uint32_t role) {
    if (xyzzu_is_identity(p)) return xyzzu_identity();
    const uint32_t BETA_I[9] = {0x0a337995u, 0x158d1d23u, 0x189c9b98u, 0x12fa4e45u, 0x185faadcu, 0x0176f16du, 0x0eed93bau, 0x14291140u, 0x000c0afeu};
    XYZZu p1 = p, p2 = p;
    p2.x = fu_mul<QU>(p.x, fu_const<QU>(BETA_I));
    if (k.neg1) p1.y = fu_norm(fu_neg(p1.y));
    if (k.neg2) p2.y = fu_norm(fu_neg(p2.y));
    XYZZu tab[16];  // tab[a + 4 b] = a * p1 + b * p2
    tab[0] = xyzzu_identity();
    tab[1] = p1;
    tab[2] = xyzzu_double_q(p1, role);
    tab[3] = xyzzu_sum_q(tab[2], p1, role);
    tab[4] = p2;
    tab[8] = xyzzu_double_q(p2, role);
    tab[12] = xyzzu_sum_q(tab[8], p2, role);
    for (int b = 1; b < 4; b++)
        for (int a = 1; a < 4; a++) tab[a + 4 * b] = xyzzu_sum_q(tab[a + 4 * (b - 1)], p2, role);
    XYZZu acc = xyzzu_identity();
    for (int i = 64; i >= 0; i--) {
        acc = xyzzu_double_q(xyzzu_double_q(acc, role), role);
        const uint32_t d1 = (k.k1[i >> 4] >> ((i & 15) * 2)) & 3, d2 = (k.k2[i >> 4] >> ((i & 15) * 2)) & 3;
        const uint32_t d = d1 | (d2 << 2);
        if (d) xyzzu_add_q(acc, tab[d], role);
    }
    return acc;
}

struct EcfftLayerX {
    const Affine* in;     // layer 0 reads the affine input (bit-reversed); later layers read buf
    XYZZ* buf;            // the points of the transform, in place
    const GlvScalar* tw;
    uint32_t log_n, s;
};

__global__ void __launch_bounds__(256) ecfft_layer_quad_x_kernel(EcfftLayerX L) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t tid = gid >> 2;
    const uint32_t role = (uint32_t)gid & 3;
    const uint64_t n = 1ull << L.log_n;
    if (tid >= n / 2) return;
    const uint64_t half = 1ull << L.s;
    const uint64_t i = tid & (half - 1), blk = tid >> L.s;
    const uint64_t ia = (blk << (L.s + 1)) + i, ib = ia + half;
    XYZZ a, b;
    if (L.s == 0) {  // best_fft's swap loop (arithmetic.rs:186-191), folded into the first layer's loads
        a = xyzz_from_affine(L.in[__brevll(ia) >> (64 - L.log_n)]);
        b = xyzz_from_affine(L.in[__brevll(ib) >> (64 - L.log_n)]);
    } else {
        a = L.buf[ia];
        b = L.buf[ib];
    }
    XYZZ t = b;  // twiddle one
    if (i != 0) t = xyzzu_to_ext(ec_mul_xyzz_glv_q(xyzzu_from_ext(b), L.tw[i << (L.log_n - 1 - L.s)], role));
    if (role != 0) return;  // the two closing additions are canonical arithmetic: one lane
    XYZZ hi = a, lo = a;
    xyzz_add(hi, t);  // a + t
    t.y = fe_neg<FqP>(t.y);
    xyzz_add(lo, t);  // a - t
    L.buf[ia] = hi;
    L.buf[ib] = lo;
}

// [e] p for XYZZ points (the final 1 / n of g_to_lagrange on the un-normalised path): the scalar is the same for every point, so it is
// decomposed once on the host and every point takes the quad GLV ladder of the butterflies -- 130 quad doublings + 65 quad additions
// (~0.5 ms of latency) where plain double-and-add on one lane per point walked 254 doublings + ~127 additions (~2 ms, a third of the call at k = 8)
__global__ void __launch_bounds__(256) ec_scale_xyzz_quad_kernel(XYZZ* buf, uint64_t n, GlvScalar e) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t tid = gid >> 2;
    const uint32_t role = (uint32_t)gid & 3;
    if (tid >= n) return;
    const XYZZ r = xyzzu_to_ext(ec_mul_xyzz_glv_q(xyzzu_from_ext(buf[tid]), e, role));
    if (role == 0) buf[tid] = r;
}

__global__ void __launch_bounds__(256) ec_scale_kernel(const Affine* in, XYZZ* out, uint64_t n, Scalar256 e) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n) return;
    out[tid] = xyzzu_to_ext(ec_mul_affine_uniform(in[tid], e));
}

#define EC_NORM_CHUNK 8
// lane t normalises points t, t + L, t + 2L, ... (L = lanes): one inversion per EC_NORM_CHUNK points
__global__ void __launch_bounds__(256) ec_normalize_kernel(const XYZZ* in, Affine* out, uint64_t n, uint64_t lanes) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= lanes) return;
    Fe prefix[EC_NORM_CHUNK];
    Fe acc = fe_one<FqP>();
#pragma unroll
    for (int j = 0; j < EC_NORM_CHUNK; j++) {
        const uint64_t idx = tid + (uint64_t)j * lanes;
        prefix[j] = acc;
        if (idx < n) {
            const Fe zzz = in[idx].zzz;
            if (!fe_is_zero(zzz)) acc = fe_mul<FqP>(acc, zzz);
        }
    }
    // the one inversion per lane on the unsaturated multiplier (E-form in: fu_from_ext gives the I-form limbs for free;
    // canonical E-form out through the exact reduction): 2.6x fewer instructions than fe_inv's saturated Fermat chain
    Fe inv = fu_mul_canon<FqU>(fu_inv<FqU>(fu_from_ext(acc)), fu_one_e<FqU>());
#pragma unroll
    for (int j = EC_NORM_CHUNK - 1; j >= 0; j--) {
        const uint64_t idx = tid + (uint64_t)j * lanes;
        if (idx >= n) continue;
        const XYZZ p = in[idx];
        Affine o;
        if (fe_is_zero(p.zzz)) {  // identity (zz == 0 <=> zzz == 0): (0, 0)
            o.x = fe_zero<FqP>();
            o.y = fe_zero<FqP>();
        } else {
            const Fe zzz_inv = fe_mul<FqP>(inv, prefix[j]);  // 1 / Z^3
            inv = fe_mul<FqP>(inv, p.zzz);
            const Fe zz_inv = fe_mul<FqP>(fe_sqr<FqP>(zzz_inv), fe_sqr<FqP>(p.zz));  // Z^-6 * Z^4
            o.x = fe_mul<FqP>(p.x, zz_inv);
            o.y = fe_mul<FqP>(p.y, zzz_inv);
        }
        out[idx] = o;
    }
}

__global__ void __launch_bounds__(256) ecfft_twiddle_kernel(GlvScalar* tw, uint64_t count, Fe omega_inv /* the transform's root: omega^-1 for g_to_lagrange */) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= count) return;
    const Fe c = fe_to_canonical<FrP>(fe_pow_u64<FrP>(omega_inv, tid));
    tw[tid] = glv_decompose(c.l);
}

static bool g_ecfft_quad = true, g_ecfft_lazy = true;
void ecfft_set_quad(bool on) { g_ecfft_quad = on; }
void ecfft_set_lazy(bool on) { g_ecfft_lazy = on; }  // k <= 14: no normalisation between the layers (round 4); off: round 3's path

static int normalize_launch(const XYZZ* in, Affine* out, uint64_t n, hipStream_t s) {
    const uint64_t lanes = (n + EC_NORM_CHUNK - 1) / EC_NORM_CHUNK;
    hipLaunchKernelGGL(ec_normalize_kernel, dim3((uint32_t)((lanes + 255) / 256)), dim3(256), 0, s, in, out, n, lanes);
    H2_CHECK(hipGetLastError());
    return 0;
}

int ec_normalize_device(const XYZZ* d_in, Affine* d_out, uint64_t n, hipStream_t s) { return normalize_launch(d_in, d_out, n, s); }

// Jacobian (x, y, z) -> XYZZ (x, y, z^2, z^3): the form ec_normalize_kernel takes; identity (z = 0) -> zz = zzz = 0
__global__ void __launch_bounds__(256) jac_to_xyzz_kernel(const Jac* in, XYZZ* out, uint64_t n) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n) return;
    const Jac p = in[tid];
    XYZZ o;
    o.x = p.x;
    o.y = p.y;
    o.zz = fe_sqr<FqP>(p.z);
    o.zzz = fe_mul<FqP>(o.zz, p.z);
    out[tid] = o;
}

// affine -> Jacobian with z = 1; identity (0, 0) -> (0, 1, 0), halo2curves' G1::identity()
__global__ void __launch_bounds__(256) affine_to_jac_kernel(const Affine* in, Jac* out, uint64_t n) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n) return;
    const Affine a = in[tid];
    Jac o;
    if (affine_is_identity(a)) {
        o.x = fe_zero<FqP>();
        o.y = fe_one<FqP>();
        o.z = fe_zero<FqP>();
    } else {
        o.x = a.x;
        o.y = a.y;
        o.z = fe_one<FqP>();
    }
    out[tid] = o;
}

// best_fft with G = G1 (arithmetic.rs:171-234): the layers of the curve-point FFT with twiddles omega^i, then -- scale != nullptr --
// every point times *scale (g_to_lagrange's 1 / n).  d_in: n affine points (read only), d_out: n affine points.  Queued on s.
static int ecfft_device(Ctx* c, const Affine* d_in, uint32_t k, Affine* d_out, const Fe& omega, const Fe* scale, hipStream_t s) {
    const uint64_t n = 1ull << k;
    // workspace: XYZZ[n] | twiddles[n / 2]; the affine points of the current layer live in d_out
    const size_t xyzz_bytes = n * sizeof(XYZZ), tw_bytes = (n / 2 + 1) * sizeof(GlvScalar);
    int rc = c->ecfft_ws.ensure(xyzz_bytes + tw_bytes + 256);
    if (rc) return rc;
    if ((rc = c->ws_acquire(s))) return rc;
    WsGuard guard(c, s);
    XYZZ* d_xyzz = (XYZZ*)c->ecfft_ws.p;
    GlvScalar* d_tw = (GlvScalar*)((char*)c->ecfft_ws.p + ((xyzz_bytes + 255) / 256) * 256);
    int tid = c->timer_begin("g_to_lagrange", s);
    if (n >= 2) {
        hipLaunchKernelGGL(ecfft_twiddle_kernel, dim3((uint32_t)((n / 2 + 255) / 256)), dim3(256), 0, s, d_tw, n / 2, omega);
        H2_CHECK(hipGetLastError());
    }
    const Affine* src = d_in;
    if (g_ecfft_quad && g_ecfft_lazy && k >= 1 && k <= 14) {
        // small transforms: the points stay XYZZ across the layers (in place), one normalisation at the very end
        for (uint32_t layer = 0; layer < k; layer++) {
            EcfftLayerX L = {d_in, d_xyzz, d_tw, k, layer};
            hipLaunchKernelGGL(ecfft_layer_quad_x_kernel, dim3((uint32_t)((2 * n + 255) / 256)), dim3(256), 0, s, L);
            H2_CHECK(hipGetLastError());
        }
        if (scale) {
            const Fe sc = fe_to_canonical<FrP>(*scale);
            const GlvScalar e = glv_decompose(sc.l);
            hipLaunchKernelGGL(ec_scale_xyzz_quad_kernel, dim3((uint32_t)((4 * n + 255) / 256)), dim3(256), 0, s, d_xyzz, n, e);
            H2_CHECK(hipGetLastError());
        }
        if ((rc = normalize_launch(d_xyzz, d_out, n, s))) return rc;
        c->timer_end(tid, s);
        return guard.release();
    }
    for (uint32_t layer = 0; layer < k; layer++) {
        EcfftLayer L = {src, d_xyzz, d_tw, k, layer};
        if (g_ecfft_quad && k <= 14)  // few butterflies per layer (<= 2^13: fewer lanes than SIMD slots even four to a butterfly)
            hipLaunchKernelGGL(ecfft_layer_quad_kernel, dim3((uint32_t)((2 * n + 255) / 256)), dim3(256), 0, s, L);
        else
            hipLaunchKernelGGL(ecfft_layer_kernel, dim3((uint32_t)((n / 2 + 255) / 256)), dim3(256), 0, s, L);
        H2_CHECK(hipGetLastError());
        if ((rc = normalize_launch(d_xyzz, d_out, n, s))) return rc;
        src = d_out;
    }
    if (scale) {
        const Fe sc = fe_to_canonical<FrP>(*scale);
        Scalar256 e;
        for (int i = 0; i < 8; i++) e.w[i] = sc.l[i];
        hipLaunchKernelGGL(ec_scale_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, src, d_xyzz, n, e);
        H2_CHECK(hipGetLastError());
        if ((rc = normalize_launch(d_xyzz, d_out, n, s))) return rc;
    } else if (k == 0 && d_out != d_in) {
        H2_CHECK(hipMemcpyAsync(d_out, d_in, sizeof(Affine), hipMemcpyDeviceToDevice, s));
    }
    c->timer_end(tid, s);
    return guard.release();
}

// d_g: n affine points in (read only); d_out: n affine points out.  Queued on s; does not wait.
int g_to_lagrange_device(Ctx* c, const Affine* d_g, uint32_t k, Affine* d_out, hipStream_t s) {
    if (k > FrP::S) {
        set_error("g_to_lagrange: k = %u exceeds the 2-adicity of Fr", k);
        return 1;
    }
    // arithmetic.rs:278-282
    Fe omega_inv;
    memcpy(omega_inv.l, FrP::ROOT_OF_UNITY_INV, sizeof(omega_inv.l));  // Montgomery form, as stored
    for (uint32_t i = k; i < FrP::S; i++) omega_inv = fe_sqr<FrP>(omega_inv);
    const Fe two_inv = fe_inv<FrP>(fe_from_u64<FrP>(2));
    const Fe n_inv = fe_pow_u64<FrP>(two_inv, k);
    return ecfft_device(c, d_g, k, d_out, omega_inv, &n_inv, s);
}

// best_fft::<G1>(a, omega, log_n) on device-resident Jacobian points, in place: normalise to affine (batched inversion), the FFT
// layers, back to Jacobian with z = 1 (identity: z = 0).  Only the group elements are defined by the reference -- its Jacobian
// coordinates depend on the butterfly order and the thread count.  d_tmp: 2^log_n affine points of scratch.
int fft_g1_device(Ctx* c, Jac* d_a, const Fe& omega, uint32_t log_n, hipStream_t s) {
    if (log_n > FrP::S) {
        set_error("fft_g1: log_n = %u exceeds the 2-adicity of Fr", log_n);
        return 1;
    }
    const uint64_t n = 1ull << log_n;
    int rc = c->misc.ensure(2 * n * sizeof(Affine) + 256);
    if (rc) return rc;
    Affine* d_in = (Affine*)c->misc.p;
    Affine* d_out = d_in + n;
    {   // Jacobian -> affine through the FFT's own XYZZ workspace
        if ((rc = c->ecfft_ws.ensure(n * sizeof(XYZZ) + (n / 2 + 1) * sizeof(GlvScalar) + 256))) return rc;
        if ((rc = c->ws_acquire(s))) return rc;
        WsGuard guard(c, s);
        hipLaunchKernelGGL(jac_to_xyzz_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, (const Jac*)d_a, (XYZZ*)c->ecfft_ws.p, n);
        H2_CHECK(hipGetLastError());
        if ((rc = normalize_launch((const XYZZ*)c->ecfft_ws.p, d_in, n, s))) return rc;
        if ((rc = guard.release())) return rc;
    }
    if ((rc = ecfft_device(c, d_in, log_n, d_out, omega, nullptr, s))) return rc;
    hipLaunchKernelGGL(affine_to_jac_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, (const Affine*)d_out, d_a, n);
    H2_CHECK(hipGetLastError());
    return 0;
}

}  // namespace h2
