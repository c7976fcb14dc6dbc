// ecu.h -- BN254 G1 in XYZZ coordinates over the unsaturated field of fieldu.h (I-form
// values, lazily reduced).  Same group law and the same exceptional-case handling as ec.h
// (which stays the host-side / reference implementation and the thing this file is fuzzed
// against in tests/cpp/test_fieldu.cpp); this is what the MSM kernels run.
//
// Invariants of an XYZZu accumulator: every coordinate is "normalised" (limbs 0..7 in [0, 2^29),
// small signed top limb) with |x|, |y| < 4.5 p and zz, zzz in (-0.1 p, 1.4 p); the identity is
// marked by zz having all limbs exactly zero (a valid point never has zz == 0 as an integer).
// Bound bookkeeping (units of p; a Montgomery product of magnitudes a, b lies in
// (-ab/169, ab/169 + 1)): mixed add with a freshly loaded point (|32 x| < 32):
//   U2, S2 in (-0.25, 1.25); |P|, |R| < 5.8; PP, RR < 1.4; |PPP|, |Q| < 1.1; |X3| < 4.5; |Y3| < 2.3.
// Limb bookkeeping: products always see one operand with |l| < 2^29 and the other < 2^30.
#pragma once
#include "ec.h"
#include "fieldu.h"

namespace h2 {

typedef FqU QU;

struct XYZZu {
    Fu x, y, zz, zzz;
};

H2_HD XYZZu xyzzu_identity() {
    XYZZu o;
    o.x = fu_zero();
    o.y = fu_zero();
    o.zz = fu_zero();
    o.zzz = fu_zero();
    return o;
}

H2_HD bool xyzzu_is_identity(const XYZZu& p) { return fu_all_zero(p.zz); }

// The three functions of the bucket-accumulation inner loop take the field flavour as a template parameter: the
// accumulate kernel instantiates them with FqUA (explicit-mad multiplier), everything else with QU.

// 2 * (px, py) for a non-identity affine point in I-form (mdbl-2008-s-1)
template <class QU = FqU>
H2_HD XYZZu xyzzu_double_affine(const Fu& px, const Fu& py) {
    XYZZu o;
    const Fu one = fu_one_i<QU>();
    Fu x = fu_mul<QU>(px, one), y = fu_mul<QU>(py, one);  // 32x -> (-0.2 p, 1.2 p), same residue
    Fu u = fu_norm(fu_dbl(y));
    Fu v = fu_sqr<QU>(u);
    Fu w = fu_mul<QU>(u, v);
    Fu s = fu_mul<QU>(x, v);
    Fu xx = fu_sqr<QU>(x);
    Fu m = fu_norm(fu_add(fu_dbl(xx), xx));
    o.x = fu_sqr_sub<QU>(m, fu_dbl(s));                       // M^2 - 2S, one reduction, normalised
    o.y = fu_mul_sub<QU>(m, fu_sub(s, o.x), w, y);            // M*(S - X3) - W*Y1, one reduction
    o.zz = v;
    o.zzz = w;
    return o;
}

// dbl-2008-s-1
template <class QU = FqU>
H2_HD XYZZu xyzzu_double(const XYZZu& p) {
    if (xyzzu_is_identity(p)) return p;
    XYZZu o;
    Fu u = fu_norm(fu_dbl(p.y));
    Fu v = fu_sqr<QU>(u);
    Fu w = fu_mul<QU>(u, v);
    Fu s = fu_mul<QU>(p.x, v);
    Fu xx = fu_sqr<QU>(p.x);
    Fu m = fu_norm(fu_add(fu_dbl(xx), xx));
    o.x = fu_sqr_sub<QU>(m, fu_dbl(s));
    o.y = fu_mul_sub<QU>(m, fu_sub(s, o.x), w, p.y);
    o.zz = fu_mul<QU>(v, p.zz);
    o.zzz = fu_mul<QU>(w, p.zzz);
    return o;
}

// acc += (px, py): madd-2008-s.  (px, py) is a non-identity affine point in I-form with limbs of
// magnitude < 2^29 (fu_from_ext output, possibly negated).  All exceptional cases of the group
// law are exact: the cheap residue filter on P sends possible hits to an exact reduction.
template <class QU = FqU>
H2_HD void xyzzu_add_mixed(XYZZu& acc, const Fu& px, const Fu& py) {
    if (xyzzu_is_identity(acc)) {
        // first point of a bucket: bring 32x, 32y (|.| < 32 p) inside the accumulator bounds
        const Fu one = fu_one_i<QU>();
        acc.x = fu_mul<QU>(px, one);
        acc.y = fu_mul<QU>(py, one);
        acc.zz = one;
        acc.zzz = one;
        return;
    }
    Fu u2 = fu_mul<QU>(px, acc.zz);
    Fu s2 = fu_mul<QU>(py, acc.zzz);
    Fu p_ = fu_sub(u2, acc.x);
    Fu r = fu_sub(s2, acc.y);
    if (fu_maybe_zero_mod_p<QU>(p_)) {
        if (fu_is_zero_mod_p<QU>(p_)) {
            if (fu_is_zero_mod_p<QU>(r)) {
                acc = xyzzu_double_affine<QU>(px, py);
            } else {
                acc = xyzzu_identity();
            }
            return;
        }
    }
    Fu pp = fu_sqr<QU>(p_);
    Fu ppp = fu_mul<QU>(p_, pp);
    Fu q = fu_mul<QU>(acc.x, pp);
    Fu x3 = fu_sqr_sub<QU>(r, fu_add(ppp, fu_dbl(q)));    // R^2 - PPP - 2Q, one reduction, already normalised
    Fu t = fu_sub(q, x3);
    Fu y3 = fu_mul_sub<QU>(r, t, acc.y, ppp);             // R*(Q - X3) - Y1*PPP, one reduction
    acc.x = x3;
    acc.y = y3;
    acc.zz = fu_mul<QU>(acc.zz, pp);
    acc.zzz = fu_mul<QU>(acc.zzz, ppp);
}

// a += b: add-2008-s with exceptional cases
template <class QU = FqU>
H2_HD void xyzzu_add(XYZZu& a, const XYZZu& b) {
    if (xyzzu_is_identity(b)) return;
    if (xyzzu_is_identity(a)) {
        a = b;
        return;
    }
    Fu u1 = fu_mul<QU>(a.x, b.zz);
    Fu u2 = fu_mul<QU>(b.x, a.zz);
    Fu s1 = fu_mul<QU>(a.y, b.zzz);
    Fu s2 = fu_mul<QU>(b.y, a.zzz);
    Fu p_ = fu_sub(u2, u1);
    Fu r = fu_sub(s2, s1);
    if (fu_maybe_zero_mod_p<QU>(p_)) {
        if (fu_is_zero_mod_p<QU>(p_)) {
            if (fu_is_zero_mod_p<QU>(r)) {
                a = xyzzu_double<QU>(a);
            } else {
                a = xyzzu_identity();
            }
            return;
        }
    }
    Fu pp = fu_sqr<QU>(p_);
    Fu ppp = fu_mul<QU>(p_, pp);
    Fu q = fu_mul<QU>(u1, pp);
    Fu x3 = fu_sqr_sub<QU>(r, fu_add(ppp, fu_dbl(q)));
    Fu t = fu_sub(q, x3);
    Fu y3 = fu_mul_sub<QU>(r, t, s1, ppp);
    a.x = x3;
    a.y = y3;
    a.zz = fu_mul<QU>(fu_mul<QU>(a.zz, b.zz), pp);
    a.zzz = fu_mul<QU>(fu_mul<QU>(a.zzz, b.zzz), ppp);
}

// acc += p for an affine point in the reference's layout (E-form Fe, identity = (0,0)), optionally negated
template <class QU = FqU>
H2_HD void xyzzu_add_affine(XYZZu& acc, const Affine& p, bool negate) {
    if (affine_is_identity(p)) return;
    Fu px = fu_from_ext(p.x);
    Fu py = fu_from_ext(p.y);
    if (negate) py = fu_neg(py);
    xyzzu_add_mixed<QU>(acc, px, py);
}

// I-form accumulator -> the E-form XYZZ of ec.h with canonical coordinates
H2_HD XYZZ xyzzu_to_ext(const XYZZu& p) {
    if (xyzzu_is_identity(p)) return xyzz_identity();
    XYZZ o;
    const Fu one_e = fu_one_e<QU>();
    o.x = fu_mul_canon<QU>(p.x, one_e);
    o.y = fu_mul_canon<QU>(p.y, one_e);
    o.zz = fu_mul_canon<QU>(p.zz, one_e);
    o.zzz = fu_mul_canon<QU>(p.zzz, one_e);
    return o;
}

H2_HD XYZZu xyzzu_from_ext(const XYZZ& p) {
    XYZZu o;
    if (xyzz_is_identity(p)) return xyzzu_identity();
    const Fu one = fu_one_i<QU>();
    o.x = fu_mul<QU>(fu_from_ext(p.x), one);
    o.y = fu_mul<QU>(fu_from_ext(p.y), one);
    o.zz = fu_mul<QU>(fu_from_ext(p.zz), one);
    o.zzz = fu_mul<QU>(fu_from_ext(p.zzz), one);
    return o;
}

// k * p for a small non-negative integer k < 2^nbits (double-and-add, vartime)
H2_HD XYZZu xyzzu_mul_small(const XYZZu& p, uint32_t k, uint32_t nbits = 32) {
    XYZZu acc = xyzzu_identity();
    for (int i = (int)nbits - 1; i >= 0; i--) {
        acc = xyzzu_double(acc);
        if ((k >> i) & 1) xyzzu_add(acc, p);
    }
    return acc;
}

// k * p for k < 32 (5-bit double-and-add)
H2_HD XYZZu xyzzu_mul_small5(const XYZZu& p, uint32_t k) {
    XYZZu acc = xyzzu_identity();
    for (int i = 4; i >= 0; i--) {
        acc = xyzzu_double(acc);
        if ((k >> i) & 1) xyzzu_add(acc, p);
    }
    return acc;
}

}  // namespace h2
