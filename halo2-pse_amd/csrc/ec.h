// ec.h -- BN254 G1 (y^2 = x^3 + 3 over Fq) group law for gfx950 and the library's host side.
//
// Replaces halo2curves 0.3.1 bn256::{G1Affine, G1} as used by multiexp_serial
// (halo2_proofs/src/arithmetic.rs:48 double, :62-65 affine+affine / mixed add, :74-77, :98 add,
// :153 fold).  Boundary layouts are the reference's: G1Affine = x||y (64 B, identity = (0,0)),
// G1 = Jacobian x||y||z (96 B, identity z = 0).  Internally buckets use extended Jacobian
// "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; identity ZZ = 0): a mixed add costs
// 8M+2S against 7M+4S for Jacobian and needs no per-add field doublings.  Any correct formulas
// give the same group element; parity is on the affine value.
#pragma once
#include "field.h"

namespace h2 {

struct alignas(16) Affine {
    Fe x, y;
};
struct alignas(16) Jac {
    Fe x, y, z;
};
struct alignas(16) XYZZ {
    Fe x, y, zz, zzz;
};

typedef FqP Q;

H2_HD bool affine_is_identity(const Affine& p) { return fe_is_zero(p.x) && fe_is_zero(p.y); }
H2_HD bool xyzz_is_identity(const XYZZ& p) { return fe_is_zero(p.zz); }

H2_HD XYZZ xyzz_identity() {
    XYZZ o;
    o.x = fe_zero<Q>();
    o.y = fe_zero<Q>();
    o.zz = fe_zero<Q>();
    o.zzz = fe_zero<Q>();
    return o;
}

H2_HD XYZZ xyzz_from_affine(const Affine& p) {
    XYZZ o;
    if (affine_is_identity(p)) return xyzz_identity();
    o.x = p.x;
    o.y = p.y;
    o.zz = fe_one<Q>();
    o.zzz = fe_one<Q>();
    return o;
}

H2_HD Affine affine_neg(const Affine& p) {
    Affine o;
    o.x = p.x;
    o.y = fe_neg<Q>(p.y);  // (0,0) stays (0,0)
    return o;
}

// 2*(x,y) for an affine non-identity point with y != 0 (always true on BN254 G1: no 2-torsion):
// mdbl-2008-s-1
H2_HD XYZZ xyzz_double_affine(const Affine& p) {
    XYZZ o;
    Fe u = fe_dbl<Q>(p.y);
    Fe v = fe_sqr<Q>(u);
    Fe w = fe_mul<Q>(u, v);
    Fe s = fe_mul<Q>(p.x, v);
    Fe xx = fe_sqr<Q>(p.x);
    Fe m = fe_add<Q>(fe_dbl<Q>(xx), xx);
    o.x = fe_sub<Q>(fe_sqr<Q>(m), fe_dbl<Q>(s));
    o.y = fe_sub<Q>(fe_mul<Q>(m, fe_sub<Q>(s, o.x)), fe_mul<Q>(w, p.y));
    o.zz = v;
    o.zzz = w;
    return o;
}

// dbl-2008-s-1
H2_HD XYZZ xyzz_double(const XYZZ& p) {
    if (xyzz_is_identity(p)) return p;
    XYZZ o;
    Fe u = fe_dbl<Q>(p.y);
    Fe v = fe_sqr<Q>(u);
    Fe w = fe_mul<Q>(u, v);
    Fe s = fe_mul<Q>(p.x, v);
    Fe xx = fe_sqr<Q>(p.x);
    Fe m = fe_add<Q>(fe_dbl<Q>(xx), xx);
    o.x = fe_sub<Q>(fe_sqr<Q>(m), fe_dbl<Q>(s));
    o.y = fe_sub<Q>(fe_mul<Q>(m, fe_sub<Q>(s, o.x)), fe_mul<Q>(w, p.y));
    o.zz = fe_mul<Q>(v, p.zz);
    o.zzz = fe_mul<Q>(w, p.zzz);
    return o;
}

// acc += p (p affine): madd-2008-s, with every exceptional case of the group law handled:
// p identity, acc identity, p == acc (doubling), p == -acc (identity).  This is the Bucket
// state machine of arithmetic.rs:58-68 collapsed into one accumulator type.
H2_HD void xyzz_add_mixed(XYZZ& acc, const Affine& p) {
    if (affine_is_identity(p)) return;
    if (xyzz_is_identity(acc)) {
        acc.x = p.x;
        acc.y = p.y;
        acc.zz = fe_one<Q>();
        acc.zzz = fe_one<Q>();
        return;
    }
    Fe u2 = fe_mul<Q>(p.x, acc.zz);
    Fe s2 = fe_mul<Q>(p.y, acc.zzz);
    Fe pp_ = fe_sub<Q>(u2, acc.x);
    Fe r = fe_sub<Q>(s2, acc.y);
    if (fe_is_zero(pp_)) {
        if (fe_is_zero(r)) {
            acc = xyzz_double_affine(p);
        } else {
            acc = xyzz_identity();
        }
        return;
    }
    Fe pp = fe_sqr<Q>(pp_);
    Fe ppp = fe_mul<Q>(pp_, pp);
    Fe q = fe_mul<Q>(acc.x, pp);
    Fe x3 = fe_sub<Q>(fe_sub<Q>(fe_sqr<Q>(r), ppp), fe_dbl<Q>(q));
    Fe y3 = fe_sub<Q>(fe_mul<Q>(r, fe_sub<Q>(q, x3)), fe_mul<Q>(acc.y, ppp));
    acc.x = x3;
    acc.y = y3;
    acc.zz = fe_mul<Q>(acc.zz, pp);
    acc.zzz = fe_mul<Q>(acc.zzz, ppp);
}

// a += b: add-2008-s with exceptional cases
H2_HD void xyzz_add(XYZZ& a, const XYZZ& b) {
    if (xyzz_is_identity(b)) return;
    if (xyzz_is_identity(a)) {
        a = b;
        return;
    }
    Fe u1 = fe_mul<Q>(a.x, b.zz);
    Fe u2 = fe_mul<Q>(b.x, a.zz);
    Fe s1 = fe_mul<Q>(a.y, b.zzz);
    Fe s2 = fe_mul<Q>(b.y, a.zzz);
    Fe pp_ = fe_sub<Q>(u2, u1);
    Fe r = fe_sub<Q>(s2, s1);
    if (fe_is_zero(pp_)) {
        if (fe_is_zero(r)) {
            a = xyzz_double(a);
        } else {
            a = xyzz_identity();
        }
        return;
    }
    Fe pp = fe_sqr<Q>(pp_);
    Fe ppp = fe_mul<Q>(pp_, pp);
    Fe q = fe_mul<Q>(u1, pp);
    Fe x3 = fe_sub<Q>(fe_sub<Q>(fe_sqr<Q>(r), ppp), fe_dbl<Q>(q));
    Fe y3 = fe_sub<Q>(fe_mul<Q>(r, fe_sub<Q>(q, x3)), fe_mul<Q>(s1, ppp));
    a.x = x3;
    a.y = y3;
    a.zz = fe_mul<Q>(fe_mul<Q>(a.zz, b.zz), pp);
    a.zzz = fe_mul<Q>(fe_mul<Q>(a.zzz, b.zzz), ppp);
}

// XYZZ -> Jacobian without inversion: Z = ZZ*ZZZ, X' = X*ZZ*ZZZ^2, Y' = Y*ZZ^3*ZZZ^2
// (x = X'/Z^2, y = Y'/Z^3).  Identity -> (0, 1, 0) as halo2curves' G1::identity().
H2_HD Jac xyzz_to_jac(const XYZZ& p) {
    Jac o;
    if (xyzz_is_identity(p)) {
        o.x = fe_zero<Q>();
        o.y = fe_one<Q>();
        o.z = fe_zero<Q>();
        return o;
    }
    Fe zzz2 = fe_sqr<Q>(p.zzz);
    Fe t = fe_mul<Q>(p.zz, zzz2);            // ZZ*ZZZ^2
    o.x = fe_mul<Q>(p.x, t);
    Fe zz2 = fe_sqr<Q>(p.zz);
    o.y = fe_mul<Q>(p.y, fe_mul<Q>(zz2, t));  // Y*ZZ^3*ZZZ^2
    o.z = fe_mul<Q>(p.zz, p.zzz);
    return o;
}

H2_HD XYZZ jac_to_xyzz(const Jac& p) {
    XYZZ o;
    if (fe_is_zero(p.z)) return xyzz_identity();
    o.x = p.x;
    o.y = p.y;
    o.zz = fe_sqr<Q>(p.z);
    o.zzz = fe_mul<Q>(o.zz, p.z);
    return o;
}

// to affine; identity -> (0,0)  (Curve::to_affine)
H2_HD Affine xyzz_to_affine(const XYZZ& p) {
    Affine o;
    if (xyzz_is_identity(p)) {
        o.x = fe_zero<Q>();
        o.y = fe_zero<Q>();
        return o;
    }
    Fe zi3 = fe_inv<Q>(p.zzz);                      // ZZZ^-1
    Fe zi2 = fe_sqr<Q>(fe_mul<Q>(zi3, p.zz));       // (ZZ/ZZZ)^2 = ZZ^-1  (ZZ^3 = ZZZ^2)
    o.x = fe_mul<Q>(p.x, zi2);
    o.y = fe_mul<Q>(p.y, zi3);
    return o;
}

H2_HD bool affine_on_curve(const Affine& p) {
    if (affine_is_identity(p)) return true;
    Fe y2 = fe_sqr<Q>(p.y);
    Fe x3 = fe_mul<Q>(fe_sqr<Q>(p.x), p.x);
    Fe b = fe_from_u64<Q>(3);
    return fe_eq(y2, fe_add<Q>(x3, b));
}

// k * p for a small non-negative integer k (double-and-add, vartime)
H2_HD XYZZ xyzz_mul_small(const XYZZ& p, uint32_t k) {
    XYZZ acc = xyzz_identity();
    for (int i = 31; i >= 0; i--) {
        acc = xyzz_double(acc);
        if ((k >> i) & 1) xyzz_add(acc, p);
    }
    return acc;
}

}  // namespace h2
