// host64.h -- host-only Fq / G1 arithmetic on 4 x 64-bit limbs (unsigned __int128), same byte layout as
// field.h's Fe.  Used for the few hundred group operations that finish an MSM on the CPU: the Horner
// over window sums (arithmetic.rs:46-49) and the fold of partials (arithmetic.rs:153).  field.h's
// portable 8 x 32 code is ~4x slower on x86-64 and made that tail ~0.2 ms per MSM.
#pragma once
#include <stdint.h>
#include <string.h>

#include "ec.h"

namespace h2 {
namespace h64 {

typedef unsigned __int128 u128;
struct F {
    uint64_t l[4];
};
static const uint64_t QMOD[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const uint64_t QINV = 0x87d20782e4866389ULL;  // -q^-1 mod 2^64

static inline bool is_zero(const F& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
static inline bool geq(const uint64_t a[4]) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > QMOD[i]) return true;
        if (a[i] < QMOD[i]) return false;
    }
    return true;
}
static inline void subq(uint64_t a[4]) {
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - QMOD[i] - (uint64_t)br;
        a[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
}
static inline F add(const F& a, const F& b) {
    F o;
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)a.l[i] + b.l[i];
        o.l[i] = (uint64_t)c;
        c >>= 64;
    }
    if (geq(o.l)) subq(o.l);
    return o;
}
static inline F sub(const F& a, const F& b) {
    F o;
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a.l[i] - b.l[i] - (uint64_t)br;
        o.l[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
    if (br) {
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)o.l[i] + QMOD[i];
            o.l[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    return o;
}
static inline F dbl(const F& a) { return add(a, a); }
static inline F mul(const F& a, const F& b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a.l[j] * b.l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * QINV;
        c = (u128)m * QMOD[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * QMOD[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    if (t[4] || geq(t)) subq(t);
    F o;
    memcpy(o.l, t, 32);
    return o;
}
static inline F sqr(const F& a) { return mul(a, a); }

struct P {  // XYZZ, E-form (Montgomery R = 2^256) coordinates, identity zz = 0
    F x, y, zz, zzz;
};
static inline P from_xyzz(const XYZZ& p) {
    P o;
    memcpy(&o, &p, sizeof(P));
    return o;
}
static inline XYZZ to_xyzz(const P& p) {
    XYZZ o;
    memcpy(&o, &p, sizeof(P));
    return o;
}
static inline P identity() {
    P o;
    memset(&o, 0, sizeof(o));
    return o;
}
static inline P pdouble(const P& p) {  // dbl-2008-s-1
    if (is_zero(p.zz)) return p;
    P o;
    F u = dbl(p.y), v = sqr(u), w = mul(u, v), s = mul(p.x, v), xx = sqr(p.x), m = add(dbl(xx), xx);
    o.x = sub(sqr(m), dbl(s));
    o.y = sub(mul(m, sub(s, o.x)), mul(w, p.y));
    o.zz = mul(v, p.zz);
    o.zzz = mul(w, p.zzz);
    return o;
}
static inline void padd(P& a, const P& b) {  // add-2008-s with the exceptional cases
    if (is_zero(b.zz)) return;
    if (is_zero(a.zz)) {
        a = b;
        return;
    }
    F u1 = mul(a.x, b.zz), u2 = mul(b.x, a.zz), s1 = mul(a.y, b.zzz), s2 = mul(b.y, a.zzz);
    F pp_ = sub(u2, u1), r = sub(s2, s1);
    if (is_zero(pp_)) {
        if (is_zero(r)) a = pdouble(a); else a = identity();
        return;
    }
    F pp = sqr(pp_), ppp = mul(pp_, pp), q = mul(u1, pp);
    F x3 = sub(sub(sqr(r), ppp), dbl(q));
    F y3 = sub(mul(r, sub(q, x3)), mul(s1, ppp));
    a.x = x3;
    a.y = y3;
    a.zz = mul(mul(a.zz, b.zz), pp);
    a.zzz = mul(mul(a.zzz, b.zzz), ppp);
}

}  // namespace h64
}  // namespace h2
