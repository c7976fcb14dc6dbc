// ecq.h -- BN254 G1 group operations spread over the four lanes of a quad (device only).
//
// The tail of a bucket reduction is a chain of a few dozen DEPENDENT group operations on a handful of points; a lone
// wave issues one instruction per ~4.6 cycles whatever its lane count, so such a chain costs ~3.5 k instructions = 8 us
// per general addition.  Here the four lanes of a quad hold the SAME point(s) and share one operation: an
// XYZZ addition is 14 field products in four levels of at most four independent products (4 + 4 + 3 + 3), a doubling 9 in
// three levels (2 + 4 + 3); at each level lane r of the quad computes product r and the results go back to all four
// lanes with quad-permute DPP moves (9 limbs x <= 4 moves).  ~1.3 k instructions per addition instead of ~3.5 k, at
// four times the lanes -- for phases that have lanes to spare and none for phases that fill the chip.
//
// Same formulas, same exceptional-case handling and the same value bounds as ecu.h (every coordinate normalised, |x|,
// |y| < 4.5 p, zz, zzz in (-0.1 p, 1.4 p)); the products are plain fu_mul (no fused two-product reductions), so X3 and Y3
// are carried through fu_norm where ecu.h gets a normalised value from the fused form.  All lanes of a quad take the
// same branches because they hold the same data.
#pragma once
#include "ecu.h"

namespace h2 {

template <int K>
__device__ __forceinline__ int32_t quad_lane(int32_t v) {  // the value lane K of this lane's quad holds
    return __builtin_amdgcn_update_dpp(0, v, K * 0x55, 0xF, 0xF, true);
}

template <int K>
__device__ __forceinline__ Fu quad_lane_fu(const Fu& x) {
    Fu o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.l[i] = quad_lane<K>(x.l[i]);
    return o;
}

// Lane r of a quad takes a_r.  Written as explicit v_cndmask on three lane masks: LLVM turns the equivalent compare-and-select chain
// into a scratch array indexed by the role (36 stores and 9 indexed loads per pick, in the middle of a latency chain).
struct QuadMasks {
    uint64_t m1, m2, m3;  // lanes with role 1, 2, 3
};
__device__ __forceinline__ QuadMasks quad_masks(uint32_t role) {
    QuadMasks q;
    q.m1 = __ballot(role == 1);
    q.m2 = __ballot(role == 2);
    q.m3 = __ballot(role == 3);
    return q;
}
__device__ __forceinline__ Fu pick_fu(const QuadMasks& q, const Fu& a0, const Fu& a1, const Fu& a2, const Fu& a3) {
    Fu o;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        int32_t v;
        asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(v) : "v"(a0.l[i]), "v"(a1.l[i]), "s"(q.m1));
        asm("v_cndmask_b32 %0, %0, %1, %2" : "+v"(v) : "v"(a2.l[i]), "s"(q.m2));
        asm("v_cndmask_b32 %0, %0, %1, %2" : "+v"(v) : "v"(a3.l[i]), "s"(q.m3));
        o.l[i] = v;
    }
    return o;
}

// One level: product k = ak * bk for k < N, computed by lane k of the quad, returned to every lane.
template <int N>
__device__ __forceinline__ void quad_products(uint32_t role, const Fu& a0, const Fu& b0, const Fu& a1, const Fu& b1, const Fu& a2, const Fu& b2,
                                              const Fu& a3, const Fu& b3, Fu& r0, Fu& r1, Fu& r2, Fu& r3) {
    const QuadMasks q = quad_masks(role);
    const Fu m = fu_mul<FqU>(pick_fu(q, a0, a1, a2, a3), pick_fu(q, b0, b1, b2, b3));
    r0 = quad_lane_fu<0>(m);
    if (N > 1) r1 = quad_lane_fu<1>(m);
    if (N > 2) r2 = quad_lane_fu<2>(m);
    if (N > 3) r3 = quad_lane_fu<3>(m);
}

// dbl-2008-s-1 over a quad
__device__ __forceinline__ XYZZu xyzzu_double_q(const XYZZu& p, uint32_t role) {
    if (xyzzu_is_identity(p)) return p;
    XYZZu o;
    const Fu u = fu_norm(fu_dbl(p.y));
    Fu v, xx, w, s, mm, d0, d1;
    quad_products<2>(role, u, u, p.x, p.x, u, u, u, u, v, xx, d0, d1);
    const Fu m = fu_norm(fu_add(fu_dbl(xx), xx));
    quad_products<4>(role, u, v, p.x, v, v, p.zz, m, m, w, s, o.zz, mm);
    o.x = fu_norm(fu_sub(mm, fu_dbl(s)));  // M^2 - 2S
    Fu a, b;
    quad_products<3>(role, m, fu_sub(s, o.x), w, p.y, w, p.zzz, w, w, a, b, o.zzz, d0);
    o.y = fu_norm(fu_sub(a, b));           // M*(S - X3) - W*Y1
    return o;
}

// a + b: add-2008-s over a quad, exceptional cases as in xyzzu_add.  By value: with an in/out reference the accumulator
// ended up in scratch memory (the DPP moves are convergent operations and kept the aggregate from being split).
__device__ __forceinline__ XYZZu xyzzu_sum_q(const XYZZu& a, const XYZZu& b, uint32_t role) {
    if (xyzzu_is_identity(b)) return a;
    if (xyzzu_is_identity(a)) return b;
    Fu u1, u2, s1, s2;
    quad_products<4>(role, a.x, b.zz, b.x, a.zz, a.y, b.zzz, b.y, a.zzz, u1, u2, s1, s2);
    const Fu p_ = fu_sub(u2, u1), r = fu_sub(s2, s1);
    if (fu_maybe_zero_mod_p<FqU>(p_)) {
        if (fu_is_zero_mod_p<FqU>(p_)) {
            if (fu_is_zero_mod_p<FqU>(r)) return xyzzu_double_q(a, role);
            return xyzzu_identity();
        }
    }
    XYZZu o;
    Fu pp, rr, zz12, zzz12;
    quad_products<4>(role, p_, p_, r, r, a.zz, b.zz, a.zzz, b.zzz, pp, rr, zz12, zzz12);
    Fu ppp, q, d0;
    quad_products<3>(role, p_, pp, u1, pp, zz12, pp, pp, pp, ppp, q, o.zz, d0);
    o.x = fu_norm(fu_sub(fu_sub(rr, ppp), fu_dbl(q)));  // R^2 - PPP - 2Q
    Fu ya, yb;
    quad_products<3>(role, r, fu_sub(q, o.x), s1, ppp, zzz12, ppp, pp, pp, ya, yb, o.zzz, d0);
    o.y = fu_norm(fu_sub(ya, yb));                       // R*(Q - X3) - S1*PPP
    return o;
}

__device__ __forceinline__ void xyzzu_add_q(XYZZu& a, const XYZZu& b, uint32_t role) { a = xyzzu_sum_q(a, b, role); }

// k * p for a small non-negative integer k < 2^nbits (double-and-add, vartime) over a quad
__device__ __forceinline__ XYZZu xyzzu_mul_small_q(const XYZZu& p, uint32_t k, uint32_t nbits, uint32_t role) {
    XYZZu acc = xyzzu_identity();
    for (int i = (int)nbits - 1; i >= 0; i--) {
        acc = xyzzu_double_q(acc, role);
        if ((k >> i) & 1) xyzzu_add_q(acc, p, role);
    }
    return acc;
}

}  // namespace h2
