// fieldu.h -- "unsaturated" BN254 field arithmetic for the hot kernels: 9 signed limbs of
// 29 bits (value = sum l[i] * 2^(29 i)), lazily reduced, Montgomery radix 2^261.
//
// Why: on gfx950 every VALU instruction costs about the same (tools/instr_rate.hip), so a modular
// multiply costs its instruction count.  With 29-bit limbs a whole column of 9 products plus 9
// reduction products fits a 64-bit accumulator, so v_mad_i64_i32 accumulates with no carry
// handling at all: ~250 instructions per multiply against ~535 for the saturated 8 x 32 CIOS of
// field.h.  Additions and subtractions are 9 independent 32-bit ops with no carries and no
// conditional subtraction; carries are propagated only where a bound below requires it.
//
// Forms (p = modulus, a = the field element):
//   E-form  value == a * 2^256 (mod p)  -- what the reference stores (field.h's Fe, RawBytes)
//   I-form  value == a * 2^261 (mod p)  -- Montgomery form for radix 2^261
//   fu_mul(x, y) = x*y / 2^261 (mod p):  I*I -> I,  E*I -> E  (so NTT data stays in E-form against
//   I-form twiddles, and E-form constants convert I-form results back for free).
//
// Bounds contract (checked by tests/cpp/test_fieldu.cpp on the host with H2_FU_CHECK):
//   fu_mul needs 9 * max|a.l| * max|b.l| + 9 * 2^58 + 2^36 < 2^63  (e.g. |a.l| < 2^30, |b.l| < 2^29.9,
//   or 2^29 x 2^30.4); its result has limbs 0..7 in [0, 2^29), a small signed top limb, and value in
//   (a*b/2^261, a*b/2^261 + p).  fu_add / fu_sub are limb-wise on int32: callers keep |l| < 2^31.
//   fu_norm propagates carries: limbs 0..7 back in [0, 2^29), value unchanged.
#pragma once
#if !defined(__HIPCC_RTC__)
#include <utility>
#else
namespace std {  // hiprtc has no <utility>: the two templates this file uses, on the compiler's builtin
template <class T, T... I>
struct integer_sequence {};
template <class T, T N>
using make_integer_sequence = __make_integer_seq<integer_sequence, T, N>;
}  // namespace std
#endif

#include "field.h"

namespace h2 {

#define H2_MASK29 0x1fffffffu

struct Fu {
    int32_t l[9];
};

struct FqU {  // base field, 29-bit limbs
    typedef FqP Sat;
    static constexpr bool ASM = false;
    static constexpr uint32_t P[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u,
                                      0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
    static constexpr uint32_t INV = 0x04866389u;   // -p^-1 mod 2^29
    static constexpr uint32_t PINV = 0x1b799c77u;  //  p^-1 mod 2^29
    static constexpr uint32_t ONE_I[9] = {0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x014c0419u, 0x0aa36fb9u,
                                          0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};  // 2^261 mod p
    static constexpr uint32_t ONE_E[9] = {0x058f0d9du, 0x1aea1c6eu, 0x11c2cf74u, 0x11d651ebu, 0x1462c0a7u,
                                          0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u};  // 2^256 mod p
    static constexpr uint32_t P16[9] = {0x07cfd470u, 0x10460b6cu, 0x072a34f0u, 0x0d522d0eu, 0x185d9781u,
                                        0x0db40c0au, 0x0a6e1411u, 0x05c26340u, 0x030644e7u};    // 16 p
};

struct FrU {  // scalar field, 29-bit limbs
    typedef FrP Sat;
    static constexpr bool ASM = false;
    static constexpr uint32_t P[9] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u,
                                      0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
    static constexpr uint32_t INV = 0x0fffffffu;
    static constexpr uint32_t PINV = 0x10000001u;
    static constexpr uint32_t ONE_I[9] = {0x0fffff57u, 0x1ea70ab4u, 0x052c068bu, 0x17504f49u, 0x0aa8075bu,
                                          0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
    static constexpr uint32_t ONE_E[9] = {0x0ffffffbu, 0x04b1a0e2u, 0x18334a6bu, 0x18ed2b3eu, 0x1462e36fu,
                                          0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u};
    static constexpr uint32_t P16[9] = {0x00000010u, 0x10fac9f8u, 0x05c2450fu, 0x1d090f37u, 0x185d2833u,
                                        0x0db40c0au, 0x0a6e1411u, 0x05c26340u, 0x030644e7u};
};

// Same fields with the multiplier's 162 multiply-adds written as explicit v_mad instructions in column order
// (device only).  Measured (tools/mul_rate.hip): 179 instead of 162 G multiplies/s at >= 4 waves per SIMD, but a
// lone wave is slower (673 vs 450 ns: one dependent chain, wait states between the asm statements), so only the
// throughput-bound kernels (bucket accumulation, NTT passes) use these; everything latency-bound uses the plain
// form, whose accumulator is pinned after each column by an empty asm so LLVM keeps the column order.
struct FqUA : FqU {
    static constexpr bool ASM = true;
};
struct FrUA : FrU {
    static constexpr bool ASM = true;
};

#ifdef H2_FU_CHECK
#include <assert.h>
#define H2_FU_ASSERT(x) assert(x)
#else
#define H2_FU_ASSERT(x) ((void)0)
#endif

H2_HD Fu fu_zero() {
    Fu o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.l[i] = 0;
    return o;
}

template <class U>
H2_HD Fu fu_const(const uint32_t c[9]) {
    Fu o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.l[i] = (int32_t)c[i];
    return o;
}

template <class U>
H2_HD Fu fu_one_i() {
    Fu o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.l[i] = (int32_t)U::ONE_I[i];
    return o;
}

template <class U>
H2_HD Fu fu_one_e() {
    Fu o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.l[i] = (int32_t)U::ONE_E[i];
    return o;
}

// all limbs exactly zero (used for the explicit identity marker, not a mod-p test)
H2_HD bool fu_all_zero(const Fu& a) {
    int32_t x = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) x |= a.l[i];
    return x == 0;
}

// bits [pos, pos+29) of the 256-bit little-endian integer x (pos may be negative: low bits are zero)
H2_HD uint32_t fe_bits29(const Fe& x, int pos) {
    if (pos < 0) return (x.l[0] << (-pos)) & H2_MASK29;
    int w = pos >> 5, sh = pos & 31;
    uint64_t lo = x.l[w];
    uint64_t hi = (w + 1 < 8) ? x.l[w + 1] : 0;
    return (uint32_t)(((lo | (hi << 32)) >> sh) & H2_MASK29);
}

// Slice an Fe (canonical limbs of some integer v < 2^256) into 29-bit limbs of the same integer:
// an E-form Fe stays E-form.  Limbs in [0, 2^29), top limb < 2^24.
H2_HD Fu fu_slice(const Fe& x) {
    Fu o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.l[i] = (int32_t)fe_bits29(x, 29 * i);
    return o;
}

// E-form Fe -> I-form Fu for free: the limbs of 32*x (< 32 p, not reduced: fu_mul does not care)
H2_HD Fu fu_from_ext(const Fe& x) {
    Fu o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.l[i] = (int32_t)fe_bits29(x, 29 * i - 5);
    return o;
}

H2_HD Fu fu_add(const Fu& a, const Fu& b) {
    Fu o;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        H2_FU_ASSERT((int64_t)a.l[i] + b.l[i] < ((int64_t)1 << 31) && (int64_t)a.l[i] + b.l[i] >= -((int64_t)1 << 31));
        o.l[i] = a.l[i] + b.l[i];
    }
    return o;
}

H2_HD Fu fu_sub(const Fu& a, const Fu& b) {
    Fu o;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        H2_FU_ASSERT((int64_t)a.l[i] - b.l[i] < ((int64_t)1 << 31) && (int64_t)a.l[i] - b.l[i] >= -((int64_t)1 << 31));
        o.l[i] = a.l[i] - b.l[i];
    }
    return o;
}

H2_HD Fu fu_dbl(const Fu& a) { return fu_add(a, a); }

H2_HD Fu fu_neg(const Fu& a) {
    Fu o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.l[i] = -a.l[i];
    return o;
}

// carry propagation: limbs 0..7 -> [0, 2^29), top limb takes the (signed) rest
H2_HD Fu fu_norm(const Fu& a) {
    Fu o;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        int32_t v = a.l[i] + c;
        o.l[i] = (int32_t)((uint32_t)v & H2_MASK29);
        c = v >> 29;
    }
    o.l[8] = a.l[8] + c;
    return o;
}

// acc += a * b (signed limbs) / acc += m * p (non-negative, p a compile-time limb of the modulus)
template <class U>
H2_HD void fu_mad_ss(int64_t& acc, int32_t a, int32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (U::ASM) {
        uint64_t sink;
        asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(sink) : "v"(a), "v"(b));
        return;
    }
#endif
    acc += (int64_t)a * (int64_t)b;
}

template <class U>
H2_HD void fu_mad_mp(int64_t& acc, uint32_t m, uint32_t p) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (U::ASM) {
        uint64_t sink;
        asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(sink) : "v"(m), "s"(p));
        return;
    }
#endif
    acc += (int64_t)m * (int64_t)p;
}

// keep the column order: without this LLVM reassociates the products into a row-wise multi-accumulator schedule
// with ~35 more instructions (240 vs 205)
template <class U>
H2_HD void fu_column_done(int64_t& acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (!U::ASM) asm("" : "+v"(acc));
#endif
    (void)acc;
}

// Montgomery product x*y/2^261 (mod p), product scanning with one 64-bit accumulator
template <class U, bool SQR, bool TWO, bool HI>
H2_HD Fu fu_fused(const Fu& a, const Fu& b, const Fu& c, const Fu& d, const Fu& h);

#if defined(__HIP_DEVICE_COMPILE__)
#include "fieldu_chain.inc"

// The explicit-mad flavour of fu_fused (U::ASM): column K of the product scan, its multiply-adds issued as one asm statement
// per operand kind (a*b, -c*d, m*p) so that no hazard padding lands between them.
template <class U, bool SQR, bool TWO, bool HI, int K>
struct FuAsmColumn {
    static constexpr int LO = K > 8 ? K - 8 : 0, TOP = K < 8 ? K : 8;
    static constexpr int NP = SQR ? (TOP - LO) / 2 + 1 : TOP - LO + 1;  // products of this column (the square's symmetric half)
    static constexpr int NR = K < 9 ? K : 17 - K;                       // reduction terms m[i] * P[K - i] already known

    template <int... I>
    __device__ __forceinline__ static void products(int64_t& acc, const Fu& a, const Fu& b, const int32_t (&a2)[9], std::integer_sequence<int, I...>) {
        if constexpr (SQR) {
            const int32_t x[NP] = {((LO + I) < (K - LO - I) ? a2[LO + I] : a.l[LO + I])...};
            const int32_t y[NP] = {a.l[K - LO - I]...};
            fu_chain_ss<NP>(acc, x, y);
        } else {
            const int32_t x[NP] = {a.l[LO + I]...};
            const int32_t y[NP] = {b.l[K - LO - I]...};
            fu_chain_ss<NP>(acc, x, y);
        }
    }
    template <int... I>
    __device__ __forceinline__ static void second(int64_t& acc, const int32_t (&nc)[9], const Fu& d, std::integer_sequence<int, I...>) {
        const int32_t x[TOP - LO + 1] = {nc[LO + I]...};
        const int32_t y[TOP - LO + 1] = {d.l[K - LO - I]...};
        fu_chain_ss<TOP - LO + 1>(acc, x, y);
    }
    template <int... I>
    __device__ __forceinline__ static void reduction(int64_t& acc, const uint32_t (&m)[9], std::integer_sequence<int, I...>) {
        constexpr int FIRST = K < 9 ? 0 : K - 8;  // i runs FIRST .. FIRST + NR - 1
        const uint32_t x[NR] = {m[FIRST + I]...};
        const uint32_t y[NR] = {U::P[K - FIRST - I]...};
        fu_chain_mp<NR>(acc, x, y);
    }
    __device__ __forceinline__ static void run(int64_t& acc, uint32_t (&m)[9], Fu& r, const Fu& a, const Fu& b, const int32_t (&a2)[9],
                                                const int32_t (&nc)[9], const Fu& d, const Fu& h) {
        products(acc, a, b, a2, std::make_integer_sequence<int, NP>());
        if constexpr (TWO) second(acc, nc, d, std::make_integer_sequence<int, TOP - LO + 1>());
        if constexpr (NR > 0) reduction(acc, m, std::make_integer_sequence<int, NR>());
        if constexpr (K < 9) {
            m[K] = ((uint32_t)acc * U::INV) & H2_MASK29;
            acc += (int64_t)((uint64_t)m[K] * U::P[0]);  // one add between two asm statements: nothing to reassociate
            acc >>= 29;                                  // exact
        } else {
            if constexpr (HI) acc -= (int64_t)h.l[K - 9];
            r.l[K - 9] = (int32_t)((uint32_t)acc & H2_MASK29);
            acc >>= 29;
        }
    }
};

template <class U, bool SQR, bool TWO, bool HI, int... K>
__device__ __forceinline__ Fu fu_fused_asm(const Fu& a, const Fu& b, const Fu& c, const Fu& d, const Fu& h, std::integer_sequence<int, K...>) {
    int64_t acc = 0;
    uint32_t m[9];
    int32_t a2[9], nc[9];
    Fu r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        a2[i] = SQR ? a.l[i] * 2 : 0;
        nc[i] = TWO ? -c.l[i] : 0;
    }
    (FuAsmColumn<U, SQR, TWO, HI, K>::run(acc, m, r, a, b, a2, nc, d, h), ...);
    if (HI) acc -= (int64_t)h.l[8];
    r.l[8] = (int32_t)acc;
    return r;
}
#endif

template <class U>
H2_HD Fu fu_mul(const Fu& a, const Fu& b) {
    return fu_fused<U, false, false, false>(a, b, a, a, a);
}

// General fused form: (a*b [- c*d] [- h * 2^261]) / 2^261 (mod p) with ONE Montgomery reduction.
//   SQR : b is ignored, a*a uses the symmetric 45 products
//   TWO : subtract the second product c*d in the same columns (bound: 18 * La*Lb ... see below)
//   HI  : subtract h from the high half (h * 2^261 is h shifted up by 9 limbs), i.e. result = a*b/2^261 - h
// Column bound: (9 or 18) * max|l|^2 + 9 * 2^58 + |h| < 2^63: with |l| < 2^29 everywhere the two-product form
// peaks at 27 * 2^58 = 2^62.75.  The result always has limbs 0..7 in [0, 2^29) (no fu_norm needed after it).
template <class U, bool SQR, bool TWO, bool HI>
H2_HD Fu fu_fused(const Fu& a, const Fu& b, const Fu& c, const Fu& d, const Fu& h) {
#ifdef H2_FU_CHECK
    {
        int64_t ma = 0, mb = 0, mc = 0, md = 0;
        for (int i = 0; i < 9; i++) {
            int64_t x;
            x = a.l[i] < 0 ? -(int64_t)a.l[i] : a.l[i]; if (x > ma) ma = x;
            x = b.l[i] < 0 ? -(int64_t)b.l[i] : b.l[i]; if (x > mb) mb = x;
            x = c.l[i] < 0 ? -(int64_t)c.l[i] : c.l[i]; if (x > mc) mc = x;
            x = d.l[i] < 0 ? -(int64_t)d.l[i] : d.l[i]; if (x > md) md = x;
        }
        if (SQR) mb = ma;
        __int128 bound = (__int128)9 * ma * mb + (TWO ? (__int128)9 * mc * md : 0) + ((__int128)9 << 58) + ((__int128)1 << 36);
        assert(bound < ((__int128)1 << 63));
    }
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (U::ASM) return fu_fused_asm<U, SQR, TWO, HI>(a, b, c, d, h, std::make_integer_sequence<int, 17>());
#endif
    int64_t acc = 0;
    uint32_t m[9];
    int32_t a2[9], nc[9];
    Fu r;
    if (SQR) {
#pragma unroll
        for (int i = 0; i < 9; i++) a2[i] = a.l[i] * 2;
    }
    if (TWO) {
#pragma unroll
        for (int i = 0; i < 9; i++) nc[i] = -c.l[i];
    }
#pragma unroll
    for (int k = 0; k < 17; k++) {
        const int lo = k > 8 ? k - 8 : 0, hi = k < 8 ? k : 8;
        if (SQR) {
#pragma unroll
            for (int i = lo; i <= hi; i++) {
                int j = k - i;
                if (i < j) fu_mad_ss<U>(acc, a2[i], a.l[j]);
                else if (i == j) fu_mad_ss<U>(acc, a.l[i], a.l[i]);
            }
        } else {
#pragma unroll
            for (int i = lo; i <= hi; i++) fu_mad_ss<U>(acc, a.l[i], b.l[k - i]);
        }
        if (TWO) {
#pragma unroll
            for (int i = lo; i <= hi; i++) fu_mad_ss<U>(acc, nc[i], d.l[k - i]);
        }
        if (k < 9) {
#pragma unroll
            for (int i = 0; i < k; i++) fu_mad_mp<U>(acc, m[i], U::P[k - i]);
            fu_column_done<U>(acc);
            m[k] = ((uint32_t)acc * U::INV) & H2_MASK29;
            fu_mad_mp<U>(acc, m[k], U::P[0]);
            acc >>= 29;  // exact
        } else {
#pragma unroll
            for (int i = k - 8; i <= 8; i++) fu_mad_mp<U>(acc, m[i], U::P[k - i]);
            if (HI) acc -= (int64_t)h.l[k - 9];
            fu_column_done<U>(acc);
            r.l[k - 9] = (int32_t)((uint32_t)acc & H2_MASK29);
            acc >>= 29;
        }
    }
    if (HI) acc -= (int64_t)h.l[8];
    r.l[8] = (int32_t)acc;
    return r;
}

template <class U>
H2_HD Fu fu_sqr(const Fu& a) {
    return fu_fused<U, true, false, false>(a, a, a, a, a);
}

// (a*b - c*d) / 2^261
template <class U>
H2_HD Fu fu_mul_sub(const Fu& a, const Fu& b, const Fu& c, const Fu& d) {
    return fu_fused<U, false, true, false>(a, b, c, d, a);
}

// a*b / 2^261 - h   (h loose, |h.l| < 2^31): a product and an addend in one pass, the result normalised
template <class U>
H2_HD Fu fu_mul_subh(const Fu& a, const Fu& b, const Fu& h) {
    return fu_fused<U, false, false, true>(a, b, a, a, h);
}

// a*a / 2^261 - h   (h loose, |h.l| < 2^31)
template <class U>
H2_HD Fu fu_sqr_sub(const Fu& a, const Fu& h) {
    return fu_fused<U, true, false, true>(a, a, a, a, h);
}

// Cheap necessary condition for value == k*p with |k| <= 8 (so for value == 0 mod p when
// |value| < 8.5 p): value = k*p  =>  l0 * p^-1 == k (mod 2^29).  False positives ~2^-25.
template <class U>
H2_HD bool fu_maybe_zero_mod_p(const Fu& a) {
    uint32_t t = ((uint32_t)a.l[0] * U::PINV + 8u) & H2_MASK29;
    return t <= 16u;
}

// Pack a normalised Fu whose value is in [0, 2^256) into 8 x 32-bit words
H2_HD void fu_pack(const Fu& a, uint32_t w[8]) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        // word j = bits [32 j, 32 j + 32)
        int lo_limb = (32 * j) / 29, sh = (32 * j) % 29;
        uint64_t v = (uint64_t)(uint32_t)a.l[lo_limb] >> sh;
        int have = 29 - sh;
        if (lo_limb + 1 < 9) v |= (uint64_t)(uint32_t)a.l[lo_limb + 1] << have;
        have += 29;
        if (have < 32 && lo_limb + 2 < 9) v |= (uint64_t)(uint32_t)a.l[lo_limb + 2] << have;
        w[j] = (uint32_t)v;
    }
}

// Exact reduction: any Fu with |value| < 16 p -> the canonical integer in [0, p) as an Fe
// (same residue, so the form -- E or I -- is unchanged).  Not on the hot path.
template <class U>
H2_HD Fe fu_canon(const Fu& a) {
    Fu cur = fu_norm(fu_add(a, fu_const<U>(U::P16)));  // now in (0, 32 p) < 2^260
    // conditional subtractions of 16p, 8p, 4p, 2p, p
    for (int sh = 4; sh >= 0; sh--) {
        Fu kp;
        uint64_t c = 0;
        for (int i = 0; i < 9; i++) {
            uint64_t v = ((uint64_t)U::P[i] << sh) + c;
            kp.l[i] = (int32_t)(v & H2_MASK29);
            c = v >> 29;
        }
        kp.l[8] += (int32_t)(c << 29);
        Fu d = fu_norm(fu_sub(cur, kp));
        if (d.l[8] >= 0) cur = d;  // normalised: the sign of the value is the sign of the top limb
    }
    uint32_t w[8];
    fu_pack(cur, w);
    Fe out;
#pragma unroll
    for (int j = 0; j < 8; j++) out.l[j] = w[j];
    return out;
}

// The same reduction for the hot path: |value| < 16 p -> canonical Fe, ~110 instructions (fu_mul_canon by one: ~290).
// xp = value + 16 p lies in (0, 32 p); with T its top limb (bits 232 up, < 2^27) and P8 the modulus's,
// q = floor(xp / p) satisfies floor(T / (P8 + 1)) in {q - 1, q}: xp / p - T / (P8 + 1) < (T + P8 + 1) / (P8 (P8 + 1)) < 2^-16.
// The float estimate below is biased low by a relative 2^-20 (more than the two roundings can add), which can take it
// below floor(T / (P8 + 1)) only when T / (P8 + 1) is within 2^-14.9 ABOVE an integer -- and then floor(T / (P8 + 1)) = q,
// because the q - 1 case needs it within 2^-16 BELOW one.  So q' is q - 1 or q, xp - q' p lies in [0, 2 p), and one
// conditional subtraction finishes (fe_cond_sub).
template <class U>
H2_HD Fe fu_canon_fast(const Fu& x) {
    typedef typename U::Sat P;
    const Fu xp = fu_norm(fu_add(x, fu_const<U>(U::P16)));
    constexpr float C = (float)((1.0 / ((double)U::P[8] + 1.0)) * (1.0 - 1.0 / 1048576.0));
    const uint32_t q = (uint32_t)((float)xp.l[8] * C);
    Fu y;
    int64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        acc += (int64_t)xp.l[i] - (int64_t)q * (int64_t)U::P[i];
        y.l[i] = (int32_t)((uint32_t)acc & H2_MASK29);
        acc >>= 29;
    }
    acc += (int64_t)xp.l[8] - (int64_t)q * (int64_t)U::P[8];
    y.l[8] = (int32_t)acc;
    H2_FU_ASSERT(y.l[8] >= 0 && y.l[8] < (1 << 23));
    uint32_t w[8];
    fu_pack(y, w);
    Fe o;
    fe_cond_sub<P>(o, w);
    return o;
}

// Exact test value == 0 (mod p) for |value| < 8.5 p: cheap filter first, exact reduction on a hit
template <class U>
H2_HD bool fu_is_zero_mod_p(const Fu& a) {
    if (!fu_maybe_zero_mod_p<U>(a)) return false;
    return fe_is_zero(fu_canon<U>(a));
}

// (x * c) / 2^261 reduced to the canonical integer in [0, p), for |x| < 16 p and c normalised
// non-negative (< 2^256).  With c an E-form constant and x I-form this is the E-form result the
// reference stores; with x E-form and c I-form likewise.  x + 16p > 0 makes the Montgomery
// result land in [0, 1.2 p): one conditional subtraction finishes the job.
template <class U>
H2_HD Fe fu_mul_canon(const Fu& x, const Fu& c) {
    typedef typename U::Sat P;
    Fu xp = fu_norm(fu_add(x, fu_const<U>(U::P16)));
    Fu r = fu_mul<U>(xp, c);  // in [0, 32p * 2^256 / 2^261 + p) = [0, 2p)
    uint32_t w[8];
    fu_pack(r, w);
    Fe o;
    fe_cond_sub<P>(o, w);
    return o;
}

// a^-1 (mod p) by Fermat on the unsaturated multiplier: I-form in (|value| < 32 p, limbs < 2^29: fu_from_ext output or
// any normalised value), I-form out in (-0.2 p, 1.2 p); 0 -> 0.  253 squarings + one multiply per set bit of p - 2:
// ~78 k instructions against ~204 k for fe_inv's saturated CIOS -- the chain that sets the latency of a batched
// normalisation with few lanes.
template <class U>
H2_HD Fu fu_inv(const Fu& a) {
    typedef typename U::Sat P;
    uint32_t e[8];
#pragma unroll
    for (int i = 0; i < 8; i++) e[i] = P::MOD[i];
    e[0] -= 2;  // MOD odd, low limb >= 2
    Fu r = fu_mul<U>(a, fu_one_i<U>());  // bring |value| inside (-0.2 p, 1.2 p), same residue
    const Fu base = r;
    int top = 255;
    while (top > 0 && !((e[top >> 5] >> (top & 31)) & 1)) top--;
    for (int i = top - 1; i >= 0; i--) {
        r = fu_sqr<U>(r);
        if ((e[i >> 5] >> (i & 31)) & 1) r = fu_mul<U>(r, base);
    }
    return r;
}

}  // namespace h2
