// field.h -- BN254 base field Fq and scalar field Fr for gfx950 (and the library's host side).
//
// Replaces the halo2curves 0.3.1 bn256::{Fq, Fr} arithmetic that halo2_proofs calls on the
// hot path (reference call sites: halo2_proofs/src/arithmetic.rs:14,48,62-65,74-77,98,197,
// 214-225).  In-memory form is the reference's: 4 x u64 little-endian limbs, Montgomery
// R = 2^256, always fully reduced (SerdeFormat::RawBytes, helpers.rs:13-19) -- read here as
// 8 x u32 limbs, the natural operand width of v_mad_u64_u32.
//
// Measured on MI355X (tools/instr_rate.hip): v_mad_u64_u32 issues at ~5.5 cycles per wave64
// instruction, v_mul_lo/hi_u32, v_add_co/addc, v_lshl_add_u64 at ~4.3-4.7 -- integer multiply
// is NOT quarter-rate on gfx950, so the cost of a modular multiply is its instruction count.
#pragma once
#if defined(__HIPCC_RTC__)
// compiled at run time by hiprtc (evalh.hip's per-circuit gates kernel): the runtime's device declarations and the fixed-width
// integer types are part of hiprtc's built-in prelude, and no system header can be found from there
#define H2_HD __device__ __forceinline__
typedef unsigned char uint8_t;
typedef unsigned short uint16_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long long int64_t;
typedef unsigned long long uint64_t;
typedef unsigned long size_t;
typedef unsigned long uintptr_t;
#else
#include <stdint.h>
#endif

#if defined(__HIPCC_RTC__)
#elif defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define H2_HD __host__ __device__ __forceinline__
#else
#define H2_HD inline
#endif

namespace h2 {

struct alignas(16) Fe {
    uint32_t l[8];
};

struct FqP {  // base field modulus q
    static constexpr uint32_t MOD[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                        0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t R[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                      0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                       0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
    static constexpr uint32_t INV = 0xe4866389u;  // -q^-1 mod 2^32
};

struct FrP {  // scalar field modulus r
    static constexpr uint32_t MOD[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                        0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t R[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                      0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                       0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
    static constexpr uint32_t INV = 0xefffffffu;  // -r^-1 mod 2^32
    // PrimeField constants (halo2curves bn256::Fr), Montgomery form
    static constexpr uint32_t ROOT_OF_UNITY[8] = {0xb639feb8u, 0x9632c7c5u, 0x0d0ff299u, 0x985ce340u,
                                                  0x01b0ecd8u, 0xb2dd8800u, 0x6d98ce29u, 0x1d69070du};
    static constexpr uint32_t ROOT_OF_UNITY_INV[8] = {0xaffb3d96u, 0x05f05c05u, 0xfc3b5137u, 0xb8e594ebu,
                                                      0xb85bc4c1u, 0x60314620u, 0xbb6fc591u, 0x2a4129beu};
    static constexpr uint32_t ZETA[8] = {0x55fcd653u, 0x0363f299u, 0x5fc1e200u, 0x73e7950bu,
                                         0x576d9d24u, 0xc5fce83eu, 0xa1c3a4d4u, 0x059c805du};
    static constexpr uint32_t S = 28;  // 2-adicity
};

template <class P>
H2_HD Fe fe_zero() {
    Fe o;
#pragma unroll
    for (int i = 0; i < 8; i++) o.l[i] = 0;
    return o;
}

template <class P>
H2_HD Fe fe_one() {
    Fe o;
#pragma unroll
    for (int i = 0; i < 8; i++) o.l[i] = P::R[i];
    return o;
}

H2_HD bool fe_is_zero(const Fe& a) {
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) x |= a.l[i];
    return x == 0;
}

H2_HD bool fe_eq(const Fe& a, const Fe& b) {
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) x |= a.l[i] ^ b.l[i];
    return x == 0;
}

// o = t - MOD if t >= MOD else t   (t < 2*MOD)
template <class P>
H2_HD void fe_cond_sub(Fe& o, const uint32_t t[8]) {
    uint32_t s[8];
    uint32_t br = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        uint64_t d = (uint64_t)t[j] - P::MOD[j] - br;
        s[j] = (uint32_t)d;
        br = (uint32_t)(d >> 63);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) o.l[j] = br ? t[j] : s[j];
}

template <class P>
H2_HD Fe fe_add(const Fe& a, const Fe& b) {
    uint32_t t[8];
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        uint64_t s = (uint64_t)a.l[j] + b.l[j] + c;
        t[j] = (uint32_t)s;
        c = (uint32_t)(s >> 32);
    }
    Fe o;  // MOD < 2^254: no carry out of 256 bits
    fe_cond_sub<P>(o, t);
    return o;
}

template <class P>
H2_HD Fe fe_sub(const Fe& a, const Fe& b) {
    uint32_t t[8];
    uint32_t br = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        uint64_t d = (uint64_t)a.l[j] - b.l[j] - br;
        t[j] = (uint32_t)d;
        br = (uint32_t)(d >> 63);
    }
    Fe o;
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        uint64_t s = (uint64_t)t[j] + (br ? P::MOD[j] : 0u) + c;
        o.l[j] = (uint32_t)s;
        c = (uint32_t)(s >> 32);
    }
    return o;
}

template <class P>
H2_HD Fe fe_neg(const Fe& a) {
    return fe_sub<P>(fe_zero<P>(), a);
}

template <class P>
H2_HD Fe fe_dbl(const Fe& a) {
    return fe_add<P>(a, a);
}

// Montgomery product a*b*2^-256 mod MOD.  CIOS over 8 x 32-bit limbs; MOD's top bit is clear,
// so the running value never needs a ninth-plus-one word ("no-carry" variant).
template <class P>
H2_HD Fe fe_mul(const Fe& a, const Fe& b) {
    uint32_t t[8];
#pragma unroll
    for (int j = 0; j < 8; j++) t[j] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint64_t x = (uint64_t)a.l[j] * b.l[i] + t[j] + c;
            t[j] = (uint32_t)x;
            c = x >> 32;
        }
        uint32_t t8 = (uint32_t)c;
        uint32_t m = t[0] * P::INV;
        uint64_t x = (uint64_t)m * P::MOD[0] + t[0];
        c = x >> 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            x = (uint64_t)m * P::MOD[j] + t[j] + c;
            t[j - 1] = (uint32_t)x;
            c = x >> 32;
        }
        t[7] = t8 + (uint32_t)c;
    }
    Fe o;
    fe_cond_sub<P>(o, t);
    return o;
}

template <class P>
H2_HD Fe fe_sqr(const Fe& a) {
    return fe_mul<P>(a, a);
}

// Montgomery -> canonical integer (PrimeField::to_repr, arithmetic.rs:14): a * 1 * R^-1
template <class P>
H2_HD Fe fe_to_canonical(const Fe& a) {
    Fe one = fe_zero<P>();
    one.l[0] = 1;
    return fe_mul<P>(a, one);
}

template <class P>
H2_HD Fe fe_from_canonical(const Fe& c) {
    Fe r2;
#pragma unroll
    for (int i = 0; i < 8; i++) r2.l[i] = P::R2[i];
    return fe_mul<P>(c, r2);
}

template <class P>
H2_HD Fe fe_from_u64(uint64_t v) {
    Fe c = fe_zero<P>();
    c.l[0] = (uint32_t)v;
    c.l[1] = (uint32_t)(v >> 32);
    return fe_from_canonical<P>(c);
}

// a^e, e = 8 x u32 little-endian limbs (vartime in e)
template <class P>
H2_HD Fe fe_pow(const Fe& a, const uint32_t e[8]) {
    Fe r = fe_one<P>();
    for (int i = 255; i >= 0; i--) {
        r = fe_sqr<P>(r);
        if ((e[i >> 5] >> (i & 31)) & 1) r = fe_mul<P>(r, a);
    }
    return r;
}

template <class P>
H2_HD Fe fe_pow_u64(const Fe& a, uint64_t e) {
    Fe r = fe_one<P>();
    for (int i = 63; i >= 0; i--) {
        r = fe_sqr<P>(r);
        if ((e >> i) & 1) r = fe_mul<P>(r, a);
    }
    return r;
}

// a^-1 = a^(MOD-2); 0 -> 0
template <class P>
H2_HD Fe fe_inv(const Fe& a) {
    uint32_t e[8];
#pragma unroll
    for (int i = 0; i < 8; i++) e[i] = P::MOD[i];
    e[0] -= 2;  // MOD odd, low limb >= 2
    return fe_pow<P>(a, e);
}

template <class P>
H2_HD bool fe_is_canonical(const Fe& a) {  // a < MOD
    for (int i = 7; i >= 0; i--) {
        if (a.l[i] < P::MOD[i]) return true;
        if (a.l[i] > P::MOD[i]) return false;
    }
    return false;
}

}  // namespace h2
