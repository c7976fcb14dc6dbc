// glv.h -- the GLV decomposition of a BN254 scalar: k = k1 + k2 * LAMBDA (mod r) with |k1|, |k2| < 2^128, where
// LAMBDA is the cube root of unity of Fr for which [LAMBDA](x, y) = (BETA * x, y) on G1 (BETA a cube root of unity of Fq).
// Used by the scalar ladder of g_to_lagrange (ecfft.hip): half the doublings of a 254-bit ladder.
//
// Constants (derived with Python integers; tests/cpp/test_fieldu.cpp checks the identity k1 + k2 * LAMBDA == k (mod r) and
// the size bound over random and edge-case scalars on the host):
//   LAMBDA = 0xb3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd,  BETA = 0x59e26bcea0d48bacd4f263f1acdb5c4f5763473177fffffe
//   lattice basis of {(x, y) : x + y * LAMBDA == 0 (mod r)}:  v1 = (A1, -NB1),  v2 = (A2, B2),  det = r
//   c1 = floor(k * G1 / 2^256) ~ B2 * k / r,   c2 = floor(k * G2 / 2^256) ~ NB1 * k / r   (G = floor(2^256 * . / r))
//   k1 = k - c1 * A1 - c2 * A2,   k2 = c1 * NB1 - c2 * B2
// The identity holds for any integers c1, c2; the floors only cost a bit of size (measured maximum: 127 bits).
#pragma once
#include "field.h"

namespace h2 {

struct GlvScalar {
    uint32_t k1[5], k2[5];  // magnitudes, little-endian 32-bit limbs (< 2^130)
    uint32_t neg1, neg2;    // signs
};

namespace glv {
H2_HD void mul(const uint32_t* a, int na, const uint32_t* b, int nb, uint32_t* out) {
    for (int i = 0; i < na + nb; i++) out[i] = 0;
    for (int i = 0; i < na; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < nb; j++) {
            uint64_t t = (uint64_t)a[i] * b[j] + out[i + j] + carry;
            out[i + j] = (uint32_t)t;
            carry = t >> 32;
        }
        out[i + nb] = (uint32_t)carry;
    }
}
// |a - b| over n limbs; returns 1 when a < b
H2_HD uint32_t sub_abs(const uint32_t* a, const uint32_t* b, int n, uint32_t* out) {
    int lt = 0;
    for (int i = n - 1; i >= 0; i--) {
        if (a[i] != b[i]) {
            lt = a[i] < b[i];
            break;
        }
    }
    const uint32_t* hi = lt ? b : a;
    const uint32_t* lo = lt ? a : b;
    uint64_t borrow = 0;
    for (int i = 0; i < n; i++) {
        uint64_t t = (uint64_t)hi[i] - lo[i] - borrow;
        out[i] = (uint32_t)t;
        borrow = (t >> 63) & 1;
    }
    return (uint32_t)lt;
}
}  // namespace glv

// k: canonical scalar (< r), 8 little-endian 32-bit limbs
H2_HD GlvScalar glv_decompose(const uint32_t k[8]) {
    const uint32_t G1[3] = {0xc7e0b3d7u, 0xd91d232eu, 0x2u};
    const uint32_t G2[5] = {0x391eb18du, 0x7a7bd9d4u, 0xa773d2cfu, 0x4ccef014u, 0x2u};
    const uint32_t A1[2] = {0x94d213e3u, 0x89d32568u};
    const uint32_t A2[4] = {0x1221250bu, 0x0be4e154u, 0xeeb859fdu, 0x6f4d8248u};
    const uint32_t NB1[4] = {0x7d4f1128u, 0x8211bbebu, 0xeeb859fcu, 0x6f4d8248u};
    const uint32_t B2[2] = {0x94d213e3u, 0x89d32568u};
    uint32_t p1[11], p2[13];
    glv::mul(k, 8, G1, 3, p1);
    glv::mul(k, 8, G2, 5, p2);
    const uint32_t* c1 = p1 + 8;  // 3 limbs
    const uint32_t* c2 = p2 + 8;  // 5 limbs
    // k1 = k - (c1 * A1 + c2 * A2)
    uint32_t t1[5], t2[9], sum[9], kk[9], d[9];
    glv::mul(c1, 3, A1, 2, t1);
    glv::mul(c2, 5, A2, 4, t2);
    uint64_t carry = 0;
    for (int i = 0; i < 9; i++) {
        uint64_t t = (uint64_t)t2[i] + (i < 5 ? t1[i] : 0) + carry;
        sum[i] = (uint32_t)t;
        carry = t >> 32;
    }
    for (int i = 0; i < 9; i++) kk[i] = i < 8 ? k[i] : 0;
    GlvScalar o;
    o.neg1 = glv::sub_abs(kk, sum, 9, d);
    for (int i = 0; i < 5; i++) o.k1[i] = d[i];
    // k2 = c1 * NB1 - c2 * B2
    uint32_t u[7], v[7], e[7];
    glv::mul(c1, 3, NB1, 4, u);
    glv::mul(c2, 5, B2, 2, v);
    o.neg2 = glv::sub_abs(u, v, 7, e);
    for (int i = 0; i < 5; i++) o.k2[i] = e[i];
    return o;
}

}  // namespace h2
