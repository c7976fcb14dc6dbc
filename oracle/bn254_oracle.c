/*
 * oracle/bn254_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, gcc, unsigned __int128) of the halo2_proofs hot
 * path: BN254 G1 multi-scalar multiplication, the radix-2 NTT over BN254 Fr with
 * the EvaluationDomain conversions around it, and Evaluator::evaluate_h
 * (plonk/evaluation.rs:280-522, at the end of this file).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this file's shared object; the product (libhalo2hip.so) never does.
 *
 * PARITY UNPINNED: the reference (Rust, /root/reference) can be neither built
 * nor run here (no cargo/rustc), its field/curve arithmetic lives in the
 * un-vendored dependency halo2curves tag 0.3.1 (halo2_proofs/Cargo.toml:51),
 * and its own tests hold no BN254 byte-level vectors (they draw the SRS from
 * OsRng: poly/kzg/commitment.rs:365,381).  This restatement is therefore pinned
 * by (i) independent Python big-integer golden vectors (tests/golden/, made by
 * tests/golden/make_golden.py: naive affine double-and-add MSM, O(n^2) DFT;
 * tests/golden/make_evalh_golden.py: the gate / permutation / lookup constraint
 * formulas evaluated row by row, not through the flattened graph) and
 * (ii) the algebraic identities the reference's tests assert
 * (test_commit_lagrange, poly/kzg/commitment.rs:361-384).
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference/halo2_proofs/src/).  Field elements: 4 x u64 little-endian
 * limbs, Montgomery form R = 2^256, fully reduced -- halo2curves' in-memory and
 * SerdeFormat::RawBytes layout (helpers.rs:13-19).  G1Affine = x||y (64 B),
 * identity = (0,0).  G1 = Jacobian (x,y,z) (96 B), identity z = 0.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;
typedef struct { fe x, y; } g1a;      /* affine, identity = (0,0) */
typedef struct { fe x, y, z; } g1j;   /* Jacobian, identity z = 0 */

typedef struct {
    uint64_t p[4];
    uint64_t inv;      /* -p^-1 mod 2^64 */
    fe r, r2;          /* R mod p, R^2 mod p */
} fparams;

/* moduli: halo2curves 0.3.1 bn256::{fq,fr} */
static fparams FQ = {{0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL,
                      0xb85045b68181585dULL, 0x30644e72e131a029ULL}, 0, {{0}}, {{0}}};
static fparams FR = {{0x43e1f593f0000001ULL, 0x2833e84879b97091ULL,
                      0xb85045b68181585dULL, 0x30644e72e131a029ULL}, 0, {{0}}, {{0}}};

static fe FR_ROOT_OF_UNITY;   /* Montgomery; order 2^28 */
static fe FR_ROOT_OF_UNITY_INV;
static fe FR_ZETA;
static fe FQ_B3;              /* curve constant b = 3 */
#define FR_S 28

/* ------------------------------------------------------------------ field */

static inline int fe_is_zero(const fe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fe_eq(const fe *a, const fe *b) { return memcmp(a, b, sizeof(fe)) == 0; }

static inline int ge_p(const uint64_t a[4], const uint64_t p[4]) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > p[i]) return 1;
        if (a[i] < p[i]) return 0;
    }
    return 1;
}

static inline void sub_p(uint64_t a[4], const uint64_t p[4]) {
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - p[i] - (uint64_t)br;
        a[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
}

static inline void fe_add(fe *o, const fe *a, const fe *b, const fparams *F) {
    u128 c = 0;
    uint64_t t[4];
    for (int i = 0; i < 4; i++) {
        c += (u128)a->l[i] + b->l[i];
        t[i] = (uint64_t)c;
        c >>= 64;
    }
    /* p < 2^254 so no carry out of 256 bits */
    if (ge_p(t, F->p)) sub_p(t, F->p);
    memcpy(o->l, t, 32);
}

static inline void fe_sub(fe *o, const fe *a, const fe *b, const fparams *F) {
    uint64_t t[4];
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a->l[i] - b->l[i] - (uint64_t)br;
        t[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
    if (br) {
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)t[i] + F->p[i];
            t[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    memcpy(o->l, t, 32);
}

static inline void fe_neg(fe *o, const fe *a, const fparams *F) {
    fe z = {{0, 0, 0, 0}};
    fe_sub(o, &z, a, F);
}

static inline void fe_dbl(fe *o, const fe *a, const fparams *F) { fe_add(o, a, a, F); }

/* Montgomery product a*b*R^-1 mod p (CIOS, 4 x 64) */
static inline void fe_mul(fe *o, const fe *a, const fe *b, const fparams *F) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a->l[j] * b->l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * F->inv;
        c = (u128)m * F->p[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * F->p[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    if (t[4] || ge_p(t, F->p)) sub_p(t, F->p);
    memcpy(o->l, t, 32);
}

static inline void fe_sqr(fe *o, const fe *a, const fparams *F) { fe_mul(o, a, a, F); }

/* a^e, e given as little-endian u64 limbs */
static void fe_pow(fe *o, const fe *a, const uint64_t *e, int nlimbs, const fparams *F) {
    fe res = F->r;
    for (int i = nlimbs * 64 - 1; i >= 0; i--) {
        fe_sqr(&res, &res, F);
        if ((e[i / 64] >> (i % 64)) & 1) fe_mul(&res, &res, a, F);
    }
    *o = res;
}

/* a^-1 by Fermat (0 -> 0) */
static void fe_inv(fe *o, const fe *a, const fparams *F) {
    uint64_t e[4];
    memcpy(e, F->p, 32);
    e[0] -= 2; /* p is odd and p[0] >= 2: no borrow */
    fe_pow(o, a, e, 4, F);
}

static void fe_from_u64(fe *o, uint64_t v, const fparams *F) {
    fe t = {{v, 0, 0, 0}};
    fe_mul(o, &t, &F->r2, F);
}

/* canonical integer (non-Montgomery) -> Montgomery */
static void fe_from_canonical(fe *o, const uint64_t c[4], const fparams *F) {
    fe t;
    memcpy(t.l, c, 32);
    fe_mul(o, &t, &F->r2, F);
}

/* PrimeField::to_repr (arithmetic.rs:14): canonical little-endian limbs */
static void fe_to_canonical(uint64_t c[4], const fe *a, const fparams *F) {
    fe one = {{1, 0, 0, 0}}, t;
    fe_mul(&t, a, &one, F);
    memcpy(c, t.l, 32);
}

static void fparams_init(fparams *F) {
    /* inv = -p^-1 mod 2^64 (Newton) */
    uint64_t x = 1;
    for (int i = 0; i < 6; i++) x *= 2 - F->p[0] * x;
    F->inv = (uint64_t)0 - x;
    /* R = 2^256 mod p, R2 = 2^512 mod p by repeated doubling of 1 */
    fe t = {{1, 0, 0, 0}};
    for (int i = 0; i < 512; i++) {
        fe_dbl(&t, &t, F);
        if (i == 255) F->r = t;
    }
    F->r2 = t;
}

static int g_init_done = 0;
void oracle_init(void) {
    if (g_init_done) return;
    fparams_init(&FQ);
    fparams_init(&FR);
    /* ROOT_OF_UNITY = 7^((r-1)/2^28); ZETA = 7^(2(r-1)/3)  (verified in tests) */
    uint64_t rm1[4];
    memcpy(rm1, FR.p, 32);
    rm1[0] -= 1;
    uint64_t t[4];
    /* t = (r-1) >> 28 */
    for (int i = 0; i < 4; i++) t[i] = (rm1[i] >> 28) | (i < 3 ? rm1[i + 1] << 36 : 0);
    fe seven;
    fe_from_u64(&seven, 7, &FR);
    fe_pow(&FR_ROOT_OF_UNITY, &seven, t, 4, &FR);
    fe_inv(&FR_ROOT_OF_UNITY_INV, &FR_ROOT_OF_UNITY, &FR);
    /* (r-1)/3 then *2 */
    u128 rem = 0;
    uint64_t q[4];
    for (int i = 3; i >= 0; i--) {
        u128 cur = (rem << 64) | rm1[i];
        q[i] = (uint64_t)(cur / 3);
        rem = cur % 3;
    }
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)q[i] * 2;
        q[i] = (uint64_t)c;
        c >>= 64;
    }
    fe_pow(&FR_ZETA, &seven, q, 4, &FR);
    fe_from_u64(&FQ_B3, 3, &FQ);
    g_init_done = 1;
}

/* ------------------------------------------------------------- exported field API */
/* which: 0 = Fq, 1 = Fr */
static const fparams *sel(int which) { return which ? &FR : &FQ; }
void oracle_fe_mul(int which, const fe *a, const fe *b, fe *o) { oracle_init(); fe_mul(o, a, b, sel(which)); }
void oracle_fe_add(int which, const fe *a, const fe *b, fe *o) { oracle_init(); fe_add(o, a, b, sel(which)); }
void oracle_fe_sub(int which, const fe *a, const fe *b, fe *o) { oracle_init(); fe_sub(o, a, b, sel(which)); }
void oracle_fe_inv(int which, const fe *a, fe *o) { oracle_init(); fe_inv(o, a, sel(which)); }
void oracle_fe_from_canonical(int which, const uint64_t c[4], fe *o) { oracle_init(); fe_from_canonical(o, c, sel(which)); }
void oracle_fe_to_canonical(int which, const fe *a, uint64_t c[4]) { oracle_init(); fe_to_canonical(c, a, sel(which)); }
void oracle_fe_pow(int which, const fe *a, const uint64_t e[4], fe *o) { oracle_init(); fe_pow(o, a, e, 4, sel(which)); }
/* constants: 0=R 1=R2 2=modulus 3=ROOT_OF_UNITY 4=ROOT_OF_UNITY_INV 5=ZETA; inv returned separately */
void oracle_constant(int which, int id, uint64_t out[4]) {
    oracle_init();
    const fparams *F = sel(which);
    switch (id) {
    case 0: memcpy(out, F->r.l, 32); break;
    case 1: memcpy(out, F->r2.l, 32); break;
    case 2: memcpy(out, F->p, 32); break;
    case 3: memcpy(out, FR_ROOT_OF_UNITY.l, 32); break;
    case 4: memcpy(out, FR_ROOT_OF_UNITY_INV.l, 32); break;
    case 5: memcpy(out, FR_ZETA.l, 32); break;
    default: memset(out, 0, 32);
    }
}
uint64_t oracle_inv64(int which) { oracle_init(); return sel(which)->inv; }

/* ------------------------------------------------------------------ G1 */
/* halo2curves bn256::G1 group law; any correct formulas give the same group
 * element (SURVEY.md App. A last bullet).  y^2 = x^3 + 3, a = 0. */

static inline int g1j_is_identity(const g1j *p) { return fe_is_zero(&p->z); }
static inline int g1a_is_identity(const g1a *p) { return fe_is_zero(&p->x) && fe_is_zero(&p->y); }
static inline void g1j_set_identity(g1j *p) { memset(p, 0, sizeof(*p)); p->y = FQ.r; } /* (0,1,0) */

static void g1j_from_affine(g1j *o, const g1a *a) {
    if (g1a_is_identity(a)) { g1j_set_identity(o); return; }
    o->x = a->x; o->y = a->y; o->z = FQ.r;
}

/* Curve::double (arithmetic.rs:48): dbl-2009-l */
static void g1j_double(g1j *o, const g1j *p) {
    if (g1j_is_identity(p)) { g1j_set_identity(o); return; }
    const fparams *F = &FQ;
    fe a, b, c, d, e, f, t, x3, y3, z3;
    fe_sqr(&a, &p->x, F);
    fe_sqr(&b, &p->y, F);
    fe_sqr(&c, &b, F);
    fe_add(&d, &p->x, &b, F);
    fe_sqr(&d, &d, F);
    fe_sub(&d, &d, &a, F);
    fe_sub(&d, &d, &c, F);
    fe_dbl(&d, &d, F);
    fe_dbl(&e, &a, F);
    fe_add(&e, &e, &a, F);
    fe_sqr(&f, &e, F);
    fe_mul(&z3, &p->z, &p->y, F);
    fe_dbl(&z3, &z3, F);
    fe_dbl(&t, &d, F);
    fe_sub(&x3, &f, &t, F);
    fe_dbl(&c, &c, F);
    fe_dbl(&c, &c, F);
    fe_dbl(&c, &c, F);
    fe_sub(&t, &d, &x3, F);
    fe_mul(&y3, &e, &t, F);
    fe_sub(&y3, &y3, &c, F);
    o->x = x3; o->y = y3; o->z = z3;
}

/* G1 + G1 (arithmetic.rs:77,98,153): add-2007-bl with exceptional cases */
static void g1j_add(g1j *o, const g1j *p, const g1j *q) {
    if (g1j_is_identity(p)) { *o = *q; return; }
    if (g1j_is_identity(q)) { *o = *p; return; }
    const fparams *F = &FQ;
    fe z1z1, z2z2, u1, u2, s1, s2, h, i, j, r, v, t, x3, y3, z3;
    fe_sqr(&z1z1, &p->z, F);
    fe_sqr(&z2z2, &q->z, F);
    fe_mul(&u1, &p->x, &z2z2, F);
    fe_mul(&u2, &q->x, &z1z1, F);
    fe_mul(&s1, &p->y, &q->z, F);
    fe_mul(&s1, &s1, &z2z2, F);
    fe_mul(&s2, &q->y, &p->z, F);
    fe_mul(&s2, &s2, &z1z1, F);
    if (fe_eq(&u1, &u2)) {
        if (fe_eq(&s1, &s2)) { g1j_double(o, p); return; }
        g1j_set_identity(o);
        return;
    }
    fe_sub(&h, &u2, &u1, F);
    fe_dbl(&i, &h, F);
    fe_sqr(&i, &i, F);
    fe_mul(&j, &h, &i, F);
    fe_sub(&r, &s2, &s1, F);
    fe_dbl(&r, &r, F);
    fe_mul(&v, &u1, &i, F);
    fe_sqr(&x3, &r, F);
    fe_sub(&x3, &x3, &j, F);
    fe_sub(&x3, &x3, &v, F);
    fe_sub(&x3, &x3, &v, F);
    fe_sub(&t, &v, &x3, F);
    fe_mul(&y3, &r, &t, F);
    fe_mul(&t, &s1, &j, F);
    fe_dbl(&t, &t, F);
    fe_sub(&y3, &y3, &t, F);
    fe_add(&z3, &p->z, &q->z, F);
    fe_sqr(&z3, &z3, F);
    fe_sub(&z3, &z3, &z1z1, F);
    fe_sub(&z3, &z3, &z2z2, F);
    fe_mul(&z3, &z3, &h, F);
    o->x = x3; o->y = y3; o->z = z3;
}

/* G1 += G1Affine (arithmetic.rs:64,74): madd-2007-bl with exceptional cases */
static void g1j_add_mixed(g1j *o, const g1j *p, const g1a *q) {
    if (g1a_is_identity(q)) { *o = *p; return; }
    if (g1j_is_identity(p)) { g1j_from_affine(o, q); return; }
    const fparams *F = &FQ;
    fe z1z1, u2, s2, h, hh, i, j, r, v, t, x3, y3, z3;
    fe_sqr(&z1z1, &p->z, F);
    fe_mul(&u2, &q->x, &z1z1, F);
    fe_mul(&s2, &q->y, &p->z, F);
    fe_mul(&s2, &s2, &z1z1, F);
    if (fe_eq(&p->x, &u2)) {
        if (fe_eq(&p->y, &s2)) { g1j_double(o, p); return; }
        g1j_set_identity(o);
        return;
    }
    fe_sub(&h, &u2, &p->x, F);
    fe_sqr(&hh, &h, F);
    fe_dbl(&i, &hh, F);
    fe_dbl(&i, &i, F);
    fe_mul(&j, &h, &i, F);
    fe_sub(&r, &s2, &p->y, F);
    fe_dbl(&r, &r, F);
    fe_mul(&v, &p->x, &i, F);
    fe_sqr(&x3, &r, F);
    fe_sub(&x3, &x3, &j, F);
    fe_sub(&x3, &x3, &v, F);
    fe_sub(&x3, &x3, &v, F);
    fe_sub(&t, &v, &x3, F);
    fe_mul(&y3, &r, &t, F);
    fe_mul(&t, &p->y, &j, F);
    fe_dbl(&t, &t, F);
    fe_sub(&y3, &y3, &t, F);
    fe_add(&z3, &p->z, &h, F);
    fe_sqr(&z3, &z3, F);
    fe_sub(&z3, &z3, &z1z1, F);
    fe_sub(&z3, &z3, &hh, F);
    o->x = x3; o->y = y3; o->z = z3;
}

/* G1Affine + G1Affine -> G1 (arithmetic.rs:62) */
static void g1_add_affine_affine(g1j *o, const g1a *a, const g1a *b) {
    g1j t;
    g1j_from_affine(&t, a);
    g1j_add_mixed(o, &t, b);
}

/* Curve::to_affine; identity -> (0,0) */
static void g1j_to_affine(g1a *o, const g1j *p) {
    if (g1j_is_identity(p)) { memset(o, 0, sizeof(*o)); return; }
    const fparams *F = &FQ;
    fe zi, zi2, zi3;
    fe_inv(&zi, &p->z, F);
    fe_sqr(&zi2, &zi, F);
    fe_mul(&zi3, &zi2, &zi, F);
    fe_mul(&o->x, &p->x, &zi2, F);
    fe_mul(&o->y, &p->y, &zi3, F);
}

static int g1a_on_curve(const g1a *p) {
    if (g1a_is_identity(p)) return 1;
    fe y2, x3;
    fe_sqr(&y2, &p->y, &FQ);
    fe_sqr(&x3, &p->x, &FQ);
    fe_mul(&x3, &x3, &p->x, &FQ);
    fe_add(&x3, &x3, &FQ_B3, &FQ);
    return fe_eq(&y2, &x3);
}

/* scalar (canonical 4xu64) * point, double-and-add */
static void g1j_mul_canonical(g1j *o, const g1j *p, const uint64_t e[4]) {
    g1j acc;
    g1j_set_identity(&acc);
    for (int i = 255; i >= 0; i--) {
        g1j_double(&acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) g1j_add(&acc, &acc, p);
    }
    *o = acc;
}

void oracle_g1_to_affine(const g1j *p, g1a *o) { oracle_init(); g1j_to_affine(o, p); }
void oracle_g1_add(const g1j *p, const g1j *q, g1j *o) { oracle_init(); g1j_add(o, p, q); }
void oracle_g1_add_mixed(const g1j *p, const g1a *q, g1j *o) { oracle_init(); g1j_add_mixed(o, p, q); }
void oracle_g1_double(const g1j *p, g1j *o) { oracle_init(); g1j_double(o, p); }
int oracle_g1_on_curve(const g1a *p) { oracle_init(); return g1a_on_curve(p); }
/* scalar in Montgomery form (an Fr element), as `G1 * Fr` in the reference */
void oracle_g1_mul(const g1a *p, const fe *scalar_mont, g1j *o) {
    oracle_init();
    uint64_t e[4];
    fe_to_canonical(e, scalar_mont, &FR);
    g1j t;
    g1j_from_affine(&t, p);
    g1j_mul_canonical(o, &t, e);
}

/* ------------------------------------------------------------------ MSM */

/* get_at (arithmetic.rs:24-42) */
static inline size_t get_at(size_t segment, size_t c, const uint8_t bytes[32]) {
    size_t skip_bits = segment * c;
    size_t skip_bytes = skip_bits / 8;
    if (skip_bytes >= 32) return 0;
    uint8_t v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < 8 && skip_bytes + i < 32; i++) v[i] = bytes[skip_bytes + i];
    uint64_t tmp;
    memcpy(&tmp, v, 8); /* little-endian host */
    tmp >>= skip_bits - skip_bytes * 8;
    tmp = tmp % ((uint64_t)1 << c);
    return (size_t)tmp;
}

/* window width rule (arithmetic.rs:16-22) */
size_t oracle_window_c(size_t m) {
    if (m < 4) return 1;
    if (m < 32) return 3;
    return (size_t)ceil(log((double)(uint32_t)m));
}

/* Bucket state machine (arithmetic.rs:51-80) */
enum { B_NONE = 0, B_AFFINE = 1, B_PROJ = 2 };
typedef struct { int tag; g1a a; g1j p; } bucket_t;

/* multiexp_serial (arithmetic.rs:13-101) */
static void multiexp_serial(const fe *coeffs, const g1a *bases, size_t m, g1j *acc) {
    uint8_t *repr = (uint8_t *)malloc(m * 32 + 1);
    for (size_t i = 0; i < m; i++) fe_to_canonical((uint64_t *)(repr + 32 * i), &coeffs[i], &FR); /* :14 */
    size_t c = oracle_window_c(m);                                                                 /* :16-22 */
    size_t segments = 256 / c + 1;                                                                 /* :44 */
    size_t nb = ((size_t)1 << c) - 1;
    bucket_t *buckets = (bucket_t *)malloc(nb * sizeof(bucket_t));
    for (size_t seg = segments; seg-- > 0;) {                                                      /* :46 */
        for (size_t k = 0; k < c; k++) g1j_double(acc, acc);                                       /* :47-49 */
        for (size_t b = 0; b < nb; b++) buckets[b].tag = B_NONE;                                   /* :82 */
        for (size_t i = 0; i < m; i++) {                                                           /* :84-89 */
            size_t d = get_at(seg, c, repr + 32 * i);
            if (d != 0) {
                bucket_t *bk = &buckets[d - 1];
                if (bk->tag == B_NONE) { bk->a = bases[i]; bk->tag = B_AFFINE; }                   /* :61 */
                else if (bk->tag == B_AFFINE) { g1_add_affine_affine(&bk->p, &bk->a, &bases[i]); bk->tag = B_PROJ; } /* :62 */
                else g1j_add_mixed(&bk->p, &bk->p, &bases[i]);                                     /* :63-66 */
            }
        }
        g1j running;
        g1j_set_identity(&running);                                                                /* :95 */
        for (size_t b = nb; b-- > 0;) {                                                            /* :96-99 */
            if (buckets[b].tag == B_AFFINE) g1j_add_mixed(&running, &running, &buckets[b].a);      /* :73-76 */
            else if (buckets[b].tag == B_PROJ) g1j_add(&running, &running, &buckets[b].p);         /* :77 */
            g1j_add(acc, acc, &running);                                                           /* :98 */
        }
    }
    free(buckets);
    free(repr);
}

typedef struct { const fe *coeffs; const g1a *bases; size_t m; g1j acc; } msm_job;
static void *msm_thread(void *arg) {
    msm_job *j = (msm_job *)arg;
    g1j_set_identity(&j->acc);
    multiexp_serial(j->coeffs, j->bases, j->m, &j->acc);
    return NULL;
}

/* best_multiexp (arithmetic.rs:132-159); num_threads stands for rayon::current_num_threads() */
int oracle_best_multiexp(const fe *coeffs, const g1a *bases, size_t n, int num_threads, g1j *out) {
    oracle_init();
    if (num_threads < 1) num_threads = 1;
    g1j_set_identity(out);
    if (n > (size_t)num_threads) {
        size_t chunk = n / (size_t)num_threads;                      /* :137 */
        size_t num_chunks = (n + chunk - 1) / chunk;                 /* :138 chunks(chunk).len() */
        msm_job *jobs = (msm_job *)malloc(num_chunks * sizeof(msm_job));
        pthread_t *th = (pthread_t *)malloc(num_chunks * sizeof(pthread_t));
        for (size_t k = 0; k < num_chunks; k++) {
            size_t s = k * chunk, e = s + chunk > n ? n : s + chunk;
            jobs[k].coeffs = coeffs + s; jobs[k].bases = bases + s; jobs[k].m = e - s;
            pthread_create(&th[k], NULL, msm_thread, &jobs[k]);     /* :148 scope.spawn */
        }
        for (size_t k = 0; k < num_chunks; k++) pthread_join(th[k], NULL);
        for (size_t k = 0; k < num_chunks; k++) g1j_add(out, out, &jobs[k].acc); /* :153 */
        free(th);
        free(jobs);
    } else {
        multiexp_serial(coeffs, bases, n, out);                      /* :155-157 */
    }
    return 0;
}

/* naive sum_i s_i*P_i by per-term double-and-add: an internal cross-check, not the reference algorithm */
int oracle_naive_multiexp(const fe *coeffs, const g1a *bases, size_t n, g1j *out) {
    oracle_init();
    g1j_set_identity(out);
    for (size_t i = 0; i < n; i++) {
        g1j t;
        oracle_g1_mul(&bases[i], &coeffs[i], &t);
        g1j_add(out, out, &t);
    }
    return 0;
}

/* ------------------------------------------------------------------ NTT */

static size_t bitreverse(size_t n, size_t l) {  /* arithmetic.rs:172-179 */
    size_t r = 0;
    for (size_t i = 0; i < l; i++) { r = (r << 1) | (n & 1); n >>= 1; }
    return r;
}

static uint32_t log2_floor(size_t num) {         /* arithmetic.rs:390-400 */
    uint32_t pow = 0;
    while (((size_t)1 << (pow + 1)) <= num) pow++;
    return pow;
}

static inline void butterfly_layer(fe *left, fe *right, size_t half, size_t twiddle_chunk, const fe *tw) {
    /* arithmetic.rs:209-226 and :255-272: i = 0 has twiddle one */
    fe t = right[0];
    fe_sub(&right[0], &left[0], &t, &FR);
    fe_add(&left[0], &left[0], &t, &FR);
    for (size_t i = 1; i < half; i++) {
        fe_mul(&t, &right[i], &tw[i * twiddle_chunk], &FR);
        fe_sub(&right[i], &left[i], &t, &FR);
        fe_add(&left[i], &left[i], &t, &FR);
    }
}

typedef struct { fe *a; size_t n; size_t twiddle_chunk; const fe *tw; int depth; } rb_job;
static void recursive_butterfly(fe *a, size_t n, size_t twiddle_chunk, const fe *tw, int depth);
static void *rb_thread(void *arg) {
    rb_job *j = (rb_job *)arg;
    recursive_butterfly(j->a, j->n, j->twiddle_chunk, j->tw, j->depth);
    return NULL;
}

/* recursive_butterfly_arithmetic (arithmetic.rs:237-274); rayon::join modelled by a
 * thread per right half while depth > 0 */
static void recursive_butterfly(fe *a, size_t n, size_t twiddle_chunk, const fe *tw, int depth) {
    if (n == 2) {
        fe t = a[1];
        fe_sub(&a[1], &a[0], &t, &FR);
        fe_add(&a[0], &a[0], &t, &FR);
        return;
    }
    fe *left = a, *right = a + n / 2;
    if (depth > 0) {
        pthread_t th;
        rb_job j = {right, n / 2, twiddle_chunk * 2, tw, depth - 1};
        pthread_create(&th, NULL, rb_thread, &j);
        recursive_butterfly(left, n / 2, twiddle_chunk * 2, tw, depth - 1);
        pthread_join(th, NULL);
    } else {
        recursive_butterfly(left, n / 2, twiddle_chunk * 2, tw, 0);
        recursive_butterfly(right, n / 2, twiddle_chunk * 2, tw, 0);
    }
    butterfly_layer(left, right, n / 2, twiddle_chunk, tw);
}

/* best_fft (arithmetic.rs:171-234) for G = Fr; num_threads = rayon::current_num_threads() */
int oracle_best_fft(fe *a, const fe *omega, uint32_t log_n, int num_threads) {
    oracle_init();
    if (num_threads < 1) num_threads = 1;
    uint32_t log_threads = log2_floor((size_t)num_threads);
    size_t n = (size_t)1 << log_n;
    for (size_t k = 0; k < n; k++) {                 /* :186-191 */
        size_t rk = bitreverse(k, log_n);
        if (k < rk) { fe t = a[rk]; a[rk] = a[k]; a[k] = t; }
    }
    size_t nt = n / 2;
    fe *tw = (fe *)malloc((nt ? nt : 1) * sizeof(fe));
    fe w = FR.r;                                     /* :194-200 */
    for (size_t i = 0; i < nt; i++) { tw[i] = w; fe_mul(&w, &w, omega, &FR); }
    if (log_n <= log_threads) {                      /* :202-230 */
        size_t chunk = 2, twiddle_chunk = n / 2;
        for (uint32_t s = 0; s < log_n; s++) {
            for (size_t b = 0; b < n; b += chunk) butterfly_layer(a + b, a + b + chunk / 2, chunk / 2, twiddle_chunk, tw);
            chunk *= 2;
            twiddle_chunk /= 2;
        }
    } else {                                         /* :232 */
        recursive_butterfly(a, n, 1, tw, (int)log_threads);
    }
    free(tw);
    return 0;
}

typedef struct { fe *a; size_t start, len; const fe *c0, *c1; int mode; } scale_job;
static void *scale_thread(void *arg) {
    scale_job *j = (scale_job *)arg;
    if (j->mode == 0) {          /* uniform scale (domain.rs:355-360) */
        for (size_t i = 0; i < j->len; i++) fe_mul(&j->a[i], &j->a[i], j->c0, &FR);
    } else {                     /* distribute_powers_zeta inner loop (domain.rs:341-350) */
        size_t index = j->start;
        for (size_t i = 0; i < j->len; i++, index++) {
            size_t r = index % 3;
            if (r == 1) fe_mul(&j->a[i], &j->a[i], j->c0, &FR);
            else if (r == 2) fe_mul(&j->a[i], &j->a[i], j->c1, &FR);
        }
    }
    return NULL;
}

/* parallelize (arithmetic.rs:371-388) driving one of the two pointwise bodies above */
static void parallelize_scale(fe *a, size_t n, int num_threads, const fe *c0, const fe *c1, int mode) {
    if (n == 0) return;
    if (num_threads < 1) num_threads = 1;
    size_t chunk = n / (size_t)num_threads;
    if (chunk < (size_t)num_threads) chunk = 1;
    if (num_threads == 1 || n < 4096) {   /* same values; avoid spawning n threads for tiny inputs */
        scale_job j = {a, 0, n, c0, c1, mode};
        scale_thread(&j);
        return;
    }
    size_t nchunks = (n + chunk - 1) / chunk;
    scale_job *jobs = (scale_job *)malloc(nchunks * sizeof(scale_job));
    pthread_t *th = (pthread_t *)malloc(nchunks * sizeof(pthread_t));
    for (size_t k = 0; k < nchunks; k++) {
        size_t s = k * chunk, e = s + chunk > n ? n : s + chunk;
        scale_job j = {a + s, s, e - s, c0, c1, mode};
        jobs[k] = j;
        pthread_create(&th[k], NULL, scale_thread, &jobs[k]);
    }
    for (size_t k = 0; k < nchunks; k++) pthread_join(th[k], NULL);
    free(th);
    free(jobs);
}

/* EvaluationDomain::ifft (poly/domain.rs:353-361) */
int oracle_ifft(fe *a, const fe *omega_inv, uint32_t log_n, const fe *divisor, int num_threads) {
    oracle_best_fft(a, omega_inv, log_n, num_threads);
    parallelize_scale(a, (size_t)1 << log_n, num_threads, divisor, NULL, 0);
    return 0;
}

/* EvaluationDomain (poly/domain.rs:18-34) constants for G = Fr, built as in new() (:39-142) */
typedef struct {
    uint64_t n;
    uint32_t k, extended_k;
    uint32_t t_len;
    uint64_t quotient_poly_degree;
    fe omega, omega_inv, extended_omega, extended_omega_inv;
    fe g_coset, g_coset_inv, ifft_divisor, extended_ifft_divisor, barycentric_weight;
} oracle_domain;

/* EvaluationDomain::new (poly/domain.rs:39-142); t_evaluations (inverted, :84-124) written to
 * t_eval_out (capacity >= 2^(extended_k-k)) when non-NULL */
int oracle_domain_new(uint32_t j, uint32_t k, oracle_domain *d, fe *t_eval_out) {
    oracle_init();
    memset(d, 0, sizeof(*d));
    d->quotient_poly_degree = (uint64_t)(j - 1);                       /* :41 */
    d->n = (uint64_t)1 << k;                                           /* :44 */
    d->k = k;
    uint32_t ek = k;
    while (((uint64_t)1 << ek) < d->n * d->quotient_poly_degree) ek++; /* :49-52 */
    d->extended_k = ek;
    if (ek > FR_S) return -1;
    fe eo = FR_ROOT_OF_UNITY;                                          /* :54-61 */
    for (uint32_t i = ek; i < FR_S; i++) fe_sqr(&eo, &eo, &FR);
    d->extended_omega = eo;
    fe o = eo;                                                         /* :70-73 */
    for (uint32_t i = k; i < ek; i++) fe_sqr(&o, &o, &FR);
    d->omega = o;
    d->g_coset = FR_ZETA;                                              /* :81 */
    fe_sqr(&d->g_coset_inv, &FR_ZETA, &FR);                            /* :82 */
    fe_inv(&d->omega_inv, &d->omega, &FR);                             /* :124 (batch_invert == per-element inverse) */
    fe_inv(&d->extended_omega_inv, &d->extended_omega, &FR);
    fe t;
    fe_from_u64(&t, (uint64_t)1 << k, &FR);                            /* :109 */
    fe_inv(&d->ifft_divisor, &t, &FR);
    fe_from_u64(&t, (uint64_t)1 << ek, &FR);                           /* :110 */
    fe_inv(&d->extended_ifft_divisor, &t, &FR);
    fe_from_u64(&t, d->n, &FR);                                        /* :114 */
    fe_inv(&d->barycentric_weight, &t, &FR);
    d->t_len = (uint32_t)1 << (ek - k);
    if (t_eval_out) {                                                  /* :84-107 */
        uint64_t e[4] = {d->n, 0, 0, 0};
        fe orig, step, cur;
        fe_pow(&orig, &FR_ZETA, e, 4, &FR);
        fe_pow(&step, &d->extended_omega, e, 4, &FR);
        cur = orig;
        uint32_t cnt = 0;
        do {
            if (cnt >= d->t_len) return -2;
            t_eval_out[cnt++] = cur;
            fe_mul(&cur, &cur, &step, &FR);
        } while (!fe_eq(&cur, &orig));
        if (cnt != d->t_len) return -3;                                /* :98 */
        for (uint32_t i = 0; i < cnt; i++) {
            fe_sub(&t_eval_out[i], &t_eval_out[i], &FR.r, &FR);        /* :101-103 */
            fe_inv(&t_eval_out[i], &t_eval_out[i], &FR);               /* :117-124 */
        }
    }
    return 0;
}

/* distribute_powers_zeta (poly/domain.rs:335-351) */
int oracle_distribute_powers_zeta(const oracle_domain *d, fe *a, size_t len, int into_coset, int num_threads) {
    const fe *c0 = into_coset ? &d->g_coset : &d->g_coset_inv;
    const fe *c1 = into_coset ? &d->g_coset_inv : &d->g_coset;
    parallelize_scale(a, len, num_threads, c0, c1, 1);
    return 0;
}

/* lagrange_to_coeff (poly/domain.rs:226-236): a has 2^k entries, in place */
int oracle_lagrange_to_coeff(const oracle_domain *d, fe *a, int num_threads) {
    return oracle_ifft(a, &d->omega_inv, d->k, &d->ifft_divisor, num_threads);
}

/* coeff_to_extended (poly/domain.rs:240-254): a_in 2^k entries -> out 2^extended_k entries */
int oracle_coeff_to_extended(const oracle_domain *d, const fe *a_in, fe *out, int num_threads) {
    size_t n = (size_t)1 << d->k, en = (size_t)1 << d->extended_k;
    memcpy(out, a_in, n * sizeof(fe));
    oracle_distribute_powers_zeta(d, out, n, 1, num_threads);          /* :246 */
    memset(out + n, 0, (en - n) * sizeof(fe));                         /* :247 */
    return oracle_best_fft(out, &d->extended_omega, d->extended_k, num_threads); /* :248 */
}

/* extended_to_coeff (poly/domain.rs:281-303): a has 2^extended_k entries, in place; the
 * caller keeps the first n*quotient_poly_degree (truncate, :299-300) */
int oracle_extended_to_coeff(const oracle_domain *d, fe *a, int num_threads) {
    oracle_ifft(a, &d->extended_omega_inv, d->extended_k, &d->extended_ifft_divisor, num_threads); /* :285-290 */
    return oracle_distribute_powers_zeta(d, a, (size_t)1 << d->extended_k, 0, num_threads);        /* :294 */
}

/* divide_by_vanishing_poly (poly/domain.rs:307-326) */
int oracle_divide_by_vanishing_poly(const oracle_domain *d, const fe *t_evaluations, fe *a) {
    size_t en = (size_t)1 << d->extended_k;
    for (size_t i = 0; i < en; i++) fe_mul(&a[i], &a[i], &t_evaluations[i % d->t_len], &FR);      /* :315-320 */
    return 0;
}

/* ------------------------------------------------------------------ KZG params */

/* ParamsKZG::setup (poly/kzg/commitment.rs:61-129) with the secret s supplied (Montgomery Fr)
 * instead of drawn from an rng: g[i] = [s^i]G1, g_lagrange[i] = [l_i(s)]G1. */
int oracle_kzg_setup(uint32_t k, const fe *s, g1a *g, g1a *g_lagrange) {
    oracle_init();
    if (k > FR_S) return -1;                                            /* :64 */
    size_t n = (size_t)1 << k;
    g1a gen;
    fe_from_u64(&gen.x, 1, &FQ);
    fe_from_u64(&gen.y, 2, &FQ);
    /* :71-87 */
    fe cur = FR.r;
    for (size_t i = 0; i < n; i++) {
        g1j t;
        oracle_g1_mul(&gen, &cur, &t);
        g1j_to_affine(&g[i], &t);
        fe_mul(&cur, &cur, s, &FR);
    }
    /* :89-104 */
    fe root;
    fe_inv(&root, &FR_ROOT_OF_UNITY_INV, &FR);
    for (uint32_t i = k; i < FR_S; i++) fe_sqr(&root, &root, &FR);
    fe nfe, n_inv, sn, mult;
    fe_from_u64(&nfe, (uint64_t)n, &FR);
    fe_inv(&n_inv, &nfe, &FR);
    uint64_t e[4] = {(uint64_t)n, 0, 0, 0};
    fe_pow(&sn, s, e, 4, &FR);
    fe_sub(&sn, &sn, &FR.r, &FR);
    fe_mul(&mult, &sn, &n_inv, &FR);
    fe root_pow = FR.r;
    for (size_t i = 0; i < n; i++) {
        fe d, scalar;
        fe_sub(&d, s, &root_pow, &FR);
        fe_inv(&d, &d, &FR);
        fe_mul(&scalar, &mult, &root_pow, &FR);
        fe_mul(&scalar, &scalar, &d, &FR);
        g1j t;
        oracle_g1_mul(&gen, &scalar, &t);
        g1j_to_affine(&g_lagrange[i], &t);
        fe_mul(&root_pow, &root_pow, &root, &FR);
    }
    return 0;
}

/* ParamsKZG::commit / commit_lagrange (poly/kzg/commitment.rs:281-292, :327-334): the blind is ignored */
int oracle_kzg_commit(const fe *poly, size_t size, const g1a *bases, size_t bases_len, int num_threads, g1j *out) {
    if (bases_len < size) return -1;                                    /* :290, :332 */
    return oracle_best_multiexp(poly, bases, size, num_threads, out);
}

/* ------------------------------------------------------------------ g_to_lagrange: best_fft over G = G1 */
/* Group for G1 (halo2curves): group_add / group_sub are point addition / subtraction, group_scale multiplies by an
 * Fr scalar.  Points stay Jacobian between layers, as C::Curve does in the reference. */
static void g1j_neg(g1j *o, const g1j *p) { *o = *p; fe_neg(&o->y, &p->y, &FQ); }

static void g1_butterfly_layer(g1j *left, g1j *right, size_t half, size_t twiddle_chunk, const fe *tw) {
    /* arithmetic.rs:209-226 / :255-272 with G = G1: i = 0 has twiddle one */
    for (size_t i = 0; i < half; i++) {
        g1j t = right[i], nt;
        if (i) {
            uint64_t e[4];
            fe_to_canonical(e, &tw[i * twiddle_chunk], &FR);
            g1j_mul_canonical(&t, &right[i], e);                        /* t.group_scale(&twiddles[..]) */
        }
        g1j_neg(&nt, &t);
        g1j_add(&right[i], &left[i], &nt);                              /* b = a - t */
        g1j_add(&left[i], &left[i], &t);                                /* a = a + t */
    }
}

typedef struct { g1j *a; size_t n, lo, hi, chunk, twiddle_chunk; const fe *tw; } g1_layer_job;
static void *g1_layer_thread(void *arg) {
    g1_layer_job *j = (g1_layer_job *)arg;
    for (size_t b = j->lo; b < j->hi; b += j->chunk)
        g1_butterfly_layer(j->a + b, j->a + b + j->chunk / 2, j->chunk / 2, j->twiddle_chunk, j->tw);
    return NULL;
}

/* best_fft with G = G1 (arithmetic.rs:171-234), iterative form (:202-230), in place on Jacobian points: bit-reversal, the serial
 * twiddle scan, then the butterfly layers.  The recursive form (:232) performs the same group operations in another order, and only
 * the group elements are defined.  num_threads only splits each layer's blocks over threads (test speed). */
static void g1_fft_in_place(g1j *a, uint32_t k, const fe *omega, int num_threads) {
    size_t n = (size_t)1 << k;
    for (size_t i = 0; i < n; i++) {                                     /* best_fft :186-191 */
        size_t ri = bitreverse(i, k);
        if (i < ri) { g1j t = a[ri]; a[ri] = a[i]; a[i] = t; }
    }
    size_t nt = n / 2;
    fe *tw = (fe *)malloc((nt ? nt : 1) * sizeof(fe));
    fe w = FR.r;
    for (size_t i = 0; i < nt; i++) { tw[i] = w; fe_mul(&w, &w, omega, &FR); }   /* :194-200 */
    size_t chunk = 2, twiddle_chunk = n / 2;
    for (uint32_t s = 0; s < k; s++) {
        size_t blocks = n / chunk;
        int T = (size_t)num_threads < blocks ? num_threads : (int)blocks;
        if (T <= 1) {
            for (size_t b = 0; b < n; b += chunk) g1_butterfly_layer(a + b, a + b + chunk / 2, chunk / 2, twiddle_chunk, tw);
        } else {
            pthread_t th[64];
            g1_layer_job jobs[64];
            if (T > 64) T = 64;
            for (int t = 0; t < T; t++) {
                size_t b0 = blocks * t / T, b1 = blocks * (t + 1) / T;
                jobs[t] = (g1_layer_job){a, n, b0 * chunk, b1 * chunk, chunk, twiddle_chunk, tw};
                pthread_create(&th[t], NULL, g1_layer_thread, &jobs[t]);
            }
            for (int t = 0; t < T; t++) pthread_join(th[t], NULL);
        }
        chunk *= 2;
        twiddle_chunk /= 2;
    }
    free(tw);
}

/* pub fn best_fft<G: Group>(a: &mut [G], omega: G::Scalar, log_n: u32) for G = bn256::G1, on 2^log_n Jacobian points in place */
int oracle_best_fft_g1(g1j *a, const fe *omega, uint32_t log_n, int num_threads) {
    oracle_init();
    if (log_n > FR_S) return -1;
    if (num_threads < 1) num_threads = 1;
    g1_fft_in_place(a, log_n, omega, num_threads);
    return 0;
}

/* g_to_lagrange (arithmetic.rs:277-301): inverse FFT of the coefficient-basis SRS points, scaled by 1/n, normalised. */
int oracle_g_to_lagrange(const g1a *g, uint32_t k, g1a *g_lagrange, int num_threads) {
    oracle_init();
    if (k > FR_S) return -1;
    size_t n = (size_t)1 << k;
    if (num_threads < 1) num_threads = 1;
    fe n_inv, two, omega_inv = FR_ROOT_OF_UNITY_INV;                     /* :278-282 */
    fe_from_u64(&two, 2, &FR);
    fe_inv(&two, &two, &FR);
    { uint64_t e[4] = {k, 0, 0, 0}; fe_pow(&n_inv, &two, e, 4, &FR); }
    for (uint32_t i = k; i < FR_S; i++) fe_sqr(&omega_inv, &omega_inv, &FR);
    g1j *a = (g1j *)malloc(n * sizeof(g1j));
    for (size_t i = 0; i < n; i++) g1j_from_affine(&a[i], &g[i]);        /* g.to_curve() (commitment.rs:274) */
    g1_fft_in_place(a, k, &omega_inv, num_threads);                      /* :285 */
    uint64_t e[4];
    fe_to_canonical(e, &n_inv, &FR);
    for (size_t i = 0; i < n; i++) {                                     /* :286-290 then batch_normalize :292-298 */
        g1j t;
        g1j_mul_canonical(&t, &a[i], e);
        g1j_to_affine(&g_lagrange[i], &t);
    }
    free(a);
    return 0;
}

/* ------------------------------------------------------------------ synthetic inputs */

static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

/* 256-bit draw for element i: limb j = splitmix64(seed + 4*i + j); top 2 bits cleared, then
 * one conditional subtract brings it below the 254-bit modulus (value is taken as canonical) */
static void draw_mod(uint64_t out[4], uint64_t seed, uint64_t i, uint64_t attempt, const fparams *F) {
    for (int j = 0; j < 4; j++) out[j] = splitmix64(seed + 4 * i + (uint64_t)j + attempt * 0x632BE59BD9B4E019ULL);
    out[3] &= 0x3FFFFFFFFFFFFFFFULL;
    if (ge_p(out, F->p)) sub_p(out, F->p);
}

/* scalars: uniform-ish canonical value -> Montgomery */
void oracle_gen_scalars(uint64_t seed, size_t start, size_t n, fe *out) {
    oracle_init();
    for (size_t i = 0; i < n; i++) {
        uint64_t c[4];
        draw_mod(c, seed, start + i, 0, &FR);
        fe_from_canonical(&out[i], c, &FR);
    }
}

/* points: try-and-increment on y^2 = x^3 + 3; y = rhs^((q+1)/4); sign from bit 62 of the
 * attempt's 4th raw draw */
void oracle_gen_points(uint64_t seed, size_t start, size_t n, g1a *out) {
    oracle_init();
    uint64_t e[4];
    /* (q+1)/4 */
    u128 c = (u128)FQ.p[0] + 1;
    uint64_t t[4];
    t[0] = (uint64_t)c; c >>= 64;
    for (int i = 1; i < 4; i++) { c += FQ.p[i]; t[i] = (uint64_t)c; c >>= 64; }
    for (int i = 0; i < 4; i++) e[i] = (t[i] >> 2) | (i < 3 ? t[i + 1] << 62 : 0);
    for (size_t i = 0; i < n; i++) {
        for (uint64_t attempt = 0;; attempt++) {
            uint64_t xc[4];
            draw_mod(xc, seed, start + i, attempt, &FQ);
            uint64_t signdraw = splitmix64(seed + 4 * (start + i) + 3 + attempt * 0x632BE59BD9B4E019ULL);
            fe x, rhs, y, y2;
            fe_from_canonical(&x, xc, &FQ);
            fe_sqr(&rhs, &x, &FQ);
            fe_mul(&rhs, &rhs, &x, &FQ);
            fe_add(&rhs, &rhs, &FQ_B3, &FQ);
            fe_pow(&y, &rhs, e, 4, &FQ);
            fe_sqr(&y2, &y, &FQ);
            if (!fe_eq(&y2, &rhs)) continue;
            if ((signdraw >> 62) & 1) fe_neg(&y, &y, &FQ);
            out[i].x = x;
            out[i].y = y;
            break;
        }
    }
}

typedef struct { uint64_t seed; size_t start, n; void *out; int kind; } gen_job;
static void *gen_thread(void *arg) {
    gen_job *j = (gen_job *)arg;
    if (j->kind == 0) oracle_gen_scalars(j->seed, j->start, j->n, (fe *)j->out);
    else oracle_gen_points(j->seed, j->start, j->n, (g1a *)j->out);
    return NULL;
}
/* threaded wrapper: kind 0 = scalars, 1 = points */
void oracle_gen_parallel(int kind, uint64_t seed, size_t n, void *out, int num_threads) {
    oracle_init();
    if (num_threads < 1) num_threads = 1;
    gen_job *jobs = (gen_job *)malloc((size_t)num_threads * sizeof(gen_job));
    pthread_t *th = (pthread_t *)malloc((size_t)num_threads * sizeof(pthread_t));
    size_t per = (n + (size_t)num_threads - 1) / (size_t)num_threads;
    int used = 0;
    for (int k = 0; k < num_threads; k++) {
        size_t s = (size_t)k * per;
        if (s >= n) break;
        size_t e = s + per > n ? n : s + per;
        size_t esz = kind == 0 ? sizeof(fe) : sizeof(g1a);
        gen_job j = {seed, s, e - s, (char *)out + s * esz, kind};
        jobs[k] = j;
        pthread_create(&th[k], NULL, gen_thread, &jobs[k]);
        used++;
    }
    for (int k = 0; k < used; k++) pthread_join(th[k], NULL);
    free(th);
    free(jobs);
}

/* ------------------------------------------------------------------ evaluate_h (plonk/evaluation.rs:280-522) */
#include "../include/halo2hip.h"   /* the flattened ValueSource / Calculation / GraphEvaluator structures */

typedef struct {
    const h2hip_evalh_desc *d;
    const fe *const *fixed;   /* extended cosets */
    fe **advice, **instance;  /* extended cosets computed here */
    fe beta, gamma, theta, y;
    size_t size;
    int32_t rot_scale, isize;
} evalh_ctx;

/* get_rotation_idx (evaluation.rs:32-34) */
static inline size_t get_rotation_idx(size_t idx, int32_t rot, int32_t rot_scale, int32_t isize) {
    int64_t v = ((int64_t)idx + (int64_t)rot * rot_scale) % isize;
    if (v < 0) v += isize;
    return (size_t)v;
}

/* ValueSource::get (evaluation.rs:68-103) */
static fe vs_get(const evalh_ctx *c, const h2hip_graph *g, const h2hip_value_source *v, const size_t *rotations, const fe *inter,
                 const fe *previous) {
    switch (v->kind) {
    case H2HIP_VS_CONSTANT: return ((const fe *)g->constants)[v->a];
    case H2HIP_VS_INTERMEDIATE: return inter[v->a];
    case H2HIP_VS_FIXED: return c->fixed[v->a][rotations[v->b]];
    case H2HIP_VS_ADVICE: return c->advice[v->a][rotations[v->b]];
    case H2HIP_VS_INSTANCE: return c->instance[v->a][rotations[v->b]];
    case H2HIP_VS_CHALLENGE: return ((const fe *)c->d->challenges)[v->a];
    case H2HIP_VS_BETA: return c->beta;
    case H2HIP_VS_GAMMA: return c->gamma;
    case H2HIP_VS_THETA: return c->theta;
    case H2HIP_VS_Y: return c->y;
    default: return *previous; /* PreviousValue */
    }
}

/* GraphEvaluator::evaluate (evaluation.rs:708-749) with Calculation::evaluate (:129-178) */
static fe graph_evaluate(const evalh_ctx *c, const h2hip_graph *g, size_t *rotations, fe *inter, const fe *previous, size_t idx) {
    for (uint32_t r = 0; r < g->n_rotations; r++) rotations[r] = get_rotation_idx(idx, g->rotations[r], c->rot_scale, c->isize); /* :725-727 */
    for (uint32_t q = 0; q < g->n_calculations; q++) {                                                                        /* :730-745 */
        const h2hip_calculation *cl = &g->calculations[q];
        fe a = vs_get(c, g, &cl->x, rotations, inter, previous), b, out;
        switch (cl->op) {
        case H2HIP_CALC_ADD: b = vs_get(c, g, &cl->y, rotations, inter, previous); fe_add(&out, &a, &b, &FR); break;
        case H2HIP_CALC_SUB: b = vs_get(c, g, &cl->y, rotations, inter, previous); fe_sub(&out, &a, &b, &FR); break;
        case H2HIP_CALC_MUL: b = vs_get(c, g, &cl->y, rotations, inter, previous); fe_mul(&out, &a, &b, &FR); break;
        case H2HIP_CALC_SQUARE: fe_sqr(&out, &a, &FR); break;
        case H2HIP_CALC_DOUBLE: fe_dbl(&out, &a, &FR); break;
        case H2HIP_CALC_NEGATE: fe_neg(&out, &a, &FR); break;
        case H2HIP_CALC_HORNER: {                                                                                              /* :166-173 */
            fe factor = vs_get(c, g, &cl->y, rotations, inter, previous);
            out = a;
            for (uint32_t t = 0; t < cl->parts_count; t++) {
                fe part = vs_get(c, g, &g->parts[cl->parts_offset + t], rotations, inter, previous);
                fe_mul(&out, &out, &factor, &FR);
                fe_add(&out, &out, &part, &FR);
            }
            break;
        }
        default: out = a; /* Store */
        }
        inter[cl->target] = out;
    }
    if (g->n_calculations) return inter[g->calculations[g->n_calculations - 1].target];                                        /* :748-752 */
    fe z = {{0, 0, 0, 0}};
    return z;
}

static const fe *perm_column(const evalh_ctx *c, uint32_t j) {                                                                 /* :404-408 */
    uint32_t kind = c->d->perm_column_kind[j], col = c->d->perm_column_index[j];
    if (kind == H2HIP_ANY_ADVICE) return c->advice[col];
    if (kind == H2HIP_ANY_FIXED) return c->fixed[col];
    return c->instance[col];
}

int oracle_evaluate_h(const h2hip_evalh_desc *d, fe *values) {
    oracle_init();
    evalh_ctx c;
    memset(&c, 0, sizeof(c));
    c.d = d;
    c.size = (size_t)1 << d->extended_k;
    c.rot_scale = 1 << (d->extended_k - d->k);                                                                                 /* :295 */
    c.isize = (int32_t)c.size;
    c.fixed = (const fe *const *)d->fixed_cosets;
    memcpy(&c.beta, d->beta, 32); memcpy(&c.gamma, d->gamma, 32); memcpy(&c.theta, d->theta, 32); memcpy(&c.y, d->y, 32);
    oracle_domain dom;
    memset(&dom, 0, sizeof(dom));
    dom.k = d->k; dom.extended_k = d->extended_k; dom.n = (uint64_t)1 << d->k;
    memcpy(&dom.extended_omega, d->extended_omega, 32); memcpy(&dom.g_coset, d->g_coset, 32); memcpy(&dom.g_coset_inv, d->g_coset_inv, 32);
    const size_t size = c.size;
    /* :306-323 advice and instance cosets */
    c.advice = (fe **)calloc(d->n_advice + 1, sizeof(fe *));
    c.instance = (fe **)calloc(d->n_instance + 1, sizeof(fe *));
    for (uint32_t i = 0; i < d->n_advice; i++) { c.advice[i] = (fe *)malloc(size * sizeof(fe)); oracle_coeff_to_extended(&dom, (const fe *)d->advice_polys[i], c.advice[i], 1); }
    for (uint32_t i = 0; i < d->n_instance; i++) { c.instance[i] = (fe *)malloc(size * sizeof(fe)); oracle_coeff_to_extended(&dom, (const fe *)d->instance_polys[i], c.instance[i], 1); }
    const fe *l0 = (const fe *)d->l0, *l_last = (const fe *)d->l_last, *l_active = (const fe *)d->l_active_row;
    const fe one = FR.r;
    fe extended_omega; memcpy(&extended_omega, d->extended_omega, 32);
    size_t *rotations = (size_t *)malloc(4096 * sizeof(size_t));
    fe *inter = (fe *)malloc(65536 * sizeof(fe));
    /* :334-360 custom gates */
    for (size_t idx = 0; idx < size; idx++) {
        fe prev = values[idx];
        values[idx] = graph_evaluate(&c, &d->custom_gates, rotations, inter, &prev, idx);
    }
    /* :362-441 permutations */
    if (d->n_perm_sets) {
        fe zeta, delta, delta_start, beta_term = FR.r, t, u;
        memcpy(&zeta, d->zeta, 32); memcpy(&delta, d->delta, 32);
        fe_mul(&delta_start, &c.beta, &zeta, &FR);                                                                             /* :368 */
        const fe *const *z = (const fe *const *)d->perm_product_cosets;
        const fe *first = z[0], *last = z[d->n_perm_sets - 1];
        for (size_t idx = 0; idx < size; idx++) {
            size_t r_next = get_rotation_idx(idx, 1, c.rot_scale, c.isize);
            size_t r_last = get_rotation_idx(idx, d->last_rotation, c.rot_scale, c.isize);
            fe *value = &values[idx];
            /* l_0(X) * (1 - z_0(X)) = 0  :382-386 */
            fe_sub(&t, &one, &first[idx], &FR); fe_mul(&t, &t, &l0[idx], &FR);
            fe_mul(value, value, &c.y, &FR); fe_add(value, value, &t, &FR);
            /* l_last(X) * (z_l(X)^2 - z_l(X)) = 0  :387-393 */
            fe_mul(&t, &last[idx], &last[idx], &FR); fe_sub(&t, &t, &last[idx], &FR); fe_mul(&t, &t, &l_last[idx], &FR);
            fe_mul(value, value, &c.y, &FR); fe_add(value, value, &t, &FR);
            /* l_0(X) * (z_i(X) - z_{i-1}(omega^(last) X)) = 0  :394-404 */
            for (uint32_t s = 1; s < d->n_perm_sets; s++) {
                fe_sub(&t, &z[s][idx], &z[s - 1][r_last], &FR); fe_mul(&t, &t, &l0[idx], &FR);
                fe_mul(value, value, &c.y, &FR); fe_add(value, value, &t, &FR);
            }
            /* :405-438 */
            fe current_delta;
            fe_mul(&current_delta, &delta_start, &beta_term, &FR);
            for (uint32_t s = 0; s < d->n_perm_sets; s++) {
                uint32_t j0 = s * d->chunk_len, j1 = j0 + d->chunk_len > d->n_perm_columns ? d->n_perm_columns : j0 + d->chunk_len;
                fe left = z[s][r_next], right = z[s][idx];
                for (uint32_t j = j0; j < j1; j++) {
                    const fe *col = perm_column(&c, j);
                    fe_mul(&t, &c.beta, &((const fe *)d->perm_cosets[j])[idx], &FR); fe_add(&t, &t, &col[idx], &FR); fe_add(&t, &t, &c.gamma, &FR);
                    fe_mul(&left, &left, &t, &FR);
                }
                for (uint32_t j = j0; j < j1; j++) {
                    const fe *col = perm_column(&c, j);
                    fe_add(&u, &col[idx], &current_delta, &FR); fe_add(&u, &u, &c.gamma, &FR);
                    fe_mul(&right, &right, &u, &FR);
                    fe_mul(&current_delta, &current_delta, &delta, &FR);
                }
                fe_sub(&t, &left, &right, &FR); fe_mul(&t, &t, &l_active[idx], &FR);
                fe_mul(value, value, &c.y, &FR); fe_add(value, value, &t, &FR);
            }
            fe_mul(&beta_term, &beta_term, &extended_omega, &FR);                                                              /* :439 */
        }
    }
    /* :443-518 lookups */
    fe *product = (fe *)malloc(size * sizeof(fe)), *pin = (fe *)malloc(size * sizeof(fe)), *ptab = (fe *)malloc(size * sizeof(fe));
    for (uint32_t n = 0; n < d->n_lookups; n++) {
        oracle_coeff_to_extended(&dom, (const fe *)d->lookup_product_polys[n], product, 1);
        oracle_coeff_to_extended(&dom, (const fe *)d->lookup_permuted_input_polys[n], pin, 1);
        oracle_coeff_to_extended(&dom, (const fe *)d->lookup_permuted_table_polys[n], ptab, 1);
        const fe zero = {{0, 0, 0, 0}};
        for (size_t idx = 0; idx < size; idx++) {
            fe table_value = graph_evaluate(&c, &d->lookup_graphs[n], rotations, inter, &zero, idx);                          /* :466-480 */
            size_t r_next = get_rotation_idx(idx, 1, c.rot_scale, c.isize), r_prev = get_rotation_idx(idx, -1, c.rot_scale, c.isize);
            fe *value = &values[idx];
            fe a_minus_s, t, u, w;
            fe_sub(&a_minus_s, &pin[idx], &ptab[idx], &FR);
            /* l_0(X) * (1 - z(X)) = 0 */
            fe_sub(&t, &one, &product[idx], &FR); fe_mul(&t, &t, &l0[idx], &FR);
            fe_mul(value, value, &c.y, &FR); fe_add(value, value, &t, &FR);
            /* l_last(X) * (z(X)^2 - z(X)) = 0 */
            fe_mul(&t, &product[idx], &product[idx], &FR); fe_sub(&t, &t, &product[idx], &FR); fe_mul(&t, &t, &l_last[idx], &FR);
            fe_mul(value, value, &c.y, &FR); fe_add(value, value, &t, &FR);
            /* (1 - (l_last + l_blind)) * (z(wX)(a'+beta)(s'+gamma) - z(X) * table_value) = 0 */
            fe_add(&t, &pin[idx], &c.beta, &FR); fe_add(&u, &ptab[idx], &c.gamma, &FR);
            fe_mul(&t, &product[r_next], &t, &FR); fe_mul(&t, &t, &u, &FR);
            fe_mul(&w, &product[idx], &table_value, &FR); fe_sub(&t, &t, &w, &FR); fe_mul(&t, &t, &l_active[idx], &FR);
            fe_mul(value, value, &c.y, &FR); fe_add(value, value, &t, &FR);
            /* l_0(X) * (a'(X) - s'(X)) = 0 */
            fe_mul(&t, &a_minus_s, &l0[idx], &FR);
            fe_mul(value, value, &c.y, &FR); fe_add(value, value, &t, &FR);
            /* (1 - (l_last + l_blind)) * (a' - s') * (a'(X) - a'(w^-1 X)) = 0 */
            fe_sub(&t, &pin[idx], &pin[r_prev], &FR); fe_mul(&t, &a_minus_s, &t, &FR); fe_mul(&t, &t, &l_active[idx], &FR);
            fe_mul(value, value, &c.y, &FR); fe_add(value, value, &t, &FR);
        }
    }
    free(product); free(pin); free(ptab); free(rotations); free(inter);
    for (uint32_t i = 0; i < d->n_advice; i++) free(c.advice[i]);
    for (uint32_t i = 0; i < d->n_instance; i++) free(c.instance[i]);
    free(c.advice); free(c.instance);
    return 0;
}
