// Host-side fuzz of csrc/fieldu.h + csrc/ecu.h (the unsaturated arithmetic the kernels run)
// against csrc/field.h + csrc/ec.h (the saturated arithmetic, itself checked against the oracle
// by tests/test_abi.py).  Built with -DH2_FU_CHECK so every fu_mul / fu_add asserts its limb bounds.
// No GPU needed: the same H2_HD source compiles for the host.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <cstring>

#include "../../halo2-pse_amd/csrc/ecu.h"
#include "../../halo2-pse_amd/csrc/glv.h"

using namespace h2;

static uint64_t rs = 0x1234567;
static uint64_t rnd() {
    rs += 0x9E3779B97F4A7C15ULL;
    uint64_t x = rs;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
static int failures = 0;
#define CHECK(c)                                                   \
    do {                                                           \
        if (!(c)) {                                                \
            if (failures < 20) printf("FAIL line %d: %s\n", __LINE__, #c); \
            failures++;                                            \
        }                                                          \
    } while (0)

template <class P>
static Fe rand_fe(int kind) {
    Fe a;
    for (int j = 0; j < 8; j++) a.l[j] = (uint32_t)rnd();
    a.l[7] &= 0x1fffffff;
    if (kind == 1) a = fe_zero<P>();
    if (kind == 2) {  // p - 1
        a = fe_zero<P>();
        a.l[0] = 1;
        a = fe_neg<P>(a);
    }
    if (kind == 3) {
        a = fe_zero<P>();
        a.l[0] = 1;
    }
    if (!fe_is_canonical<P>(a)) a.l[7] = 0;
    return a;
}

template <class U>
static void test_field(const char* name) {
    typedef typename U::Sat P;
    const Fu one_e = fu_one_e<U>(), one_i = fu_one_i<U>();
    for (int it = 0; it < 20000; it++) {
        Fe a = rand_fe<P>(it < 40 ? it % 4 : 0), b = rand_fe<P>(it < 40 ? (it / 4) % 4 : 0);
        CHECK(fe_eq(fu_canon<U>(fu_slice(a)), a));
        Fe a32 = a;
        for (int k = 0; k < 5; k++) a32 = fe_dbl<P>(a32);
        CHECK(fe_eq(fu_canon<U>(fu_mul<U>(fu_from_ext(a), one_i)), a32));  // 32a < 32p is outside fu_canon's range until reduced
        Fe want = fe_mul<P>(a, b);
        // I * I -> I, then back to E-form through the E-form one
        Fu ri = fu_mul<U>(fu_from_ext(a), fu_from_ext(b));
        CHECK(fe_eq(fu_mul_canon<U>(ri, one_e), want));
        // E * I -> E
        Fu bi = fu_mul<U>(fu_from_ext(b), one_i);
        CHECK(fe_eq(fu_mul_canon<U>(fu_slice(a), bi), want));
        CHECK(fe_eq(fu_canon<U>(fu_mul<U>(fu_slice(a), bi)), want));
        // linear ops on loose values
        Fu sa = fu_slice(a), sb = fu_slice(b);
        CHECK(fe_eq(fu_canon<U>(fu_add(sa, sb)), fe_add<P>(a, b)));
        CHECK(fe_eq(fu_canon<U>(fu_sub(sa, sb)), fe_sub<P>(a, b)));
        CHECK(fe_eq(fu_canon<U>(fu_neg(sa)), fe_neg<P>(a)));
        CHECK(fe_eq(fu_canon<U>(fu_norm(fu_sub(fu_sub(sa, sb), fu_dbl(sb)))), fe_sub<P>(fe_sub<P>(a, b), fe_dbl<P>(b))));
        // a product of two differences (signed limbs through fu_mul)
        Fu d1 = fu_sub(sa, sb), d2 = fu_sub(fu_mul<U>(sb, one_i), sa);  // d2 = b_E - a_E
        Fe w2 = fe_mul<P>(fe_sub<P>(a, b), fe_sub<P>(b, a));
        Fu d2i = fu_norm(fu_sub(fu_mul<U>(fu_from_ext(b), one_i), fu_mul<U>(fu_from_ext(a), one_i)));  // I-form (b - a)
        CHECK(fe_eq(fu_canon<U>(fu_mul<U>(d1, d2i)), w2));
        (void)d2;
        // fused forms: dedicated square, (ab - cd) with one reduction, a^2/2^261 - h
        {
            Fu ai = fu_mul<U>(fu_from_ext(a), one_i), bi2 = fu_mul<U>(fu_from_ext(b), one_i);  // I-form a, b
            Fu dab = fu_sub(ai, bi2);                                                            // signed limbs
            CHECK(fe_eq(fu_canon<U>(fu_sqr<U>(dab)), fu_canon<U>(fu_mul<U>(dab, dab))));
            CHECK(fe_eq(fu_mul_canon<U>(fu_sqr<U>(ai), one_e), fe_mul<P>(a, a)));
            Fe want2 = fe_sub<P>(fe_mul<P>(a, b), fe_mul<P>(fe_sub<P>(a, b), a));              // ab - (a-b)a
            CHECK(fe_eq(fu_mul_canon<U>(fu_mul_sub<U>(ai, bi2, dab, ai), one_e), want2));
            Fe want3 = fe_sub<P>(fe_mul<P>(fe_sub<P>(a, b), fe_sub<P>(a, b)), fe_add<P>(a, fe_dbl<P>(b)));  // (a-b)^2 - (a + 2b)
            Fu hh = fu_add(ai, fu_dbl(bi2));
            CHECK(fe_eq(fu_mul_canon<U>(fu_sqr_sub<U>(dab, hh), one_e), want3));
        }
        // zero test
        Fu z = fu_sub(sa, sa);
        CHECK(fu_is_zero_mod_p<U>(z));
        if (!fe_is_zero(a)) CHECK(!fu_is_zero_mod_p<U>(fu_norm(sa)) || false);
    }
    // fu_inv (Fermat on the unsaturated multiplier) against fe_inv, E-form in and out as ec_normalize uses it
    for (int it = 0; it < 300; it++) {
        Fe a = rand_fe<P>(it < 8 ? it % 4 : 0);
        Fe got = fu_mul_canon<U>(fu_inv<U>(fu_from_ext(a)), one_e);
        CHECK(fe_eq(got, fe_inv<P>(a)));
        if (!fe_is_zero(a)) CHECK(fe_eq(fe_mul<P>(got, a), fe_one<P>()));
    }
    // k*p and k*p + 1 for |k| <= 8
    Fu pf = fu_const<U>(U::P);
    for (int k = -8; k <= 8; k++) {
        Fu v = fu_zero();
        for (int j = 0; j < (k < 0 ? -k : k); j++) v = fu_norm(k < 0 ? fu_sub(v, pf) : fu_add(v, pf));
        v = fu_norm(v);
        CHECK(fu_is_zero_mod_p<U>(v));
        Fu v1 = v;
        v1.l[0] += 1;
        CHECK(!fu_is_zero_mod_p<U>(v1));
        Fu v2 = v;
        v2.l[3] += 5;
        CHECK(!fu_is_zero_mod_p<U>(v2));
    }
    // fu_canon_fast (the NTT's closing reduction) against fu_canon over its whole contract |value| < 16 p: random loose values,
    // and every k p + d for k = -15 .. 15 and small d around the quotient boundaries
    for (int it = 0; it < 200000; it++) {
        Fu v = fu_slice(rand_fe<P>(0));
        const int k = (int)(rnd() % 31) - 15;  // value in (-15 p, 16 p)
        for (int j = 0; j < (k < 0 ? -k : k); j++) v = fu_norm(k < 0 ? fu_sub(v, pf) : fu_add(v, pf));
        if (it & 1) {  // loose limbs, same value: move 2^29 between neighbours
            for (int j = 0; j < 8; j++) {
                const int32_t t = (int32_t)(rnd() % 5) - 2;
                v.l[j] += t * (1 << 29);
                v.l[j + 1] -= t;
            }
        }
        CHECK(fe_eq(fu_canon_fast<U>(v), fu_canon<U>(v)));
    }
    for (int k = -15; k <= 15; k++)
        for (int dd = -3; dd <= 3; dd++) {
            Fu v = fu_zero();
            for (int j = 0; j < (k < 0 ? -k : k); j++) v = fu_norm(k < 0 ? fu_sub(v, pf) : fu_add(v, pf));
            v.l[0] += dd;
            CHECK(fe_eq(fu_canon_fast<U>(v), fu_canon<U>(v)));
            Fu hi = v;  // just below / above the next multiple in the top limbs
            hi.l[7] += dd * 12345;
            if (k < 15 && k > -15) CHECK(fe_eq(fu_canon_fast<U>(hi), fu_canon<U>(hi)));
        }
    printf("%s field fuzz done, failures so far %d\n", name, failures);
}

static bool same_point(const XYZZu& u, const XYZZ& s) {
    Affine a = xyzz_to_affine(xyzzu_to_ext(u)), b = xyzz_to_affine(s);
    return fe_eq(a.x, b.x) && fe_eq(a.y, b.y);
}

static void test_ec() {
    // points: multiples of the generator (1, 2)
    Affine g;
    g.x = fe_from_u64<Q>(1);
    g.y = fe_from_u64<Q>(2);
    const int NP = 48;
    std::vector<Affine> pts;
    XYZZ cur = xyzz_identity();
    for (int i = 0; i < NP; i++) {
        xyzz_add_mixed(cur, g);
        XYZZ big = cur;
        for (int k = 0; k < 40 + i; k++) big = xyzz_double(big);  // spread the x coordinates
        xyzz_add(big, cur);
        pts.push_back(xyzz_to_affine(big));
        CHECK(affine_on_curve(pts.back()));
    }
    Affine ident;
    ident.x = fe_zero<Q>();
    ident.y = fe_zero<Q>();
    // random signed accumulation chains, with repeats (doubling), inverses (identity) and identity inputs
    for (int chain = 0; chain < 60; chain++) {
        XYZZ s = xyzz_identity();
        XYZZu u = xyzzu_identity();
        int len = 1 + (int)(rnd() % 300);
        int last = 0;
        bool lastneg = false;
        for (int i = 0; i < len; i++) {
            int r = (int)(rnd() % 100);
            int idx = (int)(rnd() % NP);
            bool neg = rnd() & 1;
            if (r < 5) { idx = last; neg = lastneg; }        // same point again -> doubling when acc == point
            else if (r < 10) { idx = last; neg = !lastneg; } // inverse of the previous point
            Affine p = (r >= 10 && r < 13) ? ident : pts[idx];
            Affine ps = neg ? affine_neg(p) : p;
            xyzz_add_mixed(s, ps);
            xyzzu_add_affine(u, p, neg);
            last = idx;
            lastneg = neg;
            if ((i & 15) == 0) CHECK(same_point(u, s));
        }
        CHECK(same_point(u, s));
    }
    // P + P, P - P, from identity, after identity
    for (int i = 0; i < NP; i++) {
        XYZZ s = xyzz_identity();
        XYZZu u = xyzzu_identity();
        xyzz_add_mixed(s, pts[i]); xyzzu_add_affine(u, pts[i], false);
        xyzz_add_mixed(s, pts[i]); xyzzu_add_affine(u, pts[i], false);   // doubling
        CHECK(same_point(u, s));
        xyzz_add_mixed(s, affine_neg(pts[i])); xyzzu_add_affine(u, pts[i], true);
        xyzz_add_mixed(s, affine_neg(pts[i])); xyzzu_add_affine(u, pts[i], true);  // -> identity
        CHECK(xyzz_is_identity(s) && xyzzu_is_identity(u));
        xyzz_add_mixed(s, pts[(i + 1) % NP]); xyzzu_add_affine(u, pts[(i + 1) % NP], false);
        CHECK(same_point(u, s));
    }
    // full adds: random pairs, equal accumulators (doubling), opposite accumulators, identity operands, running sums
    for (int it = 0; it < 300; it++) {
        XYZZ s1 = xyzz_identity(), s2 = xyzz_identity();
        XYZZu u1 = xyzzu_identity(), u2 = xyzzu_identity();
        int n1 = (int)(rnd() % 5), n2 = (int)(rnd() % 5);
        for (int i = 0; i < n1; i++) { int idx = (int)(rnd() % NP); xyzz_add_mixed(s1, pts[idx]); xyzzu_add_affine(u1, pts[idx], false); }
        int mode = it % 4;
        if (mode == 0) for (int i = 0; i < n2; i++) { int idx = (int)(rnd() % NP); xyzz_add_mixed(s2, pts[idx]); xyzzu_add_affine(u2, pts[idx], false); }
        if (mode == 1) { s2 = s1; u2 = u1; }  // a + a
        if (mode == 2) {                       // a + (-a) built through a different path
            s2 = s1; u2 = u1;
            s2.y = fe_neg<Q>(s2.y);
            u2.y = fu_neg(u2.y);
        }
        if (mode == 3) {                       // same point, different representation: (a + b) - b
            int idx = (int)(rnd() % NP);
            s2 = s1; u2 = u1;
            xyzz_add_mixed(s2, pts[idx]); xyzzu_add_affine(u2, pts[idx], false);
            xyzz_add_mixed(s2, affine_neg(pts[idx])); xyzzu_add_affine(u2, pts[idx], true);
        }
        xyzz_add(s1, s2);
        xyzzu_add(u1, u2);
        CHECK(same_point(u1, s1));
        XYZZ d = xyzz_double(s1);
        XYZZu du = xyzzu_double(u1);
        CHECK(same_point(du, d));
        uint32_t k = (uint32_t)(rnd() % 3000);
        CHECK(same_point(xyzzu_mul_small(u1, k), xyzz_mul_small(s1, k)));
        // round trip through the E-form
        CHECK(same_point(xyzzu_from_ext(xyzzu_to_ext(u1)), s1));
    }
    // summation by parts, as msm_reduce1 does it
    {
        std::vector<XYZZ> bs;
        std::vector<XYZZu> bu;
        for (int i = 0; i < 64; i++) {
            XYZZ s = xyzz_identity();
            XYZZu u = xyzzu_identity();
            int n = (int)(rnd() % 4);
            for (int j = 0; j < n; j++) { int idx = (int)(rnd() % NP); xyzz_add_mixed(s, pts[idx]); xyzzu_add_affine(u, pts[idx], false); }
            bs.push_back(s);
            bu.push_back(u);
        }
        XYZZ run = xyzz_identity(), acc = xyzz_identity();
        XYZZu runu = xyzzu_identity(), accu = xyzzu_identity();
        for (int i = 64; i-- > 0;) {
            xyzz_add(run, bs[i]); xyzz_add(acc, run);
            xyzzu_add(runu, bu[i]); xyzzu_add(accu, runu);
        }
        CHECK(same_point(accu, acc));
    }
    printf("ec fuzz done, failures so far %d\n", failures);
}

// glv.h: k == k1 + k2 * LAMBDA (mod r), |k1|, |k2| < 2^128, and [LAMBDA](x, y) == (BETA * x, y) on the curve
static Fe fe_from_limbs5(const uint32_t v[5]) {
    Fe c = fe_zero<FrP>();
    for (int i = 0; i < 5; i++) c.l[i] = v[i];
    return fe_from_canonical<FrP>(c);
}
static void test_glv() {
    Fe lam_c = fe_zero<FrP>();
    const uint32_t LAM[6] = {0xb99c90ddu, 0x8b17ea66u, 0x8d8daaa7u, 0x5bfc4108u, 0x41a91758u, 0xb3c4d79du};
    for (int i = 0; i < 6; i++) lam_c.l[i] = LAM[i];
    const Fe lam = fe_from_canonical<FrP>(lam_c);
    int max_bits = 0;
    for (int it = 0; it < 20000; it++) {
        Fe k = rand_fe<FrP>(it < 8 ? it % 4 : 0);  // raw canonical integer < r
        if (it == 8) k = lam_c;
        GlvScalar g = glv_decompose(k.l);
        CHECK(g.k1[4] < 4 && g.k2[4] < 4);  // < 2^130: the ladder reads 65 two-bit windows
        for (int b = 159; b >= 0; b--)
            if (((g.k1[b >> 5] >> (b & 31)) & 1) || ((g.k2[b >> 5] >> (b & 31)) & 1)) {
                if (b + 1 > max_bits) max_bits = b + 1;
                break;
            }
        Fe k1 = fe_from_limbs5(g.k1), k2 = fe_from_limbs5(g.k2);
        if (g.neg1) k1 = fe_neg<FrP>(k1);
        if (g.neg2) k2 = fe_neg<FrP>(k2);
        Fe back = fe_add<FrP>(k1, fe_mul<FrP>(k2, lam));
        Fe want = fe_from_canonical<FrP>(k);
        CHECK(memcmp(back.l, want.l, 32) == 0);
    }
    printf("glv: max |k_i| = %d bits\n", max_bits);
    CHECK(max_bits <= 128);
    // the endomorphism: [LAMBDA] G == (BETA * 1, 2) for the generator G = (1, 2)
    Affine G;
    G.x = fe_one<FqP>();
    G.y = fe_add<FqP>(fe_one<FqP>(), fe_one<FqP>());
    XYZZ acc = xyzz_identity();
    for (int b = 191; b >= 0; b--) {
        acc = xyzz_double(acc);
        if ((LAM[b >> 5] >> (b & 31)) & 1) xyzz_add_mixed(acc, G);
    }
    Affine got = xyzz_to_affine(acc);
    Fe beta_c = fe_zero<FqP>();
    const uint32_t BETA[6] = {0x77fffffeu, 0x57634731u, 0xacdb5c4fu, 0xd4f263f1u, 0xa0d48bacu, 0x59e26bceu};
    for (int i = 0; i < 6; i++) beta_c.l[i] = BETA[i];
    Fe beta = fe_from_canonical<FqP>(beta_c);
    Fe bx = fe_mul<FqP>(beta, G.x);
    CHECK(memcmp(got.x.l, bx.l, 32) == 0 && memcmp(got.y.l, G.y.l, 32) == 0);
    // the I-form limbs of BETA that ecfft.hip hard-codes
    const uint32_t BETA_I[9] = {0x0a337995u, 0x158d1d23u, 0x189c9b98u, 0x12fa4e45u, 0x185faadcu, 0x0176f16du, 0x0eed93bau, 0x14291140u, 0x000c0afeu};
    Fe five = beta;  // E-form; I-form = E * 2^5
    for (int i = 0; i < 5; i++) five = fe_dbl<FqP>(five);
    Fu sl = fu_slice(five);
    for (int i = 0; i < 9; i++) CHECK((uint32_t)sl.l[i] == BETA_I[i]);
}

int main() {
    test_field<FqU>("Fq");
    test_field<FrU>("Fr");
    test_ec();
    test_glv();
    printf(failures ? "FIELDU TESTS FAILED (%d)\n" : "fieldu tests ok\n", failures);
    return failures ? 1 : 0;
}
