// C++ tests of halo2-pse_amd/host/halo2hip.hpp, written after the reference's own tests for this
// path (halo2_proofs/src/poly/kzg/commitment.rs:361-384 test_commit_lagrange; poly/domain.rs
// round trips).  Needs an MI355X.  usage: test_host_mirror <tests/golden dir>
#include <cstdio>
#include <fstream>
#include <string>

#include "../../halo2-pse_amd/host/halo2hip.hpp"

using namespace halo2_proofs;
using namespace halo2_proofs::poly;
using halo2_proofs::poly::kzg::ParamsKZG;

static int failures = 0;
#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);    \
            failures++;                                                    \
        }                                                                  \
    } while (0)

template <class F>
static bool panics(F f) {
    try {
        f();
    } catch (const std::logic_error&) {
        return true;
    }
    return false;
}

// fn test_commit_lagrange()  (poly/kzg/commitment.rs:361-384), SRS read from the RawBytes fixture
static void test_commit_lagrange(const std::string& dir) {
    const uint32_t K = 6;
    ParamsKZG params;
    std::ifstream f(dir + "/kzg_6_params.rawbytes", std::ios::binary);
    ParamsKZG::read(f, params);
    CHECK(params.k == K && params.g.size() == 64 && params.g_lagrange.size() == 64);
    EvaluationDomain domain(1, K);

    auto a = domain.empty_lagrange();
    for (size_t i = 0; i < a.len(); i++) a[i] = Fr::from(i);

    auto b = domain.lagrange_to_coeff(a);
    Blind alpha{Fr::from(12345)};
    CHECK(params.commit(b, alpha) == params.commit_lagrange(a, alpha));
    // the blind is ignored by KZG commit (poly/kzg/commitment.rs:284,327)
    CHECK(params.commit(b, Blind{Fr::zero()}).to_affine() == params.commit(b, alpha).to_affine());
}

// iNTT then NTT returns the input; coset round trip (poly/domain.rs:240-303)
static void test_domain_roundtrips() {
    const uint32_t k = 12;
    EvaluationDomain domain(4, k);
    CHECK(domain.extended_k == 14);
    CHECK(domain.omega * domain.omega_inv == Fr::one());
    CHECK(domain.omega.pow_vartime(uint64_t(1) << (k - 1)) == Fr::zero() - Fr::one());
    Polynomial<LagrangeCoeff> a = domain.empty_lagrange();
    Fr x = Fr::from(7);
    for (size_t i = 0; i < a.len(); i++) {
        a[i] = x;
        x = x * x + Fr::from(i);
    }
    auto coeffs = domain.lagrange_to_coeff(a);
    std::vector<Fr> back = coeffs.values;
    arithmetic::best_fft(back, domain.omega, k);
    CHECK(back == a.values);
    auto ext = domain.coeff_to_extended(coeffs);
    CHECK(ext.len() == domain.extended_len());
    auto back2 = domain.extended_to_coeff(ext);
    CHECK(back2.size() == size_t(domain.n * domain.quotient_poly_degree));
    bool same = true;
    for (size_t i = 0; i < back2.size(); i++) same = same && back2[i] == (i < coeffs.len() ? coeffs[i] : Fr::zero());
    CHECK(same);
}

// the reference panics on these (arithmetic.rs:133,184; poly/domain.rs:227; kzg/commitment.rs:290)
static void test_contract_violations() {
    std::vector<Fr> c(3, Fr::one());
    std::vector<G1Affine> bases(2);
    CHECK(panics([&] { arithmetic::best_multiexp(c, bases); }));
    std::vector<Fr> a(5, Fr::one());
    CHECK(panics([&] { arithmetic::best_fft(a, Fr::one(), 2); }));
    EvaluationDomain domain(2, 4);
    Polynomial<LagrangeCoeff> p{std::vector<Fr>(15)};
    CHECK(panics([&] { domain.lagrange_to_coeff(p); }));
}

int main(int argc, char** argv) {
    std::string dir = argc > 1 ? argv[1] : "tests/golden";
    if (h2hip_init(nullptr, 0) != 0) {
        std::printf("h2hip_init failed: %s\n", h2hip_last_error());
        return 2;
    }
    test_commit_lagrange(dir);
    test_domain_roundtrips();
    test_contract_violations();
    h2hip_shutdown();
    std::printf(failures ? "HOST MIRROR TESTS FAILED (%d)\n" : "host mirror tests ok\n", failures);
    return failures ? 1 : 0;
}
