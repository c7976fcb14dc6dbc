// C++ tests of halo2-pse_amd/host/halo2hip.hpp, written after the reference's own tests for this
// path (halo2_proofs/src/poly/kzg/commitment.rs:361-384 test_commit_lagrange; poly/domain.rs
// round trips).  Needs an MI355X.  usage: test_host_mirror <tests/golden dir>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>

#include "../../halo2-pse_amd/host/evaluation.hpp"
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
    // the fixture's g_lagrange is g_to_lagrange(g) (arithmetic.rs:277-301), and downsize (:267-275) keeps the identity
    CHECK(arithmetic::g_to_lagrange(params.g, K) == params.g_lagrange);
    {
        ParamsKZG small;
        std::ifstream f2(dir + "/kzg_6_params.rawbytes", std::ios::binary);
        ParamsKZG::read(f2, small);
        small.downsize(4);
        CHECK(small.k == 4 && small.g.size() == 16 && small.g_lagrange.size() == 16);
        EvaluationDomain d4(1, 4);
        auto a4 = d4.empty_lagrange();
        for (size_t i = 0; i < a4.len(); i++) a4[i] = Fr::from(3 * i + 1);
        CHECK(small.commit(d4.lagrange_to_coeff(a4), alpha) == small.commit_lagrange(a4, alpha));
        CHECK(panics([&] { small.downsize(5); }));
    }
    // the batched column commit equals one commit_lagrange per column
    {
        std::vector<Polynomial<LagrangeCoeff>> cols(5, domain.empty_lagrange());
        std::vector<const Polynomial<LagrangeCoeff>*> ptrs;
        for (size_t c = 0; c < cols.size(); c++) {
            for (size_t i = 0; i < cols[c].len(); i++) cols[c][i] = Fr::from(1000 * c + i * i + 7);
            ptrs.push_back(&cols[c]);
        }
        std::vector<G1> many = params.commit_lagrange_many(ptrs);
        CHECK(many.size() == cols.size());
        for (size_t c = 0; c < cols.size(); c++) CHECK(many[c] == params.commit_lagrange(cols[c], alpha));
    }
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
    // the batched forms (patch 0004) give, column by column, what the one-column methods give
    std::vector<Polynomial<LagrangeCoeff>> cols(5, a);
    for (size_t j = 0; j < cols.size(); j++) cols[j][j] = cols[j][j] + Fr::from(j + 1);
    auto batch = domain.lagrange_to_coeff_batch(cols);
    CHECK(batch.size() == cols.size());
    for (size_t j = 0; j < cols.size(); j++) CHECK(batch[j].values == domain.lagrange_to_coeff(cols[j]).values);
    auto ext_batch = domain.coeff_to_extended_batch(batch);
    for (size_t j = 0; j < cols.size(); j++) CHECK(ext_batch[j].values == domain.coeff_to_extended(batch[j]).values);
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

// ---- plonk::Evaluator (halo2-pse_amd/host/evaluation.hpp)
using namespace halo2_proofs::plonk;

// the constraint system of examples/circuit-layout.rs MyCircuit (:174-240) as tests/golden/make_evalh_golden.py states it
static void circuit_layout_system(std::vector<Expr>* gates, std::vector<LookupArgument>* lookups) {
    const uint32_t e_ = 0, a_ = 1, b_ = 2, c_ = 3, d_ = 4, sf = 0, sm = 1, sa = 2, sb = 3, sc = 4, sl = 5;
    auto A = [](uint32_t c, int32_t r = 0) { return Expression::advice(c, r); };
    auto F = [](uint32_t c, int32_t r = 0) { return Expression::fixed(c, r); };
    Expr gate = A(a_) * F(sa) + A(b_) * F(sb) + A(a_) * A(b_) * F(sm) - (A(c_) * F(sc)) + F(sf) * (A(d_, 1) * A(e_, -1));
    Expr gate2 = (A(c_) * A(c_)) * Fr::from(7) + (Expression::constant(Fr::from(2)) * A(d_) + (-Expression::constant(Fr::from(5))));
    *gates = {gate, gate2};
    *lookups = {LookupArgument{{A(a_)}, {F(sl)}}};
}

static void dump_graph(std::FILE* f, const char* name, const GraphEvaluator& g) {
    FlatGraph fg = g.flatten();
    std::fprintf(f, "%s num_intermediates %u\n", name, fg.num_intermediates);
    for (auto& c : fg.constants) std::fprintf(f, "%s constant %llu %llu %llu %llu\n", name, (unsigned long long)c.l[0], (unsigned long long)c.l[1],
                                              (unsigned long long)c.l[2], (unsigned long long)c.l[3]);
    for (auto r : fg.rotations) std::fprintf(f, "%s rotation %d\n", name, r);
    for (auto& c : fg.calculations)
        std::fprintf(f, "%s calc %u %u %u %u %u %u %u %u %u %u\n", name, c.op, c.target, c.x.kind, c.x.a, c.x.b, c.y.kind, c.y.a, c.y.b, c.parts_offset,
                     c.parts_count);
    for (auto& p : fg.parts) std::fprintf(f, "%s part %u %u %u\n", name, p.kind, p.a, p.b);
}

static int dump_graphs(const char* path) {
    std::vector<Expr> gates;
    std::vector<LookupArgument> lookups;
    circuit_layout_system(&gates, &lookups);
    Evaluator ev = Evaluator::create(gates, lookups);
    std::FILE* f = std::fopen(path, "w");
    if (!f) return 2;
    dump_graph(f, "custom", ev.custom_gates);
    dump_graph(f, "lookup0", ev.lookups[0]);
    std::fclose(f);
    return 0;
}

// Expression::evaluate (plonk/circuit.rs) at one row of the extended domain, for the check below
static Fr eval_expr(const Expr& e, const std::vector<std::vector<Fr>>& fixed, const std::vector<std::vector<Fr>>& advice, size_t idx, int rot_scale) {
    const size_t size = fixed[0].size();
    auto at = [&](const std::vector<Fr>& col, int32_t rot) { return col[(idx + size + (long)rot * rot_scale) % size]; };
    switch (e->kind) {
        case Expression::Constant: return e->scalar;
        case Expression::Fixed: return at(fixed[e->index], e->rotation);
        case Expression::Advice: return at(advice[e->index], e->rotation);
        case Expression::Negated: return Fr::zero() - eval_expr(e->a, fixed, advice, idx, rot_scale);
        case Expression::Sum: return eval_expr(e->a, fixed, advice, idx, rot_scale) + eval_expr(e->b, fixed, advice, idx, rot_scale);
        case Expression::Product: return eval_expr(e->a, fixed, advice, idx, rot_scale) * eval_expr(e->b, fixed, advice, idx, rot_scale);
        case Expression::Scaled: return eval_expr(e->a, fixed, advice, idx, rot_scale) * e->scalar;
        default: throw std::logic_error("not used here");
    }
}

// custom gates of the circuit-layout system through Evaluator::evaluate_h == the Expression trees evaluated row by row
// on the cosets (no permutation, no lookups here: tests/test_evalh.py covers those against the oracle)
static void test_evaluate_h_custom_gates() {
    const uint32_t k = 5;
    EvaluationDomain domain(4, k);
    const size_t n = size_t(1) << k, size = domain.extended_len();
    std::vector<Expr> gates;
    std::vector<LookupArgument> lookups;
    circuit_layout_system(&gates, &lookups);
    Evaluator ev = Evaluator::create(gates, {});
    Fr x = Fr::from(3);
    auto next = [&] {
        x = x * x + Fr::from(11);
        return x;
    };
    std::vector<std::vector<Fr>> fixed(6, std::vector<Fr>(size)), advice_polys(5, std::vector<Fr>(n)), advice_cosets;
    for (auto& c : fixed)
        for (auto& v : c) v = next();
    for (auto& c : advice_polys)
        for (auto& v : c) v = next();
    for (auto& c : advice_polys) advice_cosets.push_back(domain.coeff_to_extended(Polynomial<Coeff>{c}).values);
    std::vector<Fr> l0(size), l_last(size), l_active(size), values(size);
    for (auto& v : values) v = next();
    EvaluateHInputs in;
    in.domain = &domain;
    for (auto& c : fixed) in.fixed_cosets.push_back(&c);
    for (auto& c : advice_polys) in.advice_polys.push_back(&c);
    in.y = next();
    in.beta = next();
    in.gamma = next();
    in.theta = next();
    in.l0 = &l0;
    in.l_last = &l_last;
    in.l_active_row = &l_active;
    std::vector<Fr> want = values;
    const int rot_scale = 1 << (domain.extended_k - k);
    for (size_t idx = 0; idx < size; idx++)
        for (auto& g : gates) want[idx] = want[idx] * in.y + eval_expr(g, fixed, advice_cosets, idx, rot_scale);
    ev.evaluate_h(in, values);
    CHECK(values == want);
    // a polynomial of the wrong length is a contract violation
    std::vector<Fr> short_values(size / 2);
    CHECK(panics([&] { ev.evaluate_h(in, short_values); }));
}

// EvaluationDomain::new(j, k) as hex, one constant per line (no GPU needed): compared with tests/golden by the CPU suite
static int dump_domain(uint32_t j, uint32_t k) {
    poly::EvaluationDomain d(j, k);
    auto hex = [](const char* name, const Fr& f) {
        std::printf("%s %016llx%016llx%016llx%016llx\n", name, (unsigned long long)f.l[3], (unsigned long long)f.l[2], (unsigned long long)f.l[1],
                    (unsigned long long)f.l[0]);
    };
    std::printf("extended_k %u\n", d.extended_k);
    hex("omega", d.omega);
    hex("omega_inv", d.omega_inv);
    hex("extended_omega", d.extended_omega);
    hex("extended_omega_inv", d.extended_omega_inv);
    hex("g_coset", d.g_coset);
    hex("g_coset_inv", d.g_coset_inv);
    hex("ifft_divisor", d.ifft_divisor);
    hex("extended_ifft_divisor", d.extended_ifft_divisor);
    hex("barycentric_weight", d.barycentric_weight);
    for (const Fr& t : d.t_evaluations) hex("t_evaluations", t);
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 2 && std::string(argv[1]) == "--dump-graphs") return dump_graphs(argv[2]);
    if (argc > 3 && std::string(argv[1]) == "--dump-domain") return dump_domain((uint32_t)std::atoi(argv[2]), (uint32_t)std::atoi(argv[3]));
    std::string dir = argc > 1 ? argv[1] : "tests/golden";
    if (h2hip_init(nullptr, 0) != 0) {
        std::printf("h2hip_init failed: %s\n", h2hip_last_error());
        return 2;
    }
    test_commit_lagrange(dir);
    test_domain_roundtrips();
    test_contract_violations();
    test_evaluate_h_custom_gates();
    h2hip_shutdown();
    std::printf(failures ? "HOST MIRROR TESTS FAILED (%d)\n" : "host mirror tests ok\n", failures);
    return failures ? 1 : 0;
}
