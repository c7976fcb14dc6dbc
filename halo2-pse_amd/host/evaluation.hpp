// evaluation.hpp -- C++ host-side mirror of halo2_proofs::plonk::evaluation (plonk/evaluation.rs) over the C ABI.
//
//   plonk::Expression        plonk/circuit.rs Expression<F> (the variants evaluate_h can meet: no Selector, :593)
//   plonk::ValueSource       evaluation.rs:37-60   (same variant order: the derived PartialOrd is (kind, a, b))
//   plonk::Calculation       evaluation.rs:108-127
//   plonk::GraphEvaluator    evaluation.rs:191-201, :526-706  add_rotation / add_constant / add_calculation / add_expression
//   plonk::Evaluator         evaluation.rs:182-189, new :221-279, evaluate_h :280-522 -> h2hip_evaluate_h_bn254
//
// The graph builder is a faithful restatement (same deduplication by linear search, same operand ordering and
// constant folding), so the flattened graph has the calculations the reference builds, in the same order.
// There is no CPU evaluation here: evaluate_h hands the flattened description to the engine.
#pragma once
#include <memory>
#include <tuple>

#include "halo2hip.hpp"

namespace halo2_proofs {
namespace plonk {

struct Expression;
using Expr = std::shared_ptr<const Expression>;

struct Expression {
    enum Kind { Constant, Fixed, Advice, Instance, Challenge, Negated, Sum, Product, Scaled };
    Kind kind;
    Fr scalar = Fr::zero();  // Constant value / Scaled factor
    uint32_t index = 0;      // column_index / challenge index
    int32_t rotation = 0;
    Expr a, b;

    static Expr constant(const Fr& v) { return mk(Constant, v, 0, 0, nullptr, nullptr); }
    static Expr fixed(uint32_t column, int32_t rot = 0) { return mk(Fixed, Fr::zero(), column, rot, nullptr, nullptr); }
    static Expr advice(uint32_t column, int32_t rot = 0) { return mk(Advice, Fr::zero(), column, rot, nullptr, nullptr); }
    static Expr instance(uint32_t column, int32_t rot = 0) { return mk(Instance, Fr::zero(), column, rot, nullptr, nullptr); }
    static Expr challenge(uint32_t i) { return mk(Challenge, Fr::zero(), i, 0, nullptr, nullptr); }
    static Expr negated(Expr e) { return mk(Negated, Fr::zero(), 0, 0, std::move(e), nullptr); }
    static Expr sum(Expr x, Expr y) { return mk(Sum, Fr::zero(), 0, 0, std::move(x), std::move(y)); }
    static Expr product(Expr x, Expr y) { return mk(Product, Fr::zero(), 0, 0, std::move(x), std::move(y)); }
    static Expr scaled(Expr e, const Fr& f) { return mk(Scaled, f, 0, 0, std::move(e), nullptr); }

  private:
    static Expr mk(Kind k, const Fr& s, uint32_t i, int32_t r, Expr x, Expr y) {
        auto e = std::make_shared<Expression>();
        e->kind = k;
        e->scalar = s;
        e->index = i;
        e->rotation = r;
        e->a = std::move(x);
        e->b = std::move(y);
        return e;
    }
};
// impl Neg / Add / Sub / Mul for Expression<F> (plonk/circuit.rs): a - b is Sum(a, Negated(b)), e * F is Scaled
inline Expr operator-(const Expr& e) { return Expression::negated(e); }
inline Expr operator+(const Expr& x, const Expr& y) { return Expression::sum(x, y); }
inline Expr operator-(const Expr& x, const Expr& y) { return Expression::sum(x, Expression::negated(y)); }
inline Expr operator*(const Expr& x, const Expr& y) { return Expression::product(x, y); }
inline Expr operator*(const Expr& x, const Fr& f) { return Expression::scaled(x, f); }

// evaluation.rs:37-60; laid out as the ABI's h2hip_value_source
struct ValueSource : h2hip_value_source {
    ValueSource() : h2hip_value_source{H2HIP_VS_CONSTANT, 0, 0} {}
    ValueSource(uint32_t k, uint32_t a_ = 0, uint32_t b_ = 0) : h2hip_value_source{k, a_, b_} {}
    static ValueSource Constant(uint32_t i) { return {H2HIP_VS_CONSTANT, i, 0}; }
    static ValueSource Intermediate(uint32_t i) { return {H2HIP_VS_INTERMEDIATE, i, 0}; }
    static ValueSource Fixed(uint32_t c, uint32_t r) { return {H2HIP_VS_FIXED, c, r}; }
    static ValueSource Advice(uint32_t c, uint32_t r) { return {H2HIP_VS_ADVICE, c, r}; }
    static ValueSource Instance(uint32_t c, uint32_t r) { return {H2HIP_VS_INSTANCE, c, r}; }
    static ValueSource Challenge(uint32_t i) { return {H2HIP_VS_CHALLENGE, i, 0}; }
    static ValueSource Beta() { return {H2HIP_VS_BETA}; }
    static ValueSource Gamma() { return {H2HIP_VS_GAMMA}; }
    static ValueSource Theta() { return {H2HIP_VS_THETA}; }
    static ValueSource Y() { return {H2HIP_VS_Y}; }
    static ValueSource PreviousValue() { return {H2HIP_VS_PREVIOUS}; }
    std::tuple<uint32_t, uint32_t, uint32_t> key() const { return {kind, a, b}; }
    bool operator==(const ValueSource& o) const { return key() == o.key(); }
    bool operator!=(const ValueSource& o) const { return key() != o.key(); }
    bool operator<=(const ValueSource& o) const { return key() <= o.key(); }
};

// evaluation.rs:108-127
struct Calculation {
    uint32_t op = H2HIP_CALC_STORE;  // H2HIP_CALC_*
    ValueSource x, y;                // Horner(start, parts, factor): x = start, y = factor
    std::vector<ValueSource> parts;
    static Calculation Add(ValueSource a, ValueSource b) { return {H2HIP_CALC_ADD, a, b, {}}; }
    static Calculation Sub(ValueSource a, ValueSource b) { return {H2HIP_CALC_SUB, a, b, {}}; }
    static Calculation Mul(ValueSource a, ValueSource b) { return {H2HIP_CALC_MUL, a, b, {}}; }
    static Calculation Square(ValueSource a) { return {H2HIP_CALC_SQUARE, a, {}, {}}; }
    static Calculation Double(ValueSource a) { return {H2HIP_CALC_DOUBLE, a, {}, {}}; }
    static Calculation Negate(ValueSource a) { return {H2HIP_CALC_NEGATE, a, {}, {}}; }
    static Calculation Horner(ValueSource start, std::vector<ValueSource> p, ValueSource factor) { return {H2HIP_CALC_HORNER, start, factor, std::move(p)}; }
    static Calculation Store(ValueSource a) { return {H2HIP_CALC_STORE, a, {}, {}}; }
    bool operator==(const Calculation& o) const { return op == o.op && x == o.x && y == o.y && parts == o.parts; }
};

struct CalculationInfo {  // :213-219
    Calculation calculation;
    uint32_t target;
};

// a GraphEvaluator flattened for the ABI; owns the arrays h2hip_graph points into
struct FlatGraph {
    std::vector<Fr> constants;
    std::vector<int32_t> rotations;
    std::vector<h2hip_calculation> calculations;
    std::vector<h2hip_value_source> parts;
    uint32_t num_intermediates = 0;
    h2hip_graph abi() const {
        h2hip_graph g;
        g.constants = constants.empty() ? nullptr : constants[0].l;
        g.n_constants = (uint32_t)constants.size();
        g.rotations = rotations.data();
        g.n_rotations = (uint32_t)rotations.size();
        g.calculations = calculations.data();
        g.n_calculations = (uint32_t)calculations.size();
        g.parts = parts.data();
        g.n_parts = (uint32_t)parts.size();
        g.num_intermediates = num_intermediates;
        return g;
    }
};

class GraphEvaluator {
  public:
    std::vector<Fr> constants{Fr::zero(), Fr::one(), Fr::from(2)};  // Default, :526-539
    std::vector<int32_t> rotations;
    std::vector<CalculationInfo> calculations;
    uint32_t num_intermediates = 0;

    uint32_t add_rotation(int32_t rotation) {  // :543-552
        for (size_t i = 0; i < rotations.size(); i++)
            if (rotations[i] == rotation) return (uint32_t)i;
        rotations.push_back(rotation);
        return (uint32_t)rotations.size() - 1;
    }
    ValueSource add_constant(const Fr& constant) {  // :555-564
        for (size_t i = 0; i < constants.size(); i++)
            if (constants[i] == constant) return ValueSource::Constant((uint32_t)i);
        constants.push_back(constant);
        return ValueSource::Constant((uint32_t)constants.size() - 1);
    }
    ValueSource add_calculation(const Calculation& calculation) {  // :570-588
        for (const auto& c : calculations)
            if (c.calculation == calculation) return ValueSource::Intermediate(c.target);
        const uint32_t target = num_intermediates;
        calculations.push_back({calculation, target});
        num_intermediates++;
        return ValueSource::Intermediate(target);
    }
    ValueSource add_expression(const Expr& expr) {  // :591-706
        const ValueSource zero = ValueSource::Constant(0), one = ValueSource::Constant(1), two = ValueSource::Constant(2);
        switch (expr->kind) {
            case Expression::Constant: return add_constant(expr->scalar);
            case Expression::Fixed: {
                const uint32_t rot_idx = add_rotation(expr->rotation);
                return add_calculation(Calculation::Store(ValueSource::Fixed(expr->index, rot_idx)));
            }
            case Expression::Advice: {
                const uint32_t rot_idx = add_rotation(expr->rotation);
                return add_calculation(Calculation::Store(ValueSource::Advice(expr->index, rot_idx)));
            }
            case Expression::Instance: {
                const uint32_t rot_idx = add_rotation(expr->rotation);
                return add_calculation(Calculation::Store(ValueSource::Instance(expr->index, rot_idx)));
            }
            case Expression::Challenge: return add_calculation(Calculation::Store(ValueSource::Challenge(expr->index)));
            case Expression::Negated: {
                if (expr->a->kind == Expression::Constant) return add_constant(Fr::zero() - expr->a->scalar);
                const ValueSource result_a = add_expression(expr->a);
                return result_a == zero ? result_a : add_calculation(Calculation::Negate(result_a));
            }
            case Expression::Sum: {
                if (expr->b->kind == Expression::Negated) {  // undo subtraction stored as a + (-b)
                    const ValueSource result_a = add_expression(expr->a);
                    const ValueSource result_b = add_expression(expr->b->a);
                    if (result_a == zero) return add_calculation(Calculation::Negate(result_b));
                    if (result_b == zero) return result_a;
                    return add_calculation(Calculation::Sub(result_a, result_b));
                }
                const ValueSource result_a = add_expression(expr->a);
                const ValueSource result_b = add_expression(expr->b);
                if (result_a == zero) return result_b;
                if (result_b == zero) return result_a;
                return result_a <= result_b ? add_calculation(Calculation::Add(result_a, result_b))
                                            : add_calculation(Calculation::Add(result_b, result_a));
            }
            case Expression::Product: {
                const ValueSource result_a = add_expression(expr->a);
                const ValueSource result_b = add_expression(expr->b);
                if (result_a == zero || result_b == zero) return zero;
                if (result_a == one) return result_b;
                if (result_b == one) return result_a;
                if (result_a == two) return add_calculation(Calculation::Double(result_b));
                if (result_b == two) return add_calculation(Calculation::Double(result_a));
                if (result_a == result_b) return add_calculation(Calculation::Square(result_a));
                return result_a <= result_b ? add_calculation(Calculation::Mul(result_a, result_b))
                                            : add_calculation(Calculation::Mul(result_b, result_a));
            }
            case Expression::Scaled: {
                if (expr->scalar == Fr::zero()) return zero;
                if (expr->scalar == Fr::one()) return add_expression(expr->a);
                const ValueSource cst = add_constant(expr->scalar);
                const ValueSource result_a = add_expression(expr->a);
                return add_calculation(Calculation::Mul(result_a, cst));
            }
        }
        throw std::logic_error("unreachable");  // Expression::Selector => unreachable!() (:593)
    }

    FlatGraph flatten() const {
        FlatGraph f;
        f.constants = constants;
        f.rotations = rotations;
        f.num_intermediates = num_intermediates;
        for (const auto& ci : calculations) {
            h2hip_calculation c;
            c.op = ci.calculation.op;
            c.target = ci.target;
            c.x = ci.calculation.x;
            c.y = ci.calculation.y;
            c.parts_offset = (uint32_t)f.parts.size();
            c.parts_count = (uint32_t)ci.calculation.parts.size();
            for (const auto& p : ci.calculation.parts) f.parts.push_back(p);
            f.calculations.push_back(c);
        }
        return f;
    }
};

// one lookup argument's expressions (plonk/lookup.rs Argument: input_expressions, table_expressions)
struct LookupArgument {
    std::vector<Expr> input_expressions, table_expressions;
};

// the per-instance inputs of evaluate_h that are not part of the Evaluator (evaluation.rs:280-305): what the reference
// reads from pk, the domain and the prover's committed structures.  Every vector of Fr is one polynomial.
struct EvaluateHInputs {
    const poly::EvaluationDomain* domain = nullptr;
    std::vector<const std::vector<Fr>*> fixed_cosets;                  // pk.fixed_cosets (extended)
    std::vector<const std::vector<Fr>*> advice_polys, instance_polys;  // coefficient form, n each
    std::vector<Fr> challenges;
    Fr y, beta, gamma, theta;
    const std::vector<Fr>*l0 = nullptr, *l_last = nullptr, *l_active_row = nullptr;  // pk.l0 / l_last / l_active_row (extended)
    // permutation (pk.vk.cs.permutation, pk.permutation, permutation::prover::Committed)
    std::vector<std::pair<uint32_t, uint32_t>> permutation_columns;    // (H2HIP_ANY_*, index) of p.columns
    std::vector<const std::vector<Fr>*> permutation_cosets;            // pk.permutation.cosets
    std::vector<const std::vector<Fr>*> permutation_product_cosets;    // sets[i].permutation_product_coset
    uint32_t cs_degree = 3;                                            // chunk_len = cs.degree() - 2 (:364)
    uint32_t blinding_factors = 5;                                     // last_rotation = -(blinding_factors + 1) (:365)
    // lookups (lookup::prover::Committed): product_poly, permuted_input_poly, permuted_table_poly per lookup
    std::vector<std::array<const std::vector<Fr>*, 3>> lookups;
};

class Evaluator {  // :182-189
  public:
    GraphEvaluator custom_gates;
    std::vector<GraphEvaluator> lookups;

    // Evaluator::new (:221-279): gate_polys = cs.gates.iter().flat_map(|gate| gate.polynomials()), lookups = cs.lookups
    static Evaluator create(const std::vector<Expr>& gate_polys, const std::vector<LookupArgument>& lookup_arguments) {
        Evaluator ev;
        std::vector<ValueSource> parts;
        for (const auto& poly : gate_polys) parts.push_back(ev.custom_gates.add_expression(poly));
        ev.custom_gates.add_calculation(Calculation::Horner(ValueSource::PreviousValue(), parts, ValueSource::Y()));
        for (const auto& lookup : lookup_arguments) {
            GraphEvaluator graph;
            auto evaluate_lc = [&](const std::vector<Expr>& expressions) {
                std::vector<ValueSource> p;
                for (const auto& e : expressions) p.push_back(graph.add_expression(e));
                return graph.add_calculation(Calculation::Horner(ValueSource::Constant(0), p, ValueSource::Theta()));
            };
            const ValueSource compressed_input_coset = evaluate_lc(lookup.input_expressions);
            const ValueSource compressed_table_coset = evaluate_lc(lookup.table_expressions);
            const ValueSource right_gamma = graph.add_calculation(Calculation::Add(compressed_table_coset, ValueSource::Gamma()));
            const ValueSource lc = graph.add_calculation(Calculation::Add(compressed_input_coset, ValueSource::Beta()));
            graph.add_calculation(Calculation::Mul(lc, right_gamma));
            ev.lookups.push_back(std::move(graph));
        }
        return ev;
    }

    // evaluate_h for one circuit instance (:280-522): `values` (extended length) is folded in place
    void evaluate_h(const EvaluateHInputs& in, std::vector<Fr>& values) const {
        const poly::EvaluationDomain& d = *in.domain;
        const size_t n = (size_t)1 << d.k, size = d.extended_len();
        auto need = [&](const std::vector<Fr>* v, size_t len) {
            if (!v || v->size() != len) throw std::logic_error("evaluate_h: polynomial of the wrong length");
            return v->data()->l;
        };
        if (values.size() != size || in.lookups.size() != lookups.size() || in.permutation_cosets.size() != in.permutation_columns.size())
            throw std::logic_error("evaluate_h: inconsistent inputs");
        auto table = [&](const std::vector<const std::vector<Fr>*>& cols, size_t len) {
            std::vector<const uint64_t*> t;
            for (auto* c : cols) t.push_back(need(c, len));
            return t;
        };
        const auto fixed = table(in.fixed_cosets, size), advice = table(in.advice_polys, n), instance = table(in.instance_polys, n);
        const auto pcosets = table(in.permutation_cosets, size), pprod = table(in.permutation_product_cosets, size);
        std::vector<uint32_t> pkind, pindex;
        for (auto& c : in.permutation_columns) {
            pkind.push_back(c.first);
            pindex.push_back(c.second);
        }
        const FlatGraph cg = custom_gates.flatten();
        std::vector<FlatGraph> lg;
        std::vector<h2hip_graph> lg_abi;
        std::vector<const uint64_t*> lprod, lpin, lptab;
        for (size_t i = 0; i < lookups.size(); i++) {
            lg.push_back(lookups[i].flatten());
            lprod.push_back(need(in.lookups[i][0], n));
            lpin.push_back(need(in.lookups[i][1], n));
            lptab.push_back(need(in.lookups[i][2], n));
        }
        for (auto& g : lg) lg_abi.push_back(g.abi());
        const Fr zeta = Fr::zeta(), delta = Fr::delta();
        h2hip_evalh_desc desc;
        std::memset(&desc, 0, sizeof(desc));
        desc.k = d.k;
        desc.extended_k = d.extended_k;
        desc.extended_omega = d.extended_omega.l;
        desc.g_coset = d.g_coset.l;
        desc.g_coset_inv = d.g_coset_inv.l;
        desc.n_fixed = (uint32_t)fixed.size();
        desc.n_advice = (uint32_t)advice.size();
        desc.n_instance = (uint32_t)instance.size();
        desc.n_challenges = (uint32_t)in.challenges.size();
        desc.fixed_cosets = fixed.data();
        desc.advice_polys = advice.data();
        desc.instance_polys = instance.data();
        desc.challenges = in.challenges.empty() ? nullptr : in.challenges[0].l;
        desc.y = in.y.l;
        desc.beta = in.beta.l;
        desc.gamma = in.gamma.l;
        desc.theta = in.theta.l;
        desc.l0 = need(in.l0, size);
        desc.l_last = need(in.l_last, size);
        desc.l_active_row = need(in.l_active_row, size);
        desc.custom_gates = cg.abi();
        desc.n_perm_sets = (uint32_t)pprod.size();
        desc.n_perm_columns = (uint32_t)pcosets.size();
        desc.chunk_len = in.cs_degree - 2;
        desc.last_rotation = -(int32_t)(in.blinding_factors + 1);
        desc.perm_product_cosets = pprod.data();
        desc.perm_column_kind = pkind.data();
        desc.perm_column_index = pindex.data();
        desc.perm_cosets = pcosets.data();
        desc.zeta = zeta.l;
        desc.delta = delta.l;
        desc.n_lookups = (uint32_t)lookups.size();
        desc.lookup_graphs = lg_abi.data();
        desc.lookup_product_polys = lprod.data();
        desc.lookup_permuted_input_polys = lpin.data();
        desc.lookup_permuted_table_polys = lptab.data();
        {   // the proving key's constant columns stay in HBM across calls (as halo2hip-sys' try_evaluate_h does; `release_key_columns` below
            // is the counterpart of patch 0006's Drop); a failure only means they are uploaded per call
            std::vector<const uint64_t*> key = fixed;
            key.insert(key.end(), pcosets.begin(), pcosets.end());
            key.push_back(desc.l0);
            key.push_back(desc.l_last);
            key.push_back(desc.l_active_row);
            (void)h2hip_columns_pin(key.data(), key.size(), size);
        }
        engine_check(h2hip_evaluate_h_bn254(&desc, values[0].l), "h2hip_evaluate_h_bn254");
    }

    // ProvingKey's Drop (patch 0006): release the device copies of the key's constant columns
    static void release_key_columns(const std::vector<const std::vector<Fr>*>& columns) {
        std::vector<const uint64_t*> ptrs;
        for (auto* c : columns)
            if (c && !c->empty()) ptrs.push_back((*c)[0].l);
        if (!ptrs.empty()) (void)h2hip_columns_unpin(ptrs.data(), ptrs.size());
    }
};

}  // namespace plonk
}  // namespace halo2_proofs
