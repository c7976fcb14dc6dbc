// halo2hip.hpp -- C++ host-side mirror of the reference's Rust interface for the accelerated
// path, layered on the C ABI (include/halo2hip.h).  The reference is Rust and no Rust toolchain
// exists in the build image, so this header plays the role the patched `halo2_proofs` modules
// would: same names, same argument meaning, same contract checks.
//
//   halo2_proofs::arithmetic::best_multiexp      halo2_proofs/src/arithmetic.rs:132-159
//   halo2_proofs::arithmetic::best_fft           halo2_proofs/src/arithmetic.rs:171-234
//   halo2_proofs::poly::EvaluationDomain         halo2_proofs/src/poly/domain.rs:18-361
//   halo2_proofs::poly::kzg::ParamsKZG           halo2_proofs/src/poly/kzg/commitment.rs:22-339
//   halo2_proofs::plonk::{GraphEvaluator, Evaluator}   halo2_proofs/src/plonk/evaluation.rs   (in evaluation.hpp)
//
// Error behaviour: where the reference panics on a contract violation (assert_eq! / assert!),
// this mirror throws std::logic_error; a non-zero engine status throws std::runtime_error
// (the Rust shim would fall back to the CPU body instead -- there is none in this library).
// Header-only; plain C++17; link with -lhalo2hip.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <istream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/halo2hip.h"
#include "../csrc/ec.h"  // host-side Fr / Fq / G1 arithmetic for domain constants and SRS checks (no HIP needed)

namespace halo2_proofs {

// bn256::Fr, G1Affine, G1 in halo2curves' in-memory layout (4 x u64 LE limbs, Montgomery)
struct Fr {
    uint64_t l[4];
    bool operator==(const Fr& o) const { return std::memcmp(l, o.l, 32) == 0; }
    bool operator!=(const Fr& o) const { return !(*this == o); }
    static Fr zero() { return Fr{{0, 0, 0, 0}}; }
    static Fr one() { return from_fe(h2::fe_one<h2::FrP>()); }
    static Fr from(uint64_t v) { return from_fe(h2::fe_from_u64<h2::FrP>(v)); }  // Fr::from(u64)
    Fr operator*(const Fr& o) const { return from_fe(h2::fe_mul<h2::FrP>(fe(), o.fe())); }
    Fr operator+(const Fr& o) const { return from_fe(h2::fe_add<h2::FrP>(fe(), o.fe())); }
    Fr operator-(const Fr& o) const { return from_fe(h2::fe_sub<h2::FrP>(fe(), o.fe())); }
    Fr square() const { return *this * *this; }
    Fr invert() const { return from_fe(h2::fe_inv<h2::FrP>(fe())); }
    Fr pow_vartime(uint64_t e) const { return from_fe(h2::fe_pow_u64<h2::FrP>(fe(), e)); }
    static Fr root_of_unity() { return from_limbs32(h2::FrP::ROOT_OF_UNITY); }
    static Fr zeta() { return from_limbs32(h2::FrP::ZETA); }
    static Fr delta() { return from(7).pow_vartime(uint64_t(1) << S_); }  // Fr::DELTA = MULTIPLICATIVE_GENERATOR^(2^S)
    static constexpr uint32_t S = 28;
    static constexpr uint32_t S_ = 28;
    h2::Fe fe() const {
        h2::Fe f;
        std::memcpy(f.l, l, 32);
        return f;
    }
    static Fr from_fe(const h2::Fe& f) {
        Fr r;
        std::memcpy(r.l, f.l, 32);
        return r;
    }
    static Fr from_limbs32(const uint32_t v[8]) {
        Fr r;
        std::memcpy(r.l, v, 32);
        return r;
    }
};
static_assert(sizeof(Fr) == 32, "Fr layout");

struct G1Affine {
    uint64_t x[4], y[4];
    bool operator==(const G1Affine& o) const { return std::memcmp(this, &o, 64) == 0; }
};
static_assert(sizeof(G1Affine) == 64, "G1Affine layout");

struct G1 {
    uint64_t x[4], y[4], z[4];
    G1Affine to_affine() const {  // Curve::to_affine
        G1Affine a;
        if (h2hip_g1_to_affine(x, a.x) != 0) throw std::runtime_error(h2hip_last_error());
        return a;
    }
    // projective equality, as halo2curves' PartialEq for G1
    bool operator==(const G1& o) const { return to_affine() == o.to_affine(); }
};
static_assert(sizeof(G1) == 96, "G1 layout");

inline void engine_check(int rc, const char* what) {
    if (rc != 0) throw std::runtime_error(std::string(what) + ": " + h2hip_last_error());
}

namespace arithmetic {

// pub fn best_multiexp<C: CurveAffine>(coeffs: &[C::Scalar], bases: &[C]) -> C::Curve   (arithmetic.rs:132)
inline G1 best_multiexp(const std::vector<Fr>& coeffs, const G1Affine* bases, size_t bases_len) {
    if (coeffs.size() != bases_len) throw std::logic_error("assertion failed: coeffs.len() == bases.len()");  // :133
    G1 out;
    engine_check(h2hip_msm_bn254(coeffs.empty() ? nullptr : coeffs[0].l, bases_len ? bases[0].x : nullptr, coeffs.size(), out.x),
                 "best_multiexp");
    return out;
}
inline G1 best_multiexp(const std::vector<Fr>& coeffs, const std::vector<G1Affine>& bases) {
    return best_multiexp(coeffs, bases.data(), bases.size());
}

// pub fn best_fft<G: Group>(a: &mut [G], omega: G::Scalar, log_n: u32)                  (arithmetic.rs:171)
inline void best_fft(std::vector<Fr>& a, const Fr& omega, uint32_t log_n) {
    if (log_n > 63 || a.size() != (size_t(1) << log_n)) throw std::logic_error("assertion failed: n == 1 << log_n");  // :184
    engine_check(h2hip_ntt_bn254_fr(a[0].l, omega.l, log_n), "best_fft");
}

// best_fft with G = G1 (arithmetic.rs:171-234; in the crate: g_to_lagrange, :285): in place on Jacobian points, the caller's omega.
// Only the group elements are defined by the reference; they come back with z = 1 (identity: z = 0).
inline void best_fft(std::vector<G1>& a, const Fr& omega, uint32_t log_n) {
    if (log_n > 63 || a.size() != (size_t(1) << log_n)) throw std::logic_error("assertion failed: n == 1 << log_n");  // :184
    engine_check(h2hip_fft_bn254_g1(a[0].x, omega.l, log_n), "best_fft::<G1>");
}

// g_to_lagrange (arithmetic.rs:277-301); takes the affine points (the reference converts them with to_curve() at the call site)
inline std::vector<G1Affine> g_to_lagrange(const std::vector<G1Affine>& g, uint32_t k) {
    if (g.size() != (size_t(1) << k)) throw std::logic_error("assertion failed: a.len() == 1 << log_n");  // best_fft, :184
    std::vector<G1Affine> out(g.size());
    engine_check(h2hip_g_to_lagrange_bn254(g[0].x, k, out[0].x), "h2hip_g_to_lagrange_bn254");
    return out;
}
}  // namespace arithmetic

namespace poly {

struct Coeff {};
struct LagrangeCoeff {};
struct ExtendedLagrangeCoeff {};

// Polynomial<F, B> (poly.rs:66-72): a Vec<F> plus a basis marker
template <class Basis>
struct Polynomial {
    std::vector<Fr> values;
    size_t len() const { return values.size(); }
    Fr& operator[](size_t i) { return values[i]; }
    const Fr& operator[](size_t i) const { return values[i]; }
};

// EvaluationDomain<Fr> (poly/domain.rs:18-34)
class EvaluationDomain {
   public:
    uint64_t n;
    uint32_t k, extended_k;
    Fr omega, omega_inv, extended_omega, extended_omega_inv, g_coset, g_coset_inv;
    uint64_t quotient_poly_degree;
    Fr ifft_divisor, extended_ifft_divisor;
    std::vector<Fr> t_evaluations;
    Fr barycentric_weight;

    // EvaluationDomain::new (poly/domain.rs:39-142)
    EvaluationDomain(uint32_t j, uint32_t k_) {
        quotient_poly_degree = uint64_t(j - 1);                     // :41
        k = k_;
        n = uint64_t(1) << k;                                       // :44
        extended_k = k;                                             // :49-52
        while ((uint64_t(1) << extended_k) < n * quotient_poly_degree) extended_k++;
        if (extended_k > Fr::S) throw std::logic_error("extended_k exceeds the 2-adicity of Fr");
        extended_omega = Fr::root_of_unity();                       // :54-61
        for (uint32_t i = extended_k; i < Fr::S; i++) extended_omega = extended_omega.square();
        omega = extended_omega;                                     // :70-73
        for (uint32_t i = k; i < extended_k; i++) omega = omega.square();
        g_coset = Fr::zeta();                                       // :81
        g_coset_inv = g_coset.square();                             // :82
        {                                                           // :84-107
            Fr orig = Fr::zeta().pow_vartime(n), step = extended_omega.pow_vartime(n), cur = orig;
            do {
                t_evaluations.push_back(cur);
                cur = cur * step;
            } while (cur != orig);
            if (t_evaluations.size() != (size_t(1) << (extended_k - k))) throw std::logic_error("t_evaluations length");  // :98
            for (auto& c : t_evaluations) c = (c - Fr::one()).invert();  // :101-103, :117-124
        }
        ifft_divisor = Fr::from(uint64_t(1) << k).invert();               // :109
        extended_ifft_divisor = Fr::from(uint64_t(1) << extended_k).invert();  // :110
        barycentric_weight = Fr::from(n).invert();                        // :114
        extended_omega_inv = extended_omega.invert();
        omega_inv = omega.invert();
    }

    size_t extended_len() const { return size_t(1) << extended_k; }       // :374-376

    Polynomial<LagrangeCoeff> empty_lagrange() const { return {std::vector<Fr>(n, Fr::zero())}; }  // :177-182

    // lagrange_to_coeff (poly/domain.rs:226-236)
    Polynomial<Coeff> lagrange_to_coeff(Polynomial<LagrangeCoeff> a) const {
        if (a.values.size() != (size_t(1) << k)) throw std::logic_error("assertion failed: a.values.len() == 1 << self.k");  // :227
        engine_check(h2hip_ifft_bn254_fr(a.values[0].l, omega_inv.l, k, ifft_divisor.l), "lagrange_to_coeff");            // :230
        return {std::move(a.values)};
    }

    // the same for the columns create_proof converts back to back (plonk/prover.rs:476-490; patch 0004's lagrange_to_coeff_batch): one
    // pipelined engine call -- column i + 1 goes up and column i - 1 comes down while column i is transformed
    std::vector<Polynomial<Coeff>> lagrange_to_coeff_batch(std::vector<Polynomial<LagrangeCoeff>> polys) const {
        std::vector<uint64_t*> cols;
        for (auto& a : polys) {
            if (a.values.size() != (size_t(1) << k)) throw std::logic_error("assertion failed: a.values.len() == 1 << self.k");
            cols.push_back(a.values[0].l);
        }
        engine_check(h2hip_ifft_bn254_fr_batch(cols.data(), cols.size(), omega_inv.l, k, ifft_divisor.l), "lagrange_to_coeff_batch");
        std::vector<Polynomial<Coeff>> out;
        for (auto& a : polys) out.push_back({std::move(a.values)});
        return out;
    }

    // coeff_to_extended for several polynomials (plonk/evaluation.rs:306-323; patch 0004's coeff_to_extended_batch)
    std::vector<Polynomial<ExtendedLagrangeCoeff>> coeff_to_extended_batch(const std::vector<Polynomial<Coeff>>& polys) const {
        std::vector<Polynomial<ExtendedLagrangeCoeff>> out(polys.size());
        std::vector<const uint64_t*> ins;
        std::vector<uint64_t*> outs;
        for (size_t i = 0; i < polys.size(); i++) {
            if (polys[i].values.size() != (size_t(1) << k)) throw std::logic_error("assertion failed: a.values.len() == 1 << self.k");
            out[i].values.resize(extended_len());
            ins.push_back(polys[i].values[0].l);
            outs.push_back(out[i].values[0].l);
        }
        engine_check(h2hip_coeff_to_extended_bn254_fr_batch(ins.data(), k, outs.data(), ins.size(), extended_k, extended_omega.l, g_coset.l, g_coset_inv.l),
                     "coeff_to_extended_batch");
        return out;
    }

    // coeff_to_extended (poly/domain.rs:240-254)
    Polynomial<ExtendedLagrangeCoeff> coeff_to_extended(const Polynomial<Coeff>& a) const {
        if (a.values.size() != (size_t(1) << k)) throw std::logic_error("assertion failed: a.values.len() == 1 << self.k");  // :244
        Polynomial<ExtendedLagrangeCoeff> out{std::vector<Fr>(extended_len())};
        engine_check(h2hip_coeff_to_extended_bn254_fr(a.values[0].l, k, out.values[0].l, extended_k, extended_omega.l, g_coset.l, g_coset_inv.l),
                     "coeff_to_extended");
        return out;
    }

    // extended_to_coeff (poly/domain.rs:281-303)
    std::vector<Fr> extended_to_coeff(Polynomial<ExtendedLagrangeCoeff> a) const {
        if (a.values.size() != extended_len()) throw std::logic_error("assertion failed: a.values.len() == self.extended_len()");  // :282
        engine_check(h2hip_extended_to_coeff_bn254_fr(a.values[0].l, extended_k, extended_omega_inv.l, extended_ifft_divisor.l, g_coset.l,
                                                      g_coset_inv.l),
                     "extended_to_coeff");
        a.values.resize(size_t(n * quotient_poly_degree));                                                                   // :299-300
        return std::move(a.values);
    }

    // divide_by_vanishing_poly (poly/domain.rs:307-326)
    Polynomial<ExtendedLagrangeCoeff> divide_by_vanishing_poly(Polynomial<ExtendedLagrangeCoeff> a) const {
        if (a.values.size() != extended_len()) throw std::logic_error("assertion failed: a.values.len() == self.extended_len()");  // :311
        engine_check(h2hip_divide_by_vanishing_poly_bn254_fr(a.values[0].l, extended_k, t_evaluations[0].l, uint32_t(t_evaluations.size())),
                     "divide_by_vanishing_poly");
        return a;
    }
};

struct Blind {
    Fr r;
};

namespace kzg {

// ParamsKZG<Bn256> (poly/kzg/commitment.rs:22-30): g / g_lagrange are pinned on the GPU for the
// life of the object (h2hip_bases_pin), the hook INTEGRATION.md section 3 describes.
class ParamsKZG {
   public:
    uint32_t k = 0;
    uint64_t n = 0;
    std::vector<G1Affine> g, g_lagrange;
    std::array<uint8_t, 128> g2{}, s_g2{};  // carried opaquely (pairing is not on this path)

    ParamsKZG() = default;
    ParamsKZG(const ParamsKZG&) = delete;
    ParamsKZG& operator=(const ParamsKZG&) = delete;
    ~ParamsKZG() { unpin(); }

    // setup (poly/kzg/commitment.rs:61-129) with the secret supplied: the reference draws `s` from its rng argument
    // (:72, "MUST NOT be used in production"); everything after that line is what runs here, on the GPU.
    // g2 / s_g2 (:118-119) are the verifier's half and stay zero.
    static void setup(uint32_t k, const Fr& s, ParamsKZG& p) {
        if (k > Fr::S) throw std::logic_error("assertion failed: k <= E::Scalar::S");  // :64
        p.unpin();
        p.k = k;
        p.n = uint64_t(1) << k;
        p.g.resize(p.n);
        p.g_lagrange.resize(p.n);
        engine_check(h2hip_kzg_setup_bn254(k, s.l, p.g[0].x, p.g_lagrange[0].x), "h2hip_kzg_setup_bn254");
        p.pin();
    }
    // the reference's signature: `rng` is any callable returning an Fr (<E::Scalar>::random(rng), :72)
    template <class Rng>
    static void setup(uint32_t k, Rng&& rng, ParamsKZG& p) {
        const Fr s = rng();
        setup(k, s, p);
    }

    // SerdeFormat (helpers.rs:8-21); `Processed` (compressed points) is not on this path
    enum class SerdeFormat { RawBytes, RawBytesUnchecked };

    // Params::read = read_custom(reader, SerdeFormat::RawBytes) (poly/kzg/commitment.rs:160-244, :300-302):
    // k as u32 LE, then g, g_lagrange as 64-B Montgomery points, then g2, s_g2 (128 B each).  RawBytes checks that
    // every coordinate is below the modulus and every point lies on the curve (helpers.rs:15-18, read_raw);
    // RawBytesUnchecked performs no checks (:19-20).
    static void read(std::istream& reader, ParamsKZG& p) { read_custom(reader, p, SerdeFormat::RawBytes); }
    static void read_custom(std::istream& reader, ParamsKZG& p, SerdeFormat format) {
        uint8_t kb[4];
        reader.read(reinterpret_cast<char*>(kb), 4);
        if (!reader) throw std::runtime_error("ParamsKZG::read: short read");
        p.unpin();
        p.k = uint32_t(kb[0]) | uint32_t(kb[1]) << 8 | uint32_t(kb[2]) << 16 | uint32_t(kb[3]) << 24;
        if (p.k > Fr::S) throw std::runtime_error("ParamsKZG::read: k too large");
        p.n = uint64_t(1) << p.k;
        p.g.resize(p.n);
        p.g_lagrange.resize(p.n);
        reader.read(reinterpret_cast<char*>(p.g.data()), std::streamsize(p.n * 64));
        reader.read(reinterpret_cast<char*>(p.g_lagrange.data()), std::streamsize(p.n * 64));
        reader.read(reinterpret_cast<char*>(p.g2.data()), 128);
        reader.read(reinterpret_cast<char*>(p.s_g2.data()), 128);
        if (!reader) throw std::runtime_error("ParamsKZG::read: short read");
        if (format == SerdeFormat::RawBytes) {
            for (const auto* v : {&p.g, &p.g_lagrange})
                for (const G1Affine& pt : *v)
                    if (!point_is_valid(pt)) throw std::runtime_error("ParamsKZG::read: invalid point encoding");
        }
        p.pin();
    }

    // read_raw's checks for one G1Affine: limbs below the modulus, point on y^2 = x^3 + 3 (the identity is (0, 0))
    static bool point_is_valid(const G1Affine& pt) {
        h2::Affine a;
        std::memcpy(&a, &pt, 64);
        return h2::fe_is_canonical<h2::FqP>(a.x) && h2::fe_is_canonical<h2::FqP>(a.y) && h2::affine_on_curve(a);
    }

    // downsize (poly/kzg/commitment.rs:267-275)
    void downsize(uint32_t k_) {
        if (k_ > k) throw std::logic_error("assertion failed: k <= self.k");  // :268
        unpin();
        k = k_;
        n = uint64_t(1) << k;
        g.resize(n);                                                          // truncate, :273
        g_lagrange = arithmetic::g_to_lagrange(g, k);                         // :274
        pin();
    }

    // commit_lagrange (poly/kzg/commitment.rs:281-292); the blind is ignored there too
    G1 commit_lagrange(const Polynomial<LagrangeCoeff>& poly, const Blind&) const {
        if (g_lagrange.size() < poly.len()) throw std::logic_error("assertion failed: bases.len() >= size");  // :290
        return arithmetic::best_multiexp(poly.values, g_lagrange.data(), poly.len());                           // :291
    }

    // The column loop of create_proof (plonk/prover.rs:361-365: `params.commit_lagrange(poly, blind)` over all advice
    // polynomials) as one engine call; equal to calling commit_lagrange on each.
    std::vector<G1> commit_lagrange_many(const std::vector<const Polynomial<LagrangeCoeff>*>& polys) const {
        std::vector<G1> out(polys.size());
        if (polys.empty()) return out;
        const size_t size = polys[0]->len();
        if (g_lagrange.size() < size) throw std::logic_error("assertion failed: bases.len() >= size");            // :290
        std::vector<const uint64_t*> ptrs;
        for (auto* p : polys) {
            if (p->len() != size) throw std::logic_error("commit_lagrange_many: polynomials of different lengths");
            ptrs.push_back(p->values[0].l);
        }
        engine_check(h2hip_msm_bn254_batch(ptrs.data(), g_lagrange[0].x, size, ptrs.size(), out[0].x), "h2hip_msm_bn254_batch");
        return out;
    }

    // commit (poly/kzg/commitment.rs:327-334)
    G1 commit(const Polynomial<Coeff>& poly, const Blind&) const {
        if (g.size() < poly.len()) throw std::logic_error("assertion failed: bases.len() >= size");           // :332
        return arithmetic::best_multiexp(poly.values, g.data(), poly.len());                                   // :333
    }

    const std::vector<G1Affine>& get_g() const { return g; }  // :336-338

   private:
    bool pinned_ = false;
    void pin() {
        engine_check(h2hip_bases_pin(g[0].x, g.size()), "bases_pin(g)");
        if (int rc = h2hip_bases_pin(g_lagrange[0].x, g_lagrange.size())) {  // do not leave g pinned behind a failed object
            std::string msg = std::string("bases_pin(g_lagrange): ") + h2hip_last_error();
            (void)h2hip_bases_unpin(g[0].x);
            (void)rc;
            throw std::runtime_error(msg);
        }
        pinned_ = true;
    }
    void unpin() {
        if (!pinned_) return;
        (void)h2hip_bases_unpin(g[0].x);
        (void)h2hip_bases_unpin(g_lagrange[0].x);
        pinned_ = false;
    }
};

}  // namespace kzg
}  // namespace poly
}  // namespace halo2_proofs
