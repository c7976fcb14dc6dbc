// evalh.hip -- Evaluator::evaluate_h on the GPU (halo2_proofs/src/plonk/evaluation.rs:280-522).
//
// One lane per row of the extended coset.  Three kernels, in the reference's order, each folding its constraints
// into values[idx] with powers of y exactly as the reference does:
//   evalh_gates_kernel   GraphEvaluator::evaluate of Evaluator::custom_gates (:334-360, :708-749): an interpreter over
//                        the flattened calculations (include/halo2hip.h), intermediates in per-lane scratch
//   evalh_perm_kernel    the permutation argument's constraints (:362-441)
//   evalh_lookup_kernel  one lookup's constraints (:443-518), its compressed-expression graph evaluated in place
// Advice / instance / lookup polynomials arrive in coefficient form and are taken to the extended coset on the
// device with the engine's own coeff_to_extended (ntt.hip), as evaluate_h does at :306-323 and :447-457.
// Arithmetic is the saturated field.cuh (always canonical): this path is bandwidth- and latency-mixed, not the
// VALU-bound inner loop of the MSM, and canonical values make bit-exactness with the reference immediate.
#include <string.h>

#include <vector>

#include "../../include/halo2hip.h"
#include "engine.h"

namespace h2 {

#define EVALH_MAX_ROT 64

struct GraphDev {
    const Fe* constants;
    const int32_t* rotations;
    const h2hip_calculation* calcs;
    const h2hip_value_source* parts;
    uint32_t n_rot, n_calcs;
};

struct ColsDev {
    const Fe* const* fixed;
    const Fe* const* advice;
    const Fe* const* instance;
    const Fe* challenges;
    Fe beta, gamma, theta, y;
    uint32_t log_size;
    int32_t rot_scale;
};

// get_rotation_idx (evaluation.rs:32-34): size is a power of two, so rem_euclid is a mask
__device__ __forceinline__ uint32_t rot_idx(uint32_t idx, int32_t rot, int32_t rot_scale, uint32_t log_size) {
    return (uint32_t)((int32_t)idx + rot * rot_scale) & ((1u << log_size) - 1);
}

// ValueSource::get (evaluation.rs:68-103)
__device__ __forceinline__ Fe vs_get(const GraphDev& g, const ColsDev& c, const h2hip_value_source& v, const uint32_t* rot, const Fe* inter,
                                     const Fe& previous) {
    switch (v.kind) {
        case H2HIP_VS_CONSTANT: return g.constants[v.a];
        case H2HIP_VS_INTERMEDIATE: return inter[v.a];
        case H2HIP_VS_FIXED: return c.fixed[v.a][rot[v.b]];
        case H2HIP_VS_ADVICE: return c.advice[v.a][rot[v.b]];
        case H2HIP_VS_INSTANCE: return c.instance[v.a][rot[v.b]];
        case H2HIP_VS_CHALLENGE: return c.challenges[v.a];
        case H2HIP_VS_BETA: return c.beta;
        case H2HIP_VS_GAMMA: return c.gamma;
        case H2HIP_VS_THETA: return c.theta;
        case H2HIP_VS_Y: return c.y;
        default: return previous;
    }
}

// GraphEvaluator::evaluate (evaluation.rs:708-749) with Calculation::evaluate (:129-178)
template <int MAXI>
__device__ Fe graph_eval(const GraphDev& g, const ColsDev& c, uint32_t idx, const Fe& previous) {
    uint32_t rot[EVALH_MAX_ROT];
    Fe inter[MAXI];
    for (uint32_t r = 0; r < g.n_rot; r++) rot[r] = rot_idx(idx, g.rotations[r], c.rot_scale, c.log_size);
    Fe out = fe_zero<FrP>();
    for (uint32_t q = 0; q < g.n_calcs; q++) {
        const h2hip_calculation cl = g.calcs[q];
        Fe a = vs_get(g, c, cl.x, rot, inter, previous);
        switch (cl.op) {
            case H2HIP_CALC_ADD: out = fe_add<FrP>(a, vs_get(g, c, cl.y, rot, inter, previous)); break;
            case H2HIP_CALC_SUB: out = fe_sub<FrP>(a, vs_get(g, c, cl.y, rot, inter, previous)); break;
            case H2HIP_CALC_MUL: out = fe_mul<FrP>(a, vs_get(g, c, cl.y, rot, inter, previous)); break;
            case H2HIP_CALC_SQUARE: out = fe_sqr<FrP>(a); break;
            case H2HIP_CALC_DOUBLE: out = fe_dbl<FrP>(a); break;
            case H2HIP_CALC_NEGATE: out = fe_neg<FrP>(a); break;
            case H2HIP_CALC_HORNER: {
                Fe factor = vs_get(g, c, cl.y, rot, inter, previous);
                out = a;
                for (uint32_t t = 0; t < cl.parts_count; t++)
                    out = fe_add<FrP>(fe_mul<FrP>(out, factor), vs_get(g, c, g.parts[cl.parts_offset + t], rot, inter, previous));
                break;
            }
            default: out = a;  // Store
        }
        inter[cl.target] = out;
    }
    return out;  // the last calculation's value, or zero for an empty graph
}

template <int MAXI>
__global__ void __launch_bounds__(256) evalh_gates_kernel(GraphDev g, ColsDev c, Fe* values) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (1u << c.log_size)) return;
    values[idx] = graph_eval<MAXI>(g, c, idx, values[idx]);
}

struct PermDev {
    const Fe* const* z;       // permutation_product_coset per set
    const Fe* const* cols;    // the permuted columns' extended cosets, already resolved by (kind, index)
    const Fe* const* cosets;  // pk.permutation.cosets
    const Fe *l0, *l_last, *l_active;
    Fe extended_omega, delta, delta_start;  // delta_start = beta * ZETA (:368)
    uint32_t n_sets, n_cols, chunk_len;
    int32_t last_rotation;
};

__global__ void __launch_bounds__(256) evalh_perm_kernel(PermDev p, ColsDev c, Fe* values) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (1u << c.log_size)) return;
    const Fe one = fe_one<FrP>();
    const uint32_t r_next = rot_idx(idx, 1, c.rot_scale, c.log_size);
    const uint32_t r_last = rot_idx(idx, p.last_rotation, c.rot_scale, c.log_size);
    Fe v = values[idx];
    // l_0(X) * (1 - z_0(X)) = 0                                                   :382-386
    v = fe_add<FrP>(fe_mul<FrP>(v, c.y), fe_mul<FrP>(fe_sub<FrP>(one, p.z[0][idx]), p.l0[idx]));
    // l_last(X) * (z_l(X)^2 - z_l(X)) = 0                                         :387-393
    {
        Fe zl = p.z[p.n_sets - 1][idx];
        v = fe_add<FrP>(fe_mul<FrP>(v, c.y), fe_mul<FrP>(fe_sub<FrP>(fe_sqr<FrP>(zl), zl), p.l_last[idx]));
    }
    // l_0(X) * (z_i(X) - z_{i-1}(omega^(last) X)) = 0                              :394-404
    for (uint32_t s = 1; s < p.n_sets; s++)
        v = fe_add<FrP>(fe_mul<FrP>(v, c.y), fe_mul<FrP>(fe_sub<FrP>(p.z[s][idx], p.z[s - 1][r_last]), p.l0[idx]));
    // (1 - (l_last + l_blind)) * (z_i(wX) prod(p + beta s_j + gamma) - z_i(X) prod(p + delta^j beta X + gamma))   :405-438
    Fe current_delta = fe_mul<FrP>(p.delta_start, fe_pow_u64<FrP>(p.extended_omega, idx));  // beta_term = extended_omega^idx
    for (uint32_t s = 0; s < p.n_sets; s++) {
        const uint32_t j0 = s * p.chunk_len, j1 = j0 + p.chunk_len < p.n_cols ? j0 + p.chunk_len : p.n_cols;
        Fe left = p.z[s][r_next], right = p.z[s][idx];
        for (uint32_t j = j0; j < j1; j++)
            left = fe_mul<FrP>(left, fe_add<FrP>(fe_add<FrP>(p.cols[j][idx], fe_mul<FrP>(c.beta, p.cosets[j][idx])), c.gamma));
        for (uint32_t j = j0; j < j1; j++) {
            right = fe_mul<FrP>(right, fe_add<FrP>(fe_add<FrP>(p.cols[j][idx], current_delta), c.gamma));
            current_delta = fe_mul<FrP>(current_delta, p.delta);
        }
        v = fe_add<FrP>(fe_mul<FrP>(v, c.y), fe_mul<FrP>(fe_sub<FrP>(left, right), p.l_active[idx]));
    }
    values[idx] = v;
}

struct LookupDev {
    const Fe *product, *pin, *ptab;  // extended cosets of product / permuted input / permuted table
    const Fe *l0, *l_last, *l_active;
};

template <int MAXI>
__global__ void __launch_bounds__(256) evalh_lookup_kernel(GraphDev g, LookupDev l, ColsDev c, Fe* values) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (1u << c.log_size)) return;
    const Fe one = fe_one<FrP>();
    const Fe table_value = graph_eval<MAXI>(g, c, idx, fe_zero<FrP>());  // :466-480
    const uint32_t r_next = rot_idx(idx, 1, c.rot_scale, c.log_size), r_prev = rot_idx(idx, -1, c.rot_scale, c.log_size);
    const Fe z = l.product[idx], a_ = l.pin[idx], s_ = l.ptab[idx];
    const Fe a_minus_s = fe_sub<FrP>(a_, s_);
    Fe v = values[idx];
    // l_0(X) * (1 - z(X)) = 0
    v = fe_add<FrP>(fe_mul<FrP>(v, c.y), fe_mul<FrP>(fe_sub<FrP>(one, z), l.l0[idx]));
    // l_last(X) * (z(X)^2 - z(X)) = 0
    v = fe_add<FrP>(fe_mul<FrP>(v, c.y), fe_mul<FrP>(fe_sub<FrP>(fe_sqr<FrP>(z), z), l.l_last[idx]));
    // (1 - (l_last + l_blind)) * (z(wX)(a' + beta)(s' + gamma) - z(X) * table_value) = 0
    {
        Fe lhs = fe_mul<FrP>(fe_mul<FrP>(l.product[r_next], fe_add<FrP>(a_, c.beta)), fe_add<FrP>(s_, c.gamma));
        Fe rhs = fe_mul<FrP>(z, table_value);
        v = fe_add<FrP>(fe_mul<FrP>(v, c.y), fe_mul<FrP>(fe_sub<FrP>(lhs, rhs), l.l_active[idx]));
    }
    // l_0(X) * (a'(X) - s'(X)) = 0
    v = fe_add<FrP>(fe_mul<FrP>(v, c.y), fe_mul<FrP>(a_minus_s, l.l0[idx]));
    // (1 - (l_last + l_blind)) * (a' - s') * (a'(X) - a'(w^-1 X)) = 0
    v = fe_add<FrP>(fe_mul<FrP>(v, c.y), fe_mul<FrP>(fe_mul<FrP>(a_minus_s, fe_sub<FrP>(a_, l.pin[r_prev])), l.l_active[idx]));
    values[idx] = v;
}

// ---------------------------------------------------------------------------------------------- host side

static bool vs_ok(const h2hip_value_source& v, const h2hip_graph& g, const h2hip_evalh_desc& d) {
    switch (v.kind) {
        case H2HIP_VS_CONSTANT: return v.a < g.n_constants;
        case H2HIP_VS_INTERMEDIATE: return v.a < g.num_intermediates;
        case H2HIP_VS_FIXED: return v.a < d.n_fixed && v.b < g.n_rotations;
        case H2HIP_VS_ADVICE: return v.a < d.n_advice && v.b < g.n_rotations;
        case H2HIP_VS_INSTANCE: return v.a < d.n_instance && v.b < g.n_rotations;
        case H2HIP_VS_CHALLENGE: return v.a < d.n_challenges;
        case H2HIP_VS_BETA: case H2HIP_VS_GAMMA: case H2HIP_VS_THETA: case H2HIP_VS_Y: case H2HIP_VS_PREVIOUS: return true;
        default: return false;
    }
}

// every index a kernel will dereference is checked here: a malformed graph must be an error, never a GPU fault
static int graph_validate(const h2hip_graph& g, const h2hip_evalh_desc& d, const char* what) {
    if (g.num_intermediates > 256 || g.n_rotations > EVALH_MAX_ROT) {
        set_error("evaluate_h: %s graph too large for this engine (%u intermediates, %u rotations; limits 256, %d)", what, g.num_intermediates,
                  g.n_rotations, EVALH_MAX_ROT);
        return 1;
    }
    if ((g.n_constants && !g.constants) || (g.n_rotations && !g.rotations) || (g.n_calculations && !g.calculations) || (g.n_parts && !g.parts)) {
        set_error("evaluate_h: %s graph has null arrays", what);
        return 1;
    }
    for (uint32_t q = 0; q < g.n_calculations; q++) {
        const h2hip_calculation& c = g.calculations[q];
        bool ok = c.op <= H2HIP_CALC_STORE && c.target < g.num_intermediates && vs_ok(c.x, g, d);
        if (c.op == H2HIP_CALC_ADD || c.op == H2HIP_CALC_SUB || c.op == H2HIP_CALC_MUL || c.op == H2HIP_CALC_HORNER) ok = ok && vs_ok(c.y, g, d);
        if (c.op == H2HIP_CALC_HORNER) {
            ok = ok && (uint64_t)c.parts_offset + c.parts_count <= g.n_parts;
            for (uint32_t t = 0; ok && t < c.parts_count; t++) ok = vs_ok(g.parts[c.parts_offset + t], g, d);
        }
        if (!ok) {
            set_error("evaluate_h: %s graph, calculation %u is malformed", what, q);
            return 1;
        }
    }
    return 0;
}

struct Arena {
    char* base = nullptr;
    size_t off = 0, cap = 0;
    void* take(size_t bytes) {
        size_t o = off;
        off = (off + bytes + 255) / 256 * 256;
        return off <= cap ? base + o : nullptr;
    }
};

static size_t graph_bytes(const h2hip_graph& g) {
    return 4 * 256 + g.n_constants * sizeof(Fe) + g.n_rotations * 4 + g.n_calculations * sizeof(h2hip_calculation) + g.n_parts * sizeof(h2hip_value_source);
}

static int graph_upload(Arena& ar, const h2hip_graph& g, GraphDev* out, hipStream_t s) {
    Fe* dc = (Fe*)ar.take(g.n_constants * sizeof(Fe));
    int32_t* dr = (int32_t*)ar.take(g.n_rotations * 4);
    h2hip_calculation* dq = (h2hip_calculation*)ar.take(g.n_calculations * sizeof(h2hip_calculation));
    h2hip_value_source* dp = (h2hip_value_source*)ar.take(g.n_parts * sizeof(h2hip_value_source));
    if (!dc || !dr || !dq || !dp) {
        set_error("evaluate_h: arena overflow");
        return 1;
    }
    if (g.n_constants) H2_CHECK(hipMemcpyAsync(dc, g.constants, g.n_constants * sizeof(Fe), hipMemcpyHostToDevice, s));
    if (g.n_rotations) H2_CHECK(hipMemcpyAsync(dr, g.rotations, g.n_rotations * 4, hipMemcpyHostToDevice, s));
    if (g.n_calculations) H2_CHECK(hipMemcpyAsync(dq, g.calculations, g.n_calculations * sizeof(h2hip_calculation), hipMemcpyHostToDevice, s));
    if (g.n_parts) H2_CHECK(hipMemcpyAsync(dp, g.parts, g.n_parts * sizeof(h2hip_value_source), hipMemcpyHostToDevice, s));
    out->constants = dc;
    out->rotations = dr;
    out->calcs = dq;
    out->parts = dp;
    out->n_rot = g.n_rotations;
    out->n_calcs = g.n_calculations;
    return 0;
}

template <class K32, class K128, class K256>
static void launch_by_size(uint32_t num_intermediates, K32 k32, K128 k128, K256 k256) {
    if (num_intermediates <= 32) k32();
    else if (num_intermediates <= 128) k128();
    else k256();
}

int evaluate_h_host(Ctx* c, const h2hip_evalh_desc* d, uint64_t* values) {
    const uint32_t k = d->k, ek = d->extended_k;
    if (k > ek || ek > 28 || ek - k > 8) {
        set_error("evaluate_h: bad domain (k = %u, extended_k = %u)", k, ek);
        return 1;
    }
    const size_t n = (size_t)1 << k, size = (size_t)1 << ek;
    const size_t col_bytes = size * sizeof(Fe);
    if (!d->extended_omega || !d->g_coset || !d->g_coset_inv || !d->y || !d->beta || !d->gamma || !d->theta || !d->l0 || !d->l_last ||
        !d->l_active_row || !values || (d->n_fixed && !d->fixed_cosets) || (d->n_advice && !d->advice_polys) ||
        (d->n_instance && !d->instance_polys) || (d->n_challenges && !d->challenges)) {
        set_error("evaluate_h: null argument");
        return 1;
    }
    if (graph_validate(d->custom_gates, *d, "custom gates")) return 1;
    if (d->n_lookups && (!d->lookup_graphs || !d->lookup_product_polys || !d->lookup_permuted_input_polys || !d->lookup_permuted_table_polys)) {
        set_error("evaluate_h: null lookup arrays");
        return 1;
    }
    for (uint32_t i = 0; i < d->n_lookups; i++)
        if (graph_validate(d->lookup_graphs[i], *d, "lookup")) return 1;
    if (d->n_perm_sets) {
        if (!d->perm_product_cosets || !d->perm_cosets || !d->perm_column_kind || !d->perm_column_index || !d->zeta || !d->delta || d->chunk_len == 0 ||
            (uint64_t)d->n_perm_sets * d->chunk_len < d->n_perm_columns) {
            set_error("evaluate_h: malformed permutation description");
            return 1;
        }
        for (uint32_t j = 0; j < d->n_perm_columns; j++) {
            uint32_t kind = d->perm_column_kind[j], idx = d->perm_column_index[j];
            uint32_t lim = kind == H2HIP_ANY_ADVICE ? d->n_advice : kind == H2HIP_ANY_FIXED ? d->n_fixed : kind == H2HIP_ANY_INSTANCE ? d->n_instance : 0;
            if (idx >= lim) {
                set_error("evaluate_h: permutation column %u out of range", j);
                return 1;
            }
        }
    }
    // ---- device arena
    const size_t n_cols = (size_t)d->n_fixed + d->n_advice + d->n_instance + 3 /* l0, l_last, l_active */ + d->n_perm_sets + d->n_perm_columns +
                          3 /* lookup cosets, reused */ + 1 /* values */;
    size_t need = n_cols * (col_bytes + 256) + 64 * 1024 + graph_bytes(d->custom_gates) + (size_t)d->n_challenges * sizeof(Fe) +
                  8 * ((size_t)d->n_fixed + d->n_advice + d->n_instance + d->n_perm_sets + 2 * (size_t)d->n_perm_columns + 16) + 4096;
    for (uint32_t i = 0; i < d->n_lookups; i++) need += graph_bytes(d->lookup_graphs[i]);
    int rc = c->evalh_ws.ensure(need);
    if (rc) return rc;
    hipStream_t s = c->stream;
    Arena ar;
    ar.base = (char*)c->evalh_ws.p;
    ar.cap = c->evalh_ws.cap;
    auto col_upload = [&](const uint64_t* h, size_t elems, Fe** out) -> int {
        Fe* p = (Fe*)ar.take(col_bytes);
        if (!p || !h) {
            set_error("evaluate_h: null column or arena overflow");
            return 1;
        }
        H2_CHECK(hipMemcpyAsync(p, h, elems * sizeof(Fe), hipMemcpyHostToDevice, s));
        *out = p;
        return 0;
    };
    const Fe ext_omega = *(const Fe*)d->extended_omega;
    NttScale sc;
    {   // distribute_powers_zeta(into_coset) + zero-pad + NTT, as h2hip_coeff_to_extended does
        sc.in_scale = true;
        sc.in3[0] = fe_one<FrP>();
        sc.in3[1] = *(const Fe*)d->g_coset;
        sc.in3[2] = *(const Fe*)d->g_coset_inv;
        sc.in_len = n;
    }
    auto poly_to_coset = [&](const uint64_t* h, Fe** out) -> int {
        int r = col_upload(h, n, out);
        if (r) return r;
        return ntt_device(c, *out, ext_omega, ek, &sc, s);
    };
    std::vector<const Fe*> fixed(d->n_fixed), advice(d->n_advice), instance(d->n_instance);
    Fe* tmp;
    for (uint32_t i = 0; i < d->n_fixed; i++) { if ((rc = col_upload(d->fixed_cosets[i], size, &tmp))) return rc; fixed[i] = tmp; }
    for (uint32_t i = 0; i < d->n_advice; i++) { if ((rc = poly_to_coset(d->advice_polys[i], &tmp))) return rc; advice[i] = tmp; }
    for (uint32_t i = 0; i < d->n_instance; i++) { if ((rc = poly_to_coset(d->instance_polys[i], &tmp))) return rc; instance[i] = tmp; }
    Fe *l0, *l_last, *l_active, *d_values;
    if ((rc = col_upload(d->l0, size, &l0)) || (rc = col_upload(d->l_last, size, &l_last)) || (rc = col_upload(d->l_active_row, size, &l_active)) ||
        (rc = col_upload(values, size, &d_values)))
        return rc;
    auto ptrs_upload = [&](const std::vector<const Fe*>& v, const Fe* const** out) -> int {
        const Fe** p = (const Fe**)ar.take((v.size() + 1) * sizeof(Fe*));
        if (!p) {
            set_error("evaluate_h: arena overflow");
            return 1;
        }
        if (!v.empty()) H2_CHECK(hipMemcpyAsync(p, v.data(), v.size() * sizeof(Fe*), hipMemcpyHostToDevice, s));
        *out = p;
        return 0;
    };
    ColsDev cols;
    if ((rc = ptrs_upload(fixed, &cols.fixed)) || (rc = ptrs_upload(advice, &cols.advice)) || (rc = ptrs_upload(instance, &cols.instance))) return rc;
    {
        Fe* dch = (Fe*)ar.take((d->n_challenges + 1) * sizeof(Fe));
        if (!dch) {
            set_error("evaluate_h: arena overflow");
            return 1;
        }
        if (d->n_challenges) H2_CHECK(hipMemcpyAsync(dch, d->challenges, d->n_challenges * sizeof(Fe), hipMemcpyHostToDevice, s));
        cols.challenges = dch;
    }
    cols.beta = *(const Fe*)d->beta;
    cols.gamma = *(const Fe*)d->gamma;
    cols.theta = *(const Fe*)d->theta;
    cols.y = *(const Fe*)d->y;
    cols.log_size = ek;
    cols.rot_scale = 1 << (ek - k);
    const dim3 grid((uint32_t)((size + 255) / 256)), block(256);

    // ---- custom gates (:334-360)
    GraphDev gd;
    if ((rc = graph_upload(ar, d->custom_gates, &gd, s))) return rc;
    launch_by_size(
        d->custom_gates.num_intermediates, [&] { hipLaunchKernelGGL(evalh_gates_kernel<32>, grid, block, 0, s, gd, cols, d_values); },
        [&] { hipLaunchKernelGGL(evalh_gates_kernel<128>, grid, block, 0, s, gd, cols, d_values); },
        [&] { hipLaunchKernelGGL(evalh_gates_kernel<256>, grid, block, 0, s, gd, cols, d_values); });
    H2_CHECK(hipGetLastError());

    // ---- permutations (:362-441)
    if (d->n_perm_sets) {
        std::vector<const Fe*> z(d->n_perm_sets), pcols(d->n_perm_columns), pcosets(d->n_perm_columns);
        for (uint32_t i = 0; i < d->n_perm_sets; i++) { if ((rc = col_upload(d->perm_product_cosets[i], size, &tmp))) return rc; z[i] = tmp; }
        for (uint32_t j = 0; j < d->n_perm_columns; j++) {
            if ((rc = col_upload(d->perm_cosets[j], size, &tmp))) return rc;
            pcosets[j] = tmp;
            uint32_t kind = d->perm_column_kind[j], idx = d->perm_column_index[j];
            pcols[j] = kind == H2HIP_ANY_ADVICE ? advice[idx] : kind == H2HIP_ANY_FIXED ? fixed[idx] : instance[idx];  // :404-408
        }
        PermDev pd;
        if ((rc = ptrs_upload(z, &pd.z)) || (rc = ptrs_upload(pcols, &pd.cols)) || (rc = ptrs_upload(pcosets, &pd.cosets))) return rc;
        pd.l0 = l0;
        pd.l_last = l_last;
        pd.l_active = l_active;
        pd.extended_omega = ext_omega;
        pd.delta = *(const Fe*)d->delta;
        pd.delta_start = fe_mul<FrP>(cols.beta, *(const Fe*)d->zeta);
        pd.n_sets = d->n_perm_sets;
        pd.n_cols = d->n_perm_columns;
        pd.chunk_len = d->chunk_len;
        pd.last_rotation = d->last_rotation;
        hipLaunchKernelGGL(evalh_perm_kernel, grid, block, 0, s, pd, cols, d_values);
        H2_CHECK(hipGetLastError());
    }

    // ---- lookups (:443-518): the three cosets of a lookup are formed, used and their buffers reused
    if (d->n_lookups) {
        Fe* buf[3];
        for (int t = 0; t < 3; t++)
            if (!(buf[t] = (Fe*)ar.take(col_bytes))) {
                set_error("evaluate_h: arena overflow");
                return 1;
            }
        for (uint32_t i = 0; i < d->n_lookups; i++) {
            const uint64_t* polys[3] = {d->lookup_product_polys[i], d->lookup_permuted_input_polys[i], d->lookup_permuted_table_polys[i]};
            for (int t = 0; t < 3; t++) {
                if (!polys[t]) {
                    set_error("evaluate_h: null lookup polynomial");
                    return 1;
                }
                H2_CHECK(hipMemcpyAsync(buf[t], polys[t], n * sizeof(Fe), hipMemcpyHostToDevice, s));
                if ((rc = ntt_device(c, buf[t], ext_omega, ek, &sc, s))) return rc;
            }
            GraphDev lg;
            if ((rc = graph_upload(ar, d->lookup_graphs[i], &lg, s))) return rc;
            LookupDev ld = {buf[0], buf[1], buf[2], l0, l_last, l_active};
            launch_by_size(
                d->lookup_graphs[i].num_intermediates, [&] { hipLaunchKernelGGL(evalh_lookup_kernel<32>, grid, block, 0, s, lg, ld, cols, d_values); },
                [&] { hipLaunchKernelGGL(evalh_lookup_kernel<128>, grid, block, 0, s, lg, ld, cols, d_values); },
                [&] { hipLaunchKernelGGL(evalh_lookup_kernel<256>, grid, block, 0, s, lg, ld, cols, d_values); });
            H2_CHECK(hipGetLastError());
        }
    }
    H2_CHECK(hipMemcpyAsync(values, d_values, col_bytes, hipMemcpyDeviceToHost, s));
    H2_CHECK(hipStreamSynchronize(s));
    return 0;
}

}  // namespace h2
