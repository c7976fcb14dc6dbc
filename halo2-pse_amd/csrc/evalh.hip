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
// Arithmetic is the saturated field.h (always canonical): this path is bandwidth- and latency-mixed, not the
// VALU-bound inner loop of the MSM, and canonical values make bit-exactness with the reference immediate.
#include <dlfcn.h>
#include <hip/hiprtc.h>
#include <stdlib.h>
#include <string.h>

#include <stdarg.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/halo2hip.h"
#include "engine.h"
#include "evalh_dev.h"

namespace h2 {

#define EVALH_VS_ZERO 11  // internal operand kind: the field's zero (an empty graph's value, evaluation.rs:745-749)

// The flattened graph is compiled on the host (compile_graph below) into a short register-machine program before it is
// run: Store calculations become direct column operands, a Horner calculation becomes one FMA per part placed as soon
// as that part exists, dead calculations are dropped and the surviving intermediates are packed into as few slots as
// their lifetimes allow.  A slot is 36 B of LDS (up to 8 slots), per-lane scratch (up to 256) or a row of a global workspace, so a
// circuit with tens of thousands of calculations still runs with a few dozen slots per lane.  The field operations
// performed per row are the reference's, operation for operation; only where a value waits between them differs.
//
// Arithmetic: the unsaturated 9 x 29-bit multiplier of fieldu.h (the MSM's and the NTT's), about half the
// instructions of the saturated CIOS.  Inside a kernel every value is I-form (a * 2^261, lazily reduced): a column
// element (E-form, canonical) becomes I-form for free as the limbs of 32 * x (fu_from_ext), I * I -> I, constants and
// challenges are converted on the host, and the one value a row writes back goes through the exact reduction
// fu_mul_canon(x, 2^256) -> canonical E-form.  So what is stored is the reference's field element, limb for limb.
// Magnitudes (in units of the modulus r) are tracked statically: a loaded column is < 32, a product of magnitudes a, b is
// < a*b/169 + 1, sums add.  For the interpreter the host walks the straight-line program with these rules and marks the
// operations after which a reduction (one multiplication by 2^261) must be inserted to stay below EVALH_MAG_LIMIT; the
// hand-written constraint code below carries its bounds in comments.  Every addition is followed by a carry propagation so
// that limbs 0..7 are back in [0, 2^29) (fu_mul's operand contract).
enum { OP_ADD = 0, OP_SUB, OP_MUL, OP_SQR, OP_DBL, OP_NEG, OP_MOV, OP_FMA /* dst = dst * y + x */, OP_REDUCE_FLAG = 0x100 };
#define EVALH_MAG_LIMIT 100.0   // |value| < 100 r keeps the top limb below 2^28.3
#define EVALH_MAG_RESULT 15.0   // fu_mul_canon needs |value| < 16 r
#define EVALH_MAG_COLUMN 32.0   // fu_from_ext of a canonical element

struct DevOp {
    uint32_t op, dst;
    h2hip_value_source x, y;  // kind INTERMEDIATE: a = slot
};

struct ProgDev {
    const Fu* constants;  // I-form, canonical
    const int32_t* rotations;
    const DevOp* ops;
    uint32_t n_ops;
    h2hip_value_source result;
};

__device__ __forceinline__ DevOp ld_op(const DevOp* ops, uint32_t q) {
    const H2_CONST_AS u32x8* p = (const H2_CONST_AS u32x8*)(uintptr_t)ops;
    const u32x8 w = p[q];
    DevOp o;
    o.op = w[0];
    o.dst = w[1];
    o.x.kind = w[2];
    o.x.a = w[3];
    o.x.b = w[4];
    o.y.kind = w[5];
    o.y.a = w[6];
    o.y.b = w[7];
    return o;
}
// slot storage: MAXI > 0 -> per-lane scratch; MAXI == 0 -> a global workspace laid out [slot][lane] (coalesced per slot)
extern __shared__ int32_t evalh_lds[];
#define EVALH_LDS_HOT 4  // slots of a scratch-tier program that live in LDS (measured, 100 / 400 gates at 2^20 rows: 2 -> 13.1 / 38.7 ms,
                         // 3 -> 11.9 / 38.3, 4 -> 11.5 / 36.3, 8 -> 12.2 / 40.7; all scratch: 14.6 / 45.4)
__device__ __forceinline__ Fu lds_slot_get(uint32_t i) {
    Fu r;
#pragma unroll
    for (int k = 0; k < 9; k++) r.l[k] = evalh_lds[(i * 9 + k) * 256 + threadIdx.x];
    return r;
}
__device__ __forceinline__ void lds_slot_set(uint32_t i, const Fu& x) {
#pragma unroll
    for (int k = 0; k < 9; k++) evalh_lds[(i * 9 + k) * 256 + threadIdx.x] = x.l[k];
}
// 16 / 64 / 256 slots: the first EVALH_LDS_HOT in LDS, the rest in per-lane scratch; compile_graph numbers the slots by
// how often the program touches them, busiest first.
template <int MAXI>
struct Slots {
    Fu v[MAXI - EVALH_LDS_HOT];
    __device__ __forceinline__ Slots(Fu*, size_t) {}
    __device__ __forceinline__ Fu get(uint32_t i) const { return i < EVALH_LDS_HOT ? lds_slot_get(i) : v[i - EVALH_LDS_HOT]; }
    __device__ __forceinline__ void set(uint32_t i, const Fu& x) {
        if (i < EVALH_LDS_HOT) lds_slot_set(i, x);
        else v[i - EVALH_LDS_HOT] = x;
    }
};
// few slots (the common case: a gate polynomial is folded as soon as it exists): LDS, laid out [slot][limb][thread] so
// that a wave's access is one conflict-free row.  (Registers would be better still, but the slot index is only known at
// run time and LLVM turns any select chain over would-be register slots back into an indexed scratch access.)
template <int N>
struct LdsSlots {
    __device__ __forceinline__ LdsSlots(Fu*, size_t) {}
    __device__ __forceinline__ Fu get(uint32_t i) const { return lds_slot_get(i); }
    __device__ __forceinline__ void set(uint32_t i, const Fu& x) { lds_slot_set(i, x); }
};
template <>
struct Slots<4> : LdsSlots<4> {
    __device__ __forceinline__ Slots(Fu* b, size_t s) : LdsSlots<4>(b, s) {}
};
template <>
struct Slots<8> : LdsSlots<8> {
    __device__ __forceinline__ Slots(Fu* b, size_t s) : LdsSlots<8>(b, s) {}
};
template <>
struct Slots<0> {
    Fu* base;
    size_t stride;
    __device__ __forceinline__ Slots(Fu* b, size_t s) : base(b), stride(s) {}
    __device__ __forceinline__ Fu get(uint32_t i) const { return base[i * stride]; }
    __device__ __forceinline__ void set(uint32_t i, const Fu& x) { base[i * stride] = x; }
};

// ValueSource::get (evaluation.rs:68-103)
template <class S>
__device__ __forceinline__ Fu vs_get(const ProgDev& g, const ColsDev& c, const h2hip_value_source& v, uint32_t idx, const S& slots,
                                     const Fu& previous) {
    switch (v.kind) {
        case H2HIP_VS_CONSTANT: return ld_const_fu(g.constants, v.a);
        case H2HIP_VS_INTERMEDIATE: return slots.get(v.a);
        case H2HIP_VS_FIXED: return ld_i(ld_const_col(c.fixed, v.a)[rot_idx(idx, ld_const_i32(g.rotations, v.b), c.rot_scale, c.log_size)]);
        case H2HIP_VS_ADVICE: return ld_i(ld_const_col(c.advice, v.a)[rot_idx(idx, ld_const_i32(g.rotations, v.b), c.rot_scale, c.log_size)]);
        case H2HIP_VS_INSTANCE: return ld_i(ld_const_col(c.instance, v.a)[rot_idx(idx, ld_const_i32(g.rotations, v.b), c.rot_scale, c.log_size)]);
        case H2HIP_VS_CHALLENGE: return ld_const_fu(c.challenges, v.a);
        case H2HIP_VS_BETA: return c.beta;
        case H2HIP_VS_GAMMA: return c.gamma;
        case H2HIP_VS_THETA: return c.theta;
        case H2HIP_VS_Y: return c.y;
        case H2HIP_VS_PREVIOUS: return previous;
        default: return fu_zero();
    }
}

// GraphEvaluator::evaluate (evaluation.rs:708-749) with Calculation::evaluate (:129-178), over the compiled program.
// Returns the I-form result; the host guarantees its magnitude is below EVALH_MAG_RESULT when it is a slot (a column or a
// constant can also be the result of a degenerate graph: < 32).
template <class S>
__device__ __forceinline__ Fu prog_eval(const ProgDev& g, const ColsDev& c, uint32_t idx, const Fu& previous, S& slots) {
    for (uint32_t q = 0; q < g.n_ops; q++) {
        const DevOp o = ld_op(g.ops, q);
        const Fu a = vs_get(g, c, o.x, idx, slots, previous);
        Fu out;
        switch (o.op & 0xff) {
            case OP_ADD: out = addn(a, vs_get(g, c, o.y, idx, slots, previous)); break;
            case OP_SUB: out = subn(a, vs_get(g, c, o.y, idx, slots, previous)); break;
            case OP_MUL: out = mul_i(a, vs_get(g, c, o.y, idx, slots, previous)); break;
            case OP_SQR: out = fu_sqr<UF>(a); break;
            case OP_DBL: out = fu_norm(fu_dbl(a)); break;
            case OP_NEG: out = fu_norm(fu_neg(a)); break;
            case OP_FMA: out = addn(mul_i(slots.get(o.dst), vs_get(g, c, o.y, idx, slots, previous)), a); break;
            default: out = a;  // OP_MOV
        }
        if (o.op & OP_REDUCE_FLAG) out = mul_i(out, fu_one_i<UF>());
        slots.set(o.dst, out);
    }
    return vs_get(g, c, g.result, idx, slots, previous);
}

// the value a row stores: a column / PreviousValue result is already the canonical element
template <class S>
__device__ __forceinline__ Fe prog_result_e(const ProgDev& g, const ColsDev& c, uint32_t idx, const Fe& previous_e, const Fu& r) {
    switch (g.result.kind) {
        case H2HIP_VS_FIXED: return ld_const_col(c.fixed, g.result.a)[rot_idx(idx, ld_const_i32(g.rotations, g.result.b), c.rot_scale, c.log_size)];
        case H2HIP_VS_ADVICE: return ld_const_col(c.advice, g.result.a)[rot_idx(idx, ld_const_i32(g.rotations, g.result.b), c.rot_scale, c.log_size)];
        case H2HIP_VS_INSTANCE: return ld_const_col(c.instance, g.result.a)[rot_idx(idx, ld_const_i32(g.rotations, g.result.b), c.rot_scale, c.log_size)];
        case H2HIP_VS_PREVIOUS: return previous_e;
        case EVALH_VS_ZERO: return fe_zero<FrP>();
        default: return out_e(r);
    }
}

// lanes = threads in the grid; rows beyond it are taken grid-stride (only the global-workspace form launches fewer lanes than rows)
template <int MAXI>
__global__ void __launch_bounds__(256) evalh_gates_kernel(ProgDev g, ColsDev c, Fe* values, Fu* gws, uint32_t lanes) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= lanes) return;
    Slots<MAXI> slots(gws + tid, lanes);
    for (uint64_t row = tid; row < (1ull << c.log_size); row += lanes) {
        const uint32_t idx = (uint32_t)row;
        const Fe prev = values[idx];
        const Fu r = prog_eval(g, c, idx, ld_i(prev), slots);
        values[idx] = prog_result_e<Slots<MAXI>>(g, c, idx, prev, r);
    }
}

struct PermDev {
    const Fe* const* z;       // permutation_product_coset per set
    const Fe* const* cols;    // the permuted columns' extended cosets, already resolved by (kind, index)
    const Fe* const* cosets;  // pk.permutation.cosets
    const Fe *l0, *l_last, *l_active;
    Fu delta, delta_start;    // I-form canonical; delta_start = beta * ZETA (:368)
    const Fu *pow_lo, *pow_hi;  // the extended domain's two-level power table (the NTT's: omega^i, i < 2^pow_bits; omega^(i << pow_bits)), I-form canonical
    uint32_t pow_bits;
    uint32_t n_sets, n_cols, chunk_len;
    int32_t last_rotation;
};

// Magnitudes in units of r are given in brackets.
__global__ void __launch_bounds__(256) evalh_perm_kernel(PermDev p, ColsDev c, Fe* values) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (1u << c.log_size)) return;
    const Fu one = fu_one_i<UF>();
    const uint32_t r_next = rot_idx(idx, 1, c.rot_scale, c.log_size);
    const uint32_t r_last = rot_idx(idx, p.last_rotation, c.rot_scale, c.log_size);
    Fu v = ld_i(values[idx]);  // [32]
    // l_0(X) * (1 - z_0(X)) = 0                                                   :382-386
    v = fold_y(v, c.y, subn(one, ld_i(ld_const_col(p.z, 0)[idx])), ld_i(p.l0[idx]));  // [1.2] + [33 * 32 / 169 + 1 = 7.3] = [8.5]
    // l_last(X) * (z_l(X)^2 - z_l(X)) = 0                                         :387-393
    {
        const Fu zl = ld_i(ld_const_col(p.z, p.n_sets - 1)[idx]);                                          // [32]
        v = fold_y(v, c.y, subn(fu_sqr<UF>(zl), zl), ld_i(p.l_last[idx]));                  // [1.1] + [(7.1 + 32) * 32 / 169 + 1 = 8.4] = [9.5]
    }
    // l_0(X) * (z_i(X) - z_{i-1}(omega^(last) X)) = 0                              :394-404
    for (uint32_t s = 1; s < p.n_sets; s++)
        v = fold_y(v, c.y, subn(ld_i(ld_const_col(p.z, s)[idx]), ld_i(ld_const_col(p.z, s - 1)[r_last])), ld_i(p.l0[idx]));  // [1.1] + [64 * 32 / 169 + 1 = 13.2] = [14.3]
    // (1 - (l_last + l_blind)) * (z_i(wX) prod(p + beta s_j + gamma) - z_i(X) prod(p + delta^j beta X + gamma))   :405-438
    Fu current_delta = p.delta_start;  // beta * ZETA * extended_omega^idx (beta_term, :366-368 and :412)   [1 .. 2]
    current_delta = mul_i(current_delta, p.pow_lo[idx & ((1u << p.pow_bits) - 1)]);
    if (idx >> p.pow_bits) current_delta = mul_i(current_delta, p.pow_hi[idx >> p.pow_bits]);  // uniform over a workgroup
    for (uint32_t s = 0; s < p.n_sets; s++) {
        const uint32_t j0 = s * p.chunk_len, j1 = j0 + p.chunk_len < p.n_cols ? j0 + p.chunk_len : p.n_cols;
        const Fe* zs = ld_const_col(p.z, s);
        Fu left = ld_i(zs[r_next]), right = ld_i(zs[idx]);  // [32]
        Fu diff = subn(left, right);  // (a set without columns)
        for (uint32_t j = j0; j < j1; j++) {  // both products in one walk over the set's columns: a column's value is read once
            const Fu cg = fu_add(ld_i(ld_const_col(p.cols, j)[idx]), c.gamma);                                // column + gamma, limbs below 2^30 (not normalised)
            const Fu lterm = fu_mul_subh<UF>(c.beta, ld_i(ld_const_col(p.cosets, j)[idx]), fu_neg(cg));        // beta * s_j + (column + gamma) in one pass   [32 + 1.2 + 1 = 34.2]
            const Fu rterm = addn(cg, current_delta);                                                          // [35]
            current_delta = mul_i(current_delta, p.delta);                                                     // [1.1]
            if (j + 1 == j1) {
                // the last column's two products leave as their difference, one reduction for both: [(32 * 34.2 + 32 * 35) / 169 + 1 = 14.1] for a set of
                // one column, [(7.5 * 34.2 + 7.7 * 35) / 169 + 1 = 4.1] otherwise
                diff = fu_mul_sub<UF>(left, lterm, right, rterm);
                break;
            }
            left = mul_i(left, lterm);                                                                         // [32 * 34.2 / 169 + 1 = 7.5], then smaller
            right = mul_i(right, rterm);                                                                       // [7.7]
        }
        v = fold_y(v, c.y, diff, ld_i(p.l_active[idx]));                              // [1.1] + [14.1 * 32 / 169 + 1 = 3.7] = [4.8]
    }
    values[idx] = out_e(v);  // [< 15]
}

template <int MAXI>
__global__ void __launch_bounds__(256) evalh_lookup_kernel(ProgDev g, LookupDev l, ColsDev c, Fe* values, Fu* gws, uint32_t lanes) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= lanes) return;
    Slots<MAXI> slots(gws + tid, lanes);
    for (uint64_t row = tid; row < (1ull << c.log_size); row += lanes) {
        const uint32_t idx = (uint32_t)row;
        const Fu table_value = prog_eval(g, c, idx, fu_zero(), slots);  // :466-480   [< 32]
        lookup_row(l, c, idx, table_value, values);
    }
}

// ---------------------------------------------------------------------------------------------- host side

// caller memory is only 8-byte aligned (4 x u64); Fe is alignas(16)
static inline Fe load_fe(const uint64_t* v) {
    Fe o;
    memcpy(o.l, v, sizeof(o.l));
    return o;
}

// E-form canonical element -> I-form canonical limbs (x * 2^5 mod r, sliced): how constants, challenges and the
// y / beta / gamma / theta scalars enter the kernels
static inline Fu to_i(const Fe& x) {
    Fe t = x;
    for (int k = 0; k < 5; k++) t = fe_dbl<FrP>(t);
    return fu_slice(t);
}

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

// Every index a kernel will dereference is checked here: a malformed graph must be an error, never a GPU fault.  The
// graph must also be in the single-assignment form GraphEvaluator builds (add_calculation, evaluation.rs:570-588:
// every calculation writes a fresh intermediate, operands refer to earlier ones) -- compile_graph relies on it.
static int graph_validate(const h2hip_graph& g, const h2hip_evalh_desc& d, const char* what) {
    if ((g.n_constants && !g.constants) || (g.n_rotations && !g.rotations) || (g.n_calculations && !g.calculations) || (g.n_parts && !g.parts)) {
        set_error("evaluate_h: %s graph has null arrays", what);
        return 1;
    }
    if (g.num_intermediates > (1u << 24) || g.n_calculations > (1u << 24)) {
        set_error("evaluate_h: %s graph too large (%u calculations)", what, g.n_calculations);
        return 1;
    }
    std::vector<uint8_t> defined(g.num_intermediates, 0);
    auto ok_src = [&](const h2hip_value_source& v) { return vs_ok(v, g, d) && (v.kind != H2HIP_VS_INTERMEDIATE || defined[v.a]); };
    for (uint32_t q = 0; q < g.n_calculations; q++) {
        const h2hip_calculation& c = g.calculations[q];
        bool ok = c.op <= H2HIP_CALC_STORE && c.target < g.num_intermediates && !defined[c.target] && ok_src(c.x);
        if (c.op == H2HIP_CALC_ADD || c.op == H2HIP_CALC_SUB || c.op == H2HIP_CALC_MUL || c.op == H2HIP_CALC_HORNER) ok = ok && ok_src(c.y);
        if (c.op == H2HIP_CALC_HORNER) {
            ok = ok && (uint64_t)c.parts_offset + c.parts_count <= g.n_parts;
            for (uint32_t t = 0; ok && t < c.parts_count; t++) ok = ok_src(g.parts[c.parts_offset + t]);
        }
        if (!ok) {
            set_error("evaluate_h: %s graph, calculation %u is malformed (bad index, or not in single-assignment order)", what, q);
            return 1;
        }
        defined[c.target] = 1;
    }
    return 0;
}

// ---- graph -> program (see the comment at DevOp).  Assumes graph_validate passed.
struct Program {
    std::vector<DevOp> ops;
    uint32_t n_slots = 0;
    h2hip_value_source result = {EVALH_VS_ZERO, 0, 0};
};

static Program compile_graph(const h2hip_graph& g) {
    const uint32_t n = g.n_calculations;
    const h2hip_value_source none = {EVALH_VS_ZERO, 0, 0};
    Program P;
    if (n == 0) return P;
    std::vector<uint32_t> def_of(g.num_intermediates, 0);
    for (uint32_t q = 0; q < n; q++) def_of[g.calculations[q].target] = q;
    // a Store is not materialised: whoever reads its target reads its source (ValueSource::get is a pure load)
    auto resolve = [&](h2hip_value_source v) {
        while (v.kind == H2HIP_VS_INTERMEDIATE && g.calculations[def_of[v.a]].op == H2HIP_CALC_STORE) v = g.calculations[def_of[v.a]].x;
        return v;
    };
    // 1. emission order; Horner steps are hoisted to the earliest point their part exists.  The value is unchanged:
    //    ((x*y + p0)*y + p1)... is the same field element whenever each step is executed.
    struct Hoist { uint32_t q, next; bool started; };
    std::vector<Hoist> hs;
    for (uint32_t q = 0; q < n; q++)
        if (g.calculations[q].op == H2HIP_CALC_HORNER) hs.push_back({q, 0, false});
    std::vector<DevOp> ops;
    ops.reserve(n + 16);
    auto avail = [&](const h2hip_value_source& v, uint32_t q) { return v.kind != H2HIP_VS_INTERMEDIATE || def_of[v.a] <= q; };
    auto advance = [&](Hoist& h, uint32_t q) {
        const h2hip_calculation& c = g.calculations[h.q];
        const h2hip_value_source x = resolve(c.x), y = resolve(c.y);
        if (!h.started) {
            if (!avail(x, q) || !avail(y, q)) return;
            ops.push_back({OP_MOV, c.target, x, none});
            h.started = true;
        }
        while (h.next < c.parts_count) {
            const h2hip_value_source p = resolve(g.parts[c.parts_offset + h.next]);
            if (!avail(p, q)) break;
            ops.push_back({OP_FMA, c.target, p, y});
            h.next++;
        }
    };
    static const uint32_t op_of[6] = {OP_ADD, OP_SUB, OP_MUL, OP_SQR, OP_DBL, OP_NEG};
    size_t h_lo = 0;  // Horners before hs[h_lo] are complete
    for (uint32_t q = 0; q < n; q++) {
        const h2hip_calculation& c = g.calculations[q];
        if (c.op <= H2HIP_CALC_NEGATE) {
            const bool binary = c.op <= H2HIP_CALC_MUL;
            ops.push_back({op_of[c.op], c.target, resolve(c.x), binary ? resolve(c.y) : none});
        }
        while (h_lo < hs.size() && hs[h_lo].q <= q) {
            if (hs[h_lo].q == q) advance(hs[h_lo], q);  // its own position: everything it reads exists, so this completes it
            h_lo++;
        }
        if (c.op != H2HIP_CALC_STORE || q + 1 == n)
            for (size_t i = h_lo; i < hs.size() && i < h_lo + 4; i++) advance(hs[i], q);  // look a few Horners ahead (the reference builds 1-2)
    }
    h2hip_value_source result = resolve({H2HIP_VS_INTERMEDIATE, g.calculations[n - 1].target, 0});
    // 2. drop what nothing reads (walking back, every reader of a value is seen before its definition)
    std::vector<uint8_t> needed(g.num_intermediates, 0);
    if (result.kind == H2HIP_VS_INTERMEDIATE) needed[result.a] = 1;
    std::vector<uint8_t> keep(ops.size(), 0);
    for (size_t i = ops.size(); i-- > 0;) {
        const DevOp& o = ops[i];
        if (!needed[o.dst]) continue;
        keep[i] = 1;
        if (o.x.kind == H2HIP_VS_INTERMEDIATE) needed[o.x.a] = 1;
        if (o.y.kind == H2HIP_VS_INTERMEDIATE) needed[o.y.a] = 1;
    }
    // 3. lifetimes -> slots.  An operand whose last reader is this op gives its slot to the op's result (operands are
    //    read into registers before the result is stored).
    const uint32_t never = 0xffffffffu;
    std::vector<uint32_t> last_use(g.num_intermediates, 0), slot(g.num_intermediates, never);
    {
        uint32_t pos = 0;
        for (size_t i = 0; i < ops.size(); i++) {
            if (!keep[i]) continue;
            const DevOp& o = ops[i];
            if (o.x.kind == H2HIP_VS_INTERMEDIATE) last_use[o.x.a] = pos;
            if (o.y.kind == H2HIP_VS_INTERMEDIATE) last_use[o.y.a] = pos;
            if (o.op == OP_FMA) last_use[o.dst] = pos;
            pos++;
        }
        if (result.kind == H2HIP_VS_INTERMEDIATE) last_use[result.a] = never;
    }
    // 4. magnitudes (see the comment at DevOp): walked with the slots, a reduction is requested wherever a sum would
    //    pass EVALH_MAG_LIMIT, and the result is brought below EVALH_MAG_RESULT for the exact final reduction
    std::vector<double> slot_mag;
    auto mag = [&](const h2hip_value_source& v) -> double {
        switch (v.kind) {
            case H2HIP_VS_INTERMEDIATE: return slot_mag[v.a];
            case H2HIP_VS_FIXED: case H2HIP_VS_ADVICE: case H2HIP_VS_INSTANCE: case H2HIP_VS_PREVIOUS: return EVALH_MAG_COLUMN;
            case EVALH_VS_ZERO: return 0.0;
            default: return 1.0;  // constants, challenges, beta / gamma / theta / y: canonical
        }
    };
    std::vector<uint32_t> free_slots;
    uint32_t pos = 0;
    for (size_t i = 0; i < ops.size(); i++) {
        if (!keep[i]) continue;
        DevOp o = ops[i];
        const bool xi = o.x.kind == H2HIP_VS_INTERMEDIATE, yi = o.y.kind == H2HIP_VS_INTERMEDIATE;
        const uint32_t xv = o.x.a, yv = o.y.a;
        if (xi) o.x.a = slot[xv];
        if (yi) o.y.a = slot[yv];
        const double mx = mag(o.x), my = mag(o.y), md = (o.op == OP_FMA) ? slot_mag[slot[o.dst]] : 0.0;
        if (xi && last_use[xv] == pos) free_slots.push_back(slot[xv]);
        if (yi && last_use[yv] == pos && !(xi && yv == xv)) free_slots.push_back(slot[yv]);
        if (slot[o.dst] == never) {  // a definition (FMA steps reuse the slot their MOV took)
            if (!free_slots.empty()) {
                slot[o.dst] = free_slots.back();
                free_slots.pop_back();
            } else {
                slot[o.dst] = P.n_slots++;
            }
        }
        o.dst = slot[o.dst];
        if (slot_mag.size() < P.n_slots) slot_mag.resize(P.n_slots, 0.0);
        double m;
        switch (o.op) {
            case OP_ADD: case OP_SUB: m = mx + my; break;
            case OP_MUL: m = mx * my / 169.0 + 1.0; break;
            case OP_SQR: m = mx * mx / 169.0 + 1.0; break;
            case OP_DBL: m = 2.0 * mx; break;
            case OP_FMA: m = md * my / 169.0 + 1.0 + mx; break;
            default: m = mx;  // NEG, MOV
        }
        if (m > EVALH_MAG_LIMIT) {
            o.op |= OP_REDUCE_FLAG;
            m = m / 169.0 + 1.0;
        }
        slot_mag[o.dst] = m;
        P.ops.push_back(o);
        pos++;
    }
    if (result.kind == H2HIP_VS_INTERMEDIATE) {
        result.a = slot[result.a];
        if (slot_mag[result.a] > EVALH_MAG_RESULT) P.ops.push_back({OP_MOV | OP_REDUCE_FLAG, result.a, result, none});
    }
    // 5. renumber the slots by how often the program touches them, busiest first: the kernels keep the lowest-numbered
    //    slots in LDS and the rest in scratch
    {
        std::vector<uint64_t> touches(P.n_slots, 0);
        for (const DevOp& o : P.ops) {
            touches[o.dst] += (o.op & 0xff) == OP_FMA ? 2 : 1;
            if (o.x.kind == H2HIP_VS_INTERMEDIATE) touches[o.x.a]++;
            if (o.y.kind == H2HIP_VS_INTERMEDIATE) touches[o.y.a]++;
        }
        std::vector<uint32_t> order(P.n_slots), renum(P.n_slots);
        for (uint32_t i = 0; i < P.n_slots; i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a_, uint32_t b_) { return touches[a_] > touches[b_]; });
        for (uint32_t i = 0; i < P.n_slots; i++) renum[order[i]] = i;
        for (DevOp& o : P.ops) {
            o.dst = renum[o.dst];
            if (o.x.kind == H2HIP_VS_INTERMEDIATE) o.x.a = renum[o.x.a];
            if (o.y.kind == H2HIP_VS_INTERMEDIATE) o.y.a = renum[o.y.a];
        }
        if (result.kind == H2HIP_VS_INTERMEDIATE) result.a = renum[result.a];
    }
    P.result = result;
    return P;
}

// ---- the gates kernel generated per circuit (round 4) -------------------------------------------------------------------------------
// The interpreter above pays, per operation and wave, an eight-dword scalar load of the op, two wave-uniform switches and nine LDS
// dwords in and out of a slot: 38 k VALU instructions per wave for a 24-gate program whose arithmetic is ~31 k, at 64-75 % of the
// issue rate because every operation waits for its LDS traffic.  A circuit's program never changes, so it is also emitted as
// straight-line HIP source -- one statement per operation, slots as local variables (registers, allocated by the compiler),
// constants as literals, rotations folded into index variables -- and compiled for gfx950 by hiprtc (dlopen'ed; in-memory headers =
// the library's own field.h / fieldu.h / evalh_dev.h, embedded as text).  The arithmetic per row is the interpreter's,
// operation for operation, so the values are the same limb for limb.  Compilation takes a second or more: it runs on a background
// thread the first time a program is seen (HALO2_HIP_EVALH_CODEGEN=1, default) while the interpreter serves the calls in between;
// code objects are cached per process by source hash and, with HALO2_HIP_CACHE_DIR set, on disk.  =2 compiles inline (tests), =0 off.
static int g_evalh_codegen = 1;
// Programs beyond these stay with the interpreter.  hiprtc takes ~20 ms per operation (0.6 s for a dozen, 4 s for 190, 7 s for 315:
// every field multiplication is ~200 inlined instructions), and a program with many values alive at once spends minutes in the
// register allocator (100 live slots = 900 VGPRs of state: 205 s) for a kernel that would spill anyway.
static uint32_t g_evalh_codegen_max_ops = 1200;
#define EVALH_CODEGEN_MAX_SLOTS 48
static bool g_evalh_gen_fuse = true;  // two products meeting in a sum share one reduction (gen_source); mode + 16 turns it off (A/B, tests)
void evalh_debug_set_codegen(int mode, uint32_t max_ops) {
    g_evalh_codegen = mode & 15;
    g_evalh_gen_fuse = !(mode & 16);
    g_evalh_codegen_max_ops = max_ops ? max_ops : 1200;
}
struct RtcStats {
    std::atomic<uint64_t> compiled{0}, failed{0}, launches{0}, interpreted{0}, disk_hits{0};
};
static RtcStats g_rtc_stats;
void evalh_debug_codegen_stats(uint64_t out[5]) {
    out[0] = g_rtc_stats.compiled;
    out[1] = g_rtc_stats.failed;
    out[2] = g_rtc_stats.launches;
    out[3] = g_rtc_stats.interpreted;
    out[4] = g_rtc_stats.disk_hits;
}

#include "rtc_headers.inc"

struct Hiprtc {
    void* lib = nullptr;
    hiprtcResult (*Create)(hiprtcProgram*, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
    hiprtcResult (*Compile)(hiprtcProgram, int, const char* const*) = nullptr;
    hiprtcResult (*GetCodeSize)(hiprtcProgram, size_t*) = nullptr;
    hiprtcResult (*GetCode)(hiprtcProgram, char*) = nullptr;
    hiprtcResult (*GetLogSize)(hiprtcProgram, size_t*) = nullptr;
    hiprtcResult (*GetLog)(hiprtcProgram, char*) = nullptr;
    hiprtcResult (*Destroy)(hiprtcProgram*) = nullptr;
};
static Hiprtc g_hiprtc;
static std::mutex g_rtc_mu;  // g_hiprtc, g_rtc

static bool hiprtc_load() {  // under g_rtc_mu
    Hiprtc& r = g_hiprtc;
    if (r.lib) return true;
    void* h = dlopen("libhiprtc.so.7", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("libhiprtc.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/libhiprtc.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return false;
    r.Create = (decltype(r.Create))dlsym(h, "hiprtcCreateProgram");
    r.Compile = (decltype(r.Compile))dlsym(h, "hiprtcCompileProgram");
    r.GetCodeSize = (decltype(r.GetCodeSize))dlsym(h, "hiprtcGetCodeSize");
    r.GetCode = (decltype(r.GetCode))dlsym(h, "hiprtcGetCode");
    r.GetLogSize = (decltype(r.GetLogSize))dlsym(h, "hiprtcGetProgramLogSize");
    r.GetLog = (decltype(r.GetLog))dlsym(h, "hiprtcGetProgramLog");
    r.Destroy = (decltype(r.Destroy))dlsym(h, "hiprtcDestroyProgram");
    if (!r.Create || !r.Compile || !r.GetCodeSize || !r.GetCode || !r.GetLogSize || !r.GetLog || !r.Destroy) return false;
    r.lib = h;
    return true;
}

// the program as HIP source.  Operands: constants are literals (I-form limbs), slots are locals, a column operand is a load at the
// row index of its rotation (one index variable per distinct rotation), challenges and beta / gamma / theta / y come with the launch.
static bool g_evalh_gen_barriers = true;
// lookup: the program is a lookup argument's compressed table expression (evaluated with PreviousValue = 0, :466-480) and the kernel ends
// with that argument's five constraints (lookup_row, evalh_dev.h) instead of storing the program's result.
static std::string gen_source(const h2hip_graph& g, const Program& P, bool lookup = false) {
    const bool sched_barriers = g_evalh_gen_barriers;
    std::string src;
    src.reserve(64 * P.ops.size() + 4096);
    char buf[512];
    auto add = [&](const char* fmt, ...) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        src += buf;
    };
    std::vector<uint8_t> use_const(g.n_constants, 0), use_rot(g.n_rotations, 0);
    std::vector<std::vector<uint8_t>> use_col(3);
    auto note = [&](const h2hip_value_source& v) {
        if (v.kind == H2HIP_VS_CONSTANT) use_const[v.a] = 1;
        if (v.kind == H2HIP_VS_FIXED || v.kind == H2HIP_VS_ADVICE || v.kind == H2HIP_VS_INSTANCE) {
            std::vector<uint8_t>& u = use_col[v.kind - H2HIP_VS_FIXED];
            if (u.size() <= v.a) u.resize((size_t)v.a + 1, 0);
            u[v.a] = 1;
            use_rot[v.b] = 1;
        }
    };
    for (const DevOp& o : P.ops) {
        note(o.x);
        note(o.y);
    }
    note(P.result);
    static const char* const tab_name[3] = {"fixed", "advice", "instance"};
    static const char col_name[3] = {'F', 'A', 'N'};
    auto operand = [&](const h2hip_value_source& v) -> std::string {
        char t[96];
        switch (v.kind) {
            case H2HIP_VS_CONSTANT: snprintf(t, sizeof(t), "K%u", v.a); break;
            case H2HIP_VS_INTERMEDIATE: snprintf(t, sizeof(t), "s%u", v.a); break;
            case H2HIP_VS_FIXED: case H2HIP_VS_ADVICE: case H2HIP_VS_INSTANCE:
                snprintf(t, sizeof(t), "ld_i(%c%u[r%u])", col_name[v.kind - H2HIP_VS_FIXED], v.a, v.b);
                break;
            case H2HIP_VS_CHALLENGE: snprintf(t, sizeof(t), "ld_const_fu(c.challenges, %u)", v.a); break;
            case H2HIP_VS_BETA: return "c.beta";
            case H2HIP_VS_GAMMA: return "c.gamma";
            case H2HIP_VS_THETA: return "c.theta";
            case H2HIP_VS_Y: return "c.y";
            case H2HIP_VS_PREVIOUS: return "prev";
            default: return "fu_zero()";
        }
        return t;
    };
    src += "#include \"evalh_dev.h\"\nusing namespace h2;\n";
    src += lookup ? "extern \"C\" __global__ void __launch_bounds__(256) evalh_lookup_gen(ColsDev c, LookupDev l, Fe* values) {\n"
                  : "extern \"C\" __global__ void __launch_bounds__(256) evalh_gates_gen(ColsDev c, Fe* __restrict__ values) {\n";
    src += "    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;\n"
           "    if (idx >= (1u << c.log_size)) return;\n";
    src += lookup ? "    const Fu prev = fu_zero();\n"
                  : "    const Fe prev_e = values[idx];\n"
                    "    const Fu prev = ld_i(prev_e);\n";
    src += "    (void)prev;\n";
    for (uint32_t i = 0; i < g.n_constants; i++)
        if (use_const[i]) {
            const Fu k = to_i(load_fe(g.constants + 4 * (size_t)i));
            add("    const Fu K%u = {{%d, %d, %d, %d, %d, %d, %d, %d, %d}};\n", i, k.l[0], k.l[1], k.l[2], k.l[3], k.l[4], k.l[5], k.l[6], k.l[7], k.l[8]);
        }
    for (int t = 0; t < 3; t++)
        for (uint32_t i = 0; i < use_col[t].size(); i++)
            if (use_col[t][i]) add("    const Fe* __restrict__ %c%u = ld_const_col(c.%s, %u);\n", col_name[t], i, tab_name[t], i);
    for (uint32_t i = 0; i < g.n_rotations; i++)
        if (use_rot[i]) add("    const uint32_t r%u = rot_idx(idx, %d, c.rot_scale, c.log_size);\n", i, g.rotations[i]);
    if (P.n_slots) {
        src += "    Fu s0";
        for (uint32_t i = 1; i < P.n_slots; i++) add(", s%u", i);
        src += ";\n";
    }
    // Two products that meet in a sum or difference share ONE Montgomery reduction (fu_mul_sub: (ab - cd) / 2^261, the columns of both
    // products in one accumulator; a sum passes -c): a product whose only reader is the next ADD / SUB / Horner step on its slot is not
    // emitted where the program has it but inside that reader.  Per gate of the usual shape -- selector * (product - product + ...)
    // folded with y -- that is two of six reductions.  The value is the same field element (the reduced representative may differ by a
    // multiple of r before the closing exact reduction, which the interpreter's magnitude bounds cover: the fused form's bound,
    // (|a||b| + |c||d|) / 169 + 1, is below the sum of the two separate ones).
    const size_t n_ops = P.ops.size();
    std::vector<int> fused_into(n_ops, -1);  // product i is emitted inside op fused_into[i]
    if (g_evalh_gen_fuse) {
        auto is_slot = [](const h2hip_value_source& v) { return v.kind == H2HIP_VS_INTERMEDIATE; };
        for (size_t i = 0; i < n_ops; i++) {
            const DevOp& m = P.ops[i];
            if (m.op != OP_MUL) continue;  // (a product carrying a reduction request stays where it is)
            const uint32_t X = m.dst;
            int reader = -1;
            bool ok = true;
            for (size_t j = i + 1; j < n_ops && ok; j++) {
                const DevOp& o = P.ops[j];
                const uint32_t k = o.op & 0xff;
                const bool rx = is_slot(o.x) && o.x.a == X, ry = is_slot(o.y) && o.y.a == X, rd = k == OP_FMA && o.dst == X;
                if (reader < 0) {
                    if (rx || ry || rd) {
                        // the one reader: an ADD / SUB taking it as either operand (not both), or a Horner step taking it as the addend
                        if ((k == OP_ADD || k == OP_SUB) && rx != ry && !rd) reader = (int)j;
                        else if (k == OP_FMA && rx && !ry && !rd) reader = (int)j;
                        else ok = false;
                        if (ok && o.dst == X) break;  // the reader overwrites the slot: nothing later sees the product
                        continue;
                    }
                    if (o.dst == X) ok = false;  // overwritten unread: dead code the compiler kept (cannot happen), leave it alone
                    // the product's own operands must still hold their values when the reader runs
                    if ((is_slot(m.x) && o.dst == m.x.a) || (is_slot(m.y) && o.dst == m.y.a)) ok = false;
                } else {
                    if (rx || ry || rd) ok = false;   // a second reader
                    else if (o.dst == X) break;       // overwritten: the product was dead after its one reader
                }
            }
            if (ok && reader >= 0 && !(is_slot(P.result) && P.result.a == X && P.ops[reader].dst != X)) fused_into[i] = reader;
        }
        // an ADD / SUB fuses only when BOTH operands are such products; a Horner step when its addend is
        std::vector<int> cnt(n_ops, 0);
        for (size_t i = 0; i < n_ops; i++)
            if (fused_into[i] >= 0) cnt[fused_into[i]]++;
        for (size_t i = 0; i < n_ops; i++)
            if (fused_into[i] >= 0) {
                const uint32_t k = P.ops[fused_into[i]].op & 0xff;
                if ((k == OP_ADD || k == OP_SUB) && cnt[fused_into[i]] != 2) fused_into[i] = -1;
            }
    }
    auto product_of = [&](size_t j, const h2hip_value_source& v) -> const DevOp* {  // the product fused into op j that v names
        for (size_t i = 0; i < j; i++)
            if (fused_into[i] == (int)j && v.kind == H2HIP_VS_INTERMEDIATE && P.ops[i].dst == v.a) return &P.ops[i];
        return nullptr;
    };
    for (size_t q = 0; q < n_ops; q++) {
        const DevOp& o = P.ops[q];
        if (fused_into[q] >= 0) continue;  // emitted inside its reader
        const std::string x = operand(o.x), y = operand(o.y);
        const uint32_t dst = o.dst;
        const DevOp *px = product_of(q, o.x), *py = product_of(q, o.y);
        const uint32_t kind = o.op & 0xff;
        if ((kind == OP_ADD || kind == OP_SUB) && px && py) {
            add("    s%u = fu_mul_sub<UF>(%s, %s, ", dst, operand(px->x).c_str(), operand(px->y).c_str());
            add(kind == OP_SUB ? "%s, %s);\n" : "fu_neg(%s), %s);\n", operand(py->x).c_str(), operand(py->y).c_str());
        } else if (kind == OP_FMA && px) {
            add("    s%u = fu_mul_sub<UF>(s%u, %s, fu_neg(%s), %s);\n", dst, dst, y.c_str(), operand(px->x).c_str(), operand(px->y).c_str());
        } else
        switch (kind) {
            case OP_ADD: add("    s%u = addn(%s, %s);\n", dst, x.c_str(), y.c_str()); break;
            case OP_SUB: add("    s%u = subn(%s, %s);\n", dst, x.c_str(), y.c_str()); break;
            case OP_MUL: add("    s%u = mul_i(%s, %s);\n", dst, x.c_str(), y.c_str()); break;
            case OP_SQR: add("    s%u = fu_sqr<UF>(%s);\n", dst, x.c_str()); break;
            case OP_DBL: add("    s%u = fu_norm(fu_dbl(%s));\n", dst, x.c_str()); break;
            case OP_NEG: add("    s%u = fu_norm(fu_neg(%s));\n", dst, x.c_str()); break;
            case OP_FMA: add("    s%u = addn(mul_i(s%u, %s), %s);\n", dst, dst, y.c_str(), x.c_str()); break;
            default: add("    s%u = %s;\n", dst, x.c_str());  // OP_MOV
        }
        if (o.op & OP_REDUCE_FLAG) add("    s%u = mul_i(s%u, fu_one_i<UF>());\n", dst, dst);
        // one scheduling region per operation: the whole program as ONE basic block (tens of thousands of instructions) costs the
        // scheduler and the register allocator minutes and buys nothing -- a field multiplication already fills the pipeline
        if (sched_barriers) src += "    __builtin_amdgcn_sched_barrier(0);\n";
    }
    if (lookup) {
        add("    lookup_row(l, c, idx, %s, values);\n}\n", operand(P.result).c_str());
        return src;
    }
    switch (P.result.kind) {  // prog_result_e
        case H2HIP_VS_FIXED: case H2HIP_VS_ADVICE: case H2HIP_VS_INSTANCE:
            add("    values[idx] = %c%u[r%u];\n", col_name[P.result.kind - H2HIP_VS_FIXED], P.result.a, P.result.b);
            break;
        case H2HIP_VS_PREVIOUS: src += "    (void)prev_e;\n"; break;
        case EVALH_VS_ZERO: src += "    values[idx] = fe_zero<FrP>();\n"; break;
        default: add("    values[idx] = out_e(%s);\n", operand(P.result).c_str());
    }
    src += "}\n";
    return src;
}

static uint64_t fnv1a64_raw(const char* p, size_t n, uint64_t h) {
    for (size_t i = 0; i < n; i++) {
        h ^= (unsigned char)p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}
// key of a generated kernel: its source AND the embedded headers it is compiled against (a code object left in HALO2_HIP_CACHE_DIR by
// another build of the library, whose arithmetic headers differ, must not be picked up)
static uint64_t fnv1a64(const std::string& s) {
    static const uint64_t headers = [] {
        uint64_t h = 0xcbf29ce484222325ull;
        for (int i = 0; i < H2_RTC_N_HEADERS; i++) h = fnv1a64_raw(H2_RTC_HEADER_TEXT[i], strlen(H2_RTC_HEADER_TEXT[i]), h);
        return h;
    }();
    return fnv1a64_raw(s.data(), s.size(), headers);
}

struct RtcEntry {
    std::mutex m;
    int state = 0;  // 0 compiling, 1 ready, 2 failed
    std::vector<char> code;
    std::string log;
    std::thread th;
    double compile_s = 0.0;
    // a process that exits without h2hip_shutdown still destroys the cache (static destruction): a joinable std::thread must not meet its destructor
    ~RtcEntry() {
        if (th.joinable()) th.join();
    }
};
static std::map<uint64_t, std::shared_ptr<RtcEntry>> g_rtc;

static std::string rtc_cache_path(uint64_t key) {
    const char* dir = getenv("HALO2_HIP_CACHE_DIR");
    if (!dir || !*dir) return std::string();
    char name[64];
    snprintf(name, sizeof(name), "/evalh_gates_%016llx_gfx950.co", (unsigned long long)key);
    return std::string(dir) + name;
}

static void rtc_compile(std::shared_ptr<RtcEntry> e, std::string src, uint64_t key) {
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<char> code;
    std::string log;
    bool ok = false;
    const std::string path = rtc_cache_path(key);
    if (!path.empty()) {  // a code object an earlier process left for this very source
        if (FILE* f = fopen(path.c_str(), "rb")) {
            fseek(f, 0, SEEK_END);
            const long sz = ftell(f);
            fseek(f, 0, SEEK_SET);
            if (sz > 0) {
                code.resize((size_t)sz);
                ok = fread(code.data(), 1, (size_t)sz, f) == (size_t)sz;
            }
            fclose(f);
            if (ok) g_rtc_stats.disk_hits++;
        }
    }
    if (!ok) {
        // one compilation at a time: a circuit brings a gates kernel and one kernel per lookup, each on its own background thread, and
        // nothing here relies on the run-time compiler being re-entrant
        static std::mutex compile_mu;
        std::lock_guard<std::mutex> one(compile_mu);
        hiprtcProgram prog = nullptr;
        hiprtcResult r = g_hiprtc.Create(&prog, src.c_str(), "evalh_gates_gen.hip", H2_RTC_N_HEADERS, H2_RTC_HEADER_TEXT, H2_RTC_HEADER_NAMES);
        if (r == HIPRTC_SUCCESS) {
            const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
            r = g_hiprtc.Compile(prog, 3, opts);
            size_t ls = 0;
            if (g_hiprtc.GetLogSize(prog, &ls) == HIPRTC_SUCCESS && ls > 1) {
                log.resize(ls);
                (void)g_hiprtc.GetLog(prog, &log[0]);
            }
            size_t cs = 0;
            if (r == HIPRTC_SUCCESS && g_hiprtc.GetCodeSize(prog, &cs) == HIPRTC_SUCCESS && cs) {
                code.resize(cs);
                ok = g_hiprtc.GetCode(prog, code.data()) == HIPRTC_SUCCESS;
            }
            (void)g_hiprtc.Destroy(&prog);
        }
        if (ok && !path.empty()) {  // write-then-rename: a concurrent reader sees the whole file or none
            const std::string tmp = path + ".tmp" + std::to_string((unsigned long long)key ^ (unsigned long long)(uintptr_t)&code);
            if (FILE* f = fopen(tmp.c_str(), "wb")) {
                const bool w = fwrite(code.data(), 1, code.size(), f) == code.size();
                fclose(f);
                if (!w || rename(tmp.c_str(), path.c_str()) != 0) (void)remove(tmp.c_str());
            }
        }
    }
    std::lock_guard<std::mutex> lk(e->m);
    e->compile_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    e->log.swap(log);
    if (ok) {
        e->code.swap(code);
        e->state = 1;
        g_rtc_stats.compiled++;
    } else {
        e->state = 2;
        g_rtc_stats.failed++;
    }
}

// test / tooling hook, no GPU needed: the source a graph's program is emitted as, and (compile != 0) what hiprtc makes of it
int evalh_debug_codegen_source(const h2hip_graph* g, char* buf, size_t cap, size_t* len, int compile, double* seconds, size_t* code_bytes) {
    if (!g || !len) {
        set_error("evaluate_h: null argument");
        return 1;
    }
    h2hip_evalh_desc any;
    memset(&any, 0, sizeof(any));
    any.n_fixed = any.n_advice = any.n_instance = any.n_challenges = 0xffffffffu;
    if (graph_validate(*g, any, "given")) return 1;
    const Program P = compile_graph(*g);
    const std::string src = gen_source(*g, P, (compile & 2) != 0);  // bit 1: the graph as a lookup's table expression (evalh_lookup_gen)
    compile &= 1;
    *len = src.size();
    if (buf && cap) {
        const size_t m = src.size() < cap - 1 ? src.size() : cap - 1;
        memcpy(buf, src.data(), m);
        buf[m] = 0;
    }
    if (!compile) return 0;
    {
        std::lock_guard<std::mutex> lk(g_rtc_mu);
        if (!hiprtc_load()) {
            set_error("evaluate_h: libhiprtc.so could not be loaded");
            return 2;
        }
    }
    auto e = std::make_shared<RtcEntry>();
    rtc_compile(e, src, fnv1a64(src));
    if (seconds) *seconds = e->compile_s;
    if (code_bytes) *code_bytes = e->code.size();
    if (e->state != 1) {
        set_error("evaluate_h: hiprtc rejected the generated kernel: %.300s", e->log.c_str());
        return 2;
    }
    return 0;
}

// join the compile threads (h2hip_shutdown, after the contexts are released); the compiled code stays cached for a later init
void evalh_rtc_shutdown() {
    std::lock_guard<std::mutex> lk(g_rtc_mu);
    for (auto& kv : g_rtc)
        if (kv.second->th.joinable()) kv.second->th.join();
}

void evalh_modules_free(Ctx* c) {
    for (auto& kv : c->evalh_mods)
        if (kv.second.first) (void)hipModuleUnload((hipModule_t)kv.second.first);
    c->evalh_mods.clear();
}

// the generated kernel for this program on this device, or nullptr: not wanted, not compiled yet (the interpreter serves this
// call), or not compilable (it serves every call)
// what gen_source reads of a program, hashed: every call of evaluate_h comes here, and writing (and hashing) 20 KB of source each time to
// find a kernel that is already loaded is ~150 us of host time per call
static uint64_t program_fingerprint(const h2hip_graph& g, const Program& P, bool lookup) {
    uint64_t h = 0xcbf29ce484222325ull;
    if (!P.ops.empty()) h = fnv1a64_raw((const char*)P.ops.data(), P.ops.size() * sizeof(DevOp), h);
    if (g.n_constants) h = fnv1a64_raw((const char*)g.constants, (size_t)g.n_constants * 32, h);
    if (g.n_rotations) h = fnv1a64_raw((const char*)g.rotations, (size_t)g.n_rotations * sizeof(int32_t), h);
    const uint32_t tail[6] = {P.result.kind, P.result.a, P.result.b, P.n_slots, lookup ? 1u : 0u, (g_evalh_gen_fuse ? 1u : 0u) | (g_evalh_gen_barriers ? 2u : 0u)};
    return fnv1a64_raw((const char*)tail, sizeof(tail), h);
}
static std::map<uint64_t, uint64_t> g_rtc_key_of;  // program fingerprint -> source key (under g_rtc_mu)

static hipFunction_t gates_kernel_for(Ctx* c, const h2hip_graph& g, const Program& P, bool lookup = false) {
    if (g_evalh_codegen <= 0 || (P.ops.empty() && !lookup) || P.ops.size() > g_evalh_codegen_max_ops || P.n_slots > EVALH_CODEGEN_MAX_SLOTS) return nullptr;
    const uint64_t fp = program_fingerprint(g, P, lookup);
    {
        std::lock_guard<std::mutex> lk(g_rtc_mu);
        auto known = g_rtc_key_of.find(fp);
        if (known != g_rtc_key_of.end()) {
            auto hit = c->evalh_mods.find(known->second);
            if (hit != c->evalh_mods.end()) return (hipFunction_t)hit->second.second;
        }
    }
    std::string src = gen_source(g, P, lookup);
    const uint64_t key = fnv1a64(src);
    {
        std::lock_guard<std::mutex> lk(g_rtc_mu);
        g_rtc_key_of[fp] = key;
    }
    auto hit = c->evalh_mods.find(key);
    if (hit != c->evalh_mods.end()) return (hipFunction_t)hit->second.second;
    std::shared_ptr<RtcEntry> e;
    bool mine = false;  // this call created the entry and compiles it inline
    {
        std::lock_guard<std::mutex> lk(g_rtc_mu);
        auto it = g_rtc.find(key);
        if (it != g_rtc.end()) {
            e = it->second;
            if (g_evalh_codegen >= 2 && e->th.joinable()) e->th.join();  // tests: wait for a compile another call started
        } else {
            if (!hiprtc_load()) return nullptr;
            try {
                e = std::make_shared<RtcEntry>();
                g_rtc[key] = e;
            } catch (...) {
                return nullptr;
            }
            mine = true;
            if (g_evalh_codegen == 1) {
                try {
                    e->th = std::thread(rtc_compile, e, src, key);
                    mine = false;
                } catch (...) {  // no thread to be had: compile inline
                }
            }
        }
    }
    if (mine) rtc_compile(e, src, key);
    std::lock_guard<std::mutex> lk(e->m);
    if (e->state != 1) return nullptr;
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    if (hipModuleLoadData(&mod, e->code.data()) != hipSuccess || hipModuleGetFunction(&fn, mod, lookup ? "evalh_lookup_gen" : "evalh_gates_gen") != hipSuccess) {
        (void)hipGetLastError();
        if (mod) (void)hipModuleUnload(mod);
        c->evalh_mods[key] = {nullptr, nullptr};  // do not try again on this device
        return nullptr;
    }
    c->evalh_mods[key] = {(void*)mod, (void*)fn};
    return fn;
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

static size_t prog_bytes(const h2hip_graph& g, const Program& P) {
    return 3 * 256 + g.n_constants * sizeof(Fu) + g.n_rotations * 4 + P.ops.size() * sizeof(DevOp);
}

// Where the slots of a program live: per-lane scratch in three sizes, or (past 256) a global workspace of
// n_slots x lanes elements with the rows taken grid-stride by `lanes` threads.
struct SlotPlan {
    int tier;        // 4, 8 (LDS), 16, 64, 256 (scratch) or 0 (global workspace)
    uint32_t lanes;  // threads launched
    size_t ws_bytes; // global workspace (tier 0)
};

static uint32_t g_evalh_max_local_slots = 256;  // debug knob (tests force the global-workspace form with it)

static int slot_plan(uint32_t n_slots, size_t size, SlotPlan* out) {
    const uint32_t all = (uint32_t)((size + 255) / 256 * 256);
    if (n_slots <= g_evalh_max_local_slots && n_slots <= 256) {
        out->tier = n_slots <= 4 ? 4 : n_slots <= 8 ? 8 : n_slots <= 16 ? 16 : n_slots <= 64 ? 64 : 256;
        out->lanes = all;
        out->ws_bytes = 0;
        return 0;
    }
    uint32_t lanes = all < 256u * 2048u ? all : 256u * 2048u;  // at most every wave slot of the chip
    const size_t budget = (size_t)32 << 30;
    while ((size_t)n_slots * lanes * sizeof(Fu) > budget && lanes > 16384) lanes /= 2;
    if ((size_t)n_slots * lanes * sizeof(Fu) > budget) {
        set_error("evaluate_h: a graph with %u simultaneously live intermediates does not fit this engine's workspace", n_slots);
        return H2HIP_ENOMEM;
    }
    out->tier = 0;
    out->lanes = lanes;
    out->ws_bytes = (size_t)n_slots * lanes * sizeof(Fu);
    return 0;
}

void evalh_debug_set_max_local_slots(uint32_t v) { g_evalh_max_local_slots = v; }
static size_t g_evalh_lookup_group_bytes = (size_t)2 << 30;  // HBM one group of lookup cosets may take (tests shrink it to force several groups)
void evalh_debug_set_lookup_group_bytes(uint64_t v) { g_evalh_lookup_group_bytes = v ? (size_t)v : (size_t)2 << 30; }

// field multiplications one row of the compiled program performs (products, squares, Horner steps, inserted reductions, and the
// exact reduction of the stored value): the numerator of the gates kernel's valu_roofline in bench.py
int evalh_debug_program_muls(const h2hip_graph* g, uint32_t* n_mul) {
    if (!g || !n_mul) {
        set_error("evaluate_h: null argument");
        return 1;
    }
    h2hip_evalh_desc any;
    memset(&any, 0, sizeof(any));
    any.n_fixed = any.n_advice = any.n_instance = any.n_challenges = 0xffffffffu;
    if (graph_validate(*g, any, "given")) return 1;
    const Program P = compile_graph(*g);
    uint32_t m = 0;
    for (const DevOp& o : P.ops) {
        const uint32_t k = o.op & 0xff;
        if (k == OP_MUL || k == OP_SQR || k == OP_FMA) m++;
        if (o.op & OP_REDUCE_FLAG) m++;
    }
    if (P.result.kind == H2HIP_VS_INTERMEDIATE || P.result.kind <= H2HIP_VS_CONSTANT || (P.result.kind >= H2HIP_VS_CHALLENGE && P.result.kind <= H2HIP_VS_Y)) m++;
    *n_mul = m;
    return 0;
}

int evalh_debug_compile_stats(const h2hip_graph* g, uint32_t* n_ops, uint32_t* n_slots) {
    if (!g || !n_ops || !n_slots) {
        set_error("evaluate_h: null argument");
        return 1;
    }
    h2hip_evalh_desc any;  // column counts are not known here: accept every column index
    memset(&any, 0, sizeof(any));
    any.n_fixed = any.n_advice = any.n_instance = any.n_challenges = 0xffffffffu;
    if (graph_validate(*g, any, "given")) return 1;
    const Program P = compile_graph(*g);
    *n_ops = (uint32_t)P.ops.size();
    *n_slots = P.n_slots;
    return 0;
}

// Everything a kernel will dereference is checked here, on the host, before any device work: a malformed description is
// H2HIP_EINVAL, never a GPU fault.  Needs no device (api.hip calls it before the engine is entered).
int evaluate_h_validate(const h2hip_evalh_desc* d, const void* values) {
    const uint32_t k = d->k, ek = d->extended_k;
    if (k > ek || ek > 28 || ek - k > 8) {
        set_error("evaluate_h: bad domain (k = %u, extended_k = %u)", k, ek);
        return 1;
    }
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
    return 0;
}

// Everything small the kernels read (programs, constants, rotations, column-pointer tables, challenges, the power table)
// is laid out in one host image of a metadata region at the head of the arena and uploaded with ONE copy before any kernel
// is queued.  A copy from pageable memory is synchronous in HIP and ordered after the stream's earlier work, so a copy
// between two kernels would make the host wait for the first and leave the GPU idle until the second is launched.
struct MetaBlob {
    std::vector<char> host;
    char* dev_base = nullptr;
    size_t cap = 0;
    bool overflow = false;
    template <class T>
    T* put(const T* src, size_t count) {
        size_t off = (host.size() + 255) / 256 * 256;
        const size_t bytes = count * sizeof(T);
        if (off + bytes > cap) {
            overflow = true;
            return (T*)dev_base;
        }
        host.resize(off + bytes);
        if (bytes) memcpy(host.data() + off, src, bytes);
        return (T*)(dev_base + off);
    }
};

static void prog_put(MetaBlob& mb, const h2hip_graph& g, const Program& P, ProgDev* out) {
    std::vector<Fu> hc(g.n_constants);
    for (uint32_t i = 0; i < g.n_constants; i++) hc[i] = to_i(load_fe(g.constants + 4 * (size_t)i));
    out->constants = mb.put(hc.data(), hc.size());
    out->rotations = mb.put(g.rotations, g.n_rotations);
    out->ops = mb.put(P.ops.data(), P.ops.size());
    out->n_ops = (uint32_t)P.ops.size();
    out->result = P.result;
}

// dev = false: every column pointer in `d` and `values` are host memory (the reference's Vec<F>s); values is updated in place
//              and the call returns when it is.
// dev = true:  the columns and `values` are device memory (extended cosets are used where they lie, the first NTT pass reads
//              coefficient-form polynomials where they lie); the graphs, challenges and scalars stay host pointers; the
//              kernels are queued on `s` and the call returns without waiting for them.
int evaluate_h_host(Ctx* c, const h2hip_evalh_desc* d, uint64_t* values, bool dev, hipStream_t s) {
    const uint32_t k = d->k, ek = d->extended_k;
    const size_t n = (size_t)1 << k, size = (size_t)1 << ek;
    const size_t col_bytes = size * sizeof(Fe);
    // programs first: their slot counts size the workspaces
    const Program gates_prog = compile_graph(d->custom_gates);
    std::vector<Program> lookup_progs(d->n_lookups);
    for (uint32_t i = 0; i < d->n_lookups; i++) lookup_progs[i] = compile_graph(d->lookup_graphs[i]);
    int rc;
    SlotPlan gates_plan;
    std::vector<SlotPlan> lookup_plans(d->n_lookups);
    if ((rc = slot_plan(gates_prog.n_slots, size, &gates_plan))) return rc;
    size_t slots_ws = gates_plan.ws_bytes;
    for (uint32_t i = 0; i < d->n_lookups; i++) {
        if ((rc = slot_plan(lookup_progs[i].n_slots, size, &lookup_plans[i]))) return rc;
        if (lookup_plans[i].ws_bytes > slots_ws) slots_ws = lookup_plans[i].ws_bytes;
    }
    // ---- device arena: [metadata | columns the engine owns]
    // lookup cosets: three buffers per lookup for as many lookups as 2 GB hold (a group); the first group's transforms join the advice /
    // instance batch, later groups reuse the buffers once the kernels of the group before are queued
    size_t lk_group = d->n_lookups;
    if (lk_group) {
        const size_t fit = g_evalh_lookup_group_bytes / (3 * col_bytes);
        if (lk_group > fit) lk_group = fit ? fit : 1;
    }
    const size_t n_cols = dev ? (size_t)d->n_advice + d->n_instance + 3 * lk_group
                              : (size_t)d->n_fixed + d->n_advice + d->n_instance + 3 /* l0, l_last, l_active */ + d->n_perm_sets + d->n_perm_columns +
                                    3 * lk_group /* lookup cosets, reused per group */ + 1 /* values */;
    size_t meta_cap = 64 * 1024 + prog_bytes(d->custom_gates, gates_prog) + ((size_t)d->n_challenges + 28) * sizeof(Fu) +
                      (8 + 256) * ((size_t)d->n_fixed + d->n_advice + d->n_instance + d->n_perm_sets + 2 * (size_t)d->n_perm_columns + 16);
    for (uint32_t i = 0; i < d->n_lookups; i++) meta_cap += prog_bytes(d->lookup_graphs[i], lookup_progs[i]);
    meta_cap = (meta_cap + 255) / 256 * 256;
    if ((rc = c->evalh_ws.ensure(meta_cap + n_cols * (col_bytes + 256) + 4096))) return rc;
    if (slots_ws && (rc = c->evalh_slots.ensure(slots_ws))) return rc;
    Fu* const gws = (Fu*)c->evalh_slots.p;
    if ((rc = c->ws_acquire(s))) return rc;
    WsGuard guard(c, s);
    Arena ar;
    ar.base = (char*)c->evalh_ws.p;
    ar.cap = c->evalh_ws.cap;
    MetaBlob mb;
    mb.dev_base = (char*)ar.take(meta_cap);
    mb.cap = meta_cap;
    if (!mb.dev_base) {
        set_error("evaluate_h: arena overflow");
        return 1;
    }
    // where every column will live: an extended coset already in HBM is used in place, anything else gets arena space
    struct Upload { Fe* dst; const uint64_t* src; size_t elems; bool late; };
    std::vector<Upload> uploads;  // host -> device column copies, issued after the metadata; `late`: only the permutation kernel reads it
    bool bad = false;
    auto place = [&](const uint64_t* h, size_t elems, bool late = false) -> Fe* {
        if (!h) {
            bad = true;
            return nullptr;
        }
        if (dev && elems == size) return (Fe*)h;
        if (!dev && elems == size)  // a proving key's constant column the caller pinned (h2hip_columns_pin): already in HBM, fingerprint checked
            if (const Fe* hit = pinned_column_lookup(c, h, elems)) return (Fe*)hit;
        Fe* p = (Fe*)ar.take(col_bytes);
        if (!p) bad = true;
        else if (!dev) uploads.push_back({p, h, elems, late});
        return p;
    };
    std::vector<const Fe*> fixed(d->n_fixed), advice(d->n_advice), instance(d->n_instance);
    for (uint32_t i = 0; i < d->n_fixed; i++) fixed[i] = place(d->fixed_cosets[i], size);
    const size_t n_polys = (size_t)d->n_advice + d->n_instance;
    std::vector<Fe*> poly_dst(n_polys);
    std::vector<const Fe*> poly_src(n_polys, nullptr);
    for (size_t i = 0; i < n_polys; i++) {
        const uint64_t* h = i < d->n_advice ? d->advice_polys[i] : d->instance_polys[i - d->n_advice];
        Fe* p = (Fe*)ar.take(col_bytes);
        if (!p || !h) bad = true;
        if (dev) poly_src[i] = (const Fe*)h;  // the first NTT pass reads the coefficients where they lie
        else uploads.push_back({p, h, n, false});
        poly_dst[i] = p;
        (i < d->n_advice ? advice[i] : instance[i - d->n_advice]) = p;
    }
    Fe* const l0 = place(d->l0, size);
    Fe* const l_last = place(d->l_last, size);
    Fe* const l_active = place(d->l_active_row, size);
    Fe* const d_values = place(values, size);
    std::vector<const Fe*> z(d->n_perm_sets), pcols(d->n_perm_sets ? d->n_perm_columns : 0), pcosets(d->n_perm_sets ? d->n_perm_columns : 0);
    if (d->n_perm_sets) {
        for (uint32_t i = 0; i < d->n_perm_sets; i++) z[i] = place(d->perm_product_cosets[i], size, true);
        for (uint32_t j = 0; j < d->n_perm_columns; j++) {
            pcosets[j] = place(d->perm_cosets[j], size, true);
            const uint32_t kind = d->perm_column_kind[j], idx = d->perm_column_index[j];
            pcols[j] = kind == H2HIP_ANY_ADVICE ? advice[idx] : kind == H2HIP_ANY_FIXED ? fixed[idx] : instance[idx];  // :404-408
        }
    }
    std::vector<Fe*> lbuf(3 * lk_group, nullptr);
    for (size_t t = 0; t < lbuf.size(); t++)
        if (!(lbuf[t] = (Fe*)ar.take(col_bytes))) bad = true;
    for (uint32_t i = 0; i < d->n_lookups; i++)
        if (!d->lookup_product_polys[i] || !d->lookup_permuted_input_polys[i] || !d->lookup_permuted_table_polys[i]) bad = true;
    auto lookup_poly = [&](uint32_t i, int t) -> const uint64_t* {
        return t == 0 ? d->lookup_product_polys[i] : t == 1 ? d->lookup_permuted_input_polys[i] : d->lookup_permuted_table_polys[i];
    };
    if (!bad)
        for (size_t i = 0; i < lk_group; i++)  // the first group rides in the main batch
            for (int t = 0; t < 3; t++) {
                poly_dst.push_back(lbuf[3 * i + t]);
                poly_src.push_back(dev ? (const Fe*)lookup_poly((uint32_t)i, t) : nullptr);
                if (!dev) uploads.push_back({lbuf[3 * i + t], lookup_poly((uint32_t)i, t), n, false});
            }
    if (bad) {
        set_error("evaluate_h: null column or arena overflow");
        return 1;
    }
    // ---- metadata image
    ColsDev cols;
    cols.fixed = mb.put(fixed.data(), fixed.size());
    cols.advice = mb.put(advice.data(), advice.size());
    cols.instance = mb.put(instance.data(), instance.size());
    {
        std::vector<Fu> hch(d->n_challenges);
        for (uint32_t i = 0; i < d->n_challenges; i++) hch[i] = to_i(load_fe(d->challenges + 4 * (size_t)i));
        cols.challenges = mb.put(hch.data(), hch.size());
    }
    cols.beta = to_i(load_fe(d->beta));
    cols.gamma = to_i(load_fe(d->gamma));
    cols.theta = to_i(load_fe(d->theta));
    cols.y = to_i(load_fe(d->y));
    cols.log_size = ek;
    cols.rot_scale = 1 << (ek - k);
    const Fe ext_omega = load_fe(d->extended_omega);
    ProgDev gd;
    prog_put(mb, d->custom_gates, gates_prog, &gd);
    std::vector<ProgDev> lgs(d->n_lookups);
    for (uint32_t i = 0; i < d->n_lookups; i++) prog_put(mb, d->lookup_graphs[i], lookup_progs[i], &lgs[i]);
    PermDev pd;
    memset(&pd, 0, sizeof(pd));
    if (d->n_perm_sets) {
        pd.z = mb.put(z.data(), z.size());
        pd.cols = mb.put(pcols.data(), pcols.size());
        pd.cosets = mb.put(pcosets.data(), pcosets.size());
        if ((rc = ntt_power_table(c, ext_omega, ek, s, &pd.pow_lo, &pd.pow_hi, &pd.pow_bits))) return rc;
        pd.l0 = l0;
        pd.l_last = l_last;
        pd.l_active = l_active;
        pd.delta = to_i(load_fe(d->delta));
        pd.delta_start = to_i(fe_mul<FrP>(load_fe(d->beta), load_fe(d->zeta)));
        pd.n_sets = d->n_perm_sets;
        pd.n_cols = d->n_perm_columns;
        pd.chunk_len = d->chunk_len;
        pd.last_rotation = d->last_rotation;
    }
    if (mb.overflow) {
        set_error("evaluate_h: metadata region overflow");
        return 1;
    }
    if (!mb.host.empty()) {  // through the pinned ring (or a synchronous copy when large): mb.host dies with this frame
        int rc_up = c->stage_h2d(mb.dev_base, mb.host.data(), mb.host.size(), s);
        if (rc_up) return rc_up;
    }
    // ---- columns: host -> device copies (host-pointer form), then advice / instance polynomials -> extended cosets
    //      (:306-323) in one batched transform: distribute_powers_zeta(into_coset) + zero-pad + NTT, as h2hip_coeff_to_extended does
    // (the permutation argument's own columns -- the product cosets of this proof, and the key's permutation cosets when they are not
    // pinned -- cross later, on a second stream under the coset transforms and the gates kernel: a copy from pageable memory blocks the
    // calling thread, not the kernels already queued)
    bool any_late = false;
    for (const Upload& u : uploads) {
        if (u.late) any_late = true;
        else H2_CHECK(hipMemcpyAsync(u.dst, u.src, u.elems * sizeof(Fe), hipMemcpyHostToDevice, s));
    }
    if (any_late && (rc = c->ensure_aux(2))) return rc;
    NttScale sc;
    sc.in_scale = true;
    sc.in3[0] = fe_one<FrP>();
    sc.in3[1] = load_fe(d->g_coset);
    sc.in3[2] = load_fe(d->g_coset_inv);
    sc.in_len = n;
    int t_c = c->timer_begin("evalh_cosets", s);
    if ((rc = ntt_device_batch(c, poly_dst.data(), poly_src.data(), poly_dst.size(), ext_omega, ek, &sc, s))) return rc;
    c->timer_end(t_c, s);
    const dim3 grid((uint32_t)((size + 255) / 256)), block(256);

    // ---- custom gates (:334-360): the kernel generated for this circuit once it is compiled, the interpreter until then
    int t_g = c->timer_begin("evalh_gates", s);
    hipFunction_t gen = gates_plan.tier != 0 ? gates_kernel_for(c, d->custom_gates, gates_prog) : nullptr;
    if (gen) {
        Fe* vals = d_values;
        void* args[] = {(void*)&cols, (void*)&vals};
        H2_CHECK(hipModuleLaunchKernel(gen, (uint32_t)((size + 255) / 256), 1, 1, 256, 1, 1, 0, s, args, nullptr));
        g_rtc_stats.launches++;
    } else {
        g_rtc_stats.interpreted++;
        const dim3 g(gates_plan.lanes / 256);
        switch (gates_plan.tier) {
            case 4: hipLaunchKernelGGL(evalh_gates_kernel<4>, g, block, 4 * 9 * 256 * 4, s, gd, cols, d_values, gws, gates_plan.lanes); break;
            case 8: hipLaunchKernelGGL(evalh_gates_kernel<8>, g, block, 8 * 9 * 256 * 4, s, gd, cols, d_values, gws, gates_plan.lanes); break;
            case 16: hipLaunchKernelGGL(evalh_gates_kernel<16>, g, block, EVALH_LDS_HOT * 9 * 256 * 4, s, gd, cols, d_values, gws, gates_plan.lanes); break;
            case 64: hipLaunchKernelGGL(evalh_gates_kernel<64>, g, block, EVALH_LDS_HOT * 9 * 256 * 4, s, gd, cols, d_values, gws, gates_plan.lanes); break;
            case 256: hipLaunchKernelGGL(evalh_gates_kernel<256>, g, block, EVALH_LDS_HOT * 9 * 256 * 4, s, gd, cols, d_values, gws, gates_plan.lanes); break;
            default: hipLaunchKernelGGL(evalh_gates_kernel<0>, g, block, 0, s, gd, cols, d_values, gws, gates_plan.lanes);
        }
    }
    H2_CHECK(hipGetLastError());
    c->timer_end(t_g, s);

    // ---- permutations (:362-441)
    if (any_late) {
        for (const Upload& u : uploads)
            if (u.late) H2_CHECK(hipMemcpyAsync(u.dst, u.src, u.elems * sizeof(Fe), hipMemcpyHostToDevice, c->aux2));
        H2_CHECK(hipEventRecord(c->aux_events[0], c->aux2));
        H2_CHECK(hipStreamWaitEvent(s, c->aux_events[0], 0));
    }
    if (d->n_perm_sets) {
        int t_p = c->timer_begin("evalh_perm", s);
        hipLaunchKernelGGL(evalh_perm_kernel, grid, block, 0, s, pd, cols, d_values);
        H2_CHECK(hipGetLastError());
        c->timer_end(t_p, s);
    }

    // ---- lookups (:443-518): the three cosets of a lookup are formed in groups (the first group with the columns above), used, and
    //      the group's buffers reused
    int t_l = d->n_lookups ? c->timer_begin("evalh_lookups", s) : -1;
    for (uint32_t i = 0; i < d->n_lookups; i++) {
        const size_t gi = i % lk_group;
        if (gi == 0 && i) {  // the next group's cosets
            const size_t cnt = d->n_lookups - i < lk_group ? d->n_lookups - i : lk_group;
            std::vector<const Fe*> lsrc(3 * cnt, nullptr);
            for (size_t q = 0; q < 3 * cnt; q++) {
                const uint64_t* h = lookup_poly(i + (uint32_t)(q / 3), (int)(q % 3));
                if (dev) lsrc[q] = (const Fe*)h;
                else H2_CHECK(hipMemcpyAsync(lbuf[q], h, n * sizeof(Fe), hipMemcpyHostToDevice, s));
            }
            if ((rc = ntt_device_batch(c, lbuf.data(), lsrc.data(), 3 * cnt, ext_omega, ek, &sc, s))) return rc;
        }
        const ProgDev& lg = lgs[i];
        LookupDev ld = {lbuf[3 * gi], lbuf[3 * gi + 1], lbuf[3 * gi + 2], l0, l_last, l_active};
        const SlotPlan& lp = lookup_plans[i];
        if (hipFunction_t lgen = gates_kernel_for(c, d->lookup_graphs[i], lookup_progs[i], true)) {  // generated for this circuit, as the gates kernel is
            Fe* vals = d_values;
            void* args[] = {(void*)&cols, (void*)&ld, (void*)&vals};
            H2_CHECK(hipModuleLaunchKernel(lgen, (uint32_t)((size + 255) / 256), 1, 1, 256, 1, 1, 0, s, args, nullptr));
            g_rtc_stats.launches++;
            continue;
        }
        g_rtc_stats.interpreted++;
        const dim3 g(lp.lanes / 256);
        switch (lp.tier) {
            case 4: hipLaunchKernelGGL(evalh_lookup_kernel<4>, g, block, 4 * 9 * 256 * 4, s, lg, ld, cols, d_values, gws, lp.lanes); break;
            case 8: hipLaunchKernelGGL(evalh_lookup_kernel<8>, g, block, 8 * 9 * 256 * 4, s, lg, ld, cols, d_values, gws, lp.lanes); break;
            case 16: hipLaunchKernelGGL(evalh_lookup_kernel<16>, g, block, EVALH_LDS_HOT * 9 * 256 * 4, s, lg, ld, cols, d_values, gws, lp.lanes); break;
            case 64: hipLaunchKernelGGL(evalh_lookup_kernel<64>, g, block, EVALH_LDS_HOT * 9 * 256 * 4, s, lg, ld, cols, d_values, gws, lp.lanes); break;
            case 256: hipLaunchKernelGGL(evalh_lookup_kernel<256>, g, block, EVALH_LDS_HOT * 9 * 256 * 4, s, lg, ld, cols, d_values, gws, lp.lanes); break;
            default: hipLaunchKernelGGL(evalh_lookup_kernel<0>, g, block, 0, s, lg, ld, cols, d_values, gws, lp.lanes);
        }
        H2_CHECK(hipGetLastError());
    }
    c->timer_end(t_l, s);
    if ((rc = guard.release())) return rc;
    if (dev) return 0;
    H2_CHECK(hipMemcpyAsync(values, d_values, col_bytes, hipMemcpyDeviceToHost, s));
    H2_CHECK(hipStreamSynchronize(s));
    return 0;
}

}  // namespace h2
