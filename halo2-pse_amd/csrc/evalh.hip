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

#include <algorithm>
#include <vector>

#include "../../include/halo2hip.h"
#include "engine.h"

namespace h2 {

#define EVALH_VS_ZERO 11  // internal operand kind: the field's zero (an empty graph's value, evaluation.rs:745-749)

// The flattened graph is compiled on the host (compile_graph below) into a short register-machine program before it is
// run: Store calculations become direct column operands, a Horner calculation becomes one FMA per part placed as soon
// as that part exists, dead calculations are dropped and the surviving intermediates are packed into as few slots as
// their lifetimes allow.  A slot is 36 B of LDS (up to 8 slots), per-lane scratch (up to 256) or a row of a global workspace, so a
// circuit with tens of thousands of calculations still runs with a few dozen slots per lane.  The field operations
// performed per row are the reference's, operation for operation; only where a value waits between them differs.
//
// Arithmetic: the unsaturated 9 x 29-bit multiplier of fieldu.cuh (the MSM's and the NTT's), about half the
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

typedef FrUA UF;

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

struct ColsDev {
    const Fe* const* fixed;
    const Fe* const* advice;
    const Fe* const* instance;
    const Fu* challenges;  // I-form, canonical
    Fu beta, gamma, theta, y;
    uint32_t log_size;
    int32_t rot_scale;
};

// Wave-uniform reads of data that no kernel writes (the program, its constants and rotations, the column pointer
// tables): read through the constant address space so that they are scalar loads.  Through a generic pointer the
// compiler must assume the kernel's own stores may alias them and issues one vector load per lane instead (measured:
// 13 scalar against 1 900 vector memory instructions per wave in the gates kernel).
#define H2_CONST_AS __attribute__((address_space(4)))
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
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
__device__ __forceinline__ Fu ld_const_fu(const Fu* tab, uint32_t i) {
    const H2_CONST_AS int32_t* p = (const H2_CONST_AS int32_t*)(uintptr_t)tab;
    Fu o;
#pragma unroll
    for (int k = 0; k < 9; k++) o.l[k] = p[9 * (size_t)i + k];
    return o;
}
__device__ __forceinline__ int32_t ld_const_i32(const int32_t* tab, uint32_t i) { return ((const H2_CONST_AS int32_t*)(uintptr_t)tab)[i]; }
__device__ __forceinline__ const Fe* ld_const_col(const Fe* const* tab, uint32_t i) {
    return (const Fe*)(uintptr_t)((const H2_CONST_AS uint64_t*)(uintptr_t)tab)[i];
}

__device__ __forceinline__ Fu ld_i(const Fe& x) { return fu_from_ext(x); }                    // E canonical -> I, < 32 r
__device__ __forceinline__ Fu addn(const Fu& a, const Fu& b) { return fu_norm(fu_add(a, b)); }
__device__ __forceinline__ Fu subn(const Fu& a, const Fu& b) { return fu_norm(fu_sub(a, b)); }
__device__ __forceinline__ Fu mul_i(const Fu& a, const Fu& b) { return fu_mul<UF>(a, b); }
__device__ __forceinline__ Fe out_e(const Fu& a) { return fu_mul_canon<UF>(a, fu_one_e<UF>()); }  // |a| < 16 r -> canonical E

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

// get_rotation_idx (evaluation.rs:32-34): size is a power of two, so rem_euclid is a mask
__device__ __forceinline__ uint32_t rot_idx(uint32_t idx, int32_t rot, int32_t rot_scale, uint32_t log_size) {
    return (uint32_t)((int32_t)idx + rot * rot_scale) & ((1u << log_size) - 1);
}

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
    v = addn(mul_i(v, c.y), mul_i(subn(one, ld_i(ld_const_col(p.z, 0)[idx])), ld_i(p.l0[idx])));  // [1.2] + [33 * 32 / 169 + 1 = 7.3] = [8.5]
    // l_last(X) * (z_l(X)^2 - z_l(X)) = 0                                         :387-393
    {
        const Fu zl = ld_i(ld_const_col(p.z, p.n_sets - 1)[idx]);                                          // [32]
        v = addn(mul_i(v, c.y), mul_i(subn(fu_sqr<UF>(zl), zl), ld_i(p.l_last[idx])));      // [1.1] + [(7.1 + 32) * 32 / 169 + 1 = 8.4] = [9.5]
    }
    // l_0(X) * (z_i(X) - z_{i-1}(omega^(last) X)) = 0                              :394-404
    for (uint32_t s = 1; s < p.n_sets; s++)
        v = addn(mul_i(v, c.y), mul_i(subn(ld_i(ld_const_col(p.z, s)[idx]), ld_i(ld_const_col(p.z, s - 1)[r_last])), ld_i(p.l0[idx])));  // [1.1] + [64 * 32 / 169 + 1 = 13.2] = [14.3]
    // (1 - (l_last + l_blind)) * (z_i(wX) prod(p + beta s_j + gamma) - z_i(X) prod(p + delta^j beta X + gamma))   :405-438
    Fu current_delta = p.delta_start;  // beta * ZETA * extended_omega^idx (beta_term, :366-368 and :412)   [1 .. 2]
    current_delta = mul_i(current_delta, p.pow_lo[idx & ((1u << p.pow_bits) - 1)]);
    if (idx >> p.pow_bits) current_delta = mul_i(current_delta, p.pow_hi[idx >> p.pow_bits]);  // uniform over a workgroup
    for (uint32_t s = 0; s < p.n_sets; s++) {
        const uint32_t j0 = s * p.chunk_len, j1 = j0 + p.chunk_len < p.n_cols ? j0 + p.chunk_len : p.n_cols;
        const Fe* zs = ld_const_col(p.z, s);
        Fu left = ld_i(zs[r_next]), right = ld_i(zs[idx]);  // [32]
        for (uint32_t j = j0; j < j1; j++) {
            const Fu term = addn(addn(ld_i(ld_const_col(p.cols, j)[idx]), mul_i(c.beta, ld_i(ld_const_col(p.cosets, j)[idx]))), c.gamma);  // [32 + 1.2 + 1 = 34.2]
            left = mul_i(left, term);                                                                          // [32 * 34.2 / 169 + 1 = 7.5], then smaller
        }
        for (uint32_t j = j0; j < j1; j++) {
            const Fu term = addn(addn(ld_i(ld_const_col(p.cols, j)[idx]), current_delta), c.gamma);  // [35]
            right = mul_i(right, term);                                                // [7.7]
            current_delta = mul_i(current_delta, p.delta);                             // [1.1]
        }
        v = addn(mul_i(v, c.y), mul_i(subn(left, right), ld_i(p.l_active[idx])));     // [1.1] + [64 * 32 / 169 + 1 = 13.2] = [14.3]
    }
    values[idx] = out_e(v);  // [< 15]
}

struct LookupDev {
    const Fe *product, *pin, *ptab;  // extended cosets of product / permuted input / permuted table
    const Fe *l0, *l_last, *l_active;
};

template <int MAXI>
__global__ void __launch_bounds__(256) evalh_lookup_kernel(ProgDev g, LookupDev l, ColsDev c, Fe* values, Fu* gws, uint32_t lanes) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= lanes) return;
    Slots<MAXI> slots(gws + tid, lanes);
    const Fu one = fu_one_i<UF>();
    for (uint64_t row = tid; row < (1ull << c.log_size); row += lanes) {
        const uint32_t idx = (uint32_t)row;
        const Fu table_value = prog_eval(g, c, idx, fu_zero(), slots);  // :466-480   [< 32]
        const uint32_t r_next = rot_idx(idx, 1, c.rot_scale, c.log_size), r_prev = rot_idx(idx, -1, c.rot_scale, c.log_size);
        const Fu z = ld_i(l.product[idx]), a_ = ld_i(l.pin[idx]), s_ = ld_i(l.ptab[idx]);  // [32]
        const Fu l0 = ld_i(l.l0[idx]), l_active = ld_i(l.l_active[idx]);                  // [32]
        const Fu a_minus_s = subn(a_, s_);                                                // [64]
        Fu v = ld_i(values[idx]);                                                         // [32]
        // l_0(X) * (1 - z(X)) = 0
        v = addn(mul_i(v, c.y), mul_i(subn(one, z), l0));                                             // [1.2 + 7.3 = 8.5]
        // l_last(X) * (z(X)^2 - z(X)) = 0
        v = addn(mul_i(v, c.y), mul_i(subn(fu_sqr<UF>(z), z), ld_i(l.l_last[idx])));                  // [1.1 + 8.4 = 9.5]
        // (1 - (l_last + l_blind)) * (z(wX)(a' + beta)(s' + gamma) - z(X) * table_value) = 0
        {
            const Fu lhs = mul_i(mul_i(ld_i(l.product[r_next]), addn(a_, c.beta)), addn(s_, c.gamma));  // [32 * 33 / 169 + 1 = 7.3] -> [7.3 * 33 / 169 + 1 = 2.5]
            const Fu rhs = mul_i(z, table_value);                                                       // [32 * 32 / 169 + 1 = 7.1]
            v = addn(mul_i(v, c.y), mul_i(subn(lhs, rhs), l_active));                                   // [1.1] + [9.6 * 32 / 169 + 1 = 2.9] = [4]
        }
        // l_0(X) * (a'(X) - s'(X)) = 0
        v = addn(mul_i(v, c.y), mul_i(a_minus_s, l0));                                                  // [1.1] + [64 * 32 / 169 + 1 = 13.2] = [14.3]
        // (1 - (l_last + l_blind)) * (a' - s') * (a'(X) - a'(w^-1 X)) = 0
        v = addn(mul_i(v, c.y), mul_i(mul_i(a_minus_s, subn(a_, ld_i(l.pin[r_prev]))), l_active));      // [1.1] + [(64 * 64 / 169 + 1 = 25.3) * 32 / 169 + 1 = 5.8] = [6.9]
        values[idx] = out_e(v);
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
    struct Upload { Fe* dst; const uint64_t* src; size_t elems; };
    std::vector<Upload> uploads;  // host -> device column copies, issued after the metadata
    bool bad = false;
    auto place = [&](const uint64_t* h, size_t elems) -> Fe* {
        if (!h) {
            bad = true;
            return nullptr;
        }
        if (dev && elems == size) return (Fe*)h;
        Fe* p = (Fe*)ar.take(col_bytes);
        if (!p) bad = true;
        else if (!dev) uploads.push_back({p, h, elems});
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
        else uploads.push_back({p, h, n});
        poly_dst[i] = p;
        (i < d->n_advice ? advice[i] : instance[i - d->n_advice]) = p;
    }
    Fe* const l0 = place(d->l0, size);
    Fe* const l_last = place(d->l_last, size);
    Fe* const l_active = place(d->l_active_row, size);
    Fe* const d_values = place(values, size);
    std::vector<const Fe*> z(d->n_perm_sets), pcols(d->n_perm_sets ? d->n_perm_columns : 0), pcosets(d->n_perm_sets ? d->n_perm_columns : 0);
    if (d->n_perm_sets) {
        for (uint32_t i = 0; i < d->n_perm_sets; i++) z[i] = place(d->perm_product_cosets[i], size);
        for (uint32_t j = 0; j < d->n_perm_columns; j++) {
            pcosets[j] = place(d->perm_cosets[j], size);
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
                if (!dev) uploads.push_back({lbuf[3 * i + t], lookup_poly((uint32_t)i, t), n});
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
    for (const Upload& u : uploads) H2_CHECK(hipMemcpyAsync(u.dst, u.src, u.elems * sizeof(Fe), hipMemcpyHostToDevice, s));
    NttScale sc;
    sc.in_scale = true;
    sc.in3[0] = fe_one<FrP>();
    sc.in3[1] = load_fe(d->g_coset);
    sc.in3[2] = load_fe(d->g_coset_inv);
    sc.in_len = n;
    if ((rc = ntt_device_batch(c, poly_dst.data(), poly_src.data(), poly_dst.size(), ext_omega, ek, &sc, s))) return rc;
    const dim3 grid((uint32_t)((size + 255) / 256)), block(256);

    // ---- custom gates (:334-360)
    {
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

    // ---- permutations (:362-441)
    if (d->n_perm_sets) {
        hipLaunchKernelGGL(evalh_perm_kernel, grid, block, 0, s, pd, cols, d_values);
        H2_CHECK(hipGetLastError());
    }

    // ---- lookups (:443-518): the three cosets of a lookup are formed in groups (the first group with the columns above), used, and
    //      the group's buffers reused
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
    if ((rc = guard.release())) return rc;
    if (dev) return 0;
    H2_CHECK(hipMemcpyAsync(values, d_values, col_bytes, hipMemcpyDeviceToHost, s));
    H2_CHECK(hipStreamSynchronize(s));
    return 0;
}

}  // namespace h2
