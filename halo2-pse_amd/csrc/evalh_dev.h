// evalh_dev.h -- what the evaluate_h kernels share with the per-circuit gates kernel that evalh.hip generates and compiles at
// run time (hiprtc): the columns descriptor, the scalar-load helpers and the I-form field helpers.  Device only.  This file, field.h,
// fieldu.h and fieldu_chain.inc are embedded in the library as text (csrc/rtc_headers.inc, made by tools/embed_headers.py) and
// handed to hiprtc as in-memory headers, so the generated kernel is built from the very arithmetic the rest of the engine runs.
#pragma once
#include "fieldu.h"

namespace h2 {

typedef FrUA UF;

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
// One step of the fold with y (evaluation.rs: value * y + constraint) where the constraint is itself a product a * b: v y + a b with ONE
// Montgomery reduction (the two products' columns in one accumulator, fu_mul_sub with -a) instead of two reductions and an addition.
// Operands normalised (limbs below 2^29 in magnitude: addn / subn / mul_i / ld_i results); the magnitude of the result is below
// (|v| |y| + |a| |b|) / 169 + 1, the sum of the two separate bounds minus one.
__device__ __forceinline__ Fu fold_y(const Fu& v, const Fu& y, const Fu& a, const Fu& b) { return fu_mul_sub<UF>(v, y, fu_neg(a), b); }

// get_rotation_idx (evaluation.rs:32-34): size is a power of two, so rem_euclid is a mask
__device__ __forceinline__ uint32_t rot_idx(uint32_t idx, int32_t rot, int32_t rot_scale, uint32_t log_size) {
    return (uint32_t)((int32_t)idx + rot * rot_scale) & ((1u << log_size) - 1);
}

// One lookup argument's share of a row (evaluation.rs:443-518), given the compressed table expression of that row: shared by the
// interpreter's lookup kernel and the one generated per circuit.  Magnitudes in units of r are given in brackets.
struct LookupDev {
    const Fe *product, *pin, *ptab;  // extended cosets of product / permuted input / permuted table
    const Fe *l0, *l_last, *l_active;
};
__device__ __forceinline__ void lookup_row(const LookupDev& l, const ColsDev& c, uint32_t idx, const Fu& table_value, Fe* values) {
    const Fu one = fu_one_i<UF>();
    const uint32_t r_next = rot_idx(idx, 1, c.rot_scale, c.log_size), r_prev = rot_idx(idx, -1, c.rot_scale, c.log_size);
    const Fu z = ld_i(l.product[idx]), a_ = ld_i(l.pin[idx]), s_ = ld_i(l.ptab[idx]);  // [32]
    const Fu l0 = ld_i(l.l0[idx]), l_active = ld_i(l.l_active[idx]);                  // [32]
    const Fu a_minus_s = subn(a_, s_);                                                // [64]
    Fu v = ld_i(values[idx]);                                                         // [32]
    // l_0(X) * (1 - z(X)) = 0
    v = fold_y(v, c.y, subn(one, z), l0);                                                         // [1.2 + 7.3 = 8.5]
    // l_last(X) * (z(X)^2 - z(X)) = 0
    v = fold_y(v, c.y, subn(fu_sqr<UF>(z), z), ld_i(l.l_last[idx]));                              // [1.1 + 8.4 = 9.5]
    // (1 - (l_last + l_blind)) * (z(wX)(a' + beta)(s' + gamma) - z(X) * table_value) = 0
    {
        // the difference of the two products with one reduction: (z(wX)(a' + beta)) (s' + gamma) - z table_value
        const Fu zab = mul_i(ld_i(l.product[r_next]), addn(a_, c.beta));                            // [32 * 33 / 169 + 1 = 7.3]
        const Fu diff = fu_mul_sub<UF>(zab, addn(s_, c.gamma), z, table_value);                     // [(7.3 * 33 + 32 * 32) / 169 + 1 = 8.5]
        v = fold_y(v, c.y, diff, l_active);                                                         // [1.1] + [8.5 * 32 / 169 + 1 = 2.6] = [3.7]
    }
    // l_0(X) * (a'(X) - s'(X)) = 0
    v = fold_y(v, c.y, a_minus_s, l0);                                                              // [1.1] + [64 * 32 / 169 + 1 = 13.2] = [14.3]
    // (1 - (l_last + l_blind)) * (a' - s') * (a'(X) - a'(w^-1 X)) = 0
    v = fold_y(v, c.y, mul_i(a_minus_s, subn(a_, ld_i(l.pin[r_prev]))), l_active);                  // [1.1] + [(64 * 64 / 169 + 1 = 25.3) * 32 / 169 + 1 = 5.8] = [6.9]
    values[idx] = out_e(v);
}

}  // namespace h2
