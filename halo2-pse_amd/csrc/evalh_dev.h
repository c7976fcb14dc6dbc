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

// get_rotation_idx (evaluation.rs:32-34): size is a power of two, so rem_euclid is a mask
__device__ __forceinline__ uint32_t rot_idx(uint32_t idx, int32_t rot, int32_t rot_scale, uint32_t log_size) {
    return (uint32_t)((int32_t)idx + rot * rot_scale) & ((1u << log_size) - 1);
}

}  // namespace h2
