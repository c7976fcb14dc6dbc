// gen.hip -- synthetic workload generator on the GPU: counter-based SplitMix64
// scalars (uniform mod r) and G1 points (try-and-increment on y^2 = x^3 + 3), bit-identical to
// oracle_gen_scalars / oracle_gen_points so the CPU baseline and the parity tests see the same
// inputs the bench generates in HBM.  The reference has no counterpart (its benches draw from
// OsRng: poly/kzg/commitment.rs:365).
#include "engine.h"

namespace h2 {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

// 256-bit draw for element i: limb j = splitmix64(seed + 4i + j + attempt*K); top two bits
// cleared; one conditional subtract brings it below the 254-bit modulus
template <class P>
__device__ __forceinline__ Fe draw_mod(uint64_t seed, uint64_t i, uint64_t attempt, uint64_t* raw3) {
    Fe c;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint64_t v = splitmix64(seed + 4 * i + (uint64_t)j + attempt * 0x632BE59BD9B4E019ULL);
        if (j == 3) {
            *raw3 = v;
            v &= 0x3FFFFFFFFFFFFFFFULL;
        }
        c.l[2 * j] = (uint32_t)v;
        c.l[2 * j + 1] = (uint32_t)(v >> 32);
    }
    if (!fe_is_canonical<P>(c)) {
        uint32_t br = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint64_t d = (uint64_t)c.l[j] - P::MOD[j] - br;
            c.l[j] = (uint32_t)d;
            br = (uint32_t)(d >> 63);
        }
    }
    return c;
}

__global__ void gen_scalars_kernel(uint64_t seed, uint64_t start, uint64_t n, Fe* out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t raw3;
    Fe c = draw_mod<FrP>(seed, start + i, 0, &raw3);
    out[i] = fe_from_canonical<FrP>(c);
}

__global__ void gen_points_kernel(uint64_t seed, uint64_t start, uint64_t n, Affine* out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t e[8] = {0xb61f3f52u, 0x4f082305u, 0x5a1c72a3u, 0x65e05aa4u,
                           0xa0605617u, 0x6e14116du, 0xb84c680au, 0x0c19139cu};  // (q+1)/4
    const Fe b3 = fe_from_u64<Q>(3);
    for (uint64_t attempt = 0; attempt < 256; attempt++) {  // P(fail) = 2^-256; bounded so every lane exits
        uint64_t raw3;
        Fe x = fe_from_canonical<Q>(draw_mod<Q>(seed, start + i, attempt, &raw3));
        Fe rhs = fe_add<Q>(fe_mul<Q>(fe_sqr<Q>(x), x), b3);
        Fe y = fe_pow<Q>(rhs, e);
        if (!fe_eq(fe_sqr<Q>(y), rhs)) continue;
        if ((raw3 >> 62) & 1) y = fe_neg<Q>(y);
        Affine p;
        p.x = x;
        p.y = y;
        out[i] = p;
        return;
    }
    Affine id;
    id.x = fe_zero<Q>();
    id.y = fe_zero<Q>();
    out[i] = id;
}

int gen_scalars_device(uint64_t seed, uint64_t start, size_t n, Fe* d_out, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(gen_scalars_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, seed, start, (uint64_t)n, d_out);
    H2_CHECK(hipGetLastError());
    return 0;
}

int gen_points_device(uint64_t seed, uint64_t start, size_t n, Affine* d_out, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(gen_points_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, seed, start, (uint64_t)n, d_out);
    H2_CHECK(hipGetLastError());
    return 0;
}

}  // namespace h2
