// tools/mul_rate.hip -- throughput/latency of fu_mul variants on gfx950 at 1, 2, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../halo2-pse_amd/csrc/fieldu.h"
using namespace h2;

// variant 1: two accumulators per column (a*b chain and m*p chain)
template <class U>
__device__ __forceinline__ Fu fu_mul2(const Fu& a, const Fu& b) {
    int64_t carry = 0;
    uint32_t m[9];
    Fu r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        int64_t s0 = carry, s1 = 0;
#pragma unroll
        for (int i = 0; i <= k; i++) s0 += (int64_t)a.l[i] * (int64_t)b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) s1 += (int64_t)m[i] * (int64_t)U::P[k - i];
        int64_t acc = s0 + s1;
        m[k] = ((uint32_t)acc * U::INV) & H2_MASK29;
        acc += (int64_t)m[k] * (int64_t)U::P[0];
        carry = acc >> 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
        int64_t s0 = carry, s1 = 0;
#pragma unroll
        for (int i = k - 8; i <= 8; i++) s0 += (int64_t)a.l[i] * (int64_t)b.l[k - i];
#pragma unroll
        for (int i = k - 8; i <= 8; i++) s1 += (int64_t)m[i] * (int64_t)U::P[k - i];
        int64_t acc = s0 + s1;
        r.l[k - 9] = (int32_t)((uint32_t)acc & H2_MASK29);
        carry = acc >> 29;
    }
    r.l[8] = (int32_t)carry;
    return r;
}

// variant 2: full product first (17 independent column sums), then reduction
template <class U>
__device__ __forceinline__ Fu fu_mul3(const Fu& a, const Fu& b) {
    int64_t col[17];
#pragma unroll
    for (int k = 0; k < 17; k++) {
        int64_t s = 0;
#pragma unroll
        for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); i++) s += (int64_t)a.l[i] * (int64_t)b.l[k - i];
        col[k] = s;
    }
    uint32_t m[9];
    int64_t carry = 0;
    Fu r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        int64_t s1 = carry;
#pragma unroll
        for (int i = 0; i < k; i++) s1 += (int64_t)m[i] * (int64_t)U::P[k - i];
        int64_t acc = col[k] + s1;
        m[k] = ((uint32_t)acc * U::INV) & H2_MASK29;
        acc += (int64_t)m[k] * (int64_t)U::P[0];
        carry = acc >> 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
        int64_t s1 = carry;
#pragma unroll
        for (int i = k - 8; i <= 8; i++) s1 += (int64_t)m[i] * (int64_t)U::P[k - i];
        int64_t acc = col[k] + s1;
        r.l[k - 9] = (int32_t)((uint32_t)acc & H2_MASK29);
        carry = acc >> 29;
    }
    r.l[8] = (int32_t)carry;
    return r;
}

// variant 3: column order, explicit v_mad instructions (one accumulator), nothing else left to the compiler
__device__ __forceinline__ void mad_ss(int64_t& acc, int32_t a, int32_t b) { uint64_t sink; asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(sink) : "v"(a), "v"(b)); }
__device__ __forceinline__ void mad_us(int64_t& acc, uint32_t a, uint32_t b) { uint64_t sink; asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(sink) : "v"(a), "s"(b)); }
template <class U>
__device__ __forceinline__ Fu fu_mul4(const Fu& a, const Fu& b) {
    int64_t acc = 0;
    uint32_t m[9];
    Fu r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) mad_ss(acc, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = 0; i < k; i++) mad_us(acc, m[i], U::P[k - i]);
        m[k] = ((uint32_t)acc * U::INV) & H2_MASK29;
        mad_us(acc, m[k], U::P[0]);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i <= 8; i++) mad_ss(acc, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = k - 8; i <= 8; i++) mad_us(acc, m[i], U::P[k - i]);
        r.l[k - 9] = (int32_t)((uint32_t)acc & H2_MASK29);
        acc >>= 29;
    }
    r.l[8] = (int32_t)acc;
    return r;
}

// variant 5: plain C in column order with an empty-asm barrier on the accumulator after every column, which stops
// LLVM from reassociating the 162 products into its row-wise multi-accumulator form
template <class U>
__device__ __forceinline__ Fu fu_mul6(const Fu& a, const Fu& b) {
    int64_t acc = 0;
    uint32_t m[9];
    Fu r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (int64_t)a.l[i] * (int64_t)b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (int64_t)m[i] * (int64_t)U::P[k - i];
        asm("" : "+v"(acc));
        m[k] = ((uint32_t)acc * U::INV) & H2_MASK29;
        acc += (int64_t)m[k] * (int64_t)U::P[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i <= 8; i++) acc += (int64_t)a.l[i] * (int64_t)b.l[k - i];
#pragma unroll
        for (int i = k - 8; i <= 8; i++) acc += (int64_t)m[i] * (int64_t)U::P[k - i];
        asm("" : "+v"(acc));
        r.l[k - 9] = (int32_t)((uint32_t)acc & H2_MASK29);
        acc >>= 29;
    }
    r.l[8] = (int32_t)acc;
    return r;
}

// variant 4: as 3 with two accumulators per column (a*b chain, m*p chain)
template <class U>
__device__ __forceinline__ Fu fu_mul5(const Fu& a, const Fu& b) {
    int64_t carry = 0;
    uint32_t m[9];
    Fu r;
#pragma unroll
    for (int k = 0; k < 17; k++) {
        int64_t s0 = carry, s1 = 0;
        const int lo = k > 8 ? k - 8 : 0, hi = k < 8 ? k : 8;
#pragma unroll
        for (int i = lo; i <= hi; i++) mad_ss(s0, a.l[i], b.l[k - i]);
        if (k < 9) {
#pragma unroll
            for (int i = 0; i < k; i++) mad_us(s1, m[i], U::P[k - i]);
            int64_t acc = s0 + s1;
            m[k] = ((uint32_t)acc * U::INV) & H2_MASK29;
            mad_us(acc, m[k], U::P[0]);
            carry = acc >> 29;
        } else {
#pragma unroll
            for (int i = k - 8; i <= 8; i++) mad_us(s1, m[i], U::P[k - i]);
            int64_t acc = s0 + s1;
            r.l[k - 9] = (int32_t)((uint32_t)acc & H2_MASK29);
            carry = acc >> 29;
        }
    }
    r.l[8] = (int32_t)carry;
    return r;
}

template <int V>
__global__ void __launch_bounds__(256) k(Fu* a, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    Fu x = a[i], y = a[i + 1];
    for (int it = 0; it < iters; it++) {
        if (V == 0) x = fu_mul<FqU>(x, y);
        if (V == 1) x = fu_mul2<FqU>(x, y);
        if (V == 2) x = fu_mul3<FqU>(x, y);
        if (V == 3) x = fu_mul4<FqU>(x, y);
        if (V == 4) x = fu_mul5<FqU>(x, y);
        if (V == 5) x = fu_mul6<FqU>(x, y);
        if (V == 6) x = fu_mul<FqUA>(x, y);  // the library's explicit-mad flavour: one asm statement per column and operand kind
    }
    a[i] = x;
}

template <int V>
void run(Fu* d, int waves_per_simd, int cus) {
    int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = 1 wave per SIMD per block
    int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<V><<<blocks, 256>>>(d, 10);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(e0);
        k<V><<<blocks, 256>>>(d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double muls = (double)blocks * 256 * iters;
    printf("variant %d  waves/SIMD %d: %.3f ms  %.2f Gmul/s  latency per mul (one wave) %.0f ns\n", V, waves_per_simd, best, muls / best / 1e6,
           best * 1e6 / iters);
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    Fu* d; hipMalloc(&d, (size_t)cus * 8 * 256 * sizeof(Fu) + 64); hipMemset(d, 1, (size_t)cus * 8 * 256 * sizeof(Fu) + 64);
    // measured: lone wave 468 ns (variant 0), 669 (3: an s_nop after every asm statement), 467 (6); 8 waves per SIMD 167 / 177 / 183 G multiplies/s.
    // Alternating two accumulators inside the column statements (tried, not kept) is slower everywhere: a lone wave is bound by issue
    // (4.8 cycles per instruction), not by the accumulator dependency.
    for (int w : {1, 2, 4, 8}) { run<0>(d, w, cus); run<3>(d, w, cus); run<6>(d, w, cus); }
    return 0;
}
