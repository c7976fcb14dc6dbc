// tools/instr_rate.hip -- VALU issue-rate microbenchmark for gfx950 (MI355X).
// Measures wave64 instructions/clk/SIMD for the integer and FP64 ops a 256-bit Montgomery
// multiplier can be built from, so that the limb representation is chosen from measurements.
// Build: hipcc --offload-arch=gfx950 -O3 tools/instr_rate.hip -o tools/instr_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum { MAD64, MULLO, MULHI, MAD24, MULHI24, LSHLADD64, ADDCO, ADDC, FMA64, ADD64F, FMA32, MUL24, ADD3, CNDMASK, N_OPS };
static const char* NAMES[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_mul_hi_u32_u24",
                              "v_lshl_add_u64", "v_add_co_u32", "v_addc_co_u32", "v_fma_f64", "v_add_f64", "v_fma_f32",
                              "v_mul_u32_u24", "v_add3_u32", "v_cndmask_b32"};

template <int OP>
__global__ void __launch_bounds__(256) bench(uint64_t* out, int iters, uint32_t seed) {
    uint64_t r[8];
    uint32_t a = threadIdx.x * 2654435761u + seed, b = (threadIdx.x ^ seed) * 40503u + 12345u;
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = ((uint64_t)(a + i) << 20) ^ b;
    double da = 1.0 + 1e-9 * threadIdx.x, db = 1.0 - 1e-9 * threadIdx.x;
    for (int it = 0; it < iters; it++) {
#define X(i)                                                                                                         \
    if (OP == MAD64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b) : "vcc");        \
    if (OP == MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(*(uint32_t*)&r[i]) : "v"(a));                     \
    if (OP == MULHI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(*(uint32_t*)&r[i]) : "v"(a));                     \
    if (OP == MAD24) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(*(uint32_t*)&r[i]) : "v"(a), "v"(b));        \
    if (OP == MULHI24) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(*(uint32_t*)&r[i]) : "v"(a));               \
    if (OP == LSHLADD64) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 7]));            \
    if (OP == ADDCO) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(*(uint32_t*)&r[i]) : "v"(a) : "vcc");        \
    if (OP == ADDC) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(*(uint32_t*)&r[i]) : "v"(a) : "vcc");   \
    if (OP == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(*(double*)&r[i]) : "v"(da), "v"(db));            \
    if (OP == ADD64F) asm volatile("v_add_f64 %0, %0, %1" : "+v"(*(double*)&r[i]) : "v"(da));                        \
    if (OP == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(*(float*)&r[i]) : "v"(a), "v"(b));               \
    if (OP == MUL24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(*(uint32_t*)&r[i]) : "v"(a));                    \
    if (OP == ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(*(uint32_t*)&r[i]) : "v"(a), "v"(b));            \
    if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(*(uint32_t*)&r[i]) : "v"(a) : "vcc");
        REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s ^= r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
static void run(uint64_t* d_out, int blocks, int iters, double clk_ghz, int cus) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    bench<OP><<<blocks, 256>>>(d_out, 16, 1);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        bench<OP><<<blocks, 256>>>(d_out, iters, rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double wave_instr = (double)blocks * 4 * (double)iters * 32;  // 4 waves/block, 32 instr/iter
    double per_s = wave_instr / (best * 1e-3);
    double per_clk_simd = per_s / (clk_ghz * 1e9) / (cus * 4);
    printf("%-18s %8.3f ms  %.3e wave-instr/s  %.4f wave-instr/clk/SIMD @%.2fGHz (=> %.2f cyc per wave64 instr)\n",
           NAMES[OP], best, per_s, per_clk_simd, clk_ghz, 1.0 / per_clk_simd);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    double clk = prop.clockRate * 1e-6;
    printf("device %s CUs=%d clock=%.2f GHz\n", prop.name, cus, clk);
    int blocks = cus * 8;  // 8 blocks x 4 waves = 32 waves/CU = 8 waves/SIMD
    uint64_t* d_out;
    hipMalloc(&d_out, (size_t)blocks * 256 * 8);
    int iters = 4000;
    run<FMA32>(d_out, blocks, iters, clk, cus);
    run<MAD64>(d_out, blocks, iters, clk, cus);
    run<MULLO>(d_out, blocks, iters, clk, cus);
    run<MULHI>(d_out, blocks, iters, clk, cus);
    run<MAD24>(d_out, blocks, iters, clk, cus);
    run<MULHI24>(d_out, blocks, iters, clk, cus);
    run<MUL24>(d_out, blocks, iters, clk, cus);
    run<LSHLADD64>(d_out, blocks, iters, clk, cus);
    run<ADDCO>(d_out, blocks, iters, clk, cus);
    run<ADDC>(d_out, blocks, iters, clk, cus);
    run<ADD3>(d_out, blocks, iters, clk, cus);
    run<CNDMASK>(d_out, blocks, iters, clk, cus);
    run<FMA64>(d_out, blocks, iters, clk, cus);
    run<ADD64F>(d_out, blocks, iters, clk, cus);
    hipFree(d_out);
    return 0;
}
