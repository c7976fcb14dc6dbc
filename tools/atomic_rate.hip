// atomic_rate.hip -- device-scope atomicAdd throughput on a small table (would a small MSM's sort count its buckets directly?):
// N lanes x A atomics each onto T counters at pseudo-random indices; and the same with returning atomics (cursor reservation).
//   hipcc -O2 --offload-arch=gfx950 -o atomic_rate atomic_rate.hip && ./atomic_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
template <bool RET>
__global__ void atom_kernel(uint32_t* tab, uint32_t mask, uint32_t per_lane, uint32_t* sink) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t k = 0; k < per_lane; k++) {
        const uint32_t i = mix(t * 31u + k) & mask;
        if (RET) acc += atomicAdd(&tab[i], 1u);
        else atomicAdd(&tab[i], 1u);
    }
    if (RET && acc == 0xffffffffu) sink[0] = acc;
}

int main() {
    uint32_t *tab, *sink;
    CK(hipMalloc(&tab, 4u << 20));
    CK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("{\"runs\": [");
    bool first = true;
    for (uint32_t log_t : {13u, 16u, 19u})
        for (uint32_t log_lanes : {17u, 20u})
            for (int ret = 0; ret < 2; ret++) {
                const uint32_t per = 15, lanes = 1u << log_lanes;
                CK(hipMemset(tab, 0, 4u << 20));
                float best = 1e9f;
                for (int r = 0; r < 6; r++) {
                    CK(hipEventRecord(e0));
                    if (ret) hipLaunchKernelGGL(atom_kernel<true>, dim3(lanes / 256), dim3(256), 0, 0, tab, (1u << log_t) - 1, per, sink);
                    else hipLaunchKernelGGL(atom_kernel<false>, dim3(lanes / 256), dim3(256), 0, 0, tab, (1u << log_t) - 1, per, sink);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (r && ms < best) best = ms;
                }
                printf("%s\n  {\"counters\": \"2^%u\", \"lanes\": \"2^%u\", \"atomics\": %u, \"returning\": %d, \"us\": %.1f, \"G_atomics_per_s\": %.2f}", first ? "" : ",", log_t, log_lanes,
                       lanes * per, ret, best * 1e3, lanes * (double)per / best / 1e6);
                first = false;
            }
    printf("\n]}\n");
    return 0;
}
