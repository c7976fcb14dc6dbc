// graph_chain.hip -- does a hipGraph shorten a chain of small dependent kernels (a lone 2^17-pair commit is twelve of them, 4-140 us each)?
// N kernels that each spin for a given time, launched (a) one by one into a stream and (b) as one captured graph; wall time from the first
// launch call to the stream's completion, and the GPU-side span from the first kernel's start to the last one's end (wall_clock64 stamps).
//   hipcc -O2 --offload-arch=gfx950 -o graph_chain graph_chain.hip && ./graph_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <vector>

#define CK(x)                                                       \
    do {                                                            \
        hipError_t e = (x);                                         \
        if (e != hipSuccess) {                                      \
            printf("%s -> %s\n", #x, hipGetErrorString(e));         \
            exit(1);                                                \
        }                                                           \
    } while (0)

__global__ void spin_kernel(unsigned long long ticks, unsigned long long* stamps, int slot) {
    const unsigned long long t0 = wall_clock64();  // 100 MHz
    while (wall_clock64() - t0 < ticks) {
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        stamps[2 * slot] = t0;
        stamps[2 * slot + 1] = wall_clock64();
    }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned long long* d_st;
    CK(hipMalloc(&d_st, 64 * 2 * sizeof(unsigned long long)));
    std::vector<unsigned long long> h_st(128);
    const int reps = 200;
    printf("{\"runs\": [");
    bool first = true;
    for (int n : {4, 12, 24})
        for (int us : {2, 10, 40}) {
            for (int grid : {1, 1024}) {
                const unsigned long long ticks = (unsigned long long)us * 100;
                auto chain = [&] {
                    for (int k = 0; k < n; k++) hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), 0, s, ticks, d_st, k);
                };
                hipGraph_t g;
                hipGraphExec_t ge;
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                chain();
                CK(hipStreamEndCapture(s, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                double res[2][2];
                for (int mode = 0; mode < 2; mode++) {
                    std::vector<double> wall, span;
                    for (int r = 0; r < reps + 10; r++) {
                        CK(hipStreamSynchronize(s));
                        const double t0 = now();
                        if (mode == 0) chain();
                        else CK(hipGraphLaunch(ge, s));
                        CK(hipStreamSynchronize(s));
                        const double t1 = now();
                        CK(hipMemcpy(h_st.data(), d_st, n * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                        if (r >= 10) {
                            wall.push_back((t1 - t0) * 1e6);
                            span.push_back((double)(h_st[2 * (n - 1) + 1] - h_st[0]) / 100.0);
                        }
                    }
                    std::sort(wall.begin(), wall.end());
                    std::sort(span.begin(), span.end());
                    res[mode][0] = wall[wall.size() / 2];
                    res[mode][1] = span[span.size() / 2];
                }
                printf("%s\n  {\"kernels\": %d, \"spin_us\": %d, \"grid\": %d, \"stream\": {\"wall_us\": %.1f, \"gpu_span_us\": %.1f}, \"graph\": {\"wall_us\": %.1f, \"gpu_span_us\": %.1f}, \"sum_of_spins_us\": %d}",
                       first ? "" : ",", n, us, grid, res[0][0], res[0][1], res[1][0], res[1][1], n * us);
                first = false;
                CK(hipGraphExecDestroy(ge));
                CK(hipGraphDestroy(g));
            }
        }
    printf("\n]}\n");
    return 0;
}
