// pcie_rate.hip -- what the host link gives a caller whose columns are pageable memory (a prover's Vec<F>): hipMemcpyAsync to and
// from pageable and pinned buffers, one direction and both at once, one or two issuing threads per direction, and the cost of
// registering the caller's pages for the duration of a call.  The floor of every host-pointer entry point is read off these.
//   hipcc -O2 --offload-arch=gfx950 -o pcie_rate pcie_rate.hip -lpthread && ./pcie_rate [MiB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <thread>
#include <vector>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess) {                                                 \
            printf("%s -> %s\n", #x, hipGetErrorString(e));                    \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Job {
    void* d;
    void* h;
    size_t bytes;
    bool down;
    int pieces;
};

#include <atomic>
static std::atomic<int> g_ready{0};
static std::atomic<int> g_go{0};

static double run_job(const Job& j, int reps, int n_jobs) {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto once = [&] {
        size_t piece = j.bytes / j.pieces;
        for (int p = 0; p < j.pieces; p++) {
            char* h = (char*)j.h + p * piece;
            char* d = (char*)j.d + p * piece;
            if (j.down)
                CK(hipMemcpyAsync(h, d, piece, hipMemcpyDeviceToHost, s));
            else
                CK(hipMemcpyAsync(d, h, piece, hipMemcpyHostToDevice, s));
        }
        CK(hipStreamSynchronize(s));
    };
    once();  // warm
    g_ready.fetch_add(1);
    while (g_go.load() == 0) {
    }
    double t0 = now();
    for (int r = 0; r < reps; r++) once();
    double t = (now() - t0) / reps;
    CK(hipStreamDestroy(s));
    (void)n_jobs;
    return t;
}

static double timed(std::vector<Job> jobs, int reps) {
    g_ready.store(0);
    g_go.store(0);
    std::vector<double> ts(jobs.size());
    std::vector<std::thread> th;
    for (size_t i = 0; i < jobs.size(); i++) th.emplace_back([&, i] { ts[i] = run_job(jobs[i], reps, (int)jobs.size()); });
    while (g_ready.load() < (int)jobs.size()) {
    }
    g_go.store(1);
    for (auto& t : th) t.join();
    double m = 0;
    for (double t : ts) m = t > m ? t : m;
    return m;
}

int main(int argc, char** argv) {
    size_t mib = argc > 1 ? (size_t)atoi(argv[1]) : 128;
    size_t bytes = mib << 20;
    const int reps = 6;
    void *d0, *d1, *d2, *d3;
    CK(hipMalloc(&d0, bytes));
    CK(hipMalloc(&d1, bytes));
    CK(hipMalloc(&d2, bytes));
    CK(hipMalloc(&d3, bytes));
    void *p0, *p1, *q0, *q1;  // pageable
    p0 = aligned_alloc(4096, bytes);
    p1 = aligned_alloc(4096, bytes);
    q0 = aligned_alloc(4096, bytes);
    q1 = aligned_alloc(4096, bytes);
    memset(p0, 1, bytes);
    memset(p1, 2, bytes);
    memset(q0, 3, bytes);
    memset(q1, 4, bytes);
    void *h0, *h1;  // pinned
    CK(hipHostMalloc(&h0, bytes, hipHostMallocDefault));
    CK(hipHostMalloc(&h1, bytes, hipHostMallocDefault));
    memset(h0, 1, bytes);
    memset(h1, 2, bytes);
    auto gbps = [&](double t, double n) { return n * bytes / t / 1e9; };
    double t;
    printf("{\"mib\": %zu", mib);
    t = timed({{d0, p0, bytes, false, 1}}, reps);
    printf(", \"pageable_h2d\": {\"ms\": %.3f, \"GBps\": %.1f}", t * 1e3, gbps(t, 1));
    t = timed({{d1, p1, bytes, true, 1}}, reps);
    printf(", \"pageable_d2h\": {\"ms\": %.3f, \"GBps\": %.1f}", t * 1e3, gbps(t, 1));
    t = timed({{d0, p0, bytes, false, 1}, {d1, p1, bytes, true, 1}}, reps);
    printf(", \"pageable_both\": {\"ms\": %.3f, \"GBps_each\": %.1f}", t * 1e3, gbps(t, 1));
    t = timed({{d0, p0, bytes, false, 1}, {d2, q0, bytes, false, 1}}, reps);
    printf(", \"pageable_h2d_two_threads\": {\"ms\": %.3f, \"GBps_total\": %.1f}", t * 1e3, gbps(t, 2));
    t = timed({{d1, p1, bytes, true, 1}, {d3, q1, bytes, true, 1}}, reps);
    printf(", \"pageable_d2h_two_threads\": {\"ms\": %.3f, \"GBps_total\": %.1f}", t * 1e3, gbps(t, 2));
    t = timed({{d0, p0, bytes, false, 1}, {d2, q0, bytes, false, 1}, {d1, p1, bytes, true, 1}, {d3, q1, bytes, true, 1}}, reps);
    printf(", \"pageable_both_two_threads_each\": {\"ms\": %.3f, \"GBps_each_direction\": %.1f}", t * 1e3, gbps(t, 2));
    t = timed({{d0, p0, bytes, false, 8}}, reps);
    printf(", \"pageable_h2d_8_pieces\": {\"ms\": %.3f, \"GBps\": %.1f}", t * 1e3, gbps(t, 1));
    t = timed({{d0, h0, bytes, false, 1}}, reps);
    printf(", \"pinned_h2d\": {\"ms\": %.3f, \"GBps\": %.1f}", t * 1e3, gbps(t, 1));
    t = timed({{d1, h1, bytes, true, 1}}, reps);
    printf(", \"pinned_d2h\": {\"ms\": %.3f, \"GBps\": %.1f}", t * 1e3, gbps(t, 1));
    t = timed({{d0, h0, bytes, false, 1}, {d1, h1, bytes, true, 1}}, reps);
    printf(", \"pinned_both\": {\"ms\": %.3f, \"GBps_each\": %.1f}", t * 1e3, gbps(t, 1));
    {  // register the caller's pages, copy, unregister
        double t0 = now();
        for (int r = 0; r < reps; r++) {
            CK(hipHostRegister(p0, bytes, hipHostRegisterDefault));
            CK(hipHostUnregister(p0));
        }
        double treg = (now() - t0) / reps;
        printf(", \"register_unregister_ms\": %.3f", treg * 1e3);
        CK(hipHostRegister(p0, bytes, hipHostRegisterDefault));
        CK(hipHostRegister(p1, bytes, hipHostRegisterDefault));
        t = timed({{d0, p0, bytes, false, 1}}, reps);
        printf(", \"registered_h2d\": {\"ms\": %.3f, \"GBps\": %.1f}", t * 1e3, gbps(t, 1));
        t = timed({{d0, p0, bytes, false, 1}, {d1, p1, bytes, true, 1}}, reps);
        printf(", \"registered_both\": {\"ms\": %.3f, \"GBps_each\": %.1f}", t * 1e3, gbps(t, 1));
        CK(hipHostUnregister(p0));
        CK(hipHostUnregister(p1));
    }
    {  // host threads staging through pinned memory: memcpy rate of T threads
        for (int T : {1, 2, 4, 8}) {
            double t0 = now();
            for (int r = 0; r < reps; r++) {
                std::vector<std::thread> th;
                for (int i = 0; i < T; i++)
                    th.emplace_back([&, i] { memcpy((char*)h0 + i * (bytes / T), (char*)p0 + i * (bytes / T), bytes / T); });
                for (auto& x : th) x.join();
            }
            double tm = (now() - t0) / reps;
            printf(", \"memcpy_%d_threads_GBps\": %.1f", T, bytes / tm / 1e9);
        }
    }
    printf("}\n");
    return 0;
}
