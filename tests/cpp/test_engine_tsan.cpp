// Host logic of csrc/api.hip under ThreadSanitizer, no GPU: the file is compiled as plain C++ against a stub HIP
// runtime (tests/cpp/hipstub: N fake gfx950 devices, device memory = host memory) with the compute kernels' host drivers replaced by
// stand-ins (engine_stubs.cpp), and driven from several threads at once:
//   * entry points that lock ONE context (device-pointer transforms on either device) next to entry points that lock ALL of them
//     (host-pointer MSMs sharded over the devices, pin / unpin / pinned-info), lock order = device-list order;
//   * the single-slot hand-off to the per-device Worker threads (on_devices) from concurrent callers;
//   * the copier threads (upload / download) of the host-pointer batched transforms, incl. several runs per call;
//   * h2hip_init / h2hip_shutdown racing with running entry points and with the threshold getters the Rust shim calls per best_multiexp;
//   * the no-GPU path (H2_STUB_DEVICES=0): cached failure, getters answer "never".
// Results are not checked for arithmetic (no kernels run): what is checked is status codes, that copies through the pipeline arrive
// intact (the stand-in transform adds one to every word), and that TSan reports nothing (it makes the process exit non-zero).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "../../include/halo2hip.h"
#include "../../include/halo2hip_debug.h"

static std::atomic<int> failures{0};
#define CHECK(c)                                              \
    do {                                                      \
        if (!(c)) {                                           \
            printf("FAIL line %d: %s (%s)\n", __LINE__, #c, h2hip_last_error()); \
            failures++;                                       \
        }                                                     \
    } while (0)

extern "C" void* h2stub_device_alloc(int device, size_t bytes);  // engine_stubs.cpp: hipMalloc on a given fake device

static const uint64_t ONE[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};  // R mod r: Fr one, reduced

int main() {
    const size_t n = 1 << 12;
    // ---- two devices
    int ids[2] = {0, 1};
    CHECK(h2hip_init(ids, 2) == 0);
    CHECK(h2hip_num_devices() == 2);
    std::vector<uint64_t> bases(n * 8, 0), scalars(n * 4, 0);
    for (size_t i = 0; i < n; i++) scalars[4 * i] = i;
    std::atomic<bool> stop{false};
    std::vector<std::thread> th;
    // host-pointer MSMs (all-device lock, on_devices) + pin / unpin of the same array
    for (int t = 0; t < 3; t++)
        th.emplace_back([&, t] {
            uint64_t out[12];
            for (int it = 0; it < 40; it++) {
                CHECK(h2hip_msm_bn254(scalars.data(), bases.data(), n, out) == 0);
                if (t == 0 && it % 8 == 0) {
                    CHECK(h2hip_bases_pin(bases.data(), n) == 0);
                    size_t np = 0, bytes = 0;
                    uint32_t c = 0, w = 0;
                    CHECK(h2hip_bases_pinned_info(bases.data(), &np, &c, &w, &bytes) == 0 && np == n);
                }
                if (t == 0 && it % 8 == 4) (void)h2hip_bases_unpin(bases.data());
                const uint64_t* cols[3] = {scalars.data(), scalars.data(), scalars.data()};
                uint64_t outs[36];
                CHECK(h2hip_msm_bn254_batch(cols, bases.data(), n, 3, outs) == 0);
            }
        });
    // device-pointer transforms on each device (single-context lock), batches split by owner (mixed owners: all-device lock)
    for (int t = 0; t < 2; t++)
        th.emplace_back([&, t] {
            void* d = h2stub_device_alloc(t, n * 32);
            void* e = h2stub_device_alloc(1 - t, n * 32);
            for (int it = 0; it < 60; it++) {
                CHECK(h2hip_ntt_bn254_fr_device(d, ONE, 12, nullptr) == 0);
                void* both[2] = {d, e};
                CHECK(h2hip_ntt_bn254_fr_batch_device(both, 2, ONE, 12, nullptr) == 0);
            }
            h2hip_device_free(d);
            h2hip_device_free(e);
        });
    // host-pointer batched transforms: copier threads, shares dealt over the devices, several runs per call
    for (int t = 0; t < 2; t++)
        th.emplace_back([&, t] {
            const size_t m = 1 << 10;
            std::vector<std::vector<uint64_t>> cols(7, std::vector<uint64_t>(m * 4));
            for (int it = 0; it < 30; it++) {
                uint64_t* ptrs[7];
                for (int j = 0; j < 7; j++) {
                    for (size_t i = 0; i < m * 4; i++) cols[j][i] = (uint64_t)(it * 1000 + j * 10) + i;
                    ptrs[j] = cols[j].data();
                }
                CHECK(h2hip_ntt_bn254_fr_batch(ptrs, 7, ONE, 10) == 0);
                for (int j = 0; j < 7; j++)  // the stand-in transform adds one to every word: up, through, and down again intact
                    for (size_t i = 0; i < m * 4; i += 97) CHECK(cols[j][i] == (uint64_t)(it * 1000 + j * 10) + i + 1);
            }
        });
    // pinned proving-key columns: pin (idempotent), look at the cache, unpin, from two threads with their own columns
    for (int t = 0; t < 2; t++)
        th.emplace_back([&, t] {
            const size_t m = 1 << 9;
            std::vector<std::vector<uint64_t>> cols(4, std::vector<uint64_t>(m * 4, (uint64_t)(t + 1)));
            const uint64_t* ptrs[4];
            for (int j = 0; j < 4; j++) ptrs[j] = cols[j].data();
            for (int it = 0; it < 40; it++) {
                CHECK(h2hip_columns_pin(ptrs, 4, m) == 0);
                CHECK(h2hip_columns_pin(ptrs, 4, m) == 0);
                size_t nc = 0, nb = 0;
                CHECK(h2hip_columns_pinned_info(&nc, &nb) == 0 && nc >= 4 && nb >= 4 * m * 32);
                cols[it % 4][0] ^= 0x55;  // a rewritten column: the next pin drops the stale copy and uploads again
                CHECK(h2hip_columns_pin(ptrs, 4, m) == 0);
                CHECK(h2hip_columns_unpin(ptrs, 4) == 0);
            }
        });
    // the getters the shim calls on every best_multiexp / best_fft
    th.emplace_back([&] {
        while (!stop.load()) {
            CHECK(h2hip_msm_min_n() > 0);
            CHECK(h2hip_ntt_min_log_n() > 0);
            (void)h2hip_lazy_pin_after();
            (void)h2hip_num_devices();
        }
    });
    // idempotent re-init racing with everything
    th.emplace_back([&] {
        for (int it = 0; it < 50; it++) CHECK(h2hip_init(ids, 2) == 0);
    });
    for (size_t i = 0; i + 2 < th.size(); i++) th[i].join();  // (the last two threads are the getters loop and the re-init loop)
    stop.store(true);
    th[th.size() - 2].join();
    th[th.size() - 1].join();
    th.clear();
    // small runs force several pipelined runs and grouped steps in one call
    h2hip_debug_set_ntt_host_batch(3 * (32 << 10), 2 * (32 << 10));
    {
        const size_t m = 1 << 10;
        std::vector<std::vector<uint64_t>> cols(9, std::vector<uint64_t>(m * 4, 5));
        uint64_t* ptrs[9];
        for (int j = 0; j < 9; j++) ptrs[j] = cols[j].data();
        CHECK(h2hip_ntt_bn254_fr_batch(ptrs, 9, ONE, 10) == 0);
        for (int j = 0; j < 9; j++) CHECK(cols[j][0] == 6 && cols[j][m * 4 - 1] == 6);
    }
    h2hip_debug_set_ntt_host_batch(0, 0);

    // ---- shutdown / init cycles racing with callers (every call either runs or re-initialises lazily; none may crash or deadlock)
    stop.store(false);
    for (int t = 0; t < 3; t++)
        th.emplace_back([&] {
            uint64_t out[12];
            while (!stop.load()) {
                int rc = h2hip_msm_bn254(scalars.data(), bases.data(), n, out);
                CHECK(rc == 0);
                (void)h2hip_msm_min_n();
                // glibc's rwlock prefers readers: three threads re-entering back to back would starve h2hip_shutdown's exclusive
                // lock for as long as they keep it up (a prover does not call in a closed loop while it shuts the engine down)
                std::this_thread::sleep_for(std::chrono::microseconds(300));
            }
        });
    for (int it = 0; it < 20; it++) {
        h2hip_shutdown();
        int rc = h2hip_init(ids, it % 2 ? 2 : 1);  // a caller may have re-initialised lazily in between: a different list is then EINVAL, by contract
        CHECK(rc == 0 || rc == H2HIP_EINVAL);
    }
    stop.store(true);
    for (auto& t : th) t.join();
    th.clear();
    h2hip_shutdown();

    // ---- no GPU: the failure is cached, the getters say "never", explicit init retries
    setenv("H2_STUB_DEVICES", "0", 1);
    uint64_t out[12];
    CHECK(h2hip_msm_bn254(scalars.data(), bases.data(), n, out) == H2HIP_EDEVICE);
    CHECK(h2hip_msm_min_n() == (size_t)-1);
    CHECK(h2hip_ntt_min_log_n() == 0xffffffffu);
    for (int t = 0; t < 4; t++)
        th.emplace_back([&] {
            uint64_t o[12];
            for (int it = 0; it < 200; it++) {
                CHECK(h2hip_msm_bn254(scalars.data(), bases.data(), n, o) == H2HIP_EDEVICE);
                CHECK(h2hip_msm_min_n() == (size_t)-1);
            }
        });
    for (auto& t : th) t.join();
    th.clear();
    CHECK(h2hip_init(nullptr, 0) == H2HIP_EDEVICE);
    setenv("H2_STUB_DEVICES", "2", 1);
    CHECK(h2hip_init(ids, 2) == 0);  // an explicit init looks again
    CHECK(h2hip_msm_bn254(scalars.data(), bases.data(), n, out) == 0);
    h2hip_shutdown();
    printf(failures.load() ? "FAILED (%d)\n" : "engine host logic under tsan: ok\n", failures.load());
    return failures.load() ? 1 : 0;
}
