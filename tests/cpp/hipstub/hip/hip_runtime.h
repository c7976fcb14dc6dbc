// A stand-in for <hip/hip_runtime.h> that lets the HOST logic of csrc/api.hip (locks, worker and copier threads, the pinned-bases
// cache, init / shutdown) be compiled with g++ and run under ThreadSanitizer in a container without a GPU
// (tests/test_host_cpu.py::test_engine_host_logic_under_tsan).  It is test infrastructure: nothing under halo2-pse_amd/ includes it.
// "Devices" are counters, device memory is host memory, streams run everything at the call (in order by construction), events are
// no-ops, kernels are not launched.  H2_STUB_DEVICES (environment) = number of fake gfx950 devices, default 2.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__

typedef enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidDevice = 101, hipErrorUnknown = 999 } hipError_t;
typedef struct h2stub_stream* hipStream_t;
typedef struct h2stub_event* hipEvent_t;
typedef enum { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 } hipMemcpyKind;
struct dim3 {
    uint32_t x, y, z;
    dim3(uint32_t a = 1, uint32_t b = 1, uint32_t c = 1) : x(a), y(b), z(c) {}
};
static const dim3 threadIdx, blockIdx, blockDim, gridDim;
struct hipDeviceProp_t {
    char gcnArchName[256];
    int multiProcessorCount;
};
struct hipPointerAttribute_t {
    int device;
};
#define hipStreamNonBlocking 1
#define hipEventDisableTiming 2
#define hipHostMallocDefault 0

namespace h2stub {
inline int n_devices() {
    const char* v = getenv("H2_STUB_DEVICES");
    int n = v ? atoi(v) : 2;
    return n < 0 ? 0 : n;
}
inline int& current() {
    static thread_local int dev = 0;
    return dev;
}
struct Allocs {
    std::mutex m;
    std::map<const void*, std::pair<size_t, int>> by_ptr;  // base -> (bytes, device)
};
inline Allocs& allocs() {
    static Allocs a;
    return a;
}
}  // namespace h2stub

inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "stub error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) {
    *n = h2stub::n_devices();
    return *n > 0 ? hipSuccess : hipErrorInvalidDevice;
}
inline hipError_t hipSetDevice(int d) {
    if (d < 0 || d >= h2stub::n_devices()) return hipErrorInvalidDevice;
    h2stub::current() = d;
    return hipSuccess;
}
inline hipError_t hipGetDevice(int* d) {
    *d = h2stub::current();
    return hipSuccess;
}
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof(*p));
    strcpy(p->gcnArchName, "gfx950:sramecc+:xnack-");
    p->multiProcessorCount = 256;
    return hipSuccess;
}
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t bytes) {
    *p = calloc(1, bytes ? bytes : 1);
    if (!*p) return hipErrorOutOfMemory;
    std::lock_guard<std::mutex> lk(h2stub::allocs().m);
    h2stub::allocs().by_ptr[*p] = {bytes, h2stub::current()};
    return hipSuccess;
}
inline hipError_t hipFree(void* p) {
    if (!p) return hipSuccess;
    {
        std::lock_guard<std::mutex> lk(h2stub::allocs().m);
        h2stub::allocs().by_ptr.erase(p);
    }
    free(p);
    return hipSuccess;
}
inline hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) {
    *p = calloc(1, bytes ? bytes : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
inline hipError_t hipHostFree(void* p) {
    free(p);
    return hipSuccess;
}
inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t* at, const void* p) {
    std::lock_guard<std::mutex> lk(h2stub::allocs().m);
    auto& m = h2stub::allocs().by_ptr;
    auto it = m.upper_bound(p);
    if (it == m.begin()) return hipErrorInvalidValue;
    --it;
    if ((const char*)p >= (const char*)it->first + it->second.first) return hipErrorInvalidValue;
    at->device = it->second.second;
    return hipSuccess;
}
inline hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t) {
    memmove(dst, src, bytes);
    return hipSuccess;
}
inline hipError_t hipMemsetAsync(void* dst, int v, size_t bytes, hipStream_t) {
    memset(dst, v, bytes);
    return hipSuccess;
}
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
    *s = (hipStream_t)malloc(8);
    return hipSuccess;
}
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, uint32_t, const uint32_t*) { return hipStreamCreateWithFlags(s, 0); }
inline hipError_t hipStreamDestroy(hipStream_t s) {
    free(s);
    return hipSuccess;
}
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) {
    *e = (hipEvent_t)malloc(8);
    return hipSuccess;
}
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) {
    free(e);
    return hipSuccess;
}
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) {
    *ms = 0.f;
    return hipSuccess;
}
// kernels are not run: the host logic under test does not depend on their results
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
    do {                                                          \
        (void)(grid);                                             \
        (void)(block);                                            \
        (void)(stream);                                           \
    } while (0)
#define hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, e0, e1, flags, ...) hipLaunchKernelGGL(kernel, grid, block, lds, stream)
