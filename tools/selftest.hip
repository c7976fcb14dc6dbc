// tools/selftest.hip -- device-vs-host self check of field.h / ec.h on the GPU box: the same
// H2_HD source runs on the host (validated against the oracle by tests/test_abi.py) and in a kernel.
#include <stdio.h>
#include <string.h>
#include <vector>
#include "../halo2-pse_amd/csrc/ec.h"
using namespace h2;

struct Out { Fe mul, add, sub, sqr, pow, inv, canon; };

template <class P>
H2_HD Out compute(const Fe& a, const Fe& b) {
    Out o;
    o.mul = fe_mul<P>(a, b);
    o.add = fe_add<P>(a, b);
    o.sub = fe_sub<P>(a, b);
    o.sqr = fe_sqr<P>(a);
    uint32_t e[8];
    for (int i = 0; i < 8; i++) e[i] = b.l[i];
    o.pow = fe_pow<P>(a, e);
    o.inv = fe_inv<P>(a);
    o.canon = fe_to_canonical<P>(a);
    return o;
}

template <class P>
__global__ void k(const Fe* a, const Fe* b, Out* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = compute<P>(a[i], b[i]);
}

static uint64_t sm(uint64_t& s) { s += 0x9E3779B97F4A7C15ULL; uint64_t x = s; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL; x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; return x ^ (x >> 31); }

template <class P>
int run(const char* name) {
    const int n = 512;
    std::vector<Fe> a(n), b(n);
    uint64_t s = 42;
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < 8; j++) { a[i].l[j] = (uint32_t)sm(s); b[i].l[j] = (uint32_t)sm(s); }
        a[i].l[7] &= 0x1fffffff; b[i].l[7] &= 0x1fffffff;
    }
    Fe *da, *db; Out* dout;
    hipMalloc(&da, n * sizeof(Fe)); hipMalloc(&db, n * sizeof(Fe)); hipMalloc(&dout, n * sizeof(Out));
    hipMemcpy(da, a.data(), n * sizeof(Fe), hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), n * sizeof(Fe), hipMemcpyHostToDevice);
    k<P><<<(n + 63) / 64, 64>>>(da, db, dout, n);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%s: kernel error %s\n", name, hipGetErrorString(e)); return 1; }
    std::vector<Out> out(n);
    hipMemcpy(out.data(), dout, n * sizeof(Out), hipMemcpyDeviceToHost);
    int bad[7] = {0};
    for (int i = 0; i < n; i++) {
        Out h = compute<P>(a[i], b[i]);
        const Fe* hp = (const Fe*)&h; const Fe* dp = (const Fe*)&out[i];
        for (int f = 0; f < 7; f++) if (!fe_eq(hp[f], dp[f])) bad[f]++;
    }
    printf("%s mismatches: mul=%d add=%d sub=%d sqr=%d pow=%d inv=%d canon=%d\n", name, bad[0], bad[1], bad[2], bad[3], bad[4], bad[5], bad[6]);
    int t = 0; for (int f = 0; f < 7; f++) t += bad[f];
    return t;
}

int main() {
    int r = run<FqP>("Fq") + run<FrP>("Fr");
    printf(r ? "SELFTEST FAIL\n" : "SELFTEST OK\n");
    return r ? 1 : 0;
}
