// msm.hip -- Pippenger bucket MSM over BN254 G1 for gfx950.
//
// Computes the group element halo2_proofs::arithmetic::best_multiexp returns
// (halo2_proofs/src/arithmetic.rs:132-159, inner loop multiexp_serial :13-101):
// sum_i coeffs[i] * bases[i].  The reference splits the pairs over rayon threads and runs an
// unsigned-window bucket method per chunk; its Jacobian coordinates depend on the thread count,
// only the group element is defined (SURVEY.md App. B rule 3), and that is what this engine
// reproduces bit-exactly after normalisation to affine.
//
// GPU schedule (one stream, no host sync until the W window sums come back):
//   K1 msm_digits_kernel     to_repr + get_at (arithmetic.rs:14,24-42): Montgomery -> canonical,
//                            signed c-bit digits; emits (bucket key, point index | sign) pairs
//   sort                     radix sort of the pairs by key (rocPRIM via hipCUB)
//   K2a msm_bounds_kernel    bucket -> [start, end) in the sorted pairs
//   K2  msm_accum_kernel     one lane per bucket: XYZZ mixed adds over its pairs (arithmetic.rs:84-89);
//                            over-full buckets (skewed scalars) go to a chunk list
//   K2h msm_heavy_*          workgroup-per-chunk accumulation + LDS tree for over-full buckets
//   K3  msm_reduce1/2        summation by parts (arithmetic.rs:95-99), segmented: per-lane running
//                            sums over 2^s buckets, then one workgroup per window
//   K4  host                 Horner over the W window sums with c doublings each (arithmetic.rs:46-49)
#include <hipcub/hipcub.hpp>

#include "engine.h"

namespace h2 {

#define MSM_MAX_WINDOWS 64

struct MsmPlan {
    uint32_t c;        // window bits
    uint32_t W;        // number of windows
    uint32_t NB;       // buckets per window = 2^(c-1)
    uint32_t log_s1;   // level-1 segment = 2^log_s1 buckets per lane
    uint32_t heavy_t;  // bucket size above which the chunked path is used
    uint32_t chunk;    // pairs per heavy chunk
};

__device__ __forceinline__ uint32_t scalar_bits(const Fe& s, uint32_t bit, uint32_t c) {
    uint32_t limb = bit >> 5, sh = bit & 31;
    if (limb >= 8) return 0;
    uint64_t lo = s.l[limb];
    uint64_t hi = (limb + 1 < 8) ? s.l[limb + 1] : 0;
    return (uint32_t)(((lo | (hi << 32)) >> sh) & ((1u << c) - 1));
}

// K1: one lane per scalar
__global__ void __launch_bounds__(256) msm_digits_kernel(const Fe* __restrict__ scalars, uint32_t n, uint32_t c, uint32_t W,
                                                         uint32_t NB, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe s = fe_to_canonical<FrP>(scalars[i]);
    uint32_t carry = 0;
    const uint32_t half = 1u << (c - 1);
    const uint32_t sentinel = W * NB;
    for (uint32_t w = 0; w < W; w++) {
        uint32_t d = scalar_bits(s, w * c, c) + carry;
        uint32_t neg = 0;
        carry = 0;
        if (d > half) {
            d = (1u << c) - d;
            neg = 1;
            carry = 1;
        }
        size_t e = (size_t)w * n + i;
        keys[e] = d ? (w * NB + d - 1) : sentinel;
        vals[e] = i | (neg << 31);
    }
}

// K2a: start[b] = first sorted position with key >= b, for b in [0, W*NB]
__global__ void msm_bounds_kernel(const uint32_t* __restrict__ keys, uint32_t n_entries, uint32_t n_keys, uint32_t* __restrict__ start) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > n_keys) return;
    uint32_t lo = 0, hi = n_entries;
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (keys[mid] < b) lo = mid + 1; else hi = mid;
    }
    start[b] = lo;
}

struct HeavyBucket {
    uint32_t bucket, first_chunk, n_chunks, pad;
};
struct HeavyChunk {
    uint32_t begin, end;
};

// one bucket-accumulation step: acc += (+/-) bases[index], in the unsaturated arithmetic of ecu.cuh
__device__ __forceinline__ void accum_signed(XYZZu& acc, const Affine* __restrict__ bases, uint32_t v) {
    Affine p = bases[v & 0x7fffffffu];
    xyzzu_add_affine(acc, p, (v >> 31) != 0);
}

// K2: one lane per bucket
__global__ void __launch_bounds__(256) msm_accum_kernel(const Affine* __restrict__ bases, const uint32_t* __restrict__ vals,
                                                        const uint32_t* __restrict__ start, uint32_t n_buckets, uint32_t heavy_t,
                                                        uint32_t chunk, XYZZu* __restrict__ buckets, uint32_t* __restrict__ heavy_counts,
                                                        HeavyBucket* __restrict__ heavy_buckets, HeavyChunk* __restrict__ heavy_chunks) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_buckets) return;
    uint32_t s = start[b], e = start[b + 1];
    XYZZu acc = xyzzu_identity();
    if (e - s > heavy_t) {
        uint32_t nch = (e - s + chunk - 1) / chunk;
        uint32_t slot = atomicAdd(&heavy_counts[1], nch);
        uint32_t hb = atomicAdd(&heavy_counts[0], 1u);
        HeavyBucket h = {b, slot, nch, 0};
        heavy_buckets[hb] = h;
        for (uint32_t q = 0; q < nch; q++) {
            HeavyChunk ch = {s + q * chunk, (s + (q + 1) * chunk < e) ? s + (q + 1) * chunk : e};
            heavy_chunks[slot + q] = ch;
        }
    } else {
        for (uint32_t i = s; i < e; i++) accum_signed(acc, bases, vals[i]);
    }
    buckets[b] = acc;
}

// tree-sum 256 XYZZu values through LDS; result valid in thread 0
__device__ __forceinline__ XYZZu block_tree_sum(XYZZu v, XYZZu* sh) {
    sh[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t stride = blockDim.x >> 1; stride >= 1; stride >>= 1) {
        if (threadIdx.x < stride) {
            XYZZu a = sh[threadIdx.x];
            xyzzu_add(a, sh[threadIdx.x + stride]);
            sh[threadIdx.x] = a;
        }
        __syncthreads();
    }
    return sh[0];
}

// K2h-1: workgroups stride over the chunk list; every wave exits once its index passes the count
__global__ void __launch_bounds__(256) msm_heavy_chunk_kernel(const Affine* __restrict__ bases, const uint32_t* __restrict__ vals,
                                                              const uint32_t* __restrict__ heavy_counts, const HeavyChunk* __restrict__ heavy_chunks,
                                                              XYZZu* __restrict__ chunk_sums) {
    __shared__ XYZZu sh[256];
    const uint32_t total = heavy_counts[1];
    for (uint32_t ci = blockIdx.x; ci < total; ci += gridDim.x) {
        HeavyChunk ch = heavy_chunks[ci];
        XYZZu acc = xyzzu_identity();
        for (uint32_t i = ch.begin + threadIdx.x; i < ch.end; i += blockDim.x) accum_signed(acc, bases, vals[i]);
        XYZZu r = block_tree_sum(acc, sh);
        if (threadIdx.x == 0) chunk_sums[ci] = r;
        __syncthreads();
    }
}

// K2h-2: one workgroup per over-full bucket sums its chunk sums into the bucket
__global__ void __launch_bounds__(256) msm_heavy_final_kernel(const uint32_t* __restrict__ heavy_counts, const HeavyBucket* __restrict__ heavy_buckets,
                                                              const XYZZu* __restrict__ chunk_sums, XYZZu* __restrict__ buckets) {
    __shared__ XYZZu sh[256];
    const uint32_t total = heavy_counts[0];
    for (uint32_t hi = blockIdx.x; hi < total; hi += gridDim.x) {
        HeavyBucket h = heavy_buckets[hi];
        XYZZu acc = xyzzu_identity();
        for (uint32_t q = threadIdx.x; q < h.n_chunks; q += blockDim.x) xyzzu_add(acc, chunk_sums[h.first_chunk + q]);
        XYZZu r = block_tree_sum(acc, sh);
        if (threadIdx.x == 0) buckets[h.bucket] = r;
        __syncthreads();
    }
}

// K3 level 1: lane t of window w folds buckets [t*s1, (t+1)*s1): run = sum B_i, acc = sum (i - t*s1 + 1) B_i
__global__ void __launch_bounds__(256) msm_reduce1_kernel(const XYZZu* __restrict__ buckets, uint32_t n_seg_total, uint32_t log_s1,
                                                          XYZZu* __restrict__ acc_out, XYZZu* __restrict__ run_out) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_seg_total) return;
    const uint32_t s1 = 1u << log_s1;
    const XYZZu* seg = buckets + ((size_t)t << log_s1);
    XYZZu run = xyzzu_identity(), acc = xyzzu_identity();
    for (uint32_t i = s1; i-- > 0;) {
        xyzzu_add(run, seg[i]);
        xyzzu_add(acc, run);
    }
    acc_out[t] = acc;
    run_out[t] = run;
}

// K3 level 2: one workgroup per window over its m1 (acc, run) pairs:
//   window sum = sum_t ACC[t] + s1 * sum_t t * RUN[t]
__global__ void __launch_bounds__(256) msm_reduce2_kernel(const XYZZu* __restrict__ acc_in, const XYZZu* __restrict__ run_in, uint32_t m1,
                                                          uint32_t log_s1, XYZZ* __restrict__ window_sums) {
    __shared__ XYZZu sh[256];
    const uint32_t w = blockIdx.x;
    const XYZZu* A = acc_in + (size_t)w * m1;
    const XYZZu* Rn = run_in + (size_t)w * m1;
    const uint32_t s2 = (m1 + 255) / 256;
    const uint32_t lo = threadIdx.x * s2;
    XYZZu v = xyzzu_identity();
    if (lo < m1) {
        uint32_t hi = lo + s2 < m1 ? lo + s2 : m1;
        XYZZu a = xyzzu_identity(), run2 = xyzzu_identity(), acc2 = xyzzu_identity();
        for (uint32_t t = hi; t-- > lo;) {
            xyzzu_add(a, A[t]);
            xyzzu_add(run2, Rn[t]);
            if (t != lo) xyzzu_add(acc2, run2);
        }
        // sum_{t in [lo,hi)} t*RUN[t] = acc2 + lo*run2
        XYZZu wsum = xyzzu_mul_small(run2, lo);
        xyzzu_add(wsum, acc2);
        for (uint32_t k = 0; k < log_s1; k++) wsum = xyzzu_double(wsum);
        xyzzu_add(wsum, a);
        v = wsum;
    }
    XYZZu r = block_tree_sum(v, sh);
    if (threadIdx.x == 0) window_sums[w] = xyzzu_to_ext(r);  // canonical E-form for the host Horner
}

static uint32_t g_window_override = 0;
void msm_set_window(uint32_t c) { g_window_override = c; }

static MsmPlan make_plan(size_t n) {
    MsmPlan p;
    uint32_t c;
    if (g_window_override) {
        c = g_window_override;
    } else {
        uint32_t lg = 0;
        while (((size_t)1 << (lg + 1)) <= n) lg++;
        // enough buckets to fill 256 CUs, few enough that the reduction stays small
        if (lg <= 8) c = 6;
        else if (lg <= 12) c = 9;
        else if (lg <= 16) c = 12;
        else if (lg <= 19) c = 14;
        else c = 16;
    }
    if (c < 2) c = 2;
    if (c > 22) c = 22;
    p.c = c;
    p.W = 254 / c + 1;
    p.NB = 1u << (c - 1);
    p.log_s1 = (c - 1) < 4 ? (c - 1) : 4;
    size_t avg = (n * p.W) / ((size_t)p.W * p.NB) + 1;
    size_t t = 4 * avg;
    if (t < 2048) t = 2048;
    p.heavy_t = (uint32_t)t;
    p.chunk = 4096;
    return p;
}

uint32_t msm_get_window(size_t n) { return make_plan(n).c; }

// host Horner over window sums (arithmetic.rs:46-49): acc = sum_w 2^(c*w) * S_w
static XYZZ combine_windows(const XYZZ* ws, const MsmPlan& p) {
    XYZZ acc = xyzz_identity();
    for (uint32_t w = p.W; w-- > 0;) {
        for (uint32_t k = 0; k < p.c; k++) acc = xyzz_double(acc);
        xyzz_add(acc, ws[w]);
    }
    return acc;
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static int msm_device_chunk(Ctx* c, const Fe* d_scalars, const Affine* d_bases, size_t n, XYZZ* h_out, hipStream_t s) {
    MsmPlan p = make_plan(n);
    const size_t E = n * p.W;
    if (E >= ((size_t)1 << 31)) {
        set_error("msm: n*W = %zu pairs exceeds the 2^31 sort limit (window override too small?)", E);
        return 1;
    }
    const uint32_t n_buckets = p.W * p.NB;
    const uint32_t m1 = p.NB >> p.log_s1;
    const size_t max_chunks = E / p.chunk + E / p.heavy_t + 16;  // sum of ceil(cnt/chunk) over buckets with cnt > heavy_t
    const size_t max_heavy = E / p.heavy_t + 16;

    size_t cub_bytes = 0;
    uint32_t end_bit = 1;
    while ((1ull << end_bit) <= (uint64_t)n_buckets) end_bit++;
    H2_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, cub_bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                                (uint32_t*)nullptr, (int)E, 0, (int)end_bit, s));
    // carve the workspace
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    size_t o_keys0 = carve(E * 4), o_keys1 = carve(E * 4), o_vals0 = carve(E * 4), o_vals1 = carve(E * 4);
    size_t o_cub = carve(cub_bytes);
    size_t o_start = carve(((size_t)n_buckets + 2) * 4);
    size_t o_buckets = carve((size_t)n_buckets * sizeof(XYZZu));
    size_t o_acc = carve((size_t)p.W * m1 * sizeof(XYZZu)), o_run = carve((size_t)p.W * m1 * sizeof(XYZZu));
    size_t o_wsum = carve((size_t)p.W * sizeof(XYZZ));
    size_t o_hcnt = carve(16);
    size_t o_hb = carve(max_heavy * sizeof(HeavyBucket)), o_hc = carve(max_chunks * sizeof(HeavyChunk));
    size_t o_hs = carve(max_chunks * sizeof(XYZZu));
    int rc = c->msm_ws.ensure(off);
    if (rc) return rc;
    char* base = (char*)c->msm_ws.p;
    uint32_t *keys0 = (uint32_t*)(base + o_keys0), *keys1 = (uint32_t*)(base + o_keys1);
    uint32_t *vals0 = (uint32_t*)(base + o_vals0), *vals1 = (uint32_t*)(base + o_vals1);
    uint32_t* start = (uint32_t*)(base + o_start);
    XYZZu* buckets = (XYZZu*)(base + o_buckets);
    XYZZu *accs = (XYZZu*)(base + o_acc), *runs = (XYZZu*)(base + o_run);
    XYZZ* wsum = (XYZZ*)(base + o_wsum);
    uint32_t* hcnt = (uint32_t*)(base + o_hcnt);
    HeavyBucket* hb = (HeavyBucket*)(base + o_hb);
    HeavyChunk* hc = (HeavyChunk*)(base + o_hc);
    XYZZu* hs = (XYZZu*)(base + o_hs);

    rc = c->ws_acquire(s);
    if (rc) return rc;
    int t_all = c->timer_begin("msm_total", s);
    int t0 = c->timer_begin("msm_digits", s);
    H2_CHECK(hipMemsetAsync(hcnt, 0, 16, s));
    hipLaunchKernelGGL(msm_digits_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, d_scalars, (uint32_t)n, p.c, p.W, p.NB, keys0, vals0);
    H2_CHECK(hipGetLastError());
    c->timer_end(t0, s);

    int t1 = c->timer_begin("msm_sort", s);
    H2_CHECK(hipcub::DeviceRadixSort::SortPairs(base + o_cub, cub_bytes, keys0, keys1, vals0, vals1, (int)E, 0, (int)end_bit, s));
    hipLaunchKernelGGL(msm_bounds_kernel, dim3((n_buckets + 1 + 255) / 256), dim3(256), 0, s, keys1, (uint32_t)E, n_buckets, start);
    H2_CHECK(hipGetLastError());
    c->timer_end(t1, s);

    int t2 = c->timer_begin("msm_accum", s);
    hipLaunchKernelGGL(msm_accum_kernel, dim3((n_buckets + 255) / 256), dim3(256), 0, s, d_bases, vals1, start, n_buckets, p.heavy_t, p.chunk,
                       buckets, hcnt, hb, hc);
    H2_CHECK(hipGetLastError());
    c->timer_end(t2, s);

    int t3 = c->timer_begin("msm_heavy", s);
    uint32_t hgrid = (uint32_t)(max_chunks < (size_t)c->sm_count * 4 ? max_chunks : (size_t)c->sm_count * 4);
    hipLaunchKernelGGL(msm_heavy_chunk_kernel, dim3(hgrid), dim3(256), 0, s, d_bases, vals1, hcnt, hc, hs);
    H2_CHECK(hipGetLastError());
    uint32_t fgrid = (uint32_t)(max_heavy < (size_t)c->sm_count ? max_heavy : (size_t)c->sm_count);
    hipLaunchKernelGGL(msm_heavy_final_kernel, dim3(fgrid), dim3(256), 0, s, hcnt, hb, hs, buckets);
    H2_CHECK(hipGetLastError());
    c->timer_end(t3, s);

    int t4 = c->timer_begin("msm_reduce", s);
    uint32_t n_seg = p.W * m1;
    hipLaunchKernelGGL(msm_reduce1_kernel, dim3((n_seg + 255) / 256), dim3(256), 0, s, buckets, n_seg, p.log_s1, accs, runs);
    H2_CHECK(hipGetLastError());
    hipLaunchKernelGGL(msm_reduce2_kernel, dim3(p.W), dim3(256), 0, s, accs, runs, m1, p.log_s1, wsum);
    H2_CHECK(hipGetLastError());
    c->timer_end(t4, s);
    c->timer_end(t_all, s);

    XYZZ h_ws[MSM_MAX_WINDOWS * 2];
    if (p.W > MSM_MAX_WINDOWS * 2) {
        set_error("msm: too many windows");
        return 1;
    }
    H2_CHECK(hipMemcpyAsync(h_ws, wsum, (size_t)p.W * sizeof(XYZZ), hipMemcpyDeviceToHost, s));
    H2_CHECK(hipStreamSynchronize(s));
    *h_out = combine_windows(h_ws, p);
    return c->ws_release(s);
}

// Sum of coeffs[i]*bases[i] for device-resident inputs; result (XYZZ) to host memory.
int msm_device(Ctx* c, const Fe* d_scalars, const Affine* d_bases, size_t n, XYZZ* h_out, hipStream_t s) {
    *h_out = xyzz_identity();
    if (n == 0) return 0;
    // the pair index lives in 31 bits and the sort counts in int: split very large inputs
    const size_t max_chunk = (size_t)1 << 26;
    for (size_t o = 0; o < n; o += max_chunk) {
        size_t m = n - o < max_chunk ? n - o : max_chunk;
        XYZZ part;
        int rc = msm_device_chunk(c, d_scalars + o, d_bases + o, m, &part, s);
        if (rc) return rc;
        xyzz_add(*h_out, part);
    }
    return 0;
}

}  // namespace h2
