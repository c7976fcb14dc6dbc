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
#include <string.h>

#include <vector>

#include "engine.h"
#include "host64.h"

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
// A fused batch launches it once with gridDim.y = MSMs: MSM y reads list[y] and writes windows [y * W, (y + 1) * W).
__global__ void __launch_bounds__(256) msm_digits_kernel(const Fe* __restrict__ scalars_one, const Fe* const* __restrict__ list, uint32_t n, uint32_t c,
                                                         uint32_t W, uint32_t NB, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fe* __restrict__ scalars = list ? list[blockIdx.y] : scalars_one;
    const uint32_t w_base = blockIdx.y * W;
    Fe s = fe_to_canonical<FrP>(scalars[i]);
    uint32_t carry = 0;
    const uint32_t half = 1u << (c - 1);
    // key = (window << c) | slot, slot = |digit| - 1 in [0, NB) or NB for a zero digit (skipped): pairs are
    // written window-major, so each window's n pairs can also be sorted on their own on the low c bits
    for (uint32_t w = 0; w < W; w++) {
        uint32_t d = scalar_bits(s, w * c, c) + carry;
        uint32_t neg = 0;
        carry = 0;
        if (d > half) {
            d = (1u << c) - d;
            neg = 1;
            carry = 1;
        }
        size_t e = (size_t)(w_base + w) * n + i;
        keys[e] = ((w_base + w) << c) | (d ? d - 1 : NB);
        vals[e] = i | (neg << 31);
    }
}

// K2a: bucket bounds by boundary detection over the sorted keys (coalesced, no searches): the lane that sees
// a key change records where the new key's run starts and where the previous key's run ended.  start/end are
// zeroed beforehand, so an empty bucket reads as [0, 0).
__global__ void __launch_bounds__(256) msm_bounds_kernel(const uint32_t* __restrict__ keys, size_t e_begin, size_t e_end, uint32_t c,
                                                         uint32_t* __restrict__ start, uint32_t* __restrict__ end) {
    const uint32_t NB = 1u << (c - 1), slot_mask = (1u << c) - 1;
    for (size_t i = e_begin + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < e_end; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t k = keys[i];
        const bool first = (i == e_begin);
        const uint32_t kp = first ? 0xffffffffu : keys[i - 1];
        if (first || kp != k) {
            if ((k & slot_mask) < NB) start[((k >> c) << (c - 1)) | (k & slot_mask)] = (uint32_t)i;
            if (!first && (kp & slot_mask) < NB) end[((kp >> c) << (c - 1)) | (kp & slot_mask)] = (uint32_t)i;
        }
        if (i + 1 == e_end && (k & slot_mask) < NB) end[((k >> c) << (c - 1)) | (k & slot_mask)] = (uint32_t)(i + 1);
    }
}

// K2b/K2c: order the buckets by size (descending, in 256 classes of width 2^bin_shift) with a counting sort, so that
// the 64 lanes of an accumulate wave get buckets of near-equal size.  hist[0..256) = class counts, hist[256..512)
// = per-class cursors; both zeroed beforehand.
__device__ __forceinline__ uint32_t size_class(uint32_t cnt, uint32_t bin_shift) {
    uint32_t b = cnt >> bin_shift;
    return 255u - (b < 255u ? b : 255u);  // big buckets first
}

__global__ void __launch_bounds__(256) msm_bucket_hist_kernel(const uint32_t* __restrict__ start, const uint32_t* __restrict__ end, uint32_t gb_base,
                                                              uint32_t n_buckets, uint32_t bin_shift, uint32_t* __restrict__ counts,
                                                              uint32_t* __restrict__ hist) {
    __shared__ uint32_t lh[256];
    lh[threadIdx.x] = 0;
    __syncthreads();
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_buckets) {
        uint32_t gb = gb_base + t;
        uint32_t cnt = end[gb] - start[gb];
        counts[gb] = cnt;
        atomicAdd(&lh[size_class(cnt, bin_shift)], 1u);
    }
    __syncthreads();
    if (lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
}

__global__ void __launch_bounds__(256) msm_bucket_scatter_kernel(const uint32_t* __restrict__ counts, uint32_t gb_base, uint32_t n_buckets,
                                                                 uint32_t bin_shift, uint32_t* __restrict__ hist, uint32_t* __restrict__ perm) {
    __shared__ uint32_t scan[256], lh[256], lbase[256];
    // exclusive scan of the 256 class counts (every block repeats it: 256 values)
    uint32_t v = hist[threadIdx.x];
    scan[threadIdx.x] = v;
    lh[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {
        uint32_t add = threadIdx.x >= off ? scan[threadIdx.x - off] : 0;
        __syncthreads();
        scan[threadIdx.x] += add;
        __syncthreads();
    }
    const uint32_t excl = scan[threadIdx.x] - v;
    __syncthreads();
    scan[threadIdx.x] = excl;
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t cls = 0, rank = 0;
    const bool live = t < n_buckets;
    if (live) {
        cls = size_class(counts[gb_base + t], bin_shift);
        rank = atomicAdd(&lh[cls], 1u);
    }
    __syncthreads();
    if (lh[threadIdx.x]) lbase[threadIdx.x] = atomicAdd(&hist[256 + threadIdx.x], lh[threadIdx.x]);
    __syncthreads();
    if (live) perm[scan[cls] + lbase[cls] + rank] = gb_base + t;
}

struct HeavyBucket {
    uint32_t bucket, first_chunk, n_chunks, pad;
};
struct HeavyChunk {
    uint32_t begin, end;
};

// one bucket-accumulation step: acc += (+/-) bases[index], in the unsaturated arithmetic of ecu.cuh
template <class F>
__device__ __forceinline__ void accum_signed(XYZZu& acc, const Affine* __restrict__ bases, uint32_t v) {
    Affine p = bases[v & 0x7fffffffu];
    xyzzu_add_affine<F>(acc, p, (v >> 31) != 0);
}

// K2: one lane per bucket
__global__ void __launch_bounds__(256) msm_accum_kernel(const Affine* __restrict__ bases, const uint32_t* __restrict__ vals,
                                                        const uint32_t* __restrict__ start, const uint32_t* __restrict__ counts,
                                                        const uint32_t* __restrict__ perm, uint32_t n_buckets, uint32_t heavy_t,
                                                        uint32_t chunk, XYZZu* __restrict__ buckets, uint32_t* __restrict__ heavy_counts,
                                                        HeavyBucket* __restrict__ heavy_buckets, HeavyChunk* __restrict__ heavy_chunks) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_buckets) return;
    const uint32_t b = perm[t];  // buckets in descending size order
    uint32_t s = start[b], e = s + counts[b];
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
        for (uint32_t i = s; i < e; i++) accum_signed<FqUA>(acc, bases, vals[i]);  // throughput-bound: explicit-mad multiplier
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
        for (uint32_t i = ch.begin + threadIdx.x; i < ch.end; i += blockDim.x) accum_signed<FqU>(acc, bases, vals[i]);
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

// K3: summation by parts (arithmetic.rs:95-99), restated so that no lane runs a long chain of dependent
// EC additions (one XYZZ add is ~3.5 k instructions = ~6 us for a lone wave):
//   R1  lane t of window w folds s1 = 2^log_s1 consecutive buckets by running sums:
//         RUN[t] = sum B_i,  ACC[t] = sum (i - t*s1 + 1) B_i         (chain 2*s1)
//       window sum = sum_t ACC[t] + s1 * sum_t t * RUN[t]
//   R2  the weighted sum over t is taken digit by digit in radix 32: t = sum_d t_d 32^d, so
//         sum_t t*RUN[t] = sum_d 32^d sum_v v * T[d][v],   T[d][v] = sum_{t : t_d = v} RUN[t]
//       one workgroup per (window, d, v) forms T[d][v] (plain sum: per-lane partial + LDS tree); slot d = D
//       holds 32 partial plain sums of ACC.
//   R3  one workgroup per window: v*T[d][v] by double-and-add, tree over v, Horner over d, times s1, plus ACC.
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

struct DigitPlan {
    uint32_t D;          // number of digits of the segment index t (<= 7)
    uint32_t width[8];   // bits of digit d (<= 5), split as evenly as possible so every T[d][v] sums equally many terms
    uint32_t shift[8];   // bit position of digit d
};

// T laid out [w][d][v]; LPT lanes of a single-wave workgroup form one T.  One wave64 already saturates its SIMD's issue
// rate (a lone wave's EC add takes 6.5 us = 3.5 k instructions at ~4 cycles), and a wave pays full price for a tree level
// however few of its lanes are active, so what counts is waves x chain length against the 1024 SIMDs:
//   LPT = 32, two T's per wave: a lone MSM (2 048 T's -> one wave per SIMD, chain = sums/32 + 5 tree levels; measured 170 -> 158 us
//   at 2^20 and 167 -> 120 us at 2^17 against one T per wave);
//   LPT = 16, four T's per wave: fused batches, tens of thousands of T's, throughput-bound (a third of the wave-additions of LPT = 64).
template <int LPT>
__global__ void __launch_bounds__(64) msm_reduce2_kernel(const XYZZu* __restrict__ acc_in, const XYZZu* __restrict__ run_in, uint32_t m1,
                                                         DigitPlan dp, uint32_t n_T, XYZZu* __restrict__ T) {
    __shared__ XYZZu sh[64];
    const uint32_t D = dp.D;
    const uint32_t sub = threadIdx.x % LPT;
    const uint32_t ti = blockIdx.x * (64 / LPT) + threadIdx.x / LPT;  // which T this lane works for
    const bool live = ti < n_T;
    const uint32_t v = ti & 31;
    const uint32_t d = (ti >> 5) % (D + 1);
    const uint32_t w = (ti >> 5) / (D + 1);
    XYZZu acc = xyzzu_identity();
    if (!live) {
    } else if (d == D) {
        // plain sum of ACC[t] over t == v (mod 32)
        const XYZZu* A = acc_in + (size_t)w * m1;
        for (uint32_t t = v + 32 * sub; t < m1; t += 32 * LPT) xyzzu_add(acc, A[t]);
    } else if (v < (1u << dp.width[d])) {
        // RUN[t] over the t whose digit d is v: t = hi << (shift + width) | v << shift | lo
        const XYZZu* Rn = run_in + (size_t)w * m1;
        const uint32_t sh_d = dp.shift[d], wd = dp.width[d];
        const uint32_t n_sel = (((m1 - 1) >> (sh_d + wd)) + 1) << sh_d;  // (hi, lo) combinations that can land below m1
        for (uint32_t q = sub; q < n_sel; q += LPT) {
            uint32_t lo = q & ((1u << sh_d) - 1), hi = q >> sh_d;
            uint32_t t = (hi << (sh_d + wd)) | (v << sh_d) | lo;
            if (t < m1) xyzzu_add(acc, Rn[t]);
        }
    }
    // tree over the LPT lanes of each group
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t stride = LPT >> 1; stride >= 1; stride >>= 1) {
        if (sub < stride) {
            XYZZu a = sh[threadIdx.x];
            xyzzu_add(a, sh[threadIdx.x + stride]);
            sh[threadIdx.x] = a;
        }
        __syncthreads();
    }
    if (sub == 0 && live) T[ti] = sh[threadIdx.x];
}

// one workgroup of 256 per window: lane (d, v) scales T[d][v] by v << shift[d] (double-and-add), one tree over
// all lanes sums them together with the ACC partial sums, then times s1
__global__ void __launch_bounds__(256) msm_reduce3_kernel(const XYZZu* __restrict__ T, DigitPlan dp, uint32_t log_s1,
                                                          XYZZ* __restrict__ window_sums) {
    __shared__ XYZZu sh[256];
    const uint32_t D = dp.D;
    const uint32_t w = blockIdx.x;
    const XYZZu* Tw = T + (size_t)w * (D + 1) * 32;
    const uint32_t d = threadIdx.x >> 5, v = threadIdx.x & 31;  // D + 1 <= 8 slots of 32 lanes
    XYZZu x = xyzzu_identity();
    if (d < D) {
        if (v != 0 && v < (1u << dp.width[d])) x = xyzzu_mul_small(Tw[d * 32 + v], (v << dp.shift[d]) << log_s1, dp.shift[d] + dp.width[d] + log_s1);
    } else if (d == D) {
        x = Tw[D * 32 + v];
    }
    XYZZu r = block_tree_sum(x, sh);
    if (threadIdx.x == 0) window_sums[w] = xyzzu_to_ext(r);  // canonical E-form for the host Horner
}

static uint32_t g_window_override = 0;
static size_t g_max_chunk = (size_t)1 << 26;
void msm_set_max_chunk(size_t m) { g_max_chunk = m ? m : ((size_t)1 << 26); }
static uint32_t g_reserved_cus = 0;  // measured on MI355X: every partition (16..96 CUs) was slower than none
void msm_set_reserved_cus(uint32_t k) { g_reserved_cus = k; }
uint32_t msm_get_reserved_cus() { return g_reserved_cus; }
void msm_set_window(uint32_t c) { g_window_override = c; }

static MsmPlan make_plan(size_t n, bool fused = false) {
    MsmPlan p;
    uint32_t c;
    if (g_window_override) {
        c = g_window_override;
    } else {
        uint32_t lg = 0;
        while (((size_t)1 << (lg + 1)) <= n) lg++;
        // enough buckets to fill 256 CUs, few enough that the reduction stays small, and 254 mod c large so
        // that the top window is not a handful of over-full buckets
        // a fused batch has count times the buckets for the same chain lengths: narrower windows pay (measured per MSM,
        // 16 x 2^17: c = 13 0.42 ms, c = 15 0.47 ms; 16 x 2^13: c = 10 0.156, c = 13 0.183)
        if (fused && lg >= 13 && lg <= 18) c = lg <= 13 ? 10 : lg <= 16 ? 12 : lg == 17 ? 13 : 14;
        else if (lg <= 8) c = 7;
        else if (lg <= 12) c = 10;
        else if (lg <= 15) c = 13;
        else if (lg <= 18) c = 15;
        else c = 16;  // re-swept after the sort went to two passes: 2^16 0.79 (c = 15) vs 0.83 ms (13); 2^19 1.43 (16) vs 1.53 ms (15)
    }
    if (c < 2) c = 2;
    if (c > 22) c = 22;
    p.c = c;
    p.W = 254 / c + 1;
    p.NB = 1u << (c - 1);
    p.log_s1 = (c - 1) < 3 ? (c - 1) : 3;  // short chains when there are few buckets, 8-bucket segments when many
    // A lone lane adds ~6 us per pair, and the kernel cannot finish faster than 2 * n*W / 65536 add-times
    // anyway (64 lanes x 1024 SIMDs): buckets above that go to the chunked path, or one lane's chain
    // (e.g. the few buckets of a narrow top window) sets the kernel's duration.
    size_t t = (n * p.W) / 32768;
    if (t < 32) t = 32;
    p.heavy_t = (uint32_t)t;
    p.chunk = 4096;
    return p;
}

uint32_t msm_get_window(size_t n) { return make_plan(n).c; }

// host Horner over window sums (arithmetic.rs:46-49): acc = sum_w 2^(c*w) * S_w
static XYZZ combine_windows(const XYZZ* ws, const MsmPlan& p) {
    h64::P acc = h64::identity();
    for (uint32_t w = p.W; w-- > 0;) {
        for (uint32_t k = 0; k < p.c; k++) acc = h64::pdouble(acc);
        h64::padd(acc, h64::from_xyzz(ws[w]));
    }
    return h64::to_xyzz(acc);
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Workspace layout of one in-flight MSM ("slot")
struct MsmLayout {
    uint32_t fuse, Wt;  // MSMs sharing this workspace as one run (1 normally) and their windows in total: fuse * p.W
    MsmPlan p;
    DigitPlan dplan;
    size_t n, E;
    uint32_t n_buckets, m1, D, bin_shift;
    int end_bit;
    size_t cub_bytes, max_chunks, max_heavy;
    size_t o_keys0, o_keys1, o_vals0, o_vals1, o_cub, o_start, o_end, o_counts, o_perm, o_hist, o_buckets, o_acc, o_run, o_wsum, o_T, o_hcnt,
        o_hb, o_hc, o_hs, o_zero_end, o_ptrs, total;
};

static int msm_layout(size_t n, hipStream_t s, MsmLayout* L, uint32_t fuse = 1) {
    MsmPlan p = make_plan(n, fuse > 1);
    L->p = p;
    L->n = n;
    L->fuse = fuse;
    L->Wt = p.W * fuse;
    L->E = n * L->Wt;
    if (L->E >= ((size_t)1 << 31)) {
        set_error("msm: n*W = %zu pairs exceeds the 2^31 sort limit (window override too small?)", L->E);
        return 1;
    }
    if (p.W > MSM_MAX_WINDOWS * 2 || L->Wt > 4096) {
        set_error("msm: too many windows");
        return 1;
    }
    if ((uint64_t)L->Wt * p.NB >= (1ull << 31)) {
        set_error("msm: too many buckets");
        return 1;
    }
    L->n_buckets = L->Wt * p.NB;
    L->m1 = p.NB >> p.log_s1;
    // Only the slot bits are sorted.  The pairs leave the digits kernel window-major and the radix sort is stable, so after
    // sorting on the low c bits the entries of one (window, slot) bucket are still one contiguous run (ordered by slot, then
    // window, then pair index) -- which is all the bounds kernel and the accumulate kernel need.  That is 2 radix passes for
    // c <= 16 instead of the 3 a sort on (window, slot) takes.  One sort of all n*W pairs at every size (one sort per window,
    // the earlier choice above 2^22 pairs, measured slower: 8.1 vs 7.5 ms at 2^22, 27.9 vs 27.1 ms at 2^24).
    L->end_bit = (int)p.c;
    L->cub_bytes = 0;
    H2_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, L->cub_bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                                (uint32_t*)nullptr, (int)L->E, 0, L->end_bit, s));
    L->max_chunks = L->E / p.chunk + L->E / p.heavy_t + 16;  // sum of ceil(cnt/chunk) over buckets with cnt > heavy_t
    L->max_heavy = L->E / p.heavy_t + 16;
    // digits of the segment index t < m1: ceil(bits / 5) digits of near-equal width
    uint32_t tbits = 0;
    while ((1u << tbits) < L->m1) tbits++;
    memset(&L->dplan, 0, sizeof(L->dplan));
    L->D = tbits ? (tbits + 4) / 5 : 1;
    L->dplan.D = L->D;
    for (uint32_t d = 0, pos = 0; d < L->D; d++) {
        uint32_t wd = tbits / L->D + (d < tbits % L->D ? 1 : 0);
        L->dplan.width[d] = wd;
        L->dplan.shift[d] = pos;
        pos += wd;
    }
    // size classes: width 2^bin_shift pairs, the mean bucket size lands in classes 50..100 (of 256)
    L->bin_shift = 0;
    while (((n / p.NB) >> L->bin_shift) > 100) L->bin_shift++;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t E = L->E;
    const uint32_t nb = L->n_buckets;
    L->o_keys0 = carve(E * 4); L->o_keys1 = carve(E * 4); L->o_vals0 = carve(E * 4); L->o_vals1 = carve(E * 4);
    L->o_cub = carve(L->cub_bytes);
    L->o_start = carve(((size_t)nb + 2) * 4);
    L->o_end = carve(((size_t)nb + 2) * 4);
    L->o_hist = carve(512 * 4);
    L->o_hcnt = carve(16);
    L->o_zero_end = off;  // start[], end[], hist[], heavy counters are contiguous: one memset clears them all
    L->o_counts = carve((size_t)nb * 4);
    L->o_perm = carve((size_t)nb * 4);
    L->o_buckets = carve((size_t)nb * sizeof(XYZZu));
    L->o_acc = carve((size_t)L->Wt * L->m1 * sizeof(XYZZu));
    L->o_run = carve((size_t)L->Wt * L->m1 * sizeof(XYZZu));
    L->o_wsum = carve((size_t)L->Wt * sizeof(XYZZ));
    L->o_T = carve((size_t)L->Wt * (L->D + 1) * 32 * sizeof(XYZZu));
    L->o_hb = carve(L->max_heavy * sizeof(HeavyBucket));
    L->o_hc = carve(L->max_chunks * sizeof(HeavyChunk));
    L->o_hs = carve(L->max_chunks * sizeof(XYZZu));
    L->o_ptrs = carve((size_t)fuse * sizeof(void*));
    L->total = off;
    return 0;
}

// stage A (memory-bound): digits, radix sort, bucket bounds, size order
static int msm_stage_a(Ctx* c, const MsmLayout& L, char* base, const Fe* const* d_scalars_list, hipStream_t s) {
    const MsmPlan& p = L.p;
    const size_t n = L.n;
    uint32_t *keys0 = (uint32_t*)(base + L.o_keys0), *keys1 = (uint32_t*)(base + L.o_keys1);
    uint32_t *vals0 = (uint32_t*)(base + L.o_vals0), *vals1 = (uint32_t*)(base + L.o_vals1);
    uint32_t *start = (uint32_t*)(base + L.o_start), *endp = (uint32_t*)(base + L.o_end);
    uint32_t *counts = (uint32_t*)(base + L.o_counts), *perm = (uint32_t*)(base + L.o_perm), *hist = (uint32_t*)(base + L.o_hist);
    int t0 = c->timer_begin("msm_digits", s);
    H2_CHECK(hipMemsetAsync(start, 0, L.o_zero_end - L.o_start, s));  // start[], end[], hist[], heavy counters
    const Fe* const* d_list = nullptr;
    if (L.fuse > 1) {  // the scalar arrays' addresses, read by the kernel per blockIdx.y
        H2_CHECK(hipMemcpyAsync(base + L.o_ptrs, d_scalars_list, L.fuse * sizeof(void*), hipMemcpyHostToDevice, s));  // pageable: staged before return
        d_list = (const Fe* const*)(base + L.o_ptrs);
    }
    hipLaunchKernelGGL(msm_digits_kernel, dim3((uint32_t)((n + 255) / 256), L.fuse), dim3(256), 0, s, d_scalars_list[0], d_list, (uint32_t)n, p.c, p.W, p.NB,
                       keys0, vals0);
    H2_CHECK(hipGetLastError());
    c->timer_end(t0, s);
    int t1 = c->timer_begin("msm_sort", s);
    H2_CHECK(hipcub::DeviceRadixSort::SortPairs(base + L.o_cub, const_cast<size_t&>(L.cub_bytes), keys0, keys1, vals0, vals1, (int)L.E, 0, L.end_bit, s));
    {
        size_t blocks = (L.E + 255) / 256;
        uint32_t grid = (uint32_t)(blocks < (size_t)c->sm_count * 16 ? blocks : (size_t)c->sm_count * 16);
        hipLaunchKernelGGL(msm_bounds_kernel, dim3(grid), dim3(256), 0, s, keys1, (size_t)0, L.E, p.c, start, endp);
        H2_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(msm_bucket_hist_kernel, dim3((L.n_buckets + 255) / 256), dim3(256), 0, s, start, endp, 0u, L.n_buckets, L.bin_shift, counts, hist);
    H2_CHECK(hipGetLastError());
    hipLaunchKernelGGL(msm_bucket_scatter_kernel, dim3((L.n_buckets + 255) / 256), dim3(256), 0, s, counts, 0u, L.n_buckets, L.bin_shift, hist, perm);
    H2_CHECK(hipGetLastError());
    c->timer_end(t1, s);
    return 0;
}

// stage B (VALU-bound): bucket accumulation, plus the chunked path for over-full buckets
static int msm_stage_b(Ctx* c, const MsmLayout& L, char* base, const Affine* d_bases, hipStream_t s) {
    const MsmPlan& p = L.p;
    uint32_t* vals1 = (uint32_t*)(base + L.o_vals1);
    uint32_t *start = (uint32_t*)(base + L.o_start), *counts = (uint32_t*)(base + L.o_counts), *perm = (uint32_t*)(base + L.o_perm);
    XYZZu* buckets = (XYZZu*)(base + L.o_buckets);
    uint32_t* hcnt = (uint32_t*)(base + L.o_hcnt);
    HeavyBucket* hb = (HeavyBucket*)(base + L.o_hb);
    HeavyChunk* hc = (HeavyChunk*)(base + L.o_hc);
    XYZZu* hs = (XYZZu*)(base + L.o_hs);
    int t2 = c->timer_begin("msm_accum", s);
    hipLaunchKernelGGL(msm_accum_kernel, dim3((L.n_buckets + 255) / 256), dim3(256), 0, s, d_bases, vals1, start, counts, perm,
                       L.n_buckets, p.heavy_t, p.chunk, buckets, hcnt, hb, hc);
    H2_CHECK(hipGetLastError());
    c->timer_end(t2, s);
    int t3 = c->timer_begin("msm_heavy", s);
    uint32_t hgrid = (uint32_t)(L.max_chunks < (size_t)c->sm_count * 4 ? L.max_chunks : (size_t)c->sm_count * 4);
    hipLaunchKernelGGL(msm_heavy_chunk_kernel, dim3(hgrid), dim3(256), 0, s, d_bases, vals1, hcnt, hc, hs);
    H2_CHECK(hipGetLastError());
    uint32_t fgrid = (uint32_t)(L.max_heavy < (size_t)c->sm_count ? L.max_heavy : (size_t)c->sm_count);
    hipLaunchKernelGGL(msm_heavy_final_kernel, dim3(fgrid), dim3(256), 0, s, hcnt, hb, hs, buckets);
    H2_CHECK(hipGetLastError());
    c->timer_end(t3, s);
    return 0;
}

// stage C (latency-bound): bucket reduction to W window sums, copied to h_ws (host)
static int msm_stage_c(Ctx* c, const MsmLayout& L, char* base, XYZZ* h_ws, hipStream_t s) {
    const MsmPlan& p = L.p;
    XYZZu* buckets = (XYZZu*)(base + L.o_buckets);
    XYZZu *accs = (XYZZu*)(base + L.o_acc), *runs = (XYZZu*)(base + L.o_run), *Tsum = (XYZZu*)(base + L.o_T);
    XYZZ* wsum = (XYZZ*)(base + L.o_wsum);
    int t4 = c->timer_begin("msm_reduce", s);
    uint32_t n_seg = L.Wt * L.m1;
    hipLaunchKernelGGL(msm_reduce1_kernel, dim3((n_seg + 255) / 256), dim3(256), 0, s, buckets, n_seg, p.log_s1, accs, runs);
    H2_CHECK(hipGetLastError());
    {
        const uint32_t n_T = L.Wt * (L.D + 1) * 32;
        if (L.fuse > 1)
            hipLaunchKernelGGL(msm_reduce2_kernel<16>, dim3((n_T + 3) / 4), dim3(64), 0, s, accs, runs, L.m1, L.dplan, n_T, Tsum);
        else
            hipLaunchKernelGGL(msm_reduce2_kernel<32>, dim3((n_T + 1) / 2), dim3(64), 0, s, accs, runs, L.m1, L.dplan, n_T, Tsum);
    }
    H2_CHECK(hipGetLastError());
    hipLaunchKernelGGL(msm_reduce3_kernel, dim3(L.Wt), dim3(256), 0, s, Tsum, L.dplan, p.log_s1, wsum);
    H2_CHECK(hipGetLastError());
    c->timer_end(t4, s);
    H2_CHECK(hipMemcpyAsync(h_ws, wsum, (size_t)L.Wt * sizeof(XYZZ), hipMemcpyDeviceToHost, s));
    return 0;
}

// `count` independent MSMs of n <= 2^26 pairs each over the same bases (ParamsKZG::commit_lagrange for the
// advice columns of one proof, plonk/prover.rs:361-365).  count == 1 runs the three stages back to back on the
// caller's stream.  count > 1 pipelines whole MSMs over three streams with three workspace slots:
//   aux1:  A(0) | A(1) | A(2) | ...            digits + sort + bounds      (memory-bound)
//   s   :       | B(0) | B(1) | ...            accumulate                  (VALU-bound)
//   aux2:              | C(0) | C(1) | ...     reduce + copy-out           (latency-bound, few waves)
// so that every accumulate launch is full-size while the sort of the next MSM and the reduction of the previous
// one run underneath it.
// scalars_on_host: d_scalars[j] are host pointers, uploaded into three device slots ahead of stage A.
static int msm_batch_chunk(Ctx* c, const Fe* const* d_scalars, bool scalars_on_host, const Affine* d_bases, size_t n, size_t count, XYZZ* h_out,
                           hipStream_t s) {
    MsmLayout L;
    int rc = msm_layout(n, s, &L);
    if (rc) return rc;
    const size_t n_slots = count < 3 ? count : 3;
    for (size_t k = 0; k < n_slots; k++) {
        rc = c->msm_slot[k].ensure(L.total);
        if (rc) return rc;
        if (scalars_on_host) {
            rc = c->msm_scalars[k].ensure(n * sizeof(Fe));
            if (rc) return rc;
        }
    }
    auto scalars_for = [&](size_t j, hipStream_t st, const Fe** out) -> int {
        if (!scalars_on_host) {
            *out = d_scalars[j];
            return 0;
        }
        Fe* dst = (Fe*)c->msm_scalars[j % 3].p;  // stage A(j-3) read this slot earlier on the same stream
        H2_CHECK(hipMemcpyAsync(dst, d_scalars[j], n * sizeof(Fe), hipMemcpyHostToDevice, st));
        *out = dst;
        return 0;
    };
    rc = c->host_ws.ensure(count * L.p.W * sizeof(XYZZ));
    if (rc) return rc;
    XYZZ* h_ws = (XYZZ*)c->host_ws.p;
    rc = c->ws_acquire(s);
    if (rc) return rc;
    int t_all = c->timer_begin("msm_total", s);
    if (count == 1) {
        char* base = (char*)c->msm_slot[0].p;
        const Fe* sc;
        if ((rc = scalars_for(0, s, &sc))) return rc;
        if ((rc = msm_stage_a(c, L, base, &sc, s))) return rc;
        if ((rc = msm_stage_b(c, L, base, d_bases, s))) return rc;
        if ((rc = msm_stage_c(c, L, base, h_ws, s))) return rc;
    } else {
        // Three internal streams.  A full-size accumulate holds every wave slot for ~0.65 ms at a time, so the other
        // stages only partly overlap with it (trace: profiles/): measured gain 17 % at 2^20, 37 % at 2^17.  Reserving
        // CUs for stages A/C with CU-masked streams (h2hip_debug_set_reserved_cus) was slower for every split tried.
        rc = c->ensure_aux(3 * count + 2);
        if (rc) return rc;
        hipStream_t a1 = c->aux1, a2 = c->aux2, sb = c->aux_b;
        hipEvent_t* ev = c->aux_events.data();  // [0] start, [1+3j] A(j) done, [2+3j] B(j) done, [3+3j] C(j) done
        H2_CHECK(hipEventRecord(ev[0], s));
        H2_CHECK(hipStreamWaitEvent(a1, ev[0], 0));
        H2_CHECK(hipStreamWaitEvent(a2, ev[0], 0));
        H2_CHECK(hipStreamWaitEvent(sb, ev[0], 0));
        for (size_t j = 0; j < count; j++) {
            char* base = (char*)c->msm_slot[j % 3].p;
            if (j >= 3) H2_CHECK(hipStreamWaitEvent(a1, ev[3 + 3 * (j - 3)], 0));  // slot reuse: C(j-3) has drained it
            const Fe* sc;
            if ((rc = scalars_for(j, a1, &sc))) return rc;
            if ((rc = msm_stage_a(c, L, base, &sc, a1))) return rc;
            H2_CHECK(hipEventRecord(ev[1 + 3 * j], a1));
            H2_CHECK(hipStreamWaitEvent(sb, ev[1 + 3 * j], 0));
            if ((rc = msm_stage_b(c, L, base, d_bases, sb))) return rc;
            H2_CHECK(hipEventRecord(ev[2 + 3 * j], sb));
            H2_CHECK(hipStreamWaitEvent(a2, ev[2 + 3 * j], 0));
            if ((rc = msm_stage_c(c, L, base, h_ws + j * L.p.W, a2))) return rc;
            H2_CHECK(hipEventRecord(ev[3 + 3 * j], a2));
        }
        H2_CHECK(hipStreamWaitEvent(s, ev[3 + 3 * (count - 1)], 0));  // C stages are in order on aux2
    }
    c->timer_end(t_all, s);
    H2_CHECK(hipStreamSynchronize(s));
    for (size_t j = 0; j < count; j++) h_out[j] = combine_windows(h_ws + j * L.p.W, L.p);
    return c->ws_release(s);
}

// Small MSMs over the same bases, fused: the `count` MSMs run as ONE pass of the three stages whose windows are the
// windows of all of them (window j * W + w of the run is window w of MSM j).  The bucket reduction is a chain of ~60
// dependent EC operations -- 0.43 ms whatever the size -- and dominates a 2^17-pair MSM; pipelining whole MSMs over streams
// only overlaps those chains, fusing pays for one.  Sort and accumulate become one large launch each.
static int msm_fused_chunk(Ctx* c, const Fe* const* d_scalars, bool scalars_on_host, const Affine* d_bases, size_t n, size_t count, XYZZ* h_out,
                           hipStream_t s) {
    MsmLayout L;
    int rc = msm_layout(n, s, &L, (uint32_t)count);
    if (rc) return rc;
    if ((rc = c->msm_slot[0].ensure(L.total))) return rc;
    std::vector<const Fe*> list(d_scalars, d_scalars + count);
    if ((rc = c->ws_acquire(s))) return rc;
    if (scalars_on_host) {
        if ((rc = c->msm_scalars[0].ensure(count * n * sizeof(Fe)))) return rc;
        for (size_t j = 0; j < count; j++) {
            Fe* dst = (Fe*)c->msm_scalars[0].p + j * n;
            H2_CHECK(hipMemcpyAsync(dst, d_scalars[j], n * sizeof(Fe), hipMemcpyHostToDevice, s));
            list[j] = dst;
        }
    }
    if ((rc = c->host_ws.ensure((size_t)L.Wt * sizeof(XYZZ)))) return rc;
    XYZZ* h_ws = (XYZZ*)c->host_ws.p;
    char* base = (char*)c->msm_slot[0].p;
    int t_all = c->timer_begin("msm_total", s);
    if ((rc = msm_stage_a(c, L, base, list.data(), s))) return rc;
    if ((rc = msm_stage_b(c, L, base, d_bases, s))) return rc;
    if ((rc = msm_stage_c(c, L, base, h_ws, s))) return rc;
    c->timer_end(t_all, s);
    H2_CHECK(hipStreamSynchronize(s));
    for (size_t j = 0; j < count; j++) h_out[j] = combine_windows(h_ws + j * L.p.W, L.p);
    return c->ws_release(s);
}

static bool g_fuse_small = true;
void msm_set_fuse_small(bool on) { g_fuse_small = on; }

// count MSMs over the same bases for device-resident inputs; results (XYZZ) to host memory.
int msm_batch_device(Ctx* c, const Fe* const* d_scalars, bool scalars_on_host, const Affine* d_bases, size_t n, size_t count, XYZZ* h_out,
                     hipStream_t s) {
    for (size_t j = 0; j < count; j++) h_out[j] = xyzz_identity();
    if (n == 0 || count == 0) return 0;
    // the pair index lives in 31 bits and the sort counts in int: split very large inputs
    const size_t max_chunk = g_max_chunk;
    std::vector<const Fe*> ptrs(count);
    std::vector<XYZZ> part(count);
    for (size_t o = 0; o < n; o += max_chunk) {
        size_t m = n - o < max_chunk ? n - o : max_chunk;
        for (size_t j = 0; j < count; j++) ptrs[j] = d_scalars[j] + o;
        // up to 2^18 pairs each: fused runs of at most 2^26 (pair, window) entries; larger MSMs: pipelined over streams
        const size_t per_msm = m * make_plan(m, true).W;
        const size_t fuse_max = per_msm ? ((size_t)1 << 26) / per_msm : 0;
        if (g_fuse_small && count > 1 && m <= ((size_t)1 << 18) && fuse_max >= 2) {
            for (size_t j0 = 0; j0 < count; j0 += fuse_max) {
                const size_t g = count - j0 < fuse_max ? count - j0 : fuse_max;
                int rc = msm_fused_chunk(c, ptrs.data() + j0, scalars_on_host, d_bases + o, m, g, part.data() + j0, s);
                if (rc) return rc;
            }
        } else {
            int rc = msm_batch_chunk(c, ptrs.data(), scalars_on_host, d_bases + o, m, count, part.data(), s);
            if (rc) return rc;
        }
        for (size_t j = 0; j < count; j++) xyzz_add(h_out[j], part[j]);
    }
    return 0;
}

// Sum of coeffs[i]*bases[i] for device-resident inputs; result (XYZZ) to host memory.
int msm_device(Ctx* c, const Fe* d_scalars, const Affine* d_bases, size_t n, XYZZ* h_out, hipStream_t s) {
    return msm_batch_device(c, &d_scalars, false, d_bases, n, 1, h_out, s);
}

}  // namespace h2
