// msm.hip -- Pippenger bucket MSM over BN254 G1 for gfx950.
//
// Computes the group element halo2_proofs::arithmetic::best_multiexp returns
// (halo2_proofs/src/arithmetic.rs:132-159, inner loop multiexp_serial :13-101):
// sum_i coeffs[i] * bases[i].  The reference splits the pairs over rayon threads and runs an
// unsigned-window bucket method per chunk; its Jacobian coordinates depend on the thread count,
// only the group element is defined (SURVEY.md App. B rule 3), and that is what this engine
// reproduces bit-exactly after normalisation to affine.
//
// Two forms of the same pipeline:
//   plain        the W windows of an MSM have one bucket set each (2^(c-1) signed-digit buckets); the window sums
//                come back to the host, which finishes with the Horner of arithmetic.rs:46-49.
//   fixed-base   for bases that live as long as a ParamsKZG (g, g_lagrange: poly/kzg/commitment.rs:26-27) the
//                engine keeps the table 2^(c j) * P_i, j < W, in HBM (built once by h2hip_bases_pin*).  Digit j of
//                scalar i then selects table[j][i], every window adds into the SAME bucket set, the windows can
//                be wide (c = 20 at 2^20 pairs: W = 13 instead of 16 bucket additions per pair) and the one set
//                sum IS the result: no Horner.
//
// GPU schedule (one stream, no host sync until the set sums come back):
//   A  msm_l1_count_kernel     to_repr + get_at (arithmetic.rs:14,24-42): Montgomery -> canonical, signed c-bit digits; one
//                              workgroup per tile of <= 1024 scalars leaves the tile's histogram over the coarse bins
//                              (bucket id >> L) in HBM -- no global atomics anywhere in level 1
//      msm_l1_chunk / _scan / _offsets   the [tiles][bins] matrix -> every (tile, bin)'s offset; the scan kernel also clears
//                              the run's counters
//      msm_l1_scatter_kernel   digits again; the tile's entries sorted by bin in LDS, every bin's share written as one run of
//                              (entry, bucket id) records (or entries / key bits apart when bins span several level-2 tiles)
//      msm_l2_kernel           one workgroup per coarse bin: LDS counting sort on the low L bits -> entries grouped
//                              by bucket, bucket start / size, size-class histogram
//      msm_bucket_scatter      buckets ordered by size, so the 64 lanes of a wave get equal-length buckets
//   B  msm_accum_kernel        one lane per bucket: XYZZ mixed adds over its entries (arithmetic.rs:84-89); cont = 1: into the
//                              sums the earlier chunks of a streamed host-slice MSM left there
//      msm_heavy_*             workgroup-per-chunk accumulation + LDS tree for over-full buckets
//   C  msm_rowcol_*_kernel     summation by parts (arithmetic.rs:95-99) restated as plain sums: with the bucket
//                              index b = hi * 2^s + lo,  sum (b+1) B_b = 2^s sum hi R_hi + sum (lo+1) C_lo  for the
//                              row sums R and column sums C; applied twice (first pass: msm_rowcol_qtree_kernel,
//                              lane chains then a tree of quad-cooperative additions), then
//      msm_final_quad_kernel   <= 256 small multiples + trees per bucket set; the last workgroup to arrive sums the partials
//                              (msm_final_kernel: the lane-per-operation form for runs with many sets)
//   D  host                    plain form only: Horner over the W set sums with c doublings each
// Host-resident scalars (the call best_multiexp makes) stream in under stages A and B: msm_stream_host, msm_fused_groups_host.
#include <hip/hip_ext.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <thread>
#include <vector>

#include "ecq.h"
#include "engine.h"
#include "host64.h"

namespace h2 {

#define MSM_MAX_C1 1024u   // coarse bins of the level-1 pass
#define MSM_MAX_L 12u      // low bucket-id bits sorted by the level-2 pass (LDS histogram of 2^L counters)
#define MSM_STAGE 16384u   // entries of one LDS-staged tile (level 1: 128 KB of (entry, bucket); level 2: 96 KB)
#define MSM_COUNT_TILE 4096u  // scalars per workgroup of the count pass

// Windows: the 255 bits a signed-digit recoding of a scalar < 2^254 needs are cut into W = ceil(255 / c) windows, the
// first q of them c bits wide and the rest c - 1, q = 255 - W (c - 1).  With equal widths the top window would hold only
// 254 mod c bits: its n digits would crowd into 2^(254 mod c) buckets (64 extra entries in each of 2^14 buckets at
// c = 20, n = 2^20), and the lanes of those buckets would set the duration of the accumulation.
struct MsmPlan {
    uint32_t c;        // window bits (wide windows)
    uint32_t W;        // number of windows
    uint32_t q;        // windows 0..q-1 are c bits wide, windows q..W-1 are c - 1
    uint32_t cb;       // c - 1
    uint32_t NB;       // buckets per set = 2^(c-1)
    uint32_t shared;   // fixed-base form: the windows share one bucket set
    uint32_t heavy_t;  // bucket size above which the chunked path is used
    uint32_t chunk;    // entries per heavy chunk
};

// get_at (arithmetic.rs:24-42) with signed digits: f(window, |digit| - 1, negative) for every non-zero digit.
// The limbs are walked with compile-time indices through a 64-bit bit buffer (a run-time limb index would put the
// scalar into scratch memory); c <= 24, so the buffer never holds more than 23 + 32 bits.
template <class F>
__device__ __forceinline__ void for_each_digit(const Fe& s, uint32_t c, uint32_t q, uint32_t W, F&& f) {
    uint64_t buf = 0;
    uint32_t nb = 0, w = 0, carry = 0;
    uint32_t width = q ? c : c - 1;
    auto emit = [&]() {
        uint32_t d = ((uint32_t)buf & ((1u << width) - 1)) + carry;
        uint32_t neg = 0;
        carry = 0;
        if (d > (1u << (width - 1))) {
            d = (1u << width) - d;
            neg = 1;
            carry = 1;
        }
        if (d) f(w, d - 1, neg);
        w++;
        buf >>= width;
        nb = nb > width ? nb - width : 0;
        width = w < q ? c : c - 1;
    };
#pragma unroll
    for (int limb = 0; limb < 8; limb++) {
        buf |= (uint64_t)s.l[limb] << nb;
        nb += 32;
        while (nb >= width && w < W) emit();
    }
    while (w < W) emit();  // the bits left over (zero-extended) and any windows above them
}

struct L1Args {
    const Fe* scalars_one;
    const Fe* const* list;  // fused batch: MSM y reads list[y]
    uint32_t n, c, q, W, cb, L, C1;
    uint32_t shared, stride;  // fixed-base: entry index = window * stride + pair
    uint32_t tile;                // scalars per tile: one workgroup of the count pass and one of the scatter pass each
    uint16_t* tile_hist;          // [tiles][C1] entries of the tile per coarse bin (count pass writes, scatter pass reads)
    const uint32_t* tile_off;     // [tiles][C1] where the tile's run of each bin starts in tmp (scatter pass)
    uint2* tmp;                   // (entry, bucket id) grouped by coarse bin
    uint32_t* tmp_e;              // split records (runs whose bins span several level-2 tiles): the entries ...
    uint16_t* tmp_k;              // ... and the low L bits of their bucket ids, apart
    // device-key pin guard (api.hip msm_device_keyed): the count pass's first workgroup compares the sampled points
    const Affine* chk_bases;
    const Affine* chk_samples;
    uint32_t* chk_flag;           // null: no check
    size_t chk_n, chk_total;
};

// PrimeField::to_repr (arithmetic.rs:14) on the unsaturated multiplier: the stored a * 2^256 times 32, divided by the
// Montgomery radix 2^261 and reduced exactly -- ~220 instructions instead of the ~535 of the saturated CIOS
__device__ __forceinline__ Fe fr_to_canonical(const Fe& x) {
    Fu c32 = fu_zero();
    c32.l[0] = 32;
    return fu_mul_canon<FrU>(fu_slice(x), c32);
}

// Level 1 is a counting sort by coarse bin (bucket id >> L) WITHOUT global atomics: the scalars are cut into tiles (one
// workgroup of the count pass and one of the scatter pass each); the count pass leaves every tile's histogram in HBM, three
// small kernels turn the [tiles][C1] matrix into every (tile, bin)'s offset (column-wise exclusive sums: tiles in chunks of
// MSM_TCHUNK, then the chunks, then the bins), and the scatter pass writes its runs there.  With one atomic per (workgroup,
// bin) instead -- non-returning in the count pass, returning in the scatter pass, 2 x 1 M of them at 2^20 pairs -- the atomics
// WERE the two passes: they top out near 60 and 20 G/s chip-wide.  Bucket id of (MSM y, window w, slot) =
// (set << cb) | slot with set = y (fixed-base) or y * W + w.
#define MSM_TCHUNK 64u  // tiles per chunk of the offset computation
#ifndef MSM_SCATTER_XG
#define MSM_SCATTER_XG 4u  // log2 of the tiles of the scatter pass that share an XCD (0: workgroups take the tiles in order); A/B builds: 2, 3, 4, 6 -> sort of 2^24 pairs 1.89, 1.87, 1.85, 1.84 ms (2.02 in order)
#endif

// A1: histogram of the tile over the coarse bins
__global__ void __launch_bounds__(256) msm_l1_count_kernel(L1Args a) {
    __shared__ uint32_t lh[MSM_MAX_C1];
    if (a.chk_flag && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 4 * H2_PIN_SAMPLES) {  // 16 points x 4 pieces of 16 B
        const uint32_t k = threadIdx.x >> 2, part = threadIdx.x & 3;
        const size_t idx = pin_sample_index(a.chk_total, k);
        if (idx < a.chk_n) {
            const uint4 u = reinterpret_cast<const uint4*>(&a.chk_bases[idx])[part], v = reinterpret_cast<const uint4*>(&a.chk_samples[k])[part];
            if (u.x != v.x || u.y != v.y || u.z != v.z || u.w != v.w) atomicOr(a.chk_flag, 1u);
        }
    }
    for (uint32_t b = threadIdx.x; b < a.C1; b += 256) lh[b] = 0;
    __syncthreads();
    const Fe* __restrict__ scalars = a.list ? a.list[blockIdx.y] : a.scalars_one;
    const uint32_t set0 = a.shared ? blockIdx.y : blockIdx.y * a.W;
    for (uint32_t t = threadIdx.x; t < a.tile; t += 256) {
        const uint32_t i = blockIdx.x * a.tile + t;
        if (i < a.n) {
            const Fe sc = fr_to_canonical(scalars[i]);
            for_each_digit(sc, a.c, a.q, a.W, [&](uint32_t w, uint32_t slot, uint32_t) {
                const uint32_t gb = ((set0 + (a.shared ? 0u : w)) << a.cb) | slot;
                atomicAdd(&lh[gb >> a.L], 1u);
            });
        }
    }
    __syncthreads();
    uint16_t* row = a.tile_hist + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * a.C1;
    for (uint32_t b = threadIdx.x; b < a.C1; b += 256) row[b] = (uint16_t)lh[b];  // a tile holds at most MSM_STAGE entries
}

// A2a: entries per (chunk of tiles, bin); lane = bin
__global__ void __launch_bounds__(256) msm_l1_chunk_kernel(const uint16_t* __restrict__ tile_hist, uint32_t n_tiles, uint32_t C1,
                                                           uint32_t* __restrict__ chunk_sum) {
    const uint32_t b = blockIdx.y * 256 + threadIdx.x;
    if (b >= C1) return;
    const uint32_t t0 = blockIdx.x * MSM_TCHUNK, t1 = t0 + MSM_TCHUNK < n_tiles ? t0 + MSM_TCHUNK : n_tiles;
    uint32_t sum = 0;
#pragma unroll 8
    for (uint32_t t = t0; t < t1; t++) sum += tile_hist[(size_t)t * C1 + b];
    chunk_sum[(size_t)blockIdx.x * C1 + b] = sum;
}

// exclusive scan of cnt[0..n) over a workgroup of BS lanes (n <= per * BS): lane t owns elements [t * per, (t + 1) * per).
// Returns this lane's first offset; ps is BS words of LDS scratch; *total gets the sum.
template <int BS>
__device__ __forceinline__ uint32_t block_scan_base(const uint32_t* cnt, uint32_t n, uint32_t per, uint32_t* ps, uint32_t* total) {
    const uint32_t t = threadIdx.x;
    uint32_t sum = 0;
    for (uint32_t j = 0; j < per; j++) {
        const uint32_t k = t * per + j;
        if (k < n) sum += cnt[k];
    }
    // inclusive scan inside the wave with lane shuffles, then over the BS / 64 wave totals through LDS
    uint32_t inc = sum;
#pragma unroll
    for (uint32_t off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(inc, off, 64);
        if ((t & 63) >= off) inc += up;
    }
    if ((t & 63) == 63) ps[t >> 6] = inc;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (uint32_t w = 0; w < BS / 64; w++) {
        const uint32_t v = ps[w];
        if (w < (t >> 6)) before += v;
        all += v;
    }
    __syncthreads();
    *total = all;
    return before + inc - sum;
}

// A3: one scalar per lane (the first a.tile lanes).  The workgroup sorts its entries by coarse bin in LDS and writes
// every bin's share as one contiguous run at the offset A2c computed for (tile, bin).  What bounds this pass is how HBM takes
// the writes: with 3-entry runs (or a scattered 8-byte store per entry) every run is a partial-line write and the pass ran at
// 0.6 TB/s; hence at most MSM_MAX_C1 bins for a tile of up to MSM_STAGE entries.
template <bool SPLIT>
__global__ void __launch_bounds__(1024) msm_l1_scatter_kernel(L1Args a) {
    __shared__ uint2 stage[MSM_STAGE];
    __shared__ uint32_t cur[MSM_MAX_C1];   // entries per bin, then the bin's cursor in `stage`
    __shared__ uint32_t delta[MSM_MAX_C1]; // (position in tmp) - (position in stage) of the bin's run (mod 2^32)
    __shared__ uint32_t ps[16];
    // consecutive tiles write consecutive runs of every bin: groups of 2^XG tiles are dealt to ONE XCD (workgroups b, b + 8, ...), so that the
    // cache lines two runs share meet in one L2
    constexpr uint32_t XG = MSM_SCATTER_XG;
    uint32_t tx = blockIdx.x;
    if (XG && tx < (gridDim.x & ~((8u << XG) - 1))) {
        const uint32_t q = tx >> (3 + XG), r = tx & ((8u << XG) - 1);
        tx = ((q * 8 + (r & 7)) << XG) | (r >> 3);
    }
    const size_t row = (size_t)(blockIdx.y * gridDim.x + tx) * a.C1;
    for (uint32_t b = threadIdx.x; b < a.C1; b += 1024) cur[b] = a.tile_hist[row + b];
    const Fe* __restrict__ scalars = a.list ? a.list[blockIdx.y] : a.scalars_one;
    const uint32_t set0 = a.shared ? blockIdx.y : blockIdx.y * a.W;
    const uint32_t i = tx * a.tile + threadIdx.x;
    const bool live = threadIdx.x < a.tile && i < a.n;
    Fe sc;
    if (live) sc = fr_to_canonical(scalars[i]);
    __syncthreads();
    uint32_t total;
    uint32_t run = block_scan_base<1024>(cur, a.C1, 1, ps, &total);  // C1 <= 1024: one bin per lane
    if (threadIdx.x < a.C1) {
        const uint32_t b = threadIdx.x;
        delta[b] = a.tile_off[row + b] - run;
        cur[b] = run;
    }
    __syncthreads();
    if (live) {
        for_each_digit(sc, a.c, a.q, a.W, [&](uint32_t w, uint32_t slot, uint32_t neg) {
            const uint32_t gb = ((set0 + (a.shared ? 0u : w)) << a.cb) | slot;
            const uint32_t pos = atomicAdd(&cur[gb >> a.L], 1u);
            stage[pos] = make_uint2((a.shared ? w * a.stride + i : i) | (neg << 31), gb);
        });
    }
    __syncthreads();
    for (uint32_t pos = threadIdx.x; pos < total; pos += 1024) {
        const uint2 e = stage[pos];
        const uint32_t at = pos + delta[e.y >> a.L];
        if (SPLIT) {
            a.tmp_e[at] = e.x;
            a.tmp_k[at] = (uint16_t)(e.y & ((1u << a.L) - 1));
        } else {
            a.tmp[at] = e;
        }
    }
}

// A2b: exclusive sums down the chunks (in place) and the total of every bin.  Workgroup = 64 bins x 16 parts of the chunk list
// (a lone lane per bin walking 256 chunks at 2^24 pairs took 39 us).
// Its first workgroup also clears the run's counters (size-class histogram, heavy counters, final-stage arrival counters): no
// memset launch.
__global__ void __launch_bounds__(1024) msm_l1_scan_kernel(uint32_t* __restrict__ chunk_sum, uint32_t n_chunks, uint32_t C1,
                                                           uint32_t* __restrict__ bin_total, uint32_t* __restrict__ zero, uint32_t zero_words) {
    __shared__ uint32_t part_sum[16][64];
    if (blockIdx.x == 0)
        for (uint32_t i = threadIdx.x; i < zero_words; i += 1024) zero[i] = 0;
    const uint32_t lb = threadIdx.x & 63, p = threadIdx.x >> 6, b = blockIdx.x * 64 + lb;
    const uint32_t per = (n_chunks + 15) / 16, c0 = p * per, c1 = c0 + per < n_chunks ? c0 + per : n_chunks;
    uint32_t sum = 0;
    if (b < C1)
        for (uint32_t ch = c0; ch < c1; ch++) sum += chunk_sum[(size_t)ch * C1 + b];
    part_sum[p][lb] = sum;
    __syncthreads();
    uint32_t acc = 0, all = 0;
#pragma unroll
    for (uint32_t q = 0; q < 16; q++) {
        const uint32_t v = part_sum[q][lb];
        if (q < p) acc += v;
        all += v;
    }
    if (b < C1) {
        for (uint32_t ch = c0; ch < c1; ch++) {
            const uint32_t v = chunk_sum[(size_t)ch * C1 + b];
            chunk_sum[(size_t)ch * C1 + b] = acc;
            acc += v;
        }
        if (p == 0) bin_total[b] = all;
    }
}

// A2c: tile_off[t][b] = coarse_start[b] + (entries of bin b in the tiles before t); lane = bin, one chunk of tiles per workgroup.
// coarse_start = exclusive scan of the C1 <= 1024 bin totals, which every workgroup forms for itself in LDS (the first column
// of workgroups also writes it out for level 2).
__global__ void __launch_bounds__(256) msm_l1_offsets_kernel(const uint16_t* __restrict__ tile_hist, uint32_t n_tiles, uint32_t C1,
                                                             const uint32_t* __restrict__ chunk_sum, const uint32_t* __restrict__ bin_total,
                                                             uint32_t* __restrict__ coarse_start, uint32_t* __restrict__ tile_off) {
    __shared__ uint32_t cs[MSM_MAX_C1], ps[4];
    const uint32_t per = (C1 + 255) / 256;
    uint32_t total;
    uint32_t run = block_scan_base<256>(bin_total, C1, per, ps, &total);
    for (uint32_t j = 0; j < per; j++) {
        const uint32_t k = threadIdx.x * per + j;
        if (k < C1) {
            cs[k] = run;
            run += bin_total[k];
        }
    }
    __syncthreads();
    const uint32_t b = blockIdx.y * 256 + threadIdx.x;
    if (b >= C1) return;
    if (blockIdx.x == 0) {
        coarse_start[b] = cs[b];
        if (b == 0) coarse_start[C1] = total;
    }
    const uint32_t t0 = blockIdx.x * MSM_TCHUNK, t1 = t0 + MSM_TCHUNK < n_tiles ? t0 + MSM_TCHUNK : n_tiles;
    uint32_t running = cs[b] + chunk_sum[(size_t)blockIdx.x * C1 + b];
#pragma unroll 8
    for (uint32_t t = t0; t < t1; t++) {
        tile_off[(size_t)t * C1 + b] = running;
        running += tile_hist[(size_t)t * C1 + b];
    }
}

// size classes of the accumulate order: 256 classes of width 2^bin_shift entries, big buckets first
__device__ __forceinline__ uint32_t size_class(uint32_t cnt, uint32_t bin_shift) {
    uint32_t b = cnt >> bin_shift;
    return 255u - (b < 255u ? b : 255u);
}

// A4: one workgroup per coarse bin -- counting sort of its entries on the low L bits of the bucket id, through LDS, one
// tile of MSM_STAGE entries at a time (16 per lane, held in registers between the tile's count and its placement);
// every bucket's share of a tile leaves as one contiguous run.  A bin of at most one tile is read from HBM once, a
// larger one twice (count, place).  Also writes, for the bin's 2^L buckets: start / counts (empty buckets included) and
// the accumulate order -- the bin's buckets by descending size class, so that the 64 lanes of an accumulate wave (which
// takes 64 consecutive entries of perm) get buckets of equal length.
// SPLIT: the records come as two arrays (entries, low key bits): a bin of several tiles is read twice -- once for its buckets'
// sizes, once to place -- and the first read then takes 2 instead of 8 bytes per entry (at 2^24 pairs: 0.4 instead of 1.6 GB).
template <bool SPLIT>
__global__ void __launch_bounds__(1024) msm_l2_kernel(const uint2* __restrict__ tmp, const uint32_t* __restrict__ tmp_e, const uint16_t* __restrict__ tmp_k,
                                                      const uint32_t* __restrict__ coarse_start, uint32_t L,
                                                      uint32_t bin_shift, uint32_t* __restrict__ vals, uint32_t* __restrict__ start,
                                                      uint32_t* __restrict__ counts, uint32_t* __restrict__ perm, uint32_t* __restrict__ class_hist) {
    constexpr uint32_t NBMAX = 1u << MSM_MAX_L, EPT = MSM_STAGE / 1024;
    __shared__ uint32_t sval[MSM_STAGE];
    __shared__ uint16_t skey[MSM_STAGE];
    __shared__ uint32_t gpos[NBMAX];    // where the bucket's next entry goes in vals
    __shared__ uint32_t tcur[NBMAX];    // entries of the tile per bucket, then the bucket's cursor in the staged tile
    __shared__ uint32_t tdelta[NBMAX];  // (position in vals) - (position in the staged tile) of the bucket's run
    __shared__ uint32_t ps[16], cls[256];
    const uint32_t nb = 1u << L, t = threadIdx.x, per = (nb + 1023) / 1024;
    const uint32_t cs = coarse_start[blockIdx.x], ce = coarse_start[blockIdx.x + 1];
    const bool single = ce - cs <= MSM_STAGE;
    // The loads of a phase are all issued before the first LDS atomic that needs one of them: written load-then-count per record, the
    // compiler kept them in program order and every record paid a full memory latency (measured with clock64 in round 3: 15 us of a
    // 29-us single-tile workgroup at 2^20, and 75 % of a 0.6-ms twelve-tile workgroup at 2^24, went into these waits).
    auto load_tile = [&](uint2 (&e)[EPT], uint32_t first, uint32_t tile_n) {
#pragma unroll
        for (uint32_t j = 0; j < EPT; j++) {
            const uint32_t idx = t + j * 1024;
            if (idx < tile_n) e[j] = SPLIT ? make_uint2(tmp_e[first + idx], tmp_k[first + idx]) : tmp[first + idx];
        }
    };
    auto count_tile = [&](const uint2 (&e)[EPT], uint32_t tile_n) {
#pragma unroll
        for (uint32_t j = 0; j < EPT; j++)
            if (t + j * 1024 < tile_n) atomicAdd(&tcur[e[j].y & (nb - 1)], 1u);
    };
    uint2 e[EPT], e_next[EPT];
    const uint32_t first_n = ce - cs < MSM_STAGE ? ce - cs : MSM_STAGE;
    load_tile(e, cs, first_n);  // the first tile's records: in flight while the counters are cleared
    for (uint32_t k = t; k < nb; k += 1024) tcur[k] = 0;
    if (t < 256) cls[t] = 0;
    __syncthreads();
    if (single) {
        count_tile(e, first_n);
    } else {
        // sizes of the bin's buckets: the whole bin's keys, eight loads in flight per lane
        uint32_t idx = cs + t;
        for (; idx + 7 * 1024 < ce; idx += 8 * 1024) {
            uint32_t key[8];
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) key[j] = SPLIT ? (uint32_t)tmp_k[idx + j * 1024] : tmp[idx + j * 1024].y;
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) atomicAdd(&tcur[key[j] & (nb - 1)], 1u);
        }
        for (; idx < ce; idx += 1024) atomicAdd(&tcur[(SPLIT ? (uint32_t)tmp_k[idx] : tmp[idx].y) & (nb - 1)], 1u);
    }
    __syncthreads();
    uint32_t total;
    uint32_t run = block_scan_base<1024>(tcur, nb, per, ps, &total);
    const uint32_t gb0 = blockIdx.x << L;
    for (uint32_t j = 0; j < per; j++) {
        const uint32_t k = t * per + j;
        if (k < nb) {
            const uint32_t cnt = tcur[k];
            gpos[k] = cs + run;
            start[gb0 + k] = cs + run;
            counts[gb0 + k] = cnt;
            atomicAdd(&cls[size_class(cnt, bin_shift)], 1u);
            run += cnt;
        }
    }
    __syncthreads();
    if (class_hist) {
        if (t < 256 && cls[t]) atomicAdd(&class_hist[t], cls[t]);
    } else {
        // the bin's buckets in descending size-class order: exclusive scan of the 256 class counts (tdelta is free here)
        uint32_t dummy;
        const uint32_t c0 = block_scan_base<1024>(cls, 256, 1, ps, &dummy);
        if (t < 256) tdelta[t] = c0;
        __syncthreads();
        for (uint32_t j = 0; j < per; j++) {
            const uint32_t k = t * per + j;
            if (k < nb) perm[gb0 + atomicAdd(&tdelta[size_class(tcur[k], bin_shift)], 1u)] = gb0 + k;
        }
        __syncthreads();
    }
    for (uint32_t base = cs; base < ce; base += MSM_STAGE) {
        const uint32_t tile_n = ce - base < MSM_STAGE ? ce - base : MSM_STAGE;
        // the next tile's records are fetched while this one is counted, placed and written out
        const uint32_t next = base + MSM_STAGE;
        const uint32_t next_n = next < ce ? (ce - next < MSM_STAGE ? ce - next : MSM_STAGE) : 0;
        if (next_n) load_tile(e_next, next, next_n);
        if (!single) {
            for (uint32_t k = t; k < nb; k += 1024) tcur[k] = 0;
            __syncthreads();
            count_tile(e, tile_n);
            __syncthreads();
        }
        run = block_scan_base<1024>(tcur, nb, per, ps, &total);
        for (uint32_t j = 0; j < per; j++) {
            const uint32_t k = t * per + j;
            if (k < nb) {
                const uint32_t cnt = tcur[k];
                tdelta[k] = gpos[k] - run;
                gpos[k] += cnt;
                tcur[k] = run;
                run += cnt;
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < EPT; j++) {
            const uint32_t idx = t + j * 1024;
            if (idx < tile_n) {
                const uint32_t k = e[j].y & (nb - 1);
                const uint32_t pos = atomicAdd(&tcur[k], 1u);
                sval[pos] = e[j].x;
                skey[pos] = (uint16_t)k;
            }
        }
        __syncthreads();
        for (uint32_t pos = t; pos < tile_n; pos += 1024) vals[pos + tdelta[skey[pos]]] = sval[pos];
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < EPT; j++) e[j] = e_next[j];
    }
}

// A5: order the buckets by size (descending, 256 classes) with a counting sort.  hist[0..256) = class counts (from
// msm_l2_kernel), hist[256..512) = per-class cursors, zeroed beforehand.  1024 lanes x 4 buckets per workgroup: one
// returning atomic per (workgroup, class) on 256 addresses -- with 256-bucket workgroups those atomics, serialised per
// address, were the whole 31 us of this kernel at 2^19 buckets.
#define MSM_BS_PER 4u
// Over-full buckets (more than heavy_t entries: the zeros, ones and twos of a prover's column) are summed chunk by chunk by
// workgroups of their own (msm_accum_kernel's first `hgrid` workgroups).  The list of those buckets and of their chunks is made
// here, where every bucket's size passes by anyway; `arrived` counts a bucket's finished chunks.
struct HeavyBucket {
    uint32_t bucket, first_chunk, n_chunks, arrived;
};
struct HeavyChunk {
    uint32_t begin, end, owner;  // entries [begin, end) of heavy bucket number `owner`
};

__global__ void __launch_bounds__(1024) msm_bucket_scatter_kernel(const uint32_t* __restrict__ counts, const uint32_t* __restrict__ start, uint32_t n_buckets,
                                                                  uint32_t bin_shift, uint32_t heavy_t, uint32_t chunk, uint32_t* __restrict__ hist,
                                                                  uint32_t* __restrict__ perm, uint32_t* __restrict__ heavy_counts,
                                                                  HeavyBucket* __restrict__ heavy_buckets, HeavyChunk* __restrict__ heavy_chunks) {
    __shared__ uint32_t scan[256], lh[256], lbase[256], ps[16];
    const uint32_t t = threadIdx.x;
    for (uint32_t j = 0; j < MSM_BS_PER; j++) {
        const uint32_t b = (blockIdx.x * 1024 + t) * MSM_BS_PER + j;
        const uint32_t cnt = b < n_buckets ? counts[b] : 0;
        if (cnt > heavy_t) {
            const uint32_t s0 = start[b], e0 = s0 + cnt, nch = (cnt + chunk - 1) / chunk;
            const uint32_t slot = atomicAdd(&heavy_counts[1], nch);
            const uint32_t hb = atomicAdd(&heavy_counts[0], 1u);
            const HeavyBucket h = {b, slot, nch, 0};
            heavy_buckets[hb] = h;
            for (uint32_t q = 0; q < nch; q++) {
                const HeavyChunk ch = {s0 + q * chunk, (s0 + (q + 1) * chunk < e0) ? s0 + (q + 1) * chunk : e0, hb};
                heavy_chunks[slot + q] = ch;
            }
        }
    }
    if (!perm) return;  // bucket order local to the sort bins (debug knob): only the list above
    if (t < 256) lh[t] = 0;
    uint32_t total;
    const uint32_t excl = block_scan_base<1024>(hist, 256, 1, ps, &total);  // exclusive scan of the class counts (every block repeats it)
    if (t < 256) scan[t] = excl;
    __syncthreads();
    uint32_t cls[MSM_BS_PER], rank[MSM_BS_PER];
    const uint32_t b0 = (blockIdx.x * 1024 + t) * MSM_BS_PER;
#pragma unroll
    for (uint32_t j = 0; j < MSM_BS_PER; j++) {
        cls[j] = 0;
        rank[j] = 0;
        if (b0 + j < n_buckets) {
            cls[j] = size_class(counts[b0 + j], bin_shift);
            rank[j] = atomicAdd(&lh[cls[j]], 1u);
        }
    }
    __syncthreads();
    if (t < 256 && lh[t]) lbase[t] = atomicAdd(&hist[256 + t], lh[t]);
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < MSM_BS_PER; j++)
        if (b0 + j < n_buckets) perm[scan[cls[j]] + lbase[cls[j]] + rank[j]] = b0 + j;
}

// the heavy role of msm_accum_kernel (defined below, after the tree sums it uses)
__device__ __noinline__ void msm_heavy_role(const Affine* __restrict__ bases, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ heavy_counts,
                                            HeavyBucket* __restrict__ heavy_buckets, const HeavyChunk* __restrict__ heavy_chunks,
                                            XYZZu* __restrict__ chunk_sums, uint32_t split_log, uint32_t cont, XYZZu* __restrict__ parts, uint32_t hgrid,
                                            XYZZu* sh);

// B: 2^split_log lanes per bucket (one when split_log = 0): lane (bucket, sub) adds the bucket's entries sub, sub + S,
// sub + 2S, ... into parts[bucket * S + sub].  A run with few buckets (a lone MSM of <= 2^17 pairs over a window table
// has 2^16) would otherwise be one wave per SIMD walking ~30 dependent additions per lane.
// `bases` is the caller's point array (plain form) or the window table (fixed-base form).
#ifdef H2_ACCUM_TIMELINE  // measurement builds only (tools/accum_timeline.py): entry / exit time, first lane's bucket size and HW_ID of every accumulate wave
__device__ unsigned long long g_accum_tl[6 * 16384];
#endif
__global__ void __launch_bounds__(256, 4) msm_accum_kernel(const Affine* __restrict__ bases, const uint32_t* __restrict__ vals,
                                                        const uint32_t* __restrict__ start, const uint32_t* __restrict__ counts,
                                                        const uint32_t* __restrict__ perm, uint32_t n_buckets, uint32_t split_log, uint32_t heavy_t,
                                                        uint32_t cont, XYZZu* __restrict__ parts, uint32_t hgrid, const uint32_t* __restrict__ heavy_counts,
                                                        HeavyBucket* __restrict__ heavy_buckets, const HeavyChunk* __restrict__ heavy_chunks,
                                                        XYZZu* __restrict__ chunk_sums) {
    // one LDS block for both roles: the heavy role's 256 sums (27 KB) or the accumulate role's point buffer (16 KB); four workgroups per CU
    __shared__ __align__(16) unsigned char lds_raw[256 * sizeof(XYZZu)];
    if (blockIdx.x < hgrid) {
        msm_heavy_role(bases, vals, heavy_counts, heavy_buckets, heavy_chunks, chunk_sums, split_log, cont, parts, hgrid, reinterpret_cast<XYZZu*>(lds_raw));
        return;
    }
    uint32_t t = (blockIdx.x - hgrid) * blockDim.x + threadIdx.x;
    if ((t >> split_log) >= n_buckets) return;
    const uint32_t S = 1u << split_log, sub = t & (S - 1);
    const uint32_t b = perm[t >> split_log];  // buckets in descending size order
    uint32_t s = start[b], e = s + counts[b];
#ifdef H2_ACCUM_TIMELINE
    const unsigned long long tl0 = wall_clock64(), tc0 = clock64();  // constant 100 MHz / shader clock
    const uint32_t tl_cnt = e - s;
#endif
    XYZZu acc = xyzzu_identity();
    XYZZu* const mine = parts + (((size_t)b << split_log) + sub);
    bool store = !cont;  // cont: the parts hold the sums of the earlier chunks of a streamed MSM; a lane with nothing to add leaves its part alone
    if (e - s > heavy_t) {
        store = store && sub != 0;  // the bucket's first part belongs to the heavy role (which runs beside this lane); the others are identities
    } else if (s + sub < e) {
        // The point of the next entry is fetched before the current addition starts: with a window table the points
        // are gathers from hundreds of MB of HBM, and one addition (~2.3 k instructions) hides the whole miss.  The fetch
        // is an LDS-DMA (global_load_lds_dwordx4, per-lane source address, lane-linear destination): holding the next
        // point in registers instead costs 16 of them and with that the fourth wave per SIMD.
        typedef uint4 (*PBuf)[4][64];  // [wave][16-byte chunk of the point][lane]
        const PBuf pbuf = reinterpret_cast<PBuf>(lds_raw);
        const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        auto issue = [&](uint32_t v) {
            const char* g = reinterpret_cast<const char*>(&bases[v & 0x7fffffffu]);
#pragma unroll
            for (int ch = 0; ch < 4; ch++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 16 * ch),
                                                 (__attribute__((address_space(3))) void*)&pbuf[wave][ch][0], 16, 0, 0);
        };
        auto take = [&]() {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            Affine p;
#pragma unroll
            for (int ch = 0; ch < 4; ch++) {
                const uint4 q = pbuf[wave][ch][lane];
                Fe& f = ch < 2 ? p.x : p.y;
                f.l[4 * (ch & 1)] = q.x;
                f.l[4 * (ch & 1) + 1] = q.y;
                f.l[4 * (ch & 1) + 2] = q.z;
                f.l[4 * (ch & 1) + 3] = q.w;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the buffer is free for the next fetch
            return p;
        };
        uint32_t v = vals[s + sub];
        issue(v);
        if (cont) acc = *mine;
        store = true;
        for (uint32_t i = s + sub + S; i < e; i += S) {
            const uint32_t vn = vals[i];
            const Affine p = take();
            issue(vn);
            xyzzu_add_affine<FqUA>(acc, p, (v >> 31) != 0);  // throughput-bound: explicit-mad multiplier
            v = vn;
        }
        const Affine p = take();
        xyzzu_add_affine<FqUA>(acc, p, (v >> 31) != 0);
    }
    if (store) *mine = acc;
#ifdef H2_ACCUM_TIMELINE
    if ((threadIdx.x & 63) == 0) {
        const uint32_t w = (blockIdx.x - hgrid) * 4 + (threadIdx.x >> 6);
        if (w < 16384) {
            g_accum_tl[6 * w] = tl0;
            g_accum_tl[6 * w + 1] = wall_clock64();
            g_accum_tl[6 * w + 2] = tl_cnt;
            g_accum_tl[6 * w + 3] = __builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
            g_accum_tl[6 * w + 4] = tc0;
            g_accum_tl[6 * w + 5] = clock64();
        }
    }
#endif
}
#ifdef H2_ACCUM_TIMELINE
}  // namespace h2
extern "C" int h2hip_debug_accum_timeline(unsigned long long* out, size_t n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(h2::g_accum_tl), n * sizeof(unsigned long long));
}
namespace h2 {
#endif

// B': bucket = sum of its 2^split_log parts (msm_heavy_final has added an over-full bucket's sum to its first part by then).
// Half as many lanes as parts: every lane adds one pair, the pair sums of a bucket then meet in a tree through LDS -- log2(S)
// dependent general additions per bucket instead of S - 1 in one lane (2^16 buckets x 4 parts: 29 -> ~19 us).
__global__ void __launch_bounds__(256) msm_combine_kernel(const XYZZu* __restrict__ parts, uint32_t n_buckets, uint32_t split_log,
                                                          XYZZu* __restrict__ buckets) {
    __shared__ XYZZu sh[256];
    const uint32_t hl = split_log - 1, H = 1u << hl;  // lanes per bucket (split_log >= 1)
    const uint32_t b = (blockIdx.x * 256 + threadIdx.x) >> hl, j = threadIdx.x & (H - 1);
    XYZZu acc = xyzzu_identity();
    if (b < n_buckets) {
        const XYZZu* P = parts + ((size_t)b << split_log) + 2 * j;
        acc = P[0];
        xyzzu_add(acc, P[1]);
    }
    for (uint32_t st = H >> 1; st >= 1; st >>= 1) {
        sh[threadIdx.x] = acc;
        __syncthreads();
        if (j < st) xyzzu_add(acc, sh[threadIdx.x + st]);
        __syncthreads();
    }
    if (b < n_buckets && j == 0) buckets[b] = acc;
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

// the same sum with one QUAD of lanes per addition (ecq.h): 64 quads, 2 + 1 + ... + 1 = 9 dependent quad additions instead of 8
// lane additions at twice their latency (an over-full bucket's sum is nothing but such trees: 15 levels for 26 k entries)
__device__ __forceinline__ XYZZu block_tree_sum_q(const XYZZu& v, XYZZu* sh) {
    sh[threadIdx.x] = v;
    __syncthreads();
    const uint32_t quad = threadIdx.x >> 2, role = threadIdx.x & 3;
    for (uint32_t stride = 128; stride >= 1; stride >>= 1) {
        for (uint32_t a = quad; a < stride; a += 64) {
            const XYZZu x = xyzzu_sum_q(sh[a], sh[a + stride], role);
            if (role == 0) sh[a] = x;
        }
        __syncthreads();
    }
    return sh[0];
}

// The heavy role: the first `hgrid` workgroups of msm_accum_kernel stride over the chunk list (on uniform scalars it is empty and they
// leave at once).  A lane's entries (four of a 1024-entry chunk) are read up front and the next point is fetched before the current
// addition; the chunk's 256 lane sums meet in a tree of quad additions.  The workgroup that finishes a bucket's LAST chunk (arrival
// counter, as in msm_final_quad_kernel) sums the bucket's chunk sums and puts the total into the bucket's first part -- stored in a
// fresh run, added in a continued one; the accumulate lanes leave that part alone, so the two roles never touch the same word.
__device__ __noinline__ void msm_heavy_role(const Affine* __restrict__ bases, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ heavy_counts,
                                            HeavyBucket* __restrict__ heavy_buckets, const HeavyChunk* __restrict__ heavy_chunks,
                                            XYZZu* __restrict__ chunk_sums, uint32_t split_log, uint32_t cont, XYZZu* __restrict__ parts, uint32_t hgrid,
                                            XYZZu* sh) {
    __shared__ uint32_t last_flag;
    const uint32_t total = heavy_counts[1];
    for (uint32_t ci = blockIdx.x; ci < total; ci += hgrid) {
        const HeavyChunk ch = heavy_chunks[ci];
        XYZZu acc = xyzzu_identity();
        uint32_t i = ch.begin + threadIdx.x;
        if (i < ch.end) {
            uint32_t v = vals[i];
            Affine p = bases[v & 0x7fffffffu];
            for (i += blockDim.x; i < ch.end; i += blockDim.x) {
                const uint32_t vn = vals[i];
                const Affine pn = bases[vn & 0x7fffffffu];
                xyzzu_add_affine<FqU>(acc, p, (v >> 31) != 0);
                v = vn;
                p = pn;
            }
            xyzzu_add_affine<FqU>(acc, p, (v >> 31) != 0);
        }
        const XYZZu r = block_tree_sum_q(acc, sh);
        if (threadIdx.x == 0) {
            chunk_sums[ci] = r;
            __threadfence();
            last_flag = atomicAdd(&heavy_buckets[ch.owner].arrived, 1u) + 1 == heavy_buckets[ch.owner].n_chunks;
        }
        __syncthreads();
        if (last_flag) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const HeavyBucket h = heavy_buckets[ch.owner];
            XYZZu sum = xyzzu_identity();
            for (uint32_t q = threadIdx.x; q < h.n_chunks; q += blockDim.x) xyzzu_add(sum, chunk_sums[h.first_chunk + q]);
            const XYZZu tot = block_tree_sum_q(sum, sh);
            if (threadIdx.x == 0) {
                XYZZu* dst = parts + ((size_t)h.bucket << split_log);
                XYZZu cur = tot;
                if (cont) {
                    cur = *dst;
                    xyzzu_add(cur, tot);
                }
                *dst = cur;
            }
        }
        __syncthreads();
    }
}

// C: summation by parts (arithmetic.rs:95-99) as plain sums.  For an array X of rows x cols elements (cols = 2^t)
// with weights (i + o) << k on element i = h * cols + l:
//     sum_i ((i + o) << k) X_i = sum_h (h << (k + t)) R_h + sum_l ((l + o) << k) C_l,   R_h = sum_l X, C_l = sum_h X
// so a weighted sum over N elements becomes two over sqrt(N) after one pass of N-term plain sums (2 additions per
// element, every one of them independent of the others -- no lane runs a long chain).  The buckets of a set carry
// (o, k) = (1, 0); two such passes leave at most four arrays of <= 64 elements per set, which msm_final_kernel finishes.
struct RowColJob {
    const XYZZu* in;
    XYZZu* out;           // row sums [n_arr][rows] or column sums [n_arr][cols]
    uint32_t n_arr, log_rows, log_cols;
    uint32_t cols_kind;   // 0: this job forms the row sums, 1: the column sums
    uint32_t g_log;       // 2^g_log lanes share one sum
    uint32_t first_block; // this job's first workgroup
};
struct RowColArgs {
    RowColJob job[4];
    uint32_t n_jobs;
};

template <class F>
__global__ void __launch_bounds__(256) msm_rowcol_kernel(RowColArgs args) {
    __shared__ XYZZu sh[256];
    uint32_t ji = 0;
    for (uint32_t q = 1; q < args.n_jobs; q++)
        if (blockIdx.x >= args.job[q].first_block) ji = q;
    const RowColJob& J = args.job[ji];
    const uint32_t G = 1u << J.g_log, g = threadIdx.x & (G - 1);
    const uint32_t rows = 1u << J.log_rows, cols = 1u << J.log_cols;
    const uint32_t per_arr = J.cols_kind ? cols : rows;
    const uint32_t task = (blockIdx.x - J.first_block) * (256u >> J.g_log) + (threadIdx.x >> J.g_log);
    const bool live = task < J.n_arr * per_arr;
    XYZZu acc = xyzzu_identity();
    if (live) {
        const uint32_t a = task / per_arr, idx = task - a * per_arr;
        const XYZZu* X = J.in + ((size_t)a << (J.log_rows + J.log_cols));
        // element i of this lane's sum sits at first + i * step; the next one is fetched before the current addition
        const XYZZu* first = J.cols_kind ? X + idx + ((size_t)g << J.log_cols) : X + ((size_t)idx << J.log_cols) + g;
        const size_t step = J.cols_kind ? ((size_t)G << J.log_cols) : (size_t)G;
        const uint32_t terms = J.cols_kind ? rows : cols;
        if (g < terms) {
            XYZZu cur = first[0];
            for (uint32_t i = g + G; i < terms; i += G) {
                first += step;
                const XYZZu nxt = first[0];
                xyzzu_add<F>(acc, cur);
                cur = nxt;
            }
            xyzzu_add<F>(acc, cur);
        }
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t stride = G >> 1; stride >= 1; stride >>= 1) {
        if (g < stride) {
            XYZZu x = sh[threadIdx.x];
            xyzzu_add<F>(x, sh[threadIdx.x + stride]);
            sh[threadIdx.x] = x;
        }
        __syncthreads();
    }
    if (live && g == 0) J.out[task] = sh[threadIdx.x];
}

// The same sums for the first pass over a single bucket set (the fixed-base form's 2^19..2^21 buckets), where the work is
// throughput -- 2 general additions per bucket -- and the lane kernel above loses a third of it to five tree levels on mostly
// idle lanes at one wave per SIMD: here every lane chains only a few terms (four waves per SIMD reach the issue rate a lone wave
// cannot), and the 2^g_log partials of a sum are then added pairwise by QUADS of lanes (ecq.h: a third of a lane's latency per
// addition), 64 quads per workgroup, level by level through LDS.
template <class F>
__global__ void __launch_bounds__(256) msm_rowcol_qtree_kernel(RowColArgs args) {
    __shared__ XYZZu sh[256];
    uint32_t ji = 0;
    for (uint32_t q = 1; q < args.n_jobs; q++)
        if (blockIdx.x >= args.job[q].first_block) ji = q;
    const RowColJob& J = args.job[ji];
    const uint32_t G = 1u << J.g_log, g = threadIdx.x & (G - 1);
    const uint32_t rows = 1u << J.log_rows, cols = 1u << J.log_cols;
    const uint32_t per_arr = J.cols_kind ? cols : rows;
    const uint32_t task = (blockIdx.x - J.first_block) * (256u >> J.g_log) + (threadIdx.x >> J.g_log);
    const bool live = task < J.n_arr * per_arr;
    XYZZu acc = xyzzu_identity();
    if (live) {
        const uint32_t a = task / per_arr, idx = task - a * per_arr;
        const XYZZu* X = J.in + ((size_t)a << (J.log_rows + J.log_cols));
        const XYZZu* first = J.cols_kind ? X + idx + ((size_t)g << J.log_cols) : X + ((size_t)idx << J.log_cols) + g;
        const size_t step = J.cols_kind ? ((size_t)G << J.log_cols) : (size_t)G;
        const uint32_t terms = J.cols_kind ? rows : cols;
        if (g < terms) {
            XYZZu cur = first[0];
            for (uint32_t i = g + G; i < terms; i += G) {
                first += step;
                const XYZZu nxt = first[0];
                xyzzu_add<F>(acc, cur);
                cur = nxt;
            }
            xyzzu_add<F>(acc, cur);
        }
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    const uint32_t quad = threadIdx.x >> 2, role = threadIdx.x & 3;
    for (uint32_t stride = G >> 1; stride >= 1; stride >>= 1) {
        // this level adds slot i + stride into slot i for the first `stride` slots of every sum: (256 / G) * stride additions
        const uint32_t log_st = 31 - __clz(stride);
        const uint32_t n_adds = (256u >> J.g_log) << log_st;
        for (uint32_t a = quad; a < n_adds; a += 64) {
            const uint32_t i = ((a >> log_st) << J.g_log) + (a & (stride - 1));
            const XYZZu x = xyzzu_sum_q(sh[i], sh[i + stride], role);  // operands straight from LDS: a local copy of either went to scratch
            if (role == 0) sh[i] = x;
        }
        __syncthreads();
    }
    if (live && g == 0) J.out[task] = sh[threadIdx.x];
}

struct FinalArr {
    const XYZZu* base;  // array of set 0
    uint32_t stride;    // elements between consecutive sets
    uint32_t len;       // <= 64
    uint32_t o, k;      // weight of element v: (v + o) << k
    uint32_t nbits;     // bits of the largest weight
};
struct FinalArgs {
    FinalArr arr[4];
    uint32_t n_arr;
};

// one workgroup of 256 per bucket set: lane (array, v) scales its element (double-and-add), one tree sums everything
__global__ void __launch_bounds__(256) msm_final_kernel(FinalArgs args, XYZZ* __restrict__ set_sums) {
    __shared__ XYZZu sh[256];
    const uint32_t set = blockIdx.x, ai = threadIdx.x >> 6, v = threadIdx.x & 63;
    XYZZu x = xyzzu_identity();
    if (ai < args.n_arr) {
        const FinalArr& A = args.arr[ai];
        if (v < A.len) {
            const uint32_t wgt = (v + A.o) << A.k;
            if (wgt) x = xyzzu_mul_small(A.base[(size_t)set * A.stride + v], wgt, A.nbits);
        }
    }
    XYZZu r = block_tree_sum(x, sh);
    if (threadIdx.x == 0) set_sums[set] = xyzzu_to_ext(r);  // canonical E-form
}

// ---- the same two kernels with one QUAD of lanes per group operation (ecq.h), for runs with few bucket sets, whose
// ---- reduction tail is a latency chain on a handful of waves.  Workgroups are single waves so that the chains spread
// ---- over the chip's SIMDs instead of sharing one CU's issue slots.

// row / column sums, 16 quads per workgroup, 2^g_log quads per sum
__global__ void __launch_bounds__(64) msm_rowcol_quad_kernel(RowColArgs args) {
    __shared__ XYZZu sh[16];
    uint32_t ji = 0;
    for (uint32_t q = 1; q < args.n_jobs; q++)
        if (blockIdx.x >= args.job[q].first_block) ji = q;
    const RowColJob& J = args.job[ji];
    const uint32_t quad = threadIdx.x >> 2, role = threadIdx.x & 3;
    const uint32_t G = 1u << J.g_log, g = quad & (G - 1);
    const uint32_t rows = 1u << J.log_rows, cols = 1u << J.log_cols;
    const uint32_t per_arr = J.cols_kind ? cols : rows;
    const uint32_t task = (blockIdx.x - J.first_block) * (16u >> J.g_log) + (quad >> J.g_log);
    const bool live = task < J.n_arr * per_arr;
    XYZZu acc = xyzzu_identity();
    if (live) {
        const uint32_t a = task / per_arr, idx = task - a * per_arr;
        const XYZZu* X = J.in + ((size_t)a << (J.log_rows + J.log_cols));
        const XYZZu* first = J.cols_kind ? X + idx + ((size_t)g << J.log_cols) : X + ((size_t)idx << J.log_cols) + g;
        const size_t step = J.cols_kind ? ((size_t)G << J.log_cols) : (size_t)G;
        const uint32_t terms = J.cols_kind ? rows : cols;
        for (uint32_t i = g; i < terms; i += G) {
            xyzzu_add_q(acc, first[0], role);
            first += step;
        }
    }
    if (role == 0) sh[quad] = acc;
    __syncthreads();
    for (uint32_t stride = G >> 1; stride >= 1; stride >>= 1) {
        if (g < stride) {
            XYZZu x = sh[quad];
            xyzzu_add_q(x, sh[quad + stride], role);
            if (role == 0) sh[quad] = x;
        }
        __syncthreads();
    }
    if (live && g == 0 && role == 0) J.out[task] = sh[quad];
}

// final: workgroup (set, group) scales 16 of the set's remaining points (one per quad) and sums them; the workgroup that
// arrives last for its set (arrival counter, cleared with the run's other counters) then sums the set's <= 16 partials -- the
// second launch this used to be cost ~10 us of a lone MSM's latency chain.
__global__ void __launch_bounds__(64) msm_final_quad_kernel(FinalArgs args, uint32_t n_groups, XYZZu* __restrict__ partials, uint32_t* __restrict__ done,
                                                            XYZZ* __restrict__ set_sums) {
    __shared__ XYZZu sh[16];
    __shared__ uint32_t is_last;
    const uint32_t set = blockIdx.x / n_groups, group = blockIdx.x - set * n_groups;
    const uint32_t quad = threadIdx.x >> 2, role = threadIdx.x & 3;
    uint32_t e = group * 16 + quad;  // index into the set's arrays laid end to end
    // locate the array with selects (a run-time index into the kernel argument would go through scratch)
    const XYZZu* base = nullptr;
    uint32_t stride = 0, o = 0, k = 0, nbits = 0;
    bool found = false;
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
        if (i < args.n_arr && !found) {
            if (e < args.arr[i].len) {
                found = true;
                base = args.arr[i].base;
                stride = args.arr[i].stride;
                o = args.arr[i].o;
                k = args.arr[i].k;
                nbits = args.arr[i].nbits;
            } else {
                e -= args.arr[i].len;
            }
        }
    }
    XYZZu x = xyzzu_identity();
    if (found) {
        const uint32_t wgt = (e + o) << k;
        if (wgt) x = xyzzu_mul_small_q(base[(size_t)set * stride + e], wgt, nbits, role);
    }
    if (role == 0) sh[quad] = x;
    __syncthreads();
    for (uint32_t st = 8; st >= 1; st >>= 1) {
        if (quad < st) {
            XYZZu a = sh[quad];
            xyzzu_add_q(a, sh[quad + st], role);
            if (role == 0) sh[quad] = a;
        }
        __syncthreads();
    }
    if (n_groups == 1) {
        if (threadIdx.x == 0) set_sums[set] = xyzzu_to_ext(sh[0]);
        return;
    }
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = sh[0];
        __threadfence();
        is_last = atomicAdd(&done[set], 1u) == n_groups - 1;
    }
    __syncthreads();
    if (!is_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the other workgroups' partials, written before their arrivals
    x = xyzzu_identity();
    if (quad < n_groups) x = partials[(size_t)set * n_groups + quad];
    if (role == 0) sh[quad] = x;
    __syncthreads();
    for (uint32_t st = 8; st >= 1; st >>= 1) {
        if (quad < st) {
            XYZZu a = sh[quad];
            xyzzu_add_q(a, sh[quad + st], role);
            if (role == 0) sh[quad] = a;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) set_sums[set] = xyzzu_to_ext(sh[0]);
}

// ---- the tail of a run with few bucket sets (a lone commit), round 4: bit planes on the GPU, Horner on the host ----
// After the first row / column pass a set is sum_h (h << s) R_h + sum_l (l + 1) C_l over 2^rb + 2^s points.  Until round 4 a second
// pass and msm_final_quad_kernel finished it on a handful of waves: every remaining point scaled by its weight (~cb dependent quad
// doublings at 1.7 us, a few quad additions at 4 us) and two trees -- 0.12 ms of a 2^20-pair MSM, 0.09 of a 2^17-pair one, all latency.
// The weights are only cb bits wide: with T_j = the plain sum of the points whose weight has bit j set, the set is sum_j 2^j T_j.
// The T_j are independent trees (this kernel: one workgroup per 128 points of a plane, seven levels of quad additions; the workgroup
// that arrives last for its plane adds the plane's partials), and the cb doublings + cb additions that remain are a dependent chain
// of exactly the kind a host core runs eight times faster than a quad of lanes (0.35 / 0.5 us per 4 x 64-bit-limb doubling / addition
// against 1.7 / 4 us): the plane sums go to pinned host memory and msm_planes_finish runs the Horner there.
#define MSM_PLANES_MAX 32      // planes per set the host buffer is sized for (cb <= 23 -> at most s + 1 + rb <= 25)
#define MSM_PLANE_SETS_MAX 4   // runs with more sets keep the GPU tail (the host chain is serial per set)
struct PlaneJob {
    const XYZZu* base;    // array of set 0: R (weights i << s) or C (weights i + 1)
    uint32_t log_len;     // 2^log_len elements per set
    uint32_t o;           // weight of element i is (i + o), o in {0, 1}; the shift of R is applied by the host
    uint32_t n_planes;    // log_len for o = 0; log_len + 1 for o = 1 (the weight 2^log_len of the last element)
    uint32_t wgs;         // workgroups per plane
    uint32_t first_plane; // index of this job's plane 0 among the set's planes
    uint32_t first_block; // of set 0; a set takes blocks_per_set blocks
};
struct PlaneArgs {
    PlaneJob job[2];
    uint32_t blocks_per_set, planes_per_set;
};

__global__ void __launch_bounds__(256) msm_planes_kernel(PlaneArgs args, XYZZu* __restrict__ partials, uint32_t* __restrict__ done,
                                                         XYZZ* __restrict__ plane_sums) {
    __shared__ XYZZu sh[64];
    __shared__ uint32_t is_last;
    const uint32_t set = blockIdx.x / args.blocks_per_set, b = blockIdx.x - set * args.blocks_per_set;
    const PlaneJob& J = args.job[b >= args.job[1].first_block ? 1 : 0];
    const uint32_t pb = b - J.first_block, plane = pb / J.wgs, part = pb - plane * J.wgs;
    const XYZZu* X = J.base + ((size_t)set << J.log_len);
    const uint32_t quad = threadIdx.x >> 2, role = threadIdx.x & 3;
    // element t of plane j < log_len: the t-th value v in [0, 2^log_len) with bit j set, i.e. a one inserted at bit j of t; index v - o.
    // Plane log_len (o = 1 only) is the single element whose weight is 2^log_len: the last one.
    const bool top = plane == J.log_len;
    const uint32_t count = top ? 1u : 1u << (J.log_len - 1);
    auto elem = [&](uint32_t t) -> XYZZu {
        if (t >= count) return xyzzu_identity();
        const uint32_t v = top ? (1u << J.log_len) : ((t >> plane) << (plane + 1)) | (1u << plane) | (t & ((1u << plane) - 1));
        return X[v - J.o];
    };
    const uint32_t t0 = part * 128 + 2 * quad;
    const XYZZu x = xyzzu_sum_q(elem(t0), elem(t0 + 1), role);
    if (role == 0) sh[quad] = x;
    __syncthreads();
    for (uint32_t st = 32; st >= 1; st >>= 1) {
        if (quad < st) {
            const XYZZu a = xyzzu_sum_q(sh[quad], sh[quad + st], role);
            if (role == 0) sh[quad] = a;
        }
        __syncthreads();
    }
    const uint32_t out = set * args.planes_per_set + J.first_plane + plane;
    if (J.wgs == 1) {
        if (threadIdx.x == 0) plane_sums[out] = xyzzu_to_ext(sh[0]);
        return;
    }
    if (threadIdx.x == 0) {
        partials[(size_t)out * 16 + part] = sh[0];
        __threadfence();
        is_last = atomicAdd(&done[out], 1u) == J.wgs - 1;
    }
    __syncthreads();
    if (!is_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the other workgroups' partials, written before their arrivals
    XYZZu y = xyzzu_identity();
    if (quad < J.wgs) y = partials[(size_t)out * 16 + quad];
    if (role == 0 && quad < 16) sh[quad] = y;
    __syncthreads();
    for (uint32_t st = 8; st >= 1; st >>= 1) {
        if (quad < st) {
            const XYZZu a = xyzzu_sum_q(sh[quad], sh[quad + st], role);
            if (role == 0) sh[quad] = a;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) plane_sums[out] = xyzzu_to_ext(sh[0]);
}

// Fixed-base table, one step: out[i] = 2^c * prev[i] (XYZZ; normalised to affine by ec_normalize afterwards)
__global__ void __launch_bounds__(256) msm_table_step_kernel(const Affine* __restrict__ prev, XYZZ* __restrict__ out, uint32_t n, uint32_t c) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Affine p = prev[i];
    if (affine_is_identity(p)) {
        out[i] = xyzz_identity();
        return;
    }
    XYZZu q = xyzzu_double_affine<FqUA>(fu_from_ext(p.x), fu_from_ext(p.y));
    for (uint32_t k = 1; k < c; k++) q = xyzzu_double(q);
    out[i] = xyzzu_to_ext(q);
}

static uint32_t g_window_override = 0;
static size_t g_heavy_div = 32768;
static size_t g_bin_entries = 8192;
static bool g_split_records = true;
void msm_set_split_records(bool on) { g_split_records = on; }
static uint64_t g_rowcol_lanes = 65536;
static bool g_rowcol_asm = true, g_rowcol_qtree = true;  // first pass: chains on the explicit-mad multiplier, then the quad tree
void msm_set_rowcol(uint64_t lanes, uint32_t flavour) {
    g_rowcol_lanes = lanes ? lanes : 65536;
    g_rowcol_asm = flavour & 1;
    g_rowcol_qtree = flavour & 2;
}
static bool g_split_buckets = true;
void msm_set_split_buckets(bool on) { g_split_buckets = on; }
static bool g_quad_tail = true;
void msm_set_quad_tail(bool on) { g_quad_tail = on; }
static bool g_global_order = true;  // false: bucket order local to a sort bin (measured 1.3-1.8x slower accumulation)
void msm_set_bucket_order(int local) { g_global_order = local == 0; }
void msm_set_bin_entries(size_t d) { g_bin_entries = d ? d : 8192; }
void msm_set_heavy_div(size_t d) { g_heavy_div = d ? d : 32768; }
static size_t g_max_chunk = (size_t)1 << 26;
void msm_set_max_chunk(size_t m) { g_max_chunk = m ? m : ((size_t)1 << 26); }
static uint32_t g_reserved_cus = 0;  // measured on MI355X: every partition (16..96 CUs) was slower than none
void msm_set_reserved_cus(uint32_t k) { g_reserved_cus = k; }
uint32_t msm_get_reserved_cus() { return g_reserved_cus; }
void msm_set_window(uint32_t c) { g_window_override = c; }

static uint32_t floor_log2(size_t n) {
    uint32_t lg = 0;
    while (((size_t)1 << (lg + 1)) <= n) lg++;
    return lg;
}

// window width of the plain form
static uint32_t plain_window(size_t n, bool fused) {
    if (g_window_override) return g_window_override;
    const uint32_t lg = floor_log2(n);
    // enough buckets to fill 256 CUs, few enough that the reduction stays small, and 254 mod c large so
    // that the top window is not a handful of over-full buckets
    // a fused batch has count times the buckets for the same chain lengths: narrower windows pay
    if (fused && lg >= 13 && lg <= 18) return lg <= 13 ? 10 : lg <= 16 ? 12 : lg == 17 ? 13 : 14;
    // swept on MI355X after the round-2 sort / reduction (tools/msm_bench.py --no-fixed --windows ...).  255 = 15 x 17: with
    // c = 15 or 17 every window is full width (no narrow windows crowding half of the buckets), which is worth more than
    // the window count alone says: 2^20 1.96 ms at c = 16, 1.73 ms at c = 17; 2^24 24.4 -> 21.6 ms
    if (lg <= 8) return 7;
    if (lg <= 12) return 10;
    if (lg <= 14) return 13;
    if (lg <= 17) return 15;
    return 17;
}

// window width a fixed-base table is built with for n pinned points: all windows share one bucket set, so the
// reduction costs 2 * 2^(c-1) additions once instead of per window and c follows n
static uint32_t normalise_window(uint32_t c) {
    if (c < 2) c = 2;
    if (c > 22) c = 22;  // one bucket set of 2^21 buckets is the most a run sorts (MSM_MAX_C1 << MSM_MAX_L)
    const uint32_t W = (255 + c - 1) / c;
    return (255 + W - 1) / W;
}

uint32_t msm_table_window(size_t n) {
    if (g_window_override) return normalise_window(g_window_override);
    const uint32_t lg = floor_log2(n);
    // swept on MI355X (tools/msm_bench.py --windows): up to 2^17 points c = 17 (a lone MSM would take c = 20 -- 0.70 instead
    // of 0.78 ms at 2^17 -- but the prover's fused batches pay the 2 * 2^(c-1) additions of the reduction once per MSM:
    // 0.18 ms per MSM of a 16 x 2^17 batch at c = 17); 2^18..2^21: c = 20; from 2^22: c = 22 (4.87 against 5.00 ms there)
    uint32_t c = lg <= 12 ? 13 : lg <= 17 ? 17 : lg <= 21 ? 20 : 22;  // up to 2^12 points the MSM is all reduction tail: few buckets
    return normalise_window(c);
}

static MsmPlan make_plan(size_t n, bool fused, const MsmTable* tab, uint32_t force_c = 0) {
    MsmPlan p;
    uint32_t c = tab ? tab->c : force_c ? force_c : plain_window(n, fused);
    if (c < 2) c = 2;
    if (c > 24) c = 24;
    p.W = (255 + c - 1) / c;
    c = (255 + p.W - 1) / p.W;  // the smallest width that gives W windows (18 -> 17, 21 -> 20, 23 -> 22)
    p.c = c;
    p.q = 255 - p.W * (c - 1);
    p.cb = c - 1;
    p.NB = 1u << (c - 1);
    p.shared = tab ? 1 : 0;
    // A lone lane adds ~6 us per entry, and the kernel cannot finish faster than 2 * n*W / 65536 add-times
    // anyway (64 lanes x 1024 SIMDs): buckets above that go to the chunked path, or one lane's chain
    // (e.g. the few buckets of a narrow top window) sets the kernel's duration.
    size_t t = (n * p.W) / g_heavy_div;
    if (t < 32) t = 32;
    const size_t mean = (p.shared ? n * p.W : n) >> p.cb;  // few buckets (narrow windows): the ordinary bucket is not "over-full"
    if (t < 8 * mean) t = 8 * mean;  // the buckets the narrow windows use hold twice the mean
    p.heavy_t = (uint32_t)t;
    p.chunk = 1024;  // a heavy bucket's chunk: 256 lanes x 4 entries, then a tree (4096 measured 30-50 us slower on prover-like columns: 16-entry chains)
    return p;
}

uint32_t msm_get_window(size_t n) { return make_plan(n, false, nullptr).c; }

// host Horner over the set sums of one MSM in the plain form (arithmetic.rs:46-49): acc = sum_w 2^(pos_w) * S_w
static XYZZ combine_windows(const XYZZ* ws, const MsmPlan& p) {
    h64::P acc = h64::identity();
    for (uint32_t w = p.W; w-- > 0;) {
        const uint32_t width = w < p.q ? p.c : p.c - 1;  // window w + 1 starts `width` bits above window w
        for (uint32_t k = 0; k < width; k++) acc = h64::pdouble(acc);
        h64::padd(acc, h64::from_xyzz(ws[w]));
    }
    return h64::to_xyzz(acc);
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static uint32_t bits_of(uint32_t v) {
    uint32_t b = 0;
    while (v) {
        b++;
        v >>= 1;
    }
    return b;
}

// Workspace layout of one in-flight run ("slot")
struct MsmLayout {
    uint32_t fuse;      // MSMs sharing this workspace as one run (1 normally)
    uint32_t n_sets;    // bucket sets of the run: fuse (fixed-base) or fuse * W
    MsmPlan p;
    size_t n, E;        // pairs per MSM; upper bound of the entries of the run
    uint32_t K;         // buckets in total
    uint32_t L, C1, bin_shift;
    uint32_t tile, tiles_x, n_tiles, n_tchunks;  // level-1 tiling: scalars per tile, tiles per MSM, tiles and tile chunks of the run
    uint32_t split_log;  // lanes per bucket of the accumulation = 2^split_log
    // reduction: levels 0..2 of row/column passes before the final kernel
    uint32_t levels, s, rb, s2, s3;
    size_t max_chunks, max_heavy;
    size_t o_zero, o_zero_end, o_bintot, o_hist, o_hcnt, o_fdone, o_cstart, o_tmp, o_vals, o_start, o_counts, o_perm, o_buckets, o_parts, o_RA, o_CA, o_RR,
        o_RC, o_CR, o_CC, o_sums, o_partials, o_ppart, o_hb, o_hc, o_hs, o_ptrs, o_thist, o_toff, o_chsum, total;
};

// force_c / force_split: the chunks of a streamed MSM (msm_stream_host) all use the window width and the lanes-per-bucket
// split of the whole MSM, so that they add into the same parts; -1 / 0 = derive them from n.
// The regions whose size depends only on the plan (buckets, parts, reduction arrays) come first: every chunk layout of a
// streamed MSM has them at the same offsets, and stages B and C take their base address separately (`bbase`).
static int msm_layout(size_t n, MsmLayout* L, uint32_t fuse, const MsmTable* tab, uint32_t force_c = 0, int force_split = -1) {
    MsmPlan p = make_plan(n, fuse > 1, tab, force_c);
    L->p = p;
    L->n = n;
    L->fuse = fuse;
    L->n_sets = p.shared ? fuse : fuse * p.W;
    L->E = n * p.W * fuse;
    if (L->E >= ((size_t)1 << 31)) {
        set_error("msm: %zu entries exceed the 2^31 limit of one run (window override too small?)", L->E);
        return 1;
    }
    if (p.shared && (uint64_t)p.W * tab->stride >= (1ull << 31)) {
        set_error("msm: fixed-base table index exceeds 31 bits");
        return 1;
    }
    const uint64_t K64 = (uint64_t)L->n_sets << p.cb;
    if (K64 > ((uint64_t)MSM_MAX_C1 << MSM_MAX_L)) {
        set_error("msm: too many buckets (%llu) for one run", (unsigned long long)K64);
        return 1;
    }
    L->K = (uint32_t)K64;
    // level-1 coarse bins: about one level-2 tile (MSM_STAGE entries) each, at most MSM_MAX_C1 of them (so that the
    // runs a level-1 workgroup writes stay long), at most 2^MSM_MAX_L buckets each, never straddling two sets
    uint32_t want = 1;
    while (want < MSM_MAX_C1 && (size_t)want * g_bin_entries < L->E) want <<= 1;
    uint32_t Lb = 0;
    while (Lb < p.cb && Lb < MSM_MAX_L && (K64 >> Lb) > want) Lb++;
    while ((K64 >> Lb) > MSM_MAX_C1) {
        if (Lb >= p.cb || Lb >= MSM_MAX_L) {
            set_error("msm: too many bucket sets (%u) for one run", L->n_sets);
            return 1;
        }
        Lb++;
    }
    L->L = Lb;
    L->C1 = (uint32_t)(K64 >> Lb);
    L->tile = (MSM_STAGE / p.W) & ~63u;  // tile * W entries fit the LDS stage of the scatter pass
    if (L->tile > 1024) L->tile = 1024;
    L->tiles_x = (uint32_t)((n + L->tile - 1) / L->tile);
    L->n_tiles = L->tiles_x * fuse;
    L->n_tchunks = (L->n_tiles + MSM_TCHUNK - 1) / MSM_TCHUNK;
    // size classes: width 2^bin_shift entries, the mean bucket size lands in classes 50..100 (of 256)
    L->bin_shift = 0;
    const size_t mean_bucket = (p.shared ? n * p.W : n) >> p.cb;
    while ((mean_bucket >> L->bin_shift) > 100) L->bin_shift++;
    // lanes per bucket: up to 8, while the run has fewer than 2^18 buckets (4 waves per SIMD) and the mean bucket holds at
    // least two entries per lane
    L->split_log = 0;
    while (L->split_log < 3 && ((uint64_t)L->K << (L->split_log + 1)) <= (1u << 18) && (mean_bucket >> (L->split_log + 1)) >= 2) L->split_log++;
    if (!g_split_buckets) L->split_log = 0;
    if (force_split >= 0) L->split_log = (uint32_t)force_split;
    L->max_chunks = L->E / p.chunk + L->E / p.heavy_t + 16;  // sum of ceil(cnt/chunk) over buckets with cnt > heavy_t
    L->max_heavy = L->E / p.heavy_t + 16;
    // reduction geometry
    const uint32_t cb = p.cb;
    L->levels = cb <= 6 ? 0 : cb <= 12 ? 1 : 2;
    L->s = L->levels ? (cb + 1) / 2 : 0;  // level A: cols = 2^s, rows = 2^rb
    L->rb = cb - L->s;
    L->s2 = L->levels == 2 ? (L->rb + 1) / 2 : 0;  // level B on the row sums: cols = 2^s2
    L->s3 = L->levels == 2 ? (L->s + 1) / 2 : 0;   // level B on the column sums: cols = 2^s3
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t E = L->E;
    const uint32_t K = L->K, ns = L->n_sets;
    // -- plan-sized regions (offsets independent of n)
    L->o_zero = off;  // cleared by msm_l1_scan_kernel: size-class histogram and cursors, heavy counters, arrivals per set of the final stage
    L->o_hist = carve(512 * 4);
    L->o_hcnt = carve(16);
    L->o_fdone = carve((size_t)MSM_MAX_C1 * 4);
    L->o_zero_end = off;
    L->o_cstart = carve(((size_t)MSM_MAX_C1 + 1) * 4);
    L->o_bintot = carve((size_t)MSM_MAX_C1 * 4);
    L->o_start = carve((size_t)K * 4);
    L->o_counts = carve((size_t)K * 4);
    L->o_perm = carve((size_t)K * 4);
    L->o_buckets = carve((size_t)K * sizeof(XYZZu));
    L->o_parts = L->split_log ? carve(((size_t)K << L->split_log) * sizeof(XYZZu)) : L->o_buckets;
    L->o_RA = carve(((size_t)ns << L->rb) * sizeof(XYZZu));
    L->o_CA = carve(((size_t)ns << L->s) * sizeof(XYZZu));
    L->o_RR = carve(((size_t)ns << (L->rb - L->s2)) * sizeof(XYZZu));
    L->o_RC = carve(((size_t)ns << L->s2) * sizeof(XYZZu));
    L->o_CR = carve(((size_t)ns << (L->s - L->s3)) * sizeof(XYZZu));
    L->o_CC = carve(((size_t)ns << L->s3) * sizeof(XYZZu));
    L->o_sums = carve((size_t)ns * sizeof(XYZZ));
    L->o_partials = carve((size_t)ns * 16 * sizeof(XYZZu));
    L->o_ppart = carve(ns <= MSM_PLANE_SETS_MAX ? (size_t)ns * MSM_PLANES_MAX * 16 * sizeof(XYZZu) : 0);  // plane partials of the host-finished tail
    L->o_ptrs = carve((size_t)fuse * sizeof(void*));
    // -- regions sized by the entries of the run
    L->o_thist = carve((size_t)L->n_tiles * L->C1 * 2);
    L->o_toff = carve((size_t)L->n_tiles * L->C1 * 4);
    L->o_chsum = carve((size_t)L->n_tchunks * L->C1 * 4);
    L->o_tmp = carve(E * 8);
    L->o_vals = carve(E * 4);
    L->o_hb = carve(L->max_heavy * sizeof(HeavyBucket));
    L->o_hc = carve(L->max_chunks * sizeof(HeavyChunk));
    L->o_hs = carve(L->max_chunks * sizeof(XYZZu));
    L->total = off;
    return 0;
}

// stage A (memory-bound): digits, two-level counting sort by bucket, size order
static int msm_stage_a(Ctx* c, const MsmLayout& L, char* base, const Fe* const* d_scalars_list, const MsmTable* tab, hipStream_t s) {
    const MsmPlan& p = L.p;
    const size_t n = L.n;
    uint32_t* cstart = (uint32_t*)(base + L.o_cstart);
    uint32_t *vals = (uint32_t*)(base + L.o_vals), *start = (uint32_t*)(base + L.o_start);
    uint32_t *counts = (uint32_t*)(base + L.o_counts), *perm = (uint32_t*)(base + L.o_perm);
    int t0 = c->timer_begin("msm_digits", s);
    L1Args a;
    a.scalars_one = d_scalars_list[0];
    a.list = nullptr;
    if (L.fuse > 1) {  // the scalar arrays' addresses, read by the kernels per blockIdx.y (staged through pinned memory)
        int rc = c->stage_h2d(base + L.o_ptrs, d_scalars_list, L.fuse * sizeof(void*), s);
        if (rc) return rc;
        a.list = (const Fe* const*)(base + L.o_ptrs);
    }
    a.n = (uint32_t)n;
    a.c = p.c;
    a.q = p.q;
    a.W = p.W;
    a.cb = p.cb;
    a.L = L.L;
    a.C1 = L.C1;
    a.shared = p.shared;
    a.stride = tab ? (uint32_t)tab->stride : 0;
    a.tile = L.tile;
    a.tile_hist = (uint16_t*)(base + L.o_thist);
    a.tile_off = (const uint32_t*)(base + L.o_toff);
    a.tmp = (uint2*)(base + L.o_tmp);
    a.chk_bases = c->pin_chk.bases;
    a.chk_samples = c->pin_chk.samples;
    a.chk_flag = c->pin_chk.flag;
    a.chk_n = c->pin_chk.n;
    a.chk_total = c->pin_chk.total;
    c->pin_chk = Ctx::PinCheck();  // once per call
    uint32_t* chsum = (uint32_t*)(base + L.o_chsum);
    const dim3 tiles(L.tiles_x, L.fuse), chunks(L.n_tchunks, (L.C1 + 255) / 256);
    hipLaunchKernelGGL(msm_l1_count_kernel, tiles, dim3(256), 0, s, a);
    H2_CHECK(hipGetLastError());
    c->timer_end(t0, s);
    int t1 = c->timer_begin("msm_sort", s);
    hipLaunchKernelGGL(msm_l1_chunk_kernel, chunks, dim3(256), 0, s, (const uint16_t*)a.tile_hist, L.n_tiles, L.C1, chsum);
    H2_CHECK(hipGetLastError());
    uint32_t* bin_total = (uint32_t*)(base + L.o_bintot);
    hipLaunchKernelGGL(msm_l1_scan_kernel, dim3((L.C1 + 63) / 64), dim3(1024), 0, s, chsum, L.n_tchunks, L.C1, bin_total,
                       (uint32_t*)(base + L.o_zero), (uint32_t)((L.o_zero_end - L.o_zero) / 4));
    H2_CHECK(hipGetLastError());
    hipLaunchKernelGGL(msm_l1_offsets_kernel, chunks, dim3(256), 0, s, (const uint16_t*)a.tile_hist, L.n_tiles, L.C1, (const uint32_t*)chsum,
                       (const uint32_t*)bin_total, cstart, (uint32_t*)(base + L.o_toff));
    H2_CHECK(hipGetLastError());
    // bins of several level-2 tiles: split records (see msm_l2_kernel)
    const bool split = g_split_records && L.E / L.C1 > MSM_STAGE;
    a.tmp_e = (uint32_t*)(base + L.o_tmp);
    a.tmp_k = (uint16_t*)(base + L.o_tmp + L.E * 4);
    if (split)
        hipLaunchKernelGGL(msm_l1_scatter_kernel<true>, tiles, dim3(1024), 0, s, a);
    else
        hipLaunchKernelGGL(msm_l1_scatter_kernel<false>, tiles, dim3(1024), 0, s, a);
    H2_CHECK(hipGetLastError());
    uint32_t* hist = (uint32_t*)(base + L.o_hist);
    if (split)
        hipLaunchKernelGGL(msm_l2_kernel<true>, dim3(L.C1), dim3(1024), 0, s, (const uint2*)nullptr, (const uint32_t*)a.tmp_e, (const uint16_t*)a.tmp_k, cstart,
                           L.L, L.bin_shift, vals, start, counts, perm, g_global_order ? hist : (uint32_t*)nullptr);
    else
        hipLaunchKernelGGL(msm_l2_kernel<false>, dim3(L.C1), dim3(1024), 0, s, (const uint2*)(base + L.o_tmp), (const uint32_t*)nullptr,
                           (const uint16_t*)nullptr, cstart, L.L, L.bin_shift, vals, start, counts, perm, g_global_order ? hist : (uint32_t*)nullptr);
    H2_CHECK(hipGetLastError());
    // size-ordered bucket permutation (skipped under the local-order debug knob) and the list of over-full buckets with their chunks
    hipLaunchKernelGGL(msm_bucket_scatter_kernel, dim3((L.K + 1024 * MSM_BS_PER - 1) / (1024 * MSM_BS_PER)), dim3(1024), 0, s, (const uint32_t*)counts,
                       (const uint32_t*)start, L.K, L.bin_shift, p.heavy_t, p.chunk, hist, g_global_order ? perm : (uint32_t*)nullptr,
                       (uint32_t*)(base + L.o_hcnt), (HeavyBucket*)(base + L.o_hb), (HeavyChunk*)(base + L.o_hc));
    H2_CHECK(hipGetLastError());
    c->timer_end(t1, s);
    return 0;
}

// stage B (VALU-bound): bucket accumulation, plus the chunked path for over-full buckets.  `base` holds the sorted entries of
// this run, `bbase` the parts / buckets they are added to (the same arena unless the run is one chunk of a streamed MSM);
// cont: the parts already hold sums (msm_accum_kernel); combine: fold the parts of every bucket afterwards.
static int msm_stage_b(Ctx* c, const MsmLayout& L, char* base, const Affine* d_points, hipStream_t s, char* bbase = nullptr, bool cont = false,
                       bool combine = true) {
    const MsmPlan& p = L.p;
    if (!bbase) bbase = base;
    uint32_t* vals = (uint32_t*)(base + L.o_vals);
    uint32_t *start = (uint32_t*)(base + L.o_start), *counts = (uint32_t*)(base + L.o_counts), *perm = (uint32_t*)(base + L.o_perm);
    XYZZu* buckets = (XYZZu*)(bbase + L.o_buckets);
    uint32_t* hcnt = (uint32_t*)(base + L.o_hcnt);
    HeavyBucket* hb = (HeavyBucket*)(base + L.o_hb);
    HeavyChunk* hc = (HeavyChunk*)(base + L.o_hc);
    XYZZu* hs = (XYZZu*)(base + L.o_hs);
    int t2 = c->timer_begin("msm_accum", s);
    hipEvent_t ke0 = nullptr, ke1 = nullptr;
    XYZZu* parts = (XYZZu*)(bbase + L.o_parts);
    const uint32_t lanes = L.K << L.split_log;
    const uint32_t cont_u = cont ? 1u : 0u;
    // the first hgrid workgroups take the over-full buckets' chunks (none on uniform scalars: they leave at once), the rest one lane per part
    const uint32_t hgrid = (uint32_t)(L.max_chunks < (size_t)c->sm_count ? L.max_chunks : (size_t)c->sm_count);
    const dim3 agrid(hgrid + (lanes + 255) / 256);
    if (c->timer_kernel("msm_accum", &ke0, &ke1) >= 0)  // the dispatch's own begin / end timestamps: no marker packets around it
        hipExtLaunchKernelGGL(msm_accum_kernel, agrid, dim3(256), 0, s, ke0, ke1, 0, d_points, (const uint32_t*)vals, (const uint32_t*)start,
                              (const uint32_t*)counts, (const uint32_t*)perm, L.K, L.split_log, p.heavy_t, cont_u, parts, hgrid, (const uint32_t*)hcnt, hb,
                              (const HeavyChunk*)hc, hs);
    else
        hipLaunchKernelGGL(msm_accum_kernel, agrid, dim3(256), 0, s, d_points, (const uint32_t*)vals, (const uint32_t*)start, (const uint32_t*)counts,
                           (const uint32_t*)perm, L.K, L.split_log, p.heavy_t, cont_u, parts, hgrid, (const uint32_t*)hcnt, hb, (const HeavyChunk*)hc, hs);
    H2_CHECK(hipGetLastError());
    c->timer_end(t2, s);
    if (L.split_log && combine) {
        const uint32_t comb_lanes = L.K << (L.split_log - 1);
        hipLaunchKernelGGL(msm_combine_kernel, dim3((comb_lanes + 255) / 256), dim3(256), 0, s, (const XYZZu*)parts, L.K, L.split_log, buckets);
        H2_CHECK(hipGetLastError());
    }
    return 0;
}

// the two jobs (row sums, column sums) of one row/column pass over n_arr arrays of 2^log_rows x 2^log_cols elements.
// Lanes per sum: every lane gets the same number of terms t, with t the smallest power of two that keeps the pass
// within the lane budget below.
static void rowcol_jobs(RowColArgs* ra, const XYZZu* in, XYZZu* out_rows, XYZZu* out_cols, uint32_t n_arr, uint32_t log_rows, uint32_t log_cols,
                        uint32_t* n_blocks, bool quad = false) {
    const uint64_t elems = (uint64_t)n_arr << (log_rows + log_cols);
    // one wave per SIMD while that keeps the chains at 16 terms (a second wave only shares the SIMD: measured 0.273 against 0.276 ms
    // for a 2^20-pair MSM), two once they would grow longer (2^21 buckets: 0.66 against 0.70 ms)
    const uint64_t budget = (2 * elems) >> 4 > g_rowcol_lanes ? 2 * g_rowcol_lanes : g_rowcol_lanes;
    uint32_t t_log = 1;
    while (!quad && (2 * elems) >> t_log > budget) t_log++;
    for (uint32_t kind = 0; kind < 2; kind++) {
        RowColJob* j = &ra->job[ra->n_jobs++];
        const uint32_t log_terms = kind ? log_rows : log_cols, log_sums = kind ? log_cols : log_rows;
        uint32_t g_log = log_terms > t_log ? log_terms - t_log : 0;
        const uint32_t g_max = quad ? 4 : 8;  // quads (of 16) or lanes (of 256) per sum
        if (g_log > g_max) g_log = g_max;
        j->in = in;
        j->out = kind ? out_cols : out_rows;
        j->n_arr = n_arr;
        j->log_rows = log_rows;
        j->log_cols = log_cols;
        j->cols_kind = kind;
        j->g_log = g_log;
        j->first_block = *n_blocks;
        const uint64_t tasks = (uint64_t)n_arr << log_sums;
        const uint32_t per_block = (quad ? 16u : 256u) >> g_log;
        *n_blocks += (uint32_t)((tasks + per_block - 1) / per_block);
    }
}

static bool g_plane_tail = true;
void msm_set_plane_tail(bool on) { g_plane_tail = on; }
// whether a run's tail is finished on the host from bit-plane sums (msm_planes_kernel): few sets, one row / column level at least,
// at most 16 workgroups (2^11 points) per plane, and the sums not wanted in HBM for the RCCL gather
static bool msm_plane_tail(const Ctx* c, const MsmLayout& L) {
    return g_plane_tail && g_quad_tail && L.levels >= 1 && L.n_sets <= MSM_PLANE_SETS_MAX && L.rb <= 11 && L.s <= 11 && L.rb >= 1 && L.s >= 1 &&
           L.s + 1 + L.rb <= MSM_PLANES_MAX && !c->gather_want;
}
static inline uint32_t msm_planes_per_set(const MsmLayout& L) { return L.rb + L.s + 1; }

// host side of the plane tail: set = sum_j 2^j C-plane_j + sum_j 2^(s + j) R-plane_j, as one Horner from the top (planes: [C 0..s | R 0..rb-1])
static void msm_planes_finish(const MsmLayout& L, const XYZZ* planes, XYZZ* sums) {
    const uint32_t pps = msm_planes_per_set(L);
    for (uint32_t set = 0; set < L.n_sets; set++) {
        const XYZZ* P = planes + (size_t)set * pps;
        h64::P acc = h64::identity();
        for (uint32_t m = L.s + L.rb; m-- > 0;) {
            acc = h64::pdouble(acc);
            if (m >= L.s) h64::padd(acc, h64::from_xyzz(P[L.s + 1 + (m - L.s)]));
            if (m <= L.s) h64::padd(acc, h64::from_xyzz(P[m]));
        }
        sums[set] = h64::to_xyzz(acc);
    }
}

// stage C (latency-bound): bucket reduction to one sum per set, copied to h_sums (host).  h_planes (optional, pinned host memory for
// n_sets * MSM_PLANES_MAX points): when the run qualifies (msm_plane_tail) the kernels stop at the bit-plane sums and deliver those
// instead; the caller then completes h_sums with msm_planes_finish once the stream has drained.
static int msm_stage_c(Ctx* c, const MsmLayout& L, char* base, XYZZ* h_sums, hipStream_t s, XYZZ* h_planes = nullptr) {
    XYZZu* buckets = (XYZZu*)(base + L.o_buckets);
    XYZZu *RA = (XYZZu*)(base + L.o_RA), *CA = (XYZZu*)(base + L.o_CA), *RR = (XYZZu*)(base + L.o_RR), *RC = (XYZZu*)(base + L.o_RC);
    XYZZu *CR = (XYZZu*)(base + L.o_CR), *CC = (XYZZu*)(base + L.o_CC);
    // The sums go straight into the caller's pinned host buffer (a HostBuf: device-visible, as pin_flag is) -- no copy launch
    // behind the last kernel; a multi-device run that gathers over RCCL keeps them in HBM and copies (api.hip gather_rccl).
    const bool direct = !c->gather_want;
    XYZZ* sums = direct ? h_sums : (XYZZ*)(base + L.o_sums);
    const uint32_t ns = L.n_sets, cb = L.p.cb;
    // few sets: the second row/column pass and the final scaling are latency chains on a handful of waves -- one quad of
    // lanes per group operation (ecq.h).  Many sets (fused batches of the plain form) fill the chip: one lane each.
    const bool quad = g_quad_tail && ns <= 64;
    int t4 = c->timer_begin("msm_reduce", s);
    FinalArgs fa;
    memset(&fa, 0, sizeof(fa));
    auto set_arr = [&](uint32_t i, const XYZZu* b, uint32_t log_len, uint32_t o, uint32_t k) {
        fa.arr[i].base = b;
        fa.arr[i].stride = 1u << log_len;
        fa.arr[i].len = 1u << log_len;
        fa.arr[i].o = o;
        fa.arr[i].k = k;
        fa.arr[i].nbits = bits_of(((1u << log_len) - 1 + o) << k);
    };
    if (L.levels == 0) {
        set_arr(0, buckets, cb, 1, 0);
        fa.n_arr = 1;
    } else {
        RowColArgs ra;
        memset(&ra, 0, sizeof(ra));
        uint32_t nblk = 0;
        rowcol_jobs(&ra, buckets, RA, CA, ns, L.rb, L.s, &nblk);
        if (g_rowcol_qtree)
            hipLaunchKernelGGL(msm_rowcol_qtree_kernel<FqUA>, dim3(nblk), dim3(256), 0, s, ra);
        else if (g_rowcol_asm)
            hipLaunchKernelGGL(msm_rowcol_kernel<FqUA>, dim3(nblk), dim3(256), 0, s, ra);
        else
            hipLaunchKernelGGL(msm_rowcol_kernel<FqU>, dim3(nblk), dim3(256), 0, s, ra);
        H2_CHECK(hipGetLastError());
        if (h_planes && msm_plane_tail(c, L)) {
            PlaneArgs pa;
            memset(&pa, 0, sizeof(pa));
            uint32_t blk = 0, pl = 0;
            for (uint32_t k = 0; k < 2; k++) {  // [0] the column sums (weights l + 1), [1] the row sums (weights h, shifted by s on the host)
                PlaneJob& J = pa.job[k];
                J.base = k ? RA : CA;
                J.log_len = k ? L.rb : L.s;
                J.o = k ? 0 : 1;
                J.n_planes = J.log_len + J.o;
                J.wgs = J.log_len > 7 ? 1u << (J.log_len - 8) : 1;  // 128 points of a plane's 2^(log_len - 1) per workgroup
                J.first_plane = pl;
                J.first_block = blk;
                pl += J.n_planes;
                blk += J.n_planes * J.wgs;
            }
            pa.blocks_per_set = blk;
            pa.planes_per_set = pl;
            hipLaunchKernelGGL(msm_planes_kernel, dim3(ns * blk), dim3(256), 0, s, pa, (XYZZu*)(base + L.o_ppart), (uint32_t*)(base + L.o_fdone), h_planes);
            H2_CHECK(hipGetLastError());
            c->timer_end(t4, s);
            return 0;
        }
        if (L.levels == 1) {
            set_arr(0, RA, L.rb, 0, L.s);
            set_arr(1, CA, L.s, 1, 0);
            fa.n_arr = 2;
        } else {
            memset(&ra, 0, sizeof(ra));
            nblk = 0;
            rowcol_jobs(&ra, RA, RR, RC, ns, L.rb - L.s2, L.s2, &nblk, quad);
            rowcol_jobs(&ra, CA, CR, CC, ns, L.s - L.s3, L.s3, &nblk, quad);
            if (quad)
                hipLaunchKernelGGL(msm_rowcol_quad_kernel, dim3(nblk), dim3(64), 0, s, ra);
            else
                hipLaunchKernelGGL(msm_rowcol_kernel<FqU>, dim3(nblk), dim3(256), 0, s, ra);
            H2_CHECK(hipGetLastError());
            set_arr(0, RR, L.rb - L.s2, 0, L.s + L.s2);
            set_arr(1, RC, L.s2, 0, L.s);
            set_arr(2, CR, L.s - L.s3, 0, L.s3);
            set_arr(3, CC, L.s3, 1, 0);
            fa.n_arr = 4;
        }
    }
    if (quad) {
        uint32_t n_el = 0;
        for (uint32_t i = 0; i < fa.n_arr; i++) n_el += fa.arr[i].len;
        const uint32_t n_groups = (n_el + 15) / 16;  // <= 16: at most 4 arrays of 64
        XYZZu* partials = (XYZZu*)(base + L.o_partials);
        hipLaunchKernelGGL(msm_final_quad_kernel, dim3(ns * n_groups), dim3(64), 0, s, fa, n_groups, partials, (uint32_t*)(base + L.o_fdone), sums);
    } else {
        hipLaunchKernelGGL(msm_final_kernel, dim3(ns), dim3(256), 0, s, fa, sums);
    }
    H2_CHECK(hipGetLastError());
    c->timer_end(t4, s);
    if (!direct) H2_CHECK(hipMemcpyAsync(h_sums, sums, (size_t)ns * sizeof(XYZZ), hipMemcpyDeviceToHost, s));
    if (c->gather_want) {  // the same sums stay on the device for the RCCL gather of a multi-device MSM (api.hip gather_rccl)
        const size_t bytes = (size_t)ns * sizeof(XYZZ);
        if (c->gather_off + bytes <= H2_GATHER_OWN && c->gather.p)
            H2_CHECK(hipMemcpyAsync((char*)c->gather.p + c->gather_off, sums, bytes, hipMemcpyDeviceToDevice, s));
        c->gather_off += bytes;
    }
    return 0;
}

// sets per MSM and the host-side finish
static inline uint32_t sets_per_msm(const MsmPlan& p) { return p.shared ? 1 : p.W; }
static inline XYZZ finish_msm(const XYZZ* sums, const MsmPlan& p) { return p.shared ? sums[0] : combine_windows(sums, p); }

// `count` independent MSMs of n <= 2^26 pairs each over the same bases (ParamsKZG::commit_lagrange for the
// advice columns of one proof, plonk/prover.rs:361-365).  count == 1 runs the three stages back to back on the
// caller's stream.  count > 1 pipelines whole MSMs over three streams with three workspace slots:
//   aux1:  A(0) | A(1) | A(2) | ...            digits + sort                (memory-bound)
//   s   :       | B(0) | B(1) | ...            accumulate                  (VALU-bound)
//   aux2:              | C(0) | C(1) | ...     reduce + copy-out           (latency-bound, few waves)
// so that every accumulate launch is full-size while the sort of the next MSM and the reduction of the previous
// one run underneath it.
// scalars_on_host: d_scalars[j] are host pointers, uploaded into three device slots ahead of stage A.
static int msm_batch_chunk(Ctx* c, const Fe* const* d_scalars, bool scalars_on_host, const Affine* d_points, const MsmTable* tab, size_t n,
                           size_t count, XYZZ* h_out, hipStream_t s) {
    MsmLayout L;
    int rc = msm_layout(n, &L, 1, tab);
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
    const uint32_t spm = sets_per_msm(L.p);
    rc = c->host_ws.ensure(count * spm * sizeof(XYZZ));
    if (rc) return rc;
    XYZZ* h_ws = (XYZZ*)c->host_ws.p;
    XYZZ* h_pl = nullptr;  // a lone MSM: the reduction's tail is finished on the host from bit-plane sums
    if (count == 1 && msm_plane_tail(c, L)) {
        if ((rc = c->host_planes.ensure((size_t)L.n_sets * MSM_PLANES_MAX * sizeof(XYZZ)))) return rc;
        h_pl = (XYZZ*)c->host_planes.p;
    }
    rc = c->ws_acquire(s);
    if (rc) return rc;
    WsGuard guard(c, s);
    int t_all = c->timer_begin("msm_total", s);
    if (count == 1) {
        char* base = (char*)c->msm_slot[0].p;
        const Fe* sc;
        if ((rc = scalars_for(0, s, &sc))) return rc;
        if ((rc = msm_stage_a(c, L, base, &sc, tab, s))) return rc;
        if ((rc = msm_stage_b(c, L, base, d_points, s))) return rc;
        if ((rc = msm_stage_c(c, L, base, h_ws, s, h_pl))) return rc;
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
            if ((rc = msm_stage_a(c, L, base, &sc, tab, a1))) return rc;
            H2_CHECK(hipEventRecord(ev[1 + 3 * j], a1));
            H2_CHECK(hipStreamWaitEvent(sb, ev[1 + 3 * j], 0));
            if ((rc = msm_stage_b(c, L, base, d_points, sb))) return rc;
            H2_CHECK(hipEventRecord(ev[2 + 3 * j], sb));
            H2_CHECK(hipStreamWaitEvent(a2, ev[2 + 3 * j], 0));
            if ((rc = msm_stage_c(c, L, base, h_ws + j * spm, a2))) return rc;
            H2_CHECK(hipEventRecord(ev[3 + 3 * j], a2));
        }
        H2_CHECK(hipStreamWaitEvent(s, ev[3 + 3 * (count - 1)], 0));  // C stages are in order on aux2
    }
    c->timer_end(t_all, s);
    H2_CHECK(hipStreamSynchronize(s));
    if (h_pl) msm_planes_finish(L, h_pl, h_ws);
    for (size_t j = 0; j < count; j++) h_out[j] = finish_msm(h_ws + j * spm, L.p);
    return guard.release();
}

// Small MSMs over the same bases, fused: the `count` MSMs run as ONE pass of the three stages whose bucket sets are
// the sets of all of them.  The bucket reduction is a chain of dependent EC operations -- its duration hardly
// depends on the size -- and dominates a 2^17-pair MSM; pipelining whole MSMs over streams only overlaps those
// chains, fusing pays for one.  Sort and accumulate become one large launch each.
static int msm_fused_chunk(Ctx* c, const Fe* const* d_scalars, bool scalars_on_host, const Affine* d_points, const MsmTable* tab, size_t n,
                           size_t count, XYZZ* h_out, hipStream_t s) {
    MsmLayout L;
    int rc = msm_layout(n, &L, (uint32_t)count, tab);
    if (rc) return rc;
    if ((rc = c->msm_slot[0].ensure(L.total))) return rc;
    std::vector<const Fe*> list(d_scalars, d_scalars + count);
    if ((rc = c->ws_acquire(s))) return rc;
    WsGuard guard(c, s);
    if (scalars_on_host) {
        if ((rc = c->msm_scalars[0].ensure(count * n * sizeof(Fe)))) return rc;
        for (size_t j = 0; j < count; j++) {
            Fe* dst = (Fe*)c->msm_scalars[0].p + j * n;
            H2_CHECK(hipMemcpyAsync(dst, d_scalars[j], n * sizeof(Fe), hipMemcpyHostToDevice, s));
            list[j] = dst;
        }
    }
    if ((rc = c->host_ws.ensure((size_t)L.n_sets * sizeof(XYZZ)))) return rc;
    XYZZ* h_ws = (XYZZ*)c->host_ws.p;
    XYZZ* h_pl = nullptr;
    if (msm_plane_tail(c, L)) {
        if ((rc = c->host_planes.ensure((size_t)L.n_sets * MSM_PLANES_MAX * sizeof(XYZZ)))) return rc;
        h_pl = (XYZZ*)c->host_planes.p;
    }
    char* base = (char*)c->msm_slot[0].p;
    int t_all = c->timer_begin("msm_total", s);
    if ((rc = msm_stage_a(c, L, base, list.data(), tab, s))) return rc;
    if ((rc = msm_stage_b(c, L, base, d_points, s))) return rc;
    if ((rc = msm_stage_c(c, L, base, h_ws, s, h_pl))) return rc;
    c->timer_end(t_all, s);
    H2_CHECK(hipStreamSynchronize(s));
    if (h_pl) msm_planes_finish(L, h_pl, h_ws);
    const uint32_t spm = sets_per_msm(L.p);
    for (size_t j = 0; j < count; j++) h_out[j] = finish_msm(h_ws + j * spm, L.p);
    return guard.release();
}

// ---- host-resident scalars: the upload runs under the work -------------------------------------------------------------------
// best_multiexp hands over host slices (arithmetic.rs:132; one call per column at plonk/prover.rs:361-365), 32 B per pair over
// PCIe (~0.55 ms per 2^20 pairs).  A lone MSM is cut into K chunks of growing size; chunk k + 1 crosses PCIe (copy stream) and is
// sorted (sort stream) while chunk k is accumulated, and every chunk adds into the SAME parts / buckets (msm_accum_kernel with
// cont = 1), so the bucket reduction is paid once.  Chunk sizes form the geometric ladder f_k ~ r^-k with r = (upload time) /
// (compute time) per pair: each upload then ends as the previous chunk's work does, and only the first -- the smallest --
// upload is exposed.  What streaming costs: a bucket's first addition of a chunk is a full mixed addition (into a fresh bucket
// it is two multiplications), ~7 multiplications per bucket and extra chunk, hence few chunks.
// The copies are hipMemcpyAsync from the caller's (pageable) memory in chunk-sized pieces: HIP pins the pages and the DMA runs at
// PCIe speed, the call returns when its piece has left -- by which time the previous chunk's kernels are queued and running.
// swept on MI355X (tools/stream_sweep.py; 2^20 / 2^22 / 2^24 pairs, pinned bases): 3 chunks at ratio 0.6 -- 1.62 / 5.36 / 19.4 ms
// against 1.87 / 7.35 / 27.8 ms with the whole upload first and 1.27 / 4.60 / 17.0 ms device-resident.  With the bases crossing
// too (unpinned, 96 B per pair) the run is upload-bound: more and nearly equal chunks (6 at 3 x 0.4), 2.55 / 8.3 / 31.3 ms
// against 3.44 / 13.2 / 51.2 ms.
static uint32_t g_stream_chunks = 3;
static bool g_stream_chunks_default = true;  // below 2^20 pairs the default is two chunks; an explicit setting is taken as given
static double g_stream_ratio = 0.6;
static size_t g_stream_min_n = (size_t)1 << 19;  // 2^19 pairs: 1.13 -> 1.04 ms (two chunks); 2^18: 0.72 -> 0.80 ms, not worth it
void msm_set_stream(uint32_t chunks, double ratio, size_t min_n) {
    g_stream_chunks = chunks ? chunks : 3;
    g_stream_chunks_default = chunks == 0;
    g_stream_ratio = ratio > 0 ? ratio : 0.6;
    g_stream_min_n = min_n ? min_n : ((size_t)1 << 19);
}

// The copies are issued by a helper thread (engine.h, copier_*): hipMemcpyAsync from pageable memory returns only when
// its piece has left the host, and the thread that enqueues the kernels must not sit in it -- with one thread doing both, chunk
// k + 1 started to cross PCIe only after chunk k's launches were queued and the sort of chunk k + 1 only a launch latency after
// its copy had ended: no overlap was left (measured, rocprofv3 timeline of round 3).  The enqueueing thread follows the copier's
// progress counter -- a chunk is at most a few hundred microseconds away -- and then makes its stream wait on the job's event.

// a streamed run owns the copier until every job has been issued: an early error return must not leave it reading the caller's
// arrays (or this frame's events) behind the call's back
struct CopierDrain {
    Ctx* c;
    size_t n_jobs;
    ~CopierDrain() { copier_abort(c, false, n_jobs); }
};

// units cut into at most K pieces with sizes ~ r^-k, each a multiple of `quantum` (the last takes the remainder)
static void stream_ladder(size_t units, uint32_t K, double r, size_t quantum, std::vector<size_t>* out) {
    out->clear();
    if (K < 1) K = 1;
    while (K > 1 && units < (size_t)K * quantum * 2) K--;
    std::vector<double> f(K);
    double w = 1.0, tot = 0.0;
    for (uint32_t k = 0; k < K; k++) {
        f[k] = w;
        tot += w;
        w /= r;
    }
    size_t used = 0;
    for (uint32_t k = 0; k + 1 < K; k++) {
        size_t m = (size_t)((double)units * f[k] / tot / (double)quantum + 0.5) * quantum;
        if (m < quantum) m = quantum;
        if (used + m + quantum > units) break;
        out->push_back(m);
        used += m;
    }
    out->push_back(units - used);
}

// test hook (host only, no GPU): the chunk sizes a streamed MSM of n pairs would use
size_t msm_debug_ladder(size_t n, uint32_t chunks, double ratio, bool with_bases, size_t* out, size_t cap) {
    std::vector<size_t> sz;
    const uint32_t k0 = chunks ? chunks : g_stream_chunks;
    const double r = ratio > 0 ? ratio : g_stream_ratio;
    const uint32_t kk = !chunks && g_stream_chunks_default && n < ((size_t)1 << 20) ? 2 : k0;
    stream_ladder(n, with_bases ? 2 * kk : kk, with_bases ? 2.0 * r : r, 4096, &sz);
    for (size_t i = 0; i < sz.size() && i < cap; i++) out[i] = sz[i];
    return sz.size();
}

static int msm_stream_host(Ctx* c, const Fe* h_scalars, const Affine* h_bases, const Affine* d_points, const MsmTable* tab, size_t n, XYZZ* h_out,
                           hipStream_t s) {
    std::vector<size_t> sz;
    const uint32_t kk = g_stream_chunks_default && n < ((size_t)1 << 20) ? 2 : g_stream_chunks;
    stream_ladder(n, h_bases ? 2 * kk : kk, h_bases ? 2.0 * g_stream_ratio : g_stream_ratio, 4096, &sz);
    const size_t K = sz.size();
    MsmLayout Lt;
    int rc = msm_layout(n, &Lt, 1, tab);
    if (rc) return rc;
    std::vector<MsmLayout> Lk(K);
    size_t need = 0;
    for (size_t k = 0; k < K; k++) {
        if ((rc = msm_layout(sz[k], &Lk[k], 1, tab, Lt.p.c, (int)Lt.split_log))) return rc;
        if (Lk[k].total > need) need = Lk[k].total;
    }
    if ((rc = c->msm_slot[0].ensure(need))) return rc;
    if ((rc = c->msm_scalars[0].ensure(n * sizeof(Fe)))) return rc;
    Fe* d_sc = (Fe*)c->msm_scalars[0].p;
    Affine* d_bs = nullptr;
    if (h_bases) {
        if ((rc = c->msm_bases.ensure(n * sizeof(Affine)))) return rc;
        d_bs = (Affine*)c->msm_bases.p;
        d_points = d_bs;
    }
    const uint32_t spm = sets_per_msm(Lt.p);
    if ((rc = c->host_ws.ensure(spm * sizeof(XYZZ)))) return rc;
    XYZZ* h_ws = (XYZZ*)c->host_ws.p;
    XYZZ* h_pl = nullptr;
    if (msm_plane_tail(c, Lk[K - 1])) {
        if ((rc = c->host_planes.ensure((size_t)Lk[K - 1].n_sets * MSM_PLANES_MAX * sizeof(XYZZ)))) return rc;
        h_pl = (XYZZ*)c->host_planes.p;
    }
    if ((rc = c->ensure_aux(K + 2))) return rc;
    if ((rc = c->ws_acquire(s))) return rc;
    WsGuard guard(c, s);
    int t_all = c->timer_begin("msm_total", s);
    // Sort and accumulation of all chunks run in order on ONE stream.  Putting the sort of chunk k + 1 on a second stream under
    // the accumulation of chunk k was slower: the accumulation holds every wave slot and 64 KB of LDS per CU, the sort's
    // workgroups (1024 lanes, 136 KB) only got onto the chip in its tail, both kernels ran 10-30 % longer and the next
    // accumulation still waited for the sort (timeline in profiles/r03_stream_*).
    hipStream_t cs = c->aux2;
    hipEvent_t* ev = c->aux_events.data();  // [0] start, [1 + k] chunk k uploaded
    H2_CHECK(hipEventRecord(ev[0], s));
    H2_CHECK(hipStreamWaitEvent(cs, ev[0], 0));
    char* base = (char*)c->msm_slot[0].p;
    const Affine* points0 = tab ? tab->table : d_points;
    size_t o = 0;
    std::vector<CopyJob> jobs;
    for (size_t k = 0; k < K; k++) {
        if (h_bases) jobs.push_back(CopyJob{d_bs + o, h_bases + o, sz[k] * sizeof(Affine), nullptr, nullptr, false});
        jobs.push_back(CopyJob{d_sc + o, h_scalars + o, sz[k] * sizeof(Fe), ev[1 + k], nullptr, false});
        o += sz[k];
    }
    const size_t per_chunk = h_bases ? 2 : 1;
    H2_CHECK(hipStreamSynchronize(s));  // the copies start now: whatever was queued ahead of this call has to be done with the buffers
    if ((rc = copier_begin(c, false, cs))) return rc;
    CopierDrain drain{c, K * per_chunk};
    if ((rc = copier_push(c, false, jobs))) return rc;
    o = 0;
    for (size_t k = 0; k < K; k++) {
        if ((rc = copier_wait(c, false, (k + 1) * per_chunk))) return rc;
        H2_CHECK(hipStreamWaitEvent(s, ev[1 + k], 0));
        const Fe* sc = d_sc + o;
        if ((rc = msm_stage_a(c, Lk[k], base, &sc, tab, s))) return rc;
        if ((rc = msm_stage_b(c, Lk[k], base, points0 + o, s, base, k > 0, k + 1 == K))) return rc;
        o += sz[k];
    }
    if ((rc = msm_stage_c(c, Lk[K - 1], base, h_ws, s, h_pl))) return rc;
    c->timer_end(t_all, s);
    H2_CHECK(hipStreamSynchronize(s));
    if (h_pl) msm_planes_finish(Lk[K - 1], h_pl, h_ws);
    h_out[0] = finish_msm(h_ws, Lt.p);
    return guard.release();
}

// The fused batch with host-resident columns: groups of columns (a ladder again), each group one fused run on its own stream and
// workspace slot -- group g + 1 crosses PCIe while group g is sorted and accumulated, and its reduction tail (a latency chain on a
// few waves) runs under the next group's accumulation.
static int msm_fused_groups_host(Ctx* c, const Fe* const* h_scalars, const Affine* d_points, const MsmTable* tab, size_t n, size_t count,
                                 size_t fuse_max, XYZZ* h_out, hipStream_t s) {
    std::vector<size_t> ladder, groups;
    stream_ladder(count, g_stream_chunks, g_stream_ratio * 0.85, 1, &ladder);  // 16 columns: 2 + 5 + 9
    for (size_t g : ladder)
        for (size_t o = 0; o < g; o += fuse_max) groups.push_back(g - o < fuse_max ? g - o : fuse_max);
    const size_t G = groups.size();
    std::vector<MsmLayout> Lg(G);
    size_t need = 0, max_sets = 0;
    int rc;
    for (size_t g = 0; g < G; g++) {
        if ((rc = msm_layout(n, &Lg[g], (uint32_t)groups[g], tab))) return rc;
        if (Lg[g].total > need) need = Lg[g].total;
        max_sets += Lg[g].n_sets;
    }
    for (int k = 0; k < 2; k++)
        if ((rc = c->msm_slot[k].ensure(need))) return rc;
    if ((rc = c->msm_scalars[0].ensure(count * n * sizeof(Fe)))) return rc;
    if ((rc = c->host_ws.ensure(max_sets * sizeof(XYZZ)))) return rc;
    XYZZ* h_ws = (XYZZ*)c->host_ws.p;
    if ((rc = c->ensure_aux(3 * G + 2))) return rc;
    if ((rc = c->ws_acquire(s))) return rc;
    WsGuard guard(c, s);
    int t_all = c->timer_begin("msm_total", s);
    // sort + accumulate of every group in order on one stream (they would only fight for the same wave slots side by side), the
    // reduction of group g -- a latency chain on a few waves -- on a second stream under group g + 1; two workspace slots
    hipStream_t cs = c->aux2, sa = c->aux_b, sc_ = c->aux1;
    hipEvent_t* ev = c->aux_events.data();  // [0] start, [1 + 3g] group g uploaded, [2 + 3g] accumulated, [3 + 3g] reduced
    H2_CHECK(hipEventRecord(ev[0], s));
    H2_CHECK(hipStreamWaitEvent(cs, ev[0], 0));
    H2_CHECK(hipStreamWaitEvent(sa, ev[0], 0));
    H2_CHECK(hipStreamWaitEvent(sc_, ev[0], 0));
    std::vector<const Fe*> list(count);
    size_t j0 = 0, set0 = 0;
    std::vector<CopyJob> jobs;
    for (size_t g = 0; g < G; g++) {
        for (size_t j = j0; j < j0 + groups[g]; j++) {
            Fe* dst = (Fe*)c->msm_scalars[0].p + j * n;
            jobs.push_back(CopyJob{dst, h_scalars[j], n * sizeof(Fe), j + 1 == j0 + groups[g] ? ev[1 + 3 * g] : nullptr, nullptr, false});
            list[j] = dst;
        }
        j0 += groups[g];
    }
    H2_CHECK(hipStreamSynchronize(s));  // as in msm_stream_host
    if ((rc = copier_begin(c, false, cs))) return rc;
    CopierDrain drain{c, count};
    if ((rc = copier_push(c, false, jobs))) return rc;
    j0 = 0;
    for (size_t g = 0; g < G; g++) {
        const size_t cnt = groups[g];
        if ((rc = copier_wait(c, false, j0 + cnt))) return rc;
        char* base = (char*)c->msm_slot[g & 1].p;
        H2_CHECK(hipStreamWaitEvent(sa, ev[1 + 3 * g], 0));
        if (g >= 2) H2_CHECK(hipStreamWaitEvent(sa, ev[3 + 3 * (g - 2)], 0));  // the slot is free once group g - 2 is reduced
        if ((rc = msm_stage_a(c, Lg[g], base, list.data() + j0, tab, sa))) return rc;
        if ((rc = msm_stage_b(c, Lg[g], base, d_points, sa))) return rc;
        H2_CHECK(hipEventRecord(ev[2 + 3 * g], sa));
        H2_CHECK(hipStreamWaitEvent(sc_, ev[2 + 3 * g], 0));
        if ((rc = msm_stage_c(c, Lg[g], base, h_ws + set0, sc_))) return rc;
        H2_CHECK(hipEventRecord(ev[3 + 3 * g], sc_));
        j0 += cnt;
        set0 += Lg[g].n_sets;
    }
    H2_CHECK(hipStreamWaitEvent(s, ev[3 + 3 * (G - 1)], 0));  // the reductions are in order on their stream
    c->timer_end(t_all, s);
    H2_CHECK(hipStreamSynchronize(s));
    j0 = set0 = 0;
    for (size_t g = 0; g < G; g++) {  // a group of one column has the lone MSM's window width, not the fused one
        const uint32_t spm = sets_per_msm(Lg[g].p);
        for (size_t j = 0; j < groups[g]; j++) h_out[j0 + j] = finish_msm(h_ws + set0 + j * spm, Lg[g].p);
        j0 += groups[g];
        set0 += Lg[g].n_sets;
    }
    return guard.release();
}

// measured (tools/fuse_big.py, 8 MSMs per batch, per MSM): 2^19 pairs fused 0.66 ms / pipelined 0.69 ms, 2^20 pairs fused 1.24 / pipelined 1.16
static size_t g_fuse_entries = (size_t)1 << 26, g_fuse_max_n = (size_t)1 << 19;
void msm_set_fuse_limits(size_t entries, size_t max_n) { g_fuse_entries = entries ? entries : ((size_t)1 << 26); g_fuse_max_n = max_n ? max_n : ((size_t)1 << 19); }
static bool g_fuse_small = true;
void msm_set_fuse_small(bool on) { g_fuse_small = on; }

// count MSMs over the same bases; results (XYZZ) to host memory.  tab != nullptr: the fixed-base form over tab's table
// (d_bases is then unused).  scalars_on_host: scalars[j] are host pointers.  h_bases != nullptr: the bases are the
// caller's host array and cross PCIe inside the call (d_bases unused).
int msm_batch_device(Ctx* c, const Fe* const* d_scalars, bool scalars_on_host, const Affine* d_bases, size_t n, size_t count, XYZZ* h_out,
                     hipStream_t s, const MsmTable* tab, const Affine* h_bases) {
    for (size_t j = 0; j < count; j++) h_out[j] = xyzz_identity();
    if (n == 0 || count == 0) return 0;
    if (tab && n > tab->stride) {
        set_error("msm: %zu pairs exceed the %zu points of the fixed-base table", n, tab->stride);
        return 1;
    }
    if (tab) h_bases = nullptr;
    // the entry index lives in 31 bits: split very large inputs
    const size_t max_chunk = g_max_chunk;
    std::vector<const Fe*> ptrs(count);
    std::vector<XYZZ> part(count);
    if (n > max_chunk) c->gather_off = SIZE_MAX / 2;  // several runs per MSM: their sums are added on the host
    for (size_t o = 0; o < n; o += max_chunk) {
        size_t m = n - o < max_chunk ? n - o : max_chunk;
        for (size_t j = 0; j < count; j++) ptrs[j] = d_scalars[j] + o;
        MsmTable sub;
        const MsmTable* tsub = nullptr;
        const Affine* points = d_bases ? d_bases + o : nullptr;
        if (tab) {
            sub = *tab;
            sub.table = tab->table + o;  // row j of the sub-table starts at table + j * stride + o
            tsub = &sub;
            points = sub.table;
        }
        const bool stream = scalars_on_host && g_stream_chunks > 1 && copier_ready(c, false);
        if (stream && count == 1 && m >= g_stream_min_n) {  // a lone host-resident MSM: chunks stream in under the work
            int rc = msm_stream_host(c, ptrs[0], h_bases ? h_bases + o : nullptr, points, tsub, m, part.data(), s);
            if (rc) return rc;
            xyzz_add(h_out[0], part[0]);
            continue;
        }
        if (h_bases) {  // whole upload ahead of the run
            int rc = c->msm_bases.ensure(m * sizeof(Affine));
            if (rc) return rc;
            if ((rc = c->ws_acquire(s))) return rc;
            H2_CHECK(hipMemcpyAsync(c->msm_bases.p, h_bases + o, m * sizeof(Affine), hipMemcpyHostToDevice, s));
            points = (const Affine*)c->msm_bases.p;
        }
        // up to 2^19 pairs each: fused runs of at most 2^26 entries, MSM_MAX_C1 << MSM_MAX_L buckets and MSM_MAX_C1
        // bucket sets; larger MSMs: pipelined over streams
        const MsmPlan fp = make_plan(m, true, tsub);
        const size_t per_msm = m * fp.W;
        const size_t sets = fp.shared ? 1 : fp.W;
        size_t fuse_max = per_msm ? g_fuse_entries / per_msm : 0;
        const size_t by_buckets = (((size_t)MSM_MAX_C1 << MSM_MAX_L) >> fp.cb) / sets, by_sets = MSM_MAX_C1 / sets;
        if (fuse_max > by_buckets) fuse_max = by_buckets;
        if (fuse_max > by_sets) fuse_max = by_sets;
        if (g_fuse_small && count > 1 && m <= g_fuse_max_n && fuse_max >= 2) {
            // (up to seven host columns the whole upload first is faster than groups: 6 x 2^17 1.64 against 1.72 ms, 10: 2.59 / 2.23, 16: 4.01 / 3.23 -- tools/fused_host_sweep.py)
            if (stream && count * m >= g_stream_min_n && (count >= 8 || !g_stream_chunks_default)) {
                int rc = msm_fused_groups_host(c, ptrs.data(), points, tsub, m, count, fuse_max, part.data(), s);
                if (rc) return rc;
            } else {
                for (size_t j0 = 0; j0 < count; j0 += fuse_max) {
                    const size_t g = count - j0 < fuse_max ? count - j0 : fuse_max;
                    int rc = g == 1 ? msm_batch_chunk(c, ptrs.data() + j0, scalars_on_host, points, tsub, m, 1, part.data() + j0, s)
                                    : msm_fused_chunk(c, ptrs.data() + j0, scalars_on_host, points, tsub, m, g, part.data() + j0, s);
                    if (rc) return rc;
                }
            }
        } else {
            int rc = msm_batch_chunk(c, ptrs.data(), scalars_on_host, points, tsub, m, count, part.data(), s);
            if (rc) return rc;
        }
        for (size_t j = 0; j < count; j++) xyzz_add(h_out[j], part[j]);
    }
    return 0;
}

// Sum of coeffs[i]*bases[i] for device-resident inputs; result (XYZZ) to host memory.
int msm_device(Ctx* c, const Fe* d_scalars, const Affine* d_bases, size_t n, XYZZ* h_out, hipStream_t s, const MsmTable* tab) {
    return msm_batch_device(c, &d_scalars, false, d_bases, n, 1, h_out, s, tab, nullptr);
}

// Build the fixed-base table for n device-resident points: rows 0..W-1 of n points each, row j = 2^(pos_j) * P with pos_j the
// first bit of window j (MsmPlan).
// Row 0 is a copy of the points.  Rows are built one from the other (c doublings in XYZZ, then one batched
// normalisation back to affine), in slices of 2^22 points so the XYZZ scratch stays at 512 MB.
int msm_table_build(Ctx* c, const Affine* d_points, size_t n, uint32_t cw, Affine* d_table, hipStream_t s) {
    const uint32_t W = (255 + cw - 1) / cw, q = 255 - W * (cw - 1);
    if (d_table != d_points) H2_CHECK(hipMemcpyAsync(d_table, d_points, n * sizeof(Affine), hipMemcpyDeviceToDevice, s));  // row 0 = the points
    const size_t slice = (size_t)1 << 22;
    const size_t m_max = n < slice ? n : slice;
    int rc = c->ecfft_ws.ensure(m_max * sizeof(XYZZ));
    if (rc) return rc;
    if ((rc = c->ws_acquire(s))) return rc;
    WsGuard guard(c, s);
    XYZZ* tmp = (XYZZ*)c->ecfft_ws.p;
    for (uint32_t j = 1; j < W; j++) {
        for (size_t o = 0; o < n; o += slice) {
            const size_t m = n - o < slice ? n - o : slice;
            hipLaunchKernelGGL(msm_table_step_kernel, dim3((uint32_t)((m + 255) / 256)), dim3(256), 0, s, d_table + (size_t)(j - 1) * n + o, tmp,
                               (uint32_t)m, (j - 1) < q ? cw : cw - 1);  // window j starts width(j - 1) bits above window j - 1
            H2_CHECK(hipGetLastError());
            if ((rc = ec_normalize_device(tmp, d_table + (size_t)j * n + o, m, s))) return rc;
        }
    }
    return guard.release();
}

}  // namespace h2
