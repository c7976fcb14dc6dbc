// ntt.hip -- radix-2^s multi-pass NTT over BN254 Fr for gfx950.
//
// Computes exactly what halo2_proofs::arithmetic::best_fft computes for G = Fr
// (halo2_proofs/src/arithmetic.rs:171-234): a[i] <- sum_j a[j] * omega^(i*j), natural order in
// and out, in place, for omega of exact order 2^log_n.  Field arithmetic is exact, so any
// butterfly schedule gives the reference's limbs bit for bit.
//
// Schedule (decimation in frequency, one HBM round trip per pass): pass t takes blocks of length
// M, views each as R x (M/R) (R = 2^s rows at stride M/R), loads J adjacent columns into LDS,
// runs the R-point DFT per column in LDS, multiplies output row k of column lo by w_M^(k*lo) and
// stores it back to row k.  After the strided passes the result is in digit-reversed order by
// blocks; the last pass (M = R, contiguous blocks) undoes that on its store, writing J consecutive
// outputs per row so stores stay coalesced.  The pointwise steps that surround best_fft in
// poly/domain.rs (zeta coset scaling + zero padding on the way in, :246-247; 1/n and zeta^-1
// scaling on the way out, :294, :355-360) are fused into the first load and the last store.
//
// Two plans.  Three (or more) passes of 2^7..2^9-point tiles, 256 lanes and J = 4 or 2 columns at
// once (ntt_strided_kernel / ntt_final_kernel), for every size but 2^20..2^22; those take two
// passes of 2^10 / 2^11-point tiles (ntt2_*_kernel: R / 4 lanes, one column at a time, first and
// last butterfly rounds in registers): 13 instead of 14.5 field multiplications per element and
// one HBM round trip fewer.  Both are bound by VALU issue (the 9 x 29-bit multiplier is ~80 % of
// the instructions), so what matters is the instruction count and keeping four waves per SIMD
// busy: the tile twiddles w_R^i come from a per-domain table in global memory (every workgroup
// reads the same few KB; keeping them out of LDS is what lets a fourth 256-point workgroup, or a
// second 2^11-point one, share a CU), the LDS image is swizzled conflict-free (lds_swz), and rounds
// whose groups stay inside one wave's 256 points synchronise the wave only.
#include <stddef.h>
#include <string.h>

#include <vector>

#include "engine.h"
#include "fieldu.h"

namespace h2 {

struct NttPass {
    const Fe* src;
    Fe* dst;
    const Fu* tw_lo;
    const Fu* tw_hi;
    const Fu* tw_full;   // this pass's inter-pass twiddles as one table -- two-pass plan: omega^(k lo) at [lo << s | k]; strided passes of the
                         // other plan: w_M^(k lo) at [k << log_l | lo] -- or null (tw_pow then combines the two-level table: one multiplication more)
    const Fu* stage_tw;  // this pass's tile twiddles w_R^i, i < R / 2 (every workgroup reads the same few KB: L1 / L2 hits)
    uint64_t in_len;
    uint32_t log_n, log_m, s, log_j, lo_bits;
    uint32_t first, in_scale;
    uint32_t out_scale;  // 0: none, 1: out3[oi % 3], 2: out3[0] for every output
    uint32_t n_prev;
    uint32_t xgroup;     // two-pass plan, one column / block per workgroup: 2^xgroup neighbouring ones go to workgroups b, b + 8, ... (one XCD); 0: in order
    uint32_t prev_s[4];
    Fu in3[3], out3[3];  // I-form constants
    // batched launch (gridDim.y transforms of the same size): transform y reads srcs[y], writes dsts[y]
    const Fe* const* srcs;
    Fe* const* dsts;
};

// the kernel argument itself is left untouched (a modified copy of the struct would leave the scalar registers)
__device__ __forceinline__ const Fe* ntt_src(const NttPass& p) { return p.srcs ? p.srcs[blockIdx.y] : p.src; }
__device__ __forceinline__ Fe* ntt_dst(const NttPass& p) { return p.dsts ? p.dsts[blockIdx.y] : p.dst; }

#define NTT_THREADS 256
#ifndef NTT2_OWN_FINAL
#define NTT2_OWN_FINAL 0  // the same in pass 2, whose rows are contiguous: the lanes' 32-byte loads scatter and the pass measures 1 % slower (A/B builds)
#endif
#ifndef NTT2_OWN_ROWS
#define NTT2_OWN_ROWS 1  // two-pass plan, pass 1: a lane loads the rows of its own points (dft_col<true>); 0 = rows t + m R/4 and a workgroup barrier (A/B builds)
#endif

// Arithmetic: data stays in the reference's E-form (value == a * 2^256 mod r) as lazily reduced
// 9 x 29-bit limbs (fieldu.h); twiddles and scale constants are I-form, so every
// fu_mul(data, twiddle) is again E-form and the last multiply of a pass doubles as the exact
// reduction back to canonical limbs (fu_mul_canon).

__device__ __forceinline__ Fu tw_pow(const NttPass& p, uint64_t e) {
    uint32_t lo = (uint32_t)(e & ((1ull << p.lo_bits) - 1));
    uint64_t hi = e >> p.lo_bits;
    Fu a = p.tw_lo[lo];
    if (hi) a = fu_mul<FrUA>(a, p.tw_hi[hi]);
    return a;
}

__device__ __forceinline__ uint32_t bitrev(uint32_t k, uint32_t s) { return __brev(k) >> (32 - s); }

// One of three kernel-argument constants by a lane-dependent residue.  The constants stay where kernel arguments live
// (scalar registers) and the pick is bit-mask arithmetic: a compare-and-select chain is turned by LLVM into an indexed
// load from a scratch copy of the array, and reading them through a pointer costs a vector load per use.
__device__ __forceinline__ Fu pick3(const Fu (&c)[3], uint32_t m) {
    const int32_t m1 = -(int32_t)(m == 1), m2 = -(int32_t)(m == 2);
    Fu r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = c[0].l[i] ^ ((c[0].l[i] ^ c[1].l[i]) & m1) ^ ((c[0].l[i] ^ c[2].l[i]) & m2);
    return r;
}
// the constant of output index oi (< 2^28): 1 (forward transform), one constant (1/n), or one of three by oi mod 3 (coset)
__device__ __forceinline__ Fu out_const(const NttPass& p, uint64_t oi) {
    if (p.out_scale == 0) return fu_one_i<FrUA>();
    if (p.out_scale == 2) return p.out3[0];
    return pick3(p.out3, (uint32_t)oi % 3u);
}

// LDS image index of point e.  A point is 9 dwords, so the bank of its dword c is (9 e + c) mod 32: the 32 lanes of a DS
// access group are conflict-free exactly when their e differ mod 32.  Every access pattern of these kernels varies five
// bit positions of e across such a group -- {2..6}, {0,1,4,5,6}, {0..3,6} in the rounds of half-size 1, 4, 16, five
// consecutive positions in the bit-reversed loads, {0,1,2,s,s+1} in the row-major stores -- so the low five bits are
// XORed with a GF(2)-linear image of the higher ones whose columns c5, c6, ... continue the sequence e0..e4 by
// c[k+5] = c[k] ^ c[k+2] ^ c[k+4]: any five consecutive columns are independent, and so is each of the other sets.
__device__ __forceinline__ uint32_t lds_swz(uint32_t e) {
    const uint32_t t = e >> 5;
    uint32_t m = (0u - (t & 1)) & 21u;
    m ^= (0u - ((t >> 1) & 1)) & 31u;
    m ^= (0u - ((t >> 2) & 1)) & 11u;
    m ^= (0u - ((t >> 3) & 1)) & 22u;
    m ^= (0u - ((t >> 4) & 1)) & 25u;
    m ^= (0u - ((t >> 5) & 1)) & 7u;
    m ^= (0u - ((t >> 6) & 1)) & 14u;
    m ^= (0u - ((t >> 7) & 1)) & 28u;
    return e ^ m;
}

// A lane's radix-4 group of half-size h <= 64 lies inside the 256 consecutive points its own wave works on (lane q of a
// round takes point group q), so rounds up to there hand their results on within the wave: the LDS image only has to be
// ordered for the wave (DS operations of one wave complete in issue order), not for the workgroup.
__device__ __forceinline__ void round_sync(uint32_t next_log_h, bool next_is_radix4) {
    if (next_is_radix4 && next_log_h <= 6) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// R-point DFT on J columns held in LDS as x[col*R + bitrev(r)] on entry, x[col*R + k] on exit (decimation in
// time).  Two butterfly stages per LDS round trip: a lane takes the four elements base + {0, h, 2h, 3h}, runs
// the stage of half-size h on (x0,x1), (x2,x3) and the stage of half-size 2h on (y0,y2), (y1,y3) in registers.
// Every stage multiplies by its twiddle, w^0 = 1 included (keeps |value| growing by at most ~1.2 r per stage
// instead of doubling), except in the first round, whose inputs are fresh: stage 0's twiddles are all 1 and so
// is one of stage 1's two (four fresh values add up to < 4.5 r; the canonicalising store accepts 16 r); limbs are
// re-normalised after each stage pair, which keeps every fu_mul operand below 2^30.  An odd s ends with one
// single stage whose loose outputs the canonicalising store accepts.
__device__ __forceinline__ void dft_lds(Fu* x, const Fu* wtab, uint32_t s, uint32_t log_j) {
    const uint32_t R = 1u << s;
    uint32_t log_h = 0;
    const uint32_t nq = (R << log_j) >> 2;
    for (; log_h + 2 <= s; log_h += 2) {
        const uint32_t h = 1u << log_h;
        const uint32_t d1 = lds_swz(h), d2 = lds_swz(2 * h), d3 = d1 ^ d2;
        for (uint32_t q = threadIdx.x; q < nq; q += NTT_THREADS) {
            uint32_t col = q >> (s - 2);
            uint32_t i = q & ((R >> 2) - 1);
            uint32_t off = i & (h - 1);
            uint32_t blk = i >> log_h;
            const uint32_t base = lds_swz((col << s) + (blk << (log_h + 2)) + off);  // the bits of j * h are clear: swz(base + j h) = swz(base) ^ swz(j h)
            const uint32_t e1 = base ^ d1, e2 = base ^ d2, e3 = base ^ d3;
            Fu x0 = x[base], x1 = x[e1], x2 = x[e2], x3 = x[e3];
            if (log_h) {
                Fu wa = wtab[off << (s - 1 - log_h)];
                x1 = fu_mul<FrUA>(x1, wa);
                x3 = fu_mul<FrUA>(x3, wa);
            }
            Fu y0 = fu_add(x0, x1), y1 = fu_sub(x0, x1), y2 = fu_add(x2, x3), y3 = fu_sub(x2, x3);
            Fu u2 = log_h ? fu_mul<FrUA>(y2, wtab[off << (s - 2 - log_h)]) : y2;  // first round: the twiddle is 1 (dft_col's note on the bound)
            Fu u3 = fu_mul<FrUA>(y3, wtab[(off + h) << (s - 2 - log_h)]);
            x[base] = fu_norm(fu_add(y0, u2));
            x[e2] = fu_norm(fu_sub(y0, u2));
            x[e1] = fu_norm(fu_add(y1, u3));
            x[e3] = fu_norm(fu_sub(y1, u3));
        }
        round_sync(log_h + 2, log_h + 4 <= s);
    }
    if (log_h < s) {  // odd s: last stage on its own
        const uint32_t h = 1u << log_h;
        const uint32_t nbf = (R << log_j) >> 1;
        for (uint32_t bf = threadIdx.x; bf < nbf; bf += NTT_THREADS) {
            uint32_t col = bf >> (s - 1);
            uint32_t i = bf & ((R >> 1) - 1);
            uint32_t off = i & (h - 1);
            uint32_t blk = i >> log_h;
            uint32_t i0 = lds_swz((col << s) + (blk << (log_h + 1)) + off);
            uint32_t i1 = i0 ^ lds_swz(h);
            Fu a = x[i0], t = x[i1];
            if (log_h) t = fu_mul<FrUA>(t, wtab[off << (s - 1 - log_h)]);
            x[i0] = fu_add(a, t);
            x[i1] = fu_sub(a, t);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ Fu ntt_load(const NttPass& p, const Fe* src, uint64_t gi) {
    if (p.first) {
        if (gi >= p.in_len) return fu_zero();
        Fu v = fu_slice(src[gi]);
        if (p.in_scale) {
            const uint32_t m = (uint32_t)gi % 3u;  // gi < 2^28
            if (m) v = fu_mul<FrUA>(v, pick3(p.in3, m));
        }
        return v;
    }
    return fu_slice(src[gi]);
}

extern __shared__ __align__(16) unsigned char ntt_lds_raw[];

__global__ void __launch_bounds__(NTT_THREADS) ntt_strided_kernel(NttPass p) {
    const Fe* src = ntt_src(p);
    Fe* dst = ntt_dst(p);
    Fu* x = reinterpret_cast<Fu*>(ntt_lds_raw);
    const uint32_t R = 1u << p.s, J = 1u << p.log_j;
    const uint32_t log_l = p.log_m - p.s;                 // L = M/R columns per block
    const uint32_t log_gpb = log_l - p.log_j;             // column groups per block
    const uint64_t u = blockIdx.x;
    const uint64_t base = (u >> log_gpb) << p.log_m;
    const uint64_t lo0 = (u & ((1ull << log_gpb) - 1)) << p.log_j;
    for (uint32_t idx = threadIdx.x; idx < (R << p.log_j); idx += NTT_THREADS) {
        uint32_t r = idx >> p.log_j, jj = idx & (J - 1);
        x[lds_swz((jj << p.s) + bitrev(r, p.s))] = ntt_load(p, src, base + ((uint64_t)r << log_l) + lo0 + jj);
    }
    __syncthreads();
    dft_lds(x, p.stage_tw, p.s, p.log_j);
    for (uint32_t idx = threadIdx.x; idx < (R << p.log_j); idx += NTT_THREADS) {
        uint32_t k = idx >> p.log_j, jj = idx & (J - 1);
        uint64_t lo = lo0 + jj;
        uint64_t e = ((uint64_t)k * lo) << (p.log_n - p.log_m);  // w_M^(k*lo) = omega^((N/M)*k*lo)
        const Fu w = p.tw_full ? p.tw_full[((uint64_t)k << log_l) + lo] : tw_pow(p, e);
        dst[base + ((uint64_t)k << log_l) + lo] = fu_mul_canon<FrUA>(x[lds_swz((jj << p.s) + k)], w);
    }
}

__global__ void __launch_bounds__(NTT_THREADS) ntt_final_kernel(NttPass p) {
    const Fe* src = ntt_src(p);
    Fe* dst = ntt_dst(p);
    Fu* x = reinterpret_cast<Fu*>(ntt_lds_raw);
    const uint32_t R = 1u << p.s, J = 1u << p.log_j;
    const uint64_t g = blockIdx.x;
    for (uint32_t idx = threadIdx.x; idx < (R << p.log_j); idx += NTT_THREADS) {
        uint32_t jj = idx >> p.s, r = idx & (R - 1);
        // block whose digit-reversed index is g*J + jj
        uint64_t v = (g << p.log_j) + jj, bi = 0;
        for (uint32_t t = 0; t < p.n_prev; t++) {
            bi = (bi << p.prev_s[t]) | (v & ((1ull << p.prev_s[t]) - 1));
            v >>= p.prev_s[t];
        }
        x[lds_swz((jj << p.s) + bitrev(r, p.s))] = ntt_load(p, src, (bi << p.s) + r);
    }
    __syncthreads();
    dft_lds(x, p.stage_tw, p.s, p.log_j);
    const uint32_t log_nb = p.log_n - p.s;
    for (uint32_t idx = threadIdx.x; idx < (R << p.log_j); idx += NTT_THREADS) {
        uint32_t k = idx >> p.log_j, jj = idx & (J - 1);
        uint64_t oi = ((uint64_t)k << log_nb) + (g << p.log_j) + jj;
        const Fu v = x[lds_swz((jj << p.s) + k)];
        dst[oi] = p.out_scale ? fu_mul_canon<FrUA>(v, out_const(p, oi)) : fu_canon_fast<FrUA>(v);  // no constant: reduce directly
    }
}

// E-form canonical Fe -> I-form canonical limbs (x * 2^5 mod r), sliced
__host__ __device__ __forceinline__ Fu fu_i_from_fe(const Fe& x) {
    Fe t = x;
    for (int k = 0; k < 5; k++) t = fe_dbl<FrP>(t);
    return fu_slice(t);
}

// ---- two-pass plan for 2^20 <= n <= 2^22: tiles of R = 2^10 or 2^11 points -------------------------------------------
// One HBM round trip and one inter-pass twiddle fewer than three passes of 2^7..2^8-point tiles.  A 2^11-point image is
// 72 KB of limbs, so a CU holds two workgroups of R / 4 lanes; each runs its columns one after the other through its one
// image (two columns per workgroup: 64-byte rows).
// One column's R-point DFT, registers to registers: lane t (of R / 4) brings rows t + m R/4 (m = 0..3) and leaves with
// outputs t + m R/4.  Bit-reversed, its four rows are the consecutive points 4 i' .. 4 i' + 3 (i' = bitrev(t)) in the order
// m = 0, 2, 1, 3, so the first stage pair (or, for odd s, the lone first stage) runs on the loaded values; and the last
// round's group of half-size R / 4 is exactly {t + m R/4}, so its results never go back to LDS.  In between, the rounds
// exchange through the image x as dft_lds does.
// `quarter` (uniform): rows R / 4 and up are zero (a coefficient vector padded to four times its length, coeff_to_extended's
// first pass), i.e. v[1] = v[2] = v[3] = 0 in every lane: the first stage pair copies v[0] to its four outputs.
// OWN: the lane brought rows bitrev(t) + m R/4 instead, so its points are 4 t .. 4 t + 3 -- inside the 256 points its own wave works on
// in the rounds of half-size <= 64 -- and the first stage's results need no workgroup barrier before the next round reads them.
template <bool OWN = false>
__device__ __forceinline__ void dft_col(Fu* x, const Fu* wtab, uint32_t s, const Fu (&v)[4], Fu (&res)[4], bool quarter = false) {
    const uint32_t t = threadIdx.x;
    const uint32_t b0 = lds_swz((OWN ? t : bitrev(t, s - 2)) << 2);  // swz(4 i' + j) = swz(4 i') ^ j
    // stages 0 and 1 on the loaded values.  Stage 0's twiddles are all 1 and stage 1's are 1 and w_4: the product by 1 is not formed (the sum
    // of four fresh values stays below 4.5 p -- the rounds that follow add at most 2.3 p each, the canonicalising store takes 16 p -- and
    // below 2^31 per limb).  An odd s ends with a lone stage instead (below): 19 instead of 20 multiplications per lane at s = 11.
    if (quarter) {
        const Fu n0 = fu_norm(v[0]);
        x[b0] = n0;
        x[b0 ^ 1] = n0;
        x[b0 ^ 2] = n0;
        x[b0 ^ 3] = n0;
    } else {
        const Fu y0 = fu_add(v[0], v[2]), y1 = fu_sub(v[0], v[2]), y2 = fu_add(v[1], v[3]), y3 = fu_sub(v[1], v[3]);
        const Fu u3 = fu_mul<FrUA>(y3, wtab[1u << (s - 2)]);
        x[b0] = fu_norm(fu_add(y0, y2));
        x[b0 ^ 2] = fu_norm(fu_sub(y0, y2));
        x[b0 ^ 1] = fu_norm(fu_add(y1, u3));
        x[b0 ^ 3] = fu_norm(fu_sub(y1, u3));
    }
    uint32_t log_h = 2;
    if (OWN) round_sync(log_h, log_h + 2 <= s);
    else __syncthreads();
    for (; log_h + 2 <= s; log_h += 2) {
        const bool last = log_h + 2 == s;
        const uint32_t h = 1u << log_h;
        const uint32_t d1 = lds_swz(h), d2 = lds_swz(2 * h), d3 = d1 ^ d2;
        const uint32_t off = t & (h - 1), blk = t >> log_h;
        const uint32_t base = lds_swz((blk << (log_h + 2)) + off);
        const uint32_t e1 = base ^ d1, e2 = base ^ d2, e3 = base ^ d3;
        Fu x0 = x[base], x1 = x[e1], x2 = x[e2], x3 = x[e3];
        const Fu wa = wtab[off << (s - 1 - log_h)];
        x1 = fu_mul<FrUA>(x1, wa);
        x3 = fu_mul<FrUA>(x3, wa);
        const Fu y0 = fu_add(x0, x1), y1 = fu_sub(x0, x1), y2 = fu_add(x2, x3), y3 = fu_sub(x2, x3);
        const Fu u2 = fu_mul<FrUA>(y2, wtab[off << (s - 2 - log_h)]);
        const Fu u3 = fu_mul<FrUA>(y3, wtab[(off + h) << (s - 2 - log_h)]);
        if (last) {  // loose limbs (< 1.5 * 2^30 in magnitude): the canonicalising multiply that follows normalises its operand itself
            res[0] = fu_add(y0, u2);
            res[2] = fu_sub(y0, u2);
            res[1] = fu_add(y1, u3);
            res[3] = fu_sub(y1, u3);
            break;
        }
        x[base] = fu_norm(fu_add(y0, u2));
        x[e2] = fu_norm(fu_sub(y0, u2));
        x[e1] = fu_norm(fu_add(y1, u3));
        x[e3] = fu_norm(fu_sub(y1, u3));
        round_sync(log_h + 2, log_h + 4 <= s);
    }
    if (s & 1) {  // the lone last stage, half-size R / 2: the lane's own outputs t + m R/4 pair up as (0, 2) and (1, 3)
        const uint32_t q = 1u << (s - 2);
        const Fu a0 = x[lds_swz(t)], a1 = x[lds_swz(t + q)];
        const Fu t0 = fu_mul<FrUA>(x[lds_swz(t + 2 * q)], wtab[t]);
        const Fu t1 = fu_mul<FrUA>(x[lds_swz(t + 3 * q)], wtab[t + q]);
        res[0] = fu_add(a0, t0);
        res[2] = fu_sub(a0, t0);
        res[1] = fu_add(a1, t1);
        res[3] = fu_sub(a1, t1);
    }
    __syncthreads();  // the image is free for the next column
}

template <class F>
__device__ __forceinline__ void four(F&& f) {
    f(0u);
    f(1u);
    f(2u);
    f(3u);
}

// workgroup b of a grid whose size is a multiple of 8 << g  ->  the item it takes, such that items i, i + 1, ..., i + 2^g - 1 (i a multiple
// of 2^g) are taken by workgroups b, b + 8, ..., b + 8 (2^g - 1): the same XCD under the round-robin dispatch
__device__ __forceinline__ uint64_t xcd_group(uint32_t b, uint32_t g) {
    const uint32_t q = b >> (3 + g), r = b & ((8u << g) - 1);
    return ((uint64_t)(q * 8 + (r & 7)) << g) | (r >> 3);
}

// pass 1 of 2: columns lo0 .. lo0 + J - 1 of the R x L view (L = N / R), rows at stride L.  A column's 32-byte pieces of a
// row are read and written one column at a time.  Measured (FETCH_SIZE): the pass fetches 2.04 x what it consumes -- every
// piece brings its 64-byte sector and the neighbouring column asks for it again after 16 MB have passed through the XCD's
// L2 -- which the Infinity Cache absorbs; holding the second column's pieces in registers meanwhile cost more in spills.
__global__ void __launch_bounds__(512, 4) ntt2_strided_kernel(NttPass p) {
    const Fe* src = ntt_src(p);
    Fe* dst = ntt_dst(p);
    Fu* x = reinterpret_cast<Fu*>(ntt_lds_raw);
    const uint32_t T = blockDim.x;  // R / 4
    const uint32_t log_l = p.log_n - p.s;
    // One column per workgroup.  Neighbouring columns share their rows' cache lines (a row piece is 32 B of a 128-byte line), and workgroups
    // are dealt to the eight XCDs -- eight L2s -- in turn: 2^xgroup neighbouring columns go to workgroups b, b + 8, b + 16, ..., which land on
    // ONE XCD at nearly the same time, so a line is fetched into (and written back from) one L2 once.  Measured against two columns per
    // workgroup one after the other, whose second column found its sectors evicted: 2^20 0.143 -> 0.120 ms, 2^21 0.247 -> 0.219, 2^22 0.496 -> 0.462.
    uint64_t lo0 = (uint64_t)blockIdx.x << p.log_j;
    if (p.xgroup) lo0 = xcd_group(blockIdx.x, p.xgroup);
    const bool quarter = p.first && (p.in_len << 2) <= (1ull << p.log_n);  // rows T = R / 4 and up lie at T * L = N / 4 and beyond
#pragma unroll 1
    for (uint32_t c = 0; c < (1u << p.log_j); c++) {
        const uint64_t lo = lo0 + c;
        Fu v[4], y[4];
        four([&](uint32_t m) __attribute__((always_inline)) {
            const uint64_t r = (NTT2_OWN_ROWS ? bitrev(threadIdx.x, p.s - 2) : threadIdx.x) + m * T;  // rows are a stride apart whichever lane takes them: the lane takes those of its own points
            v[m] = ntt_load(p, src, (r << log_l) + lo);
        });
        dft_col<NTT2_OWN_ROWS>(x, p.stage_tw, p.s, v, y, quarter);
        four([&](uint32_t m) __attribute__((always_inline)) {
            const uint64_t k = threadIdx.x + m * T;
            const Fu w = p.tw_full ? p.tw_full[(lo << p.s) + k] : tw_pow(p, k * lo);  // w_N^(k * lo)
            dst[(k << log_l) + lo] = fu_mul_canon<FrUA>(y[m], w);
        });
    }
}

// pass 2 of 2: contiguous blocks v0 .. v0 + J - 1 of R points; output k of block v goes to k * (N / R) + v
__global__ void __launch_bounds__(512, 4) ntt2_final_kernel(NttPass p) {
    const Fe* src = ntt_src(p);
    Fe* dst = ntt_dst(p);
    Fu* x = reinterpret_cast<Fu*>(ntt_lds_raw);
    const uint32_t T = blockDim.x;  // R / 4
    const uint32_t log_nb = p.log_n - p.s;
    uint64_t v0 = (uint64_t)blockIdx.x << p.log_j;
    if (p.xgroup) v0 = xcd_group(blockIdx.x, p.xgroup);  // neighbouring blocks write neighbouring 32-byte pieces: pairs on one XCD (up to 2^21 points: -2 to -4 %; at 2^22 +2 %, not used)
#pragma unroll 1
    for (uint32_t c = 0; c < (1u << p.log_j); c++) {
        const uint64_t vb = v0 + c;
        Fu v[4], y[4];
        four([&](uint32_t m) __attribute__((always_inline)) {
            const uint64_t r = (NTT2_OWN_FINAL ? bitrev(threadIdx.x, p.s - 2) : threadIdx.x) + m * T;
            v[m] = fu_slice(src[(vb << p.s) + r]);
        });
        dft_col<NTT2_OWN_FINAL>(x, p.stage_tw, p.s, v, y);
        four([&](uint32_t m) __attribute__((always_inline)) {
            const uint64_t k = threadIdx.x + m * T;
            const uint64_t oi = (k << log_nb) + vb;
            dst[oi] = p.out_scale ? fu_mul_canon<FrUA>(y[m], out_const(p, oi)) : fu_canon_fast<FrUA>(y[m]);  // no constant: reduce directly
        });
    }
}

// the two-pass plan's inter-pass twiddles as one table: entry (lo << s | k) = omega^(k * lo)
__global__ void __launch_bounds__(256) full_twiddle_build_kernel(NttPass p, Fu* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >> p.log_n) return;
    const uint64_t k = i & ((1ull << p.s) - 1), lo = i >> p.s;
    out[i] = tw_pow(p, k * lo);
}

// a strided pass's inter-pass twiddles as one table: entry (k << log_l | lo) = w_M^(k * lo), M = 2^log_m = R * L
__global__ void __launch_bounds__(256) pass_twiddle_build_kernel(NttPass p, Fu* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >> p.log_m) return;
    const uint32_t log_l = p.log_m - p.s;
    const uint64_t lo = i & ((1ull << log_l) - 1), k = i >> log_l;
    out[i] = tw_pow(p, (k * lo) << (p.log_n - p.log_m));
}

__global__ void stage_twiddle_build_kernel(Fe omega, uint32_t shift, uint32_t count, Fu* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = fu_i_from_fe(fe_pow_u64<FrP>(omega, (uint64_t)i << shift));
}

// n = 1: best_fft is the identity; only the fused scales apply
__global__ void ntt_n1_kernel(NttPass p) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        Fu v = ntt_load(p, ntt_src(p), 0);
        ntt_dst(p)[0] = fu_mul_canon<FrUA>(v, p.out_scale ? p.out3[0] : fu_one_i<FrUA>());
    }
}

__global__ void twiddle_build_kernel(Fe omega, Fu* lo, uint32_t n_lo, Fu* hi, uint32_t n_hi, uint32_t lo_bits) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_lo) {
        lo[i] = fu_i_from_fe(fe_pow_u64<FrP>(omega, i));
    } else if (i < n_lo + n_hi) {
        uint32_t j = i - n_lo;
        hi[j] = fu_i_from_fe(fe_pow_u64<FrP>(omega, (uint64_t)j << lo_bits));
    }
}

// every table of the twiddle cache (the caller has made sure no kernel still reads them)
void ntt_twiddles_free(Ctx* c) {
    for (auto& kv : c->twiddles) {
        TwiddleTable& t = kv.second;
        (void)hipFree(t.lo);
        (void)hipFree(t.hi);
        for (int k = 0; k < 13; k++) (void)hipFree(t.stage_of[k]);
        for (int k = 0; k < H2_TW_FULL; k++) (void)hipFree(t.full[k]);
        (void)hipFree(t.lo_scaled);
    }
    c->twiddles.clear();
    c->tw_full_bytes = 0;
}

static uint64_t g_ntt_full_budget = (uint64_t)4 << 30;  // HALO2_HIP_NTT_TWIDDLE_MB: HBM the full inter-pass tables may take per device (of 288 GB)
void ntt_set_full_twiddle_budget(uint64_t bytes) { g_ntt_full_budget = bytes; }
static size_t g_ntt_batch_bytes = (size_t)2 << 30;  // columns + workspace one batched launch may span (ntt_device_batch)
void ntt_set_batch_bytes(uint64_t bytes) { g_ntt_batch_bytes = bytes ? (size_t)bytes : (size_t)2 << 30; }
// Strided passes of the three-pass plan read their inter-pass twiddles from a table of up to 2^this entries (36 B each: 0.3 / 0.6 GB per domain
// and direction at 2^23 / 2^24 for the first pass's, 2.4 MB for the second's) while the budget lasts: one multiplication per point fewer than
// combining the two-level table (2^23 -5 %, 2^24 -2.5 %; at 2^25 and beyond the pass has no memory bandwidth to spare for it and nothing is gained).
static uint32_t g_ntt_full_max_log_m = 24;
void ntt_set_full_max_log_m(uint32_t v) { g_ntt_full_max_log_m = v ? v : 24; }
static bool g_ntt_fold_tables = true;  // the inverse's 1/n rides in a scaled copy of the first pass's table (A/B: h2hip_debug_set_ntt_fold_tables)
void ntt_set_fold_tables(bool on) { g_ntt_fold_tables = on; }
static int g_ntt2_log_j = -1;  // tuning: columns per workgroup of the two-pass kernels (-1 = the plan's choice)
void ntt_set_two_pass_log_j(int v) { g_ntt2_log_j = v; }

static int get_twiddles(Ctx* c, const Fe& omega, uint32_t log_n, hipStream_t s, TwiddleTable* out) {
    TwiddleKey key;
    for (int i = 0; i < 8; i++) key.omega[i] = omega.l[i];
    key.log_n = log_n;
    auto it = c->twiddles.find(key);
    if (it != c->twiddles.end()) {
        *out = it->second;
        return 0;
    }
    TwiddleTable t;
    t.lo_bits = log_n < 10 ? log_n : 10;
    uint32_t n_lo = 1u << t.lo_bits, n_hi = 1u << (log_n - t.lo_bits);
    H2_CHECK(hipMalloc((void**)&t.lo, (size_t)n_lo * sizeof(Fu)));
    H2_CHECK(hipMalloc((void**)&t.hi, (size_t)n_hi * sizeof(Fu)));
    uint32_t total = n_lo + n_hi;
    hipLaunchKernelGGL(twiddle_build_kernel, dim3((total + 255) / 256), dim3(256), 0, s, omega, t.lo, n_lo, t.hi, n_hi, t.lo_bits);
    H2_CHECK(hipGetLastError());
    H2_CHECK(hipStreamSynchronize(s));  // built once per (omega, log_n); later calls may use another stream
    if (c->twiddles.size() >= 64) {  // bounded cache: a prover uses a handful of domains
        H2_CHECK(hipDeviceSynchronize());
        ntt_twiddles_free(c);
    }
    c->twiddles[key] = t;
    *out = t;
    return 0;
}

__global__ void twiddle_scale_kernel(const Fu* __restrict__ in, Fu* __restrict__ out, uint32_t n, Fu scale_i) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fu_norm(fu_mul<FrU>(in[i], scale_i));  // I * I -> I
}

// the domain's `lo` table times `scale` (cached; one constant per domain: a second one replaces it)
static int get_scaled_lo(Ctx* c, const Fe& omega, uint32_t log_n, const Fe& scale, hipStream_t s, const Fu** out) {
    TwiddleKey key;
    for (int i = 0; i < 8; i++) key.omega[i] = omega.l[i];
    key.log_n = log_n;
    auto it = c->twiddles.find(key);
    if (it == c->twiddles.end()) {
        set_error("ntt: scaled twiddles asked before the domain's table");
        return 1;
    }
    TwiddleTable& t = it->second;
    if (t.lo_scaled && memcmp(&t.lo_scale, &scale, sizeof(Fe)) != 0) {
        H2_CHECK(hipDeviceSynchronize());
        (void)hipFree(t.lo_scaled);
        t.lo_scaled = nullptr;
    }
    if (!t.lo_scaled) {
        const uint32_t n_lo = 1u << t.lo_bits;
        H2_CHECK(hipMalloc((void**)&t.lo_scaled, (size_t)n_lo * sizeof(Fu)));
        hipLaunchKernelGGL(twiddle_scale_kernel, dim3((n_lo + 255) / 256), dim3(256), 0, s, (const Fu*)t.lo, t.lo_scaled, n_lo, fu_i_from_fe(scale));
        H2_CHECK(hipGetLastError());
        H2_CHECK(hipStreamSynchronize(s));  // built once per domain; later calls may use another stream
        t.lo_scale = scale;
    }
    *out = t.lo_scaled;
    return 0;
}

// omega^e for e < 2^log_n as lo[e & (2^lo_bits - 1)] * hi[e >> lo_bits] (I-form canonical limbs): the domain's cached table, for kernels
// outside this file (evaluate_h's permutation argument walks extended_omega^row).  Valid until the cache is cleared, which only a
// transform over a domain not yet in the cache can do.
int ntt_power_table(Ctx* c, const Fe& omega, uint32_t log_n, hipStream_t s, const Fu** lo, const Fu** hi, uint32_t* lo_bits) {
    TwiddleTable t;
    int rc = get_twiddles(c, omega, log_n, s, &t);
    if (rc) return rc;
    *lo = t.lo;
    *hi = t.hi;
    *lo_bits = t.lo_bits;
    return 0;
}

// the tile twiddles w_R^i (i < R / 2) for R = 2^s1 and 2^s2, each built once per domain and radix
static int get_stage_twiddles(Ctx* c, const Fe& omega, uint32_t log_n, uint32_t s1, uint32_t s2, hipStream_t s, TwiddleTable* out) {
    TwiddleKey key;
    for (int i = 0; i < 8; i++) key.omega[i] = omega.l[i];
    key.log_n = log_n;
    auto it = c->twiddles.find(key);
    if (it == c->twiddles.end()) {
        set_error("ntt: stage twiddles asked before the domain's table");
        return 1;
    }
    TwiddleTable& t = it->second;
    const uint32_t want[2] = {s1, s2};
    for (int k = 0; k < 2; k++) {
        if (want[k] < 1 || want[k] > 12) {
            set_error("ntt: tile radix 2^%u out of range", want[k]);
            return 1;
        }
        if (t.stage_of[want[k]]) continue;
        const uint32_t cnt = 1u << (want[k] - 1);
        Fu* d = nullptr;
        H2_CHECK(hipMalloc((void**)&d, (size_t)cnt * sizeof(Fu)));
        hipLaunchKernelGGL(stage_twiddle_build_kernel, dim3((cnt + 255) / 256), dim3(256), 0, s, omega, log_n - want[k], cnt, d);
        H2_CHECK(hipGetLastError());
        H2_CHECK(hipStreamSynchronize(s));  // built once per domain; later calls may use another stream
        t.stage_of[want[k]] = d;
    }
    *out = t;
    return 0;
}

// A pass's inter-pass twiddles as one table, built once per domain while the budget lasts: it replaces the multiplication that
// combines the two-level table -- one of a pass's six or seven per element -- by a 36-byte read.  two_pass: the two-pass plan's
// pass 1 (2^log_n entries [lo][k]: 38 MB at 2^20, 75 MB at 2^21); otherwise a strided pass of the other plan (2^log_m entries
// [k][lo]).  `p` carries the two-level table, log_m and s.  *table stays null when the budget (or the domain's H2_TW_FULL slots)
// is spent: the kernels fall back to tw_pow.
static int get_full_twiddles(Ctx* c, const Fe& omega, const NttPass& p, bool two_pass, hipStream_t s, const Fu** table, const Fe* scale = nullptr) {
    *table = nullptr;
    TwiddleKey key;
    for (int i = 0; i < 8; i++) key.omega[i] = omega.l[i];
    key.log_n = p.log_n;
    auto it = c->twiddles.find(key);
    if (it == c->twiddles.end()) return 0;
    TwiddleTable& t = it->second;
    const size_t bytes = sizeof(Fu) << p.log_m;
    const uint32_t tag = (two_pass ? 0x80000000u : 0u) | (scale ? 0x40000000u : 0u) | (p.log_m << 8) | p.s;
    int free_slot = -1;
    for (int k = 0; k < H2_TW_FULL; k++) {
        if (t.full[k] && t.full_tag[k] == tag) {
            // scaled: one constant per domain and pass (the inverse's 1/n); a caller with another one takes the unscaled table and multiplies
            if (!scale || memcmp(&t.full_scale[k], scale, sizeof(Fe)) == 0) *table = t.full[k];
            return 0;
        }
        if (!t.full[k] && free_slot < 0) free_slot = k;
    }
    if (free_slot < 0 || c->tw_full_bytes + bytes > g_ntt_full_budget) return 0;
    NttPass q = p;
    if (scale) {  // built from the scaled copy of `lo`: tw_pow(q, e) = scale * omega^e for every e
        const Fu* lo_s = nullptr;
        int rc = get_scaled_lo(c, omega, p.log_n, *scale, s, &lo_s);
        if (rc) return rc;
        q.tw_lo = lo_s;
    }
    Fu* d = nullptr;
    if (hipMalloc((void**)&d, bytes) != hipSuccess) {
        (void)hipGetLastError();  // no room: not an error, the two-level table still serves
        return 0;
    }
    const dim3 grid((uint32_t)((((uint64_t)1 << p.log_m) + 255) / 256));
    if (two_pass)
        hipLaunchKernelGGL(full_twiddle_build_kernel, grid, dim3(256), 0, s, q, d);
    else
        hipLaunchKernelGGL(pass_twiddle_build_kernel, grid, dim3(256), 0, s, q, d);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        (void)hipFree(d);
        set_error("ntt: building an inter-pass twiddle table failed");
        return 2;
    }
    t.full[free_slot] = d;
    t.full_tag[free_slot] = tag;
    if (scale) t.full_scale[free_slot] = *scale;
    c->tw_full_bytes += bytes;
    *table = d;
    return 0;
}

// EvaluationDomain::divide_by_vanishing_poly (poly/domain.rs:307-326): a[i] *= t_evaluations[i % t_len]
__global__ void __launch_bounds__(256) scale_periodic_kernel(Fe* a, uint64_t n, const Fu* t_i, uint32_t t_len) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        a[i] = fu_mul_canon<FrU>(fu_slice(a[i]), t_i[i % t_len]);
}

int scale_periodic_device(Ctx* c, Fe* d_a, uint64_t n, const uint64_t* h_t, uint32_t t_len, hipStream_t s) {
    if (t_len == 0 || t_len > 4096) {
        set_error("divide_by_vanishing_poly: t_len %u out of range", t_len);
        return 1;
    }
    int rc = c->ws_acquire(s);
    if (rc) return rc;
    WsGuard guard(c, s);
    rc = c->misc.ensure((size_t)t_len * sizeof(Fu));
    if (rc) return rc;
    std::vector<Fu> t(t_len);
    for (uint32_t i = 0; i < t_len; i++) {
        Fe e;  // caller memory is only 8-byte aligned
        memcpy(e.l, h_t + 4 * (size_t)i, sizeof(e.l));
        t[i] = fu_i_from_fe(e);
    }
    H2_CHECK(hipMemcpyAsync(c->misc.p, t.data(), (size_t)t_len * sizeof(Fu), hipMemcpyHostToDevice, s));
    H2_CHECK(hipStreamSynchronize(s));  // t lives on this stack frame
    uint64_t blocks = (n + 255) / 256;
    uint32_t grid = (uint32_t)(blocks < (uint64_t)c->sm_count * 16 ? blocks : (uint64_t)c->sm_count * 16);
    hipLaunchKernelGGL(scale_periodic_kernel, dim3(grid), dim3(256), 0, s, d_a, n, (const Fu*)c->misc.p, t_len);
    H2_CHECK(hipGetLastError());
    return guard.release();
}

static uint32_t g_ntt_smax = 8;
void ntt_set_smax(uint32_t v) { g_ntt_smax = v < 4 ? 4 : (v > 10 ? 10 : v); }
// Sizes 2^lo..2^hi take the two-pass plan (tiles of 2^10 or 2^11 points, ntt2_*_kernel): measured on MI355X
// (tools/ntt_two_pass.py) 14 % faster than three passes at 2^20, 9 % at 2^21, 5 % at 2^22 with two columns per workgroup, 25 / 19 / 12 % with one
// (round 3); below 2^19 a lone transform brings too few workgroups.
// Batched columns of 2^17 / 2^18 points take it too once the batch brings g_ntt_two_batch_wgs workgroup pairs per pass (tools/ntt_batch_rate.py, with one
// column per workgroup: a lone 2^18 transform 0.068 against 0.052 ms for three passes, two of them 0.041 against 0.048 each; four of 2^17 0.021 / 0.026;
// a lone 2^19 already 0.078 / 0.091).
static uint32_t g_ntt_two_lo = 19, g_ntt_two_hi = 22, g_ntt_two_batch_lo = 17;
static uint64_t g_ntt_two_batch_wgs = 512;
static bool g_ntt2_attr[64];
void ntt_set_two_pass(uint32_t lo, uint32_t hi) {
    if (lo == 0 && hi == 0) {  // the defaults
        g_ntt_two_lo = 19;
        g_ntt_two_hi = 22;
        g_ntt_two_batch_lo = 17;
        return;
    }
    g_ntt_two_lo = lo < 16 ? 16 : lo;
    g_ntt_two_hi = hi > 22 ? 22 : hi;
    g_ntt_two_batch_lo = g_ntt_two_lo < 17 ? g_ntt_two_lo : hi < lo ? 64 : 17;  // hi < lo turns the plan off altogether
}
void ntt_set_two_pass_batch_wgs(uint64_t v) { g_ntt_two_batch_wgs = v ? v : 512; }

// pass radices: one pass up to 2^10, otherwise ceil(log_n / smax) passes of near-equal radix
static int plan_passes(uint32_t log_n, uint32_t s_out[4]) {
    if (log_n <= 10) {
        s_out[0] = log_n;
        return 1;
    }
    // measured (tools/ntt_sweep.py): 256-point tiles are best up to 2^24; beyond, 512-point tiles save a whole pass
    const uint32_t smax = (g_ntt_smax == 8 && log_n > 24) ? 9 : g_ntt_smax;
    int P = (int)((log_n + smax - 1) / smax);
    if (P > 4) P = 4;
    uint32_t base = log_n / P, rem = log_n % P;
    for (int t = 0; t < P; t++) s_out[t] = base + ((uint32_t)t < rem ? 1 : 0);
    return P;
}

// The passes of `count` same-size transforms, each launched once with gridDim.y = count.  count == 1: plain pointers
// (d_data, first_src); count > 1: h_datas / h_firsts are host arrays of device pointers.
static int ntt_run(Ctx* c, size_t count, Fe* const* h_datas, const Fe* const* h_firsts, const Fe& omega, uint32_t log_n, const NttScale* sc,
                   hipStream_t s) {
    if (log_n > FrP::S) {
        set_error("ntt: log_n=%u exceeds the 2-adicity of Fr (28)", log_n);
        return 1;
    }
    if (count == 0) return 0;
    if (count > 65535) {
        set_error("ntt: batch of %zu exceeds the grid limit", count);
        return 1;
    }
    NttPass p;
    memset(&p, 0, sizeof(p));
    p.log_n = log_n;
    p.in_len = 1ull << log_n;
    if (sc) {
        p.in_scale = sc->in_scale;
        p.out_scale = !sc->out_scale ? 0 : (memcmp(&sc->out3[0], &sc->out3[1], sizeof(Fe)) == 0 && memcmp(&sc->out3[0], &sc->out3[2], sizeof(Fe)) == 0) ? 2 : 1;
        if (sc->in_len) p.in_len = sc->in_len;
        for (int i = 0; i < 3; i++) {
            p.in3[i] = fu_i_from_fe(sc->in3[i]);
            p.out3[i] = fu_i_from_fe(sc->out3[i]);
        }
    }
    int rc0 = c->ws_acquire(s);
    if (rc0) return rc0;
    WsGuard guard(c, s);
    int tid = c->timer_begin("ntt", s);
    const bool two = log_n <= g_ntt_two_hi &&
                     (log_n >= g_ntt_two_lo || (log_n >= g_ntt_two_batch_lo && ((uint64_t)count << (log_n - (log_n + 1) / 2 - 1)) >= g_ntt_two_batch_wgs));
    uint32_t S[4];
    int P = log_n ? plan_passes(log_n, S) : 1;
    if (two) {
        P = 2;
        S[0] = (log_n + 1) / 2;
        S[1] = log_n - S[0];
    }
    Fe* ws = nullptr;
    int rc;
    if (P > 1) {
        rc = c->ntt_ws.ensure((sizeof(Fe) << log_n) * count);
        if (rc) return rc;
        ws = (Fe*)c->ntt_ws.p;
    }
    // device pointer lists for a batch: [data | first source | workspace]
    Fe* const* l_data = nullptr;
    const Fe* const* l_first = nullptr;
    Fe* const* l_ws = nullptr;
    if (count > 1) {
        rc = c->ntt_ptrs.ensure(3 * count * sizeof(void*));
        if (rc) return rc;
        std::vector<const void*> h(3 * count);
        for (size_t i = 0; i < count; i++) {
            h[i] = h_datas[i];
            h[count + i] = h_firsts[i];
            h[2 * count + i] = ws ? ws + (i << log_n) : nullptr;
        }
        if ((rc = c->stage_h2d(c->ntt_ptrs.p, h.data(), h.size() * sizeof(void*), s))) return rc;  // through the pinned ring: `h` dies with this frame
        l_data = (Fe* const*)c->ntt_ptrs.p;
        l_first = (const Fe* const*)c->ntt_ptrs.p + count;
        l_ws = (Fe* const*)c->ntt_ptrs.p + 2 * count;
    }
    Fe* const d_data = h_datas[0];
    const Fe* const first_src = h_firsts[0];
    enum { FIRST, DATA, WS };
    auto bind = [&](int from, int to) {
        p.src = from == FIRST ? first_src : from == DATA ? d_data : ws;
        p.dst = to == DATA ? d_data : ws;
        if (count > 1) {
            p.srcs = from == FIRST ? l_first : from == DATA ? (const Fe* const*)l_data : (const Fe* const*)l_ws;
            p.dsts = to == DATA ? l_data : l_ws;
        }
    };
    if (log_n == 0) {
        bind(FIRST, DATA);
        p.first = 1;
        hipLaunchKernelGGL(ntt_n1_kernel, dim3(1, (uint32_t)count), dim3(64), 0, s, p);
        H2_CHECK(hipGetLastError());
        c->timer_end(tid, s);
        return guard.release();
    }
    TwiddleTable tw;
    rc = get_twiddles(c, omega, log_n, s, &tw);
    if (rc) return rc;
    if ((rc = get_stage_twiddles(c, omega, log_n, S[0], S[P - 1], s, &tw))) return rc;  // the plans hand out at most two radices
    p.tw_lo = tw.lo;
    p.tw_hi = tw.hi;
    p.lo_bits = tw.lo_bits;
    bool folded = false;
    if (two) {
        if (c->device >= 0 && c->device < 64 && !g_ntt2_attr[c->device]) {  // a 2^11-point image: more than the default 64 KB of dynamic LDS
            H2_CHECK(hipFuncSetAttribute((const void*)ntt2_strided_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 73728));
            H2_CHECK(hipFuncSetAttribute((const void*)ntt2_final_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 73728));
            g_ntt2_attr[c->device] = true;
        }
        p.s = S[0];
        // up to 2^21 points the table (36 B per point), the data and the workspace share the 256 MB Infinity Cache: -6 %; at 2^22 they
        // no longer do and the table costs 7 % instead
        p.log_m = log_n;
        const Fu* full0 = nullptr;
        if (log_n <= 21 && g_ntt_full_budget) {
            // one output constant (the inverse's 1/n): a copy of the table that carries it, and pass 2 closes with the direct reduction
            if (p.out_scale == 2 && sc && g_ntt_fold_tables) {
                if ((rc = get_full_twiddles(c, omega, p, true, s, &full0, &sc->out3[0]))) return rc;
                folded = full0 != nullptr;
            }
            if (!full0 && (rc = get_full_twiddles(c, omega, p, true, s, &full0))) return rc;
        }
        p.log_j = 0;  // one column / block per workgroup, neighbours grouped per XCD (ntt2_strided_kernel); two columns one after the other: see there
        if (g_ntt2_log_j >= 0 && g_ntt2_log_j <= 3) p.log_j = (uint32_t)g_ntt2_log_j;
        for (int t = 0; t < 2; t++) {
            p.s = S[t];
            p.first = (t == 0);
            {
                const uint32_t log_grid = log_n - p.s - p.log_j;
                const uint32_t want = t == 0 ? 2u : (log_n <= 21 ? 1u : 0u);  // pass 1: four columns = one 128-byte line; pass 2: pairs while the transform stays in the Infinity Cache
                p.xgroup = (p.log_j == 0 && log_grid >= 3 + want) ? want : 0;
            }
            p.stage_tw = tw.stage_of[S[t]];
            p.tw_full = t == 0 ? full0 : nullptr;  // (a zero budget also sets existing tables aside)
            if (t == 0 && !p.tw_full && p.out_scale == 2 && sc) {
                // one output constant (the inverse's 1/n) and pass 1 combining its twiddles from the two-level table: the constant rides in a
                // scaled copy of `lo`, and pass 2 closes with the direct reduction instead of the multiply
                const Fu* lo_s = nullptr;
                if ((rc = get_scaled_lo(c, omega, log_n, sc->out3[0], s, &lo_s))) return rc;
                p.tw_lo = lo_s;
                folded = true;
            }
            if (t == 1 && folded) {
                p.tw_lo = tw.lo;
                p.out_scale = 0;
            }
            bind(t == 0 ? FIRST : WS, t == 0 ? WS : DATA);
            const dim3 grid(1u << (log_n - p.s - p.log_j), (uint32_t)count), block(1u << (p.s - 2));
            const size_t lds = sizeof(Fu) << p.s;
            if (t == 0)
                hipLaunchKernelGGL(ntt2_strided_kernel, grid, block, lds, s, p);
            else
                hipLaunchKernelGGL(ntt2_final_kernel, grid, block, lds, s, p);
            H2_CHECK(hipGetLastError());
        }
        c->timer_end(tid, s);
        return guard.release();
    }
    uint32_t log_m = log_n;
    for (int t = 0; t < P; t++) {
        p.s = S[t];
        p.stage_tw = tw.stage_of[p.s];
        p.log_m = log_m;
        p.first = (t == 0);
        bool final = (t == P - 1);
        if (final) {
            p.tw_full = nullptr;
            p.tw_lo = tw.lo;
            if (folded) p.out_scale = 0;
            bind(P == 1 ? FIRST : WS, DATA);
            uint32_t log_nb = log_n - p.s;
            const uint32_t want_j = p.s <= 8 ? 2 : 1;  // 4 columns (128 B rows) up to 256-point tiles, 2 beyond: LDS
            p.log_j = log_nb < want_j ? log_nb : want_j;
            p.n_prev = (uint32_t)(P - 1);
            for (int q = 0; q < P - 1; q++) p.prev_s[q] = S[q];
        } else {
            // the last strided pass goes out of place so the final pass lands in the data buffer; passes own
            // disjoint tiles, so reading elsewhere than they write is safe
            bind(t == 0 ? FIRST : DATA, t == P - 2 ? WS : DATA);
            // a table of up to 2^20 entries (38 MB) stays cache-resident next to the data: every strided pass but the first of a large transform
            p.tw_full = nullptr;
            p.tw_lo = tw.lo;  // (before a table is built from it: pass_twiddle_build_kernel reads p)
            if (log_m <= g_ntt_full_max_log_m && g_ntt_full_budget) {
                const Fu* full = nullptr;
                if (t == 0 && p.out_scale == 2 && sc && g_ntt_fold_tables) {  // as in the two-pass plan: the first pass's table carries the one output constant
                    if ((rc = get_full_twiddles(c, omega, p, false, s, &full, &sc->out3[0]))) return rc;
                    folded = full != nullptr;
                }
                if (!full && (rc = get_full_twiddles(c, omega, p, false, s, &full))) return rc;
                p.tw_full = full;
            }
            if (t == 0 && !p.tw_full && p.out_scale == 2 && sc) {  // as in the two-pass plan: the one output constant rides in the first pass's twiddles
                const Fu* lo_s = nullptr;
                if ((rc = get_scaled_lo(c, omega, log_n, sc->out3[0], s, &lo_s))) return rc;
                p.tw_lo = lo_s;
                folded = true;
            }
            uint32_t log_l = log_m - p.s;
            const uint32_t want_j = p.s <= 8 ? 2 : 1;
            p.log_j = log_l < want_j ? log_l : want_j;
        }
        size_t lds = sizeof(Fu) << (p.s + p.log_j);
        uint64_t grid = 1ull << (log_n - p.s - p.log_j);
        if (grid > 0x7fffffffull) {
            set_error("ntt: grid too large");
            return 1;
        }
        if (final)
            hipLaunchKernelGGL(ntt_final_kernel, dim3((uint32_t)grid, (uint32_t)count), dim3(NTT_THREADS), lds, s, p);
        else
            hipLaunchKernelGGL(ntt_strided_kernel, dim3((uint32_t)grid, (uint32_t)count), dim3(NTT_THREADS), lds, s, p);
        H2_CHECK(hipGetLastError());
        log_m -= p.s;
    }
    c->timer_end(tid, s);
    return guard.release();
}

int ntt_device(Ctx* c, Fe* d_data, const Fe& omega, uint32_t log_n, const NttScale* sc, hipStream_t s, const Fe* d_src) {
    const Fe* first = d_src ? d_src : d_data;
    return ntt_run(c, 1, &d_data, &first, omega, log_n, sc, s);
}

// `count` transforms of the same size and scale, one launch per pass.  h_srcs (optional, host array): transform i reads its
// input at h_srcs[i] (nullptr entries and a null array mean "in place").
int ntt_device_batch(Ctx* c, Fe* const* h_datas, const Fe* const* h_srcs, size_t count, const Fe& omega, uint32_t log_n, const NttScale* sc,
                     hipStream_t s) {
    if (log_n > FrP::S) {
        set_error("ntt: log_n=%u exceeds the 2-adicity of Fr (28)", log_n);
        return 1;
    }
    // One launch per pass for as many columns as 2 GB of columns + workspace hold.  (Until round 3 the cut was 96 MB -- one 2^20 column
    // per launch -- from a measurement on the three-pass kernels of round 1; with the two-pass kernels a lone 2^20 column is 512
    // workgroups, two per CU where four fit: eighteen 2^18 -> 2^20 coset NTTs 0.139 -> 0.105 ms each in one launch per pass, ten
    // 2^17 -> 2^19 0.073 -> 0.062; from 2^22 up it makes no difference: tools/ntt_batch_rate.py.)
    size_t per = g_ntt_batch_bytes / (2 * (sizeof(Fe) << log_n));
    if (per < 1) per = 1;
    std::vector<const Fe*> firsts(count);
    for (size_t i = 0; i < count; i++) firsts[i] = (h_srcs && h_srcs[i]) ? h_srcs[i] : h_datas[i];
    for (size_t i = 0; i < count; i += per) {
        size_t m = count - i < per ? count - i : per;
        int rc = ntt_run(c, m, h_datas + i, firsts.data() + i, omega, log_n, sc, s);
        if (rc) return rc;
    }
    return 0;
}

}  // namespace h2
