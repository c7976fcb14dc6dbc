/*
 * halo2hip.h -- C ABI of libhalo2hip.so, the MI355X (gfx950) engine for the halo2_proofs
 * proving hot path: BN254 G1 multi-scalar multiplication and the radix-2 NTT over BN254 Fr.
 *
 * The reference (eldenpark/halo2-pse, Rust) has no FFI layer: the seam is two generic free
 * functions plus the KZG commit methods built on them.  Each entry point below names the
 * reference interface it replaces (paths relative to halo2_proofs/src/ in the reference);
 * the Rust side ships as files: halo2hip-sys/ (extern block generated from this header, safe wrappers) and
 * patches/0001..0003 against the reference.
 *
 * Data layout at the boundary (identical to halo2curves 0.3.1 in memory and to
 * SerdeFormat::RawBytes, helpers.rs:13-19):
 *   Fr / Fq element : 4 x uint64_t little-endian limbs, Montgomery form (R = 2^256), < modulus
 *   G1Affine        : x || y            (8 x uint64_t, 64 B), identity = all zero
 *   G1 (projective) : x || y || z       (12 x uint64_t, 96 B), Jacobian, identity z = 0
 *
 * Conventions: every function returns 0 on success and a non-zero H2HIP_E* code otherwise
 * (never throws / unwinds; the Rust shim falls back to the original CPU body on non-zero).
 * h2hip_last_error() describes the last failure on the calling thread.  All entry points are
 * thread-safe and blocking.  One process drives the GPUs named at h2hip_init: host-pointer MSMs
 * are sharded over all of them inside the call (the fan-out and fold best_multiexp performs over
 * rayon threads, arithmetic.rs:137-153), everything else runs on the first device, or -- for the
 * _device entry points -- on the device that owns the pointers, under that device's lock only (calls
 * on different devices overlap; a batched transform whose columns live on several devices is split by owner).
 * There is no CPU fallback inside the library: without a usable GPU every compute entry
 * point fails with H2HIP_EDEVICE.
 *
 * Environment (read at init): HALO2_HIP_DEVICES="0,1,.." device list when h2hip_init gets none;
 * HALO2_HIP_MULTI_GPU_MIN_N (default 2^18) smallest MSM that is sharded; HALO2_HIP_GATHER=host|rccl
 * how the devices' partial sums meet (default host: each device's sum has come back with its run, the
 * calling thread folds them; rccl: ncclAllGather from the devices' HBM over xGMI, fixed-base form);
 * HALO2_HIP_MSM_WINDOW; HALO2_HIP_STREAM=0 uploads a host-slice MSM's arrays whole instead of streaming them in chunks
 * (no copier thread), HALO2_HIP_STREAM_MIN_N (default 2^19) is the smallest MSM that streams; HALO2_HIP_FIXED_BASE=0 and
 * HALO2_HIP_TABLE_MAX_GB for h2hip_bases_pin's window tables; HALO2_HIP_MSM_MIN_N /
 * HALO2_HIP_NTT_MIN_LOGN thresholds the Rust shim reads back through h2hip_msm_min_n() /
 * h2hip_ntt_min_log_n(); HALO2_HIP_ROCTX=1 roctx ranges around every entry point;
 * HALO2_HIP_NTT_TWIDDLE_MB (default 4096) HBM per device for the full inter-pass twiddle tables of 2^20-, 2^21-, 2^23- and 2^24-point
 * transforms (36 bytes per point, domain and direction; without room the two-level table serves, one multiplication more per point);
 * HALO2_HIP_LAZY_PIN=k (default 0 = off) lets the library pin a host bases array by itself once a
 * host-pointer MSM has seen it k times (see h2hip_bases_pin); HALO2_HIP_EVALH_CODEGEN=0|1|2 the per-circuit custom-gates kernel of
 * h2hip_evaluate_h_bn254 (0: byte-code interpreter only; 1, default: generated and compiled by hiprtc on a background thread, the interpreter
 * serves until the code object is ready; 2: compiled inline) and HALO2_HIP_CACHE_DIR a directory that keeps those code objects across processes
 * (files there are loaded as GPU code, keyed by a hash of their source: the directory must be writable by the prover's user only).
 */
#ifndef HALO2HIP_H
#define HALO2HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define H2HIP_OK 0
#define H2HIP_EINVAL 1   /* contract violation (the reference would panic: arithmetic.rs:133,184) */
#define H2HIP_EDEVICE 2  /* HIP runtime / device error, or no GPU */
#define H2HIP_ENOMEM 3

/* ---- lifecycle ------------------------------------------------------------------------- */

/* Bind the engine to n_devices GPUs (HIP device ordinals).  device_ids == NULL or n_devices == 0: the
 * HALO2_HIP_DEVICES list, else the calling thread's current HIP device.  Idempotent for the same list;
 * a different list needs h2hip_shutdown first (H2HIP_EINVAL otherwise).  With more than one device the
 * engine starts one host thread and one stream per device and, with HALO2_HIP_GATHER=rccl and librccl
 * present, one RCCL communicator per device (ncclCommInitAll) for the gather of the MSM partials.  An id out of range or
 * listed twice is H2HIP_EINVAL. */
int h2hip_init(const int* device_ids, int n_devices);
void h2hip_shutdown(void);
const char* h2hip_last_error(void);
const char* h2hip_version(void);
/* number of visible HIP devices, or -1 when the runtime reports an error */
int h2hip_device_count(void);
/* devices the engine is bound to (0 before init) */
int h2hip_num_devices(void);
/* dispatch thresholds for the shim: below them the reference's CPU body is the faster path.  Without a usable GPU they answer
 * "never" (SIZE_MAX / UINT32_MAX) from a cached failure -- no lock, no HIP call -- so the CPU-fallback path of a prover on a GPU-less host costs nothing;
 * h2hip_init or h2hip_shutdown makes the library look for a device again. */
size_t h2hip_msm_min_n(void);
uint32_t h2hip_ntt_min_log_n(void);

/* ---- best_multiexp: arithmetic.rs:132-159 ------------------------------------------------ */

/* out = sum_i scalars[i] * bases[i].  Replaces
 *   pub fn best_multiexp<C: CurveAffine>(coeffs: &[C::Scalar], bases: &[C]) -> C::Curve
 * for C = bn256::G1Affine, and through it ParamsKZG::commit / commit_lagrange
 * (poly/kzg/commitment.rs:281-292, :327-334) and MSMKZG::eval (poly/kzg/msm.rs:65-70).
 * Host pointers; n == 0 gives the identity.  The group element equals the reference's for
 * every thread count; Jacobian coordinates are not specified by the reference
 * (they depend on rayon's thread count, arithmetic.rs:153) and may differ from call to call here too.
 * The slices are the caller's (pageable) memory, borrowed for the call.  From 2^19 pairs up the scalars -- and, when
 * the bases are not pinned, the points -- cross PCIe in chunks that stream in under the work and add into one bucket
 * set, so only the first chunk's upload is exposed (2^20 pairs over pinned bases: 1.6 ms against 1.25 ms for
 * device-resident inputs).  With several devices every device's share streams over its own link.  Blocking and
 * thread-safe: concurrent callers (rayon workers) are serialised per device. */
int h2hip_msm_bn254(const uint64_t* scalars, const uint64_t* bases_xy, size_t n, uint64_t out_xyz[12]);

/* Keep `bases_xy[0..n)` on the GPU(s), keyed by the host pointer: later h2hip_msm_bn254[_batch] calls
 * whose bases pointer equals `bases_xy` (and n' <= n) skip the upload.  For ParamsKZG::{g, g_lagrange}
 * (poly/kzg/commitment.rs:26-27), which live as long as the params; call unpin before the Vec is dropped
 * or mutated (downsize, :267-275).  With several devices each one keeps its contiguous share.
 * Pinning also builds the fixed-base window table 2^(pos_j) * P_i (pos_j = first bit of window j, W = ceil(255/c) windows; n <= 2^26; W * n * 64
 * bytes of HBM, e.g. 0.8 GB at 2^20, 12 GB at 2^24): all windows of an MSM then share one bucket set, the
 * windows are wider (c = 20 instead of 16 at 2^20) and no Horner pass is needed.
 * The cache is safe against stale pointers: every lookup compares 16 sampled points of the caller's array (the first
 * point and a geometric ladder of indices up to the last, so that a shorter prefix still sees several) with the ones seen
 * at pin time and falls back to a plain upload (dropping the entry) on a mismatch. */
int h2hip_bases_pin(const uint64_t* bases_xy, size_t n);
/* The lazy cache (HALO2_HIP_LAZY_PIN=k): for the unpatched two-line drop-in, whose best_multiexp(coeffs, bases) has no
 * handle in its signature.  A host array that reaches h2hip_msm_bn254[_batch] unpinned for the k-th time with the same
 * address, length and sampled points is pinned as above by the library (at most 4 such arrays, least recently used
 * dropped first; h2hip_bases_unpin removes one, h2hip_shutdown all).  Lookups validate it like any pinned entry.  What the
 * samples cannot see is a caller rewriting part of a live array in place; ParamsKZG never does, hence opt-in. */
uint32_t h2hip_lazy_pin_after(void);
/* the same for points that already live in HBM (keyed by the device pointer, used by the _device MSMs).  The points are
 * copied into the table, but the caller's buffer stays the KEY: keep it alive and unchanged until h2hip_bases_unpin, as a
 * ParamsKZG keeps g.  Should it be freed and its address reused all the same, the cache notices: a one-wave kernel ahead
 * of every MSM compares 16 sampled points of the buffer with the ones seen at pin time (no extra synchronisation; the
 * verdict is read with the MSM's result), and on a mismatch the entry is dropped and the MSM runs over the caller's points. */
int h2hip_bases_pin_device(const void* d_bases_xy, size_t n, void* stream);
/* drops a pinned host array or device buffer; H2HIP_EINVAL when the pointer is not pinned (also when the engine is not
 * initialised: the call never starts it) */
int h2hip_bases_unpin(const void* bases_xy);
/* what a pinned pointer holds: points, window width / windows of its table (0 / 0 without one), bytes of HBM */
int h2hip_bases_pinned_info(const void* bases_xy, size_t* n_points, uint32_t* window_bits, uint32_t* windows, size_t* device_bytes);

/* Same computation on device-resident inputs (scalars n x 32 B, bases n x 64 B in HBM);
 * `stream` is a hipStream_t; NULL is HIP's default (null) stream, as everywhere in HIP.  All
 * kernels are enqueued on that stream, so the inputs may be produced by earlier work on it;
 * the call returns after the stream is synchronised and the result is in host memory. */
int h2hip_msm_bn254_device(const void* d_scalars, const void* d_bases_xy, size_t n, uint64_t out_xyz[12], void* stream);

/* `count` independent MSMs of n pairs each over the SAME bases: out_xyz[12*j..] = sum_i scalars[j][i] * bases[i].
 * This is what create_proof does when it commits its columns back to back (plonk/prover.rs:361-365:
 * `params.commit_lagrange(poly, blind)` for every advice polynomial; lookup/prover.rs:127-132, :291;
 * vanishing/prover.rs:104).  Results equal `count` separate h2hip_msm_bn254 calls.  Up to 2^19 pairs the MSMs of a batch
 * run fused (one sort / accumulate / reduce over the windows of all of them); larger ones are pipelined whole over three
 * streams (sort of j+1 and reduction of j-1 under the accumulation of j). */
int h2hip_msm_bn254_batch(const uint64_t* const* scalars, const uint64_t* bases_xy, size_t n, size_t count, uint64_t* out_xyz);
int h2hip_msm_bn254_batch_device(const void* const* d_scalars, const void* d_bases_xy, size_t n, size_t count, uint64_t* out_xyz, void* stream);

/* CurveExt::batch_normalize (plonk/prover.rs:368-371 normalises the advice commitments before writing them to the
 * transcript): k Jacobian points -> k affine points with one field inversion; identity -> (0,0).  Host-side. */
int h2hip_g1_batch_normalize(const uint64_t* xyz, size_t k, uint64_t* xy);

/* Left fold of k Jacobian partial sums from the identity -- the fold at arithmetic.rs:153.
 * Inside one process h2hip_msm_bn254 does this itself over the devices of h2hip_init.  With one process per
 * GPU (bench.py --gpus N) each rank computes its shard's partial with h2hip_msm_bn254[_device], the
 * 96-byte partials are all-gathered (RCCL, bytes), and every rank folds them with this. */
int h2hip_g1_fold(const uint64_t* partials_xyz, size_t k, uint64_t out_xyz[12]);
/* Curve::to_affine; identity -> (0,0) */
int h2hip_g1_to_affine(const uint64_t xyz[12], uint64_t xy[8]);

/* ---- best_fft and the EvaluationDomain conversions built on it --------------------------- */

/* In-place radix-2 NTT, natural order in and out: a[i] <- sum_j a[j] * omega^(i*j).  Replaces
 *   pub fn best_fft<G: Group>(a: &mut [G], omega: G::Scalar, log_n: u32)       arithmetic.rs:171
 * for G = bn256::Fr.  `a` has 2^log_n elements; omega must have exact order 2^log_n (true for
 * every in-crate caller: poly/domain.rs:248,354); log_n <= 28. */
int h2hip_ntt_bn254_fr(uint64_t* a, const uint64_t omega[4], uint32_t log_n);
int h2hip_ntt_bn254_fr_device(void* d_a, const uint64_t omega[4], uint32_t log_n, void* stream);

/* EvaluationDomain::ifft (poly/domain.rs:353-361): best_fft(a, omega_inv, log_n) then
 * a[i] *= divisor, fused into one device round trip.  lagrange_to_coeff (:226-236) is this
 * with (omega_inv, k, ifft_divisor). */
int h2hip_ifft_bn254_fr(uint64_t* a, const uint64_t omega_inv[4], uint32_t log_n, const uint64_t divisor[4]);
int h2hip_ifft_bn254_fr_device(void* d_a, const uint64_t omega_inv[4], uint32_t log_n, const uint64_t divisor[4], void* stream);

/* EvaluationDomain::coeff_to_extended (poly/domain.rs:240-254): out[0..2^extended_k) =
 * best_fft(zero-pad(distribute_powers_zeta(a[0..2^k), into_coset = true)), extended_omega).
 * g_coset = ZETA, g_coset_inv = ZETA^2 as stored in the domain (:81-82). */
int h2hip_coeff_to_extended_bn254_fr(const uint64_t* a, uint32_t k, uint64_t* out, uint32_t extended_k,
                                     const uint64_t extended_omega[4], const uint64_t g_coset[4], const uint64_t g_coset_inv[4]);
/* device form: d_a holds 2^extended_k elements of which the first 2^k are the input */
int h2hip_coeff_to_extended_bn254_fr_device(void* d_a, uint32_t k, uint32_t extended_k, const uint64_t extended_omega[4],
                                            const uint64_t g_coset[4], const uint64_t g_coset_inv[4], void* stream);

/* EvaluationDomain::extended_to_coeff (poly/domain.rs:281-303) without the final truncate:
 * ifft(a, extended_omega_inv, extended_k, extended_ifft_divisor) then
 * distribute_powers_zeta(a, into_coset = false), in place on 2^extended_k elements. */
int h2hip_extended_to_coeff_bn254_fr(uint64_t* a, uint32_t extended_k, const uint64_t extended_omega_inv[4],
                                     const uint64_t extended_ifft_divisor[4], const uint64_t g_coset[4], const uint64_t g_coset_inv[4]);
int h2hip_extended_to_coeff_bn254_fr_device(void* d_a, uint32_t extended_k, const uint64_t extended_omega_inv[4],
                                            const uint64_t extended_ifft_divisor[4], const uint64_t g_coset[4],
                                            const uint64_t g_coset_inv[4], void* stream);

/* EvaluationDomain::divide_by_vanishing_poly (poly/domain.rs:307-326): a[i] *= t_evaluations[i % t_len] on the
 * 2^extended_k coset evaluations, t_evaluations = the domain's inverted t(X) = X^n - 1 values (:84-124). */
int h2hip_divide_by_vanishing_poly_bn254_fr(uint64_t* a, uint32_t extended_k, const uint64_t* t_evaluations, uint32_t t_len);
int h2hip_divide_by_vanishing_poly_bn254_fr_device(void* d_a, uint32_t extended_k, const uint64_t* t_evaluations, uint32_t t_len, void* stream);

/* ---- device-resident polynomials ---------------------------------------------------------
 * A caller without a HIP binding (the Rust shim) keeps columns on the GPU across calls -- commit a column with
 * h2hip_msm_bn254_device, then h2hip_ifft_bn254_fr_device / h2hip_coeff_to_extended_bn254_fr_device on the same
 * buffer -- instead of crossing PCIe for every call.  Plain hipMalloc'd memory; `stream` as elsewhere. */
int h2hip_device_alloc(size_t bytes, void** d_ptr);
int h2hip_device_free(void* d_ptr);
int h2hip_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes, void* stream);
int h2hip_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes, void* stream);
int h2hip_memset_zero(void* d_dst, size_t bytes, void* stream);
int h2hip_stream_synchronize(void* stream);

/* ---- batched device-resident transforms: `count` columns of one size in one launch per NTT pass.  d_a is a HOST array of
 * `count` device pointers; semantics per column are those of the unbatched _device entry points.  create_proof converts
 * its columns back to back (plonk/prover.rs:487 lagrange_to_coeff per advice column; plonk/evaluation.rs:306-323
 * coeff_to_extended per advice / instance column): at 2^17..2^20 points per column one transform does not fill the GPU. */
int h2hip_ntt_bn254_fr_batch_device(void* const* d_a, size_t count, const uint64_t omega[4], uint32_t log_n, void* stream);
int h2hip_ifft_bn254_fr_batch_device(void* const* d_a, size_t count, const uint64_t omega_inv[4], uint32_t log_n,
                                     const uint64_t divisor[4], void* stream);
int h2hip_coeff_to_extended_bn254_fr_batch_device(void* const* d_a, size_t count, uint32_t k, uint32_t extended_k,
                                                  const uint64_t extended_omega[4], const uint64_t g_coset[4],
                                                  const uint64_t g_coset_inv[4], void* stream);

/* ---- batched transforms on HOST columns: what the prover has when its polynomials are `Vec<F>`s (plonk/prover.rs:476-490 converts
 * every advice column with lagrange_to_coeff; plonk/evaluation.rs:306-323 extends every advice / instance column with coeff_to_extended).
 * `a` / `out` are host arrays of `count` host pointers; semantics per column are those of the unbatched host entry points.  The columns
 * run as a three-stage pipeline -- column i + 1 crosses PCIe upwards and column i - 1 downwards while column i is transformed -- so a
 * column costs max(upload, transform, download) instead of their sum (2^22: 5.4 ms one call each).  With several devices the columns
 * are dealt to them in contiguous shares, each share over its own PCIe link (NTT replicas; no column crosses xGMI).  Blocking. */
int h2hip_ntt_bn254_fr_batch(uint64_t* const* a, size_t count, const uint64_t omega[4], uint32_t log_n);
int h2hip_ifft_bn254_fr_batch(uint64_t* const* a, size_t count, const uint64_t omega_inv[4], uint32_t log_n, const uint64_t divisor[4]);
/* a[i]: 2^k coefficients in, out[i]: 2^extended_k evaluations out (a[i] and out[i] may not overlap unless equal) */
int h2hip_coeff_to_extended_bn254_fr_batch(const uint64_t* const* a, uint32_t k, uint64_t* const* out, size_t count, uint32_t extended_k,
                                           const uint64_t extended_omega[4], const uint64_t g_coset[4], const uint64_t g_coset_inv[4]);
int h2hip_extended_to_coeff_bn254_fr_batch(uint64_t* const* a, size_t count, uint32_t extended_k, const uint64_t extended_omega_inv[4],
                                           const uint64_t extended_ifft_divisor[4], const uint64_t g_coset[4], const uint64_t g_coset_inv[4]);

/* ---- g_to_lagrange: arithmetic.rs:277-301 (best_fft with G = G1, then 1/n and batch_normalize); called by
 * ParamsKZG::downsize, poly/kzg/commitment.rs:267-275.  g_xy: 2^k affine points (the coefficient-basis SRS, possibly
 * truncated); g_lagrange_xy: 2^k affine points out.  k <= 28.  Input and output may not overlap. */
int h2hip_g_to_lagrange_bn254(const uint64_t* g_xy, uint32_t k, uint64_t* g_lagrange_xy);
int h2hip_g_to_lagrange_bn254_device(const void* d_g_xy, uint32_t k, void* d_g_lagrange_xy, void* stream);

/* best_fft with G = bn256::G1 and a caller-supplied omega (arithmetic.rs:171-234; in the crate only g_to_lagrange calls it, :285): in place on
 * 2^log_n Jacobian points (96 B each, identity z = 0), a[i] <- sum_j [omega^(i*j)] a[j].  omega of exact order 2^log_n; log_n <= 28.  Only
 * the group elements are defined by the reference (its Jacobian coordinates depend on the butterfly order and rayon's thread count): the
 * points come back with z = 1 (identity: (0, 1, 0)), compare after normalising. */
int h2hip_fft_bn254_g1(uint64_t* a_xyz, const uint64_t omega[4], uint32_t log_n);
int h2hip_fft_bn254_g1_device(void* d_a_xyz, const uint64_t omega[4], uint32_t log_n, void* stream);

/* ---- ParamsKZG::setup: poly/kzg/commitment.rs:61-129, with the secret supplied by the caller (the reference draws it from
 * an rng at :72): g[i] = [s^i] G1 and g_lagrange[i] = [l_i(s)] G1 for i < 2^k, affine, as batch_normalize leaves them.
 * k <= 28; a secret that is a 2^k-th root of unity is H2HIP_EINVAL (the reference panics on the inversion at :100).
 * The G2 half of the parameters (g2, s_g2: :118-119) belongs to the verifier and is not produced here. */
int h2hip_kzg_setup_bn254(uint32_t k, const uint64_t secret[4], uint64_t* g_xy, uint64_t* g_lagrange_xy);
int h2hip_kzg_setup_bn254_device(uint32_t k, const uint64_t secret[4], void* d_g_xy, void* d_g_lagrange_xy, void* stream);

/* ---- Evaluator::evaluate_h: plonk/evaluation.rs:280-522 ---------------------------------------
 * The quotient numerator h(X) on the extended coset: per row, the custom-gate graph (GraphEvaluator, :191-201,
 * :708-749), the permutation argument's constraints (:362-441) and every lookup's constraints (:443-518), folded
 * with powers of y.  The reference's Rust structures are passed flattened:
 *   ValueSource  (evaluation.rs:37-60)   -> h2hip_value_source {kind, a, b}
 *   Calculation + CalculationInfo (:108-127, :213-219) -> h2hip_calculation
 *   GraphEvaluator (:191-201)            -> h2hip_graph
 * One call handles one circuit instance; `values` is read and written (the reference folds all instances into
 * the same polynomial, :326-332), so pass zeros for the first instance. */
enum {
    H2HIP_VS_CONSTANT = 0, H2HIP_VS_INTERMEDIATE = 1, H2HIP_VS_FIXED = 2, H2HIP_VS_ADVICE = 3, H2HIP_VS_INSTANCE = 4,
    H2HIP_VS_CHALLENGE = 5, H2HIP_VS_BETA = 6, H2HIP_VS_GAMMA = 7, H2HIP_VS_THETA = 8, H2HIP_VS_Y = 9, H2HIP_VS_PREVIOUS = 10
};
enum {
    H2HIP_CALC_ADD = 0, H2HIP_CALC_SUB = 1, H2HIP_CALC_MUL = 2, H2HIP_CALC_SQUARE = 3, H2HIP_CALC_DOUBLE = 4,
    H2HIP_CALC_NEGATE = 5, H2HIP_CALC_HORNER = 6, H2HIP_CALC_STORE = 7
};
enum { H2HIP_ANY_ADVICE = 0, H2HIP_ANY_FIXED = 1, H2HIP_ANY_INSTANCE = 2 };

typedef struct {
    uint32_t kind; /* H2HIP_VS_* */
    uint32_t a;    /* constant / intermediate / challenge index, or column index */
    uint32_t b;    /* index into the graph's rotations for Fixed / Advice / Instance */
} h2hip_value_source;

typedef struct {
    uint32_t op;     /* H2HIP_CALC_* */
    uint32_t target; /* intermediate written (CalculationInfo::target) */
    h2hip_value_source x, y; /* operands; Horner: x = start value, y = factor */
    uint32_t parts_offset, parts_count; /* Horner parts in h2hip_graph::parts */
} h2hip_calculation;

typedef struct {
    const uint64_t* constants; /* n_constants x 4 */
    uint32_t n_constants;
    const int32_t* rotations;
    uint32_t n_rotations;
    const h2hip_calculation* calculations;
    uint32_t n_calculations;
    const h2hip_value_source* parts;
    uint32_t n_parts;
    uint32_t num_intermediates; /* <= 256 in this engine */
} h2hip_graph;

typedef struct {
    /* domain (poly/domain.rs:18-34) */
    uint32_t k, extended_k;
    const uint64_t *extended_omega, *g_coset, *g_coset_inv;
    /* columns: pk.fixed_cosets (extended), advice / instance polynomials in coefficient form (2^k each; their cosets
     * are formed here with coeff_to_extended as at evaluation.rs:306-323) */
    uint32_t n_fixed, n_advice, n_instance, n_challenges;
    const uint64_t* const* fixed_cosets;
    const uint64_t* const* advice_polys;
    const uint64_t* const* instance_polys;
    const uint64_t* challenges;              /* n_challenges x 4 */
    const uint64_t *y, *beta, *gamma, *theta;
    const uint64_t *l0, *l_last, *l_active_row; /* pk.l0 / l_last / l_active_row, extended */
    h2hip_graph custom_gates;                /* Evaluator::custom_gates */
    /* permutation argument (evaluation.rs:362-441); n_perm_sets == 0 skips it */
    uint32_t n_perm_sets, n_perm_columns, chunk_len; /* chunk_len = cs.degree() - 2 */
    int32_t last_rotation;                   /* -(blinding_factors + 1) */
    const uint64_t* const* perm_product_cosets; /* sets[i].permutation_product_coset, extended */
    const uint32_t* perm_column_kind;        /* H2HIP_ANY_* of p.columns[j] */
    const uint32_t* perm_column_index;
    const uint64_t* const* perm_cosets;      /* pk.permutation.cosets[j], extended */
    const uint64_t *zeta, *delta;            /* Fr::ZETA, Fr::DELTA */
    /* lookups (evaluation.rs:443-518) */
    uint32_t n_lookups;
    const h2hip_graph* lookup_graphs;        /* Evaluator::lookups[n] */
    const uint64_t* const* lookup_product_polys;        /* coefficient form, 2^k each */
    const uint64_t* const* lookup_permuted_input_polys;
    const uint64_t* const* lookup_permuted_table_polys;
} h2hip_evalh_desc;

/* The constant columns of a proving key -- pk.fixed_cosets, pk.l0 / l_last / l_active_row, pk.permutation.cosets (plonk.rs:262-272): extended-coset
 * form, `elems` = 2^extended_k elements each -- kept in HBM across h2hip_evaluate_h_bn254 calls, keyed by their host pointers: the host-pointer
 * evaluate_h then uploads only what changes from proof to proof (at k = 18 in the bench system 22 of its 27 full-size columns are constant:
 * 2.8 of 3.5 GB per call).  Idempotent; guarded like h2hip_bases_pin (16 sampled elements of the caller's memory are compared on every use, a
 * mismatch drops the copy and uploads); HALO2_HIP_COLUMN_CACHE_GB (default 64) bounds the HBM taken, least recently used columns go first.
 * Unpin before the Vecs are dropped (ProvingKey's Drop); unknown pointers are ignored; neither call changes any result. */
int h2hip_columns_pin(const uint64_t* const* cols, size_t count, size_t elems);
int h2hip_columns_unpin(const uint64_t* const* cols, size_t count);
int h2hip_columns_pinned_info(size_t* n_columns, size_t* device_bytes);

/* values: 2^extended_k elements, in/out, host memory; every column pointer in desc is host memory */
int h2hip_evaluate_h_bn254(const h2hip_evalh_desc* desc, uint64_t* values);
/* Device-resident form for a prover whose columns already live in HBM: every column pointer in desc (fixed_cosets[i],
 * advice_polys[i], instance_polys[i], l0 / l_last / l_active_row, perm_product_cosets[i], perm_cosets[j], the lookup
 * polynomials) and d_values are device pointers; extended cosets are read where they lie, coefficient-form polynomials
 * are copied device-to-device before they are extended, so no input is modified.  The pointer tables themselves,
 * the graphs, challenges and scalars stay host memory.  Kernels are queued on `stream` (NULL = the default stream);
 * the call does not wait for them. */
int h2hip_evaluate_h_bn254_device(const h2hip_evalh_desc* desc, void* d_values, void* stream);

/* ---- synthetic workload; same streams as oracle_gen_{scalars,points} ---- */

int h2hip_gen_scalars_device(uint64_t seed, uint64_t start, size_t n, void* d_out, void* stream);
int h2hip_gen_points_device(uint64_t seed, uint64_t start, size_t n, void* d_out, void* stream);

/* ---- tuning and measurement ----------------------------------------------------------------- */

/* MSM window width in bits (2..24); 0 restores the size-based default.  Applies to the plain form at once
 * and to window tables built by later h2hip_bases_pin* calls. */
int h2hip_set_msm_window(uint32_t c);
/* window width the engine would use for n pairs: plain form / a window table over n pinned points */
uint32_t h2hip_get_msm_window(size_t n);
uint32_t h2hip_get_msm_window_fixed_base(size_t n);
/* Per-stage HIP-event timers recorded on the stream each kernel group is launched on.
 * Stages: "ntt", "msm_total", "msm_digits", "msm_sort", "msm_accum" (over-full buckets included), "msm_reduce", "g_to_lagrange", "kzg_setup". */
/* on = 1: every stage (each event record costs the stream ~10 us of gap); on = 2: only the dominant kernel ("msm_accum"),
 * timed through its own dispatch packet with no gap; 0: off */
int h2hip_profile_enable(int on);
int h2hip_profile_reset(void);
int h2hip_profile_get(const char* stage, double* total_ms, uint64_t* count);

/* Test and tuning hooks (h2hip_debug_*) are declared in halo2hip_debug.h; they are not part of the drop-in surface. */

#ifdef __cplusplus
}
#endif
#endif /* HALO2HIP_H */
