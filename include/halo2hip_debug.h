/* halo2hip_debug.h -- test and tuning hooks of libhalo2hip.so.
 *
 * Not part of the drop-in surface (include/halo2hip.h): nothing a halo2 prover calls.  tests/ and tools/ reach these
 * through ctypes to force code paths (chunking, window widths, kernel variants) and to sweep tuning constants; every
 * hook restores its default when passed 0.  They change process-wide state and are not meant for concurrent use.
 */
#ifndef HALO2HIP_DEBUG_H
#define HALO2HIP_DEBUG_H
#include "halo2hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* a host-resident MSM of at least min_n pairs (default 2^19) streams across PCIe in `chunks` pieces (default 3; 1 = off) whose
 * sizes grow by 1000 / ratio_permille (default 600 = upload time over compute time per pair); with the bases crossing too
 * (unpinned) twice the chunks at twice the ratio; zeros restore the defaults */
int h2hip_debug_set_msm_stream(uint32_t chunks, uint32_t ratio_permille, size_t min_n);
/* needs no GPU: the chunk sizes a streamed MSM of n pairs is cut into (0 = the values in force); returns their number */
size_t h2hip_debug_msm_stream_ladder(size_t n, uint32_t chunks, uint32_t ratio_permille, int with_bases, size_t* sizes, size_t cap);
/* split MSM inputs above m pairs into consecutive chunks (default 2^26, the 31-bit pair-index limit; 0 restores it) */
int h2hip_debug_set_msm_max_chunk(size_t m);
/* push `count` Jacobian partials per engine device through the library's RCCL all-gather (communicators created on
 * demand, also for one device) and fold them on the host: exercises the multi-GPU gather on any box */
int h2hip_debug_rccl_gather_selftest(const uint64_t* partials_xyz, size_t count, uint64_t* out_xyz);
/* buckets with more than (entries of the MSM) / d entries take the chunked path (default d = 32768; 0 restores it) */
int h2hip_debug_set_msm_heavy_div(size_t d);
/* g_to_lagrange / fft_g1 up to k = 14: bit 0 -- one quad of lanes per butterfly (1, default) or one lane (0); bit 1 set -- normalise to affine
 * after every layer as round 3 did (default clear: the points stay XYZZ between the layers, one normalisation at the end) */
int h2hip_debug_set_g2l_quad(int on);
/* fused batches: at most `entries` entries per fused run (0 = 2^26), MSMs of at most `max_n` pairs are fused (0 = 2^19) */
int h2hip_debug_set_msm_fuse_limits(size_t entries, size_t max_n);
/* first row/column pass of the reduction: lane budget (0 = 65536, one wave per SIMD) and explicit-mad multiplier (1) or plain (0) */
int h2hip_debug_set_msm_rowcol(uint64_t lanes, int use_asm);
/* accumulation of runs with fewer than 2^18 buckets: up to 8 lanes per bucket (1, default) or one (0) */
int h2hip_debug_set_msm_split_buckets(int on);
/* reduction tail: one quad of lanes per group operation (1, default) or one lane each (0) */
/* runs with at most 4 bucket sets: finish the reduction on the host from bit-plane sums (1, default) or keep the GPU tail (0) */
int h2hip_debug_set_msm_plane_tail(int on);
int h2hip_debug_set_msm_quad_tail(int on);
/* 1: accumulate order = buckets by size inside each sort bin only; 0 (default): global size order */
int h2hip_debug_set_msm_bucket_order(int local);
/* level-1 sort records of runs whose bins span several level-2 tiles: entries and key bits as two arrays (1, default) or 8-byte pairs (0) */
int h2hip_debug_set_msm_split_records(int on);
/* target entries per coarse bin of the MSM's two-level sort (default 8192; 0 restores it) */
int h2hip_debug_set_msm_bin_entries(size_t d);
/* CUs reserved for the sort / reduce streams of a batched MSM (0 = none: every split measured slower) */
int h2hip_debug_set_reserved_cus(uint32_t k);
/* batches of MSMs of up to 2^19 pairs: fused into one run (1, default) or pipelined over streams (0) */
int h2hip_debug_set_msm_fuse_small(int on);
/* largest log2 tile of an NTT pass (4..10; default 8, 9 beyond 2^24 points) */
int h2hip_debug_set_ntt_smax(uint32_t v);
int h2hip_debug_set_lazy_pin(uint32_t after);
/* sizes 2^lo..2^hi take the two-pass plan (default 19..22; hi < lo: never; 0, 0: back to the defaults) */
int h2hip_debug_set_ntt_two_pass(uint32_t lo, uint32_t hi);
int h2hip_debug_set_ntt_twiddle_budget(uint64_t bytes);
/* batched transforms: bytes of columns + workspace one launch spans (0 = default) */
int h2hip_debug_set_ntt_batch_bytes(uint64_t bytes);
/* two-pass plan for batched columns of 2^17 / 2^18 points: from this many pairs of workgroups per pass (0 = default 512) */
int h2hip_debug_set_ntt_two_pass_batch_wgs(uint64_t v);
/* three-pass plan: strided passes up to 2^v points read their inter-pass twiddles from a per-domain table (0 = default 24) */
int h2hip_debug_set_ntt_full_max_log_m(uint32_t v);
/* 0: no scaled copies of the inter-pass tables (the inverse's 1/n is then a multiplication in the last pass); 1 = default */
int h2hip_debug_set_ntt_fold_tables(int on);
/* host-pointer batched transforms: device bytes one pipelined run may hold (0 = default 4 GB; smaller forces several runs) and the
 * size below which columns are grouped per pipeline step (0 = default 2 MB) */
int h2hip_debug_set_ntt_host_batch(uint64_t run_bytes, uint64_t group_bytes);
/* two-pass plan: log2 columns per workgroup (-1 = default) */
int h2hip_debug_set_ntt_two_pass_log_j(int v);
/* evaluate_h: programs needing more slots than v use the global-workspace form of the kernels (default 256) */
int h2hip_debug_set_evalh_max_local_slots(uint32_t v);
/* evaluate_h: HBM one group of lookup cosets may take (0 = default 2 GB; the first group is transformed with the advice columns) */
int h2hip_debug_set_evalh_lookup_group_bytes(uint64_t v);
/* evaluate_h: field multiplications per row of the program a graph compiles to; needs no GPU */
int h2hip_debug_evalh_program_muls(const h2hip_graph* g, uint32_t* n_mul);
/* evaluate_h: the per-circuit gates kernel (hiprtc): 0 off, 1 compiled on a background thread while the interpreter serves (default;
 * HALO2_HIP_EVALH_CODEGEN), 2 compiled inline; + 16: without the fusion of two products into one reduction; max_ops: programs with more operations (or more than 48
 * values alive at once) stay with the interpreter (0 = default 1200) */
int h2hip_debug_set_evalh_codegen(int mode, uint32_t max_ops);
/* the HIP source a graph's program is emitted as (buf may be NULL: *len alone), and with bit 0 of `compile` set what hiprtc makes of it for
 * gfx950 (seconds, bytes of code object); bit 1: the graph as a lookup argument's table expression (evalh_lookup_gen) instead of the custom
 * gates (evalh_gates_gen); needs no GPU.  Returns 0, 1 (malformed graph) or 2 (hiprtc missing / rejected the source). */
int h2hip_debug_evalh_codegen_source(const h2hip_graph* g, char* buf, size_t cap, size_t* len, int compile, double* seconds, size_t* code_bytes);
/* out: programs compiled, compilations failed, launches of a generated kernel, launches of the interpreter, disk-cache hits */
int h2hip_debug_evalh_codegen_stats(uint64_t out[5]);
/* evaluate_h: compile a graph as the engine would and report the program's size; needs no GPU */
int h2hip_debug_evalh_compile_stats(const h2hip_graph* g, uint32_t* n_ops, uint32_t* n_slots);

#ifdef __cplusplus
}
#endif
#endif /* HALO2HIP_DEBUG_H */
