"""Randomised cross-checks on the GPU box (not part of the test suite; run by hand):
  * MSM: random sizes (not powers of two) and scalar shapes, plain form, fixed-base form and the CPU oracle agree in affine;
  * NTT: every size 2^1..2^22, both plans and both twiddle sources agree limb for limb, round trips restore the input;
  * round 4: random window widths through both reduction tails (host-finished and all-GPU), host-pointer batched transforms with random
    column counts / run cuts / step grouping against the device-resident transform, lone and batched calls on one domain interleaved."""
import ctypes, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_pkg
h2 = load_pkg(); h2.init(0)
import torch
from oracle import oracle
NT = min(16, os.cpu_count() or 1)
L = h2.lib()
rng = random.Random(int(os.environ.get("SEED", "7")))

bad = 0
for it in range(int(os.environ.get("MSM_CASES", "24"))):
    n = rng.choice([1, 2, 3, 63, 64, 65, 1000, 4097, rng.randrange(1, 70000), rng.randrange(1, 300000)])
    bs = oracle.gen_points(1000 + it, n, num_threads=NT)
    kind = rng.randrange(4)
    sc = oracle.gen_scalars(2000 + it, n, num_threads=NT)
    if kind == 1:
        sc[rng.randrange(n):] = 0                                  # a zero tail
    elif kind == 2:
        m = np.array([rng.random() < 0.9 for _ in range(n)])
        sc[m] = 0                                                  # prover-like: mostly zero
        sc[~m, 1:] = 0; sc[~m, 0] &= np.uint64(3)                  # small values in Montgomery limbs are still big integers: fine, just skewed
    elif kind == 3:
        sc[:] = sc[0]                                              # one scalar everywhere: every point lands in the same bucket per window
    want = oracle.g1_to_affine(oracle.best_multiexp(sc, bs, NT))
    # round 3: the host-slice call streams in a random number of chunks (ladder ratio and threshold random too), or not at all
    chunks = rng.choice([1, 2, 3, 4, 6])
    L.h2hip_debug_set_msm_stream(ctypes.c_uint32(chunks), ctypes.c_uint32(rng.choice([300, 600, 1000, 1700])), ctypes.c_size_t(rng.choice([1024, 4096, 1 << 16])))
    got_plain = h2.g1_to_affine(h2.best_multiexp(sc, bs))
    h2.bases_pin(bs)
    try:
        got_fixed = h2.g1_to_affine(h2.best_multiexp(sc, bs))
    finally:
        h2.bases_unpin(bs)
    ok = np.array_equal(got_plain, want) and np.array_equal(got_fixed, want)
    bad += not ok
    if n >= 2 and it % 3 == 0:  # and a batch of columns over the same bases (fused groups stream in)
        cols = [sc, np.ascontiguousarray(sc[::-1]), oracle.gen_scalars(3000 + it, n, num_threads=NT)][:rng.choice([2, 3])]
        h2.bases_pin(bs)
        try:
            gb = h2.best_multiexp_batch(cols, bs)
        finally:
            h2.bases_unpin(bs)
        for j, col in enumerate(cols):
            ok = ok and np.array_equal(h2.g1_to_affine(gb[j]), oracle.g1_to_affine(oracle.best_multiexp(col, bs, NT)))
        bad += not ok and bad == 0
    print("msm n=%d kind=%d chunks=%d %s" % (n, kind, chunks, "ok" if ok else "MISMATCH"), flush=True)

L.h2hip_debug_set_msm_stream(ctypes.c_uint32(0), ctypes.c_uint32(0), ctypes.c_size_t(0))
for k in range(1, 23):
    d = h2.EvaluationDomain.new(2, k)
    a = h2.gen_scalars_device(40 + k, 1 << k)
    outs = []
    for lo, hi, budget in ((1, 0, 1 << 30), (16, 22, 1 << 30), (1, 0, 0), (16, 22, 0)):
        L.h2hip_debug_set_ntt_two_pass(ctypes.c_uint32(lo), ctypes.c_uint32(hi))
        L.h2hip_debug_set_ntt_twiddle_budget(ctypes.c_uint64(budget))
        x = a.clone(); h2.ntt_device(x, d.omega, k)
        y = x.clone(); h2.ifft_device(y, d.omega_inv, k, d.ifft_divisor)
        torch.cuda.synchronize()
        outs.append((x, torch.equal(y, a)))
    L.h2hip_debug_set_ntt_two_pass(ctypes.c_uint32(0), ctypes.c_uint32(0))
    L.h2hip_debug_set_ntt_twiddle_budget(ctypes.c_uint64(4 << 30))
    ok = all(torch.equal(o[0], outs[0][0]) and o[1] for o in outs)
    if k <= 16:
        ok = ok and np.array_equal(h2.to_numpy_u64(outs[0][0]), oracle.best_fft(h2.to_numpy_u64(a).copy(), d.omega, k, NT))
    bad += not ok
    print("ntt 2^%d %s" % (k, "ok" if ok else "MISMATCH"), flush=True)
# round 4, late: the first pass's table up to 2^24 points and the inverse's 1/n folded into a scaled copy of it -- against the same transform
# with the table limited to 2^20 entries (round 3), without the fold, and on the two-level table alone; forward and scaled inverse
for k in [int(x) for x in os.environ.get("BIG_NTT", "19,21,23,24").split(",") if x]:
    d = h2.EvaluationDomain.new(2, k)
    a = h2.gen_scalars_device(70 + k, 1 << k)
    res = []
    for full_max, fold, budget in ((0, 1, 4 << 30), (20, 1, 4 << 30), (0, 0, 4 << 30), (0, 1, 0)):
        L.h2hip_debug_set_ntt_full_max_log_m(ctypes.c_uint32(full_max))
        L.h2hip_debug_set_ntt_fold_tables(ctypes.c_int(fold))
        L.h2hip_debug_set_ntt_twiddle_budget(ctypes.c_uint64(budget))
        x = a.clone(); h2.ntt_device(x, d.omega, k)
        y = a.clone(); h2.ifft_device(y, d.omega_inv, k, d.ifft_divisor)
        z = x.clone(); h2.ifft_device(z, d.omega_inv, k, d.ifft_divisor)
        torch.cuda.synchronize()
        res.append((x, y, torch.equal(z, a)))
    L.h2hip_debug_set_ntt_full_max_log_m(ctypes.c_uint32(0))
    L.h2hip_debug_set_ntt_fold_tables(ctypes.c_int(1))
    L.h2hip_debug_set_ntt_twiddle_budget(ctypes.c_uint64(4 << 30))
    ok = all(torch.equal(r[0], res[0][0]) and torch.equal(r[1], res[0][1]) and r[2] for r in res)
    bad += not ok
    print("big ntt 2^%d tables / fold / budget %s" % (k, "ok" if ok else "MISMATCH"), flush=True)
    del res, a
    torch.cuda.empty_cache()
# round 3: batched zero-padded coset transforms (coeff_to_extended) -- padding ratios 2, 4, 8, random column counts, the two-pass plan
# forced from 2^16 points and one workgroup up (first stage pair skipped at ratio >= 4, rows taken by the lanes of their own points)
# against one column at a time on the three-pass plan
for it in range(int(os.environ.get("COSET_CASES", "24"))):
    k = rng.randrange(8, 19)
    rlog = rng.choice([1, 2, 2, 3])
    ek = min(k + rlog, 21)
    rlog = ek - k
    cnt = rng.randrange(1, 13)
    d = h2.EvaluationDomain.new(2 ** rlog + 1, k)
    assert d.extended_k == ek, (d.extended_k, ek)
    cols = [h2.gen_scalars_device(5000 + 16 * it + j, 1 << ek) for j in range(cnt)]
    L.h2hip_debug_set_ntt_two_pass(ctypes.c_uint32(1), ctypes.c_uint32(0))
    want = [c.clone() for c in cols]
    for w in want:
        h2.coeff_to_extended_device(w, k, ek, d.extended_omega, d.g_coset, d.g_coset_inv)
    L.h2hip_debug_set_ntt_two_pass(ctypes.c_uint32(16), ctypes.c_uint32(22))
    L.h2hip_debug_set_ntt_two_pass_batch_wgs(ctypes.c_uint64(1))
    got = [c.clone() for c in cols]
    h2.coeff_to_extended_batch_device(got, k, ek, d.extended_omega, d.g_coset, d.g_coset_inv)
    L.h2hip_debug_set_ntt_two_pass_batch_wgs(ctypes.c_uint64(0))
    L.h2hip_debug_set_ntt_two_pass(ctypes.c_uint32(0), ctypes.c_uint32(0))
    got2 = [c.clone() for c in cols]
    h2.coeff_to_extended_batch_device(got2, k, ek, d.extended_omega, d.g_coset, d.g_coset_inv)  # the library's own choice of plan
    torch.cuda.synchronize()
    ok = all(torch.equal(g_, w) and torch.equal(g2, w) for g_, g2, w in zip(got, got2, want))
    bad += not ok
    print("coset 2^%d -> 2^%d x %d %s" % (k, ek, cnt, "ok" if ok else "MISMATCH"), flush=True)
# round 4 (a): window widths x reduction tails.  The host-finished tail (bit planes + Horner on the host) and the all-GPU tail must give the
# oracle's group element for every width the table builder accepts, on uniform, mostly-zero and all-equal columns.
for it in range(int(os.environ.get("TAIL_CASES", "16"))):
    n = rng.choice([257, 4096, 20000, rng.randrange(2, 100000)])
    c = rng.choice([5, 8, 11, 13, 14, 16, 17, 19, 20, 22])
    bs = oracle.gen_points(7000 + it, n, num_threads=NT)
    sc = oracle.gen_scalars(7100 + it, n, num_threads=NT)
    kind = rng.randrange(3)
    if kind == 1:
        sc[np.array([rng.random() < 0.95 for _ in range(n)])] = 0
    elif kind == 2:
        sc[:] = sc[0]
    want = oracle.g1_to_affine(oracle.best_multiexp(sc, bs, NT))
    h2.set_msm_window(c)
    h2.bases_pin(bs)
    try:
        got = []
        for on in (1, 0):
            L.h2hip_debug_set_msm_plane_tail(ctypes.c_int(on))
            got.append(h2.g1_to_affine(h2.best_multiexp(sc, bs)))
    finally:
        L.h2hip_debug_set_msm_plane_tail(ctypes.c_int(1))
        h2.bases_unpin(bs)
        h2.set_msm_window(0)
    ok = np.array_equal(got[0], want) and np.array_equal(got[1], want)
    bad += not ok
    print("tail n=%d c=%d kind=%d %s" % (n, c, kind, "ok" if ok else "MISMATCH"), flush=True)
# round 4 (b): host-pointer batched transforms (three-stage pipeline, copier threads) against the device-resident transform of the same
# columns; random run cuts and step grouping; a lone call on the same domain in between (the plans' twiddle tables coexist)
for it in range(int(os.environ.get("HOSTBATCH_CASES", "16"))):
    k = rng.randrange(1, 19)
    cnt = rng.randrange(1, 12)
    d = h2.EvaluationDomain.new(4, k)
    ek = d.extended_k
    cols_d = [h2.gen_scalars_device(9000 + 16 * it + j, 1 << k) for j in range(cnt)]
    cols_h = [h2.to_numpy_u64(c_).copy() for c_ in cols_d]
    col_bytes = 32 << ek
    L.h2hip_debug_set_ntt_host_batch(ctypes.c_uint64(rng.choice([0, 0, 2 * col_bytes, 3 * col_bytes + 7, 1])), ctypes.c_uint64(rng.choice([0, 1, 4 * col_bytes, 64 << 20])))
    want_i = []
    for c_ in cols_d:
        x = c_.clone()
        h2.ifft_device(x, d.omega_inv, k, d.ifft_divisor)
        want_i.append(x)
    lone = d.lagrange_to_coeff(cols_h[0])
    got_i = d.lagrange_to_coeff_batch(cols_h)
    got_e = d.coeff_to_extended_batch(got_i)
    want_e = []
    for x in want_i:
        e = torch.zeros((1 << ek, 4), dtype=torch.int64, device="cuda")
        e[:1 << k] = x.view(-1, 4)
        h2.coeff_to_extended_device(e, k, ek, d.extended_omega, d.g_coset, d.g_coset_inv)
        want_e.append(e)
    torch.cuda.synchronize()
    ok = np.array_equal(lone, h2.to_numpy_u64(want_i[0]))
    ok = ok and all(np.array_equal(g_, h2.to_numpy_u64(w)) for g_, w in zip(got_i, want_i))
    ok = ok and all(np.array_equal(g_, h2.to_numpy_u64(w)) for g_, w in zip(got_e, want_e))
    L.h2hip_debug_set_ntt_host_batch(ctypes.c_uint64(0), ctypes.c_uint64(0))
    bad += not ok
    print("host batch 2^%d x %d %s" % (k, cnt, "ok" if ok else "MISMATCH"), flush=True)
print("FAILURES", bad)
sys.exit(1 if bad else 0)
