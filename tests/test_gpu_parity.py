"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle and the
golden vectors.  Bit-exact: NTT outputs limb for limb, MSM outputs after normalising to affine
(SURVEY.md Appendix B).  Run with `pytest -m gpu` on an MI355X."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MSM_CASES = ["1", "2", "3", "4", "31", "32", "33", "100", "1024", "zeros", "ones", "rm1", "single", "sparse", "cancel"]
NT = min(16, os.cpu_count() or 1)


@pytest.fixture(scope="module", autouse=True)
def _engine(h2):
    h2.init()
    yield
    h2.set_msm_window(0)


def aff(h2, xyz):
    return h2.g1_to_affine(xyz)


# ---------------------------------------------------------------------------- MSM
@pytest.mark.parametrize("case", MSM_CASES)
def test_msm_golden(h2, golden, case):
    got = aff(h2, h2.best_multiexp(golden[f"msm_{case}_scalars"], golden[f"msm_{case}_bases"]))
    assert np.array_equal(got, golden[f"msm_{case}_result"])


@pytest.mark.parametrize("c", [2, 3, 5, 8, 11, 13, 16])
def test_msm_golden_all_window_widths(h2, golden, c):
    h2.set_msm_window(c)
    try:
        for case in ("33", "1024", "sparse", "rm1"):
            got = aff(h2, h2.best_multiexp(golden[f"msm_{case}_scalars"], golden[f"msm_{case}_bases"]))
            assert np.array_equal(got, golden[f"msm_{case}_result"]), (c, case)
    finally:
        h2.set_msm_window(0)


def test_msm_empty(h2):
    out = h2.best_multiexp(np.zeros((0, 4), dtype=np.uint64), np.zeros((0, 8), dtype=np.uint64))
    assert np.array_equal(aff(h2, out), np.zeros(8, dtype=np.uint64))


@pytest.mark.parametrize("n", [1 << 10, (1 << 12) + 37, 1 << 14, 1 << 16])
def test_msm_vs_oracle_seeded(h2, oracle, n):
    sc = oracle.gen_scalars(0x5EED0001, n, num_threads=NT)
    bs = oracle.gen_points(0x5EED0002, n, num_threads=NT)
    want = oracle.g1_to_affine(oracle.best_multiexp(sc, bs, NT))
    got = aff(h2, h2.best_multiexp(sc, bs))
    assert np.array_equal(got, want)


def _prover_like(oracle, n, seed):
    # 90 % zero, 5 % in {1,2}, 5 % uniform
    rng = np.random.default_rng(seed)
    sc = oracle.gen_scalars(seed, n, num_threads=NT)
    u = rng.random(n)
    one = oracle.fe_from_int(oracle.FR, 1)
    two = oracle.fe_from_int(oracle.FR, 2)
    sc[u < 0.90] = 0
    sc[(u >= 0.90) & (u < 0.925)] = one
    sc[(u >= 0.925) & (u < 0.95)] = two
    return sc


def test_msm_skewed_scalars_heavy_buckets(h2, oracle):
    n = 1 << 14
    bs = oracle.gen_points(77, n, num_threads=NT)
    # every scalar equal: one over-full bucket per window (chunked path, several chunks)
    sc = np.repeat(oracle.gen_scalars(78, 1), n, axis=0)
    want = oracle.g1_to_affine(oracle.best_multiexp(sc, bs, NT))
    assert np.array_equal(aff(h2, h2.best_multiexp(sc, bs)), want)
    # prover-like sparse column
    sc = _prover_like(oracle, n, 79)
    want = oracle.g1_to_affine(oracle.best_multiexp(sc, bs, NT))
    assert np.array_equal(aff(h2, h2.best_multiexp(sc, bs)), want)
    # all bases equal and all scalars equal: doubling inside buckets and inside the tree
    bs1 = np.repeat(bs[:1], n, axis=0)
    want = oracle.g1_to_affine(oracle.best_multiexp(sc, bs1, NT))
    assert np.array_equal(aff(h2, h2.best_multiexp(sc, bs1)), want)


def test_msm_pinned_bases_and_prefix(h2, oracle):
    n = 1 << 12
    sc = oracle.gen_scalars(5, n, num_threads=NT)
    bs = oracle.gen_points(6, n, num_threads=NT)
    h2.bases_pin(bs)
    try:
        for m in (n, n // 2, 100):
            want = oracle.g1_to_affine(oracle.best_multiexp(sc[:m], bs[:m], NT))
            got = h2.best_multiexp(sc[:m], bs[:m])  # a view that starts at the pinned pointer: no re-upload
            assert np.array_equal(aff(h2, got), want)
    finally:
        h2.bases_unpin(bs)
    with pytest.raises(h2.H2HipError):
        h2.bases_unpin(bs)


def test_generators_match_oracle(h2, oracle, golden):
    ds = h2.gen_scalars_device(0x5EED0001, 4096)
    dp = h2.gen_points_device(0x5EED0002, 4096)
    s, p = h2.to_numpy_u64(ds), h2.to_numpy_u64(dp)
    assert np.array_equal(s[:64], golden["gen_scalars_5EED0001"])
    assert np.array_equal(p[:64], golden["gen_points_5EED0002"])
    assert np.array_equal(s, oracle.gen_scalars(0x5EED0001, 4096, num_threads=NT))
    assert np.array_equal(p, oracle.gen_points(0x5EED0002, 4096, num_threads=NT))
    ds = h2.gen_scalars_device(0x5EED0001, 8, start=1000)
    assert np.array_equal(h2.to_numpy_u64(ds), golden["gen_scalars_offset1000"])
    dp = h2.gen_points_device(0x5EED0002, 8, start=1000)
    assert np.array_equal(h2.to_numpy_u64(dp), golden["gen_points_offset1000"])


def test_msm_device_entry_point(h2, oracle):
    n = 1 << 13
    ds = h2.gen_scalars_device(0x5EED0001, n)
    dp = h2.gen_points_device(0x5EED0002, n)
    got = aff(h2, h2.msm_device(ds, dp))
    want = oracle.g1_to_affine(oracle.best_multiexp(h2.to_numpy_u64(ds), h2.to_numpy_u64(dp), NT))
    assert np.array_equal(got, want)


def test_msm_full_size_2p20(h2, oracle):
    """BASELINE.json configs[1]: 2^20 pairs on one MI355X, bit-exact vs the CPU path; plus the
    size-independent shard/fold property used by the multi-GPU path."""
    n = 1 << 20
    ds = h2.gen_scalars_device(0x5EED0001, n)
    dp = h2.gen_points_device(0x5EED0002, n)
    full = h2.msm_device(ds, dp)
    sc, bs = h2.to_numpy_u64(ds), h2.to_numpy_u64(dp)
    want = oracle.g1_to_affine(oracle.best_multiexp(sc, bs, NT))
    assert np.array_equal(aff(h2, full), want)
    parts = [h2.msm_device(ds[i * (n // 4):(i + 1) * (n // 4)], dp[i * (n // 4):(i + 1) * (n // 4)]) for i in range(4)]
    assert np.array_equal(aff(h2, h2.g1_fold(np.stack(parts))), want)


# ---------------------------------------------------------------------------- NTT
@pytest.mark.parametrize("k", range(0, 11))
def test_ntt_golden(h2, golden, k):
    a = golden[f"ntt_{k}_in"].copy()
    h2.best_fft(a, golden[f"ntt_{k}_omega"], k)
    assert np.array_equal(a, golden[f"ntt_{k}_out"])


@pytest.mark.parametrize("k", [11, 12, 13, 15, 16, 17, 18, 19, 20, 21])
def test_ntt_vs_oracle(h2, oracle, k):
    d, _ = oracle.domain_new(2, k)
    a = oracle.gen_scalars(0x5EED0003, 1 << k, num_threads=NT)
    want = oracle.best_fft(a, d.fe("omega"), k, NT)
    got = a.copy()
    h2.best_fft(got, d.fe("omega"), k)
    assert np.array_equal(got, want)
    # and back: ifft returns the input (round trip) and equals the oracle's ifft
    back = got.copy()
    dom = _domain(h2, d)
    back = dom.lagrange_to_coeff(back)
    assert np.array_equal(back, a)


def _domain(h2, d):
    """the product-side domain computes its own constants (EvaluationDomain.new); the oracle's are only compared with them"""
    dom = h2.EvaluationDomain.new(d.quotient_poly_degree + 1, d.k)
    assert dom.extended_k == d.extended_k
    for f in h2.EvaluationDomain.FIELDS:
        assert np.array_equal(getattr(dom, f), d.fe(f)), f
    return dom


@pytest.mark.parametrize("jk", [(4, 5), (3, 4), (2, 3)])
def test_domain_golden(h2, oracle, golden, jk):
    j, k = jk
    d, _ = oracle.domain_new(j, k)
    dom = _domain(h2, d)
    assert np.array_equal(dom.coeff_to_extended(golden[f"ext_{j}_{k}_coeffs"]), golden[f"ext_{j}_{k}_extended"])
    assert np.array_equal(dom.extended_to_coeff(golden[f"ext_{j}_{k}_h_extended"]), golden[f"ext_{j}_{k}_h_coeffs"])


@pytest.mark.parametrize("jk", [(4, 10), (4, 14), (3, 12), (2, 11), (4, 17), (4, 18), (2, 20)])
def test_domain_vs_oracle(h2, oracle, jk):
    j, k = jk
    d, _ = oracle.domain_new(j, k)
    dom = _domain(h2, d)
    a = oracle.gen_scalars(1234 + k, 1 << k, num_threads=NT)
    ext = dom.coeff_to_extended(a)
    assert np.array_equal(ext, oracle.coeff_to_extended(d, a, NT))
    h = oracle.gen_scalars(4321 + k, 1 << d.extended_k, num_threads=NT)
    assert np.array_equal(dom.extended_to_coeff(h), oracle.extended_to_coeff(d, h, NT))
    lag = oracle.gen_scalars(999 + k, 1 << k, num_threads=NT)
    assert np.array_equal(dom.lagrange_to_coeff(lag), oracle.lagrange_to_coeff(d, lag, NT))


def test_ntt_full_size_2p22_roundtrip_and_oracle(h2, oracle):
    """BASELINE.json configs[2]: k = 22 NTT + iNTT on one MI355X (device-resident), each
    direction equal to the oracle, and the round trip returns the input."""
    import torch
    k = 22
    d, _ = oracle.domain_new(2, k)
    da = h2.gen_scalars_device(0x5EED0003, 1 << k)
    a = h2.to_numpy_u64(da).copy()
    h2.ntt_device(da, d.fe("omega"), k)
    torch.cuda.synchronize()
    fwd = h2.to_numpy_u64(da).copy()
    assert np.array_equal(fwd, oracle.best_fft(a, d.fe("omega"), k, NT))
    h2.ifft_device(da, d.fe("omega_inv"), k, d.fe("ifft_divisor"))
    torch.cuda.synchronize()
    assert np.array_equal(h2.to_numpy_u64(da), a)


@pytest.mark.parametrize("k", [18, 19, 20, 21, 22, 23])
def test_ntt_two_pass_plan_equals_three_pass(h2, oracle, k):
    """The two plans of ntt.hip (two passes of 2^9..2^11-point tiles, three of 2^6..2^8) give the same limbs for the plain
    transform, the scaled inverse and the zero-padded coset transform, with the inter-pass twiddles read from their per-domain table
    or combined from the two-level one; the tuning hooks force each at every size.  Where a first pass combines its twiddles from the
    two-level table, the inverse's 1/n rides in a scaled copy of that table and the last pass closes with the direct reduction
    (2^21 up on three passes -- with a table-fed second pass behind it --, 2^22 on two, every size under a zero budget); 2^23 has
    only the three-pass plan, with and without tables."""
    import ctypes
    import torch
    L = h2.lib()
    d, _ = oracle.domain_new(4, k - 2)  # extended_k = k
    assert d.extended_k == k
    a = h2.gen_scalars_device(77 + k, 1 << k)
    out = {}
    try:
        for plan, (lo, hi, budget) in (("three", (1, 0, 1 << 30)), ("two", (18, 22, 1 << 30)), ("three, two-level twiddles", (1, 0, 0)), ("two, two-level twiddles", (18, 22, 0))):
            L.h2hip_debug_set_ntt_two_pass(ctypes.c_uint32(lo), ctypes.c_uint32(hi))
            L.h2hip_debug_set_ntt_twiddle_budget(ctypes.c_uint64(budget))  # 0: inter-pass twiddles from the two-level table (one multiplication more)
            f = a.clone()
            h2.ntt_device(f, d.fe("extended_omega"), k)
            i = a.clone()
            h2.ifft_device(i, d.fe("extended_omega_inv"), k, d.fe("extended_ifft_divisor"))
            e = a.clone()
            h2.coeff_to_extended_device(e, k - 2, k, d.fe("extended_omega"), d.fe("g_coset"), d.fe("g_coset_inv"))
            b = a.clone()
            h2.extended_to_coeff_device(b, k, d.fe("extended_omega_inv"), d.fe("extended_ifft_divisor"), d.fe("g_coset"), d.fe("g_coset_inv"))
            torch.cuda.synchronize()
            out[plan] = (f, i, e, b)
    finally:
        L.h2hip_debug_set_ntt_two_pass(ctypes.c_uint32(0), ctypes.c_uint32(0))
        L.h2hip_debug_set_ntt_twiddle_budget(ctypes.c_uint64(4 << 30))
    for plan, res in out.items():
        for x, y in zip(res, out["three"]):
            assert torch.equal(x, y), plan


@pytest.mark.parametrize("k", [11, 14, 17, 19, 20, 21, 23, 24])
def test_inverse_scale_folded_into_the_first_pass_table(h2, oracle, k):
    """round 4: where the first pass reads its inter-pass twiddles from a per-domain table, the inverse's 1/n rides in a scaled copy of
    that table (get_full_twiddles(.., scale)) and the last pass closes with the direct reduction.  Same limbs as the multiplication in
    the last pass (h2hip_debug_set_ntt_fold_tables(0)), lone and as a batch; a second constant on the same domain (not a prover's case)
    finds the table taken, multiplies instead and is right as well; the round trip restores the input."""
    import ctypes
    import torch
    L = h2.lib()
    d, _ = oracle.domain_new(1, k)
    a = h2.gen_scalars_device(4100 + k, 1 << k)
    other = d.fe("omega")  # any second constant
    try:
        got = {}
        for on in (1, 0):
            L.h2hip_debug_set_ntt_fold_tables(ctypes.c_int(on))
            x = a.clone()
            h2.ifft_device(x, d.fe("omega_inv"), k, d.fe("ifft_divisor"))
            y = a.clone()
            h2.ifft_device(y, d.fe("omega_inv"), k, other)
            cols = [a.clone() for _ in range(3)] if k <= 21 else []
            if cols:
                h2.ifft_batch_device(cols, d.fe("omega_inv"), k, d.fe("ifft_divisor"))
            torch.cuda.synchronize()
            got[on] = (x, y, cols)
        assert torch.equal(got[1][0], got[0][0])
        assert torch.equal(got[1][1], got[0][1])
        for c1 in got[1][2] + got[0][2]:
            assert torch.equal(c1, got[0][0])
        back = got[1][0].clone()
        h2.ntt_device(back, d.fe("omega"), k)
        torch.cuda.synchronize()
        assert torch.equal(back, a)
        if k <= 17:
            assert np.array_equal(h2.to_numpy_u64(got[1][0]), oracle.ifft(h2.to_numpy_u64(a).copy(), d.fe("omega_inv"), k, d.fe("ifft_divisor"), NT))
    finally:
        L.h2hip_debug_set_ntt_fold_tables(ctypes.c_int(1))


@pytest.mark.parametrize("k", [6])
def test_commit_lagrange_identity(h2, oracle, golden, k):
    """poly/kzg/commitment.rs:361-384 test_commit_lagrange through the engine:
    commit(lagrange_to_coeff(a)) == commit_lagrange(a), a[i] = i"""
    params = h2.ParamsKZG(k, golden[f"kzg_{k}_g"], golden[f"kzg_{k}_g_lagrange"])
    try:
        d, _ = oracle.domain_new(1, k)
        dom = _domain(h2, d)
        a = golden[f"kzg_{k}_poly_lagrange"]
        b = dom.lagrange_to_coeff(a)
        assert np.array_equal(b, golden[f"kzg_{k}_poly_coeff"])
        c1 = aff(h2, params.commit(b))
        c2 = aff(h2, params.commit_lagrange(a))
        assert np.array_equal(c1, c2)
        assert np.array_equal(c1, golden[f"kzg_{k}_commit_lagrange"])
    finally:
        params.close()


def test_contract_violations_are_errors_not_crashes(h2):
    # the reference panics (assert_eq!, arithmetic.rs:133,184); the C ABI returns H2HIP_EINVAL
    bad_omega = np.full(4, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
    a = np.zeros((4, 4), dtype=np.uint64)
    with pytest.raises(h2.H2HipError):
        h2.best_fft(a, bad_omega, 2)
    with pytest.raises(AssertionError):
        h2.best_fft(a, np.zeros(4, dtype=np.uint64), 3)
    with pytest.raises(AssertionError):
        h2.best_multiexp(np.zeros((2, 4), dtype=np.uint64), np.zeros((3, 8), dtype=np.uint64))


def test_cpp_host_mirror(h2):
    """halo2-pse_amd/host/halo2hip.hpp (C++ mirror of the reference's Rust API): its own test program,
    written after poly/kzg/commitment.rs:361-384, must pass on the GPU."""
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_mirror")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host mirror tests ok" in r.stdout


def test_reentrancy_from_threads(h2, oracle):
    """keygen_pk re-enters best_fft from several rayon workers at once
    (plonk/permutation/keygen.rs:216-233): concurrent callers must each get the right answer."""
    import threading
    k = 12
    d, _ = oracle.domain_new(2, k)
    inputs = [oracle.gen_scalars(100 + t, 1 << k) for t in range(6)]
    bases = oracle.gen_points(55, 1 << k, num_threads=NT)
    want_fft = [oracle.best_fft(a, d.fe("omega"), k, 2) for a in inputs]
    want_msm = [oracle.g1_to_affine(oracle.best_multiexp(a, bases, 4)) for a in inputs]
    errs = []

    def worker(t):
        try:
            for _ in range(3):
                a = inputs[t].copy()
                h2.best_fft(a, d.fe("omega"), k)
                if not np.array_equal(a, want_fft[t]):
                    errs.append(("fft", t))
                got = aff(h2, h2.best_multiexp(inputs[t], bases))
                if not np.array_equal(got, want_msm[t]):
                    errs.append(("msm", t))
        except Exception as e:  # noqa: BLE001
            errs.append((repr(e), t))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errs, errs


def test_create_proof_call_trace_k17(h2, oracle):
    """BASELINE.json configs[4] as a call-trace replay (the Rust create_proof itself
    cannot run here): the MSM / NTT calls one KZG proof of examples/circuit-layout.rs's MyCircuit makes at
    k = 17 -- 16 commits of 2^17, 10 lagrange_to_coeff of 2^17, 10 coeff_to_extended 2^17 -> 2^19,
    1 extended_to_coeff of 2^19 -- on prover-like and dense columns, every call checked against the oracle."""
    k = 17
    n = 1 << k
    d, _ = oracle.domain_new(4, k)
    assert d.extended_k == 19
    dom = _domain(h2, d)
    g = oracle.gen_points(0xABCD, n, num_threads=NT)          # stands for params.g
    g_lagrange = oracle.gen_points(0xABCE, n, num_threads=NT)  # stands for params.g_lagrange
    params = h2.ParamsKZG(k, g, g_lagrange)
    try:
        sparse = [_prover_like(oracle, n, 500 + i) for i in range(5)]               # advice columns (plonk/prover.rs:361-365)
        dense = [oracle.gen_scalars(600 + i, n, num_threads=NT) for i in range(5)]  # lookup / permutation products
        lag_cols = sparse + dense
        # commit_lagrange x10 (advice, lookup permuted, permutation z, lookup z), commit x6 (random poly, h pieces, shplonk)
        for col in lag_cols:
            want = oracle.g1_to_affine(oracle.best_multiexp(col, g_lagrange, NT))
            assert np.array_equal(aff(h2, params.commit_lagrange(col)), want)
        coeff_cols = []
        for col in lag_cols:                                                        # lagrange_to_coeff x10
            c = dom.lagrange_to_coeff(col)
            assert np.array_equal(c, oracle.lagrange_to_coeff(d, col, NT))
            coeff_cols.append(c)
        for c in coeff_cols[:6]:                                                    # commit x6
            want = oracle.g1_to_affine(oracle.best_multiexp(c, g, NT))
            assert np.array_equal(aff(h2, params.commit(c)), want)
        ext = None
        for c in coeff_cols:                                                        # coeff_to_extended x10
            ext = dom.coeff_to_extended(c)
            assert np.array_equal(ext, oracle.coeff_to_extended(d, c, NT))
        h = dom.extended_to_coeff(ext)                                              # extended_to_coeff x1
        assert np.array_equal(h, oracle.extended_to_coeff(d, ext, NT))
    finally:
        params.close()


def test_msm_2p24_shard_fold_property(h2, oracle):
    """BASELINE.json configs[3] shape on one GPU: 2^24 pairs; the full MSM equals the fold of 8 shard MSMs
    (the multi-GPU partition), and one shard equals the oracle."""
    n = 1 << 24
    ds = h2.gen_scalars_device(0x5EED0001, n)
    dp = h2.gen_points_device(0x5EED0002, n)
    full = aff(h2, h2.msm_device(ds, dp))
    per = n // 8
    parts = [h2.msm_device(ds[i * per:(i + 1) * per], dp[i * per:(i + 1) * per]) for i in range(8)]
    assert np.array_equal(aff(h2, h2.g1_fold(np.stack(parts))), full)
    sc, bs = h2.to_numpy_u64(ds[:per]), h2.to_numpy_u64(dp[:per])
    assert np.array_equal(aff(h2, parts[0]), oracle.g1_to_affine(oracle.best_multiexp(sc, bs, NT)))


def test_msm_2p24_fixed_base_equals_eight_oracle_checked_shards(h2, oracle):
    """VERDICT r2 gap: the fixed-base form at the width it uses from 2^23 pairs up (c = 22, 2^21 buckets, 12 windows) met the
    oracle only at <= 1024 points.  Here at BASELINE.json configs[3]'s full size: EVERY one of the 8 plain-form shards (the
    multi-GPU partition, 2^21 pairs each) is checked against the oracle, and the fixed-base MSM of all 2^24 pairs -- device
    entry point and streamed host-pointer entry point -- must equal their fold."""
    n = 1 << 24
    ds = h2.gen_scalars_device(0x5EED0001, n)
    dp = h2.gen_points_device(0x5EED0002, n)
    per = n // 8
    parts = []
    for i in range(8):
        part = h2.msm_device(ds[i * per:(i + 1) * per], dp[i * per:(i + 1) * per])
        sc, bs = h2.to_numpy_u64(ds[i * per:(i + 1) * per]), h2.to_numpy_u64(dp[i * per:(i + 1) * per])
        assert np.array_equal(aff(h2, part), oracle.g1_to_affine(oracle.best_multiexp(sc, bs, NT))), i
        parts.append(part)
    want = aff(h2, h2.g1_fold(np.stack(parts)))
    h2.bases_pin_device(dp)
    try:
        assert h2.bases_pinned_info(dp)[1] == 22
        assert np.array_equal(aff(h2, h2.msm_device(ds, dp)), want)
    finally:
        h2.bases_unpin_device(dp)
    sc, bs = h2.to_numpy_u64(ds).copy(), h2.to_numpy_u64(dp).copy()
    del ds, dp
    h2.bases_pin(bs)
    try:
        assert np.array_equal(aff(h2, h2.best_multiexp(sc, bs)), want)  # chunks stream into one persistent bucket set
    finally:
        h2.bases_unpin(bs)


def test_ntt_2p24_forward_and_inverse_vs_oracle(h2, oracle):
    """VERDICT r2 gap: above 2^22 the NTT was covered by a round trip only, which a transform that is consistently wrong in a
    way its inverse undoes would pass.  2^24 points, forward and scaled inverse, limb for limb against the oracle's best_fft
    (a few seconds on the host's threads)."""
    k = 24
    d, _ = oracle.domain_new(2, k)
    da = h2.gen_scalars_device(0x5EED0003, 1 << k)
    a = h2.to_numpy_u64(da).copy()
    h2.ntt_device(da, d.fe("omega"), k)
    want = oracle.best_fft(a, d.fe("omega"), k, NT)
    assert np.array_equal(h2.to_numpy_u64(da), want)
    db = h2.gen_scalars_device(0x5EED0004, 1 << k)
    b = h2.to_numpy_u64(db).copy()
    h2.ifft_device(db, d.fe("omega_inv"), k, d.fe("ifft_divisor"))
    assert np.array_equal(h2.to_numpy_u64(db), oracle.ifft(b, d.fe("omega_inv"), k, d.fe("ifft_divisor"), NT))


@pytest.mark.parametrize("k", [24, 26])
def test_ntt_large_roundtrip(h2, oracle, k):
    """metric range 2^20..2^26: NTT then iNTT returns the input (2 GiB at k = 26); linearity spot check"""
    import torch
    d, _ = oracle.domain_new(2, k)
    da = h2.gen_scalars_device(0x5EED0003, 1 << k)
    keep = da[:4096].clone()
    ref = da.clone()
    h2.ntt_device(da, d.fe("omega"), k)
    # X[0] = sum of inputs: check against a device-side independent reduction is not available; use DC term
    # via the inverse instead: round trip must restore every limb
    h2.ifft_device(da, d.fe("omega_inv"), k, d.fe("ifft_divisor"))
    torch.cuda.synchronize()
    assert torch.equal(da, ref)
    assert torch.equal(da[:4096], keep)


@pytest.mark.parametrize("count", [1, 2, 3, 4, 7])
def test_msm_batch_matches_single_calls(h2, oracle, count):
    """h2hip_msm_bn254_batch: the back-to-back commits of plonk/prover.rs:361-365 pipelined over three
    streams and three workspace slots must equal `count` separate best_multiexp calls (and the oracle)."""
    n = (1 << 13) + 5
    bs = oracle.gen_points(900, n, num_threads=NT)
    cols = [(_prover_like(oracle, n, 910 + j) if j % 2 else oracle.gen_scalars(910 + j, n, num_threads=NT)) for j in range(count)]
    want = [oracle.g1_to_affine(oracle.best_multiexp(c, bs, NT)) for c in cols]
    got = h2.best_multiexp_batch(cols, bs)
    for j in range(count):
        assert np.array_equal(aff(h2, got[j]), want[j]), j
    # device-resident form, pinned-bases form
    import torch
    dcols = [torch.from_numpy(c.view(np.int64)).cuda() for c in cols]
    dbs = torch.from_numpy(bs.view(np.int64)).cuda()
    got = h2.msm_batch_device(dcols, dbs)
    for j in range(count):
        assert np.array_equal(aff(h2, got[j]), want[j]), j
    h2.bases_pin(bs)
    try:
        got = h2.best_multiexp_batch(cols, bs)
    finally:
        h2.bases_unpin(bs)
    for j in range(count):
        assert np.array_equal(aff(h2, got[j]), want[j]), j


def test_msm_batch_2p20_pipelined(h2, oracle):
    n = 1 << 20
    dp = h2.gen_points_device(0x5EED0002, n)
    dcols = [h2.gen_scalars_device(0x5EED0001, n, start=j * n) for j in range(5)]
    got = h2.msm_batch_device(dcols, dp)
    for j in range(5):
        assert np.array_equal(aff(h2, got[j]), aff(h2, h2.msm_device(dcols[j], dp))), j


@pytest.mark.parametrize("jk", [(4, 5), (3, 4), (4, 12), (2, 9)])
def test_divide_by_vanishing_poly(h2, oracle, golden, jk):
    j, k = jk
    d, t_eval = oracle.domain_new(j, k)
    dom = _domain(h2, d)
    if (j, k) in ((4, 5), (3, 4)):
        got = dom.divide_by_vanishing_poly(golden[f"ext_{j}_{k}_h_extended"], t_eval)
        assert np.array_equal(got, golden[f"ext_{j}_{k}_h_divided"])
    h = oracle.gen_scalars(777 + k, 1 << d.extended_k, num_threads=NT)
    assert np.array_equal(dom.divide_by_vanishing_poly(h, t_eval), oracle.divide_by_vanishing_poly(d, t_eval, h))


def test_lifecycle_and_edge_cases(h2, oracle):
    # shutdown + lazy re-init; count = 0 batch; n = 0; null-free error reporting
    sc = oracle.gen_scalars(1, 100)
    bs = oracle.gen_points(2, 100)
    want = oracle.g1_to_affine(oracle.best_multiexp(sc, bs, 2))
    assert np.array_equal(aff(h2, h2.best_multiexp(sc, bs)), want)
    h2.shutdown()
    assert np.array_equal(aff(h2, h2.best_multiexp(sc, bs)), want)  # entry points re-initialise lazily
    assert h2.best_multiexp_batch([], bs).shape == (0, 12)
    z = h2.best_multiexp_batch([sc[:0], sc[:0]], bs[:0])
    assert np.array_equal(aff(h2, z[0]), np.zeros(8, dtype=np.uint64)) and np.array_equal(aff(h2, z[1]), np.zeros(8, dtype=np.uint64))
    assert "halo2hip" in h2.version()
    assert h2.device_count() >= 1


def test_msm_chunked_inputs(h2, oracle):
    """inputs above the 2^26-pair sort limit are split into consecutive chunks whose results are added; the
    split is exercised here with the limit lowered to 1000 pairs (single and batched entry points)."""
    import ctypes
    n = 4321
    sc = [oracle.gen_scalars(40 + j, n) for j in range(3)]
    bs = oracle.gen_points(44, n, num_threads=NT)
    want = [oracle.g1_to_affine(oracle.best_multiexp(s_, bs, NT)) for s_ in sc]
    h2.lib().h2hip_debug_set_msm_max_chunk(ctypes.c_size_t(1000))
    try:
        assert np.array_equal(aff(h2, h2.best_multiexp(sc[0], bs)), want[0])
        got = h2.best_multiexp_batch(sc, bs)
        for j in range(3):
            assert np.array_equal(aff(h2, got[j]), want[j])
    finally:
        h2.lib().h2hip_debug_set_msm_max_chunk(ctypes.c_size_t(0))


def test_msm_2p26_quarters_property(h2):
    """largest size of the metric range: 2^26 pairs (2^30 sorted pairs, 17 GB of workspace) equals the fold of its quarters"""
    n = 1 << 26
    ds = h2.gen_scalars_device(0x5EED0001, n)
    dp = h2.gen_points_device(0x5EED0002, n)
    full = aff(h2, h2.msm_device(ds, dp))
    q = n // 4
    parts = [h2.msm_device(ds[i * q:(i + 1) * q], dp[i * q:(i + 1) * q]) for i in range(4)]
    assert np.array_equal(aff(h2, h2.g1_fold(np.stack(parts))), full)


def test_device_resident_column_flow(h2, oracle):
    """SURVEY.md 8(f).2 through the buffer entry points a HIP-less caller would use: upload one column, commit it
    (plonk/prover.rs:361-365), lagrange_to_coeff it in place (:487), coeff_to_extended it (evaluation.rs:311) -- one
    upload, no PCIe traffic in between -- each step checked against the oracle."""
    import ctypes
    L = h2.lib()
    k = 13
    n = 1 << k
    d, _ = oracle.domain_new(4, k)
    col = _prover_like(oracle, n, 4242)
    bases = oracle.gen_points(4243, n, num_threads=NT)
    d_col, d_bases = ctypes.c_void_p(), ctypes.c_void_p()
    en = 1 << d.extended_k
    assert L.h2hip_device_alloc(ctypes.c_size_t(en * 32), ctypes.byref(d_col)) == 0
    assert L.h2hip_device_alloc(ctypes.c_size_t(n * 64), ctypes.byref(d_bases)) == 0
    try:
        assert L.h2hip_memcpy_h2d(d_col, col.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n * 32), None) == 0
        assert L.h2hip_memcpy_h2d(d_bases, bases.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n * 64), None) == 0
        out = np.zeros(12, dtype=np.uint64)
        assert L.h2hip_msm_bn254_device(d_col, d_bases, ctypes.c_size_t(n), out.ctypes.data_as(ctypes.c_void_p), None) == 0
        assert np.array_equal(aff(h2, out), oracle.g1_to_affine(oracle.best_multiexp(col, bases, NT)))
        fe = lambda name: np.ascontiguousarray(d.fe(name)).ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        keep = [d.fe(x) for x in ("omega_inv", "ifft_divisor", "extended_omega", "g_coset", "g_coset_inv")]
        ptr = [np.ascontiguousarray(a) for a in keep]
        p = [a.ctypes.data_as(ctypes.c_void_p) for a in ptr]
        assert L.h2hip_ifft_bn254_fr_device(d_col, p[0], ctypes.c_uint32(k), p[1], None) == 0
        coeff = np.zeros((n, 4), dtype=np.uint64)
        assert L.h2hip_memcpy_d2h(coeff.ctypes.data_as(ctypes.c_void_p), d_col, ctypes.c_size_t(n * 32), None) == 0
        assert np.array_equal(coeff, oracle.lagrange_to_coeff(d, col, NT))
        assert L.h2hip_coeff_to_extended_bn254_fr_device(d_col, ctypes.c_uint32(k), ctypes.c_uint32(d.extended_k), p[2], p[3], p[4], None) == 0
        ext = np.zeros((en, 4), dtype=np.uint64)
        assert L.h2hip_memcpy_d2h(ext.ctypes.data_as(ctypes.c_void_p), d_col, ctypes.c_size_t(en * 32), None) == 0
        assert np.array_equal(ext, oracle.coeff_to_extended(d, coeff, NT))
        assert L.h2hip_stream_synchronize(None) == 0
    finally:
        L.h2hip_device_free(d_col)
        L.h2hip_device_free(d_bases)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [0, 3, 9, 14, 15, 17, 18])
def test_batched_transforms_equal_single_ones(h2, oracle, k):
    """h2hip_{ntt,ifft,coeff_to_extended}_bn254_fr_batch_device: every column of the batch equals the unbatched entry point
    (which is checked against the oracle above) -- 1-, 2- and 3-pass sizes, nine columns.  Nine columns of 2^17..2^19 points take
    the two-pass plan where a lone one takes three passes (k = 15: the coset transform; k = 17: all three), so this also sets
    the two plans side by side at those sizes."""
    import torch
    d, _ = oracle.domain_new(4, k)
    ek = d.extended_k
    cols = [h2.gen_scalars_device(4000 + i, 1 << ek) for i in range(9)]
    torch.cuda.synchronize()
    # NTT over the first 2^k elements of each column
    want = []
    for c in cols:
        t = c[:1 << k].clone()
        h2.ntt_device(t, d.fe("omega"), k)
        want.append(t)
    got = [c[:1 << k].clone() for c in cols]
    h2.ntt_batch_device(got, d.fe("omega"), k)
    torch.cuda.synchronize()
    for g, w in zip(got, want):
        assert torch.equal(g, w)
    # iNTT
    want = [w.clone() for w in got]
    for w in want:
        h2.ifft_device(w, d.fe("omega_inv"), k, d.fe("ifft_divisor"))
    h2.ifft_batch_device(got, d.fe("omega_inv"), k, d.fe("ifft_divisor"))
    torch.cuda.synchronize()
    for g, w, c in zip(got, want, cols):
        assert torch.equal(g, w) and torch.equal(g, c[:1 << k])
    # coeff_to_extended in place on the extended-size buffers
    want = [c.clone() for c in cols]
    for w in want:
        h2.coeff_to_extended_device(w, k, ek, d.fe("extended_omega"), d.fe("g_coset"), d.fe("g_coset_inv"))
    got = [c.clone() for c in cols]
    h2.coeff_to_extended_batch_device(got, k, ek, d.fe("extended_omega"), d.fe("g_coset"), d.fe("g_coset_inv"))
    torch.cuda.synchronize()
    for g, w in zip(got, want):
        assert torch.equal(g, w)


@pytest.mark.gpu
def test_batched_transforms_cut_into_several_launches(h2, oracle):
    """A batch larger than the bytes one launch may span (2 GB by default; lowered to three columns' worth here with the tuning
    hook) is cut into several launches per pass: nine columns of 2^14 -> 2^16, every one equal to the unbatched entry point"""
    import ctypes
    import torch
    k = 14
    d, _ = oracle.domain_new(4, k)
    ek = d.extended_k
    cols = [h2.gen_scalars_device(7000 + i, 1 << ek) for i in range(9)]
    want = [c.clone() for c in cols]
    for w in want:
        h2.coeff_to_extended_device(w, k, ek, d.fe("extended_omega"), d.fe("g_coset"), d.fe("g_coset_inv"))
    got = [c.clone() for c in cols]
    try:
        h2.lib().h2hip_debug_set_ntt_batch_bytes(ctypes.c_uint64(3 * 2 * (32 << ek)))  # columns + workspace of three
        h2.coeff_to_extended_batch_device(got, k, ek, d.fe("extended_omega"), d.fe("g_coset"), d.fe("g_coset_inv"))
        inv = [g.clone() for g in got]
        h2.ifft_batch_device(inv, d.fe("extended_omega_inv"), ek, d.fe("extended_ifft_divisor"))
    finally:
        h2.lib().h2hip_debug_set_ntt_batch_bytes(ctypes.c_uint64(0))
    torch.cuda.synchronize()
    for g, w in zip(got, want):
        assert torch.equal(g, w)
    one = inv[0].clone()
    h2.ntt_device(one, d.fe("extended_omega"), ek)
    torch.cuda.synchronize()
    assert torch.equal(one, got[0])  # and the cut inverse batch undoes a forward transform


@pytest.mark.gpu
@pytest.mark.parametrize("j,k", [(9, 15), (9, 17), (3, 18)])
def test_coeff_to_extended_other_padding_ratios_vs_oracle(h2, oracle, j, k):
    """coeff_to_extended with the extended domain 8 x (quotient degree 8: the first pass skips its first stage pair as it does at 4 x,
    two-pass plan at 2^20, three passes at 2^18) and 2 x (nothing to skip) the polynomial's length, against the oracle"""
    import torch
    d, _ = oracle.domain_new(j, k)
    ek = d.extended_k
    assert ek == k + (3 if j == 9 else 1)
    a = oracle.gen_scalars(8100 + k, 1 << k)
    want = oracle.coeff_to_extended(d, a, NT)
    buf = torch.zeros((1 << ek, 4), dtype=torch.int64, device="cuda")
    buf[:1 << k] = torch.from_numpy(a.view(np.int64)).cuda()
    buf[1 << k:] = 0x5A5A  # whatever lies beyond the coefficients is not read
    h2.coeff_to_extended_device(buf, k, ek, d.fe("extended_omega"), d.fe("g_coset"), d.fe("g_coset_inv"))
    torch.cuda.synchronize()
    assert np.array_equal(h2.to_numpy_u64(buf), want)
    cols = [buf.clone() for _ in range(5)]  # and batched (2^19 / 2^20 columns: two-pass plan either way)
    for c in cols:
        c[:1 << k] = torch.from_numpy(a.view(np.int64)).cuda()
    h2.coeff_to_extended_batch_device(cols, k, ek, d.fe("extended_omega"), d.fe("g_coset"), d.fe("g_coset_inv"))
    torch.cuda.synchronize()
    for c in cols:
        assert np.array_equal(h2.to_numpy_u64(c), want)


def test_batched_transforms_reject_bad_arguments(h2):
    import ctypes
    L = h2.lib()
    one = (ctypes.c_uint64 * 4)(1, 0, 0, 0)
    nul = (ctypes.c_void_p * 2)(None, None)
    assert L.h2hip_ntt_bn254_fr_batch_device(nul, ctypes.c_size_t(2), one, ctypes.c_uint32(4), None) == 1
    assert L.h2hip_ntt_bn254_fr_batch_device(None, ctypes.c_size_t(2), one, ctypes.c_uint32(4), None) == 1
    assert L.h2hip_ntt_bn254_fr_batch_device(nul, ctypes.c_size_t(2), one, ctypes.c_uint32(29), None) == 1
    assert L.h2hip_coeff_to_extended_bn254_fr_batch_device(nul, ctypes.c_size_t(1), ctypes.c_uint32(5), ctypes.c_uint32(4), one, one, one, None) == 1


def test_msm_batch_fused_equals_pipelined(h2, oracle):
    """batches of small MSMs run as one fused pass over all their windows; with the fusion switched off the same batch goes
    through the stream pipeline.  Both must give the same points as separate calls -- 16 columns of 2^15 + 3 pairs (dense and
    prover-like, so over-full buckets too), and a batch larger than one fused run holds (2^18 pairs x 15)"""
    import ctypes
    L = h2.lib()
    n = (1 << 15) + 3
    dp = h2.gen_points_device(321, n)
    dense = [h2.gen_scalars_device(700 + j, n) for j in range(12)]
    import torch
    sparse = [torch.from_numpy(_prover_like(oracle, n, 800 + j).view(np.int64)).cuda() for j in range(4)]
    cols = dense + sparse
    want = [aff(h2, h2.msm_device(c, dp)) for c in cols]
    fused = h2.msm_batch_device(cols, dp)
    try:
        L.h2hip_debug_set_msm_fuse_small(0)
        piped = h2.msm_batch_device(cols, dp)
    finally:
        L.h2hip_debug_set_msm_fuse_small(1)
    for j in range(len(cols)):
        assert np.array_equal(aff(h2, fused[j]), want[j]), j
        assert np.array_equal(aff(h2, piped[j]), want[j]), j
    n = 1 << 18
    dp = h2.gen_points_device(322, n)
    cols = [h2.gen_scalars_device(900 + j, n) for j in range(15)]  # 2^18 x 19 windows each: 13 fit one fused run, so two runs
    got = h2.msm_batch_device(cols, dp)
    for j in (0, 3, 12, 13, 14):
        assert np.array_equal(aff(h2, got[j]), aff(h2, h2.msm_device(cols[j], dp))), j


def test_alt_bn128_published_known_answer_on_the_gpu(h2, oracle):
    """EIP-196's ecMul((1, 2), 2) through the engine's MSM (plain and fixed-base form): a known answer published outside this
    repository (see tests/test_oracle_golden.py::test_alt_bn128_published_known_answers)."""
    from test_oracle_golden import ALT_BN128_2G, ALT_BN128_Q, ALT_BN128_R, _affine_ints, _g1
    g = _g1(oracle)
    n = 1500
    bs = np.ascontiguousarray(np.repeat(g[None, :], n, axis=0))
    sc = np.zeros((n, 4), dtype=np.uint64)
    sc[0] = oracle.fe_from_int(oracle.FR, 2)                      # 2 G + 0 + ...
    assert _affine_ints(oracle, aff(h2, h2.best_multiexp(sc, bs))) == ALT_BN128_2G
    sc[:] = oracle.fe_from_int(oracle.FR, 1)
    sc[7] = oracle.fe_from_int(oracle.FR, ALT_BN128_R - (n - 3))  # the scalars sum to 2 mod r: every bucket path, same answer
    assert _affine_ints(oracle, aff(h2, h2.best_multiexp(sc, bs))) == ALT_BN128_2G
    h2.bases_pin(bs)
    try:
        assert _affine_ints(oracle, aff(h2, h2.best_multiexp(sc, bs))) == ALT_BN128_2G
        sc[7] = oracle.fe_from_int(oracle.FR, ALT_BN128_R - (n - 1))  # sum = 0: the identity, encoded (0, 0)
        assert not aff(h2, h2.best_multiexp(sc, bs)).any()
        sc[7] = oracle.fe_from_int(oracle.FR, ALT_BN128_R - n)        # sum = -1: -G = (1, q - 2)
        assert _affine_ints(oracle, aff(h2, h2.best_multiexp(sc, bs))) == (1, ALT_BN128_Q - 2)
    finally:
        h2.bases_unpin(bs)
