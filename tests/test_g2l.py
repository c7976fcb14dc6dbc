"""g_to_lagrange (arithmetic.rs:277-301; best_fft with G = G1, used by ParamsKZG::downsize, poly/kzg/commitment.rs:267-275):
the oracle against the definition-minted golden vectors and against ParamsKZG::setup's closed-form g_lagrange (CPU), and
the HIP implementation against the oracle (GPU)."""
import os

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def g2l_golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "g2l.npz"), allow_pickle=False)


@pytest.mark.parametrize("k", range(0, 6))
def test_oracle_g_to_lagrange_golden(oracle, g2l_golden, k):
    got = oracle.g_to_lagrange(g2l_golden["g2l_k%d_g" % k], k)
    assert np.array_equal(got, g2l_golden["g2l_k%d_out" % k])


def test_oracle_g_to_lagrange_matches_setup(oracle):
    """ParamsKZG::setup computes g_lagrange from the secret in closed form (poly/kzg/commitment.rs:89-117); g_to_lagrange of
    the same setup's g must give the same points -- two independent routes to [l_i(s)]G"""
    k = 5
    s = oracle.gen_scalars(4242, 1)[0]
    g, gl = oracle.kzg_setup(k, s)
    assert np.array_equal(oracle.g_to_lagrange(g, k, num_threads=4), gl)


def test_golden_kzg_params_g_lagrange(oracle, golden):
    """the RawBytes SRS fixture (k = 6): its g_lagrange is g_to_lagrange(g)"""
    raw = np.fromfile(os.path.join(ROOT, "tests", "golden", "kzg_6_params.rawbytes"), dtype=np.uint8)
    k = int(raw[:4].view("<u4")[0])
    n = 1 << k
    pts = raw[4:4 + 2 * n * 64].view("<u8").reshape(2, n, 8)
    assert np.array_equal(oracle.g_to_lagrange(pts[0].copy(), k, num_threads=4), pts[1])


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(0, 6))
def test_gpu_g_to_lagrange_golden(h2, g2l_golden, k):
    got = h2.g_to_lagrange(g2l_golden["g2l_k%d_g" % k], k)
    assert np.array_equal(got, g2l_golden["g2l_k%d_out" % k])


@pytest.mark.gpu
@pytest.mark.parametrize("k", [7, 10, 12])
def test_gpu_g_to_lagrange_vs_oracle(h2, oracle, k):
    g = oracle.gen_points(900 + k, 1 << k, num_threads=8)
    if k == 7:
        g[5] = 0  # an identity among the inputs
    want = oracle.g_to_lagrange(g, k, num_threads=16)
    assert np.array_equal(h2.g_to_lagrange(g, k), want)


@pytest.mark.gpu
def test_gpu_downsize_commit_lagrange_identity(h2, oracle):
    """ParamsKZG::downsize (poly/kzg/commitment.rs:267-275) then the reference's own test_commit_lagrange identity
    (:361-384) at the smaller size: commit(lagrange_to_coeff(a)) == commit_lagrange(a) with g_lagrange from the GPU"""
    K, k = 12, 9
    # g = [s^i]G for i < 2^K is too slow to mint at K = 12 on one core with the oracle's scalar ladder; random points
    # are as good for the identity, which is linear algebra over the group
    g = oracle.gen_points(31337, 1 << K, num_threads=8)
    params = h2.ParamsKZG(K, g, g_lagrange=None)
    params.downsize(k)
    assert params.k == k and params.g.shape[0] == 1 << k and params.g_lagrange.shape[0] == 1 << k
    od, _ = oracle.domain_new(2, k)
    d = h2.EvaluationDomain.new(od.quotient_poly_degree + 1, od.k)
    a = oracle.gen_scalars(5, 1 << k)
    coeffs = d.lagrange_to_coeff(a)
    lhs = h2.g1_to_affine(params.commit(coeffs))
    rhs = h2.g1_to_affine(params.commit_lagrange(a))
    assert np.array_equal(lhs, rhs)
    params.close()


def test_g_to_lagrange_rejects_bad_arguments(h2):
    """contract violations are H2HIP_EINVAL before any device work"""
    import ctypes
    L = h2.lib()
    buf = np.zeros((2, 8), dtype=np.uint64)
    p = buf.ctypes.data_as(ctypes.c_void_p)
    assert L.h2hip_g_to_lagrange_bn254(None, ctypes.c_uint32(1), p) == 1
    assert L.h2hip_g_to_lagrange_bn254(p, ctypes.c_uint32(1), None) == 1
    assert L.h2hip_g_to_lagrange_bn254(p, ctypes.c_uint32(29), p) == 1   # beyond the 2-adicity of Fr
    assert b"g_to_lagrange" in L.h2hip_last_error()
    with pytest.raises(AssertionError):
        h2.g_to_lagrange(np.zeros((3, 8), dtype=np.uint64), 2)           # a.len() == 1 << log_n (arithmetic.rs:184)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [3, 6])
def test_kzg_setup_golden(h2, golden, k):
    """ParamsKZG::setup (poly/kzg/commitment.rs:61-129) on the GPU against the textbook SRS minted with Python integers"""
    h2.init()
    params = h2.ParamsKZG.setup(k, golden[f"kzg_{k}_secret"])
    try:
        assert np.array_equal(params.g, golden[f"kzg_{k}_g"])
        assert np.array_equal(params.g_lagrange, golden[f"kzg_{k}_g_lagrange"])
        # the reference's own identity (:361-384) over the fresh parameters
        got = h2.g1_to_affine(params.commit_lagrange(golden[f"kzg_{k}_poly_lagrange"]))
        assert np.array_equal(got, golden[f"kzg_{k}_commit_lagrange"])
    finally:
        params.close()


@pytest.mark.gpu
def test_kzg_setup_vs_oracle_and_contract(h2, oracle):
    h2.init()
    for k, seed in ((0, 11), (1, 12), (9, 13)):
        s = oracle.gen_scalars(seed, 1)[0]
        g, gl = oracle.kzg_setup(k, s)
        params = h2.ParamsKZG.setup(k, s)
        try:
            assert np.array_equal(params.g, g) and np.array_equal(params.g_lagrange, gl), k
        finally:
            params.close()
    # g_lagrange from setup equals g_to_lagrange(g) (what downsize recomputes, :274)
    s = oracle.gen_scalars(14, 1)[0]
    params = h2.ParamsKZG.setup(7, s)
    try:
        assert np.array_equal(h2.g_to_lagrange(params.g, 7), params.g_lagrange)
    finally:
        params.close()
    # a secret that is an n-th root of unity makes the reference panic at the inversion (:100): an error here
    with pytest.raises(h2.H2HipError):
        h2.ParamsKZG.setup(4, 1)
    with pytest.raises(h2.H2HipError):
        h2.ParamsKZG.setup(29, 5)


# ---------------------------------------------------------------------------- best_fft::<G1> with a caller-supplied omega
def _jac(affine):
    """(n, 8) affine points -> (n, 12) Jacobian with z = one (Montgomery form), identity (0, 0) -> z = 0"""
    from oracle import oracle as orc
    one = orc.fe_from_int(orc.FQ, 1)
    out = np.zeros((affine.shape[0], 12), dtype=np.uint64)
    out[:, :8] = affine
    ident = ~affine.any(axis=1)
    out[~ident, 8:] = one
    out[ident, 4:8] = one
    return out


def _aff_all(orc, jac):
    return np.stack([orc.g1_to_affine(p) for p in jac])


@pytest.mark.parametrize("k", [0, 1, 3, 5])
def test_oracle_best_fft_g1_is_the_scalar_ntt_in_the_exponent(oracle, k):
    """an independent pin of the curve-point FFT: for P_i = [a_i]G, best_fft::<G1>(P, omega) = [best_fft::<Fr>(a, omega)_j]G -- the
    oracle's G1 butterflies against its Fr transform (itself pinned by the O(n^2) golden DFTs) and plain scalar multiplication"""
    d, _ = oracle.domain_new(2, max(k, 1))
    omega = d.fe("omega") if k else oracle.fe_from_int(oracle.FR, 1)
    n = 1 << k
    a = oracle.gen_scalars(31 + k, n)
    gen = np.zeros(8, dtype=np.uint64)
    gen[:4], gen[4:] = oracle.fe_from_int(oracle.FQ, 1), oracle.fe_from_int(oracle.FQ, 2)
    pts = np.stack([oracle.g1_to_affine(oracle.g1_mul(gen, a[i])) for i in range(n)])
    got = _aff_all(oracle, oracle.best_fft_g1(_jac(pts), omega, k, num_threads=2))
    fa = oracle.best_fft(a, omega, k, 1) if k else a
    want = np.stack([oracle.g1_to_affine(oracle.g1_mul(gen, fa[i])) for i in range(n)])
    assert np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [0, 1, 4, 8, 11, 15])
def test_gpu_best_fft_g1_vs_oracle(h2, oracle, k):
    """arithmetic.rs:171-234 with G = bn256::G1 and the caller's omega (forward root here; g_to_lagrange passes the inverse one):
    the same group elements as the oracle's iterative butterflies, identities and non-trivial z among the inputs"""
    h2.init()
    d, _ = oracle.domain_new(2, max(k, 1))
    omega = d.fe("omega") if k else oracle.fe_from_int(oracle.FR, 1)
    n = 1 << k
    pts = oracle.gen_points(700 + k, n, num_threads=8)
    a = _jac(pts)
    if n >= 8:
        a[3] = _jac(np.zeros((1, 8), dtype=np.uint64))[0]     # an identity
        a[5] = oracle.best_multiexp(oracle.gen_scalars(9, 4), pts[:4], 1)  # a point with z != 1
    want = _aff_all(oracle, oracle.best_fft_g1(a, omega, k, num_threads=16))
    got = a.copy()
    h2.best_fft_g1(got, omega, k)
    assert np.array_equal(_aff_all(oracle, got), want)
    assert all((p[8:] == oracle.fe_from_int(oracle.FQ, 1)).all() or not p[8:].any() for p in got)  # z = 1, or the identity's z = 0


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 2, 5, 9, 13, 14])
def test_gpu_small_transforms_lazy_and_normalised_paths_agree(h2, oracle, k):
    """k <= 14: the layers keep their points XYZZ and normalise once at the end (round 4) -- same affine output as round 3's path (a
    normalisation after every layer), as the one-lane ladder, and as the oracle; identities and repeated points among the inputs"""
    import ctypes
    h2.init()
    L = h2.lib()
    n = 1 << k
    g = oracle.gen_points(1300 + k, n, num_threads=8)
    if n >= 8:
        g[2] = 0
        g[6] = g[1]
    want = oracle.g_to_lagrange(g, k, num_threads=16)
    try:
        for mode in (1, 3, 0):  # lazy quad (default), normalised quad, one lane per butterfly
            L.h2hip_debug_set_g2l_quad(ctypes.c_int(mode))
            assert np.array_equal(h2.g_to_lagrange(g, k), want), mode
    finally:
        L.h2hip_debug_set_g2l_quad(ctypes.c_int(1))
