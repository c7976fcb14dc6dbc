"""CPU: pin oracle/bn254_oracle.c against the Python big-int golden vectors
(tests/golden/make_golden.py).  The reference holds no BN254 byte vectors,
so these vectors plus the algebraic identities of the reference's own tests are the pin."""
import numpy as np
import pytest

MSM_CASES = ["1", "2", "3", "4", "31", "32", "33", "100", "1024", "zeros", "ones", "rm1", "single", "sparse", "cancel"]


def test_constants(oracle, golden):
    for which, name in ((oracle.FQ, "fq"), (oracle.FR, "fr")):
        assert np.array_equal(oracle.constant(which, 0), golden[f"const_{name}_R"])
        assert np.array_equal(oracle.constant(which, 1), golden[f"const_{name}_R2"])
        assert np.array_equal(oracle.constant(which, 2), golden[f"const_{name}_modulus"])
        assert oracle.inv64(which) == int(golden[f"const_{name}_inv64"][0])
    assert np.array_equal(oracle.constant(oracle.FR, 3), golden["const_fr_root_of_unity"])
    assert np.array_equal(oracle.constant(oracle.FR, 4), golden["const_fr_root_of_unity_inv"])
    assert np.array_equal(oracle.constant(oracle.FR, 5), golden["const_fr_zeta"])
    # Appendix A literals
    assert oracle.inv64(oracle.FQ) == 0x87d20782e4866389
    assert oracle.inv64(oracle.FR) == 0xc2e1f593efffffff


@pytest.mark.parametrize("name", ["fq", "fr"])
def test_field_ops(oracle, golden, name):
    which = oracle.FQ if name == "fq" else oracle.FR
    a, b = golden[f"field_{name}_a"], golden[f"field_{name}_b"]
    assert np.array_equal(oracle.fe_binop("mul", which, a, b), golden[f"field_{name}_mul"])
    assert np.array_equal(oracle.fe_binop("add", which, a, b), golden[f"field_{name}_add"])
    assert np.array_equal(oracle.fe_binop("sub", which, a, b), golden[f"field_{name}_sub"])
    assert np.array_equal(oracle.fe_to_canonical(which, a), golden[f"field_{name}_canon"])
    assert np.array_equal(oracle.fe_from_canonical(which, golden[f"field_{name}_canon"]), a)


@pytest.mark.parametrize("case", MSM_CASES)
@pytest.mark.parametrize("threads", [1, 3, 8])
def test_best_multiexp_golden(oracle, golden, case, threads):
    sc, bs = golden[f"msm_{case}_scalars"], golden[f"msm_{case}_bases"]
    got = oracle.g1_to_affine(oracle.best_multiexp(sc, bs, threads))
    assert np.array_equal(got, golden[f"msm_{case}_result"])


def test_window_rule(oracle):
    # arithmetic.rs:16-22 and the SURVEY.md 8(a) table (m=8192 -> c=10, 131072 -> 12, 2097152 -> 15)
    assert [oracle.window_c(m) for m in (1, 3, 4, 31, 32, 8192, 131072, 2097152)] == [1, 1, 3, 3, 4, 10, 12, 15]


def test_known_answers(oracle):
    gen = np.concatenate([oracle.fe_from_int(oracle.FQ, 1), oracle.fe_from_int(oracle.FQ, 2)])
    assert oracle.g1_on_curve(gen)
    r_minus_1 = oracle.fe_from_int(oracle.FR, -1)
    one = oracle.fe_from_int(oracle.FR, 1)
    # [r-1]G + G = identity  ([r]G = inf)
    p = oracle.g1_add(oracle.g1_mul(gen, r_minus_1), oracle.g1_mul(gen, one))
    assert np.array_equal(oracle.g1_to_affine(p), np.zeros(8, dtype=np.uint64))
    # omega^(2^27) = -1
    rou = oracle.constant(oracle.FR, 3)
    x = rou.reshape(1, 4)
    for _ in range(27):
        x = oracle.fe_binop("mul", oracle.FR, x, x)
    assert np.array_equal(x[0], r_minus_1)


@pytest.mark.parametrize("k", range(0, 11))
@pytest.mark.parametrize("threads", [1, 8])
def test_best_fft_golden(oracle, golden, k, threads):
    # threads=1 exercises the recursive variant (log_n > log_threads) except k=0;
    # threads=8 exercises the iterative variant for k <= 3 (arithmetic.rs:202)
    got = oracle.best_fft(golden[f"ntt_{k}_in"], golden[f"ntt_{k}_omega"], k, threads)
    assert np.array_equal(got, golden[f"ntt_{k}_out"])
    got = oracle.ifft(golden[f"ntt_{k}_in"], golden[f"ifft_{k}_omega_inv"], k, golden[f"ifft_{k}_divisor"], threads)
    assert np.array_equal(got, golden[f"ifft_{k}_out"])


@pytest.mark.parametrize("jk", [(4, 5), (3, 4), (2, 3)])
def test_domain_golden(oracle, golden, jk):
    j, k = jk
    d, t_eval = oracle.domain_new(j, k)
    tag = f"domain_{j}_{k}"
    assert d.extended_k == int(golden[tag + "_extended_k"][0])
    for f in ("omega", "omega_inv", "extended_omega", "extended_omega_inv", "g_coset", "g_coset_inv",
              "ifft_divisor", "extended_ifft_divisor", "barycentric_weight"):
        assert np.array_equal(d.fe(f), golden[f"{tag}_{f}"]), f
    assert np.array_equal(t_eval, golden[tag + "_t_evaluations"])
    ext = oracle.coeff_to_extended(d, golden[f"ext_{j}_{k}_coeffs"])
    assert np.array_equal(ext, golden[f"ext_{j}_{k}_extended"])
    h = oracle.extended_to_coeff(d, golden[f"ext_{j}_{k}_h_extended"])
    assert np.array_equal(h, golden[f"ext_{j}_{k}_h_coeffs"])
    div = oracle.divide_by_vanishing_poly(d, t_eval, golden[f"ext_{j}_{k}_h_extended"])
    assert np.array_equal(div, golden[f"ext_{j}_{k}_h_divided"])


def test_plonk_api_pinned_extended_k(oracle):
    # tests/plonk_api.rs:629-632 pins `k: 5, extended_k: 7` for a degree-5.. circuit; cs.degree()=4 -> j=4
    d, _ = oracle.domain_new(4, 5)
    assert (d.k, d.extended_k) == (5, 7)
    # EvaluationDomain::new(4, 17) -> extended_k = 19
    d, _ = oracle.domain_new(4, 17)
    assert d.extended_k == 19


@pytest.mark.parametrize("k", [3, 6])
def test_kzg_setup_and_commit_lagrange(oracle, golden, k):
    """poly/kzg/commitment.rs:361-384 test_commit_lagrange, with the secret fixed"""
    g, gl = oracle.kzg_setup(k, golden[f"kzg_{k}_secret"])
    assert np.array_equal(g, golden[f"kzg_{k}_g"])
    assert np.array_equal(gl, golden[f"kzg_{k}_g_lagrange"])
    a = golden[f"kzg_{k}_poly_lagrange"]
    d, _ = oracle.domain_new(1, k)
    b = oracle.lagrange_to_coeff(d, a)
    assert np.array_equal(b, golden[f"kzg_{k}_poly_coeff"])
    c1 = oracle.g1_to_affine(oracle.best_multiexp(b, g, 4))      # params.commit(&b)
    c2 = oracle.g1_to_affine(oracle.best_multiexp(a, gl, 4))     # params.commit_lagrange(&a)
    assert np.array_equal(c1, c2)
    assert np.array_equal(c1, golden[f"kzg_{k}_commit_lagrange"])


def test_generator_golden(oracle, golden):
    assert np.array_equal(oracle.gen_scalars(0x5EED0001, 64), golden["gen_scalars_5EED0001"])
    assert np.array_equal(oracle.gen_points(0x5EED0002, 64), golden["gen_points_5EED0002"])
    assert np.array_equal(oracle.gen_scalars(0x5EED0003, 64), golden["gen_ntt_5EED0003"])
    assert np.array_equal(oracle.gen_scalars(0x5EED0001, 8, start=1000), golden["gen_scalars_offset1000"])
    assert np.array_equal(oracle.gen_points(0x5EED0002, 8, start=1000), golden["gen_points_offset1000"])
    assert np.array_equal(oracle.gen_points(0x5EED0002, 64, num_threads=4), golden["gen_points_5EED0002"])
    for p in golden["gen_points_5EED0002"][:8]:
        assert oracle.g1_on_curve(p)


def test_multiexp_thread_independence_and_naive(oracle):
    # App. B rule 3: the group element is independent of T; also equals per-term double-and-add
    sc = oracle.gen_scalars(11, 300)
    bs = oracle.gen_points(12, 300)
    ref = oracle.g1_to_affine(oracle.naive_multiexp(sc, bs))
    for t in (1, 2, 7, 8, 64, 299, 300, 301):
        assert np.array_equal(oracle.g1_to_affine(oracle.best_multiexp(sc, bs, t)), ref), t


def test_fft_roundtrip_and_linearity(oracle):
    k = 12
    a = oracle.gen_scalars(21, 1 << k)
    b = oracle.gen_scalars(22, 1 << k)
    d, _ = oracle.domain_new(2, k)
    fa = oracle.best_fft(a, d.fe("omega"), k, 8)
    back = oracle.ifft(fa, d.fe("omega_inv"), k, d.fe("ifft_divisor"), 8)
    assert np.array_equal(back, a)
    fb = oracle.best_fft(b, d.fe("omega"), k, 1)
    fab = oracle.best_fft(oracle.fe_binop("add", oracle.FR, a, b), d.fe("omega"), k, 4)
    assert np.array_equal(fab, oracle.fe_binop("add", oracle.FR, fa, fb))


# ---- published known answers (not minted by this repository) -------------------------------------------------------------------
# bn256 of halo2curves is alt_bn128, the curve of Ethereum's EIP-196 / EIP-197 precompiles: y^2 = x^3 + 3 over F_q, generator (1, 2),
# group order r.  The constants below are the ones those EIPs publish (decimal, as printed there); they pin the oracle's moduli,
# Montgomery conversion and group law to bytes that no script in this repository produced.
ALT_BN128_Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
ALT_BN128_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
ALT_BN128_2G = (1368015179489954701390400359078579693043519447331113978918064868415326638035,
                9918110051302171585080402603319702774565515993150576347155970296011118125764)


def _affine_ints(oracle, xy):
    c = oracle.fe_to_canonical(oracle.FQ, np.ascontiguousarray(xy).reshape(2, 4))
    return oracle.int_from_limbs(c[0]), oracle.int_from_limbs(c[1])


def _g1(oracle):
    return np.concatenate([oracle.fe_from_int(oracle.FQ, 1), oracle.fe_from_int(oracle.FQ, 2)])


def test_alt_bn128_published_known_answers(oracle):
    assert oracle.int_from_limbs(oracle.constant(oracle.FQ, 2)) == ALT_BN128_Q
    assert oracle.int_from_limbs(oracle.constant(oracle.FR, 2)) == ALT_BN128_R
    g = _g1(oracle)
    assert oracle.g1_on_curve(g)
    two = oracle.fe_from_int(oracle.FR, 2)
    assert _affine_ints(oracle, oracle.g1_to_affine(oracle.g1_mul(g, two))) == ALT_BN128_2G            # EIP-196 ecMul((1, 2), 2)
    assert _affine_ints(oracle, oracle.g1_to_affine(oracle.best_multiexp(two[None, :], g[None, :]))) == ALT_BN128_2G
    # (r - 1) G = -G = (1, q - 2); r G = identity, encoded (0, 0)
    assert _affine_ints(oracle, oracle.g1_to_affine(oracle.g1_mul(g, oracle.fe_from_int(oracle.FR, ALT_BN128_R - 1)))) == (1, ALT_BN128_Q - 2)
    gg = np.stack([g, g])
    sc = np.stack([oracle.fe_from_int(oracle.FR, ALT_BN128_R - 1), oracle.fe_from_int(oracle.FR, 1)])
    assert not oracle.g1_to_affine(oracle.best_multiexp(sc, gg)).any()
    # G + G through the addition path, and 2G + 2G = 4G = [4]G
    dbl = oracle.g1_add(np.concatenate([g, oracle.fe_from_int(oracle.FQ, 1)]), np.concatenate([g, oracle.fe_from_int(oracle.FQ, 1)]))
    assert _affine_ints(oracle, oracle.g1_to_affine(dbl)) == ALT_BN128_2G
    four = oracle.g1_to_affine(oracle.g1_mul(g, oracle.fe_from_int(oracle.FR, 4)))
    assert np.array_equal(oracle.g1_to_affine(oracle.g1_add(dbl, dbl)), four)
