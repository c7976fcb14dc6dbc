"""Evaluator::evaluate_h (plonk/evaluation.rs:280-522): the oracle's restatement against the
Python big-integer golden values (CPU), and the HIP implementation against the oracle (GPU)."""
import ctypes

import numpy as np
import pytest

from evalh_util import DescHolder


def load_case(z, tag):
    p = tag + "_"
    g = lambda name: {k: z[p + name + "_" + k] for k in ("constants", "rotations", "calcs", "parts")} | {  # noqa: E731
        "num_intermediates": int(z[p + name + "_num_intermediates"][0])}
    k, ek = (int(v) for v in z[p + "k"])
    chunk_len, last_rotation = (int(v) for v in z[p + "perm_params"])
    case = {
        "k": k, "extended_k": ek,
        **{f: z[p + f] for f in ("extended_omega", "g_coset", "g_coset_inv", "zeta", "delta", "y", "beta", "gamma", "theta",
                                 "l0", "l_last", "l_active_row", "perm_column_kind", "perm_column_index")},
        "fixed_cosets": list(z[p + "fixed_cosets"]), "advice_polys": list(z[p + "advice_polys"]), "instance_polys": [],
        "challenges": np.zeros((0, 4), dtype=np.uint64),
        "custom": g("custom"),
        "perm_product_cosets": list(z[p + "perm_product_cosets"]), "perm_cosets": list(z[p + "perm_cosets"]),
        "chunk_len": chunk_len, "last_rotation": last_rotation,
        "lookups": [(g("lookup0"), z[p + "lookup0_product_poly"], z[p + "lookup0_permuted_input_poly"], z[p + "lookup0_permuted_table_poly"])],
    }
    return case, z[p + "values_in"], z[p + "values_out"]


@pytest.fixture(scope="module")
def evalh_golden():
    import os
    from conftest import ROOT
    return np.load(os.path.join(ROOT, "tests", "golden", "evalh.npz"), allow_pickle=False)


@pytest.mark.parametrize("tag", ["k3", "k4"])
def test_oracle_evaluate_h_golden(oracle, evalh_golden, tag):
    case, vin, vout = load_case(evalh_golden, tag)
    h = DescHolder(case)
    values = vin.copy()
    rc = oracle.lib().oracle_evaluate_h(h.byref(), values.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    assert np.array_equal(values, vout)


def test_graph_builder_matches_reference_shapes():
    """the flattened custom-gate graph of the circuit-layout gate: one Store per distinct query, the products / sums of
    add_expression (evaluation.rs:591-706), and the final Horner over the gate polynomials with y"""
    from evalh_util import CALC_HORNER, CALC_STORE, VS_PREVIOUS, VS_Y, custom_gates_graph
    A = lambda col, rot=0: ('advice', col, rot)  # noqa: E731
    F = lambda col, rot=0: ('fixed', col, rot)  # noqa: E731
    gate = ('sum', ('prod', A(1), F(2)), ('neg', ('prod', A(3), F(4))))
    g = custom_gates_graph([gate])
    ops = [c[0][0] for c in g.calculations]
    assert ops.count(CALC_STORE) == 4 and ops[-1] == CALC_HORNER
    last = g.calculations[-1][0]
    assert last[1] == (VS_PREVIOUS, 0, 0) and last[2] == (VS_Y, 0, 0) and len(last[3]) == 1
    assert g.constants[:3] == [0, 1, 2] and g.rotations == [0]


def _random_case(oracle, k, seed, n_gates=3):
    """a larger synthetic constraint system over random columns (same shape as the golden one, more gates)"""
    from evalh_util import custom_gates_graph, flatten_graph, lookup_graph
    rng = np.random.default_rng(seed)
    ek = k + 2
    n, size = 1 << k, 1 << ek
    d, _ = oracle.domain_new(4, k)
    assert d.extended_k == ek
    col = lambda m, s: oracle.gen_scalars(seed * 1000 + s, m)  # noqa: E731
    A = lambda c, r=0: ('advice', c, r)  # noqa: E731
    F = lambda c, r=0: ('fixed', c, r)  # noqa: E731
    I = lambda c, r=0: ('instance', c, r)  # noqa: E731
    gates = []
    for gi in range(n_gates):
        a, b, c = (int(x) for x in rng.integers(0, 5, 3))
        f1, f2 = (int(x) for x in rng.integers(0, 6, 2))
        r1, r2 = (int(x) for x in rng.integers(-2, 3, 2))
        gates.append(('sum', ('prod', ('prod', A(a, r1), A(b)), F(f1)),
                      ('sum', ('neg', ('prod', A(c, r2), F(f2, r1))), ('scaled', ('sum', I(0), ('challenge', 0)), 3 + gi))))
    lookups = []
    for li in range(2):
        lg = lookup_graph([A(li), ('prod', A(li + 1), F(li))], [F(5), F(li + 2, 1)])
        lookups.append((flatten_graph(lg), col(n, 50 + 3 * li), col(n, 51 + 3 * li), col(n, 52 + 3 * li)))
    fr = lambda v: oracle.fe_from_int(oracle.FR, v)  # noqa: E731
    zeta = oracle.constant(oracle.FR, 5)
    delta = fr(pow(7, 1 << 28, 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001))
    case = {
        "k": k, "extended_k": ek, "extended_omega": d.fe("extended_omega"), "g_coset": d.fe("g_coset"), "g_coset_inv": d.fe("g_coset_inv"),
        "zeta": zeta, "delta": delta, "y": col(1, 1)[0], "beta": col(1, 2)[0], "gamma": col(1, 3)[0], "theta": col(1, 4)[0],
        "l0": col(size, 5), "l_last": col(size, 6), "l_active_row": col(size, 7),
        "fixed_cosets": [col(size, 10 + i) for i in range(6)], "advice_polys": [col(n, 20 + i) for i in range(5)],
        "instance_polys": [col(n, 30)], "challenges": col(2, 31),
        "custom": flatten_graph(custom_gates_graph(gates)),
        "perm_product_cosets": [col(size, 40 + i) for i in range(3)], "perm_cosets": [col(size, 44 + i) for i in range(5)],
        "perm_column_kind": np.array([0, 0, 1, 2, 0], dtype=np.uint32), "perm_column_index": np.array([1, 2, 3, 0, 4], dtype=np.uint32),
        "chunk_len": 2, "last_rotation": -6, "lookups": lookups,
    }
    return case, col(size, 99)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["k3", "k4"])
def test_gpu_evaluate_h_golden(h2, oracle, evalh_golden, tag):
    case, vin, vout = load_case(evalh_golden, tag)
    h = DescHolder(case)
    values = vin.copy()
    rc = h2.lib().h2hip_evaluate_h_bn254(h.byref(), values.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, h2.lib().h2hip_last_error()
    assert np.array_equal(values, vout)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [5, 9, 13])
def test_gpu_evaluate_h_vs_oracle(h2, oracle, k):
    """bigger synthetic systems (instance column, challenges, mixed permutation column kinds, two lookups, three
    permutation sets with a ragged last chunk): GPU == oracle, limb for limb"""
    case, vin = _random_case(oracle, k, seed=k)
    h = DescHolder(case)
    want = vin.copy()
    assert oracle.lib().oracle_evaluate_h(h.byref(), want.ctypes.data_as(ctypes.c_void_p)) == 0
    got = vin.copy()
    rc = h2.lib().h2hip_evaluate_h_bn254(h.byref(), got.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, h2.lib().h2hip_last_error()
    assert np.array_equal(got, want)
