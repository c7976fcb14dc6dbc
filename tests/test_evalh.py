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


@pytest.fixture(params=["interpreter", "generated"])
def gates_kernel(request, h2):
    """run a GPU test twice: custom gates through the byte-code interpreter (code generation off) and through the kernel generated for
    the circuit and compiled inline by hiprtc (mode 2) -- the counters must show that the generated kernel is the one that ran"""
    L = h2.lib()
    h2.init()
    before = (ctypes.c_uint64 * 5)()
    L.h2hip_debug_evalh_codegen_stats(before)
    L.h2hip_debug_set_evalh_codegen(ctypes.c_int(0 if request.param == "interpreter" else 2), ctypes.c_uint32(0))
    mode = {"kind": request.param, "expect_generated": request.param == "generated"}  # a test whose program is beyond the generator's limits clears the flag
    yield mode
    after = (ctypes.c_uint64 * 5)()
    L.h2hip_debug_evalh_codegen_stats(after)
    L.h2hip_debug_set_evalh_codegen(ctypes.c_int(1), ctypes.c_uint32(0))
    if request.param == "generated":
        assert after[1] == before[1], "hiprtc rejected a generated kernel: " + L.h2hip_last_error().decode()
        if mode["expect_generated"]:
            assert after[2] > before[2], "the generated kernel never ran"
    else:
        assert after[2] == before[2]


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["k3", "k4"])
def test_gpu_evaluate_h_golden(h2, oracle, evalh_golden, tag, gates_kernel):
    case, vin, vout = load_case(evalh_golden, tag)
    if tag == "k4":  # scalars at addresses that are only 8-byte aligned, as inside a Rust struct
        for f in ("extended_omega", "g_coset", "g_coset_inv", "zeta", "delta", "y", "beta", "gamma", "theta"):
            buf = np.zeros(7, dtype=np.uint64)
            off = 1 if buf.ctypes.data % 16 == 0 else 2
            buf[off:off + 4] = case[f]
            case[f] = buf[off:off + 4]
            assert case[f].ctypes.data % 16 == 8
    h = DescHolder(case)
    values = vin.copy()
    rc = h2.lib().h2hip_evaluate_h_bn254(h.byref(), values.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, h2.lib().h2hip_last_error()
    assert np.array_equal(values, vout)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [5, 9, 13])
def test_gpu_evaluate_h_vs_oracle(h2, oracle, k, gates_kernel):
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


def test_evaluate_h_rejects_malformed_descriptions(h2, oracle, evalh_golden):
    """a description a kernel could fault on is H2HIP_EINVAL before any device work (no GPU needed): an intermediate
    beyond num_intermediates, a column beyond n_advice, a Horner part range past the parts array, a graph that is
    not in single-assignment order, a permutation column out of range, too few permutation sets"""
    import copy
    L = h2.lib()
    case, vin, _ = load_case(evalh_golden, "k3")

    def rc_of(mutate):
        c = copy.deepcopy(case)
        mutate(c)
        h = DescHolder(c)
        v = vin.copy()
        return L.h2hip_evaluate_h_bn254(h.byref(), v.ctypes.data_as(ctypes.c_void_p))

    def bad_target(c): c["custom"]["calcs"][0, 1] = 10_000
    def bad_column(c): c["custom"]["calcs"][0, 3] = 99          # x.a of the first Store: advice column 99
    def bad_rotation(c): c["custom"]["calcs"][0, 4] = 99        # x.b: rotation index
    def bad_parts(c): c["custom"]["calcs"][-1, 9] = 1000        # Horner parts_count
    def bad_op(c): c["custom"]["calcs"][0, 0] = 8
    def rewritten_target(c): c["custom"]["calcs"][1, 1] = c["custom"]["calcs"][0, 1]   # not single-assignment
    def read_before_write(c): c["custom"]["calcs"][0, 2:5] = (1, c["custom"]["calcs"][-1, 1], 0)  # x = a later intermediate
    def bad_perm_col(c): c["perm_column_index"] = np.array([0, 1, 77], dtype=np.uint32)
    def few_sets(c): c["chunk_len"] = 1
    def bad_domain(c): c["extended_k"] = 2
    def bad_lookup(c): c["lookups"][0][0]["calcs"][0, 3] = 99

    for m in (bad_target, bad_column, bad_rotation, bad_parts, bad_op, rewritten_target, read_before_write, bad_perm_col, few_sets, bad_domain, bad_lookup):
        assert rc_of(m) == 1, m.__name__
        assert b"evaluate_h" in L.h2hip_last_error()
    assert L.h2hip_evaluate_h_bn254(None, None) == 1


def test_evaluator_new_builds_the_golden_graphs(evalh_golden):
    """Evaluator.new over the circuit-layout gate polynomials reproduces the graphs stored with the golden vectors"""
    from evalh_util import Evaluator, flatten_graph
    A = lambda col, rot=0: ('advice', col, rot)  # noqa: E731
    F = lambda col, rot=0: ('fixed', col, rot)  # noqa: E731
    e_, a_, b_, c_, d_ = 0, 1, 2, 3, 4
    sf, sm, sa, sb, sc, sl = 0, 1, 2, 3, 4, 5
    gate = ('sum',
            ('sum', ('sum', ('sum', ('prod', A(a_), F(sa)), ('prod', A(b_), F(sb))), ('prod', ('prod', A(a_), A(b_)), F(sm))),
             ('neg', ('prod', A(c_), F(sc)))),
            ('prod', F(sf), ('prod', A(d_, 1), A(e_, -1))))
    gate2 = ('sum', ('scaled', ('prod', A(c_), A(c_)), 7), ('sum', ('prod', ('const', 2), A(d_)), ('neg', ('const', 5))))
    ev = Evaluator.new([gate, gate2], [([A(a_)], [F(sl)])])
    for name, g in (("custom", ev.custom_gates), ("lookup0", ev.lookups[0])):
        f = flatten_graph(g)
        for key in ("constants", "rotations", "calcs", "parts"):
            assert np.array_equal(f[key], evalh_golden["k3_" + name + "_" + key]), (name, key)


@pytest.mark.gpu
@pytest.mark.parametrize("group_bytes", [0, 1])
def test_gpu_evaluate_h_device_resident(h2, oracle, group_bytes, gates_kernel):
    """h2hip_evaluate_h_bn254_device: every column already in HBM (torch tensors), values folded in place on the current
    stream, inputs untouched; equal to the oracle and to the host-pointer entry point.  group_bytes = 1: one lookup per group
    of coset buffers (the second lookup's cosets are formed after the first lookup's kernel, in the reused buffers) instead of
    all lookups' cosets in the batch of the advice columns"""
    import torch
    h2.lib().h2hip_debug_set_evalh_lookup_group_bytes(ctypes.c_uint64(group_bytes))
    case, vin = _random_case(oracle, 11, seed=77)
    h = DescHolder(case)
    want = vin.copy()
    assert oracle.lib().oracle_evaluate_h(h.byref(), want.ctypes.data_as(ctypes.c_void_p)) == 0
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()  # noqa: E731
    keep = []
    tens = {}
    for key in ("fixed_cosets", "advice_polys", "instance_polys", "perm_product_cosets", "perm_cosets"):
        tens[key] = [dev(a) for a in case[key]]
    for key in ("l0", "l_last", "l_active_row"):
        tens[key] = dev(case[key])
    tens["lookups"] = [[dev(p) for p in l[1:]] for l in case["lookups"]]
    before = {k: ([t.clone() for t in v] if isinstance(v, list) and not isinstance(v[0], list) else v) for k, v in tens.items() if k != "lookups"}
    hd = DescHolder(case)  # host description first, then swap in the device addresses
    d = hd.desc

    def table(ts):
        arr = (ctypes.c_void_p * max(1, len(ts)))(*[t.data_ptr() for t in ts])
        keep.append(arr)
        return ctypes.addressof(arr)

    d.fixed_cosets, d.advice_polys, d.instance_polys = table(tens["fixed_cosets"]), table(tens["advice_polys"]), table(tens["instance_polys"])
    d.perm_product_cosets, d.perm_cosets = table(tens["perm_product_cosets"]), table(tens["perm_cosets"])
    d.l0, d.l_last, d.l_active_row = tens["l0"].data_ptr(), tens["l_last"].data_ptr(), tens["l_active_row"].data_ptr()
    d.lookup_product_polys = table([l[0] for l in tens["lookups"]])
    d.lookup_permuted_input_polys = table([l[1] for l in tens["lookups"]])
    d.lookup_permuted_table_polys = table([l[2] for l in tens["lookups"]])
    d_values = dev(vin)
    stream = torch.cuda.current_stream().cuda_stream
    try:
        rc = h2.lib().h2hip_evaluate_h_bn254_device(hd.byref(), ctypes.c_void_p(d_values.data_ptr()), ctypes.c_void_p(stream))
        assert rc == 0, h2.lib().h2hip_last_error()
        torch.cuda.synchronize()
        got_host = vin.copy()  # the host-pointer entry point under the same grouping
        rc = h2.lib().h2hip_evaluate_h_bn254(h.byref(), got_host.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0, h2.lib().h2hip_last_error()
    finally:
        h2.lib().h2hip_debug_set_evalh_lookup_group_bytes(ctypes.c_uint64(0))
    got = d_values.cpu().numpy().view(np.uint64)
    assert np.array_equal(got, want)
    assert np.array_equal(got_host, want)
    for key in ("fixed_cosets", "advice_polys", "instance_polys", "perm_product_cosets", "perm_cosets"):
        for t0, t1 in zip(before[key], tens[key]):
            assert torch.equal(t0, t1), key  # inputs are read-only


def _compile_stats(h2, graph_dict):
    """(n_ops, n_slots) of the program evaluate_h compiles a flattened graph into (host-only debug entry point)"""
    from evalh_util import DescHolder as DH
    holder = DH.__new__(DH)
    holder.keep = []
    G = holder._graph(graph_dict)
    n_ops, n_slots = ctypes.c_uint32(), ctypes.c_uint32()
    rc = h2.lib().h2hip_debug_evalh_compile_stats(ctypes.byref(G), ctypes.byref(n_ops), ctypes.byref(n_slots))
    assert rc == 0, h2.lib().h2hip_last_error()
    return n_ops.value, n_slots.value


def _many_live_graph(n_live):
    """a hand-made single-assignment graph whose n_live products are all alive at once: they are consumed in reverse
    order of creation, so no slot can be reused before the last product exists"""
    from evalh_util import CALC_ADD, CALC_HORNER, CALC_MUL, GraphEvaluator, VS_ADVICE, VS_FIXED, VS_PREVIOUS, VS_Y
    g = GraphEvaluator()
    r0, r1 = g.add_rotation(0), g.add_rotation(1)
    prods = [g.add_calculation((CALC_MUL, (VS_ADVICE, i % 5, r0 if i % 2 else r1), (VS_FIXED, i % 6, r0), ())) for i in range(n_live)]
    # distinct coefficients keep add_calculation from deduplicating: t_i = p_i * const_i
    terms = [g.add_calculation((CALC_MUL, p, g.add_constant(1000 + i), ())) for i, p in enumerate(prods)]
    acc = terms[-1]
    for t in reversed(terms[:-1]):
        acc = g.add_calculation((CALC_ADD, acc, t, ()))
    g.add_calculation((CALC_HORNER, (VS_PREVIOUS, 0, 0), (VS_Y, 0, 0), (acc,)))
    return g


def test_compiled_program_is_compact(h2, oracle):
    """Stores are folded into operands, the Horner is spread over its parts, and intermediates share slots: the
    three-gate system needs a handful of slots; the reversed-consumption graph needs one per live product"""
    from evalh_util import flatten_graph
    case, _ = _random_case(oracle, 5, seed=5, n_gates=40)
    g = case["custom"]
    assert g["num_intermediates"] > 256
    n_ops, n_slots = _compile_stats(h2, g)
    assert n_slots <= 32 and n_ops <= g["calcs"].shape[0] + g["parts"].shape[0] + 1
    n_ops, n_slots = _compile_stats(h2, flatten_graph(_many_live_graph(300)))
    assert 300 <= n_slots <= 302
    assert _compile_stats(h2, {"constants": np.zeros((0, 4), np.uint64), "rotations": np.zeros(0, np.int32),
                               "calcs": np.zeros((0, 10), np.uint32), "parts": np.zeros((0, 3), np.uint32), "num_intermediates": 0}) == (0, 0)


@pytest.mark.gpu
def test_gpu_evaluate_h_large_graph(h2, oracle, gates_kernel):
    """more intermediates than any per-lane scratch tier holds one-to-one (the reference's Vec is unbounded): slots are
    shared by lifetime"""
    case, vin = _random_case(oracle, 8, seed=21, n_gates=40)
    assert case["custom"]["num_intermediates"] > 256
    h = DescHolder(case)
    want = vin.copy()
    assert oracle.lib().oracle_evaluate_h(h.byref(), want.ctypes.data_as(ctypes.c_void_p)) == 0
    got = vin.copy()
    assert h2.lib().h2hip_evaluate_h_bn254(h.byref(), got.ctypes.data_as(ctypes.c_void_p)) == 0, h2.lib().h2hip_last_error()
    assert np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("forced", [False, True])
def test_gpu_evaluate_h_global_slot_workspace(h2, oracle, forced):
    """programs whose live values exceed 256 slots run from a global workspace with rows taken grid-stride; `forced`
    pushes every program of an ordinary system through that form with the debug knob, the other case gets there by
    itself with 300 simultaneously live products"""
    from evalh_util import flatten_graph
    L = h2.lib()
    case, vin = _random_case(oracle, 9, seed=33)
    if not forced:
        case["custom"] = flatten_graph(_many_live_graph(300))
    h = DescHolder(case)
    want = vin.copy()
    assert oracle.lib().oracle_evaluate_h(h.byref(), want.ctypes.data_as(ctypes.c_void_p)) == 0
    got = vin.copy()
    try:
        if forced:
            L.h2hip_debug_set_evalh_max_local_slots(0)
        rc = L.h2hip_evaluate_h_bn254(h.byref(), got.ctypes.data_as(ctypes.c_void_p))
    finally:
        L.h2hip_debug_set_evalh_max_local_slots(256)
    assert rc == 0, L.h2hip_last_error()
    assert np.array_equal(got, want)


def test_cpp_graph_evaluator_matches_golden_graphs(evalh_golden, tmp_path):
    """halo2-pse_amd/host/evaluation.hpp (C++ mirror of GraphEvaluator / Evaluator::new): the circuit-layout system built
    with the C++ Expression operators flattens to the graphs stored with the golden vectors (host only)"""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_mirror")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    out = tmp_path / "graphs.txt"
    r = subprocess.run([exe, "--dump-graphs", str(out)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    got = {}
    for line in out.read_text().splitlines():
        name, kind, *vals = line.split()
        got.setdefault((name, kind), []).append([int(v) for v in vals])
    for name in ("custom", "lookup0"):
        p = "k3_" + name + "_"
        assert got[(name, "num_intermediates")] == [[int(evalh_golden[p + "num_intermediates"][0])]]
        assert np.array_equal(np.array(got[(name, "constant")], dtype=np.uint64), evalh_golden[p + "constants"])
        assert np.array_equal(np.array(got[(name, "rotation")], dtype=np.int32).reshape(-1), evalh_golden[p + "rotations"])
        assert np.array_equal(np.array(got[(name, "calc")], dtype=np.uint32), evalh_golden[p + "calcs"])
        assert np.array_equal(np.array(got.get((name, "part"), []), dtype=np.uint32).reshape(-1, 3), evalh_golden[p + "parts"])


def _strip(case, perm=True, lookups=True):
    c = dict(case)
    if not perm:
        c["perm_product_cosets"], c["perm_cosets"] = [], []
        c["perm_column_kind"] = np.zeros(0, dtype=np.uint32)
        c["perm_column_index"] = np.zeros(0, dtype=np.uint32)
    if not lookups:
        c["lookups"] = []
    return c


@pytest.mark.parametrize("variant", ["no_perm", "no_lookups", "gates_only", "empty_graph", "horner_without_parts"])
def test_oracle_evaluate_h_degenerate_systems(oracle, variant):
    """the reference's loops simply do not run for a system without permutation / lookups (evaluation.rs:362, :443); an
    Evaluator with no gates still holds Horner(PreviousValue, [], y) and returns the previous value; a GraphEvaluator with
    no calculations evaluates to zero (:745-749).  Pinned here on the oracle by hand-computable outcomes."""
    from evalh_util import GraphEvaluator, custom_gates_graph, flatten_graph
    case, vin = _random_case(oracle, 4, seed=9)
    if variant == "horner_without_parts":
        case = _strip(case, perm=False, lookups=False)
        case["custom"] = flatten_graph(custom_gates_graph([]))
        want = vin  # unchanged
    elif variant == "empty_graph":
        case = _strip(case, perm=False, lookups=False)
        case["custom"] = flatten_graph(GraphEvaluator())
        want = np.zeros_like(vin)
    else:
        case = _strip(case, perm=variant not in ("no_perm", "gates_only"), lookups=variant not in ("no_lookups", "gates_only"))
        want = None
    h = DescHolder(case)
    got = vin.copy()
    assert oracle.lib().oracle_evaluate_h(h.byref(), got.ctypes.data_as(ctypes.c_void_p)) == 0
    if want is not None:
        assert np.array_equal(got, want)
    else:
        assert not np.array_equal(got, vin)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["no_perm", "no_lookups", "gates_only", "empty_graph", "horner_without_parts"])
def test_gpu_evaluate_h_degenerate_systems(h2, oracle, variant, gates_kernel):
    from evalh_util import GraphEvaluator, custom_gates_graph, flatten_graph
    case, vin = _random_case(oracle, 6, seed=10)
    if variant == "horner_without_parts":
        case = _strip(case, perm=False, lookups=False)
        case["custom"] = flatten_graph(custom_gates_graph([]))
    elif variant == "empty_graph":
        case = _strip(case, perm=False, lookups=False)
        case["custom"] = flatten_graph(GraphEvaluator())
        gates_kernel["expect_generated"] = False  # no operations: nothing to generate, the interpreter's empty loop stores the zero
    else:
        case = _strip(case, perm=variant not in ("no_perm", "gates_only"), lookups=variant not in ("no_lookups", "gates_only"))
    h = DescHolder(case)
    want = vin.copy()
    assert oracle.lib().oracle_evaluate_h(h.byref(), want.ctypes.data_as(ctypes.c_void_p)) == 0
    got = vin.copy()
    assert h2.lib().h2hip_evaluate_h_bn254(h.byref(), got.ctypes.data_as(ctypes.c_void_p)) == 0, h2.lib().h2hip_last_error()
    assert np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("n_live", [2, 6, 12, 40, 100])
def test_gpu_evaluate_h_every_slot_tier(h2, oracle, n_live, gates_kernel):
    """slots live in registers (up to 4 / 8), per-lane scratch (16 / 64 / 256) or the global workspace: one graph per tier,
    each with n_live products alive at once, against the oracle"""
    from evalh_util import flatten_graph
    case, vin = _random_case(oracle, 7, seed=50 + n_live)
    case["custom"] = flatten_graph(_many_live_graph(n_live))
    _, n_slots = _compile_stats(h2, case["custom"])
    assert n_live <= n_slots <= n_live + 2
    if n_slots > 48:
        gates_kernel["expect_generated"] = False  # more live values than the generator takes on: the interpreter serves them
    h = DescHolder(case)
    want = vin.copy()
    assert oracle.lib().oracle_evaluate_h(h.byref(), want.ctypes.data_as(ctypes.c_void_p)) == 0
    got = vin.copy()
    assert h2.lib().h2hip_evaluate_h_bn254(h.byref(), got.ctypes.data_as(ctypes.c_void_p)) == 0, h2.lib().h2hip_last_error()
    assert np.array_equal(got, want)


def _random_graph(rng, n_calcs, n_adv=5, n_fix=6):
    """a random single-assignment graph straight in calculation form (not via add_expression): every operation kind, operands
    drawn from columns / constants / challenges / earlier intermediates / beta-gamma-theta-y / PreviousValue, Store chains,
    dead calculations, Horners with 0..6 parts anywhere in the graph, long runs of additions of un-reduced column values"""
    from evalh_util import (CALC_ADD, CALC_DOUBLE, CALC_HORNER, CALC_MUL, CALC_NEGATE, CALC_SQUARE, CALC_STORE, CALC_SUB, GraphEvaluator,
                            VS_ADVICE, VS_BETA, VS_CHALLENGE, VS_CONSTANT, VS_FIXED, VS_GAMMA, VS_INSTANCE, VS_INTERMEDIATE, VS_PREVIOUS, VS_THETA,
                            VS_Y)
    g = GraphEvaluator()
    for r in (0, 1, -1, 2, -3):
        g.add_rotation(r)
    for _ in range(4):
        g.add_constant(int(rng.integers(0, 1 << 62)) ** 3)
    targets = []

    def src():
        kind = int(rng.integers(0, 10))
        if kind <= 3 and targets:
            return (VS_INTERMEDIATE, targets[int(rng.integers(0, len(targets)))], 0)
        if kind <= 5:
            return (VS_ADVICE, int(rng.integers(0, n_adv)), int(rng.integers(0, 5)))
        if kind == 6:
            return (VS_FIXED, int(rng.integers(0, n_fix)), int(rng.integers(0, 5)))
        if kind == 7:
            return (VS_CONSTANT, int(rng.integers(0, len(g.constants))), 0)
        if kind == 8:
            return [(VS_INSTANCE, 0, int(rng.integers(0, 5))), (VS_CHALLENGE, int(rng.integers(0, 2)), 0), (VS_PREVIOUS, 0, 0)][int(rng.integers(0, 3))]
        return [(VS_BETA, 0, 0), (VS_GAMMA, 0, 0), (VS_THETA, 0, 0), (VS_Y, 0, 0)][int(rng.integers(0, 4))]

    def emit(calc):
        t = g.num_intermediates
        g.calculations.append((calc, t))  # no deduplication: repeated calculations are legal input too
        g.num_intermediates += 1
        targets.append(t)

    for _ in range(n_calcs):
        op = [CALC_ADD, CALC_ADD, CALC_SUB, CALC_MUL, CALC_MUL, CALC_SQUARE, CALC_DOUBLE, CALC_NEGATE, CALC_HORNER, CALC_STORE][int(rng.integers(0, 10))]
        if op in (CALC_ADD, CALC_SUB, CALC_MUL):
            emit((op, src(), src(), ()))
        elif op == CALC_HORNER:
            emit((op, src(), src(), tuple(src() for _ in range(int(rng.integers(0, 7))))))
        else:
            emit((op, src(), None, ()))
    return g


def _codegen(h2, graph_dict, compile_it=True, lookup=False):
    """(source, seconds, code bytes) of the per-circuit gates kernel for a flattened graph: host-only, hiprtc cross-compiles gfx950 without a GPU"""
    from evalh_util import DescHolder as DH
    holder = DH.__new__(DH)
    holder.keep = []
    G = holder._graph(graph_dict)
    L = h2.lib()
    n = ctypes.c_size_t()
    assert L.h2hip_debug_evalh_codegen_source(ctypes.byref(G), None, ctypes.c_size_t(0), ctypes.byref(n), ctypes.c_int(2 if lookup else 0), None, None) == 0
    buf = ctypes.create_string_buffer(n.value + 1)
    secs, size = ctypes.c_double(), ctypes.c_size_t()
    rc = L.h2hip_debug_evalh_codegen_source(ctypes.byref(G), buf, ctypes.c_size_t(n.value + 1), ctypes.byref(n), ctypes.c_int((1 if compile_it else 0) | (2 if lookup else 0)),
                                            ctypes.byref(secs), ctypes.byref(size))
    assert rc == 0, L.h2hip_last_error().decode()
    return buf.value.decode(), secs.value, size.value


def test_generated_gates_kernel_compiles_for_gfx950(h2, oracle, evalh_golden):
    """CPU: the straight-line source evalh.hip emits for a circuit's custom gates -- the reference's own example circuit (k = 4 golden
    graph), a 40-gate system, a graph with 40 values alive at once and random graphs with every operation kind -- is accepted by hiprtc
    for gfx950 from the embedded headers alone; one statement per program operation."""
    from evalh_util import flatten_graph
    case, _, _ = load_case(evalh_golden, "k4")
    src, secs, size = _codegen(h2, case["custom"])
    n_ops, n_slots = _compile_stats(h2, case["custom"])
    assert 'extern "C" __global__ void' in src and "evalh_gates_gen" in src and size > 1000
    # one statement per operation, except the products emitted inside the sum that consumes them (two per fused ADD / SUB, one per Horner step)
    assert sum(1 for ln in src.splitlines() if ln.startswith("    s") and " = " in ln) >= n_ops - 2 * src.count("fu_mul_sub<UF>")
    big, _ = _random_case(oracle, 5, seed=5, n_gates=40)
    src, secs, size = _codegen(h2, big["custom"])
    assert size > 1000 and secs < 120
    _, _, size = _codegen(h2, flatten_graph(_many_live_graph(40)))
    assert size > 1000
    rng = np.random.default_rng(11)
    for _ in range(4):
        _, _, size = _codegen(h2, flatten_graph(_random_graph(rng, int(rng.integers(5, 90)))))
        assert size > 1000
    # the lookup arguments' kernels: the compressed table expression, then the argument's five constraints (lookup_row)
    for lk in list(case["lookups"]) + list(big["lookups"][:1]):
        src, _, size = _codegen(h2, lk[0], lookup=True)
        assert "evalh_lookup_gen" in src and "lookup_row(l, c, idx," in src and size > 1000


def test_compile_survives_random_graphs(h2):
    """host-only: validation + compilation of 200 random graphs neither fails nor needs more slots than intermediates"""
    from evalh_util import flatten_graph
    rng = np.random.default_rng(2024)
    for i in range(200):
        g = flatten_graph(_random_graph(rng, int(rng.integers(1, 120))))
        n_ops, n_slots = _compile_stats(h2, g)
        assert n_slots <= g["num_intermediates"] and n_ops <= g["calcs"].shape[0] + g["parts"].shape[0] + 2


@pytest.mark.gpu
def test_gpu_evaluate_h_random_graphs(h2, oracle, gates_kernel):
    """40 random graphs (custom gates and one lookup each) against the oracle: exercises Store folding, Horner hoisting, dead
    code, slot reuse and the inserted magnitude reductions far from the shapes add_expression produces"""
    from evalh_util import flatten_graph
    rng = np.random.default_rng(77)
    case, vin = _random_case(oracle, 5, seed=123)
    for i in range(40):
        case["custom"] = flatten_graph(_random_graph(rng, int(rng.integers(1, 90))))
        lk = list(case["lookups"][0])
        lk[0] = flatten_graph(_random_graph(rng, int(rng.integers(1, 40))))
        case["lookups"] = [tuple(lk), case["lookups"][1]]
        h = DescHolder(case)
        want = vin.copy()
        assert oracle.lib().oracle_evaluate_h(h.byref(), want.ctypes.data_as(ctypes.c_void_p)) == 0
        got = vin.copy()
        assert h2.lib().h2hip_evaluate_h_bn254(h.byref(), got.ctypes.data_as(ctypes.c_void_p)) == 0, h2.lib().h2hip_last_error()
        assert np.array_equal(got, want), "graph %d" % i


@pytest.mark.gpu
def test_background_compile_switches_kernels_without_changing_values(h2, oracle):
    """HALO2_HIP_EVALH_CODEGEN=1, the default: the first calls on a circuit the process has not seen run the interpreter while hiprtc
    compiles the circuit's kernels on a background thread; once they are there the generated kernels take over.  Every call, before and
    after the switch, returns the oracle's values; the switch is seen in the counters within two minutes."""
    import time
    L = h2.lib()
    h2.init()
    L.h2hip_debug_set_evalh_codegen(ctypes.c_int(1), ctypes.c_uint32(0))
    case, vin = _random_case(oracle, 6, seed=int(time.time()) % 100000 + 31337, n_gates=5)  # a program no earlier test has compiled
    h = DescHolder(case)
    want = vin.copy()
    assert oracle.lib().oracle_evaluate_h(h.byref(), want.ctypes.data_as(ctypes.c_void_p)) == 0
    st = (ctypes.c_uint64 * 5)()
    L.h2hip_debug_evalh_codegen_stats(st)
    launches0, interp0, failed0 = st[2], st[3], st[1]
    deadline = time.time() + 120
    switched = False
    calls = 0
    while time.time() < deadline:
        got = vin.copy()
        assert L.h2hip_evaluate_h_bn254(h.byref(), got.ctypes.data_as(ctypes.c_void_p)) == 0, L.h2hip_last_error()
        assert np.array_equal(got, want), "call %d" % calls
        calls += 1
        L.h2hip_debug_evalh_codegen_stats(st)
        if st[2] > launches0:
            switched = True
            break
        time.sleep(0.2)
    assert st[1] == failed0, L.h2hip_last_error().decode()
    assert switched, "no generated kernel after %d calls in 120 s" % calls
    assert st[3] > interp0, "the first call cannot have found a compiled kernel"
    for _ in range(3):  # and after the switch
        got = vin.copy()
        assert L.h2hip_evaluate_h_bn254(h.byref(), got.ctypes.data_as(ctypes.c_void_p)) == 0
        assert np.array_equal(got, want)


_DISK_CACHE = r"""
import ctypes, json, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
from conftest import load_pkg
from oracle import oracle
import test_evalh as T
h2 = load_pkg()
h2.init()
L = h2.lib()
L.h2hip_debug_set_evalh_codegen(ctypes.c_int(2), ctypes.c_uint32(0))
case, vin = T._random_case(oracle, 6, seed=77)
h = T.DescHolder(case)
want = vin.copy()
assert oracle.lib().oracle_evaluate_h(h.byref(), want.ctypes.data_as(ctypes.c_void_p)) == 0
got = vin.copy()
assert L.h2hip_evaluate_h_bn254(h.byref(), got.ctypes.data_as(ctypes.c_void_p)) == 0, L.h2hip_last_error()
st = (ctypes.c_uint64 * 5)()
L.h2hip_debug_evalh_codegen_stats(st)
print("RESULT " + json.dumps({"match": bool(np.array_equal(got, want)), "compiled": st[0], "failed": st[1], "launches": st[2], "disk_hits": st[4],
                              "files": sorted(os.listdir(os.environ["HALO2_HIP_CACHE_DIR"]))}))
"""


@pytest.mark.gpu
def test_generated_kernel_disk_cache_across_processes(tmp_path):
    """HALO2_HIP_CACHE_DIR: the first process compiles the circuit's gates kernel and leaves the code object behind, a second process
    finds it (no hiprtc compile: `disk_hits`), and both match the oracle"""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, HALO2_HIP_CACHE_DIR=str(tmp_path))
    outs = []
    for _ in range(2):
        r = subprocess.run([sys.executable, "-c", _DISK_CACHE % {"root": ROOT}], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        outs.append(json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:]))
    first, second = outs
    assert first["match"] and second["match"]
    # one code object for the custom gates and one per lookup argument whose table expression is distinct
    assert first["disk_hits"] == 0 and first["compiled"] >= 1 and first["launches"] >= 1 and len(first["files"]) == first["compiled"]
    assert all(f.endswith("_gfx950.co") for f in first["files"])
    assert second["disk_hits"] == first["compiled"] and second["launches"] >= 1 and second["files"] == first["files"]


@pytest.mark.gpu
def test_gpu_evaluate_h_with_pinned_key_columns(h2, oracle):
    """h2hip_columns_pin: the proving key's constant columns (fixed cosets, l0 / l_last / l_active_row, permutation cosets) stay in HBM across
    host-pointer evaluate_h calls.  Same values as the oracle with and without the pins; pinning is idempotent; a column whose memory was
    rewritten (what a freed and reused Vec looks like) is noticed and uploaded again; unpin releases everything and ignores strangers."""
    h2.init()
    case, vin = _random_case(oracle, 9, seed=31)
    h = DescHolder(case)
    want = vin.copy()
    assert oracle.lib().oracle_evaluate_h(h.byref(), want.ctypes.data_as(ctypes.c_void_p)) == 0

    def run():
        got = vin.copy()
        assert h2.lib().h2hip_evaluate_h_bn254(h.byref(), got.ctypes.data_as(ctypes.c_void_p)) == 0, h2.lib().h2hip_last_error()
        return got

    assert np.array_equal(run(), want)
    key_cols = list(case["fixed_cosets"]) + list(case["perm_cosets"]) + [case["l0"], case["l_last"], case["l_active_row"]]
    key_cols = [c_ for c_ in key_cols if c_.shape[0] == 1 << case["extended_k"]]
    before = h2.columns_pinned_info()
    h2.columns_pin(key_cols)
    h2.columns_pin(key_cols)  # idempotent
    n_pinned, nbytes = h2.columns_pinned_info()
    assert n_pinned == before[0] + len(key_cols) and nbytes == before[1] + len(key_cols) * (32 << case["extended_k"])
    assert np.array_equal(run(), want) and np.array_equal(run(), want)
    # a pinned column's memory now holds other values (its first element is among the sampled ones): the stale copy must not be used
    victim = case["fixed_cosets"][0]
    saved = victim.copy()
    victim[:] = oracle.gen_scalars(4242, victim.shape[0], num_threads=4)
    want2 = vin.copy()
    assert oracle.lib().oracle_evaluate_h(h.byref(), want2.ctypes.data_as(ctypes.c_void_p)) == 0
    assert np.array_equal(run(), want2) and not np.array_equal(want2, want)
    assert h2.columns_pinned_info()[0] == n_pinned - 1  # the lookup dropped the stale entry
    victim[:] = saved
    assert np.array_equal(run(), want)
    h2.columns_unpin(key_cols + [vin])  # a pointer that was never pinned is ignored
    assert h2.columns_pinned_info() == before
    assert np.array_equal(run(), want)
