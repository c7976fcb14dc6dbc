"""Host-side mirror of halo2_proofs::plonk::evaluation (plonk/evaluation.rs) over the C ABI of include/halo2hip.h.

  GraphEvaluator   the graph builder (evaluation.rs:191-201, :526-706): add_rotation / add_constant / add_calculation /
                   add_expression with the reference's deduplication and operand ordering, so a flattened graph has the
                   same calculations, in the same order, as the reference builds for the same Expression
  Evaluator        Evaluator::new (:221-279) from gate polynomials + lookup argument expressions, and evaluate_h
                   (:280-522) which hands the flattened description to h2hip_evaluate_h_bn254[_device]
  ValueSource / Calculation / Graph / EvalhDesc    ctypes mirrors of the h2hip_* structs;  DescHolder builds one from numpy arrays

Expression trees are tuples:  ('const', int) ('fixed', col, rot) ('advice', col, rot) ('instance', col, rot)
('challenge', i) ('neg', e) ('sum', e, e) ('prod', e, e) ('scaled', e, int)   (plonk/circuit.rs Expression).
ValueSource = (kind, a, b) with the enum order of evaluation.rs:37-60, so tuple comparison equals the derived PartialOrd.
There is no CPU evaluation here: evaluate_h raises if the HIP library or the GPU is missing.
"""
import ctypes

import numpy as np

R_MOD = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
RR = 1 << 256

VS_CONSTANT, VS_INTERMEDIATE, VS_FIXED, VS_ADVICE, VS_INSTANCE, VS_CHALLENGE, VS_BETA, VS_GAMMA, VS_THETA, VS_Y, VS_PREVIOUS = range(11)
CALC_ADD, CALC_SUB, CALC_MUL, CALC_SQUARE, CALC_DOUBLE, CALC_NEGATE, CALC_HORNER, CALC_STORE = range(8)
ANY_ADVICE, ANY_FIXED, ANY_INSTANCE = 0, 1, 2


class GraphEvaluator:
    """GraphEvaluator (evaluation.rs:191-201, Default :526-539)"""

    def __init__(self):
        self.constants = [0, 1, 2]
        self.rotations = []
        self.calculations = []  # (calc tuple, target)
        self.num_intermediates = 0

    def add_rotation(self, rot):                      # :543-552
        if rot in self.rotations:
            return self.rotations.index(rot)
        self.rotations.append(rot)
        return len(self.rotations) - 1

    def add_constant(self, c):                        # :555-564
        c %= R_MOD
        if c in self.constants:
            return (VS_CONSTANT, self.constants.index(c), 0)
        self.constants.append(c)
        return (VS_CONSTANT, len(self.constants) - 1, 0)

    def add_calculation(self, calc):                  # :570-588
        for c, target in self.calculations:
            if c == calc:
                return (VS_INTERMEDIATE, target, 0)
        target = self.num_intermediates
        self.calculations.append((calc, target))
        self.num_intermediates += 1
        return (VS_INTERMEDIATE, target, 0)

    def add_expression(self, e):                      # :591-706
        zero, one, two = (VS_CONSTANT, 0, 0), (VS_CONSTANT, 1, 0), (VS_CONSTANT, 2, 0)
        tag = e[0]
        if tag == 'const':
            return self.add_constant(e[1])
        if tag in ('fixed', 'advice', 'instance'):
            kind = {'fixed': VS_FIXED, 'advice': VS_ADVICE, 'instance': VS_INSTANCE}[tag]
            rot_idx = self.add_rotation(e[2])
            return self.add_calculation((CALC_STORE, (kind, e[1], rot_idx), None, ()))
        if tag == 'challenge':
            return self.add_calculation((CALC_STORE, (VS_CHALLENGE, e[1], 0), None, ()))
        if tag == 'neg':
            if e[1][0] == 'const':
                return self.add_constant(-e[1][1])
            ra = self.add_expression(e[1])
            return ra if ra == zero else self.add_calculation((CALC_NEGATE, ra, None, ()))
        if tag == 'sum':
            a, b = e[1], e[2]
            if b[0] == 'neg':                         # undo subtraction stored as a + (-b)
                ra = self.add_expression(a)
                rb = self.add_expression(b[1])
                if ra == zero:
                    return self.add_calculation((CALC_NEGATE, rb, None, ()))
                if rb == zero:
                    return ra
                return self.add_calculation((CALC_SUB, ra, rb, ()))
            ra = self.add_expression(a)
            rb = self.add_expression(b)
            if ra == zero:
                return rb
            if rb == zero:
                return ra
            return self.add_calculation((CALC_ADD, ra, rb, ()) if ra <= rb else (CALC_ADD, rb, ra, ()))
        if tag == 'prod':
            ra = self.add_expression(e[1])
            rb = self.add_expression(e[2])
            if ra == zero or rb == zero:
                return zero
            if ra == one:
                return rb
            if rb == one:
                return ra
            if ra == two:
                return self.add_calculation((CALC_DOUBLE, rb, None, ()))
            if rb == two:
                return self.add_calculation((CALC_DOUBLE, ra, None, ()))
            if ra == rb:
                return self.add_calculation((CALC_SQUARE, ra, None, ()))
            return self.add_calculation((CALC_MUL, ra, rb, ()) if ra <= rb else (CALC_MUL, rb, ra, ()))
        if tag == 'scaled':
            f = e[2] % R_MOD
            if f == 0:
                return zero
            if f == 1:
                return self.add_expression(e[1])
            cst = self.add_constant(f)
            ra = self.add_expression(e[1])
            return self.add_calculation((CALC_MUL, ra, cst, ()))
        raise ValueError(tag)


def custom_gates_graph(gate_polys):
    """Evaluator::new, custom gates (evaluation.rs:225-239)"""
    g = GraphEvaluator()
    parts = tuple(g.add_expression(p) for p in gate_polys)
    g.add_calculation((CALC_HORNER, (VS_PREVIOUS, 0, 0), (VS_Y, 0, 0), parts))
    return g


def lookup_graph(input_exprs, table_exprs):
    """Evaluator::new, one lookup (evaluation.rs:242-275)"""
    g = GraphEvaluator()

    def evaluate_lc(exprs):
        parts = tuple(g.add_expression(e) for e in exprs)
        return g.add_calculation((CALC_HORNER, (VS_CONSTANT, 0, 0), (VS_THETA, 0, 0), parts))

    cin = evaluate_lc(input_exprs)
    ctab = evaluate_lc(table_exprs)
    right_gamma = g.add_calculation((CALC_ADD, ctab, (VS_GAMMA, 0, 0), ()))
    lc = g.add_calculation((CALC_ADD, cin, (VS_BETA, 0, 0), ()))
    g.add_calculation((CALC_MUL, lc, right_gamma, ()))
    return g


# ------------------------------------------------------------------ flat arrays <-> ctypes
def to_mont_limbs(vals):
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        m = (v % R_MOD) * RR % R_MOD
        for j in range(4):
            out[i, j] = (m >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return out


def flatten_graph(g):
    """-> dict of numpy arrays: constants (n,4) u64 Montgomery, rotations i32, calcs (n,10) u32, parts (m,3) u32"""
    calcs, parts = [], []
    for (op, x, y, hp), target in g.calculations:
        x = x or (0, 0, 0)
        y = y or (0, 0, 0)
        calcs.append([op, target, *x, *y, len(parts), len(hp)])
        parts.extend(list(p) for p in hp)
    return {
        "constants": to_mont_limbs(g.constants),
        "rotations": np.array(g.rotations, dtype=np.int32).reshape(-1),
        "calcs": np.array(calcs, dtype=np.uint32).reshape(-1, 10),
        "parts": np.array(parts, dtype=np.uint32).reshape(-1, 3),
        "num_intermediates": g.num_intermediates,
    }


class ValueSource(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_uint32), ("a", ctypes.c_uint32), ("b", ctypes.c_uint32)]


class Calculation(ctypes.Structure):
    _fields_ = [("op", ctypes.c_uint32), ("target", ctypes.c_uint32), ("x", ValueSource), ("y", ValueSource),
                ("parts_offset", ctypes.c_uint32), ("parts_count", ctypes.c_uint32)]


class Graph(ctypes.Structure):
    _fields_ = [("constants", ctypes.c_void_p), ("n_constants", ctypes.c_uint32), ("rotations", ctypes.c_void_p),
                ("n_rotations", ctypes.c_uint32), ("calculations", ctypes.c_void_p), ("n_calculations", ctypes.c_uint32),
                ("parts", ctypes.c_void_p), ("n_parts", ctypes.c_uint32), ("num_intermediates", ctypes.c_uint32)]


class EvalhDesc(ctypes.Structure):
    _fields_ = [
        ("k", ctypes.c_uint32), ("extended_k", ctypes.c_uint32),
        ("extended_omega", ctypes.c_void_p), ("g_coset", ctypes.c_void_p), ("g_coset_inv", ctypes.c_void_p),
        ("n_fixed", ctypes.c_uint32), ("n_advice", ctypes.c_uint32), ("n_instance", ctypes.c_uint32), ("n_challenges", ctypes.c_uint32),
        ("fixed_cosets", ctypes.c_void_p), ("advice_polys", ctypes.c_void_p), ("instance_polys", ctypes.c_void_p),
        ("challenges", ctypes.c_void_p),
        ("y", ctypes.c_void_p), ("beta", ctypes.c_void_p), ("gamma", ctypes.c_void_p), ("theta", ctypes.c_void_p),
        ("l0", ctypes.c_void_p), ("l_last", ctypes.c_void_p), ("l_active_row", ctypes.c_void_p),
        ("custom_gates", Graph),
        ("n_perm_sets", ctypes.c_uint32), ("n_perm_columns", ctypes.c_uint32), ("chunk_len", ctypes.c_uint32),
        ("last_rotation", ctypes.c_int32),
        ("perm_product_cosets", ctypes.c_void_p), ("perm_column_kind", ctypes.c_void_p), ("perm_column_index", ctypes.c_void_p),
        ("perm_cosets", ctypes.c_void_p), ("zeta", ctypes.c_void_p), ("delta", ctypes.c_void_p),
        ("n_lookups", ctypes.c_uint32), ("lookup_graphs", ctypes.c_void_p),
        ("lookup_product_polys", ctypes.c_void_p), ("lookup_permuted_input_polys", ctypes.c_void_p),
        ("lookup_permuted_table_polys", ctypes.c_void_p),
    ]


def _ptr(a):
    return a.ctypes.data if a is not None and a.size else None


class DescHolder:
    """Builds an EvalhDesc from numpy arrays and keeps every buffer alive.

    case: dict with k, extended_k, extended_omega, g_coset, g_coset_inv, zeta, delta, y, beta, gamma, theta (4,) u64;
    fixed_cosets (nf, size, 4), advice_polys (na, n, 4), instance_polys (ni, n, 4), challenges (nc, 4),
    l0, l_last, l_active_row (size, 4); custom (flattened graph dict);
    perm_product_cosets (ns, size, 4), perm_column_kind / perm_column_index u32, perm_cosets (ncols, size, 4), chunk_len, last_rotation;
    lookups: list of (graph dict, product_poly, permuted_input_poly, permuted_table_poly)"""

    def __init__(self, case):
        self.keep = []
        d = EvalhDesc()
        c = lambda a, dt=np.uint64: self._c(a, dt)  # noqa: E731
        d.k, d.extended_k = int(case["k"]), int(case["extended_k"])
        for f in ("extended_omega", "g_coset", "g_coset_inv", "y", "beta", "gamma", "theta", "zeta", "delta"):
            setattr(d, f, _ptr(c(case[f])))
        d.l0, d.l_last, d.l_active_row = _ptr(c(case["l0"])), _ptr(c(case["l_last"])), _ptr(c(case["l_active_row"]))
        d.n_fixed, d.fixed_cosets = self._ptr_array(case["fixed_cosets"])
        d.n_advice, d.advice_polys = self._ptr_array(case["advice_polys"])
        d.n_instance, d.instance_polys = self._ptr_array(case["instance_polys"])
        ch = c(case["challenges"])
        d.n_challenges, d.challenges = ch.shape[0] if ch.ndim == 2 else 0, _ptr(ch)
        d.custom_gates = self._graph(case["custom"])
        d.n_perm_sets, d.perm_product_cosets = self._ptr_array(case["perm_product_cosets"])
        d.n_perm_columns, d.perm_cosets = self._ptr_array(case["perm_cosets"])
        d.perm_column_kind = _ptr(c(case["perm_column_kind"], np.uint32))
        d.perm_column_index = _ptr(c(case["perm_column_index"], np.uint32))
        d.chunk_len, d.last_rotation = int(case["chunk_len"]), int(case["last_rotation"])
        lookups = case["lookups"]
        d.n_lookups = len(lookups)
        graphs = (Graph * max(1, len(lookups)))()
        for i, (g, _, _, _) in enumerate(lookups):
            graphs[i] = self._graph(g)
        self.keep.append(graphs)
        d.lookup_graphs = ctypes.addressof(graphs)
        _, d.lookup_product_polys = self._ptr_array([l[1] for l in lookups])
        _, d.lookup_permuted_input_polys = self._ptr_array([l[2] for l in lookups])
        _, d.lookup_permuted_table_polys = self._ptr_array([l[3] for l in lookups])
        self.desc = d

    def _c(self, a, dt=np.uint64):
        a = np.ascontiguousarray(a, dtype=dt)
        self.keep.append(a)
        return a

    def _ptr_array(self, arrs):
        arrs = [self._c(a) for a in arrs]
        n = len(arrs)
        p = (ctypes.c_void_p * max(1, n))(*[a.ctypes.data for a in arrs])
        self.keep.append(p)
        return n, ctypes.addressof(p)

    def _graph(self, g):
        G = Graph()
        consts, rots = self._c(g["constants"]), self._c(g["rotations"], np.int32)
        calcs, parts = self._c(g["calcs"], np.uint32), self._c(g["parts"], np.uint32)
        G.constants, G.n_constants = _ptr(consts), consts.shape[0]
        G.rotations, G.n_rotations = _ptr(rots), rots.shape[0]
        G.calculations, G.n_calculations = _ptr(calcs), calcs.shape[0]
        G.parts, G.n_parts = _ptr(parts), parts.shape[0]
        G.num_intermediates = int(g["num_intermediates"])
        return G

    def byref(self):
        return ctypes.byref(self.desc)


GraphBuilder = GraphEvaluator  # older name used by the tests


class Evaluator:
    """Evaluator (evaluation.rs:182-189): custom_gates + one GraphEvaluator per lookup."""

    def __init__(self, custom_gates, lookups):
        self.custom_gates, self.lookups = custom_gates, lookups

    @classmethod
    def new(cls, gate_polys, lookup_arguments=()):
        """Evaluator::new (:221-279).  gate_polys: every gate's polynomials in cs.gates order, flattened;
        lookup_arguments: [(input_expressions, table_expressions)]"""
        return cls(custom_gates_graph(list(gate_polys)), [lookup_graph(i, t) for i, t in lookup_arguments])

    def describe(self, case):
        """case as for DescHolder minus `custom` and the graphs inside `lookups` (3-tuples of polynomials there)"""
        full = dict(case)
        full["custom"] = flatten_graph(self.custom_gates)
        full["lookups"] = [(flatten_graph(g), *polys) for g, polys in zip(self.lookups, case["lookups"])]
        return DescHolder(full)

    def evaluate_h(self, case, values):
        """evaluate_h for one circuit instance (:280-522): values (2^extended_k, 4) u64 is folded in place and returned"""
        from . import _check, lib
        h = self.describe(case)
        values = np.ascontiguousarray(values, dtype=np.uint64)
        _check(lib().h2hip_evaluate_h_bn254(h.byref(), values.ctypes.data_as(ctypes.c_void_p)), "evaluate_h")
        return values
