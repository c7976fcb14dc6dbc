#!/usr/bin/python3
"""Golden vectors for Evaluator::evaluate_h (halo2_proofs/src/plonk/evaluation.rs:280-522), minted with Python
big integers straight from the constraint formulas -- the gate Expression tree evaluated directly per row
(Expression::evaluate semantics, evaluation.rs:755-786), the permutation and lookup constraints from the
comments at :382-438 and :484-515 -- not through the flattened graph.  The graph handed to the oracle / GPU
is built by tests/evalh_util.py's port of GraphEvaluator::add_expression, so the test also pins that the
flattening reproduces the direct evaluation.

Constraint system: the one of examples/circuit-layout.rs MyCircuit (:174-240), the BASELINE.json configs[4]
circuit: advice e,a,b,c,d; fixed sf,sm,sa,sb,sc,sl; gate a*sa + b*sb + a*b*sm - c*sc + sf*(d(next)*e(prev));
lookup a in sl; permutation over (a,b,c) => cs.degree() = 4, chunk_len = 2, two permutation sets,
blinding_factors = 5.  Columns are random (evaluate_h is polynomial evaluation; it does not need a satisfying
witness).  Sizes: k = 3 and 4 with extended_k = k + 2.

Run: python tests/golden/make_evalh_golden.py   (writes tests/golden/evalh.npz)
"""
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from evalh_util import R_MOD, custom_gates_graph, flatten_graph, lookup_graph, to_mont_limbs  # noqa: E402

S = 28
ROOT = pow(7, (R_MOD - 1) >> S, R_MOD)
ZETA = pow(7, 2 * (R_MOD - 1) // 3, R_MOD)
DELTA = pow(7, 1 << S, R_MOD)


def eval_expr(e, cols, idx, rot_scale, size, challenges):
    tag = e[0]
    if tag == 'const':
        return e[1] % R_MOD
    if tag in ('fixed', 'advice', 'instance'):
        return cols[tag][e[1]][(idx + e[2] * rot_scale) % size]
    if tag == 'challenge':
        return challenges[e[1]]
    if tag == 'neg':
        return (-eval_expr(e[1], cols, idx, rot_scale, size, challenges)) % R_MOD
    if tag == 'sum':
        return (eval_expr(e[1], cols, idx, rot_scale, size, challenges) + eval_expr(e[2], cols, idx, rot_scale, size, challenges)) % R_MOD
    if tag == 'prod':
        return eval_expr(e[1], cols, idx, rot_scale, size, challenges) * eval_expr(e[2], cols, idx, rot_scale, size, challenges) % R_MOD
    if tag == 'scaled':
        return eval_expr(e[1], cols, idx, rot_scale, size, challenges) * e[2] % R_MOD
    raise ValueError(tag)


def coset_eval(coeffs, ext_omega, size):
    """coeff_to_extended by definition: a(zeta * w^i)"""
    out = []
    for i in range(size):
        x = ZETA * pow(ext_omega, i, R_MOD) % R_MOD
        acc = 0
        for c in reversed(coeffs):
            acc = (acc * x + c) % R_MOD
        out.append(acc)
    return out


def make_case(k, rng, tag, out):
    n = 1 << k
    ek = k + 2
    size = 1 << ek
    rot_scale = 1 << (ek - k)
    ext_omega = pow(ROOT, 1 << (S - ek), R_MOD)
    rnd = lambda: rng.randrange(R_MOD)  # noqa: E731

    # columns
    e_, a_, b_, c_, d_ = 0, 1, 2, 3, 4
    sf, sm, sa, sb, sc, sl = 0, 1, 2, 3, 4, 5
    advice_polys = [[rnd() for _ in range(n)] for _ in range(5)]
    fixed_cosets = [[rnd() for _ in range(size)] for _ in range(6)]
    cols = {'advice': [coset_eval(p, ext_omega, size) for p in advice_polys], 'fixed': fixed_cosets, 'instance': []}
    challenges = []
    y, beta, gamma, theta = rnd(), rnd(), rnd(), rnd()
    l0, l_last, l_active = ([rnd() for _ in range(size)] for _ in range(3))

    A = lambda col, rot=0: ('advice', col, rot)  # noqa: E731
    F = lambda col, rot=0: ('fixed', col, rot)  # noqa: E731
    # a*sa + b*sb + a*b*sm - (c*sc) + sf*(d(next)*e(prev));  `x - y` is Sum(x, Negated(y)) for Expression
    gate = ('sum',
            ('sum', ('sum', ('sum', ('prod', A(a_), F(sa)), ('prod', A(b_), F(sb))), ('prod', ('prod', A(a_), A(b_)), F(sm))),
             ('neg', ('prod', A(c_), F(sc)))),
            ('prod', F(sf), ('prod', A(d_, 1), A(e_, -1))))
    # a second, smaller gate with a constant, a scaling and a doubled term to exercise those graph paths
    gate2 = ('sum', ('scaled', ('prod', A(c_), A(c_)), 7), ('sum', ('prod', ('const', 2), A(d_)), ('neg', ('const', 5))))
    gates = [gate, gate2]
    lookup_inputs, lookup_tables = [A(a_)], [F(sl)]

    # permutation over (a, b, c)
    perm_cols = [(0, a_), (0, b_), (0, c_)]  # (H2HIP_ANY_ADVICE, index)
    chunk_len, last_rotation = 2, -6
    n_sets = 2
    perm_product = [[rnd() for _ in range(size)] for _ in range(n_sets)]
    perm_cosets = [[rnd() for _ in range(size)] for _ in range(3)]
    # lookup polys (coefficient form)
    product_poly, pin_poly, ptab_poly = ([rnd() for _ in range(n)] for _ in range(3))
    product, pin, ptab = (coset_eval(p, ext_omega, size) for p in (product_poly, pin_poly, ptab_poly))

    values_in = [rnd() for _ in range(size)]  # a previous circuit instance's contribution (PreviousValue path)
    values = list(values_in)
    # custom gates: value = value * y + poly, per gate polynomial in order (Horner(PreviousValue, parts, Y))
    for idx in range(size):
        v = values[idx]
        for g in gates:
            v = (v * y + eval_expr(g, cols, idx, rot_scale, size, challenges)) % R_MOD
        values[idx] = v
    # permutation constraints (evaluation.rs:362-441)
    delta_start = beta * ZETA % R_MOD
    for idx in range(size):
        r_next = (idx + rot_scale) % size
        r_last = (idx + last_rotation * rot_scale) % size
        v = values[idx]
        v = (v * y + (1 - perm_product[0][idx]) * l0[idx]) % R_MOD
        zl = perm_product[-1][idx]
        v = (v * y + (zl * zl - zl) * l_last[idx]) % R_MOD
        for s in range(1, n_sets):
            v = (v * y + (perm_product[s][idx] - perm_product[s - 1][r_last]) * l0[idx]) % R_MOD
        current_delta = delta_start * pow(ext_omega, idx, R_MOD) % R_MOD
        for s in range(n_sets):
            chunk = perm_cols[s * chunk_len:(s + 1) * chunk_len]
            cos = perm_cosets[s * chunk_len:(s + 1) * chunk_len]
            left = perm_product[s][r_next]
            for (kind, ci), pc in zip(chunk, cos):
                left = left * (cols['advice'][ci][idx] + beta * pc[idx] + gamma) % R_MOD
            right = perm_product[s][idx]
            for (kind, ci) in chunk:
                right = right * (cols['advice'][ci][idx] + current_delta + gamma) % R_MOD
                current_delta = current_delta * DELTA % R_MOD
            v = (v * y + (left - right) * l_active[idx]) % R_MOD
        values[idx] = v
    # lookup constraints (evaluation.rs:443-518)
    for idx in range(size):
        r_next = (idx + rot_scale) % size
        r_prev = (idx - rot_scale) % size
        cin = 0
        for e in lookup_inputs:
            cin = (cin * theta + eval_expr(e, cols, idx, rot_scale, size, challenges)) % R_MOD
        ctab = 0
        for e in lookup_tables:
            ctab = (ctab * theta + eval_expr(e, cols, idx, rot_scale, size, challenges)) % R_MOD
        table_value = (cin + beta) * (ctab + gamma) % R_MOD
        a_minus_s = (pin[idx] - ptab[idx]) % R_MOD
        v = values[idx]
        v = (v * y + (1 - product[idx]) * l0[idx]) % R_MOD
        v = (v * y + (product[idx] * product[idx] - product[idx]) * l_last[idx]) % R_MOD
        v = (v * y + (product[r_next] * (pin[idx] + beta) * (ptab[idx] + gamma) - product[idx] * table_value) * l_active[idx]) % R_MOD
        v = (v * y + a_minus_s * l0[idx]) % R_MOD
        v = (v * y + a_minus_s * (pin[idx] - pin[r_prev]) * l_active[idx]) % R_MOD
        values[idx] = v

    m = lambda vals: to_mont_limbs(vals)  # noqa: E731
    p = tag + "_"
    out[p + "k"] = np.array([k, ek], dtype=np.uint32)
    for name, val in (("extended_omega", ext_omega), ("g_coset", ZETA), ("g_coset_inv", ZETA * ZETA % R_MOD), ("zeta", ZETA), ("delta", DELTA),
                      ("y", y), ("beta", beta), ("gamma", gamma), ("theta", theta)):
        out[p + name] = m([val])[0]
    out[p + "fixed_cosets"] = np.stack([m(c) for c in fixed_cosets])
    out[p + "advice_polys"] = np.stack([m(c) for c in advice_polys])
    out[p + "l0"], out[p + "l_last"], out[p + "l_active_row"] = m(l0), m(l_last), m(l_active)
    cg = flatten_graph(custom_gates_graph(gates))
    lg = flatten_graph(lookup_graph(lookup_inputs, lookup_tables))
    for gname, g in (("custom", cg), ("lookup0", lg)):
        for key in ("constants", "rotations", "calcs", "parts"):
            out[p + gname + "_" + key] = g[key]
        out[p + gname + "_num_intermediates"] = np.array([g["num_intermediates"]], dtype=np.uint32)
    out[p + "perm_product_cosets"] = np.stack([m(c) for c in perm_product])
    out[p + "perm_cosets"] = np.stack([m(c) for c in perm_cosets])
    out[p + "perm_column_kind"] = np.array([c[0] for c in perm_cols], dtype=np.uint32)
    out[p + "perm_column_index"] = np.array([c[1] for c in perm_cols], dtype=np.uint32)
    out[p + "perm_params"] = np.array([chunk_len, last_rotation], dtype=np.int32)
    out[p + "lookup0_product_poly"], out[p + "lookup0_permuted_input_poly"], out[p + "lookup0_permuted_table_poly"] = m(product_poly), m(pin_poly), m(ptab_poly)
    out[p + "values_in"] = m(values_in)
    out[p + "values_out"] = m(values)


def main():
    rng = random.Random(0xE7A1)
    out = {}
    make_case(3, rng, "k3", out)
    make_case(4, rng, "k4", out)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "evalh.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
