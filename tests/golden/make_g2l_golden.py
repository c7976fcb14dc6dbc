#!/usr/bin/python3
"""Golden vectors for g_to_lagrange (halo2_proofs/src/arithmetic.rs:277-301), minted from the definition with Python
integers and naive affine curve arithmetic (no FFT): g_lagrange[i] = [1/n] * sum_j [omega^(-i*j)] g[j], where omega is
the 2^k-th root of unity EvaluationDomain uses.  Inputs: random curve points, with the identity and a repeated point
mixed in for k >= 2 (a truncated / degenerate SRS must not break the butterflies' exceptional cases).

Run: python tests/golden/make_g2l_golden.py   (writes tests/golden/g2l.npz)
"""
import os
import random

import numpy as np

from make_golden import R_MOD, ec_add, ec_mul, omega_for, pt_arr, rand_point


def g_to_lagrange_def(g, k):
    n = 1 << k
    w_inv = pow(omega_for(k), -1, R_MOD)
    n_inv = pow(n, -1, R_MOD)
    out = []
    for i in range(n):
        acc = None
        for j in range(n):
            acc = ec_add(acc, ec_mul(pow(w_inv, i * j, R_MOD), g[j]))
        out.append(ec_mul(n_inv, acc))
    return out


def main():
    rng = random.Random(0x6217)
    out = {}
    for k in range(0, 6):
        n = 1 << k
        g = [rand_point(rng) for _ in range(n)]
        if k >= 2:
            g[1] = None          # identity (0, 0)
            g[n - 1] = g[0]      # a repeated point: the first layer adds and subtracts equal points
        out["g2l_k%d_g" % k] = pt_arr(g)
        out["g2l_k%d_out" % k] = pt_arr(g_to_lagrange_def(g, k))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "g2l.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
