#!/usr/bin/python3
"""Mint golden vectors for the BN254 MSM / Fr-NTT hot path with Python big integers.

Independent of oracle/bn254_oracle.c and of the HIP code: everything here is plain
`int` arithmetic -- naive per-term affine double-and-add for the MSM, the O(n^2)
definition sum_j a_j w^(ij) for the NTT, textbook formulas for the KZG SRS.  The
reference itself (Rust) cannot be run in this pipeline (no cargo/rustc; halo2curves
0.3.1 un-vendored), and its tests hold no BN254 byte vectors, so these fixtures are
what pins the oracle ("parity unpinned" w.r.t. reference-produced bytes; see
DESIGN.md).  What each fixture restates (paths under /root/reference/halo2_proofs/src):

  msm_*      value of best_multiexp(coeffs, bases)            arithmetic.rs:132-159
  ntt_*      value of best_fft(a, omega, log_n)               arithmetic.rs:171-234
  ifft_*     EvaluationDomain::ifft                           poly/domain.rs:353-361
  domain_*   EvaluationDomain::new constants                  poly/domain.rs:39-142
  ext_*      coeff_to_extended / extended_to_coeff            poly/domain.rs:240-303
  kzg_*      ParamsKZG::setup with a fixed secret, commit     poly/kzg/commitment.rs:61-129,281-334
  gen_*      the synthetic-input generator of SURVEY.md 8(d)

Layout: field elements are 4 x u64 little-endian limbs in Montgomery form (R = 2^256),
G1Affine = x||y with identity (0,0) -- halo2curves' RawBytes layout (helpers.rs:13-19).

Run:  python tests/golden/make_golden.py     (writes tests/golden/golden.npz, ~1 MB)
"""
import os
import random
import numpy as np

Q = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
R_MOD = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
RR = 1 << 256
S = 28
GEN7 = 7


def to_limbs(v):
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def mont(v, p):
    return (v * RR) % p


def fe_arr(vals, p):
    """canonical ints -> (n,4) uint64 Montgomery limbs"""
    return np.array([to_limbs(mont(v, p)) for v in vals], dtype=np.uint64).reshape(len(vals), 4)


def pt_arr(pts):
    """affine points (None = identity) -> (n,8) uint64"""
    out = []
    for P in pts:
        if P is None:
            out.append([0] * 8)
        else:
            out.append(to_limbs(mont(P[0], Q)) + to_limbs(mont(P[1], Q)))
    return np.array(out, dtype=np.uint64).reshape(len(pts), 8)


# ---------------------------------------------------------------- curve (affine, naive)
def ec_add(P, T):
    if P is None:
        return T
    if T is None:
        return P
    x1, y1 = P
    x2, y2 = T
    if x1 == x2:
        if (y1 + y2) % Q == 0:
            return None
        lam = (3 * x1 * x1) * pow(2 * y1, -1, Q) % Q
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, Q) % Q
    x3 = (lam * lam - x1 - x2) % Q
    y3 = (lam * (x1 - x3) - y1) % Q
    return (x3, y3)


def ec_neg(P):
    return None if P is None else (P[0], (-P[1]) % Q)


def ec_mul(k, P):
    acc = None
    for bit in bin(k)[2:] if k else "":
        acc = ec_add(acc, acc)
        if bit == "1":
            acc = ec_add(acc, P)
    return acc


def naive_msm(scalars, points):
    acc = None
    for s, P in zip(scalars, points):
        acc = ec_add(acc, ec_mul(s % R_MOD, P))
    return acc


def rand_point(rng):
    while True:
        x = rng.randrange(Q)
        rhs = (x * x * x + 3) % Q
        y = pow(rhs, (Q + 1) // 4, Q)
        if y * y % Q == rhs:
            if rng.getrandbits(1):
                y = (-y) % Q
            return (x, y)


# ---------------------------------------------------------------- NTT (definition)
def root_of_unity():
    return pow(GEN7, (R_MOD - 1) >> S, R_MOD)


def omega_for(k):
    return pow(root_of_unity(), 1 << (S - k), R_MOD)


def naive_dft(a, w):
    n = len(a)
    pw = [pow(w, i, R_MOD) for i in range(n)]
    return [sum(a[j] * pw[(i * j) % n] for j in range(n)) % R_MOD for i in range(n)]


# ---------------------------------------------------------------- splitmix generator
M64 = (1 << 64) - 1
ATT = 0x632BE59BD9B4E019


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M64
    return x ^ (x >> 31)


def draw_mod(seed, i, attempt, p):
    limbs = [splitmix64((seed + 4 * i + j + attempt * ATT) & M64) for j in range(4)]
    raw3 = limbs[3]
    limbs[3] &= 0x3FFFFFFFFFFFFFFF
    v = sum(l << (64 * k) for k, l in enumerate(limbs))
    if v >= p:
        v -= p
    return v, raw3


def gen_scalar(seed, i):
    return draw_mod(seed, i, 0, R_MOD)[0]


def gen_point(seed, i):
    attempt = 0
    while True:
        x, raw3 = draw_mod(seed, i, attempt, Q)
        rhs = (x * x * x + 3) % Q
        y = pow(rhs, (Q + 1) // 4, Q)
        if y * y % Q == rhs:
            if (raw3 >> 62) & 1:
                y = (-y) % Q
            return (x, y)
        attempt += 1


# ---------------------------------------------------------------- main
def main():
    out = {}
    rng = random.Random(0xB254)

    # constants (recomputed)
    for name, p in (("fq", Q), ("fr", R_MOD)):
        out[f"const_{name}_modulus"] = np.array(to_limbs(p), dtype=np.uint64)
        out[f"const_{name}_R"] = np.array(to_limbs(RR % p), dtype=np.uint64)
        out[f"const_{name}_R2"] = np.array(to_limbs(RR * RR % p), dtype=np.uint64)
        out[f"const_{name}_inv64"] = np.array([(-pow(p, -1, 1 << 64)) % (1 << 64)], dtype=np.uint64)
    rou = root_of_unity()
    assert pow(rou, 1 << S, R_MOD) == 1 and pow(rou, 1 << (S - 1), R_MOD) == R_MOD - 1
    zeta = pow(GEN7, 2 * (R_MOD - 1) // 3, R_MOD)
    assert pow(zeta, 3, R_MOD) == 1 and zeta != 1
    out["const_fr_root_of_unity"] = fe_arr([rou], R_MOD)[0]
    out["const_fr_root_of_unity_inv"] = fe_arr([pow(rou, -1, R_MOD)], R_MOD)[0]
    out["const_fr_zeta"] = fe_arr([zeta], R_MOD)[0]
    # Appendix A literal values (canonical) that the recomputation must reproduce
    assert rou == 0x03ddb9f5166d18b798865ea93dd31f743215cf6dd39329c8d34f1ed960c37c9c
    assert zeta == 0x30644e72e131a029048b6e193fd84104cc37a73fec2bc5e9b8ca0b2d36636f23
    assert RR % Q == 0x0e0a77c19a07df2f666ea36f7879462c0a78eb28f5c70b3dd35d438dc58f0d9d
    assert RR % R_MOD == 0x0e0a77c19a07df2f666ea36f7879462e36fc76959f60cd29ac96341c4ffffffb

    # field products: Montgomery limbs of a, b and of a*b
    for name, p in (("fq", Q), ("fr", R_MOD)):
        a = [0, 1, p - 1, 2, p - 2] + [rng.randrange(p) for _ in range(27)]
        b = [p - 1, p - 1, p - 1, (p + 1) // 2, 3] + [rng.randrange(p) for _ in range(27)]
        out[f"field_{name}_a"] = fe_arr(a, p)
        out[f"field_{name}_b"] = fe_arr(b, p)
        out[f"field_{name}_mul"] = fe_arr([x * y % p for x, y in zip(a, b)], p)
        out[f"field_{name}_add"] = fe_arr([(x + y) % p for x, y in zip(a, b)], p)
        out[f"field_{name}_sub"] = fe_arr([(x - y) % p for x, y in zip(a, b)], p)
        out[f"field_{name}_canon"] = np.array([to_limbs(x) for x in a], dtype=np.uint64)

    # MSM cases
    G = (1, 2)
    assert ec_mul(R_MOD, G) is None  # [r]G = identity
    sizes = [1, 2, 3, 4, 31, 32, 33, 100, 1024]
    for n in sizes:
        pts = [rand_point(rng) for _ in range(n)]
        sc = [rng.randrange(R_MOD) for _ in range(n)]
        # sprinkle edge cases (SURVEY.md App. B rule 5)
        if n >= 31:
            sc[0] = 0
            sc[1] = 1
            sc[2] = R_MOD - 1
            sc[3] = (1 << 253) + 12345
            pts[4] = None                 # identity base
            pts[6] = pts[5]               # repeated base ...
            sc[6] = sc[5]                 # ... same digits -> same bucket (doubling case)
            pts[8] = ec_neg(pts[7])       # base and its negation ...
            sc[8] = sc[7]                 # ... in the same buckets (-> identity)
            pts[9] = G
            sc[10] = 2
            sc[11] = 3
        out[f"msm_{n}_scalars"] = fe_arr(sc, R_MOD)
        out[f"msm_{n}_bases"] = pt_arr(pts)
        out[f"msm_{n}_result"] = pt_arr([naive_msm(sc, pts)])[0]
    # degenerate inputs
    n = 64
    pts = [rand_point(rng) for _ in range(n)]
    for tag, sc in (
        ("zeros", [0] * n),
        ("ones", [1] * n),
        ("rm1", [R_MOD - 1] * n),
        ("single", [0] * 17 + [rng.randrange(R_MOD)] + [0] * (n - 18)),
        ("sparse", [rng.choice([0] * 18 + [1, 2]) if rng.random() < 0.95 else rng.randrange(R_MOD) for _ in range(n)]),
    ):
        out[f"msm_{tag}_scalars"] = fe_arr(sc, R_MOD)
        out[f"msm_{tag}_bases"] = pt_arr(pts)
        out[f"msm_{tag}_result"] = pt_arr([naive_msm(sc, pts)])[0]
    # all bases equal + cancelling pair summing to identity
    P = rand_point(rng)
    out["msm_cancel_scalars"] = fe_arr([5, 5], R_MOD)
    out["msm_cancel_bases"] = pt_arr([P, ec_neg(P)])
    out["msm_cancel_result"] = pt_arr([None])[0]

    # NTT cases: forward with omega, inverse with omega^-1 and 2^-k
    for k in range(0, 11):
        n = 1 << k
        w = omega_for(k)
        a = [rng.randrange(R_MOD) for _ in range(n)]
        if n >= 4:
            a[0] = 0
            a[1] = R_MOD - 1
        fwd = naive_dft(a, w)
        winv = pow(w, -1, R_MOD)
        ninv = pow(n, -1, R_MOD)
        inv = [x * ninv % R_MOD for x in naive_dft(a, winv)]
        out[f"ntt_{k}_in"] = fe_arr(a, R_MOD)
        out[f"ntt_{k}_omega"] = fe_arr([w], R_MOD)[0]
        out[f"ntt_{k}_out"] = fe_arr(fwd, R_MOD)
        out[f"ifft_{k}_omega_inv"] = fe_arr([winv], R_MOD)[0]
        out[f"ifft_{k}_divisor"] = fe_arr([ninv], R_MOD)[0]
        out[f"ifft_{k}_out"] = fe_arr(inv, R_MOD)

    # EvaluationDomain::new(j, k) constants + coset round trip, for (j,k) = (4,5) -> extended_k 7
    # (the k: 5, extended_k: 7 pair pinned in tests/plonk_api.rs:629-632) and (3,4) -> extended_k 5
    for (j, k) in ((4, 5), (3, 4), (2, 3)):
        n = 1 << k
        qd = j - 1
        ek = k
        while (1 << ek) < n * qd:
            ek += 1
        eo = omega_for(ek)
        o = pow(eo, 1 << (ek - k), R_MOD)
        assert o == omega_for(k)
        tl = 1 << (ek - k)
        t_eval = [(pow(zeta, n, R_MOD) * pow(eo, n * i, R_MOD) - 1) % R_MOD for i in range(tl)]
        t_eval_inv = [pow(t, -1, R_MOD) for t in t_eval]
        tag = f"domain_{j}_{k}"
        out[tag + "_extended_k"] = np.array([ek], dtype=np.uint64)
        out[tag + "_omega"] = fe_arr([o], R_MOD)[0]
        out[tag + "_omega_inv"] = fe_arr([pow(o, -1, R_MOD)], R_MOD)[0]
        out[tag + "_extended_omega"] = fe_arr([eo], R_MOD)[0]
        out[tag + "_extended_omega_inv"] = fe_arr([pow(eo, -1, R_MOD)], R_MOD)[0]
        out[tag + "_g_coset"] = fe_arr([zeta], R_MOD)[0]
        out[tag + "_g_coset_inv"] = fe_arr([zeta * zeta % R_MOD], R_MOD)[0]
        out[tag + "_ifft_divisor"] = fe_arr([pow(n, -1, R_MOD)], R_MOD)[0]
        out[tag + "_extended_ifft_divisor"] = fe_arr([pow(1 << ek, -1, R_MOD)], R_MOD)[0]
        out[tag + "_barycentric_weight"] = fe_arr([pow(n, -1, R_MOD)], R_MOD)[0]
        out[tag + "_t_evaluations"] = fe_arr(t_eval_inv, R_MOD)
        # coeff_to_extended: evaluations of a(X) over the coset zeta*<extended_omega>
        a = [rng.randrange(R_MOD) for _ in range(n)]
        en = 1 << ek
        ext = [sum(c * pow(zeta * pow(eo, i, R_MOD) % R_MOD, d, R_MOD) for d, c in enumerate(a)) % R_MOD for i in range(en)]
        out[f"ext_{j}_{k}_coeffs"] = fe_arr(a, R_MOD)
        out[f"ext_{j}_{k}_extended"] = fe_arr(ext, R_MOD)
        # extended_to_coeff of the evaluations of a degree < n*qd polynomial returns its coefficients
        h = [rng.randrange(R_MOD) for _ in range(n * qd)]
        hext = [sum(c * pow(zeta * pow(eo, i, R_MOD) % R_MOD, d, R_MOD) for d, c in enumerate(h)) % R_MOD for i in range(en)]
        out[f"ext_{j}_{k}_h_coeffs"] = fe_arr(h, R_MOD)
        out[f"ext_{j}_{k}_h_extended"] = fe_arr(hext, R_MOD)
        out[f"ext_{j}_{k}_h_divided"] = fe_arr([v * t_eval_inv[i % tl] % R_MOD for i, v in enumerate(hext)], R_MOD)

    # KZG SRS with a fixed secret (poly/kzg/commitment.rs:61-129) + test_commit_lagrange inputs (:361-384)
    s = 0x1234567890abcdef1122334455667788 % R_MOD
    for k in (3, 6):
        n = 1 << k
        w = omega_for(k)
        g = [ec_mul(pow(s, i, R_MOD), G) for i in range(n)]
        mult = (pow(s, n, R_MOD) - 1) * pow(n, -1, R_MOD) % R_MOD
        gl = []
        for i in range(n):
            rp = pow(w, i, R_MOD)
            scalar = mult * rp % R_MOD * pow(s - rp, -1, R_MOD) % R_MOD
            gl.append(ec_mul(scalar, G))
        a = list(range(n))  # a[i] = i as in test_commit_lagrange
        out[f"kzg_{k}_secret"] = fe_arr([s], R_MOD)[0]
        out[f"kzg_{k}_g"] = pt_arr(g)
        out[f"kzg_{k}_g_lagrange"] = pt_arr(gl)
        out[f"kzg_{k}_poly_lagrange"] = fe_arr(a, R_MOD)
        out[f"kzg_{k}_commit_lagrange"] = pt_arr([naive_msm(a, gl)])[0]
        # commitment = [a(s)]G where a is the interpolant: must equal commit(iNTT(a))
        winv = pow(w, -1, R_MOD)
        ninv = pow(n, -1, R_MOD)
        coeffs = [x * ninv % R_MOD for x in naive_dft(a, winv)]
        a_at_s = sum(c * pow(s, d, R_MOD) for d, c in enumerate(coeffs)) % R_MOD
        assert ec_mul(a_at_s, G) == naive_msm(a, gl) == naive_msm(coeffs, g)
        out[f"kzg_{k}_poly_coeff"] = fe_arr(coeffs, R_MOD)

    # synthetic generator (seeds of SURVEY.md 8(d))
    out["gen_scalars_5EED0001"] = fe_arr([gen_scalar(0x5EED0001, i) for i in range(64)], R_MOD)
    out["gen_points_5EED0002"] = pt_arr([gen_point(0x5EED0002, i) for i in range(64)])
    out["gen_ntt_5EED0003"] = fe_arr([gen_scalar(0x5EED0003, i) for i in range(64)], R_MOD)
    out["gen_scalars_offset1000"] = fe_arr([gen_scalar(0x5EED0001, 1000 + i) for i in range(8)], R_MOD)
    out["gen_points_offset1000"] = pt_arr([gen_point(0x5EED0002, 1000 + i) for i in range(8)])

    # the k = 6 SRS in the reference's on-disk format, ParamsKZG::write = write_custom(RawBytes)
    # (poly/kzg/commitment.rs:142-157): k u32 LE | g | g_lagrange | g2 | s_g2.  The G2 points are not on
    # this path (pairing stays on the CPU) and are written as zeros.
    import struct
    raw = struct.pack("<I", 6) + out["kzg_6_g"].tobytes() + out["kzg_6_g_lagrange"].tobytes() + bytes(256)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kzg_6_params.rawbytes"), "wb") as f:
        f.write(raw)

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
