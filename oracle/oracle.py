"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never
from the product package (halo2-pse_amd/).  Arrays are numpy uint64: Fr/Fq elements
(n,4), G1Affine (n,8), G1 Jacobian (12,) -- the RawBytes/Montgomery layout described in
bn254_oracle.c.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    src = os.path.join(_HERE, "bn254_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_init()
        _lib.oracle_inv64.restype = ctypes.c_uint64
        _lib.oracle_window_c.restype = ctypes.c_size_t
        _lib.oracle_window_c.argtypes = [ctypes.c_size_t]
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a


FQ, FR = 0, 1


def constant(which, cid):
    out = np.zeros(4, dtype=np.uint64)
    lib().oracle_constant(which, cid, _p(out))
    return out


def inv64(which):
    return int(lib().oracle_inv64(which))


def fe_binop(name, which, a, b):
    a, b = _c(a), _c(b)
    out = np.zeros_like(a)
    f = getattr(lib(), "oracle_fe_" + name)
    for i in range(a.shape[0]):
        f(which, _p(a[i]), _p(b[i]), _p(out[i]))
    return out


def fe_to_canonical(which, a):
    a = _c(a)
    out = np.zeros_like(a)
    for i in range(a.shape[0]):
        lib().oracle_fe_to_canonical(which, _p(a[i]), _p(out[i]))
    return out


def fe_from_canonical(which, a):
    a = _c(a)
    out = np.zeros_like(a)
    for i in range(a.shape[0]):
        lib().oracle_fe_from_canonical(which, _p(a[i]), _p(out[i]))
    return out


def fe_inv(which, a):
    a = _c(a)
    out = np.zeros(4, dtype=np.uint64)
    lib().oracle_fe_inv(which, _p(a), _p(out))
    return out


def fe_from_int(which, v):
    mod = int_from_limbs(constant(which, 2))
    c = np.array([((v % mod) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
    return fe_from_canonical(which, c.reshape(1, 4))[0]


def int_from_limbs(l):
    return sum(int(x) << (64 * i) for i, x in enumerate(l))


def g1_to_affine(xyz):
    xyz = _c(xyz)
    out = np.zeros(8, dtype=np.uint64)
    lib().oracle_g1_to_affine(_p(xyz), _p(out))
    return out


def g1_add(p, q):
    p, q = _c(p), _c(q)
    out = np.zeros(12, dtype=np.uint64)
    lib().oracle_g1_add(_p(p), _p(q), _p(out))
    return out


def g1_mul(p_affine, scalar_mont):
    out = np.zeros(12, dtype=np.uint64)
    lib().oracle_g1_mul(_p(_c(p_affine)), _p(_c(scalar_mont)), _p(out))
    return out


def g1_on_curve(p_affine):
    return bool(lib().oracle_g1_on_curve(_p(_c(p_affine))))


def window_c(m):
    return int(lib().oracle_window_c(m))


def best_multiexp(scalars, bases, num_threads=1):
    """arithmetic.rs:132-159; returns Jacobian (12,) uint64"""
    scalars, bases = _c(scalars), _c(bases)
    assert scalars.shape[0] == bases.shape[0]  # assert_eq! arithmetic.rs:133
    out = np.zeros(12, dtype=np.uint64)
    lib().oracle_best_multiexp(_p(scalars), _p(bases), ctypes.c_size_t(scalars.shape[0]), int(num_threads), _p(out))
    return out


def naive_multiexp(scalars, bases):
    scalars, bases = _c(scalars), _c(bases)
    out = np.zeros(12, dtype=np.uint64)
    lib().oracle_naive_multiexp(_p(scalars), _p(bases), ctypes.c_size_t(scalars.shape[0]), _p(out))
    return out


def best_fft(a, omega, log_n, num_threads=1):
    """arithmetic.rs:171-234; returns a new array"""
    a = _c(a).copy()
    assert a.shape[0] == 1 << log_n  # assert_eq! arithmetic.rs:184
    lib().oracle_best_fft(_p(a), _p(_c(omega)), ctypes.c_uint32(log_n), int(num_threads))
    return a


def ifft(a, omega_inv, log_n, divisor, num_threads=1):
    a = _c(a).copy()
    lib().oracle_ifft(_p(a), _p(_c(omega_inv)), ctypes.c_uint32(log_n), _p(_c(divisor)), int(num_threads))
    return a


class Domain(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_uint64),
        ("k", ctypes.c_uint32),
        ("extended_k", ctypes.c_uint32),
        ("t_len", ctypes.c_uint32),
        ("quotient_poly_degree", ctypes.c_uint64),
    ] + [(nm, ctypes.c_uint64 * 4) for nm in (
        "omega", "omega_inv", "extended_omega", "extended_omega_inv", "g_coset", "g_coset_inv",
        "ifft_divisor", "extended_ifft_divisor", "barycentric_weight")]

    def fe(self, name):
        return np.array(list(getattr(self, name)), dtype=np.uint64)


def domain_new(j, k):
    """EvaluationDomain::new (poly/domain.rs:39-142) -> (Domain, t_evaluations)"""
    d = Domain()
    ek = k
    while (1 << ek) < (1 << k) * (j - 1):
        ek += 1
    t = np.zeros((1 << (ek - k), 4), dtype=np.uint64)
    rc = lib().oracle_domain_new(ctypes.c_uint32(j), ctypes.c_uint32(k), ctypes.byref(d), _p(t))
    if rc != 0:
        raise ValueError("oracle_domain_new rc=%d" % rc)
    return d, t


def lagrange_to_coeff(d, a, num_threads=1):
    a = _c(a).copy()
    lib().oracle_lagrange_to_coeff(ctypes.byref(d), _p(a), int(num_threads))
    return a


def coeff_to_extended(d, a, num_threads=1):
    a = _c(a)
    out = np.zeros((1 << d.extended_k, 4), dtype=np.uint64)
    lib().oracle_coeff_to_extended(ctypes.byref(d), _p(a), _p(out), int(num_threads))
    return out


def extended_to_coeff(d, a, num_threads=1):
    a = _c(a).copy()
    lib().oracle_extended_to_coeff(ctypes.byref(d), _p(a), int(num_threads))
    return a[: d.n * d.quotient_poly_degree]


def divide_by_vanishing_poly(d, t_eval, a):
    a = _c(a).copy()
    lib().oracle_divide_by_vanishing_poly(ctypes.byref(d), _p(_c(t_eval)), _p(a))
    return a


def kzg_setup(k, secret_mont):
    n = 1 << k
    g = np.zeros((n, 8), dtype=np.uint64)
    gl = np.zeros((n, 8), dtype=np.uint64)
    rc = lib().oracle_kzg_setup(ctypes.c_uint32(k), _p(_c(secret_mont)), _p(g), _p(gl))
    if rc != 0:
        raise ValueError("oracle_kzg_setup rc=%d" % rc)
    return g, gl


def best_fft_g1(a_xyz, omega, log_n, num_threads=1):
    """arithmetic.rs:171-234 with G = bn256::G1: (n, 12) Jacobian points -> (n, 12) Jacobian points (a copy is transformed)"""
    a = _c(a_xyz).copy()
    assert a.shape == (1 << log_n, 12)
    rc = lib().oracle_best_fft_g1(_p(a), _p(_c(omega)), ctypes.c_uint32(log_n), int(num_threads))
    if rc != 0:
        raise ValueError("oracle_best_fft_g1 rc=%d" % rc)
    return a


def g_to_lagrange(g, k, num_threads=1):
    """arithmetic.rs:277-301: (n, 8) affine coefficient-basis points -> (n, 8) affine Lagrange-basis points"""
    g = _c(g)
    assert g.shape[0] == 1 << k
    out = np.zeros((1 << k, 8), dtype=np.uint64)
    rc = lib().oracle_g_to_lagrange(_p(g), ctypes.c_uint32(k), _p(out), int(num_threads))
    if rc != 0:
        raise ValueError("oracle_g_to_lagrange rc=%d" % rc)
    return out


def gen_scalars(seed, n, start=0, num_threads=1):
    out = np.zeros((n, 4), dtype=np.uint64)
    if start == 0 and num_threads > 1:
        lib().oracle_gen_parallel(0, ctypes.c_uint64(seed), ctypes.c_size_t(n), _p(out), int(num_threads))
    else:
        lib().oracle_gen_scalars(ctypes.c_uint64(seed), ctypes.c_size_t(start), ctypes.c_size_t(n), _p(out))
    return out


def gen_points(seed, n, start=0, num_threads=1):
    out = np.zeros((n, 8), dtype=np.uint64)
    if start == 0 and num_threads > 1:
        lib().oracle_gen_parallel(1, ctypes.c_uint64(seed), ctypes.c_size_t(n), _p(out), int(num_threads))
    else:
        lib().oracle_gen_points(ctypes.c_uint64(seed), ctypes.c_size_t(start), ctypes.c_size_t(n), _p(out))
    return out
