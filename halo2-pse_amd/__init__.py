"""halo2_pse_amd -- thin ctypes plumbing over libhalo2hip.so (include/halo2hip.h).

The product is the C-ABI library (HIP kernels for gfx950 + C++ host code); the C++ mirror of
the reference's Rust interface lives in host/halo2hip.hpp.  This module only exists so that
tests/, bench.py and __graft_entry__.py can drive the C ABI from Python; names follow the
reference (halo2_proofs::arithmetic::{best_multiexp, best_fft}, poly::EvaluationDomain,
poly::kzg::ParamsKZG).  There is no CPU fallback here: if the library or the GPU is
missing, calls raise.

Arrays: numpy uint64 -- Fr elements (n,4), G1Affine (n,8), G1 Jacobian (12,): 4 x u64 LE limbs,
Montgomery form (halo2curves' in-memory / RawBytes layout).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HALO2_HIP_LIB") or os.path.join(_HERE, "libhalo2hip.so")  # HALO2_HIP_LIB: another build of the library (A/B measurements)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "halo2hip.h")

_lib = None


class H2HipError(RuntimeError):
    pass


def build(force=False, jobs=4):
    """compile libhalo2hip.so in-tree (hipcc --offload-arch=gfx950)"""
    args = ["make", "-C", _HERE, "-j%d" % jobs]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise H2HipError("libhalo2hip.so is not built (run __graft_entry__.build()); no CPU fallback exists")
        # PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64; two HIP
        # runtimes in one process cannot both open the GPU.  Import torch first so that this
        # library's NEEDED libamdhip64.so.7 binds to the runtime torch already loaded (same
        # SONAME).  A pure C++/Rust consumer gets /opt/rocm's runtime through the RUNPATH.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        L.h2hip_last_error.restype = ctypes.c_char_p
        L.h2hip_version.restype = ctypes.c_char_p
        L.h2hip_get_msm_window.restype = ctypes.c_uint32
        L.h2hip_get_msm_window.argtypes = [ctypes.c_size_t]
        L.h2hip_get_msm_window_fixed_base.restype = ctypes.c_uint32
        L.h2hip_get_msm_window_fixed_base.argtypes = [ctypes.c_size_t]
        L.h2hip_msm_min_n.restype = ctypes.c_size_t
        L.h2hip_ntt_min_log_n.restype = ctypes.c_uint32
        L.h2hip_lazy_pin_after.restype = ctypes.c_uint32
        # release streams, workspaces and worker threads while the HIP runtime is still alive (a profiler's own
        # finalisation otherwise meets them in the static destructors at process exit)
        import atexit
        atexit.register(L.h2hip_shutdown)
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise H2HipError("%s failed (rc=%d): %s" % (what, rc, lib().h2hip_last_error().decode()))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _u64(a, cols=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if cols is not None and (a.ndim != 2 or a.shape[1] != cols):
        raise ValueError("expected shape (n,%d), got %s" % (cols, a.shape))
    return a


def _fe(a):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(4)
    return a


def init(device=None):
    """device: None (HALO2_HIP_DEVICES or the current HIP device), one ordinal, or a list of ordinals (multi-GPU MSM)"""
    if device is None:
        _check(lib().h2hip_init(None, 0), "h2hip_init")
    else:
        devs = [int(d) for d in device] if isinstance(device, (list, tuple)) else [int(device)]
        ids = (ctypes.c_int * len(devs))(*devs)
        _check(lib().h2hip_init(ids, len(devs)), "h2hip_init")


def num_devices():
    return int(lib().h2hip_num_devices())


def shutdown():
    lib().h2hip_shutdown()


def device_count():
    return int(lib().h2hip_device_count())


def version():
    return lib().h2hip_version().decode()


# ------------------------------------------------------------------ arithmetic.rs
def best_multiexp(coeffs, bases):
    """halo2_proofs::arithmetic::best_multiexp (arithmetic.rs:132-159) for C = bn256::G1Affine.
    Returns the Jacobian result (12,) uint64."""
    coeffs, bases = _u64(coeffs, 4), _u64(bases, 8)
    assert coeffs.shape[0] == bases.shape[0]  # assert_eq!(coeffs.len(), bases.len()), arithmetic.rs:133
    out = np.zeros(12, dtype=np.uint64)
    _check(lib().h2hip_msm_bn254(_p(coeffs), _p(bases), ctypes.c_size_t(coeffs.shape[0]), _p(out)), "h2hip_msm_bn254")
    return out


def best_multiexp_batch(coeffs_list, bases):
    """`len(coeffs_list)` MSMs over the same bases (the back-to-back commits of plonk/prover.rs:361-365);
    returns (count, 12) uint64 Jacobian results."""
    bases = _u64(bases, 8)
    cols = [_u64(c, 4) for c in coeffs_list]
    n = bases.shape[0]
    for c in cols:
        assert c.shape[0] == n  # arithmetic.rs:133
    count = len(cols)
    ptrs = (ctypes.c_void_p * count)(*[c.ctypes.data for c in cols])
    out = np.zeros((count, 12), dtype=np.uint64)
    _check(lib().h2hip_msm_bn254_batch(ptrs, _p(bases), ctypes.c_size_t(n), ctypes.c_size_t(count), _p(out)), "h2hip_msm_bn254_batch")
    return out


def best_fft(a, omega, log_n):
    """halo2_proofs::arithmetic::best_fft (arithmetic.rs:171-234) for G = bn256::Fr; in place on `a`."""
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    assert a.shape == (1 << log_n, 4)  # assert_eq!(n, 1 << log_n), arithmetic.rs:184
    _check(lib().h2hip_ntt_bn254_fr(_p(a), _p(_fe(omega)), ctypes.c_uint32(log_n)), "h2hip_ntt_bn254_fr")


def _host_ptrs(arrays):
    return (ctypes.c_void_p * len(arrays))(*[a.ctypes.data for a in arrays])


def best_fft_batch(columns, omega, log_n):
    """`len(columns)` in-place best_fft calls on host columns as one pipelined call (upload i + 1 | transform i | download i - 1)"""
    for a in columns:
        assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"] and a.shape == (1 << log_n, 4)
    if columns:
        _check(lib().h2hip_ntt_bn254_fr_batch(_host_ptrs(columns), ctypes.c_size_t(len(columns)), _p(_fe(omega)), ctypes.c_uint32(log_n)),
               "h2hip_ntt_bn254_fr_batch")


def g_to_lagrange(g, k):
    """arithmetic::g_to_lagrange (arithmetic.rs:277-301): (2^k, 8) affine points -> (2^k, 8) affine points"""
    g = _u64(g, 8)
    assert g.shape[0] == 1 << k
    out = np.zeros((1 << k, 8), dtype=np.uint64)
    _check(lib().h2hip_g_to_lagrange_bn254(_p(g), ctypes.c_uint32(k), _p(out)), "h2hip_g_to_lagrange_bn254")
    return out


def best_fft_g1(a_xyz, omega, log_n):
    """halo2_proofs::arithmetic::best_fft for G = bn256::G1 (arithmetic.rs:171-234): in place on (2^log_n, 12) Jacobian points"""
    assert a_xyz.dtype == np.uint64 and a_xyz.flags["C_CONTIGUOUS"] and a_xyz.shape == (1 << log_n, 12)
    _check(lib().h2hip_fft_bn254_g1(_p(a_xyz), _p(_fe(omega)), ctypes.c_uint32(log_n)), "h2hip_fft_bn254_g1")


def g1_to_affine(xyz):
    xyz = np.ascontiguousarray(xyz, dtype=np.uint64).reshape(12)
    out = np.zeros(8, dtype=np.uint64)
    _check(lib().h2hip_g1_to_affine(_p(xyz), _p(out)), "h2hip_g1_to_affine")
    return out


def g1_fold(partials):
    partials = np.ascontiguousarray(partials, dtype=np.uint64).reshape(-1, 12)
    out = np.zeros(12, dtype=np.uint64)
    _check(lib().h2hip_g1_fold(_p(partials), ctypes.c_size_t(partials.shape[0]), _p(out)), "h2hip_g1_fold")
    return out


def bases_pin(bases):
    assert bases.dtype == np.uint64 and bases.flags["C_CONTIGUOUS"]
    _check(lib().h2hip_bases_pin(_p(bases), ctypes.c_size_t(bases.shape[0])), "h2hip_bases_pin")


def bases_unpin(bases):
    _check(lib().h2hip_bases_unpin(_p(bases)), "h2hip_bases_unpin")


_device_pins = {}  # device address -> weakref.finalize on the STORAGE of the tensor that was pinned


def _unpin_address(addr):
    _device_pins.pop(addr, None)
    try:
        lib().h2hip_bases_unpin(ctypes.c_void_p(addr))
    except Exception:
        pass


def bases_pin_device(d_bases, n=None):
    """pin device-resident points (torch CUDA tensor): copies them and builds the fixed-base window table.  The buffer's
    address is the cache key, so the entry is tied to the lifetime of the tensor's STORAGE (not of the tensor object: pinning
    through a view or a temporary -- `t.view(-1)`, `t[:n]` -- must not unpin while `t` is alive): when the storage is freed
    without bases_unpin_device the entry goes with it (the library's own fingerprint check is the second line of defence)."""
    import weakref
    n = d_bases.numel() * d_bases.element_size() // 64 if n is None else int(n)
    _check(lib().h2hip_bases_pin_device(_dptr(d_bases), ctypes.c_size_t(n), _stream()), "h2hip_bases_pin_device")
    addr = d_bases.data_ptr()
    old = _device_pins.pop(addr, None)
    if old is not None:
        old.detach()
    # torch keeps one Python wrapper per storage alive as long as the storage is (checked on this image: a finalizer attached to
    # `t.view(-1).untyped_storage()` fires when the last tensor over the storage goes, not when the temporary view does)
    _device_pins[addr] = weakref.finalize(d_bases.untyped_storage(), _unpin_address, addr)


def bases_unpin_device(d_bases):
    fin = _device_pins.pop(d_bases.data_ptr(), None)
    if fin is not None:
        fin.detach()
    _check(lib().h2hip_bases_unpin(_dptr(d_bases)), "h2hip_bases_unpin")


def columns_pin(columns):
    """keep a proving key's constant columns (pk.fixed_cosets, pk.l0 / l_last / l_active_row, pk.permutation.cosets: (2^extended_k, 4)
    uint64 each) in HBM across host-pointer evaluate_h calls, keyed by their host addresses (h2hip_columns_pin)"""
    cols = list(columns)
    if not cols:
        return
    for a in cols:
        assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"] and a.shape == cols[0].shape and a.shape[1] == 4
    _check(lib().h2hip_columns_pin(_host_ptrs(cols), ctypes.c_size_t(len(cols)), ctypes.c_size_t(cols[0].shape[0])), "h2hip_columns_pin")


def columns_unpin(columns):
    cols = list(columns)
    if cols:
        _check(lib().h2hip_columns_unpin(_host_ptrs(cols), ctypes.c_size_t(len(cols))), "h2hip_columns_unpin")


def columns_pinned_info():
    """(columns, bytes of HBM) the pinned-column cache holds"""
    n, b = ctypes.c_size_t(0), ctypes.c_size_t(0)
    _check(lib().h2hip_columns_pinned_info(ctypes.byref(n), ctypes.byref(b)), "h2hip_columns_pinned_info")
    return n.value, b.value


def bases_pinned_info(bases):
    """(points, window bits, windows, bytes of HBM) of a pinned host array or device tensor"""
    ptr = _p(bases) if isinstance(bases, np.ndarray) else _dptr(bases)
    n, c, w, b = ctypes.c_size_t(0), ctypes.c_uint32(0), ctypes.c_uint32(0), ctypes.c_size_t(0)
    _check(lib().h2hip_bases_pinned_info(ptr, ctypes.byref(n), ctypes.byref(c), ctypes.byref(w), ctypes.byref(b)), "h2hip_bases_pinned_info")
    return n.value, c.value, w.value, b.value


# ------------------------------------------------------------------ bn256::Fr constants (halo2curves 0.3.1)
FR_MODULUS = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
FR_S = 28
FR_ROOT_OF_UNITY = 0x03ddb9f5166d18b798865ea93dd31f743215cf6dd39329c8d34f1ed960c37c9c
FR_ZETA = 0x30644e72e131a029048b6e193fd84104cc37a73fec2bc5e9b8ca0b2d36636f23
_FR_R = (1 << 256) % FR_MODULUS


class PrimeField:
    """what EvaluationDomain::new needs of `G::Scalar` (ff::PrimeField + FieldExt): modulus, 2-adicity S, ROOT_OF_UNITY
    (a primitive 2^S-th root), ZETA (a primitive cube root), and the 4 x u64 Montgomery limbs (R = 2^256) of an integer"""

    def __init__(self, name, modulus, S, root_of_unity, zeta):
        self.name, self.modulus, self.S, self.root_of_unity, self.zeta = name, modulus, S, root_of_unity, zeta
        self._R = (1 << 256) % modulus

    def from_int(self, v):
        m = (int(v) % self.modulus) * self._R % self.modulus
        return np.array([(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


BN254_FR = PrimeField("bn256::Fr", FR_MODULUS, FR_S, FR_ROOT_OF_UNITY, FR_ZETA)
# pasta_curves Fp (the scalar field of vesta / EqAffine, the curve of the reference's pinned verification key in
# tests/plonk_api.rs:624-632): multiplicative generator 5, 2-adicity 32, ROOT_OF_UNITY = 5^((p - 1) / 2^32), ZETA a cube root
# of unity (5^((p - 1) / 3); which of the two the crate fixes is not visible in the reference and does not enter omega).
_PASTA_P = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001
PASTA_FP = PrimeField("pasta::Fp", _PASTA_P, 32, pow(5, (_PASTA_P - 1) >> 32, _PASTA_P), pow(5, (_PASTA_P - 1) // 3, _PASTA_P))


def fr_from_int(v):
    """integer -> the 4 x u64 Montgomery limbs the reference stores"""
    m = (int(v) % FR_MODULUS) * _FR_R % FR_MODULUS
    return np.array([(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def fr_to_int(limbs):
    m = sum(int(x) << (64 * i) for i, x in enumerate(np.asarray(limbs, dtype=np.uint64).reshape(4)))
    return m * pow(_FR_R, -1, FR_MODULUS) % FR_MODULUS


# ------------------------------------------------------------------ poly/domain.rs
class EvaluationDomain:
    """poly::EvaluationDomain<Fr> (poly/domain.rs:18-34): holds the constants `new` computes (:39-142) and routes the
    conversions through the fused device entry points.  `EvaluationDomain.new(j, k)` computes them here with Python
    integers (field inversions are not on the accelerated path)."""

    FIELDS = ("omega", "omega_inv", "extended_omega", "extended_omega_inv", "g_coset", "g_coset_inv",
              "ifft_divisor", "extended_ifft_divisor")

    def __init__(self, k, extended_k, quotient_poly_degree, t_evaluations=None, **consts):
        self.k, self.extended_k, self.quotient_poly_degree = int(k), int(extended_k), int(quotient_poly_degree)
        self.n = 1 << self.k
        for f in self.FIELDS:
            setattr(self, f, _fe(consts[f]))
        self.t_evaluations = None if t_evaluations is None else _u64(t_evaluations, 4)
        self.field = BN254_FR

    def _device_field(self):
        if self.field is not BN254_FR:
            raise H2HipError("the engine transforms bn256::Fr only (domain over %s)" % self.field.name)

    @classmethod
    def new(cls, j, k, field=None):
        """EvaluationDomain::new (poly/domain.rs:39-142), generic over the scalar field as the reference is (`G::Scalar`):
        `field` is a PrimeField (default BN254_FR, the only field the device entry points serve; PASTA_FP exists so that the
        constructor can be checked against the one domain constant the reference's own tests pin, tests/plonk_api.rs:629-632)."""
        field = BN254_FR if field is None else field
        r = field.modulus
        quotient_poly_degree = j - 1                                    # :41
        n = 1 << k                                                      # :44
        extended_k = k                                                  # :49-52
        while (1 << extended_k) < n * quotient_poly_degree:
            extended_k += 1
        if extended_k > field.S:
            raise ValueError("extended_k exceeds the 2-adicity of the field")
        extended_omega = field.root_of_unity                            # :54-61
        for _ in range(extended_k, field.S):
            extended_omega = extended_omega * extended_omega % r
        omega = extended_omega                                          # :70-73
        for _ in range(k, extended_k):
            omega = omega * omega % r
        g_coset = field.zeta                                            # :81
        g_coset_inv = g_coset * g_coset % r                             # :82
        orig = pow(field.zeta, n, r)                                    # :84-107
        step = pow(extended_omega, n, r)
        t_evaluations, cur = [], orig
        while True:
            t_evaluations.append(cur)
            cur = cur * step % r
            if cur == orig:
                break
        assert len(t_evaluations) == 1 << (extended_k - k)             # :98
        t_evaluations = [pow((c - 1) % r, -1, r) for c in t_evaluations]   # :101-103, batch_invert :117-124
        consts = {
            "omega": omega, "omega_inv": pow(omega, -1, r), "extended_omega": extended_omega,
            "extended_omega_inv": pow(extended_omega, -1, r), "g_coset": g_coset, "g_coset_inv": g_coset_inv,
            "ifft_divisor": pow(1 << k, -1, r), "extended_ifft_divisor": pow(1 << extended_k, -1, r),   # :109-110
        }
        d = cls(k, extended_k, quotient_poly_degree, t_evaluations=np.stack([field.from_int(c) for c in t_evaluations]),
                **{f: field.from_int(v) for f, v in consts.items()})
        d.barycentric_weight = field.from_int(pow(n, -1, r))            # :114
        d.field = field
        d.ints = dict(consts)  # the same constants as integers (canonical form)
        return d

    def extended_len(self):
        return 1 << self.extended_k

    def lagrange_to_coeff(self, a):
        """poly/domain.rs:226-236"""
        self._device_field()
        a = _u64(a, 4).copy()
        assert a.shape[0] == 1 << self.k
        _check(lib().h2hip_ifft_bn254_fr(_p(a), _p(self.omega_inv), ctypes.c_uint32(self.k), _p(self.ifft_divisor)),
               "h2hip_ifft_bn254_fr")
        return a

    def coeff_to_extended(self, a):
        """poly/domain.rs:240-254"""
        self._device_field()
        a = _u64(a, 4)
        assert a.shape[0] == 1 << self.k
        out = np.zeros((self.extended_len(), 4), dtype=np.uint64)
        _check(lib().h2hip_coeff_to_extended_bn254_fr(_p(a), ctypes.c_uint32(self.k), _p(out), ctypes.c_uint32(self.extended_k),
                                                      _p(self.extended_omega), _p(self.g_coset), _p(self.g_coset_inv)),
               "h2hip_coeff_to_extended_bn254_fr")
        return out

    def extended_to_coeff(self, a):
        """poly/domain.rs:281-303 (including the truncate at :299-300)"""
        self._device_field()
        a = _u64(a, 4).copy()
        assert a.shape[0] == self.extended_len()
        _check(lib().h2hip_extended_to_coeff_bn254_fr(_p(a), ctypes.c_uint32(self.extended_k), _p(self.extended_omega_inv),
                                                      _p(self.extended_ifft_divisor), _p(self.g_coset), _p(self.g_coset_inv)),
               "h2hip_extended_to_coeff_bn254_fr")
        return a[: self.n * self.quotient_poly_degree]


    # the same conversions for a list of host columns, pipelined over PCIe inside one call (h2hip_*_batch): what a patched
    # prover calls where the reference maps the single-column method over its polynomials (plonk/prover.rs:476-490,
    # plonk/evaluation.rs:306-323)
    def lagrange_to_coeff_batch(self, polys):
        self._device_field()
        cols = [_u64(a, 4).copy() for a in polys]
        for a in cols:
            assert a.shape[0] == 1 << self.k
        if cols:
            _check(lib().h2hip_ifft_bn254_fr_batch(_host_ptrs(cols), ctypes.c_size_t(len(cols)), _p(self.omega_inv), ctypes.c_uint32(self.k),
                                                   _p(self.ifft_divisor)), "h2hip_ifft_bn254_fr_batch")
        return cols

    def coeff_to_extended_batch(self, polys):
        self._device_field()
        cols = [_u64(a, 4) for a in polys]
        for a in cols:
            assert a.shape[0] == 1 << self.k
        outs = [np.empty((self.extended_len(), 4), dtype=np.uint64) for _ in cols]
        if cols:
            _check(lib().h2hip_coeff_to_extended_bn254_fr_batch(_host_ptrs(cols), ctypes.c_uint32(self.k), _host_ptrs(outs), ctypes.c_size_t(len(cols)),
                                                                ctypes.c_uint32(self.extended_k), _p(self.extended_omega), _p(self.g_coset),
                                                                _p(self.g_coset_inv)), "h2hip_coeff_to_extended_bn254_fr_batch")
        return outs

    def extended_to_coeff_batch(self, polys):
        self._device_field()
        cols = [_u64(a, 4).copy() for a in polys]
        for a in cols:
            assert a.shape[0] == self.extended_len()
        if cols:
            _check(lib().h2hip_extended_to_coeff_bn254_fr_batch(_host_ptrs(cols), ctypes.c_size_t(len(cols)), ctypes.c_uint32(self.extended_k),
                                                                _p(self.extended_omega_inv), _p(self.extended_ifft_divisor), _p(self.g_coset),
                                                                _p(self.g_coset_inv)), "h2hip_extended_to_coeff_bn254_fr_batch")
        return [a[: self.n * self.quotient_poly_degree] for a in cols]

    def divide_by_vanishing_poly(self, a, t_evaluations=None):
        """poly/domain.rs:307-326"""
        self._device_field()
        a = _u64(a, 4).copy()
        t = self.t_evaluations if t_evaluations is None else _u64(t_evaluations, 4)
        assert a.shape[0] == self.extended_len()
        _check(lib().h2hip_divide_by_vanishing_poly_bn254_fr(_p(a), ctypes.c_uint32(self.extended_k), _p(t), ctypes.c_uint32(t.shape[0])),
               "h2hip_divide_by_vanishing_poly_bn254_fr")
        return a


# ------------------------------------------------------------------ poly/kzg/commitment.rs
class ParamsKZG:
    """poly::kzg::commitment::ParamsKZG<Bn256> (poly/kzg/commitment.rs:22-30): g and g_lagrange
    are pinned on the GPU for the life of the object; commit / commit_lagrange are
    best_multiexp over them (:281-292, :327-334; the blind is ignored there too)."""

    def __init__(self, k, g, g_lagrange=None):
        """g_lagrange None: derived from g with g_to_lagrange, as downsize does (:274)"""
        self.k, self.n = int(k), 1 << int(k)
        self.g = _u64(g, 8).copy()
        self.g_lagrange = g_to_lagrange(self.g, self.k) if g_lagrange is None else _u64(g_lagrange, 8).copy()
        assert self.g.shape[0] == self.n and self.g_lagrange.shape[0] == self.n
        bases_pin(self.g)
        bases_pin(self.g_lagrange)

    @classmethod
    def setup(cls, k, secret):
        """ParamsKZG::setup (poly/kzg/commitment.rs:61-129) with the secret given (an int, or 4 Montgomery limbs) instead
        of drawn from an rng: g[i] = [s^i] G1, g_lagrange[i] = [l_i(s)] G1, computed on the GPU"""
        if not 0 <= int(k) <= FR_S:
            raise H2HipError("kzg_setup: assertion failed: k <= Fr::S")  # :64 (before any allocation of 2^k points)
        s = fr_from_int(secret) if isinstance(secret, int) else _fe(secret)
        n = 1 << int(k)
        g = np.zeros((n, 8), dtype=np.uint64)
        gl = np.zeros((n, 8), dtype=np.uint64)
        _check(lib().h2hip_kzg_setup_bn254(ctypes.c_uint32(k), _p(s), _p(g), _p(gl)), "h2hip_kzg_setup_bn254")
        return cls(k, g, gl)

    def commit_lagrange(self, poly, blind=None):
        poly = _u64(poly, 4)
        size = poly.shape[0]
        assert self.g_lagrange.shape[0] >= size  # assert!(bases.len() >= size), :290
        out = np.zeros(12, dtype=np.uint64)
        _check(lib().h2hip_msm_bn254(_p(poly), _p(self.g_lagrange), ctypes.c_size_t(size), _p(out)), "h2hip_msm_bn254")
        return out

    def commit_lagrange_many(self, polys):
        """the advice-column loop of plonk/prover.rs:361-365 as one pipelined batch"""
        return best_multiexp_batch(polys, self.g_lagrange)

    def commit(self, poly, blind=None):
        poly = _u64(poly, 4)
        size = poly.shape[0]
        assert self.g.shape[0] >= size  # :332
        out = np.zeros(12, dtype=np.uint64)
        _check(lib().h2hip_msm_bn254(_p(poly), _p(self.g), ctypes.c_size_t(size), _p(out)), "h2hip_msm_bn254")
        return out

    def downsize(self, k):
        """ParamsKZG::downsize (poly/kzg/commitment.rs:267-275)"""
        assert k <= self.k  # :268
        self.close()
        self.k, self.n = int(k), 1 << int(k)
        self.g = self.g[:self.n].copy()                       # self.g.truncate(self.n)
        self.g_lagrange = g_to_lagrange(self.g, self.k)       # :274
        bases_pin(self.g)
        bases_pin(self.g_lagrange)

    def close(self):
        """unpin both arrays (idempotent); also runs when the object is dropped or leaves a `with` block, so a dead
        ParamsKZG never leaves its device copies behind under host addresses numpy may hand out again"""
        for b in (getattr(self, "g", None), getattr(self, "g_lagrange", None)):
            if b is None:
                continue
            try:
                bases_unpin(b)
            except H2HipError:
                pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------ device-resident entry points
def _dptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def msm_device(d_scalars, d_bases, n=None):
    """d_scalars / d_bases: torch CUDA tensors holding n x 32 B and n x 64 B (any dtype)"""
    nb = d_scalars.numel() * d_scalars.element_size()
    n = nb // 32 if n is None else int(n)
    out = np.zeros(12, dtype=np.uint64)
    _check(lib().h2hip_msm_bn254_device(_dptr(d_scalars), _dptr(d_bases), ctypes.c_size_t(n), _p(out), _stream()), "h2hip_msm_bn254_device")
    return out


def msm_batch_device(d_scalars_list, d_bases, n=None):
    """count MSMs over the same device-resident bases; returns (count, 12) uint64"""
    count = len(d_scalars_list)
    if n is None:
        n = d_scalars_list[0].numel() * d_scalars_list[0].element_size() // 32
    ptrs = (ctypes.c_void_p * count)(*[t.data_ptr() for t in d_scalars_list])
    out = np.zeros((count, 12), dtype=np.uint64)
    _check(lib().h2hip_msm_bn254_batch_device(ptrs, _dptr(d_bases), ctypes.c_size_t(n), ctypes.c_size_t(count), _p(out), _stream()),
           "h2hip_msm_bn254_batch_device")
    return out


def ntt_device(d_a, omega, log_n):
    _check(lib().h2hip_ntt_bn254_fr_device(_dptr(d_a), _p(_fe(omega)), ctypes.c_uint32(log_n), _stream()), "h2hip_ntt_bn254_fr_device")


def ifft_device(d_a, omega_inv, log_n, divisor):
    _check(lib().h2hip_ifft_bn254_fr_device(_dptr(d_a), _p(_fe(omega_inv)), ctypes.c_uint32(log_n), _p(_fe(divisor)), _stream()),
           "h2hip_ifft_bn254_fr_device")


def coeff_to_extended_device(d_a, k, extended_k, extended_omega, g_coset, g_coset_inv):
    _check(lib().h2hip_coeff_to_extended_bn254_fr_device(_dptr(d_a), ctypes.c_uint32(k), ctypes.c_uint32(extended_k), _p(_fe(extended_omega)),
                                                         _p(_fe(g_coset)), _p(_fe(g_coset_inv)), _stream()),
           "h2hip_coeff_to_extended_bn254_fr_device")


def extended_to_coeff_device(d_a, extended_k, extended_omega_inv, extended_ifft_divisor, g_coset, g_coset_inv):
    _check(lib().h2hip_extended_to_coeff_bn254_fr_device(_dptr(d_a), ctypes.c_uint32(extended_k), _p(_fe(extended_omega_inv)),
                                                         _p(_fe(extended_ifft_divisor)), _p(_fe(g_coset)), _p(_fe(g_coset_inv)), _stream()),
           "h2hip_extended_to_coeff_bn254_fr_device")


def _ptr_array(tensors):
    return (ctypes.c_void_p * max(1, len(tensors)))(*[t.data_ptr() for t in tensors])


def ntt_batch_device(d_list, omega, log_n):
    """the same NTT over every tensor of d_list, one launch per pass"""
    _check(lib().h2hip_ntt_bn254_fr_batch_device(_ptr_array(d_list), ctypes.c_size_t(len(d_list)), _p(_fe(omega)), ctypes.c_uint32(log_n), _stream()),
           "h2hip_ntt_bn254_fr_batch_device")


def ifft_batch_device(d_list, omega_inv, log_n, divisor):
    _check(lib().h2hip_ifft_bn254_fr_batch_device(_ptr_array(d_list), ctypes.c_size_t(len(d_list)), _p(_fe(omega_inv)), ctypes.c_uint32(log_n),
                                                  _p(_fe(divisor)), _stream()), "h2hip_ifft_bn254_fr_batch_device")


def coeff_to_extended_batch_device(d_list, k, extended_k, extended_omega, g_coset, g_coset_inv):
    _check(lib().h2hip_coeff_to_extended_bn254_fr_batch_device(_ptr_array(d_list), ctypes.c_size_t(len(d_list)), ctypes.c_uint32(k),
                                                               ctypes.c_uint32(extended_k), _p(_fe(extended_omega)), _p(_fe(g_coset)),
                                                               _p(_fe(g_coset_inv)), _stream()), "h2hip_coeff_to_extended_bn254_fr_batch_device")


def gen_scalars_device(seed, n, start=0, device="cuda"):
    import torch
    out = torch.empty((n, 4), dtype=torch.int64, device=device)
    _check(lib().h2hip_gen_scalars_device(ctypes.c_uint64(seed), ctypes.c_uint64(start), ctypes.c_size_t(n), _dptr(out), _stream()),
           "h2hip_gen_scalars_device")
    return out


def gen_points_device(seed, n, start=0, device="cuda"):
    import torch
    out = torch.empty((n, 8), dtype=torch.int64, device=device)
    _check(lib().h2hip_gen_points_device(ctypes.c_uint64(seed), ctypes.c_uint64(start), ctypes.c_size_t(n), _dptr(out), _stream()),
           "h2hip_gen_points_device")
    return out


def to_numpy_u64(t):
    return t.detach().cpu().numpy().view(np.uint64)


# ------------------------------------------------------------------ tuning / measurement
def set_msm_window(c):
    _check(lib().h2hip_set_msm_window(ctypes.c_uint32(c)), "h2hip_set_msm_window")


def get_msm_window(n):
    return int(lib().h2hip_get_msm_window(n))


def get_msm_window_fixed_base(n):
    return int(lib().h2hip_get_msm_window_fixed_base(n))


def msm_min_n():
    return int(lib().h2hip_msm_min_n())


def ntt_min_log_n():
    return int(lib().h2hip_ntt_min_log_n())


def lazy_pin_after():
    """HALO2_HIP_LAZY_PIN: unpinned sightings of a host bases array after which the library pins it itself (0 = never)"""
    return int(lib().h2hip_lazy_pin_after())


def profile_enable(on=True):
    """True / 1: every stage; 2: only the dominant kernel, through its own dispatch (no gaps); False / 0: off"""
    _check(lib().h2hip_profile_enable(int(on)), "h2hip_profile_enable")


def profile_reset():
    _check(lib().h2hip_profile_reset(), "h2hip_profile_reset")


def profile_get(stage):
    ms = ctypes.c_double(0)
    cnt = ctypes.c_uint64(0)
    _check(lib().h2hip_profile_get(stage.encode(), ctypes.byref(ms), ctypes.byref(cnt)), "h2hip_profile_get")
    return ms.value, cnt.value
