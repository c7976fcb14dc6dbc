"""CPU: the C-ABI library loads, exports every symbol include/halo2hip.h declares, and refuses
to compute without a GPU (no CPU fallback in the product path)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared_symbols(headers=("halo2hip.h", "halo2hip_debug.h")):
    syms = set()
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        syms |= set(re.findall(r"\b(h2hip_[a-z0-9_]+)\s*\(", text))
    return sorted(syms)


def test_debug_hooks_live_in_their_own_header():
    assert not [s for s in _declared_symbols(("halo2hip.h",)) if s.startswith("h2hip_debug_")]
    assert all(s.startswith("h2hip_debug_") for s in _declared_symbols(("halo2hip_debug.h",)))


def test_header_declares_expected_entry_points():
    syms = _declared_symbols()
    for must in ("h2hip_init", "h2hip_msm_bn254", "h2hip_ntt_bn254_fr", "h2hip_bases_pin", "h2hip_bases_unpin",
                 "h2hip_shutdown", "h2hip_msm_bn254_device", "h2hip_ntt_bn254_fr_device", "h2hip_g1_fold"):
        assert must in syms


def test_library_exports_every_declared_symbol(h2):
    L = h2.lib()
    for s in _declared_symbols():
        assert hasattr(L, s), "libhalo2hip.so does not export " + s


def test_product_does_not_link_or_import_the_oracle(h2):
    # the oracle is test infrastructure: nothing under halo2-pse_amd/ may reference it
    pkg = os.path.join(ROOT, "halo2-pse_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in txt and "bn254_oracle" not in txt, os.path.join(dirpath, f)
                for line in txt.splitlines():
                    assert not re.match(r"\s*(from|import)\s+oracle\b", line), os.path.join(dirpath, f)


def test_no_gpu_means_loud_failure(h2):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    a = np.zeros((4, 4), dtype=np.uint64)
    omega = np.zeros(4, dtype=np.uint64)
    with pytest.raises(h2.H2HipError):
        h2.best_fft(a, omega, 2)
    with pytest.raises(h2.H2HipError):
        h2.best_multiexp(np.zeros((2, 4), dtype=np.uint64), np.zeros((2, 8), dtype=np.uint64))


def test_host_group_helpers_match_oracle(h2, oracle, golden):
    # h2hip_g1_fold / h2hip_g1_to_affine are host-side (the fold of arithmetic.rs:153)
    sc, bs = golden["msm_33_scalars"], golden["msm_33_bases"]
    parts = np.stack([oracle.best_multiexp(sc[:10], bs[:10]), oracle.best_multiexp(sc[10:20], bs[10:20]),
                      oracle.best_multiexp(sc[20:], bs[20:])])
    folded = h2.g1_fold(parts)
    assert np.array_equal(h2.g1_to_affine(folded), golden["msm_33_result"])
    assert np.array_equal(oracle.g1_to_affine(folded), golden["msm_33_result"])
    ident = h2.g1_fold(np.zeros((0, 12), dtype=np.uint64))
    assert np.array_equal(h2.g1_to_affine(ident), np.zeros(8, dtype=np.uint64))
    # P + (-P) and P + P through the fold
    p = oracle.best_multiexp(sc[:1], bs[:1])
    aff = oracle.g1_to_affine(p)
    neg = aff.copy()
    q = oracle.constant(oracle.FQ, 2)
    yneg = oracle.fe_binop("sub", oracle.FQ, np.zeros((1, 4), dtype=np.uint64), aff[4:].reshape(1, 4))[0]
    neg[4:] = yneg
    one = oracle.constant(oracle.FQ, 0)
    negj = np.concatenate([neg, one])
    assert np.array_equal(h2.g1_to_affine(h2.g1_fold(np.stack([p, negj]))), np.zeros(8, dtype=np.uint64))
    dbl = h2.g1_to_affine(h2.g1_fold(np.stack([p, p])))
    assert np.array_equal(dbl, oracle.g1_to_affine(oracle.g1_add(p, p)))


def test_batch_normalize_host(h2, oracle, golden):
    import ctypes
    sc, bs = golden["msm_33_scalars"], golden["msm_33_bases"]
    pts = [oracle.best_multiexp(sc[i:i + 5], bs[i:i + 5]) for i in range(0, 30, 5)]
    pts.insert(2, np.zeros(12, dtype=np.uint64))  # an identity in the middle (z = 0)
    xyz = np.stack(pts)
    out = np.zeros((len(pts), 8), dtype=np.uint64)
    rc = h2.lib().h2hip_g1_batch_normalize(xyz.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(len(pts)), out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    for i, p in enumerate(pts):
        assert np.array_equal(out[i], oracle.g1_to_affine(p)), i
