"""GPU parity of the host-pointer batched transforms (h2hip_{ntt,ifft,coeff_to_extended,extended_to_coeff}_bn254_fr_batch):
the three-stage PCIe pipeline (upload i + 1 | transform i | download i - 1, one copier thread per direction) must return, column by
column, exactly what the oracle's restatement of poly/domain.rs returns -- whatever the number of columns, the grouping of small
columns, the cut of a batch into several runs, and with host threads calling at once.  Run with `pytest -m gpu`."""
import ctypes
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NT = min(16, os.cpu_count() or 1)


@pytest.fixture(scope="module", autouse=True)
def _engine(h2):
    h2.init()
    yield
    h2.lib().h2hip_debug_set_ntt_host_batch(ctypes.c_uint64(0), ctypes.c_uint64(0))


def _dom(h2, oracle, j, k):
    d, _ = oracle.domain_new(j, k)
    dom = h2.EvaluationDomain.new(j, k)
    assert dom.extended_k == d.extended_k
    return d, dom


@pytest.mark.parametrize("k,count", [(3, 5), (9, 7), (12, 1), (12, 2), (14, 9), (17, 10), (19, 3), (20, 4)])
def test_lagrange_to_coeff_batch_vs_oracle(h2, oracle, k, count):
    d, dom = _dom(h2, oracle, 4, k)
    cols = [oracle.gen_scalars(7000 + 13 * k + i, 1 << k, num_threads=NT) for i in range(count)]
    keep = [c.copy() for c in cols]
    got = dom.lagrange_to_coeff_batch(cols)
    assert all(np.array_equal(a, b) for a, b in zip(cols, keep))  # inputs are the caller's: untouched
    for c, g in zip(cols, got):
        assert np.array_equal(g, oracle.lagrange_to_coeff(d, c, NT))


@pytest.mark.parametrize("j,k,count", [(4, 5, 3), (4, 10, 6), (3, 12, 4), (4, 14, 5), (4, 17, 10), (2, 18, 3), (9, 15, 2)])
def test_coeff_to_extended_batch_vs_oracle(h2, oracle, j, k, count):
    d, dom = _dom(h2, oracle, j, k)
    cols = [oracle.gen_scalars(8000 + 17 * k + i, 1 << k, num_threads=NT) for i in range(count)]
    got = dom.coeff_to_extended_batch(cols)
    for c, g in zip(cols, got):
        assert g.shape == (1 << d.extended_k, 4)
        assert np.array_equal(g, oracle.coeff_to_extended(d, c, NT))


@pytest.mark.parametrize("j,k,count", [(4, 4, 3), (4, 12, 5), (4, 17, 2), (3, 16, 3)])
def test_extended_to_coeff_batch_vs_oracle(h2, oracle, j, k, count):
    d, dom = _dom(h2, oracle, j, k)
    cols = [oracle.gen_scalars(9000 + 19 * k + i, 1 << d.extended_k, num_threads=NT) for i in range(count)]
    got = dom.extended_to_coeff_batch(cols)
    for c, g in zip(cols, got):
        assert np.array_equal(g, oracle.extended_to_coeff(d, c, NT))


@pytest.mark.parametrize("k,count", [(0, 3), (1, 2), (10, 4), (16, 6), (21, 2)])
def test_best_fft_batch_in_place_vs_oracle(h2, oracle, k, count):
    d, _ = oracle.domain_new(2, max(k, 1))
    omega = d.fe("omega") if k >= 1 else h2.fr_from_int(1)
    if k == 0:
        cols = [oracle.gen_scalars(100 + i, 1, num_threads=1) for i in range(count)]
        want = [c.copy() for c in cols]  # a one-point transform is the identity
    else:
        cols = [oracle.gen_scalars(6000 + i, 1 << k, num_threads=NT) for i in range(count)]
        want = [oracle.best_fft(c, omega, k, NT) for c in cols]
    h2.best_fft_batch(cols, omega, k)
    for g, w in zip(cols, want):
        assert np.array_equal(g, w)


def test_batch_cut_into_runs_and_grouped_steps(h2, oracle):
    """force (a) several pipelined runs per call (a 3-column budget), (b) several columns per pipeline step"""
    k = 13
    d, dom = _dom(h2, oracle, 4, k)
    cols = [oracle.gen_scalars(5000 + i, 1 << k, num_threads=NT) for i in range(11)]
    want_c = [oracle.lagrange_to_coeff(d, c, NT) for c in cols]
    want_e = [oracle.coeff_to_extended(d, c, NT) for c in cols]
    L = h2.lib()
    col_bytes = 32 << d.extended_k
    for run_bytes, group_bytes in [(3 * col_bytes, 1), (3 * col_bytes, 4 * col_bytes), (1, 1), (0, 64 << 20), (2 * col_bytes + 5, 2 * col_bytes)]:
        L.h2hip_debug_set_ntt_host_batch(ctypes.c_uint64(run_bytes), ctypes.c_uint64(group_bytes))
        try:
            assert all(np.array_equal(g, w) for g, w in zip(dom.lagrange_to_coeff_batch(cols), want_c)), (run_bytes, group_bytes)
            assert all(np.array_equal(g, w) for g, w in zip(dom.coeff_to_extended_batch(cols), want_e)), (run_bytes, group_bytes)
        finally:
            L.h2hip_debug_set_ntt_host_batch(ctypes.c_uint64(0), ctypes.c_uint64(0))


def test_batch_2p22_columns_full_size(h2, oracle):
    """BASELINE.json configs[2]'s size through the host batch: four 2^22 columns, forward then scaled inverse returns the input, and
    one column of each direction equals the oracle"""
    k = 22
    d, dom = _dom(h2, oracle, 2, k)
    cols = [oracle.gen_scalars(0x5EED0003 + i, 1 << k, num_threads=NT) for i in range(4)]
    work = [c.copy() for c in cols]
    h2.best_fft_batch(work, d.fe("omega"), k)
    assert np.array_equal(work[2], oracle.best_fft(cols[2], d.fe("omega"), k, NT))
    back = dom.lagrange_to_coeff_batch(work)
    for b, c in zip(back, cols):
        assert np.array_equal(b, c)


def test_host_batches_from_threads_and_next_to_other_entry_points(h2, oracle):
    """rayon workers may call in at once: batches, lone transforms and MSMs from six threads"""
    k = 12
    d, dom = _dom(h2, oracle, 4, k)
    cols = [oracle.gen_scalars(4000 + i, 1 << k, num_threads=NT) for i in range(6)]
    want_c = [oracle.lagrange_to_coeff(d, c, NT) for c in cols]
    want_e = [oracle.coeff_to_extended(d, c, NT) for c in cols]
    pts = oracle.gen_points(0xBEEF, 1 << k, num_threads=NT)
    want_m = oracle.g1_to_affine(oracle.best_multiexp(cols[0], pts, NT))
    errs = []

    def worker(t):
        try:
            for _ in range(3):
                if t % 3 == 0:
                    got = dom.lagrange_to_coeff_batch(cols)
                    assert all(np.array_equal(g, w) for g, w in zip(got, want_c))
                elif t % 3 == 1:
                    got = dom.coeff_to_extended_batch(cols)
                    assert all(np.array_equal(g, w) for g, w in zip(got, want_e))
                else:
                    assert np.array_equal(h2.g1_to_affine(h2.best_multiexp(cols[0], pts)), want_m)
                    assert np.array_equal(dom.lagrange_to_coeff(cols[1]), want_c[1])
        except Exception as e:  # noqa: BLE001
            errs.append((t, repr(e)))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errs, errs


def test_host_batch_rejects_bad_arguments(h2):
    L = h2.lib()
    one = h2.fr_from_int(1)
    a = np.zeros((4, 4), dtype=np.uint64)
    ptrs = (ctypes.c_void_p * 2)(a.ctypes.data, None)
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    assert L.h2hip_ntt_bn254_fr_batch(ptrs, ctypes.c_size_t(2), p(one), ctypes.c_uint32(2)) == 1       # null column
    assert L.h2hip_ntt_bn254_fr_batch(ptrs, ctypes.c_size_t(1), None, ctypes.c_uint32(2)) == 1          # null omega
    assert L.h2hip_ntt_bn254_fr_batch(ptrs, ctypes.c_size_t(1), p(one), ctypes.c_uint32(29)) == 1       # log_n > 28
    assert L.h2hip_ntt_bn254_fr_batch(None, ctypes.c_size_t(0), p(one), ctypes.c_uint32(2)) == 0        # empty batch
    bad = np.full(4, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
    assert L.h2hip_ifft_bn254_fr_batch(ptrs, ctypes.c_size_t(1), p(bad), ctypes.c_uint32(2), p(one)) == 1   # unreduced omega
    assert L.h2hip_coeff_to_extended_bn254_fr_batch(ptrs, ctypes.c_uint32(5), ptrs, ctypes.c_size_t(1), ctypes.c_uint32(4), p(one), p(one), p(one)) == 1
    assert b"" != L.h2hip_last_error()
