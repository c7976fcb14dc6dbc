"""GPU parity tests of the streamed host-slice MSM (round 3): h2hip_msm_bn254 / h2hip_msm_bn254_batch with
host-resident scalars -- the call best_multiexp actually makes (arithmetic.rs:132, one per column at
plonk/prover.rs:361-365) -- cut into chunks that cross PCIe under the work and add into ONE persistent bucket set
(msm_accum_kernel, cont = 1), and the fused batch cut into groups of columns.  Every case is bit-exact (after
normalising to affine) against the CPU oracle; the full-size cases against the device-resident path, which
test_gpu_parity.py / test_msm_fixed_base.py pin to the oracle.  The chunk count, ladder ratio and size threshold are
forced through h2hip_debug_set_msm_stream so that small inputs take the streamed path.
Run with `pytest -m gpu` on an MI355X."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NT = min(16, os.cpu_count() or 1)


def aff(h2, xyz):
    return h2.g1_to_affine(xyz)


def set_stream(h2, chunks, permille=0, min_n=0):
    assert h2.lib().h2hip_debug_set_msm_stream(ctypes.c_uint32(chunks), ctypes.c_uint32(permille), ctypes.c_size_t(min_n)) == 0


@pytest.fixture(autouse=True)
def _engine(h2):
    h2.init()
    yield
    set_stream(h2, 0)
    h2.set_msm_window(0)


def prover_like(oracle, n, seed):
    # 90 % zero, 5 % in {1,2}, 5 % uniform
    rng = np.random.default_rng(seed)
    sc = oracle.gen_scalars(seed, n, num_threads=NT)
    u = rng.random(n)
    sc[u < 0.90] = 0
    sc[(u >= 0.90) & (u < 0.925)] = oracle.fe_from_int(oracle.FR, 1)
    sc[(u >= 0.925) & (u < 0.95)] = oracle.fe_from_int(oracle.FR, 2)
    return sc


def neg_points(oracle, pts):
    """(x, -y) for an array of affine points in the reference's layout"""
    out = pts.copy()
    y = np.ascontiguousarray(pts[:, 4:])
    out[:, 4:] = oracle.fe_binop("sub", oracle.FQ, np.zeros_like(y), y)
    return out


def both_forms(h2, sc, bs, want, tag):
    """the unpinned call (plain form; the bases stream in with the scalars) and the pinned one (fixed-base form)"""
    assert np.array_equal(aff(h2, h2.best_multiexp(sc, bs)), want), (tag, "plain")
    h2.bases_pin(bs)
    try:
        assert np.array_equal(aff(h2, h2.best_multiexp(sc, bs)), want), (tag, "fixed")
    finally:
        h2.bases_unpin(bs)


@pytest.mark.parametrize("chunks,permille", [(2, 600), (3, 600), (4, 600), (7, 900), (5, 1500)])
def test_streamed_msm_vs_oracle(h2, oracle, chunks, permille):
    n = 40000 + 123
    bs = oracle.gen_points(91, n, num_threads=NT)
    sc = oracle.gen_scalars(92, n, num_threads=NT)
    want = oracle.g1_to_affine(oracle.best_multiexp(sc, bs, NT))
    set_stream(h2, chunks, permille, 1024)
    both_forms(h2, sc, bs, want, (chunks, permille))


def test_streamed_msm_skewed_columns(h2, oracle):
    """prover-like columns (over-full buckets: the chunked heavy path adds into the persistent parts), a 99.98 %-zero column,
    all scalars equal, and a column of r - 1"""
    n = 1 << 15
    bs = oracle.gen_points(93, n, num_threads=NT)
    set_stream(h2, 4, 600, 1024)
    cols = {"prover": prover_like(oracle, n, 94)}
    z = np.zeros((n, 4), dtype=np.uint64)
    z[::5000] = oracle.gen_scalars(95, len(z[::5000]))
    cols["sparse"] = z
    cols["equal"] = np.repeat(oracle.gen_scalars(96, 1), n, axis=0)
    cols["rm1"] = np.repeat(oracle.fe_from_int(oracle.FR, -1)[None, :], n, axis=0)
    cols["ones"] = np.repeat(oracle.fe_from_int(oracle.FR, 1)[None, :], n, axis=0)
    for name, sc in cols.items():
        sc = np.ascontiguousarray(sc)
        want = oracle.g1_to_affine(oracle.best_multiexp(sc, bs, NT))
        both_forms(h2, sc, bs, want, name)


def test_streamed_msm_cancellation_and_doubling_across_chunks(h2, oracle):
    """A bucket that holds P after chunk 0 meets -P in a later chunk (identity in the persistent part, then a fresh first
    addition), and meets P again (the doubling case of the mixed addition into a loaded accumulator)."""
    n = 1 << 14
    bs = oracle.gen_points(97, n, num_threads=NT)
    sc = oracle.gen_scalars(98, n, num_threads=NT)
    q = n // 4
    # pairs of the last quarter repeat pairs of the first: half of them negated (cancel), half as they are (double)
    bs[3 * q:3 * q + q // 2] = neg_points(oracle, bs[:q // 2])
    bs[3 * q + q // 2:] = bs[q // 2:q]
    sc[3 * q:] = sc[:q]
    # and a block of the third quarter undoes the cancelling pairs' partners again
    bs[2 * q:2 * q + 64] = bs[:64]
    sc[2 * q:2 * q + 64] = sc[:64]
    want = oracle.g1_to_affine(oracle.best_multiexp(sc, bs, NT))
    for chunks, permille in ((4, 1000), (2, 1000), (8, 1000)):
        set_stream(h2, chunks, permille, 1024)
        both_forms(h2, sc, bs, want, chunks)
    # everything cancels: the result is the identity
    bs2 = np.concatenate([bs[:q], neg_points(oracle, bs[:q])])
    sc2 = np.concatenate([sc[:q], sc[:q]])
    set_stream(h2, 2, 1000, 1024)
    both_forms(h2, np.ascontiguousarray(sc2), np.ascontiguousarray(bs2), np.zeros(8, dtype=np.uint64), "all-cancel")


@pytest.mark.parametrize("c", [5, 9, 13, 16])
def test_streamed_msm_window_widths(h2, oracle, c):
    """few buckets: several lanes per bucket (split parts stay persistent across chunks, one combine at the end)"""
    n = 30000
    bs = oracle.gen_points(99, n, num_threads=NT)
    sc = prover_like(oracle, n, 100) if c == 9 else oracle.gen_scalars(100, n, num_threads=NT)
    want = oracle.g1_to_affine(oracle.best_multiexp(sc, bs, NT))
    h2.set_msm_window(c)
    set_stream(h2, 3, 700, 1024)
    both_forms(h2, sc, bs, want, c)


def test_streamed_msm_2p17_split_parts(h2, oracle):
    """the prover's size: 2^17 pairs over a c = 17 table has 2^16 buckets, accumulated by up to 8 lanes each"""
    n = 1 << 17
    bs = oracle.gen_points(101, n, num_threads=NT)
    sc = oracle.gen_scalars(102, n, num_threads=NT)
    want = oracle.g1_to_affine(oracle.best_multiexp(sc, bs, NT))
    set_stream(h2, 4, 600, 1024)
    both_forms(h2, sc, bs, want, "2^17")


@pytest.mark.parametrize("count,n", [(2, 5000), (5, 1 << 13), (16, 1 << 12), (37, 3000)])
def test_fused_batch_groups_vs_oracle(h2, oracle, count, n):
    """h2hip_msm_bn254_batch with host columns: groups of columns stream in, one fused run each (a group of one column runs
    the lone MSM's plan); plain and fixed-base"""
    bs = oracle.gen_points(103, n, num_threads=NT)
    cols = [prover_like(oracle, n, 200 + j) if j % 3 == 0 else oracle.gen_scalars(200 + j, n, num_threads=NT) for j in range(count)]
    want = [oracle.g1_to_affine(oracle.best_multiexp(s_, bs, NT)) for s_ in cols]
    set_stream(h2, 4, 600, 1024)
    got = h2.best_multiexp_batch(cols, bs)
    for j in range(count):
        assert np.array_equal(aff(h2, got[j]), want[j]), ("plain", j)
    h2.bases_pin(bs)
    try:
        got = h2.best_multiexp_batch(cols, bs)
        for j in range(count):
            assert np.array_equal(aff(h2, got[j]), want[j]), ("fixed", j)
    finally:
        h2.bases_unpin(bs)


def test_fused_batch_groups_split_by_run_capacity(h2, oracle):
    """a group larger than one fused run holds is cut again (capacity lowered through the fuse-limit hook)"""
    n, count = 2000, 24
    bs = oracle.gen_points(104, n, num_threads=NT)
    cols = [oracle.gen_scalars(300 + j, n) for j in range(count)]
    want = [oracle.g1_to_affine(oracle.best_multiexp(s_, bs, NT)) for s_ in cols]
    set_stream(h2, 3, 600, 1024)
    L = h2.lib()
    L.h2hip_debug_set_msm_fuse_limits(ctypes.c_size_t(5 * n * 26), ctypes.c_size_t(0))
    try:
        got = h2.best_multiexp_batch(cols, bs)
    finally:
        L.h2hip_debug_set_msm_fuse_limits(ctypes.c_size_t(0), ctypes.c_size_t(0))
    for j in range(count):
        assert np.array_equal(aff(h2, got[j]), want[j]), j


@pytest.mark.parametrize("log_n", [20, 22, 26])
def test_streamed_msm_full_size_equals_device_resident(h2, log_n):
    """BASELINE.json configs[1] -- and the top of the metric range, 2^26 pairs: 2 GiB of scalars in three chunks of up to 34 M
    pairs -- through the host-pointer entry point with the default chunk ladder: same group element as the device-resident call
    (which test_msm_full_size_2p20 / test_msm_2p26_quarters_property tie to the oracle), pinned and unpinned"""
    n = 1 << log_n
    ds = h2.gen_scalars_device(0x5EED0001, n)
    dp = h2.gen_points_device(0x5EED0002, n)
    ref = aff(h2, h2.msm_device(ds, dp))
    sc, bs = h2.to_numpy_u64(ds).copy(), h2.to_numpy_u64(dp).copy()
    set_stream(h2, 0)
    if log_n <= 22:  # unpinned: the bases stream in too (at 2^26 that is 4 GiB per call: pinned only)
        assert np.array_equal(aff(h2, h2.best_multiexp(sc, bs)), ref)
    del ds, dp
    import torch
    torch.cuda.empty_cache()
    h2.bases_pin(bs)
    try:
        assert np.array_equal(aff(h2, h2.best_multiexp(sc, bs)), ref)
        set_stream(h2, 1)  # streaming off: the whole-upload path still agrees
        assert np.array_equal(aff(h2, h2.best_multiexp(sc, bs)), ref)
    finally:
        h2.bases_unpin(bs)


def test_streamed_msm_with_31bit_split(h2, oracle):
    """the 2^26-pair split (lowered to 9000 pairs) composes with streaming inside each piece"""
    n = 30011
    bs = oracle.gen_points(105, n, num_threads=NT)
    sc = oracle.gen_scalars(106, n, num_threads=NT)
    want = oracle.g1_to_affine(oracle.best_multiexp(sc, bs, NT))
    set_stream(h2, 3, 600, 1024)
    h2.lib().h2hip_debug_set_msm_max_chunk(ctypes.c_size_t(9000))
    try:
        both_forms(h2, sc, bs, want, "split")
    finally:
        h2.lib().h2hip_debug_set_msm_max_chunk(ctypes.c_size_t(0))


def test_streamed_msm_degenerate_inputs(h2, oracle):
    """all-zero scalars (no entry in any chunk), zero scalars in whole chunks only, identity points among the bases, and a size
    that is not a multiple of anything -- through the default ladder at a size that streams by itself (>= 2^19 pairs)"""
    n = (1 << 19) + 12345
    ds = h2.gen_scalars_device(0x5EED0001, n)
    dp = h2.gen_points_device(0x5EED0002, n)
    sc, bs = h2.to_numpy_u64(ds).copy(), h2.to_numpy_u64(dp).copy()
    zero = np.zeros_like(sc)
    ident = np.zeros(8, dtype=np.uint64)
    both_forms(h2, zero, bs, ident, "all-zero")
    tail = zero.copy()
    tail[-1000:] = sc[-1000:]  # only the last chunk has entries
    want = aff(h2, h2.msm_device(ds[-1000:], dp[-1000:]))
    both_forms(h2, tail, bs, want, "last-chunk-only")
    head = zero.copy()
    head[:1000] = sc[:1000]    # only the first chunk has entries: later chunks must leave the parts alone
    want = aff(h2, h2.msm_device(ds[:1000], dp[:1000]))
    both_forms(h2, head, bs, want, "first-chunk-only")
    bs2 = bs.copy()
    bs2[::7] = 0               # identity points (0, 0) among the bases
    keep = np.ones(n, dtype=bool)
    keep[::7] = False
    import torch
    idx = torch.from_numpy(np.nonzero(keep)[0]).cuda()
    want = aff(h2, h2.msm_device(ds[idx].contiguous(), dp[idx].contiguous()))
    both_forms(h2, sc, bs2, want, "identity-bases")


def test_host_pointer_entry_points_from_several_threads(h2, oracle):
    """What a rayon pool does to the drop-in (shplonk/prover.rs:180-196 commits from `par_iter`, permutation/keygen.rs:216-233
    transforms from several workers): host-pointer MSMs -- streamed, pinned and unpinned -- and host-pointer NTTs entered from
    four threads at once.  The engine serialises them per device; every result must be the serial one."""
    import threading
    n, k = (1 << 16) + 77, 14
    bs = oracle.gen_points(401, n, num_threads=NT)
    cols = [oracle.gen_scalars(410 + j, n, num_threads=NT) for j in range(4)]
    d = h2.EvaluationDomain.new(2, k)
    polys = [oracle.gen_scalars(420 + j, 1 << k, num_threads=NT) for j in range(4)]
    set_stream(h2, 3, 600, 4096)
    h2.bases_pin(bs)
    try:
        want_msm = [aff(h2, h2.best_multiexp(c_, bs)) for c_ in cols]
        want_ntt = []
        for p_ in polys:
            a = p_.copy()
            h2.best_fft(a, d.omega, k)
            want_ntt.append(a)
        assert np.array_equal(want_msm[0], oracle.g1_to_affine(oracle.best_multiexp(cols[0], bs, NT)))
        assert np.array_equal(want_ntt[0], oracle.best_fft(polys[0], d.omega, k, NT))
        bad = []

        def msm_worker(j):
            for it in range(6):
                if not np.array_equal(aff(h2, h2.best_multiexp(cols[j], bs)), want_msm[j]):
                    bad.append(("msm", j, it))

        def ntt_worker(j):
            for it in range(6):
                a = polys[j].copy()
                h2.best_fft(a, d.omega, k)
                if not np.array_equal(a, want_ntt[j]):
                    bad.append(("ntt", j, it))

        def unpinned_worker(j):
            other = np.ascontiguousarray(bs[::-1])
            w = oracle.g1_to_affine(oracle.best_multiexp(cols[j], other, 2))
            for it in range(3):
                if not np.array_equal(aff(h2, h2.best_multiexp(cols[j], other)), w):
                    bad.append(("unpinned", j, it))

        ths = [threading.Thread(target=msm_worker, args=(j,)) for j in range(4)]
        ths += [threading.Thread(target=ntt_worker, args=(j,)) for j in range(4)]
        ths += [threading.Thread(target=unpinned_worker, args=(0,))]
        for t_ in ths:
            t_.start()
        for t_ in ths:
            t_.join()
        assert not bad, bad[:5]
    finally:
        h2.bases_unpin(bs)


@pytest.mark.parametrize("env", [{"HALO2_HIP_STREAM": "0"}, {"HALO2_HIP_STREAM_MIN_N": "4096"}])
def test_streaming_environment_knobs(env):
    """HALO2_HIP_STREAM=0 (whole uploads, no copier thread) and HALO2_HIP_STREAM_MIN_N (threshold), read at init"""
    import subprocess
    import sys
    from conftest import ROOT
    code = r"""
import os, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
from conftest import load_pkg
from oracle import oracle
h2 = load_pkg()
h2.init()
n = 30000
sc = oracle.gen_scalars(7, n, num_threads=8); bs = oracle.gen_points(8, n, num_threads=8)
want = oracle.g1_to_affine(oracle.best_multiexp(sc, bs, 8))
ok = np.array_equal(h2.g1_to_affine(h2.best_multiexp(sc, bs)), want)
h2.bases_pin(bs)
ok = ok and np.array_equal(h2.g1_to_affine(h2.best_multiexp(sc, bs)), want)
print("RESULT", ok)
""" % (ROOT, ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stderr[-2000:]
    assert [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][-1].split()[1] == "True"
