import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg
h2 = load_pkg(); h2.init(0)
import torch
for ln in (17, 20, 22):
    n = 1 << ln
    ds = h2.gen_scalars_device(0x5EED0001, n); dp = h2.gen_points_device(0x5EED0002, n)
    sc, bs = h2.to_numpy_u64(ds).copy(), h2.to_numpy_u64(dp).copy()
    ref = h2.g1_to_affine(h2.msm_device(ds, dp))
    for pinned in (False, True):
        if pinned: h2.bases_pin(bs)
        h2.best_multiexp(sc, bs)
        t = []
        for _ in range(5):
            t0 = time.perf_counter(); out = h2.best_multiexp(sc, bs); t.append(time.perf_counter() - t0)
        assert np.array_equal(h2.g1_to_affine(out), ref)
        print("msm 2^%d host-pointer pinned_bases=%s: %.3f ms (min %.3f)" % (ln, pinned, 1e3 * sorted(t)[2], 1e3 * min(t)))
        if pinned: h2.bases_unpin(bs)
    t0 = time.perf_counter(); h2.msm_device(ds, dp); print("   device-resident: %.3f ms" % (1e3 * (time.perf_counter() - t0)))
from oracle import oracle
for k in (17, 19, 22):
    d, _ = oracle.domain_new(2, k)
    a = h2.to_numpy_u64(h2.gen_scalars_device(3, 1 << k)).copy()
    h2.best_fft(a, d.fe("omega"), k)
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); h2.best_fft(a, d.fe("omega"), k); t.append(time.perf_counter() - t0)
    print("ntt 2^%d host-pointer: %.3f ms" % (k, 1e3 * sorted(t)[2]))
