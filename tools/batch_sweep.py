import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg
h2 = load_pkg(); h2.init(0)
import torch, numpy as np
for ln in (17, 20, 22):
    n = 1 << ln
    dp = h2.gen_points_device(0x5EED0002, n)
    cols = [h2.gen_scalars_device(0x5EED0001, n, start=j * n) for j in range(8)]
    ref = None
    for K in (0, 16, 32, 48, 64, 96):
        h2.lib().h2hip_debug_set_reserved_cus(ctypes.c_uint32(K))
        out = h2.msm_batch_device(cols, dp); torch.cuda.synchronize()
        if ref is None: ref = out
        assert np.array_equal(np.stack([h2.g1_to_affine(o) for o in out]), np.stack([h2.g1_to_affine(o) for o in ref]))
        t0 = time.perf_counter()
        for _ in range(3): h2.msm_batch_device(cols, dp)
        torch.cuda.synchronize()
        print("2^%d reserved_cus=%3d: %.3f ms per MSM" % (ln, K, (time.perf_counter() - t0) / 24 * 1e3), flush=True)
