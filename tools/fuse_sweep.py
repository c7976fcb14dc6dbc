#!/usr/bin/python3
"""Batches of 8 fixed-base MSMs, fused into one run against pipelined over streams, per size (the library fuses up to 2^19 pairs).
  python3 tools/fuse_sweep.py [log_n ...]     (run on the GPU box)"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg  # noqa: E402

h2 = load_pkg()
h2.init(0)
import numpy as np  # noqa: E402
import torch  # noqa: E402

L = h2.lib()
for ln in [int(a) for a in sys.argv[1:]] or [16, 17, 18, 19, 20]:
    n = 1 << ln
    dp = h2.gen_points_device(0x5EED0002, n)
    h2.bases_pin_device(dp)
    cols = [h2.gen_scalars_device(0x5EED0001, n, start=j * n) for j in range(8)]
    ref = None
    res = {}
    for name, fuse in (("fused", 1), ("pipelined", 0), ("fused", 1), ("pipelined", 0)):
        L.h2hip_debug_set_msm_fuse_small(ctypes.c_int(fuse))
        L.h2hip_debug_set_msm_fuse_limits(ctypes.c_size_t(0), ctypes.c_size_t(1 << 20))
        out = h2.msm_batch_device(cols, dp)
        aff = np.stack([h2.g1_to_affine(o) for o in out])
        ref = aff if ref is None else ref
        assert np.array_equal(aff, ref)
        for _ in range(3):
            h2.msm_batch_device(cols, dp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            h2.msm_batch_device(cols, dp)
        torch.cuda.synchronize()
        res.setdefault(name, []).append(round((time.perf_counter() - t0) / 80 * 1e3, 4))
    L.h2hip_debug_set_msm_fuse_small(ctypes.c_int(1))
    L.h2hip_debug_set_msm_fuse_limits(ctypes.c_size_t(0), ctypes.c_size_t(0))
    print("8 x 2^%d fixed-base, ms per MSM: %s" % (ln, res), flush=True)
    h2.bases_unpin_device(dp)
