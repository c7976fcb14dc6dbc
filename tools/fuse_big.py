import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg
import numpy as np, torch
h2 = load_pkg(); h2.init(0)
for lg, B in ((19, 8), (20, 8), (20, 4)):
    n = 1 << lg
    dp = h2.gen_points_device(0x5EED0002, n)
    cols = [h2.gen_scalars_device(100 + j, n) for j in range(B)]
    for pinned in (False, True):
        if pinned: h2.bases_pin_device(dp)
        res = {}
        for name, ent, mx in (("pipelined", 0, 0), ("fused", 1 << 28, 1 << 20)):
            h2.lib().h2hip_debug_set_msm_fuse_limits(ctypes.c_size_t(ent), ctypes.c_size_t(mx))
            out = h2.msm_batch_device(cols, dp); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3): out = h2.msm_batch_device(cols, dp)
            torch.cuda.synchronize()
            res[name] = ((time.perf_counter() - t0) / 3 / B * 1e3, [h2.g1_to_affine(o).tolist() for o in out])
        same = res["pipelined"][1] == res["fused"][1]
        print("2^%d x %d pinned=%s: pipelined %.3f ms/MSM, fused %.3f ms/MSM, same=%s" % (lg, B, pinned, res["pipelined"][0], res["fused"][0], same), flush=True)
        if pinned: h2.bases_unpin_device(dp)
    h2.lib().h2hip_debug_set_msm_fuse_limits(ctypes.c_size_t(0), ctypes.c_size_t(0))
