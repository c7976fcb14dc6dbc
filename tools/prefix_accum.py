"""Accumulate time of an MSM over the first m pairs of a 2^20-point window table (c = 20, 2^19 buckets): how the
dominant kernel's rate depends on the entries per bucket.  python3 tools/prefix_accum.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

h2 = load_pkg()
h2.init(0)
import torch  # noqa: E402

n = 1 << 20
ds = h2.gen_scalars_device(0x5EED0001, n)
dp = h2.gen_points_device(0x5EED0002, n)
h2.bases_pin_device(dp)
for m in (n, n * 3 // 4, n // 2, n // 4, n // 8, n // 16):
    for _ in range(3):
        h2.msm_device(ds[:m], dp, n=m)
    h2.profile_enable(True)
    h2.profile_reset()
    for _ in range(10):
        h2.msm_device(ds[:m], dp, n=m)
    torch.cuda.synchronize()
    h2.profile_enable(False)
    st = {s: h2.profile_get(s) for s in ("msm_total", "msm_digits", "msm_sort", "msm_accum", "msm_reduce")}
    acc = st["msm_accum"][0] / st["msm_accum"][1]
    print("m = %7d (%.3f of n): total %.3f digits %.3f sort %.3f accum %.3f reduce %.3f ms; accum per unit %.3f ms" % (
        m, m / n, *(st[s][0] / st[s][1] for s in ("msm_total", "msm_digits", "msm_sort")), acc, st["msm_reduce"][0] / st["msm_reduce"][1], acc * n / m), flush=True)
