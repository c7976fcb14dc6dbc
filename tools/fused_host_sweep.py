#!/usr/bin/python3
"""Host-column commit batches (h2hip_msm_bn254_batch over pinned bases, 2^17 pairs per column): the group ladder of msm_fused_groups_host
swept over chunk counts and ratios, against the whole upload first and against device-resident columns.   python tools/fused_host_sweep.py"""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg  # noqa: E402

h2 = load_pkg()
h2.init(0)
L = h2.lib()
n = 1 << 17
dcols = [h2.gen_scalars_device(0x5EED0001, n, start=(j + 1) * n) for j in range(16)]
cols = [h2.to_numpy_u64(c).copy() for c in dcols]
dbs = h2.gen_points_device(0x5EED0002, n)
bs = h2.to_numpy_u64(dbs).copy()
h2.bases_pin(bs)
h2.bases_pin_device(dbs)


def med(f, reps=9):
    f()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        t.append(time.perf_counter() - t0)
    return sorted(t)[len(t) // 2] * 1e3


out = {}
for cnt in (6, 10, 16):
    ent = {"device_resident_ms": med(lambda: h2.msm_batch_device(dcols[:cnt], dbs))}
    for chunks, ratio in ((1, 600), (2, 600), (2, 400), (3, 600), (3, 400), (3, 300), (4, 600), (4, 400)):
        L.h2hip_debug_set_msm_stream(ctypes.c_uint32(chunks), ctypes.c_uint32(ratio), ctypes.c_size_t(0))
        ent["chunks%d_ratio%d" % (chunks, ratio)] = med(lambda: h2.best_multiexp_batch(cols[:cnt], bs))
    L.h2hip_debug_set_msm_stream(ctypes.c_uint32(0), ctypes.c_uint32(0), ctypes.c_size_t(0))
    ent["default"] = med(lambda: h2.best_multiexp_batch(cols[:cnt], bs))
    out["%d_columns" % cnt] = ent
print(json.dumps(out))
