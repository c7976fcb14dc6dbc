import sys, os, time, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg
h2 = load_pkg(); h2.init(0)
n = 1 << 17
cols = [h2.to_numpy_u64(h2.gen_scalars_device(0x5EED0001, n, start=(j + 1) * n)).copy() for j in range(16)]
bs = h2.to_numpy_u64(h2.gen_points_device(0x5EED0002, n)).copy()
h2.bases_pin(bs)
def med(f, reps=9):
    f(); t = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); t.append(time.perf_counter() - t0)
    return sorted(t)[len(t) // 2] * 1e3
print("16 x 2^17 host columns: %.3f ms per batch" % med(lambda: h2.best_multiexp_batch(cols, bs)))
n = 1 << 15
cols = [c_[:n].copy() for c_ in cols]
print("16 x 2^15 host columns (prefix): %.3f ms per batch" % med(lambda: h2.best_multiexp_batch(cols, bs[:n])))
