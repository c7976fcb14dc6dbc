#!/usr/bin/python3
"""Host-pointer transforms, PCIe included: one call per column (h2hip_ntt_bn254_fr) against the pipelined batch
(h2hip_ntt_bn254_fr_batch: upload i + 1 | transform i | download i - 1).  Columns are pageable numpy arrays, as a prover's Vec<F>.
  python tools/host_ntt_batch.py [log_n ...]      (run on the GPU box)"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402


def bench(h2, log_ns=(17, 20, 22), counts=(1, 2, 4, 8), reps=5):
    import numpy as np
    L = h2.lib()
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    out = {}
    for k in log_ns:
        d = h2.EvaluationDomain.new(2, k)
        n = 1 << k
        src = h2.to_numpy_u64(h2.gen_scalars_device(0x5EED0003, n)).copy()
        ent = {"mib_per_column": n * 32 / 2**20}
        cols = [src.copy() for _ in range(max(counts))]

        def med(f):
            f()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                f()
                ts.append((time.perf_counter() - t0) * 1e3)
            return sorted(ts)[len(ts) // 2]

        ent["one_call_ms"] = med(lambda: L.h2hip_ntt_bn254_fr(P(cols[0]), P(d.omega), ctypes.c_uint32(k)))
        for cnt in counts:
            if cnt < 2:
                continue
            ptrs = (ctypes.c_void_p * cnt)(*[c.ctypes.data for c in cols[:cnt]])
            t = med(lambda: L.h2hip_ntt_bn254_fr_batch(ptrs, ctypes.c_size_t(cnt), P(d.omega), ctypes.c_uint32(k)))
            ent["batch_%d_ms_per_column" % cnt] = t / cnt
        out["2^%d" % k] = ent
        del cols, src
    return out


if __name__ == "__main__":
    h2 = load_pkg()
    h2.init()
    ks = [int(a) for a in sys.argv[1:]] or [17, 20, 22]
    print(json.dumps(bench(h2, ks)))
