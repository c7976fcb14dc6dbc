#!/usr/bin/python3
"""g_to_lagrange (arithmetic.rs:277-301) timing: device-resident call by HIP events, the oracle on the host cores beside it
at a size it finishes in seconds.   python tools/g2l_bench.py --k 16 [--cpu-k 12]   (run on the GPU box)"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402


def bench(args):
    """args: namespace with k (list), cpu_k, threads -> result dict"""
    import torch
    h2 = load_pkg()
    from oracle import oracle as orc
    orc.build()
    L = h2.lib()
    out = {}
    for k in args.k:
        n = 1 << k
        g = h2.gen_points_device(7, n)
        res = torch.empty_like(g)
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

        def call():
            rc = L.h2hip_g_to_lagrange_bn254_device(ctypes.c_void_p(g.data_ptr()), ctypes.c_uint32(k), ctypes.c_void_p(res.data_ptr()), stream)
            assert rc == 0, L.h2hip_last_error()

        call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        r = {"gpu_ms": ms, "scalar_muls_per_s": (n / 2 * max(k - 1, 0) + n) / (ms * 1e-3)}
        if k == args.cpu_k:
            gh = h2.to_numpy_u64(g)
            t0 = time.perf_counter()
            want = orc.g_to_lagrange(gh, k, num_threads=args.threads)
            r["oracle_s"] = time.perf_counter() - t0
            r["oracle_threads"] = args.threads
            r["match"] = bool(np.array_equal(h2.to_numpy_u64(res), want))
            assert r["match"]
        out["k%d" % k] = r
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, nargs="+", default=[12, 16])
    ap.add_argument("--cpu-k", type=int, default=12)
    ap.add_argument("--threads", type=int, default=16)
    print(json.dumps(bench(ap.parse_args())))


if __name__ == "__main__":
    main()
