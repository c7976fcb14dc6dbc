#!/usr/bin/python3
"""Forward transform, scaled inverse and the two alternating (as bench.py's ntt leg runs them), ms per transform by HIP events.
  python3 tools/ntt_fwd_inv.py [log_n ...]     (run on the GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg  # noqa: E402

h2 = load_pkg()
h2.init(0)
import torch  # noqa: E402


def timed(f, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


if "--after-msm" in sys.argv:  # as bench.py orders its legs: MSM work (tables, workspaces, a warm chip) before the transforms
    sys.argv.remove("--after-msm")
    ds, dp = h2.gen_scalars_device(0x5EED0001, 1 << 20), h2.gen_points_device(0x5EED0002, 1 << 20)
    h2.bases_pin_device(dp)
    print("2^20 fixed-base MSM first: %.4f ms" % timed(lambda: h2.msm_device(ds, dp), 200), flush=True)

for k in [int(a) for a in sys.argv[1:]] or [20, 22]:
    d = h2.EvaluationDomain.new(2, k)
    a = h2.gen_scalars_device(3, 1 << k)
    fwd = lambda: h2.ntt_device(a, d.omega, k)  # noqa: E731
    inv = lambda: h2.ifft_device(a, d.omega_inv, k, d.ifft_divisor)  # noqa: E731

    def both():
        fwd()
        inv()
    for rnd in range(2):
        print("2^%d: forward %.4f  inverse %.4f  alternating %.4f ms per transform" % (k, timed(fwd), timed(inv), timed(both) / 2), flush=True)
