"""Per-repetition times (ms, each between its own events) of the forward and the scaled inverse transform of 2^k points, four rounds of 24: shows the
clock ramp of the first repetitions and whether any later one stands out.   python3 tools/ntt_diag.py [k]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg
h2 = load_pkg(); h2.init(0)
import torch
k = int(sys.argv[1]) if len(sys.argv) > 1 else 24
d = h2.EvaluationDomain.new(2, k)
a = h2.gen_scalars_device(3, 1 << k)
def per_rep(f, reps):
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1))
    return out
fwd = lambda: h2.ntt_device(a, d.omega, k)
inv = lambda: h2.ifft_device(a, d.omega_inv, k, d.ifft_divisor)
for rnd in range(4):
    for name, f in (("fwd", fwd), ("inv", inv)):
        t = per_rep(f, 24)
        print(rnd, name, " ".join("%.2f" % x for x in t), flush=True)
print("mem", torch.cuda.mem_get_info())
