"""The two plans of ntt.hip on one box: three passes of 2^6..2^8-point tiles against two of 2^9..2^11, with and
without the full inter-pass twiddle table (same limbs; ms per transform).
KS=18,20,22 picks the sizes."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg
h2 = load_pkg(); h2.init(0)
import torch
lib = h2.lib()


def timed(f, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    f(); torch.cuda.synchronize()
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for k in [int(v) for v in os.environ.get("KS", "18,19,20,21,22").split(",")]:
    d = h2.EvaluationDomain.new(2, k)
    da = h2.gen_scalars_device(3, 1 << k)
    out = {}
    for name, (lo, hi, budget) in (("three", (1, 0, 0)), ("two", (18, 22, 0)), ("three+table", (1, 0, 1 << 30)), ("two+table", (18, 22, 1 << 30))):
        lib.h2hip_debug_set_ntt_two_pass(ctypes.c_uint32(lo), ctypes.c_uint32(hi))
        lib.h2hip_debug_set_ntt_twiddle_budget(ctypes.c_uint64(budget))
        x = da.clone(); h2.ntt_device(x, d.omega, k)
        y = da.clone(); h2.ntt_device(y, d.omega_inv, k)
        torch.cuda.synchronize()
        out[name] = (x, y, timed(lambda: h2.ntt_device(x, d.omega, k), 20))
    lib.h2hip_debug_set_ntt_two_pass(ctypes.c_uint32(0), ctypes.c_uint32(0))
    ok = all(torch.equal(v[0], out["three"][0]) and torch.equal(v[1], out["three"][1]) for v in out.values())
    print("2^%d: " % k + "  ".join("%s %.4f" % (n, v[2]) for n, v in out.items()) + "  same=%s" % ok, flush=True)
