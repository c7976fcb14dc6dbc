import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg
h2 = load_pkg(); h2.init(0)
import torch
for k in (12, 14, 16, 17, 18, 19, 20, 22, 24, 26):
    d = h2.EvaluationDomain.new(2, k)
    da = h2.gen_scalars_device(3, 1 << k)
    ref = None
    line = "2^%d:" % k
    for smax in (7, 8, 9, 10):
        h2.lib().h2hip_debug_set_ntt_smax(ctypes.c_uint32(smax))
        x = da.clone()
        h2.ntt_device(x, d.omega, k); torch.cuda.synchronize()
        if ref is None: ref = x.clone()
        assert torch.equal(x, ref), (k, smax)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10 if k <= 22 else 3
        e0.record()
        for _ in range(reps): h2.ntt_device(x, d.omega, k)
        e1.record(); torch.cuda.synchronize()
        line += "  smax=%d %.4f ms" % (smax, e0.elapsed_time(e1) / reps)
    print(line, flush=True)
