import os, sys, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg
import numpy as np, torch
h2 = load_pkg(); h2.init(0)
L = h2.lib()
for k in (8, 12, 14, 16, 17, 18):
    n = 1 << k
    g = h2.gen_points_device(7, n)
    outs = {}
    for q in (0, 1):
        L.h2hip_debug_set_g2l_quad(q)
        res = torch.empty_like(g)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        def call():
            rc = L.h2hip_g_to_lagrange_bn254_device(ctypes.c_void_p(g.data_ptr()), ctypes.c_uint32(k), ctypes.c_void_p(res.data_ptr()), st)
            assert rc == 0
        call(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record(); torch.cuda.synchronize()
        outs[q] = (e0.elapsed_time(e1), res.clone())
    print("k=%d: lane %.2f ms, quad %.2f ms, equal=%s" % (k, outs[0][0], outs[1][0], bool(torch.equal(outs[0][1], outs[1][1]))), flush=True)
