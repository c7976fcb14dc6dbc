import sys, os, json
sys.path.insert(0, "/root/repo")
from __graft_entry__ import load_pkg
h2 = load_pkg(); h2.init(0)
import torch
out = {}
for log_n, count in ((17, 16), (17, 6), (15, 16), (13, 16), (18, 8)):
    n = 1 << log_n
    dp = h2.gen_points_device(1, n)
    cols = [h2.gen_scalars_device(10 + j, n) for j in range(count)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for c in (0, 10, 11, 12, 13, 14, 15, 16):
        h2.set_msm_window(c)
        if c and (254 // c + 1) * count > 4096:
            continue
        h2.msm_batch_device(cols, dp); torch.cuda.synchronize()
        e0.record()
        for _ in range(3):
            h2.msm_batch_device(cols, dp)
        e1.record(); torch.cuda.synchronize()
        out["2p%d_x%d_c%d" % (log_n, count, c)] = round(e0.elapsed_time(e1) / 3 / count, 4)
h2.set_msm_window(0)
print(json.dumps(out))
