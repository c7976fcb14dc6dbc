import sys, os
sys.path.insert(0, "/root/repo")
from __graft_entry__ import load_pkg
h2 = load_pkg(); h2.init(0)
import torch
n = 1 << 17
dp = h2.gen_points_device(1, n)
cols = [h2.gen_scalars_device(10 + j, n) for j in range(16)]
h2.msm_batch_device(cols, dp); torch.cuda.synchronize()
for _ in range(3):
    h2.msm_batch_device(cols, dp)
torch.cuda.synchronize()
