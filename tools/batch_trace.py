import sys, time, ctypes, os
sys.path.insert(0, "/root/repo")
from __graft_entry__ import load_pkg
h2 = load_pkg(); h2.init(0)
import torch
n = 1 << 20
h2.lib().h2hip_debug_set_room_lds(ctypes.c_uint32(int(os.environ.get("ROOM_LDS", "0"))))
dp = h2.gen_points_device(0x5EED0002, n)
cols = [h2.gen_scalars_device(0x5EED0001, n, start=j * n) for j in range(6)]
h2.msm_batch_device(cols, dp); torch.cuda.synchronize()
h2.msm_batch_device(cols, dp); torch.cuda.synchronize()
