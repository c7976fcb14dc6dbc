"""Entry points from several Python threads at once (per-device locks): python3 tools/thread_check.py [dup|one] [both|ntt|msm]"""
import os, sys, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_pkg
h2 = load_pkg()
import torch
mode = sys.argv[1] if len(sys.argv) > 1 else "dup"
if mode == "dup":
    h2.init([0, 0])
else:
    h2.init(0)
d_big = [h2.gen_scalars_device(77 + j, 1 << 16) for j in range(4)]
dp_t = h2.gen_points_device(78, 1 << 14)
ds_t = [h2.gen_scalars_device(79 + j, 1 << 14) for j in range(4)]
exp_ntt = []
dom16 = h2.EvaluationDomain.new(2, 16)
for t_ in d_big:
    c_ = t_.clone()
    h2.ntt_device(c_, dom16.omega, 16)
    exp_ntt.append(h2.to_numpy_u64(c_).copy())
exp_msm = [h2.g1_to_affine(h2.msm_device(s_, dp_t)) for s_ in ds_t]
which = sys.argv[2] if len(sys.argv) > 2 else "both"
bad = []
def work_ntt(j):
    for it in range(int(os.environ.get("ITERS", "20"))):
        c_ = d_big[j].clone()
        h2.ntt_device(c_, dom16.omega, 16)
        torch.cuda.synchronize()
        if not np.array_equal(h2.to_numpy_u64(c_), exp_ntt[j]):
            bad.append(("ntt", j, it))
def work_msm(j):
    for it in range(int(os.environ.get("ITERS", "20"))):
        try:
            r = h2.msm_device(ds_t[j], dp_t)
        except Exception as e:
            bad.append(("msm-exc", j, it, str(e)))
            continue
        if not np.array_equal(h2.g1_to_affine(r), exp_msm[j]):
            bad.append(("msm", j, it))
ths = []
if which in ("both", "ntt"):
    ths += [threading.Thread(target=work_ntt, args=(j,)) for j in range(4)]
if which in ("both", "msm"):
    ths += [threading.Thread(target=work_msm, args=(j,)) for j in range(4)]
for t_ in ths: t_.start()
for t_ in ths: t_.join()
print(mode, which, "bad:", bad[:10], len(bad))
