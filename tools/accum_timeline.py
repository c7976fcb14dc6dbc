#!/usr/bin/python3
"""Per-wave timeline of msm_accum_kernel (a library built with -DH2_ACCUM_TIMELINE, passed as HALO2_HIP_LIB): every wave of the accumulate
role records wall_clock64 (100 MHz) at entry and exit, its first lane's bucket size and HW_ID.  Prints how many waves are resident over
the kernel's duration and per SIMD.   HALO2_HIP_LIB=ab/libtl.so python3 tools/accum_timeline.py [log_n]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_pkg
h2 = load_pkg(); h2.init(0)
import torch
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
ds, dp = h2.gen_scalars_device(0x5EED0001, n), h2.gen_points_device(0x5EED0002, n)
h2.bases_pin_device(dp)
for _ in range(5):
    h2.msm_device(ds, dp)
torch.cuda.synchronize()
L = h2.lib()
N = 16384
buf = np.zeros(6 * N, dtype=np.uint64)
rc = L.h2hip_debug_accum_timeline(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(6 * N))
assert rc == 0, rc
a = buf.reshape(N, 6)
a = a[a[:, 1] > 0]
t0 = a[:, 0].min()
st, en = (a[:, 0] - t0).astype(np.int64), (a[:, 1] - t0).astype(np.int64)  # 10 ns ticks
print("waves %d, kernel span %.1f us" % (len(a), en.max() / 100.0))
span = en.max()
edges = np.linspace(0, span, 41)
res = []
for lo, hi in zip(edges[:-1], edges[1:]):
    ov = np.clip(np.minimum(en, hi) - np.maximum(st, lo), 0, None).sum() / (hi - lo)
    res.append(ov)
print("resident waves by 2.5 % slices of the span (of 4096 slots):")
print(" ".join("%4d" % r for r in res))
print("mean resident %.0f = %.2f per SIMD" % (np.mean(res), np.mean(res) / 1024))
dur = (en - st) / 100.0
cnt = a[:, 2].astype(np.int64)
print("wave duration us: min %.0f median %.0f max %.0f; first-lane bucket size min %d median %d max %d" % (dur.min(), np.median(dur), dur.max(), cnt.min(), np.median(cnt), cnt.max()))
per = dur / np.maximum(cnt, 1)
print("us per addition (duration / size): median %.2f; first 256 waves %.2f, last 256 waves %.2f" % (np.median(per), np.median(per[:256]), np.median(per[-256:])))
order = np.argsort(st)
print("start times us (every 512th wave in start order):", " ".join("%.0f" % (st[order][i] / 100.0) for i in range(0, len(a), 512)))
print("end times us, percentiles 1 10 50 90 99 100:", " ".join("%.0f" % (np.percentile(en, q) / 100.0) for q in (1, 10, 50, 90, 99, 100)))
hw = a[:, 3] & 0xffffffff
xcc = (a[:, 3] >> 32) & 0xf
simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = ((xcc * 8 + se) * 2 + sh) * 16 * 4 + cu * 4 + simd
u, c_ = np.unique(key, return_counts=True)
print("distinct SIMDs seen %d; waves per SIMD min %d median %d max %d" % (len(u), c_.min(), np.median(c_), c_.max()))
busy = np.zeros(len(u))
for i, k in enumerate(u):
    m = key == k
    busy[i] = (en[m] - st[m]).sum() / span
print("wave-residency per SIMD (sum of wave durations / span): min %.2f median %.2f max %.2f" % (busy.min(), np.median(busy), busy.max()))
# shader clock over the kernel: cycles of clock64 per 10-ns tick of wall_clock64, per wave, against the time the wave ended
ghz = (a[:, 5] - a[:, 4]).astype(np.float64) / np.maximum((a[:, 1] - a[:, 0]).astype(np.float64), 1.0) / 10.0
o = np.argsort(en)
q = len(o) // 8
print("shader clock GHz (clock64 / wall_clock64 over each wave's life), by octile of end time:", " ".join("%.2f" % np.median(ghz[o[i * q:(i + 1) * q]]) for i in range(8)))
late = st > np.percentile(st, 90)
print("waves that started in the last tenth: clock %.2f GHz, us per addition %.2f; waves that ended in the first tenth: clock %.2f GHz, us per addition %.2f" % (
    np.median(ghz[late]), np.median(per[late]), np.median(ghz[en < np.percentile(en, 10)]), np.median(per[en < np.percentile(en, 10)])))
# how the resident waves spread over the SIMDs as the grid drains: histogram of waves per SIMD at a few instants
for frac in (0.3, 0.6, 0.75, 0.8, 0.85, 0.9, 0.95):
    T = frac * span
    m = (st <= T) & (en > T)
    per_simd = np.zeros(len(u), dtype=np.int64)
    idx = np.searchsorted(u, key[m])
    np.add.at(per_simd, idx, 1)
    print("t = %.0f us: %d waves; SIMDs holding 0 / 1 / 2 / 3 / 4+ waves: %s" % (T / 100.0, m.sum(), " ".join(str(int((per_simd == k).sum())) if k < 4 else str(int((per_simd >= 4).sum())) for k in range(5))))
