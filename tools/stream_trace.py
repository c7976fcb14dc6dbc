"""A few host-pointer MSMs with the chunk ladder, for a rocprofv3 --kernel-trace --memory-copy-trace timeline:
python3 tools/stream_trace.py LOG_N CHUNKS PERMILLE [unpinned]; tools/timeline.py prints the trace of the last call."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

h2 = load_pkg()
h2.init(0)
ln, chunks, pm = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = 1 << ln
sc = h2.to_numpy_u64(h2.gen_scalars_device(0x5EED0001, n)).copy()
bs = h2.to_numpy_u64(h2.gen_points_device(0x5EED0002, n)).copy()
if len(sys.argv) < 5:
    h2.bases_pin(bs)
h2.lib().h2hip_debug_set_msm_stream(ctypes.c_uint32(chunks), ctypes.c_uint32(pm), ctypes.c_size_t(0))
for _ in range(4):
    h2.best_multiexp(sc, bs)
