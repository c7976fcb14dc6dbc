"""Host-pointer MSM (h2hip_msm_bn254 / _batch, scalars in pageable host memory) against the device-resident call, over the
chunk ladder's knobs (h2hip_debug_set_msm_stream).  python3 tools/stream_sweep.py [log_n ...]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

h2 = load_pkg()
h2.init(0)
import torch  # noqa: E402

L = h2.lib()


def set_stream(chunks, permille=0, min_n=0):
    L.h2hip_debug_set_msm_stream(ctypes.c_uint32(chunks), ctypes.c_uint32(permille), ctypes.c_size_t(min_n))


def med(f, reps=7):
    f()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        t.append(time.perf_counter() - t0)
    return sorted(t)[len(t) // 2] * 1e3


def main():
    logs = [int(a) for a in sys.argv[1:]] or [20]
    for ln in (18, 19):  # where streaming starts to pay
        n = 1 << ln
        sc = h2.to_numpy_u64(h2.gen_scalars_device(0x5EED0001, n)).copy()
        bs = h2.to_numpy_u64(h2.gen_points_device(0x5EED0002, n)).copy()
        h2.bases_pin(bs)
        for chunks, pm in ((1, 0), (2, 600), (3, 600)):
            set_stream(chunks, pm, 1 << 16)
            print("2^%d pinned host-pointer chunks=%d ratio=%.2f: %.3f ms" % (ln, chunks, pm / 1000, med(lambda: h2.best_multiexp(sc, bs))), flush=True)
        h2.bases_unpin(bs)
        set_stream(0)
    for ln in logs:
        n = 1 << ln
        ds = h2.gen_scalars_device(0x5EED0001, n)
        dp = h2.gen_points_device(0x5EED0002, n)
        sc, bs = h2.to_numpy_u64(ds).copy(), h2.to_numpy_u64(dp).copy()
        ref = h2.g1_to_affine(h2.msm_device(ds, dp))
        print("2^%d plain device-resident: %.3f ms" % (ln, med(lambda: h2.msm_device(ds, dp))), flush=True)
        for chunks, pm in ((1, 0), (3, 600), (2, 600), (4, 600), (3, 500)):
            set_stream(chunks, pm)
            out = h2.best_multiexp(sc, bs)
            assert np.array_equal(h2.g1_to_affine(out), ref)
            print("2^%d unpinned host-pointer chunks=%d ratio=%.2f: %.3f ms" % (ln, 2 * chunks, 2 * pm / 1000, med(lambda: h2.best_multiexp(sc, bs), 5)), flush=True)
        h2.bases_pin(bs)
        h2.bases_pin_device(dp)
        print("2^%d fixed device-resident: %.3f ms" % (ln, med(lambda: h2.msm_device(ds, dp))), flush=True)
        h2.bases_unpin_device(dp)
        for chunks, pm in ((1, 0), (3, 600), (3, 700), (4, 600)):
            set_stream(chunks, pm)
            out = h2.best_multiexp(sc, bs)
            assert np.array_equal(h2.g1_to_affine(out), ref)
            print("2^%d pinned host-pointer chunks=%d ratio=%.2f: %.3f ms" % (ln, chunks, pm / 1000, med(lambda: h2.best_multiexp(sc, bs))), flush=True)
        h2.bases_unpin(bs)
        set_stream(0)
        del ds, dp, sc, bs
        torch.cuda.empty_cache()
    # the prover's batch: 16 columns of 2^17 in host memory over pinned bases
    n = 1 << 17
    cols = [h2.to_numpy_u64(h2.gen_scalars_device(0x5EED0001, n, start=(j + 1) * n)).copy() for j in range(16)]
    bs = h2.to_numpy_u64(h2.gen_points_device(0x5EED0002, n)).copy()
    h2.bases_pin(bs)
    for chunks, pm in ((1, 0), (3, 600), (3, 500), (3, 800), (2, 600), (2, 900), (4, 700)):
        set_stream(chunks, pm)
        print("16 x 2^17 host columns, pinned bases, chunks=%d ratio=%.2f: %.3f ms per batch" % (chunks, pm / 1000, med(lambda: h2.best_multiexp_batch(cols, bs), 5)),
              flush=True)
    set_stream(1)
    print("one 2^17 host column at a time: %.3f ms per MSM" % (med(lambda: [h2.best_multiexp(c_, bs) for c_ in cols], 3) / 16))
    set_stream(0, 0, 1 << 16)
    print("one 2^17 host column at a time, streamed: %.3f ms per MSM" % (med(lambda: [h2.best_multiexp(c_, bs) for c_ in cols], 3) / 16))
    h2.bases_unpin(bs)


if __name__ == "__main__":
    main()
