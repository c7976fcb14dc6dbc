#!/usr/bin/python3
"""Column conversions of the SURVEY.md 3.4 call trace (k = 17: 10 iNTT 2^17, 10 coset NTT 2^17 -> 2^19), one call per column vs
the batched entry points.  python tools/ntt_batch_bench.py [--k 17] [--cols 10]   (run on the GPU box)"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, nargs="+", default=[14, 17, 20])
    ap.add_argument("--cols", type=int, default=10)
    args = ap.parse_args()
    import torch
    h2 = load_pkg()
    from oracle import oracle as orc
    orc.build()
    out = {}
    for k in args.k:
        d, _ = orc.domain_new(4, k)
        ek = d.extended_k
        cols = [h2.gen_scalars_device(100 + i, 1 << ek) for i in range(args.cols)]
        small = [c[:1 << k] for c in cols]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

        def timed(f, reps=5):
            f()
            torch.cuda.synchronize()
            ev[0].record()
            for _ in range(reps):
                f()
            ev[1].record()
            torch.cuda.synchronize()
            return ev[0].elapsed_time(ev[1]) / reps

        def ifft_single():
            for c in small:
                h2.ifft_device(c, d.fe("omega_inv"), k, d.fe("ifft_divisor"))

        def ext_single():
            for c in cols:
                h2.coeff_to_extended_device(c, k, ek, d.fe("extended_omega"), d.fe("g_coset"), d.fe("g_coset_inv"))

        out["k%d" % k] = {
            "cols": args.cols, "extended_k": ek,
            "ifft_single_ms": timed(ifft_single),
            "ifft_batch_ms": timed(lambda: h2.ifft_batch_device(small, d.fe("omega_inv"), k, d.fe("ifft_divisor"))),
            "coeff_to_extended_single_ms": timed(ext_single),
            "coeff_to_extended_batch_ms": timed(lambda: h2.coeff_to_extended_batch_device(cols, k, ek, d.fe("extended_omega"), d.fe("g_coset"),
                                                                                           d.fe("g_coset_inv"))),
        }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
