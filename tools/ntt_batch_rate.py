#!/usr/bin/python3
"""evaluate_h's column conversions as the library runs them: COLS coset NTTs 2^k -> 2^(k+2) in one batched call, against the same
number of plain forward NTTs of 2^(k+2) batched, and one of them alone.  ms per transform by HIP events.
  python3 tools/ntt_batch_rate.py [--k 18] [--cols 18]     (run on the GPU box; under tools/prof.sh for per-kernel times)"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=18)
    ap.add_argument("--cols", type=int, default=18)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--batch-mb", type=int, nargs="*", default=[0], help="columns + workspace per batched launch (0 = library default)")
    ap.add_argument("--log-j", type=int, default=-1, help="two-pass plan: log2 columns per workgroup (-1 = default)")
    ap.add_argument("--two-lo", type=int, default=0, help="smallest log2 size on the two-pass plan (0 = library default)")
    ap.add_argument("--two-wgs", type=int, default=0, help="workgroups per pass from which batches of 2^17..2^19 take the two-pass plan (0 = default)")
    ap.add_argument("--only", default="", help="comma list of legs: lone,batch,coset")
    args = ap.parse_args()
    import torch
    h2 = load_pkg()
    h2.init(0)
    k, ek = args.k, args.k + 2
    d = h2.EvaluationDomain.new(4, k)
    cols = [h2.gen_scalars_device(100 + i, 1 << ek) for i in range(args.cols)]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

    def timed(f):
        f()
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(args.reps):
            f()
        ev[1].record()
        torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]) / args.reps

    legs = {
        "lone": lambda: h2.ntt_device(cols[0], d.extended_omega, ek),
        "batch": lambda: h2.ntt_batch_device(cols, d.extended_omega, ek),
        "coset": lambda: h2.coeff_to_extended_batch_device(cols, k, ek, d.extended_omega, d.g_coset, d.g_coset_inv),
    }
    per = {"lone": 1, "batch": args.cols, "coset": args.cols}
    import ctypes
    h2.lib().h2hip_debug_set_ntt_two_pass_log_j(ctypes.c_int(args.log_j))
    h2.lib().h2hip_debug_set_ntt_two_pass_batch_wgs(ctypes.c_uint64(args.two_wgs))
    if args.two_lo:
        h2.lib().h2hip_debug_set_ntt_two_pass(ctypes.c_uint32(args.two_lo), ctypes.c_uint32(22))
    for mb in args.batch_mb:
        h2.lib().h2hip_debug_set_ntt_batch_bytes(ctypes.c_uint64(mb << 20))
        out = {"k": k, "extended_k": ek, "cols": args.cols, "batch_mb": mb}
        for name, f in legs.items():
            if args.only and name not in args.only.split(","):
                continue
            out[name + "_ms_per_transform"] = round(timed(f) / per[name], 4)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
