#!/usr/bin/python3
"""tools/msm_bench.py -- stage times of one device-resident MSM, plain form and fixed-base form (window table built by
h2hip_bases_pin_device), for a list of sizes and optional window overrides.  Prints one JSON object per line."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from __graft_entry__ import load_pkg  # noqa: E402

STAGES = ("msm_total", "msm_digits", "msm_sort", "msm_accum", "msm_reduce")  # over-full buckets are summed inside the accumulate launch


def time_msm(h2, ds, dp, reps, stages=True):
    for _ in range(2):
        r = h2.msm_device(ds, dp)
    if not stages:  # wall clock only: the stage timers put ~10 us of gap on the stream at every stage boundary
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = h2.msm_device(ds, dp)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, {}, r
    h2.profile_enable(True)
    h2.profile_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = h2.msm_device(ds, dp)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    h2.profile_enable(False)
    st = {}
    for s in STAGES:
        tot, cnt = h2.profile_get(s)
        st[s] = round(tot / cnt, 4) if cnt else None
    return ms, st, r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, nargs="+", default=[17, 20])
    ap.add_argument("--windows", type=int, nargs="*", default=[0], help="window overrides to try (0 = default)")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--no-plain", action="store_true")
    ap.add_argument("--no-fixed", action="store_true")
    ap.add_argument("--heavy-div", type=int, default=0)
    ap.add_argument("--bin-entries", type=int, default=0)
    ap.add_argument("--local-order", type=int, default=0)
    ap.add_argument("--quad-tail", type=int, default=1)
    ap.add_argument("--split", type=int, default=1)
    ap.add_argument("--rowcol-lanes", type=int, default=0)
    ap.add_argument("--rowcol-asm", type=int, default=3, help="first row/column pass: bit 0 explicit-mad multiplier, bit 1 quad tree (3 = library default)")
    ap.add_argument("--split-records", type=int, default=1, help="level-1 sort records as two arrays where bins span several tiles (library default 1)")
    ap.add_argument("--no-stages", action="store_true", help="wall clock only, no per-stage event timers")
    ap.add_argument("--prover-like", action="store_true", help="SURVEY.md 8(d): 90 %% zero, 5 %% in {1, 2}, 5 %% uniform (over-full buckets: the heavy path)")
    ap.add_argument("--sparse", type=int, default=0, help="all scalars zero but this many (examples/circuit-layout.rs at k = 17: ~26 assigned or blinding rows of 2^17)")
    ap.add_argument("--skew", action="store_true", help="prover-like scalars: 90 %% zero, 5 %% in {1, 2}, 5 %% uniform")
    args = ap.parse_args()
    h2 = load_pkg()
    h2.init(0)
    import ctypes
    h2.lib().h2hip_debug_set_msm_heavy_div(ctypes.c_size_t(args.heavy_div))
    h2.lib().h2hip_debug_set_msm_bin_entries(ctypes.c_size_t(args.bin_entries))
    h2.lib().h2hip_debug_set_msm_split_records(ctypes.c_int(args.split_records))
    h2.lib().h2hip_debug_set_msm_bucket_order(ctypes.c_int(args.local_order))
    h2.lib().h2hip_debug_set_msm_quad_tail(ctypes.c_int(args.quad_tail))
    h2.lib().h2hip_debug_set_msm_split_buckets(ctypes.c_int(args.split))
    h2.lib().h2hip_debug_set_msm_rowcol(ctypes.c_uint64(args.rowcol_lanes), ctypes.c_int(args.rowcol_asm))
    for lg in args.log_n:
        n = 1 << lg
        ds = h2.gen_scalars_device(0x5EED0001, n)
        dp = h2.gen_points_device(0x5EED0002, n)
        if args.skew:
            r = torch.rand(n, device="cuda")
            small = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
            ds = torch.where((r < 0.9)[:, None], small, ds)
        if args.sparse:
            keep = torch.zeros(n, dtype=torch.bool, device="cuda")
            keep[torch.randperm(n, device="cuda")[:args.sparse]] = True
            ds = torch.where(keep[:, None], ds.view(n, 4), torch.zeros_like(ds.view(n, 4)))
        if args.prover_like:
            r = torch.rand(n, device="cuda")
            one = torch.from_numpy(h2.fr_from_int(1).view(np.int64)).cuda()
            two = torch.from_numpy(h2.fr_from_int(2).view(np.int64)).cuda()
            ds = ds.view(n, 4).clone()
            ds[r < 0.9] = 0
            ds[(r >= 0.9) & (r < 0.925)] = one
            ds[(r >= 0.925) & (r < 0.95)] = two
        torch.cuda.synchronize()
        ref = None
        for c in args.windows:
            h2.set_msm_window(c)
            if not args.no_plain and not (c >= 20):
                ms, st, r = time_msm(h2, ds, dp, args.reps, not args.no_stages)
                aff = h2.g1_to_affine(r)
                ref = aff if ref is None else ref
                print(json.dumps({"form": "plain", "log_n": lg, "c": c or h2.get_msm_window(n), "ms": round(ms, 4), "stages": st,
                                  "same_result": bool(np.array_equal(aff, ref))}), flush=True)
            if not args.no_fixed:
                t0 = time.perf_counter()
                h2.bases_pin_device(dp)
                t_pin = time.perf_counter() - t0
                info = h2.bases_pinned_info(dp)
                ms, st, r = time_msm(h2, ds, dp, args.reps, not args.no_stages)
                aff = h2.g1_to_affine(r)
                ref = aff if ref is None else ref
                print(json.dumps({"form": "fixed-base", "log_n": lg, "c": info[1], "W": info[2], "table_MB": round(info[3] / 2**20, 1),
                                  "pin_s": round(t_pin, 3), "ms": round(ms, 4), "stages": st, "same_result": bool(np.array_equal(aff, ref))}),
                      flush=True)
                h2.bases_unpin_device(dp)
        h2.set_msm_window(0)
        del ds, dp


if __name__ == "__main__":
    main()
