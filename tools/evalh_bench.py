#!/usr/bin/python3
"""evaluate_h (plonk/evaluation.rs:280-522) on device-resident columns: time per call by HIP events, and the oracle on the
host cores beside it.  The constraint system is synthetic but shaped like a mid-sized PLONKish circuit: `--gates` degree-4
gate polynomials over `--advice` advice and `--fixed` fixed columns with rotations in [-2, 2], a permutation over
`--perm` columns, `--lookups` lookups.  Columns come from the engine's on-device generator.

  python tools/evalh_bench.py --k 18 [--check-k 12]     (run on the GPU box)
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_pkg  # noqa: E402

R_MOD = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001


def build_system(ev, args, rng):
    A = lambda c, r=0: ('advice', c, r)  # noqa: E731
    F = lambda c, r=0: ('fixed', c, r)  # noqa: E731
    gates = []
    for gi in range(args.gates):
        a, b, c, d = (int(x) for x in rng.integers(0, args.advice, 4))
        f1, f2, f3 = (int(x) for x in rng.integers(0, args.fixed, 3))
        r1, r2 = (int(x) for x in rng.integers(-2, 3, 2))
        gates.append(('prod', F(f1), ('sum', ('sum', ('prod', ('prod', A(a, r1), A(b)), A(c)), ('neg', ('prod', A(d, r2), F(f2)))),
                                      ('scaled', F(f3), 3 + gi))))
    lookups = [([A(li % args.advice), ('prod', A((li + 1) % args.advice), F(li % args.fixed))], [F((li + 1) % args.fixed), F((li + 2) % args.fixed, 1)])
               for li in range(args.lookups)]
    return ev.Evaluator.new(gates, lookups)


def bench(args):
    """args: namespace with k, check_k, gates, advice, fixed, perm, lookups, iters, cpu -> result dict"""
    import importlib
    import torch
    h2 = load_pkg()
    ev = importlib.import_module("halo2_pse_amd.evaluation")
    from oracle import oracle as orc
    orc.build()
    L = h2.lib()
    rng = np.random.default_rng(1)
    evaluator = build_system(ev, args, rng)
    chunk_len = 2                      # cs.degree() = 4
    n_sets = -(-args.perm // chunk_len)
    out = {"gates": args.gates, "advice": args.advice, "fixed": args.fixed, "perm_columns": args.perm, "lookups": args.lookups,
           "custom_calcs": len(evaluator.custom_gates.calculations), "custom_intermediates": evaluator.custom_gates.num_intermediates,
           "custom_rotations": len(evaluator.custom_gates.rotations)}

    def run(k, check, time_cpu, host_timing=False):
        ek = k + 2
        n, size = 1 << k, 1 << ek
        d, _ = orc.domain_new(4, k)
        gen = lambda m, s: h2.gen_scalars_device(1000 + s, m)  # noqa: E731
        fe1 = lambda s: orc.gen_scalars(s, 1)[0]  # noqa: E731
        cols = {
            "fixed_cosets": [gen(size, 10 + i) for i in range(args.fixed)],
            "advice_polys": [gen(n, 100 + i) for i in range(args.advice)],
            "instance_polys": [],
            "perm_product_cosets": [gen(size, 200 + i) for i in range(n_sets)],
            "perm_cosets": [gen(size, 300 + i) for i in range(args.perm)],
            "l0": gen(size, 1), "l_last": gen(size, 2), "l_active_row": gen(size, 3),
            "lookups": [[gen(n, 400 + 3 * i + t) for t in range(3)] for i in range(args.lookups)],
        }
        d_values = gen(size, 999)
        v0 = d_values.clone()
        torch.cuda.synchronize()
        host = lambda t: h2.to_numpy_u64(t)  # noqa: E731
        scal = {"extended_omega": d.fe("extended_omega"), "g_coset": d.fe("g_coset"), "g_coset_inv": d.fe("g_coset_inv"),
                "zeta": orc.constant(orc.FR, 5), "delta": orc.fe_from_int(orc.FR, pow(7, 1 << 28, R_MOD)),
                "y": fe1(1), "beta": fe1(2), "gamma": fe1(3), "theta": fe1(4)}
        perm_kind = np.zeros(args.perm, dtype=np.uint32)
        perm_index = (np.arange(args.perm) % args.advice).astype(np.uint32)
        # a host description with tiny stand-in columns gives the struct; device addresses are swapped in below
        stub = np.zeros((1, 4), dtype=np.uint64)
        case = {"k": k, "extended_k": ek, **scal, "l0": stub, "l_last": stub, "l_active_row": stub,
                "fixed_cosets": [stub] * args.fixed, "advice_polys": [stub] * args.advice, "instance_polys": [],
                "challenges": np.zeros((0, 4), dtype=np.uint64),
                "perm_product_cosets": [stub] * n_sets, "perm_cosets": [stub] * args.perm, "perm_column_kind": perm_kind, "perm_column_index": perm_index,
                "chunk_len": chunk_len, "last_rotation": -6, "lookups": [(stub, stub, stub)] * args.lookups}
        hd = evaluator.describe(case)
        keep = []

        def table(ts):
            arr = (ctypes.c_void_p * max(1, len(ts)))(*[t.data_ptr() for t in ts])
            keep.append(arr)
            return ctypes.addressof(arr)

        dd = hd.desc
        dd.fixed_cosets, dd.advice_polys = table(cols["fixed_cosets"]), table(cols["advice_polys"])
        dd.perm_product_cosets, dd.perm_cosets = table(cols["perm_product_cosets"]), table(cols["perm_cosets"])
        dd.l0, dd.l_last, dd.l_active_row = cols["l0"].data_ptr(), cols["l_last"].data_ptr(), cols["l_active_row"].data_ptr()
        for t, name in enumerate(("lookup_product_polys", "lookup_permuted_input_polys", "lookup_permuted_table_polys")):
            setattr(dd, name, table([l[t] for l in cols["lookups"]]))
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

        def call():
            rc = L.h2hip_evaluate_h_bn254_device(hd.byref(), ctypes.c_void_p(d_values.data_ptr()), stream)
            assert rc == 0, L.h2hip_last_error()

        call()
        torch.cuda.synchronize()
        res = {}
        if check or time_cpu or host_timing:
            case_h = dict(case)
            for key in ("fixed_cosets", "advice_polys", "perm_product_cosets", "perm_cosets"):
                case_h[key] = [host(t) for t in cols[key]]
            for key in ("l0", "l_last", "l_active_row"):
                case_h[key] = host(cols[key])
            case_h["lookups"] = [tuple(host(t) for t in l) for l in cols["lookups"]]
            hh = evaluator.describe(case_h)
        if host_timing:
            # the host-pointer call (what patch 0003's evaluate_h_gpu makes: every column a Vec<F>), PCIe included, without and with the
            # proving key's constant columns pinned in HBM (h2hip_columns_pin: fixed cosets, l0 / l_last / l_active_row, permutation cosets)
            vals = host(v0).copy()

            def host_call():
                np.copyto(vals, host(v0))
                t0 = time.perf_counter()
                assert L.h2hip_evaluate_h_bn254(hh.byref(), vals.ctypes.data_as(ctypes.c_void_p)) == 0, L.h2hip_last_error()
                return (time.perf_counter() - t0) * 1e3

            def med(nrep=5):
                host_call()
                return float(np.median([host_call() for _ in range(nrep)]))

            key_cols = list(case_h["fixed_cosets"]) + list(case_h["perm_cosets"]) + [case_h["l0"], case_h["l_last"], case_h["l_active_row"]]
            full = 32 << ek
            n_full = len(key_cols) + len(case_h["perm_product_cosets"]) + 1
            res["host_pointer"] = {"ms_per_call_unpinned": med(), "full_size_columns": n_full, "key_columns": len(key_cols),
                                   "upload_MiB_unpinned": (n_full * full + (args.advice + 3 * args.lookups) * (32 << k)) / 2**20}
            h2.columns_pin(key_cols)
            res["host_pointer"]["ms_per_call_key_columns_pinned"] = med()
            res["host_pointer"]["upload_MiB_pinned"] = ((n_full - len(key_cols)) * full + (args.advice + 3 * args.lookups) * (32 << k)) / 2**20
            res["host_pointer"]["pinned_bytes"] = h2.columns_pinned_info()[1]
            h2.columns_unpin(key_cols)
        if check or time_cpu:
            want = host(v0).copy()
            t0 = time.perf_counter()
            assert orc.lib().oracle_evaluate_h(hh.byref(), want.ctypes.data_as(ctypes.c_void_p)) == 0
            res["oracle_s"] = time.perf_counter() - t0
            res["match"] = bool(np.array_equal(host(d_values), want))
            assert res["match"], "GPU evaluate_h differs from the oracle"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def timed():
            ts = []
            for _ in range(args.iters):
                e0.record()
                call()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            return float(np.median(ts))

        def stages():  # per-stage HIP-event times of the same call (their records put ~10 us of gap on the stream each)
            h2.profile_enable(True)
            h2.profile_reset()
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            h2.profile_enable(False)
            st = {}
            for name in ("evalh_cosets", "evalh_gates", "evalh_perm", "evalh_lookups"):
                tot, cnt = h2.profile_get(name)
                st[name] = tot / cnt if cnt else None
            return st

        # custom gates through the byte-code interpreter, then through the kernel generated for this circuit (compiled inline by hiprtc)
        L.h2hip_debug_set_evalh_codegen(ctypes.c_int(0), ctypes.c_uint32(0))
        call()
        res["interpreter"] = {"gpu_ms": timed(), "stage_ms": stages()}
        stats0 = (ctypes.c_uint64 * 5)()
        L.h2hip_debug_evalh_codegen_stats(stats0)
        L.h2hip_debug_set_evalh_codegen(ctypes.c_int(2), ctypes.c_uint32(0))
        t0 = time.perf_counter()
        call()
        torch.cuda.synchronize()
        first_call_s = time.perf_counter() - t0
        stats1 = (ctypes.c_uint64 * 5)()
        L.h2hip_debug_evalh_codegen_stats(stats1)
        generated = stats1[2] > stats0[2]
        res["gpu_ms"] = timed()
        res["stage_ms"] = stages()
        res["gates_kernel"] = "generated per circuit (hiprtc)" if generated else "interpreter (the program is beyond the generator's limits, or hiprtc is missing)"
        res["first_call_with_inline_compile_s"] = first_call_s
        if check or time_cpu:
            L.h2hip_debug_set_evalh_codegen(ctypes.c_int(2), ctypes.c_uint32(0))
            d_values.copy_(v0)
            call()
            torch.cuda.synchronize()
            res["match_generated"] = bool(np.array_equal(host(d_values), want))
            assert res["match_generated"], "GPU evaluate_h (generated gates kernel) differs from the oracle"
        L.h2hip_debug_set_evalh_codegen(ctypes.c_int(1), ctypes.c_uint32(0))
        res["rows_per_s"] = size / (res["gpu_ms"] * 1e-3)
        # the gates kernel against the multiplier: field multiplications per row of the compiled program x rows / kernel time
        n_mul = ctypes.c_uint32()
        if L.h2hip_debug_evalh_program_muls(ctypes.byref(hd.desc.custom_gates), ctypes.byref(n_mul)) == 0 and res["stage_ms"].get("evalh_gates"):
            gmul = n_mul.value * size / (res["stage_ms"]["evalh_gates"] * 1e-3) / 1e9
            res["gates_valu_roofline"] = {"bound": "valu-int", "kernel": "evalh_gates_gen" if generated else "evalh_gates_kernel", "field_mul_per_row": n_mul.value,
                                          "achieved": gmul, "peak": 179.0, "unit": "Gmul/s", "frac": gmul / 179.0}
        return res

    if args.check_k:
        out["check_k%d" % args.check_k] = run(args.check_k, True, False)
    out["k%d" % args.k] = run(args.k, False, args.cpu)
    if getattr(args, "host_k", 0):
        out["host_k%d" % args.host_k] = {k_: v for k_, v in run(args.host_k, False, False, host_timing=True).items() if k_ in ("host_pointer", "gpu_ms")}
    return out


def default_args(**over):
    ns = argparse.Namespace(k=18, check_k=12, gates=24, advice=12, fixed=10, perm=9, lookups=2, iters=5, cpu=False, host_k=16)
    for key, v in over.items():
        setattr(ns, key, v)
    return ns


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=18)
    ap.add_argument("--check-k", type=int, default=12, help="size at which the GPU result is compared with the oracle (0 = skip)")
    ap.add_argument("--gates", type=int, default=24)
    ap.add_argument("--advice", type=int, default=12)
    ap.add_argument("--fixed", type=int, default=10)
    ap.add_argument("--perm", type=int, default=9)
    ap.add_argument("--lookups", type=int, default=2)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--cpu", action="store_true", help="time the oracle at --k too (single thread)")
    ap.add_argument("--host-k", type=int, default=16, help="also time the host-pointer call at this k, with and without the key's columns pinned (0 = skip)")
    args = ap.parse_args()
    print(json.dumps(bench(args)))


if __name__ == "__main__":
    main()
