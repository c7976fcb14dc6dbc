#!/usr/bin/python3
"""The hot-path calls of one KZG proof of examples/circuit-layout.rs's MyCircuit at k = 17 (BASELINE.json
configs[4]): 10 commit_lagrange + 6 commit of 2^17, 10 lagrange_to_coeff of 2^17, 10 coeff_to_extended 2^17 -> 2^19, one
extended_to_coeff of 2^19 -- device-resident columns, timed call by call and through the batched entry points, with the
oracle's time for the same calls on the host cores beside it (bounded: two MSMs and two transforms, scaled).
  python tools/trace_bench.py     (run on the GPU box)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402


def prover_like(h2, col, seed):
    """SURVEY.md 8(d)'s prover-like column: 90 % zero, 5 % in {1, 2}, 5 % uniform (what a sparse advice column looks like in Lagrange form)"""
    import numpy as np
    import torch
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    r = torch.rand(col.shape[0], device="cuda", generator=g)
    one = torch.from_numpy(h2.fr_from_int(1).view(np.int64)).cuda()
    two = torch.from_numpy(h2.fr_from_int(2).view(np.int64)).cuda()
    col = col.clone()
    col[r < 0.9] = 0
    col[(r >= 0.9) & (r < 0.925)] = one
    col[(r >= 0.925) & (r < 0.95)] = two
    return col


def run(h2, cpu=True, fixed_base=True, scalars="dense", idle_s=0.0, prewarm_ms=100.0, detail=False):
    """returns the result dict; cpu=True also times the oracle on the host cores (imports oracle/: bench/tools only)"""
    import torch
    k = 17
    n = 1 << k
    d = h2.EvaluationDomain.new(4, k)
    ek = d.extended_k
    g = h2.gen_points_device(0xABCD, n)
    gl = h2.gen_points_device(0xABCE, n)
    if fixed_base:  # ParamsKZG pins g and g_lagrange for its lifetime (window tables built once)
        h2.bases_pin_device(g)
        h2.bases_pin_device(gl)
    lag = [h2.gen_scalars_device(600 + i, n) for i in range(10)]
    if scalars == "prover-like":
        lag = [prover_like(h2, c, 900 + i) for i, c in enumerate(lag)]
    lag0 = [c.clone() for c in lag]
    ext = [torch.zeros((1 << ek, 4), dtype=torch.int64, device="cuda") for _ in range(10)]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

    rep_log = {}

    def timed(f, reps=5, name=None):
        """median of `reps` repetitions, each between its own pair of events (round 3 took the mean of three between ONE pair: a single
        stall owned the figure and nothing showed it was one); the list rides along in the result.  A named leg first runs ~prewarm_ms
        of untimed repetitions: a chip that has idled (the legs of bench.py ahead of this one are host-bound) runs its first tens of
        milliseconds at idle clocks -- `--idle S` reproduces that on purpose."""
        f()
        torch.cuda.synchronize()
        import gc
        keep_gc = os.environ.get("H2_TRACE_GC") == "1"
        if not keep_gc:  # ahead of the warm-up, not between it and the repetitions: the 25-45 ms the pass takes are GPU idle time, and the
            # first repetitions after such a gap run 5-15 % slow (clocks) -- measured when the pass sat right before them
            t_gc = time.perf_counter()
            gc.collect()
            if name:  # how long ONE full pass takes in this process: the size of the stall a repetition would have carried
                rep_log.setdefault("full_gc_pass_ms", []).append(round((time.perf_counter() - t_gc) * 1e3, 2))
        if name and idle_s:
            time.sleep(idle_s)
        if name and prewarm_ms:
            t_end = time.perf_counter() + prewarm_ms * 1e-3
            while time.perf_counter() < t_end:
                f()
            torch.cuda.synchronize()
        ts, gen2 = [], []
        # The blocking calls leave the GPU idle whenever the interpreter pauses, so a full (generation-2) garbage collection of a
        # process that has torch, numpy and half a dozen bench modules loaded -- tens of milliseconds, once in a long while -- lands
        # inside whichever repetition happens to trigger it.  Round 3's driver line carried one (36 ms "per repetition" = 3 x 7.6 + 85).
        # The collector is therefore run once up front and kept off during the repetitions; the count of full collections per
        # repetition rides along as evidence (0 with the collector off; `H2_TRACE_GC=1` leaves it on to show the effect).
        was_enabled = gc.isenabled()
        if not keep_gc:
            gc.disable()
        try:
            for _ in range(reps):
                g2 = gc.get_stats()[2]["collections"]
                ev[0].record()
                f()
                ev[1].record()
                torch.cuda.synchronize()
                ts.append(ev[0].elapsed_time(ev[1]))
                gen2.append(gc.get_stats()[2]["collections"] - g2)
        finally:
            if was_enabled:
                gc.enable()
        if name:
            rep_log[name] = [round(t, 4) for t in ts]
            rep_log[name + "_full_gc_passes"] = gen2
        return sorted(ts)[len(ts) // 2]

    def fill_ext():
        for e, c in zip(ext, lag):
            e[:n] = c

    def restore():  # the iNTTs below run in place: every repetition starts from the Lagrange-form columns again
        for c, c0 in zip(lag, lag0):
            c.copy_(c0)

    def single():
        restore()
        for c in lag:
            h2.msm_device(c, gl)
        for c in lag:
            h2.ifft_device(c, d.omega_inv, k, d.ifft_divisor)
        for c in lag[:6]:
            h2.msm_device(c, g)
        fill_ext()
        for e in ext:
            h2.coeff_to_extended_device(e, k, ek, d.extended_omega, d.g_coset, d.g_coset_inv)
        h2.extended_to_coeff_device(ext[0], ek, d.extended_omega_inv, d.extended_ifft_divisor, d.g_coset, d.g_coset_inv)

    def batched():
        restore()
        h2.msm_batch_device(lag, gl)
        h2.ifft_batch_device(lag, d.omega_inv, k, d.ifft_divisor)
        h2.msm_batch_device(lag[:6], g)
        fill_ext()
        h2.coeff_to_extended_batch_device(ext, k, ek, d.extended_omega, d.g_coset, d.g_coset_inv)
        h2.extended_to_coeff_device(ext[0], ek, d.extended_omega_inv, d.extended_ifft_divisor, d.g_coset, d.g_coset_inv)

    out = {"k": k, "extended_k": ek, "fixed_base": fixed_base, "scalars": scalars, "single_ms": timed(single, name="single"),
           "batched_ms": timed(batched, name="batched"), "fill_ext_ms": timed(fill_ext) + timed(restore), "reps_ms": rep_log, "stat": "median of 5",
           "calls": "16 MSMs of 2^17 (10 commit_lagrange + 6 commit), 10 iNTTs, 10 coset NTTs 2^17 -> 2^19, one inverse coset NTT of 2^19",
           "prewarm_ms": prewarm_ms, "idle_s": idle_s}
    if detail:  # per-call GPU times of two consecutive repetitions of `single` after an idle gap: is a slow repetition one stall or slow throughout?
        calls = []

        def single_logged():
            evs = [torch.cuda.Event(enable_timing=True)]
            evs[0].record()

            def mark():
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                evs.append(e)
            restore(); mark()
            for c in lag:
                h2.msm_device(c, gl); mark()
            for c in lag:
                h2.ifft_device(c, d.omega_inv, k, d.ifft_divisor); mark()
            for c in lag[:6]:
                h2.msm_device(c, g); mark()
            fill_ext(); mark()
            for e_ in ext:
                h2.coeff_to_extended_device(e_, k, ek, d.extended_omega, d.g_coset, d.g_coset_inv); mark()
            h2.extended_to_coeff_device(ext[0], ek, d.extended_omega_inv, d.extended_ifft_divisor, d.g_coset, d.g_coset_inv); mark()
            torch.cuda.synchronize()
            calls.append([round(a.elapsed_time(b), 4) for a, b in zip(evs, evs[1:])])
        time.sleep(max(idle_s, 2.0))
        single_logged()
        single_logged()
        single_logged()
        out["detail_after_idle_ms"] = {"order": "restore, 10 msm(gl), 10 ifft, 6 msm(g), fill_ext, 10 coset, 1 ext_to_coeff", "reps": calls,
                                       "totals": [round(sum(c_), 3) for c_ in calls]}
    if fixed_base:
        h2.bases_unpin_device(g)
        h2.bases_unpin_device(gl)
    if cpu:
        # the oracle on the host cores: one MSM, one iNTT, one coset NTT, scaled to the trace's counts
        from oracle import oracle as orc
        orc.build()
        do, _ = orc.domain_new(4, k)
        T = min(16, os.cpu_count() or 1)
        sc, bs = h2.to_numpy_u64(lag[0]), h2.to_numpy_u64(g)
        t0 = time.perf_counter()
        orc.best_multiexp(sc, bs, T)
        t_msm = time.perf_counter() - t0
        t0 = time.perf_counter()
        co = orc.lagrange_to_coeff(do, sc, T)
        t_ifft = time.perf_counter() - t0
        t0 = time.perf_counter()
        orc.coeff_to_extended(do, co, T)
        t_ext = time.perf_counter() - t0
        out["cpu_port"] = {"threads": T, "msm_s": t_msm, "ifft_s": t_ifft, "coeff_to_extended_s": t_ext,
                           "trace_s": 16 * t_msm + 10 * t_ifft + 11 * t_ext}
    return out


def run_host(h2, scalars="dense", reps=5):
    """The same 37 calls with HOST columns (numpy arrays = pageable caller memory, as a Rust prover's Vec<F>), PCIe included:
    `single_ms` one C-ABI call per reference call (what patches 0001 + 0002 yield), `batched_ms` through the host batch entry points
    (h2hip_msm_bn254_batch, h2hip_{ifft,coeff_to_extended}_bn254_fr_batch: patches 0004 + 0005).  g / g_lagrange are pinned host arrays."""
    import ctypes
    import numpy as np
    k = 17
    n = 1 << k
    d = h2.EvaluationDomain.new(4, k)
    ek = d.extended_k
    L = h2.lib()
    g = h2.to_numpy_u64(h2.gen_points_device(0xABCD, n)).copy()
    gl = h2.to_numpy_u64(h2.gen_points_device(0xABCE, n)).copy()
    h2.bases_pin(g)
    h2.bases_pin(gl)
    lag_d = [h2.gen_scalars_device(600 + i, n) for i in range(10)]
    if scalars == "prover-like":
        lag_d = [prover_like(h2, c, 900 + i) for i, c in enumerate(lag_d)]
    lag0 = [h2.to_numpy_u64(c).copy() for c in lag_d]
    del lag_d
    lag = [c.copy() for c in lag0]
    ext = [np.zeros((1 << ek, 4), dtype=np.uint64) for _ in range(10)]
    out1 = np.zeros(12, dtype=np.uint64)
    out10 = np.zeros((10, 12), dtype=np.uint64)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    ptrs = lambda arrs: (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])  # noqa: E731
    u32, sz = ctypes.c_uint32, ctypes.c_size_t

    def ok(rc):
        if rc:
            raise RuntimeError(L.h2hip_last_error().decode())

    def restore():
        for c, c0 in zip(lag, lag0):
            np.copyto(c, c0)

    def single():
        for c in lag:
            ok(L.h2hip_msm_bn254(P(c), P(gl), sz(n), P(out1)))
        for c in lag:
            ok(L.h2hip_ifft_bn254_fr(P(c), P(d.omega_inv), u32(k), P(d.ifft_divisor)))
        for c in lag[:6]:
            ok(L.h2hip_msm_bn254(P(c), P(g), sz(n), P(out1)))
        for c, e in zip(lag, ext):
            ok(L.h2hip_coeff_to_extended_bn254_fr(P(c), u32(k), P(e), u32(ek), P(d.extended_omega), P(d.g_coset), P(d.g_coset_inv)))
        ok(L.h2hip_extended_to_coeff_bn254_fr(P(ext[0]), u32(ek), P(d.extended_omega_inv), P(d.extended_ifft_divisor), P(d.g_coset), P(d.g_coset_inv)))

    def batched():
        ok(L.h2hip_msm_bn254_batch(ptrs(lag), P(gl), sz(n), sz(10), P(out10)))
        ok(L.h2hip_ifft_bn254_fr_batch(ptrs(lag), sz(10), P(d.omega_inv), u32(k), P(d.ifft_divisor)))
        ok(L.h2hip_msm_bn254_batch(ptrs(lag[:6]), P(g), sz(n), sz(6), P(out10)))
        ok(L.h2hip_coeff_to_extended_bn254_fr_batch(ptrs(lag), u32(k), ptrs(ext), sz(10), u32(ek), P(d.extended_omega), P(d.g_coset), P(d.g_coset_inv)))
        ok(L.h2hip_extended_to_coeff_bn254_fr(P(ext[0]), u32(ek), P(d.extended_omega_inv), P(d.extended_ifft_divisor), P(d.g_coset), P(d.g_coset_inv)))

    legs = {}

    def timed(f, name):
        restore()
        f()
        ts = []
        for _ in range(reps):
            restore()  # outside the clock: the reference's lagrange_to_coeff consumes its Vec, there is no copy to charge
            t0 = time.perf_counter()
            f()
            ts.append((time.perf_counter() - t0) * 1e3)
        legs[name] = [round(t, 4) for t in ts]
        return sorted(ts)[len(ts) // 2]

    try:
        res = {"k": k, "extended_k": ek, "scalars": scalars, "single_ms": timed(single, "single"), "batched_ms": timed(batched, "batched"),
               "reps_ms": legs, "stat": "median of %d, host clock around blocking calls" % reps,
               "pcie_bytes": {"up": (16 + 10 + 10) * n * 32 + (32 << ek), "down": 10 * n * 32 + 11 * (32 << ek)},
               "calls": "the 37 calls of trace_k17 on host columns (pageable numpy arrays), g / g_lagrange pinned host arrays"}
        # where the batched time goes, call by call (one extra untimed pass)
        parts = {}
        restore()
        for name, f in (("msm_batch_10", lambda: ok(L.h2hip_msm_bn254_batch(ptrs(lag), P(gl), sz(n), sz(10), P(out10)))),
                        ("ifft_batch_10", lambda: ok(L.h2hip_ifft_bn254_fr_batch(ptrs(lag), sz(10), P(d.omega_inv), u32(k), P(d.ifft_divisor)))),
                        ("msm_batch_6", lambda: ok(L.h2hip_msm_bn254_batch(ptrs(lag[:6]), P(g), sz(n), sz(6), P(out10)))),
                        ("coeff_to_extended_batch_10", lambda: ok(L.h2hip_coeff_to_extended_bn254_fr_batch(ptrs(lag), u32(k), ptrs(ext), sz(10), u32(ek), P(d.extended_omega), P(d.g_coset), P(d.g_coset_inv)))),
                        ("extended_to_coeff_1", lambda: ok(L.h2hip_extended_to_coeff_bn254_fr(P(ext[0]), u32(ek), P(d.extended_omega_inv), P(d.extended_ifft_divisor), P(d.g_coset), P(d.g_coset_inv))))):
            f()
            t0 = time.perf_counter()
            f()
            parts[name] = round((time.perf_counter() - t0) * 1e3, 4)
        res["batched_parts_ms"] = parts
        return res
    finally:
        h2.bases_unpin(g)
        h2.bases_unpin(gl)


def main():
    h2 = load_pkg()
    h2.init()
    if "--host" in sys.argv:
        print(json.dumps(run_host(h2, scalars="prover-like" if "--prover-like" in sys.argv else "dense")))
        return
    idle = float(sys.argv[sys.argv.index("--idle") + 1]) if "--idle" in sys.argv else 0.0
    print(json.dumps(run(h2, cpu="--no-cpu" not in sys.argv, fixed_base="--plain" not in sys.argv,
                         scalars="prover-like" if "--prover-like" in sys.argv else "dense", idle_s=idle,
                         prewarm_ms=0.0 if "--no-prewarm" in sys.argv else 100.0, detail="--detail" in sys.argv)))


if __name__ == "__main__":
    main()
