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


def run(h2, cpu=True, fixed_base=True, scalars="dense"):
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

    def timed(f, reps=3):
        f()
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(reps):
            f()
        ev[1].record()
        torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]) / reps

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

    out = {"k": k, "extended_k": ek, "fixed_base": fixed_base, "scalars": scalars, "single_ms": timed(single), "batched_ms": timed(batched),
           "fill_ext_ms": timed(fill_ext) + timed(restore),
           "calls": "16 MSMs of 2^17 (10 commit_lagrange + 6 commit), 10 iNTTs, 10 coset NTTs 2^17 -> 2^19, one inverse coset NTT of 2^19"}
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


def main():
    h2 = load_pkg()
    h2.init()
    print(json.dumps(run(h2, cpu="--no-cpu" not in sys.argv, fixed_base="--plain" not in sys.argv,
                         scalars="prover-like" if "--prover-like" in sys.argv else "dense")))


if __name__ == "__main__":
    main()
