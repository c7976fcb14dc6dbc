#!/usr/bin/env python3
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


def main():
    import torch
    h2 = load_pkg()
    from oracle import oracle as orc
    orc.build()
    k = 17
    n = 1 << k
    d, _ = orc.domain_new(4, k)
    ek = d.extended_k
    g = h2.gen_points_device(0xABCD, n)
    gl = h2.gen_points_device(0xABCE, n)
    lag = [h2.gen_scalars_device(600 + i, n) for i in range(10)]
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

    def single():
        for c in lag:
            h2.msm_device(c, gl)
        for c in lag:
            h2.ifft_device(c, d.fe("omega_inv"), k, d.fe("ifft_divisor"))
        for c in lag[:6]:
            h2.msm_device(c, g)
        fill_ext()
        for e in ext:
            h2.coeff_to_extended_device(e, k, ek, d.fe("extended_omega"), d.fe("g_coset"), d.fe("g_coset_inv"))
        h2.extended_to_coeff_device(ext[0], ek, d.fe("extended_omega_inv"), d.fe("extended_ifft_divisor"), d.fe("g_coset"), d.fe("g_coset_inv"))

    def batched():
        h2.msm_batch_device(lag, gl)
        h2.ifft_batch_device(lag, d.fe("omega_inv"), k, d.fe("ifft_divisor"))
        h2.msm_batch_device(lag[:6], g)
        fill_ext()
        h2.coeff_to_extended_batch_device(ext, k, ek, d.fe("extended_omega"), d.fe("g_coset"), d.fe("g_coset_inv"))
        h2.extended_to_coeff_device(ext[0], ek, d.fe("extended_omega_inv"), d.fe("extended_ifft_divisor"), d.fe("g_coset"), d.fe("g_coset_inv"))

    out = {"k": k, "extended_k": ek, "single_ms": timed(single), "batched_ms": timed(batched), "fill_ext_ms": timed(fill_ext)}
    # the oracle on the host cores: one MSM, one iNTT, one coset NTT, scaled to the trace's counts
    T = min(16, os.cpu_count() or 1)
    sc, bs = h2.to_numpy_u64(lag[0]), h2.to_numpy_u64(g)
    t0 = time.perf_counter()
    orc.best_multiexp(sc, bs, T)
    t_msm = time.perf_counter() - t0
    t0 = time.perf_counter()
    co = orc.lagrange_to_coeff(d, sc, T)
    t_ifft = time.perf_counter() - t0
    t0 = time.perf_counter()
    orc.coeff_to_extended(d, co, T)
    t_ext = time.perf_counter() - t0
    out["cpu_port"] = {"threads": T, "msm_s": t_msm, "ifft_s": t_ifft, "coeff_to_extended_s": t_ext,
                       "trace_s": 16 * t_msm + 10 * t_ifft + 11 * t_ext}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
