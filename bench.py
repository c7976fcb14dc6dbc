#!/usr/bin/env python3
"""bench.py -- BN254 MSM (+ Fr NTT) throughput on MI355X, one process per GPU.

Step = one pass of the hot path over one batch of synthetic input: one BN254 G1 MSM of
2^LOG_N pairs per GPU (BASELINE.json configs[1]: 2^20 on one MI355X), inputs generated in HBM
before the timed region (SplitMix64 scalars / try-and-increment points).
With N > 1 ranks each rank runs the same-size shard (weak scaling), the 96-byte partials are
all-gathered over RCCL and folded on every rank (arithmetic.rs:153).

Prints ONE JSON line on rank 0.  `value` = bucket-accumulation G1 adds per second over the
whole job = N * n * W / t (W = 254//c + 1 signed windows at the engine's window width c;
SURVEY.md 8(d)); pairs/s and the NTT figure ride along as extra keys.  `roofline` prices the
dominant kernel (msm_accum_kernel) at its algorithmic 96 B per pair against 8 TB/s, timed with
HIP events on the stream it is launched on; `cpu_baseline` is the oracle's restatement of the
reference's rayon path timed on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from __graft_entry__ import load_pkg  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
MSM_BYTES_PER_PAIR = 96  # 32 B scalar + 64 B affine point, read once
NTT_BYTES_PER_ELEM = 64  # one 32 B read + one 32 B write per transform
FIELD_MUL_PER_BUCKET_ADD = 9.2  # XYZZ mixed add: 7 products + 2 squares + one two-product single-reduction form (csrc/ecu.cuh)
FIELD_MUL_PEAK_G = 179.0  # measured peak of csrc/fieldu.cuh's multiplier on MI355X, G multiplies/s (tools/mul_rate.hip)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def pmc_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc pass
    (profiles/*_pmc_traffic.json, made by tools/pmc_traffic.py), or None when absent."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        return d.get(workload, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def spawn_ranks(n_ranks):
    """Start `python -m torch.distributed.run --nproc-per-node N bench.py <same flags>` as a child process and return its
    exit code.  Called before anything in this process has touched the GPU (torch is imported, no device call made)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-n", type=int, default=20, help="pairs per GPU = 2^log_n")
    ap.add_argument("--ntt-log-n", type=int, default=22)
    ap.add_argument("--window", type=int, default=0, help="MSM window bits (0 = engine default)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the "
                    "N > 1 path on a single-GPU box, where every rank then uses cuda:0)")
    ap.add_argument("--batch", type=int, default=8, help="also time a pipelined batch of this many MSMs (extra key; 0/1 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ntt", action="store_true")
    ap.add_argument("--no-next-rows", action="store_true", help="skip the evaluate_h / g_to_lagrange legs (SURVEY.md 8(f).3, (f).4)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: this process has made no GPU call yet, so it may start the N ranks as fresh
        # child processes (one per GPU, the launch shape the driver uses) and hand their output and exit code through
        raise SystemExit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libhalo2hip has no CPU fallback")
    dev_index = local_rank if args.backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)
    gather_dev = dev if args.backend == "nccl" else None

    h2 = load_pkg()
    from importlib import import_module
    h2dist = import_module("halo2_pse_amd.dist")
    h2.init(dev_index)
    if args.window:
        h2.set_msm_window(args.window)

    n = 1 << args.log_n
    c = h2.get_msm_window(n)
    W = 254 // c + 1
    # every rank draws its own shard of one global sequence (element index offset = rank * n)
    d_scalars = h2.gen_scalars_device(0x5EED0001, n, start=rank * n, device=dev)
    d_points = h2.gen_points_device(0x5EED0002, n, start=rank * n, device=dev)
    torch.cuda.synchronize()

    def step():
        part = h2.msm_device(d_scalars, d_points)
        if world > 1:
            return h2dist.allgather_fold(part, h2, device=gather_dev)
        return part

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        result = step()
    h2.profile_enable(True)
    h2.profile_reset()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        result = step()
    sync_all()
    elapsed = time.perf_counter() - t0
    h2.profile_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    stages = {}
    for st in ("msm_total", "msm_digits", "msm_sort", "msm_accum", "msm_heavy", "msm_reduce"):
        ms, cnt = h2.profile_get(st)
        stages[st] = ms / cnt if cnt else None

    # ---- batched commit: B MSMs over the same bases in one pipelined call, rank 0 only ----
    batched = None
    if rank == 0 and world == 1 and args.batch > 1 and args.log_n <= 22:
        B = args.batch
        cols = [h2.gen_scalars_device(0x5EED0001, n, start=(j + 1) * n, device=dev) for j in range(B)]
        h2.msm_batch_device(cols, d_points)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            h2.msm_batch_device(cols, d_points)
        torch.cuda.synchronize()
        tb = (time.perf_counter() - t1) / (reps * B)
        batched = {"count": B, "ms_per_msm": tb * 1e3, "value": n * W / tb, "unit": "G1-adds/s",
                   "note": "h2hip_msm_bn254_batch_device: whole MSMs pipelined over three streams"}
        del cols
        # the prover's size (BASELINE.json configs[4], k = 17): 16 column commits as one fused batch, against one call each
        n17 = 1 << 17
        cols = [h2.gen_scalars_device(0x5EED0001, n17, start=(j + 1) * n17, device=dev) for j in range(16)]
        pts17 = d_points[:n17]

        def timed(f, reps=3):
            f()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):
                f()
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / reps

        t_fused = timed(lambda: h2.msm_batch_device(cols, pts17)) / 16
        t_single = timed(lambda: [h2.msm_device(c_, pts17) for c_ in cols]) / 16
        batched["k17"] = {"count": 16, "ms_per_msm": t_fused * 1e3, "ms_per_msm_one_call_each": t_single * 1e3,
                          "note": "batches of up to 2^18 pairs run fused: one sort / accumulate / reduce over the windows of all MSMs"}
        del cols

    # ---- NTT leg (BASELINE.json configs[2]: k = 22 NTT + iNTT), outside the MSM timed region ----
    ntt = None
    if not args.no_ntt and rank == 0:
        from oracle import oracle
        k = args.ntt_log_n
        d, _ = oracle.domain_new(2, k)
        d_a = h2.gen_scalars_device(0x5EED0003, 1 << k, device=dev)
        for _ in range(2):
            h2.ntt_device(d_a, d.fe("omega"), k)
            h2.ifft_device(d_a, d.fe("omega_inv"), k, d.fe("ifft_divisor"))
        torch.cuda.synchronize()
        reps = 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            h2.ntt_device(d_a, d.fe("omega"), k)
            h2.ifft_device(d_a, d.fe("omega_inv"), k, d.fe("ifft_divisor"))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / (2 * reps)
        ntt = {
            "log_n": k,
            "ms_per_transform": ms,
            "elems_per_s": (1 << k) / (ms * 1e-3),
            "roofline": {"bound": "hbm", "achieved": NTT_BYTES_PER_ELEM * (1 << k) / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": NTT_BYTES_PER_ELEM * (1 << k) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         # PMC bytes of one transform = two strided passes + the final pass (committed --pmc runs, 2^22 only)
                         "traffic": (2 * st + fi) if (k == 22 and (st := pmc_traffic("ntt_2p22_strided")) and (fi := pmc_traffic("ntt_2p22_final")))
                         else None},
        }
        del d_a

    # ---- SURVEY.md 8(f).3 / (f).4 legs (extra keys; rank 0, N = 1 only), outside the MSM timed region ----
    next_rows = None
    if rank == 0 and world == 1 and not args.no_next_rows:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import evalh_bench
        import g2l_bench
        eh = evalh_bench.bench(evalh_bench.default_args(k=18, check_k=12, iters=5))
        gl = g2l_bench.bench(argparse.Namespace(k=[12, 16], cpu_k=12, threads=min(16, os.cpu_count() or 1)))
        next_rows = {
            "evaluate_h": {"workload": "2^20 extended rows (k = 18), %d gate polynomials, %d advice + %d fixed columns, %d permutation columns, "
                                       "%d lookups; device-resident" % (eh["gates"], eh["advice"], eh["fixed"], eh["perm_columns"], eh["lookups"]),
                           "ms_per_call": eh["k18"]["gpu_ms"], "rows_per_s": eh["k18"]["rows_per_s"],
                           "parity_vs_cpu_k12": eh["check_k12"]["match"], "cpu_port_s_k12": eh["check_k12"]["oracle_s"],
                           "gpu_ms_k12": eh["check_k12"]["gpu_ms"]},
            "g_to_lagrange": {"k16_ms": gl["k16"]["gpu_ms"], "k16_scalar_muls_per_s": gl["k16"]["scalar_muls_per_s"],
                              "k12_ms": gl["k12"]["gpu_ms"], "cpu_port_s_k12": gl["k12"]["oracle_s"], "cpu_threads": gl["k12"]["oracle_threads"],
                              "parity_vs_cpu_k12": gl["k12"]["match"]},
        }

    # ---- CPU baseline (rank 0, N = 1 only): the oracle's best_multiexp on the same inputs ----
    cpu = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        cores = min(16, os.cpu_count() or 1)  # the box's CPU share for one GPU
        sc, bs = h2.to_numpy_u64(d_scalars), h2.to_numpy_u64(d_points)
        oracle.best_multiexp(sc[:4096], bs[:4096], cores)  # warm up
        times = []
        cpu_out = None
        budget = time.perf_counter() + 30.0
        while len(times) < 5 and time.perf_counter() < budget:
            t1 = time.perf_counter()
            cpu_out = oracle.best_multiexp(sc, bs, cores)
            times.append(time.perf_counter() - t1)
        tmed = sorted(times)[len(times) // 2]
        cpu_c = oracle.window_c(n // cores)
        cpu = {
            "value": n * W / tmed,  # same unit as `value`: the job's n*W adds per second of CPU time
            "unit": "G1-adds/s",
            "pairs_per_s": n / tmed,
            "cores": cores,
            "cpu_model": cpu_model(),
            "kind": "port",
            "sample": "full 2^%d-pair MSM, median of %d runs, %.3f s each; C restatement of best_multiexp "
                      "(chunk = n/T per thread, c = %d unsigned windows), not the Rust binary" % (args.log_n, len(times), tmed, cpu_c),
        }
        parity = bool(np.array_equal(oracle.g1_to_affine(cpu_out), h2.g1_to_affine(result)))

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        adds = world * n * W * args.steps
        accum_ms = stages.get("msm_accum")
        roof = None
        if accum_ms:
            ach = MSM_BYTES_PER_PAIR * n / (accum_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                    "traffic": pmc_traffic("msm_2p%d" % args.log_n), "kernel": "msm_accum_kernel", "kernel_ms": accum_ms,
                    "algorithmic_bytes_per_launch": MSM_BYTES_PER_PAIR * n}
        # second roofline, the one that actually binds: 256-bit modular multiplies per second against the
        # multiplier's measured chip-wide peak (tools/mul_rate.hip: 179 G/s for the explicit-mad form at >= 4 waves/SIMD)
        valu = None
        if accum_ms:
            gmul = n * W * FIELD_MUL_PER_BUCKET_ADD / (accum_ms * 1e-3) / 1e9
            valu = {"bound": "valu-int", "kernel": "msm_accum_kernel", "field_mul_per_add": FIELD_MUL_PER_BUCKET_ADD,
                    "achieved": gmul, "peak": FIELD_MUL_PEAK_G, "unit": "Gmul/s", "frac": gmul / FIELD_MUL_PEAK_G}
        out = {
            "metric": "bn254_msm_g1_adds_per_sec",
            "value": adds / elapsed,
            "unit": "G1-adds/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32x8 (256-bit Montgomery integers)",
            "data": "synthetic",
            "config": {"workload": "bn254_g1_msm_2p%d_per_gpu" % args.log_n, "pairs_per_gpu": n, "window_bits": c, "windows": W,
                       "signed_digits": True, "parallelism": "shard%d+allgather96B" % world},
            "pairs_per_s": world * n * args.steps / elapsed,
            "stage_ms": stages,
            "roofline": roof,
            "valu_roofline": valu,
            "cpu_baseline": cpu,
            "parity_vs_cpu": parity,
            "batched": batched,
            "ntt": ntt,
            "next_rows": next_rows,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
