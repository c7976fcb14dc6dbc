#!/usr/bin/python3
"""bench.py -- BN254 MSM (+ Fr NTT) throughput on MI355X, one process per GPU.

Step = one pass of the hot path over one batch of synthetic input: one BN254 G1 MSM of
2^LOG_N pairs per GPU (BASELINE.json configs[1]: 2^20 on one MI355X), inputs generated in HBM
before the timed region (SplitMix64 scalars / try-and-increment points).
The bases are pinned the way a ParamsKZG pins g / g_lagrange (h2hip_bases_pin_device: the
engine's fixed-base window table, built before the timed region; `--form plain` times the
unpinned form).  With N > 1 ranks each rank runs the same-size shard (weak scaling), the
96-byte partials are all-gathered over RCCL (asynchronously: the gather of step i travels under
the MSM of step i + 1; all K folds complete inside the timed region) and folded on every rank
(arithmetic.rs:153).
`python bench.py --gpus N` started bare spawns the N ranks itself.

Prints ONE JSON line on rank 0.  `value` = bucket-accumulation G1 adds per second over the
whole job = N * n * W / t (W = ceil(255 / c) signed windows at the engine's window width c;
SURVEY.md 8(d)); pairs/s, the other sizes and the NTT figures ride along as extra keys.
`roofline` prices the dominant kernel (msm_accum_kernel) at its algorithmic 96 B per pair
against 8 TB/s, timed with HIP events on the stream it is launched on; `cpu_baseline` is the
oracle's restatement of the reference's rayon path timed on this box's host cores.
"""
import argparse
import ctypes
import json
import os
import subprocess
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
FIELD_MUL_PER_BUCKET_ADD = 9.2  # XYZZ mixed add: 7 products + 2 squares + one two-product single-reduction form (csrc/ecu.h)
FIELD_MUL_PEAK_G = 179.0  # measured peak of csrc/fieldu.h's multiplier on MI355X, G multiplies/s (tools/mul_rate.hip)
# Fr multiplications per element of one transform: (log2 n) / 2 butterfly products, plus per pass boundary one to apply the
# inter-pass twiddle and -- where it is combined from the two-level table instead of read from a per-domain table -- one to combine it;
# the closing reduction is not a multiplication; each pass's first round forms one product fewer per four points (its twiddle is 1).
# 2^22 on two passes: 11 + 2 - 0.5 = 12.5; 2^24 on three: 12 + 2.5 (one boundary reads a table) - 0.75 = 13.75
NTT_FIELD_MUL_PER_ELEM = {20: 10.5, 22: 12.5, 24: 13.75, 26: 14.75}
STAGES = ("msm_total", "msm_digits", "msm_sort", "msm_accum", "msm_reduce")  # over-full buckets are summed inside the accumulate launch


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
    (profiles/pmc_traffic.json, made by tools/pmc_traffic.py), or None when absent."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        return d.get(workload, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


_PMC_BROKEN = [False]  # a counter pass that failed or timed out: no further passes in this run (the committed figures stand in)


def pmc_pass(child_flags, counters, names, timeout=60):
    """One child process of this file under `rocprofv3 --pmc <counters>` (no trace domain), running nothing but one leg (`--only-step` /
    `--only-ntt`).  Returns {kernel name substring: {counter: (mean per launch, launches)}} or None when rocprofv3 is missing, the pass
    fails or a named kernel was not seen."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if _PMC_BROKEN[0] or not os.path.exists(exe):
        return None
    env = dict(os.environ, TMPDIR="/tmp")
    for k_ in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k_, None)
    tmp = tempfile.mkdtemp(prefix="h2pmc_", dir="/tmp")
    try:
        cmd = [exe, "--pmc"] + list(counters) + ["--output-format", "csv", "-d", tmp, "-o", "p", "--", sys.executable, os.path.abspath(__file__)] + list(child_flags)
        # its own process group: a pass that overruns is ended together with the program rocprofv3 started
        pr = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env, cwd="/tmp", start_new_session=True)
        try:
            rcode = pr.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            import signal
            try:
                os.killpg(pr.pid, signal.SIGKILL)
            except OSError:
                pass
            pr.wait()
            rcode = -1
        if rcode != 0:
            _PMC_BROKEN[0] = True
            return None
        acc = {name: {c_: [0.0, 0] for c_ in counters} for name in names}
        for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    c_ = row.get("Counter_Name")
                    if c_ not in counters:
                        continue
                    for name in names:
                        if name in row.get("Kernel_Name", ""):
                            acc[name][c_][0] += float(row["Counter_Value"])
                            acc[name][c_][1] += 1
        out = {}
        for name in names:
            if any(acc[name][c_][1] == 0 for c_ in counters):
                return None
            out[name] = {c_: (acc[name][c_][0] / acc[name][c_][1], acc[name][c_][1]) for c_ in counters}
        return out
    except (OSError, subprocess.SubprocessError, ValueError):
        _PMC_BROKEN[0] = True
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def measure_traffic(child_flags, kernels, timeout=60):
    """HBM bytes per launch of the named kernels, measured by THIS run: two child processes under `rocprofv3 --pmc` (FETCH_SIZE and
    WRITE_SIZE in separate passes, no trace domain in either, as the MI355X guide's HBM section prescribes).  kernels: [(name substring,
    read factor)] -- counter units are KiB; the guide's x2 on FETCH_SIZE applies to wide coalesced reads (the NTT passes), per-lane
    64-B gathers take factor 1 (the raw counter already exceeds the known gather volume, profiles/pmc_traffic.json's note); WRITE_SIZE
    is exact.  Returns {name: dict} or None (the committed figure then stands in, labelled as such)."""
    names = [name for name, _ in kernels]
    rd_ = pmc_pass(child_flags, ["FETCH_SIZE"], names, timeout)
    wr_ = rd_ and pmc_pass(child_flags, ["WRITE_SIZE"], names, timeout)
    if not wr_:
        return None
    out = {}
    for name, factor in kernels:
        rd, wr = rd_[name]["FETCH_SIZE"][0] * 1024.0 * factor, wr_[name]["WRITE_SIZE"][0] * 1024.0
        out[name] = {"hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr, "read_factor": factor,
                     "launches": [rd_[name]["FETCH_SIZE"][1], wr_[name]["WRITE_SIZE"][1]]}
    return out


N_SIMD = 256 * 4  # MI355X: 256 CUs x 4 SIMDs; a SIMD issues one wave64 vector instruction per four cycles


def measure_issue(child_flags, names, timeout=60):
    """Share of the chip's vector-issue slots the named kernels fill, measured by THIS run: one more `rocprofv3 --pmc` child pass with the
    SQ counters.  SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES count quad-cycles summed over the waves (MI355X guide, PMC table); GRBM_GUI_ACTIVE
    is the kernel's busy cycles summed over the 8 XCDs.  issue_frac = SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 / 4): the
    kernel's VALU instructions against one per SIMD per four cycles for as long as the chip was busy with it (GRBM_GUI_ACTIVE includes the
    dispatch's ramp and reads a few % long on launches under ~0.3 ms -- the guide's note -- so the figure errs low).  Also: waves resident per
    SIMD on average (SQ_WAVE_CYCLES over the same denominator) and the instructions per wave."""
    ctr = ["SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAVES", "GRBM_GUI_ACTIVE"]
    r = pmc_pass(child_flags, ctr, names, timeout)
    if not r:
        return None
    out = {}
    for name in names:
        v = {c_: r[name][c_][0] for c_ in ctr}
        slots = N_SIMD * v["GRBM_GUI_ACTIVE"] / 8.0 / 4.0
        out[name] = {"issue_frac": v["SQ_ACTIVE_INST_VALU"] / slots, "waves_per_simd_resident": v["SQ_WAVE_CYCLES"] / slots,
                     "valu_insts_per_wave": v["SQ_INSTS_VALU"] / v["SQ_WAVES"], "wave_parked_frac": v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"],
                     "counters": v, "launches": r[name]["SQ_WAVES"][1],
                     "source": "measured by this run: one rocprofv3 --pmc child pass (SQ counters, no trace domain); issue_frac = SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 / 4)"}
    return out


def windows_of(c):
    return (255 + c - 1) // c


def spawn_ranks(n_ranks):
    """Start `python -m torch.distributed.run --nproc-per-node N bench.py <same flags>` as a child process and return its
    exit code.  Called before anything in this process has touched the GPU (torch is imported, no device call made)."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def timed_msm(h2, ds, dp, reps, warm=2):
    """ms per h2hip_msm_bn254_device call (wall clock around the blocking calls) and the per-stage HIP-event times"""
    for _ in range(warm):
        r = h2.msm_device(ds, dp)
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
        st[s] = tot / cnt if cnt else None
    return ms, st, r


def msm_roofline(n, accum_ms, workload=None):
    ach = MSM_BYTES_PER_PAIR * n / (accum_ms * 1e-3) / 1e9
    traffic = pmc_traffic(workload) if workload else None
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
            "traffic": traffic, "traffic_source": "profiles/pmc_traffic.json (committed rocprofv3 --pmc passes of this workload; not measured by this run)"
            if traffic is not None else None, "kernel": "msm_accum_kernel", "kernel_ms": accum_ms,
            "algorithmic_bytes_per_launch": MSM_BYTES_PER_PAIR * n}


def inlib_child(args):
    """--inlib N: ONE process drives N GPUs through the C ABI -- h2hip_init(ids, N), host-pointer h2hip_msm_bn254 over bases
    pinned across the devices, every device's shard streaming in over its own PCIe link, partials folded on the host
    (HALO2_HIP_GATHER=rccl: all-gathered over xGMI from the devices' HBM instead).  Prints one JSON object."""
    h2 = load_pkg()
    n_dev = args.inlib
    ids = list(range(n_dev))
    if h2.device_count() < n_dev:
        ids = [i % max(1, h2.device_count()) for i in range(n_dev)]  # rehearsal on a smaller box (needs the switch below)
        os.environ.setdefault("HALO2_HIP_ALLOW_DUPLICATE_DEVICES", "1")
    h2.init(ids)
    n = n_dev << args.log_n  # weak scaling: 2^log_n pairs per device
    sc = np.empty((n, 4), dtype=np.uint64)
    bs = np.empty((n, 8), dtype=np.uint64)
    torch.cuda.set_device(ids[0])
    chunk = 1 << 22
    for o in range(0, n, chunk):  # the synthetic inputs, generated on the first device, handed over as host arrays
        m = min(chunk, n - o)
        sc[o:o + m] = h2.to_numpy_u64(h2.gen_scalars_device(0x5EED0001, m, start=o))
        bs[o:o + m] = h2.to_numpy_u64(h2.gen_points_device(0x5EED0002, m, start=o))
    t0 = time.perf_counter()
    h2.bases_pin(bs)
    pin_s = time.perf_counter() - t0
    info = h2.bases_pinned_info(bs)
    out = h2.best_multiexp(sc, bs)
    times = []
    for _ in range(args.steps):
        t0 = time.perf_counter()
        out = h2.best_multiexp(sc, bs)
        times.append(time.perf_counter() - t0)
    t = sorted(times)[len(times) // 2]
    W = info[2]
    res = {"n_devices": n_dev, "device_ids": ids, "pairs_total": n, "window_bits": info[1], "windows": W, "pin_s": pin_s,
           "table_bytes_total": info[3], "ms_per_msm": t * 1e3, "value": n * W / t, "unit": "G1-adds/s", "pairs_per_s": n / t,
           "gather": "host fold (duplicate device ids: rehearsal)" if len(set(ids)) < len(ids) else os.environ.get("HALO2_HIP_GATHER", "host") + " (HALO2_HIP_GATHER)",
           "note": "host-pointer h2hip_msm_bn254 (scalars cross PCIe inside the call: %d MiB per device), bases pinned per device" % ((32 << args.log_n) >> 20),
           "result_affine_x0": int(h2.g1_to_affine(out)[0])}
    h2.bases_unpin(bs)
    print("INLIB " + json.dumps(res))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--prewarm-ms", type=float, default=100.0, help="untimed steps run ahead of the --warmup steps until about this much GPU work has "
                    "passed: an idle MI355X takes ~50 ms of load to reach its running clocks (3 warm-up steps of 1.25 ms: 1.276 ms per step; 50: 1.245)")
    ap.add_argument("--log-n", type=int, default=20, help="pairs per GPU = 2^log_n")
    ap.add_argument("--ntt-log-n", type=int, default=22)
    ap.add_argument("--window", type=int, default=0, help="MSM window bits (0 = engine default)")
    ap.add_argument("--form", default="fixed", choices=["fixed", "plain"], help="fixed: bases pinned with their window table (the KZG "
                    "commit path); plain: arbitrary bases, one bucket set per window")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the "
                    "N > 1 path on a single-GPU box, where every rank then uses cuda:0)")
    ap.add_argument("--batch", type=int, default=8, help="also time a pipelined batch of this many MSMs (extra key; 0/1 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ntt", action="store_true")
    ap.add_argument("--no-sizes", action="store_true", help="skip the other sizes (2^22..2^26 MSM, 2^20..2^26 NTT, host-pointer figures, k = 17 trace)")
    ap.add_argument("--no-next-rows", action="store_true", help="skip the evaluate_h / g_to_lagrange legs (SURVEY.md 8(f).3, (f).4)")
    ap.add_argument("--config4-log-n", type=int, default=24, help="N > 1: BASELINE.json configs[3], an MSM of 2^this pairs in TOTAL sharded over the N GPUs "
                    "(strong scaling; 0 = skip)")
    ap.add_argument("--no-inlib", action="store_true", help="N > 1: skip the single-process N-device leg (h2hip_init with N ids)")
    ap.add_argument("--no-measure-traffic", action="store_true", help="take roofline.traffic from profiles/pmc_traffic.json instead of measuring it with two rocprofv3 --pmc child passes")
    ap.add_argument("--only-step", action="store_true", help="nothing but the timed step (counter passes: one kernel mix per run)")
    ap.add_argument("--only-ntt", action="store_true", help="nothing but the NTT leg (counter passes)")
    ap.add_argument("--inlib", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.only_step or args.only_ntt:
        args.no_cpu_baseline = args.no_sizes = args.no_next_rows = args.no_inlib = args.no_measure_traffic = True
        args.batch = 0
        args.config4_log_n = 0
        args.no_ntt = args.only_step
    if args.inlib:
        return inlib_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: this process has made no GPU call yet, so it may start the N ranks as fresh
        # child processes (one per GPU, the launch shape the driver uses) and hand their output and exit code through
        raise SystemExit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # The contract is ONE JSON line on stdout.  Libraries underneath write to file descriptor 1 on their own (gloo announces its
    # ranks there): from here on descriptor 1 is stderr, and rank 0 writes its line to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
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
    # every rank draws its own shard of one global sequence (element index offset = rank * n)
    d_scalars = h2.gen_scalars_device(0x5EED0001, n, start=rank * n, device=dev)
    d_points = h2.gen_points_device(0x5EED0002, n, start=rank * n, device=dev)
    torch.cuda.synchronize()
    pin = None
    if args.form == "fixed":
        t0 = time.perf_counter()
        h2.bases_pin_device(d_points)
        pin_s = time.perf_counter() - t0
        info = h2.bases_pinned_info(d_points)
        c, W = info[1], info[2]
        pin = {"seconds": pin_s, "table_bytes": info[3], "note": "h2hip_bases_pin_device: W x n x 64 B window table, built once per ParamsKZG"}
    else:
        c = h2.get_msm_window(n)
        W = windows_of(c)

    # N > 1: the 96-byte all-gather of step i travels while step i + 1's shard MSM runs (async RCCL collective, two staging
    # slots); every step's fold is still completed inside the timed region -- the last one by drain() before the closing barrier.
    pending = [None, 0]

    def step():
        part = h2.msm_device(d_scalars, d_points)
        if world == 1:
            return part
        done = h2dist.allgather_finish(pending[0], h2) if pending[0] is not None else None
        pending[0] = h2dist.allgather_start(part, device=gather_dev, slot=pending[1])
        pending[1] ^= 1
        return done

    def drain():
        if pending[0] is None:
            return None
        out = h2dist.allgather_finish(pending[0], h2)
        pending[0] = None
        return out

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.only_ntt:
        args.steps, args.warmup = 1, 0
    result = None
    # the same count on every rank (the steps hold a collective): from the size, not from a clock
    prewarm_steps = 0 if args.only_ntt else min(400, int(args.prewarm_ms / (1.25 * 2.0 ** (args.log_n - 20))))
    for _ in range(prewarm_steps):
        step()
    for _ in range(args.warmup):
        result = step()
    if world > 1:
        last = drain()
        result = last if last is not None else result
    # timed region: only the dominant kernel is timed inside it (HIP events on its own dispatch, no marker packets: the
    # per-stage timers below put ~10 us of gap on the stream at every stage boundary)
    h2.profile_enable(2)
    h2.profile_reset()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        result = step()
    if world > 1:
        result = drain()
    sync_all()
    elapsed = time.perf_counter() - t0
    h2.profile_enable(False)
    accum_tot, accum_cnt = h2.profile_get("msm_accum")
    accum_ms_live = accum_tot / accum_cnt if accum_cnt else None
    # per-stage times: the same step, untimed, with every stage bracketed by events
    h2.profile_enable(True)
    h2.profile_reset()
    for _ in range(min(args.steps, 10)):
        step()
    if world > 1:
        drain()
    sync_all()
    h2.profile_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    stages = {}
    for st in STAGES:
        ms, cnt = h2.profile_get(st)
        stages[st] = ms / cnt if cnt else None

    # ---- N > 1: BASELINE.json configs[3] -- ONE MSM of 2^24 pairs in total, sharded over the N GPUs (strong scaling: 2^24 / N pairs per
    # rank, fixed-base, 96-byte partials all-gathered over RCCL and folded on every rank), next to the weak-scaling figure above ----
    config4 = None
    if world > 1 and args.config4_log_n:
        n4 = 1 << args.config4_log_n
        lo4, hi4 = h2dist.shard_range(n4, rank, world)
        ds4 = h2.gen_scalars_device(0x5EED0001, hi4 - lo4, start=lo4, device=dev)
        dp4 = h2.gen_points_device(0x5EED0002, hi4 - lo4, start=lo4, device=dev)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        h2.bases_pin_device(dp4)
        pin4_s = time.perf_counter() - t1
        info4 = h2.bases_pinned_info(dp4)
        pend4 = [None, 0]

        def step4():
            part = h2.msm_device(ds4, dp4)
            done = h2dist.allgather_finish(pend4[0], h2) if pend4[0] is not None else None
            pend4[0] = h2dist.allgather_start(part, device=gather_dev, slot=pend4[1])
            pend4[1] ^= 1
            return done

        def drain4():
            out4 = h2dist.allgather_finish(pend4[0], h2)
            pend4[0] = None
            return out4

        for _ in range(2):
            step4()
        drain4()
        k4 = 5
        sync_all()
        t1 = time.perf_counter()
        for _ in range(k4):
            step4()
        r4 = drain4()
        sync_all()
        el4 = time.perf_counter() - t1
        aff4 = h2.g1_to_affine(r4)
        # every rank folded the same gathered partials itself: all N folds must be one group element
        lim = torch.from_numpy(aff4.view(np.int64).copy())
        red_dev = dev if args.backend == "nccl" else "cpu"
        tmax, tmin = lim.to(red_dev).clone(), lim.to(red_dev).clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        te = torch.tensor([el4], dtype=torch.float64, device=red_dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        el4 = float(te.item())
        h2.bases_unpin_device(dp4)
        del ds4, dp4
        torch.cuda.empty_cache()
        config4 = {"workload": "bn254_g1_msm_2p%d_total_sharded_over_%d_gpus" % (args.config4_log_n, world), "scaling": "strong", "pairs_total": n4,
                   "pairs_per_rank": n4 // world, "window_bits": info4[1], "windows": info4[2], "pin_s": pin4_s, "table_bytes_per_rank": info4[3],
                   "steps": k4, "ms_per_msm": el4 / k4 * 1e3, "value": n4 * info4[2] / (el4 / k4), "unit": "G1-adds/s", "pairs_per_s": n4 / (el4 / k4),
                   "same_group_element_on_every_rank": bool(torch.equal(tmax, tmin)),
                   "note": "time = max over ranks between two barriers; the gather of step i travels under the MSM of step i + 1, all folds inside"}
        if rank == 0 and n4 <= (1 << 26):  # and the sharded sum equals the same MSM computed whole on ONE GPU (plain form, no table)
            whole = None
            for o in range(0, n4, 1 << 24):
                m_ = min(1 << 24, n4 - o)
                ds_ = h2.gen_scalars_device(0x5EED0001, m_, start=o, device=dev)
                dp_ = h2.gen_points_device(0x5EED0002, m_, start=o, device=dev)
                part_ = h2.msm_device(ds_, dp_)
                whole = part_ if whole is None else h2.g1_fold(np.stack([whole, part_]))
                del ds_, dp_
            config4["same_group_element_as_one_gpu"] = bool(np.array_equal(h2.g1_to_affine(whole), aff4))
            torch.cuda.empty_cache()
        sync_all()

    solo = rank == 0 and world == 1

    # ---- the other form of the same MSM (plain when the step is fixed-base and vice versa), rank 0, N = 1 ----
    other_form = None
    if solo and not (args.only_step or args.only_ntt):
        if args.form == "fixed":
            h2.bases_unpin_device(d_points)
        else:
            h2.bases_pin_device(d_points)
        ms_o, st_o, r_o = timed_msm(h2, d_scalars, d_points, 10)
        c_o = h2.get_msm_window(n) if args.form == "fixed" else h2.bases_pinned_info(d_points)[1]
        other_form = {"form": "plain" if args.form == "fixed" else "fixed", "window_bits": c_o, "windows": windows_of(c_o), "ms_per_msm": ms_o,
                      "value": n * windows_of(c_o) / (ms_o * 1e-3), "unit": "G1-adds/s", "stage_ms": st_o,
                      "same_group_element": bool(np.array_equal(h2.g1_to_affine(r_o), h2.g1_to_affine(result)))}
        if args.form == "fixed":
            h2.bases_pin_device(d_points)
        else:
            h2.bases_unpin_device(d_points)

    # ---- batched commit: B MSMs over the same bases in one call, rank 0 only ----
    batched = None
    if solo and args.batch > 1 and args.log_n <= 22:
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
        pts17 = h2.gen_points_device(0x5EED0002, n17, device=dev)

        def timed(f, reps=3):
            f()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):
                f()
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / reps

        k17 = {"count": 16}
        for form in ("plain", "fixed"):
            if form == "fixed":
                h2.bases_pin_device(pts17)
            k17[form] = {"ms_per_msm_fused_batch": timed(lambda: h2.msm_batch_device(cols, pts17)) / 16 * 1e3,
                         "ms_per_msm_one_call_each": timed(lambda: [h2.msm_device(c_, pts17) for c_ in cols]) / 16 * 1e3}
        h2.bases_unpin_device(pts17)
        k17["note"] = "batches of up to 2^18 pairs run fused: one sort / accumulate / reduce over the bucket sets of all MSMs"
        batched["k17"] = k17
        del cols, pts17

    # ---- NTT leg (BASELINE.json configs[2]: k = 22 NTT + iNTT), outside the MSM timed region ----
    def ntt_point(k, reps):
        d = h2.EvaluationDomain.new(2, k)
        d_a = h2.gen_scalars_device(0x5EED0003, 1 << k, device=dev)
        for _ in range(max(2, int(50.0 / 2.0 ** (k - 22)))):  # ~50 ms of untimed transforms: the clocks of an idle chip take that long to come up
            h2.ntt_device(d_a, d.omega, k)
            h2.ifft_device(d_a, d.omega_inv, k, d.ifft_divisor)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            h2.ntt_device(d_a, d.omega, k)
            h2.ifft_device(d_a, d.omega_inv, k, d.ifft_divisor)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / (2 * reps)
        gbps = NTT_BYTES_PER_ELEM * (1 << k) / (ms * 1e-3) / 1e9
        pt = {"log_n": k, "ms_per_transform": ms, "elems_per_s": (1 << k) / (ms * 1e-3),
              "roofline": {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                           # PMC bytes of one transform = sum over its passes (committed --pmc runs, 2^22 only)
                           "traffic": pmc_traffic("ntt_2p22") if k == 22 else None}}
        if k in NTT_FIELD_MUL_PER_ELEM:  # the roofline that binds: Fr multiplications per second against the multiplier's measured peak
            gmul = NTT_FIELD_MUL_PER_ELEM[k] * (1 << k) / (ms * 1e-3) / 1e9
            pt["valu_roofline"] = {"bound": "valu-int", "field_mul_per_elem": NTT_FIELD_MUL_PER_ELEM[k], "achieved": gmul, "peak": FIELD_MUL_PEAK_G,
                                   "unit": "Gmul/s", "frac": gmul / FIELD_MUL_PEAK_G}
        return pt

    def ntt_cpu_baseline(k):
        """the oracle's best_fft (arithmetic.rs:171-234 restated, same thread split) on the host cores: forward transform of 2^k elements,
        median of up to 5 runs within ~10 s; also the parity check of the GPU's forward transform on the same input"""
        from oracle import oracle
        d = h2.EvaluationDomain.new(2, k)
        d_a = h2.gen_scalars_device(0x5EED0003, 1 << k, device=dev)
        a = h2.to_numpy_u64(d_a).copy()
        h2.ntt_device(d_a, d.omega, k)
        torch.cuda.synchronize()
        got = h2.to_numpy_u64(d_a)
        hw = os.cpu_count() or 1
        res = {}
        want = None
        for T, budget in ((hw, 10.0), (min(16, hw), 8.0)):
            if T in res:
                continue
            oracle.best_fft(a[:1 << 12], h2.EvaluationDomain.new(2, 12).omega, 12, T)  # warm the thread pool
            ts = []
            stop = time.perf_counter() + budget
            while len(ts) < 5 and time.perf_counter() < stop:
                t1 = time.perf_counter()
                want = oracle.best_fft(a, d.omega, k, T)
                ts.append(time.perf_counter() - t1)
            res[T] = (sorted(ts)[len(ts) // 2], len(ts))
        t_hw, runs = res[hw]
        out = {"value": (1 << k) / t_hw, "unit": "elems/s", "cores": hw, "cpu_model": cpu_model(), "kind": "port",
               "sample": "full 2^%d-element forward transform, median of %d runs, %.3f s each, T = os.cpu_count() = %d threads; C restatement of "
                         "best_fft (serial bit-reversal and twiddle scan, then the rayon::join recursion), not the Rust binary" % (k, runs, t_hw, hw),
               "parity_vs_cpu": bool(np.array_equal(got, want))}
        sh = min(16, hw)
        if sh != hw:
            out["share_16_threads"] = {"cores": sh, "seconds": res[sh][0], "elems_per_s": (1 << k) / res[sh][0]}
        return out

    ntt = None
    if not args.no_ntt and rank == 0:
        ntt = ntt_point(args.ntt_log_n, 10)
        if solo and not args.no_cpu_baseline:
            ntt["cpu_baseline"] = ntt_cpu_baseline(args.ntt_log_n)
        if solo and not args.no_measure_traffic and 19 <= args.ntt_log_n <= 22:  # the two-pass plan's kernels; re-observed by this run as well
            live = measure_traffic(["--only-ntt", "--ntt-log-n", str(args.ntt_log_n)], [("ntt2_strided_kernel", 2.0), ("ntt2_final_kernel", 2.0)])
            if live:
                ntt["roofline"]["traffic_committed"] = ntt["roofline"]["traffic"]
                ntt["roofline"]["traffic"] = live["ntt2_strided_kernel"]["hbm_bytes_per_launch"] + live["ntt2_final_kernel"]["hbm_bytes_per_launch"]
                ntt["roofline"]["traffic_per_pass"] = live
                ntt["roofline"]["traffic_source"] = ("measured by this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE as two separate child passes of `bench.py --only-ntt` "
                                                     "(one transform = ntt2_strided_kernel + ntt2_final_kernel; KiB counters, the guide's x2 on the wide coalesced reads)")
            iss = measure_issue(["--only-ntt", "--ntt-log-n", str(args.ntt_log_n)], ["ntt2_strided_kernel", "ntt2_final_kernel"])
            if iss:
                ntt["valu_roofline"]["issue_slots"] = iss

    # ---- the sizes DESIGN.md quotes, measured by this run (rank 0, N = 1) ----
    sizes = None
    if solo and not args.no_sizes:
        sizes = {"msm": {}, "ntt": {}}
        for lg in (22, 24, 26):
            m = 1 << lg
            ds = h2.gen_scalars_device(0x5EED0001, m, device=dev)
            dp = h2.gen_points_device(0x5EED0002, m, device=dev)
            torch.cuda.synchronize()
            ent = {}
            reps = 3 if lg >= 24 else 5
            ms_p, st_p, r_p = timed_msm(h2, ds, dp, reps, warm=1)
            cp = h2.get_msm_window(m)
            ent["plain"] = {"window_bits": cp, "windows": windows_of(cp), "ms": ms_p, "g1_adds_per_s": m * windows_of(cp) / (ms_p * 1e-3),
                            "pairs_per_s": m / (ms_p * 1e-3), "stage_ms": st_p, "roofline_frac": msm_roofline(m, st_p["msm_accum"])["frac"]}
            t0 = time.perf_counter()
            h2.bases_pin_device(dp)
            pin_t = time.perf_counter() - t0
            inf = h2.bases_pinned_info(dp)
            ms_f, st_f, r_f = timed_msm(h2, ds, dp, reps, warm=1)
            ent["fixed"] = {"window_bits": inf[1], "windows": inf[2], "table_bytes": inf[3], "pin_s": pin_t, "ms": ms_f,
                            "g1_adds_per_s": m * inf[2] / (ms_f * 1e-3), "pairs_per_s": m / (ms_f * 1e-3), "stage_ms": st_f,
                            "roofline_frac": msm_roofline(m, st_f["msm_accum"])["frac"],
                            "same_group_element_as_plain": bool(np.array_equal(h2.g1_to_affine(r_p), h2.g1_to_affine(r_f)))}
            h2.bases_unpin_device(dp)
            sizes["msm"]["2^%d" % lg] = ent
            del ds, dp
            torch.cuda.empty_cache()
        for k in (20, 24, 26):
            sizes["ntt"]["2^%d" % k] = ntt_point(k, 5 if k < 26 else 3)
            torch.cuda.empty_cache()
        # host-pointer entry points (PCIe-inclusive, pageable caller memory): what the unpatched two-line drop-in pays
        sc, bs = h2.to_numpy_u64(d_scalars).copy(), h2.to_numpy_u64(d_points).copy()

        def host_ms(f, reps=5):
            f()
            ts = []
            for _ in range(reps):
                t1 = time.perf_counter()
                f()
                ts.append(time.perf_counter() - t1)
            return sorted(ts)[len(ts) // 2] * 1e3

        hp = {"msm_2p%d_unpinned_ms" % args.log_n: host_ms(lambda: h2.best_multiexp(sc, bs))}
        h2.bases_pin(bs)
        hp["msm_2p%d_pinned_ms" % args.log_n] = host_ms(lambda: h2.best_multiexp(sc, bs))
        # the same call with the whole upload ahead of the run (round 2's path), for the gain of streaming
        h2.lib().h2hip_debug_set_msm_stream(ctypes.c_uint32(1), ctypes.c_uint32(0), ctypes.c_size_t(0))
        hp["msm_2p%d_pinned_no_streaming_ms" % args.log_n] = host_ms(lambda: h2.best_multiexp(sc, bs))
        h2.lib().h2hip_debug_set_msm_stream(ctypes.c_uint32(0), ctypes.c_uint32(0), ctypes.c_size_t(0))
        h2.bases_unpin(bs)
        # the same unpatched call with HALO2_HIP_LAZY_PIN=2 (set here through the library's hook): the second sighting pins the array
        h2.lib().h2hip_debug_set_lazy_pin(ctypes.c_uint32(2))
        try:
            hp["msm_2p%d_lazy_pin_ms" % args.log_n] = host_ms(lambda: h2.best_multiexp(sc, bs))
        finally:
            h2.lib().h2hip_debug_set_lazy_pin(ctypes.c_uint32(0))
            try:
                h2.bases_unpin(bs)
            except h2.H2HipError:
                pass
        # BASELINE.json's north_star size through the same entry point: 2^24 pairs, 512 MiB of scalars crossing PCIe
        del sc, bs
        n24 = 1 << 24
        ds24 = h2.gen_scalars_device(0x5EED0001, n24, device=dev)
        dp24 = h2.gen_points_device(0x5EED0002, n24, device=dev)
        sc, bs = h2.to_numpy_u64(ds24).copy(), h2.to_numpy_u64(dp24).copy()
        h2.bases_pin_device(dp24)
        ms24, _, r24 = timed_msm(h2, ds24, dp24, 3, warm=1)
        h2.bases_unpin_device(dp24)
        del ds24, dp24
        torch.cuda.empty_cache()
        h2.bases_pin(bs)
        hp["msm_2p24_pinned_ms"] = host_ms(lambda: h2.best_multiexp(sc, bs), reps=3)
        hp["msm_2p24_device_resident_ms"] = ms24
        hp["msm_2p24_same_group_element"] = bool(np.array_equal(h2.g1_to_affine(h2.best_multiexp(sc, bs)), h2.g1_to_affine(r24)))
        h2.bases_unpin(bs)
        dk = h2.EvaluationDomain.new(2, args.ntt_log_n)
        a = h2.to_numpy_u64(h2.gen_scalars_device(3, 1 << args.ntt_log_n, device=dev)).copy()
        hp["ntt_2p%d_ms" % args.ntt_log_n] = host_ms(lambda: h2.best_fft(a, dk.omega, args.ntt_log_n), reps=3)
        hp["note"] = ("h2hip_msm_bn254 / h2hip_ntt_bn254_fr with host pointers (pageable caller memory): the scalars (and, unpinned, the bases) cross PCIe inside "
                      "the call, in chunks that stream under the work; lazy_pin = no h2hip_bases_pin call, HALO2_HIP_LAZY_PIN=2")
        sizes["host_pointer"] = hp
        del sc, bs, a
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import trace_bench
        sizes["trace_k17"] = trace_bench.run(h2, cpu=False)
        pl = trace_bench.run(h2, cpu=False, scalars="prover-like")  # SURVEY.md 8(d) config 5's second distribution
        sizes["trace_k17_prover_like"] = {k_: pl[k_] for k_ in ("scalars", "single_ms", "batched_ms", "reps_ms")}
        # the same 37 calls on HOST columns (what the shipped patches produce: 0001 + 0002 one call each, + 0004 / 0005 batched), PCIe included
        sizes["trace_k17_host"] = trace_bench.run_host(h2)
        import host_ntt_batch
        sizes["host_pointer"]["ntt_batch"] = host_ntt_batch.bench(h2, log_ns=(args.ntt_log_n,), counts=(4, 8))

    # ---- SURVEY.md 8(f).3 / (f).4 legs (extra keys; rank 0, N = 1 only), outside the MSM timed region ----
    next_rows = None
    if solo and not args.no_next_rows:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import evalh_bench
        import g2l_bench
        eh = evalh_bench.bench(evalh_bench.default_args(k=18, check_k=12, iters=5))
        gl = g2l_bench.bench(argparse.Namespace(k=[12, 16], cpu_k=12, threads=min(16, os.cpu_count() or 1)))
        next_rows = {
            "evaluate_h": {"workload": "2^20 extended rows (k = 18), %d gate polynomials, %d advice + %d fixed columns, %d permutation columns, "
                                       "%d lookups; device-resident" % (eh["gates"], eh["advice"], eh["fixed"], eh["perm_columns"], eh["lookups"]),
                           "ms_per_call": eh["k18"]["gpu_ms"], "rows_per_s": eh["k18"]["rows_per_s"], "gates_kernel": eh["k18"].get("gates_kernel"),
                           "stage_ms": eh["k18"].get("stage_ms"), "gates_valu_roofline": eh["k18"].get("gates_valu_roofline"),
                           "with_the_interpreter": eh["k18"].get("interpreter"), "first_call_with_inline_compile_s": eh["k18"].get("first_call_with_inline_compile_s"),
                           "parity_vs_cpu_k12_generated": eh["check_k12"].get("match_generated"),
                           "host_pointer_k16": (eh.get("host_k16") or {}).get("host_pointer"),
                           "parity_vs_cpu_k12": eh["check_k12"]["match"], "cpu_port_s_k12": eh["check_k12"]["oracle_s"],
                           "gpu_ms_k12": eh["check_k12"]["gpu_ms"]},
            "g_to_lagrange": {"k16_ms": gl["k16"]["gpu_ms"], "k16_scalar_muls_per_s": gl["k16"]["scalar_muls_per_s"],
                              "k12_ms": gl["k12"]["gpu_ms"], "cpu_port_s_k12": gl["k12"]["oracle_s"], "cpu_threads": gl["k12"]["oracle_threads"],
                              "parity_vs_cpu_k12": gl["k12"]["match"]},
        }

    # ---- CPU baseline (rank 0, N = 1 only): the oracle's best_multiexp on the same inputs ----
    cpu = None
    parity = None
    if solo and not args.no_cpu_baseline:
        from oracle import oracle
        sc, bs = h2.to_numpy_u64(d_scalars), h2.to_numpy_u64(d_points)
        hw = os.cpu_count() or 1  # rayon uses every hardware thread unless RAYON_NUM_THREADS says otherwise
        share = min(16, hw)       # the box's CPU share for one GPU

        def cpu_time(threads, budget_s):
            oracle.best_multiexp(sc[:4096], bs[:4096], threads)  # warm up the thread pool
            times, out = [], None
            stop = time.perf_counter() + budget_s
            while len(times) < 5 and time.perf_counter() < stop:
                t1 = time.perf_counter()
                out = oracle.best_multiexp(sc, bs, threads)
                times.append(time.perf_counter() - t1)
            return sorted(times)[len(times) // 2], len(times), out

        t_hw, runs_hw, cpu_out = cpu_time(hw, 15.0)
        cpu_c = oracle.window_c(n // hw)
        cpu = {
            "value": n * W / t_hw,  # same unit as `value`: the job's n*W adds per second of CPU time
            "unit": "G1-adds/s",
            "pairs_per_s": n / t_hw,
            "cores": hw,
            "cpu_model": cpu_model(),
            "kind": "port",
            "sample": "full 2^%d-pair MSM, median of %d runs, %.3f s each, T = os.cpu_count() = %d threads; C restatement of best_multiexp "
                      "(chunk = n/T per thread, c = %d unsigned windows), not the Rust binary" % (args.log_n, runs_hw, t_hw, hw, cpu_c),
        }
        if share != hw:
            t_sh, runs_sh, _ = cpu_time(share, 10.0)
            cpu["share_16_threads"] = {"cores": share, "seconds": t_sh, "pairs_per_s": n / t_sh, "value": n * W / t_sh,
                                       "note": "the same MSM on the 16 threads that are this GPU's share of the host"}
        parity = bool(np.array_equal(oracle.g1_to_affine(cpu_out), h2.g1_to_affine(result)))

    # ---- N > 1: ONE process driving all N GPUs through the C ABI (h2hip_init with N ids), while the ranks idle ----
    inlib = None
    if world > 1 and not args.no_inlib:
        # The other ranks wait on the host (a marker file; an RCCL barrier would spin on their GPUs, a gloo group prints to
        # stdout) while rank 0's child process uses all N GPUs.
        marker = os.path.join(os.environ.get("TMPDIR", "/tmp"), "h2bench_inlib_%s_%s.done" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "x")))
        if rank == 0:
            env = dict(os.environ)
            for k_ in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE", "ROLE_RANK"):
                env.pop(k_, None)
            if args.backend != "nccl":
                env["HALO2_HIP_ALLOW_DUPLICATE_DEVICES"] = "1"
            def run_inlib(extra_env, timeout):
                try:
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--inlib", str(world), "--log-n", str(args.log_n), "--steps", "7"],
                                       capture_output=True, text=True, timeout=timeout, env=dict(env, **extra_env))
                    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("INLIB ")]
                    return json.loads(lines[-1][6:]) if lines else {"error": (r.stdout + r.stderr)[-600:]}
                except subprocess.TimeoutExpired:
                    return {"error": "timed out after %d s" % timeout}
                except Exception as e:  # whatever happens here, the other ranks must be released
                    return {"error": repr(e)[:300]}

            try:
                inlib = run_inlib({"HALO2_HIP_GATHER": "host"}, 240)
                want_x0 = int(h2.g1_to_affine(result)[0])  # the ranks' shards are the same global sequence the child draws
                if "result_affine_x0" in inlib:
                    inlib["same_group_element_as_process_per_gpu"] = inlib["result_affine_x0"] == want_x0
                if args.backend == "nccl" and "error" not in inlib:  # distinct GPUs: the xGMI exchange of the same call
                    rc_ = run_inlib({"HALO2_HIP_GATHER": "rccl"}, 120)
                    if "result_affine_x0" in rc_:
                        rc_["same_group_element_as_process_per_gpu"] = rc_["result_affine_x0"] == want_x0
                    inlib["gather_rccl"] = {k_: rc_.get(k_) for k_ in ("ms_per_msm", "gather", "same_group_element_as_process_per_gpu", "error") if k_ in rc_}
            finally:
                with open(marker, "w") as f:
                    f.write("done")
        else:
            stop = time.time() + 420
            while not os.path.exists(marker) and time.time() < stop:
                time.sleep(0.05)
        sync_all()
        if rank == 0:
            try:
                os.remove(marker)
            except OSError:
                pass

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        adds = world * n * W * args.steps
        accum_ms = accum_ms_live or stages.get("msm_accum")
        roof = msm_roofline(n, accum_ms, "msm_2p%d_%s" % (args.log_n, args.form)) if accum_ms else None
        if roof and solo and not args.no_measure_traffic:  # the dominant kernel's HBM bytes, re-observed by this run
            live = measure_traffic(["--only-step", "--log-n", str(args.log_n), "--form", args.form, "--steps", "5", "--warmup", "2", "--prewarm-ms", "0"],
                                   [("msm_accum_kernel", 1.0)])
            live = live and live["msm_accum_kernel"]
            if live:
                roof["traffic_committed"] = roof["traffic"]
                roof["traffic"] = live["hbm_bytes_per_launch"]
                roof["traffic_read_write_bytes"] = [live["read_bytes"], live["write_bytes"]]
                roof["traffic_source"] = ("measured by this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE as two separate child passes of `bench.py --only-step` "
                                          "(%d / %d launches of msm_accum_kernel averaged; KiB counters, read factor 1 for 64-B gathers: profiles/pmc_traffic.json's note)" % tuple(live["launches"]))
        # second roofline, the one that actually binds: 256-bit modular multiplies per second against the
        # multiplier's measured chip-wide peak (tools/mul_rate.hip: 179 G/s for the explicit-mad form at >= 4 waves/SIMD)
        valu = None
        if accum_ms:
            gmul = n * W * FIELD_MUL_PER_BUCKET_ADD / (accum_ms * 1e-3) / 1e9
            valu = {"bound": "valu-int", "kernel": "msm_accum_kernel", "field_mul_per_add": FIELD_MUL_PER_BUCKET_ADD,
                    "achieved": gmul, "peak": FIELD_MUL_PEAK_G, "unit": "Gmul/s", "frac": gmul / FIELD_MUL_PEAK_G}
        if valu and solo and not args.no_measure_traffic:  # and how full the vector-issue slots were while it ran (SQ counters, one more child pass)
            iss = measure_issue(["--only-step", "--log-n", str(args.log_n), "--form", args.form, "--steps", "5", "--warmup", "2", "--prewarm-ms", "0"], ["msm_accum_kernel"])
            if iss:
                valu["issue_slots"] = iss["msm_accum_kernel"]
        out = {
            "metric": "bn254_msm_g1_adds_per_sec",
            "value": adds / elapsed,
            "unit": "G1-adds/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "prewarm_steps": prewarm_steps,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32x8 (256-bit Montgomery integers)",
            "data": "synthetic",
            "config": {"workload": "bn254_g1_msm_2p%d_per_gpu" % args.log_n, "pairs_per_gpu": n, "window_bits": c, "windows": W,
                       "signed_digits": True, "form": "fixed-base window table (bases pinned as ParamsKZG pins g / g_lagrange)" if args.form == "fixed"
                       else "plain (one bucket set per window)", "parallelism": "shard%d+allgather96B%s" % (world, "(async: gather of step i under the MSM of step i+1)" if world > 1 else "")},
            "pairs_per_s": world * n * args.steps / elapsed,
            "stage_ms": stages,
            "pin": pin,
            "roofline": roof,
            "valu_roofline": valu,
            "cpu_baseline": cpu,
            "parity_vs_cpu": parity,
            "other_form": other_form,
            "batched": batched,
            "ntt": ntt,
            "sizes": sizes,
            "next_rows": next_rows,
            "in_library_multi_gpu": inlib,
            "config4": config4,
        }
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
