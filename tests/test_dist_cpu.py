"""CPU, world_size 2 over gloo: the N > 1 path of bench.py -- shard by contiguous range,
all-gather the 96-byte partials, fold on every rank (arithmetic.rs:137-153) -- with the oracle
standing in for the GPU shard MSM (no GPU here)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, n, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_pkg
    from importlib import import_module
    from oracle import oracle
    h2 = load_pkg()
    h2dist = import_module("halo2_pse_amd.dist")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = h2dist.shard_range(n, rank, world)
    sc = oracle.gen_scalars(0x5EED0001, hi - lo, start=lo)
    bs = oracle.gen_points(0x5EED0002, hi - lo, start=lo)
    part = oracle.best_multiexp(sc, bs, 2)
    total = h2dist.allgather_fold(part, h2)
    # the pipelined form bench.py uses for N > 1: two gathers in flight on alternating staging slots, finished in order
    h0 = h2dist.allgather_start(part, slot=0)
    h1 = h2dist.allgather_start(oracle.best_multiexp(sc[:7], bs[:7], 1), slot=1)
    assert np.array_equal(h2.g1_to_affine(h2dist.allgather_finish(h0, h2)), h2.g1_to_affine(total))
    h2dist.allgather_finish(h1, h2)
    q.put((rank, h2.g1_to_affine(total).tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [1000, 1001])
def test_shard_allgather_fold_world2(oracle, n):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port + n % 7, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sc = oracle.gen_scalars(0x5EED0001, n)
    bs = oracle.gen_points(0x5EED0002, n)
    want = oracle.g1_to_affine(oracle.best_multiexp(sc, bs, 4)).tolist()
    assert got[0] == want and got[1] == want


def test_shard_range_covers_everything(h2):
    from importlib import import_module
    h2dist = import_module("halo2_pse_amd.dist")
    for n in (0, 1, 7, 8, 9, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [h2dist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
