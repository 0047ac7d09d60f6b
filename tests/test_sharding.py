"""Multi-process path (N>1 ranks) on CPU with gloo, world_size 2: shard -> local solve -> one all_gather.
The local solve is injected: here the TEST-ONLY host twin stands in for the HIP path (which needs a GPU)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, B, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import __graft_entry__ as G
    from helpers import twin_solve
    pkg = G.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    params = pkg.params_from_json(os.path.join(ROOT, "tests", "golden", "config-fast.json"))
    wp = pkg.scenarios.load_waypoints(os.path.join(ROOT, "tests", "golden", "lake_track_waypoints.csv"))
    batch = pkg.scenarios.lake_track_batch(B, params, wp, seed=31)      # same on every rank (counter-based PRNG)
    shard, (lo, hi) = pkg.sharding.shard_batch(batch, world, rank)
    twin = C.CDLL(os.path.join(ROOT, "tests", "host_twin", "libhost_twin.so"))
    r = twin_solve(twin, params, shard)
    local = {k: (torch.from_numpy(v) if v is not None else None) for k, v in r.items()}
    full = pkg.sharding.gather_results(local, B)
    dist.barrier()
    if rank == 0:
        ref = twin_solve(twin, params, batch)
        q.put({k: (np.array_equal(full[k].numpy(), ref[k]) if ref[k] is not None else True) for k in ref})
        q.put((lo, hi))
    dist.destroy_process_group()


def test_shard_bounds(pkg):
    sb = pkg.sharding.shard_bounds
    for B in (0, 1, 7, 64, 65536, 65537):
        for ws in (1, 2, 3, 8):
            cuts = [sb(B, ws, r) for r in range(ws)]
            assert cuts[0][0] == 0 and cuts[-1][1] == B
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(ws - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("B", [64, 37])          # even and ragged split
def test_two_rank_shard_solve_gather(host_twin, B):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + B
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    same = q.get(timeout=240)
    cut = q.get(timeout=60)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(same.values()), same
    assert cut == (0, (B + 1) // 2)
