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


def _worker_packed(rank, world, port, b, q, root_only=False, gather_traj=True, per=1):
    """PackedGather (the bench's zero-copy gather) on CPU tensors over gloo: two alternating buffer sets; all-gather or
    gather to rank 0, with or without the trajectories in the payload."""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as G
    pkg = G.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N = 10
    pg = pkg.sharding.PackedGather(b, N, True, torch.device("cpu"), root_only=root_only, gather_traj=gather_traj, batches_per_collective=per)
    ok = pg.active and pg.ws == world and not pg.overlap          # overlap needs RCCL
    ok = ok and pg.bytes_sent_per_rank == 8 * b * (10 + (2 * N if gather_traj else 0))
    for step in range(3):
        slot = step & 1
        pg.wait(slot)
        o = pg.outputs(slot)
        # what a solve would write: values that identify (rank, step, row, instance)
        o["out"].copy_(torch.arange(9 * b, dtype=torch.float64).reshape(9, b) + 1000.0 * rank + 1e5 * step)
        o["traj"].copy_(torch.arange(2 * N * b, dtype=torch.float64).reshape(2 * N, b) - 7.0 * rank)
        o["status"].copy_(torch.full((b,), rank, dtype=torch.int32))
        o["iters"].copy_(torch.arange(b, dtype=torch.int32) + 100 * step)
        pg.start(slot)
    pg.finish()
    r = pg.result(2 & 1)
    if root_only and rank != 0:                                    # a rank that only sends holds no gathered copy
        ok = ok and r is None and pg.full_res is None
        dist.barrier()
        q.put(bool(ok))
        dist.destroy_process_group()
        return
    if not gather_traj:
        ok = ok and r["traj"] is None
    for rr in range(world):
        ok = ok and torch.equal(r["out"][rr], torch.arange(9 * b, dtype=torch.float64).reshape(9, b) + 1000.0 * rr + 2e5)
        if gather_traj:
            ok = ok and torch.equal(r["traj"][rr], torch.arange(2 * N * b, dtype=torch.float64).reshape(2 * N, b) - 7.0 * rr)
        ok = ok and torch.equal(r["status"][rr], torch.full((b,), rr, dtype=torch.int32))
        ok = ok and torch.equal(r["iters"][rr], torch.arange(b, dtype=torch.int32) + 200)
    r1 = pg.result(1)                                              # the other buffer set still holds step 1
    ok = ok and torch.equal(r1["iters"][1 - rank], torch.arange(b, dtype=torch.int32) + 100)
    dist.barrier()
    q.put(bool(ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("root_only,gather_traj,per", [(False, True, 1), (True, True, 1), (True, False, 1), (True, True, 2), (False, False, 2)])
def test_packed_gather_two_ranks(root_only, gather_traj, per):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + 7 + 3 * int(root_only) + int(gather_traj) + 11 * per
    procs = [ctx.Process(target=_worker_packed, args=(r, 2, port, 48, q, root_only, gather_traj, per)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(res)


def test_packed_gather_single_process(pkg):
    pg = pkg.sharding.PackedGather(16, 10, False, torch.device("cpu"))
    assert not pg.active
    o = pg.outputs(0)
    assert o["traj"] is None and o["out"].shape == (9, 16) and o["status"].dtype == torch.int32
    o["out"].fill_(3.0); o["status"].fill_(2); o["iters"].fill_(11)
    pg.start(0); pg.finish()
    r = pg.result(0)
    assert r["out"].shape == (1, 9, 16) and float(r["out"].sum()) == 3.0 * 9 * 16
    assert int(r["status"].sum()) == 32 and int(r["iters"].sum()) == 176


@pytest.mark.gpu
def test_direct_gather_two_processes_one_gpu():
    """bench.py --gather direct with two ranks sharing the one GPU (--single-device; gloo carries the control messages, CUDA/HIP IPC
    the data): rank 1's solver writes its results straight into rank 0's buffers, rank 0 checks its copy against rank 1's own sums."""
    import json, subprocess, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-device", "--backend", "gloo", "--steps", "4", "--warmup", "2",
                        "--batch", "8192", "--gather", "direct", "--no-legs", "--no-cpu-baseline", "--no-host-leg", "--full-json", "/tmp/direct_gather_full.json"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    r = json.loads(line[-1])
    assert r["n_gpus"] == 2 and r["config"]["gather_checked"] is True
    assert r["config"]["collective_mode"].startswith("direct"), r["config"]["collective_mode"]
    assert r["converged_fraction"] > 0.99 and r["value"] > 0
