"""Multi-GPU sharding of a batch of MPC instances.

Instances are fully independent, so the batch is cut into contiguous shards, one per rank (one process
per GPU), solved with NO communication, and the per-instance results are gathered ONCE at the end with
a single all_gather (RCCL over xGMI when the backend is "nccl"; "gloo" in the CPU tests).  Shards may be
ragged (B not divisible by the world size): the gather pads to the largest shard.
"""
import numpy as np


def shard_bounds(B, world_size, rank):
    """Contiguous split of range(B): the first B % world_size ranks get one extra instance."""
    q, r = divmod(int(B), int(world_size))
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_batch(batch, world_size, rank, keys=("state", "coeffs", "yaw_lo", "yaw_hi", "weights")):
    B = batch["state"].shape[1]
    lo, hi = shard_bounds(B, world_size, rank)
    out = {}
    for k in keys:
        if batch.get(k) is None:
            continue
        a = batch[k]
        out[k] = np.ascontiguousarray(a[..., lo:hi])
    return out, (lo, hi)


def gather_results(local, B, dist=None, group=None):
    """local: dict of torch tensors of the local shard: out [9,b], status [b], iters [b], traj [2N,b] or None.
    Returns the same dict for the full batch on every rank (one all_gather of one packed buffer)."""
    import torch
    if dist is None:
        import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    ws, rank = dist.get_world_size(group), dist.get_rank(group)
    bmax = max(shard_bounds(B, ws, r)[1] - shard_bounds(B, ws, r)[0] for r in range(ws))
    rows = [local["out"]]
    if local.get("traj") is not None:
        rows.append(local["traj"])
    rows.append(local["status"].to(torch.float64)[None, :])
    rows.append(local["iters"].to(torch.float64)[None, :])
    packed = torch.cat(rows, dim=0)
    R, b = packed.shape
    buf = torch.zeros((R, bmax), dtype=torch.float64, device=packed.device)
    buf[:, :b] = packed
    # RCCL ("nccl") gathers device tensors in place; gloo (CPU tests / single-GPU rehearsal) goes through the host
    if dist.get_backend(group) != "nccl" and buf.is_cuda:
        dev = buf.device
        hbuf = buf.cpu()
        hg = [torch.empty_like(hbuf) for _ in range(ws)]
        dist.all_gather(hg, hbuf, group=group)
        gathered = [g.to(dev) for g in hg]
    else:
        gathered = [torch.empty_like(buf) for _ in range(ws)]
        dist.all_gather(gathered, buf, group=group)
    parts = []
    for r in range(ws):
        lo, hi = shard_bounds(B, ws, r)
        parts.append(gathered[r][:, :hi - lo])
    full = torch.cat(parts, dim=1)
    res = {"out": full[:9].contiguous()}
    ofs = 9
    if local.get("traj") is not None:
        nt = local["traj"].shape[0]
        res["traj"] = full[ofs:ofs + nt].contiguous(); ofs += nt
    else:
        res["traj"] = None
    res["status"] = full[ofs].to(torch.int32); res["iters"] = full[ofs + 1].to(torch.int32)
    return res
