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


class PackedGather:
    """Zero-copy gather for equal shards (the weak-scaling bench, and any batch divisible by the world size).

    The solver writes its results straight into one packed buffer per rank,
        rows 0..8 out | status and iters as 2 x int32 per instance (one float64 row, two float32 rows) | 2N trajectory rows (optional),
    of which the first `gather_rows` rows -- everything, or without the trajectories (gather_traj=False: results only) --
    are collected with ONE collective per batch: no packing kernels, no re-layout.
      root_only=False   all_gather_into_tensor: every rank ends up with full[rank, row, instance]
      root_only=True    gather to rank 0 (what BASELINE.json's north_star asks for: "RCCL only for a final gather"): the other
                        ranks only send -- 1/ws of the inbound bytes per rank, and nothing is written on them
    `slots` buffer sets alternate, and with overlap=True the collective is issued asynchronously: the gather of batch i
    runs while batch i+1 is being solved, and a slot is reused only after its gather has completed.
    """

    def __init__(self, b, N, want_traj, device, dist=None, group=None, overlap=True, slots=2, dtype=None, force=False,
                 root_only=False, gather_traj=True):
        import torch
        if dist is None:
            import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        # force=True runs the collective even with a single rank (rehearsal of the RCCL path on a one-GPU box)
        self.active = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force)
        self.ws = dist.get_world_size(group) if self.active else 1
        self.rank = dist.get_rank(group) if self.active else 0
        self.b, self.N, self.want_traj = int(b), int(N), bool(want_traj)
        self.dtype = dtype if dtype is not None else torch.float64      # float32 for an MPC_PRECISION_F32 handle
        # status and iters are 2 x int32 per instance: one row of float64, two rows of float32
        self.int_rows = 1 if self.dtype == torch.float64 else 2
        self.rows = 9 + self.int_rows + (2 * self.N if want_traj else 0)
        self.gather_traj = bool(gather_traj) and self.want_traj
        self.gather_rows = 9 + self.int_rows + (2 * self.N if self.gather_traj else 0)
        self.root_only = bool(root_only)
        self.nccl = self.active and dist.get_backend(group) == "nccl"
        self.overlap = bool(overlap) and self.nccl
        self.slots = int(slots)
        self.pack = [torch.zeros((self.rows, self.b), dtype=self.dtype, device=device) for _ in range(self.slots)]
        holds_full = self.active and (not self.root_only or self.rank == 0)
        self.full = [torch.zeros((self.ws, self.gather_rows, self.b), dtype=self.dtype, device=device) if holds_full else None
                     for _ in range(self.slots)]
        self.work = [None] * self.slots
        self.collective_name = "gather to rank 0" if self.root_only else "all_gather_into_tensor"
        self.bytes_sent_per_rank = self.gather_rows * self.b * (8 if self.dtype == torch.float64 else 4)
        # which collective the timed region really ran: reported in bench.py's JSON line (never a silent fallback)
        if not self.active:
            self.mode = "none (single rank)"
        elif self.nccl:
            self.mode = "rccl %s, %s" % (self.collective_name, "async (overlapped with the next solve)" if self.overlap else "synchronous")
        else:
            self.mode = "%s %s through host memory, synchronous (rehearsal backend)" % (dist.get_backend(group), self.collective_name)
        if self.active and not self.gather_traj and self.want_traj:
            self.mode += ", results only (trajectories stay on their rank)"

    def outputs(self, slot):
        """The tensors to hand to BatchedMPC.solve_torch(outputs=...): views into the packed buffer of `slot`."""
        p = self.pack[slot]
        ints = p[9:9 + self.int_rows].reshape(-1).view(self.torch.int32)   # 2*b int32
        return {"out": p[0:9], "traj": p[9 + self.int_rows:9 + self.int_rows + 2 * self.N] if self.want_traj else None,
                "status": ints[:self.b], "iters": ints[self.b:2 * self.b]}

    def wait(self, slot):
        """Make the current stream wait for the gather that last used `slot` (before the slot is written again)."""
        w = self.work[slot]
        if w is not None:
            w.wait()
            self.work[slot] = None

    def _collective(self, slot, async_op):
        dist, src = self.dist, self.pack[slot][:self.gather_rows]
        if self.root_only:
            lst = [self.full[slot][r] for r in range(self.ws)] if self.rank == 0 else None
            return dist.gather(src, gather_list=lst, dst=dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                               group=self.group, async_op=async_op)
        return dist.all_gather_into_tensor(self.full[slot], src, group=self.group, async_op=async_op)

    def start(self, slot):
        """Gather the packed buffer of `slot` (call after the solve that filled it has been enqueued)."""
        if not self.active:
            return
        dist, torch = self.dist, self.torch
        if self.nccl:
            try:
                w = self._collective(slot, self.overlap)
            except RuntimeError as e:
                if not self.overlap:
                    raise
                self.overlap = False                      # fall back to the synchronous collective -- and say so
                self.mode = "rccl %s, synchronous (async_op failed: %s)" % (self.collective_name, str(e).splitlines()[0][:80])
                w = self._collective(slot, False)
            self.work[slot] = w if self.overlap else None
        else:   # gloo (CPU tests, single-GPU rehearsal): through host memory, synchronous
            src = self.pack[slot][:self.gather_rows]
            h = (src.cpu() if src.is_cuda else src).contiguous()
            if self.root_only:
                parts = [torch.empty_like(h) for _ in range(self.ws)] if self.rank == 0 else None
                dist.gather(h, gather_list=parts, dst=0, group=self.group)
                if self.rank == 0:
                    self.full[slot].copy_(torch.stack(parts, dim=0))
            else:
                parts = [torch.empty_like(h) for _ in range(self.ws)]
                dist.all_gather(parts, h, group=self.group)
                self.full[slot].copy_(torch.stack(parts, dim=0))

    def finish(self):
        for s in range(self.slots):
            self.wait(s)

    def result(self, slot):
        """Views of the gathered batch: out [ws,9,b], traj [ws,2N,b] or None, status/iters [ws,b] (int32); None on a rank
        that only sent (root_only)."""
        if not self.active:
            o = self.outputs(slot)
            return {k: (v[None] if v is not None else None) for k, v in o.items()}
        f = self.full[slot]
        if f is None:
            return None
        ints = f[:, 9:9 + self.int_rows].contiguous().reshape(self.ws, -1).view(self.torch.int32).reshape(self.ws, 2 * self.b)
        return {"out": f[:, 0:9], "traj": f[:, 9 + self.int_rows:9 + self.int_rows + 2 * self.N] if self.gather_traj else None,
                "status": ints[:, :self.b], "iters": ints[:, self.b:]}
