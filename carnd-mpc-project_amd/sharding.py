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

    The solver writes its results straight into packed buffers per rank,
        results [slot][9 out rows | status and iters as 2 x int32 per instance (one float64 row, two float32 rows)][instance]
        trajectories [slot][2N][instance]   (optional)
    which are collected with ONE collective per `group` consecutive batches (two when the trajectories travel too): no
    packing kernels, no re-layout.
      root_only=False   all_gather_into_tensor: every rank ends up with the results of every rank
      root_only=True    gather to rank 0 (what BASELINE.json's north_star asks for: "RCCL only for a final gather"): the other
                        ranks only send -- 1/ws of the inbound bytes per rank, and nothing is written on them
      gather_traj=False results only: the trajectories stay on their rank
      group=g           a collective costs the solve 8-10 % by being there, whatever it carries (one-rank rehearsal,
                        DESIGN.md section 7): g batches share one.  A batch's results are then available when its group's
                        collective has finished (finish() flushes a partial group).
    `slots` buffer sets alternate (slots must be a multiple of group), and with overlap=True the collective is issued
    asynchronously: it runs while the next batches are being solved, and a slot is reused only after its collective has
    completed.
    """

    def __init__(self, b, N, want_traj, device, dist=None, group=None, overlap=True, slots=2, dtype=None, force=False,
                 root_only=False, gather_traj=True, batches_per_collective=1, direct=False):
        import torch
        if dist is None:
            import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        # force=True runs the collective even with a single rank (rehearsal of the RCCL path on a one-GPU box)
        self.active = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force)
        import os as _os
        if _os.environ.get("MPC_BENCH_NO_COLLECTIVE_CALLS"):      # measurement aid: the communicator exists, nothing is ever issued
            self.active = False
        self.ws = dist.get_world_size(group) if self.active else 1
        self.rank = dist.get_rank(group) if self.active else 0
        self.b, self.N, self.want_traj = int(b), int(N), bool(want_traj)
        self.dtype = dtype if dtype is not None else torch.float64      # float32 for an MPC_PRECISION_F32 handle
        # status and iters are 2 x int32 per instance: one row of float64, two rows of float32
        self.int_rows = 1 if self.dtype == torch.float64 else 2
        self.res_rows = 9 + self.int_rows
        self.gather_traj = bool(gather_traj) and self.want_traj
        self.root_only = bool(root_only)
        self.nccl = self.active and dist.get_backend(group) == "nccl"
        self.overlap = bool(overlap) and self.nccl
        self.g = max(1, int(batches_per_collective))
        self.slots = (int(slots) + self.g - 1) // self.g * self.g
        self.is_cuda = torch.device(device).type == "cuda"
        # direct=True (GPUs of one node): NO collective in the data path.  Rank 0 owns the gathered buffers, every other rank maps
        # them (CUDA/HIP IPC: dmabuf handles, HSA_ENABLE_IPC_MODE_LEGACY=0) and its solver -- launches and tail slices alike -- writes
        # its results straight into its own region of rank 0's memory over xGMI.  A batch is gathered the moment it is final.
        self.direct = False
        if direct and self.active and self.is_cuda:
            try:
                self._map_direct(device)
                self.direct = True
            except Exception as e:                              # never a silent change of path: `mode` says what ran and why
                self.direct_error = "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:120] if str(e) else "")
        if self.direct:
            self.work, self.ready, self.filled = [], [], []
            self.collective_name = "direct remote write"
            self.bytes_sent_per_rank = (self.res_rows + (2 * self.N if self.gather_traj else 0)) * self.b * (8 if self.dtype == torch.float64 else 4)
            self.mode = ("direct: every rank's solver writes its results into rank 0's buffers (IPC-mapped, over xGMI); no collective in the "
                         "data path" + ("" if self.gather_traj or not self.want_traj else ", results only (trajectories stay on their rank)"))
            return
        self.res = torch.zeros((self.slots, self.res_rows, self.b), dtype=self.dtype, device=device)
        self.trj = torch.zeros((self.slots, 2 * self.N, self.b), dtype=self.dtype, device=device) if self.want_traj else None
        holds_full = self.active and (not self.root_only or self.rank == 0)
        # gathered copies: [rank][slot][row][instance]
        self.full_res = torch.zeros((self.ws, self.slots, self.res_rows, self.b), dtype=self.dtype, device=device) if holds_full else None
        self.full_trj = torch.zeros((self.ws, self.slots, 2 * self.N, self.b), dtype=self.dtype, device=device) if holds_full and self.gather_traj else None
        self.work = [None] * (self.slots // self.g)         # per group: list of outstanding collectives
        self.ready = [None] * self.slots                    # per slot: event recorded behind the solve that filled it
        self.filled = [0] * (self.slots // self.g)          # per group: slots handed to start() since the last collective
        self.collective_name = "gather to rank 0" if self.root_only else "all_gather_into_tensor"
        self.bytes_sent_per_rank = (self.res_rows + (2 * self.N if self.gather_traj else 0)) * self.b * (8 if self.dtype == torch.float64 else 4)
        # which collective the timed region really ran: reported in bench.py's JSON line (never a silent fallback)
        if not self.active:
            self.mode = "none (single rank)"
        elif self.nccl:
            self.mode = "rccl %s, %s" % (self.collective_name, "async (overlapped with the next solve)" if self.overlap else "synchronous")
        else:
            self.mode = "%s %s through host memory, synchronous (rehearsal backend)" % (dist.get_backend(group), self.collective_name)
        if self.active and not self.gather_traj and self.want_traj:
            self.mode += ", results only (trajectories stay on their rank)"
        if self.active and self.g > 1:
            self.mode += ", one collective per %d batches" % self.g
        if direct and not self.direct and self.active:
            self.mode += " (direct remote write not available: %s)" % getattr(self, "direct_error", "not a CUDA device")

    def _map_direct(self, device):
        torch, dist = self.torch, self.dist
        shapes = {"res": (self.ws, self.slots, self.res_rows, self.b)}
        if self.want_traj:
            shapes["trj"] = (self.ws, self.slots, 2 * self.N, self.b)
        full = {}
        if self.ws == 1:
            for k, shp in shapes.items():
                full[k] = torch.zeros(shp, dtype=self.dtype, device=device)
        else:
            metas = [None]
            if self.rank == 0:
                try:
                    for k, shp in shapes.items():
                        full[k] = torch.zeros(shp, dtype=self.dtype, device=device)
                    torch.cuda.synchronize(device)
                    metas = [{k: t.untyped_storage()._share_cuda_() for k, t in full.items()}]
                except Exception as e:
                    metas = [("error", "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:100] if str(e) else ""))]
            dist.broadcast_object_list(metas, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            if isinstance(metas[0], tuple) and metas[0][0] == "error":
                raise RuntimeError("rank 0 could not export its buffers: " + metas[0][1])
            mapped, why = 1, ""
            if self.rank != 0:
                try:
                    for k, shp in shapes.items():
                        m = list(metas[0][k])
                        # opened on THIS rank's device (lazy peer access from it to rank 0's memory), not on the exporting one
                        m[0] = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
                        st = torch.UntypedStorage._new_shared_cuda(*m)
                        n = 1
                        for d in shp:
                            n *= d
                        full[k] = torch.empty(0, dtype=self.dtype, device=device).set_(st, 0, (n,)).view(shp)
                except Exception as e:                       # every rank must take the same path: the verdict is agreed on below
                    mapped, why = 0, "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:100] if str(e) else "")
            flags = [None] * self.ws
            dist.all_gather_object(flags, (mapped, why), group=self.group)
            if not all(f[0] for f in flags):
                raise RuntimeError("mapping failed on rank(s) %s" % [(r, f[1]) for r, f in enumerate(flags) if not f[0]])
            dist.barrier(group=self.group)
            # probe: every rank writes its number into the first words of its region through the mapping; rank 0 must read them back
            if self.rank != 0:
                full["res"][self.rank, 0, 0, :8].fill_(float(self.rank))
                torch.cuda.synchronize(device)
            dist.barrier(group=self.group)
            ok = torch.ones(1, dtype=torch.int32, device=device)
            if self.rank == 0:
                torch.cuda.synchronize(device)
                for r in range(1, self.ws):
                    if not bool((full["res"][r, 0, 0, :8] == float(r)).all()):
                        ok.zero_()
            if dist.get_backend(self.group) == "nccl":
                dist.broadcast(ok, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            else:
                okc = ok.cpu(); dist.broadcast(okc, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group); ok = okc
            if int(ok.item()) != 1:
                raise RuntimeError("a rank's write through the mapping did not arrive in rank 0's buffer")
            if self.rank != 0:
                full["res"][self.rank, 0, 0, :8].zero_()
                torch.cuda.synchronize(device)
            dist.barrier(group=self.group)
        self.full_res = full["res"]
        self.full_trj = full.get("trj") if self.gather_traj else None
        self.res = full["res"][self.rank]
        # trajectories that are not gathered stay in a buffer of the rank's own
        self.trj = (full["trj"][self.rank] if self.gather_traj else torch.zeros((self.slots, 2 * self.N, self.b), dtype=self.dtype, device=device)) if self.want_traj else None

    def outputs(self, slot):
        """The tensors to hand to BatchedMPC.solve_torch(outputs=...): views into the packed buffers of `slot`."""
        p = self.res[slot]
        ints = p[9:9 + self.int_rows].reshape(-1).view(self.torch.int32)   # 2*b int32
        return {"out": p[0:9], "traj": self.trj[slot] if self.want_traj else None, "status": ints[:self.b], "iters": ints[self.b:2 * self.b]}

    def wait(self, slot):
        """Make the current stream wait for the collective that last carried `slot` (before the slot is written again)."""
        if self.direct:
            return
        gi = slot // self.g
        if self.work[gi]:
            for w in self.work[gi]:
                w.wait()
            self.work[gi] = None

    def _one(self, full, src, lo, hi, async_op):
        dist = self.dist
        part = src[lo:hi]                                     # [g, rows, b], contiguous
        if self.root_only:
            lst = [full[r, lo:hi] for r in range(self.ws)] if self.rank == 0 else None
            return dist.gather(part, gather_list=lst, dst=dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                               group=self.group, async_op=async_op)
        if self.g == self.slots:
            return dist.all_gather_into_tensor(full, part, group=self.group, async_op=async_op)
        return dist.all_gather([full[r, lo:hi] for r in range(self.ws)], part, group=self.group, async_op=async_op)

    def _collect(self, gi, async_op):
        lo, hi = gi * self.g, (gi + 1) * self.g
        ws = [self._one(self.full_res, self.res, lo, hi, async_op)]
        if self.gather_traj:
            ws.append(self._one(self.full_trj, self.trj, lo, hi, async_op))
        return ws

    def _issue(self, gi):
        dist, torch = self.dist, self.torch
        lo, hi = gi * self.g, (gi + 1) * self.g
        self.filled[gi] = 0
        if self.nccl:
            if self.is_cuda:                                   # the members of the group were solved on other streams
                cur = torch.cuda.current_stream()
                for s in range(lo, hi):
                    if self.ready[s] is not None:
                        cur.wait_event(self.ready[s])
            try:
                ws = self._collect(gi, self.overlap)
            except RuntimeError as e:
                if not self.overlap:
                    raise
                self.overlap = False                      # fall back to the synchronous collective -- and say so
                self.mode = "rccl %s, synchronous (async_op failed: %s)" % (self.collective_name, str(e).splitlines()[0][:80])
                ws = self._collect(gi, False)
            self.work[gi] = ws if self.overlap else None
        else:   # gloo (CPU tests, single-GPU rehearsal): through host memory, synchronous
            for full, src in ((self.full_res, self.res), (self.full_trj, self.trj) if self.gather_traj else (None, None)):
                if src is None:
                    continue
                part = src[lo:hi]
                h = (part.cpu() if part.is_cuda else part).contiguous()
                if self.root_only:
                    parts = [torch.empty_like(h) for _ in range(self.ws)] if self.rank == 0 else None
                    dist.gather(h, gather_list=parts, dst=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
                    if self.rank == 0:
                        full[:, lo:hi].copy_(torch.stack(parts, dim=0))
                else:
                    parts = [torch.empty_like(h) for _ in range(self.ws)]
                    dist.all_gather(parts, h, group=self.group)
                    full[:, lo:hi].copy_(torch.stack(parts, dim=0))

    def start(self, slot):
        """Hand the packed buffers of `slot` to the gather (call on the stream the solve that filled them was enqueued on, after
        it).  The collective goes out when the last slot of the group has been handed over."""
        if not self.active or self.direct:
            return
        if self.nccl and self.is_cuda:
            ev = self.torch.cuda.Event()
            ev.record()
            self.ready[slot] = ev
        gi = slot // self.g
        self.filled[gi] += 1
        if self.filled[gi] >= self.g:
            self._issue(gi)

    def finish(self):
        """Flush partial groups and wait for every collective."""
        if self.direct:
            return
        if self.active:
            for gi in range(len(self.filled)):
                if self.filled[gi] > 0:
                    self._issue(gi)      # (slots of the group not filled this round carry what they held before)
        for gi in range(len(self.work)):
            self.wait(gi * self.g)

    def result(self, slot):
        """Views of the gathered batch: out [ws,9,b], traj [ws,2N,b] or None, status/iters [ws,b] (int32); None on a rank
        that only sent (root_only)."""
        if not self.active:
            o = self.outputs(slot)
            return {k: (v[None] if v is not None else None) for k, v in o.items()}
        if self.full_res is None:
            return None
        f = self.full_res[:, slot]
        ints = f[:, 9:9 + self.int_rows].contiguous().reshape(self.ws, -1).view(self.torch.int32).reshape(self.ws, 2 * self.b)
        return {"out": f[:, 0:9], "traj": self.full_trj[:, slot] if self.gather_traj else None, "status": ints[:, :self.b], "iters": ints[:, self.b:]}
