#!/usr/bin/env python3
"""bench.py -- MPC solves/sec of the batched HIP path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path (the batched replacement of MPC::solve(), src/control/MPC.cpp:183-325)
over one batch of synthetic inputs that are ALREADY RESIDENT IN HBM.  Workload at every N: BASELINE.json
configs[2] per GPU -- 65 536 lake-track states with 100 ms latency compensation, N=10, dt=0.1,
config-fast.json, fp64, trajectories requested -- i.e. weak scaling: each rank solves its own 65 536
instances (different PRNG streams), and every batch's per-instance results are gathered with a single
all_gather_into_tensor (RCCL) inside the timed region.  Rank 0 prints ONE JSON line.

Steps are pipelined the way a serving loop would run them: `--inflight` (default 2) handles on separate streams,
so the next batch's waves take the SIMDs that the previous launch frees in its tail, and the gather of batch i
overlaps the solve of batch i+1 (sharding.PackedGather).  All K steps, solves and gathers, complete inside the
timed region (barrier + synchronize on both sides).

roofline: bound "hbm" with ALGORITHMIC bytes = 336 B/solve (in 104 + out 72 +
trajectory 160) x solves per launch / the solve kernel's average launch duration measured live with HIP
events on the launch stream.  The path is fp64-VALU/latency bound, so the HBM fraction is tiny by
construction; `fp64_valu_frac` next to it prices the timed region against the 78.6 TFLOP/s vector peak
using the algorithmic flop count of section 8d (2.5 kflop x stages x iterations).
cpu_baseline: the oracle (oracle/mpc_oracle.c, kind "port"), one thread, a bounded sample of the same batch.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_SOLVE = 8 * (6 + 5 + 2) + 8 * 9 + 8 * 2 * 10   # 104 in + 72 out + 160 trajectory = 336 (N=10)
HBM_PEAK_GBS = 8000.0                                          # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_VALU_PEAK_TFLOPS = 78.6
KFLOP_PER_STAGE_ITER = 2.5


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=65536, help="instances per GPU")
    ap.add_argument("--config", default="config-fast.json")
    ap.add_argument("--no-traj", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--N", type=int, default=0, help="override Config::N (BASELINE.json configs[3]: 25)")
    ap.add_argument("--dt", type=float, default=0.0, help="override Config::dt (configs[3]: 0.05)")
    ap.add_argument("--weights-sweep", action="store_true", help="per-instance Config::weights (configs[4])")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo only to "
                    "rehearse the multi-process path on a single GPU)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--inflight", type=int, default=2, help="batches in flight per GPU (handles on separate streams)")
    ap.add_argument("--no-overlap", action="store_true", help="issue the gather synchronously on the solve stream")
    ap.add_argument("--no-priority-stream", action="store_true", help="run the solves on a normal-priority stream")
    args = ap.parse_args()

    import numpy as np
    import torch
    import __graft_entry__ as G
    pkg = G.load_package()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("--gpus %d needs torch.distributed.run (one process per GPU)" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    golden = os.path.join(ROOT, "tests", "golden")
    over = {}
    if args.N:
        over["N"] = args.N
    if args.dt:
        over["dt"] = args.dt
    params = pkg.params_from_json(os.path.join(golden, args.config), **over)
    wp = pkg.scenarios.load_waypoints(os.path.join(golden, "lake_track_waypoints.csv"))
    B = args.batch
    want_traj = not args.no_traj
    # weak scaling: every rank draws its own instances (rank-specific PRNG stream)
    batch = pkg.scenarios.lake_track_batch(B, params, wp, stream=3 + 16 * rank)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_state, d_coef, d_ylo, d_yhi = t(batch["state"]), t(batch["coeffs"]), t(batch["yaw_lo"]), t(batch["yaw_hi"])
    w_np = pkg.scenarios.weight_sweep(B, params, seed=1234 + rank) if args.weights_sweep else None
    d_w = t(w_np) if w_np is not None else None
    # Batches in flight: a launch of 65 536 instances is exactly one wave per SIMD and lasts as long as its slowest wave
    # (25 iterations) while the average wave is done after ~70 % of that time; a second handle on a second stream lets
    # the next batch's waves take the SIMDs as they become free (measured: 2.2 -> 1.5 ms per batch).
    nfl = max(1, args.inflight)
    mpcs = [pkg.BatchedMPC(params, B, device=local_rank) for _ in range(nfl)]
    mpc = mpcs[0]
    # results go straight into a packed buffer that is gathered with one all_gather_into_tensor; two buffer sets
    # alternate so that the gather of batch i overlaps the solve of batch i+1 (sharding.PackedGather)
    pg = pkg.sharding.PackedGather(B, params.N, want_traj, dev, dist if dist is not None else None, overlap=not args.no_overlap,
                                   slots=max(2, nfl))
    outs = pg.outputs(0)

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    nstep = 0

    def step(ev=None):
        nonlocal nstep
        slot = nstep % pg.slots
        h = mpcs[nstep % nfl]
        with torch.cuda.stream(streams[nstep % nfl]):
            pg.wait(slot)                               # the gather that last read this buffer set has finished
            if ev is not None:
                ev[0].record()
            h.solve_torch(d_state, d_coef, d_ylo, d_yhi, weights=d_w, outputs=pg.outputs(slot))   # async on this stream
            if ev is not None:
                ev[1].record()
            pg.start(slot)                              # the path's only collective
        nstep += 1

    # The solves run on a high-priority stream: when a batch's gather (RCCL's own stream, normal priority) and the next
    # batch's solve become ready together, the solve's 1 024 waves are placed first and the collective's workgroups take
    # the SIMDs the solve frees in its tail, instead of holding SIMDs that 512-register waves cannot share.
    torch.cuda.synchronize(dev)
    streams = [torch.cuda.Stream(device=dev, priority=0 if args.no_priority_stream else -1) for _ in range(nfl)]
    for _ in range(args.warmup):
        step()
    pg.finish()
    sync_all()
    # HIP events on the stream the kernel is launched on: torch's current stream, whose handle is what
    # solve_torch passes to mpc_solve_batch_device (a NULL handle is HIP's null stream = torch's default)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(ev[i])
    pg.finish()
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    kernel_ms_avg = float(np.mean(kernel_ms))
    stats = mpcs[(nstep - 1) % nfl].stats()
    last = (nstep - 1) % pg.slots
    # the same launch alone on the device (nothing else in flight), for reference
    iso = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(streams[0]):
            e0.record(); mpcs[0].solve_torch(d_state, d_coef, d_ylo, d_yhi, weights=d_w, outputs=pg.outputs(0)); e1.record()
        torch.cuda.synchronize(dev)
        iso.append(e0.elapsed_time(e1))
    outs = pg.outputs(last)
    status = outs["status"].cpu().numpy()
    out_np = outs["out"].cpu().numpy()
    # the gathered copy of this rank's shard must be what the solver wrote
    g = pg.result(last)
    gather_ok = bool(torch.equal(g["out"][rank if dist is not None else 0], outs["out"]) and
                     torch.equal(g["status"][rank if dist is not None else 0], outs["status"]))

    if rank != 0:
        for h in mpcs:
            h.close()
        if dist is not None:
            dist.barrier(); dist.destroy_process_group()
        return

    total_solves = B * world * args.steps
    value = total_solves / elapsed
    mean_iters = stats.iter_sum / max(1, stats.batch)
    stages = params.N - 1
    res = {
        "metric": "MPC solves/sec (batch) at N=%d dt=%g" % (params.N, params.dt),
        "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s%d lake-track states per GPU, 100 ms latency compensation, N=%d dt=%g, %s, trajectories %s%s"
                               % ("BASELINE.json configs[2]: " if (params.N == 10 and not args.weights_sweep and B == 65536) else "",
                                  B, params.N, params.dt, args.config, "on" if want_traj else "off",
                                  ", per-instance weight sweep" if args.weights_sweep else ""),
                   "batch_per_gpu": B, "global_batch": B * world, "N": params.N, "dt": params.dt,
                   "parallelism": "%d independent shard(s), one all_gather_into_tensor of the packed results per batch%s"
                                  % (world, ", overlapped with the next batch's solve" if (pg.overlap) else ""),
                   "gather_checked": gather_ok, "batches_in_flight": nfl,
                   "branch_mode": "frozen", "tol": params.tol, "max_iter": params.max_iter},
        "converged_fraction": float((status == 0).mean()),
        "status_counts": {pkg.STATUS_NAMES[k]: int((status == k).sum()) for k in range(5)},
        "mean_iterations": mean_iters, "max_iterations": int(stats.iter_max),
    }
    per_solve = 8 * (6 + 5 + 2) + 8 * 9 + (8 * 2 * params.N if want_traj else 0) + (8 * 12 if args.weights_sweep else 0)
    algo_bytes = per_solve * B
    achieved_gbs = algo_bytes / (kernel_ms_avg * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "round1_pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            pj = json.load(open(pmc))
            if pj.get("batch") == B and pj.get("config") == args.config and params.N == 10 and not args.weights_sweep and want_traj:
                traffic = pj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    res["roofline"] = {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                       "kernel": "mpc_solve_kernel", "kernel_ms_avg": kernel_ms_avg, "kernel_ms_min": float(np.min(kernel_ms)),
                       "kernel_ms_alone": float(np.median(iso)),
                       "algorithmic_bytes_per_launch": algo_bytes,
                       "note": "fp64-VALU/latency-bound path: HBM fraction is small by construction (SURVEY.md 8d); "
                               "kernel_ms_avg is the duration of a launch that shares the device with the other batch in flight, "
                               "kernel_ms_alone the same launch by itself",
                       # whole-device rate of this rank: with several batches in flight a launch shares the SIMDs, so the
                       # flop rate is taken over the timed region, not over one launch's duration
                       "fp64_valu_tflops": B * mean_iters * stages * KFLOP_PER_STAGE_ITER * 1e3 / (elapsed / args.steps) / 1e12,
                       "fp64_valu_frac": B * mean_iters * stages * KFLOP_PER_STAGE_ITER * 1e3 / (elapsed / args.steps) / 1e12 / FP64_VALU_PEAK_TFLOPS}

    if world == 1 and not args.no_cpu_baseline:
        # the checker, timed as the CPU baseline: oracle = plain-C restatement of the reference algorithm
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        cfg = O.load_config(args.config, **over)
        n_done, worst_steer, worst_acc, t_cpu0 = 0, 0.0, 0.0, time.perf_counter()
        t_solve = 0.0
        while n_done < min(B, 4096) and (time.perf_counter() - t_cpu0) < args.cpu_seconds:
            i = n_done
            cfg.yaw_low, cfg.yaw_high = float(batch["yaw_lo"][i]), float(batch["yaw_hi"][i])
            if w_np is not None:
                for q in range(12):
                    cfg.weights[q] = float(w_np[q, i])
            ts = time.perf_counter()
            st, o9, _, _, _ = O.mpc_solve(cfg, batch["state"][:, i], batch["coeffs"][:, i])
            t_solve += time.perf_counter() - ts
            if st == 0 and status[i] == 0:
                worst_steer = max(worst_steer, abs(o9[6] - out_np[6, i]))
                worst_acc = max(worst_acc, abs(o9[7] - out_np[7, i]))
            n_done += 1
        res["cpu_baseline"] = {"value": n_done / t_solve, "unit": "solves/s", "cores": 1, "kind": "port",
                               "sample": "first %d instances of the same batch, oracle/mpc_oracle.c (dense IPOPT-style "
                                         "interior point), 1 thread of %d host cores" % (n_done, os.cpu_count() or 0)}
        res["max_abs_dsteer_vs_oracle"] = worst_steer
        res["max_abs_daccel_vs_oracle"] = worst_acc
        res["parity_sample"] = n_done
    print(json.dumps(res))
    for h in mpcs:
        h.close()
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
