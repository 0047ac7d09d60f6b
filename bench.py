#!/usr/bin/env python3
"""bench.py -- MPC solves/sec of the batched HIP path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either the driver's `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`
     or plain `python bench.py --gpus N`, which starts the N ranks itself -- see spawn_ranks)

A "step" is one pass of the hot path (the batched replacement of MPC::solve(), src/control/MPC.cpp:183-325)
over one batch of synthetic inputs that are ALREADY RESIDENT IN HBM.  Default workload at every N: BASELINE.json
configs[2] per GPU -- 65 536 lake-track states with 100 ms latency compensation, N=10, dt=0.1, config-fast.json, fp64,
trajectories requested, drawn as SURVEY.md section 8d says (rejecting only a compensated speed above Config::maxSpeed:
`--population survey`) -- i.e. weak scaling: each rank solves its own 65 536 instances (different PRNG streams), and every
batch's per-instance results are gathered to rank 0 (RCCL).  Rank 0 prints ONE JSON line of at most 2 KB on stdout; everything
measured (every leg in full) goes to bench_full.json beside this file and to stderr.

What is timed.  Steps are pipelined the way a serving loop runs them: `--inflight` handles on separate streams, stragglers
deferred to the handles' tail slices (MpcParams.tail_cut), results gathered once a batch is final.  A batch of the survey
population is final 50-150 ms after its launch (its slowest instance runs to the iteration cap), a hundred batch periods.
So `value` is measured in a PRIMED pipeline: batches are issued continuously; the clock starts when the W-th batch has become
final (in issue order) and stops when K more have -- every one of those K batches final, stragglers and gather included,
inside the clock, with later batches being issued behind them the whole time -- and the rate no longer depends on K.  The same
K batches timed from an empty device to an empty device (barrier + synchronize on both sides, nothing issued behind them) are
`strict_value`: that number carries the whole latency of the last batch and is a function of K.

roofline: bound "hbm" with ALGORITHMIC bytes per solve (fp64 336 B = in 104 + out 72 +
trajectory 160; fp32 weight sweep 136 B) x solves per launch / the solve kernel's average launch duration measured
live with HIP events on the launch stream.  `traffic` is what the kernel really moves through the L2's memory side
(rocprofv3 FETCH_SIZE/WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes): it is read from the committed PMC
summary of the same workload (profiles/), never measured in this run -- `traffic_source` says so.  The path is
VALU/latency bound by its algorithmic bytes, so `frac` is tiny by construction; `valu_frac` prices the timed region
against the vector peak of the dtype (78.6 TFLOP/s fp64, 157.3 fp32) with the flop count of section 8d.
cpu_baseline: the oracle (oracle/mpc_oracle.c, kind "port": dense IPOPT-style interior point) on a bounded sample of the same
batch, one thread (the way the reference runs).  cpu_baseline_same_algorithm: the CPU build of the device solver's own header
(tests/host_twin: the same Riccati interior point, scalar C++) on every core of this box's share -- "CPU twin, not IPOPT"
(SURVEY.md section 8d); both are baselines only, never the product path.
host_path: the same batch through mpc_solve_batch_host (pageable host arrays -> one H2D, solve, one D2H), and the
latency of a B = 1 solve, which is what the reference's MPC::solve() drop-in does per telemetry message.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                                          # MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3}
KFLOP_PER_STAGE_ITER = 2.5
DEFAULT_TAIL_CUT = -1                                          # MPC_TAIL_AUTO: the handle chooses after how many passes an instance leaves its launch


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=65536, help="instances per GPU (weak scaling)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--global-batch", type=int, default=0, help="--scaling strong: total instances, split evenly over the GPUs")
    ap.add_argument("--config", default="config-fast.json")
    ap.add_argument("--no-traj", action="store_true")
    ap.add_argument("--precision", choices=("f64", "f32"), default="f64", help="f32 = MPC_PRECISION_F32 (BASELINE.json configs[4])")
    ap.add_argument("--f32-pure", action="store_true", help="--precision f32 with MpcParams.f32_finish = 0: the pure fp32 solver of round 2 "
                    "(tol_f32, looser tolerances) instead of fp32 iterations finished in fp64")
    ap.add_argument("--f32-phase-refill", action="store_true", help="MpcParams.f32_phase_refill = 1 (mixed precision on heavy-tailed workloads)")
    ap.add_argument("--initial-state-rows", action="store_true", help="MpcParams.initial_state_rows = 1: the multipliers of the rows that pin the initial state carried "
                    "and counted in the error measure, as IPOPT does (the oracle's iteration counts on 98.5-99 % of a batch instead of 94-96 %)")
    ap.add_argument("--fp64-only", action="store_true", help="MpcParams.f64_f32_start = 0: every iteration in fp64 at every horizon (the shipped default, "
                    "MPC_F32_START_AUTO, starts horizons of 15 steps and more on the fp32 record)")
    ap.add_argument("--f64-f32-start", action="store_true", help="fp64 handle with MpcParams.f64_f32_start = 1: the early iterations on the fp32 "
                    "record, every instance finished by the fp64 solver (experimental)")
    ap.add_argument("--switch-mu", type=float, default=0.0, help="MpcParams.mixed_switch_mu (default 2e-5)")
    ap.add_argument("--tol-f32", type=float, default=0.0, help="MpcParams.tol_f32 (default 5e-4): where the fp32 phase of a mixed-precision solve hands "
                    "over at the latest, and the stopping rule of the pure fp32 mode")
    ap.add_argument("--cpu-seconds", type=float, default=14.0, help="budget of the cpu_baseline legs (one thread + all cores)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the PCIe-inclusive and B=1 latency measurements")
    ap.add_argument("--N", type=int, default=0, help="override Config::N (BASELINE.json configs[3]: 25)")
    ap.add_argument("--dt", type=float, default=0.0, help="override Config::dt (configs[3]: 0.05)")
    ap.add_argument("--weights-sweep", action="store_true", help="per-instance Config::weights (configs[4])")
    ap.add_argument("--unfiltered", action="store_true", help="draw the instances without the generator's rejection step")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo only to "
                    "rehearse the multi-process path on a single GPU or with --stub)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--inflight", type=int, default=0, help="batches in flight per GPU (handles on separate streams); default: 2 in fp64, "
                    "4 in fp32 (measured best for each)")
    ap.add_argument("--max-iter", type=int, default=0, help="MpcParams.max_iter (default 200): the iteration cap; instances that reach it are "
                    "reported with status MAXITER, as the reference returns its iterate at IPOPT's 0.5 s max_cpu_time")
    ap.add_argument("--pass-cuts", default="", help="multi-phase solve (MpcParams.pass_cut, pass_cut_next): up to four comma-separated "
                    "cuts, e.g. 16,16,32 -- instances still running after that many passes are re-packed into dense waves for "
                    "a further launch; bitwise the same results (measured: no gain in time per batch, DESIGN.md 6c)")
    ap.add_argument("--tail-cut", type=int, default=None, help="deferred tails (MpcParams.tail_cut): instances still running after this many "
                    "passes leave their launch and are finished by the handle's tail slices while later batches run; every timed batch is "
                    "final (stragglers included) inside the clock.  0 = off, -1 = MPC_TAIL_AUTO (the handle chooses; the default)")
    ap.add_argument("--tail-ring", type=int, default=128, help="batches whose tails may be outstanding per handle")
    ap.add_argument("--outstanding", type=int, default=0, help="buffer sets = batches that may be outstanding (launched, or waiting for their stragglers); "
                    "0 = min(256, batches in flight x tail ring)")
    ap.add_argument("--leg-tail-cut", type=int, default=None, help="override the tail cut of a leg that defers (measurement aid)")
    ap.add_argument("--no-legs", action="store_true", help="skip the extra legs of the default run (the unfiltered population of the "
                    "headline workload and the other BASELINE.json configs)")
    ap.add_argument("--leg-inflight", type=int, default=0, help="batches in flight of the extra legs (0: each leg's own, or the library's advice)")
    ap.add_argument("--leg", default="", help="run ONE extra leg of the default run alone in this process and print its JSON line (what the default "
                    "run starts as child processes): " + ", ".join(LEGS))
    ap.add_argument("--leg-steps", type=int, default=20, help="timed steps of an extra leg that does not set its own")
    ap.add_argument("--no-leg-tails", dest="leg_tails", action="store_false", help="run the extra legs without deferred tails")
    ap.add_argument("--population", choices=("filtered", "survey", "unfiltered"), default="survey",
                    help="instance generator: 'filtered' redraws what the reference's road model does not hold for (scenarios.py); "
                         "'survey' applies only SURVEY.md section 8d's rejection (compensated speed above Config::maxSpeed); "
                         "'unfiltered' keeps every draw with a finite fit")
    ap.add_argument("--gather", choices=("direct", "root", "all"), default="direct", help="how a batch's results reach rank 0: 'direct' (default) -- "
                    "every rank's solver writes them straight into rank 0's IPC-mapped buffers over xGMI, no collective in the data path (falls "
                    "back to 'root' where the mapping is not available, and says so); 'root' -- one RCCL gather to rank 0 per batch group; 'all' -- "
                    "all_gather_into_tensor to every rank")
    ap.add_argument("--gather-group", type=int, default=0, help="batches per collective (default 4 with more than one rank: a collective costs "
                    "the solve 8-10 %% by being there, whatever it carries; every batch is still gathered inside the timed region)")
    ap.add_argument("--gather-results-only", action="store_true", help="trajectories are computed but stay on their rank: only out[9], status, "
                    "iters travel")
    ap.add_argument("--no-overlap", action="store_true", help="issue the gather synchronously on the solve stream")
    ap.add_argument("--no-priority-stream", action="store_true", help="run the solves on a normal-priority stream")
    ap.add_argument("--force-collective", action="store_true", help="one rank only: initialise RCCL with world size 1 and run the per-batch "
                    "all_gather_into_tensor anyway (rehearses the collective path, its stream ordering and overlap, on a one-GPU box)")
    ap.add_argument("--full-json", default="", help="where the complete result goes (default: bench_full.json beside bench.py)")
    ap.add_argument("--stub", default="", help="TEST ONLY (tests/test_bench_spawn.py): 'host_twin' replaces the device solve by "
                    "the CPU build of the solver header so that the multi-process plumbing can be exercised without a GPU; "
                    "the line it prints is marked as a stub and is not a measurement")
    args = ap.parse_args(argv)
    # (--inflight 0: the library's own advice for the workload, mpc_inflight_advice, once the parameters are known)
    if args.unfiltered:
        args.population = "unfiltered"
    if args.gather_group <= 0:
        args.gather_group = 4 if (args.gpus > 1 or args.force_collective or int(os.environ.get("WORLD_SIZE", "1")) > 1) else 1
    if args.tail_cut is None:
        args.tail_cut = 0 if args.stub else DEFAULT_TAIL_CUT
    return args


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (one per GPU, the same
    environment contract as torch.distributed.run: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT) and pass
    rank 0's JSON line through.  This parent never touches the GPU: it imports neither torch nor the HIP library
    (asserted below), so nothing is ever exec'd or forked from a process that has initialised the device."""
    import socket
    assert "torch" not in sys.modules and "carnd_mpc_project_amd" not in sys.modules, "the spawning parent must stay GPU-free"
    with open("/proc/self/maps") as f:
        assert "libamdhip64" not in f.read(), "the spawning parent must stay GPU-free"
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    print("bench parent: spawned %d ranks itself (no launcher); parent made no GPU call" % args.gpus, file=sys.stderr)
    return rc


def cpu_baseline_legs(args, batch, w_np, over, status, out_np, out_p0, budget):
    """The checker, timed as the CPU baseline (kind "port"): one thread first, then every core of this box's share; the same
    sample gives the parity numbers of the run, with the termination polish (the default) and under IPOPT's own stopping rule."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import multiprocessing as mp
    import numpy as np
    import oracle_lib as O
    B = batch["state"].shape[1]
    cfg = O.load_config(args.config, **over)
    n_done, worst_steer, worst_acc, t_solve = 0, 0.0, 0.0, 0.0
    t0 = time.perf_counter()
    while n_done < min(B, 4096) and (time.perf_counter() - t0) < 0.4 * budget:
        i = n_done
        cfg.yaw_low, cfg.yaw_high = float(batch["yaw_lo"][i]), float(batch["yaw_hi"][i])
        if w_np is not None:
            for q in range(12):
                cfg.weights[q] = float(w_np[q, i])
        ts = time.perf_counter()
        st, o9, _, _, _ = O.mpc_solve(cfg, batch["state"][:, i], batch["coeffs"][:, i])
        t_solve += time.perf_counter() - ts
        if st == 0 and status[i] == 0:
            worst_steer = max(worst_steer, abs(o9[6] - float(out_np[6, i])))
            worst_acc = max(worst_acc, abs(o9[7] - float(out_np[7, i])))
        n_done += 1
    parity = {"max_abs_dsteer_vs_oracle": worst_steer, "max_abs_daccel_vs_oracle": worst_acc, "parity_sample": n_done}
    if out_p0 is not None:                                            # polish = 0 on both sides: IPOPT's first-iterate-under-tol rule
        opt0 = O.default_options(polish=0)
        ws0, wa0, n0 = 0.0, 0.0, 0
        tp = time.perf_counter()
        while n0 < min(n_done, 1024) and (time.perf_counter() - tp) < 0.15 * budget:
            i = n0
            cfg.yaw_low, cfg.yaw_high = float(batch["yaw_lo"][i]), float(batch["yaw_hi"][i])
            if w_np is not None:
                for q in range(12):
                    cfg.weights[q] = float(w_np[q, i])
            st, o9, _, _, _ = O.mpc_solve(cfg, batch["state"][:, i], batch["coeffs"][:, i], opt0)
            if st == 0 and out_p0[1][i] == 0:
                ws0 = max(ws0, abs(o9[6] - float(out_p0[0][6, i]))); wa0 = max(wa0, abs(o9[7] - float(out_p0[0][7, i])))
            n0 += 1
        parity.update({"polish0_max_dsteer": ws0, "polish0_max_daccel": wa0, "polish0_sample": n0})
    one = {"value": n_done / t_solve, "unit": "solves/s", "cores": 1, "kind": "port",
           "sample": "first %d instances of the same batch, oracle/mpc_oracle.c (dense IPOPT-style interior point), 1 thread "
                     "of %d host cores" % (n_done, os.cpu_count() or 0)}
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    per = max(4, int(one["value"] * 0.3 * budget))               # instances per worker for ~0.3 x budget of wall time
    per = min(per, max(1, min(B, 8192) // cores))
    jobs = []
    for c in range(cores):
        lo = c * per
        sl = slice(lo, lo + per)
        jobs.append((args.config, over, batch["state"][:, sl].copy(), batch["coeffs"][:, sl].copy(), batch["yaw_lo"][sl].copy(),
                     batch["yaw_hi"][sl].copy(), None if w_np is None else w_np[:, sl].copy()))
    ctx = mp.get_context("spawn")                                  # never fork a process that holds a GPU context
    with ctx.Pool(cores) as pool:
        pool.map(O.solve_chunk, [j[:2] + (j[2][:, :1], j[3][:, :1], j[4][:1], j[5][:1], None if j[6] is None else j[6][:, :1]) for j in jobs])   # start-up outside the clock
        tw = time.perf_counter()
        done = pool.map(O.solve_chunk, jobs)
        wall = time.perf_counter() - tw
    allc = {"value": sum(done) / wall, "unit": "solves/s", "cores": cores, "kind": "port",
            "sample": "%d instances of the same batch in %d worker processes (one oracle thread each) on the %d cores this "
                      "process may use" % (sum(done), cores, cores)}
    return one, allc, parity


def _twin_chunk(job):
    """Worker of cpu_twin_leg: the CPU build of the device solver's header on a chunk of instances -> how many it solved."""
    import ctypes as C
    import numpy as np
    pbytes, st, cf, yl, yh, w, N = job
    sys.path.insert(0, ROOT)
    import __graft_entry__ as G
    pkg = G.load_package()
    params = pkg.MpcParams.from_buffer_copy(pbytes)
    twin = C.CDLL(os.path.join(ROOT, "tests", "host_twin", "libhost_twin.so"))
    n = st.shape[1]
    out = np.zeros((9, n)); status = np.zeros(n, dtype=np.int32); iters = np.zeros(n, dtype=np.int32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
    rc = twin.mpc_host_twin_solve(C.byref(params), C.c_int64(n), C.c_int64(n), vp(st), vp(cf), vp(yl), vp(yh), vp(w), vp(out), None, vp(status), vp(iters))
    assert rc == 0
    return n


def cpu_twin_leg(args, pkg, params, batch, w_np, budget):
    """SURVEY.md 8d's CPU baseline of the same algorithm: tests/host_twin (the device solver's own header compiled for the CPU:
    Riccati interior point, scalar C++), one process per core of this box's share.  "CPU twin, not IPOPT"; test infrastructure
    timed as a baseline, never the product path."""
    import multiprocessing as mp
    import numpy as np
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "host_twin")])
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    B = batch["state"].shape[1]
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    p = params.copy(); p.precision = pkg.PRECISION_F64; p.tail_cut = 0
    mk = lambda lo, n: (bytes(p), f(batch["state"][:, lo:lo + n]), f(batch["coeffs"][:, lo:lo + n]), f(batch["yaw_lo"][lo:lo + n]), f(batch["yaw_hi"][lo:lo + n]),
                        None if w_np is None else f(w_np[:, lo:lo + n]), p.N)
    ctx = mp.get_context("spawn")
    with ctx.Pool(cores) as pool:
        pool.map(_twin_chunk, [mk(0, 8) for _ in range(cores)])      # start-up outside the clock
        t1 = time.perf_counter(); pool.map(_twin_chunk, [mk(0, 256)]); one_core = 256 / (time.perf_counter() - t1)
        per = int(max(64, min(B // cores, one_core * 0.5 * budget)))
        tw = time.perf_counter()
        done = pool.map(_twin_chunk, [mk(c * per, per) for c in range(cores)])
        wall = time.perf_counter() - tw
    return {"value": sum(done) / wall, "unit": "solves/s", "cores": cores, "kind": "port", "one_core": one_core,
            "sample": "CPU twin, not IPOPT: %d instances of the same batch, tests/host_twin (the device solver's header built for the CPU), "
                      "%d processes" % (sum(done), cores)}


def f32_start_on(params):
    """MpcParams.f64_f32_start as the library reads it: 1 = on, 2 (MPC_F32_START_AUTO) = on for horizons of 15 steps and more."""
    return params.f64_f32_start == 1 or (params.f64_f32_start == 2 and params.N >= 15)


class _NullCtx:
    def __enter__(self): return self
    def __exit__(self, *a): return False


class Pipeline:
    """One workload kept in flight the way a serving loop runs it: `nfl` handles on `nfl` streams, results written straight into
    the packed buffers of sharding.PackedGather, the path's one collective issued when a batch is FINAL.  With deferred tails
    (tail_cut != 0) a batch's stragglers are finished by its handle's tail slices while later batches run.  Nothing here blocks
    the host: issue() starts the next batch if a buffer set and a launch slot are free, progress() asks the handles which
    batches have become final (mpc_tail_poll: that call is also what starts the tail slices) and hands them to the gather, in
    issue order.  final_upto counts the batches that are final in issue order; marks[n] is the host time at which it reached n."""

    def __init__(self, pkg, torch, params, B, tensors, d_w, want_traj, nfl, dev, local_rank, dist, args, stub=None, tail_cut=0, tail_ring=128,
                 outstanding=0, depth=2):
        import collections
        self.torch, self.pkg, self.B, self.nfl, self.dev, self.stub = torch, pkg, B, max(1, nfl), dev, stub
        self.tensors, self.d_w, self.depth = tensors, d_w, max(1, depth)
        self.tail = int(tail_cut) if stub is None else 0
        p = params.copy()
        p.tail_cut = self.tail
        p.tail_ring = int(tail_ring)
        self.params = p
        make = (lambda: stub.BatchedMPC(p, B)) if stub else (lambda: pkg.BatchedMPC(p, B, device=local_rank))
        self.mpcs = [make() for _ in range(self.nfl)]
        # buffer sets: one per batch that may be outstanding (launched, or waiting for its stragglers, or for its gather)
        g = max(1, int(getattr(args, "gather_group", 1) or 1)) if dist is not None else 1
        n_slots = max(2 * g, 2 * self.nfl * self.depth) if not self.tail else max(int(outstanding) or self.nfl * int(tail_ring), 2 * g)
        tdt = torch.float32 if p.precision == pkg.PRECISION_F32 else torch.float64
        self.pg = pkg.sharding.PackedGather(B, p.N, want_traj, dev, dist if dist is not None else None, overlap=not args.no_overlap,
                                            slots=n_slots, dtype=tdt, force=args.force_collective, root_only=args.gather in ("root", "direct"), gather_traj=not args.gather_results_only,
                                            batches_per_collective=g, direct=args.gather == "direct")
        self.nslots = self.pg.slots
        self.streams, self.gstream = None, None
        if stub is None:
            torch.cuda.synchronize(dev)
            # The solves run on high-priority streams: when a batch's gather (RCCL's own stream, normal priority) and the next
            # batch's solve become ready together, the solve's waves are placed first.
            self.streams = [torch.cuda.Stream(device=dev, priority=0 if args.no_priority_stream else -1) for _ in range(self.nfl)]
            self.gstream = torch.cuda.Stream(device=dev)
        self.n_issued, self.final_upto = 0, 0
        self.recs = collections.deque()          # issued, not yet final: [n, handle, batch id, slot, launch event]
        self.launches = collections.deque()      # launch events of the batches whose own launch may still be running
        self.timing = collections.deque(maxlen=4096)   # (n, e0, e1): HIP events around each launch, on its stream
        self.marks = {}
        self.want_marks = set()
        self.time_events = stub is None

    def _ctx(self, stream):
        return self.torch.cuda.stream(stream) if self.stub is None else _NullCtx()

    def issue(self):
        """Starts batch number n_issued if its buffer set is free and fewer than nfl x depth launches are unfinished."""
        n = self.n_issued
        if n - self.nslots >= self.final_upto:
            return False                                    # the batch that used this buffer set last is not final yet
        while self.launches and self.launches[0].query():
            self.launches.popleft()
        if len(self.launches) >= self.nfl * self.depth:
            return False
        slot, j = n % self.nslots, n % self.nfl
        h = self.mpcs[j]
        st = self.streams[j] if self.streams else None
        ev = None
        with self._ctx(st):
            self.pg.wait(slot)                              # the gather that last read this buffer set has finished (stream-side wait)
            if self.time_events:
                e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
                e0.record()
            h.solve_torch(*self.tensors, weights=self.d_w, outputs=self.pg.outputs(slot))   # async on this stream
            if self.time_events:
                e1.record()
                self.timing.append((n, e0, e1))
                ev = e1
        if ev is not None:
            self.launches.append(ev)
        self.recs.append((n, h, h.last_batch_id() if self.tail else 0, slot, ev))
        self.n_issued += 1
        return True

    def progress(self):
        """Batches that have become final, in issue order -> the gather.  Returns how many."""
        moved = 0
        if self.tail:
            for h in self.mpcs:                             # every handle's pump gets its turn (tail slices are started by these calls)
                h.tail_flush()
        while self.recs:
            n, h, bid, slot, ev = self.recs[0]
            if self.tail:
                ok = h.tail_poll(bid)
            else:
                ok = True if ev is None else ev.query()
            if not ok:
                break
            with self._ctx(self.gstream):
                if self.tail:
                    h.tail_stream_wait(bid, self.gstream)   # final already: orders the gather's stream behind the tail stream
                elif ev is not None:
                    self.gstream.wait_event(ev)
                self.pg.start(slot)                         # the path's only collective
            self.recs.popleft()
            self.final_upto = n + 1
            moved += 1
            if self.final_upto in self.want_marks or not self.want_marks:
                self.marks[self.final_upto] = time.perf_counter()
        return moved

    def run_until(self, cond, issue=True):
        """The serving loop: issue while allowed, collect what has become final, until cond(self)."""
        idle = 0
        while not cond(self):
            did = self.issue() if issue else False
            did = bool(self.progress()) or did
            if not did:
                idle += 1
                if idle > 64:
                    time.sleep(20e-6)                       # nothing to do right now: let the GPU get on with it
            else:
                idle = 0

    def drain(self):
        """Everything issued so far becomes final and gathered (blocks)."""
        if self.tail:
            for h in self.mpcs:
                h.tail_wait(0)
        self.run_until(lambda p: not p.recs, issue=False)
        self.pg.finish()

    def last_slot(self):
        return (self.n_issued - 1) % self.nslots

    def kernel_ms(self, lo, hi):
        return [a.elapsed_time(b) for n, a, b in self.timing if lo <= n < hi]

    def close(self):
        for h in self.mpcs:
            h.close()


def timed_run(pipe, steps, warmup, sync_all, lockstep=False):
    """-> dict: `steady_s` = host time between the moment batch number P became final and the moment batch P + steps did (P >=
    warmup batches final before the clock starts, batches issued behind the timed ones all along: the primed pipeline);
    `strict_s` = the same number of batches from an empty device to an empty device, nothing issued behind them.
    lockstep (several ranks: every rank must issue the same number of batches, their gathers are collectives): P and the number
    of batches issued behind the timed ones are fixed in advance -- as many as may be outstanding -- instead of following the
    finals."""
    sync_all()
    lag = max(2 * pipe.nfl * pipe.depth, (pipe.nslots - 2 * pipe.nfl * pipe.depth) if pipe.tail else 0)
    # Primed = steady: with deferred tails the launches run ahead of the tail slices until every buffer set is taken (a backlog of
    # batches whose bulk is done and whose stragglers are being worked off: finals then come at the rate the backlog is cleared, not
    # at the rate batches are produced).  The clock starts when every buffer set has gone round once: every batch outstanding
    # then was issued by a pipeline that was already waiting for finals.
    prime = max(int(warmup), 2 * pipe.nfl, (pipe.nslots + 2 * pipe.nfl * pipe.depth) if pipe.tail else 0)
    pipe.want_marks = set()
    if lockstep:
        total = prime + steps + lag
        pipe.run_until(lambda p: p.final_upto >= prime or p.n_issued >= total)
        pipe.run_until(lambda p: p.final_upto >= prime, issue=False)
    else:
        pipe.run_until(lambda p: p.final_upto >= prime)
    n0 = pipe.final_upto
    t_a = pipe.marks.get(n0, time.perf_counter())
    issued_a = pipe.n_issued
    if lockstep:
        pipe.run_until(lambda p: p.final_upto >= n0 + steps or p.n_issued >= total)
        pipe.run_until(lambda p: p.final_upto >= n0 + steps, issue=False)
    else:
        pipe.run_until(lambda p: p.final_upto >= n0 + steps)
    n1 = pipe.final_upto
    t_b = pipe.marks[n1]
    outstanding = pipe.n_issued - n1
    steady = {"steady_s": (t_b - t_a) * steps / (n1 - n0), "first_timed_batch": n0, "batches_in_the_window": n1 - n0,
              "batches_issued_before_the_clock": issued_a, "batches_outstanding_at_the_end": outstanding}
    kms = pipe.kernel_ms(n0, n1)
    if lockstep:
        pipe.run_until(lambda p: p.n_issued >= total)
    pipe.drain()
    sync_all()
    t0 = time.perf_counter()
    first = pipe.n_issued
    pipe.run_until(lambda p: p.n_issued >= first + steps)
    pipe.drain()
    sync_all()
    steady["strict_s"] = time.perf_counter() - t0
    steady["kernel_ms"] = kms if kms else pipe.kernel_ms(first, first + steps)
    return steady


def summarize(pkg, np, pipe, B, steps, elapsed):
    outs = pipe.pg.outputs(pipe.last_slot())
    status = outs["status"].cpu().numpy()
    iters = outs["iters"].cpu().numpy()
    r = {"solves_per_s": B * steps / elapsed, "ms_per_batch": 1e3 * elapsed / steps, "batch": B, "steps": steps,
         "batches_in_flight": pipe.nfl, "tail_cut": pipe.tail,
         "status_counts": {pkg.STATUS_NAMES[k]: int((status == k).sum()) for k in sorted(pkg.STATUS_NAMES)},
         "mean_iterations": float(iters.mean()), "max_iterations": int(iters.max())}
    if pipe.tail and pipe.stub is None:
        info = [h.tail_info() for h in pipe.mpcs]
        r["tails"] = {"tail_slices": sum(i["tail_launches"] for i in info), "batches_deferred": sum(i["batches_deferred"] for i in info),
                      "ring": info[0]["ring"], "capacity_per_batch": info[0]["capacity_per_batch"],
                      "waves_per_slice_max": info[0]["waves_per_tail_launch"], "passes_per_slice": info[0]["passes_per_slice"],
                      "batches_not_deferred_survivors_full": sum(i["batches_not_deferred_survivors_full"] for i in info),
                      "tail_cut_in_use": info[0].get("tail_cut_in_use"), "queue_overflows": sum(i.get("queue_overflows", 0) for i in info),
                      "buffer_sets": pipe.nslots}
    return r, status, iters, outs


# The extra legs of the default run: name -> (description, keyword arguments of run_leg).  tail_cut -1 = MPC_TAIL_AUTO.
LEGS = {
    "filtered": ("the headline workload (configs[2]) drawn with the generator's own rejection sampling (instances the reference's road model does "
                 "not hold for are redrawn: yaw on its bound, waypoint windows that double back, fits beyond Config::maxFitError) -- rounds 1-3's headline; "
                 "no heavy tail (at most 26 iterations), tails not deferred",
                 dict(config="config-fast.json", over={}, B=65536, kind="lake", f32=False, sweep=False, want_traj=True, nfl=None, population="filtered", tail_cut=0, steps=100)),
    "headline_f32_start": ("the headline workload (survey population) with MpcParams.f64_f32_start = 1: the early iterations (barrier parameter above 2e-5) on the fp32 "
                           "record, every instance finished by the fp64 solver to the same tol and polish; four batches in flight",
                           dict(config="config-fast.json", over={}, B=65536, kind="lake", f32=False, sweep=False, want_traj=True, nfl=None, population="survey", tail_cut=-1, steps=100, f32_start=True)),
    "configs_1": ("BASELINE.json configs[1]: 4 096 straight-line-offset states, config-stable.json",
                  dict(config="config-stable.json", over={}, B=4096, kind="straight", f32=False, sweep=False, want_traj=True, nfl=None, steps=200,
                       note="a 4 096-instance launch is 64 waves, 6 % of the device: eight batches in flight")),
    "configs_3_share": ("BASELINE.json configs[3], one GPU's share of 262 144: 32 768 lake-track states, N=25 dt=0.05, fp64 handle as shipped (MPC_F32_START_AUTO: "
                        "horizons of 15 steps and more run their early iterations on the fp32 record, every instance finished by the fp64 solver), SURVEY 8d's "
                        "population, deferred tails, four batches in flight, windows of 400 batches",
                        dict(config="config-stable.json", over=dict(N=25, dt=0.05), B=32768, kind="lake", f32=False, sweep=False, want_traj=True, nfl=None, population="survey", tail_cut=-1, steps=400)),
    "configs_3_share_filtered": ("the same share drawn with the generator's rejection sampling",
                                 dict(config="config-stable.json", over=dict(N=25, dt=0.05), B=32768, kind="lake", f32=False, sweep=False, want_traj=True, nfl=None, tail_cut=-1, steps=400)),
    "configs_3_share_fp64_only": ("the same share (SURVEY 8d's population) with MpcParams.f64_f32_start = 0: every iteration in fp64 (bitwise the host twin's solve)",
                                  dict(config="config-stable.json", over=dict(N=25, dt=0.05), B=32768, kind="lake", f32=False, sweep=False, want_traj=True, nfl=None, population="survey", tail_cut=-1, steps=400,
                                       f32_start=False)),
    "configs_4_share": ("BASELINE.json configs[4], one GPU's share of 1 048 576: 131 072 lake-track states (SURVEY 8d's population), fp32 mixed precision as shipped (fp32 "
                        "iterations down to the barrier parameter 2e-5, every instance finished in fp64), per-instance weight sweep (epsi / v incl. 0 / delta / a)",
                        dict(config="config-fast.json", over={}, B=131072, kind="lake", f32=True, sweep=True, want_traj=False, nfl=None, population="survey", tail_cut=-1, steps=300, f32_refill=True)),
    "configs_4_share_filtered": ("the same share drawn with the generator's rejection sampling",
                                 dict(config="config-fast.json", over={}, B=131072, kind="lake", f32=True, sweep=True, want_traj=False, nfl=None, tail_cut=-1, steps=300, f32_refill=True)),
    "configs_4_share_pure_fp32": ("the same share (SURVEY 8d's population) with the pure fp32 solver (f32_finish = 0: stops at tol_f32 = 5e-4, looser stated tolerances)",
                                  dict(config="config-fast.json", over={}, B=131072, kind="lake", f32=True, sweep=True, want_traj=False, nfl=None, population="survey", tail_cut=-1, steps=300,
                                       f32_pure=True)),
    "weights_sweep_f64": ("the fp64 solve of the configs[4] weight sweep (65 536 instances, SURVEY 8d's population): what the fp32 mode is to be compared with",
                          dict(config="config-fast.json", over={}, B=65536, kind="lake", f32=False, sweep=True, want_traj=False, nfl=None, population="survey", tail_cut=-1, steps=300)),
}


def run_leg(pkg, torch, np, args, dev, local_rank, golden, wp, sync_all, name, config, over, B, kind, f32, sweep, want_traj, nfl, population="filtered",
            velocity_weights=(0.0, 1.0, 100.0), note=None, tail_cut=0, steps=None, f32_pure=False, f32_start=None, f32_refill=False, hw_queues=8, outstanding=0):
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype=dt)
    params = pkg.params_from_json(os.path.join(golden, config), **over)
    if f32_start is not None:                         # None: as shipped (MPC_F32_START_AUTO: on for horizons of 15 steps and more)
        params.f64_f32_start = 1 if f32_start else 0
    params.f32_phase_refill = 1 if f32_refill else 0
    if f32:
        params.precision = pkg.PRECISION_F32
        params.f32_finish = 0 if f32_pure else 1
    if args.max_iter > 0:
        params.max_iter = args.max_iter
    tdt = torch.float32 if f32 else torch.float64
    if kind == "straight":
        b = pkg.scenarios.straight_line_batch(B, params)
        b["drawn"], b["rejected"] = B, {}
    else:
        b = pkg.scenarios.lake_track_batch(B, params, wp, stream=3, filtered={"filtered": True, "survey": "survey", "unfiltered": False}[population])
    w = pkg.scenarios.weight_sweep(B, params, seed=1234, velocity_weights=velocity_weights) if sweep else None
    tens = (t(b["state"], tdt), t(b["coeffs"], tdt), t(b["yaw_lo"], tdt), t(b["yaw_hi"], tdt))
    steps = steps or args.leg_steps
    if tail_cut and not args.leg_tails:
        tail_cut, steps = 0, min(steps, 10)
    if args.leg_tail_cut is not None and tail_cut:
        tail_cut = args.leg_tail_cut
    advised = pkg.inflight_advice(params, B)
    nfl = args.leg_inflight or nfl or advised           # (nfl = None in LEGS: the library's advice, mpc_inflight_advice)
    pipe = Pipeline(pkg, torch, params, B, tens, t(w, tdt) if w is not None else None, want_traj, nfl, dev, local_rank, None, args,
                    tail_cut=tail_cut, tail_ring=args.tail_ring, outstanding=outstanding or args.outstanding or min(256, nfl * args.tail_ring))
    # (every handle's first calls allocate lazily -- second workspace, tail queues: they happen while the pipeline is primed)
    tr = timed_run(pipe, steps, 4 * pipe.nfl, sync_all)
    r, _, _, _ = summarize(pkg, np, pipe, B, steps, tr["steady_s"])
    r["strict_solves_per_s"] = B * steps / tr["strict_s"]
    r["timing"] = {k: tr[k] for k in ("first_timed_batch", "batches_in_the_window", "batches_issued_before_the_clock", "batches_outstanding_at_the_end")}
    if tr["kernel_ms"]:
        r["kernel_ms_avg"] = float(np.mean(tr["kernel_ms"]))
    pipe.close()
    r["batches_in_flight_advised"] = advised
    r.update({"workload": name, "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")), "config": config, "N": params.N, "dt": params.dt, "dtype": "f32" if f32 else "f64", "trajectories": want_traj,
              "population": population, "draws": int(b["drawn"]), "rejected": b["rejected"]})
    if note:
        r["note"] = note
    return r


def extra_legs(args):
    """The other populations the driver-timed line reports: the headline workload drawn with SURVEY.md
    8d's own rejection only, and BASELINE.json configs[1], [3] (one GPU's share) and [4] (one GPU's share).  Each leg runs in a
    child process of its own (`bench.py --leg NAME`): the mapping of a process's streams onto the 8 hardware queues depends on
    every stream it has created before, and legs run one after the other in one process measured up to 30 % below the same
    workload run alone.  (The parent has initialised the GPU, so it starts children and never execs.)"""
    legs = {}
    for name in LEGS:
        cmd = [sys.executable, os.path.abspath(__file__), "--leg", name, "--leg-steps", str(args.leg_steps)]
        if args.max_iter > 0:
            cmd += ["--max-iter", str(args.max_iter)]
        if not args.leg_tails:
            cmd += ["--no-leg-tails"]
        if args.leg_tail_cut is not None:
            cmd += ["--leg-tail-cut", str(args.leg_tail_cut)]
        cmd += ["--tail-ring", str(args.tail_ring)]
        env = dict(os.environ, GPU_MAX_HW_QUEUES=str(LEGS[name][1].get("hw_queues", 8)))     # (more than 8: two legs hung on the box)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        try:
            p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            legs[name] = json.loads(line[-1]) if p.returncode == 0 and line else {"workload": LEGS[name][0], "error": (p.stderr or "no output")[-400:]}
        except Exception as e:                                  # a leg that fails must not take the headline line with it
            legs[name] = {"workload": LEGS[name][0], "error": repr(e)[:400]}
    return legs


def leg_main(args):
    """`bench.py --leg NAME`: one extra leg alone in this process; prints its JSON line."""
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(LEGS[args.leg][1].get("hw_queues", 8)))
    import numpy as np
    import torch
    import __graft_entry__ as G
    pkg = G.load_package()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    golden = os.path.join(ROOT, "tests", "golden")
    wp = pkg.scenarios.load_waypoints(os.path.join(golden, "lake_track_waypoints.csv"))
    desc, kw = LEGS[args.leg]
    r = run_leg(pkg, torch, np, args, dev, 0, golden, wp, lambda: torch.cuda.synchronize(dev), desc, **kw)
    os.write(json_fd, (json.dumps(r) + "\n").encode())


def main():
    args = parse_args()
    if args.leg:
        return leg_main(args)
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world_env == 1:
        sys.exit(spawn_ranks(args))                                # before anything touches the GPU

    # stdout carries rank 0's ONE JSON line and nothing else: RCCL and gloo print banners to fd 1 when a communicator is
    # created, so everything below writes to stderr and the line goes to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # HIP serves 4 hardware queues per process by default; more streams in use than that need more (read at HIP start-up):
    # batches in flight, their tail streams, the gather stream, and the heavy-tailed legs of the default run
    want_q = (args.inflight if args.inflight > 0 else 4) * (2 if args.tail_cut else 1) + 1      # (--inflight 0: the library's advice, 2-8)
    if want_q > 4 or (not args.no_legs and world_env == 1 and not args.stub):
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    import numpy as np
    import torch
    import __graft_entry__ as G
    pkg = G.load_package()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = world_env
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    stub = None
    if args.stub:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import bench_stub                                          # test infrastructure, never the product path
        stub = bench_stub.make(args.stub, pkg)
    if stub is None and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    if args.single_device:
        local_rank = 0
    if stub is None:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        dev = torch.device("cpu")
    dist = None
    if world > 1 or args.force_collective:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
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
    f32 = args.precision == "f32"
    if f32:
        params.precision = pkg.PRECISION_F32
        params.f32_finish = 0 if args.f32_pure else 1
    # (as shipped: MPC_F32_START_AUTO = the fp32 start for horizons of 15 steps and more)
    if args.f64_f32_start:
        params.f64_f32_start = 1
    elif args.fp64_only:
        params.f64_f32_start = 0
    params.f32_phase_refill = 1 if args.f32_phase_refill else 0
    params.initial_state_rows = 1 if args.initial_state_rows else 0
    if args.inflight <= 0:
        args.inflight = pkg.inflight_advice(params, args.batch) if stub is None else 2
    if args.switch_mu > 0:
        params.mixed_switch_mu = args.switch_mu
    if args.tol_f32 > 0:
        params.tol_f32 = args.tol_f32
    if args.max_iter > 0:
        params.max_iter = args.max_iter
    cuts = [int(c) for c in args.pass_cuts.split(",") if c.strip()][:4]
    if cuts:
        params.pass_cut = cuts[0]
        for k, c in enumerate(cuts[1:]):
            params.pass_cut_next[k] = c
    tdt = torch.float32 if f32 else torch.float64
    wp = pkg.scenarios.load_waypoints(os.path.join(golden, "lake_track_waypoints.csv"))
    if args.scaling == "strong":
        G_total = args.global_batch or args.batch
        if G_total % world:
            raise SystemExit("--global-batch %d is not divisible by %d ranks" % (G_total, world))
        B = G_total // world
    else:
        B = args.batch
    want_traj = not args.no_traj
    # every rank draws its own instances (rank-specific PRNG stream)
    batch = pkg.scenarios.lake_track_batch(B, params, wp, stream=3 + 16 * rank,
                                           filtered={"filtered": True, "survey": "survey", "unfiltered": False}[args.population])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype=tdt)
    tensors = (t(batch["state"]), t(batch["coeffs"]), t(batch["yaw_lo"]), t(batch["yaw_hi"]))
    w_np = pkg.scenarios.weight_sweep(B, params, seed=1234 + rank, velocity_weights=(0.0, 1.0, 100.0)) if args.weights_sweep else None
    d_w = t(w_np) if w_np is not None else None
    # Batches in flight: a launch of 65 536 instances is exactly one wave per SIMD and lasts as long as its slowest wave
    # while the average wave is done after ~70 % of that time; a second handle on a second stream lets the next batch's
    # waves take the SIMDs as they become free (measured: 2.2 -> 1.3 ms per batch).
    pipe = Pipeline(pkg, torch, params, B, tensors, d_w, want_traj, args.inflight, dev, local_rank, dist, args, stub=stub,
                    tail_cut=args.tail_cut, tail_ring=args.tail_ring, outstanding=args.outstanding or min(256, args.inflight * args.tail_ring))
    nfl, pg = pipe.nfl, pipe.pg

    def sync_all():
        if stub is None:
            torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            if stub is None:
                torch.cuda.synchronize(dev)

    # HIP events on the stream the kernel is launched on: torch's current stream, whose handle is what
    # solve_torch passes to mpc_solve_batch_device (a NULL handle is HIP's null stream = torch's default)
    # (a handle's first call allocates lazily -- second workspace, tail queues: at least one untimed step per handle)
    tr = timed_run(pipe, args.steps, max(args.warmup, pipe.nfl), sync_all, lockstep=dist is not None)
    elapsed, strict = tr["steady_s"], tr["strict_s"]
    if dist is not None:
        tt = torch.tensor([elapsed, strict], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, strict = float(tt[0].item()), float(tt[1].item())
    kernel_ms = tr["kernel_ms"] if tr["kernel_ms"] else [1e3 * elapsed / args.steps]
    kernel_ms_avg = float(np.mean(kernel_ms))
    summary, status, iters_np, outs = summarize(pkg, np, pipe, B, args.steps, elapsed)
    out_np = outs["out"].cpu().numpy()
    last = pipe.last_slot()
    # the same launch alone on the device (nothing else in flight), for reference
    iso = []
    if stub is None:
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(pipe.streams[0]):
                e0.record(); pipe.mpcs[0].solve_torch(*tensors, weights=d_w, outputs=pg.outputs(last)); e1.record()
            pipe.mpcs[0].tail_wait(0)
            torch.cuda.synchronize(dev)
            iso.append(e0.elapsed_time(e1))
    # the gathered copy of this rank's shard must be what the solver wrote
    g = pg.result(last)
    gather_ok = None
    if g is not None:
        gather_ok = bool(torch.equal(g["out"][rank if dist is not None else 0], outs["out"]) and
                         torch.equal(g["status"][rank if dist is not None else 0], outs["status"]) and
                         (g["traj"] is None or torch.equal(g["traj"][rank if dist is not None else 0], outs["traj"])))

    if dist is not None and getattr(pg, "direct", False) and world > 1:
        # direct remote write: rank 0 checks that every rank's region of ITS buffers holds what that rank's solver says it wrote
        mine = torch.stack([torch.nan_to_num(outs["out"].double()).sum(), outs["status"].double().sum(), outs["iters"].double().sum()]).to(dev)
        alls = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(alls, mine)
        if rank == 0:
            ok = True
            for r_ in range(world):
                gr = pg.result(last)
                got = torch.stack([torch.nan_to_num(gr["out"][r_].double()).sum(), gr["status"][r_].double().sum(), gr["iters"][r_].double().sum()])
                ok = ok and bool(torch.equal(got.cpu(), alls[r_].cpu()))
            gather_ok = ok
    if rank != 0:
        pipe.close()
        if dist is not None:
            dist.barrier(); dist.destroy_process_group()
        return

    dtype = "f32" if f32 else "f64"
    total_solves = B * world * args.steps
    value = total_solves / elapsed
    mean_iters = summary["mean_iterations"]
    stages = params.N - 1
    is_headline = (params.N == 10 and not args.weights_sweep and B == 65536 and not f32 and args.config == "config-fast.json" and want_traj
                   and args.population == "survey")
    res = {
        "metric": "%sMPC solves/sec (batch) at N=%d dt=%g" % ("STUB (not a measurement) " if stub else "", params.N, params.dt),
        "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, nfl),
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        # the same number of batches from an empty device to an empty device, nothing issued behind them: carries the whole
        # latency of the last batch's slowest instance (a function of --steps)
        "strict_value": total_solves / strict, "strict_ms_per_step": 1e3 * strict / args.steps,
        "timing": dict({k: tr[k] for k in ("first_timed_batch", "batches_in_the_window", "batches_issued_before_the_clock", "batches_outstanding_at_the_end")},
                       definition="primed pipeline: the clock runs from the moment batch number first_timed_batch became final (in issue order) to the "
                                  "moment `steps` more had, with later batches issued behind them all along; strict_value: empty device to empty device"),
        "config": {"workload": "%s%d lake-track states per GPU, 100 ms latency compensation, N=%d dt=%g, %s, trajectories %s%s%s"
                               % ("BASELINE.json configs[2]: " if is_headline else "", B, params.N, params.dt, args.config,
                                  "on" if want_traj else "off", ", per-instance weight sweep" if args.weights_sweep else "",
                                  ", MPC_PRECISION_F32" if f32 else ""),
                   "batch_per_gpu": B, "global_batch": B * world, "N": params.N, "dt": params.dt,
                   "parallelism": "%d independent shard(s), one %s of the packed results per %s" % (world, pg.collective_name, "batch" if pg.g == 1 else "%d batches" % pg.g),
                   "collective_mode": pg.mode, "gather_checked": gather_ok, "gather_bytes_sent_per_rank_per_batch": pg.bytes_sent_per_rank if pg.active else 0, "batches_per_collective": pg.g if pg.active else None,
                   "batches_in_flight": nfl,
                   "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                   "branch_mode": "frozen", "tol": params.tol_f32 if f32 else params.tol, "max_iter": params.max_iter,
                   "termination_polish": bool(params.polish), "bound_relax_factor": params.bound_relax_factor, "pass_cuts": cuts,
                   "mixed_precision": ("fp32 iterations to mu = %g, every instance finished in fp64" % params.mixed_switch_mu) if ((f32 and params.f32_finish) or
                                                                                                                                (not f32 and f32_start_on(params))) else "no",
                   "deferred_tails": summary.get("tails", "off"), "tail_cut": pipe.tail,
                   "lane_compact": int(os.environ.get("MPC_LANE_COMPACT", params.lane_compact if params.lane_compact >= 0 else (1 if params.N >= 15 else 2))) if B >= 8192 else 0,
                   # "survey": SURVEY.md 8d's population (only a compensated speed above Config::maxSpeed is redrawn); the leg "filtered"
                   # is the same workload with the generator's own rejection sampling (scenarios.py), rounds 1-3's headline
                   "population": args.population,
                   "instance_filter": ("none (unfiltered draws)" if args.population == "unfiltered" else
                                       "%s: %d draws for %d instances" % ("SURVEY 8d's rejection only" if args.population == "survey" else "rejection sampling",
                                                                          batch["drawn"], B)),
                   "instance_filter_rejected": batch["rejected"]},
        "converged_fraction": float((status == 0).mean()),
        "status_counts": summary["status_counts"],
        "mean_iterations": mean_iters, "max_iterations": summary["max_iterations"],
        # per-instance interior-point iterations of the last batch: the launch lasts as long as its slowest instance
        "iteration_quantiles": {q: int(np.quantile(iters_np, float(q))) for q in ("0.5", "0.9", "0.99", "0.999")},
        "iteration_histogram": {"<=8": int((iters_np <= 8).sum()), "9-12": int(((iters_np > 8) & (iters_np <= 12)).sum()),
                                "13-16": int(((iters_np > 12) & (iters_np <= 16)).sum()), "17-24": int(((iters_np > 16) & (iters_np <= 24)).sum()),
                                "25-40": int(((iters_np > 24) & (iters_np <= 40)).sum()), "41-80": int(((iters_np > 40) & (iters_np <= 80)).sum()),
                                ">80": int((iters_np > 80).sum())},
    }
    if stub:
        res["stub"] = "%s: CPU build of the solver header, test infrastructure only" % args.stub
    esz = 4 if f32 else 8
    per_solve = esz * (6 + 5 + 2) + esz * 9 + (esz * 2 * params.N if want_traj else 0) + (esz * 12 if args.weights_sweep else 0)
    algo_bytes = per_solve * B
    achieved_gbs = algo_bytes / (kernel_ms_avg * 1e-3) / 1e9
    traffic, traffic_source = None, None
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
        if not name.endswith("pmc_traffic.json"):
            continue
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            continue
        for e in pj.get("entries", [pj]):
            mixed_now = (f32 and bool(params.f32_finish)) or (not f32 and f32_start_on(params))
            if (e.get("batch") == B and e.get("config") == args.config and e.get("N", 10) == params.N and e.get("dtype", "f64") == dtype
                    and bool(e.get("weights_sweep", False)) == bool(args.weights_sweep) and bool(e.get("traj", True)) == want_traj
                    and (e.get("mixed_precision", "no") != "no") == mixed_now):
                traffic = e.get("hbm_bytes_per_launch")
                traffic_source = "profiles/%s: rocprofv3 PMC passes of this workload on another run (FETCH_SIZE x2 + WRITE_SIZE), not measured in this run" % name
                break
        if traffic is not None:
            break
    flops_per_step = B * mean_iters * stages * KFLOP_PER_STAGE_ITER * 1e3
    step_s = elapsed / args.steps
    valu_frac = flops_per_step / step_s / 1e12 / VALU_PEAK_TFLOPS[dtype]
    traffic_frac = (traffic / step_s / 1e9 / HBM_PEAK_GBS) if traffic else None
    res["roofline"] = {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                       "kernel": "mpc_solve_kernel<%s>" % ("float" if f32 else "double"),
                       "kernel_ms_avg": kernel_ms_avg, "kernel_ms_min": float(np.min(kernel_ms)),
                       "kernel_ms_alone": float(np.median(iso)) if iso else None,
                       "algorithmic_bytes_per_launch": algo_bytes,
                       # against the STEP time: with several batches in flight launches overlap, so a launch lasts longer than a step
                       "frac_vs_step": algo_bytes / step_s / 1e9 / HBM_PEAK_GBS,
                       # WHAT BINDS (the mandated `frac` prices the algorithmic bytes, which are tiny against the arithmetic):
                       # the workspace the kernel streams per sweep against the HBM peak, and the arithmetic against the vector peak
                       "binds": "the workspace traffic the kernel creates (traffic_frac_of_hbm_peak) and vector-ALU issue (valu_frac); "
                                "`frac` is the algorithmic-bytes number the bench contract asks for",
                       "traffic_gbs_vs_step": (traffic / step_s / 1e9) if traffic else None,
                       "traffic_frac_of_hbm_peak": traffic_frac,
                       "note": "kernel_ms_avg is a launch that shares the device with the other batches in flight, kernel_ms_alone the same "
                               "launch by itself; with deferred tails a launch ends at the cut and the stragglers run in the tail slices",
                       "valu_tflops": flops_per_step / step_s / 1e12,
                       "valu_frac": valu_frac, "valu_peak_tflops": VALU_PEAK_TFLOPS[dtype]}

    if world == 1 and stub is None and not args.no_host_leg:
        # PCIe-inclusive: the same batch from pageable host arrays through mpc_solve_batch_host (fp64 entry point)
        if not f32:
            hs = []
            for _ in range(3):
                th = time.perf_counter()
                rh = pipe.mpcs[0].solve_numpy(batch["state"], batch["coeffs"], batch["yaw_lo"], batch["yaw_hi"], weights=w_np, want_traj=want_traj)
                hs.append(time.perf_counter() - th)
            host_ok = bool(np.array_equal(rh["out"], out_np) and np.array_equal(rh["status"], status))
            p1 = params.copy()
            with pkg.BatchedMPC(p1, 1, device=local_rank) as m1:
                one = [a[..., :1].copy() for a in (batch["state"], batch["coeffs"], batch["yaw_lo"], batch["yaw_hi"])]
                lat = []
                for _ in range(60):
                    th = time.perf_counter(); r1 = m1.solve_numpy(*one, want_traj=want_traj); lat.append(time.perf_counter() - th)
                k1 = m1.stats().kernel_ms
            res["host_path"] = {"entry": "mpc_solve_batch_host (pageable host arrays, one pinned staging copy each way)",
                                "solves_per_s_incl_pcie": B / float(np.median(hs)), "ms_per_batch": 1e3 * float(np.median(hs)),
                                "matches_device_path_bitwise": host_ok,
                                "b1_latency_ms_median": 1e3 * float(np.median(lat[10:])), "b1_latency_ms_min": 1e3 * float(np.min(lat[10:])),
                                "b1_kernel_ms": k1, "b1_iterations": int(r1["iters"][0]),
                                "b1_note": "one MPC::solve() per telemetry message is what the reference does (mpc_main.cpp:167); includes the ctypes call; launches of at most 64 instances run one instance per wavefront (mpc_solve_wave_kernel)"}
    # IPOPT's own stopping rule (polish = 0) on the same batch, once: what the device returns then (compared with the oracle below)
    out_p0 = None
    if world == 1 and stub is None and not args.no_cpu_baseline and not f32:
        p0 = params.copy(); p0.polish = 0; p0.tail_cut = 0
        with pkg.BatchedMPC(p0, B, device=local_rank) as m0:
            r0 = m0.solve_torch(*tensors, weights=d_w, want_traj=False)
            torch.cuda.synchronize(dev)
            out_p0 = (r0["out"].cpu().numpy(), r0["status"].cpu().numpy())
    pipe.close()
    if world == 1 and stub is None and not args.no_legs:
        res["legs"] = extra_legs(args)
    if world == 1 and stub is None and not args.no_cpu_baseline:
        one, allc, parity = cpu_baseline_legs(args, batch, w_np, over, status, out_np, out_p0, args.cpu_seconds)
        res["cpu_baseline"] = one
        res["cpu_baseline_all_cores"] = allc
        res.update(parity)
        res["cpu_baseline_same_algorithm"] = cpu_twin_leg(args, pkg, params, batch, w_np, 0.5 * args.cpu_seconds)
    emit(res, json_fd, args)
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


def emit(res, json_fd, args):
    """Everything goes to bench_full.json (beside this file; `--full-json PATH` elsewhere) and to stderr; stdout gets ONE line of at
    most 2 KB: the contract's keys, the roofline and the baselines in short, and every leg as [solves/s, success, maxiter,
    linesearch, acceptable]."""
    full = json.dumps(res)
    path = args.full_json or os.path.join(ROOT, "bench_full.json")
    try:
        with open(path, "w") as f:
            f.write(full + "\n")
    except OSError as e:
        path = "not written: %s" % e
    sys.stderr.write(full + "\n")
    short = lambda x, n: (x if len(x) <= n else x[:n - 1] + "~") if isinstance(x, str) else x
    c = res["config"]
    out = {k: res[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    out["config"] = {"workload": short(c["workload"], 150), "population": c["population"], "batch_per_gpu": c["batch_per_gpu"], "global_batch": c["global_batch"],
                     "N": c["N"], "dt": c["dt"], "parallelism": short(c["parallelism"], 90), "collective_mode": short(c["collective_mode"], 110),
                     "gather_checked": c["gather_checked"], "gather_bytes_sent_per_rank_per_batch": c["gather_bytes_sent_per_rank_per_batch"],
                     "batches_per_collective": c["batches_per_collective"], "batches_in_flight": c["batches_in_flight"], "tail_cut": c["tail_cut"],
                     "timed": "primed pipeline, K batches to final"}
    rf = res["roofline"]
    out["roofline"] = {k: rf[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms_avg", "algorithmic_bytes_per_launch",
                                          "valu_frac", "traffic_frac_of_hbm_peak")}
    for k in ("cpu_baseline", "cpu_baseline_same_algorithm"):
        if k in res:
            out[k] = {q: short(v, 60) for q, v in res[k].items()}
    sc = res["status_counts"]
    out.update({"strict_value": res["strict_value"], "converged_fraction": res["converged_fraction"],
                "status": [sc.get(k, 0) for k in ("success", "maxiter", "linesearch", "acceptable")],
                "mean_iterations": res["mean_iterations"], "max_iterations": res["max_iterations"]})
    for k in ("max_abs_dsteer_vs_oracle", "max_abs_daccel_vs_oracle", "polish0_max_dsteer", "polish0_max_daccel", "parity_sample", "stub"):
        if k in res:
            out[k] = short(res[k], 80)
    if "legs" in res:
        out["legs"] = {}
        for name, l in res["legs"].items():
            if "solves_per_s" in l:
                q = l["status_counts"]
                out["legs"][name] = [round(l["solves_per_s"]), q.get("success", 0), q.get("maxiter", 0), q.get("linesearch", 0), q.get("acceptable", 0)]
            else:
                out["legs"][name] = "error"
    if "host_path" in res:
        hp = res["host_path"]
        out["host_path"] = [round(hp["solves_per_s_incl_pcie"]), round(hp["b1_latency_ms_median"], 3)]
    out["full"] = os.path.basename(path) if os.path.isabs(path) else path

    def sig(x):                                                    # six significant digits are plenty for a summary line
        if isinstance(x, float):
            return float("%.6g" % x)
        if isinstance(x, dict):
            return {k: sig(v) for k, v in x.items()}
        if isinstance(x, list):
            return [sig(v) for v in x]
        return x
    out = sig(out)
    line = json.dumps(out, separators=(",", ":"))
    for shed in (lambda: [out[k].__setitem__("sample", short(out[k]["sample"], 24)) for k in ("cpu_baseline_same_algorithm", "cpu_baseline") if k in out and "sample" in out[k]],
                 lambda: out["config"].__setitem__("workload", short(out["config"]["workload"], 80)),
                 lambda: out["config"].__setitem__("collective_mode", short(out["config"]["collective_mode"], 60)),
                 lambda: out["config"].__setitem__("parallelism", short(out["config"]["parallelism"], 40)),
                 lambda: out.pop("host_path", None), lambda: out["roofline"].pop("kernel", None)):
        if len(line) <= 2040:                                      # the driver keeps 2 KB of tail: shed what is repeated in the full file
            break
        shed()
        line = json.dumps(out, separators=(",", ":"))
    sys.stdout.flush()
    os.write(json_fd, (line + "\n").encode())


if __name__ == "__main__":
    main()
