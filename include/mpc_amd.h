/*
 * mpc_amd.h -- C ABI of the MI355X-native batched MPC solver.
 *
 * This is the drop-in boundary for ONE path of dr-tony-lin/CarND-MPC-Project:
 * MPC::solve() + FG_eval (src/control/MPC.cpp:14-155, 183-325), i.e. the call
 *     CppAD::ipopt::solve<Dvector, FG_eval>(options, vars, vars_lowerbound,
 *         vars_upperbound, constraints_lowerbound, constraints_upperbound,
 *         fg_eval, solution);                      // MPC.cpp:290-292
 * together with the set-up before it (MPC.cpp:204-281) and the unpacking after
 * it (MPC.cpp:306-324), for B independent problem instances at once.
 *
 * Plain pointers and sizes only; no C++/torch types; no exceptions cross this
 * boundary (integer return codes + a per-instance status array, mirroring the
 * reference's "print and still return solution.x" behaviour, MPC.cpp:295-303).
 * There is no CPU fallback: every entry point that computes needs a gfx950
 * device and fails with MPC_ERR_NO_DEVICE / MPC_ERR_HIP otherwise.
 *
 * Batch layout is struct-of-arrays ("quantity-major", like the reference's own
 * decision vector, MPC.cpp:56-63): element (q, i) of an array with Q rows is at
 * [q * ld + i], ld >= B, so that consecutive instances are consecutive in
 * memory and a wavefront's loads coalesce.
 */
#ifndef MPC_AMD_H
#define MPC_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPC_ABI_VERSION 4
#define MPC_MAX_TABLE 16
#define MPC_NW 12      /* Config::weights entries read by FG_eval (Config.h:14-61) */
#define MPC_NCOEF 5    /* road polynomial, zero padded: fit order is 2..4 (RoadGeometry.cpp:26-34) */
#define MPC_NSTATE 6   /* x, y, psi, v, cte, epsi (MPC.cpp:213-218) */
#define MPC_NOUT 9     /* x1,y1,psi1,v1,cte1,epsi1,delta0,a0,cost (MPC.cpp:322-324) */
#define MPC_MAX_N 64   /* largest horizon the kernels are sized for */

/* return codes */
enum {
  MPC_OK = 0,
  MPC_ERR_INVALID = -1,     /* bad argument (NULL, N out of range, ld < B ...) */
  MPC_ERR_NO_DEVICE = -2,   /* no gfx950 device / HIP runtime unavailable */
  MPC_ERR_HIP = -3,         /* a HIP call failed; see mpc_last_error() */
  MPC_ERR_UNSUPPORTED = -4, /* e.g. branch_mode LIVE, or run() with max_fit_order > 5 */
  MPC_ERR_IO = -5           /* config file unreadable / malformed */
};

/* per-instance status, modelled on CppAD::ipopt::solve_result<>::status_type
 * (MPC.cpp:295): 0 is the only value the reference treats as "ok". */
enum {
  MPC_STATUS_SUCCESS = 0,
  MPC_STATUS_MAXITER = 1,
  MPC_STATUS_LINESEARCH = 2,   /* step length fell below alpha_min (IPOPT: restoration) */
  MPC_STATUS_INFEASIBLE = 3,   /* initial state outside its own bounds (MPC.cpp:229-239 vs :269-281) */
  MPC_STATUS_NUMERIC = 4,      /* NaN/Inf met */
  MPC_STATUS_PENDING = 5,      /* deferred tails only: handed to the tail queue, final after mpc_tail_wait */
  MPC_STATUS_ACCEPTABLE = 6    /* IPOPT's STOP_AT_ACCEPTABLE_POINT = CppAD's stop_at_acceptable_point, which MPC.cpp:295-303 prints
                                * ("Ipopt failed with ...") and returns like every other non-success: acceptable_iter iterates in
                                * a row within the acceptable_* tolerances, or a line search that ran out of step length at such
                                * a point */
};

enum { MPC_BRANCH_FROZEN = 0, MPC_BRANCH_LIVE = 1 };
enum { MPC_F32_START_OFF = 0, MPC_F32_START_ON = 1, MPC_F32_START_AUTO = 2 };   /* MpcParams.f64_f32_start */
#define MPC_F32_START_AUTO_N 15
enum { MPC_PRECISION_F64 = 0, MPC_PRECISION_F32 = 1 };
enum { MPC_TAIL_OFF = 0, MPC_TAIL_AUTO = -1 };   /* MpcParams.tail_cut */
enum { MPC_LANE_COMPACT_AUTO = -1 };             /* MpcParams.lane_compact */

/* Everything the reference reads from `struct Config` statics on this path
 * (src/utils/Config.h:66-177) AFTER Config::load() has applied its unit
 * conversions and table rescaling (src/utils/Config.cpp:31-87).  Passed by
 * value per batch: no global state, unlike the reference's mutable Config. */
typedef struct MpcParams {
  int32_t abi_version;       /* MPC_ABI_VERSION */
  int32_t N;                 /* Config::N,  3..MPC_MAX_N */
  double dt;                 /* Config::dt */
  double Lf;                 /* Config::Lf */
  double weights[MPC_NW];    /* Config::weights[0..11] */
  double cte_panic;          /* Config::ctePanic */
  double epsi_panic;         /* Config::epsiPanic */
  double max_steering;       /* rad  (bounds, MPC.cpp:248-251) */
  double max_acceleration;   /* m/s2 (MPC.cpp:254-257) */
  double max_deceleration;   /* m/s2, negative */
  double max_speed;          /* m/s  (MPC.cpp:236-239) */
  int32_t n_steers, n_steer_speeds;          /* Vehicle::computeSpeedTarget tables */
  double steers[MPC_MAX_TABLE];
  double steer_speeds[MPC_MAX_TABLE];
  /* used only by the run() pre/post-processing entry points */
  int32_t n_yaw_changes, n_yaw_change_speeds;
  double yaw_changes[MPC_MAX_TABLE];
  double yaw_change_speeds[MPC_MAX_TABLE];
  int32_t max_fit_order;     /* Config::maxFitOrder */
  int32_t latency_ms;        /* Config::latency */
  double max_fit_error;      /* Config::maxFitError */
  double lookahead;          /* Config::lookahead = latency * 1e-3 */
  double steer_adj_thresh;   /* Config::steerAdjustmentThresh */
  double steer_adj_ratio;    /* Config::steerAdjustmentRatio (clamped to [0,0.1]) */
  double ipopt_timeout;      /* Config::ipoptTimeout: carried, not used (we cap iterations) */
  /* solver controls (the reference's IPOPT option string, MPC.cpp:160-179, plus
   * the IPOPT defaults it leaves untouched) */
  int32_t branch_mode;       /* MPC_BRANCH_FROZEN: CppAD tape recorded once at the start point */
  int32_t precision;         /* MPC_PRECISION_F64 (default), or MPC_PRECISION_F32: see mpc_solve_batch_device_f32 */
  int32_t max_iter;          /* IPOPT default 3000; default here 200 */
  int32_t pass_cut;          /* multi-phase solve: park instances still running after this many passes and finish them,
                              * re-packed into dense waves, in a further launch (0 = single launch, the default; more
                              * cuts in pass_cut_next; results are bitwise the same) */
  double tol;                /* IPOPT "tol", default 1e-8 */
  /* Termination polish (default on).  IPOPT stops at the FIRST iterate whose scaled optimality error is <= tol
   * (MPC.cpp:290-292 leaves that default untouched).  An output that the objective determines only weakly -- an
   * interior a0: the frozen tape has no a^2 term -- then still moves by up to ~1e-4 per Newton step, so two correct
   * solvers that stop one iterate apart differ by that much.  With polish != 0 an instance is converged when
   * E_0 <= tol AND the barrier parameter has reached its floor (tol/10) AND the last accepted step moved
   * (delta0, a0) by at most out_step_tol and every primal variable (the predicted trajectory, whose far end is the
   * least determined part) by at most out_step_tol / 0.03 = 1e-5: the returned point is then the central-path point IPOPT is converging
   * to, reproducible to ~1e-7 whatever the linear algebra (and a barrier parameter within 3x of the floor goes
   * to the floor directly).  Cost: +0.4 iterations per solve.  polish = 0 is IPOPT's own stopping rule. */
  double out_step_tol;       /* default 3e-7 (rad, m/s^2): leaves delta0 within 7e-8 and a0 within 2e-8 of the limit point */
  double tol_f32;            /* "tol" of the fp32 solver, default 5e-4 (barrier floor tol_f32/25 = 2e-5, polish step 0.6 tol_f32,
                              * line search failing inside 10 x tol_f32 = solved): where the fp32 phase of an F32 handle
                              * hands over at the latest, and the stopping rule of the pure fp32 mode (f32_finish = 0) */
  int32_t polish;            /* default 1 */
  int32_t pass_cut_next[3];  /* further cuts (passes counted from the previous cut; a zero ends the list), e.g.
                              * pass_cut 16, next {16, 32, 0}: four launches (0.3x the wave passes of a heavy-tailed
                              * batch; measured: no gain in time per batch) */
  /* IPOPT defaults the reference's option string (MPC.cpp:160-179) leaves untouched and that shape the answer when a
   * solve starts on a bound (closed loops, test.cpp:79-111): every finite variable bound is relaxed by
   * bound_relax_factor * max(1, |bound|) before the solve, and the returned point is projected back into the caller's
   * bounds (honor_original_bounds = yes in IPOPT 3.12). */
  int32_t honor_original_bounds;   /* default 1 */
  double bound_relax_factor;       /* default 1e-8 */
  /* Deferred tails: a launch lasts as long as its slowest instance.  With tail_cut > 0 an instance that
   * is still running after tail_cut passes is handed to the handle's tail queue (status MPC_STATUS_PENDING) and the
   * launch ends; the queue is worked off by short tail slices on the handle's own tail stream while later batches run.
   * mpc_tail_wait / mpc_tail_poll / mpc_tail_stream_wait mark a batch final.  Results are bitwise those of the single launch. */
  int32_t tail_cut;                /* default 0 = off; MPC_TAIL_AUTO (-1): the handle's own choice (from 20 passes up to N = 12, 24 beyond; it moves
                                    * the cut out while more than 8 % of a batch are handed over; results do not depend on the cut) */
  int32_t tail_ring;               /* batches whose tails may be outstanding at once (2..512), default 128: a batch is final
                                    * only when its slowest straggler is, tens of milliseconds behind its launch */
  int64_t tail_capacity;           /* deferred instances per batch; 0 = max_batch / 8.  A batch with more keeps the rest in its launch */
  /* Mixed precision across phases.  MPC_PRECISION_F32 handles: f32_finish = 1 (default) runs the interior-point
   * iteration in fp32 until its barrier parameter has reached mixed_switch_mu and finishes every instance in fp64
   * (same state machine, tol instead of tol_f32, termination polish), fp32 at the ABI; 0 = the pure fp32 solver.
   * MPC_PRECISION_F64 handles: f64_f32_start = 1 runs the same early iterations on the fp32 record (half the
   * workspace bytes) before the fp64 solve takes over; MPC_F32_START_AUTO (2) does so for horizons of
   * N >= MPC_F32_START_AUTO_N steps, where a full device's workspace no longer lives in the Infinity Cache and the bytes
   * count in full.  The fp64 phase runs to tol and the polish.  Only a CLEAN hand-over is continued (the fp32 phase
   * reached mixed_switch_mu or tol_f32 within 16 iterations); an instance that uses the allowance up or leaves the fp32
   * phase out of trouble (line search, inertia correction, not-a-number) is solved in fp64 from the start point, and
   * so is one the fp64 phase cannot finish from the fp32 iterate: on hard instances fp32 iterates lead into other
   * local minima than fp64 ones.  With that rule the CPU build gives the single-phase solve's status and point (1e-7)
   * on every instance of SURVEY's unfiltered populations (65 536 at N = 10, 32 768 at N = 25).
   * Default MPC_F32_START_AUTO: long horizons start on the fp32 record.  (Launches of at most wave_max_batch instances do not go
   * through the two-launch solve at all: they run one instance per wavefront, every iteration in fp64 -- DESIGN.md section 6d.)  Measured in round 4 with the tail slices on the
   * tail stream's high priority (windows of 600 batches): N = 25, SURVEY's population 9.8-10.0 M against 7.1-8.8 M solves/s,
   * the filtered generator 11.0-11.9 against 7.6 M; 0 status differences and at most 4.1e-8 rad on the first steering
   * angle over the 32 768 instances of SURVEY's N = 25 population (profiles/r04_f32start_vs_plain.jsonl).  Not for short
   * horizons: at N = 10 the workspace of a full device lives in the Infinity Cache and the two launches of the mixed
   * solve cost more than the bytes save (44 against 52 M on the filtered headline): f64_f32_start = 1 stays a choice.
   * MPC_F32_START_OFF gives the single-phase fp64 solve at every horizon (bitwise the host twin's). */
  int32_t f32_finish;
  int32_t f64_f32_start;
  double mixed_switch_mu;          /* default 2e-5 */
  /* Lane compaction.  Memory is fetched in 128-byte lines = the 16-byte groups of 8 neighbouring lanes, so a
   * line is fetched as long as one of its 8 lanes still runs.  With lane_compact = g > 0 a wave whose running lanes are
   * spread over g more 8-lane groups than they need moves the ones outside its fullest groups into free lanes inside
   * them (launches of at least 8 192 instances, single-lane arithmetic unchanged: results are bitwise the same).
   * Default MPC_LANE_COMPACT_AUTO: 2, and 1 for horizons of N >= 15, whose workspace (>= 360 KB per wave) is in HBM proper, where
   * every line not fetched counts (configs[3] share: 6.90 -> 7.26 M solves/s); 0 = off. */
  int32_t lane_compact;
  /* Mixed precision, heavy-tailed workloads (weight sweeps: some instances spend the whole allowance of the fp32 phase
   * while most are handed over after 8-12 iterations).  With f32_phase_refill = 1 a lane of the fp32 phase hands its
   * instance over through a buffer of its own instead of keeping the iterate in its workspace column, so the lane is
   * free: it takes further instances while the launch has any, and its column is there for lane compaction.  Same
   * results bit for bit.  Measured: +12-20 % on the weight sweeps of configs[4], -6...-15 % where every instance needs
   * about the same number of iterations (the copy, and late refills that prolong a wave for a few lanes).  Default 0. */
  int32_t f32_phase_refill;
  /* IPOPT's termination tests beyond `tol` (OptimalityErrorConvergenceCheck; all defaults of IPOPT 3.12 that the reference's
   * option string MPC.cpp:160-179 leaves alone).  Converged = scaled error <= tol AND unscaled dual infeasibility <=
   * dual_inf_tol, constraint violation <= constr_viol_tol, complementarity <= compl_inf_tol.  Acceptable = the same with
   * acceptable_tol and the acceptable_* values; acceptable_iter acceptable iterates in a row, or a failed line search at an
   * acceptable point, end the solve with MPC_STATUS_ACCEPTABLE; a line search that fails at an almost feasible point
   * (constraint violation <= 1e-2 tol) is final without the restart that stands in for IPOPT's restoration phase (IPOPT does
   * not try it there).  acceptable_iter = 0 switches the acceptable-level machinery off (IPOPT's meaning of 0).  fp64 solver
   * only: the fp32 phases stop by tol_f32. */
  int32_t acceptable_iter;         /* default 15 */
  double dual_inf_tol;             /* default 1 */
  double constr_viol_tol;          /* default 1e-4 */
  double compl_inf_tol;            /* default 1e-4 */
  double acceptable_tol;           /* default 1e-6 */
  double acceptable_dual_inf_tol;  /* default 1e10 */
  double acceptable_constr_viol_tol;   /* default 1e-2 */
  double acceptable_compl_inf_tol;     /* default 1e-2 */
  /* IPOPT's error measure in full.  The reference's NLP keeps the initial state as six VARIABLES pinned by six equality rows
   * (MPC.cpp:116-121, 269-281).  Those rows' multipliers and the bound duals of psi_0 / v_0 decouple from the Newton step, so
   * the device solver does not need them for its iterates -- but IPOPT counts the residuals of the six variables' stationarity
   * rows in the dual infeasibility (non-zero after a step the fraction-to-the-boundary rule has cut), their multipliers in its
   * scaling and the two variables' duals in the complementarity and in the dual step length.  1: carried and counted -- the
   * solver then takes the oracle's iteration count on 98.5-99.2 % of a batch instead of 94-96 % (what is left is the last
   * step's rounding) at ~4 % of the rate (ten more fields per instance and sweep).  0 (default): not carried; the barrier
   * parameter then comes down one iteration early on a few per cent of the instances.  Same solution either way. */
  int32_t initial_state_rows;
  /* Launches of at most this many instances run ONE INSTANCE PER WAVEFRONT -- or per 16 / 32 neighbouring lanes of one, a lane
   * per stage (mpc_solve_wave_kernel): the instance's N-step variables in LDS, its stages shared between
   * the lanes, bitwise the results of the lane-per-instance kernel.  A latency mapping: one MPC::solve() 0.30 instead of 0.68 ms,
   * a launch of 1 024 instances 0.55 instead of 1.21 ms, of 4 096 (N <= 17) 0.74 instead of 1.51 ms, level at ~10 000; it spends
   * a SIMD per 1-4 instances, so many launches in flight are better off with the lane kernel.  0 (default) = 1 024; < 0 = never.
   * Not used by an explicit f64_f32_start = 1 or a mixed MPC_PRECISION_F32 handle (those ask for the two-launch solve). */
  int32_t wave_max_batch;
} MpcParams;

typedef struct MpcHandle MpcHandle;

/* Aggregate statistics of the last batch (diagnostics for bench/tests). */
typedef struct MpcBatchStats {
  int64_t batch;
  int64_t n_success, n_maxiter, n_linesearch, n_infeasible, n_numeric, n_acceptable;
  int64_t iter_sum;          /* sum of interior-point iterations over the batch */
  int32_t iter_max;
  int32_t n_pending;         /* deferred tails: instances of the batch handed to the tail queue (their iterations are not in iter_sum yet) */
  double kernel_ms;          /* hipEvent time of the solve kernel, on its own stream */
} MpcBatchStats;

/* ---- parameters ---------------------------------------------------------- */
/* Compiled-in defaults of the reference, src/utils/Config.cpp:5-29. */
int mpc_params_default(MpcParams *p);
/* Config::load(fileName), src/utils/Config.cpp:31-87, same JSON keys. */
int mpc_params_load_json(const char *path, MpcParams *p);

/* ---- lifetime -------------------------------------------------------------- */
/* device < 0: current HIP device.  max_batch sizes the device workspace (3.7 KB per instance at N=10; 2.0 KB in fp32).
 * A handle owns its workspace and launch state: calls on one handle must be ordered (one stream at a time).
 * To keep several batches in flight -- which is how to fill the device, see DESIGN.md section 6b -- create one
 * handle per stream. */
int mpc_create(const MpcParams *p, int device, int64_t max_batch, MpcHandle **out);
/* (N, precision and the mixed-precision switches size the workspaces and cannot change on a live handle; with deferred tails
 * outstanding the call first lets them finish under the parameters their batches were issued with) */
int mpc_set_params(MpcHandle *h, const MpcParams *p);
void mpc_destroy(MpcHandle *h);
const char *mpc_last_error(void);
int mpc_abi_version(void);
int mpc_handle_device(const MpcHandle *h);   /* the device the handle's memory and launches live on */

/* ---- the hot path ---------------------------------------------------------- */
/*
 * Solve B instances; all pointers are DEVICE pointers (HBM resident):
 *   state  [6][ld]   initial state per instance            (MPC.cpp:213-218)
 *   coeffs [5][ld]   road polynomial c0..c4, zero padded   (RoadGeometry, MPC.cpp:151-152)
 *   yaw_lo [ld], yaw_hi [ld]   psi bounds = Config::yawLow/yawHigh as MPC::run
 *                    sets them just before solve()          (MPC.cpp:345-352, 229-232)
 *   weights[12][ld]  per-instance Config::weights, or NULL to use p->weights
 *   out    [9][ld]   the vector MPC::solve returns          (MPC.cpp:322-324)
 *   traj   [2N][ld]  x[0..N) then y[0..N) of the solution, or NULL (MPC.cpp:306-311)
 *   status [ld]      per-instance MPC_STATUS_*
 *   iters  [ld]      per-instance iteration count, or NULL
 * stream: the hipStream_t (as void*) to launch on; NULL is HIP's default (null)
 * stream.  The call is asynchronous with respect to the host: order later work
 * on the same stream, or synchronise that stream / the device, before reading.
 */
int mpc_solve_batch_device(MpcHandle *h, int64_t B, int64_t ld, const double *state,
                           const double *coeffs, const double *yaw_lo, const double *yaw_hi,
                           const double *weights, double *out, double *traj, int32_t *status,
                           int32_t *iters, void *stream);
/* MPC_PRECISION_F32 (BASELINE.json configs[4]: "fp32 mixed precision"): the same solve for a handle created with
 * params.precision = MPC_PRECISION_F32 -- fp32 inputs and outputs (136 B of algorithmic HBM traffic per solve with
 * per-instance weights and no trajectory).  Two modes:
 *   f32_finish = 1 (default)  fp32 interior-point iterations (Riccati sweeps and trigonometry in fp32; the road polynomial
 *       and f(x) - y in fp64; residuals as differences of neighbouring states first) until the barrier parameter is about to
 *       go below mixed_switch_mu, then EVERY instance is finished by the fp64 solver (same state machine, tol, polish).
 *       Against the fp64 path, on every instance (tests/test_f32.py, tolerances stated before measuring): |d delta0| <= 1e-3
 *       rad, |d a0| <= 1e-3 m/s^2, step-1 state <= 1e-3, trajectory <= 1e-2 m, same status.  Measured on a 131 072-instance
 *       weight sweep incl. velocity weight 0: delta0 max 5.5e-8, a0 max 1.0e-4, state max 5.5e-6 (the rounding of the fp32
 *       arrays), with ~1 instance in 50 000 converging to a neighbouring local minimum of equal cost (counted by the tests).
 *   f32_finish = 0            the pure fp32 solver: stops at tol_f32 (5e-4; barrier floor tol_f32 / 25 = 2e-5; a failed line
 *       search inside 10 x tol_f32 counts as solved), about twice the rate, measured against fp64 on 65 536 instances:
 *       delta0 max 2.4e-3 (p99 1.9e-4), a0 max 6.8e-2 when interior (p99.9 3.6e-4), state max 6.8e-3, trajectory max 0.12 m;
 *       velocity weight 0 and horizons beyond N = 10 are outside its specification.
 * An fp64 handle refuses this entry point and an fp32 handle refuses the double ones (MPC_ERR_INVALID); the run(),
 * telemetry and rollout entry points are fp64 only. */
int mpc_solve_batch_device_f32(MpcHandle *h, int64_t B, int64_t ld, const float *state,
                               const float *coeffs, const float *yaw_lo, const float *yaw_hi,
                               const float *weights, float *out, float *traj, int32_t *status,
                               int32_t *iters, void *stream);
/* Same as mpc_solve_batch_device, host pointers: one copy in, the launch, one copy out, synchronises. */
int mpc_solve_batch_host(MpcHandle *h, int64_t B, int64_t ld, const double *state,
                         const double *coeffs, const double *yaw_lo, const double *yaw_hi,
                         const double *weights, double *out, double *traj, int32_t *status,
                         int32_t *iters);
/* The same for an MPC_PRECISION_F32 handle (float arrays). */
int mpc_solve_batch_host_f32(MpcHandle *h, int64_t B, int64_t ld, const float *state,
                             const float *coeffs, const float *yaw_lo, const float *yaw_hi,
                             const float *weights, float *out, float *traj, int32_t *status,
                             int32_t *iters);
/* ---- the caller of the path: MPC::run() for a batch -------------
 * Pre-processing (waypoints to the vehicle frame, adaptive polynomial fit, cte0/epsi0, yaw bounds,
 * speed tables: src/control/MPC.cpp:329-356), the solve, and the post-processing (steering adjustment,
 * acceleration clamp, steering normalisation: MPC.cpp:360-381), all on the device:
 *   pose [6][ld]      x, y, psi, v, steering, acceleration of the (latency-compensated) vehicle
 *   ptsx, ptsy [npts][ld]  global waypoints in, VEHICLE-FRAME waypoints out (the reference transforms
 *                     them in place, mpc_main.cpp:189-190 relies on it); 3 <= npts <= 8
 *   out8 [8][ld]      {x1, y1, psi1, v1, steer in [-1,1], accel, cte1, epsi1} (MPC.cpp:381)
 *   traj [2N][ld] or NULL, status [ld], iters [ld] or NULL   as in mpc_solve_batch_device
 *   pre  [15][ld] or NULL  what run() handed to solve(): state[6], coeffs[5], yaw_lo, yaw_hi,
 *                     max_yaw_change, target_speed (diagnostic) */
int mpc_run_batch_device(MpcHandle *h, int64_t B, int64_t ld, int npts, const double *pose, double *ptsx,
                         double *ptsy, double *out8, double *traj, int32_t *status, int32_t *iters,
                         double *pre, void *stream);
/* The same for host arrays (all of them; ptsx / ptsy in: global, out: vehicle frame): one copy in, the kernels, one copy out, on the
 * handle's own device and stream; synchronises.  B = 1 is MPC::run() itself (include/mpc_drop_in.hpp delegates to it). */
int mpc_run_batch_host(MpcHandle *h, int64_t B, int64_t ld, int npts, const double *pose, double *ptsx, double *ptsy,
                       double *out8, double *traj, int32_t *status, int32_t *iters, double *pre);
/* The telemetry handler around run() (src/mpc_main.cpp:126-159, 171-174):
 *   tel [6][ld]   x, y, psi [rad], speed [mph], steering_angle (simulator sign), previous throttle command
 *   extra_latency seconds added to Config::lookahead (the handler's mean solve time, mpc_main.cpp:158)
 *   ptsx, ptsy    as in mpc_run_batch_device (in: global, out: vehicle frame)
 *   cmd [2][ld]   steering_angle and throttle of the reply (mpc_main.cpp:183-184)
 *   out8 [8][ld] or NULL   run()'s own return vector */
int mpc_telemetry_batch_device(MpcHandle *h, int64_t B, int64_t ld, int npts, const double *tel, double extra_latency,
                               double *ptsx, double *ptsy, double *cmd, double *out8, int32_t *status, void *stream);
/* The same for host arrays (tel, ptsx, ptsy read only; cmd [2][ld], status [ld] written): one copy in, the kernels, one copy
 * out, on the handle's own device and stream whatever the caller's current device is; synchronises. */
int mpc_telemetry_batch_host(MpcHandle *h, int64_t B, int64_t ld, int npts, const double *tel, double extra_latency,
                             const double *ptsx, const double *ptsy, double *cmd, int32_t *status);
/* Closed-loop rollout (the pattern of src/test.cpp:79-111): `steps` times
 * solve() and feed {x1,y1,psi1,v1,cte1,epsi1} back as the next state, cold start each time as the reference does.
 *   state [6][ld]          in: start states; out: the state after the last step
 *   hist  [steps][9][ld]   or NULL: solve()'s 9-vector of every step
 *   status[ld]             worst (largest) status code over the steps;  iters[ld] or NULL: iterations summed
 * The road polynomial and the psi-bounds stay fixed over the rollout, as in test.cpp (one fit, one run()). */
int mpc_rollout_batch_device(MpcHandle *h, int64_t B, int64_t ld, int steps, double *state, const double *coeffs,
                             const double *yaw_lo, const double *yaw_hi, const double *weights, double *hist,
                             int32_t *status, int32_t *iters, void *stream);
/* ---- the wire side of the handler (src/mpc_main.cpp:26-36, 81-222, DATA.md:5-16) --------
 * Everything between the bytes of a simulator frame and the bytes of the reply; the WebSocket server itself is out of
 * scope.  See csrc/mpc_wire.cpp. */
enum { MPC_WIRE_MANUAL = 0, MPC_WIRE_TELEMETRY = 1, MPC_WIRE_IGNORE = 2 };
typedef struct MpcWireTelemetry {
  double x, y, psi, speed, steering_angle, throttle;   /* as sent: psi [rad], speed [mph], simulator's steering sign */
  int32_t npts, reserved;
  double ptsx[8], ptsy[8];                             /* global waypoints */
} MpcWireTelemetry;
/* One text frame `42["telemetry",{...}]`: MPC_WIRE_TELEMETRY and *out filled; MPC_WIRE_MANUAL when the frame carries no
 * data (the reference answers `42["manual",{}]`, mpc_main.cpp:217-219); MPC_WIRE_IGNORE for anything that is not a "42"
 * telemetry event (the reference stays silent); MPC_ERR_INVALID for a malformed telemetry object (the reference would
 * throw out of json::parse). */
int mpc_wire_parse(const char *frame, int64_t len, MpcWireTelemetry *out);
/* The reply `42["steer",{...}]` byte for byte as mpc_main.cpp:183-197 builds it with nlohmann::json 2.1.1 (build without
 * PLOT_TRAJECTORY).  Returns the length (NUL-terminated in buf), or MPC_ERR_INVALID if cap is too small. */
int64_t mpc_wire_format_steer(double steering_angle, double throttle, char *buf, int64_t cap);
int64_t mpc_wire_format_manual(char *buf, int64_t cap);
/* B parsed frames, one per connection, through mpc_telemetry_batch_device (host arrays in and out).
 * prev_throttle[B] or NULL: throttle of each connection's previous reply; cmd [2][B]: steering_angle row, throttle row. */
int mpc_wire_telemetry_batch_host(MpcHandle *h, int64_t B, const MpcWireTelemetry *tel, const double *prev_throttle,
                                  double extra_latency, double *cmd, int32_t *status);
/* ---- deferred tails (MpcParams.tail_cut > 0) ----------------------------------------------------------------
 * A launch lasts as long as its slowest instance; on heavy-tailed workloads one 200-iteration straggler prices 65 536
 * solves.  With tail_cut = n the launch of mpc_solve_batch_device(_f32) hands every instance that is still running after
 * n passes to the handle's tail queue and ends: when the call's work on `stream` is complete, every other instance has
 * its final results and the handed-over ones carry status MPC_STATUS_PENDING (mpc_tail_pending counts them).  They are
 * carried on by TAIL SLICES on the handle's own high-priority stream: short launches (a bounded number of passes each)
 * that take the survivors of the slice before them and whatever the launches since have handed over, re-packed into dense
 * waves every time, and write the same arrays; results are bitwise those of an undisturbed launch.  Slices are started by
 * the calls below and by every solve call ("the pump"): nothing runs on its own between calls, so a serving loop that has
 * stopped issuing batches finishes with mpc_tail_wait (or polls).
 * The caller keeps a batch's output arrays alive and does not read its pending entries until the batch is final:
 *   mpc_last_batch_id     id of the batch the most recent solve call issued (ids count up from 1 per handle)
 *   mpc_tail_poll         1 if batch `id` is final, 0 if not yet (runs one turn of the pump, never blocks)
 *   mpc_tail_wait         blocks the host until batch `id` is final (id <= 0: every batch issued so far); for a batch that
 *                         did not defer (small batches, tail_cut = 0, the run()/telemetry/rollout/host entry points) that is
 *                         the completion of its own launch
 *   mpc_tail_stream_wait  orders `stream` behind the batch being final.  A batch that did not defer: a stream-side wait for its
 *                         launch, the host does not block.  A deferring batch is final behind slices that do not exist yet,
 *                         so the host pumps until it is (blocks), then `stream` is ordered behind the tail stream
 *   mpc_tail_flush        one turn of the pump (optional)
 * The handle resolves the ids of its last 1 024 batches; older ones return MPC_ERR_INVALID.  At most tail_ring batches may
 * be outstanding; issuing one more blocks the host until the oldest is final.  The survivors' list is bounded: while it is
 * more than half full, new batches run without deferral (counted in mpc_tail_info).
 * The run(), telemetry, rollout and host entry points never defer. */
int64_t mpc_last_batch_id(const MpcHandle *h);
/* How many batches of B instances to keep in flight (handles on streams of their own, INTEGRATION.md "batches in flight") for
 * these parameters, from the measurements behind DESIGN.md sections 6b/6c/6f: about two devices' worth of lanes (131 072
 * instances), between 2 and 8 launches -- 8 x 16 384 and 4 x 32 768 give what 2 x 65 536 gives -- and at least 4 for two-launch
 * solves (MPC_PRECISION_F32, the fp32 start) and horizons of 15 steps and more.  No device is touched. */
int mpc_inflight_advice(const MpcParams *p, int64_t B);
int mpc_tail_poll(MpcHandle *h, int64_t batch_id);
int mpc_tail_wait(MpcHandle *h, int64_t batch_id);
int mpc_tail_stream_wait(MpcHandle *h, int64_t batch_id, void *stream);
int mpc_tail_flush(MpcHandle *h);
int mpc_tail_pending(MpcHandle *h, int64_t batch_id, int64_t *n);   /* waits for the batch's own launch, then: how many it handed over */
/* out13: batches deferred so far, tail slices so far, ring, capacity of a batch's fresh queue, waves per slice (upper bound), 1 if the
 * tail stream has high priority, number of tail streams (1), batches that ran without deferral because the survivors' list was
 * filling up, passes per slice, survivors after the most recent retired slice, the cut in use (MPC_TAIL_AUTO: the current choice),
 * running mean of the deferred share of a batch in 1/65536 (-1: none retired yet), batches that handed over more than their fresh
 * queue holds (the rest finished in the launch; MPC_TAIL_AUTO then raises its cut) */
int mpc_tail_info(const MpcHandle *h, int64_t *out13);
int mpc_synchronize(MpcHandle *h);
/* Statistics of the most recent mpc_solve_batch_* call: gathered when asked for, from the status / iters arrays that call
 * wrote (they must still be there); waits for that call's launch. */
int mpc_get_stats(MpcHandle *h, MpcBatchStats *stats);

/* ---- diagnostics ------------------------------------------------------------ */
/* Evaluates the device's own light-weight math on n host values (the solver replaces libm's
 * sincos and IEEE division by shorter sequences, see csrc/mpc_core.h): sn[i], cs[i] = sin/cos(x[i]),
 * rc[i] = 1/x[i].  Used by the tests to bound their error on the real hardware. */
int mpc_debug_math(int device, int64_t n, const double *x, double *sn, double *cs, double *rc);
/* the same plus at[i] = atan(x[i]) and lg[i] = log|x[i]| (the solver's own atan and log kernels) */
int mpc_debug_math_ext(int device, int64_t n, const double *x, double *sn, double *cs, double *rc, double *at, double *lg);
/* Tile pool (MPC_TILE_POOL=1 in the environment of mpc_create): instead of the handle's own workspace the
 * waves of a launch take their tile from a pool of the XCD they run on, shared by all handles of that device and tile
 * size.  out32[4 x + 0..3] for XCD x = free tiles now, tiles in the pool, claims so far, highest tile number used + 1.
 * Synchronises the device.  MPC_ERR_UNSUPPORTED when the handle has no pool. */
int mpc_debug_tile_pool(MpcHandle *h, int64_t *out32);

#ifdef __cplusplus
}
#endif
#endif /* MPC_AMD_H */
