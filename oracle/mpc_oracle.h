/*
 * mpc_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference hot path of
 * dr-tony-lin/CarND-MPC-Project:  MPC::solve() + FG_eval (src/control/MPC.cpp)
 * and of the small host-side pieces around it (MPC::run, RoadGeometry,
 * Vehicle, Config::load, polyfit).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the shipped HIP path
 * never does.
 *
 * PARITY STATUS: "parity unpinned" at the optimiser seam.  The reference
 * solves its NLP with IPOPT (>= 3.12.7, MUMPS) through CppAD::ipopt::solve
 * (src/control/MPC.cpp:290-292); neither library exists in this image and the
 * reference cannot be built without them, and the reference repository holds
 * no numeric golden vectors for solve().  The NLP itself (objective, residuals,
 * bounds, start point, output packing) is restated line by line from the
 * reference and pinned by the values recorded in BASELINE.md section 2; the
 * optimiser is a restatement of IPOPT's PUBLISHED algorithm (Waechter &
 * Biegler, Math. Prog. 106(1), 2006: primal-dual interior point, filter line
 * search, monotone barrier update, inertia-corrected symmetric indefinite
 * factorisation) on the dense KKT system, cross-checked against scipy
 * (tests/golden/make_golden.py) and against curves digitised from the
 * reference's own result figures (tests/golden/make_plot_anchors.py).
 *
 * All arithmetic is IEEE fp64, as in the reference.
 */
#ifndef MPC_ORACLE_H
#define MPC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_TABLE 16
#define ORC_MAX_COEF 8   /* reference fits order 2..4 => 3..5 coefficients */
#define ORC_NW 12        /* weights used by FG_eval: indices 0..11 (Config.h:14-61) */

/* Mirror of the reference's `struct Config` statics AFTER Config::load()
 * (src/utils/Config.cpp:31-87): every derived value (unit conversions, table
 * rescaling, clamps) is already applied. */
typedef struct OrcConfig {
  int N;                    /* Config::N            */
  double dt;                /* Config::dt           */
  double ipopt_timeout;     /* Config::ipoptTimeout (unused by the oracle) */
  long latency;             /* Config::latency [ms] */
  double lookahead;         /* latency * 1e-3       */
  int max_fit_order;        /* Config::maxFitOrder  */
  double max_fit_error;     /* Config::maxFitError  */
  double max_steering;      /* rad                  */
  double max_acceleration;  /* m/s^2 (MpH2MpS of the json value) */
  double max_deceleration;  /* m/s^2, negative      */
  double max_speed;         /* m/s                  */
  double yaw_low, yaw_high; /* Config::yawLow/yawHigh, written by MPC::run */
  double steer_adj_thresh;  /* Config::steerAdjustmentThresh */
  double steer_adj_ratio;   /* clamped to [0,0.1]   */
  double Lf;
  double cte_panic, epsi_panic;
  int n_weights;
  double weights[ORC_MAX_TABLE];
  int n_steers, n_steer_speeds, n_yaw_changes, n_yaw_change_speeds;
  double steers[ORC_MAX_TABLE], steer_speeds[ORC_MAX_TABLE];
  double yaw_changes[ORC_MAX_TABLE], yaw_change_speeds[ORC_MAX_TABLE];
} OrcConfig;

/* Branch handling of FG_eval's data-dependent `if`s.
 * FROZEN: decided once at the start point xi, as CppAD does when it records
 *         the tape once (no "Retape" option in MPC.cpp:160-179).
 * LIVE:   re-decided at every evaluation. */
enum { ORC_BRANCH_FROZEN = 0, ORC_BRANCH_LIVE = 1 };

/* Solver status, modelled on CppAD::ipopt::solve_result<>::status_type. */
enum {
  ORC_SUCCESS = 0,
  ORC_MAXITER_EXCEEDED = 1,
  ORC_RESTORATION_FAILURE = 2,  /* line search hit alpha_min (IPOPT would enter restoration) */
  ORC_INFEASIBLE_START = 3,     /* initial state violates its own bounds (MPC.cpp:229-239 vs :269-281) */
  ORC_NUMERIC_ERROR = 4,
  ORC_STOP_AT_ACCEPTABLE = 6    /* CppAD's stop_at_acceptable_point = IPOPT's STOP_AT_ACCEPTABLE_POINT (5 is the device's PENDING) */
};

typedef struct OrcSolveOptions {
  int branch_mode;      /* ORC_BRANCH_FROZEN (default) */
  int max_iter;         /* IPOPT default 3000; oracle default 500 */
  double tol;           /* IPOPT default 1e-8 */
  int lam_init_ls;      /* 1: least-squares multiplier start (IPOPT default); 0: zero */
  int obj_scaling;      /* 1: IPOPT gradient-based objective scaling (default) */
  int verbose;
  /* Termination polish (default 1): IPOPT returns the FIRST iterate with E_0 <= tol; an output the objective
   * determines only weakly (an interior a0: no a^2 term on the frozen tape) still moves ~1e-4 per Newton step
   * there.  With polish the solve ends when E_0 <= tol, mu has reached its floor tol/10 and the last accepted
   * step moved (delta0, a0) by <= out_step_tol and any primal variable by <= out_step_tol / 0.03: the central-path point IPOPT converges to, which any correct
   * implementation reproduces to ~1e-7 (a barrier parameter within 3x of the floor goes to the floor directly).  polish = 0 is IPOPT's own stopping rule.  (Same rule, same constants
   * as MpcParams.polish / out_step_tol of include/mpc_amd.h.) */
  int polish;
  double out_step_tol;
  /* IPOPT defaults the reference leaves untouched (MPC.cpp:160-179): every finite variable bound is relaxed by
   * bound_relax_factor * max(1, |bound|) before the solve (1e-8), and the returned point is projected back into the
   * caller's bounds (honor_original_bounds = yes, the IPOPT 3.12 default). */
  double bound_relax_factor;
  int honor_original_bounds;
  /* IPOPT's termination tests beyond `tol`, all defaults of 3.12 that MPC.cpp:160-179 leaves alone
   * (OptimalityErrorConvergenceCheck): an iterate is CONVERGED when the scaled error E_0 <= tol AND the unscaled dual
   * infeasibility <= dual_inf_tol (1), constraint violation <= constr_viol_tol (1e-4), complementarity <= compl_inf_tol
   * (1e-4).  It is ACCEPTABLE when E_0 <= acceptable_tol (1e-6) and the unscaled quantities are within
   * acceptable_dual_inf_tol (1e10) / acceptable_constr_viol_tol (1e-2) / acceptable_compl_inf_tol (1e-2);
   * acceptable_iter (15) acceptable iterates in a row end the solve with STOP_AT_ACCEPTABLE_POINT, which
   * CppAD::ipopt::solve reports as stop_at_acceptable_point and MPC.cpp:295-303 prints and returns like any other
   * non-success.  The line search keeps the most recent acceptable iterate (BacktrackingLineSearch::
   * StoreAcceptablePoint): when it runs out of step length AT an acceptable point the solve ends there with the same
   * status ("restoration phase called at acceptable point"), at an almost feasible point (constraint violation <=
   * 1e-2 tol) IPOPT does not start its restoration phase at all -- the stored point is returned if there is one,
   * otherwise the solve fails -- and a failed restoration falls back to the stored point.  acceptable_iter = 0
   * switches all of it off (IPOPT's own meaning of 0). */
  double dual_inf_tol, constr_viol_tol, compl_inf_tol;
  double acceptable_tol, acceptable_dual_inf_tol, acceptable_constr_viol_tol, acceptable_compl_inf_tol;
  int acceptable_iter;
  /* IPOPT's second-order correction (W&B section 2.4; IPOPT default max_soc = 4, kappa_soc = 0.99): when the first trial point of
   * a line search is rejected and its constraint violation is not below the current one, up to max_soc corrected steps
   * (same matrix, constraint right-hand side alpha c(x) + c(x + alpha d)) are tried before the step is halved.  Default 0:
   * NOT part of the restated algorithm (measured on SURVEY's populations: DESIGN.md section 3); an experiment switch. */
  int max_soc;
} OrcSolveOptions;

typedef struct OrcSolveInfo {
  int status;
  int iterations;
  double kkt_error;     /* final scaled E_0 */
  double mu;
  double obj;           /* unscaled objective */
  double constr_viol;   /* max |g - bound| */
  double dual_inf;      /* unscaled max |grad_x L| */
  double compl_inf;     /* unscaled max complementarity */
  int n_regularised;    /* iterations that needed delta_w > 0 */
  int n_backtracks;
  int acceptable_restored_older;   /* 1: STOP_AT_ACCEPTABLE returned a stored iterate that was not the current one */
  int no_restart;       /* 1: the line search failed at an almost feasible point: IPOPT does not try its restoration phase there */
  int n_soc_tried, n_soc_accepted;   /* line searches that tried / accepted a second-order correction (max_soc > 0) */
} OrcSolveInfo;

void orc_default_options(OrcSolveOptions *opt);

/* Config::load, src/utils/Config.cpp:31-87.  Returns 0 on success. */
int orc_config_load(const char *path, OrcConfig *cfg);
/* Compiled-in defaults, src/utils/Config.cpp:5-29. */
void orc_config_defaults(OrcConfig *cfg);

/* src/utils/utils.h:11-21, 28-47, 56-58, 87-92 */
double orc_mph2mps(double mph);
double orc_polyeval(const double *c, int nc, double x);
double orc_polyder(const double *c, int nc, double x);
double orc_normalize_angle(double a);
/* src/utils/utils.cpp:10-29 (Vandermonde + Householder QR least squares) */
int orc_polyfit(const double *x, const double *y, int n, int order, double *coef);
/* src/model/RoadGeometry.cpp:18-34: adaptive order; returns #coefficients */
int orc_road_fit(const double *x, const double *y, int n, int max_fit_order,
                 double max_fit_error, double *coef, double *fiterr);
/* src/model/RoadGeometry.cpp:41-47, 57-61 */
double orc_orientation(const double *c, int nc, double px, double dir);
double orc_orientation_change(const double *c, int nc, double x0, double x1);
/* src/model/Vehicle.cpp:34-48, 66-79, 81-103 */
double orc_speed_target(const OrcConfig *cfg, double angle, double max);
double orc_yaw_change_speed_limit(const OrcConfig *cfg, double yaw_change, double max);
double orc_compute_throttle(const OrcConfig *cfg, double accel, double target,
                            double max_accel, double max_decel);
/* src/model/Vehicle.cpp:105-114 */
void orc_global_to_vehicle(double vx, double vy, double vpsi, double *xs, double *ys, int n);
/* src/model/Vehicle.cpp:145-168; pose = {x,y,psi,v,steering,acceleration} */
void orc_vehicle_move(const OrcConfig *cfg, double *pose, double dt);

/* FG_eval::operator(), src/control/MPC.cpp:50-154.  vars has 8N-2 entries in
 * the reference's quantity-major layout; fg has 1+6N entries.  xi is the
 * point at which branches are decided (ignored in LIVE mode, may be NULL). */
void orc_fg_eval(const OrcConfig *cfg, const double *coef, int nc, int branch_mode,
                 const double *xi, const double *vars, double *fg);

/* MPC::solve(), src/control/MPC.cpp:183-325.  state[6]; out9 =
 * {x1,y1,psi1,v1,cte1,epsi1,delta0,a0,cost}; traj_x/traj_y (N each) may be
 * NULL; sol (8N-2, the full solution.x) may be NULL.  Uses cfg->yaw_low/high. */
int orc_mpc_solve(const OrcConfig *cfg, const OrcSolveOptions *opt, const double *state,
                  const double *coef, int nc, double *out9, double *traj_x, double *traj_y,
                  double *sol, OrcSolveInfo *info);

/* The pre-solve half of MPC::run(), src/control/MPC.cpp:329-356: transforms
 * ptsx/ptsy in place to the vehicle frame, fits the road, and produces the
 * solve() inputs.  pose = {x,y,psi,v,steering,acceleration}. */
typedef struct OrcRunPre {
  int nc; double coef[ORC_MAX_COEF];
  double state[6];
  double max_yaw_change, max_speed, target_speed, yaw_low, yaw_high;
} OrcRunPre;
void orc_mpc_run_pre(const OrcConfig *cfg, const double *pose, double *ptsx, double *ptsy,
                     int npts, OrcRunPre *pre);
/* The post-solve half, src/control/MPC.cpp:360-381: result9 -> out8 =
 * {x1,y1,psi1,v1,steer in [-1,1],accel,cte1,epsi1}. */
void orc_mpc_run_post(const OrcConfig *cfg, const OrcRunPre *pre, double v0,
                      const double *result9, double *out8);
/* MPC::run() whole, src/control/MPC.cpp:327-382.  cfg->yaw_low/high are
 * overwritten exactly as the reference mutates Config::yawLow/yawHigh. */
int orc_mpc_run(OrcConfig *cfg, const OrcSolveOptions *opt, const double *pose, double *ptsx,
                double *ptsy, int npts, double *out8, double *traj_x, double *traj_y,
                OrcRunPre *pre_out, OrcSolveInfo *info);

/* Derivative access for tests (analytic, verified against finite differences
 * of orc_fg_eval in tests/test_oracle.py).  Dense row-major. */
void orc_fg_grad(const OrcConfig *cfg, const double *coef, int nc, int branch_mode,
                 const double *xi, const double *vars, double *grad_f /*n*/,
                 double *jac /* 6N x n */);
void orc_lag_hess(const OrcConfig *cfg, const double *coef, int nc, int branch_mode,
                  const double *xi, const double *vars, double obj_factor,
                  const double *lam /*6N*/, double *hess /* n x n */);

/* KKT certificate of a candidate solution `vars` of the solve() NLP (frozen
 * branches at the reference start point): returns max(stationarity residual
 * with optimal multipliers estimated by bounded least squares, primal
 * infeasibility, bound violation).  Solver-independent. */
double orc_kkt_certificate(const OrcConfig *cfg, const double *state, const double *coef,
                           int nc, const double *vars, double active_tol,
                           double *stat_res, double *prim_res, double *bound_res);

#ifdef __cplusplus
}
#endif
#endif
