/*
 * mpc_run_core.h -- per-instance pre/post-processing of MPC::run().
 *
 * What the reference does around solve() in MPC::run (src/control/MPC.cpp:327-382), for one instance:
 *   pre:  Vehicle::globalToVehicle on the waypoints (Vehicle.cpp:105-114), RoadGeometry::fit with adaptive
 *         order 2..maxFitOrder-1 (RoadGeometry.cpp:18-34; polyfit = Vandermonde + Householder QR,
 *         utils.cpp:10-29), cte0 = f(0), epsi0 = -atan(c1), computeOrientationChange, speed tables,
 *         yaw bounds, state = (0,0,0,v,cte0,epsi0)                        (MPC.cpp:329-356)
 *   post: steering adjustment, acceleration clamp, steering normalisation   (MPC.cpp:360-381)
 * One instance per lane; everything in registers (the least-squares problem is at most 8 x 5).
 * Compiles for the device and, for the test-only host build, with g++.
 */
#ifndef MPC_RUN_CORE_H
#define MPC_RUN_CORE_H

#include "mpc_core.h"

namespace mpc {

enum : int { RUN_MAX_PTS = 8 };

/* least squares min ||A c - y|| with A[i][j] = x_i^j, i < n <= 8, j < NC, by Householder QR (the
 * factorisation Eigen's householderQr() performs in utils.cpp:24-26); everything unrolled */
template <int NC>
MPC_HD void polyfit_qr(const double *x, const double *y, int n, double *coef) {
  double A[RUN_MAX_PTS][NC], b[RUN_MAX_PTS];
  MPC_UNROLL
  for (int i = 0; i < RUN_MAX_PTS; i++) {
    const bool in = i < n;
    double pw = in ? 1.0 : 0.0;
    MPC_UNROLL
    for (int j = 0; j < NC; j++) { A[i][j] = pw; pw *= x[i]; }
    b[i] = in ? y[i] : 0.0;
  }
  MPC_UNROLL
  for (int k = 0; k < NC; k++) {
    double nrm2 = 0.0;
    MPC_UNROLL
    for (int i = k; i < RUN_MAX_PTS; i++) nrm2 += A[i][k] * A[i][k];
    const double nrm = sqrt(nrm2);
    const double alpha = A[k][k] > 0.0 ? -nrm : nrm;
    const double vk = A[k][k] - alpha;
    double vn2 = vk * vk;
    MPC_UNROLL
    for (int i = k + 1; i < RUN_MAX_PTS; i++) vn2 += A[i][k] * A[i][k];
    const double beta = vn2 > 0.0 ? 2.0 / vn2 : 0.0;
    MPC_UNROLL
    for (int j = k + 1; j < NC; j++) {
      double sdot = vk * A[k][j];
      MPC_UNROLL
      for (int i = k + 1; i < RUN_MAX_PTS; i++) sdot += A[i][k] * A[i][j];
      sdot *= beta;
      A[k][j] -= sdot * vk;
      MPC_UNROLL
      for (int i = k + 1; i < RUN_MAX_PTS; i++) A[i][j] -= sdot * A[i][k];
    }
    {
      double sdot = vk * b[k];
      MPC_UNROLL
      for (int i = k + 1; i < RUN_MAX_PTS; i++) sdot += A[i][k] * b[i];
      sdot *= beta;
      b[k] -= sdot * vk;
      MPC_UNROLL
      for (int i = k + 1; i < RUN_MAX_PTS; i++) b[i] -= sdot * A[i][k];
    }
    A[k][k] = alpha;
  }
  MPC_UNROLL
  for (int k = NC - 1; k >= 0; k--) {
    double sacc = b[k];
    MPC_UNROLL
    for (int j = k + 1; j < NC; j++) sacc -= A[k][j] * coef[j];
    coef[k] = (A[k][k] != 0.0) ? sacc / A[k][k] : 0.0;
  }
}

MPC_HD double poly5(const double *c, double x) { return (((c[4] * x + c[3]) * x + c[2]) * x + c[1]) * x + c[0]; }
MPC_HD double poly5_der(const double *c, double x) { return ((4.0 * c[4] * x + 3.0 * c[3]) * x + 2.0 * c[2]) * x + c[1]; }

/* table lookups of Vehicle::computeSpeedTarget / computeYawChangeSpeedLimit (Vehicle.cpp:34-79) */
MPC_HD double table_lookup(const double *xs, int nx, const double *ys, int ny, double key, double maxv) {
  const double a = fabs(key);
  for (int i = 0; i < nx; i++)
    if (a <= xs[i]) return fmin(ys[i < ny ? i : ny - 1], maxv);
  return fmin(ys[ny - 1], maxv);
}

/* RoadGeometry::orientation (RoadGeometry.cpp:41-47) */
MPC_HD double road_orientation(const double *c, double px, double dir) {
  double psi = atan(poly5_der(c, px));
  if (dir < 0.0) {
    psi += M_PI;
    while (psi >= M_PI) psi -= 2.0 * M_PI;
    while (psi < -M_PI) psi += 2.0 * M_PI;
  }
  return psi;
}

struct RunPre {
  double state[6], coef[MPC_NCOEF], yaw_lo, yaw_hi, max_yaw_change, target_speed;
  int ncoef;
};

/* The pre-solve half of MPC::run.  pose = {x, y, psi, v, steering, acceleration}; px/py hold the n global
 * waypoints on entry and the vehicle-frame waypoints on return (the reference transforms them in place). */
MPC_HD void run_pre(const MpcParams &P, const double *pose, double *px, double *py, int n, RunPre &R) {
  double sn, cs;
  fsincos(pose[2], &sn, &cs);
  MPC_UNROLL
  for (int i = 0; i < RUN_MAX_PTS; i++) {
    if (i < n) {
      const double vx = px[i] - pose[0], vy = py[i] - pose[1];                   /* Vehicle.cpp:108-112 */
      px[i] = vx * cs + vy * sn;
      py[i] = vy * cs - vx * sn;
    }
  }
  /* RoadGeometry::fit: order 2, 3, 4 while the squared error exceeds maxFitError (RoadGeometry.cpp:26-34) */
  double c[MPC_NCOEF] = {0, 0, 0, 0, 0};
  int order = 2, ncoef = 3;
  bool pending = true;
  MPC_UNROLL
  for (int pass = 0; pass < 3; pass++) {
    if (pending) {
      const int used = order++;
      double cc[MPC_NCOEF] = {0, 0, 0, 0, 0};
      if (pass == 0) polyfit_qr<3>(px, py, n, cc);
      else if (pass == 1) polyfit_qr<4>(px, py, n, cc);
      else polyfit_qr<5>(px, py, n, cc);
      double err = 0.0;
      MPC_UNROLL
      for (int i = 0; i < RUN_MAX_PTS; i++)
        if (i < n) { const double d = py[i] - poly5(cc, px[i]); err += d * d; }
      MPC_UNROLL
      for (int j = 0; j < MPC_NCOEF; j++) c[j] = cc[j];
      ncoef = used + 1;
      pending = (err > P.max_fit_error) && (order < P.max_fit_order);
    }
  }
  MPC_UNROLL
  for (int j = 0; j < MPC_NCOEF; j++) R.coef[j] = c[j];
  R.ncoef = ncoef;
  const double cte = c[0];                                                        /* f(0), MPC.cpp:334 */
  const double epsi = -atan(c[1]);                                                /* MPC.cpp:336 */
  double back = px[0], front = px[0];
  MPC_UNROLL
  for (int i = 0; i < RUN_MAX_PTS; i++) if (i == n - 1) back = px[i];
  const double dir = back - 0.0;                                                  /* computeOrientationChange(0, back) */
  R.max_yaw_change = (road_orientation(c, back, dir) - road_orientation(c, 0.0, dir)) * (back - front) / back;   /* :339 */
  const double max_speed = table_lookup(P.yaw_changes, P.n_yaw_changes, P.yaw_change_speeds, P.n_yaw_change_speeds,
                                        R.max_yaw_change, P.max_speed);           /* :340 */
  R.target_speed = table_lookup(P.steers, P.n_steers, P.steer_speeds, P.n_steer_speeds, pose[4], max_speed);   /* :342 */
  if (R.max_yaw_change < 0.0) { R.yaw_lo = R.max_yaw_change; R.yaw_hi = 0.1; }   /* :345-352 */
  else { R.yaw_lo = -0.1; R.yaw_hi = R.max_yaw_change; }
  R.state[0] = 0.0; R.state[1] = 0.0; R.state[2] = 0.0; R.state[3] = pose[3]; R.state[4] = cte; R.state[5] = epsi;   /* :355-356 */
}

/* The post-solve half (MPC.cpp:360-381): result9 -> {x1,y1,psi1,v1,steer in [-1,1],accel,cte1,epsi1} */
MPC_HD void run_post(const MpcParams &P, double max_yaw_change, double target_speed, double v0, const double *r9, double *o8) {
  double steer = r9[6];
  if (fabs(max_yaw_change) > P.steer_adj_thresh) steer += P.steer_adj_ratio * max_yaw_change;
  const double accel = fmin(r9[7], target_speed - v0);
  double sv = steer / P.max_steering;
  sv = sv < -1.0 ? -1.0 : (sv > 1.0 ? 1.0 : sv);
  o8[0] = r9[0]; o8[1] = r9[1]; o8[2] = r9[2]; o8[3] = r9[3]; o8[4] = sv; o8[5] = accel; o8[6] = r9[4]; o8[7] = r9[5];
}

/* ---- N2: the telemetry handler around run(), src/mpc_main.cpp:126-159 and :171-174 ------------------ */
/* tel = {x, y, psi (rad, any range), speed (mph), steering_angle (simulator sign), previous throttle command};
 * extra = the handler's mean solve time added to Config::lookahead (mpc_main.cpp:158).
 * -> pose {x,y,psi,v,steering,acceleration} after latency compensation (Vehicle::update + Vehicle::move) */
MPC_HD void telemetry_to_pose(const MpcParams &P, const double *tel, double extra, double *pose) {
  double psi = tel[2];
  while (psi >= M_PI) psi -= 2.0 * M_PI;                 /* normalizeAngle, mpc_main.cpp:127 */
  while (psi < -M_PI) psi += 2.0 * M_PI;
  const double v = tel[3] * 1609.34 / 3600.0;            /* MpH2MpS, :129 */
  const double steer = -tel[4];                          /* :131 */
  const double acc = (tel[5] - v / 50.0) * 6.0;          /* :156 */
  pose[0] = tel[0]; pose[1] = tel[1]; pose[2] = psi; pose[3] = v; pose[4] = steer; pose[5] = acc;
  if (P.latency_ms != 0) {                               /* :157-159, Vehicle::move (Vehicle.cpp:145-168) */
    const double dtm = P.lookahead + extra;
    const double dist = v * dtm;
    double sn, cs;
    fsincos(psi, &sn, &cs);
    pose[0] = tel[0] + dist * cs;
    pose[1] = tel[1] + dist * sn;
    pose[2] = psi + steer * dist / P.Lf;                 /* Vehicle::length = Config::Lf, mpc_main.cpp:155 */
    pose[3] = v + acc * dtm;                             /* not clamped: Vehicle.cpp:156 is overwritten at :167 */
  }
}
/* Vehicle::computeThrottle (Vehicle.cpp:81-103) */
MPC_HD double compute_throttle(const MpcParams &P, double accel, double target) {
  const double keep = target / P.max_speed;
  if (accel >= 0.0) return accel < 0.001 ? keep : fmin(1.0, keep + (1.0 - keep) * accel / P.max_acceleration);
  if (accel <= -15.0) return -1.0;
  const double base = accel < -10.0 ? 0.95 : (accel < -5.0 ? 0.9 : 0.85);
  return -base - (1.0 - base) * accel / P.max_deceleration;
}
/* mpc_main.cpp:171-174: run()'s result -> the command sent back to the simulator */
MPC_HD void command_from_run(const MpcParams &P, const double *o8, double *steer_cmd, double *throttle_cmd) {
  *steer_cmd = -o8[4];
  *throttle_cmd = compute_throttle(P, o8[5], o8[3]);
}

}  // namespace mpc
#endif
