/*
 * mpc_drop_in.hpp -- host-side C++ mirror of the reference's operator interface for the
 * accelerated path: `class MPC` with the reference's exact method signatures, on top of
 * the C ABI (mpc_amd.h).  A user of dr-tony-lin/CarND-MPC-Project swaps
 *     #include "control/MPC.h"      ->   #include "mpc_drop_in.hpp"
 * links libmpc_amd.so, and src/mpc_main.cpp / src/test.cpp compile unchanged.
 *
 * What is mirrored (names, argument meaning, return vectors, ownership, error behaviour):
 *   class MPC            src/control/MPC.h:12-56,  MPC.cpp:160-382
 *   class Vehicle        src/model/Vehicle.h:7-174, Vehicle.cpp
 *   class RoadGeometry   src/model/RoadGeometry.h:11-101, RoadGeometry.cpp
 *   struct Config        src/utils/Config.h:9-184, Config.cpp
 *   free functions       src/utils/utils.h  (MpH2MpS, polyeval, polyder, clamp, normalizeAngle, ...)
 * The numerics of MPC::solve() run on the GPU through mpc_solve_batch_host(); everything
 * else here is the small host arithmetic of MPC::run() (frame change, polynomial fit,
 * speed tables, post-processing).  Nothing is copied from the reference: the code below
 * is written against the behaviour documented in SURVEY.md and is Eigen-free (the solve()
 * state argument is any type with operator[], e.g. Eigen::VectorXd or std::vector<double>).
 *
 * Differences a maintainer should know (INTEGRATION.md has the full list):
 *   - Config is still a bag of statics (the reference's callers write Config::maxSpeed,
 *     Config::latency directly, mpc_main.cpp:238-246), but the solver itself receives an
 *     MpcParams snapshot per call, so concurrent MPC objects are safe as long as Config
 *     is not mutated concurrently.  Needs C++17 (inline static members).
 *   - IPOPT's 0.5 s wall-clock cap (MPC.cpp:176-178) becomes an iteration cap.
 *   - Failure: the reference prints "Ipopt failed with <status>" and returns the iterate
 *     (MPC.cpp:295-303); so does this class.  With EXIT_ON_IPOPT_FAILURE it throws a
 *     std::string, as the reference does.
 *   - MPC::solveBatch() is new: B independent problems in one call.
 */
#ifndef MPC_DROP_IN_HPP
#define MPC_DROP_IN_HPP

#include <cmath>
#include <cstddef>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "mpc_amd.h"

/* ---- utils.h ------------------------------------------------------------------------- */
inline double MpH2MpS(double mph) { return mph * 1609.34 / 3600.0; }   /* utils.h:11-13 */
inline double MpS2MpH(double mps) { return mps * 3600.0 / 1609.34; }   /* utils.h:19-21 */
inline double deg2rad(double x) { return x * M_PI / 180; }             /* utils.h:69 */
inline double rad2deg(double x) { return x * 180 / M_PI; }             /* utils.h:75 */
template <typename T> T square(const T &a) { return a * a; }           /* utils.h:81 */
template <typename T> T clamp(const T a, const T lo, const T hi) { return a < lo ? lo : (a > hi ? hi : a); }
template <typename T> T normalizeAngle(const T &a) {                   /* utils.h:87-92: [-pi, pi) */
  T r = a;
  while (r >= M_PI) r -= 2. * M_PI;
  while (r < -M_PI) r += 2. * M_PI;
  return r;
}
const double EPSILON = 1E-6;                                           /* utils.h:95 */

/* coefficient containers are plain std::vector<double>, lowest order first */
template <typename T> T polyeval(const std::vector<double> &c, T x) {  /* utils.h:28-34 */
  T r = 0;
  for (int i = (int)c.size() - 1; i >= 0; i--) r = r * x + c[i];
  return r;
}
template <typename T> T polyder(const std::vector<double> &c, const T &x) {  /* utils.h:41-47 */
  T r = 0;
  for (int i = (int)c.size() - 1; i >= 1; i--) r = r * x + i * c[i];
  return r;
}
/* least-squares polynomial fit (utils.cpp:10-29).  The reference builds a Vandermonde matrix and
 * calls Eigen's Householder QR; here the same least-squares problem is solved by modified
 * Gram-Schmidt QR on the (at most 6 x 5) matrix. */
inline std::vector<double> polyfit(const std::vector<double> &xs, const std::vector<double> &ys, int order) {
  const int n = (int)xs.size(), m = order + 1;
  std::vector<double> Q((size_t)n * m), R((size_t)m * m, 0.0), c(m, 0.0), qty(m, 0.0);
  for (int i = 0; i < n; i++) { double p = 1.0; for (int j = 0; j < m; j++) { Q[(size_t)i * m + j] = p; p *= xs[i]; } }
  for (int j = 0; j < m; j++) {
    for (int pass = 0; pass < 2; pass++)                                 /* re-orthogonalise once */
      for (int k = 0; k < j; k++) {
        double d = 0; for (int i = 0; i < n; i++) d += Q[(size_t)i * m + k] * Q[(size_t)i * m + j];
        R[(size_t)k * m + j] += d;
        for (int i = 0; i < n; i++) Q[(size_t)i * m + j] -= d * Q[(size_t)i * m + k];
      }
    double nr = 0; for (int i = 0; i < n; i++) nr += Q[(size_t)i * m + j] * Q[(size_t)i * m + j];
    nr = std::sqrt(nr); R[(size_t)j * m + j] = nr;
    for (int i = 0; i < n; i++) Q[(size_t)i * m + j] = nr > 0 ? Q[(size_t)i * m + j] / nr : 0.0;
  }
  for (int j = 0; j < m; j++) { double d = 0; for (int i = 0; i < n; i++) d += Q[(size_t)i * m + j] * ys[i]; qty[j] = d; }
  for (int j = m - 1; j >= 0; j--) {
    double s = qty[j];
    for (int k = j + 1; k < m; k++) s -= R[(size_t)j * m + k] * c[k];
    c[j] = R[(size_t)j * m + j] != 0 ? s / R[(size_t)j * m + j] : 0.0;
  }
  return c;
}

/* ---- Config ---------------------------------------------------------------------------- */
/* The reference's bag of mutable statics (Config.h:66-177), same member names, so that callers such as
 * mpc_main.cpp:238-246 (`Config::latency = ...; Config::maxSpeed = MpH2MpS(...)`) and test.cpp:64,76
 * compile unchanged.  snapshot() turns the current statics into the MpcParams the solver receives. */
struct Config {
  static const int WEIGHT_CTE = 0, WEIGHT_EPSI = 1, WEIGHT_V = 2, WEIGHT_DELTA = 3, WEIGHT_DDELTA = 4,
                   WEIGHT_A = 6, WEIGHT_DA = 7, WEIGHT_DECEL_LOW_V = 8, WEIGHT_NEG_V = 9, WEIGHT_LARGE_EPSI = 10,
                   WEIGHT_LARGE_CTE = 11;
  inline static size_t N = 25;
  inline static double dt = 0.025, ipoptTimeout = 0.5, lookahead = 0, maxFitError = 0.5;
  inline static long latency = 100;
  inline static int maxFitOrder = 4;
  inline static double maxSteering = deg2rad(25.0), maxAcceleration = MpH2MpS(8), maxDeceleration = MpH2MpS(-20);
  inline static double maxSpeed = MpH2MpS(100), yawLow = -0.1, yawHigh = 0.1;
  inline static double steerAdjustmentThresh = 0.6, steerAdjustmentRatio = 0.025, Lf = 2.67, ctePanic = 0.6, epsiPanic = 1;
  inline static std::vector<double> weights = {100, 100, 1, 1, 1, 5000, 1, 1000};
  inline static std::vector<double> steers = {0.1, 0.2, 0.3};
  inline static std::vector<double> steerSpeeds = {MpH2MpS(80), MpH2MpS(65), MpH2MpS(30), MpH2MpS(25)};
  inline static std::vector<double> yawChanges, yawChangeSpeeds;
  /* solver controls that have no counterpart among the reference's statics */
  inline static int maxIterations = 200;
  inline static double tolerance = 1e-8;
  /* MpcParams.f64_f32_start of the handles this class creates: 2 (default) = the library's own choice, MPC_F32_START_AUTO (long
   * horizons start on the fp32 record; every solve still finished by the fp64 solver to `tolerance` and the polish); 0 = every
   * iteration in fp64; 1 = always the two-launch solve.  One MPC::solve() per telemetry message runs one instance per wavefront
   * whatever the horizon (0.30 ms at N = 10) unless this is 1. */
  inline static int fp32Start = 2;

  /* Config::load(fileName), Config.cpp:31-87 (parsing and unit conversion live behind the C ABI) */
  static void load(const std::string &fileName) {
    MpcParams p;
    if (mpc_params_load_json(fileName.c_str(), &p) != MPC_OK) throw std::string("Config::load failed: ") + fileName;
    N = (size_t)p.N; dt = p.dt; ipoptTimeout = p.ipopt_timeout; latency = p.latency_ms; lookahead = p.lookahead;
    maxFitOrder = p.max_fit_order; maxFitError = p.max_fit_error; maxSteering = p.max_steering;
    maxAcceleration = p.max_acceleration; maxDeceleration = p.max_deceleration; maxSpeed = p.max_speed;
    steerAdjustmentThresh = p.steer_adj_thresh; steerAdjustmentRatio = p.steer_adj_ratio; Lf = p.Lf;
    ctePanic = p.cte_panic; epsiPanic = p.epsi_panic;
    weights.assign(p.weights, p.weights + MPC_NW);
    steers.assign(p.steers, p.steers + p.n_steers);
    steerSpeeds.assign(p.steer_speeds, p.steer_speeds + p.n_steer_speeds);
    yawChanges.assign(p.yaw_changes, p.yaw_changes + p.n_yaw_changes);
    yawChangeSpeeds.assign(p.yaw_change_speeds, p.yaw_change_speeds + p.n_yaw_change_speeds);
  }
  static MpcParams snapshot() {
    MpcParams p;
    mpc_params_default(&p);
    p.N = (int32_t)N; p.dt = dt; p.ipopt_timeout = ipoptTimeout; p.latency_ms = (int32_t)latency; p.lookahead = lookahead;
    p.max_fit_order = maxFitOrder; p.max_fit_error = maxFitError; p.max_steering = maxSteering;
    p.max_acceleration = maxAcceleration; p.max_deceleration = maxDeceleration; p.max_speed = maxSpeed;
    p.steer_adj_thresh = steerAdjustmentThresh; p.steer_adj_ratio = steerAdjustmentRatio; p.Lf = Lf;
    p.cte_panic = ctePanic; p.epsi_panic = epsiPanic; p.max_iter = maxIterations; p.tol = tolerance;
    p.f64_f32_start = fp32Start == 1 ? MPC_F32_START_ON : (fp32Start == 0 ? MPC_F32_START_OFF : MPC_F32_START_AUTO);
    for (int i = 0; i < MPC_NW; i++) p.weights[i] = i < (int)weights.size() ? weights[i] : 0.0;
    auto put = [](const std::vector<double> &v, double *dst, int32_t &n) {
      n = (int32_t)(v.size() < MPC_MAX_TABLE ? v.size() : MPC_MAX_TABLE);
      for (int i = 0; i < n; i++) dst[i] = v[i];
    };
    put(steers, p.steers, p.n_steers); put(steerSpeeds, p.steer_speeds, p.n_steer_speeds);
    put(yawChanges, p.yaw_changes, p.n_yaw_changes); put(yawChangeSpeeds, p.yaw_change_speeds, p.n_yaw_change_speeds);
    return p;
  }
};

/* ---- Vehicle ----------------------------------------------------------------------------- */
class Vehicle {                                                        /* Vehicle.h:7-174 */
  double x = 0, y = 0, orientation = 0, velocity = 0, steering = 0, acceleration = 0, length = 0;

 public:
  Vehicle() {}
  void setLength(double l) { length = l; }
  double getLength() const { return length; }
  double getX() const { return x; }
  double getY() const { return y; }
  double getOrientation() const { return orientation; }
  double getVelocity() const { return velocity; }
  double getSteering() const { return steering; }
  double getAcceleration() const { return acceleration; }
  void update(double x_, double y_, double psi, double v, double steer, double acc) {   /* Vehicle.cpp:25-32 */
    x = x_; y = y_; orientation = psi; velocity = v; steering = steer; acceleration = acc;
  }
  static double lookup(const std::vector<double> &xs, const std::vector<double> &ys, double key, double maxv) {
    const double a = std::fabs(key);
    for (size_t i = 0; i < xs.size(); i++)
      if (a <= xs[i]) return std::fmin(ys[i < ys.size() ? i : ys.size() - 1], maxv);
    return std::fmin(ys.back(), maxv);
  }
  double computeSpeedTarget(double angle, double maxv) const {          /* Vehicle.cpp:34-48 */
    return lookup(Config::steers, Config::steerSpeeds, angle, maxv);
  }
  double computeYawChangeSpeedLimit(double yawChange, double maxv) const {   /* Vehicle.cpp:66-79 */
    return lookup(Config::yawChanges, Config::yawChangeSpeeds, yawChange, maxv);
  }
  double computeThrottle(double accel, double target, double max_accel, double max_decel) const {  /* :81-103 */
    const double keep = target / Config::maxSpeed;
    if (accel >= 0) return accel < 0.001 ? keep : std::fmin(1, keep + (1 - keep) * accel / max_accel);
    if (accel <= -15) return -1;
    const double base = accel < -10 ? 0.95 : (accel < -5 ? 0.9 : 0.85);
    return -base - (1 - base) * accel / max_decel;
  }
  void globalToVehicle(std::vector<double> &xs, std::vector<double> &ys) const {       /* Vehicle.cpp:105-114 */
    const double c = std::cos(orientation), s = std::sin(orientation);
    for (size_t i = 0; i < xs.size(); i++) {
      const double vx = xs[i] - x, vy = ys[i] - y;
      xs[i] = vx * c + vy * s; ys[i] = vy * c - vx * s;
    }
  }
  void globalToVehicle(double &px, double &py) {                                      /* :116-123 */
    const double c = std::cos(orientation), s = std::sin(orientation), vx = px - x, vy = py - y;
    px = vx * c + vy * s; py = vy * c - vx * s;
  }
  void vehicleToGlobal(std::vector<double> &xs, std::vector<double> &ys) const {       /* :125-134 */
    const double c = std::cos(orientation), s = std::sin(orientation);
    for (size_t i = 0; i < xs.size(); i++) {
      const double gx = x + xs[i] * c - ys[i] * s, gy = y + xs[i] * s + ys[i] * c;
      xs[i] = gx; ys[i] = gy;
    }
  }
  void vehicleToGlobal(double &px, double &py) const {                                 /* :136-143 */
    const double c = std::cos(orientation), s = std::sin(orientation);
    const double gx = x + px * c - py * s, gy = y + px * s + py * c;
    px = gx; py = gy;
  }
  /* bicycle step used for latency compensation (Vehicle.cpp:145-168); the new speed is not clamped
   * (the reference's clamp at :156 is overwritten at :167) */
  void move(double dt) {
    const double dist = velocity * dt, npsi = orientation + steering * dist / length;
    x += dist * std::cos(orientation); y += dist * std::sin(orientation);
    velocity += acceleration * dt; orientation = npsi;
  }
};

/* ---- RoadGeometry -------------------------------------------------------------------------- */
class RoadGeometry {                                                   /* RoadGeometry.h:11-101 */
  std::vector<double> x, y, polynomial;

 public:
  void setCenter(std::vector<double> &xs, std::vector<double> &ys, int maxFitOrder, double maxFitError) {
    x = xs; y = ys; fit(maxFitOrder, maxFitError);                      /* RoadGeometry.cpp:4-8 */
  }
  const std::vector<double> &getPolynomial() const { return polynomial; }
  void setPolynomial(const std::vector<double> &c) { polynomial = c; }
  double centerY(double px) { return polyeval(polynomial, px); }        /* :10-12 */
  void fit(int maxFitOrder, double maxFitError) {                       /* :18-34: order 2 upward */
    double err; int order = 2;
    do {
      polynomial = polyfit(x, y, order++);
      err = 0.0;
      for (size_t i = 0; i < x.size(); i++) err += square<double>(y[i] - centerY(x[i]));
    } while (err > maxFitError && order < maxFitOrder);
  }
  double orientation(double px, double dir) {                           /* :41-47 */
    double psi = std::atan(polyder(polynomial, px));
    if (dir < 0) psi = normalizeAngle(psi + M_PI);
    return psi;
  }
  double computeOrientationChange(double x0, double x1) { return orientation(x1, x1 - x0) - orientation(x0, x1 - x0); }
  double cte(double px, double py) { return centerY(px) - py; }         /* :63-65 */
};

/* ---- MPC ------------------------------------------------------------------------------------ */
class MPC {                                                            /* MPC.h:12-56 */
  MpcHandle *handle = nullptr;
  int64_t capacity = 0;
  int handleN = 0, handleFp32Start = 0;
  Vehicle vehicle;
  RoadGeometry roadGeometry;

  void ensure(int64_t B) {
    const MpcParams p = Config::snapshot();
    if (handle && (B > capacity || p.N != handleN || p.f64_f32_start != handleFp32Start)) { mpc_destroy(handle); handle = nullptr; }
    if (!handle) {
      capacity = B < 1 ? 1 : B; handleN = p.N; handleFp32Start = p.f64_f32_start;
      int rc = mpc_create(&p, -1, capacity, &handle);
      if (rc != MPC_OK) { handle = nullptr; throw std::string("mpc_create failed: ") + mpc_last_error(); }
    } else if (mpc_set_params(handle, &p) != MPC_OK) throw std::string("mpc_set_params failed: ") + mpc_last_error();
  }

 public:
  MPC() {}
  MPC(const MPC &) = delete;
  MPC &operator=(const MPC &) = delete;
  virtual ~MPC() { if (handle) mpc_destroy(handle); }

  RoadGeometry &road() { return roadGeometry; }

  /* MPC::solve, MPC.h:42-43 / MPC.cpp:183-325.  `state` = {x,y,psi,v,cte,epsi}; target_velocity and dir
   * are accepted and ignored exactly as FG_eval ignores them.  Returns
   * {x1,y1,psi1,v1,cte1,epsi1,delta0,a0,cost}; trajectories are APPENDED (push_back), N points each. */
  template <class Vec>
  std::vector<double> solve(Vec &state, double target_velocity, std::vector<double> *x_trajectory = NULL,
                            std::vector<double> *y_trajectory = NULL, double dir = 1) {
    (void)target_velocity; (void)dir;
    ensure(1);
    const int N = (int)Config::N;
    double st[6], cf[MPC_NCOEF] = {0, 0, 0, 0, 0}, out[9];
    for (int i = 0; i < 6; i++) st[i] = state[i];
    const std::vector<double> &poly = roadGeometry.getPolynomial();
    for (size_t i = 0; i < poly.size() && i < MPC_NCOEF; i++) cf[i] = poly[i];
    std::vector<double> traj(2 * N);
    int32_t status = 0, iters = 0;
    const double ylo = Config::yawLow, yhi = Config::yawHigh;
    int rc = mpc_solve_batch_host(handle, 1, 1, st, cf, &ylo, &yhi, NULL, out, traj.data(), &status, &iters);
    if (rc != MPC_OK) throw std::string("mpc_solve_batch_host failed: ") + mpc_last_error();
    if (status != MPC_STATUS_SUCCESS) {
#ifdef EXIT_ON_IPOPT_FAILURE
      throw std::string("Ipopt failed with ") + std::to_string(status);
#else
      std::cout << "Ipopt failed with " + std::to_string(status) << std::endl;
#endif
    }
    if (x_trajectory) for (int i = 0; i < N; i++) { x_trajectory->push_back(traj[i]); y_trajectory->push_back(traj[N + i]); }
    return std::vector<double>(out, out + 9);
  }

  /* MPC::run, MPC.h:54-55 / MPC.cpp:327-382: the whole of it is the library's mpc_run_batch_host with B = 1 (frame transform,
   * fit, bounds, solve, post-processing: the N1 kernels).  What the reference's run() leaves behind is kept: ptsx/ptsy transformed
   * IN PLACE to the vehicle frame (mpc_main.cpp:189-190 relies on it), the fitted polynomial in roadGeometry, Config::yawLow /
   * yawHigh (MPC.cpp:345-352) and the copy of the vehicle.  Returns {x1,y1,psi1,v1,steer in [-1,1],accel,cte1,epsi1}. */
  std::vector<double> run(Vehicle &veh, std::vector<double> &ptsx, std::vector<double> &ptsy,
                          std::vector<double> *x_trajectory = NULL, std::vector<double> *y_trajectory = NULL) {
    ensure(1);
    const int N = (int)Config::N, npts = (int)ptsx.size();
    const double pose[6] = {veh.getX(), veh.getY(), veh.getOrientation(), veh.getVelocity(), veh.getSteering(), veh.getAcceleration()};
    double out8[8], pre[15];
    std::vector<double> traj(2 * N);
    int32_t status = 0, iters = 0;
    const int rc = mpc_run_batch_host(handle, 1, 1, npts, pose, ptsx.data(), ptsy.data(), out8, traj.data(), &status, &iters, pre);
    if (rc != MPC_OK) throw std::string("mpc_run_batch_host failed: ") + mpc_last_error();
    vehicle = veh;
    int nc = MPC_NCOEF;
    while (nc > 3 && pre[6 + nc - 1] == 0.0) --nc;                      /* the fit's order (the rows are zero padded) */
    roadGeometry.setPolynomial(std::vector<double>(pre + 6, pre + 6 + nc));
    Config::yawLow = pre[11]; Config::yawHigh = pre[12];
    if (status != MPC_STATUS_SUCCESS) {
#ifdef EXIT_ON_IPOPT_FAILURE
      throw std::string("Ipopt failed with ") + std::to_string(status);
#else
      std::cout << "Ipopt failed with " + std::to_string(status) << std::endl;
#endif
    }
    if (x_trajectory) for (int i = 0; i < N; i++) { x_trajectory->push_back(traj[i]); y_trajectory->push_back(traj[N + i]); }
    return std::vector<double>(out8, out8 + 8);
  }

  /* New: B independent instances in one call (host pointers, struct-of-arrays, see mpc_amd.h). */
  int solveBatch(int64_t B, const double *state6xB, const double *coeffs5xB, const double *yaw_lo, const double *yaw_hi,
                 double *out9xB, double *traj2NxB, int32_t *status) {
    ensure(B);
    return mpc_solve_batch_host(handle, B, B, state6xB, coeffs5xB, yaw_lo, yaw_hi, NULL, out9xB, traj2NxB, status, NULL);
  }
};

#endif /* MPC_DROP_IN_HPP */
