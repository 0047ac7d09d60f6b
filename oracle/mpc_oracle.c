/*
 * mpc_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See mpc_oracle.h for scope and the "parity unpinned" statement.
 *
 * Every function cites the reference lines it restates (paths relative to
 * /root/reference).  Nothing here is copied from the reference: the
 * reference's NLP is written on CppAD::AD<double> and solved by IPOPT; this
 * file states the same NLP on plain doubles with hand-derived derivatives and
 * solves it with a dense restatement of IPOPT's published algorithm.
 */
#include "mpc_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_NMAX 128          /* max horizon N handled by the oracle */
#define ORC_INF_BOUND 1.0e19  /* MPC.cpp:223-224: IPOPT treats |b| >= 1e19 as infinite */

/* ------------------------------------------------------------------------- */
/* utils.h                                                                    */
/* ------------------------------------------------------------------------- */

/* src/utils/utils.h:11-13 */
double orc_mph2mps(double mph) { return mph * 1609.34 / 3600.0; }
/* src/utils/utils.h:69 */
static double deg2rad(double x) { return x * M_PI / 180; }
/* src/utils/utils.h:56-58 */
static double clampd(double a, double lo, double hi) { return a < lo ? lo : (a > hi ? hi : a); }

/* src/utils/utils.h:28-34: Horner from the highest coefficient down */
double orc_polyeval(const double *c, int nc, double x) {
  double r = 0;
  for (int i = nc - 1; i >= 0; i--) r = r * x + c[i];
  return r;
}

/* src/utils/utils.h:41-47 */
double orc_polyder(const double *c, int nc, double x) {
  double r = 0;
  for (int i = nc - 1; i >= 1; i--) r = r * x + i * c[i];
  return r;
}

/* second and third derivatives: not in the reference (CppAD differentiates the
 * tape); needed for the analytic Jacobian/Hessian of the epsi and cte rows */
static double polyder2(const double *c, int nc, double x) {
  double r = 0;
  for (int i = nc - 1; i >= 2; i--) r = r * x + (double)(i * (i - 1)) * c[i];
  return r;
}
static double polyder3(const double *c, int nc, double x) {
  double r = 0;
  for (int i = nc - 1; i >= 3; i--) r = r * x + (double)(i * (i - 1) * (i - 2)) * c[i];
  return r;
}

/* src/utils/utils.h:87-92 */
double orc_normalize_angle(double a) {
  double r = a;
  while (r >= M_PI) r -= 2. * M_PI;
  while (r < -M_PI) r += 2. * M_PI;
  return r;
}

/* ------------------------------------------------------------------------- */
/* dense least squares by Householder QR (stands in for Eigen's              */
/* householderQr().solve(), src/utils/utils.cpp:24-26)                        */
/* ------------------------------------------------------------------------- */
/* A is rows x cols row-major (overwritten), b has `rows` entries (overwritten);
 * x receives `cols` entries.  Returns 0, or 1 if rank deficient. */
static int lstsq_qr(double *A, int rows, int cols, double *b, double *x) {
  int rank_def = 0;
  for (int k = 0; k < cols; k++) {
    double nrm = 0;
    for (int i = k; i < rows; i++) nrm += A[i * cols + k] * A[i * cols + k];
    nrm = sqrt(nrm);
    if (nrm == 0) { rank_def = 1; continue; }
    double alpha = A[k * cols + k] > 0 ? -nrm : nrm;
    /* v = a_k - alpha e_k, stored in column k below the diagonal */
    double vk = A[k * cols + k] - alpha;
    double vnorm2 = vk * vk;
    for (int i = k + 1; i < rows; i++) vnorm2 += A[i * cols + k] * A[i * cols + k];
    if (vnorm2 == 0) { A[k * cols + k] = alpha; continue; }
    for (int j = k + 1; j < cols; j++) {
      double s = vk * A[k * cols + j];
      for (int i = k + 1; i < rows; i++) s += A[i * cols + k] * A[i * cols + j];
      s = 2 * s / vnorm2;
      A[k * cols + j] -= s * vk;
      for (int i = k + 1; i < rows; i++) A[i * cols + j] -= s * A[i * cols + k];
    }
    {
      double s = vk * b[k];
      for (int i = k + 1; i < rows; i++) s += A[i * cols + k] * b[i];
      s = 2 * s / vnorm2;
      b[k] -= s * vk;
      for (int i = k + 1; i < rows; i++) b[i] -= s * A[i * cols + k];
    }
    A[k * cols + k] = alpha;
  }
  for (int k = cols - 1; k >= 0; k--) {
    double s = b[k];
    for (int j = k + 1; j < cols; j++) s -= A[k * cols + j] * x[j];
    double d = A[k * cols + k];
    if (fabs(d) < 1e-300) { x[k] = 0; rank_def = 1; } else x[k] = s / d;
  }
  return rank_def;
}

/* src/utils/utils.cpp:10-29 */
int orc_polyfit(const double *xv, const double *yv, int n, int order, double *coef) {
  if (!(order >= 1 && order <= n - 1)) return -1; /* utils.cpp:13 assert */
  int cols = order + 1;
  double *A = (double *)malloc(sizeof(double) * n * cols);
  double *b = (double *)malloc(sizeof(double) * n);
  for (int j = 0; j < n; j++) {
    A[j * cols + 0] = 1.0;                                             /* utils.cpp:16-18 */
    for (int i = 0; i < order; i++) A[j * cols + i + 1] = A[j * cols + i] * xv[j]; /* :20-24 */
    b[j] = yv[j];
  }
  int rc = lstsq_qr(A, n, cols, b, coef);
  free(A); free(b);
  return rc;
}

/* src/model/RoadGeometry.cpp:18-34 */
int orc_road_fit(const double *x, const double *y, int n, int max_fit_order,
                 double max_fit_error, double *coef, double *fiterr_out) {
  double fiterr;
  int order = 2, nc;
  do {
    int used = order++;
    orc_polyfit(x, y, n, used, coef);
    nc = used + 1;
    fiterr = 0.0;
    for (int i = 0; i < n; i++) {
      double d = y[i] - orc_polyeval(coef, nc, x[i]);
      fiterr += d * d;
    }
  } while (fiterr > max_fit_error && order < max_fit_order);
  if (fiterr_out) *fiterr_out = fiterr;
  return nc;
}

/* src/model/RoadGeometry.cpp:41-47 */
double orc_orientation(const double *c, int nc, double px, double dir) {
  double psi = atan(orc_polyder(c, nc, px));
  if (dir < 0) psi = orc_normalize_angle(psi + M_PI);
  return psi;
}

/* src/model/RoadGeometry.cpp:57-61 */
double orc_orientation_change(const double *c, int nc, double x0, double x1) {
  double psi0 = orc_orientation(c, nc, x0, x1 - x0);
  double psi1 = orc_orientation(c, nc, x1, x1 - x0);
  return psi1 - psi0;
}

/* ------------------------------------------------------------------------- */
/* Vehicle                                                                    */
/* ------------------------------------------------------------------------- */

/* src/model/Vehicle.cpp:34-48 (double) and :50-64 (AD, same values) */
double orc_speed_target(const OrcConfig *cfg, double angle, double max) {
  double y = fabs(angle);
  int last = cfg->n_steer_speeds - 1;
  for (int i = 0; i < cfg->n_steers; i++) {
    if (y <= cfg->steers[i]) {
      if (cfg->n_steer_speeds > i) return fmin(cfg->steer_speeds[i], max);
      return fmin(cfg->steer_speeds[last], max);
    }
  }
  return fmin(cfg->steer_speeds[last], max);
}

/* src/model/Vehicle.cpp:66-79 */
double orc_yaw_change_speed_limit(const OrcConfig *cfg, double yaw_change, double max) {
  double y = fabs(yaw_change);
  int last = cfg->n_yaw_change_speeds - 1;
  for (int i = 0; i < cfg->n_yaw_changes; i++) {
    if (y <= cfg->yaw_changes[i]) {
      if (cfg->n_yaw_change_speeds > i) return fmin(cfg->yaw_change_speeds[i], max);
      return fmin(cfg->yaw_change_speeds[last], max);
    }
  }
  return fmin(cfg->yaw_change_speeds[last], max);
}

/* src/model/Vehicle.cpp:81-103 */
double orc_compute_throttle(const OrcConfig *cfg, double accel, double target,
                            double max_accel, double max_decel) {
  double keep = target / cfg->max_speed;
  if (accel >= 0) {
    if (accel < 0.001) return keep;
    return fmin(1, keep + (1 - keep) * accel / max_accel);
  }
  if (accel <= -15) return -1;
  if (accel < -10) return -0.95 - (1 - 0.95) * accel / max_decel;
  if (accel < -5) return -0.9 - (1 - 0.9) * accel / max_decel;
  return -0.85 - (1 - 0.85) * accel / max_decel;
}

/* src/model/Vehicle.cpp:105-114 */
void orc_global_to_vehicle(double px, double py, double psi, double *xs, double *ys, int n) {
  double cosine = cos(psi), sine = sin(psi);
  for (int i = 0; i < n; i++) {
    double vx = xs[i] - px, vy = ys[i] - py;
    xs[i] = vx * cosine + vy * sine;
    ys[i] = vy * cosine - vx * sine;
  }
}

/* src/model/Vehicle.cpp:145-168.  pose = {x,y,psi,v,steering,acceleration};
 * `length` is what mpc_main.cpp:155 sets: Config::Lf.  The clamp at :156 is
 * overwritten at :167, so the new velocity is NOT clamped (restated as is). */
void orc_vehicle_move(const OrcConfig *cfg, double *pose, double dt) {
  double dist = pose[3] * dt;
  double delta_psi = pose[4] * dist / cfg->Lf;
  double new_orientation = pose[2] + delta_psi;
  double px = pose[0] + dist * cos(pose[2]);
  double py = pose[1] + dist * sin(pose[2]);
  double v = pose[3] + pose[5] * dt;
  pose[0] = px; pose[1] = py; pose[2] = new_orientation; pose[3] = v;
}

/* ------------------------------------------------------------------------- */
/* Config                                                                     */
/* ------------------------------------------------------------------------- */

/* src/utils/Config.cpp:5-29 */
void orc_config_defaults(OrcConfig *c) {
  memset(c, 0, sizeof(*c));
  c->N = 25; c->max_fit_order = 4; c->max_fit_error = 0.5; c->latency = 100; c->lookahead = 0;
  c->ipopt_timeout = 0.5; c->dt = 0.025; c->max_steering = deg2rad(25.0);
  c->max_acceleration = orc_mph2mps(8); c->max_deceleration = orc_mph2mps(-20);
  c->max_speed = orc_mph2mps(100); c->Lf = 2.67; c->epsi_panic = 1; c->cte_panic = 0.6;
  c->steer_adj_thresh = 0.6; c->steer_adj_ratio = 0.025;
  double w[] = {100, 100, 1, 1, 1, 5000, 1, 1000};
  c->n_weights = 8; memcpy(c->weights, w, sizeof(w));
  double st[] = {0.1, 0.2, 0.3};
  c->n_steers = 3; memcpy(c->steers, st, sizeof(st));
  double ss[] = {orc_mph2mps(80), orc_mph2mps(65), orc_mph2mps(30), orc_mph2mps(25)};
  c->n_steer_speeds = 4; memcpy(c->steer_speeds, ss, sizeof(ss));
}

/* minimal reader for the flat config-*.json files: number or array of numbers */
static const char *json_find(const char *txt, const char *key) {
  char pat[128];
  snprintf(pat, sizeof(pat), "\"%s\"", key);
  const char *p = strstr(txt, pat);
  if (!p) return NULL;
  p += strlen(pat);
  while (*p && *p != ':') p++;
  return *p ? p + 1 : NULL;
}
static int json_num(const char *txt, const char *key, double *out) {
  const char *p = json_find(txt, key);
  if (!p) return -1;
  char *end; double v = strtod(p, &end);
  if (end == p) return -1;
  *out = v; return 0;
}
static int json_arr(const char *txt, const char *key, double *out, int cap) {
  const char *p = json_find(txt, key);
  if (!p) return -1;
  while (*p && *p != '[') p++;
  if (!*p) return -1;
  p++;
  int n = 0;
  for (;;) {
    while (*p == ' ' || *p == ',' || *p == '\n' || *p == '\t' || *p == '\r') p++;
    if (*p == ']' || !*p) break;
    char *end; double v = strtod(p, &end);
    if (end == p) return -1;
    if (n < cap) out[n] = v;
    n++; p = end;
  }
  return n;
}

/* src/utils/Config.cpp:31-87 */
int orc_config_load(const char *path, OrcConfig *c) {
  FILE *f = fopen(path, "rb");
  if (!f) return -1;
  char *txt = (char *)malloc(1 << 16);
  size_t len = fread(txt, 1, (1 << 16) - 1, f);
  fclose(f); txt[len] = 0;
  orc_config_defaults(c);
  double v = 0.0; int rc = 0;
  rc |= json_num(txt, "N", &v); c->N = (int)v;
  rc |= json_num(txt, "dt", &c->dt);
  rc |= json_num(txt, "max acceleration", &v); c->max_acceleration = orc_mph2mps(v);
  rc |= json_num(txt, "max deceleration", &v); c->max_deceleration = orc_mph2mps(v);
  rc |= json_num(txt, "max steering", &v); c->max_steering = deg2rad(v);
  rc |= json_num(txt, "max speed", &v); c->max_speed = orc_mph2mps(v);
  double speed_scale = c->max_speed / orc_mph2mps(100.0);             /* Config.cpp:48 */
  rc |= json_num(txt, "latency", &v); c->latency = (long)v;
  c->lookahead = c->latency * 1.0E-3;                                  /* :50 */
  rc |= json_num(txt, "max polynomial fitting order", &v); c->max_fit_order = (int)v;
  rc |= json_num(txt, "max polynomial fitting error", &c->max_fit_error);
  rc |= json_num(txt, "ipopt timeout", &c->ipopt_timeout);
  rc |= json_num(txt, "Lf", &c->Lf);
  rc |= json_num(txt, "epsi panic", &c->epsi_panic);
  rc |= json_num(txt, "cte panic", &c->cte_panic);
  rc |= json_num(txt, "steer adjustment threshold", &c->steer_adj_thresh);
  rc |= json_num(txt, "steer adjustment ratio", &v);
  c->steer_adj_ratio = clampd(v, 0.0, 0.1);                            /* :59 */
  c->n_weights = json_arr(txt, "weights", c->weights, ORC_MAX_TABLE);
  if (c->n_weights <= 11) rc = -1;                                     /* :61 assert */
  c->n_steers = json_arr(txt, "steers", c->steers, ORC_MAX_TABLE);
  c->n_steer_speeds = json_arr(txt, "steer speeds", c->steer_speeds, ORC_MAX_TABLE);
  for (int i = 0; i < c->n_steer_speeds; i++) {                        /* :66-73 */
    if (speed_scale <= 1) c->steer_speeds[i] = fmin(orc_mph2mps(c->steer_speeds[i]), c->max_speed);
    else c->steer_speeds[i] = orc_mph2mps(c->steer_speeds[i]) * speed_scale;
  }
  c->n_yaw_changes = json_arr(txt, "yaw changes", c->yaw_changes, ORC_MAX_TABLE);
  c->n_yaw_change_speeds = json_arr(txt, "yaw change speeds", c->yaw_change_speeds, ORC_MAX_TABLE);
  for (int i = 0; i < c->n_yaw_change_speeds; i++) {                   /* :78-85 */
    if (speed_scale <= 1) c->yaw_change_speeds[i] = fmin(orc_mph2mps(c->yaw_change_speeds[i]), c->max_speed);
    else c->yaw_change_speeds[i] = orc_mph2mps(c->yaw_change_speeds[i]) * speed_scale;
  }
  if (c->n_steers < 0 || c->n_steer_speeds < 0 || c->n_yaw_changes < 0 || c->n_yaw_change_speeds < 0) rc = -1;
  free(txt);
  return rc ? -1 : 0;
}

/* ------------------------------------------------------------------------- */
/* FG_eval: objective + residuals, and their derivatives                      */
/* ------------------------------------------------------------------------- */

/* Weight indices, src/utils/Config.h:14-61 */
enum { W_CTE = 0, W_EPSI = 1, W_V = 2, W_DELTA = 3, W_DDELTA = 4, W_A = 6, W_DA = 7,
       W_DECEL_LOW_V = 8, W_NEG_V = 9, W_LARGE_EPSI = 10, W_LARGE_CTE = 11 };

/* The outcome of every data-dependent `if` in FG_eval::operator()
 * (MPC.cpp:72,79,87,89,98,103,111) at one point: the "tape". */
typedef struct Tape {
  double wcte[ORC_NMAX], wepsi[ORC_NMAX], vref[ORC_NMAX];
  char negv[ORC_NMAX], apos[ORC_NMAX], declow[ORC_NMAX], dapos[ORC_NMAX];
} Tape;

typedef struct Idx { int N, x, y, psi, v, cte, epsi, delta, a, n, m; } Idx;

/* MPC.cpp:56-63 / :189-202 */
static Idx make_idx(int N) {
  Idx I; I.N = N; I.x = 0; I.y = I.x + N; I.psi = I.y + N; I.v = I.psi + N; I.cte = I.v + N;
  I.epsi = I.cte + N; I.delta = I.epsi + N; I.a = I.delta + N - 1;
  I.n = 6 * N + 2 * (N - 1); I.m = 6 * N;
  return I;
}

static void tape_decide(const OrcConfig *cfg, const double *at, Tape *t) {
  Idx I = make_idx(cfg->N);
  const double *w = cfg->weights;
  int N = I.N;
  memset(t, 0, sizeof(*t));
  for (int i = 0; i < N; i++) {
    t->wcte[i] = fabs(at[I.cte + i]) < cfg->cte_panic ? w[W_CTE] : w[W_LARGE_CTE];        /* :72-77 */
    t->wepsi[i] = fabs(at[I.epsi + i]) > cfg->epsi_panic ? w[W_LARGE_EPSI] : w[W_EPSI];   /* :79-84 */
    t->vref[i] = orc_speed_target(cfg, at[I.psi + i], cfg->max_speed);                    /* :87 */
    t->negv[i] = at[I.v + i] < 0;                                                         /* :89 */
  }
  for (int i = 0; i < N - 1; i++) {
    t->apos[i] = at[I.a + i] > 0;                                                         /* :98 */
    t->declow[i] = at[I.a + i] < 0 && at[I.v + i] < at[I.a + i];                          /* :103 */
  }
  for (int i = 0; i < N - 2; i++) t->dapos[i] = at[I.a + i + 1] > at[I.a + i];            /* :111 */
}

static double sq(double a) { return a * a; } /* utils.h:81 */

/* fg[0], MPC.cpp:68-114, with the branch outcomes taken from the tape */
static double eval_f(const OrcConfig *cfg, const Tape *t, const double *z) {
  Idx I = make_idx(cfg->N);
  const double *w = cfg->weights;
  int N = I.N;
  double f = 0;
  for (int i = 0; i < N; i++) {
    f += sq(z[I.cte + i]) * t->wcte[i];
    f += sq(z[I.epsi + i]) * t->wepsi[i];
    f += sq(z[I.v + i] - t->vref[i]) * w[W_V];
    if (t->negv[i]) f += sq(z[I.v + i]) * w[W_NEG_V];
  }
  for (int i = 0; i < N - 1; i++) {
    f += sq(z[I.delta + i]) * w[W_DELTA];
    if (t->apos[i]) f += sq(z[I.a + i]) * w[W_A];
    if (t->declow[i]) f += sq(z[I.v + i] - z[I.a + i]) * w[W_DECEL_LOW_V];
  }
  for (int i = 0; i < N - 2; i++) {
    f += sq(z[I.delta + i + 1] - z[I.delta + i]) * w[W_DDELTA];
    if (t->dapos[i]) f += sq(z[I.a + i + 1] - z[I.a + i]) * w[W_DA];
  }
  return f;
}

static void eval_grad_f(const OrcConfig *cfg, const Tape *t, const double *z, double *g) {
  Idx I = make_idx(cfg->N);
  const double *w = cfg->weights;
  int N = I.N;
  memset(g, 0, sizeof(double) * I.n);
  for (int i = 0; i < N; i++) {
    g[I.cte + i] += 2 * t->wcte[i] * z[I.cte + i];
    g[I.epsi + i] += 2 * t->wepsi[i] * z[I.epsi + i];
    g[I.v + i] += 2 * w[W_V] * (z[I.v + i] - t->vref[i]);
    if (t->negv[i]) g[I.v + i] += 2 * w[W_NEG_V] * z[I.v + i];
  }
  for (int i = 0; i < N - 1; i++) {
    g[I.delta + i] += 2 * w[W_DELTA] * z[I.delta + i];
    if (t->apos[i]) g[I.a + i] += 2 * w[W_A] * z[I.a + i];
    if (t->declow[i]) {
      double d = z[I.v + i] - z[I.a + i];
      g[I.v + i] += 2 * w[W_DECEL_LOW_V] * d;
      g[I.a + i] -= 2 * w[W_DECEL_LOW_V] * d;
    }
  }
  for (int i = 0; i < N - 2; i++) {
    double dd = z[I.delta + i + 1] - z[I.delta + i];
    g[I.delta + i + 1] += 2 * w[W_DDELTA] * dd;
    g[I.delta + i] -= 2 * w[W_DDELTA] * dd;
    if (t->dapos[i]) {
      double da = z[I.a + i + 1] - z[I.a + i];
      g[I.a + i + 1] += 2 * w[W_DA] * da;
      g[I.a + i] -= 2 * w[W_DA] * da;
    }
  }
}

/* H += factor * hess(f); the objective is piecewise quadratic, so this is constant per tape */
static void add_hess_f(const OrcConfig *cfg, const Tape *t, double factor, double *H) {
  Idx I = make_idx(cfg->N);
  const double *w = cfg->weights;
  int N = I.N, n = I.n;
#define HH(r, c) H[(size_t)(r) * n + (c)]
  for (int i = 0; i < N; i++) {
    HH(I.cte + i, I.cte + i) += factor * 2 * t->wcte[i];
    HH(I.epsi + i, I.epsi + i) += factor * 2 * t->wepsi[i];
    HH(I.v + i, I.v + i) += factor * 2 * w[W_V];
    if (t->negv[i]) HH(I.v + i, I.v + i) += factor * 2 * w[W_NEG_V];
  }
  for (int i = 0; i < N - 1; i++) {
    HH(I.delta + i, I.delta + i) += factor * 2 * w[W_DELTA];
    if (t->apos[i]) HH(I.a + i, I.a + i) += factor * 2 * w[W_A];
    if (t->declow[i]) {
      double k = factor * 2 * w[W_DECEL_LOW_V];
      HH(I.v + i, I.v + i) += k; HH(I.a + i, I.a + i) += k;
      HH(I.v + i, I.a + i) -= k; HH(I.a + i, I.v + i) -= k;
    }
  }
  for (int i = 0; i < N - 2; i++) {
    double k = factor * 2 * w[W_DDELTA];
    int p = I.delta + i, q = I.delta + i + 1;
    HH(p, p) += k; HH(q, q) += k; HH(p, q) -= k; HH(q, p) -= k;
    if (t->dapos[i]) {
      k = factor * 2 * w[W_DA]; p = I.a + i; q = I.a + i + 1;
      HH(p, p) += k; HH(q, q) += k; HH(p, q) -= k; HH(q, p) -= k;
    }
  }
#undef HH
}

/* fg[1..6N], MPC.cpp:116-153, written to g[0..6N) */
static void eval_g(const OrcConfig *cfg, const double *coef, int nc, const double *z, double *g) {
  Idx I = make_idx(cfg->N);
  int N = I.N; double dt = cfg->dt, Lf = cfg->Lf;
  g[I.x] = z[I.x]; g[I.y] = z[I.y]; g[I.psi] = z[I.psi];            /* :116-121 */
  g[I.v] = z[I.v]; g[I.cte] = z[I.cte]; g[I.epsi] = z[I.epsi];
  for (int i = 1; i < N; i++) {
    double x1 = z[I.x + i], y1 = z[I.y + i], psi1 = z[I.psi + i], v1 = z[I.v + i];
    double cte1 = z[I.cte + i], epsi1 = z[I.epsi + i];
    double x0 = z[I.x + i - 1], y0 = z[I.y + i - 1], psi0 = z[I.psi + i - 1], v0 = z[I.v + i - 1];
    double epsi0 = z[I.epsi + i - 1], delta0 = z[I.delta + i - 1], a0 = z[I.a + i - 1];
    double vdt = v0 * dt;                                             /* :142 */
    double psi = psi0 + delta0 * vdt / Lf;                            /* :144 */
    g[I.x + i] = x1 - (x0 + cos(psi0) * vdt);                         /* :145 */
    g[I.y + i] = y1 - (y0 + sin(psi0) * vdt);                         /* :146 */
    g[I.psi + i] = psi1 - psi;                                        /* :147 */
    double v = v0 + a0 * dt;                                          /* :148 */
    g[I.v + i] = v1 - v;                                              /* :149 */
    g[I.cte + i] = cte1 - ((orc_polyeval(coef, nc, x0) - y0) + sin(epsi0) * vdt);   /* :151 */
    g[I.epsi + i] = epsi1 - (psi - atan(orc_polyder(coef, nc, x0)));  /* :152, dir = 1.0 */
  }
}

/* dense Jacobian of g, m x n row-major */
static void eval_jac_g(const OrcConfig *cfg, const double *coef, int nc, const double *z, double *J) {
  Idx I = make_idx(cfg->N);
  int N = I.N, n = I.n; double dt = cfg->dt, Lf = cfg->Lf;
  memset(J, 0, sizeof(double) * (size_t)I.m * n);
#define JJ(r, c) J[(size_t)(r) * n + (c)]
  JJ(I.x, I.x) = 1; JJ(I.y, I.y) = 1; JJ(I.psi, I.psi) = 1;
  JJ(I.v, I.v) = 1; JJ(I.cte, I.cte) = 1; JJ(I.epsi, I.epsi) = 1;
  for (int i = 1; i < N; i++) {
    int p = i - 1;
    double x0 = z[I.x + p], psi0 = z[I.psi + p], v0 = z[I.v + p], e0 = z[I.epsi + p], d0 = z[I.delta + p];
    double s = sin(psi0), c = cos(psi0), vdt = v0 * dt;
    double fp = orc_polyder(coef, nc, x0), fpp = polyder2(coef, nc, x0);
    JJ(I.x + i, I.x + i) = 1; JJ(I.x + i, I.x + p) = -1; JJ(I.x + i, I.psi + p) = s * vdt; JJ(I.x + i, I.v + p) = -c * dt;
    JJ(I.y + i, I.y + i) = 1; JJ(I.y + i, I.y + p) = -1; JJ(I.y + i, I.psi + p) = -c * vdt; JJ(I.y + i, I.v + p) = -s * dt;
    JJ(I.psi + i, I.psi + i) = 1; JJ(I.psi + i, I.psi + p) = -1;
    JJ(I.psi + i, I.delta + p) = -vdt / Lf; JJ(I.psi + i, I.v + p) = -d0 * dt / Lf;
    JJ(I.v + i, I.v + i) = 1; JJ(I.v + i, I.v + p) = -1; JJ(I.v + i, I.a + p) = -dt;
    JJ(I.cte + i, I.cte + i) = 1; JJ(I.cte + i, I.x + p) = -fp; JJ(I.cte + i, I.y + p) = 1;
    JJ(I.cte + i, I.epsi + p) = -cos(e0) * vdt; JJ(I.cte + i, I.v + p) = -sin(e0) * dt;
    JJ(I.epsi + i, I.epsi + i) = 1; JJ(I.epsi + i, I.psi + p) = -1;
    JJ(I.epsi + i, I.delta + p) = -vdt / Lf; JJ(I.epsi + i, I.v + p) = -d0 * dt / Lf;
    JJ(I.epsi + i, I.x + p) = fpp / (1 + fp * fp);
  }
#undef JJ
}

/* H += sum_j lam[j] * hess(g_j) */
static void add_hess_g(const OrcConfig *cfg, const double *coef, int nc, const double *z,
                       const double *lam, double *H) {
  Idx I = make_idx(cfg->N);
  int N = I.N, n = I.n; double dt = cfg->dt, Lf = cfg->Lf;
#define HS(r, c, val) do { double v_ = (val); H[(size_t)(r) * n + (c)] += v_; if ((r) != (c)) H[(size_t)(c) * n + (r)] += v_; } while (0)
  for (int i = 1; i < N; i++) {
    int p = i - 1;
    double x0 = z[I.x + p], psi0 = z[I.psi + p], v0 = z[I.v + p], e0 = z[I.epsi + p];
    double s = sin(psi0), c = cos(psi0), vdt = v0 * dt;
    double fp = orc_polyder(coef, nc, x0), fpp = polyder2(coef, nc, x0), fppp = polyder3(coef, nc, x0);
    double lx = lam[I.x + i], ly = lam[I.y + i], lp = lam[I.psi + i], lc = lam[I.cte + i], le = lam[I.epsi + i];
    /* x row: -(cos(psi0) v0 dt) */
    HS(I.psi + p, I.psi + p, lx * c * vdt); HS(I.psi + p, I.v + p, lx * s * dt);
    /* y row: -(sin(psi0) v0 dt) */
    HS(I.psi + p, I.psi + p, ly * s * vdt); HS(I.psi + p, I.v + p, -ly * c * dt);
    /* psi row: -(delta0 v0 dt / Lf) */
    HS(I.delta + p, I.v + p, -lp * dt / Lf);
    /* cte row: -(f(x0) + sin(epsi0) v0 dt) */
    HS(I.x + p, I.x + p, -lc * fpp); HS(I.epsi + p, I.epsi + p, lc * sin(e0) * vdt);
    HS(I.epsi + p, I.v + p, -lc * cos(e0) * dt);
    /* epsi row: -(delta0 v0 dt / Lf) + atan(f'(x0)) */
    HS(I.delta + p, I.v + p, -le * dt / Lf);
    double q = 1 + fp * fp;
    HS(I.x + p, I.x + p, le * (fppp * q - 2 * fp * fpp * fpp) / (q * q));
  }
#undef HS
}

void orc_fg_eval(const OrcConfig *cfg, const double *coef, int nc, int branch_mode,
                 const double *xi, const double *vars, double *fg) {
  Tape *t = (Tape *)malloc(sizeof(Tape));
  tape_decide(cfg, (branch_mode == ORC_BRANCH_LIVE || !xi) ? vars : xi, t);
  fg[0] = eval_f(cfg, t, vars);
  eval_g(cfg, coef, nc, vars, fg + 1);
  free(t);
}

void orc_fg_grad(const OrcConfig *cfg, const double *coef, int nc, int branch_mode,
                 const double *xi, const double *vars, double *grad_f, double *jac) {
  Tape *t = (Tape *)malloc(sizeof(Tape));
  tape_decide(cfg, (branch_mode == ORC_BRANCH_LIVE || !xi) ? vars : xi, t);
  eval_grad_f(cfg, t, vars, grad_f);
  eval_jac_g(cfg, coef, nc, vars, jac);
  free(t);
}

void orc_lag_hess(const OrcConfig *cfg, const double *coef, int nc, int branch_mode,
                  const double *xi, const double *vars, double obj_factor,
                  const double *lam, double *hess) {
  Idx I = make_idx(cfg->N);
  Tape *t = (Tape *)malloc(sizeof(Tape));
  tape_decide(cfg, (branch_mode == ORC_BRANCH_LIVE || !xi) ? vars : xi, t);
  memset(hess, 0, sizeof(double) * (size_t)I.n * I.n);
  add_hess_f(cfg, t, obj_factor, hess);
  add_hess_g(cfg, coef, nc, vars, lam, hess);
  free(t);
}

/* ------------------------------------------------------------------------- */
/* dense symmetric indefinite factorisation (Bunch-Kaufman pivoting) with     */
/* inertia: stands in for MUMPS (MPC.cpp:175) on the small dense KKT matrix   */
/* ------------------------------------------------------------------------- */
typedef struct Ldl { int n; double *M; int *perm; char *blk; int npos, nneg, nzero; } Ldl;

static void ldl_swap(Ldl *F, int p, int q) {
  if (p == q) return;
  int n = F->n; double *M = F->M;
  for (int c = 0; c < n; c++) { double t = M[(size_t)p * n + c]; M[(size_t)p * n + c] = M[(size_t)q * n + c]; M[(size_t)q * n + c] = t; }
  for (int r = 0; r < n; r++) { double t = M[(size_t)r * n + p]; M[(size_t)r * n + p] = M[(size_t)r * n + q]; M[(size_t)r * n + q] = t; }
  int t = F->perm[p]; F->perm[p] = F->perm[q]; F->perm[q] = t;
}

/* factor P A P^T = L D L^T in place (A full symmetric n x n, row-major, destroyed) */
static void ldl_factor(Ldl *F) {
  const double alpha = (1.0 + sqrt(17.0)) / 8.0;
  int n = F->n; double *M = F->M;
#define A_(r, c) M[(size_t)(r) * n + (c)]
  for (int i = 0; i < n; i++) { F->perm[i] = i; F->blk[i] = 0; }
  F->npos = F->nneg = F->nzero = 0;
  int k = 0;
  while (k < n) {
    int kstep = 1, kp = k;
    double absakk = fabs(A_(k, k));
    int imax = -1; double colmax = 0;
    for (int i = k + 1; i < n; i++) if (fabs(A_(i, k)) > colmax) { colmax = fabs(A_(i, k)); imax = i; }
    if (fmax(absakk, colmax) == 0.0) {
      F->blk[k] = 1; F->nzero++; k++; continue;
    }
    if (absakk >= alpha * colmax) {
      kp = k;
    } else {
      double rowmax = 0;
      for (int j = k; j < n; j++) if (j != imax && fabs(A_(imax, j)) > rowmax) rowmax = fabs(A_(imax, j));
      if (absakk >= alpha * colmax * (colmax / rowmax)) kp = k;
      else if (fabs(A_(imax, imax)) >= alpha * rowmax) kp = imax;
      else { kp = imax; kstep = 2; }
    }
    int kk = k + kstep - 1;
    ldl_swap(F, kk, kp);
    if (kstep == 1) {
      double d = A_(k, k);
      if (d > 0) F->npos++; else if (d < 0) F->nneg++; else F->nzero++;
      F->blk[k] = 1;
      if (d != 0) {
        double r1 = 1.0 / d;
        for (int j = k + 1; j < n; j++) {
          double ljk = A_(j, k) * r1;
          if (ljk != 0) for (int i = k + 1; i < n; i++) A_(i, j) -= A_(i, k) * ljk;
        }
        for (int i = k + 1; i < n; i++) A_(i, k) *= r1;
      }
    } else {
      double a = A_(k, k), b = A_(k + 1, k), c = A_(k + 1, k + 1);
      double det = a * c - b * b;
      if (det < 0) { F->npos++; F->nneg++; }
      else if (det > 0) { if (a + c > 0) F->npos += 2; else F->nneg += 2; }
      else F->nzero += 2;
      F->blk[k] = 2; F->blk[k + 1] = 0;
      /* W = [a_k a_{k+1}] D^{-1}; trailing -= W [a_k a_{k+1}]^T */
      double i11 = c / det, i12 = -b / det, i22 = a / det;
      for (int j = k + 2; j < n; j++) {
        double w1 = A_(j, k) * i11 + A_(j, k + 1) * i12;
        double w2 = A_(j, k) * i12 + A_(j, k + 1) * i22;
        for (int i = k + 2; i < n; i++) A_(i, j) -= A_(i, k) * w1 + A_(i, k + 1) * w2;
      }
      for (int j = k + 2; j < n; j++) {
        double w1 = A_(j, k) * i11 + A_(j, k + 1) * i12;
        double w2 = A_(j, k) * i12 + A_(j, k + 1) * i22;
        A_(j, k) = w1; A_(j, k + 1) = w2;
      }
    }
    k += kstep;
  }
#undef A_
}

static void ldl_solve(const Ldl *F, const double *b, double *x) {
  int n = F->n; const double *M = F->M;
  double *y = (double *)malloc(sizeof(double) * n);
  for (int i = 0; i < n; i++) y[i] = b[F->perm[i]];
#define A_(r, c) M[(size_t)(r) * n + (c)]
  for (int k = 0; k < n;) {           /* L y = Pb */
    int s = F->blk[k] == 2 ? 2 : 1;
    for (int i = k + s; i < n; i++) {
      y[i] -= A_(i, k) * y[k];
      if (s == 2) y[i] -= A_(i, k + 1) * y[k + 1];
    }
    k += s;
  }
  for (int k = 0; k < n;) {           /* D */
    if (F->blk[k] == 2) {
      double a = A_(k, k), b2 = A_(k + 1, k), c = A_(k + 1, k + 1), det = a * c - b2 * b2;
      double y0 = y[k], y1 = y[k + 1];
      y[k] = (c * y0 - b2 * y1) / det; y[k + 1] = (-b2 * y0 + a * y1) / det;
      k += 2;
    } else { double d = A_(k, k); y[k] = d != 0 ? y[k] / d : 0; k += 1; }
  }
  /* L^T x = y: walk pivot blocks backwards */
  int *starts = (int *)malloc(sizeof(int) * n); int nb = 0;
  for (int k = 0; k < n;) { starts[nb++] = k; k += F->blk[k] == 2 ? 2 : 1; }
  for (int bi = nb - 1; bi >= 0; bi--) {
    int k = starts[bi], s = F->blk[k] == 2 ? 2 : 1;
    for (int i = k + s; i < n; i++) {
      y[k] -= A_(i, k) * y[i];
      if (s == 2) y[k + 1] -= A_(i, k + 1) * y[i];
    }
  }
#undef A_
  for (int i = 0; i < n; i++) x[F->perm[i]] = y[i];
  free(starts); free(y);
}

/* ------------------------------------------------------------------------- */
/* IPOPT-style primal-dual interior point on the dense problem               */
/*   min f(x)  s.t.  g(x) = gb,  xl <= x <= xu                                */
/* Restates Waechter & Biegler (2006) with IPOPT 3.12 default constants.      */
/* ------------------------------------------------------------------------- */
typedef struct Nlp {
  const OrcConfig *cfg; const double *coef; int nc; Idx I; Tape tape; int branch_mode;
  double *xl, *xu, *gb; double df;
} Nlp;

void orc_default_options(OrcSolveOptions *o) {
  o->branch_mode = ORC_BRANCH_FROZEN; o->max_iter = 500; o->tol = 1e-8;
  o->lam_init_ls = 1; o->obj_scaling = 1; o->verbose = 0;
  o->polish = 1; o->out_step_tol = 3e-7;
  o->bound_relax_factor = 1e-8; o->honor_original_bounds = 1;
  o->dual_inf_tol = 1.0; o->constr_viol_tol = 1e-4; o->compl_inf_tol = 1e-4;
  o->acceptable_tol = 1e-6; o->acceptable_dual_inf_tol = 1e10; o->acceptable_constr_viol_tol = 1e-2;
  o->acceptable_compl_inf_tol = 1e-2; o->acceptable_iter = 15; o->max_soc = 0;
}

typedef struct Filter { double th[256], ph[256]; int n; } Filter;

/* BacktrackingLineSearch::StoreAcceptablePoint / RestoreAcceptablePoint: the most recent iterate that met the
 * acceptable-level tests.  It outlives an attempt: the caller's stand-in for the restoration phase (a restart) falls back
 * to it when it fails, as IPOPT does when its restoration phase fails. */
typedef struct AccPoint { int have; int iter; double *x, *lam, *zl, *zu; OrcSolveInfo info; } AccPoint;

static int filter_rejects(const Filter *F, double th, double ph) {
  for (int i = 0; i < F->n; i++) if (th >= F->th[i] && ph >= F->ph[i]) return 1;
  return 0;
}

static double barrier_phi(const Nlp *P, const double *x, double f_scaled, double mu) {
  int n = P->I.n; double s = 0;
  for (int i = 0; i < n; i++) {
    if (P->xl[i] > -ORC_INF_BOUND) s += log(x[i] - P->xl[i]);
    if (P->xu[i] < ORC_INF_BOUND) s += log(P->xu[i] - x[i]);
  }
  return f_scaled - mu * s;
}

static int ipm_solve(Nlp *P, const OrcSolveOptions *opt, const double *xi, double *x,
                     double *lam_out, OrcSolveInfo *info, AccPoint *acc) {
  const int n = P->I.n, m = P->I.m, nk = n + m;
  const double kappa_eps = 10, kappa_mu = 0.2, theta_mu = 1.5, tau_min = 0.99, s_max = 100;
  const double gamma_theta = 1e-5, gamma_phi = 1e-8, delta_sw = 1, s_theta = 1.1, s_phi = 2.3, eta_phi = 1e-8;
  const double gamma_alpha = 0.05, kappa_sigma = 1e10, kappa1 = 1e-2, kappa2 = 1e-2;
  const double dw_min = 1e-20, dw_0 = 1e-4, dw_max = 1e40, kw_minus = 1.0 / 3, kw_plus = 8, kw_plus_bar = 100;
  const double mu_min_abs = opt->tol / 10;
  size_t szn = sizeof(double) * n, szm = sizeof(double) * m;
  double *lam = (double *)calloc(m, sizeof(double)), *zl = (double *)calloc(n, sizeof(double));
  double *zu = (double *)calloc(n, sizeof(double)), *g = (double *)malloc(szn), *c = (double *)malloc(szm);
  double *J = (double *)malloc(sizeof(double) * (size_t)m * n), *W = (double *)malloc(sizeof(double) * (size_t)n * n);
  double *K = (double *)malloc(sizeof(double) * (size_t)nk * nk), *rhs = (double *)malloc(sizeof(double) * nk);
  double *sol = (double *)malloc(sizeof(double) * nk), *dx = sol, *dlam = sol + n;
  double *dzl = (double *)malloc(szn), *dzu = (double *)malloc(szn), *xt = (double *)malloc(szn), *ct = (double *)malloc(szm);
  double *rhs_soc = (double *)malloc(sizeof(double) * nk), *sol_soc = (double *)malloc(sizeof(double) * nk);
  double *csoc = (double *)malloc(szm), *cs = (double *)malloc(szm), *xs = (double *)malloc(szn);
  char *hl = (char *)malloc(n), *hu = (char *)malloc(n);
  double *x_keep = (double *)malloc(szn), *lam_keep = (double *)malloc(szm), *zl_keep = (double *)malloc(szn), *zu_keep = (double *)malloc(szn);
  OrcSolveInfo info_keep; memset(&info_keep, 0, sizeof(info_keep));
  Ldl F; F.n = nk; F.M = K; F.perm = (int *)malloc(sizeof(int) * nk); F.blk = (char *)malloc(nk);
  Filter *flt = (Filter *)malloc(sizeof(Filter));
  int nb = 0, status = ORC_MAXITER_EXCEEDED;
  memset(info, 0, sizeof(*info));

  /* IPOPT option bound_relax_factor (default 1e-8; the reference's option string MPC.cpp:160-179 leaves it alone):
   * "before start of the optimization, the bounds given by the user are relaxed" -- OrigIpoptNLP::relax_bounds of
   * IPOPT 3.12: x_L -= factor * max(1, |x_L|), x_U += factor * max(1, |x_U|) for every finite variable bound
   * (equality constraints are not touched).  The whole iteration, the push of the start point included, sees the
   * relaxed bounds; the caller's are restored on return, and the final point is projected back into them
   * (honor_original_bounds, default "yes" in 3.12).  It matters exactly when a solve starts ON a bound: the fixed
   * psi_0 of a closed loop whose heading has reached yawHigh (test.cpp:79-111, examples/30-01-2.png) has slack 1e-12
   * against its own variable bound without it, and 1e-8 with it. */
  double *xl_user = P->xl, *xu_user = P->xu;
  double *xl_rel = (double *)malloc(szn), *xu_rel = (double *)malloc(szn);
  for (int i = 0; i < n; i++) {
    xl_rel[i] = xl_user[i]; xu_rel[i] = xu_user[i];
    if (xl_user[i] > -ORC_INF_BOUND) xl_rel[i] -= opt->bound_relax_factor * fmax(1.0, fabs(xl_user[i]));
    if (xu_user[i] < ORC_INF_BOUND) xu_rel[i] += opt->bound_relax_factor * fmax(1.0, fabs(xu_user[i]));
  }
  P->xl = xl_rel; P->xu = xu_rel;

  for (int i = 0; i < n; i++) {
    hl[i] = P->xl[i] > -ORC_INF_BOUND; hu[i] = P->xu[i] < ORC_INF_BOUND; nb += hl[i] + hu[i];
    double v = xi[i];
    /* push the start point into the interior (W&B section 3.6) */
    if (hl[i] && hu[i]) {
      double pl = fmin(kappa1 * fmax(1, fabs(P->xl[i])), kappa2 * (P->xu[i] - P->xl[i]));
      double pu = fmin(kappa1 * fmax(1, fabs(P->xu[i])), kappa2 * (P->xu[i] - P->xl[i]));
      v = fmin(fmax(v, P->xl[i] + pl), P->xu[i] - pu);
    } else if (hl[i]) v = fmax(v, P->xl[i] + kappa1 * fmax(1, fabs(P->xl[i])));
    else if (hu[i]) v = fmin(v, P->xu[i] - kappa1 * fmax(1, fabs(P->xu[i])));
    x[i] = v; zl[i] = hl[i] ? 1.0 : 0.0; zu[i] = hu[i] ? 1.0 : 0.0;
  }
  /* branch decisions: once, at the caller's xi (CppAD tapes at xi, before IPOPT's push) */
  tape_decide(P->cfg, xi, &P->tape);
  /* gradient-based objective scaling (IPOPT nlp_scaling_method=gradient-based, max gradient 100) */
  eval_grad_f(P->cfg, &P->tape, x, g);
  P->df = 1.0;
  if (opt->obj_scaling) {
    double gmax = 0; for (int i = 0; i < n; i++) gmax = fmax(gmax, fabs(g[i]));
    if (gmax > 100) P->df = fmax(100 / gmax, 1e-8);
  }
  const double df = P->df;

#define EVAL_C(xx, cc) do { eval_g(P->cfg, P->coef, P->nc, (xx), (cc)); for (int j_ = 0; j_ < m; j_++) (cc)[j_] -= P->gb[j_]; } while (0)

  if (opt->lam_init_ls) {
    /* least-squares multipliers: [I J^T; J 0][w; lam] = -[df g - zl + zu; 0] */
    eval_jac_g(P->cfg, P->coef, P->nc, x, J);
    memset(K, 0, sizeof(double) * (size_t)nk * nk);
    for (int i = 0; i < n; i++) K[(size_t)i * nk + i] = 1;
    for (int r = 0; r < m; r++) for (int cc = 0; cc < n; cc++) {
      double v = J[(size_t)r * n + cc]; K[(size_t)(n + r) * nk + cc] = v; K[(size_t)cc * nk + n + r] = v;
    }
    for (int i = 0; i < n; i++) rhs[i] = -(df * g[i] - zl[i] + zu[i]);
    for (int r = 0; r < m; r++) rhs[n + r] = 0;
    ldl_factor(&F); ldl_solve(&F, rhs, sol);
    double lmax = 0; for (int r = 0; r < m; r++) lmax = fmax(lmax, fabs(sol[n + r]));
    if (lmax <= 1000 && lmax == lmax) memcpy(lam, sol + n, szm);
  }

  double mu = 0.1, tau = fmax(tau_min, 1 - mu), dw_last = 0;
  EVAL_C(x, c);
  double theta0 = 0; for (int j = 0; j < m; j++) theta0 += fabs(c[j]);
  const double theta_max = 1e4 * fmax(1, theta0), theta_min = 1e-4 * fmax(1, theta0);
  flt->n = 0;
  int iter, n_polish = 0, acc_count = 0;
  double out_step = DBL_MAX;   /* |alpha d(delta_0, a_0)|_inf of the last accepted step */
  for (iter = 0; iter <= opt->max_iter; iter++) {
    double f = eval_f(P->cfg, &P->tape, x);
    eval_grad_f(P->cfg, &P->tape, x, g);
    EVAL_C(x, c);
    eval_jac_g(P->cfg, P->coef, P->nc, x, J);
    /* optimality error E_mu (W&B eq. 5) */
    double l1 = 0, z1 = 0, dinf = 0, cinf = 0;
    for (int j = 0; j < m; j++) { l1 += fabs(lam[j]); cinf = fmax(cinf, fabs(c[j])); }
    for (int i = 0; i < n; i++) {
      double r = df * g[i] - zl[i] + zu[i];
      for (int j = 0; j < m; j++) r += J[(size_t)j * n + i] * lam[j];
      dinf = fmax(dinf, fabs(r)); z1 += zl[i] + zu[i];
      if (opt->verbose > 1 && (i % P->cfg->N) == 0 && i < 6 * P->cfg->N) fprintf(stderr, "      row of x_0[%d]: %.3e\n", i / P->cfg->N, r);
    }
    double sd = fmax(s_max, (l1 + z1) / (m + nb)) / s_max, sc = fmax(s_max, z1 / (nb > 0 ? nb : 1)) / s_max;
    double cmax = 0, cmin = DBL_MAX;   /* range of the complementarity products */
    for (int i = 0; i < n; i++) {
      if (hl[i]) { double p = (x[i] - P->xl[i]) * zl[i]; cmax = fmax(cmax, p); cmin = fmin(cmin, p); }
      if (hu[i]) { double p = (P->xu[i] - x[i]) * zu[i]; cmax = fmax(cmax, p); cmin = fmin(cmin, p); }
    }
    if (nb == 0) cmin = 0;
#define COMPL(mu_) (fmax(fabs(cmax - (mu_)), fabs(cmin - (mu_))))
    double E0 = fmax(fmax(dinf / sd, cinf), COMPL(0.0) / sc);
    info->kkt_error = E0; info->mu = mu; info->obj = f; info->constr_viol = cinf;
    info->dual_inf = dinf / df; info->compl_inf = COMPL(0.0) / df; info->iterations = iter;
    if (opt->verbose) fprintf(stderr, "it %3d f=%.10g theta=%.3e dinf=%.3e compl=%.3e mu=%.2e E0=%.3e\n", iter, f, cinf, dinf, COMPL(0.0), mu, E0);
    if (!(E0 == E0)) { status = ORC_NUMERIC_ERROR; break; }
    /* OptimalityErrorConvergenceCheck (IPOPT 3.12): CONVERGED needs the scaled error within tol AND the unscaled dual
     * infeasibility, constraint violation and complementarity within their own tolerances; ACCEPTABLE is the same test
     * with the acceptable_* values (see mpc_oracle.h).  (No constraint scaling here: IPOPT's gradient-based scaling would
     * scale a constraint row only if its gradient at the start point exceeded 100, i.e. a road slope |f'(0)| > 100.) */
    const double dinf_u = dinf / df, compl_u = COMPL(0.0) / df;
    const int converged = E0 <= opt->tol && dinf_u <= opt->dual_inf_tol && cinf <= opt->constr_viol_tol && compl_u <= opt->compl_inf_tol;
    const int acceptable = opt->acceptable_iter > 0 && E0 <= opt->acceptable_tol && dinf_u <= opt->acceptable_dual_inf_tol &&
                           cinf <= opt->acceptable_constr_viol_tol && compl_u <= opt->acceptable_compl_inf_tol;
    if (n_polish > 0 && !converged) {
      /* a polish step must not cost what has been reached: one that leaves tol is dropped, the converged iterate returned */
      memcpy(x, x_keep, szn); memcpy(lam, lam_keep, szm); memcpy(zl, zl_keep, szn); memcpy(zu, zu_keep, szn);
      *info = info_keep; status = ORC_SUCCESS; break;
    }
    if (converged) {
      /* IPOPT stops at the first converged iterate.  Termination polish (OrcSolveOptions.polish, see
       * mpc_oracle.h): Newton steps at the final barrier parameter until the outputs have stopped moving. */
      if (!opt->polish || n_polish >= 6 || iter == opt->max_iter || (mu <= mu_min_abs && out_step <= opt->out_step_tol)) {
        status = ORC_SUCCESS; break;
      }
      n_polish++;
      if (mu > mu_min_abs) { mu = mu_min_abs; tau = fmax(tau_min, 1 - mu); flt->n = 0; }
    } else if (acceptable) {
      /* acceptable_iter acceptable iterates in a row: CONVERGED_TO_ACCEPTABLE_POINT (tested before max_iter, as IPOPT does) */
      if (++acc_count >= opt->acceptable_iter) { status = ORC_STOP_AT_ACCEPTABLE; break; }
    } else acc_count = 0;
    if (iter == opt->max_iter) break;
    /* barrier update (W&B eq. 7), repeated while the barrier problem is already solved */
    for (;;) {
      double Emu = fmax(fmax(dinf / sd, cinf), COMPL(mu) / sc);
      if (Emu <= kappa_eps * mu && mu > mu_min_abs) {
        mu = fmax(mu_min_abs, fmin(kappa_mu * mu, pow(mu, theta_mu)));
        if (opt->polish && mu < 3.0 * mu_min_abs) mu = mu_min_abs;   /* polish ends at the floor anyway (see mpc_oracle.h) */
        tau = fmax(tau_min, 1 - mu); flt->n = 0;
      } else break;
    }
    /* search direction with inertia correction (W&B section 3.1) */
    memset(W, 0, sizeof(double) * (size_t)n * n);
    add_hess_f(P->cfg, &P->tape, df, W);
    add_hess_g(P->cfg, P->coef, P->nc, x, lam, W);
    double dw = 0, dc = 0; int tries = 0, ok = 0;
    for (;;) {
      memset(K, 0, sizeof(double) * (size_t)nk * nk);
      for (int r = 0; r < n; r++) {
        for (int cc = 0; cc < n; cc++) K[(size_t)r * nk + cc] = W[(size_t)r * n + cc];
        double sg = 0;
        if (hl[r]) sg += zl[r] / (x[r] - P->xl[r]);
        if (hu[r]) sg += zu[r] / (P->xu[r] - x[r]);
        K[(size_t)r * nk + r] += sg + dw;
      }
      for (int r = 0; r < m; r++) {
        for (int cc = 0; cc < n; cc++) { double v = J[(size_t)r * n + cc]; K[(size_t)(n + r) * nk + cc] = v; K[(size_t)cc * nk + n + r] = v; }
        K[(size_t)(n + r) * nk + n + r] = -dc;
      }
      ldl_factor(&F);
      if (F.npos == n && F.nneg == m && F.nzero == 0) { ok = 1; break; }
      if (F.nzero > 0 && dc == 0) dc = 1e-8 * pow(mu, 0.25);
      if (dw == 0) dw = dw_last == 0 ? dw_0 : fmax(dw_min, kw_minus * dw_last);
      else dw = dw_last == 0 ? kw_plus_bar * dw : kw_plus * dw;
      if (dw > dw_max || ++tries > 200) break;
    }
    if (!ok) { status = ORC_RESTORATION_FAILURE; break; }
    if (dw > 0) { dw_last = dw; info->n_regularised++; }
    double dphi = 0;
    for (int i = 0; i < n; i++) {
      double gb = df * g[i];
      if (hl[i]) gb -= mu / (x[i] - P->xl[i]);
      if (hu[i]) gb += mu / (P->xu[i] - x[i]);
      double r = gb;
      for (int j = 0; j < m; j++) r += J[(size_t)j * n + i] * lam[j];
      rhs[i] = -r; dzl[i] = gb; /* keep the barrier gradient for the directional derivative */
    }
    for (int j = 0; j < m; j++) rhs[n + j] = -c[j];
    ldl_solve(&F, rhs, sol);
    for (int i = 0; i < n; i++) dphi += dzl[i] * dx[i];
    for (int i = 0; i < n; i++) {
      dzl[i] = hl[i] ? mu / (x[i] - P->xl[i]) - zl[i] - zl[i] / (x[i] - P->xl[i]) * dx[i] : 0;
      dzu[i] = hu[i] ? mu / (P->xu[i] - x[i]) - zu[i] + zu[i] / (P->xu[i] - x[i]) * dx[i] : 0;
    }
    /* fraction to the boundary (W&B eq. 15) */
    double amax = 1, az = 1;
    for (int i = 0; i < n; i++) {
      if (hl[i] && dx[i] < 0) amax = fmin(amax, -tau * (x[i] - P->xl[i]) / dx[i]);
      if (hu[i] && dx[i] > 0) amax = fmin(amax, tau * (P->xu[i] - x[i]) / dx[i]);
      if (hl[i] && dzl[i] < 0) az = fmin(az, -tau * zl[i] / dzl[i]);
      if (hu[i] && dzu[i] < 0) az = fmin(az, -tau * zu[i] / dzu[i]);
    }
    /* BacktrackingLineSearch::FindAcceptableTrialPoint begins by storing the current iterate if it is acceptable */
    if (acceptable && acc && !converged) {
      acc->have = 1; acc->iter = iter; acc->info = *info;
      memcpy(acc->x, x, szn); memcpy(acc->lam, lam, szm); memcpy(acc->zl, zl, szn); memcpy(acc->zu, zu, szn);
    }
    /* filter line search (W&B section 2.3, algorithm A) */
    double theta_k = 0; for (int j = 0; j < m; j++) theta_k += fabs(c[j]);
    double phi_k = barrier_phi(P, x, df * f, mu);
    double amin;
    if (dphi < 0) amin = gamma_alpha * fmin(fmin(gamma_theta, gamma_phi * theta_k / (-dphi)),
                                             theta_k <= theta_min ? delta_sw * pow(theta_k, s_theta) / pow(-dphi, s_phi) : gamma_theta);
    else amin = gamma_alpha * gamma_theta;
    double alpha = amax; int accepted = 0, ftype = 0;
    double dxn = 0, xn = 0; for (int i = 0; i < n; i++) { dxn = fmax(dxn, fabs(dx[i])); xn = fmax(xn, fabs(x[i])); }
    int tiny = dxn <= 10 * DBL_EPSILON * fmax(1.0, xn);
    for (;;) {
      for (int i = 0; i < n; i++) xt[i] = x[i] + alpha * dx[i];
      /* The fraction-to-the-boundary rule keeps every slack positive in exact arithmetic; with tau = 1 - mu = 1 - 1e-9 a
       * slack of 4e-8 is asked to shrink to 4e-17, below the spacing of doubles at x ~ 1, and x + alpha dx lands ON the
       * bound.  IPOPT repairs such slacks (CalculateSafeSlack moves the bound by ~eps); restated here, as in the device
       * solver, as "a trial point without strictly positive slacks is not acceptable": the step is halved. */
      int inside = 1;
      for (int i = 0; i < n; i++) if ((hl[i] && !(xt[i] - P->xl[i] > 0)) || (hu[i] && !(P->xu[i] - xt[i] > 0))) inside = 0;
      if (!inside && !tiny) { alpha *= 0.5; info->n_backtracks++; if (alpha < amin) break; continue; }
      EVAL_C(xt, ct);
      double theta_t = 0; for (int j = 0; j < m; j++) theta_t += fabs(ct[j]);
      double phi_t = barrier_phi(P, xt, df * eval_f(P->cfg, &P->tape, xt), mu);
      double eps_phi = 10 * DBL_EPSILON * fabs(phi_k);
      if (tiny) { accepted = 1; ftype = 1; break; }
      if (theta_t < theta_max && !filter_rejects(flt, theta_t, phi_t)) {
        int sw = dphi < 0 && alpha * pow(-dphi, s_phi) > delta_sw * pow(theta_k, s_theta);
        int armijo = phi_t - phi_k - eps_phi <= eta_phi * alpha * dphi;
        if (theta_k <= theta_min && sw) {
          if (armijo) accepted = 1;
        } else if (theta_t <= (1 - gamma_theta) * theta_k || phi_t - phi_k - eps_phi <= -gamma_phi * theta_k) {
          accepted = 1;
        }
        ftype = sw && armijo;   /* filter is augmented unless both hold (W&B step A-7) */
      }
      if (accepted) break;
      /* second-order correction (W&B A-5.5 .. A-5.10; OrcSolveOptions.max_soc): only for the first trial point */
      if (opt->max_soc > 0 && alpha == amax && theta_t >= theta_k) {
        const double a0 = alpha;
        double theta_old = theta_k, a_prev = a0;
        for (int j = 0; j < m; j++) csoc[j] = a0 * c[j] + ct[j];
        info->n_soc_tried++;
        for (int p = 1; p <= opt->max_soc && !accepted; p++) {
          for (int j = 0; j < m; j++) rhs_soc[n + j] = -csoc[j];
          memcpy(rhs_soc, rhs, szn);
          ldl_solve(&F, rhs_soc, sol_soc);
          const double *dxs = sol_soc, *dls = sol_soc + n;
          double as = 1;
          for (int i = 0; i < n; i++) {
            if (hl[i] && dxs[i] < 0) as = fmin(as, -tau * (x[i] - P->xl[i]) / dxs[i]);
            if (hu[i] && dxs[i] > 0) as = fmin(as, tau * (P->xu[i] - x[i]) / dxs[i]);
          }
          for (int i = 0; i < n; i++) xs[i] = x[i] + as * dxs[i];
          int ins = 1;
          for (int i = 0; i < n; i++) if ((hl[i] && !(xs[i] - P->xl[i] > 0)) || (hu[i] && !(P->xu[i] - xs[i] > 0))) ins = 0;
          if (!ins) break;
          EVAL_C(xs, cs);
          double theta_s = 0; for (int j = 0; j < m; j++) theta_s += fabs(cs[j]);
          double phi_s = barrier_phi(P, xs, df * eval_f(P->cfg, &P->tape, xs), mu);
          int acc_s = 0, ft_s = 0;
          if (theta_s < theta_max && !filter_rejects(flt, theta_s, phi_s)) {
            int sw = dphi < 0 && a0 * pow(-dphi, s_phi) > delta_sw * pow(theta_k, s_theta);
            int armijo = phi_s - phi_k - eps_phi <= eta_phi * a0 * dphi;
            if (theta_k <= theta_min && sw) { if (armijo) acc_s = 1; }
            else if (theta_s <= (1 - gamma_theta) * theta_k || phi_s - phi_k - eps_phi <= -gamma_phi * theta_k) acc_s = 1;
            ft_s = sw && armijo;
          }
          if (acc_s) {
            /* the corrected step replaces the step: primal, multipliers and bound duals all come from it */
            memcpy(xt, xs, szn);
            for (int i = 0; i < n; i++) dx[i] = dxs[i];
            for (int j = 0; j < m; j++) dlam[j] = dls[j];
            for (int i = 0; i < n; i++) {
              dzl[i] = hl[i] ? mu / (x[i] - P->xl[i]) - zl[i] - zl[i] / (x[i] - P->xl[i]) * dx[i] : 0;
              dzu[i] = hu[i] ? mu / (P->xu[i] - x[i]) - zu[i] + zu[i] / (P->xu[i] - x[i]) * dx[i] : 0;
            }
            az = 1;
            for (int i = 0; i < n; i++) {
              if (hl[i] && dzl[i] < 0) az = fmin(az, -tau * zl[i] / dzl[i]);
              if (hu[i] && dzu[i] < 0) az = fmin(az, -tau * zu[i] / dzu[i]);
            }
            alpha = as; accepted = 1; ftype = ft_s; info->n_soc_accepted++;
            break;
          }
          if (theta_s > 0.99 * theta_old) break;                 /* kappa_soc: the correction has stopped reducing the violation */
          for (int j = 0; j < m; j++) csoc[j] = as * csoc[j] + cs[j];
          theta_old = theta_s; a_prev = as;
        }
        (void)a_prev;
        if (accepted) break;
      }
      alpha *= 0.5; info->n_backtracks++;
      if (alpha < amin) break;
    }
    if (!accepted) {
      /* a polish step that finds no acceptable length: the iterate had already met tol, it is the answer */
      if (n_polish > 0 && converged) { status = ORC_SUCCESS; break; }
      /* IPOPT would now call its restoration phase -- unless the point is acceptable ("restoration phase called at
       * acceptable point": STOP_AT_ACCEPTABLE_POINT) or almost feasible (constraint violation <= 1e-2 tol: the stored
       * acceptable point if there is one, otherwise "restoration phase called, but point is almost feasible": a failure
       * without the restoration phase being tried -- info->no_restart tells the caller's stand-in not to restart either). */
      if (acceptable) { status = ORC_STOP_AT_ACCEPTABLE; break; }
      status = ORC_RESTORATION_FAILURE;
      if (opt->acceptable_iter > 0 && cinf <= 1e-2 * opt->tol) info->no_restart = 1;
      break;
    }
    if (!ftype && flt->n < 256) { flt->th[flt->n] = (1 - gamma_theta) * theta_k; flt->ph[flt->n] = phi_k - gamma_phi * theta_k; flt->n++; }
    if (n_polish > 0) {   /* keep the converged iterate: the step about to be taken is a polish step */
      memcpy(x_keep, x, szn); memcpy(lam_keep, lam, szm); memcpy(zl_keep, zl, szn); memcpy(zu_keep, zu, szn); info_keep = *info;
    }
    if (opt->verbose > 1) fprintf(stderr, "      step: alpha %.6f alpha_z %.6f\n", alpha, az);
    memcpy(x, xt, szn);
    out_step = alpha * fmax(fmax(fabs(dx[P->I.delta]), fabs(dx[P->I.a])), 0.03 * dxn);   /* outputs, and 0.03 x any primal variable (trajectory: 1e-5 m) */
    for (int j = 0; j < m; j++) lam[j] += alpha * dlam[j];
    for (int i = 0; i < n; i++) {
      if (hl[i]) {
        zl[i] += az * dzl[i];
        double s = x[i] - P->xl[i];
        zl[i] = fmax(fmin(zl[i], kappa_sigma * mu / s), mu / (kappa_sigma * s));   /* W&B eq. 16 */
      }
      if (hu[i]) {
        zu[i] += az * dzu[i];
        double s = P->xu[i] - x[i];
        zu[i] = fmax(fmin(zu[i], kappa_sigma * mu / s), mu / (kappa_sigma * s));
      }
    }
  }
#undef COMPL
#undef EVAL_C
  /* RestoreAcceptablePoint: a restoration failure that no restart will follow (almost feasible point, or this already is
   * the restart: `acc` then also holds what the first attempt stored) ends at the stored acceptable point */
  if (status == ORC_RESTORATION_FAILURE && acc && acc->have && (info->no_restart || !opt->lam_init_ls)) {
    const int it_now = info->iterations, nreg = info->n_regularised, nbt = info->n_backtracks;
    memcpy(x, acc->x, szn); memcpy(lam, acc->lam, szm); memcpy(zl, acc->zl, szn); memcpy(zu, acc->zu, szn);
    *info = acc->info; info->iterations = it_now; info->n_regularised = nreg; info->n_backtracks = nbt;
    info->acceptable_restored_older = 1;
    status = ORC_STOP_AT_ACCEPTABLE;
  }
  info->status = status;
  P->xl = xl_user; P->xu = xu_user;
  if (opt->honor_original_bounds)   /* OrigIpoptNLP::FinalizeSolution: the returned x lies inside the bounds the user gave */
    for (int i = 0; i < n; i++) x[i] = fmin(fmax(x[i], xl_user[i]), xu_user[i]);
  free(xl_rel); free(xu_rel);
  if (lam_out) memcpy(lam_out, lam, szm);
  free(lam); free(zl); free(zu); free(g); free(c); free(J); free(W); free(K); free(rhs); free(sol);
  free(rhs_soc); free(sol_soc); free(csoc); free(cs); free(xs);
  free(dzl); free(dzu); free(xt); free(ct); free(hl); free(hu); free(F.perm); free(F.blk); free(flt);
  free(x_keep); free(lam_keep); free(zl_keep); free(zu_keep);
  return status;
}

/* ------------------------------------------------------------------------- */
/* MPC::solve                                                                 */
/* ------------------------------------------------------------------------- */
static void solve_bounds(const OrcConfig *cfg, const double *state, double *xi, double *xl,
                         double *xu, double *gb) {
  Idx I = make_idx(cfg->N);
  for (int i = 0; i < I.n; i++) xi[i] = 0.0;                                   /* MPC.cpp:207-210 */
  xi[I.x] = state[0]; xi[I.y] = state[1]; xi[I.psi] = state[2];                /* :213-218 */
  xi[I.v] = state[3]; xi[I.cte] = state[4]; xi[I.epsi] = state[5];
  for (int i = 0; i < I.psi; i++) { xl[i] = -1.0e19; xu[i] = 1.0e19; }         /* :222-225 */
  for (int i = I.psi; i < I.v; i++) { xl[i] = cfg->yaw_low; xu[i] = cfg->yaw_high; }      /* :229-232 */
  for (int i = I.v; i < I.cte; i++) { xl[i] = -cfg->max_speed; xu[i] = cfg->max_speed; }  /* :236-239 */
  for (int i = I.cte; i < I.delta; i++) { xl[i] = -1.0e19; xu[i] = 1.0e19; }   /* :242-245 */
  for (int i = I.delta; i < I.a; i++) { xl[i] = -cfg->max_steering; xu[i] = cfg->max_steering; }          /* :248-251 */
  for (int i = I.a; i < I.n; i++) { xl[i] = cfg->max_deceleration; xu[i] = cfg->max_acceleration; }       /* :254-257 */
  for (int i = 0; i < I.m; i++) gb[i] = 0;                                     /* :263-266 */
  gb[I.x] = state[0]; gb[I.y] = state[1]; gb[I.psi] = state[2];                /* :269-281 */
  gb[I.v] = state[3]; gb[I.cte] = state[4]; gb[I.epsi] = state[5];
}

int orc_mpc_solve(const OrcConfig *cfg, const OrcSolveOptions *opt_in, const double *state,
                  const double *coef, int nc, double *out9, double *traj_x, double *traj_y,
                  double *sol_out, OrcSolveInfo *info_out) {
  OrcSolveOptions opt; if (opt_in) opt = *opt_in; else orc_default_options(&opt);
  Idx I = make_idx(cfg->N);
  if (cfg->N < 3 || cfg->N > ORC_NMAX) return ORC_NUMERIC_ERROR;
  Nlp P; P.cfg = cfg; P.coef = coef; P.nc = nc; P.I = I; P.branch_mode = opt.branch_mode;
  double *buf = (double *)malloc(sizeof(double) * (4 * I.n + I.m));
  double *xi = buf, *xl = xi + I.n, *xu = xl + I.n, *x = xu + I.n, *gb = x + I.n;
  P.xl = xl; P.xu = xu; P.gb = gb;
  solve_bounds(cfg, state, xi, xl, xu, gb);
  OrcSolveInfo info; memset(&info, 0, sizeof(info));
  int status;
  const double rf = opt.bound_relax_factor;
  if (state[2] < cfg->yaw_low - rf * fmax(1.0, fabs(cfg->yaw_low)) || state[2] > cfg->yaw_high + rf * fmax(1.0, fabs(cfg->yaw_high)) ||
      fabs(state[3]) > cfg->max_speed + rf * fmax(1.0, cfg->max_speed)) {
    /* fixed initial state outside its own variable bounds: the reference NLP is infeasible */
    status = ORC_INFEASIBLE_START; info.status = status; memcpy(x, xi, sizeof(double) * I.n);
    tape_decide(cfg, xi, &P.tape); info.obj = eval_f(cfg, &P.tape, x);
  } else if (opt.branch_mode == ORC_BRANCH_LIVE) {
    /* LIVE: re-decide branches at every accepted iterate by restarting the
     * frozen solve from the previous solution until the tape stops changing.
     * (Not the reference semantics; kept for the F3 comparison only.) */
    memcpy(x, xi, sizeof(double) * I.n);
    double *x0 = (double *)malloc(sizeof(double) * I.n); Tape *prev = (Tape *)malloc(sizeof(Tape));
    status = ORC_MAXITER_EXCEEDED;
    for (int rep = 0; rep < 50; rep++) {
      memcpy(x0, rep == 0 ? xi : x, sizeof(double) * I.n);
      tape_decide(cfg, x0, prev);
      status = ipm_solve(&P, &opt, x0, x, NULL, &info, NULL);
      Tape *now = (Tape *)malloc(sizeof(Tape)); tape_decide(cfg, x, now);
      int same = !memcmp(now, prev, sizeof(Tape)); free(now);
      if (same || status != ORC_SUCCESS) break;
    }
    free(x0); free(prev);
  } else {
    AccPoint acc; acc.have = 0; acc.iter = 0;
    double *accbuf = (double *)malloc(sizeof(double) * (3 * I.n + I.m));
    acc.x = accbuf; acc.zl = accbuf + I.n; acc.zu = accbuf + 2 * I.n; acc.lam = accbuf + 3 * I.n;
    status = ipm_solve(&P, &opt, xi, x, NULL, &info, &acc);
    if (status == ORC_RESTORATION_FAILURE && opt.lam_init_ls && !info.no_restart) {
      /* IPOPT would switch to its feasibility-restoration phase here, which this oracle does not
       * restate.  Stand-in: restart from the same start point with zero equality multipliers
       * (what IPOPT itself falls back to when the least-squares estimate is rejected). */
      OrcSolveOptions o2 = opt; o2.lam_init_ls = 0;
      OrcSolveInfo i2;
      int s2 = ipm_solve(&P, &o2, xi, x, NULL, &i2, &acc);
      i2.iterations += info.iterations; i2.n_regularised += info.n_regularised; i2.n_backtracks += info.n_backtracks;
      info = i2; status = s2;
    }
    free(accbuf);
  }
  if (traj_x && traj_y) for (int i = 0; i < I.N; i++) { traj_x[i] = x[I.x + i]; traj_y[i] = x[I.y + i]; }  /* :306-311 */
  out9[0] = x[I.x + 1]; out9[1] = x[I.y + 1]; out9[2] = x[I.psi + 1]; out9[3] = x[I.v + 1];               /* :322-324 */
  out9[4] = x[I.cte + 1]; out9[5] = x[I.epsi + 1]; out9[6] = x[I.delta]; out9[7] = x[I.a]; out9[8] = info.obj;
  if (sol_out) memcpy(sol_out, x, sizeof(double) * I.n);
  if (info_out) *info_out = info;
  free(buf);
  return status;
}

/* ------------------------------------------------------------------------- */
/* MPC::run                                                                   */
/* ------------------------------------------------------------------------- */
void orc_mpc_run_pre(const OrcConfig *cfg, const double *pose, double *ptsx, double *ptsy,
                     int npts, OrcRunPre *pre) {
  memset(pre, 0, sizeof(*pre));
  orc_global_to_vehicle(pose[0], pose[1], pose[2], ptsx, ptsy, npts);               /* MPC.cpp:329 */
  pre->nc = orc_road_fit(ptsx, ptsy, npts, cfg->max_fit_order, cfg->max_fit_error, pre->coef, NULL); /* :330 */
  double cte = orc_polyeval(pre->coef, pre->nc, 0);                                 /* :334 */
  double epsi = -atan(pre->coef[1]);                                                /* :336 */
  double back = ptsx[npts - 1], front = ptsx[0];
  pre->max_yaw_change = orc_orientation_change(pre->coef, pre->nc, 0, back) * (back - front) / back; /* :339 */
  pre->max_speed = orc_yaw_change_speed_limit(cfg, pre->max_yaw_change, cfg->max_speed);  /* :340 */
  pre->target_speed = orc_speed_target(cfg, pose[4], pre->max_speed);               /* :342 */
  if (pre->max_yaw_change < 0) { pre->yaw_low = pre->max_yaw_change; pre->yaw_high = 0.1; }  /* :345-352 */
  else { pre->yaw_low = -0.1; pre->yaw_high = pre->max_yaw_change; }
  pre->state[0] = 0; pre->state[1] = 0; pre->state[2] = 0; pre->state[3] = pose[3];  /* :355-356 */
  pre->state[4] = cte; pre->state[5] = epsi;
}

void orc_mpc_run_post(const OrcConfig *cfg, const OrcRunPre *pre, double v0,
                      const double *result, double *out8) {
  double steer_angle = result[6];                                                   /* :360 */
  if (fabs(pre->max_yaw_change) > cfg->steer_adj_thresh)                            /* :364-366 */
    steer_angle += cfg->steer_adj_ratio * pre->max_yaw_change;
  double accel = fmin(result[7], pre->target_speed - v0);                           /* :369 */
  double steer_value = clampd(steer_angle / cfg->max_steering, -1.0, 1.0);          /* :371 */
  out8[0] = result[0]; out8[1] = result[1]; out8[2] = result[2]; out8[3] = result[3];  /* :381 */
  out8[4] = steer_value; out8[5] = accel; out8[6] = result[4]; out8[7] = result[5];
}

int orc_mpc_run(OrcConfig *cfg, const OrcSolveOptions *opt, const double *pose, double *ptsx,
                double *ptsy, int npts, double *out8, double *traj_x, double *traj_y,
                OrcRunPre *pre_out, OrcSolveInfo *info) {
  OrcRunPre pre; double r9[9];
  orc_mpc_run_pre(cfg, pose, ptsx, ptsy, npts, &pre);
  cfg->yaw_low = pre.yaw_low; cfg->yaw_high = pre.yaw_high;   /* the reference mutates Config here */
  int st = orc_mpc_solve(cfg, opt, pre.state, pre.coef, pre.nc, r9, traj_x, traj_y, NULL, info);
  orc_mpc_run_post(cfg, &pre, pose[3], r9, out8);
  if (pre_out) *pre_out = pre;
  return st;
}

/* ------------------------------------------------------------------------- */
/* solver-independent KKT certificate                                         */
/* ------------------------------------------------------------------------- */
double orc_kkt_certificate(const OrcConfig *cfg, const double *state, const double *coef,
                           int nc, const double *vars, double active_tol,
                           double *stat_res, double *prim_res, double *bound_res) {
  Idx I = make_idx(cfg->N);
  int n = I.n, m = I.m;
  double *buf = (double *)malloc(sizeof(double) * (3 * n + 2 * m + n));
  double *xi = buf, *xl = xi + n, *xu = xl + n, *gb = xu + n, *g = gb + m, *grad = g + m;
  solve_bounds(cfg, state, xi, xl, xu, gb);
  Tape *t = (Tape *)malloc(sizeof(Tape)); tape_decide(cfg, xi, t);
  eval_g(cfg, coef, nc, vars, g);
  double pr = 0, br = 0;
  for (int j = 0; j < m; j++) pr = fmax(pr, fabs(g[j] - gb[j]));
  int nact = 0; int *act = (int *)malloc(sizeof(int) * 2 * n); /* +i+1 upper, -(i+1) lower */
  for (int i = 0; i < n; i++) {
    if (xl[i] > -ORC_INF_BOUND) { br = fmax(br, xl[i] - vars[i]); if (vars[i] - xl[i] <= active_tol) act[nact++] = -(i + 1); }
    if (xu[i] < ORC_INF_BOUND) { br = fmax(br, vars[i] - xu[i]); if (xu[i] - vars[i] <= active_tol) act[nact++] = i + 1; }
  }
  eval_grad_f(cfg, t, vars, grad);
  double *J = (double *)malloc(sizeof(double) * (size_t)m * n);
  eval_jac_g(cfg, coef, nc, vars, J);
  /* min || grad + J^T lam + sum_act s_a e_a z_a ||  over lam free, z_a free; then require z_a >= 0 */
  int cols = m + nact;
  double *A = (double *)calloc((size_t)n * cols, sizeof(double)), *b = (double *)malloc(sizeof(double) * n);
  double *sol = (double *)malloc(sizeof(double) * cols);
  for (int i = 0; i < n; i++) { for (int j = 0; j < m; j++) A[(size_t)i * cols + j] = J[(size_t)j * n + i]; b[i] = -grad[i]; }
  for (int a = 0; a < nact; a++) { int i = abs(act[a]) - 1; A[(size_t)i * cols + m + a] = act[a] > 0 ? 1.0 : -1.0; }
  double *A2 = (double *)malloc(sizeof(double) * (size_t)n * cols), *b2 = (double *)malloc(sizeof(double) * n);
  memcpy(A2, A, sizeof(double) * (size_t)n * cols); memcpy(b2, b, sizeof(double) * n);
  lstsq_qr(A2, n, cols, b2, sol);
  double sr = 0, gscale = 1;
  for (int i = 0; i < n; i++) gscale = fmax(gscale, fabs(grad[i]));
  for (int i = 0; i < n; i++) {
    double r = -b[i];
    for (int j = 0; j < cols; j++) r += A[(size_t)i * cols + j] * sol[j];
    sr = fmax(sr, fabs(r));
  }
  double zneg = 0; for (int a = 0; a < nact; a++) zneg = fmax(zneg, -sol[m + a]);
  sr = fmax(sr, zneg) / gscale;   /* relative to the gradient scale */
  if (stat_res) *stat_res = sr; if (prim_res) *prim_res = pr; if (bound_res) *bound_res = fmax(br, 0);
  free(A); free(b); free(sol); free(A2); free(b2); free(J); free(act); free(t); free(buf);
  return fmax(sr, fmax(pr, fmax(br, 0)));
}
