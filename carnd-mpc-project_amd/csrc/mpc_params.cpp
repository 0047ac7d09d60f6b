/*
 * mpc_params.cpp -- host side: MpcParams from the reference's config JSON.
 *
 * Mirrors Config::load() (src/utils/Config.cpp:31-87) and the compiled-in
 * defaults (src/utils/Config.cpp:5-29): same keys, same unit conversions
 * (mph -> m/s with 1609.34/3600, deg -> rad), same table rescaling rule, same
 * clamp of the steer-adjustment ratio, same "weights.size() > 11" requirement.
 * The reference parses with nlohmann::json; the config files are flat objects
 * of numbers and number arrays, so a small purpose-built reader is used here.
 */
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "mpc_amd.h"

namespace {

double mph2mps(double mph) { return mph * 1609.34 / 3600.0; }  /* utils.h:11-13 */
double deg2rad(double x) { return x * M_PI / 180; }            /* utils.h:69 */

struct FlatJson {
  std::string txt;
  bool find(const char *key, size_t &pos) const {
    std::string pat = std::string("\"") + key + "\"";
    size_t p = txt.find(pat);
    if (p == std::string::npos) return false;
    p = txt.find(':', p + pat.size());
    if (p == std::string::npos) return false;
    pos = p + 1;
    return true;
  }
  bool num(const char *key, double &out) const {
    size_t p;
    if (!find(key, p)) return false;
    const char *s = txt.c_str() + p;
    char *end;
    double v = strtod(s, &end);
    if (end == s) return false;
    out = v;
    return true;
  }
  int arr(const char *key, double *out, int cap) const {
    size_t p;
    if (!find(key, p)) return -1;
    p = txt.find('[', p);
    if (p == std::string::npos) return -1;
    const char *s = txt.c_str() + p + 1;
    int n = 0;
    for (;;) {
      while (*s == ' ' || *s == ',' || *s == '\n' || *s == '\t' || *s == '\r') s++;
      if (*s == ']' || !*s) break;
      char *end;
      double v = strtod(s, &end);
      if (end == s) return -1;
      if (n < cap) out[n] = v;
      n++;
      s = end;
    }
    return n;
  }
};

}  // namespace

extern "C" int mpc_params_default(MpcParams *p) {
  if (!p) return MPC_ERR_INVALID;
  memset(p, 0, sizeof(*p));
  p->abi_version = MPC_ABI_VERSION;
  p->N = 25;                               /* Config.cpp:5 */
  p->max_fit_order = 4;                    /* :7 */
  p->max_fit_error = 0.5;                  /* :8 */
  p->latency_ms = 100;                     /* :9 */
  p->lookahead = 0;                        /* :10 */
  p->ipopt_timeout = 0.5;                  /* :11 */
  p->dt = 0.025;                           /* :12 */
  p->max_steering = deg2rad(25.0);         /* :13 */
  p->max_acceleration = mph2mps(8);        /* :14 */
  p->max_deceleration = mph2mps(-20);      /* :15 */
  p->max_speed = mph2mps(100);             /* :16 */
  p->Lf = 2.67;                            /* :17 */
  p->epsi_panic = 1;                       /* :18 */
  p->cte_panic = 0.6;                      /* :19 */
  p->steer_adj_thresh = 0.6;               /* :20 */
  p->steer_adj_ratio = 0.025;              /* :21 */
  const double w[8] = {100, 100, 1, 1, 1, 5000, 1, 1000};   /* :22 (only 8 entries by default) */
  for (int i = 0; i < 8; i++) p->weights[i] = w[i];
  const double st[3] = {0.1, 0.2, 0.3};                     /* :23 */
  p->n_steers = 3;
  for (int i = 0; i < 3; i++) p->steers[i] = st[i];
  const double ss[4] = {mph2mps(80), mph2mps(65), mph2mps(30), mph2mps(25)};  /* :24 */
  p->n_steer_speeds = 4;
  for (int i = 0; i < 4; i++) p->steer_speeds[i] = ss[i];
  p->branch_mode = MPC_BRANCH_FROZEN;
  p->precision = MPC_PRECISION_F64;
  p->max_iter = 200;
  p->tol = 1e-8;
  p->out_step_tol = 3e-7;
  p->tol_f32 = 5e-4;
  p->polish = 1;
  p->honor_original_bounds = 1;
  p->bound_relax_factor = 1e-8;            /* IPOPT default */
  p->tail_cut = 0; p->tail_ring = 128; p->tail_capacity = 0;
  p->f32_finish = 1; p->f64_f32_start = MPC_F32_START_AUTO; p->mixed_switch_mu = 2e-5;
  p->lane_compact = MPC_LANE_COMPACT_AUTO; p->f32_phase_refill = 0;
  /* IPOPT 3.12 defaults of the termination tests the reference's option string leaves alone (MPC.cpp:160-179) */
  p->acceptable_iter = 15; p->dual_inf_tol = 1.0; p->constr_viol_tol = 1e-4; p->compl_inf_tol = 1e-4;
  p->initial_state_rows = 0; p->wave_max_batch = 0;
  p->acceptable_tol = 1e-6; p->acceptable_dual_inf_tol = 1e10; p->acceptable_constr_viol_tol = 1e-2; p->acceptable_compl_inf_tol = 1e-2;
  return MPC_OK;
}

extern "C" int mpc_params_load_json(const char *path, MpcParams *p) {
  if (!path || !p) return MPC_ERR_INVALID;
  FILE *f = fopen(path, "rb");
  if (!f) return MPC_ERR_IO;
  FlatJson js;
  char buf[4096];
  size_t n;
  while ((n = fread(buf, 1, sizeof(buf), f)) > 0) js.txt.append(buf, n);
  fclose(f);
  mpc_params_default(p);
  double v;
  bool ok = true;
  ok &= js.num("N", v); p->N = (int)v;                                         /* Config.cpp:39 */
  ok &= js.num("dt", p->dt);                                                   /* :40 */
  ok &= js.num("max acceleration", v); p->max_acceleration = mph2mps(v);       /* :41-42 */
  ok &= js.num("max deceleration", v); p->max_deceleration = mph2mps(v);       /* :43-44 */
  ok &= js.num("max steering", v); p->max_steering = deg2rad(v);               /* :45 */
  ok &= js.num("max speed", v); p->max_speed = mph2mps(v);                     /* :46-47 */
  const double speed_scale = p->max_speed / mph2mps(100.0);                    /* :48 */
  ok &= js.num("latency", v); p->latency_ms = (int)v;                          /* :49 */
  p->lookahead = p->latency_ms * 1.0E-3;                                       /* :50 */
  ok &= js.num("max polynomial fitting order", v); p->max_fit_order = (int)v;  /* :51 */
  ok &= js.num("max polynomial fitting error", p->max_fit_error);              /* :52 */
  ok &= js.num("ipopt timeout", p->ipopt_timeout);                             /* :53 */
  ok &= js.num("Lf", p->Lf);                                                   /* :54 */
  ok &= js.num("epsi panic", p->epsi_panic);                                   /* :55 */
  ok &= js.num("cte panic", p->cte_panic);                                     /* :56 */
  ok &= js.num("steer adjustment threshold", p->steer_adj_thresh);             /* :57 */
  ok &= js.num("steer adjustment ratio", v);                                   /* :58 */
  p->steer_adj_ratio = v < 0.0 ? 0.0 : (v > 0.1 ? 0.1 : v);                    /* :59 */
  double tmp[MPC_MAX_TABLE];
  int nw = js.arr("weights", tmp, MPC_MAX_TABLE);                              /* :60 */
  if (nw <= 11) return MPC_ERR_IO;                                             /* :61 assert */
  for (int i = 0; i < MPC_NW; i++) p->weights[i] = tmp[i];
  p->n_steers = js.arr("steers", p->steers, MPC_MAX_TABLE);                    /* :63-64 */
  p->n_steer_speeds = js.arr("steer speeds", p->steer_speeds, MPC_MAX_TABLE);  /* :65 */
  p->n_yaw_changes = js.arr("yaw changes", p->yaw_changes, MPC_MAX_TABLE);     /* :75-76 */
  p->n_yaw_change_speeds = js.arr("yaw change speeds", p->yaw_change_speeds, MPC_MAX_TABLE); /* :77 */
  if (p->n_steers < 0 || p->n_steer_speeds < 0 || p->n_yaw_changes < 0 || p->n_yaw_change_speeds < 0) return MPC_ERR_IO;
  if (p->n_steers > MPC_MAX_TABLE || p->n_steer_speeds > MPC_MAX_TABLE || p->n_yaw_changes > MPC_MAX_TABLE ||
      p->n_yaw_change_speeds > MPC_MAX_TABLE) return MPC_ERR_IO;
  for (int i = 0; i < p->n_steer_speeds; i++)                                  /* :66-73 */
    p->steer_speeds[i] = speed_scale <= 1 ? std::fmin(mph2mps(p->steer_speeds[i]), p->max_speed)
                                          : mph2mps(p->steer_speeds[i]) * speed_scale;
  for (int i = 0; i < p->n_yaw_change_speeds; i++)                             /* :78-85 */
    p->yaw_change_speeds[i] = speed_scale <= 1 ? std::fmin(mph2mps(p->yaw_change_speeds[i]), p->max_speed)
                                               : mph2mps(p->yaw_change_speeds[i]) * speed_scale;
  return ok ? MPC_OK : MPC_ERR_IO;
}

extern "C" int mpc_inflight_advice(const MpcParams *p, int64_t B) {
  if (!p) return MPC_ERR_INVALID;
  /* two devices' worth of lanes in flight (131 072 instances: measured at N = 10, 8 192 .. 65 536 per launch, 8 x 16 384 and
   * 4 x 32 768 give what 2 x 65 536 gives), between 2 and 8 launches; two-launch solves and long horizons at least 4 */
  int64_t n = B > 0 ? (131072 + B - 1) / B : 8;
  if (n < 2) n = 2;
  if (n > 8) n = 8;
  if ((p->precision == MPC_PRECISION_F32 || p->N >= MPC_F32_START_AUTO_N || p->f64_f32_start == MPC_F32_START_ON) && n < 4) n = 4;
  return (int)n;
}
