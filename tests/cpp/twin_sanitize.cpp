/*
 * twin_sanitize.cpp -- TEST-ONLY: the device solver header (fp64 and fp32 instantiations) and the oracle under
 * AddressSanitizer + UndefinedBehaviorSanitizer on the CPU (GPU sanitizers are not available on the pool).
 * Reads a batch written by tests/test_sanitizers.py:  header int32 {N, B, has_weights}, double dt, then the config path is
 * given on the command line; arrays state[6][B], coeffs[5][B], yaw_lo[B], yaw_hi[B], weights[12][B] (if any), doubles.
 * Solves every instance with the fp64 twin, the fp32 twin and the oracle and prints the worst disagreement.
 */
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mpc_core.h"
extern "C" {
#include "mpc_oracle.h"
}

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  MpcParams p;
  if (mpc_params_load_json(argv[1], &p) != MPC_OK) return 3;
  FILE *f = fopen(argv[2], "rb");
  if (!f) return 4;
  int32_t hdr[3];
  double dt;
  if (fread(hdr, sizeof(int32_t), 3, f) != 3 || fread(&dt, sizeof(double), 1, f) != 1) return 5;
  const int N = hdr[0], B = hdr[1], hasw = hdr[2];
  p.N = N; p.dt = dt;
  std::vector<double> in((size_t)(13 + (hasw ? 12 : 0)) * B);
  if (fread(in.data(), sizeof(double), in.size(), f) != in.size()) return 6;
  fclose(f);
  OrcConfig cfg;
  if (orc_config_load(argv[1], &cfg) != 0) return 7;
  cfg.N = N; cfg.dt = dt;
  std::vector<double> ws64((size_t)mpc::workspace_fields_per_instance(N, false, true));
  std::vector<float> ws32((size_t)mpc::workspace_fields_per_instance(N, true, true));
  double worst64 = 0, worst32 = 0;
  int bad = 0;
  for (int i = 0; i < B; i++) {
    double st[6], cf[5], w[12], o64[9], tr[2 * MPC_MAX_N];
    float stf[6], cff[5], wf[12], o32[9], trf[2 * MPC_MAX_N];
    for (int q = 0; q < 6; q++) { st[q] = in[(size_t)q * B + i]; stf[q] = (float)st[q]; }
    for (int q = 0; q < 5; q++) { cf[q] = in[(size_t)(6 + q) * B + i]; cff[q] = (float)cf[q]; }
    const double yl = in[(size_t)11 * B + i], yh = in[(size_t)12 * B + i];
    for (int q = 0; q < 12; q++) { w[q] = hasw ? in[(size_t)(13 + q) * B + i] : p.weights[q]; wf[q] = (float)w[q]; cfg.weights[q] = w[q]; }
    int it64 = 0, it32 = 0;
    const int s64 = mpc::solve_instance<mpc::HostWorkspace<double>, double>(p, mpc::HostWorkspace<double>{ws64.data()}, st, cf, yl, yh, w, o64, tr, &it64);
    MpcParams q32 = p; q32.precision = MPC_PRECISION_F32;
    const int s32 = mpc::solve_instance<mpc::HostWorkspace<float>, float>(q32, mpc::HostWorkspace<float>{ws32.data()}, stf, cff, (float)yl, (float)yh, wf, o32, trf, &it32);
    cfg.yaw_low = yl; cfg.yaw_high = yh;
    double o9[9];
    OrcSolveInfo info;
    const int so = orc_mpc_solve(&cfg, NULL, st, cf, 5, o9, NULL, NULL, NULL, &info);
    if (s64 != so) bad++;
    if (s64 == 0 && so == 0) for (int q = 0; q < 8; q++) worst64 = fmax(worst64, fabs(o64[q] - o9[q]));
    if (s32 == 0 && so == 0) worst32 = fmax(worst32, fabs((double)o32[6] - o9[6]));
  }
  printf("instances %d status_mismatch %d worst_fp64 %.3e worst_fp32_steer %.3e\n", B, bad, worst64, worst32);
  return (bad == 0 && worst64 < 1e-6 && worst32 < 5e-3) ? 0 : 1;
}
