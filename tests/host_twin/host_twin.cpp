/*
 * host_twin.cpp -- TEST-ONLY CPU build of carnd-mpc-project_amd/csrc/mpc_core.h.
 *
 * The build container has no GPU, so the interior-point/Riccati logic of the
 * device solver is debugged here by compiling the very same header with g++
 * and checking it against the oracle (tests/test_host_twin.py).  This file is
 * never linked into the product library and nothing in the product path
 * loads it: the shipped path is HIP only and fails loudly without a device.
 */
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mpc_core.h"

extern "C" int mpc_host_twin_solve(const MpcParams *p, int64_t B, int64_t ld, const double *state,
                                   const double *coeffs, const double *yaw_lo, const double *yaw_hi,
                                   const double *weights, double *out, double *traj, int32_t *status,
                                   int32_t *iters) {
  if (!p || p->N < 3 || p->N > MPC_MAX_N) return MPC_ERR_INVALID;
  const int N = p->N;
  std::vector<double> wsbuf((size_t)(N - 1) * mpc::STAGE_SZ_GLOBAL);
  for (int64_t i = 0; i < B; i++) {
    double st[6], cf[MPC_NCOEF], w[MPC_NW], o9[9];
    std::vector<double> tr(2 * N);
    for (int q = 0; q < 6; q++) st[q] = state[q * ld + i];
    for (int q = 0; q < MPC_NCOEF; q++) cf[q] = coeffs[q * ld + i];
    for (int q = 0; q < MPC_NW; q++) w[q] = weights ? weights[q * ld + i] : p->weights[q];
    mpc::HostWorkspace ws{wsbuf.data()};
    int it = 0;
    int s = mpc::solve_instance(*p, ws, st, cf, yaw_lo[i], yaw_hi[i], w, o9, traj ? tr.data() : nullptr, &it);
    for (int q = 0; q < 9; q++) out[q * ld + i] = o9[q];
    if (traj) for (int q = 0; q < 2 * N; q++) traj[q * ld + i] = tr[q];
    status[i] = s;
    if (iters) iters[i] = it;
  }
  return MPC_OK;
}

extern "C" void mpc_host_twin_math(int64_t n, const double *x, double *sn, double *cs, double *rc) {
  for (int64_t i = 0; i < n; i++) { mpc::fsincos(x[i], &sn[i], &cs[i]); rc[i] = mpc::frcp(x[i]); }
}
