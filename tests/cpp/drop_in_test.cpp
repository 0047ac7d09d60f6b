// Compiles the drop-in header the way a user of the reference would use it: the call sequence of the
// reference's src/test.cpp:16-111 (Config::load, Vehicle::update, MPC::run once, then closed-loop
// MPC::solve), minus the matplotlib plotting.  Prints one line per solve; tests/test_drop_in.py checks
// the numbers against the oracle.  Exits with 3 when no GPU is present (there is no CPU fallback).
#include <cstdio>
#include <string>
#include <vector>

#include "mpc_drop_in.hpp"

int main(int argc, char **argv) {
  const std::string cfg = argc > 1 ? argv[1] : "../config-stable.json";
  const int iters = argc > 2 ? atoi(argv[2]) : 25;
  try {
    MPC mpc;
    Config::load(cfg);
    std::vector<double> ptsx = {-145.1165, -158.3417, -164.3164, -169.3365, -175.4917, -176.9617};
    std::vector<double> ptsy = {4.339378, -17.42898, -30.18062, -42.84062, -66.52898, -76.85062};
    Vehicle vehicle;
    vehicle.setLength(Config::Lf);
    vehicle.update(-146.7283, 1.660802, 4.125825, 26.6806, 0, 0);
    std::vector<double> tx, ty;
    std::vector<double> vars = mpc.run(vehicle, ptsx, ptsy, &tx, &ty);
    printf("run %.12g %.12g %.12g %.12g %.12g %.12g %.12g %.12g traj %zu %.12g\n", vars[0], vars[1], vars[2], vars[3],
           vars[4] * Config::maxSteering, vars[5], vars[6], vars[7], tx.size(), tx.size() > 1 ? tx[1] : 0.0);
    printf("ptsx0 %.9g ptsy0 %.9g yaw %.12g %.12g\n", ptsx[0], ptsy[0], Config::yawLow, Config::yawHigh);
    std::vector<double> state = {vars[0], vars[1], vars[2], vars[3], vars[6], vars[7]};
    for (int i = 0; i < iters; i++) {
      std::vector<double> v = mpc.solve(state, 40);
      printf("solve %d %.12g %.12g %.12g %.12g %.12g %.12g %.12g %.12g %.12g\n", i, v[0], v[1], v[2], v[3], v[4], v[5], v[6],
             v[7], v[8]);
      for (int k = 0; k < 6; k++) state[k] = v[k];
    }
  } catch (const std::string &e) {
    fprintf(stderr, "error: %s\n", e.c_str());
    return e.find("NO_DEVICE") != std::string::npos || e.find("no HIP device") != std::string::npos || e.find("mpc_create") != std::string::npos ? 3 : 1;
  }
  return 0;
}
