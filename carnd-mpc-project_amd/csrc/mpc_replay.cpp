/*
 * mpc_replay -- replays simulator frames through the device handler.
 *
 * Stands where the reference's uWS server stands (src/mpc_main.cpp:81-222), without the socket: frames are read from
 * stdin, one per line, exactly as the simulator sends them (`42["telemetry",{...}]`); the replies the reference would
 * send (`42["steer",{...}]`, `42["manual",{}]`) are written to stdout, one per line, nothing for frames it ignores.
 *     mpc_replay <config.json> [--cars B] [--extra-latency seconds]
 * With --cars B the input is B interleaved connections: line i belongs to car i mod B, and each group of B lines is
 * solved as ONE batch on the device; every car keeps the throttle of its own previous reply, as the reference's
 * handler does in a static (mpc_main.cpp:89-91).  The handler's running mean of its own compute time (:158,:178) is
 * replaced by the constant --extra-latency (default 0) so that a replay is reproducible.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "mpc_amd.h"

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: mpc_replay <config.json> [--cars B] [--extra-latency s]\n"); return 2; }
  int64_t cars = 1;
  double extra = 0.0;
  for (int i = 2; i + 1 < argc; i += 2) {
    if (!strcmp(argv[i], "--cars")) cars = atoll(argv[i + 1]);
    else if (!strcmp(argv[i], "--extra-latency")) extra = atof(argv[i + 1]);
  }
  if (cars < 1) cars = 1;
  MpcParams p;
  if (mpc_params_load_json(argv[1], &p) != MPC_OK) { fprintf(stderr, "cannot load %s\n", argv[1]); return 1; }
  MpcHandle *h = nullptr;
  if (mpc_create(&p, -1, cars, &h) != MPC_OK) { fprintf(stderr, "mpc_create: %s\n", mpc_last_error()); return 1; }
  std::vector<double> prev((size_t)cars, 0.0), cmd((size_t)(2 * cars));
  std::vector<int32_t> status((size_t)cars);
  std::vector<MpcWireTelemetry> tel;
  std::vector<int64_t> who;                       /* car of each telemetry frame of the current group */
  std::vector<std::string> replies;
  char buf[512];
  std::string line;
  int64_t n = 0;
  int rc = 0;
  auto flush_group = [&]() -> int {
    if (!tel.empty()) {
      std::vector<double> pt(tel.size());
      for (size_t k = 0; k < tel.size(); k++) pt[k] = prev[(size_t)who[k]];
      std::vector<double> c(2 * tel.size());
      std::vector<int32_t> st(tel.size());
      if (mpc_wire_telemetry_batch_host(h, (int64_t)tel.size(), tel.data(), pt.data(), extra, c.data(), st.data()) != MPC_OK) {
        fprintf(stderr, "mpc_wire_telemetry_batch_host: %s\n", mpc_last_error());
        return 1;
      }
      for (size_t k = 0; k < tel.size(); k++) {
        prev[(size_t)who[k]] = c[tel.size() + k];
        mpc_wire_format_steer(c[k], c[tel.size() + k], buf, sizeof(buf));
        replies[(size_t)who[k]] = buf;
      }
    }
    for (auto &r : replies) if (!r.empty()) std::cout << r << "\n";
    tel.clear(); who.clear();
    return 0;
  };
  replies.assign((size_t)cars, "");
  while (std::getline(std::cin, line)) {
    const int64_t car = n % cars;
    MpcWireTelemetry t;
    const int kind = mpc_wire_parse(line.c_str(), (int64_t)line.size(), &t);
    if (kind == MPC_WIRE_TELEMETRY) { tel.push_back(t); who.push_back(car); }
    else if (kind == MPC_WIRE_MANUAL) { mpc_wire_format_manual(buf, sizeof(buf)); replies[(size_t)car] = buf; }
    else if (kind < 0) { fprintf(stderr, "line %lld: malformed telemetry frame\n", (long long)(n + 1)); rc = 1; }
    n++;
    if (n % cars == 0) {
      if (flush_group()) { rc = 1; break; }
      replies.assign((size_t)cars, "");
    }
  }
  if (n % cars != 0 && rc == 0) rc = flush_group();
  mpc_destroy(h);
  return rc;
}
