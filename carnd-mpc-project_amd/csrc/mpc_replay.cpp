/*
 * mpc_replay -- replays simulator frames through the device handler.
 *
 * Stands where the reference's uWS server stands (src/mpc_main.cpp:81-222), without the socket: frames are read from
 * stdin, one per line, exactly as the simulator sends them (`42["telemetry",{...}]`); the replies the reference would
 * send (`42["steer",{...}]`, `42["manual",{}]`) are written to stdout, one per line, nothing for frames it ignores.
 *     mpc_replay <config.json> [--cars B] [--extra-latency seconds] [--tcp PORT]
 * With --tcp PORT the frames come from ONE TCP connection on 127.0.0.1:PORT instead of stdin (newline-delimited text, not
 * the WebSocket protocol: a relay in front of the simulator would strip that), and the replies go back on the same socket.
 * With --cars B the input is B interleaved connections: line i belongs to car i mod B, and each group of B lines is
 * solved as ONE batch on the device; every car keeps the throttle of its own previous reply.  (The reference has ONE
 * function-local static for that, mpc_main.cpp:89-91, shared by whatever connects: it serves a single simulator.  A value
 * per car is this tool's generalisation, not parity; with --cars 1 it is the reference's behaviour.)  The replies of a
 * group are written, in car order, once the whole group has been read: a last group with fewer than B frames is flushed at
 * end of input (with --tcp: when the peer closes its side).  The handler's running mean of its own compute time
 * (:158,:178) is replaced by the constant --extra-latency (default 0) so that a replay is reproducible.
 */
#include <arpa/inet.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "mpc_amd.h"

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: mpc_replay <config.json> [--cars B] [--extra-latency s] [--tcp PORT]\n"); return 2; }
  int64_t cars = 1;
  double extra = 0.0;
  int tcp_port = 0;
  for (int i = 2; i < argc; i += 2) {
    const bool has_value = i + 1 < argc;
    if (has_value && !strcmp(argv[i], "--cars")) cars = atoll(argv[i + 1]);
    else if (has_value && !strcmp(argv[i], "--extra-latency")) extra = atof(argv[i + 1]);
    else if (has_value && !strcmp(argv[i], "--tcp")) tcp_port = atoi(argv[i + 1]);
    else { fprintf(stderr, "mpc_replay: unknown or incomplete option '%s'\nusage: mpc_replay <config.json> [--cars B] [--extra-latency s] [--tcp PORT]\n", argv[i]); return 2; }
  }
  if (cars < 1) { fprintf(stderr, "mpc_replay: --cars must be >= 1\n"); return 2; }
  /* frame source / reply sink: stdin/stdout, or one TCP connection */
  int conn = -1;
  std::string pending;
  if (tcp_port > 0) {
    const int srv = socket(AF_INET, SOCK_STREAM, 0);
    int one = 1;
    setsockopt(srv, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    sockaddr_in a;
    memset(&a, 0, sizeof(a));
    a.sin_family = AF_INET; a.sin_addr.s_addr = htonl(INADDR_LOOPBACK); a.sin_port = htons((uint16_t)tcp_port);
    if (srv < 0 || bind(srv, (sockaddr *)&a, sizeof(a)) != 0 || listen(srv, 1) != 0) { perror("mpc_replay --tcp"); return 1; }
    fprintf(stderr, "mpc_replay: listening on 127.0.0.1:%d\n", tcp_port);
    conn = accept(srv, nullptr, nullptr);
    close(srv);
    if (conn < 0) { perror("accept"); return 1; }
  }
  auto read_line = [&](std::string &out) -> bool {
    if (conn < 0) return (bool)std::getline(std::cin, out);
    for (;;) {
      const size_t nl = pending.find('\n');
      if (nl != std::string::npos) { out = pending.substr(0, nl); pending.erase(0, nl + 1); return true; }
      char tmp[4096];
      const ssize_t k = recv(conn, tmp, sizeof(tmp), 0);
      if (k <= 0) { if (pending.empty()) return false; out = pending; pending.clear(); return true; }
      pending.append(tmp, (size_t)k);
    }
  };
  auto write_line = [&](const std::string &r) {
    if (conn < 0) { std::cout << r << "\n"; return; }
    const std::string m = r + "\n";
    size_t off = 0;
    while (off < m.size()) { const ssize_t k = send(conn, m.data() + off, m.size() - off, 0); if (k <= 0) break; off += (size_t)k; }
  };
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
    for (auto &r : replies) if (!r.empty()) write_line(r);
    if (conn < 0) std::cout.flush();
    tel.clear(); who.clear();
    return 0;
  };
  replies.assign((size_t)cars, "");
  while (read_line(line)) {
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
  if (conn >= 0) close(conn);
  return rc;
}
