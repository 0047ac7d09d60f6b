/*
 * mpc_wire.cpp -- the wire side of the telemetry handler.
 *
 * The reference talks to the Udacity simulator over a WebSocket with Socket.IO-style text frames
 * (src/mpc_main.cpp:81-222, DATA.md:5-16):
 *     in :  42["telemetry",{"ptsx":[..],"ptsy":[..],"psi":..,"x":..,"y":..,"steering_angle":..,"throttle":..,"speed":..}]
 *     out:  42["steer",{"mpc_x":0,"mpc_y":0,"next_x":0,"next_y":0,"steering_angle":S,"throttle":T}]     (:183-197)
 *           42["manual",{}]                                              when the frame carries no data (:217-219)
 * The uWS server itself (socket, event loop, the 100 ms sleep) is out of scope; what is built here is everything
 * between the bytes of a frame and the bytes of the reply, for B independent connections at once:
 *   mpc_wire_parse          hasData() (:26-36) + the field extraction of :110-124
 *   mpc_wire_format_steer   the reply, byte for byte as nlohmann::json 2.1.1 dumps it (keys in std::map order;
 *                           `msgJson["mpc_x"] = NULL` stores the integer 0; doubles as "%.15g", ".0" appended to
 *                           integer-looking values, 0 as "0.0"; src/utils/json.hpp:8306-8392)
 *   mpc_wire_telemetry_batch_host   parsed frames -> mpc_telemetry_batch_device -> (steering_angle, throttle)
 * Host-side C++ like the rest of the reference's server; the numbers come from the HIP path (no CPU fallback).
 */
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mpc_amd.h"

namespace {

/* value of "key": inside the flat telemetry object; keys are matched with their quotes, so "x" does not hit "ptsx" */
bool find_value(const std::string &s, const char *key, size_t &pos) {
  const std::string pat = std::string("\"") + key + "\"";
  size_t p = 0;
  for (;;) {
    p = s.find(pat, p);
    if (p == std::string::npos) return false;
    size_t q = p + pat.size();
    while (q < s.size() && (s[q] == ' ' || s[q] == '\t')) q++;
    if (q < s.size() && s[q] == ':') { pos = q + 1; return true; }
    p += pat.size();
  }
}
bool get_num(const std::string &s, const char *key, double &out) {
  size_t p;
  if (!find_value(s, key, p)) return false;
  const char *b = s.c_str() + p;
  char *end;
  const double v = strtod(b, &end);
  if (end == b) return false;
  out = v;
  return true;
}
int get_arr(const std::string &s, const char *key, double *out, int cap) {
  size_t p;
  if (!find_value(s, key, p)) return -1;
  while (p < s.size() && (s[p] == ' ' || s[p] == '\t')) p++;
  if (p >= s.size() || s[p] != '[') return -1;
  const char *c = s.c_str() + p + 1;
  int n = 0;
  for (;;) {
    while (*c == ' ' || *c == ',' || *c == '\n' || *c == '\t' || *c == '\r') c++;
    if (*c == ']' || !*c) break;
    char *end;
    const double v = strtod(c, &end);
    if (end == c) return -1;
    if (n < cap) out[n] = v;
    n++;
    c = end;
  }
  return n;
}

/* one double as nlohmann::json 2.1.1 writes it (json.hpp:8306-8392) */
std::string dump_double(double x) {
  if (!std::isfinite(x)) return "null";
  if (x == 0) return std::signbit(x) ? "-0.0" : "0.0";
  char buf[64];
  snprintf(buf, sizeof(buf), "%.*g", 15, x);
  std::string r(buf);
  if (r.find_first_of(".eE") == std::string::npos) r += ".0";
  return r;
}

}  // namespace

extern "C" int mpc_wire_parse(const char *frame, int64_t len, MpcWireTelemetry *out) {
  if (!frame || len < 0 || !out) return MPC_ERR_INVALID;
  memset(out, 0, sizeof(*out));
  const std::string sdata(frame, (size_t)len);
  if (!(sdata.size() > 2 && sdata[0] == '4' && sdata[1] == '2')) return MPC_WIRE_IGNORE;      /* mpc_main.cpp:87 */
  /* hasData(), mpc_main.cpp:26-36 */
  const size_t b1 = sdata.find_first_of("["), b2 = sdata.rfind("}]");
  if (sdata.find("null") != std::string::npos || b1 == std::string::npos || b2 == std::string::npos) return MPC_WIRE_MANUAL;
  const std::string s = sdata.substr(b1, b2 - b1 + 2);
  /* j[0]: the event name */
  size_t q0 = s.find('"');
  size_t q1 = q0 == std::string::npos ? q0 : s.find('"', q0 + 1);
  if (q1 == std::string::npos) return MPC_ERR_INVALID;
  if (s.compare(q0 + 1, q1 - q0 - 1, "telemetry") != 0) return MPC_WIRE_IGNORE;               /* :97 */
  const std::string obj = s.substr(q1 + 1);
  const int nx = get_arr(obj, "ptsx", out->ptsx, 8), ny = get_arr(obj, "ptsy", out->ptsy, 8);
  if (nx < 3 || nx > 8 || ny != nx) return MPC_ERR_INVALID;
  out->npts = nx;
  if (!get_num(obj, "x", out->x) || !get_num(obj, "y", out->y) || !get_num(obj, "psi", out->psi) ||
      !get_num(obj, "speed", out->speed) || !get_num(obj, "steering_angle", out->steering_angle))
    return MPC_ERR_INVALID;
  if (!get_num(obj, "throttle", out->throttle)) out->throttle = 0.0;      /* read only by the COLLECT_DATA build (:137) */
  return MPC_WIRE_TELEMETRY;
}

extern "C" int64_t mpc_wire_format_steer(double steering_angle, double throttle, char *buf, int64_t cap) {
  /* std::map key order; the four trajectory members are the integer 0 in the build without PLOT_TRAJECTORY (:191-194) */
  const std::string msg = "42[\"steer\",{\"mpc_x\":0,\"mpc_y\":0,\"next_x\":0,\"next_y\":0,\"steering_angle\":" +
                          dump_double(steering_angle) + ",\"throttle\":" + dump_double(throttle) + "}]";
  if (!buf || cap < (int64_t)msg.size() + 1) return MPC_ERR_INVALID;
  memcpy(buf, msg.c_str(), msg.size() + 1);
  return (int64_t)msg.size();
}

extern "C" int64_t mpc_wire_format_manual(char *buf, int64_t cap) {
  static const char kMsg[] = "42[\"manual\",{}]";                                             /* :218 */
  if (!buf || cap < (int64_t)sizeof(kMsg)) return MPC_ERR_INVALID;
  memcpy(buf, kMsg, sizeof(kMsg));
  return (int64_t)sizeof(kMsg) - 1;
}

/* B parsed frames (one per connection) through the device handler.  prev_throttle[i] is the throttle of connection i's
 * previous reply -- the reference keeps ONE function-local static for all connections (:89-91; it serves one simulator);
 * a value per connection is this tool's generalisation to many cars, 0 before a car's first message -- and extra_latency
 * is the mean handler time the reference adds to Config::lookahead (:158).  cmd is [2][B]: steering_angle row, throttle
 * row.  Runs on the handle's own device and stream (mpc_telemetry_batch_host), whatever the caller's current device. */
extern "C" void mpc_internal_set_error(const char *msg);
extern "C" int mpc_wire_telemetry_batch_host(MpcHandle *h, int64_t B, const MpcWireTelemetry *tel, const double *prev_throttle,
                                             double extra_latency, double *cmd, int32_t *status) {
  if (!h || B < 0 || (B > 0 && (!tel || !cmd || !status))) { mpc_internal_set_error("mpc_wire_telemetry_batch_host: NULL argument or B < 0"); return MPC_ERR_INVALID; }
  if (B == 0) return MPC_OK;
  const int npts = tel[0].npts;
  for (int64_t i = 0; i < B; i++)
    if (tel[i].npts != npts) {          /* one waypoint count per batch (the simulator always sends 6) */
      mpc_internal_set_error("mpc_wire_telemetry_batch_host: the frames of one batch must carry the same number of waypoints");
      return MPC_ERR_INVALID;
    }
  const int64_t rows = 6 + 2 * npts;
  std::vector<double> host((size_t)(rows * B));
  for (int64_t i = 0; i < B; i++) {
    const MpcWireTelemetry &t = tel[i];
    host[0 * B + i] = t.x; host[1 * B + i] = t.y; host[2 * B + i] = t.psi; host[3 * B + i] = t.speed;
    host[4 * B + i] = t.steering_angle; host[5 * B + i] = prev_throttle ? prev_throttle[i] : 0.0;
    for (int q = 0; q < npts; q++) { host[(6 + q) * B + i] = t.ptsx[q]; host[(6 + npts + q) * B + i] = t.ptsy[q]; }
  }
  return mpc_telemetry_batch_host(h, B, B, npts, host.data(), extra_latency, host.data() + 6 * B, host.data() + (6 + npts) * B, cmd, status);
}
