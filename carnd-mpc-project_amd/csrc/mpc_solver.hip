/*
 * mpc_solver.hip -- gfx950 kernels and the C ABI of include/mpc_amd.h.
 *
 * Batched replacement of MPC::solve() (src/control/MPC.cpp:183-325): B
 * independent instances, one per lane (64 per wavefront), struct-of-arrays in
 * HBM.  No CPU fallback exists: without a gfx950 device every compute entry
 * point returns an error.
 */
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "mpc_core.h"
#include "mpc_run_core.h"

namespace {

thread_local std::string g_last_error;

#define MPC_HIP_CHECK(expr)                                                              \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      g_last_error = std::string(#expr) + ": " + hipGetErrorString(e_);                  \
      return MPC_ERR_HIP;                                                                \
    }                                                                                    \
  } while (0)

constexpr int kBlock = 64; /* one wavefront per workgroup: lanes never synchronise */
constexpr int64_t kLdsPerCu = 160 * 1024;

/*
 * One instance per lane.  Inputs/outputs are [quantity][instance] so that a wave's access to one
 * quantity is a single contiguous 512-byte transaction; the workspace is tiled per wave with fields
 * interleaved in pairs per lane (see mpc_core.h).  STAGING: every sweep double-buffers the next
 * stage's record into this workgroup's LDS with LDS-DMA (global_load_lds) while it computes.
 */
constexpr size_t kStagingLdsBytes = 2u * mpc::STG_SLOT_PAIRS * 64u * 16u;   /* 36 KB per wave */

#if defined(__HIP_DEVICE_COMPILE__)
#define MPC_WAVE_ANY(p) (__builtin_amdgcn_ballot_w64(p) != 0ull)   /* over the active lanes of the wave */
#else
#define MPC_WAVE_ANY(p) (p)
#endif

/* Persistent form: a wave does not own 64 fixed instances.  Every lane takes the next unsolved instance from a
 * global counter, solves it, writes its results and takes another one, until the counter passes the end; the solver
 * is a per-lane state machine (Solver::step), so the lanes of a wave may be on different instances in different
 * phases.  With a grid of ceil(B/64) waves this is the plain one-instance-per-lane launch.
 *
 * Two-phase solve (MpcTwoPhase): a wave lasts as long as its slowest instance, and two thirds of the instances are
 * done after ~10 iterations while the slowest need 25.  Phase A therefore PARKS an instance that is still running
 * after `pass_cut` passes (at a pass boundary in the DIR phase: 36 scalars + its current iterate, which stays where
 * it is) and appends it to a list; phase B, launched right behind on the same stream, is the same kernel taking its
 * work from that list: a lane copies the parked iterate into its own tile, restores the scalars and carries on.  The
 * arithmetic of an instance does not change (results are bitwise identical), but the unfinished third is re-packed
 * into dense waves: ~19 % fewer wave passes, and as much less workspace traffic.
 *
 * Exit: a lane stops asking once its counter has passed the end; the wave leaves when no lane holds an instance and
 * none can get one -- every pass either advances an instance (bounded by max_iter) or consumes the counter. */
struct MpcTwoPhase {
  int32_t *ctl;            /* [0] fresh counter, [1] number of parked instances, [2] resume counter */
  int32_t *list_inst;      /* parked: instance index */
  int32_t *list_src;       /* parked: wave * 64 + lane of the tile column that holds its iterate */
  double *park;            /* [PARK_N][ld_park] solver scalars, column = position in the list */
  int64_t ld_park;
  const double *src_ws;    /* phase B: workspace of phase A */
  int32_t pass_cut;        /* phase A: park after this many passes (0 = never) */
  int32_t resume;          /* 1 = phase B */
};

template <bool STAGING>
__global__ __launch_bounds__(kBlock) void mpc_solve_kernel(
    const MpcParams P, const int64_t B, const int64_t ld, const int64_t ldo, const double *__restrict__ state,
    const double *__restrict__ coeffs, const double *__restrict__ yaw_lo, const double *__restrict__ yaw_hi,
    const double *__restrict__ weights, double *__restrict__ out, double *__restrict__ traj,
    int32_t *__restrict__ status, int32_t *__restrict__ iters, double *__restrict__ wsbase,
    const int64_t tile_doubles, const MpcTwoPhase T) {
  extern __shared__ double smem[];
  using WS = mpc::TiledWorkspace<STAGING>;
  using SV = mpc::Solver<WS>;
  WS ws;
  ws.tile = (mpc::gdouble *)(wsbase + (int64_t)blockIdx.x * tile_doubles);
  ws.lane = threadIdx.x;
  ws.lbuf = (mpc::ldouble *)smem;
  SV S(P, ws);
  int64_t i = 0;
  bool have = false, more = true;      /* holds an instance / may still get one */
  int attempt = 0, it_total = 0, passes = 0;
  const int64_t n_work = T.resume ? (int64_t)T.ctl[1] : B;
  for (;;) {
    if (!have && more) {
      const int64_t pos = (int64_t)atomicAdd(T.ctl + (T.resume ? 2 : 0), 1);
      more = pos < n_work;
      if (more) {
        i = T.resume ? (int64_t)T.list_inst[pos] : pos;
        double st[6], cf[MPC_NCOEF], w[MPC_NW];
#pragma unroll
        for (int q = 0; q < 6; q++) st[q] = state[q * ld + i];
#pragma unroll
        for (int q = 0; q < MPC_NCOEF; q++) cf[q] = coeffs[q * ld + i];
        if (weights) {
#pragma unroll
          for (int q = 0; q < MPC_NW; q++) w[q] = weights[q * ld + i];
        } else {
#pragma unroll
          for (int q = 0; q < MPC_NW; q++) w[q] = P.weights[q];
        }
        const int s0 = S.setup(st, cf, yaw_lo[i], yaw_hi[i], w, !T.resume);
        if (T.resume) {
          /* bring the parked iterate over: the column (src wave, src lane) of phase A's workspace -> own column */
          const double *pk = T.park + pos;
          const int64_t lp = T.ld_park;
          S.unpark([pk, lp](int q) -> double { return pk[q * lp]; }, attempt, it_total);
          const int src = T.list_src[pos];
          WS wsrc = ws;
          wsrc.tile = (mpc::gdouble *)(T.src_ws + (int64_t)(src >> 6) * tile_doubles);
          wsrc.lane = src & 63;
          const int I = S.cur ? mpc::IT1 : mpc::IT0;
          for (int k = 0; k < P.N - 1; ++k) {
#pragma unroll
            for (int f = 0; f < mpc::IT_SZ; f += 2) ws.store2(k, I, f, wsrc.it(k, I, f), wsrc.it(k, I, f + 1));
          }
          passes = 0; have = true;
        } else if (s0 == MPC_STATUS_SUCCESS) { S.begin(true); attempt = 0; it_total = 0; passes = 0; have = true; }
        else {
          /* rejected at set-up (initial state outside its own bounds): report the start point, ask again */
          double *o = out + i;
          double *t = traj ? traj + i : nullptr;
          const int64_t l = ldo;
          S.unpack([o, l](int q) -> double & { return o[q * l]; }, [t, l](int q) -> double & { return t[q * l]; }, traj != nullptr);
          status[i] = s0;
          if (iters) iters[i] = 0;
        }
      }
    }
    if (!MPC_WAVE_ANY(have || more)) break;
    if (have) {
      const int r = S.step();
      ++passes;
      if (r != SV::MPC_RUNNING) {
        if (r == MPC_STATUS_LINESEARCH && attempt == 0) {
          /* the stand-in for IPOPT's restoration phase: once more from the start point, zero multipliers */
          attempt = 1; it_total += S.iters;
          S.start_point();
          S.begin(false);
        } else {
          double *o = out + i;
          double *t = traj ? traj + i : nullptr;
          const int64_t l = ldo;
          S.unpack([o, l](int q) -> double & { return o[q * l]; }, [t, l](int q) -> double & { return t[q * l]; }, traj != nullptr);
          status[i] = r;
          if (iters) iters[i] = S.iters + it_total;
          have = false;
        }
      } else if (T.pass_cut > 0 && passes >= T.pass_cut && S.phase == SV::PH_DIR) {
        /* still running: park it for phase B */
        const int64_t pos = (int64_t)atomicAdd(T.ctl + 1, 1);
        T.list_inst[pos] = (int32_t)i;
        T.list_src[pos] = (int32_t)(blockIdx.x * 64u + threadIdx.x);
        double *pk = T.park + pos;
        const int64_t lp = T.ld_park;
        S.park([pk, lp](int q) -> double & { return pk[q * lp]; }, attempt, it_total);
        have = false; more = false;     /* the column keeps the parked iterate: this lane takes nothing else */
      }
    }
  }
}

/* MPC::run pre-processing, one instance per lane (mpc_run_core.h).  rows of `pre`: state 0..5, coeffs 6..10,
 * yaw_lo 11, yaw_hi 12, max_yaw_change 13, target_speed 14 */
template <bool TELEMETRY>
__global__ __launch_bounds__(256) void mpc_run_pre_kernel(const MpcParams P, int64_t B, int64_t ld, int npts,
                                                          const double *__restrict__ pose, double extra, double *__restrict__ ptsx,
                                                          double *__restrict__ ptsy, double *__restrict__ pre, int64_t ldp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  double po[6], px[mpc::RUN_MAX_PTS], py[mpc::RUN_MAX_PTS];
#pragma unroll
  for (int q = 0; q < 6; q++) po[q] = pose[q * ld + i];
  if (TELEMETRY) {   /* rows are the simulator's telemetry: latency compensation first (mpc_main.cpp:126-159) */
    double t6[6];
#pragma unroll
    for (int q = 0; q < 6; q++) t6[q] = po[q];
    mpc::telemetry_to_pose(P, t6, extra, po);
  }
#pragma unroll
  for (int q = 0; q < mpc::RUN_MAX_PTS; q++) { px[q] = q < npts ? ptsx[q * ld + i] : 0.0; py[q] = q < npts ? ptsy[q * ld + i] : 0.0; }
  mpc::RunPre R;
  mpc::run_pre(P, po, px, py, npts, R);
#pragma unroll
  for (int q = 0; q < mpc::RUN_MAX_PTS; q++) if (q < npts) { ptsx[q * ld + i] = px[q]; ptsy[q * ld + i] = py[q]; }
#pragma unroll
  for (int q = 0; q < 6; q++) pre[q * ldp + i] = R.state[q];
#pragma unroll
  for (int q = 0; q < 5; q++) pre[(6 + q) * ldp + i] = R.coef[q];
  pre[11 * ldp + i] = R.yaw_lo; pre[12 * ldp + i] = R.yaw_hi; pre[13 * ldp + i] = R.max_yaw_change; pre[14 * ldp + i] = R.target_speed;
}

__global__ __launch_bounds__(256) void mpc_run_post_kernel(const MpcParams P, int64_t B, const double *__restrict__ pre, int64_t ldp,
                                                           const double *__restrict__ out9, int64_t ld9, double *__restrict__ out8,
                                                           double *__restrict__ cmd, int64_t ld) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  double r9[9], o8[8];
#pragma unroll
  for (int q = 0; q < 9; q++) r9[q] = out9[q * ld9 + i];
  mpc::run_post(P, pre[13 * ldp + i], pre[14 * ldp + i], pre[3 * ldp + i], r9, o8);
  if (out8) {
#pragma unroll
    for (int q = 0; q < 8; q++) out8[q * ld + i] = o8[q];
  }
  if (cmd) {         /* the reply of the telemetry handler (mpc_main.cpp:171-174) */
    double sc, tc;
    mpc::command_from_run(P, o8, &sc, &tc);
    cmd[i] = sc; cmd[ld + i] = tc;
  }
}

/* rollout bookkeeping: next state <- solve()'s step-1 rows; worst status and summed iterations per instance */
__global__ __launch_bounds__(256) void mpc_rollout_step_kernel(int64_t B, int64_t ld, int first, const double *__restrict__ out9,
                                                               double *__restrict__ state, const int32_t *__restrict__ st_step,
                                                               const int32_t *__restrict__ it_step, int32_t *__restrict__ status,
                                                               int32_t *__restrict__ iters) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
#pragma unroll
  for (int q = 0; q < 6; q++) state[q * ld + i] = out9[q * ld + i];
  const int32_t s = st_step[i];
  status[i] = first ? s : (s > status[i] ? s : status[i]);
  if (iters) iters[i] = (first ? 0 : iters[i]) + it_step[i];
}

__global__ void mpc_debug_math_kernel(int64_t n, const double *x, double *sn, double *cs, double *rc, double *at, double *lg) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s, c;
  mpc::fsincos(x[i], &s, &c);
  sn[i] = s; cs[i] = c; rc[i] = mpc::frcp(x[i]);
  at[i] = mpc::fatan(x[i]); lg[i] = mpc::flog(fabs(x[i]));
}

}  // namespace

struct MpcHandle {
  MpcParams params;
  int device = 0;
  int64_t max_batch = 0;
  int64_t ws_stride = 0;   /* doubles per wavefront tile of the workspace */
  bool staging = true;
  int64_t io_stride = 0;   /* leading dimension of the handle's own staging arrays */
  double *ws = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  /* device staging for the host-pointer entry point and for statistics */
  double *d_in = nullptr;     /* state[6] coeffs[5] ylo yhi weights[12] = 25 rows */
  double *d_out = nullptr;    /* out[9] traj[2N] */
  double *d_run = nullptr;    /* run(): pre[15] rows */
  double *d_run9 = nullptr;   /* run(): solve()'s 9 rows, caller's leading dimension */
  int64_t run9_ld = 0;
  int32_t *d_status = nullptr, *d_iters = nullptr, *d_rstat = nullptr, *d_counter = nullptr;
  int inst_per_lane = 1;      /* MPC_INSTANCES_PER_LANE: waves = ceil(B / 64 / inst_per_lane) */
  /* two-phase solve: second workspace, parked-instance list and scalars (allocated on first use) */
  int pass_cut = 0;           /* MpcParams.pass_cut, or MPC_PASS_CUT in the environment (0 = single launch) */
  int64_t two_phase_min = 8192;
  double *ws2 = nullptr, *d_park = nullptr;
  int32_t *d_list = nullptr;
  /* last call */
  int64_t last_B = 0;
  const int32_t *last_status = nullptr, *last_iters = nullptr;
  bool timed = false;
};

static int validate_params(const MpcParams *p) {
  if (!p) return MPC_ERR_INVALID;
  if (p->abi_version != MPC_ABI_VERSION) { g_last_error = "MpcParams.abi_version mismatch"; return MPC_ERR_INVALID; }
  if (p->N < 3 || p->N > MPC_MAX_N) { g_last_error = "N out of range"; return MPC_ERR_INVALID; }
  if (!(p->dt > 0) || !(p->Lf > 0) || !(p->max_speed > 0) || !(p->max_steering > 0)) { g_last_error = "bad dt/Lf/limits"; return MPC_ERR_INVALID; }
  if (p->n_steers < 0 || p->n_steers > MPC_MAX_TABLE || p->n_steer_speeds < 1 || p->n_steer_speeds > MPC_MAX_TABLE) { g_last_error = "bad steer tables"; return MPC_ERR_INVALID; }
  if (p->branch_mode != MPC_BRANCH_FROZEN) { g_last_error = "branch_mode LIVE is not implemented on the device path"; return MPC_ERR_UNSUPPORTED; }
  if (p->precision != MPC_PRECISION_F64) { g_last_error = "precision F32 is not implemented yet"; return MPC_ERR_UNSUPPORTED; }
  if (p->max_iter < 1 || !(p->tol > 0)) { g_last_error = "bad max_iter/tol"; return MPC_ERR_INVALID; }
  return MPC_OK;
}

extern "C" int mpc_abi_version(void) { return MPC_ABI_VERSION; }
extern "C" const char *mpc_last_error(void) { return g_last_error.c_str(); }

extern "C" int mpc_create(const MpcParams *p, int device, int64_t max_batch, MpcHandle **out) {
  if (!out || max_batch < 1) return MPC_ERR_INVALID;
  int rc = validate_params(p);
  if (rc != MPC_OK) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_last_error = "no HIP device"; return MPC_ERR_NO_DEVICE; }
  if (device < 0) MPC_HIP_CHECK(hipGetDevice(&device));
  if (device >= ndev) { g_last_error = "device index out of range"; return MPC_ERR_NO_DEVICE; }
  MPC_HIP_CHECK(hipSetDevice(device));
  hipDeviceProp_t prop;
  MPC_HIP_CHECK(hipGetDeviceProperties(&prop, device));
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
    g_last_error = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
    return MPC_ERR_NO_DEVICE;
  }
  MpcHandle *h = new MpcHandle();
  h->params = *p; h->device = device; h->max_batch = max_batch;
  /* Launch shape: each workgroup is one wave; the kernel needs all 512 registers, so at most one wave runs
   * per SIMD (4 per CU), and the 36 KB of staging LDS per wave fit four times into a CU's 160 KB.
   * MPC_STAGING=0 selects the variant with ordinary loads (for A/B measurements). */
  h->staging = true;
  if (const char *e = getenv("MPC_STAGING")) h->staging = atoi(e) != 0;
  (void)hipFuncSetAttribute((const void *)mpc_solve_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCu);
  h->ws_stride = mpc::workspace_fields_per_instance(p->N, false) * 64;   /* doubles per wavefront tile */
  h->io_stride = (max_batch + 63) / 64 * 64;
  const size_t ws_bytes = (size_t)h->ws_stride * (size_t)(h->io_stride / 64) * sizeof(double);
  auto fail = [&](hipError_t e, const char *what) { g_last_error = std::string(what) + ": " + hipGetErrorString(e); mpc_destroy(h); return MPC_ERR_HIP; };
  hipError_t e;
  if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
  if ((e = hipEventCreate(&h->ev0)) != hipSuccess) return fail(e, "hipEventCreate");
  if ((e = hipEventCreate(&h->ev1)) != hipSuccess) return fail(e, "hipEventCreate");
  if ((e = hipMalloc((void **)&h->ws, ws_bytes)) != hipSuccess) return fail(e, "hipMalloc(workspace)");
  if ((e = hipMalloc((void **)&h->d_status, sizeof(int32_t) * h->io_stride)) != hipSuccess) return fail(e, "hipMalloc");
  if ((e = hipMalloc((void **)&h->d_iters, sizeof(int32_t) * h->io_stride)) != hipSuccess) return fail(e, "hipMalloc");
  if ((e = hipMalloc((void **)&h->d_counter, 4 * sizeof(int32_t))) != hipSuccess) return fail(e, "hipMalloc");
  h->pass_cut = p->pass_cut > 0 ? p->pass_cut : 0;
  if (const char *e3 = getenv("MPC_PASS_CUT")) { h->pass_cut = atoi(e3); if (h->pass_cut < 0) h->pass_cut = 0; }
  if (const char *e2 = getenv("MPC_INSTANCES_PER_LANE")) { h->inst_per_lane = atoi(e2); if (h->inst_per_lane < 1) h->inst_per_lane = 1; }
  *out = h;
  return MPC_OK;
}

extern "C" int mpc_set_params(MpcHandle *h, const MpcParams *p) {
  if (!h) return MPC_ERR_INVALID;
  int rc = validate_params(p);
  if (rc != MPC_OK) return rc;
  if (p->N != h->params.N) { g_last_error = "N cannot change on a live handle (workspace is sized by N)"; return MPC_ERR_INVALID; }
  h->params = *p;
  return MPC_OK;
}

extern "C" void mpc_destroy(MpcHandle *h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->ws) (void)hipFree(h->ws);
  if (h->d_in) (void)hipFree(h->d_in);
  if (h->d_out) (void)hipFree(h->d_out);
  if (h->d_run) (void)hipFree(h->d_run);
  if (h->d_run9) (void)hipFree(h->d_run9);
  if (h->d_status) (void)hipFree(h->d_status);
  if (h->d_iters) (void)hipFree(h->d_iters);
  if (h->d_rstat) (void)hipFree(h->d_rstat);
  if (h->d_counter) (void)hipFree(h->d_counter);
  if (h->ws2) (void)hipFree(h->ws2);
  if (h->d_park) (void)hipFree(h->d_park);
  if (h->d_list) (void)hipFree(h->d_list);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

/* the launch; ld = leading dimension of the inputs, ldo = of out/traj */
static int launch_solve(MpcHandle *h, int64_t B, int64_t ld, int64_t ldo, const double *state, const double *coeffs,
                        const double *yaw_lo, const double *yaw_hi, const double *weights, double *out, double *traj,
                        int32_t *status, int32_t *iters, void *stream_) {
  if (!h) { g_last_error = "NULL handle"; return MPC_ERR_INVALID; }
  if (B < 0 || ld < B || ldo < B) { g_last_error = "ld < B"; return MPC_ERR_INVALID; }
  if (B > h->max_batch) { g_last_error = "B exceeds the handle's max_batch"; return MPC_ERR_INVALID; }
  h->last_B = B; h->last_status = status; h->last_iters = iters ? iters : h->d_iters; h->timed = false;
  if (B == 0) return MPC_OK;   /* empty batch: nothing to read or write, pointers may be NULL */
  if (!state || !coeffs || !yaw_lo || !yaw_hi || !out || !status) { g_last_error = "NULL argument"; return MPC_ERR_INVALID; }
  hipStream_t s = (hipStream_t)stream_;   /* NULL = HIP's default (null) stream, exactly as passed */
  /* waves: one lane per instance, or fewer waves whose lanes take several instances in turn (instances_per_lane) */
  const int64_t waves_full = (B + kBlock - 1) / kBlock;
  int64_t waves = (waves_full + h->inst_per_lane - 1) / h->inst_per_lane;
  if (waves < 1) waves = 1;
  const bool two = h->pass_cut > 0 && h->inst_per_lane == 1 && B >= h->two_phase_min;
  if (two && !h->ws2) {
    const size_t ws_bytes = (size_t)h->ws_stride * (size_t)(h->io_stride / 64) * sizeof(double);
    MPC_HIP_CHECK(hipMalloc((void **)&h->ws2, ws_bytes));
    MPC_HIP_CHECK(hipMalloc((void **)&h->d_park, sizeof(double) * 36 * h->io_stride));
    MPC_HIP_CHECK(hipMalloc((void **)&h->d_list, sizeof(int32_t) * 2 * h->io_stride));
  }
  MpcTwoPhase T;
  T.ctl = h->d_counter; T.list_inst = h->d_list; T.list_src = h->d_list ? h->d_list + h->io_stride : nullptr;
  T.park = h->d_park; T.ld_park = h->io_stride; T.src_ws = h->ws; T.pass_cut = two ? h->pass_cut : 0; T.resume = 0;
  int32_t *it_out = iters ? iters : h->d_iters;
  MPC_HIP_CHECK(hipMemsetAsync(h->d_counter, 0, 4 * sizeof(int32_t), s));
  MPC_HIP_CHECK(hipEventRecord(h->ev0, s));
  auto launch = [&](unsigned grid, double *wsp, const MpcTwoPhase &tp) {
    if (h->staging)
      hipLaunchKernelGGL(mpc_solve_kernel<true>, dim3(grid), dim3(kBlock), kStagingLdsBytes, s, h->params, B, ld, ldo, state, coeffs,
                         yaw_lo, yaw_hi, weights, out, traj, status, it_out, wsp, h->ws_stride, tp);
    else
      hipLaunchKernelGGL(mpc_solve_kernel<false>, dim3(grid), dim3(kBlock), 0, s, h->params, B, ld, ldo, state, coeffs,
                         yaw_lo, yaw_hi, weights, out, traj, status, it_out, wsp, h->ws_stride, tp);
  };
  launch((unsigned)waves, h->ws, T);
  if (two) {
    /* phase B: the parked instances, re-packed; its grid covers half the batch, and lanes take further work from the
     * list if more than that was parked */
    MPC_HIP_CHECK(hipGetLastError());
    MpcTwoPhase R = T;
    R.resume = 1; R.pass_cut = 0;
    launch((unsigned)((waves + 1) / 2), h->ws2, R);
  }
  MPC_HIP_CHECK(hipGetLastError());
  MPC_HIP_CHECK(hipEventRecord(h->ev1, s));
  h->timed = true;
  return MPC_OK;
}

extern "C" int mpc_solve_batch_device(MpcHandle *h, int64_t B, int64_t ld, const double *state,
                                      const double *coeffs, const double *yaw_lo, const double *yaw_hi,
                                      const double *weights, double *out, double *traj, int32_t *status,
                                      int32_t *iters, void *stream_) {
  return launch_solve(h, B, ld, ld, state, coeffs, yaw_lo, yaw_hi, weights, out, traj, status, iters, stream_);
}

/* run() for a batch; `tel` selects the telemetry rows as input (with latency compensation) and `cmd` the reply */
static int run_impl(MpcHandle *h, int64_t B, int64_t ld, int npts, const double *pose, bool tel, double extra, double *ptsx,
                    double *ptsy, double *out8, double *cmd, double *traj, int32_t *status, int32_t *iters, double *pre,
                    void *stream_) {
  if (!h) { g_last_error = "NULL handle"; return MPC_ERR_INVALID; }
  if (B < 0 || ld < B || B > h->max_batch) { g_last_error = "bad B/ld"; return MPC_ERR_INVALID; }
  if (npts < 3 || npts > mpc::RUN_MAX_PTS) { g_last_error = "npts must be 3..8"; return MPC_ERR_INVALID; }
  if (B == 0) { h->last_B = 0; return MPC_OK; }
  if (!pose || !ptsx || !ptsy || !(out8 || cmd) || !status) { g_last_error = "NULL argument"; return MPC_ERR_INVALID; }
  MPC_HIP_CHECK(hipSetDevice(h->device));
  const int64_t S = h->io_stride;
  if (!h->d_run) MPC_HIP_CHECK(hipMalloc((void **)&h->d_run, sizeof(double) * 15 * S));
  if (!h->d_run9 || h->run9_ld < ld) {       /* solve()'s 9 rows, with the caller's leading dimension (traj shares it) */
    if (h->d_run9) MPC_HIP_CHECK(hipFree(h->d_run9));
    h->d_run9 = nullptr;
    MPC_HIP_CHECK(hipMalloc((void **)&h->d_run9, sizeof(double) * 9 * ld));
    h->run9_ld = ld;
  }
  hipStream_t s = (hipStream_t)stream_;
  double *d_pre = h->d_run;
  const unsigned grid = (unsigned)((B + 255) / 256);
  if (tel) hipLaunchKernelGGL(mpc_run_pre_kernel<true>, dim3(grid), dim3(256), 0, s, h->params, B, ld, npts, pose, extra, ptsx, ptsy, d_pre, S);
  else hipLaunchKernelGGL(mpc_run_pre_kernel<false>, dim3(grid), dim3(256), 0, s, h->params, B, ld, npts, pose, 0.0, ptsx, ptsy, d_pre, S);
  MPC_HIP_CHECK(hipGetLastError());
  int rc = launch_solve(h, B, S, ld, d_pre, d_pre + 6 * S, d_pre + 11 * S, d_pre + 12 * S, nullptr, h->d_run9, traj, status, iters, stream_);
  if (rc != MPC_OK) return rc;
  hipLaunchKernelGGL(mpc_run_post_kernel, dim3(grid), dim3(256), 0, s, h->params, B, d_pre, S, h->d_run9, ld, out8, cmd, ld);
  MPC_HIP_CHECK(hipGetLastError());
  if (pre) MPC_HIP_CHECK(hipMemcpy2DAsync(pre, sizeof(double) * ld, d_pre, sizeof(double) * S, sizeof(double) * B, 15, hipMemcpyDeviceToDevice, s));
  return MPC_OK;
}

extern "C" int mpc_run_batch_device(MpcHandle *h, int64_t B, int64_t ld, int npts, const double *pose, double *ptsx,
                                    double *ptsy, double *out8, double *traj, int32_t *status, int32_t *iters,
                                    double *pre, void *stream_) {
  if (h && B > 0 && !out8) { g_last_error = "NULL argument"; return MPC_ERR_INVALID; }
  return run_impl(h, B, ld, npts, pose, false, 0.0, ptsx, ptsy, out8, nullptr, traj, status, iters, pre, stream_);
}

extern "C" int mpc_telemetry_batch_device(MpcHandle *h, int64_t B, int64_t ld, int npts, const double *tel, double extra_latency,
                                          double *ptsx, double *ptsy, double *cmd, double *out8, int32_t *status, void *stream_) {
  if (h && B > 0 && !cmd) { g_last_error = "NULL argument"; return MPC_ERR_INVALID; }
  return run_impl(h, B, ld, npts, tel, true, extra_latency, ptsx, ptsy, out8, cmd, nullptr, status, nullptr, nullptr, stream_);
}

extern "C" int mpc_rollout_batch_device(MpcHandle *h, int64_t B, int64_t ld, int steps, double *state, const double *coeffs,
                                        const double *yaw_lo, const double *yaw_hi, const double *weights, double *hist,
                                        int32_t *status, int32_t *iters, void *stream_) {
  if (!h) { g_last_error = "NULL handle"; return MPC_ERR_INVALID; }
  if (B < 0 || ld < B || B > h->max_batch) { g_last_error = "bad B/ld"; return MPC_ERR_INVALID; }
  if (steps < 1) { g_last_error = "steps < 1"; return MPC_ERR_INVALID; }
  if (B == 0) { h->last_B = 0; return MPC_OK; }
  if (!state || !coeffs || !yaw_lo || !yaw_hi || !status) { g_last_error = "NULL argument"; return MPC_ERR_INVALID; }
  MPC_HIP_CHECK(hipSetDevice(h->device));
  if (!hist && (!h->d_run9 || h->run9_ld < ld)) {
    if (h->d_run9) MPC_HIP_CHECK(hipFree(h->d_run9));
    h->d_run9 = nullptr;
    MPC_HIP_CHECK(hipMalloc((void **)&h->d_run9, sizeof(double) * 9 * ld));
    h->run9_ld = ld;
  }
  if (!h->d_rstat) MPC_HIP_CHECK(hipMalloc((void **)&h->d_rstat, sizeof(int32_t) * h->io_stride));
  hipStream_t s = (hipStream_t)stream_;
  const unsigned grid = (unsigned)((B + 255) / 256);
  for (int t = 0; t < steps; t++) {
    double *o9 = hist ? hist + (int64_t)t * 9 * ld : h->d_run9;
    int rc = launch_solve(h, B, ld, ld, state, coeffs, yaw_lo, yaw_hi, weights, o9, nullptr, h->d_rstat, h->d_iters, stream_);
    if (rc != MPC_OK) return rc;
    hipLaunchKernelGGL(mpc_rollout_step_kernel, dim3(grid), dim3(256), 0, s, B, ld, t == 0, o9, state, h->d_rstat, h->d_iters, status, iters);
    MPC_HIP_CHECK(hipGetLastError());
  }
  h->last_status = status; h->last_iters = iters ? iters : h->d_iters;
  return MPC_OK;
}

extern "C" int mpc_synchronize(MpcHandle *h) {
  if (!h) return MPC_ERR_INVALID;
  MPC_HIP_CHECK(hipStreamSynchronize(h->stream));
  return MPC_OK;
}

extern "C" int mpc_solve_batch_host(MpcHandle *h, int64_t B, int64_t ld, const double *state,
                                    const double *coeffs, const double *yaw_lo, const double *yaw_hi,
                                    const double *weights, double *out, double *traj, int32_t *status,
                                    int32_t *iters) {
  if (!h) { g_last_error = "NULL handle"; return MPC_ERR_INVALID; }
  if (B < 0 || ld < B || B > h->max_batch) { g_last_error = "bad B/ld"; return MPC_ERR_INVALID; }
  if (B == 0) { h->last_B = 0; return MPC_OK; }
  if (!state || !coeffs || !yaw_lo || !yaw_hi || !out || !status) { g_last_error = "NULL argument"; return MPC_ERR_INVALID; }
  MPC_HIP_CHECK(hipSetDevice(h->device));
  const int N = h->params.N;
  const int64_t S = h->io_stride;
  if (!h->d_in) MPC_HIP_CHECK(hipMalloc((void **)&h->d_in, sizeof(double) * 25 * S));
  if (!h->d_out) MPC_HIP_CHECK(hipMalloc((void **)&h->d_out, sizeof(double) * (9 + 2 * MPC_MAX_N) * S));
  double *d_state = h->d_in, *d_coef = d_state + 6 * S, *d_ylo = d_coef + 5 * S, *d_yhi = d_ylo + S, *d_w = d_yhi + S;
  double *d_o = h->d_out, *d_t = d_o + 9 * S;
  hipStream_t s = h->stream;
  for (int q = 0; q < 6; q++) MPC_HIP_CHECK(hipMemcpyAsync(d_state + q * S, state + q * ld, sizeof(double) * B, hipMemcpyHostToDevice, s));
  for (int q = 0; q < 5; q++) MPC_HIP_CHECK(hipMemcpyAsync(d_coef + q * S, coeffs + q * ld, sizeof(double) * B, hipMemcpyHostToDevice, s));
  MPC_HIP_CHECK(hipMemcpyAsync(d_ylo, yaw_lo, sizeof(double) * B, hipMemcpyHostToDevice, s));
  MPC_HIP_CHECK(hipMemcpyAsync(d_yhi, yaw_hi, sizeof(double) * B, hipMemcpyHostToDevice, s));
  if (weights) for (int q = 0; q < MPC_NW; q++) MPC_HIP_CHECK(hipMemcpyAsync(d_w + q * S, weights + q * ld, sizeof(double) * B, hipMemcpyHostToDevice, s));
  int rc = mpc_solve_batch_device(h, B, S, d_state, d_coef, d_ylo, d_yhi, weights ? d_w : nullptr, d_o,
                                  traj ? d_t : nullptr, h->d_status, h->d_iters, (void *)s);
  if (rc != MPC_OK) return rc;
  for (int q = 0; q < 9; q++) MPC_HIP_CHECK(hipMemcpyAsync(out + q * ld, d_o + q * S, sizeof(double) * B, hipMemcpyDeviceToHost, s));
  if (traj) for (int q = 0; q < 2 * N; q++) MPC_HIP_CHECK(hipMemcpyAsync(traj + q * ld, d_t + q * S, sizeof(double) * B, hipMemcpyDeviceToHost, s));
  MPC_HIP_CHECK(hipMemcpyAsync(status, h->d_status, sizeof(int32_t) * B, hipMemcpyDeviceToHost, s));
  if (iters) MPC_HIP_CHECK(hipMemcpyAsync(iters, h->d_iters, sizeof(int32_t) * B, hipMemcpyDeviceToHost, s));
  MPC_HIP_CHECK(hipStreamSynchronize(s));
  h->last_status = h->d_status; h->last_iters = h->d_iters;
  return MPC_OK;
}

extern "C" int mpc_get_stats(MpcHandle *h, MpcBatchStats *st) {
  if (!h || !st) return MPC_ERR_INVALID;
  memset(st, 0, sizeof(*st));
  st->batch = h->last_B;
  if (h->last_B == 0 || !h->last_status) return MPC_OK;
  MPC_HIP_CHECK(hipSetDevice(h->device));
  MPC_HIP_CHECK(hipDeviceSynchronize());
  std::vector<int32_t> s(h->last_B), it(h->last_B);
  MPC_HIP_CHECK(hipMemcpy(s.data(), h->last_status, sizeof(int32_t) * h->last_B, hipMemcpyDeviceToHost));
  MPC_HIP_CHECK(hipMemcpy(it.data(), h->last_iters, sizeof(int32_t) * h->last_B, hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < h->last_B; i++) {
    switch (s[i]) {
      case MPC_STATUS_SUCCESS: st->n_success++; break;
      case MPC_STATUS_MAXITER: st->n_maxiter++; break;
      case MPC_STATUS_LINESEARCH: st->n_linesearch++; break;
      case MPC_STATUS_INFEASIBLE: st->n_infeasible++; break;
      default: st->n_numeric++; break;
    }
    st->iter_sum += it[i];
    if (it[i] > st->iter_max) st->iter_max = it[i];
  }
  if (h->timed) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) st->kernel_ms = ms;
  }
  return MPC_OK;
}

extern "C" int mpc_debug_math_ext(int device, int64_t n, const double *x, double *sn, double *cs, double *rc, double *at, double *lg) {
  if (n < 0 || (n > 0 && (!x || !sn || !cs || !rc || !at || !lg))) return MPC_ERR_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_last_error = "no HIP device"; return MPC_ERR_NO_DEVICE; }
  if (device >= 0) MPC_HIP_CHECK(hipSetDevice(device));
  if (n == 0) return MPC_OK;
  double *d = nullptr;
  MPC_HIP_CHECK(hipMalloc((void **)&d, sizeof(double) * 6 * n));
  hipError_t e = hipMemcpy(d, x, sizeof(double) * n, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(mpc_debug_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, n, d, d + n, d + 2 * n, d + 3 * n, d + 4 * n, d + 5 * n);
    e = hipGetLastError();
  }
  double *outs[5] = {sn, cs, rc, at, lg};
  for (int q = 0; q < 5 && e == hipSuccess; q++) e = hipMemcpy(outs[q], d + (q + 1) * n, sizeof(double) * n, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) { g_last_error = std::string("mpc_debug_math: ") + hipGetErrorString(e); return MPC_ERR_HIP; }
  return MPC_OK;
}

extern "C" int mpc_debug_math(int device, int64_t n, const double *x, double *sn, double *cs, double *rc) {
  if (n < 0) return MPC_ERR_INVALID;
  std::vector<double> at((size_t)n), lg((size_t)n);
  return mpc_debug_math_ext(device, n, x, sn, cs, rc, at.data(), lg.data());
}
