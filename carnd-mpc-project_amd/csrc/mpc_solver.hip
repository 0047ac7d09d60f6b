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
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
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

/* Every entry point works on the handle's device and leaves the caller's current device as it found it. */
struct DeviceGuard {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) err = hipSetDevice(dev);
  }
  ~DeviceGuard() {
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
  }
};
#define MPC_ON_DEVICE(h)                                                                 \
  DeviceGuard guard_((h)->device);                                                       \
  if (guard_.err != hipSuccess) {                                                        \
    g_last_error = std::string("hipSetDevice: ") + hipGetErrorString(guard_.err);        \
    return MPC_ERR_HIP;                                                                  \
  }

constexpr int kBlock = 64; /* one wavefront per workgroup: lanes never synchronise */
constexpr int64_t kLdsPerCu = 160 * 1024;
constexpr int kMaxCuts = 4;         /* cuts of the multi-phase solve: up to 5 launches per batch */
constexpr int kCounterInts = 16;    /* two counters per phase */
constexpr int kCounterRing = 32;    /* counter blocks: solve call n uses block n % 32 and zeroes block (n + 16) % 32 for its next user */
constexpr int kParkRows = 47;       /* Solver::PARK_N */
constexpr int kFinPromote = -2, kFinScratch = -3;   /* a lane that waits to hand its instance to the fp64 phase (promoted / to be solved from scratch) */
constexpr int kTailMaxRing = 512;   /* deferred tails: batches whose stragglers may be outstanding at once */
constexpr int kFreshRing = 24;      /* fresh queues: one per batch between its launch and the completion of the tail slice that absorbs its stragglers */
constexpr int kSliceMaxSrc = 1 + kFreshRing;   /* what one tail slice reads: the survivors of the slice before it + fresh queues */
constexpr int kSliceRing = 4;       /* slices whose events and result blocks are kept (at most two are in flight) */
constexpr int kTailInRows = 6 + MPC_NCOEF + 2 + MPC_NW;   /* the inputs of a deferred instance travel with it: 25 rows */
constexpr int kTailMetaRows = 8;    /* ... and where it belongs: instance, slot, batch id, out, traj, status, iters, ldo */
constexpr int kTailRows = kParkRows + kTailInRows + kTailMetaRows;

/* One queue of deferred instances: the fresh queue a launch hands its stragglers to, or the list of survivors a tail slice
 * leaves for the next one.  Entry e: column e of `park` ([kTailRows][cap]: Solver::park scalars, the instance's inputs, where
 * its results go) and lane e % 64 of tile e / 64 of `iter` ([cap / 64][N-1][IT_SZ][64] reals: its current iterate). */
struct MpcTailQ {
  int32_t *count;                  /* entries appended so far (may run past cap: min(count, cap) are valid) */
  int32_t cap;                     /* multiple of 64 */
  double *park;
  void *iter;
};

/*
 * One instance per lane.  Inputs/outputs are [quantity][instance] so that a wave's access to one
 * quantity is a single contiguous 512-byte transaction; the workspace is tiled per wave with fields
 * interleaved in pairs per lane (see mpc_core.h).  STAGING: every sweep double-buffers the next
 * stage's record into this workgroup's LDS with LDS-DMA (global_load_lds) while it computes.
 */
template <class R> constexpr size_t staging_lds_bytes() { return 2u * mpc::Fields<R>::STG_SLOT * 64u * 16u; }   /* 36 KB per wave (fp64), 20 KB (fp32) */

#if defined(__HIP_DEVICE_COMPILE__)
#define MPC_WAVE_ANY(p) (__builtin_amdgcn_ballot_w64(p) != 0ull)   /* over the active lanes of the wave */
#define MPC_WAVE_COUNT(p) __builtin_popcountll(__builtin_amdgcn_ballot_w64(p))
#else
#define MPC_WAVE_ANY(p) (p)
#define MPC_WAVE_COUNT(p) ((p) ? 1 : 0)
#endif

/* Persistent form: a wave does not own 64 fixed instances.  Every lane takes the next unsolved instance from a
 * global counter, solves it, writes its results and takes another one, until the counter passes the end; the solver
 * is a per-lane state machine (Solver::step), so the lanes of a wave may be on different instances in different
 * phases.  With a grid of ceil(B/64) waves this is the plain one-instance-per-lane launch.
 *
 * Multi-phase solve (MpcPhase): a wave lasts as long as its slowest instance, and two thirds of the instances are
 * done after ~10 iterations while the slowest need 25 (headline workload) or 100-200 (weight sweeps, long horizons).
 * A phase with a cut therefore PARKS an instance that is still running after `pass_cut` passes (at a pass boundary
 * in the DIR phase: 36 scalars + its current iterate, which stays where it is) and appends it to a list; the next
 * phase, launched right behind on the same stream, is the same kernel taking its work from that list: a lane copies
 * the parked iterate into its own tile, restores the scalars and carries on -- and may park it again if that phase
 * has a cut too.  The arithmetic of an instance does not change (results are bitwise identical), but the unfinished
 * part is re-packed into dense waves after every cut, so a wave slot is not held by one or two stragglers.  Lists,
 * scalars and workspaces alternate between two buffers (phase p reads what phase p-1 wrote).
 *
 * Exit: a lane stops asking once its counter has passed the end; the wave leaves when no lane holds an instance and
 * none can get one -- every pass either advances an instance (bounded by max_iter and the cut) or consumes the
 * counter.  A lane that has parked takes nothing else (its column keeps the iterate), so a phase with a cut needs as
 * many lanes as it may receive instances: the host launches every phase with the grid of the first. */
struct MpcPhase {
  int32_t *take;            /* counter this phase takes its work from */
  const int32_t *n_in;      /* resume: number of parked instances to take (written by the previous phase) */
  int32_t *n_out;           /* number of instances this phase has parked */
  const int32_t *in_inst;   /* resume: instance index ... */
  const int32_t *in_src;    /* ... and wave * 64 + lane of the tile column (of src_ws) that holds its iterate */
  const double *in_park;    /* resume: [PARK_N][ld_park] solver scalars, column = position in the list */
  int32_t *out_inst, *out_src;
  double *out_park;
  int64_t ld_park;
  const void *src_ws;       /* resume: workspace of the previous phase */
  int32_t pass_cut;         /* park after this many passes in this phase (0 = never) */
  int32_t resume;           /* 0 = first phase (fresh instances), 1 = takes parked ones */
  /* Mixed precision across phases (MpcParams.f32_finish, f64_f32_start): a phase with promote_out != 0 runs the fp32 solver
   * with Solver::promote_mu set and parks every instance that returns MPC_PROMOTE; the next phase (promote_in != 0) is the
   * fp64 solver resuming from that list: the parked iterate, in the fp32 record layout of src_ws (tiles of src_tile_reals
   * floats), is converted field by field, the point is re-evaluated in fp64 and the solve goes on to tol and the polish. */
  int32_t promote_out, promote_in;
  int32_t promote_cap;      /* iterations the fp32 phase may spend on an instance (0: Solver::kPromoteIterCap) */
  int64_t src_tile_reals;
  /* The promoted iterates travel in a buffer of their own, [list position / 64][N-1][IT_SZ][64] reals of the fp32 record:
   * a lane that has handed its instance over is free at once -- it takes the next instance while the launch has any, and
   * its column is there for lane compaction -- and the fp64 phase reads 64 consecutive entries per wave instead of 64
   * columns scattered over the fp32 workspace.  (nullptr: the iterate stays in its column, as in a cut schedule.) */
  void *p_iter;
  /* Hand-over policy.  Writing a finished instance out and fetching the next one (set-up, start point: 270 stores) is
   * divergent code that the whole wave pays for, ~4 us per event against ~80 us per pass, and with 64 lanes finishing at
   * different times nearly every pass would have one.  So finished lanes WAIT until `refill_min` lanes of the wave are
   * waiting, or `refill_wait` passes have gone by, or nothing else is running; then all of them are served at once. */
  int32_t refill_min, refill_wait;
  int32_t refill_floor;     /* no further takes once fewer lanes than this are running (0 = take whenever there is work) */
  /* Lane compaction (MPC_LANE_COMPACT=gap, measurement aid).  Memory is fetched in 128-byte lines = the 16-byte groups of 8
   * neighbouring lanes, so a line is fetched as long as ONE of its 8 lanes still runs (tools/traffic_model.py: the launch
   * fetches 1.24 x what its running lanes ask for).  Once the launch's counter is exhausted, a wave whose running lanes are
   * spread over `compact_gap` more 8-lane groups than they need moves the ones outside its fullest groups into free lanes
   * inside them: solver scalars through LDS (the staging buffers are idle between passes), set-up repeated from the inputs,
   * the iterate copied column to column -- the arithmetic of an instance does not depend on its lane. */
  int32_t compact_gap, compact_cooldown;   /* (passes without another move after one) */
  /* Tile pool (MpcTilePool, optional): instead of tile number blockIdx of the handle's own workspace a wave takes a free
   * tile from the pool of ITS XCD and gives it back when it leaves, so that the addresses the device cycles through are
   * the tiles of the resident waves and not those of every batch in flight. */
  /* Deferred tails (MpcParams.tail_cut): an instance still running after tail_cut passes is handed to the handle's tail
   * queue -- solver scalars, its inputs and its current iterate are COPIED out, so the lane and the workspace are free at
   * once -- and reported as MPC_STATUS_PENDING; mpc_tail_kernel finishes it from another stream.  A full queue (t_cap) makes
   * the instance finish here after all. */
  int32_t tail_cut, t_slot;        /* t_slot: the batch's slot in the handle's ring (its live-instance counter, its final flag) */
  /* ... and a wave does not wait for its last few lanes: once the launch has no work left to hand out and at most tail_few of
   * a wave's lanes are still running (after tail_few_from passes), they are handed over too and the wave leaves.  A launch
   * of one instance per lane lasts as long as its waves do, a wave as long as the slowest of its 64 instances: the mean of
   * that maximum is 19 passes on the survey population with a cut at 20, 17.3 when the last two lanes are not waited for
   * (2.6 % of the instances handed over instead of 1.1 %). */
  int32_t tail_few, tail_few_from;
  int64_t t_batch;                 /* the batch's id */
  MpcTailQ tq;                     /* the batch's fresh queue */
  int32_t *zero_next;              /* first launch of a solve call: kCounterInts counters to reset for a later call (no memset launches on
                                    * the stream: on a full device each of those little launches waits for a free SIMD) */
  unsigned long long *pool_bits;   /* [8][pool_words]: bit set = tile free; nullptr = no pool */
  void *pool_base;                 /* [8][pool_tiles] tiles */
  int32_t pool_tiles, pool_words;  /* per XCD */
};

/* Copies between queue entries and a lane's workspace column.  Loads first, then stores, a stage (or 16 rows) at a time: written as
 * `dst[..] = src[..]` in one loop the compiler must assume that a store aliases the next load and waits for every load before the
 * next one goes out -- 200 round trips of ~2 us per entry, which priced a wave's life at +10 % per deferred instance. */
template <class R, class WS>
__device__ __forceinline__ void tile_from_column(R *__restrict__ dst, const WS &ws, int I, int M) {      /* dst: lane's place in a [M][IT_SZ][64] tile */
  using FL = mpc::Fields<R>;
  for (int k = 0; k < M; ++k) {
    R rec[FL::IT_SZ];
#pragma unroll
    for (int f = 0; f < FL::IT_SZ; f++) rec[f] = ws.it(k, I, f);
#pragma unroll
    for (int f = 0; f < FL::IT_SZ; f++) dst[(k * FL::IT_SZ + f) * 64] = rec[f];
  }
}
__device__ __forceinline__ void rows_copy(double *__restrict__ dk, int64_t dl, const double *__restrict__ pk, int64_t lp, int q0, int q1) {
  for (int q = q0; q < q1; q += 16) {
    double t[16];
#pragma unroll
    for (int e = 0; e < 16; e++) t[e] = q + e < q1 ? pk[(q + e) * lp] : 0.0;
#pragma unroll
    for (int e = 0; e < 16; e++) if (q + e < q1) dk[(q + e) * dl] = t[e];
  }
}

/* OCC = waves per SIMD the register allocation is held to: the fp64 solver needs ~380 registers (1); the fp32 solver
 * fits 256 with a few spilled values (2), or runs unconstrained (1) */
/* what Solver::unpack writes through when the arrays at the ABI are of another type than the solver's reals */
template <class RIO, class R> struct OutRef {
  RIO *p;
  __device__ __host__ void operator=(R v) const { *p = (RIO)v; }
};

/* RIO: the type of the arrays at the ABI (inputs, outputs); R: the solver's.  They differ only in the fp64 phase of a
 * mixed-precision solve on an MPC_PRECISION_F32 handle (RIO = float, R = double).  RSRC: the reals of the workspace a
 * promote_in phase takes its iterates from. */
template <bool STAGING, class R, int OCC, class RIO = R, class RSRC = RIO>
__global__ __launch_bounds__(kBlock, OCC) void mpc_solve_kernel(
    const MpcParams P, const int64_t B, const int64_t ld, const int64_t ldo, const RIO *__restrict__ state,
    const RIO *__restrict__ coeffs, const RIO *__restrict__ yaw_lo, const RIO *__restrict__ yaw_hi,
    const RIO *__restrict__ weights, RIO *__restrict__ out, RIO *__restrict__ traj,
    int32_t *__restrict__ status, int32_t *__restrict__ iters, R *__restrict__ wsbase,
    const int64_t tile_reals, const MpcPhase T) {
  extern __shared__ double smem[];
  using WS = mpc::TiledWorkspace<STAGING, R>;
  using SV = mpc::Solver<WS, R>;
  using FL = mpc::Fields<R>;
  static_assert(SV::PARK_N == kParkRows, "park buffer rows");
  WS ws;
  ws.tile = (typename WS::greal *)(wsbase + (int64_t)blockIdx.x * tile_reals);
#if defined(__HIP_DEVICE_COMPILE__)
  int pool_word = -1, pool_bit = 0;
  unsigned long long *pool_mine = nullptr;
  if (T.pool_bits) {
    /* first free tile of this XCD's pool (lowest number first: the set of tiles in use stays compact); a tile is only
     * ever used by waves of one XCD, so everything written to it sits in one L2.  No free tile (cannot happen while the
     * pool holds more tiles than an XCD has wave slots): the wave keeps its own tile of the handle's workspace. */
    const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;   /* HW_REG_XCC_ID[3:0] */
    pool_mine = T.pool_bits + (size_t)xcc * (size_t)T.pool_words;
    int got = -1;
    if (threadIdx.x == 0) {
      for (int w = 0; w < (T.pool_tiles + 63) / 64 && got < 0; ++w) {   /* the tile bits only: the XCD's last two words are its usage record */
        unsigned long long v = __hip_atomic_load(pool_mine + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (v) {
          const int b = __builtin_ctzll(v);
          const unsigned long long m = 1ull << b;
          const unsigned long long old = atomicAnd(pool_mine + w, ~m);
          if (old & m) { got = w * 64 + b; break; }
          v = old & ~m;
        }
      }
    }
    got = __builtin_amdgcn_readfirstlane(got);
    if (got >= T.pool_tiles) got = -1;       /* (bits beyond the pool are never set; belt and braces) the wave keeps the handle's own tile */
    if (got >= 0 && threadIdx.x == 0) {      /* usage record for mpc_debug_tile_pool: claims, highest tile number + 1 */
      atomicAdd(pool_mine + T.pool_words - 2, 1ull);
      atomicMax(pool_mine + T.pool_words - 1, (unsigned long long)(got + 1));
    }
    if (got >= 0) {
      pool_word = got >> 6; pool_bit = got & 63;
      ws.tile = (typename WS::greal *)((R *)T.pool_base + ((int64_t)xcc * T.pool_tiles + got) * tile_reals);
    }
  }
#endif
  ws.lane = threadIdx.x;
  ws.lbuf = (typename WS::lreal *)smem;
  if (T.zero_next && blockIdx.x == 0 && threadIdx.x < kCounterInts) T.zero_next[threadIdx.x] = 0;
  SV S(P, ws);
  if (T.promote_out) { S.promote_mu = (R)P.mixed_switch_mu; if (T.promote_cap > 0) S.promote_cap = T.promote_cap; }
  int64_t i = 0;
  bool have = false, more = true, fin = false;   /* holds a running instance / may still get one / holds a finished one */
  bool queue_full = false;                       /* deferred tails: the batch's queue slot has no room left */
  bool col_busy = false;                         /* this lane's column holds a parked iterate */
  int attempt = 0, it_total = 0, passes = 0, fin_status = 0, waited = 0, cooldown = 0, passes_wave = 0;
  const int64_t n_work = T.resume ? (int64_t)*T.n_in : B;
  (void)col_busy; (void)cooldown;
  for (;;) {
#if defined(__HIP_DEVICE_COMPILE__)
    /* ---- lane compaction, part 1: is it worth it now?  (see MpcPhase.compact_gap) ---- */
    bool want_compact = false;
    int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, g_min = 0;
    if (STAGING && T.compact_gap > 0) {
      if (MPC_WAVE_ANY(more) && (int64_t)__hip_atomic_load(T.take, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n_work) more = false;
      if (!MPC_WAVE_ANY(more)) {
        if (cooldown > 0) --cooldown;
        else {
          const unsigned long long live = __builtin_amdgcn_ballot_w64(have);
          const int nl = __builtin_popcountll(live);
          int g_now = 0;
#pragma unroll
          for (int g = 0; g < 8; g++) { cnt[g] = __builtin_popcountll((live >> (8 * g)) & 0xffull); g_now += cnt[g] > 0 ? 1 : 0; }
          g_min = (nl + 7) >> 3;
          want_compact = nl > 0 && g_now - g_min >= T.compact_gap;
        }
      }
    }
#else
    const bool want_compact = false;
#endif
    /* ---- hand-over point (wave-uniform decision, see MpcPhase) ---- */
    const int n_wait = MPC_WAVE_COUNT(fin || (!have && more));
    if (n_wait > 0) {
      if (!MPC_WAVE_ANY(have) || n_wait >= T.refill_min || waited >= T.refill_wait || want_compact) {
        waited = 0;
        if (fin && fin_status <= kFinPromote) {
          /* mixed precision, promoted iterates in their own buffer: the hand-over to the next phase happens here, for all the
           * lanes of the wave that are waiting for it at once (done lane by lane as they promote it cost 20 % of the rate) */
          const int64_t pos = (int64_t)atomicAdd(T.n_out, 1);
          T.out_inst[pos] = (int32_t)i;
          T.out_src[pos] = (int32_t)(blockIdx.x * 64u + threadIdx.x);
          double *pk = T.out_park + pos;
          const int64_t lp = T.ld_park;
          S.park([pk, lp](int q) -> double & { return pk[q * lp]; }, attempt, it_total);
          pk[35 * lp] = fin_status == kFinPromote ? 0.0 : 1.0;
          if (fin_status == kFinPromote) {
            ws.stage_drain();                          /* the trial sweep's stores of this wave have landed */
            const int I = S.cur ? FL::IT1 : FL::IT0, M = P.N - 1;
            R *dst = (R *)T.p_iter + (pos >> 6) * (int64_t)M * FL::IT_SZ * 64 + (pos & 63);
            for (int k = 0; k < M; ++k) {
              R rec[FL::IT_SZ];
#pragma unroll
              for (int f = 0; f < FL::IT_SZ; f++) rec[f] = ws.it(k, I, f);
#pragma unroll
              for (int f = 0; f < FL::IT_SZ; f++) dst[(k * FL::IT_SZ + f) * 64] = rec[f];
            }
          }
          fin = false;
        }
        if (fin) {
          RIO *o = out + i;
          RIO *t = traj ? traj + i : nullptr;
          const int64_t l = ldo;
          S.unpack([o, l](int q) { return OutRef<RIO, R>{o + q * l}; }, [t, l](int q) { return OutRef<RIO, R>{t + q * l}; }, traj != nullptr,
                   (R)yaw_lo[i], (R)yaw_hi[i]);
          status[i] = fin_status;
          if (iters) iters[i] = S.iters + it_total;
          fin = false;
        }
        /* a wave most of whose lanes have finished starts nothing new: what it would take runs denser in a wave that starts
         * fresh (the grid has a lane for every instance), and this one would last twice as long for a handful of lanes */
        if (T.refill_floor > 0 && passes_wave > 0 && MPC_WAVE_COUNT(have) < T.refill_floor) more = false;
        bool exhausted = false;
        if (!have && more) {
          const int64_t pos = (int64_t)atomicAdd(T.take, 1);
          more = pos < n_work;
          exhausted = !more;
          if (more) {
            i = T.resume ? (int64_t)T.in_inst[pos] : pos;
            R st[6], cf[MPC_NCOEF], w[MPC_NW];
#pragma unroll
            for (int q = 0; q < 6; q++) st[q] = (R)state[q * ld + i];
#pragma unroll
            for (int q = 0; q < MPC_NCOEF; q++) cf[q] = (R)coeffs[q * ld + i];
            if (weights) {
#pragma unroll
              for (int q = 0; q < MPC_NW; q++) w[q] = (R)weights[q * ld + i];
            } else {
#pragma unroll
              for (int q = 0; q < MPC_NW; q++) w[q] = (R)P.weights[q];
            }
            const bool from_scratch = T.promote_in && T.in_park[pos + 35 * T.ld_park] != 0.0;   /* the fp32 phase gave up on it */
            const int s0 = S.setup(st, cf, (R)yaw_lo[i], (R)yaw_hi[i], w, !T.resume || from_scratch);
            if (T.promote_in && s0 != MPC_STATUS_SUCCESS) {
              /* a start state that the fp32 set-up let through (outside the relaxed bounds by less than the fp32 spacing) and this
               * solver's set-up rejects: the verdict of the single-phase solve, the start point reported */
              S.start_point();
              S.cur = 0; S.E.f = R(0.0);
              it_total = 0; S.iters = 0; fin = true; fin_status = s0;
            } else if (from_scratch) { S.begin(true); attempt = 0; it_total = (int)T.in_park[pos + 23 * T.ld_park] + (int)T.in_park[pos + 29 * T.ld_park]; passes = 0; have = true; }   /* (the fp32 phase's iterations count) */
            else if (T.resume) {
              /* bring the parked iterate over: the column (src wave, src lane) of phase A's workspace -> own column */
              const double *pk = T.in_park + pos;
              const int64_t lp = T.ld_park;
              S.unpark([pk, lp](int q) -> double { return pk[q * lp]; }, attempt, it_total);
              const int src = T.in_src[pos];
              const int I = S.cur ? FL::IT1 : FL::IT0;
              if (T.promote_in) {
                /* the iterate as the fp32 phase left it: its record layout, its tile size; then this solver's own evaluation */
                using FS = mpc::Fields<RSRC>;
                mpc::TiledWorkspace<false, RSRC> wsrc;
                wsrc.tile = (typename mpc::TiledWorkspace<false, RSRC>::greal *)((const RSRC *)T.src_ws + (int64_t)(src >> 6) * T.src_tile_reals);
                wsrc.lane = src & 63; wsrc.lbuf = nullptr;
                const int Is = S.cur ? FS::IT1 : FS::IT0;
                const RSRC *pit = T.p_iter ? (const RSRC *)T.p_iter + (pos >> 6) * (int64_t)(P.N - 1) * FS::IT_SZ * 64 + (pos & 63) : nullptr;
                for (int k = 0; k < P.N - 1; ++k) {
                  R rec[FL::IT_SZ] = {};
                  mpc::convert_iterate_record<RSRC, R>([&](int f) { return pit ? pit[(k * FS::IT_SZ + f) * 64] : (RSRC)wsrc.it(k, Is, f); }, [&](int f, R v) { rec[f] = v; });
                  ws.template store_run<0, FL::IT_SZ>(k, I, rec);
                  /* the re-evaluation is a trial sweep with step length 0: it multiplies whatever the direction record holds */
                  const R zero[FL::D_N] = {};
                  ws.template store_run<FL::F_D, FL::D_N>(k, 0, zero);
                }
                S.promoted();
                attempt = -1;      /* came in from the fp32 phase: should it fail, it is solved again the way the single-phase solve starts */
              } else {
                WS wsrc = ws;
                wsrc.tile = (typename WS::greal *)((const R *)T.src_ws + (int64_t)(src >> 6) * tile_reals);
                wsrc.lane = src & 63;
                for (int k = 0; k < P.N - 1; ++k) {
                  R rec[FL::IT_SZ];
#pragma unroll
                  for (int f = 0; f < FL::IT_SZ; f++) rec[f] = wsrc.it(k, I, f);
                  ws.template store_run<0, FL::IT_SZ>(k, I, rec);
                }
              }
              passes = 0; have = true;
            } else if (s0 == MPC_STATUS_SUCCESS) { S.begin(true); attempt = 0; it_total = 0; passes = 0; have = true; }
            else { it_total = 0; S.iters = 0; fin = true; fin_status = s0; }   /* rejected at set-up (initial state outside its own
                                                                               * bounds): the start point is reported at the next hand-over */
          }
        }
        /* The counter only grows: what one lane found empty is empty for all.  (Until then a finished lane does take again,
         * also in a launch that has a lane for every instance: with batches in flight the last waves of a grid start late,
         * and the lanes that finish early in its first waves take their instances -- those waves then find nothing and leave.
         * Declaring the counter exhausted after every lane's first take was measured: -6 % on the headline.) */
        if (MPC_WAVE_ANY(exhausted)) more = false;
      } else ++waited;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    /* ---- lane compaction, part 2 (finished lanes have just been served: their columns are free) ---- */
    if (want_compact) {
      unsigned long long keep = 0;               /* the g_min fullest groups stay where they are */
      for (int t = 0; t < g_min; ++t) {
        int best = 0, bc = -1;
#pragma unroll
        for (int g = 0; g < 8; g++) if (!((keep >> (8 * g)) & 1ull) && cnt[g] > bc) { bc = cnt[g]; best = g; }
        keep |= 0xffull << (8 * best);
      }
      const bool in_keep = (keep >> threadIdx.x) & 1ull;
      const bool movable = have && S.phase == SV::PH_DIR && !in_keep;
      const bool is_free = !have && !fin && !col_busy && in_keep;
      const unsigned long long mv = __builtin_amdgcn_ballot_w64(movable), fr = __builtin_amdgcn_ballot_w64(is_free);
      const int n_mv = __builtin_popcountll(mv), n_fr = __builtin_popcountll(fr);
      const int n = n_mv < n_fr ? n_mv : n_fr;
      if (n > 0) {
        ws.stage_drain();                        /* the trial sweep's stores have landed, the staging buffers are idle */
        static_assert(!STAGING || staging_lds_bytes<R>() >= (size_t)kParkRows * 64 * sizeof(R) + 2 * 64 * sizeof(int), "the mailbox must fit the staging buffers");
        R *mb = (R *)smem;                       /* [kParkRows][64]: Solver::park scalars (in the solver's own precision) ... */
        int *mi = (int *)(mb + kParkRows * 64);  /* ... [2][64]: instance, passes */
        const unsigned long long below = (1ull << threadIdx.x) - 1ull;
        const bool is_src = movable && __builtin_popcountll(mv & below) < n;
        const int my_f = __builtin_popcountll(fr & below);
        const bool is_dst = is_free && my_f < n;
        if (is_src) {
          const unsigned ln = threadIdx.x;
          S.park([mb, ln](int q) -> R & { return mb[q * 64 + ln]; }, attempt, it_total);
          mi[ln] = (int)i; mi[64 + ln] = passes;
          have = false;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (is_dst) {
          unsigned long long m = mv;
          for (int r = 0; r < my_f; ++r) m &= m - 1ull;
          const int src = __builtin_ctzll(m);
          i = (int64_t)mi[src]; passes = mi[64 + src];
          R st[6], cf[MPC_NCOEF], w[MPC_NW];
#pragma unroll
          for (int q = 0; q < 6; q++) st[q] = (R)state[q * ld + i];
#pragma unroll
          for (int q = 0; q < MPC_NCOEF; q++) cf[q] = (R)coeffs[q * ld + i];
          if (weights) {
#pragma unroll
            for (int q = 0; q < MPC_NW; q++) w[q] = (R)weights[q * ld + i];
          } else {
#pragma unroll
            for (int q = 0; q < MPC_NW; q++) w[q] = (R)P.weights[q];
          }
          (void)S.setup(st, cf, (R)yaw_lo[i], (R)yaw_hi[i], w, false);
          S.unpark([mb, src](int q) -> R { return mb[q * 64 + src]; }, attempt, it_total);
          const int I = S.cur ? FL::IT1 : FL::IT0;
          WS wsrc = ws;
          wsrc.lane = src;
          /* (plain loads: the column was written through this CU's own L1 by a lane of this wave, and stage_drain() has waited
           * for the stores; keeping two stages in flight was measured: no difference) */
          for (int k = 0; k < P.N - 1; ++k) {
            R rec[FL::IT_SZ];
#pragma unroll
            for (int f = 0; f < FL::IT_SZ; f++) rec[f] = wsrc.it(k, I, f);
            ws.template store_run<0, FL::IT_SZ>(k, I, rec);
          }
          have = true;
        }
        __builtin_amdgcn_wave_barrier();
        cooldown = T.compact_cooldown;
      }
    }
#endif
    if (!MPC_WAVE_ANY(have || more || fin)) break;
    ++passes_wave;
    const bool few = T.tail_few > 0 && !MPC_WAVE_ANY(more) && MPC_WAVE_COUNT(have) <= T.tail_few;
    if (have) {
      const int r = S.step();
      ++passes;
      /* Mixed precision.  Only a CLEAN hand-over is continued by the fp64 solver (the fp32 phase reached the switch value of the
       * barrier parameter, or tol_f32).  One that comes out of trouble -- the iteration allowance used up, line search or inertia
       * correction out of single precision, not-a-number -- sends the instance to the fp64 solver FROM THE START POINT (row 35 of
       * the parked scalars): on hard instances the fp32 iterates lead into other local minima than the fp64 ones (SURVEY's
       * unfiltered populations: 7 of 32 768 at N = 25, 24 of 65 536 at N = 10 with the old rule; none with this one). */
      if ((r == SV::MPC_PROMOTE || (T.promote_out && r == MPC_STATUS_NUMERIC)) && T.p_iter) {
        /* mixed precision: this phase has taken the instance as far as it is asked to; it waits for the wave's next hand-over */
        fin = true; fin_status = (r == SV::MPC_PROMOTE && S.promote_clean) ? kFinPromote : kFinScratch; have = false;
      } else if (r == SV::MPC_PROMOTE || (T.promote_out && r == MPC_STATUS_NUMERIC)) {
        /* the same with the iterate left in its column (MPC_PROMOTE_BUFFER=0): the next phase's solver takes over.
         * Not-a-number in the fp32 phase (states far from the origin late in a closed loop: x^4 terms, lost digits) is not a
         * verdict on the instance: the fp64 solver gets it from the start point (row 35 of the parked scalars says so). */
        const int64_t pos = (int64_t)atomicAdd(T.n_out, 1);
        T.out_inst[pos] = (int32_t)i;
        T.out_src[pos] = (int32_t)(blockIdx.x * 64u + threadIdx.x);
        double *pk = T.out_park + pos;
        const int64_t lp = T.ld_park;
        S.park([pk, lp](int q) -> double & { return pk[q * lp]; }, attempt, it_total);
        pk[35 * lp] = (r == SV::MPC_PROMOTE && S.promote_clean) ? 0.0 : 1.0;
        have = false; more = false; col_busy = true;     /* the column keeps the parked iterate: this lane takes nothing else */
      } else if (r != SV::MPC_RUNNING) {
        if (attempt < 0 && r != MPC_STATUS_SUCCESS) {
          /* An instance the fp32 phase started (attempt -1) and the fp64 phase could not finish (1-3 of 8 192 at N = 25: the
           * line search fails from where fp32 left it): it is solved again from the start point exactly as the single-phase
           * solve does it -- first attempt, then the restart if that fails -- so status AND returned point are that solve's
           * (the named hard instances of tests/helpers.py stay what the oracle says).  Trying the restart first is cheaper
           * for these stragglers (their chain: fp32 + the failed fp64 continuation + a whole solve) but was measured to return
           * the restart's local minimum, or its last iterate, where the single-phase solve returns the first attempt's. */
          attempt = 0; it_total += S.iters;
          S.start_point();
          S.begin(true);
        } else if (r == MPC_STATUS_LINESEARCH && attempt == 0 && !S.no_restart) {
          /* the stand-in for IPOPT's restoration phase: once more from the start point, zero multipliers */
          attempt = 1; it_total += S.iters;
          S.start_point();
          S.begin(false);
        } else { fin = true; fin_status = r; have = false; }
      } else if (T.tail_cut > 0 && (passes >= T.tail_cut || (few && passes >= T.tail_few_from)) && S.phase == SV::PH_DIR && !queue_full) {
        const int pos = atomicAdd(T.tq.count, 1);
        if (pos >= T.tq.cap) queue_full = true;     /* it finishes here, and so does whatever else this lane takes */
        else {
          double *pk = T.tq.park + pos;
          const int64_t lp = T.tq.cap;
          S.park([pk, lp](int q) -> double & { return pk[q * lp]; }, attempt, it_total);
          pk += kParkRows * lp;
          {
            double in[kTailInRows];
#pragma unroll
            for (int q = 0; q < 6; q++) in[q] = (double)state[q * ld + i];
#pragma unroll
            for (int q = 0; q < MPC_NCOEF; q++) in[6 + q] = (double)coeffs[q * ld + i];
            in[11] = (double)yaw_lo[i]; in[12] = (double)yaw_hi[i];
#pragma unroll
            for (int q = 0; q < MPC_NW; q++) in[13 + q] = weights ? (double)weights[q * ld + i] : P.weights[q];
#pragma unroll
            for (int q = 0; q < kTailInRows; q++) pk[q * lp] = in[q];
          }
          pk += kTailInRows * lp;
          pk[0] = (double)i; pk[lp] = (double)T.t_slot; pk[2 * lp] = (double)T.t_batch;
          pk[3 * lp] = __longlong_as_double((long long)out); pk[4 * lp] = __longlong_as_double((long long)traj);
          pk[5 * lp] = __longlong_as_double((long long)status); pk[6 * lp] = __longlong_as_double((long long)iters);
          pk[7 * lp] = (double)ldo;
          ws.stage_drain();                          /* the trial sweep's stores of this wave have landed */
          /* (copied by the lane itself: the whole wave copying a deferring lane's column -- two round trips instead of twenty --
           * was built in round 4 and measured: no change in the launch's duration, 2.67 ms either way) */
          tile_from_column<R>((R *)T.tq.iter + (int64_t)(pos >> 6) * (P.N - 1) * FL::IT_SZ * 64 + (pos & 63), ws, S.cur ? FL::IT1 : FL::IT0, P.N - 1);
          status[i] = MPC_STATUS_PENDING;
          if (iters) iters[i] = S.iters + it_total;
          have = false;                              /* the lane takes its next instance at the next hand-over */
        }
      } else if (T.pass_cut > 0 && passes >= T.pass_cut && S.phase == SV::PH_DIR) {
        /* still running: park it for the next phase */
        const int64_t pos = (int64_t)atomicAdd(T.n_out, 1);
        T.out_inst[pos] = (int32_t)i;
        T.out_src[pos] = (int32_t)(blockIdx.x * 64u + threadIdx.x);
        double *pk = T.out_park + pos;
        const int64_t lp = T.ld_park;
        S.park([pk, lp](int q) -> double & { return pk[q * lp]; }, attempt, it_total);
        have = false; more = false; col_busy = true;     /* the column keeps the parked iterate: this lane takes nothing else */
      }
    }
  }
#if defined(__HIP_DEVICE_COMPILE__)
  if (pool_word >= 0) {
    /* every store of this wave to the tile has been acknowledged before the tile can be handed on */
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) atomicOr(pool_mine + pool_word, 1ull << pool_bit);
  }
#endif
}

/* Deferred tails: a TAIL SLICE works on everything the launches have handed over (MpcPhase.tail_cut) for a bounded number of
 * passes.  Same solver, same arithmetic -- the results are bitwise those of an undisturbed launch -- on the handle's tail
 * stream, a few dense waves beside the launches of later batches.  Its sources are the survivors of the slice before it and
 * the fresh queues of the batches whose launches have ended since; a lane takes entries in turn, carries each one on, and
 * when the slice's budget of passes is spent, whatever is still running is parked again (at a pass boundary) in the
 * survivors' list for the next slice -- so a slice lasts a couple of milliseconds however long the longest chain is, its
 * waves are dense again at every slice, and a batch is final a slice after its own last straggler is.  Entries nobody
 * has taken by then are moved over as they are.  An entry for which the survivors' list has no room left stays with its
 * lane until it is solved (the slice lasts longer; nothing is lost).
 * Finality: remaining[slot] counts a batch's live stragglers -- the slice that absorbs its fresh queue adds their number,
 * every finished one subtracts one -- and whoever brings it to zero writes the batch id into final[slot] (the counter cannot
 * be zero before both have happened: partial sums without the addition are negative, with it positive). */
struct MpcSliceArgs {
  int32_t n_src, budget;           /* sources in use; wave passes after which the slice parks what is still running */
  int32_t ring, pad_;              /* slots of the handle's ring */
  MpcTailQ src[kSliceMaxSrc];      /* [0]: survivors of the previous slice; [1..]: fresh queues absorbed by this slice */
  int32_t fresh_slot[kSliceMaxSrc];
  int64_t fresh_batch[kSliceMaxSrc];
  MpcTailQ dst;                    /* survivors of this slice */
  int32_t *take;                   /* work counter over the concatenated sources */
  int32_t *done;                   /* waves of this slice that have left */
  int32_t *remaining;              /* [ring] */
  long long *final_id;             /* [ring], pinned host memory: the id of the batch that has become final in the slot */
  int32_t *res;                    /* pinned host memory: what the pump reads when the slice has completed -- [0] survivors left,
                                    * [j] entries of source j; [25..27]: entries finished / moved on untouched / parked again */
  int32_t *tally;                  /* [4]: those three, and the most passes a wave of the slice made */
};

template <bool STAGING, class R, int OCC, class RIO = R>
__global__ __launch_bounds__(kBlock, OCC) void mpc_tail_slice_kernel(const MpcParams P, const MpcSliceArgs A, R *__restrict__ wsbase,
                                                                     const int64_t tile_reals) {
  extern __shared__ double smem[];
  using WS = mpc::TiledWorkspace<STAGING, R>;
  using SV = mpc::Solver<WS, R>;
  using FL = mpc::Fields<R>;
  WS ws;
  ws.tile = (typename WS::greal *)(wsbase + (int64_t)blockIdx.x * tile_reals);
  ws.lane = threadIdx.x;
  ws.lbuf = (typename WS::lreal *)smem;
  SV S(P, ws);
  const int M = P.N - 1;
  int64_t total = 0;
  for (int j = 0; j < A.n_src; j++) { const int c = *A.src[j].count; total += c < A.src[j].cap ? c : A.src[j].cap; }
  if (blockIdx.x == 0 && threadIdx.x >= 1 && (int)threadIdx.x < A.n_src) {
    /* absorb a fresh queue: its stragglers now count as live instances of their batch (a batch that deferred nothing is final here) */
    const int j = threadIdx.x;
    const int c0 = *A.src[j].count, c = c0 < A.src[j].cap ? c0 : A.src[j].cap;
    const int old = atomicAdd(A.remaining + A.fresh_slot[j], c);
    if (old + c == 0) __hip_atomic_store(A.final_id + A.fresh_slot[j], (long long)A.fresh_batch[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const double *pk = nullptr;                      /* the entry this lane holds: its column in the source's rows, the rows' stride */
  int64_t lp = 0;
  R ylo_user = 0, yhi_user = 0;
  bool have = false, more = true, fin = false;
  bool keep = false;                               /* no room in the survivors' list: this lane's instance is solved here */
  int attempt = 0, it_total = 0, fin_status = 0, wp = 0;
  for (;;) {
    const bool spent = wp >= A.budget;
    if (fin) {
      const double *pm = pk + (int64_t)(kParkRows + kTailInRows) * lp;
      const int64_t i = (int64_t)pm[0];
      const int slot = (int)pm[lp];
      RIO *ob = (RIO *)__double_as_longlong(pm[3 * lp]);
      RIO *tb = (RIO *)__double_as_longlong(pm[4 * lp]);
      int32_t *st_arr = (int32_t *)__double_as_longlong(pm[5 * lp]), *it_arr = (int32_t *)__double_as_longlong(pm[6 * lp]);
      const int64_t l = (int64_t)pm[7 * lp];
      /* an entry is 80 numbers that have travelled through one or more queues: what it says about where its results go is checked
       * before it is believed (a damaged one is reported to the host -- mpc_last_error -- instead of being written through) */
      if (!(slot >= 0 && slot < A.ring && i >= 0 && i < l && ob && st_arr && l > 0)) {
        __hip_atomic_store(A.res + 28, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(A.res + 29, slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(A.res + 30, (int)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      } else {
        RIO *o = ob + i;
        RIO *t = tb ? tb + i : nullptr;
        S.unpack([o, l](int q) { return OutRef<RIO, R>{o + q * l}; }, [t, l](int q) { return OutRef<RIO, R>{t + q * l}; }, t != nullptr, ylo_user, yhi_user);
        st_arr[i] = fin_status;
        if (it_arr) it_arr[i] = S.iters + it_total;
        const int old = atomicSub(A.remaining + slot, 1);
        if (old == 1) __hip_atomic_store(A.final_id + slot, (long long)pm[2 * lp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        atomicAdd(A.tally, 1);
      }
      fin = false;
    }
    if (!have && more) {
      int64_t g = (int64_t)atomicAdd(A.take, 1);
      more = g < total;
      if (more) {
        int sj = 0;
        for (int j = 0; j < A.n_src; j++) {
          const int c0 = *A.src[j].count, c = c0 < A.src[j].cap ? c0 : A.src[j].cap;
          if (g >= c && j == sj) { g -= c; sj = j + 1; }
        }
        const R *it_src = nullptr;
        for (int j = 0; j < A.n_src; j++)
          if (j == sj) { pk = A.src[j].park + g; lp = A.src[j].cap; it_src = (const R *)A.src[j].iter + (g >> 6) * (int64_t)M * FL::IT_SZ * 64 + (g & 63); }
        int dpos = -1;
        if (spent) {                               /* the budget is spent: the entry moves to the next slice as it is, if there is room */
          dpos = atomicAdd(A.dst.count, 1);
          if (dpos >= A.dst.cap) { dpos = -1; keep = true; }
        }
        if (dpos >= 0) {
          atomicAdd(A.tally + 1, 1);
          double *dk = A.dst.park + dpos;
          const int64_t dl = A.dst.cap;
          rows_copy(dk, dl, pk, lp, 0, kTailRows);
          R *it_dst = (R *)A.dst.iter + (int64_t)(dpos >> 6) * M * FL::IT_SZ * 64 + (dpos & 63);
          for (int k = 0; k < M; ++k) {
            R rec[FL::IT_SZ];
#pragma unroll
            for (int f = 0; f < FL::IT_SZ; f++) rec[f] = it_src[(k * FL::IT_SZ + f) * 64];
#pragma unroll
            for (int f = 0; f < FL::IT_SZ; f++) it_dst[(k * FL::IT_SZ + f) * 64] = rec[f];
          }
        } else {
          const double *pin = pk + (int64_t)kParkRows * lp;
          R st[6], cf[MPC_NCOEF], w[MPC_NW];
#pragma unroll
          for (int q = 0; q < 6; q++) st[q] = (R)pin[q * lp];
#pragma unroll
          for (int q = 0; q < MPC_NCOEF; q++) cf[q] = (R)pin[(6 + q) * lp];
#pragma unroll
          for (int q = 0; q < MPC_NW; q++) w[q] = (R)pin[(13 + q) * lp];
          ylo_user = (R)pin[11 * lp]; yhi_user = (R)pin[12 * lp];
          (void)S.setup(st, cf, ylo_user, yhi_user, w, false);
          const double *pq = pk;
          const int64_t lq = lp;
          S.unpark([pq, lq](int q) -> double { return pq[q * lq]; }, attempt, it_total);
          const int I = S.cur ? FL::IT1 : FL::IT0;
          for (int k = 0; k < M; ++k) {
            R rec[FL::IT_SZ];
#pragma unroll
            for (int f = 0; f < FL::IT_SZ; f++) rec[f] = it_src[(k * FL::IT_SZ + f) * 64];
            ws.template store_run<0, FL::IT_SZ>(k, I, rec);
          }
          have = true;
        }
      }
    }
    if (!MPC_WAVE_ANY(have || more || fin)) break;
    if (have) {
      const int r = S.step();
      if (r != SV::MPC_RUNNING) {
        if (attempt < 0 && r != MPC_STATUS_SUCCESS) {        /* (as in mpc_solve_kernel: started by the fp32 phase, not finished by fp64) */
          attempt = 0; it_total += S.iters;
          S.start_point();
          S.begin(true);
        } else if (r == MPC_STATUS_LINESEARCH && attempt == 0 && !S.no_restart) {
          attempt = 1; it_total += S.iters;
          S.start_point();
          S.begin(false);
        } else { fin = true; fin_status = r; have = false; keep = false; }
      } else if (spent && !keep && S.phase == SV::PH_DIR) {
        /* still running when the slice's budget is spent: parked for the next slice (its inputs and destination come along) */
        const int dpos = atomicAdd(A.dst.count, 1);
        if (dpos >= A.dst.cap) keep = true;
        else {
          atomicAdd(A.tally + 2, 1);
          double *dk = A.dst.park + dpos;
          const int64_t dl = A.dst.cap;
          S.park([dk, dl](int q) -> double & { return dk[q * dl]; }, attempt, it_total);
          rows_copy(dk, dl, pk, lp, kParkRows, kTailRows);
          ws.stage_drain();
          tile_from_column<R>((R *)A.dst.iter + (int64_t)(dpos >> 6) * M * FL::IT_SZ * 64 + (dpos & 63), ws, S.cur ? FL::IT1 : FL::IT0, M);
          have = false;
        }
      }
    }
    ++wp;
  }
  /* The wave that leaves last closes the slice (no copies or memsets behind it on the stream: on a full device every one of those
   * little launches waits for a free SIMD): it reports the counts to the host, and resets every counter for its next user --
   * the list this slice has read is the next slice's destination, a fresh queue it has absorbed is free for its next batch,
   * its own work counters serve the slice that takes this ring position again.  (Every wave's additions to the counters have
   * returned before it arrives here: their results decided what it did.) */
  if (threadIdx.x == 0) {
    atomicMax(A.tally + 3, wp);
    const int arrived = atomicAdd(A.done, 1);
    if (arrived == (int)gridDim.x - 1) {
      for (int q = 0; q < 4; q++) {
        const int v = __hip_atomic_load(A.tally + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(A.res + (q < 3 ? 25 + q : 31), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(A.tally + q, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const int dc = __hip_atomic_load(A.dst.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(A.res, dc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      for (int j = 1; j < A.n_src; j++) {
        const int c = __hip_atomic_load(A.src[j].count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(A.res + j, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      for (int j = 0; j < A.n_src; j++) __hip_atomic_store(A.src[j].count, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(A.take, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(A.done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

/* Small launches: the N-step variables of every instance resident in LDS (mpc::LdsWorkspace), one instance per lane,
 * LANES instances per workgroup (= per wave), no workspace in HBM at all.  Same solver, bitwise the same results. */
template <class R, int LANES>
__global__ __launch_bounds__(kBlock, 1) void mpc_solve_lds_kernel(
    const MpcParams P, const int64_t B, const int64_t ld, const int64_t ldo, const R *__restrict__ state,
    const R *__restrict__ coeffs, const R *__restrict__ yaw_lo, const R *__restrict__ yaw_hi,
    const R *__restrict__ weights, R *__restrict__ out, R *__restrict__ traj,
    int32_t *__restrict__ status, int32_t *__restrict__ iters) {
  extern __shared__ double smem[];
  using WS = mpc::LdsWorkspace<R, LANES>;
  using SV = mpc::Solver<WS, R>;
  const int lane = threadIdx.x;
  const int64_t i = (int64_t)blockIdx.x * LANES + lane;
  if (lane >= LANES || i >= B) return;                 /* no barriers in this kernel: lanes are independent */
  WS ws;
  ws.base = (typename WS::lreal *)smem;
  ws.lane = lane;
  SV S(P, ws);
  R st[6], cf[MPC_NCOEF], w[MPC_NW];
#pragma unroll
  for (int q = 0; q < 6; q++) st[q] = state[q * ld + i];
#pragma unroll
  for (int q = 0; q < MPC_NCOEF; q++) cf[q] = coeffs[q * ld + i];
#pragma unroll
  for (int q = 0; q < MPC_NW; q++) w[q] = weights ? weights[q * ld + i] : (R)P.weights[q];
  int r = S.setup(st, cf, yaw_lo[i], yaw_hi[i], w, true);
  if (r == MPC_STATUS_SUCCESS) r = S.solve();
  R *o = out + i;
  R *t = traj ? traj + i : nullptr;
  const int64_t l = ldo;
  S.unpack([o, l](int q) -> R & { return o[q * l]; }, [t, l](int q) -> R & { return t[q * l]; }, traj != nullptr, yaw_lo[i], yaw_hi[i]);
  status[i] = r;
  if (iters) iters[i] = S.iters;
}

/* ONE INSTANCE PER WAVEFRONT, or per LPI = 16 / 32 neighbouring lanes of one (small launches: one MPC::solve() per telemetry
 * message is the reference's own use).  The instance's N-step variables live in the workgroup's LDS ([stage][field][instance],
 * 3.7 KB per instance at N = 10 in fp64); every lane of the group
 * runs the solver's state machine on them -- the decisions are wave-uniform -- and the sweeps share their work between the
 * lanes (mpc::Solver<WS, R, true>: backward_wave, forward_wave and the wave form of costate_trial in mpc_core.h). */
template <class R, int LPI>
__global__ __launch_bounds__(kBlock, 1) void mpc_solve_wave_kernel(
    const MpcParams P, const int64_t B, const int64_t ld, const int64_t ldo, const R *__restrict__ state,
    const R *__restrict__ coeffs, const R *__restrict__ yaw_lo, const R *__restrict__ yaw_hi,
    const R *__restrict__ weights, R *__restrict__ out, R *__restrict__ traj,
    int32_t *__restrict__ status, int32_t *__restrict__ iters) {
  extern __shared__ double smem[];
  constexpr int G = 64 / LPI;                       /* instances per wavefront: each on LPI neighbouring lanes */
  using WS = mpc::LdsWorkspace<R, G>;
  using SV = mpc::Solver<WS, R, LPI>;
  const int group = threadIdx.x / LPI;
  const int64_t i = (int64_t)blockIdx.x * G + group;
  if (i >= B) return;                               /* (whole groups: nobody reads their lanes) */
  WS ws;
  ws.base = (typename WS::lreal *)smem;
  ws.lane = group;                                  /* every lane of the group addresses the group's instance */
  SV S(P, ws);
  S.wlane = threadIdx.x % LPI; S.wbase = group * LPI;
  R st[6], cf[MPC_NCOEF], w[MPC_NW];
#pragma unroll
  for (int q = 0; q < 6; q++) st[q] = state[q * ld + i];
#pragma unroll
  for (int q = 0; q < MPC_NCOEF; q++) cf[q] = coeffs[q * ld + i];
#pragma unroll
  for (int q = 0; q < MPC_NW; q++) w[q] = weights ? weights[q * ld + i] : (R)P.weights[q];
  int r = S.setup(st, cf, yaw_lo[i], yaw_hi[i], w, true);
  if (r == MPC_STATUS_SUCCESS) r = S.solve();
  if (S.wlane == 0) {
    R *o = out + i;
    R *t = traj ? traj + i : nullptr;
    const int64_t l = ldo;
    S.unpack([o, l](int q) -> R & { return o[q * l]; }, [t, l](int q) -> R & { return t[q * l]; }, traj != nullptr, yaw_lo[i], yaw_hi[i]);
    status[i] = r;
    if (iters) iters[i] = S.iters;
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

/* Statistics of a batch, accumulated into handle-owned memory right behind the solve (mpc_get_stats never touches
 * the caller's arrays again): acc[0..4] = instances per status code 0..3 and "any other", acc[5] = sum of iterations,
 * acc[6] = max, acc[7] = pending, acc[8] = MPC_STATUS_ACCEPTABLE. */
constexpr int kStatWords = 10;
__global__ __launch_bounds__(256) void mpc_stats_kernel(int64_t B, const int32_t *__restrict__ status, const int32_t *__restrict__ iters,
                                                        unsigned long long *__restrict__ acc) {
  __shared__ unsigned int cnt[6];
  __shared__ unsigned long long isum;
  __shared__ int imax;
  if (threadIdx.x < 6) cnt[threadIdx.x] = 0;
  if (threadIdx.x == 0) { isum = 0; imax = 0; }
  __syncthreads();
  unsigned int my[6] = {0, 0, 0, 0, 0, 0}, pend = 0;
  unsigned long long ms = 0;
  int mm = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t st = status[i];
    if (st == MPC_STATUS_PENDING) { ++pend; continue; }   /* deferred: counted, its iterations are not final */
    const int c = st == MPC_STATUS_ACCEPTABLE ? 5 : ((st >= 0 && st < 4) ? st : 4);
#pragma unroll
    for (int q = 0; q < 6; q++) my[q] += (c == q);
    const int it = iters ? iters[i] : 0;
    ms += (unsigned long long)it;
    mm = it > mm ? it : mm;
  }
#pragma unroll
  for (int q = 0; q < 6; q++) if (my[q]) atomicAdd(&cnt[q], my[q]);
  if (ms) atomicAdd(&isum, ms);
  if (mm) atomicMax(&imax, mm);
  if (pend) atomicAdd(&acc[7], (unsigned long long)pend);
  __syncthreads();
  if (threadIdx.x < 5 && cnt[threadIdx.x]) atomicAdd(&acc[threadIdx.x], (unsigned long long)cnt[threadIdx.x]);
  if (threadIdx.x == 5 && isum) atomicAdd(&acc[5], isum);
  if (threadIdx.x == 6 && imax) atomicMax(&acc[6], (unsigned long long)imax);
  if (threadIdx.x == 7 && cnt[5]) atomicAdd(&acc[8], (unsigned long long)cnt[5]);
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

/* One pool of workspace tiles per (device, tile size), shared by every handle created with MPC_TILE_POOL=1 */
struct MpcTilePool {
  int device = 0;
  size_t tile_bytes = 0;
  int tiles = 0, words = 0;        /* per XCD */
  void *base = nullptr;
  unsigned long long *bits = nullptr;
  int refs = 0;
};
static std::mutex g_pool_mutex;
static std::vector<MpcTilePool *> g_pools;

static MpcTilePool *pool_acquire(int device, size_t tile_bytes, int tiles_per_xcd) {
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  for (MpcTilePool *p : g_pools)
    if (p->device == device && p->tile_bytes == tile_bytes && p->tiles == tiles_per_xcd) { ++p->refs; return p; }
  MpcTilePool *p = new MpcTilePool();
  p->device = device; p->tile_bytes = tile_bytes; p->tiles = tiles_per_xcd; p->words = (tiles_per_xcd + 63) / 64;
  /* every XCD's words on cache lines of their own; the last two words of an XCD are its usage record */
  p->words = (p->words + 2 + 15) / 16 * 16;
  if (hipMalloc(&p->base, tile_bytes * (size_t)tiles_per_xcd * 8) != hipSuccess ||
      hipMalloc((void **)&p->bits, sizeof(unsigned long long) * (size_t)p->words * 8) != hipSuccess) {
    if (p->base) (void)hipFree(p->base);
    delete p;
    return nullptr;
  }
  std::vector<unsigned long long> init((size_t)p->words * 8, 0ull);
  for (int x = 0; x < 8; x++)
    for (int t = 0; t < tiles_per_xcd; t++) init[(size_t)x * p->words + t / 64] |= 1ull << (t % 64);
  if (hipMemcpy(p->bits, init.data(), init.size() * sizeof(unsigned long long), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(p->base); (void)hipFree(p->bits); delete p; return nullptr;
  }
  p->refs = 1;
  g_pools.push_back(p);
  return p;
}
static void pool_release(MpcTilePool *p) {
  if (!p) return;
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  if (--p->refs > 0) return;
  for (size_t i = 0; i < g_pools.size(); i++) if (g_pools[i] == p) { g_pools.erase(g_pools.begin() + i); break; }
  (void)hipFree(p->base); (void)hipFree(p->bits);
  delete p;
}

struct MpcHandle {
  MpcParams params;
  int device = 0;
  int64_t max_batch = 0;
  int64_t ws_stride = 0;   /* reals (double, or float for MPC_PRECISION_F32) per wavefront tile of the workspace */
  bool staging = true;
  int lds_lanes = 0;       /* instances per workgroup of the LDS-resident kernel (0: N too large for it, or MPC_LDS=0) */
  int64_t wave_max_batch = 0;   /* launches up to this size run one instance per wavefront (mpc_solve_wave_kernel); 0: never */
  int64_t wave_whole_max = 16;  /* ... and up to this size an instance gets the whole wave (v_readlane) even where 16 lanes would do */
  int64_t lds_max_batch = 0;   /* launches up to this size take the LDS-resident kernel: lds_lanes x number of CUs */
  bool mixed = false;      /* two phases per solve: fp32 up to MpcParams.mixed_switch_mu, then fp64 to tol (f32_finish on an F32 handle,
                            * f64_f32_start on an F64 handle) */
  int64_t ws_stride_f32 = 0, ws_stride_f64 = 0;   /* reals per wavefront tile of either record layout */
  bool occ2 = false;       /* fp32: MPC_F32_OCC=2 selects the build held to 256 registers (two waves per SIMD, ~110 spill reloads per
                            * pass); the unconstrained build (296 registers, one wave per SIMD) measured 7 % faster on the final code */
  int64_t io_stride = 0;   /* leading dimension of the handle's own staging arrays */
  void *ws = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  /* staging of the host-pointer entry point: ONE device block and its pinned host mirror, rows with a per-call
   * leading dimension: in = state[6] coeffs[5] ylo yhi weights[12] (25 rows) | out = out[9] traj[2N] | one row holding
   * status and iters (int32 each) -- so a call is one copy in, the launch, one copy out */
  void *d_io = nullptr, *h_io = nullptr;
  unsigned long long *d_stats = nullptr;   /* [kStatWords] statistics of the last batch (mpc_stats_kernel) */
  hipEvent_t ev_stats = nullptr;
  bool have_stats = false;
  double *d_run = nullptr;    /* run(): pre[15] rows */
  double *d_run9 = nullptr;   /* run(): solve()'s 9 rows, caller's leading dimension */
  int64_t run9_ld = 0;
  int32_t *d_status = nullptr, *d_iters = nullptr, *d_rstat = nullptr, *d_counter = nullptr;
  int64_t counter_seq = 0;    /* solve calls that used a counter block so far */
  /* statistics are gathered when mpc_get_stats asks for them (a kernel per batch on the launch stream costs the serving loop a
   * few per cent): what the most recent call wrote, and the event behind it */
  const int32_t *st_status = nullptr, *st_iters = nullptr;
  int64_t st_B = 0;
  hipEvent_t st_ev = nullptr;
  bool stats_pending = false;
  int inst_per_lane = 1;      /* MPC_INSTANCES_PER_LANE: waves = ceil(B / 64 / inst_per_lane) */
  int refill_min = 16, refill_wait = 8;   /* hand-over policy of the persistent kernel (MpcPhase), MPC_REFILL_MIN / MPC_REFILL_WAIT */
  int refill_floor = 0, refill_floor_f32 = 0;   /* MPC_REFILL_FLOOR / MPC_REFILL_FLOOR_F32 (the fp32 phase of a mixed solve) */
  int compact_gap = 0;        /* MpcParams.lane_compact, or MPC_LANE_COMPACT in the environment (measurement aid): see MpcPhase.compact_gap */
  bool compact_env = false;
  int compact_cooldown = 2;   /* MPC_LANE_COMPACT_COOLDOWN */
  int64_t compact_min_batch = 8192;   /* smaller launches are latency-bound: the moves cost more than the lines they save */
  int finish_div = 1;         /* mixed precision: the fp64 phase runs ceil(waves / finish_div) waves whose lanes take the promoted
                               * instances in turn (MPC_FINISH_DIV) */
  int finish_refill_min = 16, finish_refill_wait = 8;   /* its hand-over policy (MPC_FINISH_REFILL_MIN / _WAIT) */
  /* multi-phase solve: second workspace, two parked-instance lists and two sets of scalars (allocated on first use) */
  int n_cuts = 0;             /* MpcParams.pass_cut + pass_cut_next[], or MPC_PASS_CUT=a,b,c,d in the environment (none = single launch) */
  int cuts[kMaxCuts] = {0, 0, 0, 0};
  int64_t two_phase_min = 8192;
  void *ws2 = nullptr;
  double *d_park = nullptr;   /* [2][PARK_ROWS][io_stride] */
  void *d_piter = nullptr;    /* mixed precision: the promoted iterates, [io_stride / 64][N-1][IT_SZ of the fp32 record][64] floats */
  int promote_cap = 0;        /* MPC_PROMOTE_CAP (measurement aid; 0 = Solver::kPromoteIterCap) */
  bool promote_buffer = false, promote_env = false;   /* MpcParams.f32_phase_refill (MPC_PROMOTE_BUFFER in the environment overrides: measurement aid) */
  int32_t *d_list = nullptr;  /* [2][2][io_stride]: instance, source column */
  MpcTilePool *pool = nullptr;   /* MPC_TILE_POOL=1 */
  /* deferred tails (MpcParams.tail_cut > 0; allocated on first use): see "tail slices" below */
  struct TailSlot {                /* one per batch whose stragglers may be outstanding (ring of tail_ring) */
    int64_t batch_id = 0;          /* 0: never used */
    bool final_ = true;
    int64_t deferred = -1;         /* instances the batch handed over (known once the absorbing slice has completed) */
    int64_t B = 0;
  };
  struct FreshQ {                  /* the queue a launch hands its stragglers to, until a slice absorbs it (ring of kFreshRing) */
    int64_t batch_id = 0;
    int slot = 0;
    int state = 0;                 /* 0 free, 1 filled by a launch (bulk recorded), 2 absorbed by slice `slice` */
    int64_t slice = -1;
    hipEvent_t bulk = nullptr;
  };
  struct BatchRec {                /* the last kBatchRecs solve calls: how mpc_tail_wait resolves an id */
    int64_t id = 0;
    int kind = 0;                  /* 0 not deferring: final behind `ev`; 1 deferring: final per its slot */
    int slot = 0;
    hipEvent_t ev = nullptr;
  };
  static constexpr int kBatchRecs = 1024;
  struct SliceRes { int32_t count[32]; };     /* [0]: survivors the slice left; [j]: entries of its source j */
  static_assert(kSliceMaxSrc <= 25, "SliceRes: [25..31] carry the tallies and the damaged-entry report");
  bool tail_ready = false;
  bool tail_double = true;     /* the solver of the tail slices: fp64, or fp32 on a pure MPC_PRECISION_F32 handle */
  int tail_ring = 0, tail_waves = 256, tail_priority = 0, slice_passes = 16;
  /* a slice's grid: a lane per survivor (they are the long chains: most use the slice's whole budget) and one per fresh_div fresh
   * entries (an instance just over the cut needs a few more passes, so a lane works off several of them in turn) */
  int fresh_div = 4;
  int64_t tail_cap = 0, surv_cap = 0, tail_min_batch = 4096;
  hipStream_t tail_stream = nullptr;
  MpcTailQ fq_dev[kFreshRing] = {}, surv_dev[2] = {};   /* device storage of the fresh queues and the two survivor lists */
  int32_t *d_tcount = nullptr;   /* [kFreshRing + 2 + 6 kSliceRing]: counts of the fresh queues, of the survivor lists, slice work and exit counters, slice tallies */
  int32_t *d_remaining = nullptr;
  long long *h_final = nullptr;  /* pinned: [tail_ring] the id of the batch that has become final in each slot (written by the slices) */
  SliceRes *h_res = nullptr;     /* pinned: [kSliceRing], written by each slice's last wave */
  void *tail_ws = nullptr;
  TailSlot tslot[kTailMaxRing];
  FreshQ fq[kFreshRing];
  BatchRec *brec = nullptr;
  hipEvent_t slice_ev[kSliceRing] = {};
  struct Absorbed { int fq; int slot; int64_t batch; };
  /* measurement aid: MPC_TAIL_TRACE=<file> appends one line per retired slice (see tail_retire) */
  int64_t slice_t0[kSliceRing] = {}, slice_est[kSliceRing] = {}, slice_surv_in[kSliceRing] = {};
  int slice_waves[kSliceRing] = {};
  int slice_nabs[kSliceRing] = {};            /* fresh queues slice k % kSliceRing absorbed, and which (queue, its batch and slot) */
  Absorbed slice_abs[kSliceRing][kFreshRing] = {};
  int64_t n_slice = 0, n_slice_done = 0;      /* slices launched / retired */
  int64_t surv_last = 0;                      /* survivors the most recent retired slice left */
  int64_t fresh_avg = 64;                     /* running estimate of a batch's deferred instances (sizes a slice's grid) */
  int64_t n_not_final = 0;                    /* deferring batches not yet known to be final */
  int64_t n_throttled = 0;                    /* batches that ran without deferral because the survivors' list was filling up */
  int64_t n_overflow = 0;                     /* batches that handed over more than their fresh queue holds */
  /* MpcParams.tail_cut = MPC_TAIL_AUTO: the handle's own choice -- it starts from a cut at 20 passes for horizons up to N = 12, 24
   * beyond (the long horizons need more iterations), moves it out while more than 8 % of a batch are handed over (or a
   * fresh queue overflows) and back while less than 1.5 % is; and a wave does not wait for its last 4 lanes (measured on the survey population:
   * cuts of 16 ... 32 within 5 % of each other, 20 best; few 0 / 2 / 4 / 8: 43.4 / 43.8 / 44.7 / 44.8 M solves/s).  The arithmetic
   * of an instance does not depend on where it is carried on, so these change timing only.  auto_share: running mean of the
   * share of a batch that was handed over, in 1/65536 (mpc_tail_info). */
  int64_t hold_batches = 0;
  int tail_few = 4, tail_few_from = 8;        /* MPC_TAIL_FEW / MPC_TAIL_FEW_FROM (see MpcPhase.tail_few) */
  int auto_cut = 20, auto_base = 20;
  int64_t auto_share = -1;
  int64_t auto_lo = 983, auto_hi = 5243;      /* 1.5 % and 8 % of a batch, in 1/65536 (MPC_TAIL_AUTO_LO / _HI) */
  int64_t batch_seq = 0;         /* id of the most recent batch (every solve call counts) */
  int64_t n_deferred = 0;        /* deferring batches so far: batch k of them uses slot k % tail_ring and fresh queue k % kFreshRing */
  double *d_tel = nullptr;       /* mpc_telemetry_batch_host: device staging, grown on demand */
  size_t tel_bytes = 0;
  /* last call */
  int64_t last_B = 0;
  bool timed = false;
};

/* the cut schedule of the multi-phase solve: MpcParams.pass_cut, pass_cut_next[0..2] (a zero ends the list), overridden
 * by MPC_PASS_CUT=a[,b[,c[,d]]] in the environment */
static void set_wave_limit(MpcHandle *h, const MpcParams *p) {
  h->wave_max_batch = p->wave_max_batch == 0 ? 1024 : (p->wave_max_batch < 0 ? 0 : p->wave_max_batch);
  if (const char *e = getenv("MPC_WAVE_MAX_BATCH")) h->wave_max_batch = atoll(e);      /* (A/B measurements) */
  if (const char *e = getenv("MPC_WAVE_WHOLE_MAX")) h->wave_whole_max = atoll(e);
  if (p->f64_f32_start == MPC_F32_START_ON || (p->precision == MPC_PRECISION_F32 && p->f32_finish != 0)) h->wave_max_batch = 0;
}

static void set_cuts(MpcHandle *h, const MpcParams *p) {
  h->n_cuts = 0;
  const int32_t given[kMaxCuts] = {p->pass_cut, p->pass_cut_next[0], p->pass_cut_next[1], p->pass_cut_next[2]};
  for (int q = 0; q < kMaxCuts && given[q] > 0; q++) h->cuts[h->n_cuts++] = given[q];
  if (const char *e = getenv("MPC_PASS_CUT")) {
    h->n_cuts = 0;
    for (const char *c = e; *c && h->n_cuts < kMaxCuts;) {
      const int v = atoi(c);
      if (v <= 0) break;
      h->cuts[h->n_cuts++] = v;
      while (*c && *c != ',') ++c;
      if (*c == ',') ++c;
    }
  }
}

/* two phases per solve (fp32 iterations, fp64 finish)?  F32 handles: f32_finish; F64 handles: f64_f32_start = 1, or 2
 * (MPC_F32_START_AUTO; the default is 0 = off) from the horizon at which the workspace of a full device no longer lives in the
 * Infinity Cache */
static bool wants_mixed(const MpcParams *p, int64_t max_batch) {
  (void)max_batch;
  if (p->precision == MPC_PRECISION_F32) return p->f32_finish != 0;
  return p->f64_f32_start == 1 || (p->f64_f32_start == MPC_F32_START_AUTO && p->N >= MPC_F32_START_AUTO_N);
}

static int validate_params(const MpcParams *p) {
  if (!p) return MPC_ERR_INVALID;
  if (p->abi_version != MPC_ABI_VERSION) { g_last_error = "MpcParams.abi_version mismatch"; return MPC_ERR_INVALID; }
  if (p->N < 3 || p->N > MPC_MAX_N) { g_last_error = "N out of range"; return MPC_ERR_INVALID; }
  if (!(p->dt > 0) || !(p->Lf > 0) || !(p->max_speed > 0) || !(p->max_steering > 0)) { g_last_error = "bad dt/Lf/limits"; return MPC_ERR_INVALID; }
  if (p->n_steers < 0 || p->n_steers > MPC_MAX_TABLE || p->n_steer_speeds < 1 || p->n_steer_speeds > MPC_MAX_TABLE) { g_last_error = "bad steer tables"; return MPC_ERR_INVALID; }
  if (p->n_yaw_changes < 0 || p->n_yaw_changes > MPC_MAX_TABLE || p->n_yaw_change_speeds < 0 || p->n_yaw_change_speeds > MPC_MAX_TABLE) { g_last_error = "bad yaw-change tables"; return MPC_ERR_INVALID; }
  if (!(p->out_step_tol >= 0)) { g_last_error = "bad out_step_tol"; return MPC_ERR_INVALID; }
  if (!(p->bound_relax_factor >= 0 && p->bound_relax_factor <= 1e-3)) { g_last_error = "bad bound_relax_factor (IPOPT default 1e-8)"; return MPC_ERR_INVALID; }
  if (p->branch_mode != MPC_BRANCH_FROZEN) { g_last_error = "branch_mode LIVE is not implemented on the device path"; return MPC_ERR_UNSUPPORTED; }
  if (p->precision != MPC_PRECISION_F64 && p->precision != MPC_PRECISION_F32) { g_last_error = "unknown precision"; return MPC_ERR_INVALID; }
  if (p->precision == MPC_PRECISION_F32 && !(p->tol_f32 >= 1e-5)) { g_last_error = "tol_f32 below 1e-5 is beyond single precision"; return MPC_ERR_INVALID; }
  if (p->max_iter < 1 || !(p->tol > 0)) { g_last_error = "bad max_iter/tol"; return MPC_ERR_INVALID; }
  if (p->tail_cut < MPC_TAIL_AUTO || p->tail_ring < 0 || p->tail_capacity < 0) { g_last_error = "bad tail_cut/tail_ring/tail_capacity"; return MPC_ERR_INVALID; }
  if (p->f64_f32_start < 0 || p->f64_f32_start > MPC_F32_START_AUTO) { g_last_error = "f64_f32_start must be 0 (off), 1 (on) or 2 (auto)"; return MPC_ERR_INVALID; }
  if (p->lane_compact < MPC_LANE_COMPACT_AUTO || p->lane_compact > 7) { g_last_error = "lane_compact must be -1 (auto), 0 (off) .. 7"; return MPC_ERR_INVALID; }
  return MPC_OK;
}

extern "C" int mpc_abi_version(void) { return MPC_ABI_VERSION; }

/* state of the handle's tile pool: per XCD (free tiles now, tiles in the pool, claims so far, highest tile number used + 1) */
extern "C" int mpc_debug_tile_pool(MpcHandle *h, int64_t *out32) {
  if (!h || !out32) return MPC_ERR_INVALID;
  for (int q = 0; q < 32; q++) out32[q] = 0;
  if (!h->pool) { g_last_error = "no tile pool on this handle (MPC_TILE_POOL=1 at mpc_create)"; return MPC_ERR_UNSUPPORTED; }
  MPC_ON_DEVICE(h);
  MPC_HIP_CHECK(hipDeviceSynchronize());
  std::vector<unsigned long long> bits((size_t)h->pool->words * 8);
  MPC_HIP_CHECK(hipMemcpy(bits.data(), h->pool->bits, bits.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  for (int x = 0; x < 8; x++) {
    const unsigned long long *b = bits.data() + (size_t)x * h->pool->words;
    int64_t fr = 0;
    for (int t = 0; t < h->pool->tiles; t++) fr += (b[t / 64] >> (t % 64)) & 1ull;
    out32[4 * x + 0] = fr; out32[4 * x + 1] = h->pool->tiles;
    out32[4 * x + 2] = (int64_t)b[h->pool->words - 2]; out32[4 * x + 3] = (int64_t)b[h->pool->words - 1];
  }
  return MPC_OK;
}
extern "C" const char *mpc_last_error(void) { return g_last_error.c_str(); }

extern "C" int mpc_create(const MpcParams *p, int device, int64_t max_batch, MpcHandle **out) {
  if (!out || max_batch < 1) return MPC_ERR_INVALID;
  int rc = validate_params(p);
  if (rc != MPC_OK) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_last_error = "no HIP device"; return MPC_ERR_NO_DEVICE; }
  if (device < 0) MPC_HIP_CHECK(hipGetDevice(&device));
  if (device >= ndev) { g_last_error = "device index out of range"; return MPC_ERR_NO_DEVICE; }
  DeviceGuard guard_(device);   /* the caller's current device is restored on return */
  MPC_HIP_CHECK(guard_.err);
  hipDeviceProp_t prop;
  MPC_HIP_CHECK(hipGetDeviceProperties(&prop, device));
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
    g_last_error = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
    return MPC_ERR_NO_DEVICE;
  }
  MpcHandle *h = new MpcHandle();
  /* a failing HIP call from here on must not leak the handle */
#define MPC_CREATE_CHECK(expr)                                                          \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) {                                                             \
      g_last_error = std::string(#expr) + ": " + hipGetErrorString(e_);                 \
      mpc_destroy(h);                                                                   \
      return MPC_ERR_HIP;                                                               \
    }                                                                                   \
  } while (0)
  h->params = *p; h->device = device; h->max_batch = max_batch;
  /* Launch shape: each workgroup is one wave; the fp64 kernel needs ~390 registers, so at most one wave runs
   * per SIMD (4 per CU), and the 36 KB of staging LDS per wave fit four times into a CU's 160 KB (fp32: 256
   * registers, two waves per SIMD, 20 KB each).
   * MPC_STAGING=0 selects the variant with ordinary loads (for A/B measurements). */
  h->staging = true;
  if (const char *e = getenv("MPC_STAGING")) h->staging = atoi(e) != 0;
  const bool f32 = p->precision == MPC_PRECISION_F32;
  if (const char *e = getenv("MPC_F32_OCC")) h->occ2 = atoi(e) == 2;
  if (f32) {
    MPC_CREATE_CHECK(hipFuncSetAttribute((const void *)mpc_solve_kernel<true, float, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCu));
    MPC_CREATE_CHECK(hipFuncSetAttribute((const void *)mpc_solve_kernel<true, float, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCu));
  } else MPC_CREATE_CHECK(hipFuncSetAttribute((const void *)mpc_solve_kernel<true, double, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCu));
  h->ws_stride = mpc::workspace_fields_per_instance(p->N, f32, p->initial_state_rows != 0) * 64;   /* reals per wavefront tile */
  h->ws_stride_f32 = mpc::workspace_fields_per_instance(p->N, true, p->initial_state_rows != 0) * 64;
  h->ws_stride_f64 = mpc::workspace_fields_per_instance(p->N, false, p->initial_state_rows != 0) * 64;
  h->mixed = wants_mixed(p, max_batch);
  /* MPC_MIXED=0/1 overrides the parameter for every handle of the process: how the whole parity suite was run with the fp32
   * start forced on (tools/r03_session.sh p); a measurement aid, not an interface */
  if (const char *e = getenv("MPC_MIXED")) h->mixed = atoi(e) != 0;
  if (h->mixed) {
    MPC_CREATE_CHECK(hipFuncSetAttribute((const void *)mpc_solve_kernel<true, double, 1, float, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCu));
    MPC_CREATE_CHECK(hipFuncSetAttribute((const void *)mpc_solve_kernel<true, float, 1, double, double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCu));
    MPC_CREATE_CHECK(hipFuncSetAttribute((const void *)mpc_solve_kernel<true, double, 1, double, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCu));
    MPC_CREATE_CHECK(hipFuncSetAttribute((const void *)mpc_solve_kernel<true, float, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCu));
  }
  {
    /* LDS-resident kernel: as many instances per workgroup as 160 KB hold (32, 16 or 8); one workgroup per CU */
    const int64_t per_inst = mpc::workspace_fields_per_instance(p->N, f32, p->initial_state_rows != 0) * (int64_t)(f32 ? sizeof(float) : sizeof(double));
    for (int lanes : {32, 16, 8})
      if (lanes * per_inst <= kLdsPerCu) { h->lds_lanes = lanes; break; }
    /* Measured on MI355X (tools/batch_sweep.py, same box, B = 1 ... 16 384): the LDS-resident kernel is 5-9 % SLOWER than the
     * streaming kernel at every size -- a lone wave pays ~12 cycles of issue per ds_write_b64 for the 44 fields a stage
     * stores per iteration, where the streaming kernel's global stores are fire-and-forget, and the stage step is bound by
     * its dependent fp64 chain, not by the record's latency.  So it is opt-in: MPC_LDS=1. */
    const char *e_lds = getenv("MPC_LDS");
    if (!(e_lds && atoi(e_lds) == 1)) h->lds_lanes = 0;
    h->lds_max_batch = (int64_t)h->lds_lanes * prop.multiProcessorCount;
    const void *fn = nullptr;
    switch (h->lds_lanes) {
      case 32: fn = f32 ? (const void *)mpc_solve_lds_kernel<float, 32> : (const void *)mpc_solve_lds_kernel<double, 32>; break;
      case 16: fn = f32 ? (const void *)mpc_solve_lds_kernel<float, 16> : (const void *)mpc_solve_lds_kernel<double, 16>; break;
      case 8: fn = f32 ? (const void *)mpc_solve_lds_kernel<float, 8> : (const void *)mpc_solve_lds_kernel<double, 8>; break;
      default: break;
    }
    if (fn) MPC_CREATE_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCu));
    if (const char *e = getenv("MPC_LDS_MAX_BATCH")) h->lds_max_batch = atoll(e);
  }
  /* Small launches run ONE INSTANCE PER WAVEFRONT, or per 16 / 32 of its lanes (mpc_solve_wave_kernel): the lane kernel puts 64
   * instances into a wave, which is bound by the instructions it issues -- one MPC::solve() 0.68 ms; with the sweeps shared
   * between the lanes 0.30 ms, bitwise the same results (MpcParams.wave_max_batch: default 1 024 instances). */
  set_wave_limit(h, p);
  h->io_stride = (max_batch + 63) / 64 * 64;
  const size_t ws_bytes = (size_t)h->ws_stride * (size_t)(h->io_stride / 64) * (f32 ? sizeof(float) : sizeof(double));
  auto fail = [&](hipError_t e, const char *what) { g_last_error = std::string(what) + ": " + hipGetErrorString(e); mpc_destroy(h); return MPC_ERR_HIP; };
  hipError_t e;
  if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
  if ((e = hipEventCreate(&h->ev0)) != hipSuccess) return fail(e, "hipEventCreate");
  if ((e = hipEventCreate(&h->ev1)) != hipSuccess) return fail(e, "hipEventCreate");
  if ((e = hipMalloc((void **)&h->ws, ws_bytes)) != hipSuccess) return fail(e, "hipMalloc(workspace)");
  if ((e = hipMalloc((void **)&h->d_status, sizeof(int32_t) * h->io_stride)) != hipSuccess) return fail(e, "hipMalloc");
  if ((e = hipMalloc((void **)&h->d_iters, sizeof(int32_t) * h->io_stride)) != hipSuccess) return fail(e, "hipMalloc");
  if ((e = hipMalloc((void **)&h->d_counter, kCounterRing * kCounterInts * sizeof(int32_t))) != hipSuccess) return fail(e, "hipMalloc");
  /* (on the handle's own stream and waited for: a plain hipMemset of device memory is ordered on the null stream only, which the
   * non-blocking streams the launches run on do not wait for) */
  if ((e = hipMemsetAsync(h->d_counter, 0, kCounterRing * kCounterInts * sizeof(int32_t), h->stream)) != hipSuccess) return fail(e, "hipMemset");
  if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) return fail(e, "hipStreamSynchronize");
  if ((e = hipMalloc((void **)&h->d_stats, kStatWords * sizeof(unsigned long long))) != hipSuccess) return fail(e, "hipMalloc");
  if ((e = hipEventCreateWithFlags(&h->ev_stats, hipEventDisableTiming)) != hipSuccess) return fail(e, "hipEventCreate");
  set_cuts(h, p);
  if (const char *ep = getenv("MPC_TILE_POOL")) {
    if (atoi(ep) > 0) {
      /* tiles per XCD: its wave slots for this kernel (CUs / 8 XCDs x 4 SIMDs x waves per SIMD) and a margin */
      /* (a partition mode that exposes fewer XCDs than 8 only makes the pools larger than needed; a wave that finds its
       * pool empty keeps the tile of the handle's own workspace) */
      const int per_xcd = (prop.multiProcessorCount + 7) / 8 * 4 * ((f32 && h->occ2) ? 2 : 1);
      int extra = 16;
      if (const char *ex = getenv("MPC_TILE_POOL_EXTRA")) extra = atoi(ex);
      h->pool = pool_acquire(device, (size_t)h->ws_stride * (f32 ? sizeof(float) : sizeof(double)), per_xcd + (extra > 0 ? extra : 0));
      if (!h->pool) { g_last_error = "hipMalloc(tile pool)"; mpc_destroy(h); return MPC_ERR_HIP; }
    }
  }
  if (const char *e3 = getenv("MPC_PHASE_MIN_BATCH")) { h->two_phase_min = atoll(e3); if (h->two_phase_min < 1) h->two_phase_min = 1; }
  if (const char *e4 = getenv("MPC_REFILL_MIN")) { h->refill_min = atoi(e4); if (h->refill_min < 1) h->refill_min = 1; }
  if (const char *e5 = getenv("MPC_REFILL_WAIT")) { h->refill_wait = atoi(e5); if (h->refill_wait < 0) h->refill_wait = 0; }
  if (const char *e2 = getenv("MPC_INSTANCES_PER_LANE")) { h->inst_per_lane = atoi(e2); if (h->inst_per_lane < 1) h->inst_per_lane = 1; }
  h->compact_gap = p->lane_compact >= 0 ? p->lane_compact : (p->N >= 15 ? 1 : 2);
  if (const char *e9 = getenv("MPC_LANE_COMPACT")) { h->compact_gap = atoi(e9); h->compact_env = true; if (h->compact_gap < 0) h->compact_gap = 0; }
  if (const char *e10 = getenv("MPC_LANE_COMPACT_COOLDOWN")) { h->compact_cooldown = atoi(e10); if (h->compact_cooldown < 0) h->compact_cooldown = 0; }
  h->promote_buffer = p->f32_phase_refill != 0;
  if (const char *e11 = getenv("MPC_PROMOTE_BUFFER")) { h->promote_buffer = atoi(e11) != 0; h->promote_env = true; }
  if (const char *e12 = getenv("MPC_REFILL_FLOOR")) h->refill_floor = atoi(e12);
  if (const char *e13 = getenv("MPC_REFILL_FLOOR_F32")) h->refill_floor_f32 = atoi(e13);
  if (const char *e15 = getenv("MPC_PROMOTE_CAP")) h->promote_cap = atoi(e15);
  if (const char *e6 = getenv("MPC_FINISH_DIV")) { h->finish_div = atoi(e6); if (h->finish_div < 1) h->finish_div = 1; }
  if (const char *e7 = getenv("MPC_FINISH_REFILL_MIN")) { h->finish_refill_min = atoi(e7); if (h->finish_refill_min < 1) h->finish_refill_min = 1; }
  if (const char *e8 = getenv("MPC_FINISH_REFILL_WAIT")) { h->finish_refill_wait = atoi(e8); if (h->finish_refill_wait < 0) h->finish_refill_wait = 0; }
  *out = h;
  return MPC_OK;
}

static int tail_drain(MpcHandle *h);

extern "C" int mpc_set_params(MpcHandle *h, const MpcParams *p) {
  if (!h) return MPC_ERR_INVALID;
  int rc = validate_params(p);
  if (rc != MPC_OK) return rc;
  if (wants_mixed(p, h->max_batch) != h->mixed && !getenv("MPC_MIXED")) {
    g_last_error = "f32_finish / f64_f32_start cannot change on a live handle (they decide the workspaces)"; return MPC_ERR_INVALID;
  }
  if (p->N != h->params.N || p->precision != h->params.precision || (p->initial_state_rows != 0) != (h->params.initial_state_rows != 0)) {
    g_last_error = "N, precision and initial_state_rows cannot change on a live handle (they size the workspace)"; return MPC_ERR_INVALID;
  }
  if (h->tail_ready && (p->tail_ring != h->params.tail_ring || p->tail_capacity != h->params.tail_capacity)) {
    g_last_error = "tail_ring and tail_capacity cannot change once the tail queue exists"; return MPC_ERR_INVALID;
  }
  if (h->tail_ready) {
    /* stragglers of earlier batches are finished under the parameters their batch was issued with: every tail slice
     * takes the handle's parameters by value when it goes out, so what is still queued is finished first */
    MPC_ON_DEVICE(h);
    const int rf = tail_drain(h);
    if (rf != MPC_OK) return rf;
  }
  h->params = *p;
  set_cuts(h, p);
  set_wave_limit(h, p);
  if (!h->compact_env) h->compact_gap = p->lane_compact >= 0 ? p->lane_compact : (p->N >= 15 ? 1 : 2);
  if (!h->promote_env) h->promote_buffer = p->f32_phase_refill != 0;
  return MPC_OK;
}

extern "C" void mpc_destroy(MpcHandle *h) {
  if (!h) return;
  DeviceGuard guard_(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->ws) (void)hipFree(h->ws);
  if (h->d_io) (void)hipFree(h->d_io);
  if (h->h_io) (void)hipHostFree(h->h_io);
  if (h->d_stats) (void)hipFree(h->d_stats);
  if (h->ev_stats) (void)hipEventDestroy(h->ev_stats);
  if (h->d_run) (void)hipFree(h->d_run);
  if (h->d_run9) (void)hipFree(h->d_run9);
  if (h->d_status) (void)hipFree(h->d_status);
  if (h->d_iters) (void)hipFree(h->d_iters);
  if (h->d_rstat) (void)hipFree(h->d_rstat);
  if (h->d_counter) (void)hipFree(h->d_counter);
  if (h->ws2) (void)hipFree(h->ws2);
  pool_release(h->pool);
  if (h->d_park) (void)hipFree(h->d_park);
  if (h->d_list) (void)hipFree(h->d_list);
  if (h->d_piter) (void)hipFree(h->d_piter);
  if (h->d_tel) (void)hipFree(h->d_tel);
  if (h->tail_ready) (void)tail_drain(h);         /* stragglers still queued are finished: their batches' arrays may be read afterwards */
  if (h->tail_stream) (void)hipStreamSynchronize(h->tail_stream);
  for (void *q : {(void *)h->d_tcount, (void *)h->d_remaining, h->tail_ws, (void *)h->surv_dev[0].park, h->surv_dev[0].iter,
                  (void *)h->surv_dev[1].park, h->surv_dev[1].iter})
    if (q) (void)hipFree(q);
  for (int q = 0; q < kFreshRing; q++) {
    if (h->fq_dev[q].park) (void)hipFree(h->fq_dev[q].park);
    if (h->fq_dev[q].iter) (void)hipFree(h->fq_dev[q].iter);
    if (h->fq[q].bulk) (void)hipEventDestroy(h->fq[q].bulk);
  }
  if (h->h_final) (void)hipHostFree(h->h_final);
  if (h->h_res) (void)hipHostFree(h->h_res);
  for (int q = 0; q < kSliceRing; q++) if (h->slice_ev[q]) (void)hipEventDestroy(h->slice_ev[q]);
  if (h->brec) {
    for (int q = 0; q < MpcHandle::kBatchRecs; q++) if (h->brec[q].ev) (void)hipEventDestroy(h->brec[q].ev);
    delete[] h->brec;
  }
  if (h->tail_stream) (void)hipStreamDestroy(h->tail_stream);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

/* statistics of (status, iters) into the handle's own counters, behind whatever wrote them on stream s */
static int record_stats(MpcHandle *h, int64_t B, const int32_t *status, const int32_t *iters, hipStream_t s) {
  MPC_HIP_CHECK(hipMemsetAsync(h->d_stats, 0, kStatWords * sizeof(unsigned long long), s));
  unsigned grid = (unsigned)((B + 256 * 8 - 1) / (256 * 8));
  if (grid > 256) grid = 256;
  hipLaunchKernelGGL(mpc_stats_kernel, dim3(grid), dim3(256), 0, s, B, status, iters, h->d_stats);
  MPC_HIP_CHECK(hipGetLastError());
  MPC_HIP_CHECK(hipEventRecord(h->ev_stats, s));
  h->have_stats = true;
  return MPC_OK;
}

template <class R, int LANES>
static int launch_lds_n(MpcHandle *h, int64_t B, int64_t ld, int64_t ldo, const R *state, const R *coeffs, const R *yaw_lo,
                        const R *yaw_hi, const R *weights, R *out, R *traj, int32_t *status, int32_t *iters, hipStream_t s) {
  const size_t lds = (size_t)mpc::workspace_fields_per_instance(h->params.N, sizeof(R) == 4, h->params.initial_state_rows != 0) * sizeof(R) * LANES;
  const unsigned grid = (unsigned)((B + LANES - 1) / LANES);
  hipLaunchKernelGGL((mpc_solve_lds_kernel<R, LANES>), dim3(grid), dim3(kBlock), lds, s, h->params, B, ld, ldo, state, coeffs, yaw_lo, yaw_hi,
                     weights, out, traj, status, iters);
  MPC_HIP_CHECK(hipGetLastError());
  return MPC_OK;
}
template <class R>
static int launch_lds(MpcHandle *h, int64_t B, int64_t ld, int64_t ldo, const R *state, const R *coeffs, const R *yaw_lo,
                      const R *yaw_hi, const R *weights, R *out, R *traj, int32_t *status, int32_t *iters, hipStream_t s) {
  switch (h->lds_lanes) {
    case 32: return launch_lds_n<R, 32>(h, B, ld, ldo, state, coeffs, yaw_lo, yaw_hi, weights, out, traj, status, iters, s);
    case 16: return launch_lds_n<R, 16>(h, B, ld, ldo, state, coeffs, yaw_lo, yaw_hi, weights, out, traj, status, iters, s);
    default: return launch_lds_n<R, 8>(h, B, ld, ldo, state, coeffs, yaw_lo, yaw_hi, weights, out, traj, status, iters, s);
  }
}

/* ---- deferred tails: queues, tail slices, the pump, waiting for a batch -------------------------------------------------
 * A launch with a cut hands its stragglers to the FRESH QUEUE of its batch.  They are carried on by TAIL SLICES
 * (mpc_tail_slice_kernel) on the handle's own high-priority stream: slice k reads the survivors slice k-1 left and the fresh
 * queues of the batches whose launches have ended since, works for a bounded number of passes, and leaves its own survivors.
 * Slices are started by the PUMP (tail_pump), which every solve call and every wait runs: it retires the slices that have
 * completed (reads their counters and the final flags), and starts the next one while at most two are in flight.  Nothing
 * runs between calls: a batch is final when mpc_tail_wait / mpc_tail_poll / mpc_tail_stream_wait says so. */
static int tail_alloc_queue(MpcHandle *h, MpcTailQ &Q, int64_t cap, int32_t *count) {
  const bool f32 = !h->tail_double;
  const size_t real_bytes = f32 ? sizeof(float) : sizeof(double);
  const size_t it_sz = f32 ? (size_t)mpc::Fields<float>::IT_SZ : (size_t)mpc::Fields<double>::IT_SZ;
  Q.cap = (int32_t)cap; Q.count = count;
  if (!Q.park) MPC_HIP_CHECK(hipMalloc((void **)&Q.park, sizeof(double) * (size_t)kTailRows * (size_t)cap));
  if (!Q.iter) MPC_HIP_CHECK(hipMalloc(&Q.iter, (size_t)(cap / 64) * (size_t)(h->params.N - 1) * it_sz * 64 * real_bytes));
  return MPC_OK;
}

static int tail_prepare(MpcHandle *h) {
  if (h->tail_ready) return MPC_OK;
  const MpcParams &P = h->params;
  h->tail_double = P.precision != MPC_PRECISION_F32 || h->mixed;      /* a mixed-precision solve defers in its fp64 phase */
  const bool f32 = !h->tail_double;
  const size_t real_bytes = f32 ? sizeof(float) : sizeof(double);
  const int64_t tail_stride = f32 ? h->ws_stride_f32 : h->ws_stride_f64;
  h->tail_ring = P.tail_ring < 2 ? 2 : (P.tail_ring > kTailMaxRing ? kTailMaxRing : P.tail_ring);
  int64_t cap = P.tail_capacity > 0 ? P.tail_capacity : h->max_batch / 8;
  if (cap < 256) cap = 256;
  if (cap > h->io_stride) cap = h->io_stride;
  h->tail_cap = (cap + 63) / 64 * 64;
  /* survivors: everything the outstanding batches may have alive at once.  A slice that finds the list full keeps the
   * instance until it is solved, and the solve calls stop deferring while the list is more than half full. */
  int64_t sc = 8 * h->tail_cap;
  if (sc < 16384) sc = 16384;
  if (const char *e = getenv("MPC_TAIL_SURVIVORS")) { sc = atoll(e); if (sc < 64) sc = 64; }
  h->surv_cap = (sc + 63) / 64 * 64;
  if (const char *e = getenv("MPC_TAIL_WAVES")) { h->tail_waves = atoi(e); if (h->tail_waves < 1) h->tail_waves = 1; }
  if (const char *e = getenv("MPC_TAIL_MIN_BATCH")) { h->tail_min_batch = atoll(e); if (h->tail_min_batch < 1) h->tail_min_batch = 1; }
  if (const char *e = getenv("MPC_SLICE_PASSES")) { h->slice_passes = atoi(e); if (h->slice_passes < 1) h->slice_passes = 1; }
  if (const char *e = getenv("MPC_TAIL_HOLD_BATCHES")) h->hold_batches = atoll(e);
  if (const char *e = getenv("MPC_TAIL_FEW")) { h->tail_few = atoi(e); if (h->tail_few < 0) h->tail_few = 0; }
  if (const char *e = getenv("MPC_TAIL_FEW_FROM")) { h->tail_few_from = atoi(e); if (h->tail_few_from < 1) h->tail_few_from = 1; }
  if (const char *e = getenv("MPC_SLICE_FRESH_DIV")) { h->fresh_div = atoi(e); if (h->fresh_div < 1) h->fresh_div = 1; }
  h->auto_cut = h->auto_base = P.N <= 12 ? 20 : 24;
  if (const char *e = getenv("MPC_TAIL_AUTO_CUT")) { h->auto_cut = atoi(e); if (h->auto_cut < 4) h->auto_cut = 4; }
  if (const char *e = getenv("MPC_TAIL_AUTO_LO")) h->auto_lo = atoll(e);
  if (const char *e = getenv("MPC_TAIL_AUTO_HI")) h->auto_hi = atoll(e);
  /* The tail stream's priority: high (MPC_TAIL_PRIORITY=low|normal|high to measure the others).  A slice is a few dozen waves that
   * must find free SIMDs on a device the launches keep full: at normal priority a slice at N = 25 waited 6-12 ms for its 1.8 ms of
   * work (p90 of the retirement interval 15-30 ms), the stragglers' backlog grew until every buffer set was taken, and the long
   * windows read 5.4 M solves/s where the launches alone give 7.6 M; at high priority 7.1-8.8 M.  At N = 10 it makes no
   * difference (46.8 M either way; an earlier build of the slices lost 3 % with it). */
  int lo = 0, hi = 0;
  MPC_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));      /* lo = least, hi = greatest priority (numerically smaller) */
  int prio = hi;
  if (const char *e = getenv("MPC_TAIL_PRIORITY")) prio = !strcmp(e, "low") ? lo : (!strcmp(e, "normal") ? 0 : hi);
  h->tail_priority = prio;
  if (!h->tail_stream) MPC_HIP_CHECK(hipStreamCreateWithPriority(&h->tail_stream, hipStreamNonBlocking, prio));
  /* the counters start from zero BEFORE the first launch that adds to them is issued: cleared on the tail stream and waited for
   * below (a plain hipMemset runs on the null stream, which neither the caller's stream nor the tail stream waits for -- on a
   * device that other handles keep full its fill kernel can start after the launch that follows this call) */
  const int n_counts = kFreshRing + 2 + 6 * kSliceRing;
  if (!h->d_tcount) MPC_HIP_CHECK(hipMalloc((void **)&h->d_tcount, sizeof(int32_t) * n_counts));
  MPC_HIP_CHECK(hipMemsetAsync(h->d_tcount, 0, sizeof(int32_t) * n_counts, h->tail_stream));
  if (!h->d_remaining) MPC_HIP_CHECK(hipMalloc((void **)&h->d_remaining, sizeof(int32_t) * kTailMaxRing));
  MPC_HIP_CHECK(hipMemsetAsync(h->d_remaining, 0, sizeof(int32_t) * kTailMaxRing, h->tail_stream));
  if (!h->h_final) MPC_HIP_CHECK(hipHostMalloc((void **)&h->h_final, sizeof(long long) * kTailMaxRing, hipHostMallocDefault));
  if (!h->h_res) MPC_HIP_CHECK(hipHostMalloc((void **)&h->h_res, sizeof(MpcHandle::SliceRes) * kSliceRing, hipHostMallocDefault));
  memset(h->h_final, 0, sizeof(long long) * kTailMaxRing);
  memset(h->h_res, 0, sizeof(MpcHandle::SliceRes) * kSliceRing);
  for (int q = 0; q < kFreshRing; q++) {
    const int rc = tail_alloc_queue(h, h->fq_dev[q], h->tail_cap, h->d_tcount + q);
    if (rc != MPC_OK) return rc;
    if (!h->fq[q].bulk) MPC_HIP_CHECK(hipEventCreateWithFlags(&h->fq[q].bulk, hipEventDisableTiming));
  }
  for (int q = 0; q < 2; q++) {
    const int rc = tail_alloc_queue(h, h->surv_dev[q], h->surv_cap, h->d_tcount + kFreshRing + q);
    if (rc != MPC_OK) return rc;
  }
  if (!h->tail_ws) MPC_HIP_CHECK(hipMalloc((void **)&h->tail_ws, (size_t)tail_stride * (size_t)h->tail_waves * real_bytes));
  for (int q = 0; q < kSliceRing; q++)
    if (!h->slice_ev[q]) MPC_HIP_CHECK(hipEventCreateWithFlags(&h->slice_ev[q], hipEventDisableTiming));
  MPC_HIP_CHECK(hipStreamSynchronize(h->tail_stream));
  h->tail_ready = true;
  return MPC_OK;
}

/* the records of the last solve calls (allocated with the first one) */
static int batch_rec(MpcHandle *h, int64_t id, MpcHandle::BatchRec **out) {
  if (!h->brec) h->brec = new MpcHandle::BatchRec[MpcHandle::kBatchRecs];
  MpcHandle::BatchRec &R = h->brec[id % MpcHandle::kBatchRecs];
  if (!R.ev) MPC_HIP_CHECK(hipEventCreateWithFlags(&R.ev, hipEventDisableTiming));
  *out = &R;
  return MPC_OK;
}

static int tail_retire(MpcHandle *h, bool block, int *n_retired);

static FILE *g_tail_trace = nullptr;
static std::once_flag g_tail_trace_once;
static int64_t now_us() { return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static FILE *tail_trace() {
  std::call_once(g_tail_trace_once, [] { if (const char *e = getenv("MPC_TAIL_TRACE")) g_tail_trace = fopen(e, "a"); });
  return g_tail_trace;
}

/* Starts tail slice number h->n_slice.  It absorbs every filled fresh queue whose launch has ended (`force`: every filled one --
 * the slice then waits for those launches on the device). */
static int tail_launch_slice(MpcHandle *h, bool force) {
  while (h->n_slice - h->n_slice_done >= kSliceRing - 1) {      /* (the result blocks are a ring: never more than kSliceRing - 1 in flight) */
    int nr = 0;
    const int rc = tail_retire(h, true, &nr);
    if (rc != MPC_OK) return rc;
  }
  const int64_t k = h->n_slice;
  const int kr = (int)(k % kSliceRing);
  hipStream_t ts = h->tail_stream;
  MpcSliceArgs A;
  memset(&A, 0, sizeof(A));
  A.src[0] = h->surv_dev[k & 1];
  A.dst = h->surv_dev[(k + 1) & 1];
  A.n_src = 1;
  h->slice_nabs[kr] = 0;
  int64_t est = h->surv_last;
  /* oldest batch first */
  for (int step = 0; step < kFreshRing; step++) {
    int best = -1;
    for (int q = 0; q < kFreshRing; q++)
      if (h->fq[q].state == 1 && (best < 0 || h->fq[q].batch_id < h->fq[best].batch_id)) best = q;
    if (best < 0) break;
    MpcHandle::FreshQ &F = h->fq[best];
    if (!force) {
      const hipError_t e = hipEventQuery(F.bulk);
      if (e == hipErrorNotReady) { (void)hipGetLastError(); break; }      /* (launches of one handle end in the order they were issued, or nearly) */
      if (e != hipSuccess) { g_last_error = std::string("hipEventQuery: ") + hipGetErrorString(e); return MPC_ERR_HIP; }
    }
    MPC_HIP_CHECK(hipStreamWaitEvent(ts, F.bulk, 0));
    const int j = A.n_src++;
    A.src[j] = h->fq_dev[best]; A.fresh_slot[j] = F.slot; A.fresh_batch[j] = F.batch_id;
    F.state = 2; F.slice = k;
    h->slice_abs[kr][h->slice_nabs[kr]++] = MpcHandle::Absorbed{best, F.slot, F.batch_id};
    est += (h->fresh_avg + h->fresh_div - 1) / h->fresh_div;
  }
  A.budget = h->slice_passes; A.ring = h->tail_ring;
  h->h_res[kr].count[28] = 0;
  A.take = h->d_tcount + kFreshRing + 2 + kr; A.done = h->d_tcount + kFreshRing + 2 + kSliceRing + kr;
  A.remaining = h->d_remaining; A.final_id = h->h_final; A.res = h->h_res[kr].count;
  A.tally = h->d_tcount + kFreshRing + 2 + 2 * kSliceRing + 4 * kr;
  int64_t waves = (est + 63) / 64 + 1;
  if (waves > h->tail_waves) waves = h->tail_waves;
  if (waves < 1) waves = 1;
  const bool f32 = !h->tail_double, io32 = h->params.precision == MPC_PRECISION_F32;
  const int64_t tail_stride = f32 ? h->ws_stride_f32 : h->ws_stride_f64;
  if (f32 && h->occ2)
    hipLaunchKernelGGL((mpc_tail_slice_kernel<true, float, 2>), dim3((unsigned)waves), dim3(kBlock), staging_lds_bytes<float>(), ts, h->params, A, (float *)h->tail_ws, tail_stride);
  else if (f32)
    hipLaunchKernelGGL((mpc_tail_slice_kernel<true, float, 1>), dim3((unsigned)waves), dim3(kBlock), staging_lds_bytes<float>(), ts, h->params, A, (float *)h->tail_ws, tail_stride);
  else if (io32)       /* the fp64 phase of a mixed-precision solve: fp32 arrays at the ABI */
    hipLaunchKernelGGL((mpc_tail_slice_kernel<true, double, 1, float>), dim3((unsigned)waves), dim3(kBlock), staging_lds_bytes<double>(), ts, h->params, A, (double *)h->tail_ws, tail_stride);
  else
    hipLaunchKernelGGL((mpc_tail_slice_kernel<true, double, 1>), dim3((unsigned)waves), dim3(kBlock), staging_lds_bytes<double>(), ts, h->params, A, (double *)h->tail_ws, tail_stride);
  MPC_HIP_CHECK(hipGetLastError());
  /* (what the pump reads when the slice has completed -- the survivors it left, what each absorbed batch handed over, the final
   * flags -- the slice's last wave writes into pinned host memory itself) */
  MPC_HIP_CHECK(hipEventRecord(h->slice_ev[kr], ts));
  if (tail_trace()) { h->slice_t0[kr] = now_us(); h->slice_est[kr] = est; h->slice_surv_in[kr] = h->surv_last; h->slice_waves[kr] = (int)waves; }
  ++h->n_slice;
  return MPC_OK;
}

/* Retires the completed slices in order (`block`: waits for the oldest one in flight first).  Returns how many it retired. */
static int tail_retire(MpcHandle *h, bool block, int *n_retired) {
  *n_retired = 0;
  while (h->n_slice_done < h->n_slice) {
    const int kr = (int)(h->n_slice_done % kSliceRing);
    if (block) MPC_HIP_CHECK(hipEventSynchronize(h->slice_ev[kr]));
    else {
      const hipError_t e = hipEventQuery(h->slice_ev[kr]);
      if (e == hipErrorNotReady) { (void)hipGetLastError(); break; }
      if (e != hipSuccess) { g_last_error = std::string("hipEventQuery: ") + hipGetErrorString(e); return MPC_ERR_HIP; }
    }
    block = false;
    const MpcHandle::SliceRes &res = h->h_res[kr];
    if (res.count[28] != 0) {
      g_last_error = "tail slice: a damaged queue entry (slot " + std::to_string(res.count[29]) + ", instance " + std::to_string(res.count[30]) + ") was dropped";
      return MPC_ERR_HIP;
    }
    h->surv_last = res.count[0] < h->surv_cap ? res.count[0] : h->surv_cap;
    if (FILE *f = tail_trace()) {
      int64_t fresh = 0;
      for (int a = 0; a < h->slice_nabs[kr]; a++) fresh += res.count[1 + a];
      /* handle, slice, issued at / seen complete at (host, microseconds), waves, fresh queues absorbed, survivors known when it was issued,
       * fresh entries, survivors left, finished, moved on untouched, parked again, most passes of a wave, batches not final, slices in flight */
      fprintf(f, "%p %lld %lld %lld %d %d %lld %lld %d %d %d %d %d %d %lld\n", (void *)h, (long long)h->n_slice_done, (long long)h->slice_t0[kr], (long long)now_us(),
              h->slice_waves[kr], h->slice_nabs[kr], (long long)h->slice_surv_in[kr], (long long)fresh, res.count[0], res.count[25], res.count[26], res.count[27],
              res.count[31], (int)h->n_not_final, (long long)(h->n_slice - h->n_slice_done));
    }
    for (int a = 0; a < h->slice_nabs[kr]; a++) {
      const MpcHandle::Absorbed &Ab = h->slice_abs[kr][a];
      MpcHandle::FreshQ &F = h->fq[Ab.fq];
      const int64_t c = res.count[1 + a] < h->tail_cap ? res.count[1 + a] : h->tail_cap;
      /* the queue was too small for what the cut sent its way: the rest of that batch finished in its launch, stragglers included.
       * MPC_TAIL_AUTO moves its cut out of the way (an explicit cut is the caller's: mpc_tail_info counts the overflows) */
      if (res.count[1 + a] > h->tail_cap) { ++h->n_overflow; if (h->params.tail_cut < 0 && h->auto_cut < h->auto_base + 40) h->auto_cut += 4; }
      /* (the queue may hold a later batch by now: a launch may take it again as soon as the slice that absorbed it has been
       * ISSUED -- its stream waits for that slice -- which can be before this retirement) */
      const bool still = F.state == 2 && F.batch_id == Ab.batch && F.slice == h->n_slice_done;
      if (h->tslot[Ab.slot].batch_id == Ab.batch) {
        h->tslot[Ab.slot].deferred = c;
        const int64_t Bs = h->tslot[Ab.slot].B > 0 ? h->tslot[Ab.slot].B : 1;
        const int64_t share = c * 65536 / Bs;
        h->auto_share = h->auto_share < 0 ? share : (3 * h->auto_share + share) / 4;
        /* MPC_TAIL_AUTO follows the workload: a cut that sends more than 8 % of a batch to the slices is too early for this
         * distribution of iteration counts (weight sweeps: mean 16-18 iterations, a fat tail) -- the slices would do the launches'
         * work; one that sends next to nothing can come back towards the handle's base value */
        if (h->params.tail_cut < 0) {
          if (h->auto_share > h->auto_hi && h->auto_cut < h->auto_base + 40) h->auto_cut += 2;
          else if (h->auto_share < h->auto_lo && h->auto_cut > h->auto_base) --h->auto_cut;
        }
      }
      h->fresh_avg = (3 * h->fresh_avg + c + 3) / 4;
      if (still) F.state = 0;
    }
    h->slice_nabs[kr] = 0;
    const volatile long long *fin = h->h_final;
    for (int q = 0; q < h->tail_ring; q++) {
      MpcHandle::TailSlot &S = h->tslot[q];
      if (S.batch_id != 0 && !S.final_ && fin[q] == (long long)S.batch_id) { S.final_ = true; --h->n_not_final; }
    }
    ++h->n_slice_done;
    ++*n_retired;
  }
  return MPC_OK;
}

static bool tail_has_filled(const MpcHandle *h) {
  for (int q = 0; q < kFreshRing; q++) if (h->fq[q].state == 1) return true;
  return false;
}

/* One turn of the pump: retire what has completed, start slices while fewer than two are in flight and there is (or may be)
 * something for them to do.  `block`: wait for the oldest slice in flight, or -- if none is -- for the launches whose fresh
 * queues are filled, so that every call makes progress towards "everything final". */
static int tail_pump(MpcHandle *h, bool block) {
  if (!h->tail_ready) return MPC_OK;
  int n = 0;
  int rc = tail_retire(h, block && h->n_slice_done < h->n_slice, &n);
  if (rc != MPC_OK) return rc;
  /* (measurement aid: MPC_TAIL_HOLD_BATCHES=n keeps the slices back until n batches have deferred, so that the launches can be
   * timed with and without slices beside them; a blocking call releases them) */
  if (!block && h->hold_batches > 0 && h->n_deferred < h->hold_batches) return MPC_OK;
  h->hold_batches = 0;
  for (int turn = 0; turn < 2 && h->n_slice - h->n_slice_done < 2; turn++) {
    const bool in_flight = h->n_slice_done < h->n_slice;
    const bool filled = tail_has_filled(h);
    const bool force = block && !in_flight && h->surv_last == 0 && n == 0;
    bool ready = false;
    if (filled && !force)
      for (int q = 0; q < kFreshRing && !ready; q++)
        if (h->fq[q].state == 1) { const hipError_t e = hipEventQuery(h->fq[q].bulk); if (e == hipSuccess) ready = true; else (void)hipGetLastError(); }
    /* (survivors: the slice in flight leaves some if the one before it did -- the second slice takes them along as its
     * source 0 whatever their number, so no time passes between two slices while the host is elsewhere) */
    if (!(ready || (filled && force) || h->surv_last > 0)) break;
    rc = tail_launch_slice(h, force);
    if (rc != MPC_OK) return rc;
  }
  return MPC_OK;
}

/* everything handed over so far is finished (blocks) */
static int tail_drain(MpcHandle *h) {
  if (!h->tail_ready) return MPC_OK;
  for (int guard = 0; guard < (1 << 24); guard++) {
    if (h->n_not_final == 0 && h->n_slice_done == h->n_slice && !tail_has_filled(h) && h->surv_last == 0) return MPC_OK;
    const int rc = tail_pump(h, true);
    if (rc != MPC_OK) return rc;
  }
  g_last_error = "tail_drain: no progress";
  return MPC_ERR_HIP;
}

extern "C" int64_t mpc_last_batch_id(const MpcHandle *h) { return h ? h->batch_seq : 0; }

extern "C" int mpc_tail_flush(MpcHandle *h) {
  if (!h) return MPC_ERR_INVALID;
  MPC_ON_DEVICE(h);
  return tail_pump(h, false);
}

/* 1: batch `id` is final, 0: not yet (after one non-blocking turn of the pump); < 0: error */
static int tail_poll_id(MpcHandle *h, int64_t id, bool pump) {
  if (id <= 0 || id > h->batch_seq) { g_last_error = "no such batch id"; return MPC_ERR_INVALID; }
  if (!h->brec) { g_last_error = "no such batch id"; return MPC_ERR_INVALID; }
  MpcHandle::BatchRec &R = h->brec[id % MpcHandle::kBatchRecs];
  if (R.id != id) { g_last_error = "batch id too old to resolve (the handle keeps the last 1024)"; return MPC_ERR_INVALID; }
  if (R.kind == 2) return 1;                           /* an empty batch */
  if (R.kind == 0) {
    const hipError_t e = hipEventQuery(R.ev);
    if (e == hipSuccess) return 1;
    (void)hipGetLastError();
    if (e == hipErrorNotReady) return 0;
    g_last_error = std::string("hipEventQuery: ") + hipGetErrorString(e);
    return MPC_ERR_HIP;
  }
  const MpcHandle::TailSlot &S = h->tslot[R.slot];
  if (S.batch_id != id || S.final_) return 1;        /* (a slot is only taken again once its batch is final) */
  if (pump) { const int rc = tail_pump(h, false); if (rc != MPC_OK) return rc; }
  return (S.batch_id != id || S.final_) ? 1 : 0;
}

extern "C" int mpc_tail_poll(MpcHandle *h, int64_t batch_id) {
  if (!h) return MPC_ERR_INVALID;
  MPC_ON_DEVICE(h);
  return tail_poll_id(h, batch_id, true);
}

extern "C" int mpc_tail_wait(MpcHandle *h, int64_t batch_id) {
  if (!h) return MPC_ERR_INVALID;
  MPC_ON_DEVICE(h);
  if (batch_id <= 0) {                         /* every batch issued so far: the deferring ones, and the launches of the rest */
    const int rc = tail_drain(h);
    if (rc != MPC_OK) return rc;
    if (h->brec)
      for (int64_t id = h->batch_seq; id > 0 && id > h->batch_seq - MpcHandle::kBatchRecs; --id) {
        MpcHandle::BatchRec &R = h->brec[id % MpcHandle::kBatchRecs];
        if (R.id == id && R.kind == 0 && R.ev) MPC_HIP_CHECK(hipEventSynchronize(R.ev));
      }
    return MPC_OK;
  }
  for (int guard = 0; guard < (1 << 24); guard++) {
    const int r = tail_poll_id(h, batch_id, false);
    if (r < 0) return r;
    if (r == 1) return MPC_OK;
    MpcHandle::BatchRec &R = h->brec[batch_id % MpcHandle::kBatchRecs];
    if (R.kind == 0) { MPC_HIP_CHECK(hipEventSynchronize(R.ev)); return MPC_OK; }
    const int rc = tail_pump(h, true);
    if (rc != MPC_OK) return rc;
  }
  g_last_error = "mpc_tail_wait: no progress";
  return MPC_ERR_HIP;
}

extern "C" int mpc_tail_stream_wait(MpcHandle *h, int64_t batch_id, void *stream) {
  if (!h) return MPC_ERR_INVALID;
  MPC_ON_DEVICE(h);
  if (batch_id <= 0 || batch_id > h->batch_seq || !h->brec) { g_last_error = "no such batch id"; return MPC_ERR_INVALID; }
  MpcHandle::BatchRec &R = h->brec[batch_id % MpcHandle::kBatchRecs];
  if (R.id != batch_id) { g_last_error = "batch id too old to resolve (the handle keeps the last 1024)"; return MPC_ERR_INVALID; }
  if (R.kind == 2) return MPC_OK;
  if (R.kind == 0) {                           /* not deferring: final behind its own launch */
    MPC_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, R.ev, 0));
    return MPC_OK;
  }
  /* a deferring batch is final behind a slice that has not been issued yet: the host brings it there (the pump is what starts
   * slices), then orders `stream` behind the tail stream's most recent slice */
  const int rc = mpc_tail_wait(h, batch_id);
  if (rc != MPC_OK) return rc;
  if (h->n_slice > 0) MPC_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, h->slice_ev[(h->n_slice - 1) % kSliceRing], 0));
  return MPC_OK;
}

/* counters of the handle's tail machinery: out[0] batches deferred so far, [1] tail slices so far, [2] ring, [3] capacity of a
 * batch's fresh queue, [4] waves per slice (upper bound), [5] 1 if the tail stream has high priority, [6] tail streams (1),
 * [7] batches that ran without deferral because the survivors' list was filling up, [8] passes per slice, [9] survivors now */
extern "C" int mpc_tail_info(const MpcHandle *h, int64_t *out) {
  if (!h || !out) return MPC_ERR_INVALID;
  out[0] = h->n_deferred; out[1] = h->n_slice; out[2] = h->tail_ring; out[3] = h->tail_cap; out[4] = h->tail_waves;
  out[5] = h->tail_priority < 0 ? 1 : 0; out[6] = 1; out[7] = h->n_throttled; out[8] = h->slice_passes; out[9] = h->surv_last;
  out[10] = h->params.tail_cut > 0 ? h->params.tail_cut : (h->params.tail_cut < 0 ? h->auto_cut : 0); out[11] = h->auto_share;
  out[12] = h->n_overflow;
  return MPC_OK;
}

extern "C" int mpc_tail_pending(MpcHandle *h, int64_t batch_id, int64_t *n) {
  if (!h || !n) return MPC_ERR_INVALID;
  *n = 0;
  if (!h->tail_ready || !h->brec || batch_id <= 0 || batch_id > h->batch_seq) return MPC_OK;
  MPC_ON_DEVICE(h);
  MpcHandle::BatchRec &R = h->brec[batch_id % MpcHandle::kBatchRecs];
  if (R.id != batch_id || R.kind == 0) return MPC_OK;
  MpcHandle::TailSlot &S = h->tslot[R.slot];
  if (S.batch_id != batch_id) return MPC_OK;            /* long final: its slot holds a later batch */
  if (S.deferred >= 0) { *n = S.deferred; return MPC_OK; }
  for (int q = 0; q < kFreshRing; q++)
    if (h->fq[q].batch_id == batch_id && h->fq[q].state == 2) {       /* absorbed: the slice reports the count (and resets it) */
      for (int guard = 0; guard < 64 && S.deferred < 0 && h->n_slice_done < h->n_slice; guard++) {
        int nr = 0;
        const int rc = tail_retire(h, true, &nr);
        if (rc != MPC_OK) return rc;
      }
      *n = S.deferred >= 0 ? S.deferred : 0;
      return MPC_OK;
    }
  for (int q = 0; q < kFreshRing; q++)
    if (h->fq[q].batch_id == batch_id && h->fq[q].state == 1) {
      MPC_HIP_CHECK(hipEventSynchronize(h->fq[q].bulk));
      int32_t c = 0;
      MPC_HIP_CHECK(hipMemcpy(&c, h->fq_dev[q].count, sizeof(c), hipMemcpyDeviceToHost));
      *n = c < h->tail_cap ? c : h->tail_cap;
      return MPC_OK;
    }
  /* absorbed and retired in the meantime */
  int nr = 0;
  const int rc = tail_retire(h, false, &nr);
  if (rc != MPC_OK) return rc;
  *n = S.deferred >= 0 ? S.deferred : 0;
  return MPC_OK;
}

/* Mixed precision across phases: phase 0 = the fp32 solver on the handle's fp32 workspace, parking every instance at
 * MPC_PROMOTE; phase 1 = the fp64 solver resuming all of them on the fp64 workspace.  RIO is the handle's own precision
 * (the type at the ABI).  Instances the fp32 phase finishes itself (rejected at set-up, not-a-number, iteration cap) are
 * final after phase 0. */
template <class RIO>
static int launch_mixed(MpcHandle *h, int64_t B, int64_t ld, int64_t ldo, const RIO *state, const RIO *coeffs, const RIO *yaw_lo,
                        const RIO *yaw_hi, const RIO *weights, RIO *out, RIO *traj, int32_t *status, int32_t *iters, hipStream_t s,
                        const MpcPhase &tail, int32_t *cb, int32_t *zero_next) {
  const int64_t tiles = h->io_stride / 64;
  /* h->ws holds the handle's own layout; the phases need one workspace of each (each allocation on its own: a failure
   * leaves the handle usable for a retry, and mpc_destroy frees whatever exists) */
  if (!h->ws2) {
    const size_t other = sizeof(RIO) == 4 ? (size_t)h->ws_stride_f64 * tiles * sizeof(double) : (size_t)h->ws_stride_f32 * tiles * sizeof(float);
    MPC_HIP_CHECK(hipMalloc((void **)&h->ws2, other));
  }
  if (!h->d_park) MPC_HIP_CHECK(hipMalloc((void **)&h->d_park, sizeof(double) * 2 * kParkRows * h->io_stride));
  if (!h->d_list) MPC_HIP_CHECK(hipMalloc((void **)&h->d_list, sizeof(int32_t) * 4 * h->io_stride));
  if (!h->d_piter && h->promote_buffer)
    MPC_HIP_CHECK(hipMalloc(&h->d_piter, sizeof(float) * (size_t)h->io_stride * (size_t)(h->params.N - 1) * mpc::Fields<float>::IT_SZ));
  float *ws32 = sizeof(RIO) == 4 ? (float *)h->ws : (float *)h->ws2;
  double *ws64 = sizeof(RIO) == 4 ? (double *)h->ws2 : (double *)h->ws;
  int32_t *it_out = iters ? iters : h->d_iters;
  MPC_HIP_CHECK(hipEventRecord(h->ev0, s));
  const unsigned waves = (unsigned)((B + kBlock - 1) / kBlock);
  MpcPhase T;
  memset(&T, 0, sizeof(T));
  T.take = cb; T.n_out = cb + 1; T.zero_next = zero_next;
  T.out_inst = h->d_list; T.out_src = h->d_list + h->io_stride; T.out_park = h->d_park; T.ld_park = h->io_stride;
  T.refill_min = h->refill_min; T.refill_wait = h->refill_wait;
  T.promote_out = 1;
  T.promote_cap = h->promote_cap;
  T.p_iter = h->promote_buffer ? h->d_piter : nullptr;
  T.refill_floor = h->refill_floor_f32;
  T.compact_gap = (T.p_iter && B >= h->compact_min_batch) ? h->compact_gap : 0; T.compact_cooldown = h->compact_cooldown;
  hipLaunchKernelGGL((mpc_solve_kernel<true, float, 1, RIO, RIO>), dim3(waves), dim3(kBlock), staging_lds_bytes<float>(), s, h->params, B, ld, ldo, state,
                     coeffs, yaw_lo, yaw_hi, weights, out, traj, status, it_out, ws32, h->ws_stride_f32, T);
  MPC_HIP_CHECK(hipGetLastError());
  MpcPhase U;
  memset(&U, 0, sizeof(U));
  U.take = cb + 2; U.n_out = cb + 3; U.n_in = cb + 1;
  U.in_inst = h->d_list; U.in_src = h->d_list + h->io_stride; U.in_park = h->d_park; U.ld_park = h->io_stride;
  U.out_inst = h->d_list + 2 * h->io_stride; U.out_src = h->d_list + 3 * h->io_stride; U.out_park = h->d_park + (int64_t)kParkRows * h->io_stride;
  U.src_ws = ws32; U.src_tile_reals = h->ws_stride_f32;
  U.resume = 1; U.promote_in = 1;
  U.p_iter = h->promote_buffer ? h->d_piter : nullptr;
  U.refill_min = h->finish_refill_min; U.refill_wait = h->finish_refill_wait;
  U.compact_gap = B >= h->compact_min_batch ? h->compact_gap : 0;
  U.compact_cooldown = h->compact_cooldown;
  const unsigned waves2 = (waves + (unsigned)h->finish_div - 1) / (unsigned)h->finish_div;
  /* deferred tails: the fp64 phase hands its stragglers over (the fp32 phase's chains end at kPromoteIterCap anyway) */
  /* (no early hand-over of a wave's last lanes here: the waves of this phase are partly filled by construction) */
  U.tail_cut = tail.tail_cut; U.t_slot = tail.t_slot; U.t_batch = tail.t_batch; U.tq = tail.tq; U.tail_few = 0; U.tail_few_from = tail.tail_few_from;
  hipLaunchKernelGGL((mpc_solve_kernel<true, double, 1, RIO, float>), dim3(waves2), dim3(kBlock), staging_lds_bytes<double>(), s, h->params, B, ld, ldo, state,
                     coeffs, yaw_lo, yaw_hi, weights, out, traj, status, it_out, ws64, h->ws_stride_f64, U);
  MPC_HIP_CHECK(hipGetLastError());
  MPC_HIP_CHECK(hipEventRecord(h->ev1, s));
  h->timed = true;
  return MPC_OK;
}

/* the launch; ld = leading dimension of the inputs, ldo = of out/traj */
template <class R>
static int launch_solve(MpcHandle *h, int64_t B, int64_t ld, int64_t ldo, const R *state, const R *coeffs,
                        const R *yaw_lo, const R *yaw_hi, const R *weights, R *out, R *traj,
                        int32_t *status, int32_t *iters, void *stream_, bool with_stats = true, bool may_defer = false) {
  if (!h) { g_last_error = "NULL handle"; return MPC_ERR_INVALID; }
  if ((h->params.precision == MPC_PRECISION_F32) != (sizeof(R) == 4)) {
    g_last_error = "this handle was created with the other precision: fp64 handles take the double entry points, "
                   "MPC_PRECISION_F32 handles mpc_solve_batch_device_f32";
    return MPC_ERR_INVALID;
  }
  if (B < 0 || ld < B || ldo < B) { g_last_error = "ld < B"; return MPC_ERR_INVALID; }
  if (B > h->max_batch) { g_last_error = "B exceeds the handle's max_batch"; return MPC_ERR_INVALID; }
  h->last_B = B; h->timed = false; h->have_stats = false; h->stats_pending = false;
  ++h->batch_seq;
  if (B == 0) {                /* empty batch: nothing to read or write, pointers may be NULL; its id resolves as final */
    if (!h->brec) h->brec = new MpcHandle::BatchRec[MpcHandle::kBatchRecs];
    MpcHandle::BatchRec &R0 = h->brec[h->batch_seq % MpcHandle::kBatchRecs];
    R0.id = h->batch_seq; R0.kind = 2;
    return MPC_OK;
  }
  if (!state || !coeffs || !yaw_lo || !yaw_hi || !out || !status) { g_last_error = "NULL argument"; return MPC_ERR_INVALID; }
  MPC_ON_DEVICE(h);   /* workspace, lazy allocations and a NULL stream all belong to the handle's device */
  hipStream_t s = (hipStream_t)stream_;   /* NULL = HIP's default (null) stream, exactly as passed */
  /* waves: one lane per instance, or fewer waves whose lanes take several instances in turn (instances_per_lane) */
  const int64_t waves_full = (B + kBlock - 1) / kBlock;
  int64_t waves = (waves_full + h->inst_per_lane - 1) / h->inst_per_lane;
  if (waves < 1) waves = 1;
  bool defer = may_defer && h->params.tail_cut != 0 && B >= h->tail_min_batch && (h->mixed || !(h->lds_lanes > 0 && B <= h->lds_max_batch)) &&
               !(h->wave_max_batch > 0 && B <= h->wave_max_batch);
  MpcHandle::BatchRec *rec = nullptr;
  { const int rc = batch_rec(h, h->batch_seq, &rec); if (rc != MPC_OK) return rc; }
  rec->id = 0;                                      /* (valid once the launch has been issued) */
  int slot_index = 0, fq_index = 0;
  if (defer) {
    int rc = tail_prepare(h);
    if (rc != MPC_OK) return rc;
    rc = tail_pump(h, false);
    if (rc != MPC_OK) return rc;
    /* the survivors' list is filling up (stragglers arrive faster than the slices finish them): this batch keeps its own */
    if (2 * h->surv_last > h->surv_cap) { defer = false; ++h->n_throttled; if (h->params.tail_cut < 0 && h->auto_cut < h->auto_base + 40) h->auto_cut += 2; }
  }
  if (defer) {
    slot_index = (int)(h->n_deferred % h->tail_ring);
    fq_index = (int)(h->n_deferred % kFreshRing);
    /* the slot is taken again: the batch it held must be final (only if the ring is shorter than the stragglers' latency) */
    for (int guard = 0; h->tslot[slot_index].batch_id != 0 && !h->tslot[slot_index].final_; guard++) {
      if (guard > (1 << 22)) { g_last_error = "deferred tails: no progress"; return MPC_ERR_HIP; }
      const int rc = tail_pump(h, true);
      if (rc != MPC_OK) return rc;
    }
    /* the fresh queue is taken again: the slice that absorbed its previous batch must have read it */
    MpcHandle::FreshQ &F = h->fq[fq_index];
    if (F.state == 1) { const int rc = tail_launch_slice(h, true); if (rc != MPC_OK) return rc; }
    /* (its count is zero again: the slice's last wave has reset it) */
    if (F.state == 2) MPC_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream_, h->slice_ev[F.slice % kSliceRing], 0));
  }
  const int n_cuts = (!defer && h->inst_per_lane == 1 && B >= h->two_phase_min) ? h->n_cuts : 0;
  auto tail_fields = [&](MpcPhase &T) {
    T.tail_cut = defer ? (h->params.tail_cut > 0 ? h->params.tail_cut : h->auto_cut) : 0; T.t_slot = slot_index; T.t_batch = h->batch_seq;
    if (defer) { T.tq = h->fq_dev[fq_index]; T.tail_few = h->tail_few; T.tail_few_from = h->tail_few_from; }
  };
  /* behind the launch: the batch's record (what mpc_tail_wait / _poll / _stream_wait resolve its id with) */
  int32_t *cb = h->d_counter + (h->counter_seq % kCounterRing) * kCounterInts;
  int32_t *zero_next = h->d_counter + ((h->counter_seq + kCounterRing / 2) % kCounterRing) * kCounterInts;
  ++h->counter_seq;
  auto stats_later = [&](const int32_t *it_arr) { h->st_status = status; h->st_iters = it_arr; h->st_B = B; h->st_ev = rec->ev; h->stats_pending = true; };
  auto tail_done = [&]() -> int {
    rec->id = h->batch_seq; rec->kind = defer ? 1 : 0; rec->slot = slot_index;
    MPC_HIP_CHECK(hipEventRecord(rec->ev, s));
    if (!defer) return MPC_OK;
    MpcHandle::FreshQ &F = h->fq[fq_index];
    F.batch_id = h->batch_seq; F.slot = slot_index; F.state = 1; F.slice = -1;
    MPC_HIP_CHECK(hipEventRecord(F.bulk, s));
    MpcHandle::TailSlot &S = h->tslot[slot_index];
    S.batch_id = h->batch_seq; S.final_ = false; S.deferred = -1; S.B = B;
    ++h->n_not_final;
    ++h->n_deferred;
    return tail_pump(h, false);
  };
  const bool wave_path = h->wave_max_batch > 0 && B <= h->wave_max_batch;
  if (h->mixed && !wave_path) {
    MpcPhase TT;
    memset(&TT, 0, sizeof(TT));
    tail_fields(TT);
    const int rc = launch_mixed<R>(h, B, ld, ldo, state, coeffs, yaw_lo, yaw_hi, weights, out, traj, status, iters, s, TT, cb, zero_next);
    if (rc != MPC_OK) return rc;
    const int rt = tail_done();
    if (rt != MPC_OK) return rt;
    if (with_stats) stats_later(iters ? iters : h->d_iters);
    return MPC_OK;
  }
  if (n_cuts > 0) {
    const size_t ws_bytes = (size_t)h->ws_stride * (size_t)(h->io_stride / 64) * sizeof(R);
    if (!h->ws2) MPC_HIP_CHECK(hipMalloc((void **)&h->ws2, ws_bytes));
    if (!h->d_park) MPC_HIP_CHECK(hipMalloc((void **)&h->d_park, sizeof(double) * 2 * kParkRows * h->io_stride));
    if (!h->d_list) MPC_HIP_CHECK(hipMalloc((void **)&h->d_list, sizeof(int32_t) * 4 * h->io_stride));
  }
  if (h->wave_max_batch > 0 && B <= h->wave_max_batch) {
    --h->counter_seq;                              /* (this path uses no counters: the block stays clean for the next call) */
    MPC_HIP_CHECK(hipEventRecord(h->ev0, s));
    /* lanes per instance: a lane per stage -- 16 up to N = 17 (four instances per wavefront), 32 up to N = 33, else the whole wave;
     * a launch of a few instances takes the whole wave anyway (its cross-lane reads are v_readlane instead of ds_bpermute) */
    const int64_t per = (int64_t)mpc::workspace_fields_per_instance(h->params.N, sizeof(R) == 4, h->params.initial_state_rows != 0) * (int64_t)sizeof(R);
    int lpi = h->params.N - 1 <= 16 ? 16 : (h->params.N - 1 <= 32 ? 32 : 64);
    if (B <= h->wave_whole_max) lpi = 64;
    if (const char *e = getenv("MPC_WAVE_LPI")) lpi = atoi(e) == 16 ? 16 : (atoi(e) == 32 ? 32 : 64);
    if (lpi < h->params.N - 1) lpi = 64;
    int32_t *it_w = iters ? iters : h->d_iters;
    if (lpi == 16)
      hipLaunchKernelGGL((mpc_solve_wave_kernel<R, 16>), dim3((unsigned)((B + 3) / 4)), dim3(kBlock), (size_t)(4 * per), s, h->params, B, ld, ldo, state, coeffs, yaw_lo, yaw_hi,
                         weights, out, traj, status, it_w);
    else if (lpi == 32)
      hipLaunchKernelGGL((mpc_solve_wave_kernel<R, 32>), dim3((unsigned)((B + 1) / 2)), dim3(kBlock), (size_t)(2 * per), s, h->params, B, ld, ldo, state, coeffs, yaw_lo, yaw_hi,
                         weights, out, traj, status, it_w);
    else
      hipLaunchKernelGGL((mpc_solve_wave_kernel<R, 64>), dim3((unsigned)B), dim3(kBlock), (size_t)per, s, h->params, B, ld, ldo, state, coeffs, yaw_lo, yaw_hi,
                         weights, out, traj, status, it_w);
    MPC_HIP_CHECK(hipGetLastError());
    MPC_HIP_CHECK(hipEventRecord(h->ev1, s));
    h->timed = true;
    { const int rt = tail_done(); if (rt != MPC_OK) return rt; }
    if (with_stats) stats_later(iters ? iters : h->d_iters);
    return MPC_OK;
  }
  if (h->lds_lanes > 0 && B <= h->lds_max_batch) {
    --h->counter_seq;                              /* (this path uses no counters: the block stays clean for the next call) */
    MPC_HIP_CHECK(hipEventRecord(h->ev0, s));
    const int rc = launch_lds<R>(h, B, ld, ldo, state, coeffs, yaw_lo, yaw_hi, weights, out, traj, status, iters ? iters : h->d_iters, s);
    if (rc != MPC_OK) return rc;
    MPC_HIP_CHECK(hipEventRecord(h->ev1, s));
    h->timed = true;
    { const int rt = tail_done(); if (rt != MPC_OK) return rt; }
    if (with_stats) stats_later(iters ? iters : h->d_iters);
    return MPC_OK;
  }
  int32_t *it_out = iters ? iters : h->d_iters;
  MPC_HIP_CHECK(hipEventRecord(h->ev0, s));
  auto launch = [&](unsigned grid, void *wsp, const MpcPhase &tp) {
    constexpr int kOcc2 = sizeof(R) == 4 ? 2 : 1;
    if (h->staging && h->occ2)
      hipLaunchKernelGGL((mpc_solve_kernel<true, R, kOcc2>), dim3(grid), dim3(kBlock), staging_lds_bytes<R>(), s, h->params, B, ld, ldo, state, coeffs,
                         yaw_lo, yaw_hi, weights, out, traj, status, it_out, (R *)wsp, h->ws_stride, tp);
    else if (h->staging)
      hipLaunchKernelGGL((mpc_solve_kernel<true, R, 1>), dim3(grid), dim3(kBlock), staging_lds_bytes<R>(), s, h->params, B, ld, ldo, state, coeffs,
                         yaw_lo, yaw_hi, weights, out, traj, status, it_out, (R *)wsp, h->ws_stride, tp);
    else
      hipLaunchKernelGGL((mpc_solve_kernel<false, R, 1>), dim3(grid), dim3(kBlock), 0, s, h->params, B, ld, ldo, state, coeffs,
                         yaw_lo, yaw_hi, weights, out, traj, status, it_out, (R *)wsp, h->ws_stride, tp);
  };
  /* phase p takes from counter [2p], parks into list p & 1 and counts its parked instances in [2p + 1]; phase p > 0
   * reads list (p - 1) & 1 and the workspace of phase p - 1; every phase has the grid of the first (see MpcPhase) */
  for (int p = 0; p <= n_cuts; ++p) {
    MpcPhase T;
    memset(&T, 0, sizeof(T));          /* every switch a phase does not set is off */
    const int wr = p & 1, rd = wr ^ 1;
    T.take = cb + 2 * p;
    T.n_out = cb + 2 * p + 1;
    T.n_in = p > 0 ? cb + 2 * (p - 1) + 1 : nullptr;
    T.zero_next = p == 0 ? zero_next : nullptr;
    T.out_inst = h->d_list ? h->d_list + (int64_t)(2 * wr) * h->io_stride : nullptr;
    T.out_src = h->d_list ? h->d_list + (int64_t)(2 * wr + 1) * h->io_stride : nullptr;
    T.in_inst = h->d_list ? h->d_list + (int64_t)(2 * rd) * h->io_stride : nullptr;
    T.in_src = h->d_list ? h->d_list + (int64_t)(2 * rd + 1) * h->io_stride : nullptr;
    T.out_park = h->d_park ? h->d_park + (int64_t)wr * kParkRows * h->io_stride : nullptr;
    T.in_park = h->d_park ? h->d_park + (int64_t)rd * kParkRows * h->io_stride : nullptr;
    T.ld_park = h->io_stride;
    T.src_ws = rd ? h->ws2 : h->ws;
    T.pass_cut = p < n_cuts ? h->cuts[p] : 0;
    T.resume = p > 0;
    T.refill_min = h->refill_min; T.refill_wait = h->refill_wait; T.refill_floor = h->refill_floor;
    T.compact_cooldown = h->compact_cooldown;
    T.compact_gap = (n_cuts == 0 && B >= h->compact_min_batch) ? h->compact_gap : 0;     /* (a phase that parks keeps iterates in its columns) */
    tail_fields(T);
    const bool pooled = h->pool && n_cuts == 0;      /* a parked iterate stays in its column: phases keep their own tiles */
    T.pool_bits = pooled ? h->pool->bits : nullptr; T.pool_base = pooled ? h->pool->base : nullptr;
    T.pool_tiles = pooled ? h->pool->tiles : 0; T.pool_words = pooled ? h->pool->words : 0;
    launch((unsigned)waves, wr ? h->ws2 : h->ws, T);
    MPC_HIP_CHECK(hipGetLastError());
  }
  MPC_HIP_CHECK(hipGetLastError());
  MPC_HIP_CHECK(hipEventRecord(h->ev1, s));
  h->timed = true;
  { const int rt = tail_done(); if (rt != MPC_OK) return rt; }
  if (with_stats) stats_later(it_out);
  return MPC_OK;
}

extern "C" int mpc_solve_batch_device(MpcHandle *h, int64_t B, int64_t ld, const double *state,
                                      const double *coeffs, const double *yaw_lo, const double *yaw_hi,
                                      const double *weights, double *out, double *traj, int32_t *status,
                                      int32_t *iters, void *stream_) {
  return launch_solve<double>(h, B, ld, ld, state, coeffs, yaw_lo, yaw_hi, weights, out, traj, status, iters, stream_, true, true);
}

/* MPC_PRECISION_F32: the same solve with fp32 inputs, outputs and workspace (handle created with precision F32) */
extern "C" int mpc_solve_batch_device_f32(MpcHandle *h, int64_t B, int64_t ld, const float *state,
                                          const float *coeffs, const float *yaw_lo, const float *yaw_hi,
                                          const float *weights, float *out, float *traj, int32_t *status,
                                          int32_t *iters, void *stream_) {
  return launch_solve<float>(h, B, ld, ld, state, coeffs, yaw_lo, yaw_hi, weights, out, traj, status, iters, stream_, true, true);
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
  /* the tables MPC::run() looks up (Vehicle.cpp:34-79) must exist, and the fit is built for orders 2..4
   * (Config::maxFitOrder <= 5, as in every config-*.json; the reference would go on to higher orders) */
  if (h->params.n_yaw_change_speeds < 1 || h->params.n_steer_speeds < 1) { g_last_error = "run(): empty speed tables (load a config-*.json)"; return MPC_ERR_INVALID; }
  if (h->params.max_fit_order > 5) { g_last_error = "run(): max_fit_order > 5 is not built (fit orders 2..4)"; return MPC_ERR_UNSUPPORTED; }
  MPC_ON_DEVICE(h);
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
  int rc = launch_solve<double>(h, B, S, ld, d_pre, d_pre + 6 * S, d_pre + 11 * S, d_pre + 12 * S, nullptr, h->d_run9, traj, status, iters, stream_);
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

/* used by the other translation units of the library (mpc_wire.cpp): the error text of this thread */
extern "C" void mpc_internal_set_error(const char *msg) { g_last_error = msg ? msg : ""; }

/* The telemetry handler for host arrays: one copy in, the kernels, one copy out, on the handle's own device and stream
 * (whatever the caller's current device is), staging kept on the handle.  rows of `tel`, `ptsx`, `ptsy` as in
 * mpc_telemetry_batch_device with leading dimension ld; the waypoint arrays are inputs only here. */
extern "C" int mpc_telemetry_batch_host(MpcHandle *h, int64_t B, int64_t ld, int npts, const double *tel, double extra_latency,
                                        const double *ptsx, const double *ptsy, double *cmd, int32_t *status) {
  if (!h) { g_last_error = "NULL handle"; return MPC_ERR_INVALID; }
  if (B < 0 || ld < B || B > h->max_batch) { g_last_error = "bad B/ld"; return MPC_ERR_INVALID; }
  if (npts < 3 || npts > mpc::RUN_MAX_PTS) { g_last_error = "npts must be 3..8"; return MPC_ERR_INVALID; }
  if (B == 0) { h->last_B = 0; return MPC_OK; }
  if (!tel || !ptsx || !ptsy || !cmd || !status) { g_last_error = "NULL argument"; return MPC_ERR_INVALID; }
  MPC_ON_DEVICE(h);
  const int64_t rows = 6 + 2 * npts, L = (B + 7) / 8 * 8;
  const size_t need = sizeof(double) * (size_t)((rows + 2) * L) + sizeof(int32_t) * (size_t)L;
  if (h->tel_bytes < need) {
    if (h->d_tel) MPC_HIP_CHECK(hipFree(h->d_tel));
    h->d_tel = nullptr; h->tel_bytes = 0;
    MPC_HIP_CHECK(hipMalloc((void **)&h->d_tel, need));
    h->tel_bytes = need;
  }
  double *d = h->d_tel, *d_cmd = d + rows * L;
  int32_t *d_st = (int32_t *)(d_cmd + 2 * L);
  hipStream_t s = h->stream;
  MPC_HIP_CHECK(hipMemcpy2DAsync(d, sizeof(double) * L, tel, sizeof(double) * ld, sizeof(double) * B, 6, hipMemcpyHostToDevice, s));
  MPC_HIP_CHECK(hipMemcpy2DAsync(d + 6 * L, sizeof(double) * L, ptsx, sizeof(double) * ld, sizeof(double) * B, npts, hipMemcpyHostToDevice, s));
  MPC_HIP_CHECK(hipMemcpy2DAsync(d + (6 + npts) * L, sizeof(double) * L, ptsy, sizeof(double) * ld, sizeof(double) * B, npts, hipMemcpyHostToDevice, s));
  const int rc = mpc_telemetry_batch_device(h, B, L, npts, d, extra_latency, d + 6 * L, d + (6 + npts) * L, d_cmd, nullptr, d_st, (void *)s);
  if (rc != MPC_OK) return rc;
  MPC_HIP_CHECK(hipMemcpy2DAsync(cmd, sizeof(double) * ld, d_cmd, sizeof(double) * L, sizeof(double) * B, 2, hipMemcpyDeviceToHost, s));
  MPC_HIP_CHECK(hipMemcpyAsync(status, d_st, sizeof(int32_t) * B, hipMemcpyDeviceToHost, s));
  MPC_HIP_CHECK(hipStreamSynchronize(s));
  return MPC_OK;
}

/* MPC::run() for host arrays (the drop-in's B = 1 case, include/mpc_drop_in.hpp): one copy in, the three kernels of
 * mpc_run_batch_device on the handle's own device and stream, one copy out; synchronises.  ptsx / ptsy are transformed in place
 * like the reference does (MPC.cpp:329; mpc_main.cpp:189-190 relies on it). */
extern "C" int mpc_run_batch_host(MpcHandle *h, int64_t B, int64_t ld, int npts, const double *pose, double *ptsx, double *ptsy,
                                  double *out8, double *traj, int32_t *status, int32_t *iters, double *pre) {
  if (!h) { g_last_error = "NULL handle"; return MPC_ERR_INVALID; }
  if (B < 0 || ld < B || B > h->max_batch) { g_last_error = "bad B/ld"; return MPC_ERR_INVALID; }
  if (npts < 3 || npts > mpc::RUN_MAX_PTS) { g_last_error = "npts must be 3..8"; return MPC_ERR_INVALID; }
  if (h->params.precision != MPC_PRECISION_F64) { g_last_error = "run() entry points are fp64 only"; return MPC_ERR_INVALID; }
  if (B == 0) { h->last_B = 0; return MPC_OK; }
  if (!pose || !ptsx || !ptsy || !out8 || !status) { g_last_error = "NULL argument"; return MPC_ERR_INVALID; }
  MPC_ON_DEVICE(h);
  const int N = h->params.N;
  const int64_t L = (B + 7) / 8 * 8;
  const int64_t in_rows = 6 + 2 * npts, out_rows = 8 + 2 * N + 15;
  const size_t need = sizeof(double) * (size_t)((in_rows + out_rows) * L) + sizeof(int32_t) * (size_t)(2 * L);
  if (h->tel_bytes < need) {
    if (h->d_tel) MPC_HIP_CHECK(hipFree(h->d_tel));
    h->d_tel = nullptr; h->tel_bytes = 0;
    MPC_HIP_CHECK(hipMalloc((void **)&h->d_tel, need));
    h->tel_bytes = need;
  }
  double *d = h->d_tel, *d_px = d + 6 * L, *d_py = d_px + (int64_t)npts * L, *d_o8 = d + in_rows * L, *d_tr = d_o8 + 8 * L, *d_pre = d_tr + 2 * (int64_t)N * L;
  int32_t *d_st = (int32_t *)(d_pre + 15 * L), *d_it = d_st + L;
  hipStream_t s = h->stream;
  MPC_HIP_CHECK(hipMemcpy2DAsync(d, sizeof(double) * L, pose, sizeof(double) * ld, sizeof(double) * B, 6, hipMemcpyHostToDevice, s));
  MPC_HIP_CHECK(hipMemcpy2DAsync(d_px, sizeof(double) * L, ptsx, sizeof(double) * ld, sizeof(double) * B, npts, hipMemcpyHostToDevice, s));
  MPC_HIP_CHECK(hipMemcpy2DAsync(d_py, sizeof(double) * L, ptsy, sizeof(double) * ld, sizeof(double) * B, npts, hipMemcpyHostToDevice, s));
  const int rc = mpc_run_batch_device(h, B, L, npts, d, d_px, d_py, d_o8, traj ? d_tr : nullptr, d_st, d_it, d_pre, (void *)s);
  if (rc != MPC_OK) return rc;
  MPC_HIP_CHECK(hipMemcpy2DAsync(ptsx, sizeof(double) * ld, d_px, sizeof(double) * L, sizeof(double) * B, npts, hipMemcpyDeviceToHost, s));
  MPC_HIP_CHECK(hipMemcpy2DAsync(ptsy, sizeof(double) * ld, d_py, sizeof(double) * L, sizeof(double) * B, npts, hipMemcpyDeviceToHost, s));
  MPC_HIP_CHECK(hipMemcpy2DAsync(out8, sizeof(double) * ld, d_o8, sizeof(double) * L, sizeof(double) * B, 8, hipMemcpyDeviceToHost, s));
  if (traj) MPC_HIP_CHECK(hipMemcpy2DAsync(traj, sizeof(double) * ld, d_tr, sizeof(double) * L, sizeof(double) * B, 2 * N, hipMemcpyDeviceToHost, s));
  if (pre) MPC_HIP_CHECK(hipMemcpy2DAsync(pre, sizeof(double) * ld, d_pre, sizeof(double) * L, sizeof(double) * B, 15, hipMemcpyDeviceToHost, s));
  MPC_HIP_CHECK(hipMemcpyAsync(status, d_st, sizeof(int32_t) * B, hipMemcpyDeviceToHost, s));
  if (iters) MPC_HIP_CHECK(hipMemcpyAsync(iters, d_it, sizeof(int32_t) * B, hipMemcpyDeviceToHost, s));
  MPC_HIP_CHECK(hipStreamSynchronize(s));
  return MPC_OK;
}

/* the device a handle lives on (mpc_create's `device`, resolved) */
extern "C" int mpc_handle_device(const MpcHandle *h) { return h ? h->device : MPC_ERR_INVALID; }

extern "C" int mpc_rollout_batch_device(MpcHandle *h, int64_t B, int64_t ld, int steps, double *state, const double *coeffs,
                                        const double *yaw_lo, const double *yaw_hi, const double *weights, double *hist,
                                        int32_t *status, int32_t *iters, void *stream_) {
  if (!h) { g_last_error = "NULL handle"; return MPC_ERR_INVALID; }
  if (B < 0 || ld < B || B > h->max_batch) { g_last_error = "bad B/ld"; return MPC_ERR_INVALID; }
  if (steps < 1) { g_last_error = "steps < 1"; return MPC_ERR_INVALID; }
  if (B == 0) { h->last_B = 0; return MPC_OK; }
  if (!state || !coeffs || !yaw_lo || !yaw_hi || !status) { g_last_error = "NULL argument"; return MPC_ERR_INVALID; }
  MPC_ON_DEVICE(h);
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
    int rc = launch_solve<double>(h, B, ld, ld, state, coeffs, yaw_lo, yaw_hi, weights, o9, nullptr, h->d_rstat, h->d_iters, stream_, false);
    if (rc != MPC_OK) return rc;
    hipLaunchKernelGGL(mpc_rollout_step_kernel, dim3(grid), dim3(256), 0, s, B, ld, t == 0, o9, state, h->d_rstat, h->d_iters, status, iters);
    MPC_HIP_CHECK(hipGetLastError());
  }
  return record_stats(h, B, status, iters, s);   /* worst status per instance, iterations summed over the steps */
}

extern "C" int mpc_synchronize(MpcHandle *h) {
  if (!h) return MPC_ERR_INVALID;
  MPC_ON_DEVICE(h);
  MPC_HIP_CHECK(hipStreamSynchronize(h->stream));
  return tail_drain(h);
}

/* rows of a host array <-> the pinned staging block: on a few threads when there is enough to move (a 65 536-instance batch is
 * 22 MB in and 16 MB out; one core copies pageable memory at ~10 GB/s, which was most of the call's 4 ms) */
template <class Fn>
static void for_rows(int n_rows, size_t row_bytes, Fn fn) {
  const size_t total = (size_t)n_rows * row_bytes;
  int nt = total >= ((size_t)4 << 20) ? 4 : 1;
  if (nt > n_rows) nt = n_rows;
  if (nt <= 1) { for (int q = 0; q < n_rows; q++) fn(q); return; }
  std::vector<std::thread> th;
  th.reserve(nt);
  for (int t = 0; t < nt; t++)
    th.emplace_back([=]() { for (int q = t; q < n_rows; q += nt) fn(q); });
  for (auto &x : th) x.join();
}

/* host pointers: one copy in, the launch(es), one copy out, on the handle's own stream; R = the handle's precision */
template <class R>
static int solve_host(MpcHandle *h, int64_t B, int64_t ld, const R *state, const R *coeffs, const R *yaw_lo, const R *yaw_hi,
                      const R *weights, R *out, R *traj, int32_t *status, int32_t *iters) {
  if (!h) { g_last_error = "NULL handle"; return MPC_ERR_INVALID; }
  if ((h->params.precision == MPC_PRECISION_F32) != (sizeof(R) == 4)) {
    g_last_error = "this handle was created with the other precision (mpc_solve_batch_host for fp64 handles, mpc_solve_batch_host_f32 for MPC_PRECISION_F32)";
    return MPC_ERR_INVALID;
  }
  if (B < 0 || ld < B || B > h->max_batch) { g_last_error = "bad B/ld"; return MPC_ERR_INVALID; }
  if (B == 0) { h->last_B = 0; h->have_stats = false; return MPC_OK; }
  if (!state || !coeffs || !yaw_lo || !yaw_hi || !out || !status) { g_last_error = "NULL argument"; return MPC_ERR_INVALID; }
  MPC_ON_DEVICE(h);
  const int N = h->params.N;
  const int64_t S = h->io_stride;
  constexpr int kInRows = 6 + MPC_NCOEF + 2 + MPC_NW;              /* 25 */
  constexpr int kIntRows = sizeof(R) == 8 ? 1 : 2;                 /* status and iters: 2 x int32 per instance */
  const int kOutRows = MPC_NOUT + 2 * N + kIntRows;                /* out, traj, status|iters (N is fixed per handle) */
  if (!h->d_io) MPC_HIP_CHECK(hipMalloc((void **)&h->d_io, sizeof(R) * (kInRows + kOutRows) * S));
  if (!h->h_io) MPC_HIP_CHECK(hipHostMalloc((void **)&h->h_io, sizeof(R) * (kInRows + kOutRows) * S, hipHostMallocDefault));
  /* rows packed with leading dimension L (B rounded up to 16: 64-byte rows), so each direction is ONE copy */
  const int64_t L = (B + 15) / 16 * 16;
  const int in_rows = weights ? kInRows : kInRows - MPC_NW;
  const int out_rows = MPC_NOUT + (traj ? 2 * N : 0) + kIntRows;
  R *hi = (R *)h->h_io, *ho = (R *)h->h_io + kInRows * S;
  R *di = (R *)h->d_io, *d_oblk = (R *)h->d_io + kInRows * S;
  for_rows(in_rows, sizeof(R) * B, [=](int q) {
    const R *src = q < 6 ? state + q * ld : q < 11 ? coeffs + (q - 6) * ld : q == 11 ? yaw_lo : q == 12 ? yaw_hi : weights + (q - 13) * ld;
    memcpy(hi + q * L, src, sizeof(R) * B);
  });
  hipStream_t s = h->stream;
  MPC_HIP_CHECK(hipMemcpyAsync(di, hi, sizeof(R) * in_rows * L, hipMemcpyHostToDevice, s));
  R *d_o = d_oblk, *d_t = d_o + MPC_NOUT * L;
  int32_t *d_st = (int32_t *)(d_o + (out_rows - kIntRows) * L), *d_it = d_st + L;
  int rc = launch_solve<R>(h, B, L, L, di, di + 6 * L, di + 11 * L, di + 12 * L, weights ? di + 13 * L : nullptr, d_o,
                           traj ? d_t : nullptr, d_st, d_it, (void *)s);
  if (rc != MPC_OK) return rc;
  MPC_HIP_CHECK(hipMemcpyAsync(ho, d_o, sizeof(R) * out_rows * L, hipMemcpyDeviceToHost, s));
  MPC_HIP_CHECK(hipStreamSynchronize(s));
  for_rows(MPC_NOUT + (traj ? 2 * N : 0), sizeof(R) * B, [=](int q) {
    R *dst = q < MPC_NOUT ? out + q * ld : traj + (q - MPC_NOUT) * ld;
    memcpy(dst, ho + q * L, sizeof(R) * B);
  });
  const int32_t *h_st = (const int32_t *)(ho + (out_rows - kIntRows) * L);
  memcpy(status, h_st, sizeof(int32_t) * B);
  if (iters) memcpy(iters, h_st + L, sizeof(int32_t) * B);
  return MPC_OK;
}

extern "C" int mpc_solve_batch_host(MpcHandle *h, int64_t B, int64_t ld, const double *state,
                                    const double *coeffs, const double *yaw_lo, const double *yaw_hi,
                                    const double *weights, double *out, double *traj, int32_t *status,
                                    int32_t *iters) {
  return solve_host<double>(h, B, ld, state, coeffs, yaw_lo, yaw_hi, weights, out, traj, status, iters);
}

extern "C" int mpc_solve_batch_host_f32(MpcHandle *h, int64_t B, int64_t ld, const float *state,
                                        const float *coeffs, const float *yaw_lo, const float *yaw_hi,
                                        const float *weights, float *out, float *traj, int32_t *status,
                                        int32_t *iters) {
  return solve_host<float>(h, B, ld, state, coeffs, yaw_lo, yaw_hi, weights, out, traj, status, iters);
}

extern "C" int mpc_get_stats(MpcHandle *h, MpcBatchStats *st) {
  if (!h || !st) return MPC_ERR_INVALID;
  memset(st, 0, sizeof(*st));
  st->batch = h->last_B;
  if (h->last_B == 0 || !(h->have_stats || h->stats_pending)) return MPC_OK;
  MPC_ON_DEVICE(h);
  if (h->stats_pending) {                      /* gathered now, behind the call's own launch (its arrays must still be there) */
    MPC_HIP_CHECK(hipStreamWaitEvent(h->stream, h->st_ev, 0));
    const int rc = record_stats(h, h->st_B, h->st_status, h->st_iters, h->stream);
    if (rc != MPC_OK) return rc;
    h->stats_pending = false;
  }
  MPC_HIP_CHECK(hipEventSynchronize(h->ev_stats));
  unsigned long long acc[kStatWords];
  MPC_HIP_CHECK(hipMemcpy(acc, h->d_stats, sizeof(acc), hipMemcpyDeviceToHost));
  st->n_success = (int64_t)acc[MPC_STATUS_SUCCESS]; st->n_maxiter = (int64_t)acc[MPC_STATUS_MAXITER];
  st->n_linesearch = (int64_t)acc[MPC_STATUS_LINESEARCH]; st->n_infeasible = (int64_t)acc[MPC_STATUS_INFEASIBLE];
  st->n_numeric = (int64_t)acc[4]; st->n_acceptable = (int64_t)acc[8]; st->iter_sum = (int64_t)acc[5]; st->iter_max = (int32_t)acc[6]; st->n_pending = (int32_t)acc[7];
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
