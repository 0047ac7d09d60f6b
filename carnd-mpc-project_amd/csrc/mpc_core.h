/*
 * mpc_core.h -- per-instance solver of the batched MPC path (device code).
 *
 * Replaces, for one problem instance, what the reference does inside
 *   MPC::solve()          src/control/MPC.cpp:183-325
 *   FG_eval::operator()   src/control/MPC.cpp:50-154
 *   CppAD::ipopt::solve   src/control/MPC.cpp:290-292  (IPOPT + MUMPS + CppAD)
 * It is NOT a translation of any of them.  The NLP is the reference's (same
 * variables, same residuals, same bounds, same start point, branch outcomes
 * frozen at the start point as CppAD's single tape recording does), but it is
 * solved by a structure-exploiting direct-multiple-shooting primal-dual
 * interior-point method:
 *   - stage structure: s_{k+1} = F(s_k, u_k), k = 0..N-2, s_0 fixed, so the
 *     KKT matrix is block-banded and every Newton system is solved by ONE
 *     Riccati sweep (backward gains, forward roll-out, backward costates)
 *     instead of a general sparse LDL^T; O(N) work, no fill, no pivoting;
 *   - analytic stage Jacobians/Hessians (no AD tape);
 *   - the steering-rate term w4 (delta_{k+1}-delta_k)^2 (MPC.cpp:110) is
 *     carried by augmenting the stage state with d_k = delta_{k-1};
 *   - cte is a pure output state (its column of dF/ds is zero), so it is kept
 *     out of the 6x6 Riccati matrix as a scalar;
 *   - inertia correction = "every 2x2 R~_k positive definite", the Riccati
 *     equivalent of IPOPT's inertia test on the full KKT matrix.
 * Interior-point logic (least-squares multiplier start, monotone barrier,
 * fraction-to-the-boundary, filter line search, kappa_sigma dual reset, error
 * scaling, gradient-based objective scaling) follows Waechter & Biegler (2006)
 * with IPOPT's default constants so that the converged point is the one
 * IPOPT's tolerance defines.
 *
 * Mapping to the hardware: ONE INSTANCE PER LANE, 64 instances per wavefront.
 * Every instruction a wave issues is useful fp64 work for 64 independent
 * problems; there is no cross-lane traffic and no divergence except in
 * iteration counts.  Per-stage data live in a per-wave tile of a workspace
 * indexed [stage][field][lane] so that each load/store of a wave is one
 * fully coalesced line.  (See DESIGN.md for why this beats one instance per
 * wavefront for 6x6 blocks, and for the traffic accounting.)
 *
 * The same header compiles with g++ for tests/host_twin.cpp, a test-only CPU
 * build used to debug the algorithm in the GPU-less build container; the
 * shipped library contains the HIP build only.
 */
#ifndef MPC_CORE_H
#define MPC_CORE_H

#include <math.h>
#include <stdint.h>
#include <type_traits>

#include "mpc_amd.h"

#if defined(__HIPCC__)
#define MPC_HD __host__ __device__ __forceinline__
#else
#define MPC_HD inline
#endif
#if defined(__HIPCC__) || defined(__clang__)
#define MPC_UNROLL _Pragma("unroll")
#define MPC_STAGE_LOOP _Pragma("clang loop unroll(disable)")   /* one copy of a sweep's stage body: no unrolling, no peeling */
#else
#define MPC_UNROLL _Pragma("GCC unroll 8")
#define MPC_STAGE_LOOP
#endif

namespace mpc {

/* ---- workspace layout ------------------------------------------------------ */
/* A "field" is one real per instance (R = double, or float for MPC_PRECISION_F32).  Stage k (0..N-2) owns:
 *   two iterate slots: s_{k+1} (6), u_k (2), lam_{k+1} (6), bound duals of (psi_{k+1}, v_{k+1}, delta_k, a_k) (4+4)
 *   the Newton direction  ds_{k+1} (6), du_k (2)
 *   the Riccati gains     K_k (2x6) and kff_k (2)   -- in the iterate slot that is not current (see below)
 * Nothing else is kept: the stage model (sin/cos/atan, road polynomial, residual) is recomputed
 * in every sweep, because the kernel is limited by workspace traffic, not by arithmetic.
 * Fields move in GROUPS of 16 bytes per lane (G = 2 doubles or 4 floats): one global_load/store_dwordx4, one
 * LDS-DMA piece.  The fp64 record is packed (22 fields = 11 groups); the fp32 record pads the multipliers to 8 so
 * that every logical block starts on a group (24 fields = 6 groups).
 * In the fp64 solver everything is fp64: storing the direction or the gains in fp32 was tried and rejected -- an
 * absolute error of ~1e-8 in a step component is fatal next to slacks of ~1e-9 at active bounds
 * (+11 % iterations and a few non-converged instances on the 65 536-instance workload).  The fp32 solver
 * runs to tol_f32 = 5e-4 with a barrier floor of tol_f32 / 25 = 2e-5, where fp32 steps are adequate; as the first phase of
 * a mixed-precision solve it hands over to the fp64 solver before that (MPC_PROMOTE). */
template <class R> struct Layout;
template <> struct Layout<double> { enum : int { G = 2, F_S = 0, F_U = 6, F_LAM = 8, F_ZL = 14, F_ZU = 18, IT_SZ = 22 }; };
template <> struct Layout<float> { enum : int { G = 4, F_S = 0, F_U = 6, F_LAM = 8, F_ZL = 16, F_ZU = 20, IT_SZ = 24 }; };
template <class R> struct Fields : Layout<R> {
  using L = Layout<R>;
  enum : int {
    IT0 = 0, IT1 = L::IT_SZ,                                   /* double-buffered iterate */
    F_D = 2 * L::IT_SZ, D_N = 8,                               /* direction (ds, du) */
    STAGE_SZ = F_D + D_N,                                      /* fields per stage: 52 (fp64), 56 (fp32) */
    /* The Riccati gains K (2x6) and kff (2) live only between the backward and the forward sweep of one pass, while
     * the iterate slot that is not the current one holds nothing (the next trial point is written there afterwards):
     * they are stored in the first fields of that slot.  The workspace is a fifth smaller for it, and the
     * workspace stream does live in the caches (bypassing them costs 40 %). */
    F_GK = 0, GK_N = 12, F_GF = F_GK + GK_N, GF_N = 2, GAIN_SZ = (GK_N + GF_N + L::G - 1) / L::G * L::G,
    D_S = 0, D_U = 6,                                          /* direction entries */
    /* Staging interface (see TiledWorkspace): a sweep asks for the record of the NEXT stage while it works on the
     * current one.  stage_fetch_it copies the iterate slot of stage k to the front of buffer `buf`, stage_fetch_g /
     * stage_fetch_d the gains or the direction behind it; sit()/sg()/sx() read them back; stage_wait<N>() waits until
     * at most the N most recent copy/store instructions are still in flight.  Counts are in 16-byte groups. */
    STG_IT_OPS = L::IT_SZ / L::G, STG_ITF_OPS = 16 / L::G, STG_X_OPS = GAIN_SZ / L::G, STG_D_OPS = D_N / L::G,
    STG_SLOT = STG_IT_OPS + (STG_X_OPS > STG_D_OPS ? STG_X_OPS : STG_D_OPS),
    /* group stores a stage issues in each sweep (all through store_run): the counted waits let exactly these
     * stay in flight besides the newest copy group */
    ST_BACKWARD = GAIN_SZ / L::G, ST_FORWARD = D_N / L::G, ST_TRIAL = L::IT_SZ / L::G
  };
  static_assert((int)GAIN_SZ <= (int)L::IT_SZ, "the gains must fit an iterate slot");
  static_assert(L::F_ZL % L::G == 0 && L::F_ZU == L::F_ZL + 4 && L::F_U == L::F_S + 6, "blocks the forward sweep fetches");
};

/* (N - 1 stages; with MpcParams.initial_state_rows one more record for the multipliers and bound duals of the initial state's
 * own rows: Solver::s0_lam) */
MPC_HD int64_t workspace_fields_per_instance(int N, bool f32, bool s0_rows = false) {
  return (int64_t)(N - 1 + (s0_rows ? 1 : 0)) * (f32 ? (int)Fields<float>::STAGE_SZ : (int)Fields<double>::STAGE_SZ);
}

/* Plain storage for the test-only host build: one instance, fields contiguous. */
template <class R>
struct HostWorkspace {
  using F = Fields<R>;
  R *base;
  /* field f of iterate slot I (or I = 0 and an absolute field) of stage k */
  MPC_HD R &it(int k, int I, int f) const { return base[k * F::STAGE_SZ + I + f]; }
  MPC_HD R getD(int k, int j) const { return base[k * F::STAGE_SZ + F::F_D + j]; }
  MPC_HD void setD(int k, int j, R v) const { base[k * F::STAGE_SZ + F::F_D + j] = v; }
  template <int F0, int COUNT> MPC_HD void store_run(int k, int I, const R *v) const {
    for (int j = 0; j < COUNT; j++) base[k * F::STAGE_SZ + I + F0 + j] = v[j];
  }
  MPC_HD void stage_fetch_it(int, int, int) const {}
  MPC_HD void stage_fetch_itf(int, int, int) const {}
  MPC_HD void stage_fetch_d(int, int) const {}
  template <int N> MPC_HD void stage_wait() const {}
  MPC_HD void stage_drain() const {}
  MPC_HD R sit(int, int k, int I, int j) const { return base[k * F::STAGE_SZ + I + j]; }
  MPC_HD R sx(int, int k, int Fo, int j) const { return base[k * F::STAGE_SZ + Fo + j]; }
  MPC_HD void stage_fetch_g(int, int, int) const {}
  MPC_HD R sg(int, int k, int J, int j) const { return base[k * F::STAGE_SZ + J + F::F_GK + j]; }
};

#if defined(__HIPCC__)
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) float gfloat;
typedef __attribute__((address_space(1))) char gchar;
typedef __attribute__((address_space(3))) double ldouble;
typedef __attribute__((address_space(3))) float lfloat;
typedef __attribute__((address_space(3))) char lchar;
#else   /* host pass of hipcc: the kernel body is parsed but never run */
typedef double gdouble;
typedef float gfloat;
typedef char gchar;
typedef double ldouble;
typedef float lfloat;
typedef char lchar;
#endif
/* Device layout: the workspace is tiled per wavefront and, inside a tile, fields are interleaved in
 * GROUPS of 16 bytes per lane:  [wave][stage][group][64 lanes][G], G = 2 doubles or 4 floats.  One wave's
 * whole working set is one contiguous block (fp64, N=10: 26 x 9 x 1 KB = 234 KB), and the fields of a group of
 * one instance are 16 contiguous bytes, so
 *   - a group moves with one 16-byte-per-lane access (global_load/store_dwordx4: the coalescing sweet spot),
 *   - `global_load_lds_dwordx4` (LDS-DMA) moves each lane's OWN data, which keeps it correct under the
 *     partial exec masks of a wave whose instances are in different solver phases.
 * Each sweep double-buffers the next stage's record into LDS with LDS-DMA while it computes the current
 * stage: the kernel needs all 512 registers (one wave per SIMD), so nothing else can hide the
 * HBM / Infinity-Cache latency, and a register prefetch does not fit (it spilled 428 VGPRs).
 * LDS: 2 buffers x 18 groups x 1 KB = 36 KB per wave in fp64 (144 KB per CU at four waves), 2 x 10 x 1 KB = 20 KB in fp32.
 * All pointers are typed into their address space: accesses are global_* / ds_* instructions, never flat_*;
 * the tile base is wave-uniform and byte offsets are formed in 32 bits (saddr + voffset addressing). */
/* Addressing: the byte offset of (stage k, field) is wave-uniform and goes through the scalar unit into the
 * instruction's SGPR base; the only vector part is `lane16` (+ the per-lane choice of the iterate slot, whose
 * two values differ between lanes in different solver phases).  Formed any other way the compiler keeps one
 * pre-computed vector offset per field alive across the sweeps, spills them, and every reload sits between a
 * prefetch and its use -- with in-order vmcnt that turns each prefetch into a synchronous load. */
#if defined(__HIP_DEVICE_COMPILE__)
/* the empty asm pins the value in an SGPR at the point of use: without it the loop-invariant addresses of
 * the sweeps' first fetches are hoisted out of the solver loop as ~20 vector pairs and spilled */
__device__ __forceinline__ unsigned mpc_uniform(unsigned x) {
  unsigned s = (unsigned)__builtin_amdgcn_readfirstlane((int)x);
  asm volatile("" : "+s"(s));
  return s;
}
#define MPC_UNIFORM(x) mpc_uniform((unsigned)(x))
#else
#define MPC_UNIFORM(x) ((unsigned)(x))
#endif
template <class R> struct AddrSpace;
template <> struct AddrSpace<double> { typedef gdouble g; typedef ldouble l; };
template <> struct AddrSpace<float> { typedef gfloat g; typedef lfloat l; };

template <bool STAGING, class R>
struct TiledWorkspace {
  using F = Fields<R>;
  typedef typename AddrSpace<R>::g greal;
  typedef typename AddrSpace<R>::l lreal;
  greal *tile;   /* this wave's tile */
  lreal *lbuf;   /* LDS staging area of this wave (STAGING) */
  int lane;
  static constexpr unsigned G = F::G;
  static constexpr unsigned GROUPS = F::STAGE_SZ / F::G;    /* 16-byte groups per stage */
  static_assert(F::STAGE_SZ % F::G == 0 && F::IT_SZ % F::G == 0, "whole groups");
  /* uniform part: row of (stage k, field f) + position inside the group; vector part: lane and slot */
  MPC_HD gchar *row(int k, int f) const {
    return (gchar *)tile + MPC_UNIFORM(((unsigned)k * GROUPS + ((unsigned)f / G)) * 1024u + ((unsigned)f % G) * (unsigned)sizeof(R));
  }
  MPC_HD unsigned voff(int I) const { return (unsigned)lane * 16u + ((unsigned)I / G) * 1024u; }
  MPC_HD greal &it(int k, int I, int f) const { return *(greal *)(row(k, f) + voff(I)); }
  /* COUNT fields from field F0 (both multiples of G) with ONE 16-byte store per group: the number of store
   * instructions per stage is then exact, which the counted waits of the staged sweeps rely on */
  template <int F0, int COUNT> MPC_HD void store_run(int k, int I, const R *v) const {
    static_assert(F0 % F::G == 0 && COUNT % F::G == 0, "whole groups");
    typedef R __attribute__((ext_vector_type(16 / sizeof(R)))) vec;
    typedef __attribute__((address_space(1))) vec gvec;
    MPC_UNROLL
    for (int g = 0; g < COUNT / (int)G; g++) {
      vec x;
      MPC_UNROLL
      for (int e = 0; e < (int)G; e++) x[e] = v[g * (int)G + e];
      *(gvec *)(row(k, F0 + g * (int)G) + voff(I)) = x;
    }
  }
  MPC_HD R getD(int k, int j) const { return it(k, 0, F::F_D + j); }
  MPC_HD void setD(int k, int j, R v) const { it(k, 0, F::F_D + j) = v; }
  /* ---- staging ---- */
  /* Copies NG consecutive groups.  Rows are 1 KB apart in the tile AND in the LDS slot, and the
   * instruction's immediate offset applies to both addresses, so four copies share one scalar row base and
   * one M0 value (immediates 0, 1024, 2048, 3072).  Written as assembly because the compiler expands the
   * builtin's offset argument back into per-copy address arithmetic (5 issue slots per copy instead of <2).
   * The compiler does not see these as memory instructions; that only makes its own vmcnt waits more
   * conservative (vmcnt completes in order), and the sweeps order everything staged with explicit waits.
   * M0 is written by every statement and is in its clobber list: the compiler keeps nothing of its own there across one. */
  /* cache policy of the staging loads (A/B builds: -DMPC_DMA_NT=1 marks them non-temporal) */
#if defined(MPC_DMA_NT) && MPC_DMA_NT
#define MPC_DMA_POLICY " nt"
#else
#define MPC_DMA_POLICY ""
#endif
  template <int NG>
  MPC_HD void dma(int buf, int k, int I, int f0, int dst_group) const {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned vo = voff(I);
    MPC_UNROLL
    for (int q0 = 0; q0 < NG; q0 += 4) {
      const unsigned m0v = MPC_UNIFORM((unsigned)(unsigned long)lbuf + (((unsigned)buf * F::STG_SLOT + (unsigned)dst_group + (unsigned)q0) * 64u) * 16u);
      const gchar *src = row(k, f0 + (int)G * q0);
      constexpr int n = (NG - 0);
      if (q0 + 4 <= n)
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" MPC_DMA_POLICY "\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024" MPC_DMA_POLICY "\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:2048" MPC_DMA_POLICY "\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072" MPC_DMA_POLICY
                     :: "v"(vo), "s"(src), "s"(m0v) : "memory", "m0");
      else if (q0 + 3 == n)
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" MPC_DMA_POLICY "\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024" MPC_DMA_POLICY "\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:2048" MPC_DMA_POLICY
                     :: "v"(vo), "s"(src), "s"(m0v) : "memory", "m0");
      else if (q0 + 2 == n)
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" MPC_DMA_POLICY "\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024" MPC_DMA_POLICY
                     :: "v"(vo), "s"(src), "s"(m0v) : "memory", "m0");
      else
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" MPC_DMA_POLICY :: "v"(vo), "s"(src), "s"(m0v) : "memory", "m0");
    }
#endif
  }
  MPC_HD void stage_fetch_it(int buf, int k, int I) const { if (STAGING) dma<F::STG_IT_OPS>(buf, k, I, 0, 0); }
  /* the forward sweep does not read the multipliers: (s, u) and the bound duals go to their usual places */
  MPC_HD void stage_fetch_itf(int buf, int k, int I) const {
    if (STAGING) { dma<8 / F::G>(buf, k, I, F::F_S, F::F_S / F::G); dma<8 / F::G>(buf, k, I, F::F_ZL, F::F_ZL / F::G); }
  }
  MPC_HD void stage_fetch_d(int buf, int k) const { if (STAGING) dma<F::STG_D_OPS>(buf, k, 0, F::F_D, F::STG_IT_OPS); }
  /* the gains, in the first fields of iterate slot J (the one that is not current) */
  MPC_HD void stage_fetch_g(int buf, int k, int J) const { if (STAGING) dma<F::STG_X_OPS>(buf, k, J, F::F_GK, F::STG_IT_OPS); }
  /* Between sweeps: a sweep's first copies read what the sweep before it stored, so all earlier stores of the
   * wave are waited for first.  (Vector memory operations of one wave are performed in order, and with
   * -DMPC_NO_DRAIN the results stay bitwise identical; the wait costs nothing measurable -- same-box A/B 2.02 vs
   * 2.04 ms at 65 536 instances -- so the explicit form is kept.) */
  MPC_HD void stage_drain() const {
#if !defined(MPC_NO_DRAIN)
    stage_wait<0>();
#endif
  }
  template <int N> MPC_HD void stage_wait() const {
#if defined(__HIP_DEVICE_COMPILE__)
    if (STAGING) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
  }
  MPC_HD R sl(int buf, int j) const {
    return lbuf[(((unsigned)buf * F::STG_SLOT + ((unsigned)j / G)) * 64u + (unsigned)lane) * G + ((unsigned)j % G)];
  }
  MPC_HD R sit(int buf, int k, int I, int j) const { return STAGING ? sl(buf, j) : (R)it(k, I, j); }
  MPC_HD R sx(int buf, int k, int Fo, int j) const { return STAGING ? sl(buf, F::IT_SZ + j) : (R)it(k, 0, Fo + j); }
  MPC_HD R sg(int buf, int k, int J, int j) const { return STAGING ? sl(buf, F::IT_SZ + j) : (R)it(k, J, F::F_GK + j); }
};

/* The N-step variables of an instance RESIDENT IN LDS: for launches of at most a few thousand instances (one or a few
 * per CU) the whole per-instance state -- two iterate slots, direction, gains: 3.7 KB at N = 10 in fp64 -- lives in the
 * workgroup's LDS, [stage][field][LANES], LANES instances per wave.  No workspace traffic leaves the CU at all and a
 * field is ~64 cycles away instead of an L2 / Infinity-Cache round trip, which is what a lone wave's stage step waits
 * for in the streaming kernel (one MPC::solve() per telemetry message is the reference's own use: B = 1).
 * The chip holds 160 KB x 256 of LDS = 10 900 instances this way (N = 10, fp64), far fewer than the lanes it has: large
 * batches stream (TiledWorkspace).  Same Solver, same arithmetic: results are bitwise those of the streaming kernel. */
template <class R, int LANES>
struct LdsWorkspace {
  using F = Fields<R>;
  typedef typename AddrSpace<R>::l lreal;
  lreal *base;   /* this workgroup's LDS */
  int lane;      /* < LANES */
  MPC_HD lreal &it(int k, int I, int f) const { return base[(unsigned)((k * F::STAGE_SZ + I + f) * LANES + lane)]; }
  MPC_HD R getD(int k, int j) const { return it(k, 0, F::F_D + j); }
  MPC_HD void setD(int k, int j, R v) const { it(k, 0, F::F_D + j) = v; }
  template <int F0, int COUNT> MPC_HD void store_run(int k, int I, const R *v) const {
    MPC_UNROLL
    for (int j = 0; j < COUNT; j++) it(k, I, F0 + j) = v[j];
  }
  MPC_HD void stage_fetch_it(int, int, int) const {}
  MPC_HD void stage_fetch_itf(int, int, int) const {}
  MPC_HD void stage_fetch_d(int, int) const {}
  MPC_HD void stage_fetch_g(int, int, int) const {}
  template <int N> MPC_HD void stage_wait() const {}
  MPC_HD void stage_drain() const {}
  MPC_HD R sit(int, int k, int I, int j) const { return it(k, I, j); }
  MPC_HD R sx(int, int k, int Fo, int j) const { return it(k, 0, Fo + j); }
  MPC_HD R sg(int, int k, int J, int j) const { return it(k, J, F::F_GK + j); }
};
#endif

/* ---- light-weight math (same code on device and in the test-only host build) ---- */
/* reciprocal: v_rcp_f64 seed (measured on gfx950: 4.6e-8 relative, tools/rcp_test.hip) + Newton steps;
 * a full IEEE division costs ~3x as many instructions and the barrier terms need dozens of 1/slack per stage.
 *   frcp   two steps, 1.1e-16: where the quotient is a result (atan, log)
 *   frcp1  one step,  2.2e-15: slack and pivot reciprocals, which only shape the Newton system (the
 *          optimality error of a point is evaluated from the duals themselves, never from 1/slack) */
MPC_HD double frcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  return r;
#else
  return 1.0 / x;
#endif
}
/* Cross-lane reads of the one-instance-per-LPI-lanes kernels.  LPI = 64 (one instance per wavefront): the value lane `lane`
 * (wave-uniform) holds, in every lane -- v_readlane_b32 per 32 bits.  LPI < 64 (several instances per wavefront, each on LPI
 * neighbouring lanes): the value lane `base + lane` holds, `base` being the first lane of the reader's own group --
 * ds_bpermute_b32 per 32 bits (the instances of a wave may be in different phases: a wave-uniform source does not exist). */
#if defined(__HIP_DEVICE_COMPILE__)
template <int LPI> __device__ __forceinline__ int wave_bcast_i(int x, int lane, int base) {
  if constexpr (LPI >= 64) return __builtin_amdgcn_readlane(x, lane);
  else return __builtin_amdgcn_ds_bpermute((base + lane) << 2, x);
}
template <int LPI> __device__ __forceinline__ double wave_bcast(double x, int lane, int base) {
  const int lo = wave_bcast_i<LPI>(__double2loint(x), lane, base), hi = wave_bcast_i<LPI>(__double2hiint(x), lane, base);
  return __hiloint2double(hi, lo);
}
template <int LPI> __device__ __forceinline__ float wave_bcast(float x, int lane, int base) { return __int_as_float(wave_bcast_i<LPI>(__float_as_int(x), lane, base)); }
template <int LPI> __device__ __forceinline__ bool wave_bcast_flag(bool x, int lane, int base) { return wave_bcast_i<LPI>(x ? 1 : 0, lane, base) != 0; }
/* does any lane of the reader's group say so? */
template <int LPI> __device__ __forceinline__ bool wave_group_any(bool x, int base) {
  const unsigned long long b = __builtin_amdgcn_ballot_w64(x);
  if constexpr (LPI >= 64) return b != 0ull;
  else return ((b >> base) & ((1ull << LPI) - 1ull)) != 0ull;
}
#else
template <int LPI> inline double wave_bcast(double x, int, int) { return x; }
template <int LPI> inline float wave_bcast(float x, int, int) { return x; }
template <int LPI> inline bool wave_bcast_flag(bool x, int, int) { return x; }
template <int LPI> inline bool wave_group_any(bool x, int) { return x; }
#endif

MPC_HD double frcp1(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(x);
  const double e = fma(-x, r, 1.0);
  return fma(r, e, r);
#else
  return 1.0 / x;
#endif
}
/* x^p for the line-search switching heuristics only (thresholds, not results): single precision */
MPC_HD double hpow(double x, double p) {
#if defined(__HIP_DEVICE_COMPILE__)
  const float xf = (float)fmin(x, 1e16);
  return (double)__builtin_exp2f((float)p * __builtin_log2f(xf));
#else
  return pow(x, p);
#endif
}
/* An fp64 constant for a polynomial kernel, materialised in a scalar register pair AT THE POINT OF USE
 * (two s_mov_b32, which issue beside the vector work).  gfx9 encodings have no 64-bit literals, so the
 * compiler otherwise parks every coefficient in an accumulation register and pays two v_accvgpr_read plus
 * two v_mov per Horner step -- four overhead instructions per FMA in the hottest loops. */
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double mpc_kc(double c) {
  asm volatile("" : "+s"(c));
  return c;
}
#define MPC_K(c) mpc_kc(c)
#else
#define MPC_K(c) (c)
#endif

/* sin and cos together for the moderate angles of this model (psi, epsi): Cody-Waite reduction by
 * pi/2 (two FMAs: the products are exact inside the FMA, so the reduction holds while the quadrant fits an
 * int, |x| < 1e9) and the classic minimax kernels on [-pi/4, pi/4]
 * (coefficients as published in fdlibm's k_sin.c / k_cos.c).  libm's general path (Payne-Hanek reduction) is
 * deliberately not linked in: it costs ~60 permanently occupied registers for its constants. */
/* one reduced argument: r in [-pi/4, pi/4] and the quadrant */
MPC_HD void fsincos_finish(double r, double z, double ps, double pc, int q, double *sn, double *cs) {
  const double s0 = fma(r * z, ps, r);
  const double c0 = fma(z * z, pc, fma(-0.5, z, 1.0));
  const double s1 = (q & 1) ? c0 : s0, c1 = (q & 1) ? s0 : c0;
  *sn = (q & 2) ? -s1 : s1;
  *cs = ((q + 1) & 2) ? -c1 : c1;
}
/* two angles at once (psi and epsi of a stage): every constant is materialised once for both, and the two
 * Horner chains interleave */
MPC_HD void fsincos2(double xa, double xb, double *sna, double *csa, double *snb, double *csb) {
  if (!(fabs(xa) < 1.0e9)) xa = NAN;   /* no such angle in this model: the evaluation is flagged, the trial step rejected */
  if (!(fabs(xb) < 1.0e9)) xb = NAN;
  double c = MPC_K(6.36619772367581382433e-01);
  const double ka = rint(xa * c), kb = rint(xb * c);
  c = MPC_K(1.57079632679489655800e+00);
  double ra = fma(-ka, c, xa), rb = fma(-kb, c, xb);
  c = MPC_K(6.12323399573676603587e-17);
  ra = fma(-ka, c, ra); rb = fma(-kb, c, rb);
  const double za = ra * ra, zb = rb * rb;
  double psa = MPC_K(1.58969099521155010221e-10), psb = psa;
#define MPC_H2(pa, pb, k) do { const double c_ = MPC_K(k); pa = fma(za, pa, c_); pb = fma(zb, pb, c_); } while (0)
  MPC_H2(psa, psb, -2.50507602534068634195e-08);
  MPC_H2(psa, psb, 2.75573137070700676789e-06);
  MPC_H2(psa, psb, -1.98412698298579493134e-04);
  MPC_H2(psa, psb, 8.33333333332248946124e-03);
  MPC_H2(psa, psb, -1.66666666666666324348e-01);
  double pca = MPC_K(-1.13596475577881948265e-11), pcb = pca;
  MPC_H2(pca, pcb, 2.08757232129817482790e-09);
  MPC_H2(pca, pcb, -2.75573143513906633035e-07);
  MPC_H2(pca, pcb, 2.48015872894767294178e-05);
  MPC_H2(pca, pcb, -1.38888888888741095749e-03);
  MPC_H2(pca, pcb, 4.16666666666666019037e-02);
#undef MPC_H2
  fsincos_finish(ra, za, psa, pca, (int)ka & 3, sna, csa);
  fsincos_finish(rb, zb, psb, pcb, (int)kb & 3, snb, csb);
}
MPC_HD void fsincos(double x, double *sn, double *cs) {
  double s2, c2;
  fsincos2(x, x, sn, cs, &s2, &c2);
}

/* atan for the road slope f'(x):  |x| > 1 -> pi/2 - atan(1/|x|);  on [0,1]  atan(t) = t + t z q(z), z = t^2,
 * q of degree 19 (interpolant at the Chebyshev nodes of [0,1], computed with 60 digits; approximation error
 * 8e-17, measured total error < 4.1e-16 relative), evaluated as two interleaved chains in z^2. */
MPC_HD double fatan(double x) {
  const double ax = fabs(x);
  const bool inv = ax > 1.0;
  const double t = inv ? frcp(ax) : ax;
  const double z = t * t, w = z * z;
  double e = MPC_K(-1.99961893779013817792e-04), o = MPC_K(1.80619546186121510795e-05);
  e = fma(e, w, MPC_K(-3.49588597391630936939e-03)); o = fma(o, w, MPC_K(1.04960350849685147764e-03));
  e = fma(e, w, MPC_K(-1.55351524754141767648e-02)); o = fma(o, w, MPC_K(8.36893117845016222545e-03));
  e = fma(e, w, MPC_K(-3.12771890669963845144e-02)); o = fma(o, w, MPC_K(2.36967315800486223731e-02));
  e = fma(e, w, MPC_K(-4.26035663260165217703e-02)); o = fma(o, w, MPC_K(3.74948681253524720991e-02));
  e = fma(e, w, MPC_K(-5.25797333428411062251e-02)); o = fma(o, w, MPC_K(4.73774957952777867054e-02));
  e = fma(e, w, MPC_K(-6.66656469928910422329e-02)); o = fma(o, w, MPC_K(5.88150687779365605179e-02));
  e = fma(e, w, MPC_K(-9.09090859089193431553e-02)); o = fma(o, w, MPC_K(7.69229897103321652585e-02));
  e = fma(e, w, MPC_K(-1.42857142853841323493e-01)); o = fma(o, w, MPC_K(1.11111110934908274839e-01));
  e = fma(e, w, MPC_K(-3.33333333333333314830e-01)); o = fma(o, w, MPC_K(1.99999999999975308640e-01));
  const double q = fma(o, z, e);
  double r = fma(t * z, q, t);
  if (inv) r = MPC_K(1.57079632679489655800e+00) - (r - MPC_K(6.12323399573676603587e-17));
  return copysign(r, x);
}

/* natural logarithm for the barrier term: x = m 2^k with m in [sqrt(1/2), sqrt(2)), s = (m-1)/(m+1),
 * log m = 2s + s R(s^2) in the compensated form of fdlibm's e_log.c (coefficients Lg1..Lg7 as published
 * there; < 1 ulp).  x > 0 and finite is the caller's business (slack products are checked before). */
MPC_HD double flog(double x) {
  int k;
  double m = frexp(x, &k);                        /* m in [0.5, 1) */
  if (m < 7.07106781186547524401e-01) { m *= 2.0; k -= 1; }
  const double f = m - 1.0;
  const double s = f * frcp(2.0 + f);
  const double z = s * s, w = z * z;
  double t1 = MPC_K(1.531383769920937332e-01), t2 = MPC_K(1.479819860511658591e-01);
  t1 = fma(t1, w, MPC_K(2.222219843214978396e-01)); t2 = fma(t2, w, MPC_K(1.818357216161805012e-01));
  t1 = fma(t1, w, MPC_K(3.999999999940941908e-01)); t2 = fma(t2, w, MPC_K(2.857142874366239149e-01));
  t2 = fma(t2, w, MPC_K(6.666666666666735130e-01));
  const double R = w * t1 + z * t2;
  const double hfsq = 0.5 * f * f, dk = (double)k;
  return dk * MPC_K(6.93147180369123816490e-01) - ((hfsq - (s * (hfsq + R) + dk * MPC_K(1.90821492927058770002e-10))) - f);
}


/* ---- single-precision versions (the MPC_PRECISION_F32 solver) ----------------------------------------------
 * v_rcp_f32 is accurate to 1 ulp, so no Newton step; sin/cos: Cody-Waite reduction by pi/2 in three parts and
 * polynomial kernels of degree 9/6 on [-pi/4, pi/4] (interpolants at Chebyshev nodes, |err| < 8e-8);
 * atan: reciprocal for |x| > 1, odd polynomial of degree 19 on [0,1]; log: the hardware's log2. */
MPC_HD float frcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(x);
#else
  return 1.0f / x;
#endif
}
MPC_HD float frcp1(float x) { return frcp(x); }
MPC_HD float hpow(float x, float p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_exp2f(p * __builtin_log2f(fminf(x, 1e16f)));
#else
  return powf(x, p);
#endif
}
MPC_HD void fsincos_finish(float r, float z, float ps, float pc, int q, float *sn, float *cs) {
  const float s0 = fmaf(r * z, ps, r);
  const float c0 = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
  const float s1 = (q & 1) ? c0 : s0, c1 = (q & 1) ? s0 : c0;
  *sn = (q & 2) ? -s1 : s1;
  *cs = ((q + 1) & 2) ? -c1 : c1;
}
MPC_HD void fsincos2(float xa, float xb, float *sna, float *csa, float *snb, float *csb) {
  if (!(fabsf(xa) < 1.0e4f)) xa = NAN;   /* the three-part reduction below is exact for quadrant counts < 2^13 */
  if (!(fabsf(xb) < 1.0e4f)) xb = NAN;
  const float ka = rintf(xa * 6.36619772e-01f), kb = rintf(xb * 6.36619772e-01f);
  /* pi/2 = 1.5703125 + 4.83751297e-4 + 7.549789954e-8 (the first two parts have few mantissa bits: k * part is exact) */
  float ra = fmaf(-ka, 1.5703125f, xa), rb = fmaf(-kb, 1.5703125f, xb);
  ra = fmaf(-ka, 4.83751297e-4f, ra); rb = fmaf(-kb, 4.83751297e-4f, rb);
  ra = fmaf(-ka, 7.549789954e-8f, ra); rb = fmaf(-kb, 7.549789954e-8f, rb);
  const float za = ra * ra, zb = rb * rb;
  float psa = 2.7249925803e-06f, psb = psa;
#define MPC_H2F(pa, pb, k) do { pa = fmaf(za, pa, k); pb = fmaf(zb, pb, k); } while (0)
  MPC_H2F(psa, psb, -1.9840086735e-04f);
  MPC_H2F(psa, psb, 8.3333318747e-03f);
  MPC_H2F(psa, psb, -1.6666666664e-01f);
  float pca = 2.4547942085e-05f, pcb = pca;
  MPC_H2F(pca, pcb, -1.3888303036e-03f);
  MPC_H2F(pca, pcb, 4.1666664660e-02f);
#undef MPC_H2F
  fsincos_finish(ra, za, psa, pca, (int)ka & 3, sna, csa);
  fsincos_finish(rb, zb, psb, pcb, (int)kb & 3, snb, csb);
}
MPC_HD void fsincos(float x, float *sn, float *cs) {
  float s2, c2;
  fsincos2(x, x, sn, cs, &s2, &c2);
}
MPC_HD float fatan(float x) {
  const float ax = fabsf(x);
  const bool inv = ax > 1.0f;
  const float t = inv ? frcp(ax) : ax;
  const float z = t * t;
  /* atan(t) = t + t z q(z) on [0,1], q of degree 8 (interpolant at the Chebyshev nodes; total error 7e-8) */
  float q = -2.3869973167e-03f;
  q = fmaf(q, z, 1.3507771427e-02f);
  q = fmaf(q, z, -3.5871538982e-02f);
  q = fmaf(q, z, 6.2501694447e-02f);
  q = fmaf(q, z, -8.6568804302e-02f);
  q = fmaf(q, z, 1.1033764115e-01f);
  q = fmaf(q, z, -1.4278568748e-01f);
  q = fmaf(q, z, 1.9999739330e-01f);
  q = fmaf(q, z, -3.3333331733e-01f);
  float r = fmaf(t * z, q, t);
  if (inv) r = 1.57079637e+00f - r;
  return copysignf(r, x);
}
MPC_HD float flog(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_logf(x) * 6.93147182e-01f;   /* v_log_f32 = log2, 1 ulp */
#else
  return logf(x);
#endif
}
/* conversions between the solver's real type and the fp64 pieces it keeps (road polynomial, merit sums) */
MPC_HD double mpc_abs(double x) { return fabs(x); }
MPC_HD float mpc_abs(float x) { return fabsf(x); }
MPC_HD double mpc_max(double a, double b) { return fmax(a, b); }
MPC_HD float mpc_max(float a, float b) { return fmaxf(a, b); }
MPC_HD double mpc_min(double a, double b) { return fmin(a, b); }
MPC_HD float mpc_min(float a, float b) { return fminf(a, b); }
MPC_HD double mpc_sqrt(double a) { return sqrt(a); }
MPC_HD float mpc_sqrt(float a) { return sqrtf(a); }

/* IPOPT default constants (Waechter & Biegler 2006; IPOPT 3.12 option defaults) */
template <class R>
struct IpmConst {
#ifndef MPC_KAPPA_MU
#define MPC_KAPPA_MU 0.2
#endif
#ifndef MPC_MU_INIT
#define MPC_MU_INIT 0.1
#endif
#ifndef MPC_KAPPA_EPS
#define MPC_KAPPA_EPS 10.0
#endif
#ifndef MPC_TAU_MIN
#define MPC_TAU_MIN 0.99
#endif
  static constexpr R kappa_eps = MPC_KAPPA_EPS, kappa_mu = MPC_KAPPA_MU, theta_mu = 1.5, tau_min = MPC_TAU_MIN, s_max = 100.0;
  static constexpr R gamma_theta = 1e-5, gamma_phi = 1e-8, delta_sw = 1.0, s_theta = 1.1, s_phi = 2.3;
  static constexpr R eta_phi = 1e-8, gamma_alpha = 0.05, kappa_sigma = 1e10, kappa1 = 1e-2, kappa2 = 1e-2;
  static constexpr R dw_min = 1e-20, dw_0 = 1e-4, dw_max = sizeof(R) == 8 ? 1e40 : 1e30, kw_minus = 1.0 / 3.0, kw_plus = 8.0;
  static constexpr R kw_plus_bar = 100.0, mu_init = MPC_MU_INIT, eps = sizeof(R) == 8 ? 2.220446049250313e-16 : 1.1920928955078125e-07;
  static constexpr R inv_kappa_sigma = 1.0 / 1e10, huge = sizeof(R) == 8 ? 1e300 : 1e37;
};

/* Vehicle::computeSpeedTarget, src/model/Vehicle.cpp:34-64 */
MPC_HD double speed_target(const MpcParams &P, double angle, double maxv) {
  double y = fabs(angle);
  int last = P.n_steer_speeds - 1;
  for (int i = 0; i < P.n_steers; i++) {
    if (y <= P.steers[i]) {
      if (P.n_steer_speeds > i) return fmin(P.steer_speeds[i], maxv);
      return fmin(P.steer_speeds[last], maxv);
    }
  }
  return fmin(P.steer_speeds[last], maxv);
}

/* what a trial-point evaluation returns */
template <class R>
struct Eval {
  R theta;  /* ||c||_1            */
  R cinf;   /* ||c||_inf          */
  R f;      /* unscaled objective without the stage-0 constant */
  R L;      /* sum of log(slack)  */
  R dinf;   /* ||grad_x Lagrangian||_inf (scaled objective) */
  R cmin, cmax; /* range of slack*dual products */
  R lsum, zsum; /* ||lam||_1, ||z||_1 */
  R du0;        /* max(|d delta_0|, |d a_0|) of the direction this trial was made with (unscaled by alpha) */
  bool ok;
};

/* linearisation of one stage at (s_k, u_k) and its residual c_{k+1} = s_{k+1} - F(s_k, u_k) */
template <class R>
struct Lin {
  R sp, cp, se, ce;   /* sin/cos of psi_k and epsi_k */
  R fp, g1, h3, fpp;  /* f'(x_k), f''/(1+f'^2), d/dx of that, f'' */
  R c[6];
};

/* WAVE = LPI > 0: one instance per LPI neighbouring lanes of a wavefront (mpc_solve_wave_kernel; 64 = the whole wave): every lane
 * of the group runs this state machine on the same instance, `wlane` is the lane's number in its group, `wbase` the group's first
 * lane, and the sweeps share their stages between the lanes of the group */
template <class WS, class R, int WAVE = 0>
struct Solver {
  using F = Fields<R>;
  using IC = IpmConst<R>;
  typedef Eval<R> EvalR;
  typedef Lin<R> LinR;
  static constexpr int F_S = F::F_S, F_U = F::F_U, F_LAM = F::F_LAM, F_ZL = F::F_ZL, F_ZU = F::F_ZU, IT_SZ = F::IT_SZ;
  static constexpr int IT0 = F::IT0, IT1 = F::IT1, F_D = F::F_D, D_N = F::D_N, D_S = F::D_S, D_U = F::D_U;
  static constexpr int F_GK = F::F_GK, GK_N = F::GK_N, F_GF = F::F_GF, GAIN_SZ = F::GAIN_SZ;
  static constexpr int STG_IT_OPS = F::STG_IT_OPS, STG_ITF_OPS = F::STG_ITF_OPS, STG_X_OPS = F::STG_X_OPS, STG_D_OPS = F::STG_D_OPS;
  static constexpr int ST_BACKWARD = F::ST_BACKWARD, ST_FORWARD = F::ST_FORWARD, ST_TRIAL = F::ST_TRIAL;
  const MpcParams &P;
  WS ws;
  int wlane = 0, wbase = 0;
  /* instance data */
  R st[6], coef[MPC_NCOEF], yl, yu;
  R wc, we, wv, wd, wdd, vref, cost0;
  /* bounds */
  R vl, vu, dl, du, al, au;
  int M;       /* number of stages = N-1 */
  R dt, dtLf, iLf, psi_start;
  /* psi_0 and v_0 as IPOPT sees them.  In the reference's NLP the initial state is not data: vars[k*N] are VARIABLES
   * (with the bounds of their blocks, MPC.cpp:229-239) pinned by equality rows (MPC.cpp:116-121, 269-281).  IPOPT
   * pushes a start value that lies within kappa1/kappa2 of a bound into the interior, and the Newton steps then bring
   * it back at the pace the fraction-to-the-boundary rule allows: x0^{k+1} = x0^k + alpha_k (state - x0^k).  For the four
   * unbounded components that never does anything; psi_0 (a closed loop whose heading has reached its bound,
   * test.cpp:79-111) and v_0 (a car at Config::maxSpeed) take a few iterations, during which stage 0 is linearised at
   * (psi_0^k, v_0^k), its residual counts in theta and its slacks limit the step.  These two scalars carry exactly that;
   * the multipliers of the six pinning rows (lam0) and the bound duals of psi_0, v_0 (z0) decouple from the step -- the rows
   * hold a free multiplier with unit coefficient -- but not from IPOPT's error measure: the stationarity rows of the six
   * initial-state variables count in the dual infeasibility (after a step the fraction-to-the-boundary rule has cut they hold
   * (1 - alpha) of their old residual plus the curvature of stage 0 along the step), lam0 counts in its scaling, z0 in the
   * complementarity and in the dual step length.  They are carried for that: costate_trial() treats them as the record of a
   * stage "-1".  -DMPC_S0_VARIABLE=0 builds the solver with the initial state as plain data (A/B
   * measurements only). */
#ifndef MPC_S0_VARIABLE
#define MPC_S0_VARIABLE 1
#endif
#if MPC_S0_VARIABLE
  R p0, v0k;
  R wc0, we0, vref0, wneg0;   /* the i = 0 terms of the objective (MPC.cpp:71-92), whose variables are the initial state's */
  bool s0_rows = false;       /* MpcParams.initial_state_rows: lam0 / z0 carried and counted in the error measure */
  /* lam0 and z0 live in the workspace, in the iterate slots of a record of their own behind the last stage's (index M: lam0 in
   * the multipliers' fields, the duals of psi_0 / v_0 where a stage keeps those of psi_{k+1} / v_{k+1}): current and trial
   * values swap with `cur` like every other record, and the solver's registers are left to the sweeps */
  MPC_HD R s0_lam(int I, int i) const { return ws.it(M, I, F_LAM + i); }
  MPC_HD R s0_z(int I, int b, int upper) const { return ws.it(M, I, (upper ? F_ZU : F_ZL) + b); }
  MPC_HD void s0_reset(int I) {
    MPC_UNROLL
    for (int i = 0; i < 6; i++) ws.it(M, I, F_LAM + i) = R(0.0);
    MPC_UNROLL
    for (int b = 0; b < 2; b++) { ws.it(M, I, F_ZL + b) = R(1.0); ws.it(M, I, F_ZU + b) = R(1.0); }      /* bound_mult_init_val */
  }
#endif
  /* interior-point state */
  int cur;     /* slot of the current iterate */
  R mu, tau, df;
  EvalR E;
  /* direction summary */
  R amax, az, dphi, dxinf, xinf;
  /* filter: four entries in registers (it is emptied at every barrier update) */
  R fth0, fth1, fth2, fth3, fph0, fph1, fph2, fph3;
  int nf;
  int iters, n_reg;
  /* true only while the least-squares multiplier start is being computed: the sweeps then solve
   * [I J^T; J 0][w; lam] = -[grad f; 0] (identity Hessian, no barrier, zero constraint rhs) */
  bool lsm;

  MPC_HD Solver(const MpcParams &p, WS w) : P(p), ws(w) {}

  MPC_HD int it(int slot) const { return slot ? IT1 : IT0; }

  /* ---- road polynomial: RoadGeometry::centerY / orientation, utils.h:28-47 */
  /* Always evaluated in fp64, also by the fp32 solver: at x ~ 80 m the terms of a degree-4 Horner scheme reach
   * ~1e7 times the size of the cte they are compared with; the cost is a dozen fp64 FMAs. */
  MPC_HD void poly(R xr, R y, R &fmy, R &fp, R &fpp, R &fppp) const {   /* fmy = f(x) - y, the cte before the step */
    const double x = xr;
    const double c0 = coef[0], c1 = coef[1], c2 = coef[2], c3 = coef[3], c4 = coef[4];
    fmy = (R)(((((c4 * x + c3) * x + c2) * x + c1) * x + c0) - (double)y);
    fp = (R)(((4.0 * c4 * x + 3.0 * c3) * x + 2.0 * c2) * x + c1);
    fpp = (R)((12.0 * c4 * x + 6.0 * c3) * x + 2.0 * c2);
    fppp = (R)(24.0 * c4 * x + 6.0 * c3);
  }

  /* Stage model at (s,u), MPC.cpp:142-152, and the residual against the successor state sn.
   * Recomputed wherever it is needed (see the layout comment). */
  MPC_HD void linearise(const R *s, R delta, R a, const R *sn, LinR &L) const {
    fsincos2(s[2], s[5], &L.sp, &L.cp, &L.se, &L.ce);
    R fmy, fp, fpp, fppp;
    poly(s[0], s[1], fmy, fp, fpp, fppp);
    const R q1 = R(1.0) + fp * fp, iq1 = frcp1(q1);
    L.fp = fp;
    L.g1 = fpp * iq1;
    L.h3 = (fppp * q1 - R(2.0) * fp * fpp * fpp) * (iq1 * iq1);
    L.fpp = fpp;
    const R vdt = s[3] * dt;
    const R psin = s[2] + delta * vdt * iLf;
    if (sizeof(R) == 8) {
      L.c[0] = sn[0] - (s[0] + L.cp * vdt);
      L.c[1] = sn[1] - (s[1] + L.sp * vdt);
    } else {
      /* fp32: first the difference of the two neighbouring states (exact, or nearly: they are close), then the
       * increment -- x + v dt cos(psi) rounded at x ~ 80 m would cost 4e-6 m of residual */
      L.c[0] = (sn[0] - s[0]) - L.cp * vdt;
      L.c[1] = (sn[1] - s[1]) - L.sp * vdt;
    }
    L.c[2] = sn[2] - psin;
    L.c[3] = sn[3] - (s[3] + a * dt);
    L.c[4] = sn[4] - (fmy + L.se * vdt);
    L.c[5] = sn[5] - (psin - fatan(fp));
  }

  /* cost + barrier terms of one state s_k (k>=1): Hessian diagonal and gradient */
  MPC_HD void state_terms(R psi, R v, R c, R e, R zlp, R zup, R zlv,
                          R zuv, R &Hpp, R &Hvv, R &Hee, R &Hcc, R &gp,
                          R &gv, R &ge, R &gc) const {
    const R islp = frcp1(psi - yl), isup = frcp1(yu - psi), islv = frcp1(v - vl), isuv = frcp1(vu - v);
    const R mub = lsm ? R(0.0) : mu;
    Hpp = lsm ? R(1.0) : zlp * islp + zup * isup;
    Hvv = lsm ? R(1.0) : df * R(2.0) * wv + zlv * islv + zuv * isuv;
    Hee = lsm ? R(1.0) : df * R(2.0) * we;
    Hcc = lsm ? R(1.0) : df * R(2.0) * wc;
    gp = mub * (isup - islp);
    gv = df * R(2.0) * wv * (v - vref) + mub * (isuv - islv);
    ge = df * R(2.0) * we * e;
    gc = df * R(2.0) * wc * c;
  }

  MPC_HD void load_state(int k, int I, R *s) const {   /* s_k; k = 0 is the fixed initial state */
    if (k == 0) {
      MPC_UNROLL
      for (int i = 0; i < 6; i++) s[i] = st[i];
#if MPC_S0_VARIABLE
      s[2] = p0; s[3] = v0k;
#endif
    } else {
      MPC_UNROLL
      for (int i = 0; i < 6; i++) s[i] = ws.it(k - 1, I, F_S + i);
    }
  }
  /* IPOPT's push of a start value into the interior of its (relaxed) bounds, W&B section 3.6 */
  MPC_HD R pushed(R x, R lo, R hi) const {
    const R pl = mpc_min(IC::kappa1 * mpc_max(R(1.0), mpc_abs(lo)), IC::kappa2 * (hi - lo));
    const R pu = mpc_min(IC::kappa1 * mpc_max(R(1.0), mpc_abs(hi)), IC::kappa2 * (hi - lo));
    return mpc_min(mpc_max(x, lo + pl), hi - pu);
  }

  /* ------------------------------------------------------------------ */
  /* Riccati backward sweep: gains K_k, kff_k for every stage.           */
  /* Returns false when some R~_k is not positive definite (wrong        */
  /* inertia): the caller raises the regularisation dw and repeats.      */
  /* ------------------------------------------------------------------ */
  /* The Riccati sweep with one instance per wavefront: lane k prepares stage k (inputs, model, reciprocal slacks, control and state
   * terms); the value-function recursion runs through the lanes in descending order -- the stage algebra of backward(), statement
   * for statement, on the value function the lane before broadcast (29 numbers) -- and lane k stores stage k's gains. */
  MPC_HD bool backward_wave(R dw) {
    const int I = it(cur), J = it(1 - cur);         /* J: where the gains go */
    /* value function of (x,y,psi,v,e,d) [+ c] at stage k+1; only the lower triangle of the symmetric
     * matrices is ever written or read (PM/MX pick it), so the other half never occupies registers */
    R Pm[6][6], p[6], Pcc, pc;
#define PM(i, j) Pm[(i) >= (j) ? (i) : (j)][(i) >= (j) ? (j) : (i)]
#define MX(i, j) Mx[(i) >= (j) ? (i) : (j)][(i) >= (j) ? (j) : (i)]
    MPC_UNROLL
    for (int i = 0; i < 6; i++) {
      p[i] = 0;
      MPC_UNROLL
      for (int j = 0; j < 6; j++) Pm[i][j] = 0;
    }
    const R rsc = lsm ? R(0.0) : -R(1.0);             /* constraint right-hand side: -c, or 0 for the LS system */
    const R hxy = (lsm ? R(1.0) : R(0.0)) + dw;       /* x and y carry no cost: only the LS identity / regularisation */
    /* the value function behind the last stage (wave-uniform: backward()'s own start) */
    R sn[6];                                    /* s_{k+1} */
    MPC_UNROLL
    for (int i = 0; i < 6; i++) sn[i] = ws.sit(0, M - 1, I, F_S + i);
    {
      R Hpp, Hvv, Hee, Hcc, gp, gv, ge, gc;
      state_terms(sn[2], sn[3], sn[4], sn[5], ws.sit(0, M - 1, I, F_ZL + 0), ws.sit(0, M - 1, I, F_ZU + 0),
                  ws.sit(0, M - 1, I, F_ZL + 1), ws.sit(0, M - 1, I, F_ZU + 1), Hpp, Hvv, Hee, Hcc, gp, gv, ge, gc);
      Pm[0][0] = hxy; Pm[1][1] = hxy; Pm[2][2] = Hpp + dw; Pm[3][3] = Hvv + dw; Pm[4][4] = Hee + dw;
      Pcc = Hcc + dw; p[2] = gp; p[3] = gv; p[4] = ge; pc = gc;
    }
    /* ---- this lane's stage: inputs, model, reciprocal slacks, the control terms and the state terms of the value function ---- */
    const int k = wlane < M ? wlane : M - 1;      /* (the lanes behind the last stage repeat it; nothing of theirs is kept) */
    MPC_UNROLL
    for (int i = 0; i < 6; i++) sn[i] = ws.it(k, I, F_S + i);
    const R delta = ws.it(k, I, F_U + 0), acc = ws.it(k, I, F_U + 1);
    const R lx = ws.it(k, I, F_LAM + 0), ly = ws.it(k, I, F_LAM + 1), lp = ws.it(k, I, F_LAM + 2);
    const R lc = ws.it(k, I, F_LAM + 4), le = ws.it(k, I, F_LAM + 5);
    const R zld = ws.it(k, I, F_ZL + 2), zud = ws.it(k, I, F_ZU + 2), zla = ws.it(k, I, F_ZL + 3), zua = ws.it(k, I, F_ZU + 3);
    R sk[6];
    R zlp = 0, zup = 0, zlv = 0, zuv = 0, delprev = 0;
    if (k > 0) {
      MPC_UNROLL
      for (int i = 0; i < 6; i++) sk[i] = ws.it(k - 1, I, F_S + i);
      zlp = ws.it(k - 1, I, F_ZL + 0); zup = ws.it(k - 1, I, F_ZU + 0);
      zlv = ws.it(k - 1, I, F_ZL + 1); zuv = ws.it(k - 1, I, F_ZU + 1);
      delprev = ws.it(k - 1, I, F_U + 0);
    } else {
      MPC_UNROLL
      for (int i = 0; i < 6; i++) sk[i] = st[i];
#if MPC_S0_VARIABLE
      sk[2] = p0; sk[3] = v0k;
#endif
    }
    const R v = sk[3];
    LinR L;
    linearise(sk, delta, acc, sn, L);
    const R sp = L.sp, cp = L.cp, se = L.se, ce = L.ce, fp = L.fp, g1 = L.g1, h3 = L.h3, fpp = L.fpp;
    const R r0 = rsc * L.c[0], r1 = rsc * L.c[1], r2 = rsc * L.c[2], r3 = rsc * L.c[3], rc = rsc * L.c[4], r4 = rsc * L.c[5];
    const R vdt = v * dt;
    const R Axp = -vdt * sp, Axv = dt * cp, Ayp = vdt * cp, Ayv = dt * sp, Apv = delta * dtLf;
    const R Acx = fp, Acv = dt * se, Ace = vdt * ce, Aex = -g1, Bp = v * dtLf;
    /* the stage's own control terms */
    const R isld = frcp1(delta - dl), isud = frcp1(du - delta), isla = frcp1(acc - al), isua = frcp1(au - acc);
    R ddl = 0, Hdd = 0;
    if (k >= 1 && !lsm) { ddl = delta - delprev; Hdd = df * R(2.0) * wdd; }   /* LS start: all delta are 0 */
    const R mub = lsm ? R(0.0) : mu;
    const R gdel = df * R(2.0) * wd * delta + Hdd * ddl + mub * (isud - isld);
    const R gacc = mub * (isua - isla);
    /* control Hessian diagonal: cost + barrier, or the identity of the LS system */
    const R Sgd = lsm ? R(1.0) : df * R(2.0) * wd + zld * isld + zud * isud, Sga = lsm ? R(1.0) : zla * isla + zua * isua;
    /* ---- value function of stage k ---- */
    R Hpp, Hvv, Hee, Hcc, gp, gv, ge, gc;
    state_terms(sk[2], v, sk[4], sk[5], zlp, zup, zlv, zuv, Hpp, Hvv, Hee, Hcc, gp, gv, ge, gc);
    /* ---- the recursion: stage t is lane t's turn; its value function is the state the next lane starts from ---- */
    bool ok_all = true;
    for (int tk = M - 1; tk >= 0; --tk) {
      bool ok_k = true;
      /* t = p + P r (the d component of r is zero) */
      R t[6];
      MPC_UNROLL
      for (int i = 0; i < 6; i++)
        t[i] = p[i] + PM(i, 0) * r0 + PM(i, 1) * r1 + PM(i, 2) * r2 + PM(i, 3) * r3 + PM(i, 4) * r4;
      const R tc = pc + Pcc * rc;
      /* G^T applied to a (6-vector, c-scalar): outputs for inputs x,y,psi,v,e,delta,a */
#define MPC_GT(w, wcs, o)                                                         \
  do {                                                                            \
    const R w24_ = (w)[2] + (w)[4];                                          \
    (o)[0] = (w)[0] + Aex * (w)[4] + Acx * (wcs);                                 \
    (o)[1] = (w)[1] - (wcs);                                                      \
    (o)[2] = Axp * (w)[0] + Ayp * (w)[1] + w24_;                                  \
    (o)[3] = Axv * (w)[0] + Ayv * (w)[1] + Apv * w24_ + (w)[3] + Acv * (wcs);     \
    (o)[4] = Ace * (wcs);                                                         \
    (o)[5] = Bp * w24_ + (w)[5];                                                  \
    (o)[6] = dt * (w)[3];                                                         \
  } while (0)
      R qt[7];
      MPC_GT(t, tc, qt);
      const R rt_d = qt[5] + gdel, rt_a = qt[6] + gacc;
      R w5[6], w6[6];
      MPC_UNROLL
      for (int i = 0; i < 6; i++) { w5[i] = Bp * (PM(i, 2) + PM(i, 4)) + PM(i, 5); w6[i] = dt * PM(i, 3); }
      if (tk == 0) {
        /* the feed-forward of u_0, and the feedback on the only components of ds_0 that can be non-zero: psi_0, v_0 */
        R o5[7], o6[7];
        MPC_GT(w5, R(0.0), o5);
        MPC_GT(w6, R(0.0), o6);
        const R Rdd = o5[5] + Sgd + dw;
        const R Rda = o6[5];
        const R Raa = o6[6] + Sga + dw;
        const R det = Rdd * Raa - Rda * Rda;
        if (!(Rdd > R(0.0)) || !(det > R(0.0))) ok_k = false;
        const R idet = frcp1(det);
        const R i11 = Raa * idet, i12 = -Rda * idet, i22 = Rdd * idet;
        const R Sd2 = o5[2], Sd3 = o5[3] - (lp + le) * dtLf, Sa2 = o6[2], Sa3 = o6[3];   /* S~ columns of psi_0, v_0 (with the Lagrangian Hessian's d-v term) */
        R gn[GAIN_SZ] = {};
        gn[2] = -(i11 * Sd2 + i12 * Sa2); gn[3] = -(i11 * Sd3 + i12 * Sa3);
        gn[6 + 2] = -(i12 * Sd2 + i22 * Sa2); gn[6 + 3] = -(i12 * Sd3 + i22 * Sa3);
        gn[GK_N] = -(Raa * rt_d - Rda * rt_a) * idet; gn[GK_N + 1] = -(-Rda * rt_d + Rdd * rt_a) * idet;
        if (wlane == 0) ws.template store_run<F_GK, GAIN_SZ>(0, J, gn);
        ok_all = wave_bcast_flag<WAVE>(ok_k, 0, wbase);
        break;
      }
      /* ---- W = P G (columns for inputs x,y,psi,v,e,delta,a) and Mx = G^T W ---- */
      R Mx[7][7];
      {
        R w[6], o[7];
        MPC_UNROLL
        for (int i = 0; i < 6; i++) w[i] = PM(i, 0) + Aex * PM(i, 4);
        MPC_GT(w, Pcc * Acx, o);
        MPC_UNROLL
        for (int i = 0; i < 7; i++) Mx[i][0] = o[i];
        MPC_UNROLL
        for (int i = 0; i < 6; i++) w[i] = PM(i, 1);
        MPC_GT(w, -Pcc, o);
        MPC_UNROLL
        for (int i = 0; i < 7; i++) Mx[i][1] = o[i];
        MPC_UNROLL
        for (int i = 0; i < 6; i++) w[i] = Axp * PM(i, 0) + Ayp * PM(i, 1) + PM(i, 2) + PM(i, 4);
        MPC_GT(w, R(0.0), o);
        MPC_UNROLL
        for (int i = 0; i < 7; i++) Mx[i][2] = o[i];
        MPC_UNROLL
        for (int i = 0; i < 6; i++) w[i] = Axv * PM(i, 0) + Ayv * PM(i, 1) + Apv * (PM(i, 2) + PM(i, 4)) + PM(i, 3);
        MPC_GT(w, Pcc * Acv, o);
        MPC_UNROLL
        for (int i = 0; i < 7; i++) Mx[i][3] = o[i];
        {
          /* input e only feeds cte+: column = G^T (0, Pcc*Ace) */
          const R wcs = Pcc * Ace;
          Mx[0][4] = Acx * wcs; Mx[1][4] = -wcs; Mx[2][4] = R(0.0); Mx[3][4] = Acv * wcs; Mx[4][4] = Ace * wcs;
          Mx[5][4] = R(0.0); Mx[6][4] = R(0.0);
        }
        MPC_GT(w5, R(0.0), o);
        MPC_UNROLL
        for (int i = 0; i < 7; i++) Mx[i][5] = o[i];
        MPC_GT(w6, R(0.0), o);
        MPC_UNROLL
        for (int i = 0; i < 7; i++) Mx[i][6] = o[i];
      }
      /* ---- add the Lagrangian Hessian of stage k: -lam_{k+1}^T d2F ---- */
      Mx[0][0] += -lc * fpp + le * h3;
      Mx[2][2] += (lx * cp + ly * sp) * vdt;
      Mx[3][2] += (lx * sp - ly * cp) * dt;
      Mx[4][4] += lc * vdt * se;
      Mx[4][3] += -lc * dt * ce;
      Mx[5][3] += -(lp + le) * dtLf;
      /* control terms */
      const R Rdd = Mx[5][5] + Hdd + Sgd + dw;
      const R Rda = Mx[6][5];
      const R Raa = Mx[6][6] + Sga + dw;
      const R det = Rdd * Raa - Rda * Rda;
      if (!(Rdd > R(0.0)) || !(det > R(0.0))) ok_k = false;
      const R idet = frcp1(det);
      const R i11 = Raa * idet, i12 = -Rda * idet, i22 = Rdd * idet;
      /* S~ (2 x 6 over x,y,psi,v,e,d) */
      R Sd[6], Sa[6], Kd[6], Ka[6];
      MPC_UNROLL
      for (int j = 0; j < 5; j++) { Sd[j] = Mx[5][j]; Sa[j] = Mx[6][j]; }
      Sd[5] = -Hdd; Sa[5] = R(0.0);
      MPC_UNROLL
      for (int j = 0; j < 6; j++) {
        Kd[j] = -(i11 * Sd[j] + i12 * Sa[j]);
        Ka[j] = -(i12 * Sd[j] + i22 * Sa[j]);
      }
      const R kfd = -(i11 * rt_d + i12 * rt_a), kfa = -(i12 * rt_d + i22 * rt_a);
      R gn[GAIN_SZ] = {};
      MPC_UNROLL
      for (int j = 0; j < 6; j++) { gn[j] = Kd[j]; gn[6 + j] = Ka[j]; }
      gn[GK_N] = kfd; gn[GK_N + 1] = kfa;
      if (wlane == tk) ws.template store_run<F_GK, GAIN_SZ>(k, J, gn);
      MPC_UNROLL
      for (int i = 0; i < 6; i++) {
        MPC_UNROLL
        for (int j = 0; j <= i; j++) {
          R q = (i < 5) ? Mx[i][j] : ((j == 5) ? Hdd : R(0.0));
          q += Sd[i] * Kd[j] + Sa[i] * Ka[j];
          Pm[i][j] = q;
        }
        p[i] = ((i < 5) ? qt[i] : -Hdd * ddl) + Sd[i] * kfd + Sa[i] * kfa;
      }
      Pm[0][0] += hxy; Pm[1][1] += hxy; Pm[2][2] += Hpp + dw; Pm[3][3] += Hvv + dw; Pm[4][4] += Hee + dw;
      p[2] += gp; p[3] += gv; p[4] += ge;
      Pcc = Hcc + dw; pc = gc;
      /* lane tk's verdict and value function are the wave's */
      if (!wave_bcast_flag<WAVE>(ok_k, tk, wbase)) { ok_all = false; break; }
      MPC_UNROLL
      for (int i = 0; i < 6; i++) {
        MPC_UNROLL
        for (int j = 0; j <= i; j++) Pm[i][j] = wave_bcast<WAVE>(Pm[i][j], tk, wbase);
        p[i] = wave_bcast<WAVE>(p[i], tk, wbase);
      }
      Pcc = wave_bcast<WAVE>(Pcc, tk, wbase); pc = wave_bcast<WAVE>(pc, tk, wbase);
    }
#undef MPC_GT
#undef PM
#undef MX
    return ok_all;
  }

  MPC_HD bool backward(R dw) {
    if constexpr (WAVE) return backward_wave(dw);
    const int I = it(cur), J = it(1 - cur);         /* J: where the gains go */
    /* value function of (x,y,psi,v,e,d) [+ c] at stage k+1; only the lower triangle of the symmetric
     * matrices is ever written or read (PM/MX pick it), so the other half never occupies registers */
    R Pm[6][6], p[6], Pcc, pc;
#define PM(i, j) Pm[(i) >= (j) ? (i) : (j)][(i) >= (j) ? (j) : (i)]
#define MX(i, j) Mx[(i) >= (j) ? (i) : (j)][(i) >= (j) ? (j) : (i)]
    MPC_UNROLL
    for (int i = 0; i < 6; i++) {
      p[i] = 0;
      MPC_UNROLL
      for (int j = 0; j < 6; j++) Pm[i][j] = 0;
    }
    const R rsc = lsm ? R(0.0) : -R(1.0);             /* constraint right-hand side: -c, or 0 for the LS system */
    const R hxy = (lsm ? R(1.0) : R(0.0)) + dw;       /* x and y carry no cost: only the LS identity / regularisation */
    /* Staging: record j of the iterate (the fields of stage j) sits in buffer (M-1-j)&1.  Stage k needs
     * (u_k, lam_{k+1}, duals of u_k) from record k -- moved to registers one iteration earlier -- and
     * (s_k, delta_{k-1}, duals of s_k) from record k-1; record k-2 is requested meanwhile. */
    ws.stage_drain();
    ws.stage_fetch_it(0, M - 1, I);
    ws.template stage_wait<0>();
    R sn[6];                                    /* s_{k+1} */
    MPC_UNROLL
    for (int i = 0; i < 6; i++) sn[i] = ws.sit(0, M - 1, I, F_S + i);
    {
      R Hpp, Hvv, Hee, Hcc, gp, gv, ge, gc;
      state_terms(sn[2], sn[3], sn[4], sn[5], ws.sit(0, M - 1, I, F_ZL + 0), ws.sit(0, M - 1, I, F_ZU + 0),
                  ws.sit(0, M - 1, I, F_ZL + 1), ws.sit(0, M - 1, I, F_ZU + 1), Hpp, Hvv, Hee, Hcc, gp, gv, ge, gc);
      Pm[0][0] = hxy; Pm[1][1] = hxy; Pm[2][2] = Hpp + dw; Pm[3][3] = Hvv + dw; Pm[4][4] = Hee + dw;
      Pcc = Hcc + dw; p[2] = gp; p[3] = gv; p[4] = ge; pc = gc;
    }
    /* inputs of stage k that live in record k, carried in registers */
    R delta = ws.sit(0, M - 1, I, F_U + 0), acc = ws.sit(0, M - 1, I, F_U + 1);
    R lx = ws.sit(0, M - 1, I, F_LAM + 0), ly = ws.sit(0, M - 1, I, F_LAM + 1), lp = ws.sit(0, M - 1, I, F_LAM + 2);
    R lc = ws.sit(0, M - 1, I, F_LAM + 4), le = ws.sit(0, M - 1, I, F_LAM + 5);
    R zld = ws.sit(0, M - 1, I, F_ZL + 2), zud = ws.sit(0, M - 1, I, F_ZU + 2);
    R zla = ws.sit(0, M - 1, I, F_ZL + 3), zua = ws.sit(0, M - 1, I, F_ZU + 3);
    if (M >= 2) ws.stage_fetch_it(1, M - 2, I);
    MPC_STAGE_LOOP
    for (int k = M - 1; k >= 0; --k) {
      /* ---- inputs of stage k ---- */
      R sk[6];
      R zlp = 0, zup = 0, zlv = 0, zuv = 0, delprev = 0;
      const int bk = (M - k) & 1;                    /* buffer of record k-1 */
      if (k > 0) {
        if (k >= 2) {
          ws.stage_fetch_it(bk ^ 1, k - 2, I);
          /* record k-1 must have landed; the gains stored by stage k+1 and the new request may stay in flight */
          if (k == M - 1) ws.template stage_wait<STG_IT_OPS>();
          else ws.template stage_wait<STG_IT_OPS + ST_BACKWARD>();
        } else ws.template stage_wait<0>();
        MPC_UNROLL
        for (int i = 0; i < 6; i++) sk[i] = ws.sit(bk, k - 1, I, F_S + i);
        zlp = ws.sit(bk, k - 1, I, F_ZL + 0); zup = ws.sit(bk, k - 1, I, F_ZU + 0);
        zlv = ws.sit(bk, k - 1, I, F_ZL + 1); zuv = ws.sit(bk, k - 1, I, F_ZU + 1);
        delprev = ws.sit(bk, k - 1, I, F_U + 0);
      } else {
        MPC_UNROLL
        for (int i = 0; i < 6; i++) sk[i] = st[i];
#if MPC_S0_VARIABLE
        sk[2] = p0; sk[3] = v0k;
#endif
      }
      const R v = sk[3];
      LinR L;
      linearise(sk, delta, acc, sn, L);
      const R sp = L.sp, cp = L.cp, se = L.se, ce = L.ce, fp = L.fp, g1 = L.g1, h3 = L.h3, fpp = L.fpp;
      const R r0 = rsc * L.c[0], r1 = rsc * L.c[1], r2 = rsc * L.c[2], r3 = rsc * L.c[3], rc = rsc * L.c[4], r4 = rsc * L.c[5];
      const R vdt = v * dt;
      const R Axp = -vdt * sp, Axv = dt * cp, Ayp = vdt * cp, Ayv = dt * sp, Apv = delta * dtLf;
      const R Acx = fp, Acv = dt * se, Ace = vdt * ce, Aex = -g1, Bp = v * dtLf;
      /* t = p + P r (the d component of r is zero) */
      R t[6];
      MPC_UNROLL
      for (int i = 0; i < 6; i++)
        t[i] = p[i] + PM(i, 0) * r0 + PM(i, 1) * r1 + PM(i, 2) * r2 + PM(i, 3) * r3 + PM(i, 4) * r4;
      const R tc = pc + Pcc * rc;
      /* G^T applied to a (6-vector, c-scalar): outputs for inputs x,y,psi,v,e,delta,a */
#define MPC_GT(w, wcs, o)                                                         \
  do {                                                                            \
    const R w24_ = (w)[2] + (w)[4];                                          \
    (o)[0] = (w)[0] + Aex * (w)[4] + Acx * (wcs);                                 \
    (o)[1] = (w)[1] - (wcs);                                                      \
    (o)[2] = Axp * (w)[0] + Ayp * (w)[1] + w24_;                                  \
    (o)[3] = Axv * (w)[0] + Ayv * (w)[1] + Apv * w24_ + (w)[3] + Acv * (wcs);     \
    (o)[4] = Ace * (wcs);                                                         \
    (o)[5] = Bp * w24_ + (w)[5];                                                  \
    (o)[6] = dt * (w)[3];                                                         \
  } while (0)
      R qt[7];
      MPC_GT(t, tc, qt);
      /* the stage's own control terms */
      const R isld = frcp1(delta - dl), isud = frcp1(du - delta), isla = frcp1(acc - al), isua = frcp1(au - acc);
      R ddl = 0, Hdd = 0;
      if (k >= 1 && !lsm) { ddl = delta - delprev; Hdd = df * R(2.0) * wdd; }   /* LS start: all delta are 0 */
      const R mub = lsm ? R(0.0) : mu;
      const R gdel = df * R(2.0) * wd * delta + Hdd * ddl + mub * (isud - isld);
      const R gacc = mub * (isua - isla);
      const R rt_d = qt[5] + gdel, rt_a = qt[6] + gacc;
      /* control Hessian diagonal: cost + barrier, or the identity of the LS system */
      const R Sgd = lsm ? R(1.0) : df * R(2.0) * wd + zld * isld + zud * isud, Sga = lsm ? R(1.0) : zla * isla + zua * isua;
      R w5[6], w6[6];
      MPC_UNROLL
      for (int i = 0; i < 6; i++) { w5[i] = Bp * (PM(i, 2) + PM(i, 4)) + PM(i, 5); w6[i] = dt * PM(i, 3); }
      if (k == 0) {
        /* the feed-forward of u_0, and the feedback on the only components of ds_0 that can be non-zero: psi_0, v_0 */
        R o5[7], o6[7];
        MPC_GT(w5, R(0.0), o5);
        MPC_GT(w6, R(0.0), o6);
        const R Rdd = o5[5] + Sgd + dw;
        const R Rda = o6[5];
        const R Raa = o6[6] + Sga + dw;
        const R det = Rdd * Raa - Rda * Rda;
        if (!(Rdd > R(0.0)) || !(det > R(0.0))) return false;
        const R idet = frcp1(det);
        const R i11 = Raa * idet, i12 = -Rda * idet, i22 = Rdd * idet;
        const R Sd2 = o5[2], Sd3 = o5[3] - (lp + le) * dtLf, Sa2 = o6[2], Sa3 = o6[3];   /* S~ columns of psi_0, v_0 (with the Lagrangian Hessian's d-v term) */
        R gn[GAIN_SZ] = {};
        gn[2] = -(i11 * Sd2 + i12 * Sa2); gn[3] = -(i11 * Sd3 + i12 * Sa3);
        gn[6 + 2] = -(i12 * Sd2 + i22 * Sa2); gn[6 + 3] = -(i12 * Sd3 + i22 * Sa3);
        gn[GK_N] = -(Raa * rt_d - Rda * rt_a) * idet; gn[GK_N + 1] = -(-Rda * rt_d + Rdd * rt_a) * idet;
        ws.template store_run<F_GK, GAIN_SZ>(0, J, gn);
        break;
      }
      /* ---- W = P G (columns for inputs x,y,psi,v,e,delta,a) and Mx = G^T W ---- */
      R Mx[7][7];
      {
        R w[6], o[7];
        MPC_UNROLL
        for (int i = 0; i < 6; i++) w[i] = PM(i, 0) + Aex * PM(i, 4);
        MPC_GT(w, Pcc * Acx, o);
        MPC_UNROLL
        for (int i = 0; i < 7; i++) Mx[i][0] = o[i];
        MPC_UNROLL
        for (int i = 0; i < 6; i++) w[i] = PM(i, 1);
        MPC_GT(w, -Pcc, o);
        MPC_UNROLL
        for (int i = 0; i < 7; i++) Mx[i][1] = o[i];
        MPC_UNROLL
        for (int i = 0; i < 6; i++) w[i] = Axp * PM(i, 0) + Ayp * PM(i, 1) + PM(i, 2) + PM(i, 4);
        MPC_GT(w, R(0.0), o);
        MPC_UNROLL
        for (int i = 0; i < 7; i++) Mx[i][2] = o[i];
        MPC_UNROLL
        for (int i = 0; i < 6; i++) w[i] = Axv * PM(i, 0) + Ayv * PM(i, 1) + Apv * (PM(i, 2) + PM(i, 4)) + PM(i, 3);
        MPC_GT(w, Pcc * Acv, o);
        MPC_UNROLL
        for (int i = 0; i < 7; i++) Mx[i][3] = o[i];
        {
          /* input e only feeds cte+: column = G^T (0, Pcc*Ace) */
          const R wcs = Pcc * Ace;
          Mx[0][4] = Acx * wcs; Mx[1][4] = -wcs; Mx[2][4] = R(0.0); Mx[3][4] = Acv * wcs; Mx[4][4] = Ace * wcs;
          Mx[5][4] = R(0.0); Mx[6][4] = R(0.0);
        }
        MPC_GT(w5, R(0.0), o);
        MPC_UNROLL
        for (int i = 0; i < 7; i++) Mx[i][5] = o[i];
        MPC_GT(w6, R(0.0), o);
        MPC_UNROLL
        for (int i = 0; i < 7; i++) Mx[i][6] = o[i];
      }
      /* ---- add the Lagrangian Hessian of stage k: -lam_{k+1}^T d2F ---- */
      Mx[0][0] += -lc * fpp + le * h3;
      Mx[2][2] += (lx * cp + ly * sp) * vdt;
      Mx[3][2] += (lx * sp - ly * cp) * dt;
      Mx[4][4] += lc * vdt * se;
      Mx[4][3] += -lc * dt * ce;
      Mx[5][3] += -(lp + le) * dtLf;
      /* control terms */
      const R Rdd = Mx[5][5] + Hdd + Sgd + dw;
      const R Rda = Mx[6][5];
      const R Raa = Mx[6][6] + Sga + dw;
      const R det = Rdd * Raa - Rda * Rda;
      if (!(Rdd > R(0.0)) || !(det > R(0.0))) return false;
      const R idet = frcp1(det);
      const R i11 = Raa * idet, i12 = -Rda * idet, i22 = Rdd * idet;
      /* S~ (2 x 6 over x,y,psi,v,e,d) */
      R Sd[6], Sa[6], Kd[6], Ka[6];
      MPC_UNROLL
      for (int j = 0; j < 5; j++) { Sd[j] = Mx[5][j]; Sa[j] = Mx[6][j]; }
      Sd[5] = -Hdd; Sa[5] = R(0.0);
      MPC_UNROLL
      for (int j = 0; j < 6; j++) {
        Kd[j] = -(i11 * Sd[j] + i12 * Sa[j]);
        Ka[j] = -(i12 * Sd[j] + i22 * Sa[j]);
      }
      const R kfd = -(i11 * rt_d + i12 * rt_a), kfa = -(i12 * rt_d + i22 * rt_a);
      R gn[GAIN_SZ] = {};
      MPC_UNROLL
      for (int j = 0; j < 6; j++) { gn[j] = Kd[j]; gn[6 + j] = Ka[j]; }
      gn[GK_N] = kfd; gn[GK_N + 1] = kfa;
      ws.template store_run<F_GK, GAIN_SZ>(k, J, gn);
      /* ---- value function of stage k ---- */
      R Hpp, Hvv, Hee, Hcc, gp, gv, ge, gc;
      state_terms(sk[2], v, sk[4], sk[5], zlp, zup, zlv, zuv, Hpp, Hvv, Hee, Hcc, gp, gv, ge, gc);
      MPC_UNROLL
      for (int i = 0; i < 6; i++) {
        MPC_UNROLL
        for (int j = 0; j <= i; j++) {
          R q = (i < 5) ? Mx[i][j] : ((j == 5) ? Hdd : R(0.0));
          q += Sd[i] * Kd[j] + Sa[i] * Ka[j];
          Pm[i][j] = q;
        }
        p[i] = ((i < 5) ? qt[i] : -Hdd * ddl) + Sd[i] * kfd + Sa[i] * kfa;
      }
      Pm[0][0] += hxy; Pm[1][1] += hxy; Pm[2][2] += Hpp + dw; Pm[3][3] += Hvv + dw; Pm[4][4] += Hee + dw;
      p[2] += gp; p[3] += gv; p[4] += ge;
      Pcc = Hcc + dw; pc = gc;
      MPC_UNROLL
      for (int i = 0; i < 6; i++) sn[i] = sk[i];
      /* inputs of stage k-1 that live in record k-1 (still staged in buffer bk): read them only now,
       * so that they do not occupy registers during the stage algebra */
      delta = delprev;
      acc = ws.sit(bk, k - 1, I, F_U + 1);
      lx = ws.sit(bk, k - 1, I, F_LAM + 0); ly = ws.sit(bk, k - 1, I, F_LAM + 1); lp = ws.sit(bk, k - 1, I, F_LAM + 2);
      lc = ws.sit(bk, k - 1, I, F_LAM + 4); le = ws.sit(bk, k - 1, I, F_LAM + 5);
      zld = ws.sit(bk, k - 1, I, F_ZL + 2); zud = ws.sit(bk, k - 1, I, F_ZU + 2);
      zla = ws.sit(bk, k - 1, I, F_ZL + 3); zua = ws.sit(bk, k - 1, I, F_ZU + 3);
    }
#undef MPC_GT
#undef PM
#undef MX
    return true;
  }

  /* ------------------------------------------------------------------ */
  /* forward sweep: Newton direction ds, du; step limits; dphi           */
  /* ------------------------------------------------------------------ */
  /* The forward sweep with one instance per wavefront: lane k prepares stage k -- model, gains, reciprocal slacks: what costs -- for
   * all stages at once; the recursion ds_k -> ds_{k+1} then runs through the lanes in order, each lane doing its stage's few
   * dozen multiply-adds on the state the lane before it broadcast (the statements of forward(), in their order: same bits). */
  MPC_HD void forward_wave() {
    const int I = it(cur), J = it(1 - cur);
    const bool mine = wlane < M;
    const int k = mine ? wlane : M - 1;             /* (the lanes behind the last stage repeat it; nothing of theirs is kept) */
    const R rsc = lsm ? R(0.0) : R(1.0);
    /* ---- this lane's stage: everything that does not depend on ds_k ---- */
    R sk[6], sn[6];
    load_state(k, I, sk);
    MPC_UNROLL
    for (int i = 0; i < 6; i++) sn[i] = ws.it(k, I, F_S + i);
    const R v = sk[3];
    const R delta = ws.it(k, I, F_U + 0), acc = ws.it(k, I, F_U + 1);
    LinR L;
    linearise(sk, delta, acc, sn, L);
    R gk[GK_N], gf0 = ws.it(k, J, F_GK + GK_N + 0), gf1 = ws.it(k, J, F_GK + GK_N + 1);
    MPC_UNROLL
    for (int j = 0; j < GK_N; j++) gk[j] = ws.it(k, J, F_GK + j);
    const R vdt = v * dt, Apv = delta * dtLf, Bp = v * dtLf;
    const R xs[4] = {sn[2], sn[3], delta, acc};
    const R lo[4] = {yl, vl, dl, al}, hi[4] = {yu, vu, du, au};
    R isl[4], isu[4], zl[4], zu[4], izl[4], izu[4];
    MPC_UNROLL
    for (int b = 0; b < 4; b++) {
      isl[b] = frcp1(xs[b] - lo[b]); isu[b] = frcp1(hi[b] - xs[b]);
      zl[b] = ws.it(k, I, F_ZL + b); zu[b] = ws.it(k, I, F_ZU + b);
      izl[b] = frcp1(zl[b]); izu[b] = frcp1(zu[b]);
    }
    const R delprev_k = k > 0 ? (R)ws.it(k - 1, I, F_U + 0) : R(0.0);
    const R xinf_k = mpc_max(mpc_max(mpc_abs(sn[0]), mpc_abs(sn[1])), mpc_max(mpc_abs(sn[3]), mpc_abs(sn[4])));
    /* ---- the start of the recursion (wave-uniform: forward()'s own preamble) ---- */
    R d0 = 0, d1 = 0, d2 = 0, d3 = 0, d5 = 0, ddprev = 0;
    R rmax = R(0.0), rzmax = R(0.0);
    dphi = R(0.0); dxinf = R(0.0); xinf = R(0.0);
#if MPC_S0_VARIABLE
    d2 = rsc * (st[2] - p0); d3 = rsc * (st[3] - v0k);
    if (d2 != R(0.0) || d3 != R(0.0)) {
      const R islp = frcp1(p0 - yl), isup = frcp1(yu - p0), islv = frcp1(v0k - vl), isuv = frcp1(vu - v0k);
      rmax = mpc_max(mpc_max(-d2 * islp, d2 * isup), mpc_max(-d3 * islv, d3 * isuv));
      dphi = mu * ((isup - islp) * d2 + (isuv - islv) * d3);
      dxinf = mpc_max(mpc_abs(d2), mpc_abs(d3));
    }
    const R s0d2 = d2, s0d3 = d3;
#endif
    R dn[D_N] = {0, 0, 0, 0, 0, 0, 0, 0};
    /* ---- the recursion: stage t is lane t's turn ---- */
    for (int t = 0; t < M; ++t) {
      R dd = gf0, da = gf1;
      dd += gk[0] * d0 + gk[1] * d1 + gk[2] * d2 + gk[3] * d3 + gk[4] * d5 + gk[5] * ddprev;
      da += gk[6] * d0 + gk[7] * d1 + gk[8] * d2 + gk[9] * d3 + gk[10] * d5 + gk[11] * ddprev;
      const R n0 = d0 - vdt * L.sp * d2 + dt * L.cp * d3 - rsc * L.c[0];
      const R n1 = d1 + vdt * L.cp * d2 + dt * L.sp * d3 - rsc * L.c[1];
      const R n2 = d2 + Apv * d3 + Bp * dd - rsc * L.c[2];
      const R n3 = d3 + dt * da - rsc * L.c[3];
      const R n4 = L.fp * d0 - d1 + dt * L.se * d3 + vdt * L.ce * d5 - rsc * L.c[4];
      const R n5 = -L.g1 * d0 + d2 + Apv * d3 + Bp * dd - rsc * L.c[5];
      const R dx[4] = {n2, n3, dd, da};
      R rm = rmax, rz = rzmax, dp = dphi;
      MPC_UNROLL
      for (int b = 0; b < 4; b++) {
        rm = mpc_max(rm, mpc_max(-dx[b] * isl[b], dx[b] * isu[b]));
        const R dzl = mu * isl[b] - zl[b] - zl[b] * isl[b] * dx[b];
        const R dzu = mu * isu[b] - zu[b] + zu[b] * isu[b] * dx[b];
        rz = mpc_max(rz, mpc_max(-dzl * izl[b], -dzu * izu[b]));
        dp += mu * (isu[b] - isl[b]) * dx[b];
      }
      R g = R(2.0) * wc * sn[4] * n4 + R(2.0) * we * sn[5] * n5 + R(2.0) * wv * (sn[3] - vref) * n3 + R(2.0) * wd * delta * dd;
      if (t > 0) g += R(2.0) * wdd * (delta - delprev_k) * (dd - ddprev);
      dp += df * g;
      const R dxs = mpc_max(dxinf, mpc_max(mpc_max(mpc_max(mpc_abs(n0), mpc_abs(n1)), mpc_max(mpc_abs(n2), mpc_abs(n3))),
                                        mpc_max(mpc_max(mpc_abs(n4), mpc_abs(n5)), mpc_max(mpc_abs(dd), mpc_abs(da)))));
      const R xis = mpc_max(xinf, xinf_k);
      if (wlane == t) { dn[0] = n0; dn[1] = n1; dn[2] = n2; dn[3] = n3; dn[4] = n4; dn[5] = n5; dn[6] = dd; dn[7] = da; }
      /* lane t's results are the state of the recursion */
      d0 = wave_bcast<WAVE>(n0, t, wbase); d1 = wave_bcast<WAVE>(n1, t, wbase); d2 = wave_bcast<WAVE>(n2, t, wbase); d3 = wave_bcast<WAVE>(n3, t, wbase); d5 = wave_bcast<WAVE>(n5, t, wbase);
      ddprev = wave_bcast<WAVE>(dd, t, wbase);
      rmax = wave_bcast<WAVE>(rm, t, wbase); rzmax = wave_bcast<WAVE>(rz, t, wbase); dphi = wave_bcast<WAVE>(dp, t, wbase); dxinf = wave_bcast<WAVE>(dxs, t, wbase); xinf = wave_bcast<WAVE>(xis, t, wbase);
    }
    if (mine) ws.template store_run<F_D, D_N>(k, 0, dn);
#if MPC_S0_VARIABLE
    if (s0_rows && !lsm) {
      const R xs0[2] = {p0, v0k}, lo0[2] = {yl, vl}, hi0[2] = {yu, vu}, dx0[2] = {s0d2, s0d3};
      MPC_UNROLL
      for (int b = 0; b < 2; b++) {
        const R isl0 = frcp1(xs0[b] - lo0[b]), isu0 = frcp1(hi0[b] - xs0[b]);
        const R zl0 = s0_z(I, b, 0), zu0 = s0_z(I, b, 1);
        const R dzl = mu * isl0 - zl0 - zl0 * isl0 * dx0[b];
        const R dzu = mu * isu0 - zu0 + zu0 * isu0 * dx0[b];
        rzmax = mpc_max(rzmax, mpc_max(-dzl * frcp1(zl0), -dzu * frcp1(zu0)));
      }
    }
#endif
    amax = (rmax > tau) ? tau / rmax : R(1.0);
    az = (rzmax > tau) ? tau / rzmax : R(1.0);
  }

  MPC_HD void forward() {
    if constexpr (WAVE) { forward_wave(); return; }
    const int I = it(cur), J = it(1 - cur);
    R d0 = 0, d1 = 0, d2 = 0, d3 = 0, d5 = 0; /* ds_k: x,y,psi,v,(c),e */
    R ddprev = 0, delprev = 0;                 /* d(delta_{k-1}), delta_{k-1} */
    R rmax = R(0.0), rzmax = R(0.0);                 /* largest step ratios: alpha = min(1, tau / ratio) */
    const R rsc = lsm ? R(0.0) : R(1.0);
    dphi = R(0.0); dxinf = R(0.0); xinf = R(0.0);
    R sk[6];
    load_state(0, I, sk);
#if MPC_S0_VARIABLE
    /* ds_0: the pinning rows are linear, the full step restores them (zero for an instance whose start was not pushed) */
    d2 = rsc * (st[2] - p0); d3 = rsc * (st[3] - v0k);
    if (d2 != R(0.0) || d3 != R(0.0)) {
      /* psi_0 and v_0 are bounded variables: their slacks limit the step and their barrier and cost terms are part of
       * the merit function's slope */
      const R islp = frcp1(p0 - yl), isup = frcp1(yu - p0), islv = frcp1(v0k - vl), isuv = frcp1(vu - v0k);
      rmax = mpc_max(mpc_max(-d2 * islp, d2 * isup), mpc_max(-d3 * islv, d3 * isuv));
      dphi = mu * ((isup - islp) * d2 + (isuv - islv) * d3);
      dxinf = mpc_max(mpc_abs(d2), mpc_abs(d3));
    }
    const R s0d2 = d2, s0d3 = d3;
#endif
    /* staging: stage k's iterate record and gains in buffer k&1, stage k+1 requested meanwhile */
    ws.stage_drain();
    ws.stage_fetch_itf(0, 0, I);
    ws.stage_fetch_g(0, 0, J);
#if MPC_S0_VARIABLE
    /* (asked for here, used behind the loop: the loads travel with the first record's) */
    R z0lp = R(1.0), z0up = R(1.0), z0lv = R(1.0), z0uv = R(1.0);
    if (s0_rows) { z0lp = s0_z(I, 0, 0); z0up = s0_z(I, 0, 1); z0lv = s0_z(I, 1, 0); z0uv = s0_z(I, 1, 1); }
#endif
    MPC_STAGE_LOOP
    for (int k = 0; k < M; ++k) {
      const int bf = k & 1;
      if (k + 1 < M) {
        ws.stage_fetch_itf(bf ^ 1, k + 1, I);
        ws.stage_fetch_g(bf ^ 1, k + 1, J);
        if (k == 0) ws.template stage_wait<STG_ITF_OPS + STG_X_OPS>();
        else ws.template stage_wait<STG_ITF_OPS + STG_X_OPS + ST_FORWARD>();
      } else ws.template stage_wait<0>();
      R sn[6];
      MPC_UNROLL
      for (int i = 0; i < 6; i++) sn[i] = ws.sit(bf, k, I, F_S + i);
      const R v = sk[3];
      const R delta = ws.sit(bf, k, I, F_U + 0), acc = ws.sit(bf, k, I, F_U + 1);
      LinR L;
      linearise(sk, delta, acc, sn, L);
      R dd = ws.sg(bf, k, J, GK_N + 0), da = ws.sg(bf, k, J, GK_N + 1);
      /* (stage 0: the record holds zeros except for the columns of psi_0, v_0, and ds_0 is zero elsewhere) */
      dd += ws.sg(bf, k, J, 0) * d0 + ws.sg(bf, k, J, 1) * d1 + ws.sg(bf, k, J, 2) * d2 + ws.sg(bf, k, J, 3) * d3 +
            ws.sg(bf, k, J, 4) * d5 + ws.sg(bf, k, J, 5) * ddprev;
      da += ws.sg(bf, k, J, 6) * d0 + ws.sg(bf, k, J, 7) * d1 + ws.sg(bf, k, J, 8) * d2 + ws.sg(bf, k, J, 9) * d3 +
            ws.sg(bf, k, J, 10) * d5 + ws.sg(bf, k, J, 11) * ddprev;
      const R vdt = v * dt, Apv = delta * dtLf, Bp = v * dtLf;
      const R n0 = d0 - vdt * L.sp * d2 + dt * L.cp * d3 - rsc * L.c[0];
      const R n1 = d1 + vdt * L.cp * d2 + dt * L.sp * d3 - rsc * L.c[1];
      const R n2 = d2 + Apv * d3 + Bp * dd - rsc * L.c[2];
      const R n3 = d3 + dt * da - rsc * L.c[3];
      const R n4 = L.fp * d0 - d1 + dt * L.se * d3 + vdt * L.ce * d5 - rsc * L.c[4];
      const R n5 = -L.g1 * d0 + d2 + Apv * d3 + Bp * dd - rsc * L.c[5];
      {
        const R dn[D_N] = {n0, n1, n2, n3, n4, n5, dd, da};
        ws.template store_run<F_D, D_N>(k, 0, dn);
      }
      const R q2 = n2, q3 = n3, q4 = n4, q5 = n5, qd = dd, qa = da;
      /* bounded variables of this stage: psi_{k+1}, v_{k+1}, delta_k, a_k */
      const R xs[4] = {sn[2], sn[3], delta, acc};
      const R lo[4] = {yl, vl, dl, al}, hi[4] = {yu, vu, du, au};
      const R dx[4] = {q2, q3, qd, qa};
      MPC_UNROLL
      for (int b = 0; b < 4; b++) {
        const R isl = frcp1(xs[b] - lo[b]), isu = frcp1(hi[b] - xs[b]);
        const R zl = ws.sit(bf, k, I, F_ZL + b), zu = ws.sit(bf, k, I, F_ZU + b);
        rmax = mpc_max(rmax, mpc_max(-dx[b] * isl, dx[b] * isu));
        const R dzl = mu * isl - zl - zl * isl * dx[b];
        const R dzu = mu * isu - zu + zu * isu * dx[b];
        rzmax = mpc_max(rzmax, mpc_max(-dzl * frcp1(zl), -dzu * frcp1(zu)));
        dphi += mu * (isu - isl) * dx[b];
      }
      /* objective part of the directional derivative */
      R g = R(2.0) * wc * sn[4] * q4 + R(2.0) * we * sn[5] * q5 + R(2.0) * wv * (sn[3] - vref) * q3 + R(2.0) * wd * delta * qd;
      if (k > 0) g += R(2.0) * wdd * (delta - delprev) * (qd - ddprev);
      dphi += df * g;
      dxinf = mpc_max(dxinf, mpc_max(mpc_max(mpc_max(mpc_abs(n0), mpc_abs(n1)), mpc_max(mpc_abs(n2), mpc_abs(n3))),
                               mpc_max(mpc_max(mpc_abs(n4), mpc_abs(n5)), mpc_max(mpc_abs(dd), mpc_abs(da)))));
      xinf = mpc_max(xinf, mpc_max(mpc_max(mpc_abs(sn[0]), mpc_abs(sn[1])), mpc_max(mpc_abs(sn[3]), mpc_abs(sn[4]))));
      d0 = n0; d1 = n1; d2 = n2; d3 = n3; d5 = n5; ddprev = dd; delprev = delta;
      MPC_UNROLL
      for (int i = 0; i < 6; i++) sk[i] = sn[i];
    }
    #if MPC_S0_VARIABLE
        if (s0_rows && !lsm) {
          /* the bound duals of psi_0, v_0 take part in the fraction-to-the-boundary rule of the duals */
          const R xs0[2] = {p0, v0k}, lo0[2] = {yl, vl}, hi0[2] = {yu, vu}, dx0[2] = {s0d2, s0d3};
          const R zl0[2] = {z0lp, z0lv}, zu0[2] = {z0up, z0uv};
          MPC_UNROLL
          for (int b = 0; b < 2; b++) {
            const R isl = frcp1(xs0[b] - lo0[b]), isu = frcp1(hi0[b] - xs0[b]);
            const R dzl = mu * isl - zl0[b] - zl0[b] * isl * dx0[b];
            const R dzu = mu * isu - zu0[b] + zu0[b] * isu * dx0[b];
            rzmax = mpc_max(rzmax, mpc_max(-dzl * frcp1(zl0[b]), -dzu * frcp1(zu0[b])));
          }
        }
    #endif
        /* fraction to the boundary, W&B eq. (15): alpha = min(1, tau / max ratio) */
        amax = (rmax > tau) ? tau / rmax : R(1.0);
    az = (rzmax > tau) ? tau / rzmax : R(1.0);
  }

  /* ------------------------------------------------------------------ */
  /* costate + trial point in ONE descending sweep.                       */
  /*  - costate: the state rows of the Newton system solved for the       */
  /*    full-step multipliers lam+_k (backward recursion), dlam = lam+ - lam */
  /*  - trial: iterate(cur) + alpha * direction -> slot 1-cur, with       */
  /*    residuals, objective, barrier and the optimality-error pieces.    */
  /* Every trial quantity is local to a stage (or to two neighbouring     */
  /* ones), so the trial point can be evaluated in the costate's order;   */
  /* dlam never goes to memory, and a backtracking trial simply repeats   */
  /* the (cheap) costate arithmetic.  alpha scales (ds,du), alpha_l scales */
  /* dlam, alpha_z the bound duals; with_costate = false evaluates the    */
  /* point as it stands (start point).  lmax returns max |dlam|.          */
  /* Step k (M..0) works on record k-1 = (s_k, u_{k-1}, lam_k, duals) and */
  /* on transition k = (s_k, u_k) -> s_{k+1} whose other inputs are       */
  /* carried in registers from step k+1.                                 */
  /* ------------------------------------------------------------------ */
  MPC_HD EvalR costate_trial(R dw, R alpha, R alpha_l, R alpha_z, bool with_costate, R &lmax) {
    const int I = it(cur), J = it(1 - cur);
    const R hxy = (lsm ? R(1.0) : R(0.0)) + dw;
    EvalR Ev;
    Ev.theta = 0; Ev.cinf = 0; Ev.f = 0; Ev.L = 0; Ev.dinf = 0; Ev.cmin = IC::huge; Ev.cmax = 0; Ev.lsum = 0; Ev.zsum = 0; Ev.du0 = 0; Ev.ok = true;
    lmax = R(0.0);
    const R ksm = IC::kappa_sigma * mu, ksi = mu * (R(1.0) / IC::kappa_sigma);
    /* carried from step k+1 -- current iterate: s_{k+1}, u_k, lam_{k+1}, d(delta_k), lam+_{k+1} */
    R sn_o[6] = {0, 0, 0, 0, 0, 0}, del_o = 0, acc_o = 0, lx = 0, ly = 0, lp = 0, lc = 0, le = 0, ddk = 0;
    R L0 = 0, L1 = 0, L2 = 0, L3 = 0, L4 = 0, L5 = 0;
    /* -- trial point: s_{k+1}, lam_{k+1}, u_k, duals of u_k, delta_{k+1} */
    R sn_t[6] = {0, 0, 0, 0, 0, 0}, ln_t[6] = {0, 0, 0, 0, 0, 0}, del_t = 0, acc_t = 0, del_nx = 0;
    R zdl_t = 0, zdu_t = 0, zal_t = 0, zau_t = 0;
    /* staging: record j (iterate + direction of stage j) in buffer (M-1-j)&1 */
    ws.stage_drain();
    ws.stage_fetch_it(0, M - 1, I);
    ws.stage_fetch_d(0, M - 1);
#if MPC_S0_VARIABLE
    /* the record of the initial state's own rows (used by the last step): asked for here, so that it travels with the first stage record */
    R c_l0 = 0, c_l1 = 0, c_l2 = 0, c_l3 = 0, c_l4 = 0, c_l5 = 0, c_zlp = 0, c_zup = 0, c_zlv = 0, c_zuv = 0;
    if (s0_rows) {
      c_l0 = s0_lam(I, 0); c_l1 = s0_lam(I, 1); c_l2 = s0_lam(I, 2); c_l3 = s0_lam(I, 3); c_l4 = s0_lam(I, 4); c_l5 = s0_lam(I, 5);
      c_zlp = s0_z(I, 0, 0); c_zup = s0_z(I, 0, 1); c_zlv = s0_z(I, 1, 0); c_zuv = s0_z(I, 1, 1);
    }
#endif
    /* the step of a stage as a function of whether it is the last one (k = 0: the initial state's own rows, no record to
     * fetch): the loop runs the general form, the epilogue the other, and neither carries the other's registers */
    auto step_k = [&](auto kz_, const int k) {
      constexpr bool KZ = decltype(kz_)::value;
        const int bk = (M - k) & 1;                    /* buffer of record k-1 */
        R s_o[6], s_t[6], lam_t[6] = {0, 0, 0, 0, 0, 0};
        R zs0 = 0, zs1 = 0, zs2 = 0, zs3 = 0;     /* trial duals of psi_k, v_k: zl_psi, zu_psi, zl_v, zu_v */
        R n_del_o = 0, n_acc_o = 0, n_ddk = 0, lo0 = 0, lo1 = 0, lo2 = 0, lo4 = 0, lo5 = 0;
        R n_del_t = 0, n_acc_t = 0, n_zdl = 0, n_zdu = 0, n_zal = 0, n_zau = 0;
        if constexpr (!KZ) {
          if (k >= 2) {
            ws.stage_fetch_it(bk ^ 1, k - 2, I);
            ws.stage_fetch_d(bk ^ 1, k - 2);
            if (k == M) ws.template stage_wait<STG_IT_OPS + STG_D_OPS>();
            else ws.template stage_wait<STG_IT_OPS + STG_D_OPS + ST_TRIAL>();
          } else ws.template stage_wait<0>();
          const int r = k - 1;
          R ds[6];
          MPC_UNROLL
          for (int i = 0; i < 6; i++) { s_o[i] = ws.sit(bk, r, I, F_S + i); ds[i] = ws.sx(bk, r, F_D, D_S + i); }
          const R lo3 = ws.sit(bk, r, I, F_LAM + 3);
          lo0 = ws.sit(bk, r, I, F_LAM + 0); lo1 = ws.sit(bk, r, I, F_LAM + 1); lo2 = ws.sit(bk, r, I, F_LAM + 2);
          lo4 = ws.sit(bk, r, I, F_LAM + 4); lo5 = ws.sit(bk, r, I, F_LAM + 5);
          n_del_o = ws.sit(bk, r, I, F_U + 0); n_acc_o = ws.sit(bk, r, I, F_U + 1);
          const R ddel = ws.sx(bk, r, F_D, D_U + 0), dacc = ws.sx(bk, r, F_D, D_U + 1);
          n_ddk = ddel;
          if (r == 0) Ev.du0 = mpc_max(mpc_abs(ddel), mpc_abs(dacc));   /* the outputs' part of the step (termination polish) */
          /* ---- costate: lam+_k ---- */
          R dl0 = 0, dl1 = 0, dl2 = 0, dl3 = 0, dl4 = 0, dl5 = 0;
          if (with_costate) {
            R Hpp, Hvv, Hee, Hcc, gp, gv, ge, gc;
            state_terms(s_o[2], s_o[3], s_o[4], s_o[5], ws.sit(bk, r, I, F_ZL + 0), ws.sit(bk, r, I, F_ZU + 0),
                        ws.sit(bk, r, I, F_ZL + 1), ws.sit(bk, r, I, F_ZU + 1), Hpp, Hvv, Hee, Hcc, gp, gv, ge, gc);
            R n0, n1, n2, n3, n4, n5;
            if (k == M) {
              n0 = -(hxy * ds[0]);
              n1 = -(hxy * ds[1]);
              n2 = -(gp + (Hpp + dw) * ds[2]);
              n3 = -(gv + (Hvv + dw) * ds[3]);
              n4 = -(gc + (Hcc + dw) * ds[4]);
              n5 = -(ge + (Hee + dw) * ds[5]);
            } else {
              const R v = s_o[3];
              LinR L;
              linearise(s_o, del_o, acc_o, sn_o, L);   /* the residual part is unused here and is eliminated */
              const R sp = L.sp, cp = L.cp, se = L.se, ce = L.ce, fp = L.fp, g1 = L.g1, h3 = L.h3, fpp = L.fpp;
              const R vdt = v * dt, Apv = del_o * dtLf;
              /* curvature of stage k */
              const R Hxx = -lc * fpp + le * h3, Hpsi2 = (lx * cp + ly * sp) * vdt, Hpv = (lx * sp - ly * cp) * dt;
              const R Hee2 = lc * vdt * se, Hev = -lc * dt * ce, Hvd = -(lp + le) * dtLf;
              const R L25 = L2 + L5;
              n0 = L0 + fp * L4 - g1 * L5 - (Hxx + hxy) * ds[0];
              n1 = L1 - L4 - hxy * ds[1];
              n2 = -vdt * sp * L0 + vdt * cp * L1 + L25 - gp - (Hpp + dw + Hpsi2) * ds[2] - Hpv * ds[3];
              n3 = dt * cp * L0 + dt * sp * L1 + Apv * L25 + L3 + dt * se * L4 - gv - Hpv * ds[2] -
                   (Hvv + dw) * ds[3] - Hev * ds[5] - Hvd * ddk;
              n4 = -gc - (Hcc + dw) * ds[4];
              n5 = vdt * ce * L4 - ge - (Hee + dw + Hee2) * ds[5] - Hev * ds[3];
            }
            L0 = n0; L1 = n1; L2 = n2; L3 = n3; L4 = n4; L5 = n5;
            dl0 = L0 - lo0; dl1 = L1 - lo1; dl2 = L2 - lo2; dl3 = L3 - lo3; dl4 = L4 - lo4; dl5 = L5 - lo5;
            lmax = mpc_max(lmax, mpc_max(mpc_max(mpc_max(mpc_abs(dl0), mpc_abs(dl1)), mpc_max(mpc_abs(dl2), mpc_abs(dl3))), mpc_max(mpc_abs(dl4), mpc_abs(dl5))));
          }
          /* ---- trial: record k-1 ---- */
          lam_t[0] = lo0 + alpha_l * dl0; lam_t[1] = lo1 + alpha_l * dl1; lam_t[2] = lo2 + alpha_l * dl2;
          lam_t[3] = lo3 + alpha_l * dl3; lam_t[4] = lo4 + alpha_l * dl4; lam_t[5] = lo5 + alpha_l * dl5;
          MPC_UNROLL
          for (int i = 0; i < 6; i++) s_t[i] = s_o[i] + alpha * ds[i];
          n_del_t = n_del_o + alpha * ddel;
          n_acc_t = n_acc_o + alpha * dacc;
          R rec[IT_SZ] = {};                           /* the trial record, stored group by group below */
          MPC_UNROLL
          for (int i = 0; i < 6; i += 2) {
            rec[F_S + i] = s_t[i]; rec[F_S + i + 1] = s_t[i + 1];
            rec[F_LAM + i] = lam_t[i]; rec[F_LAM + i + 1] = lam_t[i + 1];
            Ev.lsum += mpc_abs(lam_t[i]) + mpc_abs(lam_t[i + 1]);
          }
          rec[F_U] = n_del_t; rec[F_U + 1] = n_acc_t;
          ws.template store_run<F_S, F_ZL - F_S>(r, J, rec + F_S);      /* s, u, lam (+ padding) */
          /* duals of psi_k, v_k, delta_{k-1}, a_{k-1} */
          const R xo[4] = {s_o[2], s_o[3], n_del_o, n_acc_o};
          const R xn[4] = {s_t[2], s_t[3], n_del_t, n_acc_t};
          const R dxb[4] = {ds[2], ds[3], ddel, dacc};
          const R lo[4] = {yl, vl, dl, al}, hi[4] = {yu, vu, du, au};
          R zln[4], zun[4], prod = R(1.0);
          MPC_UNROLL
          for (int b = 0; b < 4; b++) {
            const R islo = frcp1(xo[b] - lo[b]), isuo = frcp1(hi[b] - xo[b]);
            const R zl = ws.sit(bk, r, I, F_ZL + b), zu = ws.sit(bk, r, I, F_ZU + b);
            const R dzl = mu * islo - zl - zl * islo * dxb[b];
            const R dzu = mu * isuo - zu + zu * isuo * dxb[b];
            const R sl = xn[b] - lo[b], su = hi[b] - xn[b];
            if (!(sl > R(0.0)) || !(su > R(0.0))) Ev.ok = false;
            const R isl = frcp1(sl), isu = frcp1(su);
            R a = zl + alpha_z * dzl, c = zu + alpha_z * dzu;
            /* kappa_sigma safeguard, W&B eq. (16) */
            a = mpc_max(mpc_min(a, ksm * isl), ksi * isl);
            c = mpc_max(mpc_min(c, ksm * isu), ksi * isu);
            zln[b] = a; zun[b] = c;
            Ev.zsum += a + c;
            const R pl = sl * a, pu = su * c;
            Ev.cmin = mpc_min(Ev.cmin, mpc_min(pl, pu)); Ev.cmax = mpc_max(Ev.cmax, mpc_max(pl, pu));
            prod *= sl * su;
          }
          MPC_UNROLL
          for (int b = 0; b < 4; b++) { rec[F_ZL + b] = zln[b]; rec[F_ZU + b] = zun[b]; }
          ws.template store_run<F_ZL, 8>(r, J, rec + F_ZL);
          Ev.L += flog(prod);
          zs0 = zln[0]; zs1 = zun[0]; zs2 = zln[1]; zs3 = zun[1];
          n_zdl = zln[2]; n_zdu = zun[2]; n_zal = zln[3]; n_zau = zun[3];
          /* objective terms of (s_k, u_{k-1}) */
          const R dv = s_t[3] - vref;
          Ev.f += wc * s_t[4] * s_t[4] + we * s_t[5] * s_t[5] + wv * dv * dv + wd * n_del_t * n_del_t;
        } else {
          MPC_UNROLL
          for (int i = 0; i < 6; i++) { s_o[i] = st[i]; s_t[i] = st[i]; }
#if MPC_S0_VARIABLE
          s_o[2] = p0; s_o[3] = v0k;
          s_t[2] = trial_x0(p0, st[2], alpha); s_t[3] = trial_x0(v0k, st[3], alpha);
          const R c2 = s_t[2] - st[2], c3 = s_t[3] - st[3];    /* residuals of the pinning rows of psi_0, v_0 */
          if (c2 != R(0.0) || c3 != R(0.0)) {
            Ev.theta += mpc_abs(c2) + mpc_abs(c3); Ev.cinf = mpc_max(Ev.cinf, mpc_max(mpc_abs(c2), mpc_abs(c3)));
            const R slp = s_t[2] - yl, sup = yu - s_t[2], slv = s_t[3] - vl, suv = vu - s_t[3];
            if (!(slp > R(0.0)) || !(sup > R(0.0)) || !(slv > R(0.0)) || !(suv > R(0.0))) Ev.ok = false;
            /* barrier terms of (psi_0, v_0) RELATIVE to their values at the pinned point, which are constants of the merit
             * function like the stage-0 cost.  (The cost's own dependence on v_0, w_v (v_0 - vref_0)^2, is left a constant:
             * a pushed v_0 sits at Config::maxSpeed, where vref_0 is, so its slope there is ~0.) */
            Ev.L += flog((slp * sup * slv * suv) * frcp((st[2] - yl) * (yu - st[2]) * (st[3] - vl) * (vu - st[3])));
          }
          /* ---- the record of stage "-1": lam_0 (the multipliers of the pinning rows) and the bound duals of psi_0, v_0 ---- */
          if (s0_rows) {
            const R rsc0 = lsm ? R(0.0) : R(1.0);
            const R ds2 = rsc0 * (st[2] - p0), ds3 = rsc0 * (st[3] - v0k);        /* ds_0 (zero unless a start value was pushed) */
            R dl0 = 0, dl1 = 0, dl2 = 0, dl3 = 0, dl4 = 0, dl5 = 0;
            if (with_costate) {
              /* the state rows of stage 0, solved for lam+_0 exactly as the rows of the stages above are (k < M there) */
              const R islp = frcp1(p0 - yl), isup = frcp1(yu - p0), islv = frcp1(v0k - vl), isuv = frcp1(vu - v0k);
              const R mub = lsm ? R(0.0) : mu;
              const R Hpp = lsm ? R(1.0) : c_zlp * islp + c_zup * isup;
              const R Hvv = lsm ? R(1.0) : df * R(2.0) * (wv + wneg0) + c_zlv * islv + c_zuv * isuv;
                const R gp = mub * (isup - islp);
              const R gv = df * R(2.0) * (wv * (v0k - vref0) + wneg0 * v0k) + mub * (isuv - islv);
              const R ge = df * R(2.0) * we0 * s_o[5], gc = df * R(2.0) * wc0 * s_o[4];
              const R v = s_o[3];
              LinR L;
              linearise(s_o, del_o, acc_o, sn_o, L);
              const R sp = L.sp, cp = L.cp, se = L.se, ce = L.ce, fp = L.fp, g1 = L.g1;
              const R vdt = v * dt, Apv = del_o * dtLf;
              const R Hpsi2 = (lx * cp + ly * sp) * vdt, Hpv = (lx * sp - ly * cp) * dt;
              const R Hev = -lc * dt * ce, Hvd = -(lp + le) * dtLf;      /* (ds_0 has no x, y, cte, epsi part: their curvature terms drop out) */
              const R L25 = L2 + L5;
              const R n0 = L0 + fp * L4 - g1 * L5;
              const R n1 = L1 - L4;
              const R n2 = -vdt * sp * L0 + vdt * cp * L1 + L25 - gp - (Hpp + dw + Hpsi2) * ds2 - Hpv * ds3;
              const R n3 = dt * cp * L0 + dt * sp * L1 + Apv * L25 + L3 + dt * se * L4 - gv - Hpv * ds2 -
                           (Hvv + dw) * ds3 - Hvd * ddk;
              const R n4 = -gc;
              const R n5 = vdt * ce * L4 - ge - Hev * ds3;
                dl0 = n0 - c_l0; dl1 = n1 - c_l1; dl2 = n2 - c_l2; dl3 = n3 - c_l3; dl4 = n4 - c_l4; dl5 = n5 - c_l5;
              lmax = mpc_max(lmax, mpc_max(mpc_max(mpc_max(mpc_abs(dl0), mpc_abs(dl1)), mpc_max(mpc_abs(dl2), mpc_abs(dl3))), mpc_max(mpc_abs(dl4), mpc_abs(dl5))));
            }
            lam_t[0] = c_l0 + alpha_l * dl0; lam_t[1] = c_l1 + alpha_l * dl1; lam_t[2] = c_l2 + alpha_l * dl2;
            lam_t[3] = c_l3 + alpha_l * dl3; lam_t[4] = c_l4 + alpha_l * dl4; lam_t[5] = c_l5 + alpha_l * dl5;
            MPC_UNROLL
            for (int i = 0; i < 6; i++) { ws.it(M, J, F_LAM + i) = lam_t[i]; Ev.lsum += mpc_abs(lam_t[i]); }
            R z0_t[4];
            const R xo[2] = {p0, v0k}, xn[2] = {s_t[2], s_t[3]}, dxb[2] = {ds2, ds3}, lo[2] = {yl, vl}, hi[2] = {yu, vu};
            MPC_UNROLL
            for (int b = 0; b < 2; b++) {
              const R islo = frcp1(xo[b] - lo[b]), isuo = frcp1(hi[b] - xo[b]);
              const R zl = b ? c_zlv : c_zlp, zu = b ? c_zuv : c_zup;
              const R dzl = mu * islo - zl - zl * islo * dxb[b];
              const R dzu = mu * isuo - zu + zu * isuo * dxb[b];
              const R sl = xn[b] - lo[b], su = hi[b] - xn[b];
              const R isl = frcp1(sl), isu = frcp1(su);
              R a = zl + alpha_z * dzl, c = zu + alpha_z * dzu;
              a = mpc_max(mpc_min(a, ksm * isl), ksi * isl);
              c = mpc_max(mpc_min(c, ksm * isu), ksi * isu);
              z0_t[2 * b] = a; z0_t[2 * b + 1] = c;
              ws.it(M, J, F_ZL + b) = a; ws.it(M, J, F_ZU + b) = c;
              Ev.zsum += a + c;
              const R pl = sl * a, pu = su * c;
              Ev.cmin = mpc_min(Ev.cmin, mpc_min(pl, pu)); Ev.cmax = mpc_max(Ev.cmax, mpc_max(pl, pu));
            }
            zs0 = z0_t[0]; zs1 = z0_t[1]; zs2 = z0_t[2]; zs3 = z0_t[3];
          }
#endif
        }
        if (KZ || k < M) {
          /* ---- trial: transition k, (s_k, u_k) -> s_{k+1} ---- */
          LinR L;
          linearise(s_t, del_t, acc_t, sn_t, L);
          MPC_UNROLL
          for (int i = 0; i < 6; i++) { Ev.theta += mpc_abs(L.c[i]); Ev.cinf = mpc_max(Ev.cinf, mpc_abs(L.c[i])); }
          const R ddl = !KZ ? del_t - n_del_t : R(0.0);          /* delta_k - delta_{k-1} */
          const R ddn = (k + 1 < M) ? del_nx - del_t : R(0.0);        /* delta_{k+1} - delta_k */
          if (!KZ) Ev.f += wdd * ddl * ddl;
          const R v = s_t[3], vdt = v * dt, Apv = del_t * dtLf, Bp = v * dtLf;
          const R l25 = ln_t[2] + ln_t[5];
          /* rows of u_k */
          const R rd = df * (R(2.0) * wd * del_t + R(2.0) * wdd * ddl - R(2.0) * wdd * ddn) - Bp * l25 - zdl_t + zdu_t;
          const R ra = -dt * ln_t[3] - zal_t + zau_t;
          Ev.dinf = mpc_max(Ev.dinf, mpc_max(mpc_abs(rd), mpc_abs(ra)));
          /* rows of s_k (k>=1) with A_k of the trial point */
#if MPC_S0_VARIABLE
          if (!KZ || s0_rows) {
            /* (k = 0: the rows of the initial state's own variables, with the i = 0 terms of the objective) */
            const R gvk = !KZ ? wv * (s_t[3] - vref) : wv * (s_t[3] - vref0) + wneg0 * s_t[3];
            const R wck = !KZ ? wc : wc0, wek = !KZ ? we : we0;
#else
          if (!KZ) {
            const R gvk = wv * (s_t[3] - vref), wck = wc, wek = we;
#endif
            const R r0 = lam_t[0] - (ln_t[0] + L.fp * ln_t[4] - L.g1 * ln_t[5]);
            const R r1 = lam_t[1] - (ln_t[1] - ln_t[4]);
            const R r2 = lam_t[2] - (-vdt * L.sp * ln_t[0] + vdt * L.cp * ln_t[1] + l25) - zs0 + zs1;
            const R r3 = df * R(2.0) * gvk + lam_t[3] -
                              (dt * L.cp * ln_t[0] + dt * L.sp * ln_t[1] + Apv * l25 + ln_t[3] + dt * L.se * ln_t[4]) - zs2 + zs3;
            const R r4 = df * R(2.0) * wck * s_t[4] + lam_t[4];
            const R r5 = df * R(2.0) * wek * s_t[5] + lam_t[5] - vdt * L.ce * ln_t[4];
            Ev.dinf = mpc_max(Ev.dinf, mpc_max(mpc_max(mpc_abs(r0), mpc_abs(r1)), mpc_max(mpc_max(mpc_abs(r2), mpc_abs(r3)), mpc_max(mpc_abs(r4), mpc_abs(r5)))));
          }
        } else {
          /* terminal state rows */
          const R r2 = lam_t[2] - zs0 + zs1;
          const R r3 = df * R(2.0) * wv * (s_t[3] - vref) + lam_t[3] - zs2 + zs3;
          const R r4 = df * R(2.0) * wc * s_t[4] + lam_t[4];
          const R r5 = df * R(2.0) * we * s_t[5] + lam_t[5];
          Ev.dinf = mpc_max(Ev.dinf, mpc_max(mpc_max(mpc_abs(lam_t[0]), mpc_abs(lam_t[1])), mpc_max(mpc_max(mpc_abs(r2), mpc_abs(r3)), mpc_max(mpc_abs(r4), mpc_abs(r5)))));
        }
        /* carry to step k-1 */
        MPC_UNROLL
        for (int i = 0; i < 6; i++) { sn_o[i] = s_o[i]; sn_t[i] = s_t[i]; ln_t[i] = lam_t[i]; }
        del_nx = del_t;
        del_t = n_del_t; acc_t = n_acc_t; zdl_t = n_zdl; zdu_t = n_zdu; zal_t = n_zal; zau_t = n_zau;
        del_o = n_del_o; acc_o = n_acc_o; ddk = n_ddk; lx = lo0; ly = lo1; lp = lo2; lc = lo4; le = lo5;
    };
    if constexpr (WAVE) {
      /* One instance per wavefront: lane r does step k = r + 1 (record r, transition k) for all stages at once.  What runs through
       * the lanes in order is only what the sequential sweep carries from step to step: the costate lam+ (six multiply-add rows
       * per stage) and, in a second turn, the sums of the evaluation in the order the sequential sweep adds them. */
      const bool mine = wlane < M;
      const int r = mine ? wlane : M - 1, k = r + 1;
      const bool last = k == M;
      const int rn = last ? r : r + 1;                /* record k (clamped at the terminal stage, where it is not used) */
      R s_o[6], ds[6], lo_[6], snc[6];
      MPC_UNROLL
      for (int i = 0; i < 6; i++) { s_o[i] = ws.it(r, I, F_S + i); ds[i] = ws.it(r, 0, F_D + D_S + i); lo_[i] = ws.it(r, I, F_LAM + i); snc[i] = ws.it(rn, I, F_S + i); }
      const R n_del_o = ws.it(r, I, F_U + 0), n_acc_o = ws.it(r, I, F_U + 1);
      const R ddel = ws.it(r, 0, F_D + D_U + 0), dacc = ws.it(r, 0, F_D + D_U + 1);
      R zlc[4], zuc[4];
      MPC_UNROLL
      for (int b = 0; b < 4; b++) { zlc[b] = ws.it(r, I, F_ZL + b); zuc[b] = ws.it(r, I, F_ZU + b); }
      const R del_k = ws.it(rn, I, F_U + 0), acc_k = ws.it(rn, I, F_U + 1), ddk_k = ws.it(rn, 0, F_D + D_U + 0);
      const R lxk = ws.it(rn, I, F_LAM + 0), lyk = ws.it(rn, I, F_LAM + 1), lpk = ws.it(rn, I, F_LAM + 2), lck = ws.it(rn, I, F_LAM + 4), lek = ws.it(rn, I, F_LAM + 5);
      /* ---- costate: this stage's coefficients, then the recursion ---- */
      R Lm[6] = {0, 0, 0, 0, 0, 0};
      if (with_costate) {
        R Hpp, Hvv, Hee, Hcc, gp, gv, ge, gc;
        state_terms(s_o[2], s_o[3], s_o[4], s_o[5], zlc[0], zuc[0], zlc[1], zuc[1], Hpp, Hvv, Hee, Hcc, gp, gv, ge, gc);
        const R v = s_o[3];
        LinR L;
        linearise(s_o, del_k, acc_k, snc, L);
        const R sp = L.sp, cp = L.cp, se = L.se, ce = L.ce, fp = L.fp, g1 = L.g1, h3 = L.h3, fpp = L.fpp;
        const R vdt = v * dt, Apv = del_k * dtLf;
        const R Hxx = -lck * fpp + lek * h3, Hpsi2 = (lxk * cp + lyk * sp) * vdt, Hpv = (lxk * sp - lyk * cp) * dt;
        const R Hee2 = lck * vdt * se, Hev = -lck * dt * ce, Hvd = -(lpk + lek) * dtLf;
        for (int t = M; t >= 1; --t) {
          R n0, n1, n2, n3, n4, n5;
          if (last) {
            n0 = -(hxy * ds[0]);
            n1 = -(hxy * ds[1]);
            n2 = -(gp + (Hpp + dw) * ds[2]);
            n3 = -(gv + (Hvv + dw) * ds[3]);
            n4 = -(gc + (Hcc + dw) * ds[4]);
            n5 = -(ge + (Hee + dw) * ds[5]);
          } else {
            const R L25 = L2 + L5;
            n0 = L0 + fp * L4 - g1 * L5 - (Hxx + hxy) * ds[0];
            n1 = L1 - L4 - hxy * ds[1];
            n2 = -vdt * sp * L0 + vdt * cp * L1 + L25 - gp - (Hpp + dw + Hpsi2) * ds[2] - Hpv * ds[3];
            n3 = dt * cp * L0 + dt * sp * L1 + Apv * L25 + L3 + dt * se * L4 - gv - Hpv * ds[2] -
                 (Hvv + dw) * ds[3] - Hev * ds[5] - Hvd * ddk_k;
            n4 = -gc - (Hcc + dw) * ds[4];
            n5 = vdt * ce * L4 - ge - (Hee + dw + Hee2) * ds[5] - Hev * ds[3];
          }
          if (wlane == t - 1) { Lm[0] = n0; Lm[1] = n1; Lm[2] = n2; Lm[3] = n3; Lm[4] = n4; Lm[5] = n5; }
          L0 = wave_bcast<WAVE>(n0, t - 1, wbase); L1 = wave_bcast<WAVE>(n1, t - 1, wbase); L2 = wave_bcast<WAVE>(n2, t - 1, wbase);
          L3 = wave_bcast<WAVE>(n3, t - 1, wbase); L4 = wave_bcast<WAVE>(n4, t - 1, wbase); L5 = wave_bcast<WAVE>(n5, t - 1, wbase);
        }
      }
      R dl[6] = {0, 0, 0, 0, 0, 0};
      R lmax_k = R(0.0);
      if (with_costate) {
        MPC_UNROLL
        for (int i = 0; i < 6; i++) dl[i] = Lm[i] - lo_[i];
        lmax_k = mpc_max(mpc_max(mpc_max(mpc_abs(dl[0]), mpc_abs(dl[1])), mpc_max(mpc_abs(dl[2]), mpc_abs(dl[3]))), mpc_max(mpc_abs(dl[4]), mpc_abs(dl[5])));
      }
      /* ---- trial record r ---- */
      R lam_t[6], s_t[6];
      MPC_UNROLL
      for (int i = 0; i < 6; i++) { lam_t[i] = lo_[i] + alpha_l * dl[i]; s_t[i] = s_o[i] + alpha * ds[i]; }
      const R n_del_t = n_del_o + alpha * ddel, n_acc_t = n_acc_o + alpha * dacc;
      R rec[IT_SZ] = {};
      MPC_UNROLL
      for (int i = 0; i < 6; i++) { rec[F_S + i] = s_t[i]; rec[F_LAM + i] = lam_t[i]; }
      rec[F_U] = n_del_t; rec[F_U + 1] = n_acc_t;
      const R xo[4] = {s_o[2], s_o[3], n_del_o, n_acc_o};
      const R xn[4] = {s_t[2], s_t[3], n_del_t, n_acc_t};
      const R dxb[4] = {ds[2], ds[3], ddel, dacc};
      const R lo[4] = {yl, vl, this->dl, al}, hi[4] = {yu, vu, du, au};
      R zln[4], zun[4], sl_[4], su_[4], prod = R(1.0);
      bool bad = false;
      MPC_UNROLL
      for (int b = 0; b < 4; b++) {
        const R islo = frcp1(xo[b] - lo[b]), isuo = frcp1(hi[b] - xo[b]);
        const R dzl = mu * islo - zlc[b] - zlc[b] * islo * dxb[b];
        const R dzu = mu * isuo - zuc[b] + zuc[b] * isuo * dxb[b];
        const R sl = xn[b] - lo[b], su = hi[b] - xn[b];
        if (!(sl > R(0.0)) || !(su > R(0.0))) bad = true;
        const R isl = frcp1(sl), isu = frcp1(su);
        R a = zlc[b] + alpha_z * dzl, c = zuc[b] + alpha_z * dzu;
        a = mpc_max(mpc_min(a, ksm * isl), ksi * isl);
        c = mpc_max(mpc_min(c, ksm * isu), ksi * isu);
        zln[b] = a; zun[b] = c; sl_[b] = sl; su_[b] = su;
        prod *= sl * su;
        rec[F_ZL + b] = a; rec[F_ZU + b] = c;
      }
      if (mine) {
        ws.template store_run<F_S, F_ZL - F_S>(r, J, rec + F_S);
        ws.template store_run<F_ZL, 8>(r, J, rec + F_ZL);
      }
      const R lprod = flog(prod);
      /* ---- transition k at the trial point (the neighbours' trial records are in LDS now) ---- */
      R cT[6] = {0, 0, 0, 0, 0, 0}, ddl = 0, dinf_u = 0, dinf_s = 0;
      if (!last) {
        R snt[6], lnt[6];
        MPC_UNROLL
        for (int i = 0; i < 6; i++) { snt[i] = ws.it(rn, J, F_S + i); lnt[i] = ws.it(rn, J, F_LAM + i); }
        const R dlt = ws.it(rn, J, F_U + 0), act = ws.it(rn, J, F_U + 1);
        const R zdl = ws.it(rn, J, F_ZL + 2), zdu = ws.it(rn, J, F_ZU + 2), zal = ws.it(rn, J, F_ZL + 3), zau = ws.it(rn, J, F_ZU + 3);
        const R dnx = (k + 1 < M) ? (R)ws.it(rn + 1 < M ? rn + 1 : rn, J, F_U + 0) : R(0.0);
        LinR Lt;
        linearise(s_t, dlt, act, snt, Lt);
        MPC_UNROLL
        for (int i = 0; i < 6; i++) cT[i] = Lt.c[i];
        ddl = dlt - n_del_t;
        const R ddn = (k + 1 < M) ? dnx - dlt : R(0.0);
        const R v = s_t[3], vdt = v * dt, Apv = dlt * dtLf, Bp = v * dtLf;
        const R l25 = lnt[2] + lnt[5];
        const R rd = df * (R(2.0) * wd * dlt + R(2.0) * wdd * ddl - R(2.0) * wdd * ddn) - Bp * l25 - zdl + zdu;
        const R ra = -dt * lnt[3] - zal + zau;
        dinf_u = mpc_max(mpc_abs(rd), mpc_abs(ra));
        const R r0 = lam_t[0] - (lnt[0] + Lt.fp * lnt[4] - Lt.g1 * lnt[5]);
        const R r1 = lam_t[1] - (lnt[1] - lnt[4]);
        const R r2 = lam_t[2] - (-vdt * Lt.sp * lnt[0] + vdt * Lt.cp * lnt[1] + l25) - zln[0] + zun[0];
        const R r3 = df * R(2.0) * (wv * (s_t[3] - vref)) + lam_t[3] -
                          (dt * Lt.cp * lnt[0] + dt * Lt.sp * lnt[1] + Apv * l25 + lnt[3] + dt * Lt.se * lnt[4]) - zln[1] + zun[1];
        const R r4 = df * R(2.0) * wc * s_t[4] + lam_t[4];
        const R r5 = df * R(2.0) * we * s_t[5] + lam_t[5] - vdt * Lt.ce * lnt[4];
        dinf_s = mpc_max(mpc_max(mpc_abs(r0), mpc_abs(r1)), mpc_max(mpc_max(mpc_abs(r2), mpc_abs(r3)), mpc_max(mpc_abs(r4), mpc_abs(r5))));
      } else {
        const R r2 = lam_t[2] - zln[0] + zun[0];
        const R r3 = df * R(2.0) * wv * (s_t[3] - vref) + lam_t[3] - zln[1] + zun[1];
        const R r4 = df * R(2.0) * wc * s_t[4] + lam_t[4];
        const R r5 = df * R(2.0) * we * s_t[5] + lam_t[5];
        dinf_s = mpc_max(mpc_max(mpc_abs(lam_t[0]), mpc_abs(lam_t[1])), mpc_max(mpc_max(mpc_abs(r2), mpc_abs(r3)), mpc_max(mpc_abs(r4), mpc_abs(r5))));
      }
      /* ---- the sums, in the order of the sequential sweep ---- */
      for (int t = M; t >= 1; --t) {
        R e_ls = Ev.lsum, e_zs = Ev.zsum, e_L = Ev.L, e_f = Ev.f, e_th = Ev.theta, e_ci = Ev.cinf, e_di = Ev.dinf, e_mn = Ev.cmin, e_mx = Ev.cmax, e_lm = lmax;
        e_lm = mpc_max(e_lm, lmax_k);
        MPC_UNROLL
        for (int i = 0; i < 6; i += 2) e_ls += mpc_abs(lam_t[i]) + mpc_abs(lam_t[i + 1]);
        MPC_UNROLL
        for (int b = 0; b < 4; b++) {
          e_zs += zln[b] + zun[b];
          const R pl = sl_[b] * zln[b], pu = su_[b] * zun[b];
          e_mn = mpc_min(e_mn, mpc_min(pl, pu)); e_mx = mpc_max(e_mx, mpc_max(pl, pu));
        }
        e_L += lprod;
        const R dv = s_t[3] - vref;
        e_f += wc * s_t[4] * s_t[4] + we * s_t[5] * s_t[5] + wv * dv * dv + wd * n_del_t * n_del_t;
        if (!last) {
          MPC_UNROLL
          for (int i = 0; i < 6; i++) { e_th += mpc_abs(cT[i]); e_ci = mpc_max(e_ci, mpc_abs(cT[i])); }
          e_f += wdd * ddl * ddl;
          e_di = mpc_max(e_di, dinf_u);
        }
        e_di = mpc_max(e_di, dinf_s);
        Ev.lsum = wave_bcast<WAVE>(e_ls, t - 1, wbase); Ev.zsum = wave_bcast<WAVE>(e_zs, t - 1, wbase); Ev.L = wave_bcast<WAVE>(e_L, t - 1, wbase); Ev.f = wave_bcast<WAVE>(e_f, t - 1, wbase);
        Ev.theta = wave_bcast<WAVE>(e_th, t - 1, wbase); Ev.cinf = wave_bcast<WAVE>(e_ci, t - 1, wbase); Ev.dinf = wave_bcast<WAVE>(e_di, t - 1, wbase);
        Ev.cmin = wave_bcast<WAVE>(e_mn, t - 1, wbase); Ev.cmax = wave_bcast<WAVE>(e_mx, t - 1, wbase); lmax = wave_bcast<WAVE>(e_lm, t - 1, wbase);
      }
      Ev.du0 = wave_bcast<WAVE>(mpc_max(mpc_abs(ddel), mpc_abs(dacc)), 0, wbase);
      if (wave_group_any<WAVE>(mine && bad, wbase)) Ev.ok = false;
      /* ---- what the step of k = 1 hands to the last one ---- */
      MPC_UNROLL
      for (int i = 0; i < 6; i++) { sn_o[i] = ws.it(0, I, F_S + i); sn_t[i] = ws.it(0, J, F_S + i); ln_t[i] = ws.it(0, J, F_LAM + i); }
      del_nx = M >= 2 ? (R)ws.it(1, J, F_U + 0) : R(0.0);
      del_t = ws.it(0, J, F_U + 0); acc_t = ws.it(0, J, F_U + 1);
      zdl_t = ws.it(0, J, F_ZL + 2); zdu_t = ws.it(0, J, F_ZU + 2); zal_t = ws.it(0, J, F_ZL + 3); zau_t = ws.it(0, J, F_ZU + 3);
      del_o = ws.it(0, I, F_U + 0); acc_o = ws.it(0, I, F_U + 1); ddk = ws.it(0, 0, F_D + D_U + 0);
      lx = ws.it(0, I, F_LAM + 0); ly = ws.it(0, I, F_LAM + 1); lp = ws.it(0, I, F_LAM + 2); lc = ws.it(0, I, F_LAM + 4); le = ws.it(0, I, F_LAM + 5);
    } else {
      MPC_STAGE_LOOP
      for (int k = M; k >= 1; --k) step_k(std::false_type(), k);
    }
    step_k(std::true_type(), 0);
    if (!(Ev.theta == Ev.theta) || !(Ev.f == Ev.f) || !(Ev.L == Ev.L) || !(Ev.dinf == Ev.dinf)) Ev.ok = false;
    return Ev;
  }

  /* a pinned start value after a step of length alpha; exactly the pinned value once it has arrived */
  MPC_HD R trial_x0(R x0, R pinned, R alpha_) const { return x0 == pinned ? pinned : x0 + alpha_ * (pinned - x0); }
  MPC_HD R kkt_error(const EvalR &e, R mu_) const {
#if MPC_S0_VARIABLE
    /* (with the initial state's own rows: the reference's 6N rows; bounds of psi_i, v_i (i = 0..N-1), delta_i, a_i) */
    const R m = s0_rows ? R(6.0) * (M + 1) : R(6.0) * M, nb = s0_rows ? R(8.0) * M + R(4.0) : R(8.0) * M;
#else
    const R m = R(6.0) * M, nb = R(8.0) * M;
#endif
    const R sd = mpc_max(IC::s_max, (e.lsum + e.zsum) / (m + nb)) / IC::s_max;
    const R sc = mpc_max(IC::s_max, e.zsum / nb) / IC::s_max;
    const R compl_ = mpc_max(mpc_abs(e.cmax - mu_), mpc_abs(e.cmin - mu_));
    return mpc_max(mpc_max(e.dinf / sd, e.cinf), compl_ / sc);
  }

  /* IPOPT's termination tests (OptimalityErrorConvergenceCheck, defaults of 3.12 that MPC.cpp:160-179 leaves alone): besides
   * the scaled error E_0 <= tol an iterate must have its UNSCALED dual infeasibility, constraint violation and complementarity
   * within dual_inf_tol (1) / constr_viol_tol (1e-4) / compl_inf_tol (1e-4) to be converged; with the acceptable_* values in
   * their place it is ACCEPTABLE.  Unscaled = divided by the objective scaling df (the bound duals and lam live in the scaled
   * problem); the products slack x dual are positive, so the complementarity at mu = 0 is e.cmax.  The fp32 solver keeps its
   * own rule (tol_f32 is above these tolerances). */
  MPC_HD bool converged_at(const EvalR &e, R E0) const {
    if (!(E0 <= tol)) return false;
    if (sizeof(R) == 4) return true;
    return e.dinf <= (R)P.dual_inf_tol * df && e.cinf <= (R)P.constr_viol_tol && e.cmax <= (R)P.compl_inf_tol * df;
  }
  MPC_HD bool acceptable_at(const EvalR &e, R E0) const {
    if (sizeof(R) == 4 || P.acceptable_iter <= 0) return false;
    return E0 <= (R)P.acceptable_tol && e.dinf <= (R)P.acceptable_dual_inf_tol * df && e.cinf <= (R)P.acceptable_constr_viol_tol &&
           e.cmax <= (R)P.acceptable_compl_inf_tol * df;
  }

  MPC_HD bool filter_rejects(R th, R ph) const {
    bool r = false;
    r |= (nf > 0) && th >= fth0 && ph >= fph0;
    r |= (nf > 1) && th >= fth1 && ph >= fph1;
    r |= (nf > 2) && th >= fth2 && ph >= fph2;
    r |= (nf > 3) && th >= fth3 && ph >= fph3;
    return r;
  }
  MPC_HD void filter_add(R th, R ph) {
    int slot = nf;
    if (nf >= 4) {
      /* full: overwrite the entry with the largest theta (the least restrictive one) */
      slot = 0;
      R worst = fth0;
      if (fth1 > worst) { worst = fth1; slot = 1; }
      if (fth2 > worst) { worst = fth2; slot = 2; }
      if (fth3 > worst) { worst = fth3; slot = 3; }
    } else nf++;
    if (slot == 0) { fth0 = th; fph0 = ph; }
    else if (slot == 1) { fth1 = th; fph1 = ph; }
    else if (slot == 2) { fth2 = th; fph2 = ph; }
    else { fth3 = th; fph3 = ph; }
  }

  /* the start point of MPC.cpp:207-210 (zeros), pushed into the interior, in iterate slot 0 */
  MPC_HD void start_point() {
    for (int k = 0; k < M; ++k) {
      MPC_UNROLL
      for (int i = 0; i < 6; i++) { ws.it(k, IT0, F_S + i) = R(0.0); ws.it(k, IT0, F_LAM + i) = R(0.0); }
      ws.it(k, IT0, F_S + 2) = psi_start;
      ws.it(k, IT0, F_U + 0) = R(0.0); ws.it(k, IT0, F_U + 1) = R(0.0);
      MPC_UNROLL
      for (int b = 0; b < 4; b++) { ws.it(k, IT0, F_ZL + b) = R(1.0); ws.it(k, IT0, F_ZU + b) = R(1.0); }
      MPC_UNROLL
      for (int i = 0; i < D_N; i++) ws.setD(k, i, R(0.0));
    }
  }

  /* ------------------------------------------------------------------ */
  /* set-up: instance constants, start point (MPC.cpp:204-257)           */
  /* ------------------------------------------------------------------ */
  MPC_HD int setup(const R *state6, const R *coef5, R yaw_lo, R yaw_hi, const R *w12,
                   bool write_start = true) {
    MPC_UNROLL
    for (int i = 0; i < 6; i++) st[i] = state6[i];
    MPC_UNROLL
    for (int i = 0; i < MPC_NCOEF; i++) coef[i] = coef5[i];
    M = P.N - 1; dt = (R)P.dt; iLf = (R)(1.0 / P.Lf); dtLf = (R)(P.dt / P.Lf);
    /* IPOPT's bound_relax_factor (default 1e-8, untouched by MPC.cpp:160-179): every finite variable bound is moved
     * outwards by factor * max(1, |bound|) before the solve; the start point is pushed inside the RELAXED bounds and
     * the returned point is projected back into the caller's (unpack).  This is what makes a solve that starts ON a
     * bound well posed: the psi_0 of a closed loop whose heading has reached yawHigh (test.cpp:79-111). */
    const double rf = P.bound_relax_factor;
    yl = (R)((double)yaw_lo - rf * fmax(1.0, fabs((double)yaw_lo))); yu = (R)((double)yaw_hi + rf * fmax(1.0, fabs((double)yaw_hi)));
    vu = (R)(P.max_speed + rf * fmax(1.0, P.max_speed)); vl = -vu;
    du = (R)(P.max_steering + rf * fmax(1.0, P.max_steering)); dl = -du;
    al = (R)(P.max_deceleration - rf * fmax(1.0, fabs(P.max_deceleration))); au = (R)(P.max_acceleration + rf * fmax(1.0, fabs(P.max_acceleration)));
    fth0 = fth1 = fth2 = fth3 = fph0 = fph1 = fph2 = fph3 = R(0.0);
    lsm = false; cur = 0; iters = 0; n_reg = 0; nf = 0; E.f = R(0.0);
    /* fp32: its own tolerance; the outputs cannot stop moving below the noise of an fp32 step */
    tol = (R)(sizeof(R) == 8 ? P.tol : P.tol_f32);
    out_tol = (R)(sizeof(R) == 8 ? P.out_step_tol : 0.6 * P.tol_f32);
    /* Branch outcomes at the start point xi = (state at index 0, zeros elsewhere):
     * for i >= 1 every variable is 0, so (MPC.cpp:72-112)
     *   |cte_i| < ctePanic  -> w[0] unless ctePanic <= 0
     *   |epsi_i| > epsiPanic -> w[10] only if epsiPanic < 0
     *   vref_i = computeSpeedTarget(0, maxSpeed); v_i < 0, a_i > 0, a_i < 0,
     *   a_{i+1} > a_i are all false -> those terms are not on the tape. */
    wc = (0.0 < P.cte_panic) ? w12[0] : w12[11];
    we = (0.0 > P.epsi_panic) ? w12[10] : w12[1];
    wv = w12[2]; wd = w12[3]; wdd = w12[4];
    vref = (R)speed_target(P, 0.0, P.max_speed);
    /* i = 0 terms: constants of the objective (their variables are fixed), MPC.cpp:71-92 */
#if !MPC_S0_VARIABLE
    R wc0, we0, vref0;
#endif
    wc0 = ((double)mpc_abs(st[4]) < P.cte_panic) ? w12[0] : w12[11];
    we0 = ((double)mpc_abs(st[5]) > P.epsi_panic) ? w12[10] : w12[1];
    vref0 = (R)speed_target(P, (double)st[2], P.max_speed);
#if MPC_S0_VARIABLE
    wneg0 = (st[3] < R(0.0)) ? w12[9] : R(0.0);
    s0_rows = P.initial_state_rows != 0;
#endif
    cost0 = wc0 * st[4] * st[4] + we0 * st[5] * st[5] + wv * (st[3] - vref0) * (st[3] - vref0);
    R g0 = mpc_max(mpc_abs(R(2.0) * wc0 * st[4]), mpc_abs(R(2.0) * we0 * st[5]));
    R gv0 = R(2.0) * wv * (st[3] - vref0);
    if (st[3] < R(0.0)) { cost0 += w12[9] * st[3] * st[3]; gv0 += R(2.0) * w12[9] * st[3]; }
    g0 = mpc_max(g0, mpc_abs(gv0));
    /* gradient-based objective scaling at the start point (IPOPT default) */
    g0 = mpc_max(g0, mpc_abs(R(2.0) * wv * vref));
    df = (g0 > R(100.0)) ? mpc_max(R(100.0) / g0, R(1e-8)) : R(1.0);
    /* start point: zeros (MPC.cpp:207-210), pushed into the interior like IPOPT does */
    R psi0 = R(0.0);
    {
      const R pl = mpc_min(IC::kappa1 * mpc_max(R(1.0), mpc_abs(yl)), IC::kappa2 * (yu - yl));
      const R pu = mpc_min(IC::kappa1 * mpc_max(R(1.0), mpc_abs(yu)), IC::kappa2 * (yu - yl));
      psi0 = mpc_min(mpc_max(psi0, yl + pl), yu - pu);
    }
    psi_start = psi0;
    if (write_start) start_point();   /* a parked instance that is being resumed brings its iterate along */
    /* the fixed initial state must satisfy its own bounds (MPC.cpp:229-239 vs :269-281) */
    if (!(st[2] >= yl && st[2] <= yu) || !(mpc_abs(st[3]) <= vu) || !(yl < yu)) return MPC_STATUS_INFEASIBLE;
    return MPC_STATUS_SUCCESS;
  }

  /* ------------------------------------------------------------------ */
  /* the interior-point iteration                                         */
  /* ------------------------------------------------------------------ */
  enum { MPC_RUNNING = -1, MPC_PROMOTE = -2 };   /* PROMOTE: the fp32 phase of a mixed-precision solve hands the instance to fp64 */
  enum { PH_EVAL0 = 0, PH_LS = 1, PH_DIR = 2, PH_BACKTRACK = 3, PH_REG = 4 };
  enum { kMaxPolish = 6 };
  enum { kPromoteIterCap = 16 };   /* mixed precision: the fp32 phase's allowance per instance (the bulk hands over after 8-12 iterations); one that uses it up is solved in fp64 from the start point */
  /* state of the interior-point loop (see step()) */
  int phase, iter, n_polish;
  int acc_count;   /* acceptable iterates in a row (IPOPT's acceptable_iter counter) */
  bool ls_start, tiny;
  /* set when a line search fails at an almost feasible point (constraint violation <= 1e-2 tol): IPOPT does not start its
   * restoration phase there ("restoration phase called, but point is almost feasible"), so the stand-in must not restart */
  bool no_restart;
  /* Mixed precision across phases (MpcParams.f32_finish / f64_f32_start): an fp32 solver with promote_mu > 0 stops being
   * responsible for an instance as soon as its barrier parameter is about to go below promote_mu (or the instance has met
   * tol_f32, or its line search has run out of single precision): step() returns MPC_PROMOTE at a pass boundary, the
   * instance is parked (park()), and an fp64 solver takes the iterate over (unpark() + promoted()): it re-evaluates the
   * point in fp64 and carries on with the same state machine to tol and the polish. */
  static constexpr bool kCanPromote = sizeof(R) == 4;   /* only the fp32 solver ever hands over: none of it is in the fp64 kernels */
  R promote_mu = R(0.0);
  int promote_cap = kPromoteIterCap;
  mutable bool promote_clean = true;   /* the hand-over came where it should (barrier parameter or tol_f32 reached), not out of trouble (allowance used up, line search or inertia correction out of single precision) */
  bool keep_theta = false;
  R out_step;     /* |alpha d(delta_0, a_0)|_inf of the last accepted step */
  R out_prev;     /* the same of the step before (fp32 wants two quiet steps in a row) */
  R tol, out_tol;  /* "tol" of this precision (MpcParams.tol or tol_f32) and the polish's step tolerance */
  R alpha, alpha_l, alpha_z, dw_cur, theta_max, theta_min, dw_last;
  R dw_try;        /* the regularisation the next backward sweep is tried with (phase REG) */
  int reg_tries;
  R theta_k, phi_k, pth, pdp, amin;   /* line-search state */

  /* Solve from the start point that setup()/start_point() has written.
   * A line search that runs out of step length is where IPOPT would enter its feasibility-restoration phase.
   * Stand-in (same as the oracle's): restart ONCE from the start point with zero equality multipliers. */
  MPC_HD int solve() {
    int it_total = 0, attempt = 0;
    begin(true);
    for (;;) {
      const int r = step();
      if (r == MPC_RUNNING) continue;
      if (r == MPC_STATUS_LINESEARCH && attempt == 0 && !no_restart) {
        attempt = 1; it_total += iters;
        start_point();
        begin(false);
        continue;
      }
      iters += it_total;
      return r;
    }
  }

  /* The state of an unfinished instance between two passes with phase == PH_DIR (everything else lives in the
   * current iterate slot of the workspace or is recomputed by setup()): 36 values through an accessor a(q). */
  enum { PARK_N = 47 };
  template <class A> MPC_HD void park(A a, int attempt, int it_total) const {
    a(0) = mu; a(1) = tau; a(2) = E.theta; a(3) = E.cinf; a(4) = E.f; a(5) = E.L; a(6) = E.dinf; a(7) = E.cmin; a(8) = E.cmax;
    a(9) = E.lsum; a(10) = E.zsum; a(11) = fth0; a(12) = fth1; a(13) = fth2; a(14) = fth3; a(15) = fph0; a(16) = fph1;
    a(17) = fph2; a(18) = fph3; a(19) = theta_max; a(20) = theta_min; a(21) = dw_last;
    a(22) = (R)nf; a(23) = (R)iter; a(24) = (R)n_reg; a(25) = (R)cur; a(26) = E.ok ? R(1.0) : R(0.0);
    a(27) = ls_start ? R(1.0) : R(0.0); a(28) = (R)attempt; a(29) = (R)it_total;
    a(30) = out_step; a(31) = (R)n_polish; a(32) = out_prev; a(35) = R(0.0); a(36) = (R)acc_count;
#if MPC_S0_VARIABLE
    a(33) = p0; a(34) = v0k;
    if (s0_rows) {
      const int Ic = it(cur);
      MPC_UNROLL
      for (int i = 0; i < 6; i++) a(37 + i) = s0_lam(Ic, i);
      a(43) = s0_z(Ic, 0, 0); a(44) = s0_z(Ic, 0, 1); a(45) = s0_z(Ic, 1, 0); a(46) = s0_z(Ic, 1, 1);
    } else {
      for (int q = 37; q < 47; q++) a(q) = R(0.0);
    }
#else
    a(33) = R(0.0); a(34) = R(0.0);
    for (int q = 37; q < 47; q++) a(q) = R(0.0);
#endif
  }
  template <class A> MPC_HD void unpark(A a, int &attempt, int &it_total) {
    begin(a(27) != R(0.0));
    mu = a(0); tau = a(1); E.theta = a(2); E.cinf = a(3); E.f = a(4); E.L = a(5); E.dinf = a(6); E.cmin = a(7); E.cmax = a(8);
    E.lsum = a(9); E.zsum = a(10); fth0 = a(11); fth1 = a(12); fth2 = a(13); fth3 = a(14); fph0 = a(15); fph1 = a(16);
    fph2 = a(17); fph3 = a(18); theta_max = a(19); theta_min = a(20); dw_last = a(21);
    nf = (int)a(22); iter = (int)a(23); n_reg = (int)a(24); cur = (int)a(25); E.ok = a(26) != R(0.0);
    attempt = (int)a(28); it_total = (int)a(29); out_step = a(30); n_polish = (int)a(31); out_prev = a(32); acc_count = (int)a(36);
#if MPC_S0_VARIABLE
    p0 = a(33); v0k = a(34);
    if (s0_rows) {
      const int Ic = it(cur);
      MPC_UNROLL
      for (int i = 0; i < 6; i++) ws.it(M, Ic, F_LAM + i) = a(37 + i);
      ws.it(M, Ic, F_ZL + 0) = a(43); ws.it(M, Ic, F_ZU + 0) = a(44); ws.it(M, Ic, F_ZL + 1) = a(45); ws.it(M, Ic, F_ZU + 1) = a(46);
    }
#endif
    iters = iter; phase = PH_DIR;
  }

  /* after unpark() of an instance that another precision parked with MPC_PROMOTE: evaluate the point as this solver sees it
   * (one trial sweep with alpha = 0), then carry on; theta_max / theta_min stay those of the start point */
  MPC_HD void promoted() {
    phase = PH_EVAL0; keep_theta = true; ls_start = false; nf = 0; n_polish = 0; acc_count = 0; out_step = out_prev = IC::huge;
    alpha = alpha_l = alpha_z = dw_cur = R(0.0);
#if MPC_S0_VARIABLE
    /* psi_0 / v_0 have arrived at their pinned values long before a hand-over (a factor 100 per iteration); the other
     * precision's copy of them may lie outside THIS solver's relaxed bounds by its rounding, so they are taken as arrived */
    p0 = st[2]; v0k = st[3];
#endif
  }

  MPC_HD void begin(bool ls) {
    cur = 0; mu = IC::mu_init; tau = mpc_max(IC::tau_min, R(1.0) - mu); nf = 0; iters = 0; n_reg = 0; lsm = false;
    /* with the least-squares multiplier start the first pass is the LS pass itself: its trial sweep evaluates the
     * start point (primal part unchanged) with the estimated multipliers, so a separate evaluation is only needed
     * when that estimate is rejected or not wanted */
    phase = ls ? PH_LS : PH_EVAL0; iter = 0; ls_start = ls; tiny = false; n_polish = 0; acc_count = 0; no_restart = false; out_step = out_prev = IC::huge;
#if MPC_S0_VARIABLE
    p0 = pushed(st[2], yl, yu); v0k = pushed(st[3], vl, vu);
    if (s0_rows) s0_reset(IT0);
#endif
    alpha = alpha_l = alpha_z = dw_cur = R(0.0);
    theta_max = theta_min = dw_last = R(0.0);
    theta_k = phi_k = pth = pdp = amin = R(0.0);
  }

  /* No acceptable step.  fp64: IPOPT would enter its restoration phase (the caller's stand-in: one restart).  fp32: the
   * usual reason is that the iterate sits on the noise floor of single precision (multipliers of ~1e3 carry 6e-5 of
   * rounding into the dual residual, slacks of a few ulp cannot shrink), so a point whose optimality error is within
   * IPOPT's "acceptable" band -- here 10 x tol, IPOPT's acceptable_tol/tol is 100 -- is returned as solved. */
  MPC_HD int line_search_failed() {
    if ((kCanPromote && promote_mu > R(0.0))) { promote_clean = false; return MPC_PROMOTE; }   /* out of step length in single precision */
    const R E0 = kkt_error(E, R(0.0));
    if (sizeof(R) == 4 && E0 <= R(10.0) * tol) return MPC_STATUS_SUCCESS;
    /* a polish step that finds no acceptable length: the iterate had already met tol, it is the answer */
    if (n_polish > 0 && converged_at(E, E0)) return MPC_STATUS_SUCCESS;
    /* IPOPT would call its restoration phase now -- unless the point is acceptable ("restoration phase called at acceptable
     * point": STOP_AT_ACCEPTABLE_POINT), or almost feasible, where it gives up without trying (no_restart).  Not carried: an
     * acceptable iterate that is no longer the current one (IPOPT keeps a copy and falls back to it; the oracle restates that
     * and reports it, OrcSolveInfo.acceptable_restored_older -- on none of the populations this build is measured on). */
    if (acceptable_at(E, E0)) return MPC_STATUS_ACCEPTABLE;
    if (sizeof(R) == 8 && P.acceptable_iter > 0 && E.cinf <= R(1e-2) * tol) no_restart = true;
    return MPC_STATUS_LINESEARCH;
  }

  /* One pass of the interior-point loop, written as a small state machine so that each sweep has exactly ONE
   * call site (they are force-inlined; several call sites would multiply the code size), and so that the lanes of
   * a wave can be in different phases -- or, in the device kernel, on different instances:
   *   EVAL0     evaluate the start point
   *   LS        least-squares multiplier start (W&B section R(3.6), IPOPT default): one Riccati pass
   *             with identity Hessian; estimates above constr_mult_init_max = 1000 are discarded
   *   DIR       convergence test, barrier update, search direction, first trial of the line search
   *   REG       the same search direction again with the next regularisation (the Riccati sweep found the wrong inertia): a
   *             pass of its own, so that the other 63 lanes of the wave do not stand still while one instance repeats its
   *             backward sweep (1.6 times per pass on the hard instances of the survey population: a tenth of the waves
   *             has one)
   *   BACKTRACK further trials of the same line search
   * Returns MPC_RUNNING, or the final status of this attempt. */
  MPC_HD int step() {
    /* floor of the barrier parameter: IPOPT's tol/10; the fp32 solver goes to tol/25 (2e-5 at the default tol_f32): what
     * its answers lose against fp64 is mostly the barrier's pull on weakly active bounds (mu/z), not rounding, while
     * slacks of active bounds (mu/z ~ 1e-6 on a ~ 4.47) must stay above a few ulp */
    const R mu_floor = sizeof(R) == 8 ? tol / R(10.0) : tol / R(25.0);
    if (phase == PH_LS || phase == PH_DIR || phase == PH_REG) {
      if (phase != PH_REG) { dw_try = R(0.0); reg_tries = 0; }
      if (phase == PH_DIR) {
        iters = iter;
        const R E0 = kkt_error(E, R(0.0));
        if (!(E0 == E0)) return MPC_STATUS_NUMERIC;
        /* as far as this precision is asked to go -- or an instance that is taking long: the stragglers (steps of a few per
         * cent against a bound for dozens of iterations) are where the noise of fp32 steps costs most; they go on in fp64 */
        if ((kCanPromote && promote_mu > R(0.0)) && (E0 <= tol || iter >= promote_cap)) { promote_clean = E0 <= tol; return MPC_PROMOTE; }
        if (converged_at(E, E0)) {
          /* IPOPT's own rule stops here.  Termination polish (MpcParams.polish, include/mpc_amd.h): carry on with
           * Newton steps at the final barrier parameter until the outputs (delta_0, a_0) have stopped moving, so
           * that the point returned is the central-path point itself and not whichever iterate crossed tol first
           * (an interior a_0 still moves by ~R(1e-4) per step there).  At most kMaxPolish extra iterations. */
          if (!P.polish || n_polish >= kMaxPolish || iter >= P.max_iter || (mu <= mu_floor && out_step <= out_tol))
            return MPC_STATUS_SUCCESS;
          ++n_polish;
          if (mu > mu_floor) { mu = mu_floor; tau = mpc_max(IC::tau_min, R(1.0) - mu); nf = 0; }
        } else if (acceptable_at(E, E0)) {
          /* IPOPT's acceptable-level termination: acceptable_iter (15) acceptable iterates in a row end the solve with
           * STOP_AT_ACCEPTABLE_POINT (tested before the iteration cap, as IPOPT does) */
          if (++acc_count >= P.acceptable_iter) return MPC_STATUS_ACCEPTABLE;
        } else acc_count = 0;
        if (iter >= P.max_iter) return MPC_STATUS_MAXITER;
        /* barrier update, W&B eq. (7) */
        while (kkt_error(E, mu) <= IC::kappa_eps * mu && mu > mu_floor) {
          mu = mpc_max(mu_floor, mpc_min(IC::kappa_mu * mu, mu * mpc_sqrt(mu)));
          /* with the termination polish the solve ends at the floor anyway: a value within 3x of it (IPOPT's schedule
           * lands on R(2.4e-9) before R(1e-9)) goes there directly, which saves most instances one iteration */
          if (P.polish && mu < R(3.0) * mu_floor) mu = mu_floor;
          tau = mpc_max(IC::tau_min, R(1.0) - mu);
          nf = 0;
        }
        if ((kCanPromote && promote_mu > R(0.0)) && mu <= promote_mu) { promote_clean = true; return MPC_PROMOTE; }   /* the barrier problems below are the other solver's */
#if defined(MPC_TRACE) && !defined(__HIP_DEVICE_COMPILE__)
        printf("it %3d f=%.8g theta=%.3e dinf=%.3e cmin=%.2e cmax=%.2e mu=%.2e E0=%.3e nf=%d\n", iter, E.f, E.theta,
               E.dinf, E.cmin, E.cmax, mu, E0, nf);
#endif
      }
      lsm = (phase == PH_LS);
      /* search direction with inertia correction, W&B section 3.1: ONE backward sweep per pass; the wrong inertia sends the
       * instance round again with the next regularisation (phase REG) */
      R dw = dw_try;
      bool okb = true;
      if (!backward(dw)) {
        if (lsm) okb = false;
        else {
          if (dw == R(0.0)) dw = (dw_last == R(0.0)) ? IC::dw_0 : mpc_max(IC::dw_min, IC::kw_minus * dw_last);
          else dw *= (dw_last == R(0.0)) ? IC::kw_plus_bar : IC::kw_plus;
          if (dw > IC::dw_max || ++reg_tries > 100) { promote_clean = false; return (kCanPromote && promote_mu > R(0.0)) ? (int)MPC_PROMOTE : (int)MPC_STATUS_LINESEARCH; }
          dw_try = dw; phase = PH_REG;
          return MPC_RUNNING;
        }
      }
      if (phase == PH_REG) phase = PH_DIR;
      if (okb) forward();
      dw_cur = dw;
      if (phase == PH_LS) {
        if (!okb) { lsm = false; ls_start = false; phase = PH_EVAL0; return MPC_RUNNING; }
        alpha = R(0.0); alpha_l = R(1.0); alpha_z = R(0.0);      /* lam <- lam_LS; primal point and bound duals unchanged */
      } else {
        if (dw > R(0.0)) { dw_last = dw; n_reg++; }
        /* filter line search, W&B algorithm A */
        theta_k = E.theta; phi_k = df * E.f - mu * E.L;
        pth = hpow(theta_k, IC::s_theta);                       /* theta^s_theta */
        pdp = (dphi < R(0.0)) ? hpow(-dphi, IC::s_phi) : R(0.0);      /* (-dphi)^s_phi */
        if (dphi < R(0.0)) {
          const R t3 = (theta_k <= theta_min) ? IC::delta_sw * pth / pdp : IC::gamma_theta;
          amin = IC::gamma_alpha * mpc_min(mpc_min(IC::gamma_theta, IC::gamma_phi * theta_k / (-dphi)), t3);
        } else amin = IC::gamma_alpha * IC::gamma_theta;
        tiny = dxinf <= R(10.0) * IC::eps * mpc_max(R(1.0), xinf);
        alpha = amax; alpha_l = amax; alpha_z = az;
      }
    }
    R lmax;
    const EvalR T = costate_trial(dw_cur, alpha, alpha_l, alpha_z, phase != PH_EVAL0, lmax);
    if (phase == PH_EVAL0) {
      E = T; cur = 1 - cur;                       /* (a fresh start evaluates slot 0 into slot 1) */

      if (!E.ok) return MPC_STATUS_NUMERIC;
      if (!keep_theta) { theta_max = R(1e4) * mpc_max(R(1.0), E.theta); theta_min = R(1e-4) * mpc_max(R(1.0), E.theta); }
      keep_theta = false;
      phase = ls_start ? PH_LS : PH_DIR;
      return MPC_RUNNING;
    }
    if (phase == PH_LS) {
      lsm = false; ls_start = false;
      /* estimates above constr_mult_init_max = 1000 are discarded: the start point is then evaluated as it is */
      if (lmax <= R(1000.0)) {
        E = T; cur = 1 - cur;

        if (!E.ok) return MPC_STATUS_NUMERIC;
        theta_max = R(1e4) * mpc_max(R(1.0), E.theta); theta_min = R(1e-4) * mpc_max(R(1.0), E.theta);
        phase = PH_DIR;
      } else phase = PH_EVAL0;
      return MPC_RUNNING;
    }
    /* acceptance test of the line search */
    bool accepted = false, ftype = false;
    if (tiny) { accepted = T.ok; ftype = true; }
    else if (T.ok) {
      const R phi_t = df * T.f - mu * T.L;
      const R eps_phi = R(10.0) * IC::eps * mpc_abs(phi_k);
      if (T.theta < theta_max && !filter_rejects(T.theta, phi_t)) {
        const bool sw = dphi < R(0.0) && alpha * pdp > IC::delta_sw * pth;
        const bool armijo = phi_t - phi_k - eps_phi <= IC::eta_phi * alpha * dphi;
        if (theta_k <= theta_min && sw) {
          if (armijo) accepted = true;
        } else if (T.theta <= (R(1.0) - IC::gamma_theta) * theta_k ||
                   phi_t - phi_k - eps_phi <= -IC::gamma_phi * theta_k) {
          accepted = true;
        }
        ftype = sw && armijo; /* the filter is augmented unless both hold (W&B step A-7) */
      }
    }
    /* A polish step must not cost what has been reached.  fp64: a step that leaves tol is DROPPED and the converged iterate
     * returned (the oracle does the same).  fp32: a step that throws the point out of the acceptable band (a slack of a few
     * ulp collapsing) is not taken at that length: the line search shortens it, and if no length works the converged iterate
     * is returned as it is (line_search_failed). */
    if (accepted && n_polish > 0 && !tiny) {
      const R Et = kkt_error(T, R(0.0));
      if (sizeof(R) == 8) { if (!converged_at(T, Et)) return MPC_STATUS_SUCCESS; }
      else if (!(Et <= R(10.0) * tol)) accepted = false;
    }
    if (accepted) {
      if (!ftype) filter_add((R(1.0) - IC::gamma_theta) * theta_k, phi_k - IC::gamma_phi * theta_k);
      cur = 1 - cur;
      E = T;
#if MPC_S0_VARIABLE
      p0 = trial_x0(p0, st[2], alpha); v0k = trial_x0(v0k, st[3], alpha);
#endif
      /* fp32: the outputs must have been still for two steps in a row (steps are noisy and can be short for other reasons) */
      /* what the polish watches: the step of the outputs (delta_0, a_0), and 0.03 x the largest step of any primal variable
       * -- the predicted trajectory, whose far end is the least determined part of the solution, has then moved by less than
       * out_step_tol / 0.03 = 1e-5 m */
      const R ostep = mpc_max(alpha * T.du0, R(0.03) * alpha * dxinf);
      out_step = sizeof(R) == 4 ? mpc_max(ostep, out_prev) : ostep;
      out_prev = ostep;
      ++iter;
      phase = PH_DIR;
      return MPC_RUNNING;
    }
    if (tiny) return line_search_failed();
    alpha *= R(0.5); alpha_l = alpha;
    if (alpha < amin) return line_search_failed();
    phase = PH_BACKTRACK;
    return MPC_RUNNING;
  }

  /* MPC.cpp:306-324: out9 and the optional N-point trajectory */
  /* yaw_lo / yaw_hi: the caller's psi-bounds of this instance (the solver itself holds the relaxed ones): with
   * honor_original_bounds (IPOPT 3.12 default "yes") the returned point is projected into the bounds the user gave */
  template <class OutF, class TrajF>
  MPC_HD void unpack(OutF out, TrajF traj, bool want_traj, R yaw_lo, R yaw_hi) const {
    const int I = it(cur);
    MPC_UNROLL
    for (int i = 0; i < 6; i++) out(i) = ws.it(0, I, F_S + i);
    out(6) = ws.it(0, I, F_U + 0);
    out(7) = ws.it(0, I, F_U + 1);
    if (P.honor_original_bounds) {
      const R psi1 = ws.it(0, I, F_S + 2), v1 = ws.it(0, I, F_S + 3), d0 = ws.it(0, I, F_U + 0), a0 = ws.it(0, I, F_U + 1);
      /* written so that a NaN passes through unchanged */
      out(2) = psi1 < yaw_lo ? yaw_lo : (psi1 > yaw_hi ? yaw_hi : psi1);
      out(3) = v1 < (R)-P.max_speed ? (R)-P.max_speed : (v1 > (R)P.max_speed ? (R)P.max_speed : v1);
      out(6) = d0 < (R)-P.max_steering ? (R)-P.max_steering : (d0 > (R)P.max_steering ? (R)P.max_steering : d0);
      out(7) = a0 < (R)P.max_deceleration ? (R)P.max_deceleration : (a0 > (R)P.max_acceleration ? (R)P.max_acceleration : a0);
    }
    out(8) = E.f + cost0;
    if (want_traj) {
      const int N = P.N;
      traj(0) = st[0]; traj(N) = st[1];
      for (int k = 0; k < M; ++k) { traj(k + 1) = ws.it(k, I, F_S + 0); traj(N + k + 1) = ws.it(k, I, F_S + 1); }
    }
  }
};

/* The iterate record of one stage carried from one precision's layout to the other's (s 6, u 2, lam 6, bound duals 4 + 4;
 * the fp32 record pads the multipliers): get(f) reads field f of the source record, put(f, v) writes the destination's. */
template <class RS, class RD, class Get, class Put>
MPC_HD void convert_iterate_record(Get get, Put put) {
  using FS = Fields<RS>;
  using FD = Fields<RD>;
  MPC_UNROLL
  for (int i = 0; i < 6; i++) { put(FD::F_S + i, (RD)get(FS::F_S + i)); put(FD::F_LAM + i, (RD)get(FS::F_LAM + i)); }
  put(FD::F_U, (RD)get(FS::F_U)); put(FD::F_U + 1, (RD)get(FS::F_U + 1));
  MPC_UNROLL
  for (int b = 0; b < 4; b++) { put(FD::F_ZL + b, (RD)get(FS::F_ZL + b)); put(FD::F_ZU + b, (RD)get(FS::F_ZU + b)); }
}

/* One instance, end to end (used by the test-only host build; the device kernel
 * drives Solver directly so that outputs go straight to their HBM arrays). */
template <class WS, class R>
MPC_HD int solve_instance(const MpcParams &P, WS ws, const R *state6, const R *coef5, R yaw_lo,
                          R yaw_hi, const R *w12, R *out9, R *traj2N, int *iters_out) {
  Solver<WS, R> S(P, ws);
  int status = S.setup(state6, coef5, yaw_lo, yaw_hi, w12);
  if (status == MPC_STATUS_SUCCESS) status = S.solve();
  R *o = out9;
  R *t = traj2N;
  S.unpack([o](int i) -> R & { return o[i]; }, [t](int i) -> R & { return t[i]; }, traj2N != nullptr, yaw_lo, yaw_hi);
  if (iters_out) *iters_out = S.iters;
  return status;
}

}  // namespace mpc
#endif /* MPC_CORE_H */
