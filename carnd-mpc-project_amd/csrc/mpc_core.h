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
/* A "field" is one double per instance.  Stage k (0..N-2) owns:
 *   two iterate slots: s_{k+1} (6), u_k (2), lam_{k+1} (6), bound duals of (psi_{k+1}, v_{k+1}, delta_k, a_k) (4+4)
 *   the Newton direction  ds_{k+1} (6), du_k (2), dlam_{k+1} (6)
 *   the Riccati gains     K_k (2x6) and kff_k (2)   -- in LDS on the device when the launch allows it
 * Nothing else is kept: the stage model (sin/cos/atan, road polynomial, residual) is recomputed
 * in every sweep, because the kernel is limited by workspace traffic, not by arithmetic.
 * Everything is fp64: storing the direction or the gains in fp32 was tried and rejected -- an
 * absolute error of ~1e-8 in a step component is fatal next to slacks of ~1e-9 at active bounds
 * (+11 % iterations and a few non-converged instances on the 65 536-instance workload). */
enum : int {
  F_S = 0, F_U = 6, F_LAM = 8, F_ZL = 14, F_ZU = 18, IT_SZ = 22,   /* one iterate slot */
  IT0 = 0, IT1 = IT_SZ,                                            /* double-buffered iterate */
  F_D = 2 * IT_SZ, D_N = 8,                                        /* direction (ds, du) */
  STAGE_SZ_GLOBAL = F_D + D_N,                                     /* 52 fields per stage */
  /* The Riccati gains K (2x6) and kff (2) live only between the backward and the forward sweep of one pass, while
   * the iterate slot that is not the current one holds nothing (the next trial point is written there afterwards):
   * they are stored in the first 14 fields of that slot.  The workspace is a fifth smaller for it, and the
   * workspace stream does live in the caches (bypassing them costs 40 %). */
  F_GK = 0, GK_N = 12, F_GF = F_GK + GK_N, GF_N = 2
};
static_assert(GK_N + GF_N <= IT_SZ, "the gains must fit an iterate slot");
enum : int { D_S = 0, D_U = 6 };                                   /* direction entries */

MPC_HD int64_t workspace_fields_per_instance(int N, bool) { return (int64_t)(N - 1) * STAGE_SZ_GLOBAL; }

/* Staging interface (see TiledWorkspace): a sweep asks for the record of the NEXT stage while it works on the
 * current one.  stage_fetch_it copies the IT_SZ fields of an iterate slot of stage k to the front of buffer `buf`,
 * stage_fetch_g / stage_fetch_d the gains or the direction behind it; sit()/sg()/sx() read them back; stage_wait<N>()
 * waits until at most the N most recent copy/store instructions are still in flight.  On the host build all of
 * this degenerates to direct reads. */
enum : int { STG_IT_OPS = IT_SZ / 2, STG_ITF_OPS = 8, STG_X_OPS = 7, STG_D_OPS = D_N / 2, STG_SLOT_PAIRS = (IT_SZ + 14) / 2 };
/* pair stores a stage issues in each sweep (all through store2): the counted waits let exactly these
 * stay in flight besides the newest copy group */
enum : int { ST_BACKWARD = (GK_N + GF_N) / 2, ST_FORWARD = 4, ST_TRIAL = IT_SZ / 2 };

/* Plain storage for the test-only host build: one instance, fields contiguous. */
struct HostWorkspace {
  double *base;
  /* field f of iterate slot I (or I = 0 and an absolute field) of stage k */
  MPC_HD double &it(int k, int I, int f) const { return base[k * STAGE_SZ_GLOBAL + I + f]; }
  MPC_HD double getD(int k, int j) const { return base[k * STAGE_SZ_GLOBAL + F_D + j]; }
  MPC_HD void setD(int k, int j, double v) const { base[k * STAGE_SZ_GLOBAL + F_D + j] = v; }
  MPC_HD void store2(int k, int I, int f, double a, double b) const { base[k * STAGE_SZ_GLOBAL + I + f] = a; base[k * STAGE_SZ_GLOBAL + I + f + 1] = b; }
  MPC_HD void stage_fetch_it(int, int, int) const {}
  MPC_HD void stage_fetch_itf(int, int, int) const {}
  MPC_HD void stage_fetch_d(int, int) const {}
  template <int N> MPC_HD void stage_wait() const {}
  MPC_HD void stage_drain() const {}
  MPC_HD double sit(int, int k, int I, int j) const { return base[k * STAGE_SZ_GLOBAL + I + j]; }
  MPC_HD double sx(int, int k, int F, int j) const { return base[k * STAGE_SZ_GLOBAL + F + j]; }
  MPC_HD void stage_fetch_g(int, int, int) const {}
  MPC_HD double sg(int, int k, int J, int j) const { return base[k * STAGE_SZ_GLOBAL + J + j]; }
};

#if defined(__HIPCC__)
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) char gchar;
typedef __attribute__((address_space(3))) double ldouble;
typedef __attribute__((address_space(3))) char lchar;
#else   /* host pass of hipcc: the kernel body is parsed but never run */
typedef double gdouble;
typedef char gchar;
typedef double ldouble;
typedef char lchar;
#endif
/* Device layout: the workspace is tiled per wavefront and, inside a tile, fields are interleaved in
 * PAIRS per lane:  [wave][stage][field pair][64 lanes][2].  One wave's whole working set is one
 * contiguous block (N=10: 72 x 9 x 512 B = 324 KB), and the two fields of a pair of one instance are 16
 * contiguous bytes, so
 *   - a pair moves with one 16-byte-per-lane access (global_load/store_dwordx4: the coalescing sweet spot),
 *   - `global_load_lds_dwordx4` (LDS-DMA) moves each lane's OWN data, which keeps it correct under the
 *     partial exec masks of a wave whose instances are in different solver phases.
 * Each sweep double-buffers the next stage's record into LDS with LDS-DMA while it computes the current
 * stage: the kernel needs all 512 registers (one wave per SIMD), so nothing else can hide the
 * HBM / Infinity-Cache latency, and a register prefetch does not fit (it spilled 428 VGPRs).
 * LDS: 2 buffers x 36 fields x 512 B = 36 KB per wave, 144 KB per CU at four waves.
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
template <bool STAGING>
struct TiledWorkspace {
  gdouble *tile;   /* this wave's tile */
  ldouble *lbuf;   /* LDS staging area of this wave (STAGING) */
  int lane;
  static constexpr unsigned STAGE = STAGE_SZ_GLOBAL;
  static constexpr unsigned PAIRS = STAGE_SZ_GLOBAL / 2;
  /* uniform part: row of (stage k, field f) + position inside the pair; vector part: lane and slot */
  MPC_HD gchar *row(int k, int f) const {
    return (gchar *)tile + MPC_UNIFORM(((unsigned)k * PAIRS + ((unsigned)f >> 1)) * 1024u + ((unsigned)f & 1u) * 8u);
  }
  MPC_HD unsigned voff(int I) const { return (unsigned)lane * 16u + ((unsigned)I >> 1) * 1024u; }
  MPC_HD gdouble &it(int k, int I, int f) const { return *(gdouble *)(row(k, f) + voff(I)); }
  /* both fields of a pair (f even) with ONE 16-byte store: the number of store instructions per stage is
   * then exact, which the counted waits of the staged sweeps rely on */
  MPC_HD void store2(int k, int I, int f, double a, double b) const {
    typedef double __attribute__((ext_vector_type(2))) d2;
    typedef __attribute__((address_space(1))) d2 gd2;
    d2 v; v.x = a; v.y = b;
    *(gd2 *)(row(k, f) + voff(I)) = v;
  }
  MPC_HD double getD(int k, int j) const { return it(k, 0, F_D + j); }
  MPC_HD void setD(int k, int j, double v) const { it(k, 0, F_D + j) = v; }
  /* ---- staging ---- */
  /* Copies NPAIRS consecutive pairs.  Rows are 1 KB apart in the tile AND in the LDS slot, and the
   * instruction's immediate offset applies to both addresses, so four copies share one scalar row base and
   * one M0 value (immediates 0, 1024, 2048, 3072).  Written as assembly because the compiler expands the
   * builtin's offset argument back into per-copy address arithmetic (5 issue slots per copy instead of <2).
   * The compiler does not see these as memory instructions; that only makes its own vmcnt waits more
   * conservative (vmcnt completes in order), and the sweeps order everything staged with explicit waits. */
  template <int NPAIRS>
  MPC_HD void dma(int buf, int k, int I, int f0, int dst_pair) const {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned vo = voff(I);
    MPC_UNROLL
    for (int q0 = 0; q0 < NPAIRS; q0 += 4) {
      const unsigned m0v = MPC_UNIFORM((unsigned)(unsigned long)lbuf + (((unsigned)buf * STG_SLOT_PAIRS + (unsigned)dst_pair + (unsigned)q0) * 64u) * 16u);
      const gchar *src = row(k, f0 + 2 * q0);
      constexpr int n = (NPAIRS - 0);
      if (q0 + 4 <= n)
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072"
                     :: "v"(vo), "s"(src), "s"(m0v) : "memory");
      else if (q0 + 3 == n)
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:2048"
                     :: "v"(vo), "s"(src), "s"(m0v) : "memory");
      else if (q0 + 2 == n)
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024"
                     :: "v"(vo), "s"(src), "s"(m0v) : "memory");
      else
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(vo), "s"(src), "s"(m0v) : "memory");
    }
#endif
  }
  MPC_HD void stage_fetch_it(int buf, int k, int I) const { if (STAGING) dma<STG_IT_OPS>(buf, k, I, 0, 0); }
  /* the forward sweep does not read the multipliers: (s, u) and the bound duals go to their usual places */
  MPC_HD void stage_fetch_itf(int buf, int k, int I) const {
    if (STAGING) { dma<4>(buf, k, I, F_S, F_S / 2); dma<4>(buf, k, I, F_ZL, F_ZL / 2); }
  }
  MPC_HD void stage_fetch_d(int buf, int k) const { if (STAGING) dma<STG_D_OPS>(buf, k, 0, F_D, STG_IT_OPS); }
  /* the gains, in the first fields of iterate slot J (the one that is not current) */
  MPC_HD void stage_fetch_g(int buf, int k, int J) const { if (STAGING) dma<STG_X_OPS>(buf, k, J, F_GK, STG_IT_OPS); }
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
  MPC_HD double sl(int buf, int j) const {
    return lbuf[(((unsigned)buf * STG_SLOT_PAIRS + ((unsigned)j >> 1)) * 64u + (unsigned)lane) * 2u + ((unsigned)j & 1u)];
  }
  MPC_HD double sit(int buf, int k, int I, int j) const { return STAGING ? sl(buf, j) : (double)it(k, I, j); }
  MPC_HD double sx(int buf, int k, int F, int j) const { return STAGING ? sl(buf, IT_SZ + j) : (double)it(k, 0, F + j); }
  MPC_HD double sg(int buf, int k, int J, int j) const { return STAGING ? sl(buf, IT_SZ + j) : (double)it(k, J, F_GK + j); }
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

/* IPOPT default constants (Waechter & Biegler 2006; IPOPT 3.12 option defaults) */
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
  static constexpr double kappa_eps = MPC_KAPPA_EPS, kappa_mu = MPC_KAPPA_MU, theta_mu = 1.5, tau_min = MPC_TAU_MIN, s_max = 100.0;
  static constexpr double gamma_theta = 1e-5, gamma_phi = 1e-8, delta_sw = 1.0, s_theta = 1.1, s_phi = 2.3;
  static constexpr double eta_phi = 1e-8, gamma_alpha = 0.05, kappa_sigma = 1e10, kappa1 = 1e-2, kappa2 = 1e-2;
  static constexpr double dw_min = 1e-20, dw_0 = 1e-4, dw_max = 1e40, kw_minus = 1.0 / 3.0, kw_plus = 8.0;
  static constexpr double kw_plus_bar = 100.0, mu_init = MPC_MU_INIT, eps = 2.220446049250313e-16;
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
struct Eval {
  double theta;  /* ||c||_1            */
  double cinf;   /* ||c||_inf          */
  double f;      /* unscaled objective without the stage-0 constant */
  double L;      /* sum of log(slack)  */
  double dinf;   /* ||grad_x Lagrangian||_inf (scaled objective) */
  double cmin, cmax; /* range of slack*dual products */
  double lsum, zsum; /* ||lam||_1, ||z||_1 */
  double du0;        /* max(|d delta_0|, |d a_0|) of the direction this trial was made with (unscaled by alpha) */
  bool ok;
};

/* linearisation of one stage at (s_k, u_k) and its residual c_{k+1} = s_{k+1} - F(s_k, u_k) */
struct Lin {
  double sp, cp, se, ce;   /* sin/cos of psi_k and epsi_k */
  double fp, g1, h3, fpp;  /* f'(x_k), f''/(1+f'^2), d/dx of that, f'' */
  double c[6];
};

template <class WS>
struct Solver {
  const MpcParams &P;
  WS ws;
  /* instance data */
  double st[6], coef[MPC_NCOEF], yl, yu;
  double wc, we, wv, wd, wdd, vref, cost0;
  /* bounds */
  double vl, vu, dl, du, al, au;
  int M;       /* number of stages = N-1 */
  double dt, dtLf, iLf, psi_start;
  /* interior-point state */
  int cur;     /* slot of the current iterate */
  double mu, tau, df;
  Eval E;
  /* direction summary */
  double amax, az, dphi, dxinf, xinf;
  /* filter: four entries in registers (it is emptied at every barrier update) */
  double fth0, fth1, fth2, fth3, fph0, fph1, fph2, fph3;
  int nf;
  int iters, n_reg;
  /* true only while the least-squares multiplier start is being computed: the sweeps then solve
   * [I J^T; J 0][w; lam] = -[grad f; 0] (identity Hessian, no barrier, zero constraint rhs) */
  bool lsm;

  MPC_HD Solver(const MpcParams &p, WS w) : P(p), ws(w) {}

  MPC_HD int it(int slot) const { return slot ? IT1 : IT0; }

  /* ---- road polynomial: RoadGeometry::centerY / orientation, utils.h:28-47 */
  MPC_HD void poly(double x, double &f, double &fp, double &fpp, double &fppp) const {
    const double c0 = coef[0], c1 = coef[1], c2 = coef[2], c3 = coef[3], c4 = coef[4];
    f = (((c4 * x + c3) * x + c2) * x + c1) * x + c0;
    fp = ((4.0 * c4 * x + 3.0 * c3) * x + 2.0 * c2) * x + c1;
    fpp = (12.0 * c4 * x + 6.0 * c3) * x + 2.0 * c2;
    fppp = 24.0 * c4 * x + 6.0 * c3;
  }

  /* Stage model at (s,u), MPC.cpp:142-152, and the residual against the successor state sn.
   * Recomputed wherever it is needed (see the layout comment). */
  MPC_HD void linearise(const double *s, double delta, double a, const double *sn, Lin &L) const {
    fsincos2(s[2], s[5], &L.sp, &L.cp, &L.se, &L.ce);
    double f, fp, fpp, fppp;
    poly(s[0], f, fp, fpp, fppp);
    const double q1 = 1.0 + fp * fp, iq1 = frcp1(q1);
    L.fp = fp;
    L.g1 = fpp * iq1;
    L.h3 = (fppp * q1 - 2.0 * fp * fpp * fpp) * (iq1 * iq1);
    L.fpp = fpp;
    const double vdt = s[3] * dt;
    const double psin = s[2] + delta * vdt * iLf;
    L.c[0] = sn[0] - (s[0] + L.cp * vdt);
    L.c[1] = sn[1] - (s[1] + L.sp * vdt);
    L.c[2] = sn[2] - psin;
    L.c[3] = sn[3] - (s[3] + a * dt);
    L.c[4] = sn[4] - ((f - s[1]) + L.se * vdt);
    L.c[5] = sn[5] - (psin - fatan(fp));
  }

  /* cost + barrier terms of one state s_k (k>=1): Hessian diagonal and gradient */
  MPC_HD void state_terms(double psi, double v, double c, double e, double zlp, double zup, double zlv,
                          double zuv, double &Hpp, double &Hvv, double &Hee, double &Hcc, double &gp,
                          double &gv, double &ge, double &gc) const {
    const double islp = frcp1(psi - yl), isup = frcp1(yu - psi), islv = frcp1(v - vl), isuv = frcp1(vu - v);
    const double mub = lsm ? 0.0 : mu;
    Hpp = lsm ? 1.0 : zlp * islp + zup * isup;
    Hvv = lsm ? 1.0 : df * 2.0 * wv + zlv * islv + zuv * isuv;
    Hee = lsm ? 1.0 : df * 2.0 * we;
    Hcc = lsm ? 1.0 : df * 2.0 * wc;
    gp = mub * (isup - islp);
    gv = df * 2.0 * wv * (v - vref) + mub * (isuv - islv);
    ge = df * 2.0 * we * e;
    gc = df * 2.0 * wc * c;
  }

  MPC_HD void load_state(int k, int I, double *s) const {   /* s_k; k = 0 is the fixed initial state */
    if (k == 0) {
      MPC_UNROLL
      for (int i = 0; i < 6; i++) s[i] = st[i];
    } else {
      MPC_UNROLL
      for (int i = 0; i < 6; i++) s[i] = ws.it(k - 1, I, F_S + i);
    }
  }

  /* ------------------------------------------------------------------ */
  /* Riccati backward sweep: gains K_k, kff_k for every stage.           */
  /* Returns false when some R~_k is not positive definite (wrong        */
  /* inertia): the caller raises the regularisation dw and repeats.      */
  /* ------------------------------------------------------------------ */
  MPC_HD bool backward(double dw) {
    const int I = it(cur), J = it(1 - cur);         /* J: where the gains go */
    /* value function of (x,y,psi,v,e,d) [+ c] at stage k+1; only the lower triangle of the symmetric
     * matrices is ever written or read (PM/MX pick it), so the other half never occupies registers */
    double Pm[6][6], p[6], Pcc, pc;
#define PM(i, j) Pm[(i) >= (j) ? (i) : (j)][(i) >= (j) ? (j) : (i)]
#define MX(i, j) Mx[(i) >= (j) ? (i) : (j)][(i) >= (j) ? (j) : (i)]
    MPC_UNROLL
    for (int i = 0; i < 6; i++) {
      p[i] = 0;
      MPC_UNROLL
      for (int j = 0; j < 6; j++) Pm[i][j] = 0;
    }
    const double rsc = lsm ? 0.0 : -1.0;             /* constraint right-hand side: -c, or 0 for the LS system */
    const double hxy = (lsm ? 1.0 : 0.0) + dw;       /* x and y carry no cost: only the LS identity / regularisation */
    /* Staging: record j of the iterate (the fields of stage j) sits in buffer (M-1-j)&1.  Stage k needs
     * (u_k, lam_{k+1}, duals of u_k) from record k -- moved to registers one iteration earlier -- and
     * (s_k, delta_{k-1}, duals of s_k) from record k-1; record k-2 is requested meanwhile. */
    ws.stage_drain();
    ws.stage_fetch_it(0, M - 1, I);
    ws.template stage_wait<0>();
    double sn[6];                                    /* s_{k+1} */
    MPC_UNROLL
    for (int i = 0; i < 6; i++) sn[i] = ws.sit(0, M - 1, I, F_S + i);
    {
      double Hpp, Hvv, Hee, Hcc, gp, gv, ge, gc;
      state_terms(sn[2], sn[3], sn[4], sn[5], ws.sit(0, M - 1, I, F_ZL + 0), ws.sit(0, M - 1, I, F_ZU + 0),
                  ws.sit(0, M - 1, I, F_ZL + 1), ws.sit(0, M - 1, I, F_ZU + 1), Hpp, Hvv, Hee, Hcc, gp, gv, ge, gc);
      Pm[0][0] = hxy; Pm[1][1] = hxy; Pm[2][2] = Hpp + dw; Pm[3][3] = Hvv + dw; Pm[4][4] = Hee + dw;
      Pcc = Hcc + dw; p[2] = gp; p[3] = gv; p[4] = ge; pc = gc;
    }
    /* inputs of stage k that live in record k, carried in registers */
    double delta = ws.sit(0, M - 1, I, F_U + 0), acc = ws.sit(0, M - 1, I, F_U + 1);
    double lx = ws.sit(0, M - 1, I, F_LAM + 0), ly = ws.sit(0, M - 1, I, F_LAM + 1), lp = ws.sit(0, M - 1, I, F_LAM + 2);
    double lc = ws.sit(0, M - 1, I, F_LAM + 4), le = ws.sit(0, M - 1, I, F_LAM + 5);
    double zld = ws.sit(0, M - 1, I, F_ZL + 2), zud = ws.sit(0, M - 1, I, F_ZU + 2);
    double zla = ws.sit(0, M - 1, I, F_ZL + 3), zua = ws.sit(0, M - 1, I, F_ZU + 3);
    if (M >= 2) ws.stage_fetch_it(1, M - 2, I);
    MPC_STAGE_LOOP
    for (int k = M - 1; k >= 0; --k) {
      /* ---- inputs of stage k ---- */
      double sk[6];
      double zlp = 0, zup = 0, zlv = 0, zuv = 0, delprev = 0;
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
      }
      const double v = sk[3];
      Lin L;
      linearise(sk, delta, acc, sn, L);
      const double sp = L.sp, cp = L.cp, se = L.se, ce = L.ce, fp = L.fp, g1 = L.g1, h3 = L.h3, fpp = L.fpp;
      const double r0 = rsc * L.c[0], r1 = rsc * L.c[1], r2 = rsc * L.c[2], r3 = rsc * L.c[3], rc = rsc * L.c[4], r4 = rsc * L.c[5];
      const double vdt = v * dt;
      const double Axp = -vdt * sp, Axv = dt * cp, Ayp = vdt * cp, Ayv = dt * sp, Apv = delta * dtLf;
      const double Acx = fp, Acv = dt * se, Ace = vdt * ce, Aex = -g1, Bp = v * dtLf;
      /* t = p + P r (the d component of r is zero) */
      double t[6];
      MPC_UNROLL
      for (int i = 0; i < 6; i++)
        t[i] = p[i] + PM(i, 0) * r0 + PM(i, 1) * r1 + PM(i, 2) * r2 + PM(i, 3) * r3 + PM(i, 4) * r4;
      const double tc = pc + Pcc * rc;
      /* G^T applied to a (6-vector, c-scalar): outputs for inputs x,y,psi,v,e,delta,a */
#define MPC_GT(w, wcs, o)                                                         \
  do {                                                                            \
    const double w24_ = (w)[2] + (w)[4];                                          \
    (o)[0] = (w)[0] + Aex * (w)[4] + Acx * (wcs);                                 \
    (o)[1] = (w)[1] - (wcs);                                                      \
    (o)[2] = Axp * (w)[0] + Ayp * (w)[1] + w24_;                                  \
    (o)[3] = Axv * (w)[0] + Ayv * (w)[1] + Apv * w24_ + (w)[3] + Acv * (wcs);     \
    (o)[4] = Ace * (wcs);                                                         \
    (o)[5] = Bp * w24_ + (w)[5];                                                  \
    (o)[6] = dt * (w)[3];                                                         \
  } while (0)
      double qt[7];
      MPC_GT(t, tc, qt);
      /* the stage's own control terms */
      const double isld = frcp1(delta - dl), isud = frcp1(du - delta), isla = frcp1(acc - al), isua = frcp1(au - acc);
      double ddl = 0, Hdd = 0;
      if (k >= 1 && !lsm) { ddl = delta - delprev; Hdd = df * 2.0 * wdd; }   /* LS start: all delta are 0 */
      const double mub = lsm ? 0.0 : mu;
      const double gdel = df * 2.0 * wd * delta + Hdd * ddl + mub * (isud - isld);
      const double gacc = mub * (isua - isla);
      const double rt_d = qt[5] + gdel, rt_a = qt[6] + gacc;
      /* control Hessian diagonal: cost + barrier, or the identity of the LS system */
      const double Sgd = lsm ? 1.0 : df * 2.0 * wd + zld * isld + zud * isud, Sga = lsm ? 1.0 : zla * isla + zua * isua;
      double w5[6], w6[6];
      MPC_UNROLL
      for (int i = 0; i < 6; i++) { w5[i] = Bp * (PM(i, 2) + PM(i, 4)) + PM(i, 5); w6[i] = dt * PM(i, 3); }
      if (k == 0) {
        /* only the feed-forward of u_0 is needed (ds_0 = 0) */
        double o5[7], o6[7];
        MPC_GT(w5, 0.0, o5);
        MPC_GT(w6, 0.0, o6);
        const double Rdd = o5[5] + Sgd + dw;
        const double Rda = o6[5];
        const double Raa = o6[6] + Sga + dw;
        const double det = Rdd * Raa - Rda * Rda;
        if (!(Rdd > 0.0) || !(det > 0.0)) return false;
        const double idet = frcp1(det);
        ws.store2(0, J, F_GF, -(Raa * rt_d - Rda * rt_a) * idet, -(-Rda * rt_d + Rdd * rt_a) * idet);
        break;
      }
      /* ---- W = P G (columns for inputs x,y,psi,v,e,delta,a) and Mx = G^T W ---- */
      double Mx[7][7];
      {
        double w[6], o[7];
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
        MPC_GT(w, 0.0, o);
        MPC_UNROLL
        for (int i = 0; i < 7; i++) Mx[i][2] = o[i];
        MPC_UNROLL
        for (int i = 0; i < 6; i++) w[i] = Axv * PM(i, 0) + Ayv * PM(i, 1) + Apv * (PM(i, 2) + PM(i, 4)) + PM(i, 3);
        MPC_GT(w, Pcc * Acv, o);
        MPC_UNROLL
        for (int i = 0; i < 7; i++) Mx[i][3] = o[i];
        {
          /* input e only feeds cte+: column = G^T (0, Pcc*Ace) */
          const double wcs = Pcc * Ace;
          Mx[0][4] = Acx * wcs; Mx[1][4] = -wcs; Mx[2][4] = 0.0; Mx[3][4] = Acv * wcs; Mx[4][4] = Ace * wcs;
          Mx[5][4] = 0.0; Mx[6][4] = 0.0;
        }
        MPC_GT(w5, 0.0, o);
        MPC_UNROLL
        for (int i = 0; i < 7; i++) Mx[i][5] = o[i];
        MPC_GT(w6, 0.0, o);
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
      const double Rdd = Mx[5][5] + Hdd + Sgd + dw;
      const double Rda = Mx[6][5];
      const double Raa = Mx[6][6] + Sga + dw;
      const double det = Rdd * Raa - Rda * Rda;
      if (!(Rdd > 0.0) || !(det > 0.0)) return false;
      const double idet = frcp1(det);
      const double i11 = Raa * idet, i12 = -Rda * idet, i22 = Rdd * idet;
      /* S~ (2 x 6 over x,y,psi,v,e,d) */
      double Sd[6], Sa[6], Kd[6], Ka[6];
      MPC_UNROLL
      for (int j = 0; j < 5; j++) { Sd[j] = Mx[5][j]; Sa[j] = Mx[6][j]; }
      Sd[5] = -Hdd; Sa[5] = 0.0;
      MPC_UNROLL
      for (int j = 0; j < 6; j++) {
        Kd[j] = -(i11 * Sd[j] + i12 * Sa[j]);
        Ka[j] = -(i12 * Sd[j] + i22 * Sa[j]);
      }
      const double kfd = -(i11 * rt_d + i12 * rt_a), kfa = -(i12 * rt_d + i22 * rt_a);
      MPC_UNROLL
      for (int j = 0; j < 6; j += 2) { ws.store2(k, J, F_GK + j, Kd[j], Kd[j + 1]); ws.store2(k, J, F_GK + 6 + j, Ka[j], Ka[j + 1]); }
      ws.store2(k, J, F_GF, kfd, kfa);
      /* ---- value function of stage k ---- */
      double Hpp, Hvv, Hee, Hcc, gp, gv, ge, gc;
      state_terms(sk[2], v, sk[4], sk[5], zlp, zup, zlv, zuv, Hpp, Hvv, Hee, Hcc, gp, gv, ge, gc);
      MPC_UNROLL
      for (int i = 0; i < 6; i++) {
        MPC_UNROLL
        for (int j = 0; j <= i; j++) {
          double q = (i < 5) ? Mx[i][j] : ((j == 5) ? Hdd : 0.0);
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
  MPC_HD void forward() {
    const int I = it(cur), J = it(1 - cur);
    double d0 = 0, d1 = 0, d2 = 0, d3 = 0, d5 = 0; /* ds_k: x,y,psi,v,(c),e */
    double ddprev = 0, delprev = 0;                 /* d(delta_{k-1}), delta_{k-1} */
    double rmax = 0.0, rzmax = 0.0;                 /* largest step ratios: alpha = min(1, tau / ratio) */
    const double rsc = lsm ? 0.0 : 1.0;
    dphi = 0.0; dxinf = 0.0; xinf = 0.0;
    double sk[6];
    load_state(0, I, sk);
    /* staging: stage k's iterate record and gains in buffer k&1, stage k+1 requested meanwhile */
    ws.stage_drain();
    ws.stage_fetch_itf(0, 0, I);
    ws.stage_fetch_g(0, 0, J);
    MPC_STAGE_LOOP
    for (int k = 0; k < M; ++k) {
      const int bf = k & 1;
      if (k + 1 < M) {
        ws.stage_fetch_itf(bf ^ 1, k + 1, I);
        ws.stage_fetch_g(bf ^ 1, k + 1, J);
        if (k == 0) ws.template stage_wait<STG_ITF_OPS + STG_X_OPS>();
        else ws.template stage_wait<STG_ITF_OPS + STG_X_OPS + ST_FORWARD>();
      } else ws.template stage_wait<0>();
      double sn[6];
      MPC_UNROLL
      for (int i = 0; i < 6; i++) sn[i] = ws.sit(bf, k, I, F_S + i);
      const double v = sk[3];
      const double delta = ws.sit(bf, k, I, F_U + 0), acc = ws.sit(bf, k, I, F_U + 1);
      Lin L;
      linearise(sk, delta, acc, sn, L);
      double dd = ws.sg(bf, k, J, GK_N + 0), da = ws.sg(bf, k, J, GK_N + 1);
      if (k > 0) {
        dd += ws.sg(bf, k, J, 0) * d0 + ws.sg(bf, k, J, 1) * d1 + ws.sg(bf, k, J, 2) * d2 + ws.sg(bf, k, J, 3) * d3 +
              ws.sg(bf, k, J, 4) * d5 + ws.sg(bf, k, J, 5) * ddprev;
        da += ws.sg(bf, k, J, 6) * d0 + ws.sg(bf, k, J, 7) * d1 + ws.sg(bf, k, J, 8) * d2 + ws.sg(bf, k, J, 9) * d3 +
              ws.sg(bf, k, J, 10) * d5 + ws.sg(bf, k, J, 11) * ddprev;
      }
      const double vdt = v * dt, Apv = delta * dtLf, Bp = v * dtLf;
      const double n0 = d0 - vdt * L.sp * d2 + dt * L.cp * d3 - rsc * L.c[0];
      const double n1 = d1 + vdt * L.cp * d2 + dt * L.sp * d3 - rsc * L.c[1];
      const double n2 = d2 + Apv * d3 + Bp * dd - rsc * L.c[2];
      const double n3 = d3 + dt * da - rsc * L.c[3];
      const double n4 = L.fp * d0 - d1 + dt * L.se * d3 + vdt * L.ce * d5 - rsc * L.c[4];
      const double n5 = -L.g1 * d0 + d2 + Apv * d3 + Bp * dd - rsc * L.c[5];
      ws.store2(k, 0, F_D + D_S + 0, n0, n1); ws.store2(k, 0, F_D + D_S + 2, n2, n3);
      ws.store2(k, 0, F_D + D_S + 4, n4, n5); ws.store2(k, 0, F_D + D_U + 0, dd, da);
      const double q2 = n2, q3 = n3, q4 = n4, q5 = n5, qd = dd, qa = da;
      /* bounded variables of this stage: psi_{k+1}, v_{k+1}, delta_k, a_k */
      const double xs[4] = {sn[2], sn[3], delta, acc};
      const double lo[4] = {yl, vl, dl, al}, hi[4] = {yu, vu, du, au};
      const double dx[4] = {q2, q3, qd, qa};
      MPC_UNROLL
      for (int b = 0; b < 4; b++) {
        const double isl = frcp1(xs[b] - lo[b]), isu = frcp1(hi[b] - xs[b]);
        const double zl = ws.sit(bf, k, I, F_ZL + b), zu = ws.sit(bf, k, I, F_ZU + b);
        rmax = fmax(rmax, fmax(-dx[b] * isl, dx[b] * isu));
        const double dzl = mu * isl - zl - zl * isl * dx[b];
        const double dzu = mu * isu - zu + zu * isu * dx[b];
        rzmax = fmax(rzmax, fmax(-dzl * frcp1(zl), -dzu * frcp1(zu)));
        dphi += mu * (isu - isl) * dx[b];
      }
      /* objective part of the directional derivative */
      double g = 2.0 * wc * sn[4] * q4 + 2.0 * we * sn[5] * q5 + 2.0 * wv * (sn[3] - vref) * q3 + 2.0 * wd * delta * qd;
      if (k > 0) g += 2.0 * wdd * (delta - delprev) * (qd - ddprev);
      dphi += df * g;
      dxinf = fmax(dxinf, fmax(fmax(fmax(fabs(n0), fabs(n1)), fmax(fabs(n2), fabs(n3))),
                               fmax(fmax(fabs(n4), fabs(n5)), fmax(fabs(dd), fabs(da)))));
      xinf = fmax(xinf, fmax(fmax(fabs(sn[0]), fabs(sn[1])), fmax(fabs(sn[3]), fabs(sn[4]))));
      d0 = n0; d1 = n1; d2 = n2; d3 = n3; d5 = n5; ddprev = dd; delprev = delta;
      MPC_UNROLL
      for (int i = 0; i < 6; i++) sk[i] = sn[i];
    }
    /* fraction to the boundary, W&B eq. (15): alpha = min(1, tau / max ratio) */
    amax = (rmax > tau) ? tau / rmax : 1.0;
    az = (rzmax > tau) ? tau / rzmax : 1.0;
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
  /* carried in registers from step k+1.                                  */
  /* ------------------------------------------------------------------ */
  MPC_HD Eval costate_trial(double dw, double alpha, double alpha_l, double alpha_z, bool with_costate, double &lmax) {
    const int I = it(cur), J = it(1 - cur);
    const double hxy = (lsm ? 1.0 : 0.0) + dw;
    Eval R;
    R.theta = 0; R.cinf = 0; R.f = 0; R.L = 0; R.dinf = 0; R.cmin = 1e300; R.cmax = 0; R.lsum = 0; R.zsum = 0; R.du0 = 0; R.ok = true;
    lmax = 0.0;
    const double ksm = IpmConst::kappa_sigma * mu, ksi = mu * (1.0 / IpmConst::kappa_sigma);
    /* carried from step k+1 -- current iterate: s_{k+1}, u_k, lam_{k+1}, d(delta_k), lam+_{k+1} */
    double sn_o[6] = {0, 0, 0, 0, 0, 0}, del_o = 0, acc_o = 0, lx = 0, ly = 0, lp = 0, lc = 0, le = 0, ddk = 0;
    double L0 = 0, L1 = 0, L2 = 0, L3 = 0, L4 = 0, L5 = 0;
    /* -- trial point: s_{k+1}, lam_{k+1}, u_k, duals of u_k, delta_{k+1} */
    double sn_t[6] = {0, 0, 0, 0, 0, 0}, ln_t[6] = {0, 0, 0, 0, 0, 0}, del_t = 0, acc_t = 0, del_nx = 0;
    double zdl_t = 0, zdu_t = 0, zal_t = 0, zau_t = 0;
    /* staging: record j (iterate + direction of stage j) in buffer (M-1-j)&1 */
    ws.stage_drain();
    ws.stage_fetch_it(0, M - 1, I);
    ws.stage_fetch_d(0, M - 1);
    MPC_STAGE_LOOP
    for (int k = M; k >= 0; --k) {
      const int bk = (M - k) & 1;                    /* buffer of record k-1 */
      double s_o[6], s_t[6], lam_t[6] = {0, 0, 0, 0, 0, 0};
      double zs0 = 0, zs1 = 0, zs2 = 0, zs3 = 0;     /* trial duals of psi_k, v_k: zl_psi, zu_psi, zl_v, zu_v */
      double n_del_o = 0, n_acc_o = 0, n_ddk = 0, lo0 = 0, lo1 = 0, lo2 = 0, lo4 = 0, lo5 = 0;
      double n_del_t = 0, n_acc_t = 0, n_zdl = 0, n_zdu = 0, n_zal = 0, n_zau = 0;
      if (k >= 1) {
        if (k >= 2) {
          ws.stage_fetch_it(bk ^ 1, k - 2, I);
          ws.stage_fetch_d(bk ^ 1, k - 2);
          if (k == M) ws.template stage_wait<STG_IT_OPS + STG_D_OPS>();
          else ws.template stage_wait<STG_IT_OPS + STG_D_OPS + ST_TRIAL>();
        } else ws.template stage_wait<0>();
        const int r = k - 1;
        double ds[6];
        MPC_UNROLL
        for (int i = 0; i < 6; i++) { s_o[i] = ws.sit(bk, r, I, F_S + i); ds[i] = ws.sx(bk, r, F_D, D_S + i); }
        const double lo3 = ws.sit(bk, r, I, F_LAM + 3);
        lo0 = ws.sit(bk, r, I, F_LAM + 0); lo1 = ws.sit(bk, r, I, F_LAM + 1); lo2 = ws.sit(bk, r, I, F_LAM + 2);
        lo4 = ws.sit(bk, r, I, F_LAM + 4); lo5 = ws.sit(bk, r, I, F_LAM + 5);
        n_del_o = ws.sit(bk, r, I, F_U + 0); n_acc_o = ws.sit(bk, r, I, F_U + 1);
        const double ddel = ws.sx(bk, r, F_D, D_U + 0), dacc = ws.sx(bk, r, F_D, D_U + 1);
        n_ddk = ddel;
        if (r == 0) R.du0 = fmax(fabs(ddel), fabs(dacc));   /* the outputs' part of the step (termination polish) */
        /* ---- costate: lam+_k ---- */
        double dl0 = 0, dl1 = 0, dl2 = 0, dl3 = 0, dl4 = 0, dl5 = 0;
        if (with_costate) {
          double Hpp, Hvv, Hee, Hcc, gp, gv, ge, gc;
          state_terms(s_o[2], s_o[3], s_o[4], s_o[5], ws.sit(bk, r, I, F_ZL + 0), ws.sit(bk, r, I, F_ZU + 0),
                      ws.sit(bk, r, I, F_ZL + 1), ws.sit(bk, r, I, F_ZU + 1), Hpp, Hvv, Hee, Hcc, gp, gv, ge, gc);
          double n0, n1, n2, n3, n4, n5;
          if (k == M) {
            n0 = -(hxy * ds[0]);
            n1 = -(hxy * ds[1]);
            n2 = -(gp + (Hpp + dw) * ds[2]);
            n3 = -(gv + (Hvv + dw) * ds[3]);
            n4 = -(gc + (Hcc + dw) * ds[4]);
            n5 = -(ge + (Hee + dw) * ds[5]);
          } else {
            const double v = s_o[3];
            Lin L;
            linearise(s_o, del_o, acc_o, sn_o, L);   /* the residual part is unused here and is eliminated */
            const double sp = L.sp, cp = L.cp, se = L.se, ce = L.ce, fp = L.fp, g1 = L.g1, h3 = L.h3, fpp = L.fpp;
            const double vdt = v * dt, Apv = del_o * dtLf;
            /* curvature of stage k */
            const double Hxx = -lc * fpp + le * h3, Hpsi2 = (lx * cp + ly * sp) * vdt, Hpv = (lx * sp - ly * cp) * dt;
            const double Hee2 = lc * vdt * se, Hev = -lc * dt * ce, Hvd = -(lp + le) * dtLf;
            const double L25 = L2 + L5;
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
          lmax = fmax(lmax, fmax(fmax(fmax(fabs(dl0), fabs(dl1)), fmax(fabs(dl2), fabs(dl3))), fmax(fabs(dl4), fabs(dl5))));
        }
        /* ---- trial: record k-1 ---- */
        lam_t[0] = lo0 + alpha_l * dl0; lam_t[1] = lo1 + alpha_l * dl1; lam_t[2] = lo2 + alpha_l * dl2;
        lam_t[3] = lo3 + alpha_l * dl3; lam_t[4] = lo4 + alpha_l * dl4; lam_t[5] = lo5 + alpha_l * dl5;
        MPC_UNROLL
        for (int i = 0; i < 6; i++) s_t[i] = s_o[i] + alpha * ds[i];
        n_del_t = n_del_o + alpha * ddel;
        n_acc_t = n_acc_o + alpha * dacc;
        MPC_UNROLL
        for (int i = 0; i < 6; i += 2) {
          ws.store2(r, J, F_S + i, s_t[i], s_t[i + 1]);
          ws.store2(r, J, F_LAM + i, lam_t[i], lam_t[i + 1]);
          R.lsum += fabs(lam_t[i]) + fabs(lam_t[i + 1]);
        }
        ws.store2(r, J, F_U, n_del_t, n_acc_t);
        /* duals of psi_k, v_k, delta_{k-1}, a_{k-1} */
        const double xo[4] = {s_o[2], s_o[3], n_del_o, n_acc_o};
        const double xn[4] = {s_t[2], s_t[3], n_del_t, n_acc_t};
        const double dxb[4] = {ds[2], ds[3], ddel, dacc};
        const double lo[4] = {yl, vl, dl, al}, hi[4] = {yu, vu, du, au};
        double zln[4], zun[4], prod = 1.0;
        MPC_UNROLL
        for (int b = 0; b < 4; b++) {
          const double islo = frcp1(xo[b] - lo[b]), isuo = frcp1(hi[b] - xo[b]);
          const double zl = ws.sit(bk, r, I, F_ZL + b), zu = ws.sit(bk, r, I, F_ZU + b);
          const double dzl = mu * islo - zl - zl * islo * dxb[b];
          const double dzu = mu * isuo - zu + zu * isuo * dxb[b];
          const double sl = xn[b] - lo[b], su = hi[b] - xn[b];
          if (!(sl > 0.0) || !(su > 0.0)) R.ok = false;
          const double isl = frcp1(sl), isu = frcp1(su);
          double a = zl + alpha_z * dzl, c = zu + alpha_z * dzu;
          /* kappa_sigma safeguard, W&B eq. (16) */
          a = fmax(fmin(a, ksm * isl), ksi * isl);
          c = fmax(fmin(c, ksm * isu), ksi * isu);
          zln[b] = a; zun[b] = c;
          R.zsum += a + c;
          const double pl = sl * a, pu = su * c;
          R.cmin = fmin(R.cmin, fmin(pl, pu)); R.cmax = fmax(R.cmax, fmax(pl, pu));
          prod *= sl * su;
        }
        ws.store2(r, J, F_ZL + 0, zln[0], zln[1]); ws.store2(r, J, F_ZL + 2, zln[2], zln[3]);
        ws.store2(r, J, F_ZU + 0, zun[0], zun[1]); ws.store2(r, J, F_ZU + 2, zun[2], zun[3]);
        R.L += flog(prod);
        zs0 = zln[0]; zs1 = zun[0]; zs2 = zln[1]; zs3 = zun[1];
        n_zdl = zln[2]; n_zdu = zun[2]; n_zal = zln[3]; n_zau = zun[3];
        /* objective terms of (s_k, u_{k-1}) */
        const double dv = s_t[3] - vref;
        R.f += wc * s_t[4] * s_t[4] + we * s_t[5] * s_t[5] + wv * dv * dv + wd * n_del_t * n_del_t;
      } else {
        MPC_UNROLL
        for (int i = 0; i < 6; i++) { s_o[i] = st[i]; s_t[i] = st[i]; }
      }
      if (k < M) {
        /* ---- trial: transition k, (s_k, u_k) -> s_{k+1} ---- */
        Lin L;
        linearise(s_t, del_t, acc_t, sn_t, L);
        MPC_UNROLL
        for (int i = 0; i < 6; i++) { R.theta += fabs(L.c[i]); R.cinf = fmax(R.cinf, fabs(L.c[i])); }
        const double ddl = (k >= 1) ? del_t - n_del_t : 0.0;          /* delta_k - delta_{k-1} */
        const double ddn = (k + 1 < M) ? del_nx - del_t : 0.0;        /* delta_{k+1} - delta_k */
        if (k >= 1) R.f += wdd * ddl * ddl;
        const double v = s_t[3], vdt = v * dt, Apv = del_t * dtLf, Bp = v * dtLf;
        const double l25 = ln_t[2] + ln_t[5];
        /* rows of u_k */
        const double rd = df * (2.0 * wd * del_t + 2.0 * wdd * ddl - 2.0 * wdd * ddn) - Bp * l25 - zdl_t + zdu_t;
        const double ra = -dt * ln_t[3] - zal_t + zau_t;
        R.dinf = fmax(R.dinf, fmax(fabs(rd), fabs(ra)));
        /* rows of s_k (k>=1) with A_k of the trial point */
        if (k >= 1) {
          const double r0 = lam_t[0] - (ln_t[0] + L.fp * ln_t[4] - L.g1 * ln_t[5]);
          const double r1 = lam_t[1] - (ln_t[1] - ln_t[4]);
          const double r2 = lam_t[2] - (-vdt * L.sp * ln_t[0] + vdt * L.cp * ln_t[1] + l25) - zs0 + zs1;
          const double r3 = df * 2.0 * wv * (s_t[3] - vref) + lam_t[3] -
                            (dt * L.cp * ln_t[0] + dt * L.sp * ln_t[1] + Apv * l25 + ln_t[3] + dt * L.se * ln_t[4]) - zs2 + zs3;
          const double r4 = df * 2.0 * wc * s_t[4] + lam_t[4];
          const double r5 = df * 2.0 * we * s_t[5] + lam_t[5] - vdt * L.ce * ln_t[4];
          R.dinf = fmax(R.dinf, fmax(fmax(fabs(r0), fabs(r1)), fmax(fmax(fabs(r2), fabs(r3)), fmax(fabs(r4), fabs(r5)))));
        }
      } else {
        /* terminal state rows */
        const double r2 = lam_t[2] - zs0 + zs1;
        const double r3 = df * 2.0 * wv * (s_t[3] - vref) + lam_t[3] - zs2 + zs3;
        const double r4 = df * 2.0 * wc * s_t[4] + lam_t[4];
        const double r5 = df * 2.0 * we * s_t[5] + lam_t[5];
        R.dinf = fmax(R.dinf, fmax(fmax(fabs(lam_t[0]), fabs(lam_t[1])), fmax(fmax(fabs(r2), fabs(r3)), fmax(fabs(r4), fabs(r5)))));
      }
      /* carry to step k-1 */
      MPC_UNROLL
      for (int i = 0; i < 6; i++) { sn_o[i] = s_o[i]; sn_t[i] = s_t[i]; ln_t[i] = lam_t[i]; }
      del_nx = del_t;
      del_t = n_del_t; acc_t = n_acc_t; zdl_t = n_zdl; zdu_t = n_zdu; zal_t = n_zal; zau_t = n_zau;
      del_o = n_del_o; acc_o = n_acc_o; ddk = n_ddk; lx = lo0; ly = lo1; lp = lo2; lc = lo4; le = lo5;
    }
    if (!(R.theta == R.theta) || !(R.f == R.f) || !(R.L == R.L) || !(R.dinf == R.dinf)) R.ok = false;
    return R;
  }

  MPC_HD double kkt_error(const Eval &e, double mu_) const {
    const double m = 6.0 * M, nb = 8.0 * M;
    const double sd = fmax(IpmConst::s_max, (e.lsum + e.zsum) / (m + nb)) / IpmConst::s_max;
    const double sc = fmax(IpmConst::s_max, e.zsum / nb) / IpmConst::s_max;
    const double compl_ = fmax(fabs(e.cmax - mu_), fabs(e.cmin - mu_));
    return fmax(fmax(e.dinf / sd, e.cinf), compl_ / sc);
  }

  MPC_HD bool filter_rejects(double th, double ph) const {
    bool r = false;
    r |= (nf > 0) && th >= fth0 && ph >= fph0;
    r |= (nf > 1) && th >= fth1 && ph >= fph1;
    r |= (nf > 2) && th >= fth2 && ph >= fph2;
    r |= (nf > 3) && th >= fth3 && ph >= fph3;
    return r;
  }
  MPC_HD void filter_add(double th, double ph) {
    int slot = nf;
    if (nf >= 4) {
      /* full: overwrite the entry with the largest theta (the least restrictive one) */
      slot = 0;
      double worst = fth0;
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
      for (int i = 0; i < 6; i++) { ws.it(k, IT0, F_S + i) = 0.0; ws.it(k, IT0, F_LAM + i) = 0.0; }
      ws.it(k, IT0, F_S + 2) = psi_start;
      ws.it(k, IT0, F_U + 0) = 0.0; ws.it(k, IT0, F_U + 1) = 0.0;
      MPC_UNROLL
      for (int b = 0; b < 4; b++) { ws.it(k, IT0, F_ZL + b) = 1.0; ws.it(k, IT0, F_ZU + b) = 1.0; }
      MPC_UNROLL
      for (int i = 0; i < D_N; i++) ws.setD(k, i, 0.0);
    }
  }

  /* ------------------------------------------------------------------ */
  /* set-up: instance constants, start point (MPC.cpp:204-257)           */
  /* ------------------------------------------------------------------ */
  MPC_HD int setup(const double *state6, const double *coef5, double yaw_lo, double yaw_hi, const double *w12,
                   bool write_start = true) {
    MPC_UNROLL
    for (int i = 0; i < 6; i++) st[i] = state6[i];
    MPC_UNROLL
    for (int i = 0; i < MPC_NCOEF; i++) coef[i] = coef5[i];
    yl = yaw_lo; yu = yaw_hi;
    M = P.N - 1; dt = P.dt; iLf = 1.0 / P.Lf; dtLf = P.dt / P.Lf;
    vl = -P.max_speed; vu = P.max_speed; dl = -P.max_steering; du = P.max_steering;
    al = P.max_deceleration; au = P.max_acceleration;
    fth0 = fth1 = fth2 = fth3 = fph0 = fph1 = fph2 = fph3 = 0.0;
    lsm = false; cur = 0; iters = 0; n_reg = 0; nf = 0; E.f = 0.0;
    /* Branch outcomes at the start point xi = (state at index 0, zeros elsewhere):
     * for i >= 1 every variable is 0, so (MPC.cpp:72-112)
     *   |cte_i| < ctePanic  -> w[0] unless ctePanic <= 0
     *   |epsi_i| > epsiPanic -> w[10] only if epsiPanic < 0
     *   vref_i = computeSpeedTarget(0, maxSpeed); v_i < 0, a_i > 0, a_i < 0,
     *   a_{i+1} > a_i are all false -> those terms are not on the tape. */
    wc = (0.0 < P.cte_panic) ? w12[0] : w12[11];
    we = (0.0 > P.epsi_panic) ? w12[10] : w12[1];
    wv = w12[2]; wd = w12[3]; wdd = w12[4];
    vref = speed_target(P, 0.0, P.max_speed);
    /* i = 0 terms: constants of the objective (their variables are fixed), MPC.cpp:71-92 */
    const double wc0 = (fabs(st[4]) < P.cte_panic) ? w12[0] : w12[11];
    const double we0 = (fabs(st[5]) > P.epsi_panic) ? w12[10] : w12[1];
    const double vref0 = speed_target(P, st[2], P.max_speed);
    cost0 = wc0 * st[4] * st[4] + we0 * st[5] * st[5] + wv * (st[3] - vref0) * (st[3] - vref0);
    double g0 = fmax(fabs(2.0 * wc0 * st[4]), fabs(2.0 * we0 * st[5]));
    double gv0 = 2.0 * wv * (st[3] - vref0);
    if (st[3] < 0.0) { cost0 += w12[9] * st[3] * st[3]; gv0 += 2.0 * w12[9] * st[3]; }
    g0 = fmax(g0, fabs(gv0));
    /* gradient-based objective scaling at the start point (IPOPT default) */
    g0 = fmax(g0, fabs(2.0 * wv * vref));
    df = (g0 > 100.0) ? fmax(100.0 / g0, 1e-8) : 1.0;
    /* start point: zeros (MPC.cpp:207-210), pushed into the interior like IPOPT does */
    double psi0 = 0.0;
    {
      const double pl = fmin(IpmConst::kappa1 * fmax(1.0, fabs(yl)), IpmConst::kappa2 * (yu - yl));
      const double pu = fmin(IpmConst::kappa1 * fmax(1.0, fabs(yu)), IpmConst::kappa2 * (yu - yl));
      psi0 = fmin(fmax(psi0, yl + pl), yu - pu);
    }
    psi_start = psi0;
    if (write_start) start_point();   /* a parked instance that is being resumed brings its iterate along */
    /* the fixed initial state must satisfy its own bounds (MPC.cpp:229-239 vs :269-281) */
    if (!(st[2] >= yl && st[2] <= yu) || !(fabs(st[3]) <= P.max_speed) || !(yl < yu)) return MPC_STATUS_INFEASIBLE;
    return MPC_STATUS_SUCCESS;
  }

  /* ------------------------------------------------------------------ */
  /* the interior-point iteration                                         */
  /* ------------------------------------------------------------------ */
  enum { MPC_RUNNING = -1 };
  enum { PH_EVAL0 = 0, PH_LS = 1, PH_DIR = 2, PH_BACKTRACK = 3 };
  enum { kMaxPolish = 6 };
  /* state of the interior-point loop (see step()) */
  int phase, iter, n_polish;
  bool ls_start, tiny;
  double out_step;     /* |alpha d(delta_0, a_0)|_inf of the last accepted step */
  double alpha, alpha_l, alpha_z, dw_cur, theta_max, theta_min, dw_last;
  double theta_k, phi_k, pth, pdp, amin;   /* line-search state */

  /* Solve from the start point that setup()/start_point() has written.
   * A line search that runs out of step length is where IPOPT would enter its feasibility-restoration phase.
   * Stand-in (same as the oracle's): restart ONCE from the start point with zero equality multipliers. */
  MPC_HD int solve() {
    int it_total = 0, attempt = 0;
    begin(true);
    for (;;) {
      const int r = step();
      if (r == MPC_RUNNING) continue;
      if (r == MPC_STATUS_LINESEARCH && attempt == 0) {
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
  enum { PARK_N = 36 };
  template <class A> MPC_HD void park(A a, int attempt, int it_total) const {
    a(0) = mu; a(1) = tau; a(2) = E.theta; a(3) = E.cinf; a(4) = E.f; a(5) = E.L; a(6) = E.dinf; a(7) = E.cmin; a(8) = E.cmax;
    a(9) = E.lsum; a(10) = E.zsum; a(11) = fth0; a(12) = fth1; a(13) = fth2; a(14) = fth3; a(15) = fph0; a(16) = fph1;
    a(17) = fph2; a(18) = fph3; a(19) = theta_max; a(20) = theta_min; a(21) = dw_last;
    a(22) = (double)nf; a(23) = (double)iter; a(24) = (double)n_reg; a(25) = (double)cur; a(26) = E.ok ? 1.0 : 0.0;
    a(27) = ls_start ? 1.0 : 0.0; a(28) = (double)attempt; a(29) = (double)it_total;
    a(30) = out_step; a(31) = (double)n_polish; a(32) = 0.0; a(33) = 0.0; a(34) = 0.0; a(35) = 0.0;
  }
  template <class A> MPC_HD void unpark(A a, int &attempt, int &it_total) {
    begin(a(27) != 0.0);
    mu = a(0); tau = a(1); E.theta = a(2); E.cinf = a(3); E.f = a(4); E.L = a(5); E.dinf = a(6); E.cmin = a(7); E.cmax = a(8);
    E.lsum = a(9); E.zsum = a(10); fth0 = a(11); fth1 = a(12); fth2 = a(13); fth3 = a(14); fph0 = a(15); fph1 = a(16);
    fph2 = a(17); fph3 = a(18); theta_max = a(19); theta_min = a(20); dw_last = a(21);
    nf = (int)a(22); iter = (int)a(23); n_reg = (int)a(24); cur = (int)a(25); E.ok = a(26) != 0.0;
    attempt = (int)a(28); it_total = (int)a(29); out_step = a(30); n_polish = (int)a(31);
    iters = iter; phase = PH_DIR;
  }

  MPC_HD void begin(bool ls) {
    cur = 0; mu = IpmConst::mu_init; tau = fmax(IpmConst::tau_min, 1.0 - mu); nf = 0; iters = 0; n_reg = 0; lsm = false;
    /* with the least-squares multiplier start the first pass is the LS pass itself: its trial sweep evaluates the
     * start point (primal part unchanged) with the estimated multipliers, so a separate evaluation is only needed
     * when that estimate is rejected or not wanted */
    phase = ls ? PH_LS : PH_EVAL0; iter = 0; ls_start = ls; tiny = false; n_polish = 0; out_step = 1e300;
    alpha = alpha_l = alpha_z = dw_cur = 0.0;
    theta_max = theta_min = dw_last = 0.0;
    theta_k = phi_k = pth = pdp = amin = 0.0;
  }

  /* One pass of the interior-point loop, written as a small state machine so that each sweep has exactly ONE
   * call site (they are force-inlined; several call sites would multiply the code size), and so that the lanes of
   * a wave can be in different phases -- or, in the device kernel, on different instances:
   *   EVAL0     evaluate the start point
   *   LS        least-squares multiplier start (W&B section 3.6, IPOPT default): one Riccati pass
   *             with identity Hessian; estimates above constr_mult_init_max = 1000 are discarded
   *   DIR       convergence test, barrier update, search direction, first trial of the line search
   *   BACKTRACK further trials of the same line search
   * Returns MPC_RUNNING, or the final status of this attempt. */
  MPC_HD int step() {
    const double mu_floor = P.tol / 10.0;
    if (phase == PH_LS || phase == PH_DIR) {
      if (phase == PH_DIR) {
        iters = iter;
        const double E0 = kkt_error(E, 0.0);
        if (!(E0 == E0)) return MPC_STATUS_NUMERIC;
        if (E0 <= P.tol) {
          /* IPOPT's own rule stops here.  Termination polish (MpcParams.polish, include/mpc_amd.h): carry on with
           * Newton steps at the final barrier parameter until the outputs (delta_0, a_0) have stopped moving, so
           * that the point returned is the central-path point itself and not whichever iterate crossed tol first
           * (an interior a_0 still moves by ~1e-4 per step there).  At most kMaxPolish extra iterations. */
          if (!P.polish || n_polish >= kMaxPolish || iter >= P.max_iter || (mu <= mu_floor && out_step <= P.out_step_tol))
            return MPC_STATUS_SUCCESS;
          ++n_polish;
          if (mu > mu_floor) { mu = mu_floor; tau = fmax(IpmConst::tau_min, 1.0 - mu); nf = 0; }
        }
        if (iter >= P.max_iter) return MPC_STATUS_MAXITER;
        /* barrier update, W&B eq. (7) */
        while (kkt_error(E, mu) <= IpmConst::kappa_eps * mu && mu > mu_floor) {
          mu = fmax(mu_floor, fmin(IpmConst::kappa_mu * mu, mu * sqrt(mu)));
          /* with the termination polish the solve ends at the floor anyway: a value within 3x of it (IPOPT's schedule
           * lands on 2.4e-9 before 1e-9) goes there directly, which saves most instances one iteration */
          if (P.polish && mu < 3.0 * mu_floor) mu = mu_floor;
          tau = fmax(IpmConst::tau_min, 1.0 - mu);
          nf = 0;
        }
#if defined(MPC_TRACE) && !defined(__HIP_DEVICE_COMPILE__)
        printf("it %3d f=%.8g theta=%.3e dinf=%.3e cmin=%.2e cmax=%.2e mu=%.2e E0=%.3e nf=%d\n", iter, E.f, E.theta,
               E.dinf, E.cmin, E.cmax, mu, E0, nf);
#endif
      }
      lsm = (phase == PH_LS);
      /* search direction with inertia correction, W&B section 3.1 */
      double dw = 0.0;
      int tries = 0;
      bool okb = true;
      while (!backward(dw)) {
        if (lsm) { okb = false; break; }
        if (dw == 0.0) dw = (dw_last == 0.0) ? IpmConst::dw_0 : fmax(IpmConst::dw_min, IpmConst::kw_minus * dw_last);
        else dw *= (dw_last == 0.0) ? IpmConst::kw_plus_bar : IpmConst::kw_plus;
        if (dw > IpmConst::dw_max || ++tries > 100) return MPC_STATUS_LINESEARCH;
      }
      if (okb) forward();
      dw_cur = dw;
      if (phase == PH_LS) {
        if (!okb) { lsm = false; ls_start = false; phase = PH_EVAL0; return MPC_RUNNING; }
        alpha = 0.0; alpha_l = 1.0; alpha_z = 0.0;      /* lam <- lam_LS; primal point and bound duals unchanged */
      } else {
        if (dw > 0.0) { dw_last = dw; n_reg++; }
        /* filter line search, W&B algorithm A */
        theta_k = E.theta; phi_k = df * E.f - mu * E.L;
        pth = hpow(theta_k, IpmConst::s_theta);                       /* theta^s_theta */
        pdp = (dphi < 0.0) ? hpow(-dphi, IpmConst::s_phi) : 0.0;      /* (-dphi)^s_phi */
        if (dphi < 0.0) {
          const double t3 = (theta_k <= theta_min) ? IpmConst::delta_sw * pth / pdp : IpmConst::gamma_theta;
          amin = IpmConst::gamma_alpha * fmin(fmin(IpmConst::gamma_theta, IpmConst::gamma_phi * theta_k / (-dphi)), t3);
        } else amin = IpmConst::gamma_alpha * IpmConst::gamma_theta;
        tiny = dxinf <= 10.0 * IpmConst::eps * fmax(1.0, xinf);
        alpha = amax; alpha_l = amax; alpha_z = az;
      }
    }
    double lmax;
    const Eval T = costate_trial(dw_cur, alpha, alpha_l, alpha_z, phase != PH_EVAL0, lmax);
    if (phase == PH_EVAL0) {
      E = T; cur = 1;
      if (!E.ok) return MPC_STATUS_NUMERIC;
      theta_max = 1e4 * fmax(1.0, E.theta); theta_min = 1e-4 * fmax(1.0, E.theta);
      phase = ls_start ? PH_LS : PH_DIR;
      return MPC_RUNNING;
    }
    if (phase == PH_LS) {
      lsm = false; ls_start = false;
      /* estimates above constr_mult_init_max = 1000 are discarded: the start point is then evaluated as it is */
      if (lmax <= 1000.0) {
        E = T; cur = 1 - cur;
        if (!E.ok) return MPC_STATUS_NUMERIC;
        theta_max = 1e4 * fmax(1.0, E.theta); theta_min = 1e-4 * fmax(1.0, E.theta);
        phase = PH_DIR;
      } else phase = PH_EVAL0;
      return MPC_RUNNING;
    }
    /* acceptance test of the line search */
    bool accepted = false, ftype = false;
    if (tiny) { accepted = T.ok; ftype = true; }
    else if (T.ok) {
      const double phi_t = df * T.f - mu * T.L;
      const double eps_phi = 10.0 * IpmConst::eps * fabs(phi_k);
      if (T.theta < theta_max && !filter_rejects(T.theta, phi_t)) {
        const bool sw = dphi < 0.0 && alpha * pdp > IpmConst::delta_sw * pth;
        const bool armijo = phi_t - phi_k - eps_phi <= IpmConst::eta_phi * alpha * dphi;
        if (theta_k <= theta_min && sw) {
          if (armijo) accepted = true;
        } else if (T.theta <= (1.0 - IpmConst::gamma_theta) * theta_k ||
                   phi_t - phi_k - eps_phi <= -IpmConst::gamma_phi * theta_k) {
          accepted = true;
        }
        ftype = sw && armijo; /* the filter is augmented unless both hold (W&B step A-7) */
      }
    }
    if (accepted) {
      if (!ftype) filter_add((1.0 - IpmConst::gamma_theta) * theta_k, phi_k - IpmConst::gamma_phi * theta_k);
      cur = 1 - cur;
      E = T;
      out_step = alpha * T.du0;
      ++iter;
      phase = PH_DIR;
      return MPC_RUNNING;
    }
    if (tiny) return MPC_STATUS_LINESEARCH;
    alpha *= 0.5; alpha_l = alpha;
    if (alpha < amin) return MPC_STATUS_LINESEARCH;
    phase = PH_BACKTRACK;
    return MPC_RUNNING;
  }

  /* MPC.cpp:306-324: out9 and the optional N-point trajectory */
  template <class OutF, class TrajF>
  MPC_HD void unpack(OutF out, TrajF traj, bool want_traj) const {
    const int I = it(cur);
    MPC_UNROLL
    for (int i = 0; i < 6; i++) out(i) = ws.it(0, I, F_S + i);
    out(6) = ws.it(0, I, F_U + 0);
    out(7) = ws.it(0, I, F_U + 1);
    out(8) = E.f + cost0;
    if (want_traj) {
      const int N = P.N;
      traj(0) = st[0]; traj(N) = st[1];
      for (int k = 0; k < M; ++k) { traj(k + 1) = ws.it(k, I, F_S + 0); traj(N + k + 1) = ws.it(k, I, F_S + 1); }
    }
  }
};

/* One instance, end to end (used by the test-only host build; the device kernel
 * drives Solver directly so that outputs go straight to their HBM arrays). */
template <class WS>
MPC_HD int solve_instance(const MpcParams &P, WS ws, const double *state6, const double *coef5, double yaw_lo,
                          double yaw_hi, const double *w12, double *out9, double *traj2N, int *iters_out) {
  Solver<WS> S(P, ws);
  int status = S.setup(state6, coef5, yaw_lo, yaw_hi, w12);
  if (status == MPC_STATUS_SUCCESS) status = S.solve();
  double *o = out9;
  double *t = traj2N;
  S.unpack([o](int i) -> double & { return o[i]; }, [t](int i) -> double & { return t[i]; }, traj2N != nullptr);
  if (iters_out) *iters_out = S.iters;
  return status;
}

}  // namespace mpc
#endif /* MPC_CORE_H */
