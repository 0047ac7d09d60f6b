// What the cross-lane steps of a several-lanes-per-instance Riccati stage would cost, measured:
// ONE wave, groups of 8 lanes, fp64 values moved with DPP (two 32-bit moves per value: DPP does not take 64-bit operands).
//   rs8x8: reduce-scatter of an 8x8 block held as partial sums on 8 lanes -> lane a ends up with row a (G^T W of a stage)
//   ag12:  all-gather of 12 values held two per lane on 6 lanes -> every lane has all 12 (the feedback gains K)
// Each is timed in a dependent chain (the next repetition consumes the result), as it would sit in the stage step.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CTRL> __device__ __forceinline__ double dpp(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// DPP controls: quad_perm [1,0,3,2] = 0xB1 (lane ^ 1), quad_perm [2,3,0,1] = 0x4E (lane ^ 2), row_half_mirror = 0x141 (lane -> 7 - lane)
constexpr int X1 = 0xB1, X2 = 0x4E, HM = 0x141;

__global__ void rs8x8(double *out, int iters) {
  const int lane = threadIdx.x & 7;
  double m[8][8];
  for (int a = 0; a < 8; a++)
    for (int b = 0; b < 8; b++) m[a][b] = 1e-3 * (threadIdx.x + 1) + 0.01 * a + 0.001 * b;
  double row[8];
  for (int b = 0; b < 8; b++) row[b] = 0;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    // step 1: halves exchange (partner = 7 - lane): lanes 0..3 keep rows 0..3, lanes 4..7 keep rows 4..7
    const bool upper = lane >= 4;
    double h[4][8];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 8; b++) {
        const double mine = upper ? m[7 - a][b] : m[a][b];         // the row this lane keeps (numbered from its own end)
        const double give = upper ? m[a][b] : m[7 - a][b];         // the row the partner keeps
        h[a][b] = mine + dpp<HM>(give);
      }
    // step 2: lane ^ 2 inside the half
    const bool up2 = (lane & 2) != 0;
    double q[2][8];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 8; b++) {
        const double mine = up2 ? h[2 + a][b] : h[a][b];
        const double give = up2 ? h[a][b] : h[2 + a][b];
        q[a][b] = mine + dpp<X2>(give);
      }
    // step 3: lane ^ 1
    const bool up1 = (lane & 1) != 0;
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const double mine = up1 ? q[1][b] : q[0][b];
      const double give = up1 ? q[0][b] : q[1][b];
      row[b] = mine + dpp<X1>(give);
    }
    // dependency into the next repetition
#pragma unroll
    for (int a = 0; a < 8; a++)
#pragma unroll
      for (int b = 0; b < 8; b++) m[a][b] = m[a][b] * 0.999 + row[b] * 1e-9;
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int b = 0; b < 8; b++) s += row[b];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) out[64] = (double)(t1 - t0);
}

// the dependency update alone (64 fma per lane), to subtract
__global__ void rs8x8_base(double *out, int iters) {
  double m[8][8], row[8];
  for (int a = 0; a < 8; a++)
    for (int b = 0; b < 8; b++) m[a][b] = 1e-3 * (threadIdx.x + 1) + 0.01 * a + 0.001 * b;
  for (int b = 0; b < 8; b++) row[b] = 1e-3 * b;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int a = 0; a < 8; a++)
#pragma unroll
      for (int b = 0; b < 8; b++) m[a][b] = m[a][b] * 0.999 + row[b] * 1e-9;
#pragma unroll
    for (int b = 0; b < 8; b++) row[b] = m[b][b];
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int a = 0; a < 8; a++) for (int b = 0; b < 8; b++) s += m[a][b];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) out[64] = (double)(t1 - t0);
}

__global__ void ag12(double *out, int iters) {
  const int lane = threadIdx.x & 7;
  double k0 = 1e-3 * threadIdx.x, k1 = 2e-3 * threadIdx.x, all[16];
  for (int j = 0; j < 16; j++) all[j] = 0;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    // recursive doubling inside the group of 8: 2 -> 4 -> 8 -> 16 values per lane (12 of them are gains)
    double a2[4], a4[8];
    const double p0 = dpp<X1>(k0), p1 = dpp<X1>(k1);
    const bool o1 = lane & 1;
    a2[0] = o1 ? p0 : k0; a2[1] = o1 ? p1 : k1; a2[2] = o1 ? k0 : p0; a2[3] = o1 ? k1 : p1;
    const bool o2 = lane & 2;
#pragma unroll
    for (int j = 0; j < 4; j++) { const double p = dpp<X2>(a2[j]); a4[j] = o2 ? p : a2[j]; a4[4 + j] = o2 ? a2[j] : p; }
    const bool o4 = lane & 4;
#pragma unroll
    for (int j = 0; j < 8; j++) { const double p = dpp<HM>(a4[j]); all[j] = o4 ? p : a4[j]; all[8 + j] = o4 ? a4[j] : p; }
    double s = 0;
#pragma unroll
    for (int j = 0; j < 12; j++) s += all[j];
    k0 = k0 * 0.999 + s * 1e-9; k1 = k1 * 0.999 + s * 1e-9;
  }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = k0 + k1;
  if (threadIdx.x == 0) out[64] = (double)(t1 - t0);
}

template <class F> void run(const char *name, F kern, double *d) {
  const int iters = 2000;
  hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double h[65]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-12s %8.1f ns per repetition (wall), %8.1f shader cycles ; checksum %.6g\n", name, ms * 1e6 / iters, h[64] / iters, h[0]);
}
int main() {
  double *d; hipMalloc(&d, 66 * sizeof(double));
  run("rs8x8", rs8x8, d);
  run("rs8x8_base", rs8x8_base, d);
  run("ag12", ag12, d);
  return 0;
}
