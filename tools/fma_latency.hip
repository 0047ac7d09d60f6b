// fp64 FMA dependent-issue latency with ONE wave per SIMD: cycles per FMA for 1, 2, 4, 8 independent chains
#include <hip/hip_runtime.h>
#include <cstdio>
template <int C>
__global__ void k(double *out, int iters, double a, double b) {
  double x[C];
  for (int c = 0; c < C; c++) x[c] = threadIdx.x * 1e-3 + c;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
#pragma unroll
      for (int c = 0; c < C; c++) x[c] = fma(x[c], a, b);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int c = 0; c < C; c++) s += x[c];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) out[64] = (double)(t1 - t0);
}
template <int C> void run(double *d) {
  const int iters = 4096;
  hipLaunchKernelGGL(k<C>, dim3(1), dim3(64), 0, 0, d, iters, 0.999999, 1e-7);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<C>, dim3(1), dim3(64), 0, 0, d, iters, 0.999999, 1e-7);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double h[65]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double n = (double)iters * 16 * C;
  printf("%d chain(s): %.2f ns per FMA (wall), s_memtime ticks per FMA %.3f\n", C, ms * 1e6 / n, h[64] / n);
}
int main() {
  double *d; hipMalloc(&d, 66 * sizeof(double));
  run<1>(d); run<2>(d); run<4>(d); run<8>(d);
  return 0;
}
