// accuracy of v_rcp_f64 with 0, 1, 2 Newton steps (decides how many the solver's frcp needs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(int n, const double *x, double *r0, double *r1, double *r2) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i], r = __builtin_amdgcn_rcp(v);
  r0[i] = r;
  double e = fma(-v, r, 1.0); r = fma(r, e, r); r1[i] = r;
  e = fma(-v, r, 1.0); r = fma(r, e, r); r2[i] = r;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), a(n), b(n), c(n);
  std::mt19937_64 g(1); std::uniform_real_distribution<double> u(-30, 30), m(1, 2);
  for (int i = 0; i < n; i++) x[i] = std::ldexp(m(g), (int)u(g)) * ((i & 1) ? -1 : 1);
  double *d; hipMalloc(&d, sizeof(double) * 4 * n);
  hipMemcpy(d, x.data(), sizeof(double) * n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, n, d, d + n, d + 2 * n, d + 3 * n);
  hipMemcpy(a.data(), d + n, sizeof(double) * n, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), d + 2 * n, sizeof(double) * n, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), d + 3 * n, sizeof(double) * n, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < n; i++) {
    long double t = 1.0L / (long double)x[i];
    e0 = fmax(e0, (double)fabsl(((long double)a[i] - t) / t));
    e1 = fmax(e1, (double)fabsl(((long double)b[i] - t) / t));
    e2 = fmax(e2, (double)fabsl(((long double)c[i] - t) / t));
  }
  printf("max relative error of 1/x: raw v_rcp_f64 %.3e, +1 Newton %.3e, +2 Newton %.3e\n", e0, e1, e2);
  return 0;
}
