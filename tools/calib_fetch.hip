// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for THIS kernel's access shape: 8 bytes per lane
// (global_load_dwordx2 / global_store_dwordx2), 512 B per wave-instruction, streaming, footprint well past
// the 256 MiB Infinity Cache.  MI355X_MICROARCH.md calibrates FETCH_SIZE only for 16-B-per-lane loads
// (reads exactly 1/2) and says other widths must be calibrated on a known byte count.
//   hipcc --offload-arch=gfx950 -O3 -o calib_fetch tools/calib_fetch.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE  ... -- ./calib_fetch     (and again with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void calib_copy8(const double *__restrict__ in, double *__restrict__ out, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = in[i] * 1.0000001;
}
int main() {
  const long n = 1L << 28;  // 2 GiB read + 2 GiB written
  double *a, *b;
  if (hipMalloc(&a, n * 8) != hipSuccess || hipMalloc(&b, n * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(a, 0, n * 8);
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(calib_copy8, dim3(4096), dim3(64), 0, 0, a, b, n);
  hipDeviceSynchronize();
  printf("calib_copy8: %ld bytes read and %ld bytes written per launch\n", n * 8, n * 8);
  return 0;
}
