// md_ubench.hip -- box calibration for bench.py (scema_md_box_fma_tflops): the practical FP64 FMA ceiling of the device.
// 256 CUs x 4 SIMDs, four 256-thread workgroups' worth of waves per SIMD, eight independent FMA chains per lane, no memory traffic.
#include <hip/hip_runtime.h>

#include "../../include/scema_md.h"

__global__ __launch_bounds__(256) void k_ubench_fma64(double *out, int iters, double c) {
  double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      a0 = fma(a0, c, 1e-9); a1 = fma(a1, c, 1e-9); a2 = fma(a2, c, 1e-9); a3 = fma(a3, c, 1e-9);
      a4 = fma(a4, c, 1e-9); a5 = fma(a5, c, 1e-9); a6 = fma(a6, c, 1e-9); a7 = fma(a7, c, 1e-9);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

extern "C" int scema_md_box_fma_tflops(int32_t device, double *tflops) {
  if (!tflops) return SCEMA_MD_ERR_ARG;
  *tflops = 0.0;
  if (hipSetDevice(device) != hipSuccess) return SCEMA_MD_ERR_DEVICE;
  const int blocks = 256 * 16, threads = 256, iters = 10000;
  double *d = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipMalloc(&d, sizeof(double) * blocks * threads) != hipSuccess) return SCEMA_MD_ERR_DEVICE;
  int rc = SCEMA_MD_ERR_DEVICE;
  float ms = 0.f;
  if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
    // the clocks of a chip that has just run a light workload need tens of milliseconds of full load to settle: a warm-up launch as long as
    // the measured ones, then the best of three (the figure is a ceiling)
    hipLaunchKernelGGL(k_ubench_fma64, dim3(blocks), dim3(threads), 0, 0, d, iters, 0.999999);
    for (int rep = 0; rep < 3; rep++) {
      if (hipEventRecord(e0, 0) != hipSuccess) break;
      hipLaunchKernelGGL(k_ubench_fma64, dim3(blocks), dim3(threads), 0, 0, d, iters, 0.999999);
      if (hipEventRecord(e1, 0) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || !(ms > 0.f)) break;
      const double tf = 2.0 * (double)blocks * threads * iters * 64.0 / ms / 1e9;   // 64 FMAs per lane and iteration, 2 flop each
      if (tf > *tflops) *tflops = tf;
      rc = SCEMA_MD_OK;
    }
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(d);
  return rc;
}
