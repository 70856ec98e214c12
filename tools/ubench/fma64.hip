// Practical FP64 FMA ceiling of the box: 256 CUs x 4 waves/SIMD of independent FMA chains (8 per lane), no memory traffic.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/fma64.hip -o tools/ubench/fma64 && tools/ubench/fma64
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(double *out, int iters, double c) {
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
int main() {
  const int blocks = 256 * 16, threads = 256, iters = 20000;
  double *d; hipMalloc(&d, sizeof(double) * blocks * threads);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int occ = 0; occ < 2; occ++) {
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 100, 0.999999);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, iters, 0.999999);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fmas = (double)blocks * threads * iters * 64.0;
    printf("run %d: %.1f ms, %.1f TFLOP/s FP64 (FMA = 2 flops), %.2f wave-FMA per SIMD-cycle at 2.4 GHz\n", occ, ms, 2.0 * fmas / ms / 1e9, fmas / 64.0 / (ms * 1e-3) / (1024.0 * 2.4e9));
  }
  return 0;
}
