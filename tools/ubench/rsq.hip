// micro-benchmark: issue cost of the FP64 reciprocal-square-root estimate against the FP32 estimate with two conversions
// (both followed by the third-order correction of md_pair.hip).  hipcc --offload-arch=gfx950 -O3 -o rsq rsq.hip && ./rsq
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__device__ __forceinline__ double corr(double x, double y) {
  const double e = fma(-x * y, y, 1.0);
  return fma(y, e * fma(0.375, e, 0.5), y);
}
template <int MODE>
__global__ void k(double *out, const double *in, long long *cyc, int iters) {
  double x[8], acc = 0.0;
  for (int u = 0; u < 8; u++) x[u] = in[threadIdx.x * 8 + u];
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      double y;
      if (MODE == 0) y = __builtin_amdgcn_rsq(x[u]);
      else if (MODE == 1) y = (double)__builtin_amdgcn_rsqf((float)x[u]);
      else y = x[u] * 0.5;   // baseline: one plain FP64 instruction
      if (MODE != 2) y = corr(x[u], y);
      acc += y;
      x[u] += 1.0e-3;
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  const int T = 256, iters = 2000;
  double *in, *out; long long *cyc;
  hipMalloc(&in, T * 8 * sizeof(double)); hipMalloc(&out, 1024 * T * sizeof(double)); hipMalloc(&cyc, 8);
  double h[T * 8];
  for (int i = 0; i < T * 8; i++) h[i] = 0.7 + 140.0 * (double)i / (T * 8);
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  long long c[3];
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k<0>, dim3(1024), dim3(T), 0, 0, out, in, cyc, iters); hipMemcpy(&c[0], cyc, 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k<1>, dim3(1024), dim3(T), 0, 0, out, in, cyc, iters); hipMemcpy(&c[1], cyc, 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k<2>, dim3(1024), dim3(T), 0, 0, out, in, cyc, iters); hipMemcpy(&c[2], cyc, 8, hipMemcpyDeviceToHost);
  }
  const double n = 8.0 * iters;
  printf("cycles per rsqrt (one wave of 4 per SIMD, 1024 blocks): f64 estimate %.1f, f32 estimate + 2 cvt %.1f, baseline (mul+2 add) %.1f\n", c[0] / n, c[1] / n, c[2] / n);
  // accuracy of the f32 route
  double worst = 0.0;
  for (int i = 0; i < T * 8; i++) {
    const double x = h[i];
    const float yf = 1.0f / sqrtf((float)x);
    double y = (double)yf;
    const double e = fma(-x * y, y, 1.0);
    y = fma(y, e * fma(0.375, e, 0.5), y);
    worst = fmax(worst, fabs(y * sqrt(x) - 1.0));
  }
  printf("host model of the f32 route: worst relative error %.2e\n", worst);
  return 0;
}
