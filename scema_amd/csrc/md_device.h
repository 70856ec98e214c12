// md_device.h -- device helpers shared by the kernel translation units (gfx950)
#pragma once
#include <hip/hip_runtime.h>

#include "md_types.h"

#define TPB 256

struct BoxD {
  double lo[3], h[6], hinv[6], vol;
};

__device__ __forceinline__ void box_derive(const double *b, BoxD &o) {
  o.lo[0] = b[0]; o.lo[1] = b[1]; o.lo[2] = b[2];
  o.h[0] = b[3] - b[0]; o.h[1] = b[4] - b[1]; o.h[2] = b[5] - b[2];
  o.h[3] = b[8]; o.h[4] = b[7]; o.h[5] = b[6];
  o.hinv[0] = 1.0 / o.h[0]; o.hinv[1] = 1.0 / o.h[1]; o.hinv[2] = 1.0 / o.h[2];
  o.hinv[3] = -o.h[3] / (o.h[1] * o.h[2]);
  o.hinv[4] = (o.h[3] * o.h[5] - o.h[1] * o.h[4]) / (o.h[0] * o.h[1] * o.h[2]);
  o.hinv[5] = -o.h[5] / (o.h[0] * o.h[1]);
  o.vol = o.h[0] * o.h[1] * o.h[2];
}

__device__ __forceinline__ void minimg(const BoxD &b, double &dx, double &dy, double &dz) {
  double l0 = b.hinv[0] * dx + b.hinv[5] * dy + b.hinv[4] * dz;
  double l1 = b.hinv[1] * dy + b.hinv[3] * dz;
  double l2 = b.hinv[2] * dz;
  l0 -= rint(l0); l1 -= rint(l1); l2 -= rint(l2);
  dx = b.h[0] * l0 + b.h[5] * l1 + b.h[4] * l2;
  dy = b.h[1] * l1 + b.h[3] * l2;
  dz = b.h[2] * l2;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// block-wide sum of NV values per thread over NW waves, result atomically added to dst[0..NV)
template <int NV, int NW>
__device__ __forceinline__ void block_atomic_add_n(double (&vals)[NV], double *dst, double *lds /* >= NV*NW */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; k++) {
    double s = wave_sum(vals[k]);
    if (lane == 0) lds[k * NW + wave] = s;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    double s = 0.0;
    for (int w = 0; w < NW; w++) s += lds[threadIdx.x * NW + w];
    if (s != 0.0) atomicAdd(&dst[threadIdx.x], s);
  }
}

// block-wide sum of NV values per thread, result atomically added to dst[0..NV)
template <int NV>
__device__ __forceinline__ void block_atomic_add(double (&vals)[NV], double *dst, double *lds /* >= NV*(TPB/64) */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; k++) {
    double s = wave_sum(vals[k]);
    if (lane == 0) lds[k * (TPB / 64) + wave] = s;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    double s = 0.0;
    for (int w = 0; w < TPB / 64; w++) s += lds[threadIdx.x * (TPB / 64) + w];
    if (s != 0.0) atomicAdd(&dst[threadIdx.x], s);
  }
}

// Opt-in bookkeeping for kernels that ask for more dynamic LDS than the default limit: the attribute belongs to a
// (kernel, device) pair, so the largest size already granted is remembered per device (engines of several GPUs may live
// in one process).
static inline size_t &lds_optin_slot(size_t (&table)[16]) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return table[(dev >= 0 && dev < 16) ? dev : 0];
}
