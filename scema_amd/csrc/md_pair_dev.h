// md_pair_dev.h -- device helpers and entry layouts shared by the list build (md_neigh.hip) and the pair kernel (md_pair.hip)
#pragma once
#include <hip/hip_runtime.h>

#include "md_device.h"
#include "md_kernels.h"

#define TW MD_TILE_WAVES    // waves per tile workgroup
#define TT (TW * 64)
#define NI MD_CLUSTER
#define E_LMASK 0x1FFF
#define E_TYPE_SHIFT 13
#define E_MASK_SHIFT 17
#define E_FAR (1 << 21)    // (inside k_neigh_build only: a skin-band entry of the far part, C2)
#define CODE_HOME 13       // image code of (0,0,0)

// slot records are stored as two arrays of 16-byte halves, (x,y)[npad] then (z,q)[npad]: a wave's gather
// instruction then touches 16 B per lane at stride 16
#define XQ_X(S, s) (((const double *)(S).xq)[2 * (size_t)(s)])
#define XQ_Y(S, s) (((const double *)(S).xq)[2 * (size_t)(s) + 1])
#define XQ_Z(S, s) (((const double *)(S).xq)[2 * (size_t)(S).npad + 2 * (size_t)(s)])
#define XQ_Q(S, s) (((const double *)(S).xq)[2 * (size_t)(S).npad + 2 * (size_t)(s) + 1])


// XCD-aware block -> (simulation, tile) map.  Workgroups are dealt round-robin over the 8 XCDs
// (block L lands on XCD L % 8), each with its own 4 MiB L2.  A simulation's j gathers touch its
// whole 332 KB position table and its force atomics its 250 KB force table, so all tiles of one
// simulation are placed on ONE XCD: simulation s uses the blocks with L % 8 == s % 8.  That needs
// groups of 8 simulations; the last nsims % 8 simulations (all of them in a small batch, e.g. the
// single-replica check of BASELINE config 2) spread their tiles over all XCDs instead, so no XCD
// idles.  Placement only affects speed, never results.
__device__ __forceinline__ bool xcd_map(int ntiles, int nsims, int &sim, int &tile) {
  const int L = blockIdx.x;
  const int full = nsims & ~7;
  if (L < full * ntiles) {
    const int x = L & 7, w = L >> 3;
    sim = (w / ntiles) * 8 + x;
    tile = w % ntiles;
  } else {
    const int Lr = L - full * ntiles;
    sim = full + Lr / ntiles;
    tile = Lr % ntiles;
  }
  return sim < nsims;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// 1/sqrt(x): hardware estimate (v_rsq_f64, ~2^-26 relative) + one third-order correction
// y (1 + e/2 + 3 e^2/8), e = 1 - x y^2; the remaining error is O(e^3) < 1e-22 -> correctly
// rounded to within 1 ulp, at 6 instructions instead of the ~10 of the library routine
__device__ __forceinline__ double rsqrt_f64(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double e = fma(-x * y, y, 1.0);
  return fma(y, e * fma(0.375, e, 0.5), y);
}
__device__ __forceinline__ int popc_below(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
}
// wave-wide max on the DPP path (row shifts, then row broadcasts; lanes without a source keep their own value), result from lane 63
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_keep(double v) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, ROWMASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, ROWMASK, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max_dpp(double v) {
  SCEMA_ASSERT_FULL_WAVE();   // (md_device.h: all 64 lanes active, gfx9 row broadcasts)
  v = fmax(v, dpp_keep<0x111, 0xF>(v)); v = fmax(v, dpp_keep<0x112, 0xF>(v)); v = fmax(v, dpp_keep<0x114, 0xF>(v)); v = fmax(v, dpp_keep<0x118, 0xF>(v));
  v = fmax(v, dpp_keep<0x142, 0xA>(v));   // row_bcast:15 -> rows 1, 3
  v = fmax(v, dpp_keep<0x143, 0xC>(v));   // row_bcast:31 -> rows 2, 3
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
// cross-lane move through the DPP path of the VALU (no LDS traffic): every lane reads the lane selected by CTRL
// inside its row of 16 (quad_perm / row_shr), lanes without a source read 0
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
#define DPP_QUAD_XOR1 0xB1   // quad_perm [1,0,3,2]
#define DPP_QUAD_XOR2 0x4E   // quad_perm [2,3,0,1]
#define DPP_ROW_SHR4 0x114
#define DPP_ROW_SHR8 0x118

// inclusive prefix sum over the first 32 lanes of a wave on the DPP path (row shifts inside the rows of 16, then lane 15 broadcast
// into row 1): 5 VALU instructions instead of 5 LDS round trips
__device__ __forceinline__ int scan32_incl(int v) {
  SCEMA_ASSERT_FULL_WAVE();
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1 and 3
  return v;
}
// LDS FP64 atomic add without return value (ds_add_f64)
__device__ __forceinline__ void lds_add(double *p, double v) {
  (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// min of two finite doubles as ONE instruction (fmin() quiets signalling NaNs first: a v_max_f64 x, x per operand)
__device__ __forceinline__ double vmin_f64(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmin_f32(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmin3_f32(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ double vmax_f64(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// distance of a coordinate from an interval [lo, hi] (0 inside): max(0, lo - x, x - hi)
__device__ __forceinline__ double box_excess(double lo, double hi, double x) { return vmax_f64(0.0, vmax_f64(lo - x, x - hi)); }
// block-wide sum of NV values per thread over the TW waves of a tile workgroup, atomically added to dst[0..NV)
template <int NV>
__device__ __forceinline__ void tile_atomic_add(double (&vals)[NV], double *dst, double *lds /* >= NV*TW */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; k++) {
    double s = wave_sum(vals[k]);
    if (lane == 0) lds[k * TW + wave] = s;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    double s = 0.0;
    for (int w = 0; w < TW; w++) s += lds[threadIdx.x * TW + w];
    if (s != 0.0) atomicAdd(&dst[threadIdx.x], s);
  }
}

static inline dim3 grid_xcd(int ntiles, int ns) { return dim3((unsigned)(ns * ntiles), 1, 1); }
