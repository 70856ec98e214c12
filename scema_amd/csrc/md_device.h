// md_device.h -- device helpers shared by the kernel translation units (gfx950)
#pragma once
#include <hip/hip_runtime.h>

#include "md_types.h"

// A pointer read from a structure in memory is a generic ("flat") pointer to the compiler: its loads and stores are FLAT instructions, which may
// hit LDS, tick BOTH memory counters and return out of order with everything else -- so every use of a flat load's result waits with
// s_waitcnt vmcnt(0) lgkmcnt(0): no load stays in flight across an LDS read, a prefetch written in the source is waited for at once.
// Saying that the pointer is in global memory gives global_load / global_store: vmcnt only, counted in order, prefetches stay in flight.
#define GLOBAL_AS __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ const GLOBAL_AS T *as_global(const T *p) {
  return (const GLOBAL_AS T *)p;
}
template <class T>
__device__ __forceinline__ GLOBAL_AS T *as_global_w(T *p) {
  return (GLOBAL_AS T *)p;
}
// (double2 is a class: it cannot be read through an address-space pointer; the native vector type can)
typedef double dvec2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ldg2(const GLOBAL_AS dvec2 *p, size_t k) { const dvec2 v = p[k]; return make_double2(v.x, v.y); }

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

// The box is the same for every lane of a workgroup; said so (v_readfirstlane), its derived quantities live in scalar registers
// instead of 2 x 15 vector registers per lane (k_bonded spilled 21 registers around its torsion code with the box in VGPRs).
__device__ __forceinline__ double wave_uniform(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ void box_uniform(BoxD &b) {
#pragma unroll
  for (int k = 0; k < 3; k++) b.lo[k] = wave_uniform(b.lo[k]);
#pragma unroll
  for (int k = 0; k < 6; k++) { b.h[k] = wave_uniform(b.h[k]); b.hinv[k] = wave_uniform(b.hinv[k]); }
  b.vol = wave_uniform(b.vol);
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

// Sum over the 64 lanes of a wave, returned in every lane.  On the DPP path of the VALU (quad permutes, row mirrors, then the row
// broadcasts of gfx9): 12 cross-lane moves that cost an instruction each -- as `__shfl_down` steps they were 12 LDS round trips
// (ds_bpermute), which is what the tails of the small kernels waited for (a single replica's k_pppm_solve: 11 of its 71 thousand cycles).
// All lanes of the wave must be active (a debug build, -DSCEMA_DEVICE_ASSERTS, traps otherwise), and the row broadcasts exist on
// gfx9 only: the library is built for gfx950 and refuses other targets here instead of returning garbage sums there.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "md_device.h: the DPP reductions (row_bcast) are written for gfx9 / CDNA: build with --offload-arch=gfx950"
#endif
#ifdef SCEMA_DEVICE_ASSERTS
#define SCEMA_ASSERT_FULL_WAVE() do { if (__builtin_amdgcn_read_exec() != ~0ull) __builtin_trap(); } while (0)
#else
#define SCEMA_ASSERT_FULL_WAVE() do { } while (0)
#endif
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_read0(double v) {   // the lane CTRL selects; 0 where there is none or the row is not in ROWMASK
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
  SCEMA_ASSERT_FULL_WAVE();
  v += dpp_read0<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
  v += dpp_read0<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
  v += dpp_read0<0x141, 0xF>(v);   // row_half_mirror
  v += dpp_read0<0x140, 0xF>(v);   // row_mirror: every lane of a row of 16 holds the row's sum
  v += dpp_read0<0x142, 0xA>(v);   // row_bcast:15 into rows 1 and 3
  v += dpp_read0<0x143, 0xC>(v);   // row_bcast:31 into rows 2 and 3: lane 63 holds the total
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// the first four steps of wave_sum: every lane of a row of 16 lanes holds the sum over its row (the same additions in the same order as
// wave_sum makes inside a row: a value that lives in one row only has the same sum bit for bit either way)
__device__ __forceinline__ double row_sum(double v) {
  SCEMA_ASSERT_FULL_WAVE();
  v += dpp_read0<0xB1, 0xF>(v);
  v += dpp_read0<0x4E, 0xF>(v);
  v += dpp_read0<0x141, 0xF>(v);
  v += dpp_read0<0x140, 0xF>(v);
  return v;
}
__device__ __forceinline__ double lane_value(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
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

// ------------------------------------------------------------------------------------------
// Nose-Hoover chain half step (fix nvt; one sub-cycle, no drag).  Returns the velocity factor.
// ------------------------------------------------------------------------------------------
__device__ inline double nhc_half(const SimDev &S, SimScalars &sc) {
  const int mt = S.t_chain;
  const double dt = S.dt, dthalf = 0.5 * dt, dt4 = 0.25 * dt, dt8 = 0.125 * dt;
  const double t_target = S.ramp ? sc.t_target_now : S.t_target;
  const double ke_target = S.tdof * MD_BOLTZ * t_target;
  double kecurrent = S.tdof * MD_BOLTZ * sc.t_current;
  const double tf2 = S.t_freq * S.t_freq;
  sc.eta_mass[0] = S.tdof * MD_BOLTZ * t_target / tf2;
  for (int k = 1; k < mt; k++) sc.eta_mass[k] = MD_BOLTZ * t_target / tf2;
  sc.eta_dotdot[0] = (sc.eta_mass[0] > 0.0) ? (kecurrent - ke_target) / sc.eta_mass[0] : 0.0;
  double expfac;
  for (int k = mt - 1; k > 0; k--) {
    expfac = exp(-dt8 * sc.eta_dot[k + 1]);
    sc.eta_dot[k] *= expfac;
    sc.eta_dot[k] += sc.eta_dotdot[k] * dt4;
    sc.eta_dot[k] *= expfac;
  }
  expfac = exp(-dt8 * sc.eta_dot[1]);
  sc.eta_dot[0] *= expfac;
  sc.eta_dot[0] += sc.eta_dotdot[0] * dt4;
  sc.eta_dot[0] *= expfac;
  const double factor = exp(-dthalf * sc.eta_dot[0]);
  sc.t_current *= factor * factor;
  kecurrent = S.tdof * MD_BOLTZ * sc.t_current;
  sc.eta_dotdot[0] = (sc.eta_mass[0] > 0.0) ? (kecurrent - ke_target) / sc.eta_mass[0] : 0.0;
  for (int k = 0; k < mt; k++) sc.eta[k] += dthalf * sc.eta_dot[k];
  sc.eta_dot[0] *= expfac;
  sc.eta_dot[0] += sc.eta_dotdot[0] * dt4;
  sc.eta_dot[0] *= expfac;
  for (int k = 1; k < mt; k++) {
    expfac = exp(-dt8 * sc.eta_dot[k + 1]);
    sc.eta_dot[k] *= expfac;
    sc.eta_dotdot[k] = (sc.eta_mass[k - 1] * sc.eta_dot[k - 1] * sc.eta_dot[k - 1] - MD_BOLTZ * t_target) / sc.eta_mass[k];
    sc.eta_dot[k] += sc.eta_dotdot[k] * dt4;
    sc.eta_dot[k] *= expfac;
  }
  return factor;
}

__device__ inline void box_corners(const double *box, double *c /*24*/) {
  BoxD b;
  box_derive(box, b);
  int k = 0;
  for (int iz = 0; iz < 2; iz++)
    for (int iy = 0; iy < 2; iy++)
      for (int ix = 0; ix < 2; ix++) {
        c[3 * k + 0] = b.h[0] * ix + b.h[5] * iy + b.h[4] * iz + b.lo[0];
        c[3 * k + 1] = b.h[1] * iy + b.h[3] * iz + b.lo[1];
        c[3 * k + 2] = b.h[2] * iz + b.lo[2];
        k++;
      }
}

