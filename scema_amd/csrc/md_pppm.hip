// md_pppm.hip -- PPPM reciprocal part: what `kspace_style pppm 0.0001` (in.set.lammps:36,
// ELASTIC/potential.mod.lammps:11) asks LAMMPS for, as an alternative to the Ewald sum of md_kernels.hip (scema_md_params::kspace_style).
// Published algorithm (Hockney & Eastwood; LAMMPS pppm.cpp 17Nov16 as remembered [LAMMPS-ext], restated first in
// oracle/md_oracle.c pppm_*): order-5 charge assignment on a grid in lamda coordinates (triclinic boxes need nothing special),
// optimal influence function for ik differentiation with direct alias sums, energy and virial in reciprocal space, three
// inverse transforms for the field, forces by the same weights.  Transforms: hipFFT, one batched plan for all replicas of a
// launch group (they share the grid).  One launch per stage for the whole batch:
//   k_pppm_spread   one workgroup per replica; the real grid lives in LDS while the charges are spread (ds_add_f64), then
//                   leaves as the complex input of the transform (grids too large for the LDS: global atomics)
//   k_pppm_gf       influence function of the current box, 125 alias terms per mode; only when the box has changed
//   k_pppm_poisson  energy, virial, field spectra -i k G rho(k)
//   k_pppm_force    per atom: the 125 grid points of three field grids, added to the forces the other kernels assembled
#include <hip/hip_runtime.h>

#include "md_device.h"
#include "md_pppm.h"
#include "md_types.h"

#define PP_ORDER 5
#define PP_TPB 1024

// weights of the order-5 cardinal B-spline at the 5 grid points i-2 .. i+2 around u, i = floor(u + 1/2) (oracle: pppm_weights)
__device__ __forceinline__ int pppm_weights(double u, double (&w)[PP_ORDER]) {
  const int i = (int)floor(u + 0.5);
  const double dx = (double)i - u;
#pragma unroll
  for (int k = 0; k < PP_ORDER; k++) {
    const double t = -dx - (double)(k - 2) + 0.5 * PP_ORDER;
    double m[PP_ORDER];
#pragma unroll
    for (int j = 0; j < PP_ORDER; j++) m[j] = (t - j >= 0.0 && t - j < 1.0) ? 1.0 : 0.0;
#pragma unroll
    for (int n = 2; n <= PP_ORDER; n++)
#pragma unroll
      for (int j = 0; j + n <= PP_ORDER; j++) {
        const double tj = t - j;
        m[j] = (tj * m[j] + ((double)n - tj) * m[j + 1]) / (double)(n - 1);
      }
    w[k] = m[0];
  }
  return i;
}
__device__ __forceinline__ int pmod(int a, int n) { const int r = a % n; return r < 0 ? r + n : r; }
__device__ __forceinline__ void atom_lamda(const SimDev &S, const BoxD &b, int a, double &t0, double &t1, double &t2) {
  const double d0 = S.x[3 * a] - b.lo[0], d1 = S.x[3 * a + 1] - b.lo[1], d2 = S.x[3 * a + 2] - b.lo[2];
  const double l0 = b.hinv[0] * d0 + b.hinv[5] * d1 + b.hinv[4] * d2, l1 = b.hinv[1] * d1 + b.hinv[3] * d2, l2 = b.hinv[2] * d2;
  t0 = l0 - floor(l0); t1 = l1 - floor(l1); t2 = l2 - floor(l2);
}

extern __shared__ double s_grid[];
__global__ __launch_bounds__(PP_TPB) void k_pppm_spread(const SimDev *sims, int use_lds) {
  const SimDev &S = sims[blockIdx.x];
  const int nx = S.pg[0], ny = S.pg[1], nz = S.pg[2];
  if (nx == 0) return;
  const int NG = nx * ny * nz;
  BoxD b;
  box_derive(S.sc->box, b);
  double2 *rho = (double2 *)S.pgrid;
  if (use_lds) {
    for (int k = threadIdx.x; k < NG; k += PP_TPB) s_grid[k] = 0.0;
  } else {
    for (int k = threadIdx.x; k < NG; k += PP_TPB) rho[k] = make_double2(0.0, 0.0);
    __threadfence_block();
  }
  __syncthreads();
  const double delvolinv = (double)NG / b.vol;
  for (int a = threadIdx.x; a < S.natoms; a += PP_TPB) {
    double l0, l1, l2;
    atom_lamda(S, b, a, l0, l1, l2);
    double wx[PP_ORDER], wy[PP_ORDER], wz[PP_ORDER];
    const int ix = pppm_weights(l0 * nx, wx), iy = pppm_weights(l1 * ny, wy), iz = pppm_weights(l2 * nz, wz);
    const double z0 = delvolinv * S.q[a];
    if (z0 == 0.0) continue;
#pragma unroll
    for (int c = 0; c < PP_ORDER; c++) {
      const int gz = pmod(iz + c - 2, nz);
#pragma unroll
      for (int bb = 0; bb < PP_ORDER; bb++) {
        const int gy = pmod(iy + bb - 2, ny);
        const double zy = z0 * wz[c] * wy[bb];
        const int row = (gz * ny + gy) * nx;
#pragma unroll
        for (int k = 0; k < PP_ORDER; k++) {
          const int gx = pmod(ix + k - 2, nx);
          if (use_lds) (void)__hip_atomic_fetch_add(&s_grid[row + gx], zy * wx[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          else atomicAdd(&rho[row + gx].x, zy * wx[k]);
        }
      }
    }
  }
  if (use_lds) {
    __syncthreads();
    for (int k = threadIdx.x; k < NG; k += PP_TPB) rho[k] = make_double2(s_grid[k], 0.0);
  }
}

__device__ __forceinline__ double sinc_pow10(double x) {
  if (x == 0.0) return 1.0;
  const double s = sin(x) / x, s2 = s * s, s4 = s2 * s2;
  return s4 * s4 * s2;
}
// optimal influence function of the current box (ik differentiation), alias sums |m| <= 2 per dimension
__global__ __launch_bounds__(256) void k_pppm_gf(const SimDev *sims) {
  const SimDev &S = sims[blockIdx.y];
  const int nx = S.pg[0], ny = S.pg[1], nz = S.pg[2];
  if (nx == 0) return;
  const int NG = nx * ny * nz, idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= NG) return;
  const int m1 = idx % nx, m2 = (idx / nx) % ny, m3 = idx / (nx * ny);
  const int p1 = m1 - nx * (2 * m1 / nx), p2 = m2 - ny * (2 * m2 / ny), p3 = m3 - nz * (2 * m3 / nz);
  if (p1 == 0 && p2 == 0 && p3 == 0) { S.pgf[idx] = 0.0; return; }
  BoxD b;
  box_derive(S.sc->box, b);
  const double twopi = 2.0 * MD_PI;
  const double kx = twopi * (b.hinv[0] * p1), ky = twopi * (b.hinv[5] * p1 + b.hinv[1] * p2), kz = twopi * (b.hinv[4] * p1 + b.hinv[3] * p2 + b.hinv[2] * p3);
  const double sqk = kx * kx + ky * ky + kz * kz, g2inv4 = 0.25 / (S.g_ewald * S.g_ewald);
  double wxs[5], wys[5], wzs[5];   // squared transform of the assignment function, per lattice index
#pragma unroll
  for (int a = 0; a < 5; a++) {
    wxs[a] = sinc_pow10(MD_PI * (double)(p1 + nx * (a - 2)) / nx);
    wys[a] = sinc_pow10(MD_PI * (double)(p2 + ny * (a - 2)) / ny);
    wzs[a] = sinc_pow10(MD_PI * (double)(p3 + nz * (a - 2)) / nz);
  }
  double num = 0.0, den = 0.0;
  for (int a3 = 0; a3 < 5; a3++)
    for (int a2 = 0; a2 < 5; a2++) {
      const int q2 = p2 + ny * (a2 - 2), q3 = p3 + nz * (a3 - 2);
      const double w23 = wys[a2] * wzs[a3];
#pragma unroll
      for (int a1 = 0; a1 < 5; a1++) {
        const int q1 = p1 + nx * (a1 - 2);
        const double qx = twopi * (b.hinv[0] * q1), qy = twopi * (b.hinv[5] * q1 + b.hinv[1] * q2), qz = twopi * (b.hinv[4] * q1 + b.hinv[3] * q2 + b.hinv[2] * q3);
        const double dot2 = qx * qx + qy * qy + qz * qz;
        const double w2 = wxs[a1] * w23;
        den += w2;
        num += (kx * qx + ky * qy + kz * qz) / dot2 * exp(-dot2 * g2inv4) * w2;
      }
    }
  S.pgf[idx] = 4.0 * MD_PI / sqk * num / (den * den);
}

// energy, virial and the three field spectra; grid 0 holds rho(k) (unnormalised), grids 1..3 receive E_x, E_y, E_z (k)
__global__ __launch_bounds__(256) void k_pppm_poisson(const SimDev *sims) {
  const SimDev &S = sims[blockIdx.y];
  const int nx = S.pg[0], ny = S.pg[1], nz = S.pg[2];
  if (nx == 0) return;
  __shared__ double s_red[8 * 4];
  const int NG = nx * ny * nz, idx = blockIdx.x * 256 + threadIdx.x;
  if ((int)(blockIdx.x * 256) >= NG) return;
  double v[6] = {0, 0, 0, 0, 0, 0}, e[1] = {0};
  BoxD b;
  box_derive(S.sc->box, b);
  if (idx < NG) {
    double2 *grid = (double2 *)S.pgrid;
    const int m1 = idx % nx, m2 = (idx / nx) % ny, m3 = idx / (nx * ny);
    const int p1 = m1 - nx * (2 * m1 / nx), p2 = m2 - ny * (2 * m2 / ny), p3 = m3 - nz * (2 * m3 / nz);
    const double twopi = 2.0 * MD_PI;
    const double kx = twopi * (b.hinv[0] * p1), ky = twopi * (b.hinv[5] * p1 + b.hinv[1] * p2), kz = twopi * (b.hinv[4] * p1 + b.hinv[3] * p2 + b.hinv[2] * p3);
    const double gf = S.pgf[idx], scaleinv = 1.0 / (double)NG;
    const double2 r = grid[idx];
    const double ar = r.x * scaleinv, ai = r.y * scaleinv;
    const double pr = gf * ar, pi = gf * ai;
    const size_t gs = (size_t)S.pgstride;
    grid[gs + idx] = make_double2(kx * pi, -kx * pr);       // (a + i b)(-i k) = b k - i a k
    grid[2 * gs + idx] = make_double2(ky * pi, -ky * pr);
    grid[3 * gs + idx] = make_double2(kz * pi, -kz * pr);
    if (gf != 0.0) {
      const double sqk = kx * kx + ky * ky + kz * kz;
      const double eg = 0.5 * b.vol * MD_QQRD2E * gf * (ar * ar + ai * ai);
      const double vterm = -2.0 * (1.0 / sqk + 0.25 / (S.g_ewald * S.g_ewald));
      e[0] = eg;
      v[0] = eg * (1.0 + vterm * kx * kx); v[1] = eg * (1.0 + vterm * ky * ky); v[2] = eg * (1.0 + vterm * kz * kz);
      v[3] = eg * vterm * kx * ky; v[4] = eg * vterm * kx * kz; v[5] = eg * vterm * ky * kz;
    }
    if (idx == 0)   // self energy and neutralising background
      e[0] -= MD_QQRD2E * (S.g_ewald * S.qsqsum / sqrt(MD_PI) + 0.5 * MD_PI * S.qsum * S.qsum / (S.g_ewald * S.g_ewald * b.vol));
  }
  block_atomic_add_n<6, 4>(v, S.sc->vir + P_KSPACE * 6, s_red);
  block_atomic_add_n<1, 4>(e, S.sc->eng + P_KSPACE, s_red);
}

// forces: the field (real parts of grids 1..3 after the inverse transforms) at the atom, by the assignment weights
__global__ __launch_bounds__(256) void k_pppm_force(const SimDev *sims) {
  const SimDev &S = sims[blockIdx.y];
  const int nx = S.pg[0], ny = S.pg[1], nz = S.pg[2];
  if (nx == 0) return;
  const int a = blockIdx.x * 256 + threadIdx.x;
  if (a >= S.natoms) return;
  const double qa = S.q[a];
  if (qa == 0.0) return;
  BoxD b;
  box_derive(S.sc->box, b);
  const size_t gs = (size_t)S.pgstride;
  const double2 *ex = (const double2 *)S.pgrid + gs, *ey = ex + gs, *ez = ey + gs;
  double l0, l1, l2;
  atom_lamda(S, b, a, l0, l1, l2);
  double wx[PP_ORDER], wy[PP_ORDER], wz[PP_ORDER];
  const int ix = pppm_weights(l0 * nx, wx), iy = pppm_weights(l1 * ny, wy), iz = pppm_weights(l2 * nz, wz);
  double fx = 0.0, fy = 0.0, fz = 0.0;
#pragma unroll
  for (int c = 0; c < PP_ORDER; c++) {
    const int gz = pmod(iz + c - 2, nz);
#pragma unroll
    for (int bb = 0; bb < PP_ORDER; bb++) {
      const int gy = pmod(iy + bb - 2, ny);
      const double zy = wz[c] * wy[bb];
      const size_t row = ((size_t)gz * ny + gy) * nx;
#pragma unroll
      for (int k = 0; k < PP_ORDER; k++) {
        const size_t g = row + pmod(ix + k - 2, nx);
        const double w = zy * wx[k];
        fx = fma(w, ex[g].x, fx); fy = fma(w, ey[g].x, fy); fz = fma(w, ez[g].x, fz);
      }
    }
  }
  const double qf = MD_QQRD2E * qa;
  S.f[3 * a] += qf * fx; S.f[3 * a + 1] += qf * fy; S.f[3 * a + 2] += qf * fz;
}

static inline dim3 grid2(int nx, int ns) { return dim3((unsigned)nx, (unsigned)ns, 1); }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

size_t mdk_pppm_lds_limit() { return 144 * 1024; }
void mdk_pppm_spread(hipStream_t st, const SimDev *d, int ns, int maxgrid) {
  const size_t lds = (size_t)maxgrid * sizeof(double);
  const int use_lds = lds <= mdk_pppm_lds_limit() ? 1 : 0;
  static size_t optin_tab[16] = {0};
  size_t &optin = lds_optin_slot(optin_tab);
  if (use_lds && lds > 64 * 1024 && lds > optin) { (void)hipFuncSetAttribute((const void *)k_pppm_spread, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); optin = lds; }
  hipLaunchKernelGGL(k_pppm_spread, dim3(ns), dim3(PP_TPB), use_lds ? lds : 0, st, d, use_lds);
}
void mdk_pppm_gf(hipStream_t st, const SimDev *d, int ns, int maxgrid) { hipLaunchKernelGGL(k_pppm_gf, grid2(cdiv(maxgrid, 256), ns), dim3(256), 0, st, d); }
void mdk_pppm_poisson(hipStream_t st, const SimDev *d, int ns, int maxgrid) { hipLaunchKernelGGL(k_pppm_poisson, grid2(cdiv(maxgrid, 256), ns), dim3(256), 0, st, d); }
void mdk_pppm_force(hipStream_t st, const SimDev *d, int ns, int maxatoms) { hipLaunchKernelGGL(k_pppm_force, grid2(cdiv(maxatoms, 256), ns), dim3(256), 0, st, d); }
