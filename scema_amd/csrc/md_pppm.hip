// md_pppm.hip -- PPPM reciprocal part: what `kspace_style pppm 0.0001` (in.set.lammps:36,
// ELASTIC/potential.mod.lammps:11) asks LAMMPS for, as an alternative to the Ewald sum of md_kernels.hip (scema_md_params::kspace_style).
// Published algorithm (Hockney & Eastwood; LAMMPS pppm.cpp 17Nov16 as remembered [LAMMPS-ext], restated first in
// oracle/md_oracle.c pppm_*): order-5 charge assignment on a grid in lamda coordinates (triclinic boxes need nothing special),
// optimal influence function for ik differentiation with direct alias sums, energy and virial in reciprocal space, three
// inverse transforms for the field, forces by the same weights.  Transforms: hipFFT, one batched plan for all replicas of a
// launch group (they share the grid).  One launch per stage for the whole batch:
//   k_pppm_spread   a few workgroups per replica, each with a private copy of the real grid in LDS while its atoms are spread
//                   (ds_add_f64), added to the complex input of the transform at the end (grids beyond the LDS: global atomics)
//   k_pppm_gf       influence function of the current box, 125 alias terms per mode; only when the box has changed
//   k_pppm_poisson  energy, virial, field spectra -i k G rho(k)
//   k_pppm_force    per atom: the 125 grid points of three field grids (staged in LDS when they fit), added to the forces the
//                   other kernels assembled
#include <hip/hip_runtime.h>

#include "md_device.h"
#include "md_env.h"
#include "md_pppm.h"
#include "md_types.h"

#include <algorithm>
#include <cstdlib>

static inline dim3 grid2(int nx, int ns) { return dim3((unsigned)nx, (unsigned)ns, 1); }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

#define PP_ORDER 5
#define PP_TPB 1024
#define PP_SOLVE_MAX 2900   // grid points up to which k_pppm_solve keeps three complex grids and the twiddles in 144 KB of LDS

// weights of the order-5 cardinal B-spline at the 5 grid points i-2 .. i+2 around u, i = floor(u + 1/2).  With dx = i - u in (-1/2, 1/2] the
// argument of weight k, t_k = 9/2 - k - dx, lies in the spline's piece 4 - k for every k, at the SAME local coordinate s = 1/2 - dx in [0, 1):
// the five weights are the five quartic blending polynomials of the uniform B-spline in s (they sum to 1) -- 18 multiply-adds instead of the
// recursion over indicator functions that the oracle walks (oracle: pppm_weights; 330 instructions per dimension, half of the spreading and of
// the interpolation kernel until round 5).  Same numbers to the last places (the two forms round differently: 1e-16).
static_assert(PP_ORDER == 5, "the blending polynomials below are those of order 5");
__device__ __forceinline__ int pppm_weights(double u, double (&w)[PP_ORDER]) {
  const int i = (int)floor(u + 0.5);
  const double s = 0.5 - ((double)i - u), r = 1.0 - s;
  const double s2 = s * s, r2 = r * r;
  constexpr double q = 1.0 / 24.0;
  w[4] = q * (s2 * s2);
  w[0] = q * (r2 * r2);
  w[3] = fma(fma(fma(fma(-4.0 * q, s, 4.0 * q), s, 6.0 * q), s, 4.0 * q), s, q);
  w[2] = fma(fma(fma(fma(6.0 * q, s, -12.0 * q), s, -6.0 * q), s, 12.0 * q), s, 11.0 * q);
  w[1] = fma(fma(fma(fma(-4.0 * q, s, 12.0 * q), s, -6.0 * q), s, -12.0 * q), s, 11.0 * q);
  return i;
}
__device__ __forceinline__ int pmod(int a, int n) { const int r = a % n; return r < 0 ? r + n : r; }
// the five periodic grid indices i-2 .. i+2 of one dimension (0 <= i <= n); grids of fewer than 4 points wrap more than once
__device__ __forceinline__ void pppm_wrap(int i, int n, int (&g)[PP_ORDER]) {
#pragma unroll
  for (int k = 0; k < PP_ORDER; k++) {
    const int v = i + k - 2;
    g[k] = v < 0 ? v + n : (v >= n ? v - n : v);
  }
  if (n < 4)
#pragma unroll
    for (int k = 0; k < PP_ORDER; k++) g[k] = pmod(i + k - 2, n);
}
__device__ __forceinline__ void atom_lamda(const SimDev &S, const BoxD &b, int a, double &t0, double &t1, double &t2) {
  const double d0 = S.x[3 * a] - b.lo[0], d1 = S.x[3 * a + 1] - b.lo[1], d2 = S.x[3 * a + 2] - b.lo[2];
  const double l0 = b.hinv[0] * d0 + b.hinv[5] * d1 + b.hinv[4] * d2, l1 = b.hinv[1] * d1 + b.hinv[3] * d2, l2 = b.hinv[2] * d2;
  t0 = l0 - floor(l0); t1 = l1 - floor(l1); t2 = l2 - floor(l2);
}

// grid 0 <- 0 for the spreading kernels that add to it from several workgroups
__global__ __launch_bounds__(256) void k_pppm_zero(const SimDev *sims) {
  const SimDev &S = sims[blockIdx.y];
  const int NG = S.pg[0] * S.pg[1] * S.pg[2], idx = blockIdx.x * 256 + threadIdx.x;
  if (idx < NG) ((double2 *)S.pgrid)[idx] = make_double2(0.0, 0.0);
}

// Charge assignment.  Workgroup (s, r) takes the s-th of `split` contiguous atom ranges of replica r.  Within a wave the lanes
// take atoms that are far apart in the file (lane l: atoms l*rows .. (l+1)*rows - 1 of the range, one per iteration): bonded
// neighbours share their 125 grid points, and 64 lanes adding to the same addresses would be serialised by the LDS.  LDS = true:
// a private copy of the real grid in LDS (ds_add_f64), added to grid 0 at the end (the only global atomics: one per grid point
// and workgroup; a direct store when the replica has one workgroup).  LDS = false (grid beyond the LDS): global atomics throughout.
// PADX (LDS only, nx >= 5): the LDS copy has nx + 5 points per x row -- an atom's five x points are columns i .. i + 4 of the padded row, never
// wrapped, so the address of a point is the row's plus a CONSTANT (the offset field of ds_add_f64) instead of an addition per point; the five
// pad columns are folded back onto the points they alias before the copy leaves.
extern __shared__ double s_grid[];
template <bool LDS, int TPB_, bool PADX = false>
__global__ __launch_bounds__(TPB_) void k_pppm_spread(const SimDev *sims, int split) {
  const SimDev &S = sims[blockIdx.y];
  const int nx = S.pg[0], ny = S.pg[1], nz = S.pg[2];
  if (nx == 0) return;
  const int NG = nx * ny * nz, T = (int)blockDim.x;
  const int nxp = PADX ? nx + 5 : nx, NGP = nxp * ny * nz;
  const int chunk = (S.natoms + split - 1) / split, a0 = (int)blockIdx.x * chunk, a1 = min(S.natoms, a0 + chunk);
  if (a0 >= a1) return;
  BoxD b;
  box_derive(S.sc->box, b);
  box_uniform(b);
  double2 *rho = (double2 *)S.pgrid;
  if (LDS) {
    for (int k = threadIdx.x; k < NGP; k += T) s_grid[k] = 0.0;
    __syncthreads();
  }
  const double delvolinv = (double)NG / b.vol;
  const int rows = (a1 - a0 + 63) >> 6, lane = threadIdx.x & 63;
  // Two consecutive atoms per lane and turn.  Neighbours in the file are bonded neighbours: about half of such pairs have the same
  // nearest grid point, hence the same 125 grid points, and then ONE atomic per point carries both contributions (the LDS
  // atomics, at ~26 cycles per wave instruction, are what this kernel waits for).
  for (int r = 2 * ((int)threadIdx.x >> 6); r < rows; r += 2 * (T >> 6)) {
    const int aA = a0 + lane * rows + r, aB = aA + 1;
    const bool vA = aA < a1, vB = r + 1 < rows && aB < a1;
    if (!vA) continue;
    double wxA[PP_ORDER], wyA[PP_ORDER], wzA[PP_ORDER], wxB[PP_ORDER] = {0, 0, 0, 0, 0}, wyB[PP_ORDER] = {0, 0, 0, 0, 0}, wzB[PP_ORDER] = {0, 0, 0, 0, 0};
    int gxA[PP_ORDER], gyA[PP_ORDER], gzA[PP_ORDER], gxB[PP_ORDER] = {0, 0, 0, 0, 0}, gyB[PP_ORDER] = {0, 0, 0, 0, 0}, gzB[PP_ORDER] = {0, 0, 0, 0, 0};
    double l0, l1, l2;
    atom_lamda(S, b, aA, l0, l1, l2);
    const int ixA = pppm_weights(l0 * nx, wxA), iyA = pppm_weights(l1 * ny, wyA), izA = pppm_weights(l2 * nz, wzA);
    pppm_wrap(ixA, nx, gxA); pppm_wrap(iyA, ny, gyA); pppm_wrap(izA, nz, gzA);
    const double zA = delvolinv * S.q[aA];
    double zB = 0.0;
    int ixB = -1000, iyB = 0, izB = 0;
    if (vB) {
      atom_lamda(S, b, aB, l0, l1, l2);
      ixB = pppm_weights(l0 * nx, wxB); iyB = pppm_weights(l1 * ny, wyB); izB = pppm_weights(l2 * nz, wzB);
      pppm_wrap(ixB, nx, gxB); pppm_wrap(iyB, ny, gyB); pppm_wrap(izB, nz, gzB);
      zB = delvolinv * S.q[aB];
    }
    const bool merged = vB && ixA == ixB && iyA == iyB && izA == izB;
#pragma unroll
    for (int k = 0; k < PP_ORDER; k++) { wxA[k] *= zA; if (vB) wxB[k] *= zB; }
    if (zA != 0.0 || (merged && zB != 0.0)) {
#pragma unroll
      for (int c = 0; c < PP_ORDER; c++) {
#pragma unroll
        for (int bb = 0; bb < PP_ORDER; bb++) {
          const double zyA = wzA[c] * wyA[bb], zyB = merged ? wzB[c] * wyB[bb] : 0.0;   // (B's weights are zeros where there is no B)
          const int row = (gzA[c] * ny + gyA[bb]) * nxp;
          double *prow = s_grid + row + ixA;   // PADX
#pragma unroll
          for (int k = 0; k < PP_ORDER; k++) {
            const double val = fma(zyB, wxB[k], zyA * wxA[k]);
            if (LDS && PADX) (void)__hip_atomic_fetch_add(prow + k, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (LDS) (void)__hip_atomic_fetch_add(&s_grid[row + gxA[k]], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else atomicAdd(&rho[row + gxA[k]].x, val);
          }
        }
      }
    }
    if (vB && !merged && zB != 0.0) {
#pragma unroll
      for (int c = 0; c < PP_ORDER; c++) {
#pragma unroll
        for (int bb = 0; bb < PP_ORDER; bb++) {
          const double zyB = wzB[c] * wyB[bb];
          const int row = (gzB[c] * ny + gyB[bb]) * nxp;
          double *prow = s_grid + row + ixB;   // PADX
#pragma unroll
          for (int k = 0; k < PP_ORDER; k++) {
            if (LDS && PADX) (void)__hip_atomic_fetch_add(prow + k, zyB * wxB[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (LDS) (void)__hip_atomic_fetch_add(&s_grid[row + gxB[k]], zyB * wxB[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else atomicAdd(&rho[row + gxB[k]].x, zyB * wxB[k]);
          }
        }
      }
    }
  }
  if (LDS) {
    __syncthreads();
    if (PADX) {
      // column c of a padded row holds grid point c - 2: columns 0, 1 belong to nx - 2, nx - 1 and columns nx + 2 .. nx + 4 to 0, 1, 2
      // (five distinct targets per row as nx >= 5: plain additions, one thread each)
      const int nrow = ny * nz;
      for (int k = threadIdx.x; k < 5 * nrow; k += T) {
        const int row = k / 5, pc = k - 5 * row;
        const int src = pc < 2 ? pc : nx + pc, dst = (pc < 2 ? nx - 2 + pc : pc - 2) + 2;
        s_grid[row * nxp + dst] += s_grid[row * nxp + src];
      }
      __syncthreads();
      for (int k = threadIdx.x; k < NG; k += T) {
        const int row = k / nx, x = k - row * nx;
        const double v = s_grid[row * nxp + x + 2];
        if (split == 1) rho[k] = make_double2(v, 0.0);
        else if (v != 0.0) atomicAdd(&rho[k].x, v);
      }
    } else if (split == 1) {
      for (int k = threadIdx.x; k < NG; k += T) rho[k] = make_double2(s_grid[k], 0.0);
    } else {
      for (int k = threadIdx.x; k < NG; k += T) {
        const double v = s_grid[k];
        if (v != 0.0) atomicAdd(&rho[k].x, v);
      }
    }
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

// one reciprocal mode: wave vector of grid index idx, and -- with the (unnormalised) transform r of the charge grid -- its share of
// energy and virial; returns the potential G rho(k) / NG whose products with -i k are the field spectra
__device__ __forceinline__ double2 pppm_mode(const SimDev &S, const BoxD &b, int idx, int nx, int ny, int nz, double2 r, double (&kv)[3], double (&v)[6], double &e) {
  const int NG = nx * ny * nz;
  const int m1 = idx % nx, m2 = (idx / nx) % ny, m3 = idx / (nx * ny);
  const int p1 = m1 - nx * (2 * m1 / nx), p2 = m2 - ny * (2 * m2 / ny), p3 = m3 - nz * (2 * m3 / nz);
  const double twopi = 2.0 * MD_PI;
  const double kx = twopi * (b.hinv[0] * p1), ky = twopi * (b.hinv[5] * p1 + b.hinv[1] * p2), kz = twopi * (b.hinv[4] * p1 + b.hinv[3] * p2 + b.hinv[2] * p3);
  kv[0] = kx; kv[1] = ky; kv[2] = kz;
  const double gf = S.pgf[idx], scaleinv = 1.0 / (double)NG;
  const double ar = r.x * scaleinv, ai = r.y * scaleinv;
  if (gf != 0.0) {
    const double sqk = kx * kx + ky * ky + kz * kz;
    const double eg = 0.5 * b.vol * MD_QQRD2E * gf * (ar * ar + ai * ai);
    const double vterm = -2.0 * (1.0 / sqk + 0.25 / (S.g_ewald * S.g_ewald));
    e += eg;
    v[0] += eg * (1.0 + vterm * kx * kx); v[1] += eg * (1.0 + vterm * ky * ky); v[2] += eg * (1.0 + vterm * kz * kz);
    v[3] += eg * vterm * kx * ky; v[4] += eg * vterm * kx * kz; v[5] += eg * vterm * ky * kz;
  }
  if (idx == 0)   // self energy and neutralising background
    e -= MD_QQRD2E * (S.g_ewald * S.qsqsum / sqrt(MD_PI) + 0.5 * MD_PI * S.qsum * S.qsum / (S.g_ewald * S.g_ewald * b.vol));
  return make_double2(gf * ar, gf * ai);
}

// energy, virial and the three field spectra; the charge grid holds rho(k) (unnormalised), the field grids receive E_x, E_y, E_z (k)
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
    double kv[3];
    const double2 p = pppm_mode(S, b, idx, nx, ny, nz, ((const double2 *)S.pgrid)[idx], kv, v, e[0]);
    const size_t gs = (size_t)S.pgstride;
    double2 *field = (double2 *)S.pfield;
    field[idx] = make_double2(kv[0] * p.y, -kv[0] * p.x);       // (a + i b)(-i k) = b k - i a k
    field[gs + idx] = make_double2(kv[1] * p.y, -kv[1] * p.x);
    field[2 * gs + idx] = make_double2(kv[2] * p.y, -kv[2] * p.x);
  }
  block_atomic_add_n<6, 4>(v, S.sc->vir + P_KSPACE * 6, s_red);
  block_atomic_add_n<1, 4>(e, S.sc->eng + P_KSPACE, s_red);
}

// Small grids (3 NG complex numbers + the twiddles fit the LDS: NG <= PP_SOLVE_MAX): the whole solve of a replica -- forward
// transform, energy / virial / field spectra, the inverse transforms (two: x and y ride together) -- in ONE launch, one workgroup per replica, the grid never
// leaving the LDS.  The transforms are plain DFTs along one dimension at a time (a 10 x 10 x 9 grid is 29 multiply-adds per
// point and pass; the twiddles exp(-2 pi i j / n) come from sincospi once per launch), ping-ponging between two LDS buffers;
// a third keeps rho(k) while the three field components are transformed back.  Replaces eight to ten library launches of a few
// microseconds each per step (what a single replica or a 72-replica share waits for) and leaves the charge grid zeroed for the
// next spreading pass.  Replicas of one launch may have different grids.
// Two shapes (round 5).  <1 024, false>: sixteen waves and rho(k) in a third LDS buffer -- one grid point per thread and pass for grids of up
// to 1 024 points: what a single replica or a handful waits for.  <512, true>: eight waves, two LDS buffers, rho(k) in the replica's charge
// grid in memory (written once, read three times: cache traffic) -- 56 KB and 512 threads for a 12 x 12 x 12 grid, which is what ONE retiring
// k_pair workgroup leaves free on a CU.  The first shape (83 KB, 1 024 threads) needs both of a CU's k_pair workgroups gone, so beside a pair
// launch it only ever started in that launch's tail: 899 instead of 90 us per step at 72 replicas, the k-space chain behind it stalled.
extern __shared__ double2 s_fft[];
template <int PP_SOLVE_TPB, bool BGLOBAL>
__global__ __launch_bounds__(PP_SOLVE_TPB) void k_pppm_solve(const SimDev *sims) {
  const SimDev &S = sims[blockIdx.x];
  const int nx = S.pg[0], ny = S.pg[1], nz = S.pg[2];
  if (nx == 0) return;
  __shared__ double s_red[8 * (PP_SOLVE_TPB / 64)];
  const int NG = nx * ny * nz;
  double2 *A = s_fft, *C = A + NG, *B = BGLOBAL ? (double2 *)S.pgrid : C + NG, *tw = (BGLOBAL ? C : B) + NG;   // twiddles: x at 0, y at nx, z at nx + ny
#ifdef PAIR_TIMING
  unsigned long long tq[8]; int ntq = 0;
#define PP_CLK() tq[ntq++] = __builtin_readcyclecounter()
#else
#define PP_CLK()
#endif
  PP_CLK();
  for (int k = threadIdx.x; k < nx + ny + nz; k += PP_SOLVE_TPB) {
    const int n = k < nx ? nx : (k < nx + ny ? ny : nz), j = k < nx ? k : (k < nx + ny ? k - nx : k - nx - ny);
    double sn, cs;
    sincospi(2.0 * (double)j / (double)n, &sn, &cs);
    tw[k] = make_double2(cs, -sn);
  }
  // Two workgroups per replica (gridDim.y == 2; the shape with rho(k) in LDS only): each makes the forward transform and ONE of the two
  // transforms back -- six one-dimensional passes instead of nine on the critical path of a batch of a few replicas, which waits for
  // this kernel.  Energy and virial are taken by the first.  The charge grid must leave zeroed but not before both
  // have read it: each takes a ticket once its copy is in LDS, the last one zeroes.
  const bool two = !BGLOBAL && gridDim.y == 2;
  const int c_lo = two ? (int)blockIdx.y : 0, c_hi = two ? (int)blockIdx.y + 1 : 2;
  double2 *rho = (double2 *)S.pgrid;
  for (int k = threadIdx.x; k < NG; k += PP_SOLVE_TPB) { A[k] = make_double2(rho[k].x, 0.0); if (!BGLOBAL && !two) rho[k] = make_double2(0.0, 0.0); }
  __syncthreads();
  if (two) {
    __shared__ int s_last;
    if (threadIdx.x == 0) {
      // (relaxed: this workgroup's reads of the grid are complete -- their values are in LDS, behind the barrier above --, and nothing it wrote is
      // read by the other; a release / acquire at device scope would write back and invalidate this XCD's L2)
      const int t = __hip_atomic_fetch_add(&S.sc->pppm_ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = (t == 1);
      if (t == 1) __hip_atomic_store(&S.sc->pppm_ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (s_last)
      for (int k = threadIdx.x; k < NG; k += PP_SOLVE_TPB) rho[k] = make_double2(0.0, 0.0);
  }
  PP_CLK();   // 1: twiddles + charge grid in
  // out[.., m, ..] = sum_k in[.., k, ..] w^(m k) along dimension dim (w = exp(-+2 pi i / n)); ends with a barrier.  A work item is one
  // line of the grid and one m in 0 .. n/2: it also produces the output n - m, whose twiddles are the complex conjugates (8 multiply-adds
  // and two LDS reads per k for two outputs; one output per item took 4 and two: 7 / 12 of the items, and a 12 x 12 x 12 grid is one
  // round of the 1 024 threads instead of two).  The items of a thread are the same in the three passes of a dimension: their line
  // offsets and m come from integer divisions done once.
  int it_m[3][2], it_base[3][2];
#pragma unroll
  for (int d = 0; d < 3; d++) {
    const int n = d == 0 ? nx : (d == 1 ? ny : nz), stride = d == 0 ? 1 : (d == 1 ? nx : nx * ny), nh = n / 2 + 1;
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const int it = (int)threadIdx.x + r * PP_SOLVE_TPB, L = it / nh;
      it_m[d][r] = it - L * nh;
      it_base[d][r] = (L / stride) * stride * n + (L % stride);
    }
  }
  auto pass = [&](const double2 *in, double2 *out, int dim, bool inverse) {
    const int n = dim == 0 ? nx : (dim == 1 ? ny : nz), stride = dim == 0 ? 1 : (dim == 1 ? nx : nx * ny), nh = n / 2 + 1;
    const int nitems = (NG / n) * nh;
    const double2 *t = tw + (dim == 0 ? 0 : (dim == 1 ? nx : nx + ny));
    for (int it = threadIdx.x, r = 0; it < nitems; it += PP_SOLVE_TPB, r++) {
      int m, base;
      if (r < 2) { m = r == 0 ? it_m[dim][0] : it_m[dim][1]; base = r == 0 ? it_base[dim][0] : it_base[dim][1]; }
      else { const int L = it / nh; m = it - L * nh; base = (L / stride) * stride * n + (L % stride); }
      double ar = 0.0, ai = 0.0, br = 0.0, bi = 0.0;
      int j = 0;   // (m k) mod n
#pragma unroll 4
      for (int k = 0; k < n; k++) {
        const double2 vv = in[base + k * stride], w = t[j];
        const double wi = inverse ? -w.y : w.y;
        ar = fma(vv.x, w.x, fma(-vv.y, wi, ar));
        ai = fma(vv.x, wi, fma(vv.y, w.x, ai));
        br = fma(vv.x, w.x, fma(vv.y, wi, br));      // conj(w): output n - m
        bi = fma(-vv.x, wi, fma(vv.y, w.x, bi));
        j += m;
        if (j >= n) j -= n;
      }
      out[base + m * stride] = make_double2(ar, ai);
      if (m != 0 && 2 * m != n) out[base + (n - m) * stride] = make_double2(br, bi);
    }
    __syncthreads();
  };
  pass(A, C, 0, false); pass(C, A, 1, false); pass(A, B, 2, false);   // B = rho(k) (BGLOBAL: in the charge grid's memory; the barrier that ends a pass orders it for the workgroup)
  PP_CLK();   // 2: forward passes
  BoxD b;
  box_derive(S.sc->box, b);
  double v[6] = {0, 0, 0, 0, 0, 0}, e[1] = {0};
  const size_t gs = (size_t)S.pgstride;
  double2 *field = (double2 *)S.pfield;
  // Two transforms back instead of three: the fields are real, so E_x(k) + i E_y(k) comes back as e_x(r) + i e_y(r) -- if both
  // spectra are Hermitian.  Modes with a Nyquist component are not (the grid index n/2 stands for -n/2 only), and taking the real
  // part of each field, as the three-transform form does, is the same as transforming the Hermitian part (X(k) + conj X(-k)) / 2:
  // that is what is packed here (for every other mode it equals X(k) to the last bit).
  for (int c = c_lo; c < c_hi; c++) {
    for (int idx = threadIdx.x; idx < NG; idx += PP_SOLVE_TPB) {
      double kv[3], vd[6] = {0, 0, 0, 0, 0, 0}, ed = 0.0;
      const double2 p = (c == 0) ? pppm_mode(S, b, idx, nx, ny, nz, B[idx], kv, v, e[0]) : pppm_mode(S, b, idx, nx, ny, nz, B[idx], kv, vd, ed);
      // spectrum of one component: (a + i b)(-i k) = b k - i a k
      if (c == 0) {
        const int m1 = idx % nx, m2 = (idx / nx) % ny, m3 = idx / (nx * ny);
        const int mid = (((nz - m3) % nz) * ny + (ny - m2) % ny) * nx + (nx - m1) % nx;
        double kw[3];
        const double2 q = pppm_mode(S, b, mid, nx, ny, nz, B[mid], kw, vd, ed);
        // Hermitian parts of X = (kx p.y, -kx p.x) and Y = (ky p.y, -ky p.x), with conj of the mirror mode's (kw q.y, -kw q.x)
        const double xr = 0.5 * (kv[0] * p.y + kw[0] * q.y), xi = 0.5 * (-kv[0] * p.x + kw[0] * q.x);
        const double yr = 0.5 * (kv[1] * p.y + kw[1] * q.y), yi = 0.5 * (-kv[1] * p.x + kw[1] * q.x);
        A[idx] = make_double2(xr - yi, xi + yr);   // X_H + i Y_H
      } else A[idx] = make_double2(kv[2] * p.y, -kv[2] * p.x);
    }
    __syncthreads();
    PP_CLK();   // 3, 5: spectra
    pass(A, C, 0, true); pass(C, A, 1, true); pass(A, C, 2, true);
    PP_CLK();   // 4, 6: inverse passes
    // (the fields leave as three REAL arrays at a spacing of gs doubles inside the buffer of the three complex grids: k_pppm_force
    // then stages 8 instead of 16 bytes per point)
    for (int idx = threadIdx.x; idx < NG; idx += PP_SOLVE_TPB) {
      const double2 r = C[idx];
      double *fr = (double *)field;
      if (c == 0) { fr[idx] = r.x; fr[gs + idx] = r.y; }
      else fr[2 * gs + idx] = r.x;
    }
    __syncthreads();
  }
  if (BGLOBAL)   // the charge grid leaves zeroed for the next spreading pass, as in the other shape
    for (int k = threadIdx.x; k < NG; k += PP_SOLVE_TPB) rho[k] = make_double2(0.0, 0.0);
  if (c_lo == 0) {   // (uniform over the workgroup)
    block_atomic_add_n<6, PP_SOLVE_TPB / 64>(v, S.sc->vir + P_KSPACE * 6, s_red);
    block_atomic_add_n<1, PP_SOLVE_TPB / 64>(e, S.sc->eng + P_KSPACE, s_red);
  }
#ifdef PAIR_TIMING
  PP_CLK();   // 7: field stores of the second round + sums
  if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && ntq == 8) {
    for (int k = 1; k < 8; k++) atomicAdd(&S.sc->dbg2[k - 1], tq[k] - tq[k - 1]);
    atomicAdd(&S.sc->dbg2[7], 1ull);
  }
#endif
#undef PP_CLK
}

// forces: the field (real parts of grids 1..3 after the inverse transforms) at the atom, by the assignment weights.  LDS = true:
// workgroup (s, r) stages the three real field grids of replica r in LDS once and serves the s-th of `split` atom ranges
// (consecutive atoms in consecutive lanes: neighbours read the same grid points, which the LDS broadcasts); LDS = false reads
// the grids through the caches.
// REALF: the fields are three real arrays at a spacing of gs doubles (written by k_pppm_solve) instead of the real parts of three complex grids
template <bool LDS, bool REALF = false>
__global__ __launch_bounds__(256) void k_pppm_force(const SimDev *sims, int split, int add) {
  const SimDev &S = sims[blockIdx.y];
  const int nx = S.pg[0], ny = S.pg[1], nz = S.pg[2];
  if (nx == 0) return;
  const int NG = nx * ny * nz;
  int chunk = (S.natoms + split - 1) / split;
  chunk = (chunk + 255) & ~255;
  const int a0 = (int)blockIdx.x * chunk, a1 = min(S.natoms, a0 + chunk);
  if (a0 >= a1) return;
  const size_t gs = (size_t)S.pgstride;
  const double2 *ex = (const double2 *)S.pfield, *ey = ex + gs, *ez = ey + gs;
  if (LDS) {
    if (REALF) {
      const double *fr = (const double *)S.pfield;
      for (int k = threadIdx.x; k < NG; k += 256) { s_grid[k] = fr[k]; s_grid[NG + k] = fr[gs + k]; s_grid[2 * NG + k] = fr[2 * gs + k]; }
    } else {
      for (int k = threadIdx.x; k < NG; k += 256) { s_grid[k] = ex[k].x; s_grid[NG + k] = ey[k].x; s_grid[2 * NG + k] = ez[k].x; }
    }
    __syncthreads();
  }
  BoxD b;
  box_derive(S.sc->box, b);
  box_uniform(b);
  for (int a = a0 + (int)threadIdx.x; a < a1; a += 256) {
    const double qa = S.q[a];
    if (qa == 0.0) {
      if (!add) { S.f[3 * a] = 0.0; S.f[3 * a + 1] = 0.0; S.f[3 * a + 2] = 0.0; }
      continue;
    }
    double l0, l1, l2;
    atom_lamda(S, b, a, l0, l1, l2);
    double wx[PP_ORDER], wy[PP_ORDER], wz[PP_ORDER];
    int gx[PP_ORDER], gy[PP_ORDER], gz[PP_ORDER];
    pppm_wrap(pppm_weights(l0 * nx, wx), nx, gx);
    pppm_wrap(pppm_weights(l1 * ny, wy), ny, gy);
    pppm_wrap(pppm_weights(l2 * nz, wz), nz, gz);
    double fx = 0.0, fy = 0.0, fz = 0.0;
#pragma unroll
    for (int c = 0; c < PP_ORDER; c++) {
#pragma unroll
      for (int bb = 0; bb < PP_ORDER; bb++) {
        const double zy = wz[c] * wy[bb];
        const int row = (gz[c] * ny + gy[bb]) * nx;
        double rx = 0.0, ry = 0.0, rz = 0.0;
#pragma unroll
        for (int k = 0; k < PP_ORDER; k++) {
          const int g = row + gx[k];
          if (LDS) { rx = fma(wx[k], s_grid[g], rx); ry = fma(wx[k], s_grid[NG + g], ry); rz = fma(wx[k], s_grid[2 * NG + g], rz); }
          else { rx = fma(wx[k], ex[g].x, rx); ry = fma(wx[k], ey[g].x, ry); rz = fma(wx[k], ez[g].x, rz); }
        }
        fx = fma(zy, rx, fx); fy = fma(zy, ry, fy); fz = fma(zy, rz, fz);
      }
    }
    const double qf = MD_QQRD2E * qa;
    if (add) { S.f[3 * a] += qf * fx; S.f[3 * a + 1] += qf * fy; S.f[3 * a + 2] += qf * fz; }
    else { S.f[3 * a] = qf * fx; S.f[3 * a + 1] = qf * fy; S.f[3 * a + 2] = qf * fz; }
  }
}


size_t mdk_pppm_lds_limit() { return 144 * 1024; }
// atom ranges per replica: enough workgroups to fill the 256 CUs several times over, none with fewer than 256 atoms
static inline int pppm_split(int ns, int maxatoms) {
  // small batches: down to one atom per thread (a single replica: spreading 27 -> 15 us, interpolation 27 -> 20 us)
  return std::max(1, std::min(std::min(ns < 32 ? 64 : 16, cdiv(2048, ns)), maxatoms / 256));
}
void mdk_pppm_spread(hipStream_t st, const SimDev *d, int ns, int maxgrid, int maxatoms, int zeroed, int maxgridp) {
  // maxgridp: the largest grid with five more points per x row (0: a grid of the batch has fewer than five points in x): the padded LDS copy
  static const bool padx_off = scema_env("SCEMA_MD_PPPM_PADX") && atoi(scema_env("SCEMA_MD_PPPM_PADX")) == 0;
  const bool padx = maxgridp > 0 && !padx_off && (size_t)maxgridp * sizeof(double) <= 36 * 1024;   // (7 / 19 kB of LDS for 12 x 12 x 12 points: no workgroup fewer)
  const size_t lds = (size_t)(padx ? maxgridp : maxgrid) * sizeof(double);
  const bool use_lds = lds <= mdk_pppm_lds_limit();
  const int split = pppm_split(ns, maxatoms);
  if ((!use_lds || split > 1) && !zeroed) hipLaunchKernelGGL(k_pppm_zero, grid2(cdiv(maxgrid, 256), ns), dim3(256), 0, st, d);
  if (!use_lds) {
    hipLaunchKernelGGL((k_pppm_spread<false, 256>), grid2(split, ns), dim3(256), 0, st, d, split);
    return;
  }
  static size_t optin_tab[16] = {0};
  size_t &optin = lds_optin_slot(optin_tab);
  if (lds > 64 * 1024 && lds > optin) { (void)hipFuncSetAttribute((const void *)k_pppm_spread<true, PP_TPB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); optin = lds; }
  // a small grid leaves room for several workgroups per CU; a large one gets the CU to itself and brings its own sixteen waves
  if (padx) hipLaunchKernelGGL((k_pppm_spread<true, 256, true>), grid2(split, ns), dim3(256), lds, st, d, split);
  else if (lds <= 36 * 1024) hipLaunchKernelGGL((k_pppm_spread<true, 256>), grid2(split, ns), dim3(256), lds, st, d, split);
  else hipLaunchKernelGGL((k_pppm_spread<true, PP_TPB>), grid2(split, ns), dim3(PP_TPB), lds, st, d, split);
}
int mdk_pppm_solve_max() { return PP_SOLVE_MAX; }
void mdk_pppm_solve(hipStream_t st, const SimDev *d, int ns, int maxgrid, int maxdims) {
  // batches: the shape that fits beside a pair workgroup; a few replicas: the one with the most threads per replica
  static const int shape_env = scema_env("SCEMA_MD_PPPM_SOLVE_WIDE") ? atoi(scema_env("SCEMA_MD_PPPM_SOLVE_WIDE")) : -1;
  const bool wide = shape_env < 0 ? ns < 8 : shape_env != 0;
  static const int two_env = scema_env("SCEMA_MD_PPPM_SOLVE_TWO") ? atoi(scema_env("SCEMA_MD_PPPM_SOLVE_TWO")) : -1;
  const bool two = wide && (two_env < 0 ? ns < 8 : two_env != 0);   // two workgroups per replica, one per transform back (the shape with rho(k) in LDS)
  const size_t lds = ((wide ? 3 : 2) * (size_t)maxgrid + (size_t)maxdims) * sizeof(double2);
  static size_t optin_tab[2][16] = {{0}, {0}};
  size_t &optin = lds_optin_slot(optin_tab[wide ? 1 : 0]);
  if (wide) {
    if (lds > 64 * 1024 && lds > optin) { (void)hipFuncSetAttribute((const void *)k_pppm_solve<1024, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); optin = lds; }
    hipLaunchKernelGGL((k_pppm_solve<1024, false>), dim3(ns, two ? 2 : 1), dim3(1024), lds, st, d);
  } else {
    if (lds > 64 * 1024 && lds > optin) { (void)hipFuncSetAttribute((const void *)k_pppm_solve<512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); optin = lds; }
    hipLaunchKernelGGL((k_pppm_solve<512, true>), dim3(ns), dim3(512), lds, st, d);
  }
}
void mdk_pppm_gf(hipStream_t st, const SimDev *d, int ns, int maxgrid) { hipLaunchKernelGGL(k_pppm_gf, grid2(cdiv(maxgrid, 256), ns), dim3(256), 0, st, d); }
void mdk_pppm_poisson(hipStream_t st, const SimDev *d, int ns, int maxgrid) { hipLaunchKernelGGL(k_pppm_poisson, grid2(cdiv(maxgrid, 256), ns), dim3(256), 0, st, d); }
void mdk_pppm_force(hipStream_t st, const SimDev *d, int ns, int maxgrid, int maxatoms, int add, int real_fields) {
  const size_t lds = 3 * (size_t)maxgrid * sizeof(double);
  if (lds > mdk_pppm_lds_limit()) {
    hipLaunchKernelGGL((k_pppm_force<false, false>), grid2(cdiv(maxatoms, 256), ns), dim3(256), 0, st, d, cdiv(maxatoms, 256), add);
    return;
  }
  static size_t optin_tab[16] = {0};
  size_t &optin = lds_optin_slot(optin_tab);
  if (lds > 64 * 1024 && lds > optin) {
    (void)hipFuncSetAttribute((const void *)k_pppm_force<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void *)k_pppm_force<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    optin = lds;
  }
  const int split = pppm_split(ns, maxatoms);
  if (real_fields) hipLaunchKernelGGL((k_pppm_force<true, true>), grid2(split, ns), dim3(256), lds, st, d, split, add);
  else hipLaunchKernelGGL((k_pppm_force<true, false>), grid2(split, ns), dim3(256), lds, st, d, split, add);
}
