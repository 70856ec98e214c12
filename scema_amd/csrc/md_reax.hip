// md_reax.hip -- gfx950 kernels of the ReaxFF force field path (BASELINE config 5).
//
// Replaces, for replicas run with force_field "reax", what lammps_scripts_reax/in.strain.lammps:10-12 and
// ELASTIC/potential.mod.lammps:5-7 select in LAMMPS: pair_style reax/c + fix qeq/reax 1 0.0 10.0 1e-6 [LAMMPS-ext].
// The integrator, thermostat, fix deform, pressure sampling and the batch machinery are the ones of the OPLS path
// (md_kernels.hip); only the force stage differs.
//
// Layout: all per-atom lists are entry-major rows ([k][i], reax/rx_types.h), one lane per atom: a wave reads 64 consecutive
// entries at every step of its row walk.  One launch covers every replica of the batch (blockIdx.y).
//   k_rx_prepare ........ box -> view, rebuild bookkeeping, zero the energy parts
//   k_rx_wrap, k_rx_neigh  neighbour rows inside cutoff + skin, rebuilt when an atom has moved half the skin
//                         (LAMMPS rebuilds every step, `neigh_modify every 1 delay 0 check no`: same pairs inside the cutoff)
//   k_rx_hrow, k_rx_qeq .. charge equilibration: matrix rows in HBM, the two conjugate-gradient solves of one replica in
//                         one workgroup (no host round trips), both right-hand sides per sweep over the matrix
//   k_rx_bonds ... k_rx_back2  bond orders, energy terms, reverse-mode forces (reax/rx_core.h)
#include <hip/hip_runtime.h>

#include "md_device.h"
#include "md_kernels.h"
#include "md_reax.h"
#include "md_types.h"
#include "reax/rx_core.h"

#define RX_TPB 128

// ------------------------------------------------------------------------------------------------------------------
__global__ void k_rx_prepare(const SimDev *sims, RxView *views) {
  const SimDev &S = sims[blockIdx.x];
  SimScalars &sc = *S.sc;
  RxView &V = views[blockIdx.x];
  if (threadIdx.x == 0) {
    V.h[0] = sc.box[3] - sc.box[0]; V.h[1] = sc.box[4] - sc.box[1]; V.h[2] = sc.box[5] - sc.box[2];
    V.h[3] = sc.box[8]; V.h[4] = sc.box[7]; V.h[5] = sc.box[6];
    V.lo[0] = sc.box[0]; V.lo[1] = sc.box[1]; V.lo[2] = sc.box[2];
    if (sc.rebuild) {
      sc.ago = 0;
      sc.nbuilds += 1;
      // corners of the box at build time (the neighbour trigger of k_pre takes their motion off the skin)
      int k = 0;
      for (int iz = 0; iz < 2; iz++)
        for (int iy = 0; iy < 2; iy++)
          for (int ix = 0; ix < 2; ix++) {
            sc.corners_hold[3 * k + 0] = V.h[0] * ix + V.h[5] * iy + V.h[4] * iz + V.lo[0];
            sc.corners_hold[3 * k + 1] = V.h[1] * iy + V.h[3] * iz + V.lo[1];
            sc.corners_hold[3 * k + 2] = V.h[2] * iz + V.lo[2];
            k++;
          }
    }
  }
  if (threadIdx.x < RX_NPART) V.eparts[threadIdx.x] = 0.0;
}

// wrap into the box at a rebuild (LAMMPS does the same when it reneighbours); image counts keep the unwrapped information
__global__ __launch_bounds__(TPB) void k_rx_wrap(const SimDev *sims, RxView *views) {
  const SimDev &S = sims[blockIdx.y];
  if (!S.sc->rebuild) return;
  const RxView &V = views[blockIdx.y];
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= S.natoms) return;
  double x0 = S.x[3 * i], x1 = S.x[3 * i + 1], x2 = S.x[3 * i + 2];
  const double l2 = (x2 - V.lo[2]) / V.h[2];
  const double l1 = ((x1 - V.lo[1]) - V.h[3] * l2) / V.h[1];
  const double l0 = ((x0 - V.lo[0]) - V.h[5] * l1 - V.h[4] * l2) / V.h[0];
  const double w0 = floor(l0), w1 = floor(l1), w2 = floor(l2);
  x0 -= w0 * V.h[0] + w1 * V.h[5] + w2 * V.h[4];
  x1 -= w1 * V.h[1] + w2 * V.h[3];
  x2 -= w2 * V.h[2];
  S.x[3 * i] = x0; S.x[3 * i + 1] = x1; S.x[3 * i + 2] = x2;
  S.xhold[3 * i] = x0; S.xhold[3 * i + 1] = x1; S.xhold[3 * i + 2] = x2;
  S.wrapn[3 * i] += (int)w0; S.wrapn[3 * i + 1] += (int)w1; S.wrapn[3 * i + 2] += (int)w2;
}

// Neighbour rows by tiles: every lane owns an atom and walks all atoms of the replica, staged 256 at a time through LDS.
// O(N^2) distance tests per rebuild (about 25 FP64 operations each), rebuilt every some tens of steps: a few per cent of the
// step next to the force kernels, any box shape, and rows that come out sorted by partner index (deterministic).
// mimg = 0: boxes at least two list radii wide, minimum image; otherwise all images up to mimg[d] boxes away.
__global__ __launch_bounds__(TPB) void k_rx_neigh(const SimDev *sims, RxView *views, double rlist) {
  const SimDev &S = sims[blockIdx.y];
  if (!S.sc->rebuild) return;
  const RxView &V = views[blockIdx.y];
  __shared__ double s_x[TPB], s_y[TPB], s_z[TPB];
  const int i = blockIdx.x * TPB + threadIdx.x, n = V.n, np = V.npad;
  const bool live = i < n;
  const double xi = live ? S.x[3 * i] : 0.0, yi = live ? S.x[3 * i + 1] : 0.0, zi = live ? S.x[3 * i + 2] : 0.0;
  const double rl2 = rlist * rlist;
  const double ih0 = 1.0 / V.h[0], ih1 = 1.0 / V.h[1], ih2 = 1.0 / V.h[2];
  const int m0 = V.mimg[0], m1 = V.mimg[1], m2 = V.mimg[2];
  const bool minimage = (m0 | m1 | m2) == 0;
  int cnt = 0;
  bool full = false;
  for (int j0 = 0; j0 < n; j0 += TPB) {
    __syncthreads();
    const int jl = j0 + threadIdx.x;
    if (jl < n) { s_x[threadIdx.x] = S.x[3 * jl]; s_y[threadIdx.x] = S.x[3 * jl + 1]; s_z[threadIdx.x] = S.x[3 * jl + 2]; }
    __syncthreads();
    if (!live) continue;
    const int jn = min(TPB, n - j0);
    for (int jj = 0; jj < jn; jj++) {
      const int j = j0 + jj;
      double dx = s_x[jj] - xi, dy = s_y[jj] - yi, dz = s_z[jj] - zi;
      if (minimage) {
        if (j == i) continue;
        const double n2 = rint(dz * ih2);
        dz -= n2 * V.h[2]; dy -= n2 * V.h[3]; dx -= n2 * V.h[4];
        const double n1 = rint(dy * ih1);
        dy -= n1 * V.h[1]; dx -= n1 * V.h[5];
        const double n0 = rint(dx * ih0);
        dx -= n0 * V.h[0];
        if (dx * dx + dy * dy + dz * dz > rl2) continue;
        if (cnt >= V.maxnb) { full = true; continue; }
        const int code = (2 - (int)n0) + 5 * (2 - (int)n1) + 25 * (2 - (int)n2);
        V.nb[(size_t)cnt * np + i] = j | (code << 24);
        cnt++;
      } else {
        for (int sz = -m2; sz <= m2; sz++)
          for (int sy = -m1; sy <= m1; sy++)
            for (int sx = -m0; sx <= m0; sx++) {
              if (j == i && sx == 0 && sy == 0 && sz == 0) continue;
              const double ex = dx + sx * V.h[0] + sy * V.h[5] + sz * V.h[4], ey = dy + sy * V.h[1] + sz * V.h[3], ez = dz + sz * V.h[2];
              if (ex * ex + ey * ey + ez * ez > rl2) continue;
              if (cnt >= V.maxnb) { full = true; continue; }
              V.nb[(size_t)cnt * np + i] = j | (((sx + 2) + 5 * (sy + 2) + 25 * (sz + 2)) << 24);
              cnt++;
            }
      }
    }
  }
  if (live) {
    V.nb_cnt[i] = cnt;
    if (full) atomicOr(V.overflow, 1);
    atomicMax(&S.sc->maxneigh_seen, cnt);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// charge equilibration
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void k_rx_hrow(const SimDev *sims, const RxView *views, const RxParams *P) {
  const RxView V = views[blockIdx.y];
  (void)sims;
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i < V.n) rx_qeq_row(P, &V, i);
}

#define QEQ_TPB 1024
__device__ __forceinline__ void qeq_reduce2(double &a, double &b, double *lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  a = wave_sum(a);
  b = wave_sum(b);
  __syncthreads();
  if (lane == 0) { lds[wave] = a; lds[16 + wave] = b; }
  __syncthreads();
  double sa = 0.0, sb = 0.0;
#pragma unroll
  for (int w = 0; w < QEQ_TPB / 64; w++) { sa += lds[w]; sb += lds[16 + w]; }
  a = sa;
  b = sb;
}
// fix qeq/reax: H s = -chi and H t = -1 by Jacobi-preconditioned conjugate gradients from the extrapolated previous solutions
// (init_matvec: cubic for s, quadratic for t), until sqrt(r.p)/|b| <= tol for each; q = s - (sum s / sum t) t.
// work: [8][npad] doubles per replica (r, d, Hd, for both systems + spare)
__global__ __launch_bounds__(QEQ_TPB) void k_rx_qeq(const SimDev *sims, const RxView *views, const RxParams *P, double tol, int maxiter) {
  const RxView V = views[blockIdx.x];
  (void)sims;
  __shared__ double s_red[32];
  const int n = V.n, np = V.npad, tid = threadIdx.x;
  double *s = V.s, *t = V.t;
  double *rs = V.qwork, *rt = V.qwork + np, *ds = V.qwork + 2 * (size_t)np, *dt = V.qwork + 3 * (size_t)np;
  // initial guesses
  for (int i = tid; i < n; i += QEQ_TPB) {
    const double *sh = V.s_hist, *th = V.t_hist;
    s[i] = 4.0 * (sh[i] + sh[2 * (size_t)np + i]) - (6.0 * sh[(size_t)np + i] + sh[3 * (size_t)np + i]);
    t[i] = 3.0 * (th[i] - th[(size_t)np + i]) + th[2 * (size_t)np + i];
  }
  __syncthreads();
  double bs2 = 0.0, bt2 = 0.0, sig_s = 0.0, sig_t = 0.0;
  for (int i = tid; i < n; i += QEQ_TPB) {
    const int ti = V.rtype[i];
    const double eta = P->sbp[ti].eta, chi = P->sbp[ti].chi;
    double ys = eta * s[i], yt = eta * t[i];
    const int cnt = V.nb_cnt[i];
    for (int k = 0; k < cnt; k++) {
      const size_t o = (size_t)k * np + i;
      const int j = V.nb[o] & RX_JMASK;
      const double hv = V.hval[o];
      ys += hv * s[j];
      yt += hv * t[j];
    }
    const double r1 = -chi - ys, r2 = -1.0 - yt;
    rs[i] = r1; rt[i] = r2;
    ds[i] = r1 / eta; dt[i] = r2 / eta;
    bs2 += chi * chi; bt2 += 1.0;
    sig_s += r1 * r1 / eta; sig_t += r2 * r2 / eta;
  }
  qeq_reduce2(bs2, bt2, s_red);
  qeq_reduce2(sig_s, sig_t, s_red);
  const double bn_s = sqrt(bs2), bn_t = sqrt(bt2);
  bool run_s = sqrt(sig_s) / bn_s > tol, run_t = sqrt(sig_t) / bn_t > tol;
  int it = 0;
  for (; it < maxiter && (run_s || run_t); it++) {
    // q = H d for both systems in one sweep over the rows; kept in registers (every thread owns the same rows each sweep)
    double dq_s = 0.0, dq_t = 0.0;
    for (int i = tid; i < n; i += QEQ_TPB) {
      const double eta = P->sbp[V.rtype[i]].eta;
      double ys = eta * ds[i], yt = eta * dt[i];
      const int cnt = V.nb_cnt[i];
      for (int k = 0; k < cnt; k++) {
        const size_t o = (size_t)k * np + i;
        const int j = V.nb[o] & RX_JMASK;
        const double hv = V.hval[o];
        ys += hv * ds[j];
        yt += hv * dt[j];
      }
      V.qwork[4 * (size_t)np + i] = ys;
      V.qwork[5 * (size_t)np + i] = yt;
      dq_s += ds[i] * ys;
      dq_t += dt[i] * yt;
    }
    qeq_reduce2(dq_s, dq_t, s_red);
    const double al_s = run_s ? sig_s / dq_s : 0.0, al_t = run_t ? sig_t / dq_t : 0.0;
    double sn_s = 0.0, sn_t = 0.0;
    for (int i = tid; i < n; i += QEQ_TPB) {
      const double eta = P->sbp[V.rtype[i]].eta;
      if (run_s) {
        s[i] += al_s * ds[i];
        const double r1 = rs[i] - al_s * V.qwork[4 * (size_t)np + i];
        rs[i] = r1;
        sn_s += r1 * r1 / eta;
      }
      if (run_t) {
        t[i] += al_t * dt[i];
        const double r2 = rt[i] - al_t * V.qwork[5 * (size_t)np + i];
        rt[i] = r2;
        sn_t += r2 * r2 / eta;
      }
    }
    qeq_reduce2(sn_s, sn_t, s_red);
    const double be_s = run_s ? sn_s / sig_s : 0.0, be_t = run_t ? sn_t / sig_t : 0.0;
    for (int i = tid; i < n; i += QEQ_TPB) {
      const double eta = P->sbp[V.rtype[i]].eta;
      if (run_s) ds[i] = rs[i] / eta + be_s * ds[i];
      if (run_t) dt[i] = rt[i] / eta + be_t * dt[i];
    }
    if (run_s) { sig_s = sn_s; run_s = sqrt(sig_s) / bn_s > tol; }
    if (run_t) { sig_t = sn_t; run_t = sqrt(sig_t) / bn_t > tol; }
    __syncthreads();
  }
  double ss = 0.0, st = 0.0;
  for (int i = tid; i < n; i += QEQ_TPB) { ss += s[i]; st += t[i]; }
  qeq_reduce2(ss, st, s_red);
  const double u = ss / st;
  for (int i = tid; i < n; i += QEQ_TPB) {
    V.q[i] = s[i] - u * t[i];
    double *sh = V.s_hist, *th = V.t_hist;
    sh[3 * (size_t)np + i] = sh[2 * (size_t)np + i]; sh[2 * (size_t)np + i] = sh[(size_t)np + i]; sh[(size_t)np + i] = sh[i]; sh[i] = s[i];
    th[2 * (size_t)np + i] = th[(size_t)np + i]; th[(size_t)np + i] = th[i]; th[i] = t[i];
  }
  if (tid == 0) {
    V.qstat[0] += it;
    V.qstat[1] += 1;
    if (run_s || run_t) atomicOr(V.overflow, 4);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// force passes: one lane per atom around the functions of reax/rx_core.h
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void rx_flush(double (&e)[RX_NPART], double (&w)[6], const RxView &V, SimScalars &sc, int vpart) {
#pragma unroll
  for (int k = 0; k < RX_NPART; k++) {
    const double s = wave_sum(e[k]);
    if ((threadIdx.x & 63) == 0 && s != 0.0) atomicAdd(&V.eparts[k], s);
  }
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const double s = wave_sum(w[k]);
    if ((threadIdx.x & 63) == 0 && s != 0.0) atomicAdd(&sc.vir[vpart * 6 + k], s);
  }
}

__global__ __launch_bounds__(TPB) void k_rx_bonds(const RxView *views, const RxParams *P) {
  const RxView V = views[blockIdx.y];
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i < V.n) rx_bonds_prime(P, &V, i);
}
__global__ __launch_bounds__(TPB) void k_rx_rev(const RxView *views) {
  const RxView V = views[blockIdx.y];
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i < V.n) rx_bonds_rev(&V, i);
}
__global__ __launch_bounds__(TPB) void k_rx_corr(const RxView *views, const RxParams *P) {
  const RxView V = views[blockIdx.y];
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i < V.n) rx_bonds_corrected(P, &V, i);
}
// pass: 0 atom terms, 1 angles, 2 torsions, 3 hydrogen bonds, 4 non-bonded
template <int PASS>
__global__ __launch_bounds__(RX_TPB) void k_rx_terms(const SimDev *sims, const RxView *views, const RxParams *P) {
  const RxView V = views[blockIdx.y];
  const int i = blockIdx.x * RX_TPB + threadIdx.x;
  double e[RX_NPART], w[6];
#pragma unroll
  for (int k = 0; k < RX_NPART; k++) e[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 6; k++) w[k] = 0.0;
  if (i < V.n) {
    if (PASS == 0) rx_atom_terms(P, &V, i, e);
    if (PASS == 1) rx_angle_terms(P, &V, i, e, w);
    if (PASS == 2) rx_torsion_terms(P, &V, i, e, w);
    if (PASS == 3) rx_hbond_terms(P, &V, i, e, w);
    if (PASS == 4) rx_nonbonded(P, &V, i, e, w);
  }
  rx_flush(e, w, V, *sims[blockIdx.y].sc, PASS == 4 ? P_LJ : (PASS == 1 ? P_ANGLE : (PASS == 2 ? P_DIHEDRAL : (PASS == 3 ? P_IMPROPER : P_BOND))));
}
__global__ __launch_bounds__(TPB) void k_rx_back1(const RxView *views, const RxParams *P) {
  const RxView V = views[blockIdx.y];
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i < V.n) rx_back_corr(P, &V, i);
}
__global__ __launch_bounds__(TPB) void k_rx_back2(const SimDev *sims, const RxView *views, const RxParams *P) {
  const RxView V = views[blockIdx.y];
  const int i = blockIdx.x * TPB + threadIdx.x;
  double e[RX_NPART], w[6];
#pragma unroll
  for (int k = 0; k < RX_NPART; k++) e[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 6; k++) w[k] = 0.0;
  if (i < V.n) rx_back_force(P, &V, i, w);
  rx_flush(e, w, V, *sims[blockIdx.y].sc, P_BOND);
}
// energies of the step into the engine's parts
__global__ void k_rx_finish(const SimDev *sims, const RxView *views) {
  const RxView &V = views[blockIdx.x];
  SimScalars &sc = *sims[blockIdx.x].sc;
  if (threadIdx.x != 0) return;
  const double *e = V.eparts;
  sc.eng[P_BOND] = e[RX_E_BOND] + e[RX_E_LP] + e[RX_E_OVER] + e[RX_E_UNDER];
  sc.eng[P_ANGLE] = e[RX_E_ANGLE] + e[RX_E_PEN] + e[RX_E_COA];
  sc.eng[P_DIHEDRAL] = e[RX_E_TORS] + e[RX_E_CONJ];
  sc.eng[P_IMPROPER] = e[RX_E_HB];
  sc.eng[P_LJ] = e[RX_E_VDW];
  sc.eng[P_COUL] = e[RX_E_COUL] + e[RX_E_POL];
  if (*V.overflow) atomicOr(&sc.overflow, (*V.overflow & 3) ? 1 : 32);
}
// zero the charge-equilibration history at the start of a run (a new fix qeq/reax starts from zeros)
__global__ __launch_bounds__(TPB) void k_rx_phase_init(const RxView *views) {
  const RxView V = views[blockIdx.y];
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i < V.npad) {
    for (int k = 0; k < 4; k++) V.s_hist[(size_t)k * V.npad + i] = 0.0;
    for (int k = 0; k < 3; k++) V.t_hist[(size_t)k * V.npad + i] = 0.0;
  }
  if (i == 0) { V.qstat[0] = 0; V.qstat[1] = 0; *V.overflow = 0; }
}

static inline dim3 g2(int nx, int ns) { return dim3((unsigned)nx, (unsigned)ns, 1); }
static inline int cdv(int a, int b) { return (a + b - 1) / b; }

void mdk_reax_phase_init(hipStream_t st, const RxView *v, int ns, int maxpad) {
  hipLaunchKernelGGL(k_rx_phase_init, g2(cdv(maxpad, TPB), ns), dim3(TPB), 0, st, v);
}
void mdk_reax_forces(hipStream_t st, const SimDev *d, RxView *v, const RxParams *P, int ns, int maxatoms, double rlist, double qeq_tol, int qeq_maxiter, int terms) {
  const dim3 ga = g2(cdv(maxatoms, TPB), ns), gr = g2(cdv(maxatoms, RX_TPB), ns);
  hipLaunchKernelGGL(k_rx_prepare, dim3(ns), dim3(64), 0, st, d, v);
  hipLaunchKernelGGL(k_rx_wrap, ga, dim3(TPB), 0, st, d, v);
  hipLaunchKernelGGL(k_rx_neigh, ga, dim3(TPB), 0, st, d, v, rlist);
  hipLaunchKernelGGL(k_rx_hrow, ga, dim3(TPB), 0, st, d, v, P);
  hipLaunchKernelGGL(k_rx_qeq, dim3(ns), dim3(QEQ_TPB), 0, st, d, v, P, qeq_tol, qeq_maxiter);
  hipLaunchKernelGGL(k_rx_bonds, ga, dim3(TPB), 0, st, v, P);
  hipLaunchKernelGGL(k_rx_rev, ga, dim3(TPB), 0, st, v);
  hipLaunchKernelGGL(k_rx_corr, ga, dim3(TPB), 0, st, v, P);
  if (terms & 1) hipLaunchKernelGGL(k_rx_terms<0>, gr, dim3(RX_TPB), 0, st, d, v, P);
  if (terms & 2) hipLaunchKernelGGL(k_rx_terms<1>, gr, dim3(RX_TPB), 0, st, d, v, P);
  if (terms & 4) hipLaunchKernelGGL(k_rx_terms<2>, gr, dim3(RX_TPB), 0, st, d, v, P);
  if (terms & 8) hipLaunchKernelGGL(k_rx_terms<3>, gr, dim3(RX_TPB), 0, st, d, v, P);
  if (terms & 16) hipLaunchKernelGGL(k_rx_terms<4>, gr, dim3(RX_TPB), 0, st, d, v, P);
  hipLaunchKernelGGL(k_rx_back1, ga, dim3(TPB), 0, st, v, P);
  hipLaunchKernelGGL(k_rx_back2, ga, dim3(TPB), 0, st, d, v, P);
  hipLaunchKernelGGL(k_rx_finish, dim3(ns), dim3(64), 0, st, d, v);
}
