// md_bonded.hip -- bonded terms and special pairs, atom-centric and atomic-free.
//
// Round-1 profile: the term-centric kernels (one thread per bond/angle/dihedral, FP64 atomics into
// f) were atomic-rate bound: dihedrals alone took 14 % of the GPU time.  Here every atom walks the
// list of terms it takes part in (built once per topology, sorted by kind), recomputes each term
// and keeps only the force on itself: a dihedral is evaluated four times, which is ~30x cheaper
// than its twelve atomics, the force update is a plain read-modify-write of the atom's own f, and
// the result is bitwise reproducible.
//
// Virial bookkeeping: a term's virial is sum_a (r_a - r_ref) (x) F_a over its atoms; each atom adds
// its own summand, with r_ref = atom 2 of the term for torsions, the vertex for angles, atom 2 for
// bonds / pairs.  Energies (parity hook) are counted by role 0 only.
//
// Styles (reference in.set.lammps:44-57, in.init.lammps:31): bond harmonic, angle harmonic,
// dihedral opls, improper harmonic, special_bonds weights on lj/cut/coul/long.
#include <hip/hip_runtime.h>

#include "md_device.h"
#include "md_kernels.h"

__device__ __forceinline__ void vt(double *v, const double *a, const double *f) {
  v[0] += a[0] * f[0]; v[1] += a[1] * f[1]; v[2] += a[2] * f[2];
  v[3] += a[0] * f[1]; v[4] += a[0] * f[2]; v[5] += a[1] * f[2];
}
__device__ __forceinline__ void cross3(const double *a, const double *b, double *c) {
  c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
// 1/sqrt(x) to 1 ulp in 6 instructions (hardware estimate + one third-order correction); replaces the
// sqrt + divide pairs of the textbook formulas
__device__ __forceinline__ double rsq64(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double e = fma(-x * y, y, 1.0);
  return fma(y, e * fma(0.375, e, 0.5), y);
}

// PARTS: also split virial / energy per part (parity hook); otherwise one lumped virial
template <bool PARTS>
__global__ __launch_bounds__(TPB) void k_bonded_atom(const SimDev *__restrict__ sims) {
  const SimDev &S = sims[blockIdx.y];
  SimScalars &sc = *S.sc;
  __shared__ double s_red[8 * (TPB / 64)];
  // atoms are visited in the order aterm_order: sorted by their term-count signature, so the lanes of
  // a wave walk lists of the same shape (all-carbon waves, all-hydrogen waves) instead of idling
  const int tid = blockIdx.x * TPB + threadIdx.x;
  const int i = (tid < S.natoms) ? S.aterm_order[tid] : S.natoms;
  double ftot[3] = {0, 0, 0};
  double vsum[6] = {0, 0, 0, 0, 0, 0};
  if (i < S.natoms) {
    BoxD b;
    box_derive(sc.box, b);
    const double *x = S.x;
    const int eb = S.aterm_start[i], ee = S.aterm_start[i + 1];
    for (int t = eb; t < ee; t++) {
      const int packed = S.aterm[t];
      const int kind = packed & 7, role = (packed >> 3) & 3, m = packed >> 5;
      double fo[3] = {0, 0, 0}, v[6] = {0, 0, 0, 0, 0, 0}, v2[6] = {0, 0, 0, 0, 0, 0};
      double en = 0.0, en2 = 0.0;
      int part = P_BOND;
      if (kind == AT_BOND || kind == AT_BOND_SHAKEN) {
        if (kind == AT_BOND_SHAKEN && S.use_shake) continue;  // fix shake switches these bonds off
        const int i1 = S.bond_at[2 * m], i2 = S.bond_at[2 * m + 1];
        const double K = S.bond_cf[2 * m], r0 = S.bond_cf[2 * m + 1];
        double d[3] = {x[3 * i1] - x[3 * i2], x[3 * i1 + 1] - x[3 * i2 + 1], x[3 * i1 + 2] - x[3 * i2 + 2]};
        minimg(b, d[0], d[1], d[2]);
        const double rsq = dot3(d, d);
        const double rinv = (rsq > 0.0) ? rsq64(rsq) : 0.0;
        const double r = rsq * rinv;
        const double dr = r - r0, rk = K * dr;
        const double fb = -2.0 * rk * rinv;
        const double sgn = (role == 0) ? 1.0 : -1.0;
        for (int k = 0; k < 3; k++) fo[k] = sgn * d[k] * fb;
        if (role == 0) { vt(v, d, fo); en = rk * dr; }
      } else if (kind == AT_ANGLE) {
        part = P_ANGLE;
        const int i1 = S.angle_at[3 * m], i2 = S.angle_at[3 * m + 1], i3 = S.angle_at[3 * m + 2];
        const double K = S.angle_cf[2 * m], th0 = S.angle_cf[2 * m + 1];
        double d1[3], d2[3];
        for (int k = 0; k < 3; k++) { d1[k] = x[3 * i1 + k] - x[3 * i2 + k]; d2[k] = x[3 * i3 + k] - x[3 * i2 + k]; }
        minimg(b, d1[0], d1[1], d1[2]); minimg(b, d2[0], d2[1], d2[2]);
        const double rsq1 = dot3(d1, d1), rsq2 = dot3(d2, d2);
        const double r1i = rsq64(rsq1), r2i = rsq64(rsq2);
        double c = dot3(d1, d2) * (r1i * r2i);
        c = fmin(1.0, fmax(-1.0, c));
        const double s2 = 1.0 - c * c;
        const double sni = (s2 > 1.0e-6) ? rsq64(s2) : 1000.0;  // 1/sin(theta), sin clamped at 0.001
        const double dth = acos(c) - th0, tk = K * dth;
        const double a = -2.0 * tk * sni;
        const double a11 = a * c * (r1i * r1i), a12 = -a * (r1i * r2i), a22 = a * c * (r2i * r2i);
        double f1[3], f3[3];
        for (int k = 0; k < 3; k++) { f1[k] = a11 * d1[k] + a12 * d2[k]; f3[k] = a22 * d2[k] + a12 * d1[k]; }
        if (role == 0) { for (int k = 0; k < 3; k++) fo[k] = f1[k]; vt(v, d1, f1); en = tk * dth; }
        else if (role == 2) { for (int k = 0; k < 3; k++) fo[k] = f3[k]; vt(v, d2, f3); }
        else { for (int k = 0; k < 3; k++) fo[k] = -(f1[k] + f3[k]); }
      } else if (kind == AT_DIHEDRAL || kind == AT_IMPROPER) {
        part = (kind == AT_DIHEDRAL) ? P_DIHEDRAL : P_IMPROPER;
        const int *at = (kind == AT_DIHEDRAL) ? S.dihedral_at + 4 * m : S.improper_at + 4 * m;
        // F=r1-r2, G=r2-r3, H=r4-r3, A=FxG, B=HxG, c=A.B/(|A||B|)
        double F[3], G[3], H[3];
        for (int k = 0; k < 3; k++) {
          F[k] = x[3 * at[0] + k] - x[3 * at[1] + k];
          G[k] = x[3 * at[1] + k] - x[3 * at[2] + k];
          H[k] = x[3 * at[3] + k] - x[3 * at[2] + k];
        }
        minimg(b, F[0], F[1], F[2]); minimg(b, G[0], G[1], G[2]); minimg(b, H[0], H[1], H[2]);
        double A[3], B[3];
        cross3(F, G, A); cross3(H, G, B);
        const double a2 = dot3(A, A), b2 = dot3(B, B);
        const double ia = rsq64(a2), ib = rsq64(b2);
        const double iab = ia * ib;
        double c = dot3(A, B) * iab;
        c = fmin(1.0, fmax(-1.0, c));
        const double ca = c * (ia * ia), cb = c * (ib * ib);
        double gA[3], gB[3];
        for (int k = 0; k < 3; k++) { gA[k] = B[k] * iab - ca * A[k]; gB[k] = A[k] * iab - cb * B[k]; }
        double dEdc;
        if (kind == AT_DIHEDRAL) {
          const double *K = S.dihedral_cf + 4 * m;
          const double c2 = c * c;
          dEdc = 0.5 * (K[0] - K[1] * 4.0 * c + K[2] * (12.0 * c2 - 3.0) - K[3] * (32.0 * c2 * c - 16.0 * c));
          if (role == 0) {
            const double cos2 = 2.0 * c2 - 1.0, cos3 = (4.0 * c2 - 3.0) * c, cos4 = 8.0 * c2 * c2 - 8.0 * c2 + 1.0;
            en = 0.5 * (K[0] * (1.0 + c) + K[1] * (1.0 - cos2) + K[2] * (1.0 + cos3) + K[3] * (1.0 - cos4));
          }
        } else {
          const double K = S.improper_cf[2 * m], chi0 = S.improper_cf[2 * m + 1];
          double sn = sqrt(1.0 - c * c);
          if (sn < 0.001) sn = 0.001;
          const double dchi = acos(c) - chi0;
          dEdc = -2.0 * K * dchi / sn;
          if (role == 0) en = K * dchi * dchi;
        }
        // d c / d r_role
        double dc[3], t1[3], t2[3], t3[3];
        if (role == 0) { cross3(G, gA, dc); }
        else if (role == 3) { cross3(G, gB, dc); }
        else {
          cross3(gA, F, t1); cross3(gB, H, t2);
          if (role == 1) { cross3(G, gA, t3); for (int k = 0; k < 3; k++) dc[k] = -t3[k] + t1[k] + t2[k]; }
          else { cross3(G, gB, t3); for (int k = 0; k < 3; k++) dc[k] = -(t1[k] + t2[k]) - t3[k]; }
        }
        for (int k = 0; k < 3; k++) fo[k] = -dEdc * dc[k];
        // virial relative to atom 3 of the term: r1-r3 = F+G, r2-r3 = G, r4-r3 = H
        if (role == 0) { const double FG[3] = {F[0] + G[0], F[1] + G[1], F[2] + G[2]}; vt(v, FG, fo); }
        else if (role == 1) vt(v, G, fo);
        else if (role == 3) vt(v, H, fo);
      } else {  // AT_SPECIAL: weighted real-space pair term (k-space minus (1-f_coul) q q / r)
        part = P_LJ;
        const int a0 = S.special_at[2 * m], a1 = S.special_at[2 * m + 1];
        const double wlj = S.special_cf[2 * m], wc = S.special_cf[2 * m + 1];
        double d[3] = {x[3 * a0] - x[3 * a1], x[3 * a0 + 1] - x[3 * a1 + 1], x[3 * a0 + 2] - x[3 * a1 + 2]};
        minimg(b, d[0], d[1], d[2]);
        const double rsq = dot3(d, d);
        if (rsq >= S.excl_cut2) atomicOr(&sc.overflow, 2);  // excluded pair escaped the build-time exclusion gate
        const double rinv = rsq64(rsq), r2inv = rinv * rinv;
        double flj = 0.0, fc = 0.0;
        if (rsq < S.cut_coul2 && S.g_ewald > 0.0) {
          // erf(x) - 2x/sqrt(pi) exp(-x^2) = x H(x^2): the polynomial of k_pair (md_pair.hip) instead of erf + exp
          const double g = S.g_ewald;
          const double t = fma(rsq, g * g * S.coul_uscale, -1.0);
          double p = S.coul_poly[S.coul_npoly - 1];
          for (int m = S.coul_npoly - 2; m >= 0; m--) p = fma(p, t, S.coul_poly[m]);
          const double grij = g * rsq * rinv;
          const double pref = MD_QQRD2E * S.q[a0] * S.q[a1] * rinv;
          fc = pref * fma(-grij, p, wc) * r2inv;
          if (PARTS && role == 0) en2 = pref * (wc - erf(grij));
        }
        if (rsq < S.cut_lj2 && wlj != 0.0) {
          const int nt = S.ntypes, tt = S.type[a0] * nt + S.type[a1];
          const double r6inv = r2inv * r2inv * r2inv;
          flj = wlj * r6inv * (S.lj[tt] * r6inv - S.lj[nt * nt + tt]) * r2inv;
          if (role == 0) en = wlj * r6inv * (S.lj[2 * nt * nt + tt] * r6inv - S.lj[3 * nt * nt + tt]);
        }
        const double sgn = (role == 0) ? 1.0 : -1.0;
        for (int k = 0; k < 3; k++) fo[k] = sgn * d[k] * (flj + fc);
        if (role == 0) {
          const double fl[3] = {d[0] * flj, d[1] * flj, d[2] * flj}, fq[3] = {d[0] * fc, d[1] * fc, d[2] * fc};
          vt(v, d, fl);
          vt(v2, d, fq);
        }
      }
      for (int k = 0; k < 3; k++) ftot[k] += fo[k];
      if (PARTS) {
        // parity hook: per-part sums straight to memory (slow path, never timed)
        for (int k = 0; k < 6; k++) {
          if (v[k] != 0.0) atomicAdd(&sc.vir[part * 6 + k], v[k]);
          if (v2[k] != 0.0) atomicAdd(&sc.vir[P_COUL * 6 + k], v2[k]);
        }
        if (en != 0.0) atomicAdd(&sc.eng[part], en);
        if (en2 != 0.0) atomicAdd(&sc.eng[P_COUL], en2);
      } else {
        for (int k = 0; k < 6; k++) vsum[k] += v[k] + v2[k];
      }
    }
    // total so far = pair force (slot-ordered, from k_pair) + bonded terms
    const size_t sl = (size_t)S.slot_of[i], np = (size_t)S.npad;
    S.f[3 * i] = S.fs[sl] + ftot[0]; S.f[3 * i + 1] = S.fs[np + sl] + ftot[1]; S.f[3 * i + 2] = S.fs[2 * np + sl] + ftot[2];
  }
  if (!PARTS) block_atomic_add<6>(vsum, sc.vir + P_BOND * 6, s_red);  // lumped: the pressure sums all parts anyway
}

void mdk_bonded_atom(hipStream_t st, const SimDev *d, int ns, int maxatoms, int parts) {
  const dim3 g((unsigned)((maxatoms + TPB - 1) / TPB), (unsigned)ns, 1);
  if (parts) hipLaunchKernelGGL(k_bonded_atom<true>, g, dim3(TPB), 0, st, d);
  else hipLaunchKernelGGL(k_bonded_atom<false>, g, dim3(TPB), 0, st, d);
}
