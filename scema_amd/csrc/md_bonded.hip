// md_bonded.hip -- bonded terms and special pairs, one workgroup per bonded tile.
//
// History: term-centric kernels with FP64 atomics into global memory were atomic-rate
// bound (dihedrals alone 14 % of the GPU time); the atom-centric, atomic-free kernel that replaced them
// evaluated every dihedral four times and every angle three times and sat at 23 % VALU utilisation (a C
// atom of polyethylene walks 48 terms serially); tiles that evaluated every term once and flushed halo forces with
// global FP64 atomics were bound by those atomics once the batch outgrew the caches (they execute at the memory
// side: 1.56 ms per 576-replica step against 0.11 ms per 72).  Now:
//   * a tile = BT_OWNERS atoms that are consecutive in a breadth-first ranking of the bond graph (its owners); it
//     evaluates every term that touches an owner (built once per topology, engine/engine_topo.cpp build_topo), so a term
//     that spans two tiles is evaluated twice -- a few per cent of the terms of a chain molecule -- and NO force
//     ever crosses a tile: the owners' forces leave with plain coalesced stores into fb (indexed by rank), nothing
//     is zeroed beforehand, no global atomic is issued;
//   * the tile's atoms (owners first, then the halo) get local indices; their positions are staged in LDS and the
//     forces of all terms accumulate in LDS (ds_add_f64);
//   * virial and energies of a term are counted by ONE tile (the owner of its lowest-ranked atom; the others see
//     BT_NOCOUNT on the term's first index); the lumped virial of a tile is stored per tile, without atomics;
//   * kinds are processed one after the other, so the lanes of a wave run the same formula.
//
// Virial of a term = sum_a (r_a - r_ref) (x) F_a with r_ref = atom 3 for torsions, the vertex for angles,
// atom 2 for bonds / pairs (differences by minimum image, positions are unwrapped).
//
// Styles (reference in.set.lammps:44-57, in.init.lammps:31): bond harmonic, angle harmonic,
// dihedral opls, improper harmonic, special_bonds weights on lj/cut/coul/long.
#include <hip/hip_runtime.h>

#include "md_device.h"
#include "md_kernels.h"

#define BT_TPB 256

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
__device__ __forceinline__ void lds_add3(double *f, int l, const double *v) {
  (void)__hip_atomic_fetch_add(&f[3 * l], v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  (void)__hip_atomic_fetch_add(&f[3 * l + 1], v[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  (void)__hip_atomic_fetch_add(&f[3 * l + 2], v[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

extern __shared__ double s_bt[];  // [3*maxloc] positions, [3*maxloc] forces, [maxloc] charges, [ncoef] coefficient tables, [2 or 4 x nt2] LJ table, int [maxloc] types

#ifndef BT_OCC
#define BT_OCC 4
#endif
#define BT_PRE 8   // descriptors a thread requests up front (chunks of its wave): covers 32 chunks = 2 048 terms per tile

// PARTS: also split virial / energy per part (parity hook); otherwise one lumped virial
template <bool PARTS>
__global__ __launch_bounds__(BT_TPB, BT_OCC) void k_bonded(const SimDev *__restrict__ sims, int maxloc, int maxcoef) {
  const SimDev &S = sims[blockIdx.y];
  if ((int)blockIdx.x >= S.bt_ntile) return;
  SimScalars &sc = *S.sc;
  __shared__ double s_red[8 * (BT_TPB / 64)];
  const int *desc = S.bt_desc + (size_t)blockIdx.x * BT_DESC;
  // (wave-uniform: said so, so that the loops over them run on the scalar unit)
  const int nloc = __builtin_amdgcn_readfirstlane(desc[1]), nown = __builtin_amdgcn_readfirstlane(desc[14]), nchunk = __builtin_amdgcn_readfirstlane(desc[3]);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // the wave's term descriptors first: nothing below depends on them until the positions are staged, so their latency
  // runs under the staging loads
  const unsigned long long *td = S.bt_terms + (size_t)desc[2] * 64 + lane;
  unsigned long long dq[BT_PRE];
#pragma unroll
  for (int k = 0; k < BT_PRE; k++) {
    const int c = wave + (BT_TPB / 64) * k;
    dq[k] = (c < nchunk) ? td[(size_t)c * 64] : 0ull;
  }
  const int *atoms = S.bt_atoms + desc[0];
  const int nt = S.ntypes, nt2 = nt * nt;
  double *s_x = s_bt, *s_f = s_bt + 3 * (size_t)maxloc, *s_q = s_bt + 6 * (size_t)maxloc, *s_cf = s_bt + 7 * (size_t)maxloc;
  double *s_lj = s_cf + maxcoef;
  int *s_t = (int *)(s_lj + (PARTS ? 4 : 2) * (size_t)MD_MAXTYPES * MD_MAXTYPES);
  for (int l = threadIdx.x; l < nloc; l += BT_TPB) {
    const int a = atoms[l];
    s_x[3 * l] = S.x[3 * a]; s_x[3 * l + 1] = S.x[3 * a + 1]; s_x[3 * l + 2] = S.x[3 * a + 2];
    s_f[3 * l] = 0.0; s_f[3 * l + 1] = 0.0; s_f[3 * l + 2] = 0.0;
    s_q[l] = S.q[a];
    s_t[l] = S.type[a] * nt;
  }
  for (int k = threadIdx.x; k < S.bt_ncoef; k += BT_TPB) s_cf[k] = S.bt_coef[k];
  // (lj1, lj2) of a type pair side by side (and lj3, lj4 behind them for the energies of the parity hook)
  for (int k = threadIdx.x; k < 2 * nt2; k += BT_TPB) {
    s_lj[k] = S.lj[(k & 1) * nt2 + (k >> 1)];
    if (PARTS) s_lj[2 * nt2 + k] = S.lj[(2 + (k & 1)) * nt2 + (k >> 1)];
  }
  BoxD b;
  box_derive(sc.box, b);
  box_uniform(b);
  const double g = S.g_ewald, g2u = g * g * S.coul_uscale;
  const int np = S.coul_npoly;
  const double *cf_bond = s_cf + S.bt_cf_off[0], *cf_angle = s_cf + S.bt_cf_off[1], *cf_dih = s_cf + S.bt_cf_off[2], *cf_imp = s_cf + S.bt_cf_off[3];
  __syncthreads();
  double vsum[6] = {0, 0, 0, 0, 0, 0};
  // per-part sums straight to memory (parity hook: slow path, never timed)
  auto emit = [&](bool counted, int part, const double *v, double en) {
    if (!counted) return;
    if (PARTS) {
      for (int k = 0; k < 6; k++)
        if (v[k] != 0.0) atomicAdd(&sc.vir[part * 6 + k], v[k]);
      if (en != 0.0) atomicAdd(&sc.eng[part], en);
    } else {
      for (int k = 0; k < 6; k++) vsum[k] += v[k];
    }
  };
  auto term = [&](unsigned long long d) {
    // the kind is the same for the whole chunk (kinds are padded to whole chunks): a scalar branch
    const int kind = __builtin_amdgcn_readfirstlane((int)(d >> BT_D_KIND_SHIFT) & 7);
    const bool valid = (d & BT_D_VALID) != 0ull;
    if (!__any(valid)) return;
    const bool counted = valid && !(d & BT_D_NOCOUNT);
    const int l1 = (int)(d & BT_D_LMASK), l2 = (int)((d >> 10) & BT_D_LMASK), l3 = (int)((d >> 20) & BT_D_LMASK), l4 = (int)((d >> 30) & BT_D_LMASK);
    const int ty = (int)((d >> BT_D_TYPE_SHIFT) & BT_D_TMASK);
    if (kind == BT_BOND || kind == BT_BOND_SHAKEN) {
      // ---- bonds (the SHAKE'd ones only while fix shake is off) ----
      if (kind == BT_BOND_SHAKEN && S.use_shake) return;
      if (!valid) return;
      const double K = cf_bond[2 * ty], r0 = cf_bond[2 * ty + 1];
      double dd[3] = {s_x[3 * l1] - s_x[3 * l2], s_x[3 * l1 + 1] - s_x[3 * l2 + 1], s_x[3 * l1 + 2] - s_x[3 * l2 + 2]};
      minimg(b, dd[0], dd[1], dd[2]);
      const double rsq = dot3(dd, dd);
      const double rinv = (rsq > 0.0) ? rsq64(rsq) : 0.0;
      const double r = rsq * rinv;
      const double dr = r - r0, rk = K * dr;
      const double fb = -2.0 * rk * rinv;
      const double f1[3] = {dd[0] * fb, dd[1] * fb, dd[2] * fb}, f2[3] = {-f1[0], -f1[1], -f1[2]};
      lds_add3(s_f, l1, f1);
      lds_add3(s_f, l2, f2);
      double v[6] = {0, 0, 0, 0, 0, 0};
      vt(v, dd, f1);
      emit(counted, P_BOND, v, rk * dr);
    } else if (kind == BT_ANGLE) {
      if (!valid) return;
      const double K = cf_angle[2 * ty], th0 = cf_angle[2 * ty + 1];
      double d1[3], d2[3];
      for (int k = 0; k < 3; k++) { d1[k] = s_x[3 * l1 + k] - s_x[3 * l2 + k]; d2[k] = s_x[3 * l3 + k] - s_x[3 * l2 + k]; }
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
      double f1[3], f3[3], f2[3];
      for (int k = 0; k < 3; k++) { f1[k] = a11 * d1[k] + a12 * d2[k]; f3[k] = a22 * d2[k] + a12 * d1[k]; f2[k] = -(f1[k] + f3[k]); }
      lds_add3(s_f, l1, f1);
      lds_add3(s_f, l2, f2);
      lds_add3(s_f, l3, f3);
      double v[6] = {0, 0, 0, 0, 0, 0};
      vt(v, d1, f1); vt(v, d2, f3);
      emit(counted, P_ANGLE, v, tk * dth);
    } else if (kind == BT_DIHEDRAL || kind == BT_IMPROPER) {
      // ---- dihedrals (opls) and impropers (harmonic): same geometry, different dE/dcos ----
      if (!valid) return;
      // F=r1-r2, G=r2-r3, H=r4-r3, A=FxG, B=HxG, c=A.B/(|A||B|)
      double F[3], G[3], H[3];
      for (int k = 0; k < 3; k++) {
        F[k] = s_x[3 * l1 + k] - s_x[3 * l2 + k];
        G[k] = s_x[3 * l2 + k] - s_x[3 * l3 + k];
        H[k] = s_x[3 * l4 + k] - s_x[3 * l3 + k];
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
      double dEdc, en = 0.0;
      if (kind == BT_DIHEDRAL) {
        const double *K = cf_dih + 4 * ty;
        const double c2 = c * c;
        dEdc = 0.5 * (K[0] - K[1] * 4.0 * c + K[2] * (12.0 * c2 - 3.0) - K[3] * (32.0 * c2 * c - 16.0 * c));
        if (PARTS) {
          const double cos2 = 2.0 * c2 - 1.0, cos3 = (4.0 * c2 - 3.0) * c, cos4 = 8.0 * c2 * c2 - 8.0 * c2 + 1.0;
          en = 0.5 * (K[0] * (1.0 + c) + K[1] * (1.0 - cos2) + K[2] * (1.0 + cos3) + K[3] * (1.0 - cos4));
        }
      } else {
        const double K = cf_imp[2 * ty], chi0 = cf_imp[2 * ty + 1];
        double sn = sqrt(1.0 - c * c);
        if (sn < 0.001) sn = 0.001;
        const double dchi = acos(c) - chi0;
        dEdc = -2.0 * K * dchi / sn;
        en = K * dchi * dchi;
      }
      // d c / d r_a : r1: G x gA ; r4: G x gB ; r2: -G x gA + gA x F + gB x H ; r3: -(gA x F + gB x H) - G x gB
      double tA[3], tB[3], t1[3], t2[3];
      cross3(G, gA, tA); cross3(G, gB, tB); cross3(gA, F, t1); cross3(gB, H, t2);
      double f1[3], f2[3], f3[3], f4[3];
      for (int k = 0; k < 3; k++) {
        const double u = t1[k] + t2[k];
        f1[k] = -dEdc * tA[k];
        f4[k] = -dEdc * tB[k];
        f2[k] = -dEdc * (u - tA[k]);
        f3[k] = dEdc * (u + tB[k]);
      }
      lds_add3(s_f, l1, f1);
      lds_add3(s_f, l2, f2);
      lds_add3(s_f, l3, f3);
      lds_add3(s_f, l4, f4);
      // virial relative to atom 3 of the term: r1-r3 = F+G, r2-r3 = G, r4-r3 = H
      double v[6] = {0, 0, 0, 0, 0, 0};
      const double FG[3] = {F[0] + G[0], F[1] + G[1], F[2] + G[2]};
      vt(v, FG, f1); vt(v, G, f2); vt(v, H, f4);
      emit(counted, kind == BT_DIHEDRAL ? P_DIHEDRAL : P_IMPROPER, v, en);
    } else {
      // ---- special pairs: weighted real-space pair term (k-space minus (1-f_coul) q q / r) ----
      if (!valid) return;
      const int lvl = (int)((d >> BT_D_LVL_SHIFT) & 3);
      const double wl = S.sp_w[lvl - 1], wc = S.sp_w[2 + lvl];
      const int tt = s_t[l1] + s_t[l2] / nt;
      const double qq = MD_QQRD2E * s_q[l1] * s_q[l2];
      double dd[3] = {s_x[3 * l1] - s_x[3 * l2], s_x[3 * l1 + 1] - s_x[3 * l2 + 1], s_x[3 * l1 + 2] - s_x[3 * l2 + 2]};
      minimg(b, dd[0], dd[1], dd[2]);
      const double rsq = dot3(dd, dd);
      if (rsq >= S.excl_cut2 && counted) atomicOr(&sc.overflow, 2);  // excluded pair escaped the build-time exclusion gate
      const double rinv = rsq64(rsq), r2inv = rinv * rinv;
      double flj = 0.0, fc = 0.0, en = 0.0, en2 = 0.0;
      if (rsq < S.cut_coul2 && g > 0.0) {
        // erf(x) - 2x/sqrt(pi) exp(-x^2) = x H(x^2): the polynomial of k_pair (md_pair.hip) instead of erf + exp
        const double tq = fma(rsq, g2u, -1.0);
        double p = S.coul_poly[np - 1];
        for (int k = np - 2; k >= 0; k--) p = fma(p, tq, S.coul_poly[k]);
        const double grij = g * rsq * rinv;
        const double pref = qq * rinv;
        fc = pref * fma(-grij, p, wc) * r2inv;
        if (PARTS) en2 = pref * (wc - erf(grij));
      }
      if (rsq < S.cut_lj2 && wl != 0.0) {
        const double r6inv = r2inv * r2inv * r2inv;
        flj = wl * r6inv * (s_lj[2 * tt] * r6inv - s_lj[2 * tt + 1]) * r2inv;
        if (PARTS) en = wl * r6inv * (s_lj[2 * nt2 + 2 * tt] * r6inv - s_lj[2 * nt2 + 2 * tt + 1]);
      }
      const double fp = flj + fc;
      const double f1[3] = {dd[0] * fp, dd[1] * fp, dd[2] * fp}, f2[3] = {-f1[0], -f1[1], -f1[2]};
      lds_add3(s_f, l1, f1);
      lds_add3(s_f, l2, f2);
      if (PARTS) {
        double v[6] = {0, 0, 0, 0, 0, 0}, v2[6] = {0, 0, 0, 0, 0, 0};
        const double fl[3] = {dd[0] * flj, dd[1] * flj, dd[2] * flj}, fq[3] = {dd[0] * fc, dd[1] * fc, dd[2] * fc};
        vt(v, dd, fl);
        vt(v2, dd, fq);
        emit(counted, P_LJ, v, en);
        emit(counted, P_COUL, v2, en2);
      } else if (counted) {
        vt(vsum, dd, f1);
      }
    }
  };
  // one copy of the term code (it is large): the descriptor of pass k is picked out of the preloaded registers
#pragma unroll 1
  for (int k = 0; k < BT_PRE; k++) {
    if (wave + (BT_TPB / 64) * k >= nchunk) break;
    unsigned long long d = dq[0];
#pragma unroll
    for (int q = 1; q < BT_PRE; q++) d = (k == q) ? dq[q] : d;
    term(d);
  }
  for (int c = wave + (BT_TPB / 64) * BT_PRE; c < nchunk; c += BT_TPB / 64) term(td[(size_t)c * 64]);   // very large tiles
  __syncthreads();
  // flush: the owners' forces, plain coalesced stores (consecutive ranks = consecutive local indices); halo forces
  // belong to the tiles that own those atoms, which evaluate the same terms themselves
  {
    double *fb = S.fb + 3 * (size_t)blockIdx.x * BT_OWNERS;
    for (int l = threadIdx.x; l < 3 * nown; l += BT_TPB) fb[l] = s_f[l];
  }
  if (!PARTS) {
    // lumped virial of the tile (the pressure sums all parts anyway): one partial per tile, no atomics
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const double sk = wave_sum(vsum[k]);
      if (lane == 0) s_red[k * (BT_TPB / 64) + wave] = sk;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
      double sk = 0.0;
      for (int w = 0; w < BT_TPB / 64; w++) sk += s_red[threadIdx.x * (BT_TPB / 64) + w];
      S.virb[(size_t)blockIdx.x * 6 + threadIdx.x] = sk;
    }
  }
}

void mdk_bonded(hipStream_t st, const SimDev *d, int ns, int maxtiles, int maxloc, int maxcoef, int parts) {
  const dim3 g((unsigned)maxtiles, (unsigned)ns, 1);
  const size_t lds = ((size_t)7 * maxloc + maxcoef + (parts ? 4 : 2) * MD_MAXTYPES * MD_MAXTYPES) * sizeof(double) + (size_t)maxloc * sizeof(int);
  static size_t optin_tab[16] = {0};
  size_t &optin = lds_optin_slot(optin_tab);
  if (lds > 48 * 1024 && lds > optin) {
    (void)hipFuncSetAttribute((const void *)k_bonded<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void *)k_bonded<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    optin = lds;
  }
  if (parts) hipLaunchKernelGGL(k_bonded<true>, g, dim3(BT_TPB), lds, st, d, maxloc, maxcoef);
  else hipLaunchKernelGGL(k_bonded<false>, g, dim3(BT_TPB), lds, st, d, maxloc, maxcoef);
}
