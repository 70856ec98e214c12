// md_pair.hip -- neighbour-list build and the lj/cut/coul/long pair kernel (the roofline kernel).
//
// Work decomposition (round-1 measurements in DESIGN.md): ONE WAVE PER ATOM i, the 64 lanes span
// i's neighbour row.  A thread-per-atom kernel with per-lane j gathers was L1/TA bound on gfx950
// (183 L1 accesses per 64 pairs, nothing coalesces).  With lanes over neighbours:
//   * neigh[i*maxneigh + k] : the row of atom i is contiguous -> a wave reads 256 B per instruction;
//   * inside a row the entries keep slot (= cell) order, so the 32-byte j records a wave gathers
//     fall into a few runs of consecutive slots -> a few cache lines per instruction, not 64;
//   * i's data is wave-uniform (scalar registers), the force on i is one wave reduction per atom;
//   * each row is stored in three segments by build-time distance: A (< cut_coul + m) needs
//     coulomb + LJ, B (< cut_lj + m) LJ only, C the skin.  Every lane still tests r^2 against the
//     cutoffs (results never depend on the segments); the segments only make the three regimes
//     wave-uniform so whole waves skip the expensive branches.
//   entry = image code (5 bits) | j type (5 bits) | j slot (22 bits)
//
// Reference semantics: pair_style lj/cut/coul/long 12.0 9.0 (in.set.lammps:40), neighbor 2.0 bin
// (in.set.lammps:27); special_bonds 0 0 1 pairs are excluded here and handled in k_term<T_SPECIAL>.
#include <hip/hip_runtime.h>

#include "md_device.h"
#include "md_kernels.h"

#define WPB 4               // waves per block
#define AW 4                // atoms per wave (k_pair sweeps them row-interleaved)
#define APB (WPB * AW)      // atoms per block

#define GLOBAL_AS __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ const GLOBAL_AS T *as_global(const T *p) {
  return (const GLOBAL_AS T *)p;
}
template <class T>
__device__ __forceinline__ GLOBAL_AS T *as_global_w(T *p) {
  return (GLOBAL_AS T *)p;
}

// XCD-aware block -> (simulation, tile) map.  Workgroups are dealt round-robin over the 8 XCDs
// (block L lands on XCD L % 8), each with its own 4 MiB L2.  A simulation's j gathers touch its
// whole 332 KB position table, so all tiles of one simulation are placed on ONE XCD: simulation
// s uses the blocks with L % 8 == s % 8.  Placement only affects speed, never results.
__device__ __forceinline__ bool xcd_map(int ntiles, int nsims, int &sim, int &tile) {
  const int L = blockIdx.x;
  const int x = L & 7, w = L >> 3;
  sim = (w / ntiles) * 8 + x;
  tile = w % ntiles;
  return sim < nsims;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int popc_below(unsigned long long m) {
  // number of set bits of m below this lane
  return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
}

// ------------------------------------------------------------------------------------------
// k_neigh_build : wave per atom; lanes scan runs of candidate slots (coalesced), ballots compact
// the accepted ones into the three row segments, in slot order
// ------------------------------------------------------------------------------------------
template <int PASS>
__device__ __forceinline__ void scan_atom(const SimDev &S, const BoxD &b, int i, double xi0, double xi1, double xi2, int ai,
                                          int ci, int &nA, int &nB, int &nC, int offB, int offC) {
  const int lane = lane_id();
  const int c0 = ci % S.nc[0], c1 = (ci / S.nc[0]) % S.nc[1], c2 = ci / (S.nc[0] * S.nc[1]);
  const int exb = S.ex_start[ai], exe = S.ex_start[ai + 1];
  const GLOBAL_AS double *xq = as_global((const double *)S.xq);
  const GLOBAL_AS int *stype = as_global(S.stype);
  GLOBAL_AS int *row = as_global_w(S.neigh) + (size_t)i * S.maxneigh;
  const double ra2 = S.seg_a2, rb2 = S.seg_b2;
  for (int o2 = -S.mst[2]; o2 <= S.mst[2]; o2++) {
    int a2 = c2 + o2, s2 = 0;
    while (a2 < 0) { a2 += S.nc[2]; s2 -= 1; }
    while (a2 >= S.nc[2]) { a2 -= S.nc[2]; s2 += 1; }
    if (s2 < -1 || s2 > 1) continue;
    for (int o1 = -S.mst[1]; o1 <= S.mst[1]; o1++) {
      int a1 = c1 + o1, s1 = 0;
      while (a1 < 0) { a1 += S.nc[1]; s1 -= 1; }
      while (a1 >= S.nc[1]) { a1 -= S.nc[1]; s1 += 1; }
      if (s1 < -1 || s1 > 1) continue;
      // the x range of cells [c0-m, c0+m] is one or more contiguous slot runs, one per image
      int o0 = -S.mst[0];
      while (o0 <= S.mst[0]) {
        int a0 = c0 + o0, s0 = 0;
        while (a0 < 0) { a0 += S.nc[0]; s0 -= 1; }
        while (a0 >= S.nc[0]) { a0 -= S.nc[0]; s0 += 1; }
        // extend the run while the cells stay consecutive under the same image
        int len = 1;
        while (o0 + len <= S.mst[0] && a0 + len < S.nc[0]) len++;
        o0 += len;
        if (s0 < -1 || s0 > 1) continue;
        const double sx = b.h[0] * s0 + b.h[5] * s1 + b.h[4] * s2;
        const double sy = b.h[1] * s1 + b.h[3] * s2;
        const double sz = b.h[2] * s2;
        const int code = ((s2 + 1) * 9 + (s1 + 1) * 3 + (s0 + 1)) << MD_CODE_SHIFT;
        const int cj = (a2 * S.nc[1] + a1) * S.nc[0] + a0;
        const int jb = S.cell_start[cj], je = S.cell_start[cj + len];
        for (int base = jb; base < je; base += 64) {
          const int j = base + lane;
          bool acc = false;
          double r2 = 0.0;
          if (j < je) {
            const double dx = xi0 - xq[4 * (size_t)j] - sx, dy = xi1 - xq[4 * (size_t)j + 1] - sy, dz = xi2 - xq[4 * (size_t)j + 2] - sz;
            r2 = dx * dx + dy * dy + dz * dz;
            acc = r2 < S.rlist2 && !(j == i && s0 == 0 && s1 == 0 && s2 == 0);
            if (acc && r2 < S.excl_cut2) {
              const int aj = S.perm[j];
              for (int e = exb; e < exe; e++) acc = acc && (S.ex_list[e] != aj);
            }
          }
          const bool isA = acc && r2 < ra2, isB = acc && !isA && r2 < rb2, isC = acc && !isA && !isB;
          const unsigned long long mA = __ballot(isA), mB = __ballot(isB), mC = __ballot(isC);
          if (PASS == 1 && acc) {
            const int entry = code | (stype[j] << MD_TYPE_SHIFT) | j;
            int pos;
            if (isA) pos = nA + popc_below(mA);
            else if (isB) pos = offB + nB + popc_below(mB);
            else pos = offC + nC + popc_below(mC);
            if (pos < S.maxneigh) row[pos] = entry;
          }
          nA += __popcll(mA); nB += __popcll(mB); nC += __popcll(mC);
        }
      }
    }
  }
}

__global__ __launch_bounds__(WPB * 64) void k_neigh_build(const SimDev *__restrict__ sims, int ntiles, int nsims) {
  int sim, tile;
  if (!xcd_map(ntiles, nsims, sim, tile)) return;
  const SimDev &S = sims[sim];
  SimScalars &sc = *S.sc;
  if (!sc.rebuild) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  BoxD b;
  box_derive(sc.box, b);
  unsigned long long entries = 0;
  int nmax = 0;
  for (int a = 0; a < AW; a++) {
    const int i = tile * APB + wave * AW + a;  // wave-uniform
    if (i >= S.natoms) break;
    const double xi0 = S.xq[i].x, xi1 = S.xq[i].y, xi2 = S.xq[i].z;
    const int ai = S.perm[i];
    const int ci = S.cell_of[ai];
    int nA = 0, nB = 0, nC = 0;
    scan_atom<0>(S, b, i, xi0, xi1, xi2, ai, ci, nA, nB, nC, 0, 0);
    const int n = nA + nB + nC;
    int pA = 0, pB = 0, pC = 0;
    scan_atom<1>(S, b, i, xi0, xi1, xi2, ai, ci, pA, pB, pC, nA, nA + nB);
    if (lane_id() == 0) S.numneigh[i] = (n < S.maxneigh) ? n : S.maxneigh;
    entries += n;
    nmax = max(nmax, n);
  }
  if (lane_id() == 0) {
    if (nmax > S.maxneigh) atomicOr(&sc.overflow, 1);
    atomicMax(&sc.maxneigh_seen, nmax);
    atomicAdd(&sc.nentries, entries);
  }
}

// ------------------------------------------------------------------------------------------
// k_pair
// ------------------------------------------------------------------------------------------
// NP = number of polynomial coefficients kept in scalar registers (>= fitted degree+1, 0-padded)
template <bool VIR, bool ENG, int NP>
__global__ __launch_bounds__(WPB * 64) void k_pair(const SimDev *__restrict__ sims, int ntiles, int nsims) {
  int sim, tile;
  if (!xcd_map(ntiles, nsims, sim, tile)) return;
  const SimDev &S = sims[sim];
  SimScalars &sc = *S.sc;
  __shared__ double s_shift[27 * 4];
  __shared__ double s_lj[4 * MD_MAXTYPES * MD_MAXTYPES];
  __shared__ double s_red[8 * WPB];
  if (threadIdx.x < 27) {
    BoxD b;
    box_derive(sc.box, b);
    const int s0 = threadIdx.x % 3 - 1, s1 = (threadIdx.x / 3) % 3 - 1, s2 = threadIdx.x / 9 - 1;
    s_shift[4 * threadIdx.x + 0] = b.h[0] * s0 + b.h[5] * s1 + b.h[4] * s2;
    s_shift[4 * threadIdx.x + 1] = b.h[1] * s1 + b.h[3] * s2;
    s_shift[4 * threadIdx.x + 2] = b.h[2] * s2;
    s_shift[4 * threadIdx.x + 3] = 0.0;
  }
  const int nt = S.ntypes, nt2 = nt * nt;
  for (int k = threadIdx.x; k < 4 * nt2; k += WPB * 64) s_lj[k] = S.lj[k];
  double cp[NP];
#pragma unroll
  for (int m = 0; m < NP; m++) cp[m] = S.coul_poly[m];
  __syncthreads();
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const GLOBAL_AS double *xq = as_global((const double *)S.xq);
  const double g = S.g_ewald, g2u = g * g * S.coul_uscale;
  const double cutc2 = S.cut_coul2, cutl2 = S.cut_lj2;
  const double cutmax2 = fmax(cutc2, cutl2);
  const int maxneigh = S.maxneigh;
  double vl[6] = {0, 0, 0, 0, 0, 0}, vc[6] = {0, 0, 0, 0, 0, 0};
  double elj = 0, ecoul = 0;
  // ---- row-interleaved sweep over the PA atoms of this wave ----
  // Consecutive slots are spatial neighbours, so row r of atom i and row r of atom i+1 hold almost
  // the same j slots.  One atom's row sweep touches 1490 x 32 B = 47 KB, more than the 32 KB L1, so
  // sweeping atom after atom re-fetched everything from L2 (31 % L1 misses, TCP stalled 62 % of the
  // time on pending misses).  Interleaving the atoms row by row lets the lines fetched for one
  // atom's row serve the other PA-1 atoms, and puts PA independent rows in flight per wave.
  constexpr int PA = AW;
  const int i0 = tile * APB + wave * AW;  // wave-uniform
  double xi0[PA], xi1[PA], xi2[PA], qi[PA], fx[PA], fy[PA], fz[PA];
  int ti[PA], nn[PA];
  const GLOBAL_AS int *row[PA];
  int nmax = 0;
#pragma unroll
  for (int a = 0; a < PA; a++) {
    const int i = (i0 + a < S.natoms) ? i0 + a : S.natoms - 1;
    xi0[a] = S.xq[i].x; xi1[a] = S.xq[i].y; xi2[a] = S.xq[i].z;
    qi[a] = MD_QQRD2E * S.xq[i].w;
    ti[a] = S.stype[i] * nt;
    nn[a] = (i0 + a < S.natoms) ? S.numneigh[i] : 0;
    row[a] = as_global(S.neigh) + (size_t)i * maxneigh;
    fx[a] = fy[a] = fz[a] = 0.0;
    nmax = max(nmax, nn[a]);
  }
  for (int k0 = 0; k0 < nmax; k0 += 64) {
    int e[PA];
    double xj[PA][4];
#pragma unroll
    for (int a = 0; a < PA; a++) e[a] = (k0 + lane < nn[a]) ? row[a][k0 + lane] : -1;
#pragma unroll
    for (int a = 0; a < PA; a++) {
      const size_t j4 = 4 * (size_t)((e[a] == -1) ? 0 : (e[a] & MD_JMASK));
      xj[a][0] = xq[j4]; xj[a][1] = xq[j4 + 1]; xj[a][2] = xq[j4 + 2]; xj[a][3] = xq[j4 + 3];
    }
#pragma unroll
    for (int a = 0; a < PA; a++) {
      const int ee = e[a];
      if (ee != -1) {
        const int c = 4 * (((unsigned)ee) >> MD_CODE_SHIFT);
        const double dx = xi0[a] - xj[a][0] - s_shift[c], dy = xi1[a] - xj[a][1] - s_shift[c + 1], dz = xi2[a] - xj[a][2] - s_shift[c + 2];
        const double rsq = dx * dx + dy * dy + dz * dz;
        if (rsq < cutmax2) {
          const double rinv = rsqrt(rsq);
          const double r2inv = rinv * rinv;
          double flj = 0.0, fc = 0.0;
          if (rsq < cutc2) {
            // erfc(x) + 2x/sqrt(pi) exp(-x^2) = 1 - x H(u): Horner in t = u*uscale - 1
            const double x = g * rsq * rinv;
            const double t = fma(rsq, g2u, -1.0);
            double p = cp[NP - 1];
#pragma unroll
            for (int m = NP - 2; m >= 0; m--) p = fma(p, t, cp[m]);
            const double pref = qi[a] * xj[a][3] * rinv;
            fc = pref * fma(-x, p, 1.0) * r2inv;
            if (ENG) ecoul += pref * erfc(x);
          }
          if (rsq < cutl2) {
            const int tt = ti[a] + ((ee >> MD_TYPE_SHIFT) & MD_TYPE_MASK);
            const double r6inv = r2inv * r2inv * r2inv;
            flj = r6inv * (s_lj[tt] * r6inv - s_lj[nt2 + tt]) * r2inv;
            if (ENG) elj += r6inv * (s_lj[2 * nt2 + tt] * r6inv - s_lj[3 * nt2 + tt]);
          }
          const double fp = flj + fc;
          fx[a] = fma(dx, fp, fx[a]); fy[a] = fma(dy, fp, fy[a]); fz[a] = fma(dz, fp, fz[a]);
          if (VIR) {
            const double xl = dx * flj, yl = dy * flj, zl = dz * flj;
            vl[0] = fma(dx, xl, vl[0]); vl[1] = fma(dy, yl, vl[1]); vl[2] = fma(dz, zl, vl[2]);
            vl[3] = fma(dx, yl, vl[3]); vl[4] = fma(dx, zl, vl[4]); vl[5] = fma(dy, zl, vl[5]);
            const double xc = dx * fc, yc = dy * fc, zc = dz * fc;
            vc[0] = fma(dx, xc, vc[0]); vc[1] = fma(dy, yc, vc[1]); vc[2] = fma(dz, zc, vc[2]);
            vc[3] = fma(dx, yc, vc[3]); vc[4] = fma(dx, zc, vc[4]); vc[5] = fma(dy, zc, vc[5]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int a = 0; a < PA; a++) {
    const double sx = wave_sum(fx[a]), sy = wave_sum(fy[a]), sz = wave_sum(fz[a]);
    if (lane == 0 && i0 + a < S.natoms) {
      const int at = S.perm[i0 + a];
      S.f[3 * at] = sx; S.f[3 * at + 1] = sy; S.f[3 * at + 2] = sz;
    }
  }
  if (VIR) {
    // full list: every pair is visited from both ends
    for (int k = 0; k < 6; k++) { vl[k] *= 0.5; vc[k] *= 0.5; }
    block_atomic_add<6>(vl, sc.vir + P_LJ * 6, s_red);
    block_atomic_add<6>(vc, sc.vir + P_COUL * 6, s_red);
  }
  if (ENG) {
    double e1[1];
    e1[0] = 0.5 * elj;
    block_atomic_add<1>(e1, sc.eng + P_LJ, s_red);
    e1[0] = 0.5 * ecoul;
    block_atomic_add<1>(e1, sc.eng + P_COUL, s_red);
  }
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline dim3 grid_xcd(int ntiles, int ns) { return dim3((unsigned)(cdiv(ns, 8) * 8 * ntiles), 1, 1); }

void mdk_neigh_build(hipStream_t st, const SimDev *d, int ns, int maxatoms) {
  const int ntiles = cdiv(maxatoms, APB);
  hipLaunchKernelGGL(k_neigh_build, grid_xcd(ntiles, ns), dim3(WPB * 64), 0, st, d, ntiles, ns);
}

template <int NP>
static void launch_pair(hipStream_t st, const SimDev *d, int ns, int ntiles, int vir, int eng) {
  const dim3 g = grid_xcd(ntiles, ns);
  if (eng) hipLaunchKernelGGL((k_pair<true, true, NP>), g, dim3(WPB * 64), 0, st, d, ntiles, ns);
  else if (vir) hipLaunchKernelGGL((k_pair<true, false, NP>), g, dim3(WPB * 64), 0, st, d, ntiles, ns);
  else hipLaunchKernelGGL((k_pair<false, false, NP>), g, dim3(WPB * 64), 0, st, d, ntiles, ns);
}

void mdk_pair(hipStream_t st, const SimDev *d, int ns, int maxatoms, int vir, int eng, int npoly) {
  const int ntiles = cdiv(maxatoms, APB);
  if (npoly <= 16) launch_pair<16>(st, d, ns, ntiles, vir, eng);
  else if (npoly <= 20) launch_pair<20>(st, d, ns, ntiles, vir, eng);
  else if (npoly <= 24) launch_pair<24>(st, d, ns, ntiles, vir, eng);
  else if (npoly <= 32) launch_pair<32>(st, d, ns, ntiles, vir, eng);
  else launch_pair<MD_MAXPOLY>(st, d, ns, ntiles, vir, eng);
}
