// md_pair.hip -- neighbour-list build and the lj/cut/coul/long pair kernel (the roofline kernel).
//
// Work decomposition (measurements of every step are in DESIGN.md §5):
//   * i-CLUSTERS: 4 consecutive slots of one cell (cells are padded to multiples of 4 slots, pad
//     slots hold far-away dummy records).  The four atoms are < a cell diagonal apart, so their
//     neighbour sets overlap ~87 %: the cluster owns ONE row = the union of its atoms' neighbours,
//     each entry carrying a 4-bit mask of which i atoms list that j.
//   * ONE WAVE PER CLUSTER, the 64 lanes span the row: neigh[cl*maxrow + k] is contiguous (256 B per
//     wave instruction); a lane gathers its j record once and evaluates it against up to 4 i atoms
//     held in scalar registers -> 4x fewer row bytes from HBM and 4x fewer record gathers through
//     L1 than one row per atom, and four independent pair evaluations per lane for ILP.
//   * rows keep slot (= cell) order inside three segments by build-time distance (A: may need
//     coulomb, B: LJ only, C: skin), so gathers fall into runs of consecutive slots and the three
//     regimes are wave-uniform branches.  Every lane still tests r^2 against the cutoffs: results
//     never depend on the segments or on the cluster grouping.
//   * forces: per-lane partial sums for the 4 atoms, one wave reduction per cluster, written once
//     (no atomics); virial: wave -> block -> one atomic per block and component.
//   entry = image code [31:27] | i-mask [26:23] | j slot [22:0]
//
// Reference semantics: pair_style lj/cut/coul/long 12.0 9.0 (in.set.lammps:40), neighbor 2.0 bin
// (in.set.lammps:27); special_bonds 0 0 1 pairs are excluded here and handled in k_bonded_atom.
#include <hip/hip_runtime.h>

#include "md_device.h"
#include "md_kernels.h"

#define WPB 4    // waves per block
#define CPW 2    // clusters per wave (sequential)
#define CPB (WPB * CPW)
#define NI MD_CLUSTER

// slot records are stored as two arrays of 16-byte halves, (x,y)[npad] then (z,q)[npad]: a wave's gather
// instruction then touches 16 B per lane at stride 16 (half the cache lines of 32-byte records read as two
// 16-byte halves at stride 32)
#define XQ_X(S, s) (((const double *)(S).xq)[2 * (size_t)(s)])
#define XQ_Y(S, s) (((const double *)(S).xq)[2 * (size_t)(s) + 1])
#define XQ_Z(S, s) (((const double *)(S).xq)[2 * (size_t)(S).npad + 2 * (size_t)(s)])
#define XQ_Q(S, s) (((const double *)(S).xq)[2 * (size_t)(S).npad + 2 * (size_t)(s) + 1])

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

// ------------------------------------------------------------------------------------------
// k_neigh_build : wave per cluster; lanes scan runs of candidate slots (coalesced), test them
// against the cluster's atoms, ballots compact the accepted ones into the three row segments
// ------------------------------------------------------------------------------------------
struct ClusterI {
  double x[NI], y[NI], z[NI];
  int atom[NI];   // real atom index or -1 (pad)
};

// single pass: accepted entries are compacted with ballots + prefix popcounts.  Segment A grows
// forward from the start of the cluster's row, segment C backward from its end, segment B is
// collected in a per-wave LDS list and appended behind A at the end: row = [A | B | ... | C reversed].
// Only B needs LDS (7.6 KB per wave at PE-10k), which keeps 5 blocks per CU resident.
extern __shared__ int s_lists[];  // [WPB][capB]

__global__ __launch_bounds__(WPB * 64) void k_neigh_build(const SimDev *__restrict__ sims, int ntiles, int nsims, int capB) {
  int sim, tile;
  if (!xcd_map(ntiles, nsims, sim, tile)) return;
  const SimDev &S = sims[sim];
  SimScalars &sc = *S.sc;
  if (!sc.rebuild) return;
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int *lb = s_lists + wave * capB;
  const int maxrow = S.maxneigh;
  BoxD b;
  box_derive(sc.box, b);
  const GLOBAL_AS double *xq = as_global((const double *)S.xq);               // (x,y) halves
  const GLOBAL_AS double *zq = xq + 2 * (size_t)S.npad;                          // (z,q) halves
  const double ra2 = S.seg_a2, rb2 = S.seg_b2;
  unsigned long long npairs = 0, nrowent = 0;
  int nmax = 0, over = 0;
  for (int c = 0; c < CPW; c++) {
    const int cl = tile * CPB + wave * CPW + c;  // wave-uniform
    const int s0slot = cl * NI;
    if (s0slot >= S.npad) break;
    ClusterI ci;
#pragma unroll
    for (int a = 0; a < NI; a++) {
      ci.atom[a] = S.perm[s0slot + a];
      ci.x[a] = XQ_X(S, s0slot + a); ci.y[a] = XQ_Y(S, s0slot + a); ci.z[a] = XQ_Z(S, s0slot + a);
    }
    if (ci.atom[0] < 0) {  // empty cluster (pad only)
      if (lane == 0) { S.numneigh[2 * cl] = 0; S.numneigh[2 * cl + 1] = 0; }
      continue;
    }
    GLOBAL_AS int *row = as_global_w(S.neigh) + (size_t)cl * maxrow;
    const int cell = S.cell_of[ci.atom[0]];
    const int c0 = cell % S.nc[0], c1 = (cell / S.nc[0]) % S.nc[1], c2 = cell / (S.nc[0] * S.nc[1]);
    // bounding sphere of the cluster's real atoms: one test rejects a candidate for all four atoms
    double bx = 0.0, by = 0.0, bz = 0.0, brad2 = 0.0;
    {
      int nreal = 0;
#pragma unroll
      for (int a = 0; a < NI; a++)
        if (ci.atom[a] >= 0) { bx += ci.x[a]; by += ci.y[a]; bz += ci.z[a]; nreal++; }
      bx /= nreal; by /= nreal; bz /= nreal;
#pragma unroll
      for (int a = 0; a < NI; a++)
        if (ci.atom[a] >= 0) {
          const double dx = ci.x[a] - bx, dy = ci.y[a] - by, dz = ci.z[a] - bz;
          brad2 = fmax(brad2, dx * dx + dy * dy + dz * dz);
        }
    }
    const double breach = sqrt(S.rlist2) + sqrt(brad2) * 1.0000001 + 1.0e-9;
    const double breach2 = breach * breach;
    int nA = 0, nB = 0, nC = 0;
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
          int len = 1;
          while (o0 + len <= S.mst[0] && a0 + len < S.nc[0]) len++;
          o0 += len;
          if (s0 < -1 || s0 > 1) continue;
          const double sx = b.h[0] * s0 + b.h[5] * s1 + b.h[4] * s2;
          const double sy = b.h[1] * s1 + b.h[3] * s2;
          const double sz = b.h[2] * s2;
          const bool home = (s0 == 0 && s1 == 0 && s2 == 0);
          const int code = ((s2 + 1) * 9 + (s1 + 1) * 3 + (s0 + 1)) << MD_CODE_SHIFT;
          const int cj = (a2 * S.nc[1] + a1) * S.nc[0] + a0;
          const int jb = S.cell_start[cj], je = S.cell_start[cj + len];
          for (int base = jb; base < je; base += 64) {
            const int j = base + lane;
            int mask = 0;
            double rmin = 1.0e300;
            if (j < je) {
              const double xj = xq[2 * (size_t)j] + sx, yj = xq[2 * (size_t)j + 1] + sy, zj = zq[2 * (size_t)j] + sz;
              int aj = -2;
              const double cx = bx - xj, cy = by - yj, cz = bz - zj;
              if (cx * cx + cy * cy + cz * cz < breach2)
#pragma unroll
              for (int a = 0; a < NI; a++) {
                const double dx = ci.x[a] - xj, dy = ci.y[a] - yj, dz = ci.z[a] - zj;
                const double r2 = dx * dx + dy * dy + dz * dz;
                bool acc = ci.atom[a] >= 0 && r2 < S.rlist2 && !(home && j == s0slot + a);
                if (acc && r2 < S.excl_cut2) {
                  if (aj == -2) aj = S.perm[j];
                  for (int e = S.ex_start[ci.atom[a]]; e < S.ex_start[ci.atom[a] + 1]; e++) acc = acc && (S.ex_list[e] != aj);
                }
                if (acc) {
                  mask |= 1 << a;
                  rmin = fmin(rmin, r2);
                }
              }
            }
            const bool isA = mask && rmin < ra2, isB = mask && !isA && rmin < rb2, isC = mask && !isA && !isB;
            const unsigned long long mA = __ballot(isA), mB = __ballot(isB), mC = __ballot(isC);
            if (mask) {
              const int entry = code | (mask << MD_MASK_SHIFT) | j;
              if (isA) { const int pos = nA + popc_below(mA); if (pos < maxrow) row[pos] = entry; }
              else if (isB) { const int pos = nB + popc_below(mB); if (pos < capB) lb[pos] = entry; }
              else { const int pos = maxrow - 1 - (nC + popc_below(mC)); if (pos >= 0) row[pos] = entry; }
            }
            nA += __popcll(mA); nB += __popcll(mB); nC += __popcll(mC);
            npairs += __popc(mask);
          }
        }
      }
    }
    const int n = nA + nB + nC;
    const bool bad = nB > capB || n > maxrow;
    if (bad) over = 1;
    // B: LDS list -> behind A (same-wave LDS traffic is processed in order)
    const int mB_ = bad ? 0 : nB;
    for (int k = lane; k < mB_; k += 64) row[nA + k] = lb[k];
    if (lane == 0) { S.numneigh[2 * cl] = bad ? 0 : nA + nB; S.numneigh[2 * cl + 1] = bad ? 0 : nC; }
    nmax = max(nmax, n);
    nrowent += n;
  }
  const double cnt = wave_sum((double)npairs);
  if (lane == 0) {
    if (over) atomicOr(&sc.overflow, 1);
    atomicMax(&sc.maxneigh_seen, nmax);
    atomicAdd(&sc.nentries, (unsigned long long)cnt);
    atomicAdd(&sc.nrowent, nrowent);
  }
}


// ------------------------------------------------------------------------------------------
// k_pair
// ------------------------------------------------------------------------------------------
// NP = number of polynomial coefficients kept in scalar registers (>= fitted degree+1, 0-padded)
template <bool VIR, bool ENG, int NP>
__global__ __launch_bounds__(WPB * 64, 4) void k_pair(const SimDev *__restrict__ sims, int ntiles, int nsims) {
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
  const GLOBAL_AS double *xq = as_global((const double *)S.xq);   // (x,y) halves
  const GLOBAL_AS double *zq = xq + 2 * (size_t)S.npad;              // (z,q) halves
  const GLOBAL_AS int *stype = as_global(S.stype);
  const double g = S.g_ewald, g2u = g * g * S.coul_uscale;
  const double cutc2 = S.cut_coul2, cutl2 = S.cut_lj2;
  const double cutmax2 = fmax(cutc2, cutl2);
  const int maxrow = S.maxneigh;
  double vl[6] = {0, 0, 0, 0, 0, 0}, vc[6] = {0, 0, 0, 0, 0, 0};
  double elj = 0, ecoul = 0;
  for (int c = 0; c < CPW; c++) {
    const int cl = tile * CPB + wave * CPW + c;  // wave-uniform
    const int s0 = cl * NI;
    if (s0 >= S.npad) break;
    const int nab = S.numneigh[2 * cl], nn = nab + S.numneigh[2 * cl + 1];  // [A|B] from the front, C reversed from the back
    if (nn == 0) continue;
#define ROW_AT(k) row[((k) < nab) ? (k) : (maxrow - 1 - ((k) - nab))]
    double xi[NI], yi[NI], zi[NI], qi[NI], fx[NI], fy[NI], fz[NI];
    int ti[NI];
#pragma unroll
    for (int a = 0; a < NI; a++) {
      xi[a] = XQ_X(S, s0 + a); yi[a] = XQ_Y(S, s0 + a); zi[a] = XQ_Z(S, s0 + a);
      qi[a] = MD_QQRD2E * XQ_Q(S, s0 + a);
      ti[a] = S.stype[s0 + a] * nt;
      fx[a] = fy[a] = fz[a] = 0.0;
    }
    const GLOBAL_AS int *row = as_global(S.neigh) + (size_t)cl * maxrow;
    // one row ahead: entry + record + type of row r+1 are in flight while row r is evaluated
    int e_n = (lane < nn) ? ROW_AT(lane) : 0;
    double xn0, xn1, xn2, xn3;
    int tn;
    {
      const size_t j = (size_t)(e_n & MD_JMASK);
      xn0 = xq[2 * j]; xn1 = xq[2 * j + 1]; xn2 = zq[2 * j]; xn3 = zq[2 * j + 1];
      tn = stype[j];
    }
    for (int k0 = 0; k0 < nn; k0 += 64) {
      const int e = e_n;
      const double xj = xn0, yj = xn1, zj = xn2, qj = xn3;
      const int tj = tn;
      {
        const int kn = k0 + 64 + lane;
        e_n = (kn < nn) ? ROW_AT(kn) : 0;
        const size_t j = (size_t)(e_n & MD_JMASK);
        xn0 = xq[2 * j]; xn1 = xq[2 * j + 1]; xn2 = zq[2 * j]; xn3 = zq[2 * j + 1];
        tn = stype[j];
      }
      const int mask = (e >> MD_MASK_SHIFT) & 0xF;  // 0 for the padding of the last row
      if (mask == 0) continue;
      const int cs = 4 * (((unsigned)e) >> MD_CODE_SHIFT);
      const double xs = xj + s_shift[cs], ys = yj + s_shift[cs + 1], zs = zj + s_shift[cs + 2];
#pragma unroll
      for (int a = 0; a < NI; a++) {
        if (!(mask & (1 << a))) continue;
        const double dx = xi[a] - xs, dy = yi[a] - ys, dz = zi[a] - zs;
        const double rsq = dx * dx + dy * dy + dz * dz;
        if (rsq < cutmax2) {
          const double rinv = rsqrt_f64(rsq);
          const double r2inv = rinv * rinv;
          double flj = 0.0, fc = 0.0;
          if (rsq < cutc2) {
            // erfc(x) + 2x/sqrt(pi) exp(-x^2) = 1 - x H(u): Horner in t = u*uscale - 1
            const double x = g * rsq * rinv;
            const double t = fma(rsq, g2u, -1.0);
            double p = cp[NP - 1];
#pragma unroll
            for (int m = NP - 2; m >= 0; m--) p = fma(p, t, cp[m]);
            const double pref = qi[a] * qj * rinv;
            fc = pref * fma(-x, p, 1.0) * r2inv;
            if (ENG) ecoul += pref * erfc(x);
          }
          if (rsq < cutl2) {
            const int tt = ti[a] + tj;
            const double r6inv = r2inv * r2inv * r2inv;
            flj = r6inv * (s_lj[tt] * r6inv - s_lj[nt2 + tt]) * r2inv;
            if (ENG) elj += r6inv * (s_lj[2 * nt2 + tt] * r6inv - s_lj[3 * nt2 + tt]);
          }
          const double fp = flj + fc;
          fx[a] = fma(dx, fp, fx[a]); fy[a] = fma(dy, fp, fy[a]); fz[a] = fma(dz, fp, fz[a]);
          if (VIR && !ENG) {
            // production: one lumped pair virial (the pressure sums all parts anyway)
            const double xl = dx * fp, yl = dy * fp, zl = dz * fp;
            vl[0] = fma(dx, xl, vl[0]); vl[1] = fma(dy, yl, vl[1]); vl[2] = fma(dz, zl, vl[2]);
            vl[3] = fma(dx, yl, vl[3]); vl[4] = fma(dx, zl, vl[4]); vl[5] = fma(dy, zl, vl[5]);
          }
          if (VIR && ENG) {
            // parity hook: LJ and coulomb parts separately
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
#pragma unroll
    for (int a = 0; a < NI; a++) {
      const double sx = wave_sum(fx[a]), sy = wave_sum(fy[a]), sz = wave_sum(fz[a]);
      if (lane == 0) {
        const int at = S.perm[s0 + a];
        if (at >= 0) { S.f[3 * at] = sx; S.f[3 * at + 1] = sy; S.f[3 * at + 2] = sz; }
      }
    }
  }
  if (VIR) {
    // full list: every pair is visited from both ends
    for (int k = 0; k < 6; k++) { vl[k] *= 0.5; vc[k] *= 0.5; }
    block_atomic_add<6>(vl, sc.vir + P_LJ * 6, s_red);
    if (ENG) block_atomic_add<6>(vc, sc.vir + P_COUL * 6, s_red);
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

void mdk_neigh_build(hipStream_t st, const SimDev *d, int ns, int maxpad, int maxrow) {
  const int ntiles = cdiv(maxpad / NI, CPB);
  // per-wave LDS list of segment B: well over its expected share (~40 %) of a full row
  const int capB = (int)(0.55 * maxrow) / 64 * 64 + 64;
  const size_t lds = (size_t)WPB * capB * sizeof(int);
  static size_t optin = 0;  // more than 64 KB of dynamic LDS needs an explicit opt-in (very dense systems)
  if (lds > 64 * 1024 && lds > optin) { (void)hipFuncSetAttribute((const void *)k_neigh_build, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); optin = lds; }
  hipLaunchKernelGGL(k_neigh_build, grid_xcd(ntiles, ns), dim3(WPB * 64), lds, st, d, ntiles, ns, capB);
}

template <int NP>
static void launch_pair(hipStream_t st, const SimDev *d, int ns, int ntiles, int vir, int eng) {
  const dim3 g = grid_xcd(ntiles, ns);
  if (eng) hipLaunchKernelGGL((k_pair<true, true, NP>), g, dim3(WPB * 64), 0, st, d, ntiles, ns);
  else if (vir) hipLaunchKernelGGL((k_pair<true, false, NP>), g, dim3(WPB * 64), 0, st, d, ntiles, ns);
  else hipLaunchKernelGGL((k_pair<false, false, NP>), g, dim3(WPB * 64), 0, st, d, ntiles, ns);
}

void mdk_pair(hipStream_t st, const SimDev *d, int ns, int maxpad, int vir, int eng, int npoly) {
  const int ntiles = cdiv(maxpad / NI, CPB);
  if (npoly <= 12) launch_pair<12>(st, d, ns, ntiles, vir, eng);
  else if (npoly <= 14) launch_pair<14>(st, d, ns, ntiles, vir, eng);
  else if (npoly <= 16) launch_pair<16>(st, d, ns, ntiles, vir, eng);
  else if (npoly <= 20) launch_pair<20>(st, d, ns, ntiles, vir, eng);
  else if (npoly <= 24) launch_pair<24>(st, d, ns, ntiles, vir, eng);
  else if (npoly <= 32) launch_pair<32>(st, d, ns, ntiles, vir, eng);
  else launch_pair<MD_MAXPOLY>(st, d, ns, ntiles, vir, eng);
}
