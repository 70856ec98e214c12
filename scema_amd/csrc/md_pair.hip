// md_pair.hip -- neighbour-list build and the lj/cut/coul/long pair kernel (the roofline kernel).
//
// Work decomposition (measurements of every step are in DESIGN.md §5):
//   * TILE = one cell of the binning grid, one 512-thread workgroup per tile.  Every unordered pair
//     is evaluated ONCE (Newton's third law): the tile of cell c owns the pairs (i in c, j in c+o)
//     for the cell offsets o that are lexicographically positive (o2 > 0 | o2 == 0 & o1 > 0 | ... ),
//     offsets counted before periodic wrapping, plus the pairs inside c with slot(j) > slot(i).
//     The rule is integer-only and antisymmetric, so no pair is lost or doubled.
//   * J TABLE: the tile's candidate j images (slot | image code), own cell first, are numbered
//     0..nj-1 at build time.  k_pair keeps one FP64 force accumulator per table entry in LDS
//     (24 B x ~2 100 entries for PE-10k): reaction forces are LDS atomics (ds_add_f64), the table is
//     flushed once per tile with coalesced global atomics into the slot-ordered force array.
//   * i-CLUSTERS: 4 consecutive slots of the cell (cells are padded to multiples of 4 slots, pad
//     slots hold far-away dummy records).  A cluster owns ONE row = the union of its atoms'
//     neighbours, each entry carrying a 4-bit mask of which i atoms list that j.
//   * ONE WAVE PER CLUSTER, the 64 lanes span the row (contiguous 256 B per wave instruction); a lane
//     gathers its j record once and evaluates it against up to 4 i atoms held in scalar registers.
//   * rows keep table order inside three segments by build-time distance (A: may need coulomb,
//     B: LJ only, C: skin), so the three regimes are wave-uniform branches.  Every lane still tests
//     r^2 against the cutoffs: results never depend on the segments or on the cluster grouping.
//   entry = i-mask [20:17] | type of j [16:13] | index into the tile's j table [12:0]
//   j table entry = image code [27:23] | j slot [22:0]
//
// Reference semantics: pair_style lj/cut/coul/long 12.0 9.0 (in.set.lammps:40), neighbor 2.0 bin
// (in.set.lammps:27); special_bonds 0 0 1 pairs are excluded here and handled in k_bonded (md_bonded.hip).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "md_device.h"
#include "md_env.h"
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

// ------------------------------------------------------------------------------------------
// k_neigh_build : workgroup per cell.
//   phase 0: bounding boxes of the four groups of the cell's clusters; slot runs of the half stencil (own cell first)
//   phase 1 (all waves, units of 64 candidates): candidate j images pruned against the bounding box of the cell's
//            atoms and numbered into the tile's j table, and against the groups' boxes into the groups' candidate
//            lists -- one pass, candidate order (deterministic).  An accepted candidate leaves its position as an FP32
//            record RELATIVE TO THE TILE'S ORIGIN in LDS (the table entry itself goes straight to global memory).
//   phase 2 (wave per cluster): the group's list is tested against the cluster's 4 atoms; ballots compact
//            the accepted entries into the row segments
//   the rows are dealt round robin to the waves of k_pair
// What has to be exact and what has not: k_pair tests every r^2 in FP64 against the cutoffs, so a row only has to be a SUPERSET of
// the pairs inside the list radius.  Every build but the first of a run therefore tests its candidates in FP32 on tile-relative
// coordinates (|coordinate| <= M = half the cell's extent + list radius) against a radius widened by the error bound of that
// arithmetic, eps = 2^-24 (96 M r + 8 r^2) in r^2 (derivation at nb_eps): FP32 instructions issue at twice the FP64 rate, the
// records come out of LDS instead of two gathers, an image shift and three FP64 additions per candidate.  The first build of a
// run (sc.step == 0: also every static evaluation of the parity hook) keeps the FP64 test at the exact radius and takes the
// statistics there (pairs inside the list radius, the count the tests compare with the oracle's; the reference-radius count of a
// wider list); SCEMA_MD_NEIGH_EXACT=1 makes every build exact, =0 none (test switches).
// SCEMA_MD_QCAP16 (test switch) shrinks the group lists so that they overflow: the whole-table walk of phase 2.
// ------------------------------------------------------------------------------------------
struct ClusterI {
  double x[NI], y[NI], z[NI];
  int atom[NI];   // real atom index or -1 (pad)
};

extern __shared__ int s_build[];  // [3][capj + 64] FP32 records of the j table (x, y, z relative to the tile's origin; entry capj: a far dummy), then
                                  // [TW][capB] per-wave lists (segment B from the front, the skin band from the back), then
                                  // [NQ][qcap] 16-bit table indices: the part of the table each group (quarter) of the cell's clusters can reach
#define NQ 4      // groups of a cell's clusters with their own candidate list (<= TW: one wave takes each group's bounding box)
#define NB_MAXRUN 128    // slot runs of one tile's candidates (own cell + half stencil; 20 for PE-10k)
#define NB_MAXUNIT 1024  // 64-candidate units of one tile (85 for PE-10k)
#define NB_UPW 4         // units per wave and round (TW * NB_UPW = 32: scan32_incl)
#define NB_RECPAD 64     // records behind the table: [capj] is the dummy the lanes past the end of a list read
#define NB_FAR 1.0e18f   // FP32 place of the dummy record and (negated) of the pad atoms of an i-cluster: (2e18)^2 * 3 is finite

// Error bound of the FP32 tests.  u = 2^-24.  A tile-relative coordinate X, |X| <= M, is stored as fl(X): off by <= u M.  A difference
// of two stored coordinates is exact up to its own rounding, so d = fl(xi - xj) is off the true difference by <= 2 u M + u |d| <= 3 u M;
// r^2 = fl(dx^2 + dy^2 + dz^2) carries at most 4 roundings, relative 4 u.  |r2_f - r^2| <= 2 (|dx| + |dy| + |dz|) 3 u M + 4 u r^2
// <= 6 sqrt(3) u M r + 4 u r^2 < u (11 M r + 4 r^2) for pairs at the radius r.  The box tests of phase 1 (distance of a point from a
// box whose FP32 edges are off by <= u M as well) obey the same bound.  The kernel uses u (96 M r + 8 r^2): an order of magnitude of
// slack, and the band it adds to a 14 A list is 1e-4 A wide.
__device__ __forceinline__ float nb_eps(double M, double r2) {
  const double r = sqrt(r2);
  return (float)(1.0001 * 5.9604644775390625e-8 * (96.0 * M * r + 8.0 * r2));
}
// v < t, rounded so that the FP32 test can only err towards "inside"
__device__ __forceinline__ float nb_up(double t, float eps) { return (float)(t * (1.0 + 2.4e-7)) + eps; }

// (TT, 4): at most 128 registers, so that two workgroups share a CU -- at 129 the kernel ran 1.6 times longer
__global__ __launch_bounds__(TT, 4) void k_neigh_build(const SimDev *__restrict__ sims, int ntiles, int nsims, int capj, int capB, int qcap, int exact_mode) {
  int sim, cell;
  // Which replicas rebuild in a given step is random (one in ~16 of them, each at its own time).  With the tiles of a replica pinned
  // to one XCD (k_pair's map) the XCD that happens to hold the most rebuilding replicas sets the launch time; consecutive blocks =
  // consecutive tiles of one replica instead deals every rebuilding replica's tiles over all eight XCDs (its 332 KB of positions
  // are then read by each of them: nothing next to the 10 MB of rows it writes).
  // (measured against k_pair's map in round 3: -75 us per 576-replica step)
  sim = blockIdx.x / ntiles;
  cell = blockIdx.x % ntiles;
  if (sim >= nsims) return;
  const SimDev &S = sims[sim];
  SimScalars &sc = *S.sc;
  if (!sc.rebuild) return;
  if (cell >= S.ncells) return;
#ifdef PAIR_TIMING
  const unsigned long long tb0 = __builtin_readcyclecounter();
#endif
  const int cs = __builtin_amdgcn_readfirstlane(S.cell_start[cell]), ce = __builtin_amdgcn_readfirstlane(S.cell_start[cell + 1]), nown = ce - cs;   // uniform, and said so
  if (nown == 0) {
    if (threadIdx.x == 0) S.tile_nj[cell] = 0;
    return;
  }
  __shared__ double s_shift[27 * 3];  // image shifts (the exact row loop and the candidates of phase 1, which subtract the tile's origin themselves)
  __shared__ int s_rjb[NB_MAXRUN], s_rlen[NB_MAXRUN], s_rcode[NB_MAXRUN], s_ub[NB_MAXRUN + 1];   // slot runs of the candidates, first unit of each
  __shared__ unsigned char s_urun[NB_MAXUNIT];                                                    // run of each 64-candidate unit
  __shared__ int s_ucnt[2][1 + NQ][TW * NB_UPW];                                                  // accepted per unit of a round: table, group lists
  __shared__ int s_ex[TW][NI * 16];   // exclusion lists of the cluster a wave is working on (first 16 per atom)
  __shared__ double s_qbox[NQ][6];    // bounding boxes of the quarters of the cell's clusters (k-d order: quarters are compact)
  __shared__ float s_qboxf[NQ][6];    // the same relative to the tile's origin, FP32
  __shared__ int s_qn[NQ];            // entries of a quarter's list; -1: list overflowed, the quarter walks the whole table
  const int capr = capj + NB_RECPAD;
  float *s_rx = (float *)s_build, *s_ry = s_rx + capr, *s_rz = s_ry + capr;
  unsigned short *s_qlist = (unsigned short *)(s_build + 3 * capr) + TW * capB;   // (capB is a multiple of 64)
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned short *lb = (unsigned short *)(s_build + 3 * capr) + wave * capB;   // this wave's staging list (phase 2)

  BoxD b;
  box_derive(sc.box, b);
  if (threadIdx.x < 27) {
    const int s0 = threadIdx.x % 3 - 1, s1 = (threadIdx.x / 3) % 3 - 1, s2 = threadIdx.x / 9 - 1;
    s_shift[3 * threadIdx.x + 0] = b.h[0] * s0 + b.h[5] * s1 + b.h[4] * s2;
    s_shift[3 * threadIdx.x + 1] = b.h[1] * s1 + b.h[3] * s2;
    s_shift[3 * threadIdx.x + 2] = b.h[2] * s2;
  }
  // (capj, the kernel's argument, is the largest table of the launch and lays out the LDS; a replica's own capacity S.capj <= capj bounds
  // what is written to ITS table in memory, and entry S.capj -- never a valid one -- is the dummy that lanes past the end of a list
  // name: a far record in LDS, and in memory the first word behind this tile's table: the next tile's, or the buffer's slack)
  const int capjs = S.capj;
  if (threadIdx.x == 32) { s_rx[capjs] = NB_FAR; s_ry[capjs] = NB_FAR; s_rz[capjs] = NB_FAR; }
  const GLOBAL_AS double *xq = as_global((const double *)S.xq);               // (x,y) halves
  const GLOBAL_AS double *zq = xq + 2 * (size_t)S.npad;                          // (z,q) halves
  // ---- phase 0: bounding boxes of the groups of the cell's clusters; slot runs of the half stencil ----
  // The clusters of a cell are in k-d order, so a group of consecutive clusters (a quarter of the cell's) is a compact region: a
  // candidate farther than rlist from the group's box is left out of the group's list (conservative: no atom of the group can list
  // it) and the group's clusters walk the list instead of the table (a cluster tests its candidates with 4 atoms x 64 lanes per
  // chunk whatever the outcome).  Lists keep table order, so rows do not depend on them.
  const int nclus_cell = nown / NI;
  if (wave < NQ) {
    const int c_lo = (wave * nclus_cell) / NQ, c_hi = ((wave + 1) * nclus_cell) / NQ;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int sl = cs + c_lo * NI + lane; sl < cs + c_hi * NI; sl += 64)
      if (S.perm[sl] >= 0) {
        const double x = xq[2 * (size_t)sl], y = xq[2 * (size_t)sl + 1], z = zq[2 * (size_t)sl];
        lo[0] = fmin(lo[0], x); hi[0] = fmax(hi[0], x);
        lo[1] = fmin(lo[1], y); hi[1] = fmax(hi[1], y);
        lo[2] = fmin(lo[2], z); hi[2] = fmax(hi[2], z);
      }
    for (int d = 0; d < 3; d++) {
      const double l = -wave_max_dpp(-lo[d]), h = wave_max_dpp(hi[d]);
      if (lane == 0) { s_qbox[wave][d] = l; s_qbox[wave][3 + d] = h; }   // (an empty group keeps an inverted box: nothing passes)
    }
  }
  // Candidates = the slots of the own cell (run 0: table index l <-> slot cs + l, pads included, so that the cluster atoms know their
  // own index) and of the cells of the half stencil, as runs of consecutive slots (the x range of cells at fixed (o2, o1) is one run
  // per periodic image).  The loop below is scalar arithmetic and names the runs by their cells; lane r of wave 0 then fetches the two
  // cell boundaries of run r, so that the loads of all runs are in flight together.
  // (scalars of the replica that the loops below use, read once: behind the LDS stores the compiler reloads them at every use)
  const int nc0 = S.nc[0], nc1 = S.nc[1], nc2 = S.nc[2], mst0 = S.mst[0], mst1 = S.mst[1], mst2 = S.mst[2];
  const double rl2 = S.rlist2;
  const int c0 = cell % nc0, c1 = (cell / nc0) % nc1, c2 = cell / (nc0 * nc1);
  int nrun = 1;
  if (threadIdx.x == 0) { s_rjb[0] = cs; s_rlen[0] = nown; s_rcode[0] = CODE_HOME; }
  for (int o2 = 0; o2 <= mst2; o2++) {
    int a2 = c2 + o2, s2 = 0;
    while (a2 >= nc2) { a2 -= nc2; s2 += 1; }
    if (s2 > 1) continue;
    for (int o1 = (o2 == 0 ? 0 : -mst1); o1 <= mst1; o1++) {
      int a1 = c1 + o1, s1 = 0;
      while (a1 < 0) { a1 += nc1; s1 -= 1; }
      while (a1 >= nc1) { a1 -= nc1; s1 += 1; }
      if (s1 < -1 || s1 > 1) continue;
      int o0 = (o2 == 0 && o1 == 0) ? 1 : -mst0;
      while (o0 <= mst0) {
        int a0 = c0 + o0, s0 = 0;
        while (a0 < 0) { a0 += nc0; s0 -= 1; }
        while (a0 >= nc0) { a0 -= nc0; s0 += 1; }
        int len = 1;
        while (o0 + len <= mst0 && a0 + len < nc0) len++;
        o0 += len;
        if (s0 < -1 || s0 > 1) continue;
        if (threadIdx.x == 0 && nrun < NB_MAXRUN) {   // (cells for now; wave 0 turns them into slots below, all runs' loads in flight together)
          s_rjb[nrun] = (a2 * nc1 + a1) * nc0 + a0; s_rlen[nrun] = len;
          s_rcode[nrun] = (s2 + 1) * 9 + (s1 + 1) * 3 + (s0 + 1);
        }
        nrun++;
      }
    }
  }
  __syncthreads();
  // units of 64 consecutive candidates of one run, numbered in run order: s_ub[r] = first unit of run r, s_urun[u] = run of unit u
  if (wave == 0) {
    int carry = 0;
    for (int r0 = 0; r0 < min(nrun, NB_MAXRUN); r0 += 64) {
      const int r = r0 + lane;
      if (r > 0 && r < min(nrun, NB_MAXRUN)) {
        const int cj = s_rjb[r], jb = S.cell_start[cj];
        s_rlen[r] = S.cell_start[cj + s_rlen[r]] - jb;
        s_rjb[r] = jb;
      }
      const int nu = (r < min(nrun, NB_MAXRUN)) ? (s_rlen[r] + 63) >> 6 : 0;
      int incl = nu;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
      const int first = carry + incl - nu;
      if (r < min(nrun, NB_MAXRUN)) {
        s_ub[r] = first;
        for (int k = 0; k < nu; k++) if (first + k < NB_MAXUNIT) s_urun[first + k] = (unsigned char)r;
      }
      carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) s_ub[NB_MAXRUN] = carry;
  }
  // the cell's box = union of its groups' boxes; its centre is the tile's origin; the groups' boxes relative to it in FP32
  double blo0 = 1e300, blo1 = 1e300, blo2 = 1e300, bhi0 = -1e300, bhi1 = -1e300, bhi2 = -1e300;
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    blo0 = fmin(blo0, s_qbox[q][0]); blo1 = fmin(blo1, s_qbox[q][1]); blo2 = fmin(blo2, s_qbox[q][2]);
    bhi0 = fmax(bhi0, s_qbox[q][3]); bhi1 = fmax(bhi1, s_qbox[q][4]); bhi2 = fmax(bhi2, s_qbox[q][5]);
  }
  const double ox = wave_uniform(0.5 * (blo0 + bhi0)), oy = wave_uniform(0.5 * (blo1 + bhi1)), oz = wave_uniform(0.5 * (blo2 + bhi2));
  // (a cell always holds a real atom here -- nown > 0 and pads only fill the last cluster -- so the box is a proper one)
  const double Mrel = wave_uniform(0.5 * fmax(bhi0 - blo0, fmax(bhi1 - blo1, bhi2 - blo2)) + sqrt(rl2));
  const float eps = nb_eps(Mrel, rl2);
  const float rl2e = nb_up(rl2, eps);
  if (threadIdx.x >= 64 && threadIdx.x < 64 + NQ * 6) {
    const int q = (threadIdx.x - 64) / 6, k = (threadIdx.x - 64) % 6;
    s_qboxf[q][k] = (float)(s_qbox[q][k] - (k % 3 == 0 ? ox : k % 3 == 1 ? oy : oz));
  }
  const float hx = (float)(bhi0 - ox), hy = (float)(bhi1 - oy), hz = (float)(bhi2 - oz);   // half extents of the cell's box (the origin is its centre)
  __syncthreads();
  const int nunit = s_ub[NB_MAXRUN];
#ifdef PAIR_TIMING
  const unsigned long long tb05 = __builtin_readcyclecounter();
#endif
  // ---- phase 1: candidates -> j table (pruned against the cell's box) and group lists (against the groups' boxes), one pass ----
  // Rounds of TW * NB_UPW units: a wave takes NB_UPW units per round (their loads in flight together), ballots give each unit's
  // counts, ONE barrier per round, then every wave takes the prefix over the round's units and writes its accepted entries: table and
  // lists come out in candidate order whatever wave handled what.
  // (inside this kernel the table entries also carry the type of j in bits 28..31: the rows need it per accepted candidate, and
  // one load per table entry here replaces one per cluster and entry there; k_pair masks its reads of the table)
  const GLOBAL_AS int *stype = as_global(S.stype);
  GLOBAL_AS int *gj = as_global_w(S.tile_jtab) + (size_t)cell * S.capj;
  int nj = 0;
  int qn[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) qn[q] = 0;
  const bool runs_ok = nrun <= NB_MAXRUN && nunit <= NB_MAXUNIT;   // (uniform; otherwise reported as a table overflow below)
  for (int u0 = 0, par = 0; runs_ok && u0 < nunit; u0 += TW * NB_UPW, par ^= 1) {
    int jv[NB_UPW], cv[NB_UPW];
    double px[NB_UPW], py[NB_UPW], pz[NB_UPW];
    int tv[NB_UPW];
    bool valid[NB_UPW], home[NB_UPW];
#pragma unroll
    for (int i = 0; i < NB_UPW; i++) {
      const int u = u0 + wave * NB_UPW + i;
      valid[i] = false; home[i] = false; jv[i] = 0; cv[i] = CODE_HOME;
      if (u < nunit) {
        const int r = s_urun[u];
        const int off = ((u - s_ub[r]) << 6) + lane;
        valid[i] = off < s_rlen[r];
        home[i] = r == 0;
        jv[i] = valid[i] ? s_rjb[r] + off : cs;
        cv[i] = s_rcode[r];
      }
      const size_t j = (size_t)jv[i];
      px[i] = xq[2 * j]; py[i] = xq[2 * j + 1]; pz[i] = zq[2 * j];
      tv[i] = stype[j];
    }
    bool ok[NB_UPW];
    unsigned okq[NB_UPW];
    unsigned long long m[NB_UPW];
    float xf[NB_UPW], yf[NB_UPW], zf[NB_UPW];
#pragma unroll
    for (int i = 0; i < NB_UPW; i++) {
      // the candidate's image relative to the tile's origin, FP64 up to the conversion (a pad's 1e15 stays a finite FP32 number)
      xf[i] = (float)(px[i] + (s_shift[3 * cv[i]] - ox)); yf[i] = (float)(py[i] + (s_shift[3 * cv[i] + 1] - oy)); zf[i] = (float)(pz[i] + (s_shift[3 * cv[i] + 2] - oz));
      {
        const float ex = fmaxf(0.f, fabsf(xf[i]) - hx), ey = fmaxf(0.f, fabsf(yf[i]) - hy), ez = fmaxf(0.f, fabsf(zf[i]) - hz);
        ok[i] = valid[i] && (home[i] || ex * ex + ey * ey + ez * ez < rl2e);
      }
      okq[i] = 0;
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        const float *bq = s_qboxf[q];
        const float ex = fmaxf(0.f, fmaxf(bq[0] - xf[i], xf[i] - bq[3])), ey = fmaxf(0.f, fmaxf(bq[1] - yf[i], yf[i] - bq[4])), ez = fmaxf(0.f, fmaxf(bq[2] - zf[i], zf[i] - bq[5]));
        okq[i] |= (ok[i] && ex * ex + ey * ey + ez * ez < rl2e) ? (1u << q) : 0u;
      }
      m[i] = __ballot(ok[i]);
      const int ui = wave * NB_UPW + i;
      if (lane == 0) s_ucnt[par][0][ui] = __popcll(m[i]);
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        const unsigned long long mq = __ballot((okq[i] >> q) & 1u);
        if (lane == 0) s_ucnt[par][1 + q][ui] = __popcll(mq);
      }
    }
    __syncthreads();
    // prefix over the round's units, per counter: lane u holds unit u's counts
    int ex_c[1 + NQ], tot_c[1 + NQ];
#pragma unroll
    for (int c = 0; c <= NQ; c++) {
      const int v = (lane < TW * NB_UPW) ? s_ucnt[par][c][lane] : 0;
      const int incl = scan32_incl(v);
      ex_c[c] = incl - v;
      tot_c[c] = __builtin_amdgcn_readlane(incl, TW * NB_UPW - 1);
    }
#pragma unroll
    for (int i = 0; i < NB_UPW; i++) {
      const int ui = wave * NB_UPW + i;
      const int pos = nj + __builtin_amdgcn_readlane(ex_c[0], ui) + popc_below(m[i]);
      if (ok[i] && pos < capjs) {
        gj[pos] = jv[i] | (cv[i] << 23) | (tv[i] << 28);
        s_rx[pos] = xf[i]; s_ry[pos] = yf[i]; s_rz[pos] = zf[i];
      }
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        const bool in_q = (okq[i] >> q) & 1u;
        const unsigned long long mq = __ballot(in_q);
        const int lp = qn[q] + __builtin_amdgcn_readlane(ex_c[1 + q], ui) + popc_below(mq);
        if (in_q && lp < qcap) s_qlist[q * qcap + lp] = (unsigned short)(pos | (tv[i] << 12));   // (pos < capj <= 4032)
      }
    }
    nj += tot_c[0];
#pragma unroll
    for (int q = 0; q < NQ; q++) qn[q] += tot_c[1 + q];
  }
  // k_pair keeps a wave's row headers in one VGPR triple (lane r = r-th row): at most 64 rows per wave, 64*TW clusters
  // per cell.  A denser cell is reported like a table overflow (the engine retries with smaller cells), never dropped.
  if (!runs_ok || nj > capjs || nown / NI > 64 * TW) {   // uniform: table overflow -> the engine regrows and retries
    if (threadIdx.x == 0) { S.tile_nj[cell] = 0; atomicOr(&sc.overflow, 1 | 4); atomicMax(&sc.maxj_seen, runs_ok ? nj : 2 * S.capj); }   // 4: table
    for (int cl = cs / NI + threadIdx.x; cl < ce / NI; cl += TT) { S.numneigh[2 * cl] = 0; S.numneigh[2 * cl + 1] = 0; }
    return;
  }
  if (threadIdx.x == 0) { S.tile_nj[cell] = nj; atomicMax(&sc.maxj_seen, nj); }
#pragma unroll
  for (int q = 0; q < NQ; q++)
    if ((int)threadIdx.x == q) s_qn[q] = (qn[q] > qcap) ? -1 : qn[q];
  // (the table entries of this tile written above are read back by this workgroup's row loops: stores and loads of one workgroup to
  // global memory are ordered by the barrier once the stores have left the waves)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();

  // ---- phase 2 ----
#ifdef PAIR_TIMING
  const unsigned long long tb1 = __builtin_readcyclecounter();
#endif
  const int maxrow = S.maxneigh;
  const double ra2 = S.seg_a2, rb2 = fmax(S.seg_a2, S.seg_b2), rc2 = fmax(rb2, S.seg_c2), excl2 = S.excl_cut2;
  unsigned int npairs = 0, npairs_ref = 0;   // per lane and tile: far below 2^32
  unsigned long long nrowent = 0;
  // The first build of a run is the exact one (header).  Only a list wider than the reference's needs the second count (uniform),
  // and it is a statistic (the algorithmic bytes of the roofline): taken at that build, kept until the next run (it moves by < 0.1 %
  // between the builds of a run).
  const bool wider = S.rlist_ref2 < S.rlist2;
  const bool exact = exact_mode == 1 || (exact_mode != 0 && sc.step == 0);
  const bool count_ref = wider && exact && sc.step == 0;
  const float ra2e = nb_up(ra2, eps), rb2e = nb_up(rb2, eps), rc2e = nb_up(rc2, eps), excl2e = nb_up(excl2, eps);
  const GLOBAL_AS int *gjr = (const GLOBAL_AS int *)gj;
  // The wave's staging list, 16-bit entries (position in the group's list | i-mask << 12): segment B from the front of [0, capBC),
  // the near skin band C1 from its back, the far band C2 in [capBC, capB).  (32-bit entries were 41 KB of LDS for the eight waves.)
  const int capD = capB >> 2, capBC = capB - capD;
  int nmax = 0, over = 0;
  for (int cl = cs / NI + wave; cl < ce / NI; cl += TW) {
    const int s0slot = cl * NI;
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
    // (a bounding-sphere test that skipped whole chunks out of the cluster's reach paid before the quarter lists existed; with them it
    // costs more than it saves: 1 607 against 1 549 us per step without it)
    // the four atoms' exclusion lists (1-2, 1-3 partners) into LDS once: the candidates inside the exclusion gate
    // then compare against broadcast LDS reads instead of walking the lists in global memory lane by lane
    int exb[NI], exn[NI];
#pragma unroll
    for (int a = 0; a < NI; a++) {
      exb[a] = (ci.atom[a] >= 0) ? S.ex_start[ci.atom[a]] : 0;
      exn[a] = (ci.atom[a] >= 0) ? S.ex_start[ci.atom[a] + 1] - exb[a] : 0;
    }
    {
      const int a = lane >> 4, e = lane & 15;
      const int na = (a == 0) ? exn[0] : (a == 1) ? exn[1] : (a == 2) ? exn[2] : exn[3];
      const int ba = (a == 0) ? exb[0] : (a == 1) ? exb[1] : (a == 2) ? exb[2] : exb[3];
      s_ex[wave][lane] = (e < na) ? S.slot_of[S.ex_list[ba + e]] : -1;   // as slots: what the table entries name (no perm[] lookup per candidate)
    }
    int nA = 0, nB = 0, nC = 0, nD = 0;   // segments A, B, C1 (near skin band), C2 (far skin band)
    // the part of the table this cluster's quarter can reach (or the whole table if that list overflowed)
    int qq = 0;   // quarter q holds the clusters [q n / NQ, (q + 1) n / NQ) of the cell, as its bounding box was taken
#pragma unroll
    for (int q = 1; q < NQ; q++) qq += (cl - cs / NI >= (q * nclus_cell) / NQ) ? 1 : 0;
    const int qnl = s_qn[qq];
    const bool qall = qnl < 0;
    const int nl = qall ? nj : qnl;
    const unsigned short *ql = s_qlist + qq * qcap;
    // entry k of the cluster's candidate list as (table index | type of j << 12); past the end: the dummy record behind the table.
    // (A group list holds exactly that; the whole-table walk of an overflowed list takes the type from the table entry.)
    auto list_at = [&](int k) -> int { return (k < nl) ? (qall ? k : (int)ql[k]) : capjs; };
#ifdef PAIR_TIMING
    if (lane == 0) { atomicAdd(&sc.dbg[10], (unsigned long long)((nl + 63) >> 6)); atomicAdd(&sc.dbg[11], 1ull); }
#endif
    bool own_chunks = true;
    int posA_prev = -1, entA_prev = 0;
    // What both row loops do with a chunk once the i-mask of every candidate is known: own-cell rule, exclusions, segments, stores.
    // (lt: list entry; jt: table entry; R2X(a): the candidate's squared distance from atom a in the loop's arithmetic)
#define NB_CHUNK_TAIL(RMIN, R2X, RA, RB, RC, EXCL)                                                                                       \
      const int l = lt & 0xFFF;                                                                                                           \
      const int j = jt & MD_JMASK;                                                                                                        \
      if (own_chunks) {   /* (wave-uniform: the entries of the own cell come first in the table and in every list) */                     \
        /* same cell, same image: each pair once, by slot order -- atom a of the cluster keeps j only if j > s0slot + a.  One mask per */ \
        /* candidate instead of a test per atom (rmin may then be too small: a nearer segment is always allowed) */                       \
        const bool own = l < nown;                                                                                                        \
        const int d = j - s0slot;                                                                                                         \
        const int drop = !own ? 0 : (d <= 0 ? 0xF : (d > 3 ? 0 : (0xF << d) & 0xF));                                                      \
        mask &= ~drop; refm &= ~drop;                                                                                                     \
        own_chunks = __ballot(own) != 0ull;                                                                                               \
      }                                                                                                                                   \
      /* candidates inside the exclusion gate (bonded neighbours: a few chunks per row) take the wave-uniform slow path, which */        \
      /* looks at the four distances again and walks the exclusion lists */                                                               \
      if (__ballot(mask != 0 && RMIN < EXCL) != 0ull) {                                                                                   \
        if (mask != 0 && RMIN < EXCL) {                                                                                                   \
          _Pragma("unroll") for (int a = 0; a < NI; a++)                                                                                  \
            if ((mask & (1 << a)) && R2X(a) < EXCL) {                                                                                     \
              bool keep = true;                                                                                                           \
              const int nl16 = min(exn[a], 16);                                                                                           \
              for (int e = 0; e < nl16; e++) keep = keep && (s_ex[wave][a * 16 + e] != j);                                                \
              for (int e = 16; e < exn[a]; e++) keep = keep && (S.slot_of[S.ex_list[exb[a] + e]] != j);                                   \
              if (!keep) { mask &= ~(1 << a); refm &= ~(1 << a); }   /* rmin may stay too small: only the segment choice sees it */       \
            }                                                                                                                             \
        }                                                                                                                                 \
      }                                                                                                                                   \
      /* segments: A straight into the row (the store itself at the top of the next turn); B, C1, C2 into the wave's staging list */     \
      {                                                                                                                                   \
        const bool isA = mask && RMIN < RA, isB = mask && !isA && RMIN < RB, isS = mask && !isA && !isB;                                  \
        const bool isD = isS && !(RMIN < RC), isC = isS && !isD;                                                                          \
        const unsigned long long mA = __ballot(isA), mB = __ballot(isB), mC = __ballot(isC), mD = __ballot(isD);                          \
        if (mask) {                                                                                                                       \
          const unsigned short e16 = (unsigned short)((base + lane) | (mask << 12));                                                      \
          if (isA) {                                                                                                                      \
            const int pos = nA + popc_below(mA);                                                                                          \
            const int ty = qall ? (int)((unsigned)jt >> 28) : (lt >> 12);                                                                 \
            if (pos < maxrow) { posA_prev = pos; entA_prev = l | (ty << E_TYPE_SHIFT) | (mask << E_MASK_SHIFT); }                         \
          } else if (isB) { const int pos = nB + popc_below(mB); if (pos < capBC) lb[pos] = e16; }                                        \
          else if (isC) { const int pos = capBC - 1 - (nC + popc_below(mC)); if (pos >= 0) lb[pos] = e16; }                               \
          else { const int pos = nD + popc_below(mD); if (pos < capD) lb[capBC + pos] = e16; }                                            \
        }                                                                                                                                 \
        nA += __popcll(mA); nB += __popcll(mB); nC += __popcll(mC); nD += __popcll(mD);                                                   \
      }                                                                                                                                   \
      npairs += __popc(mask);

    if (exact) {
      // ---- the exact row loop (first build of a run): FP64 distances at the exact radius, positions gathered from memory ----
      // one chunk of the list ahead: entry + record of chunk r+1 are in flight while chunk r is tested
      int lt_n = list_at(lane);
      int jt_n = gjr[lt_n & 0xFFF];
      double pn0, pn1, pn2;
      {
        const size_t jn = (size_t)(jt_n & MD_JMASK);
        pn0 = xq[2 * jn]; pn1 = xq[2 * jn + 1]; pn2 = zq[2 * jn];
      }
      auto row_loop = [&](auto cref_tag) __attribute__((always_inline)) {
        constexpr bool CREF = decltype(cref_tag)::value;
        // (segment-A entries of a chunk are STORED at the top of the next turn, before that turn's requests: the memory counter counts in
        // order and the compiler cannot count a store behind a branch, so a store at the end of the turn made the wait for the records
        // requested at its top -- due at the start of the next turn -- a wait for the store's own round trip as well.  Issued first, it has
        // the whole turn.)
        for (int base = 0; base < nl; base += 64) {
          const int lt = lt_n;
          const int jt = jt_n;
          const double px = pn0, py = pn1, pz = pn2;
          if (posA_prev >= 0) row[posA_prev] = entA_prev;
          posA_prev = -1;
          {
            lt_n = list_at(base + 64 + lane);
            jt_n = gjr[lt_n & 0xFFF];   // (past the end of the list: the word behind the table -- the next tile's, or the buffer's slack; never used)
            const size_t jn = (size_t)(jt_n & MD_JMASK);
            pn0 = xq[2 * jn]; pn1 = xq[2 * jn + 1]; pn2 = zq[2 * jn];
          }
          // Branch-free test of the candidate against the four atoms (lanes past the end of the list are masked out); only
          // candidates inside the exclusion gate -- bonded neighbours, a few chunks per row -- take the wave-uniform slow path
          // that walks the exclusion lists.
          const bool in = base + lane < nl;
          const int code = (jt >> 23) & 31;
          const double xj = px + s_shift[3 * code], yj = py + s_shift[3 * code + 1], zj = pz + s_shift[3 * code + 2];
          int mask = 0, refm = 0;   // refm: the accepted pairs that the reference's list radius would hold too
          double r2a[NI];
          // (pad atoms of the cluster sit beyond 1e15, each pad slot at its own place: never inside the list radius of anything; the
          // nearest of the four distances stands for the nearest ACCEPTED one: beyond the list radius it decides nothing, and otherwise it
          // can only be too small, which moves the entry to a nearer segment -- always allowed)
#pragma unroll
          for (int a = 0; a < NI; a++) {
            const double dx = ci.x[a] - xj, dy = ci.y[a] - yj, dz = ci.z[a] - zj;
            const double r2 = dx * dx + dy * dy + dz * dz;
            mask |= (r2 < S.rlist2) ? (1 << a) : 0;
            if (CREF) refm |= (r2 < S.rlist_ref2) ? (1 << a) : 0;
            r2a[a] = r2;
          }
          const double rmin = vmin_f64(vmin_f64(r2a[0], r2a[1]), vmin_f64(r2a[2], r2a[3]));
          if (!in) { mask = 0; refm = 0; }
#define NB_R2X(a) r2a[a]
          NB_CHUNK_TAIL(rmin, NB_R2X, ra2, rb2, rc2, excl2)
          if (CREF) npairs_ref += __popc(refm);
        }
      };
      // (the second count has its own copy of the loop, so that the builds without it do not pay for four compares per candidate that
      // the compiler would otherwise keep as predicated code)
      if (count_ref) row_loop(std::true_type{}); else row_loop(std::false_type{});
    } else {
      // ---- the row loop of every other build: FP32, records from LDS, a superset of the list radius by eps ----
      // the cluster's atoms relative to the tile's origin, in scalar registers; pads far on the other side from every record
      float cx[NI], cy[NI], cz[NI];
#pragma unroll
      for (int a = 0; a < NI; a++) {
        const bool real = ci.atom[a] >= 0;
        cx[a] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(real ? (float)(ci.x[a] - ox) : -NB_FAR)));
        cy[a] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(real ? (float)(ci.y[a] - oy) : -NB_FAR)));
        cz[a] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(real ? (float)(ci.z[a] - oz) : -NB_FAR)));
      }
      // list entries two chunks ahead, record and table entry one chunk ahead (a record's address needs its list entry: that is there
      // when the requests of the next chunk go out)
      int lt_c = list_at(lane), lt_n = list_at(64 + lane);
      float x_c = s_rx[lt_c & 0xFFF], y_c = s_ry[lt_c & 0xFFF], z_c = s_rz[lt_c & 0xFFF];
      int jt_c = gjr[lt_c & 0xFFF];
      for (int base = 0; base < nl; base += 64) {
        const int lt = lt_c;
        const int jt = jt_c;
        const float xf = x_c, yf = y_c, zf = z_c;
        if (posA_prev >= 0) row[posA_prev] = entA_prev;
        posA_prev = -1;
        lt_c = lt_n;
        x_c = s_rx[lt_c & 0xFFF]; y_c = s_ry[lt_c & 0xFFF]; z_c = s_rz[lt_c & 0xFFF];
        jt_c = gjr[lt_c & 0xFFF];
        lt_n = list_at(base + 128 + lane);
        int mask = 0, refm = 0;   // (refm: the exact loop's second count; dead here)
        (void)refm;
        float r2a[NI];
#pragma unroll
        for (int a = 0; a < NI; a++) {
          const float dx = cx[a] - xf, dy = cy[a] - yf, dz = cz[a] - zf;
          const float r2 = dx * dx + dy * dy + dz * dz;
          mask |= (r2 < rl2e) ? (1 << a) : 0;
          r2a[a] = r2;
        }
        const float rmin = fminf(fminf(r2a[0], r2a[1]), fminf(r2a[2], r2a[3]));
        NB_CHUNK_TAIL(rmin, NB_R2X, ra2e, rb2e, rc2e, excl2e)
      }
    }
#undef NB_R2X
    if (posA_prev >= 0) row[posA_prev] = entA_prev;
    const int n = nA + nB + nC + nD;
    const bool bad = nB + nC > capBC || nD > capD || n > maxrow;
    if (bad) over = 1;
    // B, C1, C2: staging list -> behind A, with the table index and the type of j back in place (same-wave LDS traffic is processed in order)
    if (!bad) {
      auto expand = [&](int e16) -> int {
        const int lt = list_at(e16 & 0xFFF), l = lt & 0xFFF;
        const int ty = qall ? (int)((unsigned)gjr[l] >> 28) : (lt >> 12);
        return l | (ty << E_TYPE_SHIFT) | ((e16 >> 12) << E_MASK_SHIFT);
      };
      for (int k = lane; k < nB; k += 64) row[nA + k] = expand(lb[k]);
      for (int k = lane; k < nC; k += 64) row[nA + nB + k] = expand(lb[capBC - 1 - k]);
      for (int k = lane; k < nD; k += 64) row[nA + nB + nC + k] = expand(lb[capBC + k]);
      // the row's last chunk is filled up with empty entries (mask 0): k_pair reads whole chunks and masks no lane.  (maxrow is a
      // multiple of 64.)  On steps without the far band it reads up to the end of the chunk that holds the last C1 entry: what
      // follows there are C2 entries, whose pairs are outside the cutoff on such a step -- evaluated to nothing in lanes that
      // would otherwise idle
      if (n + lane < ((n + 63) & ~63)) row[n + lane] = 0;
    }
    if (lane == 0) {
      S.numneigh[2 * cl] = bad ? 0 : nA + nB + nC; S.numneigh[2 * cl + 1] = bad ? 0 : nD;
    }
    nmax = max(nmax, n);
    nrowent += n;
  }
#undef NB_CHUNK_TAIL
#ifdef PAIR_TIMING
  const unsigned long long tb2 = __builtin_readcyclecounter();
#endif
  // Schedule of k_pair, fixed here: the tile's rows are dealt round robin to its TW waves.  tile_order holds the tile's clusters
  // grouped by wave, tile_wstart the TW+1 group boundaries; an entry is (cluster | a << 20 | b << 25) and stands for the chunks
  // [C a / 16, C b / 16) of the cluster's row, (0, 16) = the whole row.  The deal does not look at the rows, so nothing waits for
  // them: a wave that has written its rows is done.  (Measured against longest-row-first list scheduling with and without splitting
  // long rows, which have to wait for the tile's slowest wave first: 398 against 395 and 396 evaluations/s, profiles/HISTORY.md.)
  if (wave == 0) {
    const int c0i = cs / NI, nclus = nown / NI;
    int *wst = S.tile_wstart + (size_t)cell * (TW + 1);
    for (int w = lane; w <= TW; w += 64) {
      int st = 0;
      for (int u = 0; u < w; u++) st += (nclus - u + TW - 1) / TW;
      wst[w] = st;
    }
    for (int i = lane; i < nclus; i += 64) {
      const int w = i % TW;
      int st = 0;
      for (int u = 0; u < w; u++) st += (nclus - u + TW - 1) / TW;
      S.tile_order[2 * c0i + st + i / TW] = (c0i + i) | (16 << 25);
    }
  }
#ifdef PAIR_TIMING
  if (lane == 0) {
    const unsigned long long tb3 = __builtin_readcyclecounter();
    atomicAdd(&sc.dbg[5], tb1 - tb0); atomicAdd(&sc.dbg[6], tb2 - tb1); atomicAdd(&sc.dbg[7], tb3 - tb2);
    atomicAdd(&sc.dbg[8], tb05 - tb0); atomicAdd(&sc.dbg[9], 1ull);
  }
#endif
  // Statistics: the pairs listed (what a full per-atom list would store: every unordered pair from both ends) -- at an exact build
  // the pairs inside the list radius, the count the tests compare with the oracle's; an FP32 build also counts the few pairs of its
  // eps band (1e-5 of the list) -- and, for a list wider than the reference's, the pairs inside the reference's radius, taken at the
  // first build of a run.
  {
    const double cnt = wave_sum((double)npairs), cnt_ref = count_ref ? wave_sum((double)npairs_ref) : (wider ? 0.0 : cnt);
    if (lane == 0) {
      atomicAdd(&sc.nentries, 2ull * (unsigned long long)cnt);
      atomicAdd(&sc.nentries_ref, 2ull * (unsigned long long)cnt_ref);
    }
  }
  if (lane == 0) {
    if (over) atomicOr(&sc.overflow, 1 | 8);   // 8: a cluster row (or its segment-B list)
    atomicMax(&sc.maxneigh_seen, nmax);
    atomicAdd(&sc.nrowent, nrowent);
  }
}


// ------------------------------------------------------------------------------------------
// k_pair
// ------------------------------------------------------------------------------------------
// NP = number of polynomial coefficients kept in scalar registers (>= fitted degree+1, 0-padded)
extern __shared__ double s_pair[];  // [capj][3] reaction-force accumulators, then int [capj] j table

// CLE: the coulomb cutoff does not exceed the LJ cutoff (the reference's 9 / 12): every interacting lane has an LJ term, which then
// defines the force factor without a zero to start from, and the cutoff test is one instead of two
template <bool VIR, bool ENG, int NP, bool CLE = false>
__global__ __launch_bounds__(TT, 4) void k_pair(const SimDev *__restrict__ sims, int ntiles, int nsims, int capj) {
  int sim, cell;
  if (!xcd_map(ntiles, nsims, sim, cell)) return;
  const SimDev &S = sims[sim];
  if (cell >= S.ncells) return;
  const int cs = S.cell_start[cell], ce = S.cell_start[cell + 1];
  if (ce == cs) {   // empty cell: its virial partials are still read by k_ewald_force
    if (VIR && !ENG && threadIdx.x < TW * 6) S.virp[(size_t)cell * TW * 6 + threadIdx.x] = 0.0;
    return;
  }
  SimScalars &sc = *S.sc;
#ifdef PAIR_TIMING
  const unsigned long long tm0 = __builtin_readcyclecounter();
#endif
  __shared__ double s_shift[27 * 4];
  __shared__ __attribute__((aligned(16))) double s_lj[2 * MD_MAXTYPES * MD_MAXTYPES];
  __shared__ double s_red[8 * TW];
  double *s_f = s_pair;   // [capj][3]: the three components of an entry side by side -- one address per entry for the three LDS atomics of a chunk
  int *s_jtab = (int *)(s_pair + 3 * (size_t)capj);
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
  // (lj1, lj2) of a type pair side by side: one 16-byte LDS read per LJ evaluation
  for (int k = threadIdx.x; k < 2 * nt2; k += TT) s_lj[k] = S.lj[(k & 1) * nt2 + (k >> 1)];
  const int nj = S.tile_nj[cell];
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int maxrow = S.maxneigh;
  // far skin band: walked only on steps where some atom of the replica has moved far enough for such a pair to reach the cutoff
  const int need_far = __builtin_amdgcn_readfirstlane(S.sc->need_far);
  // Row headers of this wave and the first two chunks of its entry stream are requested before the tile's table is
  // staged: none of that needs the LDS, so their latency runs under the table load and the barrier.
  const int p_begin = S.tile_wstart[(size_t)cell * (TW + 1) + wave], p_end = S.tile_wstart[(size_t)cell * (TW + 1) + wave + 1];
  // (wave-uniform by construction; saying so lets the row cursors below live in scalar registers and branch on the scalar unit)
  const int nrows = __builtin_amdgcn_readfirstlane((nj > 0) ? min(p_end - p_begin, 64) : 0);
  if (p_end - p_begin > 64 && lane == 0) atomicOr(&sc.overflow, 1 | 4);   // cannot happen after k_neigh_build's check; loud if it ever does
  int h_cl = 0, h_nn = 0;
  if (lane < nrows) {
    // an entry of the schedule: cluster | a << 20 | b << 25 = the chunks [C a / 16, C b / 16) of that cluster's row (k_neigh_build)
    const int ent = S.tile_order[2 * (cs / NI) + p_begin + lane];
    h_cl = ent & 0xFFFFF;
    const int n = S.numneigh[2 * h_cl] + (need_far ? S.numneigh[2 * h_cl + 1] : 0);   // [A|B|C1], then C2
    const int C = (n + 63) >> 6, pa = (ent >> 20) & 31, pb = (ent >> 25) & 31;
    const int kb = 64 * ((C * pa) >> 4), ke = 64 * ((C * pb) >> 4);   // whole chunks: the row's last one is padded with empty entries
    h_nn = (max(ke, kb) << 16) | kb;
  }
#define H_KB(v) ((v) & 0xFFFF)
#define H_KE(v) ((int)((unsigned)(v) >> 16))
  const GLOBAL_AS int *neigh = as_global(S.neigh);
  // prefetch cursor: the chunk two ahead of the one being evaluated
  int pr = 0;
  int pcl = __builtin_amdgcn_readlane(h_cl, 0), pnn = __builtin_amdgcn_readlane(h_nn, 0);
  int pk = H_KB(pnn);
  pnn = H_KE(pnn);
  const unsigned lane4 = 4u * (unsigned)lane;
  auto fetch = [&]() -> int {
    int v = 0;
    if (pr < nrows) {
      // the row is contiguous ([A|B|C1|C2], k_neigh_build): scalar row base + a 32-bit byte offset per lane
      const GLOBAL_AS char *row = (const GLOBAL_AS char *)(neigh + (size_t)pcl * maxrow);
      // (an EMPTY row -- a cluster without a listed neighbour: a lone molecule in a large box -- still takes one turn of the stream; its
      // memory holds whatever an earlier build left there, so it reads nothing.  The test is on the scalar unit.)
      if (pk < pnn) v = *(const GLOBAL_AS int *)(row + (4u * (unsigned)pk + lane4));
      pk += 64;
      if (pk >= pnn) {
        pr += 1;
        const int q = min(pr, nrows - 1);
        pcl = __builtin_amdgcn_readlane(h_cl, q); pnn = __builtin_amdgcn_readlane(h_nn, q);
        pk = H_KB(pnn);
        pnn = H_KE(pnn);
      }
    }
    return v;
  };
  int e_n = 0, e_nn = 0;
  if (nrows > 0) { e_n = fetch(); e_nn = fetch(); }
  {
    const GLOBAL_AS int *gj = as_global(S.tile_jtab) + (size_t)cell * S.capj;
    for (int l = threadIdx.x; l < nj; l += TT) {
      s_jtab[l] = gj[l];
      s_f[3 * l] = 0.0; s_f[3 * l + 1] = 0.0; s_f[3 * l + 2] = 0.0;
    }
  }
  double cp[NP];
#pragma unroll
  for (int m = 0; m < NP; m++) cp[m] = S.coul_poly_g[m];   // g H(u), scaled on the host: x H = r (g H), one multiplication less per coulomb pair
  double cp_top = cp[NP - 1];   // the leading coefficient in a vector register: the first Horner step then needs no move (one scalar operand per instruction)
  asm volatile("" : "+v"(cp_top));
  __syncthreads();
  const GLOBAL_AS double *xq = as_global((const double *)S.xq);   // (x,y) halves
  const GLOBAL_AS double *zq = xq + 2 * (size_t)S.npad;              // (z,q) halves
  const double g = S.g_ewald, g2u = g * g * S.coul_uscale;
  const double cutc2 = S.cut_coul2, cutl2 = S.cut_lj2;
  const double cutmax2 = fmax(cutc2, cutl2);
  double vl[6] = {0, 0, 0, 0, 0, 0}, vc[6] = {0, 0, 0, 0, 0, 0};
  double elj = 0, ecoul = 0;
#ifdef PAIR_TIMING
  const unsigned long long tm1 = __builtin_readcyclecounter();
#endif
  // This wave's rows, fixed at build time (longest first), run as ONE stream of 64-entry chunks: all rows of the tile
  // index the same LDS j table, so the prefetch pipeline (row entries two chunks ahead, table entry + record one chunk
  // ahead) runs straight across row boundaries; a boundary only swaps the i-cluster (reduce its forces, load the next
  // cluster's records).  Row headers (cluster, counts) sit in one VGPR triple, lane r = r-th row of the wave, and are
  // read with v_readlane: no memory latency on the row switch.
#ifdef PAIR_COUNT
  // diagnostic build (make HIPFLAGS+=-DPAIR_COUNT): where the lanes of the row loop are -- per wave-chunk and per atom block, how many
  // lanes reach the distance test, the LJ block and the coulomb block (SimScalars::dbg, printed with SCEMA_MD_TIMING=1)
  unsigned pc_chunk = 0, pc_chunk_nz = 0, pc_blk = 0, pc_dist = 0, pc_lj = 0, pc_ljblk = 0, pc_coul = 0, pc_coulblk = 0;
#endif
  if (nrows > 0) {
    // evaluation cursor
    int r = 0;
    int s0 = __builtin_amdgcn_readlane(h_cl, 0) * NI, nn = __builtin_amdgcn_readlane(h_nn, 0);
    int k0 = H_KB(nn);
    nn = H_KE(nn);
    double xi[NI], yi[NI], zi[NI], qi[NI], fx[NI], fy[NI], fz[NI];
    int ti[NI];
#pragma unroll
    for (int a = 0; a < NI; a++) {
      xi[a] = XQ_X(S, s0 + a); yi[a] = XQ_Y(S, s0 + a); zi[a] = XQ_Z(S, s0 + a);
      qi[a] = MD_QQRD2E * XQ_Q(S, s0 + a);
      ti[a] = S.stype[s0 + a] * nt;
      fx[a] = fy[a] = fz[a] = 0.0;
    }
    // One chunk: the entries `e` (one per lane), their table entries `jt` and records (xj, yj, zj, qj) against the wave's current
    // i-cluster; then the row cursor moves on.
    auto chunk = [&](const int e, const int jt, const double xj, const double yj, const double zj, const double qj) __attribute__((always_inline)) {
      const int mask = (e >> E_MASK_SHIFT) & 0xF;  // 0 for the padding of a row's last chunk
#ifdef PAIR_COUNT
      if (lane == 0) { pc_chunk += 1; pc_chunk_nz += (__ballot(mask != 0) != 0ull) ? 1 : 0; }
#endif
      if (mask != 0) {
        const int cs4 = (int)(((unsigned)jt >> 21) & 0x7Cu);   // 4 * image code (bits 23..27; the type bits above them are cleared in this copy)
        const double xs = xj + s_shift[cs4], ys = yj + s_shift[cs4 + 1], zs = zj + s_shift[cs4 + 2];
        const int tj = (e >> E_TYPE_SHIFT) & 0xF;
        double gx = 0.0, gy = 0.0, gz = 0.0;   // reaction force on j
        asm volatile("" : "+v"(gx), "+v"(gy), "+v"(gz));   // one set of zeros here instead of one per nested branch of the first atom
#pragma unroll
        for (int a = 0; a < NI; a++) {
          if (!(mask & (1 << a))) continue;
          const double dx = xi[a] - xs, dy = yi[a] - ys, dz = zi[a] - zs;
#if defined(PAIR_WHATIF_FMA)    // sensitivity experiment (never in a product build): eight more dependent FP64 FMAs per distance block
          double rsq = dx * dx + dy * dy + dz * dz;
          { double t_ = rsq; _Pragma("unroll") for (int q_ = 0; q_ < 8; q_++) t_ = fma(t_, 1.0e-300, rsq); asm volatile("" : "+v"(t_)); rsq = t_; }
#elif defined(PAIR_WHATIF_FMA_ILP)   // ... the same eight FMAs as four independent chains of two
          double rsq = dx * dx + dy * dy + dz * dz;
          { double t0_ = rsq, t1_ = dx, t2_ = dy, t3_ = dz;
            _Pragma("unroll") for (int q_ = 0; q_ < 2; q_++) { t0_ = fma(t0_, 1.0e-300, rsq); t1_ = fma(t1_, 1.0e-300, rsq); t2_ = fma(t2_, 1.0e-300, rsq); t3_ = fma(t3_, 1.0e-300, rsq); }
            asm volatile("" : "+v"(t0_), "+v"(t1_), "+v"(t2_), "+v"(t3_)); rsq = t0_; }
#elif defined(PAIR_WHATIF_INT)  // ... or eight more dependent 32-bit integer operations
          double rsq = dx * dx + dy * dy + dz * dz;
          { int t_ = __double2loint(rsq); _Pragma("unroll") for (int q_ = 0; q_ < 8; q_++) { t_ = (t_ ^ 0x5bd1e995) + q_; asm volatile("" : "+v"(t_)); }
            int lo_ = __double2loint(rsq); asm volatile("" : "+v"(lo_) : "v"(t_)); rsq = __hiloint2double(__double2hiint(rsq), lo_); }   // (a dependency, not a change)
#else
          const double rsq = dx * dx + dy * dy + dz * dz;
#endif
#ifdef PAIR_COUNT
          { const unsigned long long b = __ballot(true); pc_dist += 1; if (lane == __ffsll((long long)b) - 1) pc_blk += 1; }
          if (rsq < cutl2) { const unsigned long long b = __ballot(true); pc_lj += 1; if (lane == __ffsll((long long)b) - 1) pc_ljblk += 1; }
          if (rsq < cutc2) { const unsigned long long b = __ballot(true); pc_coul += 1; if (lane == __ffsll((long long)b) - 1) pc_coulblk += 1; }
#endif
          if (CLE && !ENG) {
            if (rsq < cutl2) {
              // (1/r^2 from the reciprocal estimate + one Newton step in the blocks no lane of which is inside the coulomb cutoff -- 3 FP64
              // instructions instead of 7 in half of the LJ blocks -- was measured a third time in round 4: 8.34 / 8.37 against 8.32 / 8.26 ms)
              const double rinv = rsqrt_f64(rsq);
              const double r2inv = rinv * rinv;
              const double r6inv = r2inv * r2inv * r2inv;
              const double2 lj12 = ((const double2 *)s_lj)[ti[a] + tj];
              double fp = r6inv * (lj12.x * r6inv - lj12.y) * r2inv;
              if (rsq < cutc2) {
                // qq (1 - x H(u)) / r^3 with x H = r (g H) = r P and r / r = 1:  qq (1/r - P) / r^2
                const double t = fma(rsq, g2u, -1.0);
                double p = cp_top;
#pragma unroll
                for (int m = NP - 2; m >= 0; m--) p = fma(p, t, cp[m]);
                fp = fma(qi[a] * qj * (rinv - p), r2inv, fp);
              }
              const double tx = dx * fp, ty = dy * fp, tz = dz * fp;
              fx[a] += tx; fy[a] += ty; fz[a] += tz;
              gx -= tx; gy -= ty; gz -= tz;
            }
          } else if (rsq < cutmax2) {
            const double rinv = rsqrt_f64(rsq);
            const double r2inv = rinv * rinv;
            double fp = 0.0, flj = 0.0, fc = 0.0;
            if (rsq < cutc2) {
              // erfc(x) + 2x/sqrt(pi) exp(-x^2) = 1 - x H(u): Horner in t = u*uscale - 1
              const double rr = rsq * rinv;   // r
              const double t = fma(rsq, g2u, -1.0);
              double p = cp[NP - 1];
#pragma unroll
              for (int m = NP - 2; m >= 0; m--) p = fma(p, t, cp[m]);
              const double pref = qi[a] * qj * rinv;
              fp = pref * fma(-rr, p, 1.0) * r2inv;
              if (ENG) { fc = fp; ecoul += pref * erfc(g * rr); }
            }
            if (rsq < cutl2) {
              const double r6inv = r2inv * r2inv * r2inv;
              const double2 lj12 = ((const double2 *)s_lj)[ti[a] + tj];
              const double w = r6inv * (lj12.x * r6inv - lj12.y);
              fp = fma(w, r2inv, fp);
              if (ENG) {
                const int tt = ti[a] + tj;
                flj = w * r2inv;
                elj += r6inv * (S.lj[2 * nt2 + tt] * r6inv - S.lj[3 * nt2 + tt]);
              }
            }
            const double tx = dx * fp, ty = dy * fp, tz = dz * fp;
            fx[a] += tx; fy[a] += ty; fz[a] += tz;
            gx -= tx; gy -= ty; gz -= tz;
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
        const int l = e & E_LMASK;
        lds_add(&s_f[3 * l], gx); lds_add(&s_f[3 * l + 1], gy); lds_add(&s_f[3 * l + 2], gz);
#ifdef PAIR_WHATIF_ATOMICS   // sensitivity experiment (never in a product build): every LDS atomic of the row loop issued twice
        lds_add(&s_f[3 * l], 0.0); lds_add(&s_f[3 * l + 1], 0.0); lds_add(&s_f[3 * l + 2], 0.0);
#endif
      }
      k0 += 64;
      if (r < nrows && k0 >= nn) {
        // Row finished.  Forces on the cluster's own atoms: 12 per-lane partial sums -> the atoms' own table entries
        // (own cell first).  Transposing butterfly over the quad (lane i ends up with component c of atom i&3), then
        // a row scan: lanes 12..15 of each row of 16 hold the row totals and add them to LDS.  81 VALU instructions
        // and 3 LDS atomics per cluster instead of 144 ds_bpermute.
        const bool b0 = lane & 1, b1 = lane & 2;
        double u[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
          const double *f = (c == 0) ? fx : (c == 1) ? fy : fz;
          const double w0 = (b0 ? f[1] : f[0]) + dpp_mov<DPP_QUAD_XOR1>(b0 ? f[0] : f[1]);
          const double w1 = (b0 ? f[3] : f[2]) + dpp_mov<DPP_QUAD_XOR1>(b0 ? f[2] : f[3]);
          double t = (b1 ? w1 : w0) + dpp_mov<DPP_QUAD_XOR2>(b1 ? w0 : w1);
          t += dpp_mov<DPP_ROW_SHR4>(t);
          t += dpp_mov<DPP_ROW_SHR8>(t);
          u[c] = t;
        }
        if ((lane & 12) == 12) {
          const int l = s0 - cs + (lane & 3);
          lds_add(&s_f[3 * l], u[0]); lds_add(&s_f[3 * l + 1], u[1]); lds_add(&s_f[3 * l + 2], u[2]);
        }
        r += 1;
        if (r < nrows) {
          s0 = __builtin_amdgcn_readlane(h_cl, r) * NI; nn = __builtin_amdgcn_readlane(h_nn, r);
          k0 = H_KB(nn);
          nn = H_KE(nn);
#pragma unroll
          for (int a = 0; a < NI; a++) {
            xi[a] = XQ_X(S, s0 + a); yi[a] = XQ_Y(S, s0 + a); zi[a] = XQ_Z(S, s0 + a);
            qi[a] = MD_QQRD2E * XQ_Q(S, s0 + a);
            ti[a] = S.stype[s0 + a] * nt;
            fx[a] = fy[a] = fz[a] = 0.0;
          }
        }
      }
    };
    // The pipeline is three deep in the entries (evaluated | table entry read and record requested | row entry requested) and two deep
    // in the records.  Written as ONE loop body, every chunk paid seven register moves to shift the stages along (three entries, a
    // table entry, four 64-bit record words: 4 % of the kernel's vector instructions); the body below is two chunks with the roles
    // of the two record sets swapped, which leaves one exchange of the entry names per pair of chunks.
    struct Rec { double x, y, z, q; };
    auto deref = [&](const int e, int &jt, Rec &R) __attribute__((always_inline)) {
      jt = s_jtab[e & E_LMASK];
      const size_t j = (size_t)(jt & MD_JMASK);
      R.x = xq[2 * j]; R.y = xq[2 * j + 1]; R.z = zq[2 * j]; R.q = zq[2 * j + 1];
    };
    int eA = e_n, eB = e_nn, eC, jtX, jtY;
    Rec X, Y;
    deref(eA, jtX, X);
    // (one exit, at the end of the pair: with a second exit between the chunks the compiler moved the prefetch cursor into vector
    // registers; a wave whose stream ends after the first chunk of a pair runs the second one on empty entries -- mask 0, nothing
    // evaluated, and the row cursor stands still once the last row is done)
    do {
      deref(eB, jtY, Y);
      eC = fetch();
      chunk(eA, jtX, X.x, X.y, X.z, X.q);
      deref(eC, jtX, X);
      eA = fetch();
      chunk(eB, jtY, Y.x, Y.y, Y.z, Y.q);
      { const int t = eA; eA = eC; eB = t; }   // evaluated next: eC (its record is in X), then the entry just requested
    } while (r < nrows);
  }
#ifdef PAIR_COUNT
  {
    const double v[8] = {(double)pc_chunk, (double)pc_chunk_nz, (double)pc_blk, (double)pc_dist, (double)pc_lj, (double)pc_ljblk, (double)pc_coul, (double)pc_coulblk};
    for (int k = 0; k < 8; k++) {
      const double t = wave_sum(v[k]);
      if (lane == 0 && t != 0.0) atomicAdd(&sc.dbg[k], (unsigned long long)t);
    }
  }
#endif
#ifdef PAIR_TIMING
  const unsigned long long tm2 = __builtin_readcyclecounter();
#endif
  __syncthreads();
#ifdef PAIR_TIMING
  const unsigned long long tm3 = __builtin_readcyclecounter();
#endif
  // flush the tile's accumulators: consecutive table entries are runs of consecutive slots -> coalesced atomics.
  // Production virial (one lumped pair virial; the pressure sums all parts anyway): the tile's pairs contribute
  // sum_pairs (r_i - r_j) (x) F_ij = sum over table entries l of (x_slot(l) + shift_l) (x) F_l, F_l = force
  // accumulated on entry l.  Summed over all tiles the first part is sum_slots x_slot (x) fs_slot, which
  // k_ewald_force takes from the finished slot-ordered forces; the tile only adds the image-shift part of its
  // non-home entries (shift from LDS, no position gathers), one partial sum per wave, stored without atomics.
  if (VIR && !ENG) {
    for (int l = threadIdx.x; l < nj; l += TT) {
      const int code = (s_jtab[l] >> 23) & 31;   // (bits 28..31: the type of j, k_neigh_build's)
      if (code != CODE_HOME) {
        const double ax = s_f[3 * l], ay = s_f[3 * l + 1], az = s_f[3 * l + 2];
        const double px = s_shift[4 * code], py = s_shift[4 * code + 1], pz = s_shift[4 * code + 2];
        vl[0] = fma(px, ax, vl[0]); vl[1] = fma(py, ay, vl[1]); vl[2] = fma(pz, az, vl[2]);
        vl[3] = fma(px, ay, vl[3]); vl[4] = fma(px, az, vl[4]); vl[5] = fma(py, az, vl[5]);
      }
    }
  }
  {
    double *fs = S.fs;
    const size_t np = (size_t)S.npad;
    for (int l = threadIdx.x; l < nj; l += TT) {
      const double ax = s_f[3 * l], ay = s_f[3 * l + 1], az = s_f[3 * l + 2];
      if (ax != 0.0 || ay != 0.0 || az != 0.0) {
        const size_t slot = (size_t)(s_jtab[l] & MD_JMASK);
        atomicAdd(fs + slot, ax); atomicAdd(fs + np + slot, ay); atomicAdd(fs + 2 * np + slot, az);
      }
    }
  }
  if (VIR && !ENG) {
    double *vp = S.virp + ((size_t)cell * TW + wave) * 6;
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const double t = wave_sum(vl[k]);
      if (lane == 0) vp[k] = t;
    }
  }
#ifdef PAIR_TIMING
  {
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long tm4 = __builtin_readcyclecounter();
    if (lane == 0) {
      atomicAdd(&sc.dbg[0], tm1 - tm0); atomicAdd(&sc.dbg[1], tm2 - tm1); atomicAdd(&sc.dbg[2], tm3 - tm2);
      atomicAdd(&sc.dbg[3], tm4 - tm3); atomicAdd(&sc.dbg[4], 1ull);
    }
  }
#endif
  if (VIR && ENG) {
    tile_atomic_add<6>(vl, sc.vir + P_LJ * 6, s_red);
    tile_atomic_add<6>(vc, sc.vir + P_COUL * 6, s_red);
  }
  if (ENG) {
    double e1[1];
    e1[0] = elj;
    tile_atomic_add<1>(e1, sc.eng + P_LJ, s_red);
    e1[0] = ecoul;
    tile_atomic_add<1>(e1, sc.eng + P_COUL, s_red);
  }
}

static inline dim3 grid_xcd(int ntiles, int ns) { return dim3((unsigned)(ns * ntiles), 1, 1); }

size_t mdk_pair_lds_bytes(int capj) { return (size_t)capj * (3 * sizeof(double) + sizeof(int)); }
int mdk_neigh_capB(int maxrow) { return (int)(0.6 * maxrow) / 64 * 64 + 64; }
// capacity of one group's list (16-bit table indices): 3/4 of the table (a quarter of PE-10k's clusters reaches 72 %); a group that reaches more walks the whole table instead.  (The kernel's LDS must stay below 80 KB for two workgroups per CU: at 83 KB it ran 1.75 times longer.)
static int neigh_qcap(int capj) {
  static const int n16 = scema_env("SCEMA_MD_QCAP16") ? atoi(scema_env("SCEMA_MD_QCAP16")) : 12;   // (test switch: small values force the whole-table path)
  return (n16 * capj / 16 + 63) / 64 * 64;
}
size_t mdk_neigh_lds_bytes(int capj, int maxrow) {
  if (capj > 4032) return (size_t)1 << 30;   // group-list and staging entries hold a 12-bit table index (k_pair's own LDS bound keeps tables below 2 707 entries)
  return 3 * ((size_t)capj + NB_RECPAD) * sizeof(float) + ((size_t)TW * mdk_neigh_capB(maxrow) + (size_t)NQ * neigh_qcap(capj)) * sizeof(unsigned short);
}
// which builds test their candidates in FP64 at the exact list radius: -1 (default) the first build of a run, 1 all, 0 none
static int neigh_exact_mode() {
  static const int m = scema_env("SCEMA_MD_NEIGH_EXACT") ? atoi(scema_env("SCEMA_MD_NEIGH_EXACT")) : -1;
  return m;
}

void mdk_neigh_build(hipStream_t st, const SimDev *d, int ns, int maxcells, int maxrow, int capj) {
  // per-wave LDS list of segment B: well over its expected share (~40 %) of a full row
  const int capB = mdk_neigh_capB(maxrow);
  const size_t lds = mdk_neigh_lds_bytes(capj, maxrow);
  static size_t optin_tab[16] = {0};  // more than 64 KB of dynamic LDS needs an explicit opt-in
  size_t &optin = lds_optin_slot(optin_tab);
  if (lds > 64 * 1024 && lds > optin) { (void)hipFuncSetAttribute((const void *)k_neigh_build, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); optin = lds; }
  hipLaunchKernelGGL(k_neigh_build, grid_xcd(maxcells, ns), dim3(TT), lds, st, d, maxcells, ns, capj, capB, neigh_qcap(capj), neigh_exact_mode());
}

template <bool VIR, bool ENG, int NP, bool CLE = false>
static void launch_pair_v(hipStream_t st, const SimDev *d, int ns, int ntiles, int capj) {
  const size_t lds = mdk_pair_lds_bytes(capj);
  static size_t optin_tab[16] = {0};
  size_t &optin = lds_optin_slot(optin_tab);
  if (lds > 48 * 1024 && lds > optin) { (void)hipFuncSetAttribute((const void *)k_pair<VIR, ENG, NP, CLE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); optin = lds; }
  hipLaunchKernelGGL((k_pair<VIR, ENG, NP, CLE>), grid_xcd(ntiles, ns), dim3(TT), lds, st, d, ntiles, ns, capj);
}

template <int NP>
static void launch_pair(hipStream_t st, const SimDev *d, int ns, int ntiles, int capj, int vir, int eng, int cle) {
  if (eng) launch_pair_v<true, true, NP>(st, d, ns, ntiles, capj);
  else if (vir) { if (cle) launch_pair_v<true, false, NP, true>(st, d, ns, ntiles, capj); else launch_pair_v<true, false, NP>(st, d, ns, ntiles, capj); }
  else { if (cle) launch_pair_v<false, false, NP, true>(st, d, ns, ntiles, capj); else launch_pair_v<false, false, NP>(st, d, ns, ntiles, capj); }
}

void mdk_pair(hipStream_t st, const SimDev *d, int ns, int maxcells, int capj, int vir, int eng, int npoly, int cle) {
  if (npoly <= 6) launch_pair<6>(st, d, ns, maxcells, capj, vir, eng, cle);
  else if (npoly <= 8) launch_pair<8>(st, d, ns, maxcells, capj, vir, eng, cle);
  else if (npoly <= 10) launch_pair<10>(st, d, ns, maxcells, capj, vir, eng, cle);
  else if (npoly <= 12) launch_pair<12>(st, d, ns, maxcells, capj, vir, eng, cle);
  else if (npoly <= 14) launch_pair<14>(st, d, ns, maxcells, capj, vir, eng, cle);
  else if (npoly <= 15) launch_pair<15>(st, d, ns, maxcells, capj, vir, eng, cle);
  else if (npoly <= 16) launch_pair<16>(st, d, ns, maxcells, capj, vir, eng, cle);
  else if (npoly <= 18) launch_pair<18>(st, d, ns, maxcells, capj, vir, eng, cle);
  else if (npoly <= 20) launch_pair<20>(st, d, ns, maxcells, capj, vir, eng, cle);
  else if (npoly <= 24) launch_pair<24>(st, d, ns, maxcells, capj, vir, eng, cle);
  else if (npoly <= 32) launch_pair<32>(st, d, ns, maxcells, capj, vir, eng, cle);
  else launch_pair<MD_MAXPOLY>(st, d, ns, maxcells, capj, vir, eng, cle);
}
