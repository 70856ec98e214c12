// md_neigh.hip -- the neighbour-list build of the pair kernel: cell tiles -> j tables -> i-cluster rows (layouts: md_pair.hip's header comment).
// A translation unit of its own since round 5 (md_pair.hip had grown to 1 200 lines; the two kernels share md_pair_dev.h).
//
// What bounds the kernel (round 5): vector-instruction ISSUE.  Every vector instruction of a wave costs its SIMD four
// cycles whatever its width -- FP64, FP32 and 32-bit integer alike, packed FP32 (v_pk_*) twice that -- and the SIMDs issue during
// 76 % of the kernel's time (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x 4 waves).  What-if builds: the FP64 distance arithmetic of the row
// loop issued twice, +32 instructions per chunk, +12 % time; the same arithmetic in FP32, -0 %.  So the row loop is as fast as its
// instruction COUNT (~125 vector instructions per 64-candidate chunk, 26 chunks per row), not as its arithmetic type.
//
// Reference semantics: neighbor 2.0 bin, neigh_modify every 1 delay 5 check yes (in.set.lammps:27,32).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "md_env.h"
#include "md_pair_dev.h"

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
#define NB_LISTPAD 256   // dummy entries behind a group list (the fast loop reads up to 192 + 63 entries past the end)
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
      // a candidate enters the table if some GROUP can reach it (the groups' boxes lie inside the cell's: a test against the cell's box would
      // only admit candidates that no cluster can list); the own cell's slots all enter (table index l <-> slot cs + l)
      okq[i] = 0;
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        const float *bq = s_qboxf[q];
        const float ex = fmaxf(0.f, fmaxf(bq[0] - xf[i], xf[i] - bq[3])), ey = fmaxf(0.f, fmaxf(bq[1] - yf[i], yf[i] - bq[4])), ez = fmaxf(0.f, fmaxf(bq[2] - zf[i], zf[i] - bq[5]));
        okq[i] |= (valid[i] && ex * ex + ey * ey + ez * ez < rl2e) ? (1u << q) : 0u;
      }
      ok[i] = valid[i] && (home[i] || okq[i] != 0u);
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
  // A group list is followed by NB_LISTPAD dummy entries (the record behind the table): the row loops read their list entries two chunks
  // ahead without a bounds test.  A list that has no room for them counts as overflowed (the group walks the whole table).
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    const bool fits = qn[q] + NB_LISTPAD <= qcap;
    if ((int)threadIdx.x == q) s_qn[q] = fits ? qn[q] : -1;
    if (fits && threadIdx.x < NB_LISTPAD) s_qlist[q * qcap + qn[q] + threadIdx.x] = (unsigned short)capjs;
  }
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
#ifdef PAIR_TIMING
    const unsigned long long tr0 = __builtin_readcyclecounter();
    unsigned nslow = 0, nownc = 0;
#endif
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
    int exv;   // lane 16 a + k: the slot of partner k of atom a (as slots: what the table entries name), or -1
    {
      const int a = lane >> 4, e = lane & 15;
      const int na = (a == 0) ? exn[0] : (a == 1) ? exn[1] : (a == 2) ? exn[2] : exn[3];
      const int ba = (a == 0) ? exb[0] : (a == 1) ? exb[1] : (a == 2) ? exb[2] : exb[3];
      exv = (e < na) ? S.slot_of[S.ex_list[ba + e]] : -1;
    }
    const unsigned long long exvalid = __ballot(exv >= 0);
    const bool exlong = max(max(exn[0], exn[1]), max(exn[2], exn[3])) > 16;   // (uniform)
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
    auto list_at = [&](int k) -> int { return qall ? ((k < nl) ? k : capjs) : (int)ql[k]; };
#ifdef PAIR_TIMING
    if (lane == 0) { atomicAdd(&sc.dbg[10], (unsigned long long)((nl + 63) >> 6)); atomicAdd(&sc.dbg[11], 1ull); }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long tr1 = __builtin_readcyclecounter();
#define NB_T_SLOW() nslow++
#define NB_T_OWN() nownc++
#else
#define NB_T_SLOW()
#define NB_T_OWN()
#endif
    bool own_chunks = true;
    // Segment-A entries go straight into the row, and EVERY lane stores in EVERY turn -- a lane without such an entry into the last 64
    // words of the row's capacity, which no row may reach (`bad` below).  The vector-memory counter counts loads and stores together, in
    // issue order; a store that only some turns issue (a branch around it) cannot be counted by the compiler, which then waits for
    // "all but the newest load" -- and with that for the store issued fifty instructions earlier, a round trip to memory in every turn
    // (2 100 cycles per chunk on the wave clocks of round 5, whatever the arithmetic cost).  A store that always issues is counted, and
    // the wait for the table entry requested a turn ago leaves it in flight.
    const int dump = maxrow - 64 + lane;
    int posA_prev = dump, entA_prev = 0;
    // What both row loops do with a chunk once the i-mask of every candidate is known: own-cell rule, exclusions, segments, stores.
    // (lt: list entry; jt: table entry)
#define NB_CHUNK_TAIL(RMIN, RA, RB, RC, EXCL)                                                                                            \
      const int l = lt & 0xFFF;                                                                                                           \
      const int j = jt & MD_JMASK;                                                                                                        \
      if (own_chunks) {   /* (wave-uniform: the entries of the own cell come first in the table and in every list) */                     \
        NB_T_OWN();                                                                                                                       \
        /* same cell, same image: each pair once, by slot order -- atom a of the cluster keeps j only if j > s0slot + a.  One mask per */ \
        /* candidate instead of a test per atom (rmin may then be too small: a nearer segment is always allowed) */                       \
        const bool own = l < nown;                                                                                                        \
        const int d = j - s0slot;                                                                                                         \
        const int drop = !own ? 0 : (d <= 0 ? 0xF : (d > 3 ? 0 : (0xF << d) & 0xF));                                                      \
        mask &= ~drop; refm &= ~drop;                                                                                                     \
        own_chunks = __ballot(own) != 0ull;                                                                                               \
      }                                                                                                                                   \
      /* Chunks with a candidate inside the exclusion gate (bonded neighbours: 5 of a PE-10k row's 26) take the wave-uniform path that */ \
      /* strikes the excluded pairs out: every partner of the cluster's atoms -- lane 16 a + k of exv holds the slot of partner k of */   \
      /* atom a -- is read into a scalar register and compared with the chunk's 64 candidates at once (3 vector instructions per */      \
      /* partner, ~30 partners per cluster).  The candidates used to walk the four lists themselves, a dependent LDS read per */         \
      /* partner and atom: 8 000 cycles per such chunk, two thirds of the row loop.  No distance test is needed: the box is at */         \
      /* least two list radii wide, so of all images of a partner only the bonded one can have its bit set. */                           \
      if (__ballot(mask != 0 && RMIN < EXCL) != 0ull) {                                                                                   \
        NB_T_SLOW();                                                                                                                      \
        unsigned long long todo = exvalid;                                                                                                \
        while (todo) {                                                                                                                    \
          const int b = __builtin_ctzll(todo);                                                                                            \
          todo &= todo - 1;                                                                                                               \
          const int sp = __builtin_amdgcn_readlane(exv, b);                                                                               \
          const int bit = 1 << (b >> 4);                                                                                                  \
          if (j == sp) { mask &= ~bit; refm &= ~bit; }   /* rmin may stay too small: only the segment choice sees it */                   \
        }                                                                                                                                 \
        if (exlong) {   /* partners beyond the first 16 of an atom (none in the reference's force fields): the lists in memory */         \
          _Pragma("unroll") for (int a = 0; a < NI; a++)                                                                                  \
            for (int e = 16; e < exn[a]; e++)                                                                                             \
              if (S.slot_of[S.ex_list[exb[a] + e]] == j) { mask &= ~(1 << a); refm &= ~(1 << a); }                                        \
        }                                                                                                                                 \
      }                                                                                                                                   \
      /* Segments: A straight into the row (the store itself at the top of the next turn); B, C1, C2 into the wave's staging list with ONE store. */ \
      /* Lanes without an entry get a distance beyond every threshold, so that the four class masks are four plain compares (a ballot of a */      \
      /* composed condition costs two more vector instructions each), and the staging index is selected, clamped into the list (an index that */  \
      /* leaves its region means an overflow, which `bad` reports below) and used by one predicated store instead of three. */                    \
      {                                                                                                                                   \
        typedef decltype(RMIN) nb_rm_t;                                                                                                   \
        const nb_rm_t rmx = mask ? RMIN : (nb_rm_t)3.0e38f;                                                                               \
        const bool inA = rmx < RA, inAB = rmx < RB, inABC = rmx < RC, inAll = mask != 0;                                                  \
        const unsigned long long mA = __ballot(inA), mAB = __ballot(inAB), mABC = __ballot(inABC), mAll = __ballot(inAll);                \
        const unsigned long long mB = mAB & ~mA, mC = mABC & ~mAB, mD = mAll & ~mABC;                                                     \
        const int pB = nB + popc_below(mB), pC = capBC - 1 - (nC + popc_below(mC)), pD = capBC + nD + popc_below(mD);                     \
        const int idx = min(max(inAB ? pB : (inABC ? pC : pD), 0), capB - 1);                                                             \
        if (inAll && !inA) lb[idx] = (unsigned short)((base + lane) | (mask << 12));                                                      \
        if (inA) {                                                                                                                        \
          const int pos = nA + popc_below(mA);                                                                                            \
          const int ty = qall ? (int)((unsigned)jt >> 28) : (lt >> 12);                                                                   \
          if (pos < maxrow - 64) { posA_prev = pos; entA_prev = l | (ty << E_TYPE_SHIFT) | (mask << E_MASK_SHIFT); }                      \
        }                                                                                                                                 \
        nA += __popcll(mA); nB += __popcll(mB); nC += __popcll(mC); nD += __popcll(mD);                                                   \
      }                                                                                                                                   \
      npairs += __popc(mask);

    if (exact) {
      // ---- the exact row loop (first build of a run): FP64 distances at the exact radius, positions gathered from memory ----
      // one chunk of the list ahead: entry + record of chunk r+1 are in flight while chunk r is tested
      // (lanes past the end of the list name slot 0: their table word is the next tile's or the buffer's slack, nothing to gather with)
      int lt_n = list_at(lane);
      int jt_n = (lane < nl) ? gjr[lt_n & 0xFFF] : 0;
      double pn0, pn1, pn2;
      {
        const size_t jn = (size_t)(jt_n & MD_JMASK);
        pn0 = xq[2 * jn]; pn1 = xq[2 * jn + 1]; pn2 = zq[2 * jn];
      }
      auto row_loop = [&](auto cref_tag) __attribute__((always_inline)) {
        constexpr bool CREF = decltype(cref_tag)::value;
        // (segment-A entries of a chunk are stored at the top of the next turn, before that turn's requests)
        for (int base = 0; base < nl; base += 64) {
          const int lt = lt_n;
          const int jt = jt_n;
          const double px = pn0, py = pn1, pz = pn2;
          row[posA_prev] = entA_prev;
          posA_prev = dump; entA_prev = 0;
          {
            lt_n = list_at(base + 64 + lane);
            jt_n = (base + 64 + lane < nl) ? gjr[lt_n & 0xFFF] : 0;
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
          NB_CHUNK_TAIL(rmin, ra2, rb2, rc2, excl2)
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
      // One chunk against the cluster: the store of the previous chunk's segment-A entries first, then the four distances.
      auto process = [&](const int base, const int lt, const int jt, const float xf, const float yf, const float zf) __attribute__((always_inline)) {
        row[posA_prev] = entA_prev;
        posA_prev = dump; entA_prev = 0;
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
        const float rmin = vmin3_f32(vmin_f32(r2a[0], r2a[1]), r2a[2], r2a[3]);
        NB_CHUNK_TAIL(rmin, ra2e, rb2e, rc2e, excl2e)
      };
      // The pipeline: list entries two chunks ahead, record and table entry one chunk ahead (a record's address needs its list entry:
      // that is there when the requests of the next chunk go out).  The loop body is TWO chunks with two register sets, each loaded
      // where it is used: written as one chunk per turn, the sets were shifted along with register moves at the end of the turn, and a
      // move of a table entry still in flight is a wait for it -- every turn ended in s_waitcnt vmcnt(0), the full latency of the load
      // it had issued a hundred instructions before (2 000 cycles per chunk on the wave clocks, whatever the arithmetic cost).
      int ltA = list_at(lane), ltB = list_at(64 + lane);
      float xA = s_rx[ltA & 0xFFF], yA = s_ry[ltA & 0xFFF], zA = s_rz[ltA & 0xFFF];
      int jtA = gjr[ltA & 0xFFF];
      for (int base = 0; base < nl; base += 128) {
        const float xB = s_rx[ltB & 0xFFF], yB = s_ry[ltB & 0xFFF], zB = s_rz[ltB & 0xFFF];
        const int jtB = gjr[ltB & 0xFFF];
        const int ltC = list_at(base + 128 + lane);
        process(base, ltA, jtA, xA, yA, zA);
        // (past the end of the list the second chunk names the dummy record in every lane: no entry, nothing stored but the dump)
        xA = s_rx[ltC & 0xFFF]; yA = s_ry[ltC & 0xFFF]; zA = s_rz[ltC & 0xFFF];
        jtA = gjr[ltC & 0xFFF];
        const int ltD = list_at(base + 192 + lane);
        process(base + 64, ltB, jtB, xB, yB, zB);
        ltA = ltC; ltB = ltD;
      }
    }
    row[posA_prev] = entA_prev;
#ifdef PAIR_TIMING
    const unsigned long long tr2 = __builtin_readcyclecounter();
#endif
    const int n = nA + nB + nC + nD;
    const bool bad = nB + nC > capBC || nD > capD || n > maxrow - 64;   // (the row's last 64 words are the loop's dump zone)
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
#ifdef PAIR_TIMING
    __builtin_amdgcn_s_waitcnt(0);
    if (lane == 0) {
      const unsigned long long tr3 = __builtin_readcyclecounter();
      atomicAdd(&sc.dbg[12], tr1 - tr0); atomicAdd(&sc.dbg[13], tr2 - tr1); atomicAdd(&sc.dbg[14], tr3 - tr2);
      atomicAdd(&sc.dbg[15], (unsigned long long)nslow); atomicAdd(&sc.dbg[16], (unsigned long long)nownc);
    }
#endif
  }
#undef NB_CHUNK_TAIL
#undef NB_T_SLOW
#undef NB_T_OWN
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


int mdk_neigh_capB(int maxrow) { return (int)(0.6 * maxrow) / 64 * 64 + 64; }
// capacity of one group's list (16-bit entries): 13/16 of the table (a quarter of PE-10k's clusters reaches 72 %, and NB_LISTPAD dummy entries follow the list); a group that reaches more walks the whole table instead.  (The kernel's LDS must stay below 80 KB for two workgroups per CU: at 83 KB it ran 1.75 times longer.)
static int neigh_qcap(int capj) {
  static const int n16 = scema_env("SCEMA_MD_QCAP16") ? atoi(scema_env("SCEMA_MD_QCAP16")) : 13;   // (test switch: small values force the whole-table path)
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

