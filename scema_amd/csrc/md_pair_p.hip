// md_pair_p.hip -- the pair kernel as PERSISTENT workgroups (round 6): the production form of k_pair (md_pair.hip) for launches that fill the chip.
//
// k_pair runs one 512-thread workgroup per cell tile, two per CU (74 kB of LDS each): table load | rows | barrier | flush, and a wave that has
// finished its rows waits at the barrier for the slowest of its tile (wave clocks, cycles per wave and tile at 576 replicas: prologue 12.5 k,
// rows 96 k, barrier wait 17 k, flush 5 k: a quarter of a wave's life is not the row loop).  Here ONE 1 024-thread workgroup per CU stays for
// the whole launch and keeps TWO tiles in LDS, both being worked on:
//   * a wave takes rows off a tile's LDS counter one at a time (three ahead of the one it evaluates); when its tile has none left it takes
//     rows of the other tile, and only when neither has any does it look for other work;
//   * the wave that completes a tile's last row opens its flush: the table's entries in shares of 64, taken off a counter by every wave that
//     has nothing to compute; the wave that finishes the last share takes the next tile off the XCD's queue and stages it into the table
//     (j table, row headers, the replica's image shifts and LJ coefficients), alone, while the other tile keeps the rest busy;
//   * the queue is one counter per XCD in global memory (a relaxed atomic per tile: nothing is published through it); the tiles of a replica
//     stay on the XCD of k_pair's map.
// Nobody waits at a barrier.  The row loop -- entry layout, prefetch pipeline, the chunk's arithmetic, the DPP reduction at a row's end -- is
// k_pair's.  Results do not depend on which form runs (FP64 atomics: to the summation order).  The parity / energy form (ENG) and launches of
// fewer tiles than keep 256 workgroups busy stay with k_pair.
//
// Reference semantics: pair_style lj/cut/coul/long 12.0 9.0 (in.set.lammps:40), neighbor 2.0 bin (in.set.lammps:27), as md_pair.hip.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "md_device.h"
#include "md_env.h"
#include "md_kernels.h"

#include "md_pair_dev.h"

#define PW 16          // waves of a persistent workgroup
#define PT (PW * 64)

enum { P_EMPTY = 0, P_ACTIVE = 1, P_FLUSHING = 2, P_END = 3 };
#define P_POISON 0x7FFF4000   /* row counter of a table without a tile to take rows from: generation no tile has, row past any tile's last */
#define P_NOSHARE 0x40000000  /* flush counter of a table that is not being flushed: past any table's last share */
struct PCtl {
  int state, gen;
  int rowctr;            // (generation << 16) | next row to take
  int rowsdone;          // rows completed
  int sim, cell, cs, nclus, nj;
  int flushctr, flushdone, pad;
  double vir[6];         // image-shift part of the pair virial, summed over the flush shares
};
// what the kernel was launched with, for the functions below (they are CALLED, not inlined: each gets a register allocation of its own -- inlined
// into one body the row loop spilled 13 vector registers into its hot loop and ran 40 % slower per row than k_pair's)
struct PCfg {
  const SimDev *sims;
  unsigned long long *queue;
  unsigned long long qbase;
  int ntiles, nsims, capj, ljn, rcap, xcd;
};
__shared__ PCtl c_t[2];
__shared__ PCfg c_cfg;

extern __shared__ double s_pp[];   // [2][capj][3] accumulators | [2][27*4] shifts | [2][ljn] (lj1, lj2) pairs | int [2][capj] j tables | int [2][capj/4] row headers

__device__ __forceinline__ int lds_ld(const int *p) {   // a word every lane reads (LDS broadcast), as a scalar; never cached in a register
  return __builtin_amdgcn_readfirstlane(*(volatile const int *)p);
}
__device__ __forceinline__ int lds_inc(int *p, int lane) {   // LDS counter + 1 by one lane, the old value to all
  int v = 0;
  if (lane == 0) v = __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return __builtin_amdgcn_readfirstlane(v);
}
// The LDS layout is fixed at compile time (the tables at their largest, P_CAPJ entries: what fits two of them into a CU's 160 kB), so that
// every LDS address of the row loop is a constant plus an index, as in k_pair, and costs no register.
#define P_CAPJ 2688
#define P_LJN 32          /* doubles of (lj1, lj2) pairs per table: up to 4 atom types */
#define P_RCAP (P_CAPJ / NI)
struct PGeom {
  const SimDev *sims;
  int capj, ljn, rcap;
  double *s_fa, *s_sha, *s_lja;
  int *s_jta, *s_rha;
};
__device__ __forceinline__ const SimDev *p_sims() {
  return (const SimDev *)(((unsigned long long)(unsigned)lds_ld((const int *)&c_cfg.sims + 1) << 32) | (unsigned)lds_ld((const int *)&c_cfg.sims));
}
__device__ __forceinline__ PGeom p_geom() {
  PGeom G;
  G.sims = p_sims();
  G.capj = P_CAPJ; G.ljn = P_LJN; G.rcap = P_RCAP;
  G.s_fa = s_pp;                                               // [2][3 capj]
  G.s_sha = G.s_fa + 6 * (size_t)P_CAPJ;                       // [2][108]
  G.s_lja = G.s_sha + 2 * 108;                                 // [2][ljn]
  G.s_jta = (int *)(G.s_lja + 2 * (size_t)P_LJN);              // [2][capj]
  G.s_rha = G.s_jta + 2 * (size_t)P_CAPJ;                      // [2][rcap]
  return G;
}

// ---- the queue of this XCD: local index k -> (replica, cell); k_pair's placement (xcd_map_at): the tiles of a replica on one XCD ----
template <bool VIR>
__device__ __forceinline__ bool p_next_tile(int *sim_out, int *cell_out) {
  const int lane = lane_id();
  const PGeom G = p_geom();
  const int ntiles = lds_ld(&c_cfg.ntiles), nsims = lds_ld(&c_cfg.nsims), xcd = lds_ld(&c_cfg.xcd);
  unsigned long long *queue = (unsigned long long *)(((unsigned long long)(unsigned)lds_ld((const int *)&c_cfg.queue + 1) << 32) | (unsigned)lds_ld((const int *)&c_cfg.queue));
  const unsigned long long qbase = ((unsigned long long)(unsigned)lds_ld((const int *)&c_cfg.qbase + 1) << 32) | (unsigned)lds_ld((const int *)&c_cfg.qbase);
  const int full = nsims & ~7, nfull_tiles = (full >> 3) * ntiles, rem_tiles = (nsims - full) * ntiles;
  for (;;) {   // (one lane asks; every lane gets the answer)
    unsigned long long got = 0;
    if (lane == 0) got = __hip_atomic_fetch_add(&queue[xcd], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long got_u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(got >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)got);
    const unsigned long long kk = got_u - qbase;
    if (kk >= 0x40000000ull) return false;   // (far past the end: the counters of a launch stay below 2^30)
    const int k = (int)kk;
    int sim, cell;
    if (k < nfull_tiles) {
      sim = (k / ntiles) * 8 + xcd;
      cell = k % ntiles;
    } else {
      const long long j = (long long)(k - nfull_tiles) * 8 + xcd;
      if (j >= rem_tiles) return false;
      sim = full + (int)(j / ntiles);
      cell = (int)(j % ntiles);
    }
    const SimDev &S = G.sims[sim];
    if (cell >= S.ncells) continue;
    if (S.cell_start[cell + 1] == S.cell_start[cell]) {   // empty cell: its virial partials are still read by k_ewald_force / k_finish
      if (VIR && lane < TW * 6) S.virp[(size_t)cell * TW * 6 + lane] = 0.0;
      continue;
    }
    *sim_out = sim; *cell_out = cell;
    return true;
  }
}

// The wave that has finished the last flush share of table b (or, first = true, the one that opens the launch): the flushed tile's virial row,
// then the next tile of the queue into the table -- j table, row headers, the replica's image shifts and LJ coefficients --, then the counters
// that hand it out.  The accumulators of the table are zero: the flush leaves them so.
template <bool VIR>
__device__ __forceinline__ void p_stage(int b_in, int first_in) {
  const int lane = lane_id();
  const int b = __builtin_amdgcn_readfirstlane(b_in), first = __builtin_amdgcn_readfirstlane(first_in);
  const PGeom G = p_geom();
  PCtl &C = c_t[b];
  if (VIR && !first) {
    const SimDev &So = G.sims[lds_ld(&C.sim)];
    double *vp = So.virp + (size_t)lds_ld(&C.cell) * TW * 6;
    if (lane < 6) { vp[lane] = C.vir[lane]; C.vir[lane] = 0.0; }
    else if (lane < TW * 6) vp[lane] = 0.0;
  }
  int sim = -1, cell = 0;
  const bool have = p_next_tile<VIR>(&sim, &cell);
  if (!have) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) *(volatile int *)&C.state = P_END;
    return;
  }
  sim = __builtin_amdgcn_readfirstlane(sim); cell = __builtin_amdgcn_readfirstlane(cell);
  const SimDev &S = G.sims[sim];
  const int cs = S.cell_start[cell], ce = S.cell_start[cell + 1], nj = S.tile_nj[cell], nclus = min((ce - cs) / NI, G.rcap);
  int *jt = G.s_jta + (size_t)b * G.capj, *rh = G.s_rha + (size_t)b * G.rcap;
  const GLOBAL_AS int *gj = as_global(S.tile_jtab) + (size_t)cell * S.capj;
  for (int l = lane; l < nj; l += 64) jt[l] = gj[l];
  const int need_far = S.sc->need_far;
  for (int k = lane; k < nclus; k += 64) {
    const int cl = cs / NI + k;
    rh[k] = (nj > 0) ? S.numneigh[2 * cl] + (need_far ? S.numneigh[2 * cl + 1] : 0) : 0;   // [A|B|C1], then C2 (walked only when some atom has moved far enough)
  }
  double *sh = G.s_sha + b * 108;
  if (lane < 27) {
    BoxD bx;
    box_derive(S.sc->box, bx);
    const int s0 = lane % 3 - 1, s1 = (lane / 3) % 3 - 1, s2 = lane / 9 - 1;
    sh[4 * lane + 0] = bx.h[0] * s0 + bx.h[5] * s1 + bx.h[4] * s2;
    sh[4 * lane + 1] = bx.h[1] * s1 + bx.h[3] * s2;
    sh[4 * lane + 2] = bx.h[2] * s2;
    sh[4 * lane + 3] = 0.0;
  }
  const int nt2 = S.ntypes * S.ntypes;
  double *lj = G.s_lja + (size_t)b * G.ljn;
  for (int k = lane; k < 2 * nt2; k += 64) lj[k] = S.lj[(k & 1) * nt2 + (k >> 1)];   // (lj1, lj2) of a type pair side by side
  // (the generation first: a wave that took a number off the old tile's exhausted counter and still reads the old generation then reads the old
  // tile's fields, not a mixture)
  const int gen = (lds_ld(&C.gen) + 1) & 0x3FFF;
  if (lane == 0) *(volatile int *)&C.gen = gen;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) { C.sim = sim; C.cell = cell; C.cs = cs; C.nclus = nclus; C.nj = nj; C.rowsdone = 0; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the table is in LDS before the counter that hands its rows out
  if (lane == 0) { *(volatile int *)&C.rowctr = gen << 16; *(volatile int *)&C.state = P_ACTIVE; }
}

// The flush of table b, a share of 64 entries at a time, by whoever has nothing to compute.  Consecutive table entries are runs of consecutive
// slots -> coalesced atomics.  Production virial: the tile adds the image-shift part of its non-home entries, sum over entries of
// shift (x) F_entry (md_pair.hip).  The wave that finishes the last share stages the next tile.  Returns whether a share was taken.
template <bool VIR>
__device__ __forceinline__ int p_help_flush(int b_in) {
  const int lane = lane_id();
  const int b = __builtin_amdgcn_readfirstlane(b_in);
  const PGeom G = p_geom();
  PCtl &C = c_t[b];
  int did = 0;
  for (;;) {
    if (lds_ld(&C.state) != P_FLUSHING) break;
    const int sh = lds_inc(&C.flushctr, lane);
    // (what is read from here on belongs to the flush that handed the share out: it cannot end before the share is done.  A counter that is
    // not handing out shares holds a number past any table's last share.)
    const int nj = lds_ld(&C.nj), nsh = (nj + 63) >> 6;
    if (sh >= nsh) break;
    did = 1;
    const SimDev &S = G.sims[lds_ld(&C.sim)];
    double *const s_f = G.s_fa + 3 * (size_t)b * G.capj;
    const double *const s_shift = G.s_sha + b * 108;
    const int *const s_jtab = G.s_jta + (size_t)b * G.capj;
    double *fs = S.fs;
    const size_t np = (size_t)S.npad;
    double vl[6] = {0, 0, 0, 0, 0, 0};
    const int l = sh * 64 + lane;
    if (l < nj) {
      const double ax = s_f[3 * l], ay = s_f[3 * l + 1], az = s_f[3 * l + 2];
      if (ax != 0.0 || ay != 0.0 || az != 0.0) {
        s_f[3 * l] = 0.0; s_f[3 * l + 1] = 0.0; s_f[3 * l + 2] = 0.0;
        const int je = s_jtab[l];
        const size_t slot = (size_t)(je & MD_JMASK);
        atomicAdd(fs + slot, ax); atomicAdd(fs + np + slot, ay); atomicAdd(fs + 2 * np + slot, az);
        if (VIR) {
          const int code = (je >> 23) & 31;
          if (code != CODE_HOME) {
            const double px = s_shift[4 * code], py = s_shift[4 * code + 1], pz = s_shift[4 * code + 2];
            vl[0] = px * ax; vl[1] = py * ay; vl[2] = pz * az; vl[3] = px * ay; vl[4] = px * az; vl[5] = py * az;
          }
        }
      }
    }
    if (VIR) {
#pragma unroll
      for (int k = 0; k < 6; k++) {
        const double tsum = wave_sum(vl[k]);
        if (lane == 0 && tsum != 0.0) lds_add(&C.vir[k], tsum);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lds_inc(&C.flushdone, lane) == nsh - 1) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      if (lane == 0) { *(volatile int *)&C.flushctr = P_NOSHARE; *(volatile int *)&C.state = P_EMPTY; }
      p_stage<VIR>(b, 0);
      break;
    }
  }
  return did;
}

// This wave's visit to the tile in table b, of which it has taken row k_first with generation gen: rows off the tile's counter one at a time,
// three ahead of the one being evaluated, until the tile has none left.  k_pair's row loop.  Returns whether this wave completed the tile's last row.
template <int NP, bool CLE, int B>
__device__ __forceinline__ bool p_rows(const int k_first) {
  const int lane = lane_id();
  constexpr int b = B;
  const PGeom G = p_geom();
  double *const s_fa = G.s_fa, *const s_sha = G.s_sha, *const s_lja = G.s_lja;
  int *const s_jta = G.s_jta, *const s_rha = G.s_rha;
  constexpr int capj = P_CAPJ, ljn = P_LJN, rcap = P_RCAP;
  PCtl &C = c_t[B];
  const int sim = lds_ld(&C.sim), cs = lds_ld(&C.cs);
  const SimDev &S = G.sims[sim];
  double *const s_f = s_fa + 3 * (size_t)b * capj;
  const double *const s_shift = s_sha + b * 108;
  const double *const s_lj = s_lja + (size_t)b * ljn;
  const int *const s_jtab = s_jta + (size_t)b * capj;
  const int *const s_rh = s_rha + (size_t)b * rcap;
  const int nt = S.ntypes;
  const int maxrow = S.maxneigh;
  const GLOBAL_AS int *neigh = as_global(S.neigh);
  const GLOBAL_AS double *xq = as_global((const double *)S.xq);   // (x,y) halves
  const GLOBAL_AS double *zq = xq + 2 * (size_t)S.npad;              // (z,q) halves
  const GLOBAL_AS int *stype = as_global(S.stype);
  // (everything the row loop needs of the replica's descriptor is in registers from here on: behind the LDS and memory atomics of the loop the
  // compiler would read a field of S from memory again at every use)
  double cp[NP];
#pragma unroll
  for (int m = 0; m < NP; m++) cp[m] = S.coul_poly_g[m];
  double cp_top = cp[NP - 1];
  asm volatile("" : "+v"(cp_top));
  const double g = S.g_ewald, g2u = g * g * S.coul_uscale;
  const double cutc2 = S.cut_coul2, cutl2 = S.cut_lj2;
  const double cutmax2 = fmax(cutc2, cutl2);

  // ---- this wave's rows of the tile: taken off the tile's counter one at a time, three rows ahead of the one being evaluated ----
  int nrows = 0, h_cl = 0, h_nn = 0;   // lane r: the r-th row this wave has taken (cluster; first and last entry of its walk)
  bool exhausted = false, completed = false;
  auto take = [&](int k) {
    const int n = lds_ld(&s_rh[k]);
    const int ke = ((n + 63) >> 6) << 6;   // whole chunks: the row's last one is padded with empty entries
    if (lane == nrows) { h_cl = cs / NI + k; h_nn = ke << 16; }
    nrows += 1;
  };
  // (while this wave holds a row it has not completed the tile cannot complete, so the table keeps its generation: only the FIRST row of a
  // visit, taken above with the generation it came with, can belong to a tile other than the one the wave looked at)
  auto grab = [&]() {
    if (exhausted || nrows >= 64) return;
    const int old = lds_inc(&C.rowctr, lane);
    if ((old >> 16) != lds_ld(&C.gen) || (old & 0xFFFF) >= lds_ld(&C.nclus)) { exhausted = true; return; }   // (the table keeps its generation while this wave holds a row)
    take(old & 0xFFFF);
  };
  take(k_first);
  grab(); grab();   // (three rows ahead of the evaluation cursor: the prefetch cursor, two chunks ahead of it, never runs out of known rows)
#define H_KB(v) ((v) & 0xFFFF)
#define H_KE(v) ((int)((unsigned)(v) >> 16))
  if (nrows > 0) {
    // prefetch cursor: the chunk two ahead of the one being evaluated
    int pr = 0;
    int pcl = __builtin_amdgcn_readlane(h_cl, 0), pnn = __builtin_amdgcn_readlane(h_nn, 0);
    int pk = H_KB(pnn);
    pnn = H_KE(pnn);
    const unsigned lane4 = 4u * (unsigned)lane;
    auto fetch = [&]() -> int {
      int v = 0;
      if (pr < nrows) {
        const GLOBAL_AS char *row = (const GLOBAL_AS char *)(neigh + (size_t)pcl * maxrow);
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
    int e_n = fetch(), e_nn = fetch();
    // evaluation cursor
    int r = 0;
    int s0 = __builtin_amdgcn_readlane(h_cl, 0) * NI, nn = __builtin_amdgcn_readlane(h_nn, 0);
    int k0 = H_KB(nn);
    nn = H_KE(nn);
    double xi[NI], yi[NI], zi[NI], qi[NI], fx[NI], fy[NI], fz[NI];
    int ti[NI];
#pragma unroll
    for (int a = 0; a < NI; a++) {
      xi[a] = xq[2 * (size_t)(s0 + a)]; yi[a] = xq[2 * (size_t)(s0 + a) + 1]; zi[a] = zq[2 * (size_t)(s0 + a)];
      qi[a] = MD_QQRD2E * zq[2 * (size_t)(s0 + a) + 1];
      ti[a] = stype[s0 + a] * nt;
      fx[a] = fy[a] = fz[a] = 0.0;
    }
    auto chunk = [&](const int e, const int jt, const double xj, const double yj, const double zj, const double qj) __attribute__((always_inline)) {
      const int mask = (e >> E_MASK_SHIFT) & 0xF;  // 0 for the padding of a row's last chunk
      if (mask != 0) {
        const int cs4 = (int)(((unsigned)jt >> 21) & 0x7Cu);   // 4 * image code
        const double xs = xj + s_shift[cs4], ys = yj + s_shift[cs4 + 1], zs = zj + s_shift[cs4 + 2];
        const int tj = (e >> E_TYPE_SHIFT) & 0xF;
        double gx = 0.0, gy = 0.0, gz = 0.0;   // reaction force on j
        asm volatile("" : "+v"(gx), "+v"(gy), "+v"(gz));
#pragma unroll
        for (int a = 0; a < NI; a++) {
          if (!(mask & (1 << a))) continue;
          const double dx = xi[a] - xs, dy = yi[a] - ys, dz = zi[a] - zs;
          const double rsq = dx * dx + dy * dy + dz * dz;
          if (CLE) {
            if (rsq < cutl2) {
              const double rinv = rsqrt_f64(rsq);
              const double r2inv = rinv * rinv;
              const double r6inv = r2inv * r2inv * r2inv;
              const double2 lj12 = ((const double2 *)s_lj)[ti[a] + tj];
              double fp = r6inv * (lj12.x * r6inv - lj12.y) * r2inv;
              if (rsq < cutc2) {
                // qq (1 - x H(u)) / r^3 with x H = r (g H) = r P and r / r = 1:  qq (1/r - P) / r^2
                const double tt = fma(rsq, g2u, -1.0);
                double p = cp_top;
#pragma unroll
                for (int m = NP - 2; m >= 0; m--) p = fma(p, tt, cp[m]);
                fp = fma(qi[a] * qj * (rinv - p), r2inv, fp);
              }
              const double tx = dx * fp, ty = dy * fp, tz = dz * fp;
              fx[a] += tx; fy[a] += ty; fz[a] += tz;
              gx -= tx; gy -= ty; gz -= tz;
            }
          } else if (rsq < cutmax2) {
            const double rinv = rsqrt_f64(rsq);
            const double r2inv = rinv * rinv;
            double fp = 0.0;
            if (rsq < cutc2) {
              const double rr = rsq * rinv;   // r
              const double tt = fma(rsq, g2u, -1.0);
              double p = cp[NP - 1];
#pragma unroll
              for (int m = NP - 2; m >= 0; m--) p = fma(p, tt, cp[m]);
              fp = qi[a] * qj * rinv * fma(-rr, p, 1.0) * r2inv;
            }
            if (rsq < cutl2) {
              const double r6inv = r2inv * r2inv * r2inv;
              const double2 lj12 = ((const double2 *)s_lj)[ti[a] + tj];
              fp = fma(r6inv * (lj12.x * r6inv - lj12.y), r2inv, fp);
            }
            const double tx = dx * fp, ty = dy * fp, tz = dz * fp;
            fx[a] += tx; fy[a] += ty; fz[a] += tz;
            gx -= tx; gy -= ty; gz -= tz;
          }
        }
        const int l = e & E_LMASK;
        lds_add(&s_f[3 * l], gx); lds_add(&s_f[3 * l + 1], gy); lds_add(&s_f[3 * l + 2], gz);
      }
      k0 += 64;
      if (r < nrows && k0 >= nn) {
        // row finished: the forces on the cluster's own atoms (transposing DPP butterfly over the quad, then a row scan, as in k_pair)
        const bool b0 = lane & 1, b1 = lane & 2;
        double u[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
          const double *f = (c == 0) ? fx : (c == 1) ? fy : fz;
          const double w0 = (b0 ? f[1] : f[0]) + dpp_mov<DPP_QUAD_XOR1>(b0 ? f[0] : f[1]);
          const double w1 = (b0 ? f[3] : f[2]) + dpp_mov<DPP_QUAD_XOR1>(b0 ? f[2] : f[3]);
          double tq = (b1 ? w1 : w0) + dpp_mov<DPP_QUAD_XOR2>(b1 ? w0 : w1);
          tq += dpp_mov<DPP_ROW_SHR4>(tq);
          tq += dpp_mov<DPP_ROW_SHR8>(tq);
          u[c] = tq;
        }
        if ((lane & 12) == 12) {
          const int l = s0 - cs + (lane & 3);
          lds_add(&s_f[3 * l], u[0]); lds_add(&s_f[3 * l + 1], u[1]); lds_add(&s_f[3 * l + 2], u[2]);
        }
        if (lds_inc(&C.rowsdone, lane) == lds_ld(&C.nclus) - 1) completed = true;   // (behind this row's LDS atomics: the LDS takes a wave's instructions in order)
        grab();   // one row taken per row finished
        r += 1;
        if (r < nrows) {
          s0 = __builtin_amdgcn_readlane(h_cl, r) * NI; nn = __builtin_amdgcn_readlane(h_nn, r);
          k0 = H_KB(nn);
          nn = H_KE(nn);
#pragma unroll
          for (int a = 0; a < NI; a++) {
            xi[a] = xq[2 * (size_t)(s0 + a)]; yi[a] = xq[2 * (size_t)(s0 + a) + 1]; zi[a] = zq[2 * (size_t)(s0 + a)];
            qi[a] = MD_QQRD2E * zq[2 * (size_t)(s0 + a) + 1];
            ti[a] = stype[s0 + a] * nt;
            fx[a] = fy[a] = fz[a] = 0.0;
          }
        }
      }
    };
    struct Rec { double x, y, z, q; };
    auto deref = [&](const int e, int &jt, Rec &R) __attribute__((always_inline)) {
      jt = s_jtab[e & E_LMASK];
      const size_t j = (size_t)(jt & MD_JMASK);
      R.x = xq[2 * j]; R.y = xq[2 * j + 1]; R.z = zq[2 * j]; R.q = zq[2 * j + 1];
    };
    int eA = e_n, eB = e_nn, eC, jtX, jtY;
    Rec X, Y;
    deref(eA, jtX, X);
    do {
      deref(eB, jtY, Y);
      eC = fetch();
      chunk(eA, jtX, X.x, X.y, X.z, X.q);
      deref(eC, jtX, X);
      eA = fetch();
      chunk(eB, jtY, Y.x, Y.y, Y.z, Y.q);
      { const int tmp = eA; eA = eC; eB = tmp; }
    } while (r < nrows);
  }
#undef H_KB
#undef H_KE
  return completed;
}

template <bool VIR, int NP, bool CLE>
__global__ __launch_bounds__(PT, 4) void k_pair_p(const SimDev *__restrict__ sims, int ntiles, int nsims, int capj, int ljn, unsigned long long *queue, unsigned long long qbase) {
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (threadIdx.x == 0) {
    c_cfg.sims = sims; c_cfg.queue = queue; c_cfg.qbase = qbase; c_cfg.ntiles = ntiles; c_cfg.nsims = nsims; c_cfg.capj = capj; c_cfg.ljn = ljn;
    c_cfg.rcap = P_RCAP;   // row headers per table: a tile's own cell comes first in its j table, so its clusters number at most capj / 4
    c_cfg.xcd = blockIdx.x & 7;
  }
  for (int k = threadIdx.x; k < 6 * P_CAPJ; k += PT) s_pp[k] = 0.0;
  if (threadIdx.x < 2) {
    PCtl &C = c_t[threadIdx.x];
    C.state = P_EMPTY; C.gen = 0; C.rowctr = P_POISON; C.rowsdone = 0; C.sim = 0; C.cell = 0; C.cs = 0; C.nclus = 0; C.nj = 0; C.flushctr = P_NOSHARE; C.flushdone = 0;
    for (int k = 0; k < 6; k++) C.vir[k] = 0.0;
  }
  __syncthreads();
  if (wave == 0) { p_stage<VIR>(0, 1); p_stage<VIR>(1, 1); }   // (one wave, in queue order)
  __syncthreads();
#ifdef PAIR_TIMING
  unsigned long long tm_idle = 0, tm_rows = 0, tm_flush = 0, tm_n = 0;
#endif
  int home = wave & 1;
  for (;;) {
#ifdef PAIR_TIMING
    const unsigned long long c0 = __builtin_readcyclecounter();
#endif
    // ---- rows: this wave's home table first, the other one if that has none to give ----
    int b = -1, k_first = 0;
    for (int pass = 0; pass < 2 && b < 0; pass++) {
      const int bb = home ^ pass;
      PCtl &Cb = c_t[bb];
      if (lds_ld(&Cb.state) != P_ACTIVE) continue;
      if ((lds_ld(&Cb.rowctr) & 0xFFFF) >= lds_ld(&Cb.nclus)) continue;   // (a look before the leap: a failed attempt costs a count in the row bits)
      const int old = lds_inc(&Cb.rowctr, lane);
      if ((old >> 16) != lds_ld(&Cb.gen) || (old & 0xFFFF) >= lds_ld(&Cb.nclus)) continue;
      b = bb; k_first = old & 0xFFFF;
    }
    if (b >= 0) {
      home = b;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const bool completed = (b == 0) ? p_rows<NP, CLE, 0>(k_first) : p_rows<NP, CLE, 1>(k_first);
#ifdef PAIR_TIMING
      const unsigned long long c1 = __builtin_readcyclecounter();
      tm_rows += c1 - c0; tm_n += 1;
#endif
      // no row left for this wave in the table.  The wave that completed the tile's last row opens its flush, and starts it (whoever runs out
      // of rows meanwhile takes shares too)
      if (completed) {
        PCtl &C = c_t[b];
        if (lane == 0) { *(volatile int *)&C.rowctr = P_POISON; C.flushdone = 0; *(volatile int *)&C.flushctr = 0; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) *(volatile int *)&C.state = P_FLUSHING;
        (void)p_help_flush<VIR>(b);
#ifdef PAIR_TIMING
        tm_flush += __builtin_readcyclecounter() - c1;
#endif
      }
      continue;
    }
    // ---- nothing to compute: a flush share of a table that is being emptied; else wait for a tile, or leave when none will come ----
    int did = 0;
    for (int bb = 0; bb < 2; bb++)
      if (lds_ld(&c_t[bb].state) == P_FLUSHING) did |= __builtin_amdgcn_readfirstlane(p_help_flush<VIR>(bb));
#ifdef PAIR_TIMING
    if (did) tm_flush += __builtin_readcyclecounter() - c0;
#endif
    if (did) continue;
    if (lds_ld(&c_t[0].state) == P_END && lds_ld(&c_t[1].state) == P_END) break;
    __builtin_amdgcn_s_sleep(4);
#ifdef PAIR_TIMING
    tm_idle += __builtin_readcyclecounter() - c0;
#endif
  }
#ifdef PAIR_TIMING
  if (lane == 0) {   // (per wave, into the first replica of the launch: SCEMA_MD_TIMING prints the batch's sum)
    SimScalars &sc0 = *sims[0].sc;
    atomicAdd(&sc0.dbg[1], tm_rows); atomicAdd(&sc0.dbg[2], tm_idle); atomicAdd(&sc0.dbg[3], tm_flush); atomicAdd(&sc0.dbg[4], tm_n);
  }
#endif
}

size_t mdk_pair_p_lds_bytes(int, int) {
  return (size_t)P_CAPJ * 2 * (3 * sizeof(double) + sizeof(int)) + 2 * 108 * sizeof(double) + 2 * (size_t)P_LJN * sizeof(double) + 2 * (size_t)P_RCAP * sizeof(int);
}

template <bool VIR, int NP, bool CLE>
static void launch_pair_p(hipStream_t st, const SimDev *d, int ns, int ntiles, int capj, int ljn, unsigned long long *queue, unsigned long long qbase, int nwg) {
  const size_t lds = mdk_pair_p_lds_bytes(capj, ljn);
  static size_t optin_tab[16] = {0};
  size_t &optin = lds_optin_slot(optin_tab);
  if (lds > optin) { (void)hipFuncSetAttribute((const void *)k_pair_p<VIR, NP, CLE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); optin = lds; }
  hipLaunchKernelGGL((k_pair_p<VIR, NP, CLE>), dim3((unsigned)nwg), dim3(PT), lds, st, d, ntiles, ns, capj, ljn, queue, qbase);
}
template <int NP>
static void launch_pair_p_np(hipStream_t st, const SimDev *d, int ns, int ntiles, int capj, int ljn, int vir, int cle, unsigned long long *queue, unsigned long long qbase, int nwg) {
  if (vir) { if (cle) launch_pair_p<true, NP, true>(st, d, ns, ntiles, capj, ljn, queue, qbase, nwg); else launch_pair_p<true, NP, false>(st, d, ns, ntiles, capj, ljn, queue, qbase, nwg); }
  else { if (cle) launch_pair_p<false, NP, true>(st, d, ns, ntiles, capj, ljn, queue, qbase, nwg); else launch_pair_p<false, NP, false>(st, d, ns, ntiles, capj, ljn, queue, qbase, nwg); }
}
// queue: eight counters (one per XCD) that only ever grow; qbase: what they read when this launch's first tile is taken (the caller adds
// MDK_PAIR_P_STRIDE per launch on that queue)
bool mdk_pair_persistent(hipStream_t st, const SimDev *d, int ns, int maxcells, int capj, int ntypes_max, int vir, int npoly, int cle, unsigned long long *queue, unsigned long long qbase, int nwg) {
  const int ljn = 2 * ntypes_max * ntypes_max;
  if (capj > P_CAPJ || ljn > P_LJN || mdk_pair_p_lds_bytes(capj, ljn) > 158 * 1024 || npoly > 16 || (npoly != 14 && npoly != 15 && npoly != 16 && npoly > 12)) return false;
  if (npoly <= 12) launch_pair_p_np<12>(st, d, ns, maxcells, capj, ljn, vir, cle, queue, qbase, nwg);
  else if (npoly <= 14) launch_pair_p_np<14>(st, d, ns, maxcells, capj, ljn, vir, cle, queue, qbase, nwg);
  else if (npoly <= 15) launch_pair_p_np<15>(st, d, ns, maxcells, capj, ljn, vir, cle, queue, qbase, nwg);
  else launch_pair_p_np<16>(st, d, ns, maxcells, capj, ljn, vir, cle, queue, qbase, nwg);
  return true;
}
