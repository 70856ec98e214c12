// md_pair_p.hip -- the pair kernel as PERSISTENT workgroups (round 6): the production form of k_pair (md_pair.hip) for launches that fill the chip.
//
// k_pair runs one 512-thread workgroup per cell tile, two per CU (74 kB of LDS each): table load | rows | barrier | flush, and a wave that has
// finished its rows waits at the barrier for the slowest of its tile (12 % of a wave's life, with 6 % in the prologue and 7 % in the flush:
// a quarter of it is not the row loop).  Here ONE 1 024-thread workgroup per CU stays for the whole launch and walks a queue of
// tiles with TWO tables in LDS:
//   * the rows of a tile are not dealt to the waves in advance: a wave takes the next row off an LDS counter when it needs one, so the 16 waves
//     finish a tile within a row of each other whatever the rows' lengths;
//   * a wave that finds no row left in tile t goes on to tile t + 1 in the other table at once; the LAST wave to finish tile t flushes its
//     accumulators alone (a few microseconds of one wave in sixteen), takes the next tile off the XCD's queue, stages it into the table it has
//     just emptied and raises its ready flag -- nobody waits at a barrier, and a fast wave can be up to a tile ahead of the slowest;
//   * the queue is one counter per XCD in global memory (a relaxed atomic per tile: nothing is published through it), the tiles of a replica
//     stay on the XCD of k_pair's map, and the drain of the launch is one tile long whatever the order the tiles came in.
// The row loop -- entry layout, prefetch pipeline, the chunk's arithmetic, the DPP reduction at a row's end -- is k_pair's.  Results do not depend
// on which form runs (FP64 atomics: to the summation order).  The parity / energy form (ENG) and launches of fewer tiles than keep 256 workgroups
// busy stay with k_pair.
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

struct PTile { int sim, cell, cs, nclus, nj, seq; };

extern __shared__ double s_pp[];   // [2][capj][3] accumulators | [2][27*4] shifts | [2][ljn] (lj1, lj2) pairs | int [2][capj] j tables | int [2][capj/4] row headers

__device__ __forceinline__ int lds_load_uniform(const int *p) {   // a word every lane reads (LDS broadcast), as a scalar
  return __builtin_amdgcn_readfirstlane(*(volatile const int *)p);
}

template <bool VIR, int NP, bool CLE>
__global__ __launch_bounds__(PT, 4) void k_pair_p(const SimDev *__restrict__ sims, int ntiles, int nsims, int capj, int ljn, unsigned long long *queue, unsigned long long qbase) {
  __shared__ int c_ready[2], c_done[2], c_rowctr[2];
  __shared__ PTile c_tile[2];
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7;
  const int rcap = capj / NI;   // row headers per table: a tile's own cell comes first in its j table, so its clusters number at most capj / 4
  double *const s_fa = s_pp;                                  // [2][3 capj]
  double *const s_sha = s_fa + 6 * (size_t)capj;             // [2][108]
  double *const s_lja = s_sha + 2 * 108;                     // [2][ljn]
  int *const s_jta = (int *)(s_lja + 2 * (size_t)ljn);       // [2][capj]
  int *const s_rha = s_jta + 2 * (size_t)capj;               // [2][rcap]

  // ---- the queue of this XCD: local index k -> (replica, cell); k_pair's placement (xcd_map_at): the tiles of a replica on one XCD ----
  const int full = nsims & ~7, nfull_tiles = (full >> 3) * ntiles, rem_tiles = (nsims - full) * ntiles;
  auto next_tile = [&](int &sim, int &cell) -> bool {   // (one lane asks; every lane gets the answer)
    for (;;) {
      unsigned long long got = 0;
      if (lane == 0) got = __hip_atomic_fetch_add(&queue[xcd], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long got_u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(got >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)got);
      const unsigned long long kk = got_u - qbase;
      if (kk >= 0x40000000ull) return false;   // (far past the end: the counters of a launch stay below 2^30)
      const int k = (int)kk;
      if (k < nfull_tiles) {
        sim = (k / ntiles) * 8 + xcd;
        cell = k % ntiles;
      } else {
        const long long j = (long long)(k - nfull_tiles) * 8 + xcd;
        if (j >= rem_tiles) return false;
        sim = full + (int)(j / ntiles);
        cell = (int)(j % ntiles);
      }
      const SimDev &S = sims[sim];
      if (cell >= S.ncells) continue;
      if (S.cell_start[cell + 1] == S.cell_start[cell]) {   // empty cell: its virial partials are still read by k_ewald_force / k_finish
        if (VIR && lane < TW * 6) S.virp[(size_t)cell * TW * 6 + lane] = 0.0;
        continue;
      }
      return true;
    }
  };
  // stage the next tile of the queue into table b as sequence number seq (one wave): j table, row headers, the replica's image shifts and LJ
  // coefficients; then the ready flag.  The accumulators of the table are zero: the flush leaves them so.
  auto stage = [&](int b, int seq) {
    int sim = -1, cell = 0;
    const bool have = next_tile(sim, cell);
    if (have) {
      const SimDev &S = sims[sim];
      const int cs = S.cell_start[cell], ce = S.cell_start[cell + 1], nj = S.tile_nj[cell], nclus = (ce - cs) / NI;
      int *jt = s_jta + (size_t)b * capj, *rh = s_rha + (size_t)b * rcap;
      const GLOBAL_AS int *gj = as_global(S.tile_jtab) + (size_t)cell * S.capj;
      for (int l = lane; l < nj; l += 64) jt[l] = gj[l];
      const int need_far = S.sc->need_far;
      for (int k = lane; k < nclus; k += 64) {
        const int cl = cs / NI + k;
        rh[k] = (nj > 0) ? S.numneigh[2 * cl] + (need_far ? S.numneigh[2 * cl + 1] : 0) : 0;   // [A|B|C1], then C2 (walked only when some atom has moved far enough)
      }
      double *sh = s_sha + b * 108;
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
      double *lj = s_lja + (size_t)b * ljn;
      for (int k = lane; k < 2 * nt2; k += 64) lj[k] = S.lj[(k & 1) * nt2 + (k >> 1)];   // (lj1, lj2) of a type pair side by side
      if (lane == 0) { c_tile[b].sim = sim; c_tile[b].cell = cell; c_tile[b].cs = cs; c_tile[b].nclus = min(nclus, rcap); c_tile[b].nj = nj; c_tile[b].seq = seq; }
    } else if (lane == 0) { c_tile[b].sim = -1; c_tile[b].seq = seq; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the table is in LDS before the flag that says so
    if (lane == 0) { c_rowctr[b] = 0; c_done[b] = 0; *(volatile int *)&c_ready[b] = seq; }
  };

  for (int k = threadIdx.x; k < 6 * capj; k += PT) s_fa[k] = 0.0;
  if (threadIdx.x < 2) { c_ready[threadIdx.x] = -1; c_done[threadIdx.x] = 0; c_rowctr[threadIdx.x] = 0; }
  __syncthreads();
  // (ONE wave stages the first two tiles, in queue order: were two waves to ask the queue at once, its last tile could land in the second table
  // while the first gets the end mark, and the workgroup would leave at the first.  Later stagings are ordered by the protocol itself: the wave
  // that stages tile t + 2 has to finish tile t + 1 before anyone can stage t + 3.)
  if (wave == 0) { stage(0, 0); stage(1, 1); }
  __syncthreads();

#ifdef PAIR_TIMING
  unsigned long long tm_wait = 0, tm_pro = 0, tm_rows = 0, tm_flush = 0, tm_n = 0;
#define PT_CLK(v) const unsigned long long v = __builtin_readcyclecounter()
#else
#define PT_CLK(v)
#endif
  for (int t = 0;; t++) {
    const int b = t & 1;
    PT_CLK(c0);
    while (lds_load_uniform(&c_ready[b]) != t) __builtin_amdgcn_s_sleep(2);
    PT_CLK(c1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int sim = lds_load_uniform(&c_tile[b].sim);
    if (sim < 0) break;   // the queue is empty (the tile before this one is flushed by the last wave to leave it)
    const int cell = lds_load_uniform(&c_tile[b].cell), cs = lds_load_uniform(&c_tile[b].cs), nclus = lds_load_uniform(&c_tile[b].nclus);
    const SimDev &S = sims[sim];
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
    bool exhausted = false;
    auto grab = [&]() {
      if (exhausted || nrows >= 64) return;
      int k = 0;
      if (lane == 0) k = __hip_atomic_fetch_add(&c_rowctr[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      k = __builtin_amdgcn_readfirstlane(k);
      if (k >= nclus) { exhausted = true; return; }
      const int n = lds_load_uniform(&s_rh[k]);
      const int ke = ((n + 63) >> 6) << 6;   // whole chunks: the row's last one is padded with empty entries
      if (lane == nrows) { h_cl = cs / NI + k; h_nn = ke << 16; }
      nrows += 1;
    };
    grab(); grab(); grab();   // (three rows ahead of the evaluation cursor: the prefetch cursor, two chunks ahead of it, never runs out of known rows)
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
#ifdef PAIR_TIMING
      tm_pro += __builtin_readcyclecounter() - c1;
#endif
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
    PT_CLK(c2);
    // ---- this wave has no row left in the tile.  The last wave to say so flushes the table, stages the tile after next into it ----
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // this wave's LDS atomics precede its count
    int old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(&c_done[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old == PW - 1) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const int nj = lds_load_uniform(&c_tile[b].nj);
      // consecutive table entries are runs of consecutive slots -> coalesced atomics.  Production virial: the tile adds the image-shift part of
      // its non-home entries, sum over entries of shift (x) F_entry (md_pair.hip); one partial row per tile, the other rows of the cell zero
      double vl[6] = {0, 0, 0, 0, 0, 0};
      double *fs = S.fs;
      const size_t np = (size_t)S.npad;
      for (int l = lane; l < nj; l += 64) {
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
              vl[0] = fma(px, ax, vl[0]); vl[1] = fma(py, ay, vl[1]); vl[2] = fma(pz, az, vl[2]);
              vl[3] = fma(px, ay, vl[3]); vl[4] = fma(px, az, vl[4]); vl[5] = fma(py, az, vl[5]);
            }
          }
        }
      }
      if (VIR) {
        double *vp = S.virp + (size_t)cell * TW * 6;
#pragma unroll
        for (int k = 0; k < 6; k++) {
          const double tsum = wave_sum(vl[k]);
          if (lane == 0) vp[k] = tsum;
        }
        if (lane >= 6 && lane < TW * 6) vp[lane] = 0.0;
      }
      stage(b, t + 2);
    }
#ifdef PAIR_TIMING
    { const unsigned long long c3 = __builtin_readcyclecounter(); tm_wait += c1 - c0; tm_rows += c2 - c1; tm_flush += c3 - c2; tm_n += 1; }
#endif
  }
#ifdef PAIR_TIMING
  if (lane == 0 && tm_n) {   // (per wave and tile visit, into the first replica's counters: SCEMA_MD_TIMING prints them as k_pair's)
    SimScalars &sc0 = *sims[0].sc;
    atomicAdd(&sc0.dbg[0], tm_pro); atomicAdd(&sc0.dbg[1], tm_rows - tm_pro); atomicAdd(&sc0.dbg[2], tm_wait); atomicAdd(&sc0.dbg[3], tm_flush); atomicAdd(&sc0.dbg[4], tm_n);
  }
#endif
}

size_t mdk_pair_p_lds_bytes(int capj, int ljn) {
  return (size_t)capj * 2 * (3 * sizeof(double) + sizeof(int)) + 2 * 108 * sizeof(double) + 2 * (size_t)ljn * sizeof(double) + 2 * (size_t)(capj / NI) * sizeof(int);
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
  if (mdk_pair_p_lds_bytes(capj, ljn) > 158 * 1024 || npoly > 16 || (npoly != 14 && npoly != 15 && npoly != 16 && npoly > 12)) return false;
  if (npoly <= 12) launch_pair_p_np<12>(st, d, ns, maxcells, capj, ljn, vir, cle, queue, qbase, nwg);
  else if (npoly <= 14) launch_pair_p_np<14>(st, d, ns, maxcells, capj, ljn, vir, cle, queue, qbase, nwg);
  else if (npoly <= 15) launch_pair_p_np<15>(st, d, ns, maxcells, capj, ljn, vir, cle, queue, qbase, nwg);
  else launch_pair_p_np<16>(st, d, ns, maxcells, capj, ljn, vir, cle, queue, qbase, nwg);
  return true;
}
