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

#include "md_pair_dev.h"

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
#ifdef PAIR_WHATIF_TILE_ORDER
  cell = S.cell_fill[cell];   // (what-if: tiles dispatched in descending order of their atoms, k_cell_build)
#endif
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
#if defined(PAIR_WHATIF_NOFLUSH)      // sensitivity experiments (never in a product build; the forces are wrong): no flush at all ...
        if (ax == 1.2345e300) fs[slot] = ax;
#elif defined(PAIR_WHATIF_STOREFLUSH)  // ... or plain stores instead of the memory-side atomics
        fs[slot] = ax; fs[np + slot] = ay; fs[2 * np + slot] = az;
#else
        atomicAdd(fs + slot, ax); atomicAdd(fs + np + slot, ay); atomicAdd(fs + 2 * np + slot, az);
#endif
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


size_t mdk_pair_lds_bytes(int capj) { return (size_t)capj * (3 * sizeof(double) + sizeof(int)); }
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
