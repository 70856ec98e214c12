// md_reax.hip -- gfx950 kernels of the ReaxFF force field path (BASELINE config 5).
//
// Replaces, for replicas run with force_field "reax", what lammps_scripts_reax/in.strain.lammps:10-12 and
// ELASTIC/potential.mod.lammps:5-7 select in LAMMPS: pair_style reax/c + fix qeq/reax 1 0.0 10.0 1e-6 [LAMMPS-ext].
// The integrator, thermostat, fix deform, pressure sampling and the batch machinery are the ones of the OPLS path
// (md_kernels.hip); only the force stage differs.
//
// Layout: the per-atom lists of the lane-per-atom passes (bond rows, near rows) are entry-major ([k][i], reax/rx_types.h): a wave reads 64
// consecutive entries at every step of its row walk; the lists a wave walks ROW by row (neighbour rows, owned pairs, matrix rows) are row-major.
// One launch covers every replica of the batch (blockIdx.y).
//   k_rx_prepare ........ box -> view, rebuild bookkeeping, zero the energy parts
//   k_rx_wrap, k_rx_neigh  neighbour rows inside cutoff + skin (a wave per row), rebuilt when an atom has moved half the skin
//                         (LAMMPS rebuilds every step, `neigh_modify every 1 delay 0 check no`: same pairs inside the cutoff)
//   k_rx_hrow ........... the matrix of the charge equilibration: each pair once in its owner's row (<true>) or every pair in both rows
//   k_rx_qeq_* .......... the two conjugate-gradient solves, both right-hand sides per sweep over the matrix, no host round trips:
//                         sweep_sym + step (symmetric form: one workgroup per replica between two sweeps) or sweep + update (full rows),
//                         finish / finish_sym (stragglers, charges, history)
//   k_rx_bonds ... k_rx_back2  bond orders, energy terms (a lane per angle / torsion item), reverse-mode forces (reax/rx_core.h)
//   k_rx_nonbonded_once . tapered van der Waals + shielded Coulomb, every owned pair once
#include <hip/hip_runtime.h>

#include <algorithm>

#include <cstdlib>

#include "md_device.h"
#include "md_env.h"
#include "md_kernels.h"
#include "md_reax.h"
#include "md_types.h"
#include "reax/rx_core.h"

// LDS FP64 atomic add without return value (ds_add_f64)
__device__ __forceinline__ void lds_add_f64(double *p, double v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#define RX_TPB 128
#ifndef RX_OCC
#define RX_OCC 2   /* waves per SIMD the term kernels are compiled for (the torsion pass would take 300 VGPRs unbounded: one wave per SIMD) */
#endif

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

// Neighbour rows, a wave per row (round 5; until then a lane per atom walked all atoms of the replica: 2 000 waves for 72 replicas, 2.0 ms per
// rebuild of all of them, and two stores of four bytes per entry that no neighbour lane shared a cache line with): a workgroup owns 64
// consecutive atoms, wave w their rows 8 w .. 8 w + 7 (as the matrix build has them), its lanes over the replica's atoms as partners, 64 at a
// time.  A chunk's accepted partners leave as one contiguous store into the row (ballot + lane rank): rows come out sorted by partner index as
// before (deterministic; with several images per partner, image-major inside a chunk).  O(N^2) distance tests per rebuild, any box shape.
// Only the row-major rows are written (RxView::nbT): the entry-major copy of the FULL rows had two readers left, the hydrogen-bond pass and the
// both-ends non-bonded kernel of the tests, which read rows now (rx_nb_entry); the NEAR rows keep both copies (a dozen entries per row).
// mimg = 0: boxes at least two list radii wide, minimum image; otherwise all images up to mimg[d] boxes away.
#define RX_NBR 8   /* rows per wave */
__global__ __launch_bounds__(64 * (64 / RX_NBR)) void k_rx_neigh(const SimDev *sims, RxView *views, const RxParams *__restrict__ P, double rlist) {
  const SimDev &S = sims[blockIdx.y];
  if (!S.sc->rebuild) return;
  const RxView &V = views[blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ double s_rn2[RX_MAXT * RX_MAXT];   // the near-row radius of every type pair (RxParams::rnear2)
  if (threadIdx.x < RX_MAXT * RX_MAXT) s_rn2[threadIdx.x] = P->rnear2[threadIdx.x];
  __syncthreads();
  const int n = V.n, np = V.npad, maxnb = V.maxnb, maxnbn = V.maxnbn;
  const int r0 = blockIdx.x * 64 + wave * RX_NBR;
  if (r0 >= n) return;   // (wave-uniform; no barrier below)
  const int nr = min(RX_NBR, n - r0);
  const double rl2 = rlist * rlist;
  const double h0 = wave_uniform(V.h[0]), h1 = wave_uniform(V.h[1]), h2 = wave_uniform(V.h[2]), h3 = wave_uniform(V.h[3]), h4 = wave_uniform(V.h[4]),
               h5 = wave_uniform(V.h[5]);
  const double ih0 = 1.0 / h0, ih1 = 1.0 / h1, ih2 = 1.0 / h2;
  const int m0 = V.mimg[0], m1 = V.mimg[1], m2 = V.mimg[2];
  const bool minimage = (m0 | m1 | m2) == 0;
  double xr[RX_NBR], yr[RX_NBR], zr[RX_NBR];
  int len[RX_NBR], lenn[RX_NBR], trow[RX_NBR], lown0[RX_NBR];
#pragma unroll
  for (int g = 0; g < RX_NBR; g++) {
    const int row = min(r0 + g, n - 1);
    xr[g] = wave_uniform(S.x[3 * row]); yr[g] = wave_uniform(S.x[3 * row + 1]); zr[g] = wave_uniform(S.x[3 * row + 2]);
    trow[g] = __builtin_amdgcn_readfirstlane(V.rtype[row]) * RX_MAXT;
    len[g] = 0; lenn[g] = 0; lown0[g] = 0;
  }
  // one accepted (partner, image) per lane -> the row's next entries
  auto append = [&](int g, bool ok, bool near, int ent) __attribute__((always_inline)) {
    const unsigned long long m = __ballot(ok);
    if (m == 0) return;
    const int row = r0 + g;
    const int pos = len[g] + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
    if (ok && pos < maxnb) V.nbT[(size_t)row * maxnb + pos] = ent;
    len[g] += __popcll(m);
    lown0[g] += __popcll(__ballot(ok && !rx_owns(row, ent)));   // (minimum image: these all come before the owned ones)
    const unsigned long long mn = __ballot(ok && near);
    if (mn == 0) return;
    const int posn = lenn[g] + __builtin_amdgcn_mbcnt_hi((unsigned)(mn >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mn, 0));
    if (ok && near && posn < maxnbn) {
      V.nbnT[(size_t)row * maxnbn + posn] = ent;
      V.nbn[(size_t)posn * np + row] = ent;
    }
    lenn[g] += __popcll(mn);
  };
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int j = j0 + lane;
    const bool jl = j < n;
    const int jc = jl ? j : n - 1;
    const double xj = S.x[3 * jc], yj = S.x[3 * jc + 1], zj = S.x[3 * jc + 2];
    const int tj = V.rtype[jc];
#pragma unroll
    for (int g = 0; g < RX_NBR; g++) {
      if (g >= nr) break;   // (wave-uniform)
      const double rn2 = s_rn2[trow[g] + tj];
      double dx = xj - xr[g], dy = yj - yr[g], dz = zj - zr[g];
      if (minimage) {
        const double n2 = rint(dz * ih2);
        dz -= n2 * h2; dy -= n2 * h3; dx -= n2 * h4;
        const double n1 = rint(dy * ih1);
        dy -= n1 * h1; dx -= n1 * h5;
        const double n0 = rint(dx * ih0);
        dx -= n0 * h0;
        const double r2 = dx * dx + dy * dy + dz * dz;
        const int code = (2 - (int)n0) + 5 * (2 - (int)n1) + 25 * (2 - (int)n2);
        append(g, jl && j != r0 + g && !(r2 > rl2), r2 <= rn2, j | (code << 24));
      } else {
        for (int sz = -m2; sz <= m2; sz++)
          for (int sy = -m1; sy <= m1; sy++)
            for (int sx = -m0; sx <= m0; sx++) {
              const double ex = dx + sx * h0 + sy * h5 + sz * h4, ey = dy + sy * h1 + sz * h3, ez = dz + sz * h2;
              const double r2 = ex * ex + ey * ey + ez * ez;
              const bool self = j == r0 + g && sx == 0 && sy == 0 && sz == 0;
              append(g, jl && !self && !(r2 > rl2), r2 <= rn2, j | (((sx + 2) + 5 * (sy + 2) + 25 * (sz + 2)) << 24));
            }
      }
    }
  }
  bool full = false;
  int most = 0;
#pragma unroll
  for (int g = 0; g < RX_NBR; g++) {
    if (g >= nr) break;
    full |= len[g] > maxnb || lenn[g] > maxnbn;
    const int c = min(len[g], maxnb);
    most = max(most, c);
    if (lane == 0) { V.nb_cnt[r0 + g] = c; V.nbn_cnt[r0 + g] = min(lenn[g], maxnbn); V.nb_own0[r0 + g] = min(lown0[g], c); }
  }
  if (lane == 0) {
    if (full) atomicOr(V.overflow, 1);
    atomicMax(&S.sc->maxneigh_seen, most);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// charge equilibration
// ------------------------------------------------------------------------------------------------------------------
// Row walks are shared by RX_KS waves: a workgroup owns 64 consecutive atoms (lane = atom), wave w takes the entries
// k = w, w + RX_KS, ... of their rows and the partial results meet in LDS.  With one lane per atom alone a replica of a few
// thousand atoms gives the chip a few dozen waves; this gives it RX_KS times as many, each with a short row walk.
#define RX_KT (64 * RX_KS)

// The matrix rows of this step: a workgroup owns 64 consecutive atoms, wave w the rows 8 w .. 8 w + 7 of them, one after the other with its
// lanes over the entries of the list row (read from the row-major copy of the list: contiguous); the entries inside the taper radius
// are compacted with a ballot and leave as contiguous stores (RxView::hval).  (Lanes over rows and the entries in [k][row] planes, as the
// other passes have them, made the compacted stores scatter over the planes: 1.30 against 0.70 ms per 72-replica step.)
// The image shift of a row entry (rx_shift: two integer divisions by 5, nine multiplications) from a table of the 125 codes in LDS -- the box
// is one per replica, and a workgroup works on one replica.  Same products in the same order as rx_shift: bitwise the same shifts.
#define RX_NSHIFT 125
__device__ __forceinline__ void rx_shift_table(const RxView &V, double *s_sh) {
  for (int code = threadIdx.x; code < RX_NSHIFT; code += blockDim.x) {
    const int sx = code % 5 - 2, sy = (code / 5) % 5 - 2, sz = code / 25 - 2;
    s_sh[3 * code] = sx * V.h[0] + sy * V.h[5] + sz * V.h[4];
    s_sh[3 * code + 1] = sy * V.h[1] + sz * V.h[3];
    s_sh[3 * code + 2] = sz * V.h[2];
  }
}
// SYM (round 5, the symmetric form of the charge solve): only the pairs a row OWNS are computed -- the tail of the row from nb_own0 on -- and stored,
// matrix entry (hpk) and list entry (hown) at the same position; hlen = hownlen.  Half the arithmetic and half the stores of the full rows.
#ifndef RX_HROW_RG
#define RX_HROW_RG 2   /* rows of a wave in flight */
#endif
template <bool SYM>
__global__ __launch_bounds__(RX_KT) void k_rx_hrow(const SimDev *sims, const RxView *views, const RxParams *__restrict__ P) {
  const RxView V = views[blockIdx.y];
  (void)sims;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // the type-pair constants of the kernel in LDS: one dependent global load less per entry
  // (positions and types staged in LDS as well, as the sweep does with its vector: 418.5 against 416.5 evaluations/s, not kept)
  __shared__ double s_gamma[RX_MAXT * RX_MAXT];
  if (threadIdx.x < RX_MAXT * RX_MAXT) s_gamma[threadIdx.x] = P->tbp[threadIdx.x].gamma;
  __shared__ double s_sh[3 * RX_NSHIFT];
  rx_shift_table(V, s_sh);
  __syncthreads();
  double tap[8];
#pragma unroll
  for (int m = 0; m < 8; m++) tap[m] = P->tap[m];
  const double swb2 = P->swb * P->swb;
  // A wave walks TWO of its rows at a time (RG): their requests are independent, so a record has two chunks of arithmetic to arrive in
  // instead of one.  Per row, one turn of the loop is: (1) the STORES of the chunk computed in the turn before, (2) the requests of the
  // turn after (partner records of the next chunk, list entries of the one after it), (3) the arithmetic of this chunk (rx_qeq_entry of
  // reax/rx_core.h, same operations in the same order).  The stores come first because the memory counter counts in order and the
  // compiler cannot count stores that sit behind a branch: a wait for any load issued before them becomes a wait for everything, the
  // stores' own round trip included -- with the stores last in the turn every turn ended on that (80 % of the waves' cycles waiting,
  // rocprofv3 SQ_WAIT_ANY).  Issued first, they have the whole turn to complete.
  constexpr int RG = RX_HROW_RG;
  for (int r = 0; r < 64 / RX_KS; r += RG) {
    const int ifirst = blockIdx.x * 64 + wave * (64 / RX_KS) + r;
    if (ifirst >= V.n) return;   // (wave-uniform; no barrier below)
    int row[RG], cnt[RG], own0[RG], len[RG], lown[RG], e1[RG], e2[RG], tjn[RG], ent_prev[RG];
    size_t base[RG];
    const double *grow[RG];
    double xi0[RG], xi1[RG], xi2[RG], p0[RG], p1[RG], p2[RG], h_prev[RG];
    int cmax = 0;
#pragma unroll
    for (int g = 0; g < RG; g++) {
      const bool live = ifirst + g < V.n;
      row[g] = live ? ifirst + g : ifirst;
      own0[g] = (SYM && live) ? V.nb_own0[row[g]] : 0;
      cnt[g] = live ? V.nb_cnt[row[g]] - own0[g] : 0;   // (entries this walk takes: all of the row, or its owned tail)
      cmax = max(cmax, cnt[g]);
      base[g] = (size_t)row[g] * V.maxnb;
      grow[g] = s_gamma + V.rtype[row[g]] * RX_MAXT;
      // (the row's own position: the same for every lane; as scalars -- and waited for here, not inside the loop behind the stores)
      xi0[g] = wave_uniform(V.x[3 * row[g]]); xi1[g] = wave_uniform(V.x[3 * row[g] + 1]); xi2[g] = wave_uniform(V.x[3 * row[g] + 2]);
      len[g] = 0; lown[g] = 0; h_prev[g] = -1.0; ent_prev[g] = -1;
    }
    auto load_ent = [&](int g, int k0) -> int { const int k = k0 + lane; return (k < cnt[g]) ? V.nbT[base[g] + own0[g] + k] : -1; };
    auto gather = [&](int g) __attribute__((always_inline)) {
      const int j = (e1[g] >= 0) ? (e1[g] & RX_JMASK) : 0;
      p0[g] = V.x[3 * j]; p1[g] = V.x[3 * j + 1]; p2[g] = V.x[3 * j + 2]; tjn[g] = V.rtype[j];
    };
    auto flush = [&](int g) __attribute__((always_inline)) {
      const unsigned long long m = __ballot(h_prev[g] >= 0.0);
      if (SYM) {
        if (h_prev[g] >= 0.0) {
          const size_t o = base[g] + len[g] + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
          V.hpk[o] = rx_hpack(h_prev[g], ent_prev[g] & RX_JMASK);
          V.hown[o] = ent_prev[g];
        }
        len[g] += __popcll(m);
        lown[g] = len[g];
        return;
      }
      if (h_prev[g] >= 0.0) {
        const size_t o = base[g] + len[g] + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
        if (V.hpk) V.hpk[o] = rx_hpack(h_prev[g], ent_prev[g] & RX_JMASK);
        else { V.hval[o] = h_prev[g]; V.hcol32[o] = ent_prev[g] & RX_JMASK; }
      }
      len[g] += __popcll(m);
      // the pairs of the row that this end owns, for the non-bonded pass (each pair once)
      const bool mine = h_prev[g] >= 0.0 && rx_owns(row[g], ent_prev[g]);
      const unsigned long long mo = __ballot(mine);
      if (mine) V.hown[base[g] + lown[g] + __builtin_amdgcn_mbcnt_hi((unsigned)(mo >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mo, 0))] = ent_prev[g];
      lown[g] += __popcll(mo);
    };
#pragma unroll
    for (int g = 0; g < RG; g++) { e1[g] = load_ent(g, 0); e2[g] = load_ent(g, 64); }
#pragma unroll
    for (int g = 0; g < RG; g++) gather(g);
    // (the first records are waited for HERE: a wait left to the first use inside the loop stays in the loop, behind the stores of every turn)
#pragma unroll
    for (int g = 0; g < RG; g++) asm volatile("" : : "v"(p0[g]), "v"(p1[g]), "v"(p2[g]), "v"(tjn[g]), "v"(e2[g]));
    for (int k0 = 0; k0 < cmax; k0 += 64) {
      int ent[RG], tj[RG];
      double q0[RG], q1[RG], q2[RG];
#pragma unroll
      for (int g = 0; g < RG; g++) { ent[g] = e1[g]; q0[g] = p0[g]; q1[g] = p1[g]; q2[g] = p2[g]; tj[g] = tjn[g]; e1[g] = e2[g]; }
#pragma unroll
      for (int g = 0; g < RG; g++) flush(g);
#pragma unroll
      for (int g = 0; g < RG; g++) { gather(g); e2[g] = load_ent(g, k0 + 128); }
#pragma unroll
      for (int g = 0; g < RG; g++) {
        double h = -1.0;
        if (ent[g] >= 0) {
          const double *sh = s_sh + 3 * ((ent[g] >> 24) & 0x7F);
          const double d0 = q0[g] - xi0[g] + sh[0], d1 = q1[g] - xi1[g] + sh[1], d2 = q2[g] - xi2[g] + sh[2];
          const double r2 = d0 * d0 + d1 * d1 + d2 * d2;
          if (!(r2 > swb2)) {
            const double rr = r2 * rx_rsqrt(r2);
            double tp = tap[7];
#pragma unroll
            for (int m = 6; m >= 0; m--) tp = tp * rr + tap[m];      // rx_taper (value only)
            h = tp * RX_EV_TO_KCALPMOL * rx_icbrt(r2 * rr + grow[g][tj[g]]);
          }
        }
        h_prev[g] = h; ent_prev[g] = ent[g];
      }
    }
#pragma unroll
    for (int g = 0; g < RG; g++) {
      flush(g);
      if (lane == 0 && ifirst + g < V.n) { V.hlen[row[g]] = len[g]; V.hownlen[row[g]] = lown[g]; }
    }
  }
}

// fix qeq/reax: H s = -chi and H t = -1 by Jacobi-preconditioned conjugate gradients from the extrapolated previous solutions
// (init_matvec: cubic for s, quadratic for t), until sqrt(r.z)/|b| <= tol for each; q = s - (sum s / sum t) t.
//
// The iteration runs as launches over all replicas, two per iteration (a conjugate-gradient step has two scalar products that
// every row needs before it can go on):
//   k_rx_qeq_sweep   y = H z (the only pass over the matrix: 8 bytes per stored entry -- RxView::hpk -- both systems per entry), then row-local
//                    d = z + beta d,  q = y + beta q  (H d by linearity: only z is ever gathered),  partial sums of d.q
//   k_rx_qeq_update  alpha = sigma / d.q;  s += alpha d;  r -= alpha q;  z = r / eta;  partial sums of r.z
// Scalars never sit in one memory word: every workgroup writes its partial sums with plain stores and every workgroup of the
// NEXT launch folds them in a fixed order (deterministic, no atomics, no fences inside a launch).  The host does not know the
// iteration count: it issues as many iterations as the previous run needed plus a margin; workgroups of a converged replica
// leave at once, and k_rx_qeq_finish -- one workgroup per replica -- finishes a replica that needs more with the in-kernel loop,
// then forms the charges and shifts the history.
//
// qpart layout (doubles): [0, 2 NV) and [2 NV, 4 NV): r.z partials of even / odd iterations; [4 NV, 6 NV): b.b partials;
// [6 NV, 6 NV + 2 NB'): d.q partials; then [.., + 2 NV) and [.., + 2 NV): r.D^-1 r partials of even / odd iterations (the reference's
// convergence measure: the same sums as r.z while the preconditioner is the Jacobi one).  NV = RX_QNV (update workgroups, padded),
// NB' = npad / RX_SWR + 1 >= the sweep workgroups.
//
// Preconditioner (round 5): `fix qeq/reax 1 0.0 10.0 1e-6` (in.strain.lammps:12) fixes the answer -- stop when sqrt(r.D^-1 r) / |b| <= 1e-6 --
// not the way there.  z = M r with M a sparse approximate inverse on the bonded pattern (k_rx_qeq_pm_rows / _sym; RX_PM_RADIUS) instead of
// D^-1 takes 5.1 instead of 12.4 iterations per solve on PE-1620 (tools/qeq_precond_gate.py, gated offline before it was built), for a
// product with ~4 entries per row next to the matrix's ~545.  The stop is still taken on the reference's measure.
#define QEQ_TPB 1024
#define RX_QEQ_COLD RX_QEQ_COLD_SOLVES
#define QEQ_UT 256
#define RX_QNV(npad) (((npad) + QEQ_UT - 1) / QEQ_UT)

// fold per-block partial pairs in a fixed order; every lane of the calling wave gets the sums
__device__ __forceinline__ void qeq_fold(const double *part, int count, double &a, double &b) {
  double sa = 0.0, sb = 0.0;
  for (int k = threadIdx.x & 63; k < count; k += 64) { sa += part[2 * k]; sb += part[2 * k + 1]; }
  a = wave_sum(sa);   // (DPP: every lane holds the same sum; as shuffles these were 24 LDS round trips per fold, at the head of every launch)
  b = wave_sum(sb);
}
struct QeqScal { double sig[2], bn[2]; bool run[2]; };
// where the r.D^-1 r partials of an iteration's parity sit in qpart
__device__ __forceinline__ double RX_G *qeq_conv_part(const RxView &V, int parity) {
  return V.qpart + 6 * RX_QNV(V.npad) + 2 * (V.npad / RX_SWR + 1) + 2 * RX_QNV(V.npad) * parity;
}
// the run flags of iteration `it`: the reference's measure sqrt(r.D^-1 r) / |b| against the tolerance (r.z itself under the Jacobi preconditioner)
__device__ __forceinline__ void qeq_run_flags(const RxView &V, int it, double tol, QeqScal &Q) {
  double c0 = Q.sig[0], c1 = Q.sig[1];
  if (V.pm_on) qeq_fold((const double *)qeq_conv_part(V, it & 1), (V.n + QEQ_UT - 1) / QEQ_UT, c0, c1);
  Q.run[0] = sqrt(c0) / Q.bn[0] > tol;
  Q.run[1] = sqrt(c1) / Q.bn[1] > tol;
}
// scalars of iteration `it` from the partial sums of the launches before it
__device__ __forceinline__ QeqScal qeq_scalars(const RxView &V, int it, double tol) {
  const int nv = RX_QNV(V.npad), nvl = (V.n + QEQ_UT - 1) / QEQ_UT;   // slots, slots in use
  QeqScal Q;
  qeq_fold(V.qpart + 2 * nv * (it & 1), nvl, Q.sig[0], Q.sig[1]);
  qeq_fold(V.qpart + 4 * nv, nvl, Q.bn[0], Q.bn[1]);
  Q.bn[0] = sqrt(Q.bn[0]); Q.bn[1] = sqrt(Q.bn[1]);
  qeq_run_flags(V, it, tol, Q);
  return Q;
}

// The same scalars, and up to two more folds, from ONE round of loads: a launch used to open with three folds one after the other (three dependent
// memory round trips at the head of every workgroup of a 90 us kernel -- and of the 6 us update).  Partial sums of different kinds go to
// different rows of 16 lanes; the row sums are the additions wave_sum makes inside a row, so every scalar is the same bit for bit as folded
// alone (run flags agree between kernels whichever form they use).  For at most 16 partial pairs per kind (32 for `wide`, which takes two rows).
//   rows 0, 1: r.z of iteration it, b.b;  row 2 (+ 3): `extra`, `nextra` pairs (r.z of iteration it - 1, or the d.q partials)
__device__ __forceinline__ QeqScal qeq_scalars_with(const RxView &V, int it, double tol, const double *extra, int nextra, double &xs, double &xt) {
  const int nv = RX_QNV(V.npad), nvl = (V.n + QEQ_UT - 1) / QEQ_UT;
  QeqScal Q;
  if (nvl > 16 || nextra > 32) {   // (uniform) large replicas: one fold after the other
    Q = qeq_scalars(V, it, tol);
    xs = 0.0; xt = 0.0;
    if (extra) qeq_fold(extra, nextra, xs, xt);
    return Q;
  }
  const int lane = threadIdx.x & 63, row = lane >> 4, k = lane & 15;
  const double *src = (row == 0) ? V.qpart + 2 * nv * (it & 1) : (row == 1) ? V.qpart + 4 * nv : extra;
  const int cnt = (row < 2) ? nvl : nextra, kk = (row < 2) ? k : lane - 32;
  double a = 0.0, b = 0.0;
  if (src && kk < cnt) { a = src[2 * kk]; b = src[2 * kk + 1]; }
  a = row_sum(a); b = row_sum(b);
  Q.sig[0] = lane_value(a, 0); Q.sig[1] = lane_value(b, 0);
  Q.bn[0] = sqrt(lane_value(a, 16)); Q.bn[1] = sqrt(lane_value(b, 16));
  xs = lane_value(a, 48) + lane_value(a, 32); xt = lane_value(b, 48) + lane_value(b, 32);   // (row 3 + row 2: the order of wave_sum)
  qeq_run_flags(V, it, tol, Q);
  return Q;
}

// setup != 0: the solve of a run's step 0.  A replica whose history was kept from the run before (RxView::warm) stands where that
// run's last solve stood: its guess is the newest stored solution, not the extrapolation one step ahead
__global__ __launch_bounds__(QEQ_UT) void k_rx_qeq_guess(const RxView *views, int setup, int sym) {
  const RxView V = views[blockIdx.y];
  const int i = blockIdx.x * QEQ_UT + threadIdx.x;
  if (i >= V.n) return;
  const size_t np = V.npad;
  const double *sh = V.s_hist, *th = V.t_hist;
  const bool same_place = setup && V.warm;
  const double s0 = same_place ? sh[i] : 4.0 * (sh[i] + sh[2 * np + i]) - (6.0 * sh[np + i] + sh[3 * np + i]);
  const double t0 = same_place ? th[i] : 3.0 * (th[i] - th[np + i]) + th[2 * np + i];
  V.s[i] = s0; V.t[i] = t0;
  double2 *z = (double2 *)(V.qwork + 6 * np);
  z[i] = make_double2(s0, t0);
  if (sym) { V.qwork[8 * np + i] = 0.0; V.qwork[9 * np + i] = 0.0; }   // (the product vector the symmetric sweeps add to)
}

// it < 0: the first product H x0 of the solve (x0 sits in z), stored in q
// ZLDS: the vector the rows gather from (one (s, t) pair per atom, 16 bytes) is staged in LDS first.  A workgroup's rows gather
// ~17 000 pairs from a vector of 1 620 (PE-1620): through the caches every gather is an L2 round trip (the workgroups of a replica
// sit on different CUs, so no L1 holds a replica's vector for long), and the kernel waited for those, not for the matrix stream.
extern __shared__ double2 s_zl[];
template <bool COL16, bool ZLDS>
__global__ __launch_bounds__(RX_KT) void k_rx_qeq_sweep(const RxView *views, const RxParams *__restrict__ P, double tol, int it) {
  const RxView V = views[blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * RX_SWR + lane, n = V.n;   // (the row of lane < RX_SWR of wave 0 in the row-local part below)
  if (blockIdx.x * RX_SWR >= n) return;
  QeqScal Q;
  double beta_s = 0.0, beta_t = 0.0;
  if (it >= 0) {
    double ps, pt;
    Q = qeq_scalars_with(V, it, tol, (it > 0) ? V.qpart + 2 * RX_QNV(V.npad) * ((it - 1) & 1) : nullptr, (V.n + QEQ_UT - 1) / QEQ_UT, ps, pt);
    if (!Q.run[0] && !Q.run[1]) return;
    if (it > 0) { beta_s = Q.sig[0] / ps; beta_t = Q.sig[1] / pt; }
  }
  const size_t np = V.npad;
  const GLOBAL_AS dvec2 *z = as_global((const dvec2 *)(V.qwork + 6 * np));
  if (ZLDS) {
    // (four loads in flight per thread: one at a time, every element waited a memory round trip before its LDS write)
    for (int k0 = threadIdx.x; k0 < n; k0 += 4 * RX_KT) {
      double2 t[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const int k = k0 + u * RX_KT; t[u] = ldg2(z, k < n ? k : n - 1); }
#pragma unroll
      for (int u = 0; u < 4; u++) { const int k = k0 + u * RX_KT; if (k < n) s_zl[k] = t[u]; }
    }
    __syncthreads();
  }
  // the products of this workgroup's 64 rows: wave w takes the rows 8 w .. 8 w + 7, lanes over the entries of a row
  __shared__ double s_y[2][RX_SWR];
  const GLOBAL_AS unsigned long long *pk = as_global(V.hpk);
  const GLOBAL_AS int *c32 = as_global(V.hcol32);
  const GLOBAL_AS double *hv = as_global(V.hval);
  const GLOBAL_AS int *hlen = as_global(V.hlen);
  // (two rows at a time: their loads are independent)
#define RX_SWEEP_RG 2
  for (int r0 = 0; r0 < RX_SWR / RX_KS; r0 += RX_SWEEP_RG) {
    const int lr0 = wave * (RX_SWR / RX_KS) + r0, row0 = blockIdx.x * RX_SWR + lr0;
    int len[RX_SWEEP_RG], lmax = 0;
    size_t base[RX_SWEEP_RG];
    double ps[RX_SWEEP_RG], pt[RX_SWEEP_RG];
#pragma unroll
    for (int q = 0; q < RX_SWEEP_RG; q++) {
      const int row = row0 + q;
      len[q] = (row < n) ? hlen[row] : 0;   // (wave-uniform)
      base[q] = (size_t)min(row, n - 1) * V.maxnb;
      lmax = max(lmax, len[q]);
      ps[q] = 0.0; pt[q] = 0.0;
    }
    // the matrix stream one chunk ahead of the products (global loads: they stay in flight across the LDS gathers, md_device.h)
    int jn[RX_SWEEP_RG];
    double hn[RX_SWEEP_RG];
    auto load = [&](int c0) {
      const int c = c0 + lane;
#pragma unroll
      for (int q = 0; q < RX_SWEEP_RG; q++) {
        const bool on = c < len[q];
        const size_t o = base[q] + (on ? c : 0);
        // (a masked lane must not trust entry 0 of the row either: a row without neighbours inside the taper radius was never written)
        if (COL16) {
          const unsigned long long b = on ? pk[o] : 0ull;
          hn[q] = rx_hunpack(b, &jn[q]);
        } else {
          jn[q] = on ? c32[o] : 0;
          hn[q] = on ? hv[o] : 0.0;
        }
      }
    };
    load(0);
    for (int c0 = 0; c0 < lmax; c0 += 64) {
      int j[RX_SWEEP_RG];
      double h[RX_SWEEP_RG];
#pragma unroll
      for (int q = 0; q < RX_SWEEP_RG; q++) { j[q] = jn[q]; h[q] = hn[q]; }
      load(c0 + 64);
#pragma unroll
      for (int q = 0; q < RX_SWEEP_RG; q++) {
        const double2 zj = ZLDS ? s_zl[j[q]] : ldg2(z, j[q]);
        ps[q] = fma(h[q], zj.x, ps[q]);
        pt[q] = fma(h[q], zj.y, pt[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < RX_SWEEP_RG; q++) {
      const double a = wave_sum(ps[q]), bsum = wave_sum(pt[q]);
      if (lane == 0) { s_y[0][lr0 + q] = a; s_y[1][lr0 + q] = bsum; }
    }
  }
  __syncthreads();
  double ys = 0.0, yt = 0.0;
  if (wave != 0) return;
  double dq_s = 0.0, dq_t = 0.0;
  if (lane < RX_SWR && i < n) {
    ys = s_y[0][lane]; yt = s_y[1][lane];
    const double eta = P->sbp[V.rtype[i]].eta;
    const double2 zi = ZLDS ? s_zl[i] : ldg2(z, i);
    ys = fma(eta, zi.x, ys); yt = fma(eta, zi.y, yt);
    double2 *d = (double2 *)(V.qwork + 2 * np), *q = (double2 *)(V.qwork + 4 * np);
    if (it < 0) {
      q[i] = make_double2(ys, yt);
    } else {
      // (iteration 0: d = z, q = y -- the arrays hold what the solve before left, and q the first product H x0 that the update of
      // it < 0 and its preconditioner read)
      double2 di = (it > 0) ? d[i] : make_double2(0.0, 0.0), qi = (it > 0) ? q[i] : make_double2(0.0, 0.0);
      if (Q.run[0]) { di.x = fma(beta_s, di.x, zi.x); qi.x = fma(beta_s, qi.x, ys); dq_s = di.x * qi.x; }
      if (Q.run[1]) { di.y = fma(beta_t, di.y, zi.y); qi.y = fma(beta_t, qi.y, yt); dq_t = di.y * qi.y; }
      d[i] = di; q[i] = qi;
    }
  }
  if (it >= 0) {
    dq_s = wave_sum(dq_s); dq_t = wave_sum(dq_t);
    if (lane == 0) {
      double *pb = V.qpart + 6 * RX_QNV(V.npad) + 2 * blockIdx.x;
      pb[0] = dq_s; pb[1] = dq_t;
    }
  }
}

// ---- the preconditioner: a sparse approximate inverse on the bonded pattern ----
// Row i of M = row i of (H[P, P])^-1 with P = {i} + its neighbours within RX_PM_RADIUS (from the near rows, a superset), then M := (M + M^T) / 2
// in a second launch (a row needs its neighbours' rows).  H[P, P] is a handful of shielded Coulomb terms: a dense solve of at most 8 x 8 per
// atom (symmetric positive definite: elimination without pivoting).  Rebuilt with the neighbour rows.
__global__ __launch_bounds__(QEQ_UT) void k_rx_qeq_pm_rows(const SimDev *sims, const RxView *views, const RxParams *__restrict__ P) {
  const RxView V = views[blockIdx.y];
  // (built with the neighbour rows: a preconditioner may be stale -- one built at the first step of an evaluation and kept for all 31 saves the
  // same 59 % of the iterations as one rebuilt every step, profiles/r05_qeq_precond_gate.txt -- and every run starts with a list build)
  if (!V.pm_on || !sims[blockIdx.y].sc->rebuild) return;
  const int i = blockIdx.x * QEQ_UT + threadIdx.x;
  if (i >= V.n) return;
  const size_t np = V.npad;
  int col[RX_PM_MAX], ty[RX_PM_MAX];
  double px[RX_PM_MAX][3];
  int m = 1;
  col[0] = i; ty[0] = V.rtype[i]; px[0][0] = px[0][1] = px[0][2] = 0.0;
  const int cnt = V.nbn_cnt[i];
  for (int k = 0; k < cnt && m < RX_PM_MAX; k++) {
    const int e = V.nbn[(size_t)k * np + i];
    double d[3];
    const int j = rx_partner(&V, i, e, d);
    if (j == i) continue;
    if (d[0] * d[0] + d[1] * d[1] + d[2] * d[2] <= RX_PM_RADIUS * RX_PM_RADIUS) {
      col[m] = j; ty[m] = V.rtype[j]; px[m][0] = d[0]; px[m][1] = d[1]; px[m][2] = d[2];
      m++;
    }
  }
  double A[RX_PM_MAX][RX_PM_MAX], rhs[RX_PM_MAX];
  for (int a = 0; a < m; a++) {
    rhs[a] = (a == 0) ? 1.0 : 0.0;
    A[a][a] = P->sbp[ty[a]].eta;
    for (int b2 = a + 1; b2 < m; b2++) {
      const double d0 = px[a][0] - px[b2][0], d1 = px[a][1] - px[b2][1], d2 = px[a][2] - px[b2][2];
      const double r2 = d0 * d0 + d1 * d1 + d2 * d2;
      double h = 0.0;
      if (!(r2 > P->swb * P->swb)) {
        const double r = r2 * rx_rsqrt(r2);
        double tp = P->tap[7];
        for (int q = 6; q >= 0; q--) tp = tp * r + P->tap[q];
        h = tp * RX_EV_TO_KCALPMOL * rx_icbrt(r2 * r + P->tbp[ty[a] * RX_MAXT + ty[b2]].gamma);
      }
      A[a][b2] = h; A[b2][a] = h;
    }
  }
  for (int c = 0; c < m; c++) {          // elimination
    const double inv = 1.0 / A[c][c];
    for (int a = c + 1; a < m; a++) {
      const double f = A[a][c] * inv;
      for (int b2 = c; b2 < m; b2++) A[a][b2] -= f * A[c][b2];
      rhs[a] -= f * rhs[c];
    }
  }
  for (int a = m - 1; a >= 0; a--) {     // back substitution
    double v = rhs[a];
    for (int b2 = a + 1; b2 < m; b2++) v -= A[a][b2] * rhs[b2];
    rhs[a] = v / A[a][a];
  }
  V.pm_len[i] = m;
  for (int k = 0; k < m; k++) { V.pm_col[(size_t)k * np + i] = col[k]; V.pm_raw[(size_t)k * np + i] = rhs[k]; }
}
__global__ __launch_bounds__(QEQ_UT) void k_rx_qeq_pm_sym(const SimDev *sims, const RxView *views) {
  const RxView V = views[blockIdx.y];
  if (!V.pm_on || !sims[blockIdx.y].sc->rebuild) return;
  const int i = blockIdx.x * QEQ_UT + threadIdx.x;
  if (i >= V.n) return;
  const size_t np = V.npad;
  const int m = V.pm_len[i];
  V.pm_val[i] = V.pm_raw[i];
  for (int k = 1; k < m; k++) {
    const int j = V.pm_col[(size_t)k * np + i];
    double other = 0.0;
    bool found = false;
    const int mj = V.pm_len[j];
    for (int kk = 1; kk < mj; kk++)
      if (V.pm_col[(size_t)kk * np + j] == i) { other = V.pm_raw[(size_t)kk * np + j]; found = true; }
    // (a neighbour whose own row was full and left this atom out: the pair is dropped from both rows)
    V.pm_val[(size_t)k * np + i] = found ? 0.5 * (V.pm_raw[(size_t)k * np + i] + other) : 0.0;
  }
}
// z_i = sum_k M_ik r_k for both systems, the residuals r_k handed in by a functor (the update launches compute a neighbour's NEW residual
// from arrays that no workgroup writes in that launch)
template <class F>
__device__ __forceinline__ double2 qeq_pm_apply(const RxView &V, int i, F &&res) {
  const size_t np = V.npad;
  const int m = V.pm_len[i];
  double z0 = 0.0, z1 = 0.0;
  for (int k = 0; k < m; k++) {
    const double w = V.pm_val[(size_t)k * np + i];
    const double2 rk = res(V.pm_col[(size_t)k * np + i]);
    z0 = fma(w, rk.x, z0); z1 = fma(w, rk.y, z1);
  }
  return make_double2(z0, z1);
}
// the residual array of an iteration's parity (rx_types.h: qwork)
__device__ __forceinline__ double2 *qeq_rbuf(const RxView &V, int parity) { return (double2 *)(V.qwork + (parity ? 8 * (size_t)V.npad : 0)); }

// it < 0: r = b - H x0 (into the residual array of parity 0), z = M r, partial sums of r.z and r.D^-1 r (slots of iteration 0) and b.b
__global__ __launch_bounds__(QEQ_UT) void k_rx_qeq_update(const RxView *views, const RxParams *__restrict__ P, double tol, int it) {
  const RxView V = views[blockIdx.y];
  const int i = blockIdx.x * QEQ_UT + threadIdx.x, n = V.n;
  if (blockIdx.x * QEQ_UT >= n) return;
  const size_t np = V.npad;
  const int nv = RX_QNV(V.npad);
  const double2 *r_in = qeq_rbuf(V, it < 0 ? 0 : (it & 1));
  double2 *r_out = qeq_rbuf(V, it < 0 ? 0 : ((it + 1) & 1));
  const double2 *d = (const double2 *)(V.qwork + 2 * np), *q = (const double2 *)(V.qwork + 4 * np);
  double2 *z = (double2 *)(V.qwork + 6 * np);
  __shared__ double s_red[6][QEQ_UT / 64];
  double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0, c0 = 0.0, c1 = 0.0;
  if (it < 0) {
    if (i < n) {
      // (q holds the first product H x0, which this launch leaves alone: the neighbours' residuals come from it)
      auto res0 = [&](int k) { const int tk = V.rtype[k]; const double2 y = q[k]; return make_double2(-P->sbp[tk].chi - y.x, -1.0 - y.y); };
      const int ti = V.rtype[i];
      const double eta = P->sbp[ti].eta, chi = P->sbp[ti].chi;
      const double2 ri = res0(i);
      const double2 zi = V.pm_on ? qeq_pm_apply(V, i, res0) : make_double2(ri.x / eta, ri.y / eta);
      r_out[i] = ri;
      z[i] = zi;
      p0 = ri.x * zi.x; p1 = ri.y * zi.y; p2 = chi * chi; p3 = 1.0;
      c0 = ri.x * ri.x / eta; c1 = ri.y * ri.y / eta;
    }
  } else {
    double dq_s, dq_t;
    const QeqScal Q = qeq_scalars_with(V, it, tol, V.qpart + 6 * nv, (V.n + RX_SWR - 1) / RX_SWR, dq_s, dq_t);
    if (!Q.run[0] && !Q.run[1]) {
      // converged: the scalar products travel on unchanged, so that every later launch sees the same (converged) state
      if (threadIdx.x == 0) {
        const double *pf = V.qpart + 2 * nv * (it & 1) + 2 * blockIdx.x;
        double *pa = V.qpart + 2 * nv * ((it + 1) & 1) + 2 * blockIdx.x;
        pa[0] = pf[0]; pa[1] = pf[1];
        if (V.pm_on) {
          const double RX_G *cf = qeq_conv_part(V, it & 1) + 2 * blockIdx.x;
          double RX_G *ca = qeq_conv_part(V, (it + 1) & 1) + 2 * blockIdx.x;
          ca[0] = cf[0]; ca[1] = cf[1];
        }
      }
      return;
    }
    const double al_s = Q.run[0] ? Q.sig[0] / dq_s : 0.0, al_t = Q.run[1] ? Q.sig[1] / dq_t : 0.0;
    if (i < n) {
      // a converged system keeps r (alpha = 0), z and with them its scalar products: it stays converged
      auto res1 = [&](int k) { const double2 rk = r_in[k], qk = q[k]; return make_double2(fma(-al_s, qk.x, rk.x), fma(-al_t, qk.y, rk.y)); };
      const double eta = P->sbp[V.rtype[i]].eta;
      const double2 di = d[i];
      const double2 ri = res1(i);
      double2 zi = z[i];
      if (V.pm_on) {
        const double2 zn = qeq_pm_apply(V, i, res1);
        if (Q.run[0]) zi.x = zn.x;
        if (Q.run[1]) zi.y = zn.y;
      } else {
        if (Q.run[0]) zi.x = ri.x / eta;
        if (Q.run[1]) zi.y = ri.y / eta;
      }
      if (Q.run[0]) V.s[i] = fma(al_s, di.x, V.s[i]);
      if (Q.run[1]) V.t[i] = fma(al_t, di.y, V.t[i]);
      r_out[i] = ri; z[i] = zi;
      p0 = ri.x * zi.x; p1 = ri.y * zi.y;
      c0 = ri.x * ri.x / eta; c1 = ri.y * ri.y / eta;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) V.qstat[0] += 1;
  }
  p0 = wave_sum(p0); p1 = wave_sum(p1);
  if (it < 0) { p2 = wave_sum(p2); p3 = wave_sum(p3); }
  if (V.pm_on) { c0 = wave_sum(c0); c1 = wave_sum(c1); }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { s_red[0][wave] = p0; s_red[1][wave] = p1; s_red[2][wave] = p2; s_red[3][wave] = p3; s_red[4][wave] = c0; s_red[5][wave] = c1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0;
    for (int w = 0; w < QEQ_UT / 64; w++) { a0 += s_red[0][w]; a1 += s_red[1][w]; a2 += s_red[2][w]; a3 += s_red[3][w]; a4 += s_red[4][w]; a5 += s_red[5][w]; }
    const int par = (it < 0) ? 0 : ((it + 1) & 1);
    double *pa = V.qpart + 2 * nv * par + 2 * blockIdx.x;
    pa[0] = a0; pa[1] = a1;
    if (it < 0) { double *pc = V.qpart + 4 * nv + 2 * blockIdx.x; pc[0] = a2; pc[1] = a3; }
    if (V.pm_on) { double RX_G *ca = qeq_conv_part(V, par) + 2 * blockIdx.x; ca[0] = a4; ca[1] = a5; }
  }
}

template <int NT>
__device__ __forceinline__ void qeq_reduce2t(double &a, double &b, double *lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  a = wave_sum(a);
  b = wave_sum(b);
  __syncthreads();
  if (lane == 0) { lds[wave] = a; lds[16 + wave] = b; }
  __syncthreads();
  double sa = 0.0, sb = 0.0;
#pragma unroll
  for (int w = 0; w < NT / 64; w++) { sa += lds[w]; sb += lds[16 + w]; }
  a = sa;
  b = sb;
}
__device__ __forceinline__ void qeq_reduce2(double &a, double &b, double *lds) { qeq_reduce2t<QEQ_TPB>(a, b, lds); }
// the end of a solve, one workgroup per replica: charges q = s - (sum s / sum t) t, the history pushed down, the solver's statistics
// (`it`: iterations made by the single-workgroup loop after the launched ones; `unconverged`: the iteration limit was reached)
__device__ __forceinline__ void qeq_finish_tail(const RxView &V, int setup, int it, bool unconverged, double *s_red) {
  const int n = V.n, tid = threadIdx.x;
  const size_t np = V.npad;
  double *s = V.s, *t = V.t;
  double ss = 0.0, st = 0.0;
  for (int i = tid; i < n; i += QEQ_TPB) { ss += s[i]; st += t[i]; }
  qeq_reduce2(ss, st, s_red);
  const double u = ss / st;
  const bool same_place = setup && V.warm;   // the atoms stand where the newest stored solution was taken: it is replaced, not pushed down
  for (int i = tid; i < n; i += QEQ_TPB) {
    V.q[i] = s[i] - u * t[i];
    double *sh = V.s_hist, *th = V.t_hist;
    if (!same_place) {
      sh[3 * np + i] = sh[2 * np + i]; sh[2 * np + i] = sh[np + i]; sh[np + i] = sh[i];
      th[2 * np + i] = th[np + i]; th[np + i] = th[i];
    }
    sh[i] = s[i];
    th[i] = t[i];
  }
  // what the launched sweeps of this solve read: every stored entry of the replica's rows once per sweep it took part in
  // (the launches counted by k_rx_qeq_update, plus the first product H x0)
  double nent = 0.0, nrow = 0.0;
  for (int i = tid; i < n; i += QEQ_TPB) { nent += (double)V.hlen[i]; nrow += 1.0; }
  qeq_reduce2(nent, nrow, s_red);
  if (tid == 0) {
    const long long sweeps = (long long)(V.qstat[0] - V.qstat[4]) + 1;   // k_rx_qeq_update counted the launched iterations; plus the first product
    V.sweep_acc[0] += sweeps * (long long)(nent + 0.5);
    V.sweep_acc[1] += sweeps * (long long)n;
    const int total = V.qstat[0] + it;
    V.qstat[0] = total;
    V.qstat[1] += 1;
    const int mine = total - V.qstat[4];
    V.qstat[4] = total;
    // the first solves of a run that starts from an empty history take longer: their own record
    const int rec = (V.qstat[1] <= RX_QEQ_COLD) ? 5 : 2;
    if (mine > V.qstat[rec]) V.qstat[rec] = mine;
    if (it > 0) V.qstat[3] += 1;
    if (unconverged) atomicOr(V.overflow, 4);
  }
}
// One workgroup per replica after `done` iterations of the launches above: a replica that has not converged yet goes on here
// with the same recurrences on the same arrays (the result does not depend on how many iterations were issued as launches, only
// the summation order of the scalar products differs), then q = s - (sum s / sum t) t and the history.
__global__ __launch_bounds__(QEQ_TPB) void k_rx_qeq_finish(const SimDev *sims, const RxView *views, const RxParams *__restrict__ P, double tol, int done, int maxiter, int setup) {
  const RxView V = views[blockIdx.x];
  (void)sims;
  __shared__ double s_red[32];
  const int n = V.n, tid = threadIdx.x;
  const size_t np = V.npad;
  double *s = V.s, *t = V.t;
  const QeqScal Q = qeq_scalars(V, done, tol);
  bool run_s = Q.run[0], run_t = Q.run[1];
  int it = 0;
  if (run_s || run_t) {
    double2 *r = qeq_rbuf(V, done & 1), *d = (double2 *)(V.qwork + 2 * np), *q = (double2 *)(V.qwork + 4 * np), *z = (double2 *)(V.qwork + 6 * np);
    double sig_s = Q.sig[0], sig_t = Q.sig[1], prev_s = 1.0, prev_t = 1.0;
    if (done > 0) qeq_fold(V.qpart + 2 * RX_QNV(V.npad) * ((done - 1) & 1), (n + QEQ_UT - 1) / QEQ_UT, prev_s, prev_t);
    for (; done + it < maxiter && (run_s || run_t); it++) {
      const double be_s = (done + it > 0) ? sig_s / prev_s : 0.0, be_t = (done + it > 0) ? sig_t / prev_t : 0.0;
      double dq_s = 0.0, dq_t = 0.0;
      // (a wave per row, lanes over its entries: the matrix is row-major)
      for (int i = tid >> 6; i < n; i += QEQ_TPB / 64) {
        const int len = V.hlen[i], lane = tid & 63;
        double ys = 0.0, yt = 0.0;
        for (int c0 = 0; c0 < len; c0 += 64) {
          const bool on = c0 + lane < len;
          const size_t o = (size_t)i * V.maxnb + (on ? c0 + lane : 0);
          int col;
          double hv;
          if (V.hpk) hv = rx_hunpack(on ? V.hpk[o] : 0ull, &col);
          else { col = on ? V.hcol32[o] : 0; hv = on ? V.hval[o] : 0.0; }
          const double2 zj = z[col];
          ys = fma(hv, zj.x, ys);
          yt = fma(hv, zj.y, yt);
        }
        ys = wave_sum(ys); yt = wave_sum(yt);
        if (lane == 0) {
          const double eta = P->sbp[V.rtype[i]].eta;
          const double2 zi = z[i];
          ys = fma(eta, zi.x, ys); yt = fma(eta, zi.y, yt);
          const bool first = done + it == 0;   // (d = z, q = y: the arrays hold what the solve before left)
          double2 di = first ? make_double2(0.0, 0.0) : d[i], qi = first ? make_double2(0.0, 0.0) : q[i];
          if (run_s) { di.x = fma(be_s, di.x, zi.x); qi.x = fma(be_s, qi.x, ys); dq_s += di.x * qi.x; }
          if (run_t) { di.y = fma(be_t, di.y, zi.y); qi.y = fma(be_t, qi.y, yt); dq_t += di.y * qi.y; }
          d[i] = di; q[i] = qi;
        }
      }
      qeq_reduce2(dq_s, dq_t, s_red);   // its barriers also order the sweep (reads z) before the update (writes z)
      const double al_s = run_s ? sig_s / dq_s : 0.0, al_t = run_t ? sig_t / dq_t : 0.0;
      double sn_s = 0.0, sn_t = 0.0, cv_s = 0.0, cv_t = 0.0;
      for (int i = tid; i < n; i += QEQ_TPB) {
        const double eta = P->sbp[V.rtype[i]].eta;
        const double2 di = d[i], qi = q[i];
        double2 ri = r[i], zi = z[i];
        if (run_s) { s[i] = fma(al_s, di.x, s[i]); ri.x = fma(-al_s, qi.x, ri.x); zi.x = ri.x / eta; }
        if (run_t) { t[i] = fma(al_t, di.y, t[i]); ri.y = fma(-al_t, qi.y, ri.y); zi.y = ri.y / eta; }
        r[i] = ri;
        if (!V.pm_on) { z[i] = zi; sn_s += ri.x * zi.x; sn_t += ri.y * zi.y; }
        cv_s += ri.x * ri.x / eta; cv_t += ri.y * ri.y / eta;
      }
      if (V.pm_on) {
        __syncthreads();   // the residuals of the whole replica are in place: z = M r gathers the neighbours'
        for (int i = tid; i < n; i += QEQ_TPB) {
          const double2 ri = r[i];
          const double2 zn = qeq_pm_apply(V, i, [&](int k) { return r[k]; });
          double2 zi = z[i];
          if (run_s) zi.x = zn.x;
          if (run_t) zi.y = zn.y;
          z[i] = zi;
          sn_s += ri.x * zi.x; sn_t += ri.y * zi.y;
        }
      }
      qeq_reduce2(sn_s, sn_t, s_red);
      if (V.pm_on) qeq_reduce2(cv_s, cv_t, s_red); else { cv_s = sn_s; cv_t = sn_t; }
      if (run_s) { prev_s = sig_s; sig_s = sn_s; run_s = sqrt(cv_s) / Q.bn[0] > tol; }
      if (run_t) { prev_t = sig_t; sig_t = sn_t; run_t = sqrt(cv_t) / Q.bn[1] > tol; }
      __syncthreads();
    }
  }
  qeq_finish_tail(V, setup, it, run_s || run_t, s_red);
}

// ---- the symmetric form of the solve (round 5) ----
// H is symmetric and every pair is OWNED by one of its ends (rx_owns): k_rx_hrow<true> stores each pair once, in its owner's row, and a sweep makes both
// products of an entry -- y_i += h z_j summed over the row by the wave as before, y_j += h z_i through LDS atomics into a copy of the replica's
// vector -- for half the bytes and half the matrix arithmetic of the full rows.  A workgroup's part of y meets the others' in memory (coalesced
// atomics, once per workgroup), so y is complete only when the sweep has ENDED: what followed the product inside the full-row sweep (d, q, d.q)
// moves to the launch after it, which is one workgroup per replica and keeps every scalar of the recurrences itself:
//   k_rx_qeq_sweep_sym  y += H z (both triangles), workgroups of a converged replica leave at once
//   k_rx_qeq_step       y complete: d = z + beta d, q = y + beta q, alpha = sigma / d.q, s += alpha d, r -= alpha q, z = M r, sigma' = r.z, the stop
// The scalars of a replica's solve sit in qpart[0 .. QS_N) (no partial sums in this form).  y: the fifth array of qwork.
enum { QS_SIG = 0, QS_PREV = 2, QS_BN = 4, QS_RUN = 6, QS_N = 8 };
#ifndef RX_SYM_RG
#define RX_SYM_RG 4   /* rows of a wave in flight in the symmetric sweep */
#endif
struct QeqState { double sig[2], prev[2], bn[2]; bool run[2]; };
__device__ __forceinline__ QeqState qeq_state_load(const RxView &V) {
  QeqState S;
  const double RX_G *q = V.qpart;
  S.sig[0] = q[QS_SIG]; S.sig[1] = q[QS_SIG + 1]; S.prev[0] = q[QS_PREV]; S.prev[1] = q[QS_PREV + 1];
  S.bn[0] = q[QS_BN]; S.bn[1] = q[QS_BN + 1]; S.run[0] = q[QS_RUN] != 0.0; S.run[1] = q[QS_RUN + 1] != 0.0;
  return S;
}
__device__ __forceinline__ void qeq_state_store(const RxView &V, const QeqState &S) {
  double RX_G *q = V.qpart;
  q[QS_SIG] = S.sig[0]; q[QS_SIG + 1] = S.sig[1]; q[QS_PREV] = S.prev[0]; q[QS_PREV + 1] = S.prev[1];
  q[QS_BN] = S.bn[0]; q[QS_BN + 1] = S.bn[1]; q[QS_RUN] = S.run[0] ? 1.0 : 0.0; q[QS_RUN + 1] = S.run[1] ? 1.0 : 0.0;
}
__device__ __forceinline__ double *qeq_ybuf(const RxView &V) { return V.qwork + 8 * (size_t)V.npad; }   // [2][npad]: the s system's part, then the t system's
extern __shared__ double2 s_sym[];   // [0, npad): z of the replica; [npad, 2 npad): this workgroup's part of y
// the products of the rows [row0, row1) with NW waves, two rows of a wave at a time (their loads are independent); s_z read, s_y added to
// (s_y: the two systems' parts one after the other, [2][npad] -- the atomics of a chunk's consecutive columns fall into consecutive banks)
template <int NW>
__device__ __forceinline__ void qeq_sym_product(const RxView &V, int row0, int row1, const double2 *s_z, double *s_y) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const GLOBAL_AS unsigned long long *pk = as_global(V.hpk);
  const GLOBAL_AS int *hlen = as_global(V.hownlen);
  const size_t np = V.npad;
  // RG rows of a wave at a time: an owned row is three or four chunks long, so a row's header (its length, a dependent memory round trip before
  // the first entry can be masked) weighs as much as its entries; with four rows in flight the wave waits for half as many of them
  constexpr int RG = RX_SYM_RG;
  for (int ra = row0 + RG * wave; ra < row1; ra += RG * NW) {
    int len[RG], lmax = 0;
    size_t base[RG];
    double2 zi[RG];
    double ps[RG], pt[RG];
#pragma unroll
    for (int q = 0; q < RG; q++) {
      const int row = ra + q;
      const bool live = row < row1;
      len[q] = live ? hlen[row] : 0;   // (wave-uniform)
      base[q] = (size_t)(live ? row : ra) * V.maxnb;
      zi[q] = s_z[live ? row : ra];
      ps[q] = 0.0; pt[q] = 0.0;
    }
#pragma unroll
    for (int q = 0; q < RG; q++) lmax = max(lmax, len[q]);
    unsigned long long bn[RG];
    auto load = [&](int c0) {
      const int c = c0 + lane;
#pragma unroll
      for (int q = 0; q < RG; q++) bn[q] = (c < len[q]) ? pk[base[q] + c] : ~0ull;
    };
    load(0);
    for (int c0 = 0; c0 < lmax; c0 += 64) {
      unsigned long long b[RG];
#pragma unroll
      for (int q = 0; q < RG; q++) b[q] = bn[q];
      load(c0 + 64);
#pragma unroll
      for (int q = 0; q < RG; q++) {
        if (b[q] != ~0ull) {   // (a stored value is positive and finite: no entry has this pattern)
          int j;
          const double h = rx_hunpack(b[q], &j);
          const double2 zj = s_z[j];
          ps[q] = fma(h, zj.x, ps[q]);
          pt[q] = fma(h, zj.y, pt[q]);
          lds_add_f64(&s_y[j], h * zi[q].x);
          lds_add_f64(&s_y[np + j], h * zi[q].y);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < RG; q++) {
      const double a = wave_sum(ps[q]), bsum = wave_sum(pt[q]);
      if (lane == 0 && ra + q < row1) { lds_add_f64(&s_y[ra + q], a); lds_add_f64(&s_y[np + ra + q], bsum); }
    }
  }
}
// it < 0: the first product H x0 of a solve (x0 sits in z)
// (The replica's vector step inside this kernel -- done by the last of its workgroups to have flushed its part of y, found by a ticket behind
// device-scope fences, VERDICT r5 item 4a -- was built and measured in round 6: 578 against 1 357 evaluations/s, 0.39 instead of 0.07 ms per
// launch.  A release or acquire at device scope writes back / invalidates the XCD's L2 on this eight-XCD part.  profiles/r06_o_reax_fused_step_ab.txt)
__global__ __launch_bounds__(RX_KT) void k_rx_qeq_sweep_sym(const RxView *views, int it) {
  const RxView V = views[blockIdx.y];
  const int n = V.n, row0 = blockIdx.x * RX_SWR;
  if (row0 >= n) return;
  if (it >= 0 && V.qpart[QS_RUN] == 0.0 && V.qpart[QS_RUN + 1] == 0.0) return;   // (uniform) both systems have converged
  const size_t np = V.npad;
  double2 *s_z = s_sym;
  double *s_y = (double *)(s_sym + np);
  const GLOBAL_AS dvec2 *z = as_global((const dvec2 *)(V.qwork + 6 * np));
  for (int k0 = threadIdx.x; k0 < n; k0 += 4 * RX_KT) {
    double2 t[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const int k = k0 + u * RX_KT; t[u] = ldg2(z, k < n ? k : n - 1); }
#pragma unroll
    for (int u = 0; u < 4; u++) { const int k = k0 + u * RX_KT; if (k < n) { s_z[k] = t[u]; s_y[k] = 0.0; s_y[np + k] = 0.0; } }
  }
  __syncthreads();
  qeq_sym_product<RX_KS>(V, row0, min(row0 + RX_SWR, n), s_z, s_y);
  __syncthreads();
  double *yg = qeq_ybuf(V);   // [2][npad] as well
  for (int k = threadIdx.x; k < n; k += RX_KT) {
    const double a = s_y[k], b = s_y[np + k];
    if (a != 0.0) atomicAdd(&yg[k], a);
    if (b != 0.0) atomicAdd(&yg[np + k], b);
  }
}
// One iteration's vector work for a replica with NT threads of ONE workgroup; y (the finished product H z without the diagonal; read, and
// zeroed for the next sweep if `zero_y`) in memory or in LDS.  first: d = z, q = y.  The recurrences and their order are k_rx_qeq_finish's.
template <int NT>
__device__ __forceinline__ void qeq_sym_step(const RxView &V, const RxParams *__restrict__ P, double tol, bool first, QeqState &S, double *y, bool zero_y, double *s_red) {
  const int n = V.n, tid = threadIdx.x;
  const size_t np = V.npad;
  double *s = V.s, *t = V.t;
  double2 *r = qeq_rbuf(V, 0), *d = (double2 *)(V.qwork + 2 * np), *q = (double2 *)(V.qwork + 4 * np), *z = (double2 *)(V.qwork + 6 * np);
  const bool run_s = S.run[0], run_t = S.run[1];
  const double be_s = first ? 0.0 : S.sig[0] / S.prev[0], be_t = first ? 0.0 : S.sig[1] / S.prev[1];
  double dq_s = 0.0, dq_t = 0.0;
  for (int i = tid; i < n; i += NT) {
    const double eta = P->sbp[V.rtype[i]].eta;
    const double2 zi = z[i];
    const double y0 = y[i], y1 = y[np + i];
    if (zero_y) { y[i] = 0.0; y[np + i] = 0.0; }
    const double ys = fma(eta, zi.x, y0), yt = fma(eta, zi.y, y1);
    double2 di = first ? make_double2(0.0, 0.0) : d[i], qi = first ? make_double2(0.0, 0.0) : q[i];
    if (run_s) { di.x = fma(be_s, di.x, zi.x); qi.x = fma(be_s, qi.x, ys); dq_s += di.x * qi.x; }
    if (run_t) { di.y = fma(be_t, di.y, zi.y); qi.y = fma(be_t, qi.y, yt); dq_t += di.y * qi.y; }
    d[i] = di; q[i] = qi;
  }
  qeq_reduce2t<NT>(dq_s, dq_t, s_red);
  const double al_s = run_s ? S.sig[0] / dq_s : 0.0, al_t = run_t ? S.sig[1] / dq_t : 0.0;
  double sn_s = 0.0, sn_t = 0.0, cv_s = 0.0, cv_t = 0.0;
  for (int i = tid; i < n; i += NT) {
    const double eta = P->sbp[V.rtype[i]].eta;
    const double2 di = d[i], qi = q[i];
    double2 ri = r[i], zi = z[i];
    if (run_s) { s[i] = fma(al_s, di.x, s[i]); ri.x = fma(-al_s, qi.x, ri.x); zi.x = ri.x / eta; }
    if (run_t) { t[i] = fma(al_t, di.y, t[i]); ri.y = fma(-al_t, qi.y, ri.y); zi.y = ri.y / eta; }
    r[i] = ri;
    if (!V.pm_on) { z[i] = zi; sn_s += ri.x * zi.x; sn_t += ri.y * zi.y; }
    cv_s += ri.x * ri.x / eta; cv_t += ri.y * ri.y / eta;
  }
  if (V.pm_on) {
    __syncthreads();   // the residuals of the whole replica are in place: z = M r gathers the neighbours'
    for (int i = tid; i < n; i += NT) {
      const double2 ri = r[i];
      const double2 zn = qeq_pm_apply(V, i, [&](int k) { return r[k]; });
      double2 zi = z[i];
      if (run_s) zi.x = zn.x;
      if (run_t) zi.y = zn.y;
      z[i] = zi;
      sn_s += ri.x * zi.x; sn_t += ri.y * zi.y;
    }
  }
  qeq_reduce2t<NT>(sn_s, sn_t, s_red);
  if (V.pm_on) qeq_reduce2t<NT>(cv_s, cv_t, s_red); else { cv_s = sn_s; cv_t = sn_t; }
  if (run_s) { S.prev[0] = S.sig[0]; S.sig[0] = sn_s; S.run[0] = sqrt(cv_s) / S.bn[0] > tol; }
  if (run_t) { S.prev[1] = S.sig[1]; S.sig[1] = sn_t; S.run[1] = sqrt(cv_t) / S.bn[1] > tol; }
  __syncthreads();
}
// The launch between two sweeps (one workgroup per replica).  it < 0: r = b - H x0 from the first product (x0 sits in z), z = M r, the scalars
// of the solve; it >= 0: iteration it of the conjugate gradients (qeq_sym_step).  A converged replica leaves at once.
// (RX_STEP_TPB threads: the launch sits between two sweeps of a serial chain on a chip that other streams keep full, and a workgroup of 1 024 threads
// needs four free wave slots on EVERY SIMD of one CU before it starts)
#ifndef RX_STEP_TPB
#define RX_STEP_TPB 1024
#endif
__global__ __launch_bounds__(RX_STEP_TPB) void k_rx_qeq_step(const RxView *views, const RxParams *__restrict__ P, double tol, int it) {
  const RxView V = views[blockIdx.x];
  __shared__ double s_red[32];
  const int n = V.n, tid = threadIdx.x;
  const size_t np = V.npad;
  double *y = qeq_ybuf(V);
  if (it >= 0) {
    QeqState S = qeq_state_load(V);
    if (!S.run[0] && !S.run[1]) return;
    __syncthreads();   // (every thread has read the state before thread 0 stores the next one)
    qeq_sym_step<RX_STEP_TPB>(V, P, tol, it == 0, S, y, true, s_red);
    if (tid == 0) { qeq_state_store(V, S); V.qstat[0] += 1; }
    return;
  }
  double2 *r = qeq_rbuf(V, 0), *z = (double2 *)(V.qwork + 6 * np);
  double p0 = 0.0, p1 = 0.0, b0 = 0.0, b1 = 0.0, c0 = 0.0, c1 = 0.0;
  for (int i = tid; i < n; i += RX_STEP_TPB) {
    const int ti = V.rtype[i];
    const double eta = P->sbp[ti].eta, chi = P->sbp[ti].chi;
    const double2 x0 = z[i];
    const double y0 = y[i], y1 = y[np + i];
    y[i] = 0.0; y[np + i] = 0.0;
    const double2 ri = make_double2(-chi - fma(eta, x0.x, y0), -1.0 - fma(eta, x0.y, y1));
    r[i] = ri;
    b0 += chi * chi; b1 += 1.0;
    c0 += ri.x * ri.x / eta; c1 += ri.y * ri.y / eta;
    if (!V.pm_on) { const double2 zi = make_double2(ri.x / eta, ri.y / eta); z[i] = zi; p0 += ri.x * zi.x; p1 += ri.y * zi.y; }
  }
  if (V.pm_on) {
    __syncthreads();
    for (int i = tid; i < n; i += RX_STEP_TPB) {
      const double2 ri = r[i];
      const double2 zi = qeq_pm_apply(V, i, [&](int k) { return r[k]; });
      z[i] = zi;
      p0 += ri.x * zi.x; p1 += ri.y * zi.y;
    }
  }
  qeq_reduce2t<RX_STEP_TPB>(p0, p1, s_red);
  qeq_reduce2t<RX_STEP_TPB>(b0, b1, s_red);
  qeq_reduce2t<RX_STEP_TPB>(c0, c1, s_red);
  if (tid == 0) {
    QeqState S;
    S.sig[0] = p0; S.sig[1] = p1; S.prev[0] = 1.0; S.prev[1] = 1.0; S.bn[0] = sqrt(b0); S.bn[1] = sqrt(b1);
    S.run[0] = sqrt(c0) / S.bn[0] > tol; S.run[1] = sqrt(c1) / S.bn[1] > tol;
    qeq_state_store(V, S);
  }
}
// after `done` launched iterations: a replica that has not converged goes on in its one workgroup -- product into LDS, then the same step -- and
// every replica ends its solve (charges, history, statistics)
__global__ __launch_bounds__(QEQ_TPB) void k_rx_qeq_finish_sym(const RxView *views, const RxParams *__restrict__ P, double tol, int done, int maxiter, int setup) {
  const RxView V = views[blockIdx.x];
  __shared__ double s_red[32];
  const int n = V.n, tid = threadIdx.x;
  const size_t np = V.npad;
  QeqState S = qeq_state_load(V);
  int it = 0;
  double2 *s_z = s_sym;
  double *s_y = (double *)(s_sym + np);
  const double2 *z = (const double2 *)(V.qwork + 6 * np);
  for (; done + it < maxiter && (S.run[0] || S.run[1]); it++) {
    for (int k = tid; k < n; k += QEQ_TPB) { s_z[k] = z[k]; s_y[k] = 0.0; s_y[np + k] = 0.0; }
    __syncthreads();
    qeq_sym_product<QEQ_TPB / 64>(V, 0, n, s_z, s_y);
    __syncthreads();
    qeq_sym_step<QEQ_TPB>(V, P, tol, done + it == 0, S, s_y, false, s_red);
  }
  qeq_finish_tail(V, setup, it, S.run[0] || S.run[1], s_red);
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

// the same for a workgroup of NW waves: one set of atomics per workgroup (same-address FP64 atomics serialise at the memory side)
template <int NW>
__device__ __forceinline__ void rx_flush_block(double (&e)[RX_NPART], double (&w)[6], const RxView &V, SimScalars &sc, int vpart) {
  __shared__ double s_fl[NW][RX_NPART + 6];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < RX_NPART; k++) {
    const double s = wave_sum(e[k]);
    if (lane == 0) s_fl[wave][k] = s;
  }
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const double s = wave_sum(w[k]);
    if (lane == 0) s_fl[wave][RX_NPART + k] = s;
  }
  __syncthreads();
  if (threadIdx.x < RX_NPART + 6) {
    double s = 0.0;
#pragma unroll
    for (int m = 0; m < NW; m++) s += s_fl[m][threadIdx.x];
    if (s != 0.0) atomicAdd(threadIdx.x < RX_NPART ? &V.eparts[threadIdx.x] : &sc.vir[vpart * 6 + threadIdx.x - RX_NPART], s);
  }
}

// Uncorrected bond orders (rx_bonds_prime): a workgroup of 4 waves owns 32 consecutive atoms, a wave eight of them, TWO at a time -- a half wave
// per near row with its lanes over the entries (row-major copy of the near rows).  Since the near rows hold a type pair's candidates within reach of its
// bond order (round 5) a row has some thirty of them: a wave per row left half its lanes idle.  The few entries that are bonds are compacted with a
// ballot (each half's own 32 bits) into the atom's bond row in list order, the sum of their orders by a half-wave sum.
__global__ __launch_bounds__(TPB) void k_rx_bonds(const RxView *views, const RxParams *__restrict__ P) {
  const RxView V = views[blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, hl = lane & 31;
  const size_t np = V.npad, plane = (size_t)V.maxbd * np;
  // the type tables in LDS: the parameters of a pair are two dependent global loads away otherwise
  __shared__ RxSbp s_sbp[RX_MAXT];
  __shared__ RxTbp s_tbp[RX_MAXT * RX_MAXT];
  for (int k = threadIdx.x; k < (int)(sizeof(s_sbp) / 8); k += TPB) ((double *)s_sbp)[k] = ((const double *)P->sbp)[k];
  for (int k = threadIdx.x; k < (int)(sizeof(s_tbp) / 8); k += TPB) ((double *)s_tbp)[k] = ((const double *)P->tbp)[k];
  __shared__ double s_sh[3 * RX_NSHIFT];
  rx_shift_table(V, s_sh);
  __syncthreads();
  const unsigned below = (1u << hl) - 1u;
  for (int r = 0; r < 8; r += 2) {
    const int i0 = blockIdx.x * (8 * (TPB / 64)) + wave * 8 + r;
    if (i0 >= V.n) return;   // (wave-uniform; no barrier below)
    const bool live = i0 + half < V.n;
    const int i = live ? i0 + half : i0;
    const int cnt = live ? V.nbn_cnt[i] : 0;
    const int cmax = max(__builtin_amdgcn_readlane(cnt, 0), __builtin_amdgcn_readlane(cnt, 32));
    const size_t base = (size_t)i * V.maxnbn;
    int nb = 0;
    double sum = 0.0;
    // the row walk as in k_rx_hrow: the stores of the chunk before, the requests of the chunks after (partner records one chunk ahead, near-row
    // entries two), the arithmetic of this one
    const int ti = V.rtype[i];
    const double xi0 = V.x[3 * i], xi1 = V.x[3 * i + 1], xi2 = V.x[3 * i + 2];
    auto load_ent = [&](int k0) -> int { const int k = k0 + hl; return (k < cnt) ? V.nbnT[base + k] : -1; };
    int e1 = load_ent(0), e2 = load_ent(32);
    double p0, p1, p2;
    int tjn;
    { const int j = (e1 >= 0) ? (e1 & RX_JMASK) : 0; p0 = V.x[3 * j]; p1 = V.x[3 * j + 1]; p2 = V.x[3 * j + 2]; tjn = V.rtype[j]; }
    asm volatile("" : : "v"(p0), "v"(p1), "v"(p2), "v"(tjn), "v"(e2), "v"(xi0), "v"(xi1), "v"(xi2), "v"(ti));   // (waited for here, not inside the loop)
    int ok_prev = 0, e_prev = 0;
    double bo_p = 0, bp_p = 0, bpp_p = 0, rr_p = 0, cs_p = 0, cp_p = 0, cpp_p = 0;
    auto flush = [&]() __attribute__((always_inline)) {
      const unsigned long long m = __ballot(ok_prev);
      const unsigned mh = half ? (unsigned)(m >> 32) : (unsigned)m;
      if (ok_prev) {
        const int pos = nb + __popc(mh & below);
        if (pos < V.maxbd) {
          const size_t o = (size_t)pos * np + i;
          V.bd[o] = e_prev;
          V.bd_bop[o] = bo_p - P->bo_cut;
          V.bd_bop[plane + o] = bp_p;
          V.bd_bop[2 * plane + o] = bpp_p;
          V.bd_bop[3 * plane + o] = rr_p;
          V.bd_c[o] = cs_p;
          V.bd_c[plane + o] = cp_p;
          V.bd_c[2 * plane + o] = cpp_p;
          sum += bo_p - P->bo_cut;
        }
      }
      nb += __popc(mh);
    };
    for (int k0 = 0; k0 < cmax; k0 += 32) {
      const int ent = e1, tj = tjn;
      const double q0 = p0, q1 = p1, q2 = p2;
      e1 = e2;
      flush();
      { const int j = (e1 >= 0) ? (e1 & RX_JMASK) : 0; p0 = V.x[3 * j]; p1 = V.x[3 * j + 1]; p2 = V.x[3 * j + 2]; tjn = V.rtype[j]; }
      e2 = load_ent(k0 + 64);
      ok_prev = 0;
      if (ent >= 0) {
        const double *sh = s_sh + 3 * ((ent >> 24) & 0x7F);
        const double d0 = q0 - xi0 + sh[0], d1 = q1 - xi1 + sh[1], d2 = q2 - xi2 + sh[2];
        ok_prev = rx_bond_prime_pair(P, s_sbp, s_tbp, ti, tj, d0 * d0 + d1 * d1 + d2 * d2, &bo_p, &bp_p, &bpp_p, &rr_p, &cs_p, &cp_p, &cpp_p);
      }
      e_prev = ent;
    }
    flush();
    // the half's sum: the rows of 16 lanes, then row 0 into row 1 and row 2 into row 3 (the last lane of each half holds it)
    sum = row_sum(sum);
    sum += dpp_read0<0x142, 0xA>(sum);
    if (hl == 31 && live) {
      if (nb > V.maxbd) { atomicOr(V.overflow, 2); nb = V.maxbd; }
      V.bd_cnt[i] = nb;
      V.deltap[i] = sum - P->sbp[ti].valency;
    }
  }
}
__global__ __launch_bounds__(TPB) void k_rx_rev(const RxView *views) {
  const RxView V = views[blockIdx.y];
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i < V.n) rx_bonds_rev(&V, i);
}
// Corrected bond orders (rx_bonds_corrected) with FOUR lanes per atom, one per bond slot (a carbon has four bonds): a lane-per-atom pass is a chain
// of dependent round trips per bond -- entry, partner's Delta', the correction's exponentials -- and nothing else (2 % of the issue slots); a wave takes
// 16 atoms, row r of its 16 lanes the bond slots r, r + 4, ...; the entry-major rows make every row's loads 16 consecutive words.
__global__ __launch_bounds__(TPB) void k_rx_corr(const RxView *views, const RxParams *__restrict__ P) {
  const RxView V = views[blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, a = lane & 15, k4 = lane >> 4;
  const int i = blockIdx.x * (TPB / 4) + wave * 16 + a;
  const bool live = i < V.n;
  const int ii = live ? i : 0;
  const int np = V.npad, ti = V.rtype[ii], cnt = live ? V.bd_cnt[ii] : 0;
  const size_t plane = (size_t)V.maxbd * np;
  const double Di = V.deltap[ii];
  double sum = 0.0;
  for (int k = k4; k < cnt; k += 4) {
    const size_t o = (size_t)k * np + i;
    const int j = V.bd[o] & RX_JMASK;
    const double B = V.bd_bop[o], Bp = V.bd_bop[plane + o], Bpp = V.bd_bop[2 * plane + o];
    RxCorr c;
    rx_corr(P, ti, V.rtype[j], Di, V.deltap[j], B, &c);
    const double A0 = c.Y * c.X, A1 = A0 * c.Y;
    double bo = B * A0, bp = Bp * A1, bpp = Bpp * A1;
    if (bo < 1e-10) bo = 0.0;
    if (bp < 1e-10) bp = 0.0;
    if (bpp < 1e-10) bpp = 0.0;
    V.bd_bo[o] = bo; V.bd_bo[plane + o] = bp; V.bd_bo[2 * plane + o] = bpp;
    V.bd_g[o] = 0.0; V.bd_g[plane + o] = 0.0; V.bd_g[2 * plane + o] = 0.0;
    sum += bo;
  }
  // the atom's four lanes sit in the four rows of the wave (the order of this sum differs from the serial pass for atoms of more than two bonds)
  sum += __shfl_xor(sum, 16);
  sum += __shfl_xor(sum, 32);
  if (k4 == 0 && live) {
    V.total_bo[i] = sum;
    V.cd_delta[i] = 0.0;
    V.hd[i] = 0.0;
    V.f[3 * i] = 0.0; V.f[3 * i + 1] = 0.0; V.f[3 * i + 2] = 0.0;
  }
}
// pass: 0 atom terms, 3 hydrogen bonds (angles: k_rx_angles, torsions: k_rx_torsions, non-bonded: k_rx_nonbonded_once / k_rx_nonbonded)
template <int PASS>
__global__ __launch_bounds__(RX_TPB, RX_OCC) void k_rx_terms(const SimDev *sims, const RxView *views, const RxParams *__restrict__ P) {
  const RxView V = views[blockIdx.y];
  const int i = blockIdx.x * RX_TPB + threadIdx.x;
  double e[RX_NPART], w[6];
#pragma unroll
  for (int k = 0; k < RX_NPART; k++) e[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 6; k++) w[k] = 0.0;
  if (i < V.n) {
    if (PASS == 0) rx_atom_terms(P, &V, i, e);
    if (PASS == 3) rx_hbond_terms(P, &V, i, e, w);
  }
  rx_flush(e, w, V, *sims[blockIdx.y].sc, PASS == 4 ? P_LJ : (PASS == 1 ? P_ANGLE : (PASS == 2 ? P_DIHEDRAL : (PASS == 3 ? P_IMPROPER : P_BOND))));
}
// The torsion pass with a lane per WORK ITEM (first leg j-i, central bond j-k; reax/rx_core.h rx_torsion_item) instead of a lane per atom:
// with a lane per atom only the atoms that own a central bond work -- a third of the lanes of a hydrocarbon, each walking its nine
// torsions one after the other at 256 registers.  A workgroup takes RX_TORS_ATOMS atoms, lists their items in LDS (an LDS counter; the
// order of the list, and with it the order of the atomic sums, varies from run to run as every atomic sum here does) and then
// puts its lanes over the list.  Items beyond the list's capacity (a denser system than any tested) are done where they are found.
#define RX_TORS_ATOMS 256
#define RX_TORS_CAP 2048
// LACC: the forces and dE/dDelta sums of the items meet in LDS tables of the replica (s_acc: [3][npad] + [npad]; replicas of up to RX_NB1_MAXPAD
// atoms) and reach the work set once per workgroup (rx_core.h, rx_add_f): 20 of an item's 26 device-wide atomics, which bounded the pass.
extern __shared__ double s_acc[];
__device__ __forceinline__ void rx_acc_zero(int np) {
  for (int k = threadIdx.x; k < 4 * np; k += RX_TORS_ATOMS) s_acc[k] = 0.0;
}
__device__ __forceinline__ void rx_acc_flush(const RxView &V) {
  const int n = V.n, np = V.npad;
  for (int k = threadIdx.x; k < 3 * n; k += RX_TORS_ATOMS) {
    const int a = k / 3, c = k - 3 * a;
    const double v = s_acc[(size_t)c * np + a];
    if (v != 0.0) atomicAdd(&V.f[k], v);
  }
  for (int a = threadIdx.x; a < n; a += RX_TORS_ATOMS) {
    const double v = s_acc[(size_t)3 * np + a];
    if (v != 0.0) atomicAdd(&V.cd_delta[a], v);
  }
}
template <bool LACC>
__global__ __launch_bounds__(RX_TORS_ATOMS, RX_OCC) void k_rx_torsions(const SimDev *sims, const RxView *views, const RxParams *__restrict__ P, int cap) {
  const RxView V = views[blockIdx.y];
  if ((int)(blockIdx.x * RX_TORS_ATOMS) >= V.n) return;
  __shared__ int s_items[RX_TORS_CAP];
  __shared__ int s_n;
  double *lf = LACC ? s_acc : nullptr, *lcd = LACC ? s_acc + 3 * (size_t)V.npad : nullptr;
  if (threadIdx.x == 0) s_n = 0;
  if (LACC) rx_acc_zero(V.npad);
  __syncthreads();
  double e[RX_NPART], w[6];
#pragma unroll
  for (int k = 0; k < RX_NPART; k++) e[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 6; k++) w[k] = 0.0;
  const int j0 = blockIdx.x * RX_TORS_ATOMS, j = j0 + threadIdx.x;
  if (j < V.n) {
    const int cnt = V.bd_cnt[j];
    for (int ak = 0; ak < cnt; ak++)
      for (int ai = 0; ai < cnt; ai++)
        if (rx_torsion_item_valid(&V, j, ak, ai)) {
          const int pos = atomicAdd(&s_n, 1);
          if (pos < cap) s_items[pos] = (int)(((unsigned)threadIdx.x << 24) | ((unsigned)ak << 12) | (unsigned)ai);   // (rows hold far fewer than 4 096 bonds)
          else rx_torsion_item<LACC>(P, &V, j, ak, ai, e, w, lf, lcd);
        }
  }
  __syncthreads();
  const int nitems = min(s_n, cap);
  for (int it = threadIdx.x; it < nitems; it += RX_TORS_ATOMS) {
    const unsigned c = (unsigned)s_items[it];
    rx_torsion_item<LACC>(P, &V, j0 + (int)(c >> 24), (int)((c >> 12) & 0xFFF), (int)(c & 0xFFF), e, w, lf, lcd);
  }
  if (LACC) {
    __syncthreads();
    rx_acc_flush(V);
  }
  rx_flush(e, w, V, *sims[blockIdx.y].sc, P_DIHEDRAL);
}
// The valence-angle pass the same way: a lane per ANGLE (reax/rx_core.h rx_angle_item).  What an atom's angles share (rx_angle_pre) is
// computed by the atom's thread first and read from LDS by the items; what they sum for the atom (force on it, dE/dDelta, dE/dSBO)
// meets in LDS accumulators and is fed back by the atom's thread (rx_angle_post) after a barrier.
template <bool LACC>
__global__ __launch_bounds__(RX_TORS_ATOMS, RX_OCC) void k_rx_angles(const SimDev *sims, const RxView *views, const RxParams *__restrict__ P, int cap) {
  const RxView V = views[blockIdx.y];
  if ((int)(blockIdx.x * RX_TORS_ATOMS) >= V.n) return;
  __shared__ int s_items[RX_TORS_CAP];
  __shared__ int s_n;
  __shared__ double s_sbo[2][RX_TORS_ATOMS];   // SBO2, CSBO2 of the block's atoms
  __shared__ double s_sum[5][RX_TORS_ATOMS];   // cdd, f[3], dE/dSBO
  double *lf = LACC ? s_acc : nullptr, *lcd = LACC ? s_acc + 3 * (size_t)V.npad : nullptr;
  if (LACC) rx_acc_zero(V.npad);
  if (threadIdx.x == 0) s_n = 0;
#pragma unroll
  for (int k = 0; k < 5; k++) s_sum[k][threadIdx.x] = 0.0;
  __syncthreads();
  double e[RX_NPART], w[6];
#pragma unroll
  for (int k = 0; k < RX_NPART; k++) e[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 6; k++) w[k] = 0.0;
  const int j0 = blockIdx.x * RX_TORS_ATOMS, j = j0 + threadIdx.x;
  RxAnglePre A;
  const bool has = j < V.n && rx_angle_pre(P, &V, j, &A);
  if (has) {
    s_sbo[0][threadIdx.x] = A.SBO2; s_sbo[1][threadIdx.x] = A.CSBO2;
    const int cnt = V.bd_cnt[j];
    for (int ai = 0; ai < cnt; ai++)
      for (int ak = ai + 1; ak < cnt; ak++)
        if (rx_angle_item_valid(&V, j, ai, ak)) {
          const int pos = atomicAdd(&s_n, 1);
          if (pos < cap) s_items[pos] = (int)(((unsigned)threadIdx.x << 24) | ((unsigned)ai << 12) | (unsigned)ak);
          else {   // (a denser system than any tested: done where it is found)
            RxAngleSum S = {0.0, {0.0, 0.0, 0.0}, 0.0};
            rx_angle_item<LACC>(P, &V, j, ai, ak, A.SBO2, A.CSBO2, &S, e, w, lf, lcd);
            lds_add_f64(&s_sum[0][threadIdx.x], S.cdd); lds_add_f64(&s_sum[1][threadIdx.x], S.f[0]); lds_add_f64(&s_sum[2][threadIdx.x], S.f[1]);
            lds_add_f64(&s_sum[3][threadIdx.x], S.f[2]); lds_add_f64(&s_sum[4][threadIdx.x], S.dE_dSBO);
          }
        }
  }
  __syncthreads();
  const int nitems = min(s_n, cap);
  for (int it = threadIdx.x; it < nitems; it += RX_TORS_ATOMS) {
    const unsigned c = (unsigned)s_items[it];
    const int jl = (int)(c >> 24);
    RxAngleSum S = {0.0, {0.0, 0.0, 0.0}, 0.0};
    rx_angle_item<LACC>(P, &V, j0 + jl, (int)((c >> 12) & 0xFFF), (int)(c & 0xFFF), s_sbo[0][jl], s_sbo[1][jl], &S, e, w, lf, lcd);
    lds_add_f64(&s_sum[0][jl], S.cdd); lds_add_f64(&s_sum[1][jl], S.f[0]); lds_add_f64(&s_sum[2][jl], S.f[1]);
    lds_add_f64(&s_sum[3][jl], S.f[2]); lds_add_f64(&s_sum[4][jl], S.dE_dSBO);
  }
  __syncthreads();
  if (has) {
    RxAngleSum S;
    S.cdd = s_sum[0][threadIdx.x]; S.f[0] = s_sum[1][threadIdx.x]; S.f[1] = s_sum[2][threadIdx.x]; S.f[2] = s_sum[3][threadIdx.x]; S.dE_dSBO = s_sum[4][threadIdx.x];
    rx_angle_post<LACC>(&V, j, &A, &S, lf, lcd);
  }
  if (LACC) {
    __syncthreads();
    rx_acc_flush(V);
  }
  rx_flush(e, w, V, *sims[blockIdx.y].sc, P_ANGLE);
}
// tapered van der Waals + shielded Coulomb over the full neighbour rows, RX_KS waves per row
__global__ __launch_bounds__(RX_KT) void k_rx_nonbonded(const SimDev *sims, const RxView *views, const RxParams *__restrict__ P) {
  const RxView V = views[blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  double e[RX_NPART], w[6], fi[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < RX_NPART; k++) e[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 6; k++) w[k] = 0.0;
  if (i < V.n) rx_nonbonded_part(P, &V, i, wave, RX_KS, fi, e, w);
  __shared__ double s_f[3][RX_KS][64];
  s_f[0][wave][lane] = fi[0]; s_f[1][wave][lane] = fi[1]; s_f[2][wave][lane] = fi[2];
  __syncthreads();
  if (wave == 0 && i < V.n) {
    double f0 = 0.0, f1 = 0.0, f2 = 0.0;
#pragma unroll
    for (int m = 0; m < RX_KS; m++) { f0 += s_f[0][m][lane]; f1 += s_f[1][m][lane]; f2 += s_f[2][m][lane]; }
    atomicAdd(&V.f[3 * i], f0); atomicAdd(&V.f[3 * i + 1], f1); atomicAdd(&V.f[3 * i + 2], f2);
    const double qi = V.q[i];
    const int ti = V.rtype[i];
    e[RX_E_POL] += RX_KCALPMOL_TO_EV * (P->sbp[ti].chi * qi + 0.5 * P->sbp[ti].eta * qi * qi);
  }
  rx_flush_block<RX_KS>(e, w, V, *sims[blockIdx.y].sc, P_LJ);
}
// The same terms with every pair evaluated ONCE: a workgroup owns 64 consecutive atoms, wave w their rows 8 w .. 8 w + 7 with its lanes over
// the row's OWNED pairs (RxView::hown, compacted by k_rx_hrow: no skin entries, no lane idling behind the ownership test); the
// force on the partner goes into a table of the replica's forces in LDS (ds_add_f64; replicas of up to RX_NB1_MAXPAD atoms), flushed
// once per workgroup.  Half the transcendental arithmetic of the both-ends form (three exp, two log, a cube root per pair).
#define RX_NB1_MAXPAD 6000
extern __shared__ double s_nbf[];   // [3][npad]
#ifndef RX_NB1_ROWS
#define RX_NB1_ROWS 64   /* rows of a workgroup (a multiple of 64) */
#endif
__global__ __launch_bounds__(RX_KT) void k_rx_nonbonded_once(const SimDev *sims, const RxView *views, const RxParams *__restrict__ P) {
  const RxView V = views[blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = V.n;
  if ((int)(blockIdx.x * RX_NB1_ROWS) >= n) return;
  const size_t np = V.npad;
  __shared__ RxTbp s_tbp[RX_MAXT * RX_MAXT];   // the pair parameters in LDS (a dependent global load less per pair)
  for (int k = threadIdx.x; k < (int)(sizeof(s_tbp) / 8); k += RX_KT) ((double *)s_tbp)[k] = ((const double *)P->tbp)[k];
  for (int k = threadIdx.x; k < 3 * (int)np; k += RX_KT) s_nbf[k] = 0.0;
  __shared__ double s_sh[3 * RX_NSHIFT];
  rx_shift_table(V, s_sh);
  __syncthreads();
  double e[RX_NPART], w[6];
#pragma unroll
  for (int k = 0; k < RX_NPART; k++) e[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 6; k++) w[k] = 0.0;
  for (int r = 0; r < RX_NB1_ROWS / RX_KS; r++) {
    const int i = blockIdx.x * RX_NB1_ROWS + wave * (RX_NB1_ROWS / RX_KS) + r;
    if (i >= n) break;   // (wave-uniform)
    const int len = V.hownlen[i], ti = V.rtype[i];
    const size_t base = (size_t)i * V.maxnb;
    const double qi = RX_C_ELE * V.q[i];
    const double xi0 = V.x[3 * i], xi1 = V.x[3 * i + 1], xi2 = V.x[3 * i + 2];
    double f0 = 0.0, f1 = 0.0, f2 = 0.0;
    // (the same pipeline as the matrix build: entries two chunks ahead, the partner's record one chunk ahead)
    auto load_ent = [&](int c0) -> int { const int c = c0 + lane; return (c < len) ? V.hown[base + c] : -1; };
    int e1 = load_ent(0), e2 = load_ent(64);
    double p0, p1, p2, pq;
    int ptj;
    { const int j = (e1 >= 0) ? (e1 & RX_JMASK) : 0; p0 = V.x[3 * j]; p1 = V.x[3 * j + 1]; p2 = V.x[3 * j + 2]; pq = V.q[j]; ptj = V.rtype[j]; }
    for (int c0 = 0; c0 < len; c0 += 64) {
      const int ent = e1, tj = ptj;
      const double x0 = p0, x1 = p1, x2 = p2, qj = pq;
      e1 = e2;
      { const int j = (e1 >= 0) ? (e1 & RX_JMASK) : 0; p0 = V.x[3 * j]; p1 = V.x[3 * j + 1]; p2 = V.x[3 * j + 2]; pq = V.q[j]; ptj = V.rtype[j]; }
      e2 = load_ent(c0 + 128);
      if (ent >= 0) {
        const int j = ent & RX_JMASK;
        const double *sh = s_sh + 3 * ((ent >> 24) & 0x7F);
        const double d0 = x0 - xi0 + sh[0], d1 = x1 - xi1 + sh[1], d2 = x2 - xi2 + sh[2];
        double ev, ec, sc_;
        rx_nonbonded_pair(P, &s_tbp[ti * RX_MAXT + tj], qi * qj, d0 * d0 + d1 * d1 + d2 * d2, &ev, &ec, &sc_);
        e[RX_E_VDW] += ev; e[RX_E_COUL] += ec;
        const double g0 = sc_ * d0, g1 = sc_ * d1, g2 = sc_ * d2;   // force on i; the partner takes the opposite
        f0 += g0; f1 += g1; f2 += g2;
        (void)__hip_atomic_fetch_add(&s_nbf[j], -g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        (void)__hip_atomic_fetch_add(&s_nbf[np + j], -g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        (void)__hip_atomic_fetch_add(&s_nbf[2 * np + j], -g2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // pair virial d (x) f_j = -s d (x) d
        w[0] -= g0 * d0; w[1] -= g1 * d1; w[2] -= g2 * d2; w[3] -= g0 * d1; w[4] -= g0 * d2; w[5] -= g1 * d2;
      }
    }
    f0 = wave_sum(f0); f1 = wave_sum(f1); f2 = wave_sum(f2);
    if (lane == 0) {
      (void)__hip_atomic_fetch_add(&s_nbf[i], f0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      (void)__hip_atomic_fetch_add(&s_nbf[np + i], f1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      (void)__hip_atomic_fetch_add(&s_nbf[2 * np + i], f2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < 3 * n; k += RX_KT) {
    const int a = k / 3, c = k - 3 * a;
    const double v = s_nbf[(size_t)c * np + a];
    if (v != 0.0) atomicAdd(&V.f[k], v);
  }
  for (int i = blockIdx.x * RX_NB1_ROWS + threadIdx.x; i < min(n, (int)(blockIdx.x + 1) * RX_NB1_ROWS); i += RX_KT) {
    {
      const double qi = V.q[i];
      const int ti = V.rtype[i];
      e[RX_E_POL] += RX_KCALPMOL_TO_EV * (P->sbp[ti].chi * qi + 0.5 * P->sbp[ti].eta * qi * qi);
    }
  }
  rx_flush_block<RX_KS>(e, w, V, *sims[blockIdx.y].sc, P_LJ);
}
__global__ __launch_bounds__(TPB) void k_rx_back1(const RxView *views, const RxParams *__restrict__ P) {
  const RxView V = views[blockIdx.y];
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i < V.n) rx_back_corr(P, &V, i);
}
__global__ __launch_bounds__(TPB) void k_rx_back2(const SimDev *sims, const RxView *views, const RxParams *__restrict__ P) {
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
// Start of a run: a new fix qeq/reax starts from zeros (the reference's two LAMMPS lifetimes per evaluation each do).  The history
// only decides where the conjugate gradients START -- the answer is fixed by the tolerance -- so a replica that continues a run
// of the same state (RxView::warm: phase B after phase A, or the state's previous evaluation) keeps it and none of its solves
// counts as cold.
__global__ __launch_bounds__(TPB) void k_rx_phase_init(const RxView *views) {
  const RxView V = views[blockIdx.y];
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i < V.npad && !V.warm) {
    for (int k = 0; k < 4; k++) V.s_hist[(size_t)k * V.npad + i] = 0.0;
    for (int k = 0; k < 3; k++) V.t_hist[(size_t)k * V.npad + i] = 0.0;
  }
  if (i == 0) {
    for (int k = 0; k < 6; k++) V.qstat[k] = 0;
    if (V.warm) V.qstat[1] = RX_QEQ_COLD;
    *V.overflow = 0; V.sweep_acc[0] = 0; V.sweep_acc[1] = 0;
  }
}

static inline dim3 g2(int nx, int ns) { return dim3((unsigned)nx, (unsigned)ns, 1); }
static inline int cdv(int a, int b) { return (a + b - 1) / b; }

__global__ void k_rx_collect_stats(const RxView *views, int ns, long long *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ns) return;
  const RxView &V = views[i];
  for (int k = 0; k < 6; k++) out[8 * i + k] = V.qstat[k];
  out[8 * i + 6] = V.sweep_acc[0]; out[8 * i + 7] = V.sweep_acc[1];
}
void mdk_reax_collect_stats(hipStream_t st, const RxView *v, int ns, long long *out) {
  if (ns > 0) hipLaunchKernelGGL(k_rx_collect_stats, dim3((ns + 63) / 64), dim3(64), 0, st, v, ns, out);
}
void mdk_reax_phase_init(hipStream_t st, const RxView *v, int ns, int maxpad) {
  hipLaunchKernelGGL(k_rx_phase_init, g2(cdv(maxpad, TPB), ns), dim3(TPB), 0, st, v);
}
void mdk_reax_forces(hipStream_t st, const SimDev *d, RxView *v, const RxParams *P, int ns, int maxatoms, double rlist, double qeq_tol, int qeq_maxiter, const RxQeqPlan &plan,
                     int terms, bool col16, std::vector<hipEvent_t> *ev, size_t *ev_used, const RxSide *side) {
  // a HIP-event pair around every launch of the matrix sweep when the caller profiles (bench.py's roofline block)
  auto sweep = [&](int it) {
    const dim3 gk = g2(cdv(maxatoms, RX_SWR), ns);
    if (ev) {
      while (*ev_used + 2 > ev->size()) { hipEvent_t a; if (hipEventCreate(&a) != hipSuccess) { ev = nullptr; break; } ev->push_back(a); }
    }
    if (ev) (void)hipEventRecord((*ev)[*ev_used], st);
    // replicas of up to 4 096 atoms: the gathered vector in LDS (64 KB)
    static const bool zlds_off = scema_env("SCEMA_REAX_QEQ_ZLDS") && atoi(scema_env("SCEMA_REAX_QEQ_ZLDS")) == 0;
    const size_t lds = (size_t)((maxatoms + 63) / 64 * 64) * sizeof(double2);
    // (the kernel has 1 KB of static LDS besides; beyond 48 KB of dynamic LDS a launch needs the opt-in, as the other large-LDS kernels take it)
    static size_t optin_tab[16] = {0};
    size_t &optin = lds_optin_slot(optin_tab);
    const bool zlds = col16 && lds + 2048 <= 64 * 1024 && !zlds_off;
    if (plan.sym) {
      static size_t optin_sym[16] = {0};
      size_t &os = lds_optin_slot(optin_sym);
      if (2 * lds > 47 * 1024 && 2 * lds > os) { (void)hipFuncSetAttribute((const void *)k_rx_qeq_sweep_sym, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds)); os = 2 * lds; }
      hipLaunchKernelGGL(k_rx_qeq_sweep_sym, gk, dim3(RX_KT), 2 * lds, st, v, it);
    } else
    if (zlds && lds > 47 * 1024 && lds > optin) { (void)hipFuncSetAttribute((const void *)k_rx_qeq_sweep<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); optin = lds; }
    if (plan.sym) {}
    else if (zlds) hipLaunchKernelGGL((k_rx_qeq_sweep<true, true>), gk, dim3(RX_KT), lds, st, v, P, qeq_tol, it);
    else if (col16) hipLaunchKernelGGL((k_rx_qeq_sweep<true, false>), gk, dim3(RX_KT), 0, st, v, P, qeq_tol, it);
    else hipLaunchKernelGGL((k_rx_qeq_sweep<false, false>), gk, dim3(RX_KT), 0, st, v, P, qeq_tol, it);
    if (ev) { (void)hipEventRecord((*ev)[*ev_used + 1], st); *ev_used += 2; }
  };
  const dim3 ga = g2(cdv(maxatoms, TPB), ns), gr = g2(cdv(maxatoms, RX_TPB), ns), gk = g2(cdv(maxatoms, 64), ns), gu = g2(cdv(maxatoms, QEQ_UT), ns);
  hipLaunchKernelGGL(k_rx_prepare, dim3(ns), dim3(64), 0, st, d, v);
  hipLaunchKernelGGL(k_rx_wrap, ga, dim3(TPB), 0, st, d, v);
  hipLaunchKernelGGL(k_rx_neigh, gk, dim3(64 * (64 / RX_NBR)), 0, st, d, v, P, rlist);
  // Two chains from here to the sum of the forces.  CHARGES: matrix rows, conjugate gradients, charges, non-bonded pairs.  BOND ORDERS: bond
  // orders, corrections, bonded terms, back-propagation.  Neither reads what the other writes, except that k_rx_corr zeroes the force array the
  // non-bonded pass adds to (one event).  The second is a chain of latency-bound kernels (two waves per SIMD waiting on dependent loads, a tenth
  // of the issue rate) and the first spends half of its time in launches that find most replicas converged: side by side on two streams.
  hipStream_t sb = side ? side->st2 : st;
  if (side) { (void)hipEventRecord(side->fork, st); (void)hipStreamWaitEvent(sb, side->fork, 0); }
  hipLaunchKernelGGL(k_rx_bonds, g2(cdv(maxatoms, 8 * (TPB / 64)), ns), dim3(TPB), 0, sb, v, P);
  hipLaunchKernelGGL(k_rx_rev, ga, dim3(TPB), 0, sb, v);
  hipLaunchKernelGGL(k_rx_corr, g2(cdv(maxatoms, TPB / 4), ns), dim3(TPB), 0, sb, v, P);
  if (side) (void)hipEventRecord(side->mid, sb);
  if (terms & 1) hipLaunchKernelGGL(k_rx_terms<0>, gr, dim3(RX_TPB), 0, sb, d, v, P);
  // (test hook: a small item list forces the in-place path of the two item kernels)
  static const int item_cap = scema_env("SCEMA_MD_RX_ITEMCAP") ? std::max(0, std::min(RX_TORS_CAP, atoi(scema_env("SCEMA_MD_RX_ITEMCAP")))) : RX_TORS_CAP;
  {
    // (replicas whose force table fits in LDS next to the item list: the items' atom sums meet there)
    static const bool lacc_off = scema_env("SCEMA_MD_RX_ITEM_LDS") && atoi(scema_env("SCEMA_MD_RX_ITEM_LDS")) == 0;
    const int mp = (maxatoms + 63) / 64 * 64;
    const size_t acc_lds = 4 * (size_t)mp * sizeof(double);
    const bool lacc = !lacc_off && mp <= RX_NB1_MAXPAD / 2;
    if (lacc) {
      static size_t optin_tab[16] = {0};
      size_t &optin = lds_optin_slot(optin_tab);
      if (acc_lds > 32 * 1024 && acc_lds > optin) {
        (void)hipFuncSetAttribute((const void *)k_rx_angles<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)acc_lds);
        (void)hipFuncSetAttribute((const void *)k_rx_torsions<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)acc_lds);
        optin = acc_lds;
      }
    }
    const dim3 gi = g2(cdv(maxatoms, RX_TORS_ATOMS), ns);
    if (terms & 2) {
      if (lacc) hipLaunchKernelGGL(k_rx_angles<true>, gi, dim3(RX_TORS_ATOMS), acc_lds, sb, d, v, P, item_cap);
      else hipLaunchKernelGGL(k_rx_angles<false>, gi, dim3(RX_TORS_ATOMS), 0, sb, d, v, P, item_cap);
    }
    if (terms & 4) {
      if (lacc) hipLaunchKernelGGL(k_rx_torsions<true>, gi, dim3(RX_TORS_ATOMS), acc_lds, sb, d, v, P, item_cap);
      else hipLaunchKernelGGL(k_rx_torsions<false>, gi, dim3(RX_TORS_ATOMS), 0, sb, d, v, P, item_cap);
    }
  }
  if (terms & 8) hipLaunchKernelGGL(k_rx_terms<3>, gr, dim3(RX_TPB), 0, sb, d, v, P);
  hipLaunchKernelGGL(k_rx_back1, ga, dim3(TPB), 0, sb, v, P);
  hipLaunchKernelGGL(k_rx_back2, ga, dim3(TPB), 0, sb, d, v, P);
  if (side) (void)hipEventRecord(side->join, sb);
  if (plan.sym) hipLaunchKernelGGL(k_rx_hrow<true>, gk, dim3(RX_KT), 0, st, d, v, P);
  else hipLaunchKernelGGL(k_rx_hrow<false>, gk, dim3(RX_KT), 0, st, d, v, P);
  if (plan.precond) {
    hipLaunchKernelGGL(k_rx_qeq_pm_rows, gu, dim3(QEQ_UT), 0, st, d, v, P);
    hipLaunchKernelGGL(k_rx_qeq_pm_sym, gu, dim3(QEQ_UT), 0, st, d, v);
  }
  hipLaunchKernelGGL(k_rx_qeq_guess, gu, dim3(QEQ_UT), 0, st, v, plan.setup, plan.sym);
  const int nlaunch = plan.launch < qeq_maxiter ? plan.launch : qeq_maxiter;
  if (plan.sym) {
    // the symmetric form: a sweep, then the replica's one-workgroup step (the product is complete only when the sweep has ended)
    sweep(-1);
    hipLaunchKernelGGL(k_rx_qeq_step, dim3(ns), dim3(RX_STEP_TPB), 0, st, v, P, qeq_tol, -1);
    for (int it = 0; it < nlaunch; it++) {
      sweep(it);
      hipLaunchKernelGGL(k_rx_qeq_step, dim3(ns), dim3(RX_STEP_TPB), 0, st, v, P, qeq_tol, it);
    }
    const size_t lds2 = 2 * (size_t)((maxatoms + 63) / 64 * 64) * sizeof(double2);
    static size_t optin_fin[16] = {0};
    size_t &of = lds_optin_slot(optin_fin);
    if (lds2 > 47 * 1024 && lds2 > of) { (void)hipFuncSetAttribute((const void *)k_rx_qeq_finish_sym, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2); of = lds2; }
    hipLaunchKernelGGL(k_rx_qeq_finish_sym, dim3(ns), dim3(QEQ_TPB), lds2, st, v, P, qeq_tol, nlaunch, qeq_maxiter, plan.setup);
  } else {
    sweep(-1);
    hipLaunchKernelGGL(k_rx_qeq_update, gu, dim3(QEQ_UT), 0, st, v, P, qeq_tol, -1);
    for (int it = 0; it < nlaunch; it++) {
      sweep(it);
      hipLaunchKernelGGL(k_rx_qeq_update, gu, dim3(QEQ_UT), 0, st, v, P, qeq_tol, it);
    }
    hipLaunchKernelGGL(k_rx_qeq_finish, dim3(ns), dim3(QEQ_TPB), 0, st, d, v, P, qeq_tol, nlaunch, qeq_maxiter, plan.setup);
  }
  if (side) (void)hipStreamWaitEvent(st, side->mid, 0);
  if (terms & 16) {
    static const bool once_off = scema_env("SCEMA_MD_RX_NB_ONCE") && atoi(scema_env("SCEMA_MD_RX_NB_ONCE")) == 0;   // (test switch: the both-ends kernel)
    const int maxpad = (maxatoms + 63) / 64 * 64;
    if (!once_off && maxpad <= RX_NB1_MAXPAD) {
      const size_t lds = 3 * (size_t)maxpad * sizeof(double);
      static size_t optin_tab[16] = {0};
      size_t &optin = lds_optin_slot(optin_tab);
      if (lds > 48 * 1024 && lds > optin) { (void)hipFuncSetAttribute((const void *)k_rx_nonbonded_once, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); optin = lds; }
      hipLaunchKernelGGL(k_rx_nonbonded_once, g2(cdv(maxatoms, RX_NB1_ROWS), ns), dim3(RX_KT), lds, st, d, v, P);
    } else hipLaunchKernelGGL(k_rx_nonbonded, gk, dim3(RX_KT), 0, st, d, v, P);
  }
  if (side) (void)hipStreamWaitEvent(st, side->join, 0);
  hipLaunchKernelGGL(k_rx_finish, dim3(ns), dim3(64), 0, st, d, v);
}
