// engine_run.cpp -- one "run" of a batch with the OPLS force stage: slots, launch sequence of the MD steps, what comes back
#include "engine.h"
#include "../md_env.h"

namespace scema_eng {

// -------------------------------------------------------------------------------------------
// one "run" of a batch
// -------------------------------------------------------------------------------------------
// slots: every cell is padded to a multiple of MD_CLUSTER slots (i-clusters never straddle cells)
static int padded_slots(int natoms, int ncells) { return (natoms + (MD_CLUSTER - 1) * ncells + 255) / 256 * 256; }

int ensure_slot(scema_md_engine *e, Slot &sl, int natoms, int maxneigh, int ncells, int nk, int capj) {
  const int npad = padded_slots(natoms, ncells);
  if (natoms > sl.cap_atoms || npad > sl.cap_pad) {
    HIPCHK(sl.f.ensure(3 * (size_t)natoms * 8));
    HIPCHK(sl.slot_of.ensure((size_t)natoms * 4));
    HIPCHK(sl.fs.ensure(3 * (size_t)npad * 8));
    HIPCHK(sl.fb.ensure(3 * (size_t)npad * 8));
    HIPCHK(sl.virb.ensure(((size_t)natoms / BT_OWNERS + 2) * 6 * 8));
    HIPCHK(sl.tile_order.ensure((size_t)npad * 4));
    HIPCHK(sl.wrapn.ensure(3 * (size_t)natoms * 4));
    HIPCHK(sl.xhold.ensure(3 * (size_t)natoms * 8));
    HIPCHK(sl.cell_of.ensure((size_t)natoms * 4));
    HIPCHK(sl.ckey.ensure((size_t)natoms * 4));
    HIPCHK(sl.slot_tmp.ensure((size_t)npad * 4));   // indexed by (padded) slot
    HIPCHK(sl.xbak.ensure(3 * (size_t)natoms * 8));
    HIPCHK(sl.vbak.ensure(3 * (size_t)natoms * 8));
    HIPCHK(sl.xq.ensure((size_t)npad * 32));
    HIPCHK(sl.stype.ensure((size_t)npad * 4));
    HIPCHK(sl.perm.ensure((size_t)npad * 4));
    HIPCHK(sl.numneigh.ensure((size_t)npad * 4));
    sl.cap_atoms = natoms;
    sl.cap_pad = npad;
    sl.cap_neigh = 0;
  }
  if (maxneigh > sl.cap_neigh || sl.cap_neigh == 0) {
    // one row of maxneigh entries per cluster of MD_CLUSTER slots
    HIPCHK(sl.neigh.ensure((size_t)maxneigh * (npad / MD_CLUSTER) * 4 + 8192));
    sl.cap_neigh = maxneigh;
  }
  if (ncells + 1 > sl.cap_cells) {
    HIPCHK(sl.cell_count.ensure((size_t)(ncells + 1) * 4));
    HIPCHK(sl.cell_start.ensure((size_t)(ncells + 1) * 4));
    HIPCHK(sl.cell_fill.ensure((size_t)(ncells + 1) * 4));
    HIPCHK(sl.tile_nj.ensure((size_t)(ncells + 1) * 4));
    HIPCHK(sl.tile_wstart.ensure((size_t)(ncells + 1) * 9 * 4));
    HIPCHK(sl.virp.ensure((size_t)(ncells + 1) * MD_TILE_WAVES * 6 * 8));
    sl.cap_cells = ncells + 1;
  }
  if ((size_t)ncells * capj > sl.cap_jtab || sl.cap_jtab == 0) {
    HIPCHK(sl.tile_jtab.ensure((size_t)ncells * capj * 4 + 1024));
    sl.cap_jtab = (size_t)ncells * capj;
  }
  if (nk > sl.cap_k || sl.cap_k == 0) {
    const int kc = std::max(nk, 64);
    HIPCHK(sl.sfac.ensure((size_t)kc * 2 * 8));
    HIPCHK(sl.kvec.ensure((size_t)kc * 4 * 8));
    sl.cap_k = kc;
  }
  return SCEMA_MD_OK;
}

// After k_pair: bonded terms on the main stream, structure factors + per-k coefficients on the side stream (both
// are small, latency-bound kernels that need only the positions), joined before the per-atom reciprocal force.
static hipError_t force_stage(scema_md_engine *e, hipStream_t st, bool allow_side, const SimDev *D, int ns, int maxbt, int maxloc, int maxcoef, int maxatoms,
                              int maxk, int mmax, int maxgrp, int parts, int pairvir, bool pppm_ahead = false) {
  const bool side = allow_side && maxk > 0 && e->stream2 != nullptr && ns >= 16;   // small batches: the fork/join costs more than it hides
  if (side) {
    hipError_t rc = hipEventRecord(e->ev_fork, st);
    if (rc != hipSuccess) return rc;
    if ((rc = hipStreamWaitEvent(e->stream2, e->ev_fork, 0)) != hipSuccess) return rc;
    mdk_ewald_recip(e->stream2, D, ns, maxk, mmax, maxgrp);
    if ((rc = hipEventRecord(e->ev_join, e->stream2)) != hipSuccess) return rc;
  }
  mdk_bonded(st, D, ns, maxbt, maxloc, maxcoef, parts);
  if (side) {
    hipError_t rc = hipStreamWaitEvent(st, e->ev_join, 0);
    if (rc != hipSuccess) return rc;
  } else {
    mdk_ewald_recip(st, D, ns, maxk, mmax, maxgrp);
  }
  if (pppm_ahead) {   // the PPPM chain of this step ran on the side stream and left its forces in SimDev::f
    hipError_t rc = hipStreamWaitEvent(st, e->ev_join, 0);
    if (rc != hipSuccess) return rc;
  }
  mdk_ewald_force(st, D, ns, maxatoms, pairvir, pppm_ahead ? 1 : 0);
  return hipSuccess;
}


// Advance sims[0..ns) (already assigned to slots 0..ns-1, scalars' box valid on the device).
// On return the per-sim SimScalars are in e->h_sc.
int run_phase(scema_md_engine *e, std::vector<ActiveSim> &sims, const RunSpec &spec) {
  if (e->reax_active) return run_phase_reax(e, sims, spec);
  const int ns = (int)sims.size();
  const scema_md_params &P = e->p;
  const auto t_enter = std::chrono::steady_clock::now();
  const double cutmax_all = std::max(P.cut_lj, P.cut_coul);
  // k-space set-up of every simulation first (by simulation index, before the launch order exists): g_ewald with the k list of
  // the Ewald sum, or with the PPPM grid.  Pure functions of the box and by far the longest part of the layout (12 us per PE-10k
  // replica, 7 of 8 ms for 576 while the GPU waits), so large batches spread them over a few host threads.
  std::vector<EwaldSetup> ews_i(ns);
  {
    auto kspace_one = [&](int i) {
      const Topo &T = *sims[i].st->topo;
      const SimScalars &hsc = e->h_sc[i];
      EwaldSetup &ew = ews_i[i];
      const bool kept = spec.ew_keep && spec.keep;
      const bool pppm = P.kspace_style == 1 && T.qsqsum > 0.0 && !kept;
      if (kept && (int)spec.ew_keep->size() == ns) ew = (*spec.ew_keep)[i];   // a run keeps the k-space setup of its start
      else ewald_setup(P, T, hsc.box, ew, pppm);
      if (pppm) {
        // PPPM: the Ewald k list is not used; g_ewald is adjusted to the grid (and with it the real-space part)
        int pgd[3];
        double gp = ew.g;
        pppm_setup_host(P, T, hsc.box, gp, pgd);
        ew = EwaldSetup();
        ew.g = gp;
        for (int d = 0; d < 3; d++) ew.kmaxd[d] = -pgd[d];   // the grid travels in the set-up record (negative: not a k range)
      }
    };
    const int nthr = ns >= 64 ? std::max(1, std::min(8, (int)std::thread::hardware_concurrency())) : 1;
    if (nthr == 1) {
      for (int i = 0; i < ns; i++) kspace_one(i);
    } else {
      std::vector<std::thread> pool;
      for (int t = 0; t < nthr; t++)
        pool.emplace_back([&, t] { for (int i = t; i < ns; i += nthr) kspace_one(i); });
      for (auto &th : pool) th.join();
    }
  }
  if (spec.ew_keep && !spec.keep) *spec.ew_keep = ews_i;
  const double t_kspace_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enter).count();
  // order: longest run first, so the active simulations are always a prefix; among equally long runs the simulations that share
  // a PPPM grid stand together (one batched transform per such group; a strained batch can straddle a grid size)
  auto grid_key = [&](int i) { const int *k = ews_i[i].kmaxd; return k[0] < 0 ? ((long)(-k[0]) << 40) | ((long)(-k[1]) << 20) | (long)(-k[2]) : 0L; };
  std::vector<int> order(ns);
  for (int i = 0; i < ns; i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    if (sims[a].nsteps != sims[b].nsteps) return sims[a].nsteps > sims[b].nsteps;
    return grid_key(a) < grid_key(b);
  });
  // Part batches on streams of their own: every kernel but k_pair is latency bound and leaves most issue slots idle, while
  // k_pair saturates them and holds every wave slot of the chip; with independent parts in flight the small kernels of
  // one part fill in as the pair workgroups of another retire (the in-order streams fall out of phase by themselves).
  // The parts take every nparts-th rank of the length order, so each is itself sorted longest first.  How many parts is
  // a measured table (profiles/r06_t_parts_ab.log, same-box A/B against the whole / two-half forms): under 9 replicas
  // the batch runs whole with its PPPM chain on the side stream (8 replicas: 289 whole, 285 / 281 as three / four parts);
  // 9 replicas as three parts of three (306 against 294); 10-63 replicas as four parts (with the largest cells, below:
  // +6..12 % at 10-30 replicas, +4..6 % at 36-60; three parts 1-2 % behind, two 4-7 %); from 64 on two halves (three or
  // four parts: -0.5..+0.7 %, the chip is full either way).  Four is the most: a process has four hardware queues and further streams share them.  Five to eight
  // parts were measured -- six parts of a 36-replica batch: -12 %; with GPU_MAX_HW_QUEUES=8 -29 % -- and removed.  (They
  // also showed a bug: the hipFFT plans of the PPPM path were shared by all part streams beyond the second, pppm_plan below.)
  // SCEMA_MD_PARTS (2-4) forces a count for batches of SCEMA_MD_PART_MIN (2) replicas per part and more,
  // SCEMA_MD_SPLIT_MIN moves the lower end, SCEMA_MD_SPLIT=0 runs every batch whole.
  constexpr int MAXP = 4;
  static const int parts_env = [] { const char *s = scema_env("SCEMA_MD_PARTS"); return s ? std::min(4, std::max(2, atoi(s))) : 0; }();
  static const int part_min_env = [] { const char *s = scema_env("SCEMA_MD_PART_MIN"); return s ? std::max(1, atoi(s)) : 2; }();
  const bool can_split = e->split_streams && e->stream3 != nullptr && ns >= e->split_min && ns < e->split_max;
  int nhalf = 1;
  if (can_split) {
    nhalf = parts_env > 0 ? (ns >= part_min_env * parts_env ? parts_env : 2) : ns < 10 ? 3 : ns < 64 ? 4 : 2;
    if (e->stream2 == nullptr || e->rx_side1 == nullptr) nhalf = std::min(nhalf, 2);
    nhalf = std::max(1, std::min(nhalf, ns / 2));
  }
  while ((int)e->md_part_done.size() < nhalf - 1) {
    hipEvent_t pe = nullptr;
    HIPCHK(hipEventCreateWithFlags(&pe, hipEventDisableTiming));
    e->md_part_done.push_back(pe);
  }
  if (nhalf >= 2) {
    std::vector<int> o2;
    o2.reserve(ns);
    for (int p = 0; p < nhalf; p++)
      for (int r = p; r < ns; r += nhalf) o2.push_back(order[r]);
    order.swap(o2);
  }
  int hbeg[MAXP], hcnt[MAXP];
  for (int p = 0; p < MAXP; p++) hbeg[p] = p ? ns : 0, hcnt[p] = p ? 0 : ns;
  if (nhalf >= 2)
    for (int p = 0, b = 0; p < MAXP; p++) {
      hbeg[p] = b;
      hcnt[p] = p < nhalf ? (ns - p + nhalf - 1) / nhalf : 0;
      b += hcnt[p];
    }
  const auto part_of = [&](int pos) { int h = 0; while (h + 1 < nhalf && pos >= hbeg[h + 1]) h++; return h; };
  e->h_sims.assign(ns, SimDev());
  int maxbt = 1, maxloc = 1, maxcoef = 0;
  bool any_validate = false;   // some simulation may keep the rows its slot holds (SimDev::keep_list: 1 from the run, 2 from the update before)
  int maxrow = 64, maxcapj = 64, maxpoly = 1, maxatoms = 0, maxpad = 0, maxcells = 0, maxk = 0, mmax = 1, maxb = 0, maxa = 0, maxd = 0, maxi = 0, maxs = 0, maxclus = 0, maxunits = 0, maxsteps = 0;
  // k-vector tables of all simulations (indices, row run lengths, groups), packed into one upload
  std::vector<int> &kpack = e->h_kpack;
  kpack.clear();
  std::vector<size_t> koff(ns, 0);
  int maxgrp = 0;
  double t_lay[4] = {0, 0, 0, 0};   // host time of the layout loop by part (SCEMA_MD_TIMING): box range, cell grid, k-space set-up, the rest
  auto t_now = [] { return std::chrono::steady_clock::now(); };
  auto t_ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  std::vector<std::vector<FlipEvent>> flips(ns);   // per position: the box flips of this run (fix deform, flip yes)
  std::vector<EwaldSetup> ews(ns);   // by position
  for (int pos = 0; pos < ns; pos++) ews[pos] = std::move(ews_i[order[pos]]);
  int maxgrid = 0;   // PPPM: largest grid of the batch
  int maxgridp = 0;  // ... with five more points per x row
  bool padx_ok = true;
  // NOTE: slot index == position in `sims` (not in `order`): scalars stay attached to their slot
  for (int pos = 0; pos < ns; pos++) {
    const int i = order[pos];
    ActiveSim &A = sims[i];
    Topo &T = *A.st->topo;
    const SimScalars &hsc = e->h_sc[i];
    const auto tl0 = t_now();
    // box range over this run -> cell grid that stays valid while the box deforms (and flips: the tilt is largest just
    // before a flip, those boxes are kept as extremes)
    double box_end[9];
    std::memcpy(box_end, hsc.box, sizeof box_end);
    std::vector<HostBox> boxes(2);
    if (spec.deform) {
      std::vector<HostBox> extremes;
      if (!deform_trajectory(hsc.box, A.rates, A.dt, A.nsteps, box_end, flips[pos], extremes))
        return fail(e, SCEMA_MD_ERR_BOX, "fix deform is changing yz too much with xy: the strain would tilt yz past half the box (a yz flip changes xz by xy, "
                    "which LAMMPS refuses while xy is deformed too, as in.strain.lammps always does)");
      boxes.insert(boxes.end(), extremes.begin(), extremes.end());
    }
    box_derive(hsc.box, boxes[0]);
    box_derive(box_end, boxes[1]);
    if (spec.nh && spec.npt && spec.box_margin > 0.0)
      for (int sgn = -1; sgn <= 1; sgn += 2) {   // the barostat dilates the box (tilts with it): both ends of the allowed range
        double bx[9];
        const double f = 1.0 + sgn * spec.box_margin;
        for (int d = 0; d < 3; d++) {
          const double c = 0.5 * (hsc.box[d] + hsc.box[3 + d]);
          bx[d] = c + (hsc.box[d] - c) * f;
          bx[3 + d] = c + (hsc.box[3 + d] - c) * f;
        }
        for (int k = 6; k < 9; k++) bx[k] = hsc.box[k] * f;
        HostBox hb;
        box_derive(bx, hb);
        boxes.push_back(hb);
      }
    const HostBox &b0 = boxes[0], &b1 = boxes[1];
    double w0[3] = {1e300, 1e300, 1e300}, w1[3];   // w0 = narrowest perpendicular widths over the run
    double vol_min = 1e300, vol_max = 0.0;
    for (const HostBox &hb : boxes) {
      perp_widths(hb, w1);
      for (int d = 0; d < 3; d++) w0[d] = std::min(w0[d], w1[d]);
      vol_min = std::min(vol_min, hb.vol);
      vol_max = std::max(vol_max, hb.vol);
    }
    for (int d = 0; d < 3; d++) w1[d] = w0[d];
    (void)b0; (void)b1;
    SimDev S;
    std::memset(&S, 0, sizeof S);
    // list skin of this simulation = the reference's neighbour skin + the state's performance extra (dropped where the
    // box is too small for it)
    if (!e->skin_adapt) A.st->skin_extra = e->skin_extra_fixed;
    for (int d = 0; d < 3; d++)
      if (std::min(w0[d], w1[d]) < 2.0 * (cutmax_all + P.skin + A.st->skin_extra)) A.st->skin_extra = 0.0;
    const double skin_i = P.skin + A.st->skin_extra;
    const double rlist = cutmax_all + skin_i;
    for (int d = 0; d < 3; d++)
      if (std::min(w0[d], w1[d]) < 2.0 * rlist)
        return fail(e, SCEMA_MD_ERR_BOX, "box width %.3f < 2*(cutoff+skin) = %.3f in dim %d", std::min(w0[d], w1[d]), 2 * rlist, d);
    // Cell grid = tiling of k_pair (one workgroup per cell).  The per-tile phases of k_pair (table load, barrier,
    // flush) are amortised over the tile's rows, so cells are made as LARGE as the LDS allows: of all grids with
    // cell edges between rlist/2 and rlist, the one with the largest cells whose estimated j table (the images of
    // the half stencil within rlist of the cell, 28 B of LDS each) still fits two workgroups per CU.  PE-10k:
    // 5x6x4 cells of 8.9 x 7.4 x 10.1 A (22 clusters, 2 280 table entries) instead of 6x6x5 (14 clusters, 2 040):
    // k_pair -3.5 %, build +8 %, step -2.4 %.  Denser systems fall back to cells of rlist/3, rlist/4, ...
    const auto tl1 = t_now();
    const double rho = T.natoms / vol_min;
    int capj = 0, maxneigh = 0;
    bool fits = false;
    auto size_grid = [&](const int nc[3], int mst[3], int &cj_out, int &mn_out) {
      int ncells = 1;
      for (int d = 0; d < 3; d++) {
        const double w = std::min(w0[d], w1[d]);
        mst[d] = (int)std::ceil(rlist / (w / nc[d]) - 1e-12);
        ncells *= nc[d];
      }
      // Cartesian extents of one cell (bounding box of its edge vectors), the larger of the two boxes
      double ext[3] = {0, 0, 0};
      for (const HostBox &hbr : boxes) {
        const HostBox *hb = &hbr;
        ext[0] = std::max(ext[0], std::fabs(hb->h[0]) / nc[0] + std::fabs(hb->h[5]) / nc[1] + std::fabs(hb->h[4]) / nc[2]);
        ext[1] = std::max(ext[1], std::fabs(hb->h[1]) / nc[1] + std::fabs(hb->h[3]) / nc[2]);
        ext[2] = std::max(ext[2], std::fabs(hb->h[2]) / nc[2]);
      }
      const double r = rlist;
      // volume of (cell (+) ball of rlist); the table holds the half stencil: half of it plus half of the own cell.
      // Calibrated on PE-10k grids from 6x6x5 to 4x5x4: estimate = 1.15-1.17 x the largest table seen.
      const double vmink = ext[0] * ext[1] * ext[2] + 2.0 * r * (ext[0] * ext[1] + ext[1] * ext[2] + ext[0] * ext[2]) +
                           MD_PI * r * r * (ext[0] + ext[1] + ext[2]) + 4.0 / 3.0 * MD_PI * r * r * r;
      const double vmin = vol_min;
      const double rho_slots = (T.natoms + 1.5 * ncells) / vmin;
      const double cellvol = vol_max / ncells;
      double cj = rho_slots * (0.5 * vmink + 0.5 * cellvol) * 1.13 * e->jtab_grow;
      cj = std::min(cj, (double)padded_slots(T.natoms, ncells) * 14.0);
      cj_out = std::max(64, ((int)std::ceil(cj) + 63) / 64 * 64);
      // row capacity of one i-cluster: the union of 4 half neighbour spheres whose centres are within a cell, plus
      // headroom; regrown on overflow
      mn_out = (int)std::ceil(rho * 4.0 / 3.0 * MD_PI * rlist * rlist * rlist * 1.25 * e->neigh_grow) + 128;
      mn_out = (std::min(mn_out, cj_out) + 63) / 64 * 64 + 64;   // (+ 64: the last 64 words of a row's capacity are k_neigh_build's dump zone, md_pair.hip)
      return cj_out <= MD_MAXJTAB && mdk_pair_lds_bytes(cj_out) <= 74 * 1024 && mdk_neigh_lds_bytes(cj_out, mn_out) <= 150 * 1024;
    };
    // A run that follows another one on the same slot (the sampling run of an evaluation behind its straining run) keeps that run's
    // cell grid where it is a valid one for the new box, and with it the neighbour rows on the device: one list build in seven of an
    // evaluation less.  (Capacities are strides of the stored tables: they stay what they were.)
    static const bool keep_lists = !(scema_env("SCEMA_MD_KEEP_LIST") && atoi(scema_env("SCEMA_MD_KEEP_LIST")) == 0);
    bool keep = false;
    {
      const ListSig &g = e->slots[i]->sig;
      if (spec.keep_list && keep_lists && g.valid && g.rx_stamp == 0 && g.topo == T.id && g.rlist == rlist && g.cut_lj == P.cut_lj && g.cut_coul == P.cut_coul &&
          (spec.keep_list == 1 || g.state == A.st->id) && !hsc.force_rebuild && !hsc.overflow) {
        int mst[3], cj = 0, mn = 0;
        if (size_grid(g.nc, mst, cj, mn) && cj <= g.capj && mn <= g.maxneigh && padded_slots(T.natoms, g.nc[0] * g.nc[1] * g.nc[2]) == g.npad) {
          keep = fits = true;
          capj = g.capj; maxneigh = g.maxneigh;
          for (int d = 0; d < 3; d++) { S.nc[d] = g.nc[d]; S.mst[d] = mst[d]; }
        }
      }
    }
    S.keep_list = keep ? spec.keep_list : 0;
    any_validate = any_validate || S.keep_list != 0;
    // first among cell edges between rlist/2 and rlist; if no such grid fits, among edges down to rlist/4 (so that a
    // slightly denser system degrades gradually instead of dropping to the uniform fallback below)
    // replicas up to which the most-cells grid is taken: every launch group that runs whole (scanned again in round 6, profiles/r06_d_cells_scan.txt:
    // 180 instead of 120 cells +5.4 % at 9 replicas, +2.4 % at 18; batches of 32 and more run as two half batches that fill the chip together and
    // are fastest with the largest cells: 120 against 180 cells 378 / 372 evaluations/s at 36, 422 / 417 at 72, 441 / 429 at 144)
    static const int small_max = scema_env("SCEMA_MD_SMALL_BATCH_MAX") ? atoi(scema_env("SCEMA_MD_SMALL_BATCH_MAX")) : -1;
    const bool small_batch = small_max >= 0 ? ns <= small_max : (nhalf == 1 && ns <= 31);   // (31: a large batch issued whole -- SCEMA_MD_SPLIT=0, the chip-exclusive timing of bench.py -- keeps the grid it has as two halves; part batches fill the chip together, like the two halves of a large batch: 120 against 180 cells 345 / 330 evaluations/s at 12 replicas, 370 / 361 at 18, 391 / 386 at 24, profiles/r06_t_parts_ab.log)
    static const int cells_target = scema_env("SCEMA_MD_CELLS_TARGET") ? atoi(scema_env("SCEMA_MD_CELLS_TARGET")) : 0;
    for (int pass = 0; pass < 2 && !fits; pass++) {
      int lo[3], hi[3];
      for (int d = 0; d < 3; d++) {
        const double w = std::min(w0[d], w1[d]);
        lo[d] = std::max(2, std::min(64, (int)std::floor(w / (rlist * 1.0001))));
        hi[d] = std::max(lo[d], std::min(64, (int)std::floor(w / ((pass == 0 ? 0.5 : 0.25) * rlist * 1.0001))));
      }
      double best = -1.0e300;
      for (int n0 = lo[0]; n0 <= hi[0]; n0++)
        for (int n1 = lo[1]; n1 <= hi[1]; n1++)
          for (int n2 = lo[2]; n2 <= hi[2]; n2++) {
            const int nc[3] = {n0, n1, n2};
            int mst[3], cj, mn;
            if (!size_grid(nc, mst, cj, mn)) continue;
            // batches that fill the chip take the largest cells (per-tile phases amortised over more rows); small ones the
            // most cells: a single replica on 120 tiles leaves half of the 512 workgroup slots empty and waits for one tile
            // (SCEMA_MD_CELLS_TARGET: the fitting grid whose number of cells is closest to the target -- what-if runs of the tile size)
            const double vol = cells_target > 0 ? -std::fabs((double)n0 * n1 * n2 - cells_target) - 1e-3 * n2 : small_batch ? (double)n0 * n1 * n2 : 1.0 / ((double)n0 * n1 * n2);
            if (vol > best) {
              best = vol;
              fits = true;
              capj = cj; maxneigh = mn;
              for (int d = 0; d < 3; d++) { S.nc[d] = nc[d]; S.mst[d] = mst[d]; }
            }
          }
    }
    for (int k = 5; k <= 8 && !fits; k++) {
      int nc[3], mst[3];
      for (int d = 0; d < 3; d++) {
        const double w = std::min(w0[d], w1[d]);
        nc[d] = std::max(1, std::min((int)std::floor(w / (rlist / k * 1.0001)), 64));
      }
      fits = size_grid(nc, mst, capj, maxneigh);
      for (int d = 0; d < 3; d++) { S.nc[d] = nc[d]; S.mst[d] = mst[d]; }
    }
    if (!fits)
      return fail(e, SCEMA_MD_ERR_ARG, "the j table of a cell tile (%d entries) does not fit the LDS of the pair kernel (system too dense for the cutoff)", capj);
    S.ncells = S.nc[0] * S.nc[1] * S.nc[2];
    const auto tl2 = t_now();
    EwaldSetup &ew = ews[pos];   // from the pass above
    if (P.kspace_style == 1 && T.qsqsum > 0.0) {
      for (int d = 0; d < 3; d++) { S.pg[d] = -ew.kmaxd[d]; }
      maxgrid = std::max(maxgrid, S.pg[0] * S.pg[1] * S.pg[2]);
      if (S.pg[0] < 5) padx_ok = false;   // (the padded LDS copies of the spreading and interpolation kernels fold five distinct pad columns per row)
      maxgridp = std::max(maxgridp, (S.pg[0] + 5) * S.pg[1] * S.pg[2]);
    }
    const auto tl3 = t_now();
    S.nk = (int)ew.kn.size() / 3;
    for (int d = 0; d < 3; d++) S.kmaxd[d] = std::max(ew.kmaxd[d], 0);
    S.g_ewald = ew.g;
    {
      // H depends on u only: fit once per (rounded-up) range and share it between simulations
      const double perr = cached_coul_poly(e, ew.g, P.cut_coul, S.coul_poly, &S.coul_npoly, &S.coul_uscale);
      if (perr > 1e-12 && !scema_env("SCEMA_MD_POLY_TOL")) return fail(e, SCEMA_MD_ERR_ARG, "real-space Ewald polynomial fit error %.3e too large (g*rc = %.3f)", perr, ew.g * P.cut_coul);
      for (int m = 0; m < MD_MAXPOLY; m++) S.coul_poly_g[m] = S.coul_poly[m] * ew.g;
    }
    {
      const double m = 0.1 * P.skin;   // margin of the row segments over the cutoffs (scan 0 .. 0.6 skin: flat optimum at 0.05-0.15)
      S.seg_a2 = (P.cut_coul + m) * (P.cut_coul + m);
      S.seg_b2 = (P.cut_lj + m) * (P.cut_lj + m);
      // skin pairs listed beyond cutmax + far_band sit at the back of the rows and are skipped until an atom has moved far_band/2
      double frac = 0.65;   // scan 0.25 .. 0.85 on PE-10k (rebuild every ~33 steps, the largest displacement passes 0.5 A after ~8): optimum 0.65-0.75
      S.far_band = frac * skin_i;
      const double cm = std::max(P.cut_coul, P.cut_lj) + S.far_band;
      S.seg_c2 = cm * cm;
    }
    S.natoms = T.natoms;
    S.npad = padded_slots(T.natoms, S.ncells);
    S.ntypes = T.ntypes;
    Slot &sl = *e->slots[i];
    int rc = ensure_slot(e, sl, T.natoms, maxneigh, S.ncells, S.nk, capj);
    if (rc) return rc;
    S.maxneigh = maxneigh;
    S.capj = capj;
    {
      ListSig &g = sl.sig;   // what this run's rows are built for; valid once the run has ended without a fault
      g.valid = false;
      g.rx_stamp = 0;
      g.topo = T.id;
      for (int d = 0; d < 3; d++) g.nc[d] = S.nc[d];
      g.capj = capj; g.maxneigh = maxneigh; g.npad = S.npad;
      g.rlist = rlist; g.cut_lj = P.cut_lj; g.cut_coul = P.cut_coul;
    }
    maxrow = std::max(maxrow, maxneigh);
    maxcapj = std::max(maxcapj, capj);
    S.nbonds = T.nbonds; S.nbonds_noshake = T.nbonds_noshake; S.nangles = T.nangles; S.ndihedrals = T.ndihedrals;
    S.nimpropers = T.nimpropers; S.nspecial = T.nspecial; S.nclus = T.nclus;
    S.nsteps = A.nsteps;
    S.nav = 0; S.nwin = 0;
    if (spec.sample) {
      // in.homogenization.lammps:57 : nav = nss/10 (nss/1000 beyond 10000 steps); nss/nav windows
      S.nav = (A.nsteps > 10000) ? A.nsteps / 1000 : A.nsteps / 10;
      if (S.nav < 1) S.nav = 1;
      S.nwin = A.nsteps / S.nav;
    }
    S.nvt = spec.nvt;
    S.use_shake = (spec.use_shake && T.nclus > 0) ? 1 : 0;
    S.deform = spec.deform;
    if (spec.nh) {
      S.ramp = 1; S.npt = spec.npt; S.nh_total = std::max(spec.nh_total, 1); S.lavg_nav = spec.lavg_nav;
      S.t_start = spec.t_start; S.t_stop = spec.t_stop; S.p_target = spec.p_target; S.p_freq = 1.0 / spec.p_period; S.box_margin = spec.box_margin;
    }
    if (spec.minimize) {
      S.min_etol = spec.min_etol; S.min_ftol = spec.min_ftol; S.min_dmax = 0.1; S.min_maxiter = spec.min_maxiter; S.min_maxeval = spec.min_maxeval;
    }
    S.t_chain = std::min(P.t_chain, MD_MAXCHAIN);
    S.neigh_delay = P.neigh_delay;
    S.shake_maxiter = P.shake_maxiter;
    S.dt = A.dt;
    S.t_target = A.temperature;
    S.t_freq = 1.0 / P.t_period;
    S.tdof = 3.0 * T.natoms - 3.0 - (S.use_shake ? T.ncons : 0);
    S.qsqsum = T.qsqsum; S.qsum = T.qsum;
    S.cut_lj2 = P.cut_lj * P.cut_lj; S.cut_coul2 = P.cut_coul * P.cut_coul; S.rlist2 = rlist * rlist;
    S.skin = skin_i;
    S.rlist_ref2 = (cutmax_all + P.skin) * (cutmax_all + P.skin);   // the reference's list, for the roofline accounting
    S.excl_cut2 = std::min(T.excl_cut * T.excl_cut, S.rlist2);
    S.shake_tol = P.shake_tol;
    for (int k = 0; k < 6; k++) S.rates[k] = A.rates[k];
    S.type = T.d_type.as<int>(); S.q = T.d_q.as<double>(); S.mass = T.d_mass.as<double>(); S.lj = T.d_lj.as<double>();
    S.bt_terms = T.d_bt_terms.as<unsigned long long>(); S.bt_coef = T.d_bt_coef.as<double>(); S.bt_ncoef = T.bt_ncoef;
    for (int k = 0; k < 4; k++) S.bt_cf_off[k] = T.bt_cf_off[k];
    for (int k = 0; k < 6; k++) S.sp_w[k] = T.sp_w[k];
    S.ex_start = T.d_ex_start.as<int>(); S.ex_list = T.d_ex_list.as<int>();
    S.bt_desc = T.d_bt_desc.as<int>(); S.bt_atoms = T.d_bt_atoms.as<int>(); S.bt_rank = T.d_bt_rank.as<int>(); S.bt_ntile = T.bt_ntile;
    maxbt = std::max(maxbt, T.bt_ntile); maxloc = std::max(maxloc, T.bt_maxloc);
    maxcoef = std::max(maxcoef, T.bt_ncoef);
    S.clus_at = T.d_clus_at.as<int>(); S.clus_n = T.d_clus_n.as<int>(); S.clus_d = T.d_clus_d.as<double>();
    S.free_at = T.d_free_at.as<int>(); S.nfree = T.nfree;
    S.x = A.st->x.as<double>(); S.v = A.st->v.as<double>(); S.f = sl.f.as<double>();
    S.xq = sl.xq.as<double4>(); S.stype = sl.stype.as<int>(); S.perm = sl.perm.as<int>(); S.slot_tmp = sl.slot_tmp.as<int>();
    S.wrapn = sl.wrapn.as<int>(); S.xhold = sl.xhold.as<double>();
    S.cell_of = sl.cell_of.as<int>(); S.ckey = sl.ckey.as<int>(); S.cell_count = sl.cell_count.as<int>(); S.cell_start = sl.cell_start.as<int>();
    S.cell_fill = sl.cell_fill.as<int>(); S.numneigh = sl.numneigh.as<int>(); S.neigh = sl.neigh.as<int>();
    S.fs = sl.fs.as<double>(); S.fb = sl.fb.as<double>(); S.slot_of = sl.slot_of.as<int>(); S.tile_nj = sl.tile_nj.as<int>(); S.tile_jtab = sl.tile_jtab.as<int>(); S.tile_order = sl.tile_order.as<int>(); S.tile_wstart = sl.tile_wstart.as<int>(); S.virp = sl.virp.as<double>(); S.virb = sl.virb.as<double>();
    S.sfac = sl.sfac.as<double>(); S.kvec = sl.kvec.as<double>();
    S.sc = e->d_sc.as<SimScalars>() + i;
    if (S.nk > 0) {
      // layout per simulation: kn[3 nk] | krun[nk] | pad to 4 ints | kgrp[8 ngrp]
      koff[pos] = kpack.size();
      kpack.insert(kpack.end(), ew.kn.begin(), ew.kn.end());
      kpack.insert(kpack.end(), ew.krun.begin(), ew.krun.end());
      while (kpack.size() % 4) kpack.push_back(0);
      kpack.insert(kpack.end(), ew.kgrp.begin(), ew.kgrp.end());
      S.ngrp = (int)ew.kgrp.size() / 8;
      maxgrp = std::max(maxgrp, S.ngrp);
    }
    e->h_sims[pos] = S;
    { const auto tl4 = t_now(); t_lay[0] += t_ms(tl0, tl1); t_lay[1] += t_ms(tl1, tl2); t_lay[3] += t_ms(tl2, tl3); t_lay[3] += t_ms(tl3, tl4); }
    maxatoms = std::max(maxatoms, S.natoms); maxpad = std::max(maxpad, S.npad); maxcells = std::max(maxcells, S.ncells);
    maxk = std::max(maxk, S.nk);
    maxpoly = std::max(maxpoly, S.coul_npoly);
    for (int d = 0; d < 3; d++) mmax = std::max(mmax, S.kmaxd[d] + 1);
    maxb = std::max(maxb, S.nbonds); maxa = std::max(maxa, S.nangles); maxd = std::max(maxd, S.ndihedrals);
    maxi = std::max(maxi, S.nimpropers); maxs = std::max(maxs, S.nspecial); maxclus = std::max(maxclus, S.use_shake ? S.nclus : 0);
    maxunits = std::max(maxunits, S.use_shake ? S.nclus + S.nfree : S.natoms);
    maxsteps = std::max(maxsteps, A.nsteps);
  }
  // (Round 6 measured three ways of giving a launch that does not fill the chip more, shorter workgroups of k_pair -- every tile as 2 / 4 / 8
  // workgroups with a part of every row each; only the last replicas of a launch split that way; and the list kernels on a stream of their own
  // beside a first pair launch for the replicas whose rows stand -- and all three LOST at every batch size from 1 to 144 replicas: DESIGN.md 5.4,
  // profiles/r06_a_pair_parts_ab.log, r06_k_pair_tail_ab.log, r06_b_ab.log.  They were removed again; commit 992bf45 holds the code.)
  if ((size_t)64 * 3 * mmax * 16 + 4096 > 160 * 1024)
    return fail(e, SCEMA_MD_ERR_ARG, "k-space index range (|n| up to %d) too large for the LDS phase tables; raise cut_coul or loosen kspace_accuracy", mmax - 1);
  HIPCHK(e->d_kpack.ensure(kpack.size() * sizeof(int) + 64));
  if (!kpack.empty()) HIPCHK(hipMemcpyAsync(e->d_kpack.p, kpack.data(), kpack.size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
  for (int pos = 0; pos < ns; pos++) {
    SimDev &S = e->h_sims[pos];
    if (S.nk <= 0) continue;
    const int *base = e->d_kpack.as<int>() + koff[pos];
    S.kn = base;
    S.krun = base + 3 * (size_t)S.nk;
    S.kgrp = base + ((4 * (size_t)S.nk + 3) / 4) * 4;
  }
  // PPPM: four complex grids and the influence function per simulation.  The charge grids of the batch are contiguous, and so
  // are the field grids (three per simulation, simulation-major): one batched transform forward and ONE back for a launch
  // group whose simulations share the grid, which they do for one material
  int maxdims = 0;              // largest nx + ny + nz
  bool pppm_clean[MAXP] = {};   // per half: the charge grids hold zeros (the buffer is laid out anew for every run)
  std::vector<std::pair<int, int>> pppm_runs;   // (first position, count) of neighbours in the launch order that share a grid; none crosses a half
  if (maxgrid > 0) {
    HIPCHK(e->d_pppm.ensure((size_t)ns * maxgrid * (4 * sizeof(double2) + sizeof(double))));
    double *gbase = e->d_pppm.as<double>(), *ebase = gbase + (size_t)ns * maxgrid * 2, *fbase = gbase + (size_t)ns * maxgrid * 8;
    for (int pos = 0; pos < ns; pos++) {
      SimDev &S = e->h_sims[pos];
      S.pgrid = gbase + (size_t)pos * maxgrid * 2;
      S.pfield = ebase + (size_t)pos * maxgrid * 6;
      S.pgstride = (long long)maxgrid;
      S.pgf = fbase + (size_t)pos * maxgrid;
      maxdims = std::max(maxdims, S.pg[0] + S.pg[1] + S.pg[2]);
      const bool same = !pppm_runs.empty() && pos != hbeg[part_of(pos)] && S.pg[0] == e->h_sims[pos - 1].pg[0] && S.pg[1] == e->h_sims[pos - 1].pg[1] && S.pg[2] == e->h_sims[pos - 1].pg[2];
      if (same) pppm_runs.back().second += 1;
      else pppm_runs.push_back({pos, 1});
    }
  }
  const bool pppm_in_lds = maxgrid > 0 && maxgrid <= mdk_pppm_solve_max() && (3 * (size_t)maxgrid + (size_t)maxdims) * 16 <= 150 * 1024 && !scema_env("SCEMA_MD_PPPM_FFT");
  // Batched 3-d Z2Z plans over grids that lie maxgrid complex elements apart (the charge grids of neighbouring simulations, and
  // all their field grids: three per simulation, simulation-major).  A plan owns work space, so each stream has its own.
  auto pppm_plan = [&](const int pg[3], int batch, hipStream_t st, hipfftHandle &plan) -> int {
    if ((long long)maxgrid > 0x7fffffffLL) return fail(e, SCEMA_MD_ERR_ARG, "PPPM grid of %d points is too large", maxgrid);
    // A plan owns a work area and is bound to a stream when it runs: ONE PER STREAM that may run it.  (Until round 6 the key knew the main
    // stream, the side stream and "the other one": with three or four part batches two parts shared the plans of their common mesh sizes --
    // two replicas with the same mesh beyond the in-LDS solve, one in each, transformed through one work area at the same time.)
    const int sk = st == e->stream ? 0 : st == e->stream2 ? 1 : st == e->stream3 ? 2 : st == e->rx_side1 ? 3 : -1;
    if (sk < 0) return fail(e, SCEMA_MD_ERR_ARG, "PPPM transform on a stream the engine does not know");
    const std::array<int, 6> key = {pg[0], pg[1], pg[2], batch, sk, maxgrid};
    auto it = e->pppm_plans.find(key);
    if (it == e->pppm_plans.end()) {
      hipfftHandle h;
      int n[3] = {pg[2], pg[1], pg[0]};   // slowest dimension first
      // embed = the grid itself; the distance between consecutive grids is the batch's stride, not the grid's size
      if (hipfftPlanMany(&h, 3, n, n, 1, maxgrid, n, 1, maxgrid, HIPFFT_Z2Z, batch) != HIPFFT_SUCCESS)
        return fail(e, SCEMA_MD_ERR_DEVICE, "hipfftPlanMany failed for a %d x %d x %d grid, batch %d", pg[0], pg[1], pg[2], batch);
      it = e->pppm_plans.emplace(key, h).first;
    }
    plan = it->second;
    return SCEMA_MD_OK;
  };
  // reciprocal part by PPPM for the simulations [pos0, pos0 + na) of a launch group of `full` (md_pppm.hip); after force_stage
  auto pppm_stage = [&](hipStream_t st, int pos0, int na, bool new_box, int add = 1) -> int {
    if (maxgrid <= 0 || na <= 0) return SCEMA_MD_OK;
    const SimDev *Dp = e->d_sims.as<SimDev>() + pos0;
    bool &clean = pppm_clean[part_of(pos0)];
    mdk_pppm_spread(st, Dp, na, maxgrid, maxatoms, clean ? 1 : 0, padx_ok ? maxgridp : 0);
    clean = false;
    if (pppm_in_lds) {   // small grids: the whole solve in one launch, in LDS (md_pppm.hip k_pppm_solve); it leaves the charge grids zeroed
      if (new_box) mdk_pppm_gf(st, Dp, na, maxgrid);
      mdk_pppm_solve(st, Dp, na, maxgrid, maxdims);
      clean = true;
      mdk_pppm_force(st, Dp, na, maxgrid, maxatoms, add, 1);
      return SCEMA_MD_OK;
    }
    auto transform = [&](bool fields, int dir) -> int {   // the charge grids forward, or the three field grids of every simulation back
      const bool serial_fft = false;
      for (const auto &run : pppm_runs) {
        if (run.first + run.second <= pos0 || run.first >= pos0 + na) continue;   // outside this launch group, or none of it is active any more
        const SimDev &S0 = e->h_sims[run.first];
        if (S0.pg[0] == 0) continue;
        const int per = fields ? 3 : 1;
        for (int k = 0; k < (serial_fft ? run.second * per : 1); k++) {
          hipfftHandle plan;
          const int rc = pppm_plan(S0.pg, serial_fft ? 1 : per * run.second, st, plan);
          if (rc) return rc;
          double *g = (fields ? S0.pfield : S0.pgrid) + (serial_fft ? 2 * (size_t)k * S0.pgstride : 0);
          if (hipfftSetStream(plan, st) != HIPFFT_SUCCESS || hipfftExecZ2Z(plan, (hipfftDoubleComplex *)g, (hipfftDoubleComplex *)g, dir) != HIPFFT_SUCCESS)
            return fail(e, SCEMA_MD_ERR_DEVICE, "hipfftExecZ2Z failed");
        }
      }
      return SCEMA_MD_OK;
    };
    int rc = transform(false, HIPFFT_FORWARD);
    if (rc) return rc;
    if (new_box) mdk_pppm_gf(st, Dp, na, maxgrid);
    mdk_pppm_poisson(st, Dp, na, maxgrid);
    if ((rc = transform(true, HIPFFT_BACKWARD))) return rc;
    mdk_pppm_force(st, Dp, na, maxgrid, maxatoms, add);
    return SCEMA_MD_OK;
  };
  // With one launch group and the side stream, the whole PPPM chain of a step (it needs the positions only) runs next to
  // k_pair and the bonded kernel: its forces are stored in SimDev::f, and k_ewald_force, which assembles the force of the
  // step, adds them after the join.  Otherwise the chain follows the assembly and adds to it.  PE-10k, evaluations per second
  // with the chain on the side stream / inline: 8 replicas 210 / 183, 72: 336 / 333, 576: 369 / 368; a single replica 39.7 / 41.6
  // (round 2: its k_pair does not fill the chip and the fork/join is pure latency -- with round 5's kernels, where the chain is 67 us of
  // dependent launches beside 45 us of k_pair + k_bonded, 16.7 against 18.3 ms per evaluation).  So: batches of up to 255 replicas
  // (SCEMA_MD_PPPM_SIDE_MIN: the smallest); a batch that fills the chip many times over gains nothing, and inline its k_pair launches
  // are timed and profiled undisturbed.
  static const int side_min = scema_env("SCEMA_MD_PPPM_SIDE_MIN") ? atoi(scema_env("SCEMA_MD_PPPM_SIDE_MIN")) : 1;
  const bool pppm_side = maxgrid > 0 && nhalf == 1 && e->stream2 != nullptr && ns >= side_min && ns < 256;
  auto pppm_fork = [&](hipStream_t st, int pos0, int na, bool new_box, bool with_bonded = false) -> int {
    if (!pppm_side) return SCEMA_MD_OK;
    HIPCHK(hipEventRecord(e->ev_fork, st));
    HIPCHK(hipStreamWaitEvent(e->stream2, e->ev_fork, 0));
    const int rc = pppm_stage(e->stream2, pos0, na, new_box, 0);
    if (rc) return rc;
    if (with_bonded) mdk_bonded(e->stream2, e->d_sims.as<SimDev>() + pos0, na, maxbt, maxloc, maxcoef, 0);   // (needs the positions only, like the chain before it)
    HIPCHK(hipEventRecord(e->ev_join, e->stream2));
    return SCEMA_MD_OK;
  };
  HIPCHK(e->d_sims.ensure((size_t)ns * sizeof(SimDev)));
  HIPCHK(hipMemcpyAsync(e->d_sims.p, e->h_sims.data(), (size_t)ns * sizeof(SimDev), hipMemcpyHostToDevice, e->stream));
  const auto t_laid_out = std::chrono::steady_clock::now();
  const SimDev *D = e->d_sims.as<SimDev>();
  hipStream_t hs[MAXP] = {e->stream, nhalf >= 2 ? e->stream3 : e->stream, e->stream2, e->rx_side1};
  if (nhalf >= 2) {   // the other streams start behind the uploads
    HIPCHK(hipEventRecord(e->ev_up, e->stream));
    for (int h = 1; h < nhalf; h++) HIPCHK(hipStreamWaitEvent(hs[h], e->ev_up, 0));
  }
  const int ev = (spec.sample || spec.ev_always || (spec.nh && spec.npt)) ? 1 : 0;   // the barostat needs the virial of every step
  const bool allow_side = nhalf == 1;
  // The replicas of a launch (a half batch where the batch runs as two) rebuild their rows together (k_cell_build): same
  // box, own trigger / together, evaluations/s: 2 replicas 118.3 / 122.5, 4: 189.0 / 204.6, 9: 263.8 / 297.6, 18: 319.0 / 348.4, 24: 331.7 / 357.1,
  // 36: 378.3 / 405.7, 72: 425.4 / 438.2, 144: 447.8 / 450.3, 288: 458.4 / 458.6, 576: 462.8 / 463.3 (profiles/r06_g_ab.log, r06_h_ab.log).
  // A common trigger fires at the earliest of the launch's replicas: with 288 of them per launch a list lives 12.6 instead of 18.2 steps, and
  // the builds that adds (at full occupancy: +4 % list-build time) buy nothing where scattered rebuilds already find the chip full.  So: launches
  // of fewer than 128 replicas.  SCEMA_MD_REBUILD_TOGETHER = 0 / 1: every replica on its own trigger / together at every size.
  static const int together_env = scema_env("SCEMA_MD_REBUILD_TOGETHER") ? atoi(scema_env("SCEMA_MD_REBUILD_TOGETHER")) : -1;
  const bool nb_together = ns > 1 && (together_env < 0 ? hcnt[0] < 128 : together_env != 0);
  // ---- setup (step 0) ----
  for (int h = 0; h < nhalf; h++) {
    hipStream_t st = hs[h];
    const SimDev *Dh = D + hbeg[h];
    const int nh = hcnt[h];
    mdk_phase_init(st, Dh, nh);
    if (any_validate) mdk_keep_validate(st, Dh, nh, maxatoms);
    mdk_neighbor(st, Dh, nh, maxatoms, maxpad, maxcells, maxrow, maxcapj, true, nb_together);
    { const int rcp = pppm_fork(st, hbeg[h], nh, true); if (rcp) return rcp; }
    mdk_pair(st, Dh, nh, maxcells, maxcapj, ev, spec.ev_always, maxpoly, P.cut_coul <= P.cut_lj);
    HIPCHK(force_stage(e, st, allow_side, Dh, nh, maxbt, maxloc, maxcoef, maxatoms, maxk, mmax, maxgrp, spec.ev_always, (ev && !spec.ev_always) ? 1 : 0, pppm_side));
    if (!pppm_side) { const int rcp = pppm_stage(st, hbeg[h], nh, true); if (rcp) return rcp; }
    if (!spec.static_only) mdk_shake(st, Dh, nh, maxclus, 0.5);
    mdk_final_integrate(st, Dh, nh, maxatoms, 0);
    if (spec.nh) mdk_setup_post_nh(st, Dh, nh);
    else mdk_setup_post(st, Dh, nh);
  }
  if (spec.minimize) {
    // min_style sd: every replica runs its own line search, decided on the device between two force evaluations; the host
    // only looks now and then whether all of them have stopped.  x0 and the search direction live in the slot's backup arrays.
    hipStream_t st = e->stream;
    std::vector<double *> ptrs(2 * (size_t)ns);
    for (int pos = 0; pos < ns; pos++) {
      Slot &sl = *e->slots[order[pos]];
      ptrs[pos] = sl.xbak.as<double>();
      ptrs[ns + pos] = sl.vbak.as<double>();
      HIPCHK(hipMemsetAsync(sl.vbak.p, 0, 3 * (size_t)e->h_sims[pos].natoms * 8, st));
    }
    HIPCHK(e->d_minptr.ensure(ptrs.size() * sizeof(double *)));
    HIPCHK(hipMemcpyAsync(e->d_minptr.p, ptrs.data(), ptrs.size() * sizeof(double *), hipMemcpyHostToDevice, st));
    double *const *x0s = e->d_minptr.as<double *>(), *const *hsd = e->d_minptr.as<double *>() + ns;
    mdk_min_reduce(st, D, ns, maxatoms, hsd);
    mdk_min_decide(st, D, ns);
    const long long cap = (long long)spec.min_maxeval + 2LL * spec.min_maxiter + 8;
    bool all_done = false;
    for (long long ev_n = 0; ev_n < cap && !all_done;) {
      for (int r = 0; r < 16; r++, ev_n++) {
        mdk_min_pre(st, D, ns);
        mdk_min_move(st, D, ns, maxatoms, x0s, hsd);
        mdk_neighbor(st, D, ns, maxatoms, maxpad, maxcells, maxrow, maxcapj);
        mdk_pair(st, D, ns, maxcells, maxcapj, 1, 1, maxpoly, P.cut_coul <= P.cut_lj);
        HIPCHK(force_stage(e, st, false, D, ns, maxbt, maxloc, maxcoef, maxatoms, maxk, mmax, maxgrp, 1, 0));
        { const int rcp = pppm_stage(st, 0, ns, false); if (rcp) return rcp; }
        mdk_min_reduce(st, D, ns, maxatoms, hsd);
        mdk_min_decide(st, D, ns);
      }
      HIPCHK(hipMemcpyAsync(e->h_sc.data(), e->d_sc.p, (size_t)ns * sizeof(SimScalars), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      all_done = true;
      for (int i = 0; i < ns; i++) {
        if (e->h_sc[i].overflow) all_done = true;
        else if (e->h_sc[i].min_phase != 4) { all_done = false; }
      }
      for (int i = 0; i < ns; i++) if (e->h_sc[i].overflow) all_done = true;
    }
    HIPCHK(hipMemcpyAsync(e->h_sc.data(), e->d_sc.p, (size_t)ns * sizeof(SimScalars), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    int fault_m = 0;
    for (int i = 0; i < ns; i++) fault_m |= e->h_sc[i].overflow;
    if (fault_m & 16) return fail(e, SCEMA_MD_ERR_ARG, "a simulation became unstable during the minimisation (non-finite positions)");
    e->overflow_bits = (fault_m & 1) ? (fault_m & (4 | 8)) : 0;
    if (fault_m & 1) return SCEMA_MD_ERR_OVERFLOW;
    if (!all_done) return fail(e, SCEMA_MD_ERR_ARG, "minimiser did not stop within its evaluation budget");
    return SCEMA_MD_OK;
  }
  // ---- steps ----
  const bool prof = e->p.profile != 0;
  size_t ev_used = 0;
  std::vector<std::pair<int, int>> launch_sims;   // per timed pair launch: (first position, simulations)
  // one MD step of the first `na` simulations of half h, as a sequence of launches on that half's stream
  // Small batches are launch-bound (a single replica: ~20 launches of 5-35 us per step), so there k_initial_integrate also writes the
  // slot-ordered records that k_pack would (scattered 16-byte stores: for 576 replicas that costs what the separate, coalesced
  // k_pack costs -- 304 against 170 + 125 us -- so large batches keep k_pack)
  // (by the size of a LAUNCH instead -- the halves of a batch of 36 to 72 -- it loses 0.3-1 %: profiles/r06_s_fusepack_ab.log)
  const bool fuse_pack = ns <= 32;
  // Likewise the tail of the force stage of steps without a per-atom reciprocal sum (PPPM or no k-space; Verlet / fix nvt, production
  // virial): one pass (k_finish) instead of k_ewald_force + k_shake + k_final_integrate -- two launches less for a small batch
  // (a single replica: 14.2 against 17.4 us); a thread per SHAKE cluster gathers less well than the three kernels stream, so large
  // batches keep them (576 replicas: 452 against 426 us).  SCEMA_MD_FUSED_TAIL = 0 / 1 forces either.
  static const int fused_tail_env = scema_env("SCEMA_MD_FUSED_TAIL") ? atoi(scema_env("SCEMA_MD_FUSED_TAIL")) : -1;
  const bool fused_tail = (fused_tail_env < 0 ? hcnt[0] <= 32 : fused_tail_env != 0) && !spec.nh && maxk == 0 && !spec.ev_always;   // (by the size of a launch: a batch of 36 as two halves +0.9 %, of 72 +-0, of 144 -0.3 %, profiles/r06_i_ab.log)
  // The bonded kernel behind the PPPM chain on the side stream on steps whose chain is short (no new influence function), for batches of 8
  // replicas and more, where the pair kernel is the longer of the step's two chains of dependent launches: +4 % at 9 replicas, +2 % at 18;
  // below 8 the PPPM chain is the longer one and the move costs 4-8 % (profiles/r06_d_ab.log).  SCEMA_MD_BONDED_SIDE = 0: off.
  // (k_post's work inside k_finish -- the replica's last workgroup, found by a ticket, does the end of the step -- was built and measured in
  // the same series: nothing at 1-4 replicas, where k_finish, k_post, k_remap and k_initial_integrate already run back to back without a gap,
  // and -2.5 / -3.6 % at 9 / 18 replicas, where the device-scope release and acquire around the ticket write back and invalidate the XCD's L2
  // with the replica's freshly stored velocities and forces in it.  Removed.)
  // (The bonded tiles of such a batch as extra workgroups of the k_pair launch -- on the CUs its 180 pair tiles leave idle -- take the kernel and
  // its launch gap off the main stream, 89 -> 72 us per step, and leave the PPPM chain beside it, fork 11 + 54 + join 11 us, the longer one:
  // 76.6 against 76.2 evaluations/s for one replica, 208 / 204 for four, 257 / 259 for seven.  profiles/r06_w_bonded_in_pair_ab.log.  Removed.)
  // (The bonded kernel of a batch under 8 replicas on a THIRD stream, beside both k_pair and the PPPM chain -- on paper 17 us off a lone replica's
  // 127 us step -- lost: 66.5 against 77.1 evaluations/s for one replica, 188.7 / 201.2 for four, 290.2 / 294.2 for nine.  A second fork and join
  // per step costs more than the 11 us kernel it hides.  profiles/r06_v_bonded_third_ab.log.  Removed.)
  static const bool bonded_side_on = !(scema_env("SCEMA_MD_BONDED_SIDE") && atoi(scema_env("SCEMA_MD_BONDED_SIDE")) == 0);
  static const int bonded_side_min = scema_env("SCEMA_MD_BONDED_SIDE_MIN") ? atoi(scema_env("SCEMA_MD_BONDED_SIDE_MIN")) : 8;
  // (The pair kernel as PERSISTENT workgroups -- one 1 024-thread workgroup per CU for the whole launch, two tiles in LDS, rows taken off LDS
  // counters, no barrier between a tile's rows and its flush -- was built in two forms in round 6 to recover the quarter of a wave's life that
  // k_pair spends outside its row loop, and lost: 405 / 342 against 463 / 451 evaluations/s at 576 replicas.  k_pair sits at 120 of 128 vector
  // registers; the persistent shell's dozen extra live scalars tip the allocation into scratch reloads inside the row loop, whose every wait then
  // covers all loads in flight.  profiles/r06_q_persistent_pair.txt has the wave clocks and the ISA counts; commit 2371531 holds the code.)
  auto launch_step = [&](int h, int na, bool timed) -> int {
    hipStream_t st = hs[h];
    const SimDev *Dh = D + hbeg[h];
    if (spec.nh) { mdk_pre_nh(st, Dh, na); mdk_initial_integrate_nh(st, Dh, na, maxatoms); }
    else mdk_initial_integrate(st, Dh, na, maxatoms, fuse_pack);   // (its k_pre: at the end of the step before, in k_post; for step 1 below)
    // the PPPM chain needs the new positions only: it leaves for its side stream before the list kernels are issued, not behind them
    const bool bonded_side = bonded_side_on && pppm_side && fused_tail && ns >= bonded_side_min && !(spec.deform || (spec.nh && spec.npt));
    { const int rcp = pppm_fork(st, hbeg[h], na, spec.deform || (spec.nh && spec.npt), bonded_side); if (rcp) return rcp; }
    mdk_neighbor(st, Dh, na, maxatoms, maxpad, maxcells, maxrow, maxcapj, spec.nh != 0 || !fuse_pack, nb_together);
    if (timed) {
      if (ev_used + 2 > e->ev_pool.size()) {
        hipEvent_t a, b;
        HIPCHK(hipEventCreate(&a));
        HIPCHK(hipEventCreate(&b));
        e->ev_pool.push_back(a);
        e->ev_pool.push_back(b);
      }
      HIPCHK(hipEventRecord(e->ev_pool[ev_used], st));
    }
    mdk_pair(st, Dh, na, maxcells, maxcapj, ev, spec.ev_always, maxpoly, P.cut_coul <= P.cut_lj);
    if (timed) {
      HIPCHK(hipEventRecord(e->ev_pool[ev_used + 1], st));
      ev_used += 2;
      launch_sims.push_back({hbeg[h], na});
    }
    if (fused_tail) {
      // no per-atom reciprocal sum: the bonded kernel, the PPPM chain (its forces stored in f, from the side stream or here), then
      // assembly of f, fix shake and the second half-kick in one pass (k_finish)
      if (!bonded_side) mdk_bonded(st, Dh, na, maxbt, maxloc, maxcoef, 0);
      if (pppm_side) HIPCHK(hipStreamWaitEvent(st, e->ev_join, 0));
      else { const int rcp = pppm_stage(st, hbeg[h], na, spec.deform, 0); if (rcp) return rcp; }
      mdk_finish(st, Dh, na, maxunits, ev, maxgrid > 0 ? 1 : 0);
    } else {
      HIPCHK(force_stage(e, st, allow_side, Dh, na, maxbt, maxloc, maxcoef, maxatoms, maxk, mmax, maxgrp, spec.ev_always, (ev && !spec.ev_always) ? 1 : 0, pppm_side));
      if (!pppm_side) { const int rcp = pppm_stage(st, hbeg[h], na, spec.deform || (spec.nh && spec.npt)); if (rcp) return rcp; }
      mdk_shake(st, Dh, na, maxclus, 1.0);
      mdk_final_integrate(st, Dh, na, maxatoms, 1);
    }
    if (spec.nh) mdk_post_nh(st, Dh, na);
    else mdk_post(st, Dh, na, 1);
    if (spec.deform) mdk_remap(st, Dh, na, maxatoms);
    return SCEMA_MD_OK;
  };
  auto active = [&](int h, int step) {   // active prefix of half h at this step (sorted by nsteps)
    int na = 0;
    while (na < hcnt[h] && e->h_sims[hbeg[h] + na].nsteps >= step) na++;
    return na;
  };
  // (hipGraph replay of the steps that share an active count was measured on ROCm 7.2 / MI355X -- ms per update of 1 / 72 PE-10k
  // replicas: plain launches 25.8 / 236.7, replay 26.4 / 236.9, with the side stream inside the graph 55.2 / 251.3 -- and removed in
  // round 4: profiles/HISTORY.md)
  // box flips (fix deform, flip yes): step -> positions that flip after it
  std::map<int, std::vector<std::pair<int, int>>> flip_at;
  for (int pos = 0; pos < ns; pos++)
    for (size_t k = 0; k < flips[pos].size(); k++)
      if (flips[pos][k].step < e->h_sims[pos].nsteps) flip_at[flips[pos][k].step].push_back({pos, (int)k});
  std::vector<std::unique_ptr<DevBuf>> flip_bufs;          // k-vector tables in the new reciprocal basis, alive until the run has drained
  std::vector<std::unique_ptr<std::vector<int>>> flip_host;
  std::vector<std::unique_ptr<SimDev>> flip_desc;
  if (!spec.nh)   // the k_pre of step 1, for the simulations that have a step 1; every later one rides on the k_post of the step before
    for (int h = 0; h < nhalf; h++) {
      const int n1 = active(h, 1);
      if (n1 > 0) mdk_pre(hs[h], D + hbeg[h], n1);
    }
  for (int step = 1; step <= maxsteps;) {
    const int na = active(0, step);
    if (na == 0) break;
    int run_len = e->h_sims[na - 1].nsteps - step + 1;   // steps until the active prefix shrinks (sorted by nsteps)
    int nact[MAXP] = {na}, nsum = na;
    for (int h = 1; h < nhalf; h++) {
      nact[h] = active(h, step);
      nsum += nact[h];
      if (nact[h] > 0) run_len = std::min(run_len, e->h_sims[hbeg[h] + nact[h] - 1].nsteps - step + 1);
    }
    {
      auto nxt = flip_at.lower_bound(step);
      if (nxt != flip_at.end()) run_len = std::min(run_len, nxt->first - step + 1);   // the launch group ends with the flipping step
    }
    for (int r = 0; r < run_len; r++) {
      for (int h = 0; h < nhalf; h++) {
        if (nact[h] == 0) continue;
        const int rc_l = launch_step(h, nact[h], prof);
        if (rc_l) return rc_l;
      }
    }
    e->prof.md_steps += (long long)nsum * run_len;
    step += run_len;
    // flips detected at the end of step - 1: between the two steps the box takes its flipped tilts, the list rebuild of
    // the next step is forced and the k-vector list is re-expressed in the new reciprocal basis (same vectors:
    // n2 += f_xy n1, n3 += f_yz n2 + f_xz n1), all stream-ordered behind the launches of step - 1
    auto fl = flip_at.find(step - 1);
    if (fl != flip_at.end())
      for (const auto &pk : fl->second) {
        const int pos = pk.first;
        const FlipEvent &fe = flips[pos][pk.second];
        const int h = part_of(pos);
        SimDev &S = e->h_sims[pos];
        EwaldSetup &ew = ews[pos];
        if (S.nk > 0) {
          for (int k = 0; k < S.nk; k++) {
            const int n1 = ew.kn[3 * k], n2 = ew.kn[3 * k + 1], n3 = ew.kn[3 * k + 2];
            ew.kn[3 * k + 1] = n2 + fe.nflip[0] * n1;
            ew.kn[3 * k + 2] = n3 + fe.nflip[2] * n2 + fe.nflip[1] * n1;
          }
          ewald_tables(ew);
          flip_host.emplace_back(new std::vector<int>());
          std::vector<int> &hk = *flip_host.back();
          hk.insert(hk.end(), ew.kn.begin(), ew.kn.end());
          hk.insert(hk.end(), ew.krun.begin(), ew.krun.end());
          while (hk.size() % 4) hk.push_back(0);
          const size_t goff = hk.size();
          hk.insert(hk.end(), ew.kgrp.begin(), ew.kgrp.end());
          flip_bufs.emplace_back(new DevBuf());
          HIPCHK(flip_bufs.back()->ensure(hk.size() * sizeof(int) + 64));
          HIPCHK(hipMemcpyAsync(flip_bufs.back()->p, hk.data(), hk.size() * sizeof(int), hipMemcpyHostToDevice, hs[h]));
          const int *base = flip_bufs.back()->as<int>();
          S.kn = base;
          S.krun = base + 3 * (size_t)S.nk;
          S.kgrp = base + goff;
          S.ngrp = (int)ew.kgrp.size() / 8;
          for (int d = 0; d < 3; d++) { S.kmaxd[d] = ew.kmaxd[d]; mmax = std::max(mmax, S.kmaxd[d] + 1); }
          maxgrp = std::max(maxgrp, S.ngrp);
          if ((size_t)64 * 3 * mmax * 16 + 4096 > 160 * 1024)
            return fail(e, SCEMA_MD_ERR_ARG, "k-space index range after a box flip (|n| up to %d) too large for the LDS phase tables", mmax - 1);
          flip_desc.emplace_back(new SimDev(S));   // the source of an asynchronous upload must not change under it
          HIPCHK(hipMemcpyAsync(e->d_sims.as<SimDev>() + pos, flip_desc.back().get(), sizeof(SimDev), hipMemcpyHostToDevice, hs[h]));
        }
        mdk_flip(hs[h], D + pos, fe.tilt[0], fe.tilt[1], fe.tilt[2]);
        e->prof.box_flips += 1;
      }
  }
  for (int h = 0; h < nhalf; h++) mdk_phase_end(hs[h], D + hbeg[h], hcnt[h], maxatoms);
  for (int h = 1; h < nhalf; h++) {   // (an event of its own per part)
    HIPCHK(hipEventRecord(e->md_part_done[h - 1], hs[h]));
    HIPCHK(hipStreamWaitEvent(e->stream, e->md_part_done[h - 1], 0));
  }
  hipStream_t st = e->stream;
  HIPCHK(hipMemcpyAsync(e->h_sc.data(), e->d_sc.p, (size_t)ns * sizeof(SimScalars), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipGetLastError());
  if (prof) {
    // algorithmic bytes of one pair launch: per simulation N*(4*nbar + 56) + 48 with
    // nbar = stored neighbours per atom of the (full) list
    double per_sim_bytes = 0.0;  // averaged over the batch; active prefix differs only for ragged nts
    std::vector<double> simbytes(ns);
    for (int pos = 0; pos < ns; pos++) {
      const int i = order[pos];
      simbytes[pos] = 4.0 * (double)e->h_sc[i].nentries_ref + 56.0 * e->h_sims[pos].natoms + 48.0;   // the reference's list radius, whatever the skin used
      per_sim_bytes += simbytes[pos];
    }
    (void)per_sim_bytes;
    for (size_t l = 0; l < launch_sims.size(); l++) {
      float ms = 0.f;
      HIPCHK(hipEventElapsedTime(&ms, e->ev_pool[2 * l], e->ev_pool[2 * l + 1]));
      e->prof.pair_ms += ms;
      e->prof.pair_launches += 1;
      e->prof.pair_sims += launch_sims[l].second;
      for (int pos = launch_sims[l].first; pos < launch_sims[l].first + launch_sims[l].second; pos++) e->prof.pair_alg_bytes += simbytes[pos];
    }
    e->prof.pair_union_ms += event_union_ms(e->ev_pool, launch_sims.size());
  }
  if (scema_env("SCEMA_MD_TIMING") && ns > 0) {
    const SimScalars &c = e->h_sc[0];
    const SimDev &S0 = e->h_sims[0];
    fprintf(stderr, "[scema_md] sim 0: cells %dx%dx%d, j table max %d of %d, row max %d of %d, row entries/cluster %.1f, listed pairs/atom %.1f, builds %d\n",
            S0.nc[0], S0.nc[1], S0.nc[2], c.maxj_seen, S0.capj, c.maxneigh_seen, S0.maxneigh, (double)c.nrowent / (S0.npad / MD_CLUSTER),
            (double)c.nentries / S0.natoms, c.nbuilds);
    fprintf(stderr, "[scema_md] host: %.2f ms laying out %d simulations before the first launch of this run (k-space set-up on host threads %.2f, box range %.2f, cell grid %.2f, rest of the loop %.2f)\n",
            std::chrono::duration<double, std::milli>(t_laid_out - t_enter).count(), ns, t_kspace_ms, t_lay[0], t_lay[1], t_lay[3]);
    fprintf(stderr, "[scema_md] sim 0: far skin band walked on %d of %d steps; list skin %.2f A\n", c.nfar_steps, c.step, S0.skin);
#ifdef PAIR_COUNT
    fprintf(stderr, "[scema_md] k_pair lanes (sim 0, this run): %llu wave-chunks (%llu with work); atom blocks run %llu = %.2f per working chunk, %.1f lanes of 64 in them; "
            "LJ block run in %llu of them with %.1f lanes; coulomb block in %llu with %.1f lanes; pairs inside the LJ cutoff %llu, inside the coulomb cutoff %llu\n",
            c.dbg[0], c.dbg[1], c.dbg[2], (double)c.dbg[2] / std::max(1ull, c.dbg[1]), (double)c.dbg[3] / std::max(1ull, c.dbg[2]), c.dbg[5],
            (double)c.dbg[4] / std::max(1ull, c.dbg[5]), c.dbg[7], (double)c.dbg[6] / std::max(1ull, c.dbg[7]), c.dbg[4], c.dbg[6]);
#endif
#ifdef PAIR_TIMING
    if (c.dbg2[7])
      fprintf(stderr, "[scema_md] k_pppm_solve clocks (sim 0, thread 0, mean per launch): grid in %.0f, forward passes %.0f, spectra %.0f + %.0f, inverse passes %.0f + %.0f, out + sums %.0f\n",
              (double)c.dbg2[0] / c.dbg2[7], (double)c.dbg2[1] / c.dbg2[7], (double)c.dbg2[2] / c.dbg2[7], (double)c.dbg2[4] / c.dbg2[7], (double)c.dbg2[3] / c.dbg2[7],
              (double)c.dbg2[5] / c.dbg2[7], (double)c.dbg2[6] / c.dbg2[7]);
    fprintf(stderr, "[scema_md] k_pair wave clocks (sim 0, mean per wave): prologue %.0f, rows %.0f, barrier wait %.0f, flush %.0f (%llu waves)\n",
            (double)c.dbg[0] / c.dbg[4], (double)c.dbg[1] / c.dbg[4], (double)c.dbg[2] / c.dbg[4], (double)c.dbg[3] / c.dbg[4], c.dbg[4]);
    {   // (the same clocks summed over the whole batch)
      unsigned long long a[6] = {0, 0, 0, 0, 0, 0};
      for (int i = 0; i < ns; i++) for (int k = 0; k < 6; k++) a[k] += e->h_sc[i].dbg[k];
      if (a[4]) fprintf(stderr, "[scema_md] pair kernel wave clocks (whole batch, mean per wave and tile visit): prologue %.0f, rows %.0f, wait %.0f, flush %.0f (of which staging %.0f) (%llu visits)\n",
                        (double)a[0] / a[4], (double)a[1] / a[4], (double)a[2] / a[4], (double)a[3] / a[4], (double)a[5] / a[4], a[4]);
    }
    if (c.nbuilds > 0) {
      const double nw = (double)c.nbuilds * S0.ncells * MD_TILE_WAVES;
      fprintf(stderr, "[scema_md] k_neigh_build wave clocks (sim 0, mean per wave and build): table %.0f (boxes and runs %.0f, candidates %.0f), rows %.0f, schedule %.0f; %llu waves of %.0f; %.2f rows per wave of %.1f chunks\n",
              (double)c.dbg[5] / nw, (double)c.dbg[8] / nw, (double)(c.dbg[5] - c.dbg[8]) / nw, (double)c.dbg[6] / nw, (double)c.dbg[7] / nw, c.dbg[9], nw,
              (double)c.dbg[11] / nw, (double)c.dbg[10] / std::max(1ull, c.dbg[11]));
      const double nr = (double)std::max(1ull, c.dbg[11]);
      fprintf(stderr, "[scema_md] k_neigh_build per row (sim 0, cycles): set-up %.0f, chunk loop %.0f = %.0f per chunk, row end %.0f; of %.1f chunks %.2f walk exclusion lists, %.2f are own-cell chunks\n",
              (double)c.dbg[12] / nr, (double)c.dbg[13] / nr, (double)c.dbg[13] / std::max(1ull, c.dbg[10]), (double)c.dbg[14] / nr, (double)c.dbg[10] / nr,
              (double)c.dbg[15] / nr, (double)c.dbg[16] / nr);
    }
#endif
  }
  int fault = 0;
  for (int i = 0; i < ns; i++) {
    fault |= e->h_sc[i].overflow;
    e->prof.neigh_builds += e->h_sc[i].nbuilds;
    e->prof.unique_pairs_sum += 0.5 * (double)e->h_sc[i].nentries_ref;
    e->prof.unique_pairs_n += 1;
  }
  // A capacity that overflowed comes first: the run went on with truncated rows or tables (nothing is written past a capacity, pairs are
  // missing), so an instability or a stretched special pair later in the same run is its consequence, not the caller's input -- the retry
  // with grown capacities starts from the backup and reports them if they are real.
  e->overflow_bits = fault;
  e->overflow_need_j = e->overflow_need_row = 1.0;
  {   // (what the builds of this run saw -- a table's count runs on past its capacity, a row's stops a chunk beyond -- against the smallest capacity of the launch)
    int seen_j = 0, seen_row = 0, cap_j = 1 << 30, cap_row = 1 << 30;
    for (int i = 0; i < ns; i++) {
      seen_j = std::max(seen_j, e->h_sc[i].maxj_seen); seen_row = std::max(seen_row, e->h_sc[i].maxneigh_seen);
      cap_j = std::min(cap_j, std::max(1, e->h_sims[i].capj)); cap_row = std::min(cap_row, std::max(1, e->h_sims[i].maxneigh));
    }
    e->overflow_need_j = std::max(1.0, (double)seen_j / cap_j);
    e->overflow_need_row = std::max(1.0, (double)seen_row / cap_row);
  }
  if (fault & 1) return SCEMA_MD_ERR_OVERFLOW;
  if (fault & 16) return fail(e, SCEMA_MD_ERR_ARG, "a simulation became unstable (non-finite or runaway atom positions): overlapping atoms or parameters far from the replica's equilibrium");
  if (fault & 2) return fail(e, SCEMA_MD_ERR_ARG, "an excluded (special) pair stretched beyond the exclusion gate; topology or state is broken");
  if (fault & 64) return SCEMA_MD_ERR_OVERFLOW;   // the barostat took the box out of the range this segment was laid out for
  for (int i = 0; i < ns; i++) {   // the rows on the device hold for the positions this run ended at
    ListSig &g = e->slots[i]->sig;
    const SimScalars &c = e->h_sc[i];
    g.valid = true;
    g.state = sims[i].st->id;
    std::memcpy(g.corners_hold, c.corners_hold, sizeof g.corners_hold);
    g.ago = c.ago; g.maxj_seen = c.maxj_seen; g.nentries = c.nentries; g.nentries_ref = c.nentries_ref; g.nrowent = c.nrowent;
  }
  return SCEMA_MD_OK;
}

int prepare_slots(scema_md_engine *e, std::vector<ActiveSim> &sims) {
  const int ns = (int)sims.size();
  while ((int)e->slots.size() < ns) e->slots.emplace_back(new Slot());
  HIPCHK(e->d_sc.ensure((size_t)std::max(ns, 1) * sizeof(SimScalars)));
  e->h_sc.assign(ns, SimScalars());
  for (int i = 0; i < ns; i++) {
    std::memset(&e->h_sc[i], 0, sizeof(SimScalars));
    std::memcpy(e->h_sc[i].box, sims[i].st->box, 9 * sizeof(double));
    e->h_sc[i].vscale = 1.0;
    // the slot may still hold this state's neighbour rows from the update before: their scalars come back with them (run_phase and the
    // device decide whether the rows are kept; a slot that last served another state leaves the zeros, which force the build)
    const ListSig &g = e->slots[i]->sig;
    if (g.valid && g.state == sims[i].st->id) {
      std::memcpy(e->h_sc[i].corners_hold, g.corners_hold, sizeof g.corners_hold);
      e->h_sc[i].ago = g.ago; e->h_sc[i].maxj_seen = g.maxj_seen;
      e->h_sc[i].nentries = g.nentries; e->h_sc[i].nentries_ref = g.nentries_ref; e->h_sc[i].nrowent = g.nrowent;
    }
  }
  HIPCHK(hipMemcpyAsync(e->d_sc.p, e->h_sc.data(), (size_t)ns * sizeof(SimScalars), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return SCEMA_MD_OK;
}

// scalars that must not leak from one run into the next when h_sc is re-uploaded
int reupload_scalars(scema_md_engine *e, int ns) {
  for (int i = 0; i < ns; i++) {
    e->h_sc[i].overflow = 0;
    e->h_sc[i].nbuilds = 0;
    e->h_sc[i].maxneigh_seen = 0;
  }
  HIPCHK(hipMemcpyAsync(e->d_sc.p, e->h_sc.data(), (size_t)ns * sizeof(SimScalars), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return SCEMA_MD_OK;
}

}  // namespace scema_eng
