// engine_reax.cpp -- host side of the ReaxFF path (force_field "reax"): run_phase_reax and the ReaxFF entry points of the C ABI
#include "engine.h"
#include "../md_env.h"

namespace scema_eng {

// run_phase_reax is run_phase with another force stage: the same step sequence (k_pre, k_initial_integrate, forces,
// k_final_integrate, k_post, k_remap), the same batch rules (longest run first, active prefix), the same box flips; no
// cells, no Ewald tables, no SHAKE (lammps_scripts_reax/in.strain.lammps has no fix shake and no kspace_style).

// (the pointers of RxView are qualified as global-memory pointers in device code, reax/rx_types.h: a cast in both passes of the compiler)
#define RXSET(dst, src) dst = (decltype(dst))(src)
static int ensure_rx_slot(scema_md_engine *e, RxSlot &r, int n, int npad, int maxnb, int maxbd, int maxnbn, bool col16) {
  if (npad > r.cap_pad) {
    HIPCHK(r.nb_cnt.ensure((size_t)npad * 4));
    HIPCHK(r.bd_cnt.ensure((size_t)npad * 4));
    HIPCHK(r.deltap.ensure((size_t)npad * 8));
    HIPCHK(r.total_bo.ensure((size_t)npad * 8));
    HIPCHK(r.cd_delta.ensure((size_t)npad * 8));
    HIPCHK(r.hd.ensure((size_t)npad * 8));
    HIPCHK(r.q.ensure((size_t)npad * 8));
    HIPCHK(r.s.ensure((size_t)npad * 8));
    HIPCHK(r.t.ensure((size_t)npad * 8));
    HIPCHK(r.s_hist.ensure(4 * (size_t)npad * 8));
    HIPCHK(r.t_hist.ensure(3 * (size_t)npad * 8));
    HIPCHK(r.qwork.ensure(10 * (size_t)npad * 8));
    HIPCHK(r.qpart.ensure((10 * (size_t)((npad + 255) / 256) + 2 * (size_t)(npad / RX_SWR + 1)) * 8));   // layout: md_reax.hip
    HIPCHK(r.pm_len.ensure((size_t)npad * 4));
    HIPCHK(r.pm_col.ensure((size_t)RX_PM_MAX * npad * 4));
    HIPCHK(r.pm_raw.ensure((size_t)RX_PM_MAX * npad * 8));
    HIPCHK(r.pm_val.ensure((size_t)RX_PM_MAX * npad * 8));
    HIPCHK(r.nbn_cnt.ensure((size_t)npad * 4));
    HIPCHK(r.hlen.ensure((size_t)npad * 4));
    HIPCHK(r.hownlen.ensure((size_t)npad * 4));
    HIPCHK(r.nb_own0.ensure((size_t)npad * 4));
    HIPCHK(r.misc.ensure(256));
    r.cap_pad = npad;
    r.cap_nb = 0;
    r.cap_bd = 0;
    r.cap_nbn = 0;
  }
  if (maxnbn > r.cap_nbn) {
    HIPCHK(r.nbn.ensure((size_t)maxnbn * npad * 4));
    HIPCHK(r.nbnT.ensure((size_t)maxnbn * npad * 4));
    r.cap_nbn = maxnbn;
  }
  if (maxnb > r.cap_nb) {
    HIPCHK(r.hval.ensure((size_t)maxnb * npad * 8));
    HIPCHK(r.nbT.ensure((size_t)maxnb * npad * 4));
    HIPCHK(r.hown.ensure((size_t)maxnb * npad * 4));
    r.cap_nb = maxnb;
  }
  if (!col16 && (size_t)r.cap_nb * npad > r.cap_col) {   // (32-bit columns beside the values: replicas of more than 65 536 atoms only)
    HIPCHK(r.hcol.ensure((size_t)r.cap_nb * npad * 4));
    r.cap_col = (size_t)r.cap_nb * npad;
  }
  if (maxbd > r.cap_bd) {
    const size_t plane = (size_t)maxbd * npad;
    HIPCHK(r.bd.ensure(plane * 4));
    HIPCHK(r.bd_rev.ensure(plane * 4));
    HIPCHK(r.bd_bop.ensure(4 * plane * 8));
    HIPCHK(r.bd_c.ensure(3 * plane * 8));
    HIPCHK(r.bd_bo.ensure(3 * plane * 8));
    HIPCHK(r.bd_g.ensure(3 * plane * 8));
    HIPCHK(r.bd_cb.ensure(plane * 8));
    r.cap_bd = maxbd;
  }
  (void)n;
  return SCEMA_MD_OK;
}

// force-field type of every atom of a replica: element of its LAMMPS type as pair_coeff names them
static int ensure_rtype(scema_md_engine *e, Topo &T) {
  if (T.rtype_stamp == e->rx_stamp && T.d_rtype.p) return SCEMA_MD_OK;
  std::vector<int> rt(T.natoms);
  for (int i = 0; i < T.natoms; i++) {
    const int ty = T.original.type[i];   // the registered LAMMPS type (Topo::type holds Lennard-Jones classes)
    if (ty < 0 || ty >= (int)e->rx_type_map.size())
      return fail(e, SCEMA_MD_ERR_ARG, "atom %d has type %d but the ReaxFF element list names %zu types (pair_coeff * * ffield ...)", i, ty + 1, e->rx_type_map.size());
    rt[i] = e->rx_type_map[ty];
  }
  int rc = upload(e, T.d_rtype, rt);
  if (rc) return rc;
  T.rtype_stamp = e->rx_stamp;
  return SCEMA_MD_OK;
}

int run_phase_reax(scema_md_engine *e, std::vector<ActiveSim> &sims, const RunSpec &spec) {
  const int ns = (int)sims.size();
  const scema_md_params &P = e->p;
  if (!e->rx_ready) return fail(e, SCEMA_MD_ERR_ARG, "force field 'reax' asked for but no ReaxFF force-field file is loaded (scema_md_reax_configure)");
  const double rlist = e->rx_host.swb + e->rx_skin;
  std::vector<int> order(ns);
  for (int i = 0; i < ns; i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return sims[a].nsteps > sims[b].nsteps; });
  // The batch runs as part batches on as many streams (the step loop below).  Part p takes the ranks p, p + P, ... of the length order and sits at
  // consecutive positions: each part is itself sorted longest first, and the parts carry the same mix of run lengths.
  // (From six replicas per part on: 8 / 10 replicas whole 494 / 595 evaluations/s, as two parts 460 / 564; 12 / 16 / 18 replicas 617 / 776 / 847
  // whole, 661 / 852 / 910 as two.  Three and four parts lose at every size -- each part has a side stream too, and a process has four hardware
  // queues: 36 replicas 1 171 as two parts, 902 as three.  profiles/r06_x_reax_parts_ab.log)
  const int nparts_plan = (e->rx_halves >= 2 && ns >= 6 * e->rx_halves && !spec.minimize) ? e->rx_halves : 1;
  if (nparts_plan > 1) {
    std::vector<int> o2;
    o2.reserve(ns);
    for (int p = 0; p < nparts_plan; p++)
      for (int r = p; r < ns; r += nparts_plan) o2.push_back(order[r]);
    order.swap(o2);
  }
  e->h_sims.assign(ns, SimDev());
  e->h_rxviews.assign(ns, RxView());
  std::vector<std::vector<FlipEvent>> flips(ns);
  int maxatoms = 0, maxpad = 0, maxsteps = 0;
  // columns of the charge-equilibration matrix as 16-bit atom indices when every replica of the batch has at most 65 536 atoms
  bool col16 = !(scema_env("SCEMA_MD_RX_COL32") && atoi(scema_env("SCEMA_MD_RX_COL32")) != 0);   // (test switch: 32-bit columns for any size)
  for (int i = 0; i < ns; i++) col16 = col16 && sims[i].st->topo->natoms <= 65536;
  e->h_zerotab.clear();
  bool any_precond = false, any_validate = false, inject_precond_failure = false;
  bool all_sym = col16 && e->rx_sym;
  for (int pos = 0; pos < ns; pos++) {
    const int i = order[pos];
    ActiveSim &A = sims[i];
    Topo &T = *A.st->topo;
    int rc = ensure_rtype(e, T);
    if (rc) return rc;
    const SimScalars &hsc = e->h_sc[i];
    double box_end[9];
    std::memcpy(box_end, hsc.box, sizeof box_end);
    std::vector<HostBox> boxes(2);
    if (spec.deform) {
      std::vector<HostBox> extremes;
      if (!deform_trajectory(hsc.box, A.rates, A.dt, A.nsteps, box_end, flips[pos], extremes))
        return fail(e, SCEMA_MD_ERR_BOX, "fix deform is changing yz too much with xy: the strain would tilt yz past half the box");
      boxes.insert(boxes.end(), extremes.begin(), extremes.end());
    }
    box_derive(hsc.box, boxes[0]);
    box_derive(box_end, boxes[1]);
    if (spec.nh && spec.npt && spec.box_margin > 0.0)
      for (int sgn = -1; sgn <= 1; sgn += 2) {   // the barostat dilates the box: both ends of the range a segment is laid out for
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
    double w0[3] = {1e300, 1e300, 1e300}, w1[3], vol_min = 1e300;
    for (const HostBox &hb : boxes) {
      perp_widths(hb, w1);
      for (int d = 0; d < 3; d++) w0[d] = std::min(w0[d], w1[d]);
      vol_min = std::min(vol_min, hb.vol);
    }
    SimDev S;
    std::memset(&S, 0, sizeof S);
    RxView V;
    std::memset(&V, 0, sizeof V);
    bool small = false;
    for (int d = 0; d < 3; d++) small = small || w0[d] < 2.0 * rlist;
    for (int d = 0; d < 3; d++) {
      V.mimg[d] = small ? (int)std::ceil(rlist / w0[d]) : 0;
      if (V.mimg[d] > 2) return fail(e, SCEMA_MD_ERR_BOX, "box width %.3f < (cutoff+skin)/2 = %.3f in dim %d", w0[d], 0.5 * rlist, d);
    }
    const int n = T.natoms, npad = (n + 63) / 64 * 64;
    const double rho = n / vol_min;
    int maxnb = (int)std::ceil(rho * 4.0 / 3.0 * MD_PI * rlist * rlist * rlist * 1.2 * e->neigh_grow) + 32;
    maxnb = (maxnb + 7) / 8 * 8;
    const int maxbd = (int)std::ceil(24 * e->neigh_grow);
    double rnear = 0.0;   // (the widest near row of the force field sizes the rows)
    for (int k = 0; k < RX_MAXT * RX_MAXT; k++) rnear = std::max(rnear, std::sqrt(e->rx_host.rnear2[k]));
    int maxnbn = (int)std::ceil(rho * 4.0 / 3.0 * MD_PI * rnear * rnear * rnear * 1.5 * e->neigh_grow) + 32;
    maxnbn = (maxnbn + 7) / 8 * 8;
    Slot &sl = *e->slots[i];
    // A run that follows another ReaxFF run of the same state on the same slot (the sampling run behind the straining run of an evaluation; the
    // straining run of the next update) keeps that run's neighbour rows, as the OPLS path does (run_phase): rows, near rows, reference positions
    // and the preconditioner live in the slot, the list's scalars come back through the slot's signature (prepare_slots), and k_phase_init /
    // k_keep_validate decide on the device whether they still hold.  The rows keep the strides they were built with.
    static const bool keep_lists = !(scema_env("SCEMA_MD_KEEP_LIST") && atoi(scema_env("SCEMA_MD_KEEP_LIST")) == 0);
    bool keep = false;
    {
      const ListSig &g = sl.sig;
      const SimScalars &hsc = e->h_sc[i];
      if (spec.keep_list && keep_lists && g.valid && g.rx_stamp != 0 && g.rx_stamp == e->rx_stamp && g.topo == T.id && g.rlist == rlist && g.npad == npad &&
          (spec.keep_list == 1 || g.state == A.st->id) && !hsc.force_rebuild && !hsc.overflow && maxnb <= g.maxneigh && maxnbn <= g.capj &&
          g.rx_mimg[0] == V.mimg[0] && g.rx_mimg[1] == V.mimg[1] && g.rx_mimg[2] == V.mimg[2] && sl.rx) {
        keep = true;
        maxnb = g.maxneigh; maxnbn = g.capj;
      }
    }
    S.keep_list = keep ? spec.keep_list : 0;
    any_validate = any_validate || S.keep_list == 2;
    {
      ListSig &g = sl.sig;   // what this run's rows are built for; valid once the run has ended without a fault
      g.valid = false;
      g.rx_stamp = e->rx_stamp;
      g.topo = T.id;
      g.rlist = rlist; g.npad = npad; g.maxneigh = maxnb; g.capj = maxnbn;
      for (int d = 0; d < 3; d++) g.rx_mimg[d] = V.mimg[d];
    }
    rc = ensure_slot(e, sl, n, 64, 1, 0, 64);
    if (rc) return rc;
    if (!sl.rx) sl.rx.reset(new RxSlot());
    RxSlot &R = *sl.rx;
    if ((rc = ensure_rx_slot(e, R, n, npad, maxnb, maxbd, maxnbn, col16))) return rc;
    S.natoms = n; S.npad = npad; S.ntypes = T.ntypes;
    S.nsteps = A.nsteps;
    if (spec.sample) {
      S.nav = (A.nsteps > 10000) ? A.nsteps / 1000 : A.nsteps / 10;   // in.homogenization.lammps:57 (the reax copy is the same)
      if (S.nav < 1) S.nav = 1;
      S.nwin = A.nsteps / S.nav;
    }
    S.nvt = spec.nvt; S.use_shake = 0; S.deform = spec.deform;
    if (spec.nh) {
      S.ramp = 1; S.npt = spec.npt; S.nh_total = std::max(spec.nh_total, 1); S.lavg_nav = spec.lavg_nav;
      S.t_start = spec.t_start; S.t_stop = spec.t_stop; S.p_target = spec.p_target; S.p_freq = 1.0 / spec.p_period; S.box_margin = spec.box_margin;
    }
    if (spec.minimize) {
      S.min_etol = spec.min_etol; S.min_ftol = spec.min_ftol; S.min_dmax = 0.1; S.min_maxiter = spec.min_maxiter; S.min_maxeval = spec.min_maxeval;
      S.min_incremental = 1;   // the neighbour rebuild wraps the atoms into the box: trial points by increments
    }
    S.t_chain = std::min(P.t_chain, MD_MAXCHAIN);
    S.neigh_delay = 0;   // neigh_modify every 1 delay 0 (in.set.lammps:32 of the reax scripts); rebuilt when needed, same pairs inside the cutoff
    S.dt = A.dt; S.t_target = A.temperature; S.t_freq = 1.0 / P.t_period;
    S.tdof = 3.0 * n - 3.0;
    S.skin = e->rx_skin;
    S.far_band = e->rx_skin;
    for (int k = 0; k < 6; k++) S.rates[k] = A.rates[k];
    S.type = T.d_type.as<int>(); S.q = T.d_q.as<double>(); S.mass = T.d_mass.as<double>();
    S.x = A.st->x.as<double>(); S.v = A.st->v.as<double>(); S.f = sl.f.as<double>();
    S.wrapn = sl.wrapn.as<int>(); S.xhold = sl.xhold.as<double>();
    S.sfac = sl.sfac.as<double>(); S.cell_count = sl.cell_count.as<int>();
    S.sc = e->d_sc.as<SimScalars>() + i;
    V.n = n; V.npad = npad; V.maxnb = maxnb; V.maxbd = maxbd;
    RXSET(V.rtype, T.d_rtype.as<int>()); RXSET(V.x, S.x); RXSET(V.q, R.q.as<double>());
    RXSET(V.nbn_cnt, R.nbn_cnt.as<int>()); RXSET(V.nbn, R.nbn.as<int>()); RXSET(V.nbnT, R.nbnT.as<int>()); V.maxnbn = maxnbn; V.rnear2 = rnear * rnear;
    RXSET(V.qpart, R.qpart.as<double>());
    RXSET(V.nb_cnt, R.nb_cnt.as<int>()); RXSET(V.nb, (int *)nullptr); RXSET(V.bd_cnt, R.bd_cnt.as<int>()); RXSET(V.bd, R.bd.as<int>()); RXSET(V.bd_rev, R.bd_rev.as<int>());
    RXSET(V.bd_bop, R.bd_bop.as<double>()); RXSET(V.bd_c, R.bd_c.as<double>()); RXSET(V.bd_bo, R.bd_bo.as<double>()); RXSET(V.bd_g, R.bd_g.as<double>()); RXSET(V.bd_cb, R.bd_cb.as<double>());
    RXSET(V.deltap, R.deltap.as<double>()); RXSET(V.total_bo, R.total_bo.as<double>()); RXSET(V.cd_delta, R.cd_delta.as<double>()); RXSET(V.hd, R.hd.as<double>());
    RXSET(V.f, S.f); RXSET(V.hval, col16 ? nullptr : R.hval.as<double>()); RXSET(V.hpk, col16 ? R.hval.as<unsigned long long>() : nullptr); RXSET(V.s, R.s.as<double>()); RXSET(V.t, R.t.as<double>());
    V.warm = (spec.qeq_continue || A.st->qhist_valid) ? 1 : 0;
    RXSET(V.hcol16, (unsigned short *)nullptr); RXSET(V.hcol32, col16 ? nullptr : R.hcol.as<int>()); RXSET(V.hlen, R.hlen.as<int>()); RXSET(V.nbT, R.nbT.as<int>());
    RXSET(V.hown, R.hown.as<int>()); RXSET(V.hownlen, R.hownlen.as<int>()); RXSET(V.nb_own0, R.nb_own0.as<int>());
    RXSET(V.s_hist, R.s_hist.as<double>()); RXSET(V.t_hist, R.t_hist.as<double>()); RXSET(V.qwork, R.qwork.as<double>());
    // the bonded-pattern preconditioner needs one image per neighbour (boxes at least two list radii wide: every production replica)
    V.pm_on = (e->rx_precond && V.mimg[0] == 0 && V.mimg[1] == 0 && V.mimg[2] == 0) ? 1 : 0;
    if (V.pm_on && scema_env("SCEMA_MD_TEST_QEQ_PRECOND_FAILS")) inject_precond_failure = true;   // test hook: this run reports a solve that did not converge
    any_precond = any_precond || V.pm_on;
    // the symmetric form of the solve: rows sorted by partner (one image per neighbour), both vectors of the replica in a workgroup's LDS
    all_sym = all_sym && V.mimg[0] == 0 && V.mimg[1] == 0 && V.mimg[2] == 0 && 2 * (size_t)npad * 16 + 4096 <= 128 * 1024;
    RXSET(V.pm_len, R.pm_len.as<int>()); RXSET(V.pm_col, R.pm_col.as<int>()); RXSET(V.pm_raw, R.pm_raw.as<double>()); RXSET(V.pm_val, R.pm_val.as<double>());
    RXSET(V.eparts, R.misc.as<double>());                         // [0, 13) doubles
    RXSET(V.qstat, (int *)(R.misc.as<char>() + 128));             // 6 ints
    RXSET(V.overflow, (int *)(R.misc.as<char>() + 160));
    RXSET(V.sweep_acc, (long long *)(R.misc.as<char>() + 168));   // 2 x 8 bytes
    e->h_zerotab.push_back(MdkZero{sl.wrapn.as<int>(), 3 * (long long)n});
    e->h_sims[pos] = S;
    e->h_rxviews[pos] = V;
    maxatoms = std::max(maxatoms, n); maxpad = std::max(maxpad, npad); maxsteps = std::max(maxsteps, A.nsteps);
  }
  HIPCHK(e->d_sims.ensure((size_t)ns * sizeof(SimDev)));
  HIPCHK(e->d_rxviews.ensure((size_t)ns * sizeof(RxView)));
  HIPCHK(hipMemcpyAsync(e->d_sims.p, e->h_sims.data(), (size_t)ns * sizeof(SimDev), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(e->d_rxviews.p, e->h_rxviews.data(), (size_t)ns * sizeof(RxView), hipMemcpyHostToDevice, e->stream));
  if (!e->h_zerotab.empty()) {   // the wrap counters of every replica start from zero: one launch
    HIPCHK(e->d_zerotab.ensure(e->h_zerotab.size() * sizeof(MdkZero)));
    HIPCHK(hipMemcpyAsync(e->d_zerotab.p, e->h_zerotab.data(), e->h_zerotab.size() * sizeof(MdkZero), hipMemcpyHostToDevice, e->stream));
    mdk_zero_many(e->stream, e->d_zerotab.as<MdkZero>(), (int)e->h_zerotab.size(), 3 * (long long)maxatoms);
  }
  const SimDev *D = e->d_sims.as<SimDev>();
  RxView *VV = e->d_rxviews.as<RxView>();
  const RxParams *RP = e->d_rxparams.as<RxParams>();
  hipStream_t st = e->stream;
  const int terms = e->rx_terms;
  // ---- setup (step 0) ----
  const bool prof = e->p.profile != 0 && !spec.minimize;
  size_t ev_used = 0;
  std::vector<hipEvent_t> *evp = prof ? &e->ev_pool : nullptr;
  mdk_phase_init(st, D, ns);
  if (any_validate) mdk_keep_validate(st, D, ns, maxatoms);
  // Charge-equilibration history: kept in place when this run follows another one on the same slots; else a state that has run
  // before brings its own (one copy launch for the batch); the rest start from zeros like a new fix qeq/reax
  bool any_cold = false;
  {
    std::vector<MdkCopy> tab;
    long long maxn = 0;
    for (int pos = 0; pos < ns; pos++) {
      ActiveSim &A = sims[order[pos]];
      const RxView &V = e->h_rxviews[pos];
      if (!V.warm) any_cold = true;
      if (spec.qeq_continue || !A.st->qhist_valid) continue;
      const long long np = V.npad;
      tab.push_back(MdkCopy{A.st->qhist.as<double>(), V.s_hist, 4 * np});
      tab.push_back(MdkCopy{A.st->qhist.as<double>() + 4 * np, V.t_hist, 3 * np});
      maxn = std::max(maxn, 4 * np);
    }
    if (!tab.empty()) {
      HIPCHK(e->d_copytab.ensure(tab.size() * sizeof(MdkCopy)));
      HIPCHK(hipMemcpyAsync(e->d_copytab.p, tab.data(), tab.size() * sizeof(MdkCopy), hipMemcpyHostToDevice, st));
      HIPCHK(hipStreamSynchronize(st));   // (the table is a local)
      mdk_copy_many(st, e->d_copytab.as<MdkCopy>(), (int)tab.size(), maxn);
    }
  }
  // how a solve is issued (md_reax.h): as many iterations as the slowest solve of the last run took plus a margin; the first
  // solves of a run that has replicas without a history take longer
  // the bond-order chain of the force stage on the engine's side stream, next to the charge chain (md_reax.hip); SCEMA_REAX_OVERLAP=0: one stream
  const RxSide side = {e->stream2, e->ev_fork, e->ev_up, e->ev_join};
  const RxSide *sidep = (e->stream2 && e->ev_up && e->rx_overlap) ? &side : nullptr;
  auto plan_for = [&](int step) {
    RxQeqPlan pl;
    pl.launch = (step < 4 && any_cold) ? e->rx_qeq_launch_cold : e->rx_qeq_launch;
    pl.setup = step == 0 ? 1 : 0;
    pl.precond = any_precond ? 1 : 0;
    pl.sym = all_sym ? 1 : 0;
    return pl;
  };
  mdk_reax_phase_init(st, VV, ns, maxpad);
  mdk_reax_forces(st, D, VV, RP, ns, maxatoms, rlist, e->rx_qeq_tol, e->rx_qeq_maxiter, plan_for(0), terms, col16, evp, &ev_used, sidep);
  mdk_final_integrate(st, D, ns, maxatoms, 0);
  if (spec.nh) mdk_setup_post_nh(st, D, ns);
  else mdk_setup_post(st, D, ns);
  if (spec.minimize) {
    // min_style sd (md_equil.hip): the line search of every replica on the device, forces from the ReaxFF stage
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
        mdk_reax_forces(st, D, VV, RP, ns, maxatoms, rlist, e->rx_qeq_tol, e->rx_qeq_maxiter, plan_for(1), terms, col16, nullptr, nullptr, sidep);
        mdk_min_reduce(st, D, ns, maxatoms, hsd);
        mdk_min_decide(st, D, ns);
      }
      HIPCHK(hipMemcpyAsync(e->h_sc.data(), e->d_sc.p, (size_t)ns * sizeof(SimScalars), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      all_done = true;
      for (int i = 0; i < ns; i++)
        if (!e->h_sc[i].overflow && e->h_sc[i].min_phase != 4) all_done = false;
    }
    HIPCHK(hipGetLastError());
    int fault_m = 0;
    for (int i = 0; i < ns; i++) fault_m |= e->h_sc[i].overflow;
    if (fault_m & 16) return fail(e, SCEMA_MD_ERR_ARG, "a simulation became unstable during the minimisation (non-finite positions)");
    if (fault_m & 32) {
      e->rx_qeq_failed = true;   // (eval_chunk retries once with the reference's Jacobi preconditioner if the approximate inverse was on)
      return fail(e, SCEMA_MD_ERR_ARG, "charge equilibration did not converge to %.1e in %d iterations", e->rx_qeq_tol, e->rx_qeq_maxiter);
    }
    e->overflow_bits = (fault_m & 1) ? 8 : 0;
    if (fault_m & 1) return SCEMA_MD_ERR_OVERFLOW;
    if (!all_done) return fail(e, SCEMA_MD_ERR_ARG, "minimiser did not stop within its evaluation budget");
    return SCEMA_MD_OK;
  }
  // ---- steps ----
  std::map<int, std::vector<std::pair<int, int>>> flip_at;
  for (int pos = 0; pos < ns; pos++)
    for (size_t k = 0; k < flips[pos].size(); k++)
      if (flips[pos][k].step < e->h_sims[pos].nsteps) flip_at[flips[pos][k].step].push_back({pos, (int)k});
  // Two half batches on two streams (replicas are independent: each half runs its own sequence of steps, and one half's launch gaps, tails
  // and latency-bound kernels are filled by the other's work); each with its own side stream for the bond-order chain.  SCEMA_REAX_HALVES=0: one.
  struct Half { int off, n; hipStream_t st; RxSide side; const RxSide *sidep; hipEvent_t done; };
  std::vector<Half> halves;
  int nparts = nparts_plan;
  // (streams and events of the parts beyond the first: created once, kept)
  // (the second part runs on stream3 and the engine's fourth stream; further parts -- a measurement aid, two is the optimum -- get streams of their own)
  const int pool0 = (e->stream3 && e->rx_side1) ? 2 : 1;   // parts served without the pool
  while (nparts > pool0 && (int)e->rx_parts.size() < nparts - pool0) {
    scema_md_engine::RxPart pt;
    bool ok = hipStreamCreateWithFlags(&pt.main, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&pt.side, hipStreamNonBlocking) == hipSuccess;
    for (int k = 0; k < 4 && ok; k++) ok = hipEventCreateWithFlags(&pt.ev[k], hipEventDisableTiming) == hipSuccess;
    if (!ok) return fail(e, SCEMA_MD_ERR_DEVICE, "could not create the streams of the ReaxFF part batches");
    e->rx_parts.push_back(pt);
  }
  const bool two = nparts > 1;
  halves.reserve(nparts);
  for (int k = 0, off = 0; k < nparts; k++) {
    const int nk = (ns - off) / (nparts - k) + (((ns - off) % (nparts - k)) ? 1 : 0);
    if (k == 0) halves.push_back(Half{0, nk, st, side, nullptr, nullptr});
    else if (k == 1 && pool0 == 2) halves.push_back(Half{off, nk, e->stream3, RxSide{e->rx_side1, e->rx_side1_ev[0], e->rx_side1_ev[1], e->rx_side1_ev[2]}, nullptr, e->rx_side1_ev[3]});
    else {
      const auto &pt = e->rx_parts[k - pool0];
      halves.push_back(Half{off, nk, pt.main, RxSide{pt.side, pt.ev[0], pt.ev[1], pt.ev[2]}, nullptr, pt.ev[3]});
    }
    off += nk;
  }
  for (auto &H : halves) H.sidep = sidep ? &H.side : nullptr;
  if (two) {
    if (!e->rx_fork) HIPCHK(hipEventCreateWithFlags(&e->rx_fork, hipEventDisableTiming));
    HIPCHK(hipEventRecord(e->rx_fork, st));
    for (size_t k = 1; k < halves.size(); k++) HIPCHK(hipStreamWaitEvent(halves[k].st, e->rx_fork, 0));
  }
  for (int step = 1; step <= maxsteps; step++) {
    bool any = false;
    for (const Half &H : halves) {
      int na = 0;   // (a part is sorted longest first: its active replicas are a prefix)
      while (na < H.n && e->h_sims[H.off + na].nsteps >= step) na++;
      if (na == 0) continue;
      any = true;
      const SimDev *Dh = D + H.off;
      RxView *Vh = VV + H.off;
      hipStream_t sh = H.st;
      if (spec.nh) { mdk_pre_nh(sh, Dh, na); mdk_initial_integrate_nh(sh, Dh, na, maxatoms); }
      else { mdk_pre(sh, Dh, na); mdk_initial_integrate(sh, Dh, na, maxatoms); }
      // the first solves of a run start from an empty history (RX_QEQ_COLD in md_reax.hip: setup is solve 1)
      mdk_reax_forces(sh, Dh, Vh, RP, na, maxatoms, rlist, e->rx_qeq_tol, e->rx_qeq_maxiter, plan_for(step), terms, col16, evp, &ev_used, H.sidep);
      mdk_final_integrate(sh, Dh, na, maxatoms, 1);
      if (spec.nh) mdk_post_nh(sh, Dh, na);
      else mdk_post(sh, Dh, na);
      if (spec.deform) mdk_remap(sh, Dh, na, maxatoms);
      e->prof.md_steps += na;
    }
    if (!any) break;
    auto fl = flip_at.find(step);
    if (fl != flip_at.end())
      for (const auto &pk : fl->second) {
        const FlipEvent &fe = flips[pk.first][pk.second];
        hipStream_t sh = st;
        for (const Half &H : halves) if (pk.first >= H.off && pk.first < H.off + H.n) sh = H.st;
        mdk_flip(sh, D + pk.first, fe.tilt[0], fe.tilt[1], fe.tilt[2]);
        e->prof.box_flips += 1;
      }
  }
  for (size_t k = 1; k < halves.size(); k++) {
    HIPCHK(hipEventRecord(halves[k].done, halves[k].st));
    HIPCHK(hipStreamWaitEvent(st, halves[k].done, 0));
  }
  mdk_phase_end(st, D, ns, maxatoms);
  {  // the states keep the history for their next run (a failed update drops it: backup_states)
    std::vector<MdkCopy> tab;
    long long maxn = 0;
    for (int pos = 0; pos < ns; pos++) {
      ActiveSim &A = sims[order[pos]];
      const RxView &V = e->h_rxviews[pos];
      const long long np = V.npad;
      HIPCHK(A.st->qhist.ensure(7 * (size_t)np * 8));
      tab.push_back(MdkCopy{V.s_hist, A.st->qhist.as<double>(), 4 * np});
      tab.push_back(MdkCopy{V.t_hist, A.st->qhist.as<double>() + 4 * np, 3 * np});
      maxn = std::max(maxn, 4 * np);
      A.st->qhist_valid = true;
    }
    HIPCHK(e->d_copytab.ensure(tab.size() * sizeof(MdkCopy)));
    HIPCHK(hipMemcpyAsync(e->d_copytab.p, tab.data(), tab.size() * sizeof(MdkCopy), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    mdk_copy_many(st, e->d_copytab.as<MdkCopy>(), (int)tab.size(), maxn);
  }
  HIPCHK(hipMemcpyAsync(e->h_sc.data(), e->d_sc.p, (size_t)ns * sizeof(SimScalars), hipMemcpyDeviceToHost, st));
  // (the solver statistics of all replicas in ONE read-back: two small copies per replica were 2 x 72 launch gaps of 24 us per run)
  std::vector<int> qs(6 * (size_t)ns, 0);
  std::vector<long long> acc(2 * (size_t)ns, 0);
  HIPCHK(e->d_rxstat.ensure(8 * (size_t)ns * sizeof(long long)));
  e->h_rxstat.assign(8 * (size_t)ns, 0);
  mdk_reax_collect_stats(st, e->d_rxviews.as<RxView>(), ns, e->d_rxstat.as<long long>());
  HIPCHK(hipMemcpyAsync(e->h_rxstat.data(), e->d_rxstat.p, 8 * (size_t)ns * sizeof(long long), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  for (int pos = 0; pos < ns; pos++) {
    for (int k = 0; k < 6; k++) qs[6 * pos + k] = (int)e->h_rxstat[8 * pos + k];
    acc[2 * pos] = e->h_rxstat[8 * pos + 6]; acc[2 * pos + 1] = e->h_rxstat[8 * pos + 7];
  }
  HIPCHK(hipGetLastError());
  if (prof) {
    // the matrix sweep of the charge equilibration, the HBM-bound kernel of this path: HIP-event time of every launch, and what
    // the launches read by the algorithm: 8 bytes per stored matrix entry (value and 16-bit column in one word, RxView::hpk; 8 + 4 with 32-bit columns) and RX_SWEEP_ROW_BYTES per row,
    // for every replica and sweep it took part in (counted on the device, k_rx_qeq_finish)
    for (size_t l = 0; 2 * l + 1 < ev_used; l++) {
      float ms = 0.f;
      HIPCHK(hipEventElapsedTime(&ms, e->ev_pool[2 * l], e->ev_pool[2 * l + 1]));
      e->prof.rx_sweep_ms += ms;
      e->prof.rx_sweep_launches += 1;
    }
    e->prof.rx_sweep_union_ms += event_union_ms(e->ev_pool, ev_used / 2);
    // (the launches of both half batches are timed, each on its stream; rx_sweep_union_ms is the time with at least one of them in flight)
    for (int pos = 0; pos < ns; pos++) {
      e->prof.rx_sweep_entries += (double)acc[2 * pos];
      e->prof.rx_sweep_rows += (double)acc[2 * pos + 1];
      e->prof.rx_sweep_col_bytes = col16 ? 0 : 4;   // (packed entries: the column rides in the value's word)
      e->prof.rx_sweep_symmetric = all_sym ? 1 : 0;
    }
  }
  int fault = 0, most = 0, most_cold = 0;
  for (int i = 0; i < ns; i++) {
    fault |= e->h_sc[i].overflow;
    e->prof.neigh_builds += e->h_sc[i].nbuilds;
    e->rx_qeq_iters += qs[6 * i];
    e->rx_qeq_solves += qs[6 * i + 1] - (e->h_rxviews[i].warm ? RX_QEQ_COLD_SOLVES : 0);   // (a warm run starts its solve count past the cold ones)
    most = std::max(most, qs[6 * i + 2]);
    most_cold = std::max(most_cold, qs[6 * i + 5]);
    e->rx_qeq_slow += qs[6 * i + 3];
  }
  // iterations issued as launches in the next run: what the slowest solve of this one needed (of all replicas and steps), plus one.  A launch
  // that finds every replica converged still costs its two kernels and their gaps (28 us); a replica that needs more than was issued
  // finishes in one workgroup (80 us per iteration).  Scan on the 72-replica set, slowest solve 15: 13 launches 543, 14: 591, 15: 603,
  // 16: 598, 18: 590 evaluations/s (tools/reax_launch_scan.sh, profiles/r04_zh_launch_scan.txt)
  if (!e->rx_qeq_launch_pinned) {
    // (the floor of 8 dated from the Jacobi preconditioner's 11 iterations per solve; with 3.9 per solve -- slowest 5 -- it issued 8: scan
    // with the round-5 preconditioner, 72 replicas: 3 launches 577, 4: 747, 5: 978, 6: 969, 7: 965, 8: 959 evaluations/s)
    if (most > 0) e->rx_qeq_launch = std::max(4, most + 1);
    if (most_cold > 0) e->rx_qeq_launch_cold = std::max(4, most_cold + 1);
  }
  if (fault & 16) return fail(e, SCEMA_MD_ERR_ARG, "a simulation became unstable (non-finite or runaway atom positions): overlapping atoms or a time step too long for ReaxFF");
  if ((fault & 32) || inject_precond_failure) {
      e->rx_qeq_failed = true;   // (eval_chunk retries once with the reference's Jacobi preconditioner if the approximate inverse was on)
      return fail(e, SCEMA_MD_ERR_ARG, "charge equilibration did not converge to %.1e in %d iterations", e->rx_qeq_tol, e->rx_qeq_maxiter);
    }
  e->overflow_bits = ((fault & 1) ? (1 | 8) : 0) | (fault & 64);
  if (fault & 1) return SCEMA_MD_ERR_OVERFLOW;
  if (fault & 64) return SCEMA_MD_ERR_OVERFLOW;   // the barostat took the box out of the range this segment was laid out for
  for (int i = 0; i < ns; i++) {   // the rows on the device hold for the positions this run ended at
    ListSig &g = e->slots[i]->sig;
    const SimScalars &c = e->h_sc[i];
    g.valid = !spec.minimize;
    g.state = sims[i].st->id;
    std::memcpy(g.corners_hold, c.corners_hold, sizeof g.corners_hold);
    g.ago = c.ago;
  }
  return SCEMA_MD_OK;
}

}  // namespace scema_eng

extern "C" {

// ---- ReaxFF path ----
// The reach of the uncorrected bond order of a type pair (rx_bond_prime_pair, reax/rx_core.h: three terms exp(p_a (r / r_x)^p_b) with p_a < 0 < p_b, each
// falling with r): the r at which it passes bo_cut, plus a margin that covers the difference between this libm evaluation and the kernels' own
// exp / log by orders of magnitude.  The exact test stays in the kernel; this only keeps hopeless candidates out of the near rows
// (a row of 105 candidates within 5 + 1 A of a polyethylene atom holds 30 within reach + 1 A).  Parameters that do not fall with r: no cut.
static double rx_bond_reach(const RxParams &P, int ti, int tj) {
  const RxSbp &si = P.sbp[ti], &sj = P.sbp[tj];
  const RxTbp &t = P.tbp[ti * RX_MAXT + tj];
  const bool on_s = si.r_s > 0.0 && sj.r_s > 0.0, on_p = si.r_pi > 0.0 && sj.r_pi > 0.0, on_pp = si.r_pi_pi > 0.0 && sj.r_pi_pi > 0.0;
  if ((on_s && !(t.p_bo1 < 0.0 && t.p_bo2 > 0.0)) || (on_p && !(t.p_bo3 < 0.0 && t.p_bo4 > 0.0)) || (on_pp && !(t.p_bo5 < 0.0 && t.p_bo6 > 0.0))) return RX_BOND_CUT;
  auto bo = [&](double r) {
    const double lr = std::log(r);
    double b = 0.0;
    if (on_s) b += (1.0 + P.bo_cut) * std::exp(t.p_bo1 * std::exp(t.p_bo2 * (lr - t.lr_s)));
    if (on_p) b += std::exp(t.p_bo3 * std::exp(t.p_bo4 * (lr - t.lr_p)));
    if (on_pp) b += std::exp(t.p_bo5 * std::exp(t.p_bo6 * (lr - t.lr_pp)));
    return b;
  };
  if (!(P.bo_cut > 0.0) || bo(RX_BOND_CUT) >= P.bo_cut) return RX_BOND_CUT;
  double lo = 1e-3, hi = RX_BOND_CUT;
  if (bo(lo) < P.bo_cut) return lo;
  for (int it = 0; it < 200 && hi - lo > 1e-12; it++) {
    const double mid = 0.5 * (lo + hi);
    (bo(mid) >= P.bo_cut ? lo : hi) = mid;
  }
  return std::min((double)RX_BOND_CUT, hi * (1.0 + 1e-6) + 1e-6);
}
static void rx_near_radii(RxParams &P, double skin) {
  const bool full = scema_env("SCEMA_MD_RX_NEAR_FULL") && atoi(scema_env("SCEMA_MD_RX_NEAR_FULL")) != 0;   // (test hook: every pair inside the bond cutoff)
  for (int a = 0; a < RX_MAXT; a++)
    for (int b = 0; b < RX_MAXT; b++) {
      const bool used = a < P.nt && b < P.nt;
      const double reach = !used ? 0.0 : full ? (double)RX_BOND_CUT : std::max(rx_bond_reach(P, a, b), rx_bond_reach(P, b, a));
      if (used && b >= a && scema_env("SCEMA_MD_TIMING")) fprintf(stderr, "[scema_md] reax types %d-%d: bond order below bo_cut beyond %.4f A\n", a, b, reach);
      P.rbond[a * RX_MAXT + b] = reach;
      const double rn = std::max(reach, (double)RX_PM_RADIUS) + skin;
      P.rnear2[a * RX_MAXT + b] = used ? rn * rn : 0.0;
    }
}

int scema_md_reax_configure(scema_md_engine *e, const char *ffield_path, const char *const *elements, int32_t n_elements, double qeq_tol, double skin) {
  if (!e || !ffield_path || !elements || n_elements <= 0) return fail(e, SCEMA_MD_ERR_ARG, "bad arguments");
  HIPCHK(hipSetDevice(e->p.device));
  std::vector<std::string> el(elements, elements + n_elements);
  std::string err;
  RxParams P;
  std::vector<int> map;
  if (!scema::read_reax_ffield(ffield_path, el, P, map, err)) return fail(e, SCEMA_MD_ERR_IO, "%s", err.c_str());
  if (const char *x = scema_env("SCEMA_REAX_DROP_DSBO2")) P.lammps_dsbo2 = atoi(x) ? 1 : 0;
  e->rx_host = P;
  e->rx_type_map = map;
  if (qeq_tol > 0.0) e->rx_qeq_tol = qeq_tol;
  if (skin >= 0.0) e->rx_skin = skin;
  if (const char *x = scema_env("SCEMA_REAX_SKIN")) e->rx_skin = atof(x);
  if (const char *x = scema_env("SCEMA_REAX_QEQ_LAUNCH")) { e->rx_qeq_launch = e->rx_qeq_launch_cold = std::max(0, atoi(x)); e->rx_qeq_launch_pinned = true; }
  rx_near_radii(e->rx_host, e->rx_skin);
  HIPCHK(e->d_rxparams.ensure(sizeof(RxParams)));
  HIPCHK(hipMemcpyAsync(e->d_rxparams.p, &e->rx_host, sizeof(RxParams), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  e->rx_ready = true;
  e->rx_stamp += 1;
  e->reax_active = true;
  return SCEMA_MD_OK;
}
int scema_md_reax_activate(scema_md_engine *e, int32_t on) {
  if (!e) return SCEMA_MD_ERR_ARG;
  if (on && !e->rx_ready) return fail(e, SCEMA_MD_ERR_ARG, "no ReaxFF force field loaded");
  e->reax_active = on != 0;
  return SCEMA_MD_OK;
}
int scema_md_reax_concurrency(scema_md_engine *e, int32_t halves, int32_t overlap) {
  if (!e) return SCEMA_MD_ERR_ARG;
  if (halves >= 0) e->rx_halves = std::min(halves, 8);
  if (overlap >= 0) e->rx_overlap = overlap != 0;
  return SCEMA_MD_OK;
}
int scema_md_batch_split(scema_md_engine *e, int32_t on) {
  if (!e) return SCEMA_MD_ERR_ARG;
  if (on >= 0) e->split_streams = on != 0;
  return SCEMA_MD_OK;
}
int scema_md_get_concurrency(const scema_md_engine *e, int32_t *out) {
  if (!e || !out) return SCEMA_MD_ERR_ARG;
  out[0] = e->split_streams ? 1 : 0;
  out[1] = e->rx_halves;
  out[2] = e->rx_overlap ? 1 : 0;
  return SCEMA_MD_OK;
}
int scema_md_pppm_plan_count(const scema_md_engine *e) { return e ? (int)e->pppm_plans.size() : -SCEMA_MD_ERR_ARG; }
int scema_md_reax_set(scema_md_engine *e, int32_t exact_gradient, int32_t terms, int32_t qeq_maxiter) {
  if (!e || !e->rx_ready) return fail(e, SCEMA_MD_ERR_ARG, "no ReaxFF force field loaded");
  HIPCHK(hipSetDevice(e->p.device));
  if (exact_gradient >= 0) {
    e->rx_host.lammps_dsbo2 = exact_gradient ? 0 : 1;
    HIPCHK(hipMemcpyAsync(e->d_rxparams.p, &e->rx_host, sizeof(RxParams), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
  }
  if (terms >= 0) e->rx_terms = terms;
  if (qeq_maxiter > 0) e->rx_qeq_maxiter = qeq_maxiter;
  return SCEMA_MD_OK;
}
// static evaluation of a state (qp_id SCEMA_MD_QP_NONE: the registered replica): forces, the 13 energy parts, virial, charges
int scema_md_reax_debug_compute(scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica, double *f, double *eparts, double *virial,
                                double *q, double *info) {
  if (!e) return SCEMA_MD_ERR_ARG;
  if (!e->rx_ready) return fail(e, SCEMA_MD_ERR_ARG, "no ReaxFF force field loaded");
  HIPCHK(hipSetDevice(e->p.device));
  State *s = nullptr;
  std::unique_ptr<State> tmp;
  int rc = debug_state(e, qp_id, matid, replica, &s, tmp);
  if (rc) return rc;
  std::vector<ActiveSim> sims(1);
  sims[0].st = s;
  sims[0].nsteps = 0;
  sims[0].dt = 1.0;
  sims[0].temperature = 300.0;
  const bool saved = e->reax_active;
  e->reax_active = true;
  const long long it0 = e->rx_qeq_iters;
  for (int attempt = 0; attempt < 6; attempt++) {
    if ((rc = prepare_slots(e, sims))) break;
    RunSpec R;
    R.nvt = 0;
    R.static_only = 1;
    rc = run_phase(e, sims, R);
    if (rc != SCEMA_MD_ERR_OVERFLOW) break;
    e->neigh_grow *= 1.5;
  }
  e->reax_active = saved;
  if (rc) return rc;
  const int n = s->topo->natoms;
  const SimScalars &sc = e->h_sc[0];
  const RxView &V = e->h_rxviews[0];
  if (f) HIPCHK(hipMemcpy(f, e->slots[0]->f.p, 3 * (size_t)n * 8, hipMemcpyDeviceToHost));
  if (eparts) HIPCHK(hipMemcpy(eparts, V.eparts, RX_NPART * 8, hipMemcpyDeviceToHost));
  if (q) HIPCHK(hipMemcpy(q, V.q, (size_t)n * 8, hipMemcpyDeviceToHost));
  if (virial)
    for (int k = 0; k < 6; k++) {
      virial[k] = 0.0;
      for (int p = 0; p < MD_NPART; p++) virial[k] += sc.vir[p * 6 + k];
    }
  if (info) {
    info[0] = sc.maxneigh_seen;
    info[1] = V.maxnb;
    info[2] = V.maxbd;
    info[3] = (double)(e->rx_qeq_iters - it0);
    info[4] = V.mimg[0] + V.mimg[1] + V.mimg[2];
    std::vector<int> bc(n);
    HIPCHK(hipMemcpy(bc.data(), V.bd_cnt, (size_t)n * 4, hipMemcpyDeviceToHost));
    int mb = 0;
    for (int v : bc) mb = std::max(mb, v);
    info[5] = mb;
  }
  return SCEMA_MD_OK;
}
int scema_md_reax_stats(const scema_md_engine *e, double *out) {
  if (!e || !out) return SCEMA_MD_ERR_ARG;
  out[0] = (double)e->rx_qeq_iters;
  out[1] = (double)e->rx_qeq_solves;
  out[2] = e->rx_skin;
  out[3] = e->rx_qeq_tol;
  out[4] = (double)e->rx_qeq_slow;
  out[5] = (double)e->rx_qeq_launch;
  out[6] = (double)e->rx_precond_fallbacks;
  return SCEMA_MD_OK;
}

}  // extern "C"
