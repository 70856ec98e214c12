// engine_batch.cpp -- the hot path behind the C ABI of include/scema_md.h: scema_md_strain_batch.
//
// Replaces STMDProblem<3>::lammps_straining (reference headers/stmd_problem.h:84-383): instead of two LAMMPS lifetimes and
// three restart files per quadrature-point replica, every replica state (x, v, box) stays resident in HBM keyed by
// (qp_id, matid, replica); a whole vector of MDSim requests is advanced in lockstep by the kernels of md_kernels.hip /
// md_pair.hip / md_bonded.hip / md_pppm.hip, one launch per stage for the whole batch, with no host synchronisation inside a run.
//
// Host-side arithmetic restated here, with the reference line it follows:
//   lbdim / strain correction ........ stmd_problem.h:210-225
//   nts rule ......................... stmd_problem.h:229-232
//   "%f" dts/tempt, "%.6e" rates ..... stmd_problem.h:164,235,241
//   state branch rule ................ stmd_problem.h:116-138,185-207 (engine_state.cpp resolve_state)
//   stress = -<P> * 1.01325e5 ........ stmd_problem.h:335-341
//   Hooke fallback ................... stmd_problem.h:386-392,479-483
//   force-field check ................ stmd_problem.h:462-467
#include "engine.h"
#include "../md_env.h"

namespace scema_eng {

void hooke(const double *c, const double *eps, double *out) {
  static const int RAW_OF[3][3] = {{0, 3, 4}, {3, 1, 5}, {4, 5, 2}};
  static const int FILE_OF[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
  for (int k = 0; k < 3; k++)
    for (int l = k; l < 3; l++) {
      double acc = 0.0;
      for (int m = 0; m < 3; m++)
        for (int n = 0; n < 3; n++) acc += c[FILE_OF[k][l] * 6 + FILE_OF[m][n]] * eps[RAW_OF[m][n]];
      out[RAW_OF[k][l]] = acc;
    }
}


// positions and velocities of a chunk's states -> the engine's backup pool from position `pool_off` on (restore: the other
// way), all in one launch
static int backup_states(scema_md_engine *e, std::vector<ActiveSim> &chunk, bool restore, size_t pool_off = 0) {
  const int ns = (int)chunk.size();
  e->h_copytab.resize(2 * (size_t)ns);
  while (e->bak_x.size() < pool_off + (size_t)ns) {
    e->bak_x.emplace_back(new DevBuf());
    e->bak_v.emplace_back(new DevBuf());
  }
  long long maxn = 0;
  for (int i = 0; i < ns; i++) {
    DevBuf &bx = *e->bak_x[pool_off + i], &bv = *e->bak_v[pool_off + i];
    const long long n = 3 * (long long)chunk[i].st->topo->natoms;
    HIPCHK(bx.ensure((size_t)n * 8));
    HIPCHK(bv.ensure((size_t)n * 8));
    double *x = chunk[i].st->x.as<double>(), *v = chunk[i].st->v.as<double>(), *xb = bx.as<double>(), *vb = bv.as<double>();
    e->h_copytab[2 * i] = restore ? MdkCopy{xb, x, n} : MdkCopy{x, xb, n};
    e->h_copytab[2 * i + 1] = restore ? MdkCopy{vb, v, n} : MdkCopy{v, vb, n};
    maxn = std::max(maxn, n);
  }
  HIPCHK(e->d_copytab.ensure(2 * (size_t)std::max(ns, 1) * sizeof(MdkCopy)));
  HIPCHK(hipMemcpyAsync(e->d_copytab.p, e->h_copytab.data(), 2 * (size_t)ns * sizeof(MdkCopy), hipMemcpyHostToDevice, e->stream));
  mdk_copy_many(e->stream, e->d_copytab.as<MdkCopy>(), 2 * ns, maxn);
  if (restore)
    for (int i = 0; i < ns; i++) {
      std::memcpy(chunk[i].st->box, chunk[i].box0, sizeof chunk[i].box0);
      chunk[i].st->skin_extra = chunk[i].skin0;
      chunk[i].st->qhist_valid = false;   // (what it holds belongs to the positions that were just taken back)
    }
  else
    for (int i = 0; i < ns; i++) {
      std::memcpy(chunk[i].box0, chunk[i].st->box, sizeof chunk[i].box0);
      chunk[i].skin0 = chunk[i].st->skin_extra;
    }
  return SCEMA_MD_OK;
}

// full evaluation (phase A + phase B) of a chunk of simulations, with overflow retry
int eval_chunk(scema_md_engine *e, std::vector<ActiveSim> &chunk, const EvalOpt &opt, size_t pool_off) {
  const int ns = (int)chunk.size();
  // ReaxFF: the approximate-inverse preconditioner of the charge solve is symmetrised by hand and not guaranteed positive definite on every
  // geometry; conjugate gradients that stall on it end the run with "did not converge".  The evaluation is then repeated ONCE
  // from its backup with the preconditioner fix qeq/reax itself uses (the diagonal), which is, before the error is reported.
  struct PrecondScope { scema_md_engine *e; bool saved; ~PrecondScope() { e->rx_precond = saved; } } precond_scope{e, e->rx_precond};
  bool precond_retry = false;
  for (int attempt = 0; attempt < 6; attempt++) {
    e->rx_qeq_failed = false;
    int rc = prepare_slots(e, chunk);
    if (rc) return rc;
    // backup for a retry after neighbour overflow
    rc = backup_states(e, chunk, false, pool_off);
    if (rc) return rc;
    RunSpec A;
    A.deform = 1;
    A.use_shake = opt.shake_a;
    A.keep_list = (attempt == 0) ? 2 : 0;   // the slots may hold these states' rows from the update before (a retry after an overflow rebuilds)
    for (int i = 0; i < ns; i++) chunk[i].nsteps = chunk[i].nts;
    const double t_a0 = wall_s();
    rc = opt.phase_a ? run_phase(e, chunk, A) : SCEMA_MD_OK;
    const double t_a1 = wall_s();
    if (rc == SCEMA_MD_OK) {
      if (opt.phase_a) {
        rc = reupload_scalars(e, ns);
        if (rc) return rc;
      }
      RunSpec B;
      B.sample = 1;
      B.use_shake = opt.shake_b;
      B.keep_list = opt.phase_a ? 1 : 0;      // ... and so are the cell grids and neighbour rows (run_phase keeps them where the grid can stay)
      B.qeq_continue = opt.phase_a ? 1 : 0;   // same simulations, same slots: the ReaxFF solver history of phase A is in place
      for (int i = 0; i < ns; i++) chunk[i].nsteps = chunk[i].nss;
      rc = run_phase(e, chunk, B);
      if (scema_env("SCEMA_MD_TIMING")) fprintf(stderr, "[scema_md] chunk of %d: phase A %.1f ms, phase B %.1f ms (attempt %d)\n", ns, 1e3 * (t_a1 - t_a0), 1e3 * (wall_s() - t_a1), attempt);
    }
    if (rc == SCEMA_MD_OK) {
      for (int i = 0; i < ns; i++) {
        const SimScalars &sc = e->h_sc[i];
        std::memcpy(chunk[i].st->box, sc.box, 9 * sizeof(double));
        for (int k = 0; k < 6; k++) chunk[i].pavg[k] = sc.psum[k] / (double)std::max(sc.nsamples, 1);
        e->prof.skin_sum += e->p.skin + chunk[i].st->skin_extra;
        // steps per list rebuild of the sampling run -> list skin of this state's next evaluation (with hysteresis)
        if (e->skin_adapt && chunk[i].nss >= 50) {
          const double interval = (double)chunk[i].nss / (double)std::max(sc.nbuilds, 1);
          State &st = *chunk[i].st;
          if (st.skin_extra == 0.0 && interval < 19.0) st.skin_extra = 0.25 * e->p.skin;
          else if (st.skin_extra > 0.0 && interval > 40.0) st.skin_extra = 0.0;
        }
      }
      e->prof.evals += ns;
      return SCEMA_MD_OK;
    }
    if (rc != SCEMA_MD_ERR_OVERFLOW && e->rx_qeq_failed && e->rx_precond && !precond_retry) {
      precond_retry = true;
      e->rx_precond = false;
      e->rx_precond_fallbacks += 1;
      fprintf(stderr, "[scema_md] the charge equilibration did not converge with the approximate-inverse preconditioner: the evaluation of these %d replicas is repeated with the Jacobi preconditioner of fix qeq/reax\n", ns);
      rc = backup_states(e, chunk, true, pool_off);
      if (rc) return rc;
      HIPCHK(hipStreamSynchronize(e->stream));
      continue;
    }
    if (rc != SCEMA_MD_ERR_OVERFLOW) {
      // instability, box error, non-finite stress, device error: the reference would have stopped before write_restart
      // (stmd_problem.h:258), so the stored states must not keep the half-advanced positions
      (void)backup_states(e, chunk, true, pool_off);
      (void)hipStreamSynchronize(e->stream);
      return rc;
    }
    // restore and grow
    rc = backup_states(e, chunk, true, pool_off);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
    // (by the demand the failed run saw where that is more than the fixed step: a capacity far too small is found in one retry, not six)
    if (e->overflow_bits & 4) e->jtab_grow *= std::max(1.25, std::min(8.0, 1.1 * e->overflow_need_j));
    if ((e->overflow_bits & 8) || !(e->overflow_bits & 4)) e->neigh_grow *= std::max(1.5, std::min(8.0, 1.1 * e->overflow_need_row));
  }
  return fail(e, SCEMA_MD_ERR_OVERFLOW, "neighbour capacity exceeded after regrowth");
}

// the update that is waiting for the caller's collective (PendingUpdate): it stands (owners committed, stale copies dropped) or it
// never happened (states back to their backups, new states gone, directory as it was)
int settle_pending(scema_md_engine *e, bool failed) {
  PendingUpdate &P = e->pending;
  if (!P.active) return SCEMA_MD_OK;
  P.active = false;
  if (failed) {
    if (!P.act.empty()) {
      (void)backup_states(e, P.act, true, 0);
      (void)hipStreamSynchronize(e->stream);
    }
    for (auto it = P.created.rbegin(); it != P.created.rend(); ++it) {
      if (it->displaced) e->states[it->key] = std::move(it->displaced);
      else e->states.erase(it->key);
    }
  } else {
    e->dir.commit(P.plan, P.dst_keys);
    for (size_t i = 0; i < P.dst_keys.size(); i++)
      if (P.plan.owner[i] != P.rank) e->states.erase(P.dst_keys[i]);
  }
  P.act.clear();
  P.created.clear();
  P.dst_keys.clear();
  return SCEMA_MD_OK;
}

}  // namespace scema_eng

extern "C" {

/* The caller's collective has shown the status words of every rank (scema_md_scatter_gathered calls this itself; a host that
 * scatters on its own -- STMDSync::share_stresses over its callback -- says so here): failed != 0 takes this rank's share of the last
 * update back, 0 lets it stand.  Without a pending update (communicator attached, single rank, Hooke mode) nothing happens. */
int scema_md_settle_update(scema_md_engine *e, int32_t failed) {
  if (!e) return SCEMA_MD_ERR_ARG;
  return settle_pending(e, failed != 0);
}

// ---- the hot path ----
// straining steps of a request (stmd_problem.h:213-232) for a box of the given lengths
static int nts_rule(const scema_mdsim &m, const double lb[3], double eps[6], double *norm) {
  const double *sl = m.strain;
  eps[0] = sl[0] / lb[0]; eps[1] = sl[1] / lb[1]; eps[2] = sl[2] / lb[2];
  eps[3] = sl[3] / lb[2];  // [0][1] /= lbdim[2]
  eps[5] = sl[5] / lb[0];  // [1][2] /= lbdim[0]
  eps[4] = sl[4] / lb[1];  // [2][0] /= lbdim[1]
  // stmd_problem.h:229-232
  const double nrm = std::sqrt(eps[0] * eps[0] + eps[1] * eps[1] + eps[2] * eps[2] + 2.0 * (eps[3] * eps[3] + eps[4] * eps[4] + eps[5] * eps[5]));
  if (norm) *norm = nrm;
  if (!std::isfinite(nrm) || !(m.strain_rate > 0.0) || !(m.timestep_length > 0.0)) return 10;
  const double steps = nrm / m.strain_rate / m.timestep_length;
  if (!(steps < 1.0e7)) return 10;
  return std::max((int)(std::ceil(steps / 10.0) * 10), 10);
}

int scema_md_strain_batch(scema_md_engine *e, scema_mdsim *sims, int32_t n_sims, int32_t hooke_mode, int32_t rank, int32_t world) {
  if (!e || (!sims && n_sims > 0) || n_sims < 0 || world <= 0 || rank < 0 || rank >= world) return fail(e, SCEMA_MD_ERR_ARG, "bad arguments");
  if (e->comm.kind && (e->comm.rank != rank || e->comm.world != world))
    return fail(e, SCEMA_MD_ERR_ARG, "rank/world (%d/%d) differ from the attached communicator (%d/%d)", rank, world, e->comm.rank, e->comm.world);
  // Rank-local findings before anything runs are collected in pre_status and exchanged in the handshake: with a communicator a rank
  // that cannot go on (device error, force-field file unreadable here, replica not registered here) still enters the handshake, so
  // that every rank ends the call with the same error instead of the others waiting in it
  int pre_status = SCEMA_MD_OK;
  const bool collective_call = e->comm.kind && world > 1;
  if (hipSetDevice(e->p.device) != hipSuccess) {
    if (!collective_call) return fail(e, SCEMA_MD_ERR_DEVICE, "hipSetDevice(%d) failed", e->p.device);
    pre_status = fail(e, SCEMA_MD_ERR_DEVICE, "hipSetDevice(%d) failed on rank %d", e->p.device, rank);
  }
  if (e->pending.active) {
    // An update of a world > 1 without a communicator that nobody settled (scema_md_scatter_gathered / scema_md_settle_update): it is taken
    // to stand.  A host written against the round-3 interface, which scatters the gathered buffer itself, ends up here after every update:
    // if another rank's share had failed, that rank has rolled back and this one commits -- the directories part ways and the next plan hash
    // says so.  Counted (scema_md_unsettled_updates) and said once.
    e->unsettled_updates += 1;
    if (e->unsettled_updates == 1)
      fprintf(stderr, "[scema_md] warning: the previous update (rank %d of %d, no communicator attached) was never settled -- call scema_md_scatter_gathered or "
                      "scema_md_settle_update once the stresses of all ranks are known; it is taken to stand (INTEGRATION.md, \"Settling an update\")\n",
              e->pending.rank, e->pending.plan.world);
  }
  (void)settle_pending(e, false);    // an update nobody objected to stands
  e->last_plan = scema::SimPlan();   // a call that ends before planning leaves no plan behind
  // ---- the request itself: checked on every rank for every simulation, so that a request that cannot run is refused by
  // all ranks together, before anything is planned or moved ----
  std::vector<std::string> src_keys(n_sims), dst_keys(n_sims);
  std::vector<double> cost(n_sims, 1.0);
  int n_reax = 0, n_md = 0;
  for (int i = 0; i < n_sims; i++) {
    sims[i].stress_updated = 0;
    // stmd_problem.h:462-467
    const char *ff = sims[i].force_field ? sims[i].force_field : "";
    if (std::strcmp(ff, "opls") != 0 && std::strcmp(ff, "reax") != 0)
      return fail(e, SCEMA_MD_ERR_ARG, "Error: Force field is %s but only 'opls' and 'reax' are implemented... ", ff);
    if (hooke_mode) continue;   // sigma = C:eps has no state: the fresh-batch rule of the planner = i % world (stmd_sync.h:583)
    n_md++;
    if (std::strcmp(ff, "reax") == 0) n_reax++;
    // requests that cannot be run: LAMMPS would stop while parsing "variable ceeps_.. equal nan" or "timestep 0"
    bool finite = true;
    for (int k = 0; k < 6; k++) finite = finite && std::isfinite(sims[i].strain[k]);
    if (!finite) return fail(e, SCEMA_MD_ERR_ARG, "quadrature point %d: non-finite strain", sims[i].qp_id);
    if (!(sims[i].strain_rate > 0.0) || !std::isfinite(sims[i].strain_rate) || !(sims[i].timestep_length > 0.0) ||
        !std::isfinite(sims[i].timestep_length) || !(sims[i].temperature > 0.0) || !std::isfinite(sims[i].temperature))
      return fail(e, SCEMA_MD_ERR_ARG, "quadrature point %d: strain rate, time step and temperature must be positive and finite", sims[i].qp_id);
    if (sims[i].nsteps_sample < 1) return fail(e, SCEMA_MD_ERR_ARG, "number of sampling steps must be >= 1");
  }
  if (n_reax != 0 && n_reax != n_md) return fail(e, SCEMA_MD_ERR_ARG, "one update mixes force fields (%d of %d simulations ask for 'reax'): md_force_field is one setting per run", n_reax, n_md);
  if (n_reax && !e->rx_ready) {
    // the reference's scripts name the file and the elements: pair_coeff * * ${locs}/ffield.reax.2 H C N O
    // (lammps_scripts_reax/in.strain.lammps:11, locs = MDSim.scripts_folder, stmd_problem.h:163)
    static const char *hcno[4] = {"H", "C", "N", "O"};
    const std::string path = std::string(sims[0].scripts_folder ? sims[0].scripts_folder : ".") + "/ffield.reax.2";
    const int rc_cfg = scema_md_reax_configure(e, path.c_str(), hcno, 4, 1e-6, -1.0);
    if (rc_cfg && !collective_call) return rc_cfg;
    if (rc_cfg && !pre_status) pre_status = rc_cfg;   // (a file one rank cannot read: the others must hear of it)
  }
  // ---- who runs what (host/sim_plan.h): identical on every rank ----
  for (int i = 0; i < n_sims && !hooke_mode; i++) {
    dst_keys[i] = state_key(sims[i].qp_id, sims[i].matid, sims[i].replica);
    // stmd_problem.h:116-120: the state is read under most_recent_qp_id ("none" -> init.<mat>_<rep>.bin)
    if (sims[i].most_recent_qp_id == sims[i].qp_id) src_keys[i] = dst_keys[i];
    else if (sims[i].most_recent_qp_id != SCEMA_MD_QP_NONE) src_keys[i] = state_key(sims[i].most_recent_qp_id, sims[i].matid, sims[i].replica);
    // cost = MD steps of the evaluation, estimated with the replica's registered box (the same on every rank: replicas
    // are registered collectively; the plan hash of the handshake says so if they were not)
    Topo *t = find_topo(e, sims[i].matid, sims[i].replica);
    if (!t) {
      if (world == 1 || !e->comm.kind)
        return fail(e, SCEMA_MD_ERR_NOSTATE, "replica %s_%d is not registered (init.%s_%d.bin missing)", sims[i].matid, sims[i].replica, sims[i].matid, sims[i].replica);
      if (!pre_status) pre_status = fail(e, SCEMA_MD_ERR_NOSTATE, "replica %s_%d is not registered on rank %d (init.%s_%d.bin missing)", sims[i].matid, sims[i].replica, rank, sims[i].matid, sims[i].replica);
      continue;
    }
    const double lb0[3] = {t->init_box[3] - t->init_box[0], t->init_box[4] - t->init_box[1], t->init_box[5] - t->init_box[2]};
    double eps0[6];
    cost[i] = (double)nts_rule(sims[i], lb0, eps0, nullptr) + (double)std::max(sims[i].nsteps_sample, 1);
  }
  struct ReaxScope {   // the force field of this update; the debug entry points keep whatever scema_md_reax_activate chose
    scema_md_engine *e; bool saved;
    ReaxScope(scema_md_engine *e_, bool on) : e(e_), saved(e_->reax_active) { e->reax_active = on; }
    ~ReaxScope() { e->reax_active = saved; }
  } reax_scope(e, n_reax > 0);
  e->last_plan = e->dir.plan(src_keys, dst_keys, cost, world);
  if (e->comm.kind == 1 && world == 1 && !hooke_mode && scema_env("SCEMA_MD_TEST_SELF_MOVE")) {
    // test hook: with one rank no state ever changes GPU, so ncclSend / ncclRecv would never run on a one-GPU box.  Every simulation
    // that continues from a state held here gets a move from this rank to itself: the state travels through the RCCL group of
    // migrate_states like any other and the simulation continues from the received copy (results do not change)
    for (int i = 0; i < n_sims; i++)
      if (!src_keys[i].empty() && e->states.count(src_keys[i])) e->last_plan.moves.push_back({i, rank, rank});
  }
  const scema::SimPlan &plan = e->last_plan;
  const int per_rank = plan.cap;
  const double hash = plan_hash(plan, cost);
  e->local_stress_count = per_rank;
  const size_t nres = 6 * (size_t)std::max(per_rank, 1) + SCEMA_MD_RESULT_TRAILER;
  if (e->d_local_stress.ensure(nres * sizeof(double)) != hipSuccess) {
    if (!collective_call) return fail(e, SCEMA_MD_ERR_DEVICE, "out of device memory for the result buffer");
    if (!pre_status) pre_status = fail(e, SCEMA_MD_ERR_DEVICE, "out of device memory for the result buffer on rank %d", rank);
  }
  std::vector<double> local(nres, 0.0);
  local[nres - 1] = hash;
  const bool collective = collective_call;
  // without a communicator the caller gathers this buffer: it must say what happened to this rank's share whenever a plan exists
  auto publish = [&](int st) {
    local[nres - 2] = (double)st;
    (void)hipMemcpyAsync(e->d_local_stress.p, local.data(), local.size() * sizeof(double), hipMemcpyHostToDevice, e->stream);
    (void)hipStreamSynchronize(e->stream);
    return st;
  };
  // Everything the exchange of states needs on this rank is made BEFORE the handshake (prepare_incoming): the source states it is
  // recorded to own are looked up (one it does not hold is found before any rank posts a receive for it), the states that migrate to it
  // are allocated, the boxes that travel beside them are on the device -- a rank that cannot do its part says so in its status word
  // instead of leaving its peers inside the exchange
  std::map<int, std::unique_ptr<State>> incoming;
  if (collective && !hooke_mode && !pre_status) pre_status = prepare_incoming(e, sims, plan, src_keys, incoming);
  if (collective) {
    const int rc = handshake(e, pre_status, hash);
    if (rc) return rc;
  } else if (pre_status)
    return publish(pre_status);
  // ---- states that have to change GPU first ----
  int status = SCEMA_MD_OK;   // of this rank's share; with a communicator it travels in the trailer of the all-gather
  if (!hooke_mode && !plan.moves.empty()) {
    if (!e->comm.kind) {
      const scema::PlanMove &m = plan.moves[0];
      return publish(fail(e, SCEMA_MD_ERR_NOSTATE, "the state %s that quadrature point %d continues from lives on rank %d but the simulation is planned on rank %d: "
                          "attach a communicator (scema_md_comm_init_rccl / scema_md_comm_init_host) so that states can move between GPUs",
                          src_keys[m.sim].c_str(), sims[m.sim].qp_id, m.from, m.to));
    }
    if (!collective) status = prepare_incoming(e, sims, plan, src_keys, incoming);   // (a communicator with world 1 plans no moves of its own: the self-move test hook)
    // an error of the exchange becomes this rank's status: it still enters the stress all-gather, where every rank learns of it
    if (!status) status = migrate_states(e, plan);
  }
  // ---- this rank's share ----
  std::vector<ActiveSim> act;
  std::vector<CreatedState> created;
  auto undo = [&]() {   // a failed update leaves the state store as it found it (the reference stops before write_restart)
    for (auto it = created.rbegin(); it != created.rend(); ++it) {
      if (it->displaced) e->states[it->key] = std::move(it->displaced);
      else e->states.erase(it->key);
    }
    created.clear();
  };
  for (int i = 0; i < n_sims && !status; i++) {
    if (plan.owner[i] != rank) continue;
    if (hooke_mode) {
      hooke(sims[i].stiffness, sims[i].strain, sims[i].stress);
      sims[i].stress_updated = 1;
      continue;
    }
    ActiveSim A;
    bool was_created = false;
    std::unique_ptr<State> displaced;
    auto inc = incoming.find(i);
    status = resolve_state(e, sims[i], &A.st, inc == incoming.end() ? nullptr : &inc->second, &was_created, &displaced);
    if (status) break;
    if (was_created) created.push_back({dst_keys[i], std::move(displaced)});
    A.user_index = i;
    // stmd_problem.h:213-225
    const double lb[3] = {A.st->box[3] - A.st->box[0], A.st->box[4] - A.st->box[1], A.st->box[5] - A.st->box[2]};
    double eps[6], nrm = 0.0;
    const int nts = nts_rule(sims[i], lb, eps, &nrm);
    if (!std::isfinite(nrm)) status = fail(e, SCEMA_MD_ERR_ARG, "quadrature point %d: non-finite strain", sims[i].qp_id);
    else if (nrm / sims[i].strain_rate / sims[i].timestep_length > 1.0e7)
      status = fail(e, SCEMA_MD_ERR_ARG, "quadrature point %d: %.3g straining steps requested (strain norm %.3g at rate %.3g per fs)",
                    sims[i].qp_id, nrm / sims[i].strain_rate / sims[i].timestep_length, nrm, sims[i].strain_rate);
    if (status) break;
    A.nts = nts;
    A.nss = sims[i].nsteps_sample;
    A.dt = round_trip("%f", sims[i].timestep_length);
    A.temperature = round_trip("%f", sims[i].temperature);
    for (int k = 0; k < 6; k++) A.rates[k] = round_trip("%.6e", eps[k] / (nts * sims[i].timestep_length));
    act.push_back(A);
  }
  // max_batch = 0: as many simulations per launch group as the free HBM holds (neighbour rows dominate: about 2 KB per
  // atom at the default row capacity, plus tables, slot copies and backups), at most 1024, at least the slots that
  // exist already
  int maxb = e->p.max_batch;
  if (maxb <= 0) {
    size_t free_b = 0, total_b = 0;
    size_t maxat = 1;
    for (const ActiveSim &A : act) maxat = std::max(maxat, (size_t)A.st->topo->natoms);
    maxb = 1024;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
      const double per_sim = 3300.0 * (double)maxat + 4.0e6;
      const double fit = 0.85 * (double)free_b / per_sim + (double)e->slots.size();
      maxb = (int)std::max(1.0, std::min(1024.0, fit));
    }
  }
  // (A ragged update -- straining runs of 10 to 100 steps -- as 2 / 3 / 4 launch groups of similar length instead of one, VERDICT r5 item 6:
  // 352.5 / 350.6 / 347.7 against 353.7 evaluations/s, profiles/r06_l_cumask_ragged_ab.log.  The late straining steps of the one group run
  // with few replicas left, but every further group pays the small-batch rate for ALL its steps; the one group stays.)
  size_t n_advanced = 0;   // simulations of `act` whose states have been advanced (their backups sit in the pool)
  for (size_t off = 0; off < act.size() && !status; off += maxb) {
    std::vector<ActiveSim> chunk(act.begin() + off, act.begin() + std::min(act.size(), off + (size_t)maxb));
    status = eval_chunk(e, chunk, EvalOpt(), off);   // a failed chunk has put its own states back
    if (status) break;
    for (size_t k = 0; k < chunk.size(); k++) { std::memcpy(act[off + k].box0, chunk[k].box0, sizeof chunk[k].box0); act[off + k].skin0 = chunk[k].skin0; }
    n_advanced = off + chunk.size();
    for (auto &A : chunk) {
      scema_mdsim &m = sims[A.user_index];
      for (int k = 0; k < 6; k++) m.stress[k] = A.pavg[k] * (-1.0) * 1.01325e+05;  // stmd_problem.h:340
      // a replica that blew up (overlapping atoms, a time step far too long) must not hand NaN to the FE solver:
      // LAMMPS would stop with "lost atoms" / "bond atoms missing" at this point
      for (int k = 0; k < 6 && !status; k++)
        if (!std::isfinite(m.stress[k]))
          status = fail(e, SCEMA_MD_ERR_ARG, "simulation of quadrature point %d (material %s, replica %d) produced a non-finite stress: unstable state or parameters",
                        m.qp_id, m.matid ? m.matid : "?", m.replica);
      if (status) break;
      m.stress_updated = 1;
    }
  }
  // ---- results of this rank: stresses, status word, plan hash ----
  if (status)
    for (int i = 0; i < n_sims; i++) sims[i].stress_updated = 0;
  for (int i = 0; i < n_sims; i++)
    if (plan.owner[i] == rank && sims[i].stress_updated)
      for (int k = 0; k < 6; k++) local[6 * (size_t)plan.pos[i] + k] = sims[i].stress[k];
  local[nres - 2] = (double)status;
  // ---- the one collective of the update (replaces STMDSync::share_stresses, stmd_sync.h:620-726) ----
  // (with a communicator attached it runs for a single rank too: one 48-byte-per-simulation collective costs microseconds
  // and the one-GPU test box thereby exercises the RCCL calls).  A rank whose share failed enters it all the same: the
  // status word in the trailer ends the update on every rank.
  int rc = status;
  if (e->comm.kind) {
    const int rc_g = allgather_stresses(e, local, sims, n_sims);
    if (!rc) rc = rc_g;
  } else {
    // the caller gathers (scema_md_copy_local_stress + scema_md_scatter_gathered): the buffer carries this rank's status
    (void)publish(status);
  }
  HIPCHK(hipStreamSynchronize(e->stream));
  if (rc) {
    // every rank arrives here together (or the single rank alone): states that were advanced go back to their backups,
    // new states go, the directory keeps the owners it had
    if (n_advanced) {
      std::vector<ActiveSim> done(act.begin(), act.begin() + n_advanced);
      (void)backup_states(e, done, true, 0);
      (void)hipStreamSynchronize(e->stream);
    }
    undo();
    for (int i = 0; i < n_sims; i++) sims[i].stress_updated = 0;
    return rc;
  }
  // ---- bookkeeping: every state now lives under its own key on the rank that ran it; stale copies elsewhere go ----
  if (!hooke_mode && world > 1) {
    if (!e->comm.kind) {
      // no communicator: whether the other ranks' shares succeeded is only known after the caller's collective.  The update waits
      // (backups kept, directory uncommitted) for scema_md_scatter_gathered / scema_md_settle_update -- or for the next call, which
      // lets it stand (only the failing rank used to roll back on this transport)
      PendingUpdate &P = e->pending;
      P.active = true;
      P.act.assign(act.begin(), act.begin() + n_advanced);
      P.created = std::move(created);
      P.plan = plan;
      P.dst_keys = dst_keys;
      P.rank = rank;
      return SCEMA_MD_OK;
    }
    e->dir.commit(plan, dst_keys);
    for (int i = 0; i < n_sims; i++)
      if (plan.owner[i] != rank) e->states.erase(dst_keys[i]);
  }
  return SCEMA_MD_OK;
}

int scema_md_strain(scema_md_engine *e, scema_mdsim *sim, int32_t hooke_mode) { return scema_md_strain_batch(e, sim, 1, hooke_mode, 0, 1); }

int64_t scema_md_unsettled_updates(const scema_md_engine *e) { return e ? e->unsettled_updates : 0; }

void *scema_md_local_stress_device_ptr(scema_md_engine *e) { return e ? e->d_local_stress.p : nullptr; }
int32_t scema_md_local_stress_count(const scema_md_engine *e) { return e ? e->local_stress_count : 0; }

int scema_md_copy_local_stress(scema_md_engine *e, void *dst, int32_t dst_on_device) {
  if (!e || !dst) return SCEMA_MD_ERR_ARG;
  HIPCHK(hipSetDevice(e->p.device));
  HIPCHK(hipMemcpyAsync(dst, e->d_local_stress.p, (size_t)scema_md_local_result_doubles(e) * sizeof(double),
                        dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, e->stream));
  // the caller hands dst to a collective on another stream
  HIPCHK(hipStreamSynchronize(e->stream));
  return SCEMA_MD_OK;
}

int32_t scema_md_local_result_doubles(const scema_md_engine *e) { return e ? 6 * std::max(e->local_stress_count, 1) + SCEMA_MD_RESULT_TRAILER : 0; }

int scema_md_last_plan(const scema_md_engine *e, int32_t n_sims, int32_t *owner, int32_t *pos, int32_t *cap) {
  if (!e || n_sims != (int)e->last_plan.owner.size()) return SCEMA_MD_ERR_ARG;
  for (int i = 0; i < n_sims; i++) {
    if (owner) owner[i] = e->last_plan.owner[i];
    if (pos) pos[i] = e->last_plan.pos[i];
  }
  if (cap) *cap = e->last_plan.cap;
  return SCEMA_MD_OK;
}

int scema_md_scatter_gathered(scema_md_engine *e, const double *gathered, scema_mdsim *sims, int32_t n_sims) {
  if (!e || !gathered || !sims || n_sims != (int)e->last_plan.owner.size()) return SCEMA_MD_ERR_ARG;
  const scema::SimPlan &plan = e->last_plan;
  const size_t cnt = (size_t)scema_md_local_result_doubles(e);
  for (int i = 0; i < n_sims; i++) sims[i].stress_updated = 0;
  // this rank's own failure was reported by scema_md_strain_batch already; here: somebody else's, or a plan mismatch
  const int rc = check_gathered_trailers(e, gathered, cnt, cnt - SCEMA_MD_RESULT_TRAILER, plan.world, -1, "during the update");
  (void)settle_pending(e, rc != 0);   // another rank failed (or the plans differ): this rank's share goes back as well
  if (rc) return rc;
  for (int i = 0; i < n_sims; i++) {
    const double *src = gathered + ((size_t)plan.owner[i] * cnt + 6 * (size_t)plan.pos[i]);
    for (int k = 0; k < 6; k++) sims[i].stress[k] = src[k];
    sims[i].stress_updated = 1;
  }
  return SCEMA_MD_OK;
}

}  // extern "C"
