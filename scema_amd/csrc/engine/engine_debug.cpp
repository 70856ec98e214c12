// engine_debug.cpp -- parity / measurement entry points: static evaluations and plain runs on a stored state
#include "engine.h"

namespace scema_eng {

// ---- parity / measurement hooks ----
// qp_id == SCEMA_MD_QP_NONE: a temporary copy of the registered init state (held by `tmp`)
int debug_state(scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica, State **out, std::unique_ptr<State> &tmp) {
  Topo *t = find_topo(e, matid, replica);
  if (!t) return fail(e, SCEMA_MD_ERR_NOSTATE, "replica %s_%d not registered", matid, replica);
  if (qp_id == SCEMA_MD_QP_NONE) {
    int rc = make_state(e, t, t->init_box, t->init_x.data(), t->init_v.data(), false, tmp);
    if (rc) return rc;
    *out = tmp.get();
    return SCEMA_MD_OK;
  }
  State *s = find_state(e, qp_id, matid, replica);
  if (!s) return fail(e, SCEMA_MD_ERR_NOSTATE, "no state for qp %d", qp_id);
  *out = s;
  return SCEMA_MD_OK;
}

}  // namespace scema_eng

extern "C" {

int scema_md_debug_compute(scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica, int32_t use_shake, double *f,
                           double *energies, double *virials, double *info) {
  if (e) (void)settle_pending(e, false);   // an update that waits for its verdict (no communicator) stands once the engine is used for something else
  if (!e) return SCEMA_MD_ERR_ARG;
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
  for (int attempt = 0; attempt < 6; attempt++) {
    if ((rc = prepare_slots(e, sims))) return rc;
    RunSpec R;
    R.use_shake = use_shake;
    R.ev_always = 1;
    R.static_only = 1;
    R.nvt = 0;
    rc = run_phase(e, sims, R);
    if (rc != SCEMA_MD_ERR_OVERFLOW) break;
    if (e->overflow_bits & 4) e->jtab_grow *= 1.25;
    if ((e->overflow_bits & 8) || !(e->overflow_bits & 4)) e->neigh_grow *= 1.5;
  }
  if (rc) return rc;
  const SimScalars &sc = e->h_sc[0];
  if (f) HIPCHK(hipMemcpy(f, e->slots[0]->f.p, 3 * (size_t)s->topo->natoms * 8, hipMemcpyDeviceToHost));
  if (energies) std::memcpy(energies, sc.eng, sizeof sc.eng);
  if (virials) std::memcpy(virials, sc.vir, sizeof sc.vir);
  if (info) {
    info[0] = e->h_sims[0].g_ewald;
    info[1] = e->h_sims[0].nk;
    info[2] = 0.5 * (double)sc.nentries;
    info[3] = e->h_sims[0].tdof;
    info[4] = sc.t_current;
    info[5] = sc.maxneigh_seen;
    info[4] = (double)sc.nrowent;  // row entries stored (t_current is not needed by the callers)
    info[6] = e->h_sims[0].maxneigh;
    info[7] = s->topo->nclus;
  }
  return SCEMA_MD_OK;
}

int scema_md_debug_run(scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica, int32_t nsteps, double dt,
                       double temperature, int32_t nvt, int32_t use_shake, const double *rates, double *press_avg) {
  if (e) (void)settle_pending(e, false);   // an update that waits for its verdict (no communicator) stands once the engine is used for something else
  if (!e || nsteps < 0) return SCEMA_MD_ERR_ARG;
  HIPCHK(hipSetDevice(e->p.device));
  if (qp_id == SCEMA_MD_QP_NONE) return fail(e, SCEMA_MD_ERR_ARG, "debug_run needs a stored state (scema_md_set_state first)");
  State *s = nullptr;
  std::unique_ptr<State> tmp;
  int rc = debug_state(e, qp_id, matid, replica, &s, tmp);
  if (rc) return rc;
  std::vector<ActiveSim> sims(1);
  sims[0].st = s;
  sims[0].nsteps = nsteps;
  sims[0].dt = dt;
  sims[0].temperature = temperature;
  if (rates) for (int k = 0; k < 6; k++) sims[0].rates[k] = rates[k];
  if ((rc = prepare_slots(e, sims))) return rc;
  RunSpec R;
  R.nvt = nvt;
  R.use_shake = use_shake;
  R.deform = rates ? 1 : 0;
  R.sample = press_avg ? 1 : 0;
  rc = run_phase(e, sims, R);
  if (rc) return rc;
  const SimScalars &sc = e->h_sc[0];
  std::memcpy(s->box, sc.box, 9 * sizeof(double));
  if (press_avg) for (int k = 0; k < 6; k++) press_avg[k] = sc.psum[k] / (double)std::max(sc.nsamples, 1);
  return SCEMA_MD_OK;
}

}  // extern "C"
