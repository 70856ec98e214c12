// engine_core.cpp -- the engine object: creation, destruction, replica registration, counters
#include "engine.h"

namespace scema_eng {

int fail(scema_md_engine *e, int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (e) e->err = buf;
  return code;
}

std::string topo_key(const char *matid, int replica) { return std::string(matid ? matid : "") + "_" + std::to_string(replica); }
std::string state_key(int qp, const char *matid, int replica) { return std::to_string(qp) + "." + topo_key(matid, replica); }

}  // namespace scema_eng

extern "C" {

void scema_md_default_params(scema_md_params *p) {
  p->cut_lj = 12.0;
  p->cut_coul = 9.0;
  p->skin = 2.0;
  p->neigh_delay = 5;
  p->kspace_accuracy = 1.0e-4;
  p->shake_tol = 1.0e-3;
  p->shake_maxiter = 20;
  p->shake_mass = 1.0;
  p->t_period = 100.0;
  p->t_chain = 3;
  p->device = 0;
  p->max_batch = 0;
  p->profile = 0;
  p->kspace_style = 1;   // kspace_style pppm 0.0001 (in.set.lammps:36); 0: the plain Ewald sum at the same accuracy
}

int scema_md_create(const scema_md_params *p, scema_md_engine **out) {
  if (!out) return SCEMA_MD_ERR_ARG;
  *out = nullptr;
  scema_md_engine *e = new scema_md_engine();
  if (p) e->p = *p; else scema_md_default_params(&e->p);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    delete e;
    return SCEMA_MD_ERR_DEVICE;  // no GPU: the product path has no CPU fallback
  }
  if (hipSetDevice(e->p.device) != hipSuccess || hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) {
    delete e;
    return SCEMA_MD_ERR_DEVICE;
  }
  if (getenv("SCEMA_MD_GRAPH")) e->use_graphs = true;
  if (const char *sp = getenv("SCEMA_MD_SPLIT")) e->split_streams = atoi(sp) != 0;
  if (const char *sp = getenv("SCEMA_MD_SPLIT_MIN")) e->split_min = std::max(2, atoi(sp));
  if (hipStreamCreateWithFlags(&e->stream3, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&e->ev_up, hipEventDisableTiming) != hipSuccess)
    e->stream3 = nullptr;   // an optimisation only
  if (const char *sx = getenv("SCEMA_MD_SKIN_EXTRA")) e->skin_extra_fixed = std::max(-0.75 * e->p.skin, atof(sx));
  if (const char *sx = getenv("SCEMA_MD_SKIN_ADAPT")) e->skin_adapt = atoi(sx) != 0;
  if (!getenv("SCEMA_MD_ONE_STREAM")) {
    if (hipStreamCreateWithFlags(&e->stream2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming) != hipSuccess)
      e->stream2 = nullptr;   // side stream is an optimisation only
  }
  // test hook: start with undersized neighbour capacities, so that the overflow -> restore -> regrow path runs
  if (const char *g0 = getenv("SCEMA_MD_NEIGH_GROW0")) e->neigh_grow = e->jtab_grow = std::max(0.05, atof(g0));
  *out = e;
  return SCEMA_MD_OK;
}

void scema_md_destroy(scema_md_engine *e) {
  if (!e) return;
  (void)hipSetDevice(e->p.device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  scema_md_comm_destroy(e);
  e->comm.d_gather.release();
  e->comm.d_box.release();
  e->comm.d_word.release();
  for (hipEvent_t ev : e->ev_pool) (void)hipEventDestroy(ev);
  if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
  if (e->ev_join) (void)hipEventDestroy(e->ev_join);
  if (e->stream2) (void)hipStreamDestroy(e->stream2);
  if (e->stream3) (void)hipStreamDestroy(e->stream3);
  if (e->ev_up) (void)hipEventDestroy(e->ev_up);
  e->states.clear();
  e->topos.clear();
  e->slots.clear();
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

const char *scema_md_last_error(const scema_md_engine *e) { return e ? e->err.c_str() : "null engine"; }

int scema_md_register_replica(scema_md_engine *e, const char *matid, int32_t replica, const scema_md_system *sys) {
  if (!e || !matid || !sys) return SCEMA_MD_ERR_ARG;
  HIPCHK(hipSetDevice(e->p.device));
  std::unique_ptr<Topo> t(new Topo());
  int rc = build_topo(e, sys, *t);
  if (rc) return rc;
  // re-registering a replica invalidates every state that was derived from the old one
  const std::string suffix = "." + topo_key(matid, replica);
  for (auto it = e->states.begin(); it != e->states.end();) {
    const std::string &k = it->first;
    if (k.size() >= suffix.size() && k.compare(k.size() - suffix.size(), suffix.size(), suffix) == 0) it = e->states.erase(it);
    else ++it;
  }
  e->dir.erase_suffix(suffix);
  e->topos[topo_key(matid, replica)] = std::move(t);
  return SCEMA_MD_OK;
}


int scema_md_get_profile(scema_md_engine *e, scema_md_profile *out, int32_t reset) {
  if (!e || !out) return SCEMA_MD_ERR_ARG;
  out->pair_launches = e->prof.pair_launches;
  out->pair_ms = e->prof.pair_ms;
  out->pair_alg_bytes = e->prof.pair_alg_bytes;
  out->md_steps = e->prof.md_steps;
  out->neigh_builds = e->prof.neigh_builds;
  out->unique_pairs_per_sim = e->prof.unique_pairs_n ? e->prof.unique_pairs_sum / e->prof.unique_pairs_n : 0.0;
  out->evals = e->prof.evals;
  out->list_skin_mean = e->prof.evals ? e->prof.skin_sum / (double)e->prof.evals : 0.0;
  out->pair_sims = e->prof.pair_sims;
  out->box_flips = e->prof.box_flips;
  out->rx_sweep_launches = e->prof.rx_sweep_launches;
  out->rx_sweep_ms = e->prof.rx_sweep_ms;
  out->rx_sweep_entries = e->prof.rx_sweep_entries;
  out->rx_sweep_rows = e->prof.rx_sweep_rows;
  out->rx_sweep_col_bytes = e->prof.rx_sweep_col_bytes;
  if (reset) e->prof = Profile();
  return SCEMA_MD_OK;
}

}  // extern "C"
