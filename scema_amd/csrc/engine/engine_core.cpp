// engine_core.cpp -- the engine object: creation, destruction, replica registration, counters
#include <algorithm>
#include "engine.h"
#include "../md_env.h"

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

namespace scema_eng {
double event_union_ms(const std::vector<hipEvent_t> &ev, size_t n) {
  if (n == 0) return 0.0;
  std::vector<std::pair<float, float>> iv(n);
  for (size_t l = 0; l < n; l++) {
    float a = 0.f, b = 0.f;
    if (hipEventElapsedTime(&a, ev[0], ev[2 * l]) != hipSuccess || hipEventElapsedTime(&b, ev[0], ev[2 * l + 1]) != hipSuccess) return 0.0;
    iv[l] = {a, b};
  }
  std::sort(iv.begin(), iv.end());
  double total = 0.0;
  float lo = iv[0].first, hi = iv[0].second;
  for (size_t l = 1; l < n; l++) {
    if (iv[l].first > hi) { total += hi - lo; lo = iv[l].first; hi = iv[l].second; }
    else hi = std::max(hi, iv[l].second);
  }
  return total + (hi - lo);
}
}  // namespace scema_eng

// ---- the environment switches of the library (md_env.h): name, what it does ----
namespace {
struct EnvSwitch { const char *name, *what; };
const EnvSwitch k_env[] = {
    // performance switches with a measured default; a reported run sets none of them
    {"SCEMA_MD_SPLIT_MAX", "launch groups of this many replicas and more run whole instead of as two half batches (default: none)"},
    {"SCEMA_MD_SPLIT", "0: never run a launch group of 9 simulations and more as part batches on streams of their own"},
    {"SCEMA_MD_PPPM_SIDE_MIN", "smallest batch whose PPPM chain runs on the side stream next to the pair kernel (default 1; 4 until round 5)"},
    {"SCEMA_MD_SPLIT_MIN", "launch groups from this many replicas on run as part batches on streams of their own (default 9)"},
    {"SCEMA_MD_PART_MIN", "SCEMA_MD_PARTS applies to launch groups of this many replicas per part and more (default 2; smaller groups run as two parts)"},
    {"SCEMA_MD_PARTS", "2-4: this many part batches for every launch group that is split (default: by the size of the group, engine_run.cpp)"},
    {"SCEMA_MD_CELLS_TARGET", "what-if: take the cell grid (= tiling of the pair kernel) whose number of cells is closest to this among the grids that fit"},
    {"SCEMA_MD_ONE_STREAM", "no side stream (bonded / k-space chain beside the pair kernel)"},
    {"SCEMA_MD_KEEP_LIST", "0: the sampling run of an evaluation rebuilds its neighbour rows at its start even where those of the straining run still hold"},
    {"SCEMA_MD_SKIN_EXTRA", "list skin = params.skin + this many Angstrom (results do not depend on it)"},
    {"SCEMA_MD_SKIN_ADAPT", "1: per-state adaptation of the extra skin from the rebuild interval (round-1 behaviour)"},
    {"SCEMA_MD_PPPM_SOLVE_WIDE", "1 / 0: the in-LDS PPPM solve with 1 024 threads and three LDS grids / 512 threads and two (default: by batch size)"},
    {"SCEMA_MD_PPPM_SOLVE_TWO", "0 / 1: the in-LDS PPPM solve of the 1 024-thread shape as one / two workgroups per replica, one per transform back (default: two for batches under 8 replicas)"},
    {"SCEMA_MD_PPPM_PADX", "0: the LDS grid of the PPPM spreading kernel without the five pad points per x row (an address addition per stencil point instead of a constant offset)"},
    {"SCEMA_MD_PPPM_FFT", "hipFFT for every PPPM grid (default: grids of up to 2 900 points are solved in LDS)"},
    {"SCEMA_MD_FUSED_TAIL", "0 / 1: force assembly + SHAKE + second kick as three kernels / as k_finish (default: by batch size)"},
    {"SCEMA_MD_BONDED_SIDE", "0: the bonded kernel of a small batch always on the main stream behind the pair kernel (default: behind the PPPM chain on the side stream on steps without a new influence function)"},
    {"SCEMA_MD_BONDED_SIDE_MIN", "smallest batch whose bonded kernel follows the PPPM chain on the side stream (default 8: below, that chain is the longer one)"},
    {"SCEMA_MD_SMALL_BATCH_MAX", "largest batch that takes the cell grid with the most cells instead of the largest cells (default: launch groups of up to 31 replicas that run whole, i.e. not as part batches)"},
    {"SCEMA_MD_REBUILD_TOGETHER", "0 / 1: every replica rebuilds its neighbour rows on its own trigger / the replicas of a launch rebuild together as soon as one asks for it (default: together in launches of fewer than 128 replicas)"},
    {"SCEMA_MD_CELL_BUILD", "0: cell binning as k_bin + k_cell_scan + k_cell_fill instead of the one-launch k_cell_build"},
    {"SCEMA_MD_POLY_TOL", "fit target of the real-space Ewald polynomial (default 2e-13); parity tolerances assume the default"},
    {"SCEMA_REAX_DROP_DSBO2", "ReaxFF valence-angle gradient without the dSBO2 term, as USER-REAXC is believed to compute it"},
    {"SCEMA_MD_RX_ITEM_LDS", "0 (test hook): the ReaxFF angle and torsion items add forces and dE/dDelta to the work set with device-wide atomics instead of through a workgroup's LDS tables (replicas too large for the tables do)"},
    {"SCEMA_MD_RX_NEAR_FULL", "1 (test hook): the ReaxFF near rows hold every pair within the 5 A bond cutoff + skin instead of the pairs within reach of their type pair's bond order"},
    {"SCEMA_REAX_SKIN", "ReaxFF list skin in Angstrom"},
    {"SCEMA_REAX_HALVES", "number of part batches a ReaxFF batch runs as, each on its own stream (default 2; 0 or 1: one sequence of launches)"},
    {"SCEMA_REAX_OVERLAP", "0: the bond-order chain of the ReaxFF force stage on the same stream as the charge chain instead of next to it"},
    {"SCEMA_REAX_QEQ_PRECOND", "0: the conjugate gradients of the charge equilibration with the Jacobi preconditioner of fix qeq/reax instead of the bonded-pattern approximate inverse"},
    {"SCEMA_REAX_QEQ_SYM", "0: the matrix of the charge equilibration as full rows (every pair in both rows) instead of each pair once in its owner's row"},
    {"SCEMA_REAX_QEQ_ZLDS", "0: the matrix sweep of the charge equilibration gathers through the caches instead of from an LDS copy"},
    {"SCEMA_REAX_QEQ_LAUNCH", "conjugate-gradient iterations issued as launches per charge solve (default: adaptive)"},
    // test hooks: force rarely-taken paths
    {"SCEMA_MD_NEIGH_GROW0", "start with undersized neighbour capacities: overflow -> restore -> regrow"},
    {"SCEMA_MD_NEIGH_EXACT", "1: every list build tests its candidates in FP64 at the exact list radius; 0: none does (default: the first build of a run)"},
    {"SCEMA_MD_TEST_FAIL_INCOMING", "rank on which the allocation of a state that migrates in fails (-1: on whichever rank receives one); the failure must reach every rank through the handshake"},
    {"SCEMA_MD_TEST_FAIL_MIGRATE", "<what>[:<rank>] -- an injected failure of the exchange of replica states: dbox / upload (before the handshake: every rank ends the call, nothing is posted), enqueue / group / hostcopy (after it: the exchange is completed, the error travels in the status word of the stress all-gather)"},
    {"SCEMA_MD_TEST_SELF_MOVE", "1 (RCCL, one rank): every simulation that continues from a state held here receives it through ncclSend / ncclRecv from this very rank"},
    {"SCEMA_MD_TEST_QEQ_PRECOND_FAILS", "1: a ReaxFF run whose charge solve uses the approximate-inverse preconditioner reports that it did not converge (the evaluation must come back from its one retry with the Jacobi preconditioner)"},
    {"SCEMA_MD_QCAP16", "capacity of k_neigh_build's group lists in sixteenths of the table: small values force the whole-table walk"},
    {"SCEMA_MD_RX_COL32", "32-bit column indices in the ReaxFF charge matrix whatever the replica size"},
    {"SCEMA_MD_RX_NB_ONCE", "0: the both-ends ReaxFF non-bonded kernel"},
    {"SCEMA_MD_RX_ITEMCAP", "capacity of the work-item lists of the ReaxFF angle and torsion kernels: small values force their in-place path"},
    // diagnostics: print, never change a result
    {"SCEMA_MD_TIMING", "per-chunk wall times and the PAIR_TIMING / PAIR_COUNT counters on stderr"},
};
}  // namespace

const char *scema_env(const char *name) {
  for (const EnvSwitch &s : k_env)
    if (strcmp(s.name, name) == 0) return getenv(name);
  fprintf(stderr, "[scema_md] environment switch %s is not declared in engine_core.cpp\n", name);
  abort();
}

extern "C" {

/* the declared switches that are set in this process's environment, as "NAME=value" separated by newlines; returns the
 * number of switches set (the text is truncated to cap - 1 characters) */
int scema_md_env_overrides(char *buf, int cap) {
  std::string out;
  int n = 0;
  for (const EnvSwitch &s : k_env)
    if (const char *v = getenv(s.name)) { out += (n ? "\n" : ""); out += s.name; out += "="; out += v; n++; }
  if (buf && cap > 0) { strncpy(buf, out.c_str(), (size_t)cap - 1); buf[cap - 1] = 0; }
  return n;
}

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
  // (hipExtStreamCreateWithCUMask for the streams of the two half batches -- odd CUs for one, even CUs for the other, every XCD in both sets --
  // was measured in round 6, VERDICT r5 item 5: 439.3 / 441.9 against 441.7 / 441.9 evaluations/s at 72 replicas, 469.3 / 469.1 against 469.6 /
  // 469.6 at 576, profiles/r06_l_cumask_ragged_ab.log.  Neither half gains from not sharing CUs with the other; removed.)
  if (hipSetDevice(e->p.device) != hipSuccess || hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) {
    delete e;
    return SCEMA_MD_ERR_DEVICE;
  }
  if (const char *sp = scema_env("SCEMA_MD_SPLIT")) e->split_streams = atoi(sp) != 0;
  if (const char *sp = scema_env("SCEMA_MD_SPLIT_MIN")) e->split_min = std::max(2, atoi(sp));
  if (const char *sp = scema_env("SCEMA_MD_SPLIT_MAX")) e->split_max = std::max(0, atoi(sp));
  if (hipStreamCreateWithFlags(&e->stream3, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&e->ev_up, hipEventDisableTiming) != hipSuccess)
    e->stream3 = nullptr;   // an optimisation only
  if (const char *sx = scema_env("SCEMA_MD_SKIN_EXTRA")) e->skin_extra_fixed = std::max(-0.75 * e->p.skin, atof(sx));
  if (const char *sx = scema_env("SCEMA_MD_SKIN_ADAPT")) e->skin_adapt = atoi(sx) != 0;
  if (!scema_env("SCEMA_MD_ONE_STREAM")) {
    if (hipStreamCreateWithFlags(&e->stream2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming) != hipSuccess)
      e->stream2 = nullptr;   // side stream is an optimisation only
  }
  if (const char *sx = scema_env("SCEMA_REAX_HALVES")) e->rx_halves = std::max(0, std::min(8, atoi(sx)));
  if (const char *sx = scema_env("SCEMA_REAX_OVERLAP")) e->rx_overlap = atoi(sx) != 0;
  if (const char *sx = scema_env("SCEMA_REAX_QEQ_PRECOND")) e->rx_precond = atoi(sx) != 0;
  if (const char *sx = scema_env("SCEMA_REAX_QEQ_SYM")) e->rx_sym = atoi(sx) != 0;
  if (e->stream2 && e->stream3) {   // (fourth and last stream of an engine: see engine.h)
    bool ok = hipStreamCreateWithFlags(&e->rx_side1, hipStreamNonBlocking) == hipSuccess;
    for (int k = 0; k < 4 && ok; k++) ok = hipEventCreateWithFlags(&e->rx_side1_ev[k], hipEventDisableTiming) == hipSuccess;
    if (!ok && e->rx_side1) { (void)hipStreamDestroy(e->rx_side1); e->rx_side1 = nullptr; }
  }
  // test hook: start with undersized neighbour capacities, so that the overflow -> restore -> regrow path runs
  if (const char *g0 = scema_env("SCEMA_MD_NEIGH_GROW0")) e->neigh_grow = e->jtab_grow = std::max(0.05, atof(g0));
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
  if (e->rx_fork) (void)hipEventDestroy(e->rx_fork);
  if (e->rx_side1) (void)hipStreamDestroy(e->rx_side1);
  for (int k = 0; k < 4; k++) if (e->rx_side1_ev[k]) (void)hipEventDestroy(e->rx_side1_ev[k]);
  for (hipEvent_t pe : e->md_part_done) (void)hipEventDestroy(pe);
  for (auto &pt : e->rx_parts) {
    if (pt.main) (void)hipStreamDestroy(pt.main);
    if (pt.side) (void)hipStreamDestroy(pt.side);
    for (int k = 0; k < 4; k++) if (pt.ev[k]) (void)hipEventDestroy(pt.ev[k]);
  }
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
  if (e) (void)settle_pending(e, false);   // an update that waits for its verdict (no communicator) stands once the engine is used for something else
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
  out->rx_sweep_symmetric = e->prof.rx_sweep_symmetric;
  out->pair_union_ms = e->prof.pair_union_ms;
  out->rx_sweep_union_ms = e->prof.rx_sweep_union_ms;
  if (reset) e->prof = Profile();
  return SCEMA_MD_OK;
}

}  // extern "C"
