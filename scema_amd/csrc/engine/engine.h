// engine.h -- internal header of the batch engine behind include/scema_md.h: the types every part of the engine shares and the
// functions that cross its source files.  Not part of the C ABI; nothing outside scema_amd/csrc/engine/ includes it.
//
//   engine_core.cpp    create / destroy / register_replica / profile
//   engine_topo.cpp    topology preprocessing of a registered replica (exclusions, SHAKE clusters, bonded tiles)
//   engine_kspace.cpp  what a LAMMPS `run` sets up on the host: g_ewald, k-vectors, PPPM grid, the real-space polynomial, fix deform's box path
//   engine_run.cpp     one run of a batch with the OPLS force stage (run_phase), slots
//   engine_reax.cpp    the same with the ReaxFF force stage (run_phase_reax) and the ReaxFF entry points
//   engine_batch.cpp   the hot path: scema_md_strain_batch (request checks, plan, chunks, backups, results)
//   engine_comm.cpp    communicator, handshake, state migration, the stress all-gather, the planner's C face
//   engine_state.cpp   state store: branch rule, replica / state files
//   engine_equil.cpp   init_material: equilibration schedule, homogenisation, stiffness
//   engine_debug.cpp   parity / measurement entry points
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <thread>
#include <cstring>
#include <ctime>
#include <array>
#include <atomic>
#include <map>
#include <memory>
#include <sstream>
#include <fstream>
#include <string>
#include <vector>

#include "../../../include/scema_md.h"
#include "host/reax_ffield.h"
#include "host/sim_plan.h"
#include <hipfft/hipfft.h>

#include "md_kernels.h"
#include "md_equil.h"
#include "md_pppm.h"
#include "md_reax.h"
#include "md_types.h"

namespace scema_eng {

struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  hipError_t ensure(size_t n) {
    if (n <= bytes && p) return hipSuccess;
    release();
    if (n == 0) n = 8;
    hipError_t e = hipMalloc(&p, n);
    if (e == hipSuccess) bytes = n;
    return e;
  }
  template <class T>
  T *as() const { return reinterpret_cast<T *>(p); }
};

struct HostBox {
  double lo[3], h[6], hinv[6], vol;
};
inline void box_derive(const double *b, HostBox &o) {
  for (int d = 0; d < 3; d++) o.lo[d] = b[d];
  o.h[0] = b[3] - b[0]; o.h[1] = b[4] - b[1]; o.h[2] = b[5] - b[2];
  o.h[3] = b[8]; o.h[4] = b[7]; o.h[5] = b[6];
  o.hinv[0] = 1.0 / o.h[0]; o.hinv[1] = 1.0 / o.h[1]; o.hinv[2] = 1.0 / o.h[2];
  o.hinv[3] = -o.h[3] / (o.h[1] * o.h[2]);
  o.hinv[4] = (o.h[3] * o.h[5] - o.h[1] * o.h[4]) / (o.h[0] * o.h[1] * o.h[2]);
  o.hinv[5] = -o.h[5] / (o.h[0] * o.h[1]);
  o.vol = o.h[0] * o.h[1] * o.h[2];
}
inline void perp_widths(const HostBox &b, double w[3]) {
  w[0] = 1.0 / std::sqrt(b.hinv[0] * b.hinv[0] + b.hinv[5] * b.hinv[5] + b.hinv[4] * b.hinv[4]);
  w[1] = 1.0 / std::sqrt(b.hinv[1] * b.hinv[1] + b.hinv[3] * b.hinv[3]);
  w[2] = 1.0 / std::fabs(b.hinv[2]);
}

// -------------------------------------------------------------------------------------------
// Topology of one (material, replica): immutable, shared by every quadrature point that uses it
// -------------------------------------------------------------------------------------------
// the replica as it was registered (init.<mat>_<rep>.bin), kept so that a state can be written back in LAMMPS' own
// restart layout (last.<qp>.* / lcts.<qp>.*, stmd_problem.h:258,268)
struct SysCopy {
  scema_md_system sys;
  std::vector<int32_t> type, ba, bt, aa, at, da, dt, ia, it;
  std::vector<double> q, mass, eps, sig, bc, ac, dc, ic;
  void take(const scema_md_system &s) {
    sys = s;
    const size_t n = (size_t)s.natoms, nt = (size_t)s.ntypes;
    type.assign(s.type, s.type + n); q.assign(s.charge, s.charge + n); mass.assign(s.mass, s.mass + nt);
    eps.assign(s.eps, s.eps + nt * nt); sig.assign(s.sigma, s.sigma + nt * nt);
    ba.assign(s.bond_atoms, s.bond_atoms + 2 * (size_t)s.nbonds); bt.assign(s.bond_type, s.bond_type + s.nbonds); bc.assign(s.bond_coeff, s.bond_coeff + 2 * (size_t)s.nbondtypes);
    aa.assign(s.angle_atoms, s.angle_atoms + 3 * (size_t)s.nangles); at.assign(s.angle_type, s.angle_type + s.nangles); ac.assign(s.angle_coeff, s.angle_coeff + 2 * (size_t)s.nangletypes);
    da.assign(s.dihedral_atoms, s.dihedral_atoms + 4 * (size_t)s.ndihedrals); dt.assign(s.dihedral_type, s.dihedral_type + s.ndihedrals); dc.assign(s.dihedral_coeff, s.dihedral_coeff + 4 * (size_t)s.ndihedraltypes);
    ia.assign(s.improper_atoms, s.improper_atoms + 4 * (size_t)s.nimpropers); it.assign(s.improper_type, s.improper_type + s.nimpropers); ic.assign(s.improper_coeff, s.improper_coeff + 2 * (size_t)s.nimpropertypes);
    sys.type = type.data(); sys.charge = q.data(); sys.mass = mass.data(); sys.eps = eps.data(); sys.sigma = sig.data();
    sys.bond_atoms = ba.data(); sys.bond_type = bt.data(); sys.bond_coeff = bc.data();
    sys.angle_atoms = aa.data(); sys.angle_type = at.data(); sys.angle_coeff = ac.data();
    sys.dihedral_atoms = da.data(); sys.dihedral_type = dt.data(); sys.dihedral_coeff = dc.data();
    sys.improper_atoms = ia.data(); sys.improper_type = it.data(); sys.improper_coeff = ic.data();
    sys.x = nullptr; sys.v = nullptr;
  }
};

// Identity of a State / Topo object for the list signatures (ListSig): a number no other object of the process ever gets.  (Until round 6 the
// signatures held raw pointers: states are erased and re-created every update -- migration, branching, undo -- and a new object can land
// on a freed address, where the device's displacement test was the only guard left: ADVICE r5.)
inline unsigned long long next_object_id() { static std::atomic<unsigned long long> n{0}; return ++n; }
struct Topo {
  const unsigned long long id = next_object_id();
  int natoms = 0, ntypes = 0;
  SysCopy original;
  std::vector<int> type;
  std::vector<double> q, mass_atom, lj;
  int nbonds = 0, nbonds_noshake = 0, nangles = 0, ndihedrals = 0, nimpropers = 0, nspecial = 0, nclus = 0, ncons = 0, nfree = 0;
  double qsqsum = 0, qsum = 0, excl_cut = 0;
  double init_box[9];
  std::vector<double> init_x, init_v;
  // device copies
  DevBuf d_rtype;          // ReaxFF: force-field type per atom (element of the LAMMPS type), valid for rtype_stamp
  int rtype_stamp = -1;
  DevBuf d_type, d_q, d_mass, d_lj, d_bt_terms, d_bt_coef, d_ex_start, d_ex_list, d_clus_at, d_clus_n, d_clus_d, d_free_at, d_bt_desc, d_bt_atoms, d_bt_rank;
  int bt_ntile = 0, bt_maxloc = 1, bt_maxchunk = 1, bt_ncoef = 0, bt_cf_off[4] = {0, 0, 0, 0};
  double sp_w[6] = {0, 0, 0, 0, 0, 0};   // special_bonds weights: lj 1-2, 1-3, 1-4, coul 1-2, 1-3, 1-4
};

struct State {
  const unsigned long long id = next_object_id();
  Topo *topo = nullptr;
  double box[9];
  DevBuf x, v;
  // Performance only (results do not depend on the skin): extra list skin for this state, chosen from how often its last
  // sampling run had to rebuild the list.  A freshly built crystal rebuilds every ~30 steps and is fastest with the
  // reference's 2.0 A; once thermalised it rebuilds every ~14 steps and 0.5 A more (one rebuild in ~22 steps, the far
  // band of the rows mostly skipped) is 6 % faster.
  double skin_extra = 0.0;
  // ReaxFF, performance only (the solver's tolerance fixes the charges, not its starting point): the last solutions of the charge
  // equilibration of this state's latest run, [4][npad] s then [3][npad] t, newest first -- the next run of the state extrapolates
  // its first guesses from them instead of starting from zeros (RxView::warm)
  DevBuf qhist;
  bool qhist_valid = false;
};

// work arrays of the ReaxFF path for one batch position (reax/rx_types.h RxView points into these)
struct RxSlot {
  int cap_pad = 0, cap_nb = 0, cap_bd = 0, cap_nbn = 0;
  size_t cap_col = 0;
  DevBuf nbn_cnt, nbn, nbnT, qpart, pm_len, pm_col, pm_raw, pm_val;
  DevBuf nb_cnt, nb_own0, nbT, hval, hcol, hlen, hown, hownlen, bd_cnt, bd, bd_rev, bd_bop, bd_c, bd_bo, bd_g, bd_cb, deltap, total_bo, cd_delta, hd, q, s, t, s_hist, t_hist, qwork, misc;
};

// what the neighbour rows of a slot were built for: a run that follows on the same slot keeps them if all of it still holds
struct ListSig {
  bool valid = false;
  unsigned long long topo = 0;    // Topo::id the rows were built for
  unsigned long long state = 0;   // State::id the rows were last built for (a hint: whether they still hold is decided on the device)
  int nc[3] = {0, 0, 0}, capj = 0, maxneigh = 0, npad = 0;
  double rlist = 0.0, cut_lj = 0.0, cut_coul = 0.0;
  // the list's scalars at the end of the run that left it: what the displacement test of the next run needs, and the statistics
  double corners_hold[24];
  int ago = 0, maxj_seen = 0;
  unsigned long long nentries = 0, nentries_ref = 0, nrowent = 0;
  long long rx_stamp = 0;        // != 0: the rows are ReaxFF rows (RxSlot), built under this force-field stamp; capj holds the near rows' stride
  int rx_mimg[3] = {0, 0, 0};
};
struct Slot {
  std::unique_ptr<RxSlot> rx;
  ListSig sig;
  int cap_atoms = 0, cap_pad = 0, cap_neigh = 0, cap_cells = 0, cap_k = 0;
  size_t cap_jtab = 0;
  DevBuf virp, virb, fb, fs, slot_of, tile_nj, tile_jtab, tile_order, tile_wstart;
    DevBuf f, xq, stype, perm, slot_tmp, wrapn, xhold, cell_of, ckey, cell_count, cell_start, cell_fill, numneigh, neigh, sfac, kvec,
      xbak, vbak;
};

struct ActiveSim {
  State *st = nullptr;
  int user_index = -1;  // index into the caller's sims[]
  int nsteps = 0;
  double rates[6] = {0, 0, 0, 0, 0, 0};
  double dt = 0, temperature = 0;
  int nts = 0, nss = 0;
  double pavg[6];
  double box0[9], skin0 = 0.0;   // box and list skin of the state before the update (a failed update puts them back)
};

// A state object that an update stored under a key, with what it displaced (undone if the update fails)
struct CreatedState { std::string key; std::unique_ptr<State> displaced; };
// An update that ran on this rank WITHOUT a communicator in a world of several ranks: whether it stands is only known once the caller's
// collective has shown every rank's status word (scema_md_scatter_gathered / scema_md_settle_update).  Until then the advanced states keep
// their backups and the owner directory stays uncommitted.
struct PendingUpdate {
  bool active = false;
  std::vector<ActiveSim> act;         // the simulations whose states were advanced (backups at pool offsets 0 .. act.size() - 1)
  std::vector<CreatedState> created;
  scema::SimPlan plan;
  std::vector<std::string> dst_keys;
  int rank = 0;
};

// One process per GPU: the communicator of the engine.  RCCL (xGMI) for the GPU box, or transport callbacks of the host
// program (MPI in SCEMa, gloo in the CPU tests).  Replaces the MPI calls of stmd_sync.h:620-726.
struct Comm {
  int kind = 0;   // 0 none, 1 RCCL, 2 host callbacks
  int rank = 0, world = 1;
  ncclComm_t nccl = nullptr;
  scema_md_host_allgather_fn ag = nullptr;
  scema_md_host_send_fn send = nullptr;
  scema_md_host_recv_fn recv = nullptr;
  void *ctx = nullptr;
  DevBuf d_gather, d_box, d_word;
  std::vector<double> h_gather;
  // the exchange of states of the current update, resolved before the handshake (prepare_incoming): per move of the plan the source
  // state this rank sends / the state it receives into (else null), and the boxes that travel beside x and v
  std::vector<State *> mig_src, mig_dst;
  std::vector<double> h_box;
  long long migrations = 0, allgathers = 0, handshakes = 0;
};

// union of the intervals [ev[2l], ev[2l+1]), l < n, on the device clock (ms); events of several streams
double event_union_ms(const std::vector<hipEvent_t> &ev, size_t n);
struct Profile {
  long long pair_launches = 0;
  double pair_ms = 0, pair_alg_bytes = 0;
  long long pair_sims = 0;   // simulations summed over the timed pair launches
  double pair_union_ms = 0.0, rx_sweep_union_ms = 0.0;   // time with at least one timed launch in flight (launches of two half batches overlap)
  long long box_flips = 0;   // triclinic box flips applied (fix deform, flip yes)
  long long md_steps = 0, neigh_builds = 0, evals = 0;
  double unique_pairs_sum = 0;
  long long unique_pairs_n = 0;
  double skin_sum = 0;
  // ReaxFF: launches of k_rx_qeq_sweep
  long long rx_sweep_launches = 0;
  double rx_sweep_ms = 0, rx_sweep_entries = 0, rx_sweep_rows = 0;
  long long rx_sweep_col_bytes = 0, rx_sweep_symmetric = 0;
};

struct EwaldSetup {
  double g = 0.0;
  std::vector<int> kn;
  std::vector<int> krun;   // per k: length of the run of following k-vectors that continue its row (n3 + 1 each)
  std::vector<int> kgrp;   // groups of k-vectors (n1, +-n2, +-n3): n1, |n2|, |n3|, k index of (+,+), (-,+), (+,-), (-,-) or -1, pad
  int kmaxd[3] = {0, 0, 0};
};

struct FlipEvent {
  int step;          // the flip is detected at the end of this step and applied at the start of the next one
  double tilt[3];    // xy, xz, yz after the flip
  int nflip[3];      // lattice steps f_xy, f_xz, f_yz: a2' = a2 + f_xy a1, a3' = a3 + f_yz a2 + f_xz a1
};

struct RunSpec {
  int nvt = 1, use_shake = 1, deform = 0, sample = 0, ev_always = 0;
  int static_only = 0;  // parity hook: forces of the potential only (no constraint forces)
  // equilibration schedule of init_material (md_equil.hip): fix nvt / fix npt ... iso with a temperature ramp, issued in
  // segments (the cell grid and the k-space tables of a segment hold for box lengths within +-box_margin), and min_style sd
  int nh = 0, npt = 0, keep = 0, nh_total = 0, lavg_nav = 0;
  double t_start = 0, t_stop = 0, p_target = 1.0, p_period = 1000.0, box_margin = 0.0;
  std::vector<EwaldSetup> *ew_keep = nullptr;   // k-space setup of the run's first segment, reused by the later ones
  int minimize = 0, min_maxiter = 0, min_maxeval = 0;
  double min_etol = 0, min_ftol = 0;
  int keep_list = 0;      // 1: this run follows another one of the same simulations on the same slots (phase B after phase A): the neighbour rows on the
                          // device stand where the cell grid of the new run can be the old one (SimDev::keep_list).  2: the slots may still hold the rows
                          // of these states from the update before: kept where the device finds every atom within half the skin of the positions
                          // the rows were built for (k_keep_validate)
  int qeq_continue = 0;   // ReaxFF: this run follows another one of the same simulations on the same slots (phase B after phase A): the
                          // charge-equilibration history is still in place
};

// which phases an evaluation runs and with which constraints (the hot path: strain with SHAKE, sample with SHAKE;
// the elastic-constant runs of init_material: both without SHAKE; its homogenisation run: sampling only)
struct EvalOpt {
  bool phase_a = true;
  int shake_a = 1, shake_b = 1;
};

}  // namespace scema_eng
using namespace scema_eng;

struct scema_md_engine {
  scema_md_params p;
  hipStream_t stream = nullptr;
  double skin_extra_fixed = 0.0;          // SCEMA_MD_SKIN_EXTRA: list skin = params.skin + this (performance only; may be negative)
  bool skin_adapt = false;                // SCEMA_MD_SKIN_ADAPT=1: per-state adaptation from the rebuild interval (round-1 behaviour)
  hipStream_t stream2 = nullptr;          // side stream: structure factors next to the bonded kernel
  hipStream_t stream3 = nullptr;          // second half batch of a large launch group (run_phase)
  hipEvent_t ev_up = nullptr;
  bool split_streams = true;              // SCEMA_MD_SPLIT=0 switches the two-half pipeline off
  int split_min = 9, split_max = 1 << 30;  // launch groups from this size on are split (SCEMA_MD_SPLIT_MAX puts an upper end back: round 2 measured 336 evals/s either way
                                         // at 576 and left large groups whole; with round 4's kernels the halves give 437 against 429, profiles/r04_zs_*)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool rx_qeq_failed = false;             // the last ReaxFF run ended with a charge solve that did not converge (eval_chunk's one retry with the Jacobi preconditioner)
  long long rx_precond_fallbacks = 0;     // evaluations that were repeated that way
  bool rx_precond = true;                 // bonded-pattern sparse approximate inverse as the preconditioner of the charge equilibration (SCEMA_REAX_QEQ_PRECOND=0: the reference's Jacobi one)
  int rx_halves = 2, rx_overlap = 1;      // scema_md_reax_concurrency: part batches (1 = one sequence of launches) and side streams (initial values from SCEMA_REAX_HALVES / SCEMA_REAX_OVERLAP)
  // ReaxFF runs as rx_halves part batches on as many streams (part 0: stream + stream2), each with a side stream for its bond-order chain and
  // four events (fork / mid / join of the side stream, the part's join with the main stream); created on first use
  struct RxPart { hipStream_t main = nullptr, side = nullptr; hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr}; };
  std::vector<RxPart> rx_parts;
  std::vector<hipEvent_t> md_part_done;        // one event per part batch beyond the first: the end of its launches, which the main stream waits for
  hipStream_t rx_side1 = nullptr;         // side stream of the second part batch (its main stream is stream3): created with the engine, so that an engine has
                                          // exactly four streams in a fixed order of creation -- the runtime deals streams to its four hardware queues in that order,
                                          // and two of these four sharing a queue costs 7 % (bench: a second engine of a process drew such a deal)
  hipEvent_t rx_side1_ev[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t rx_fork = nullptr;           // main stream -> the parts' streams at the start of the step loop
  std::map<std::string, std::unique_ptr<Topo>> topos;
  std::map<std::string, std::unique_ptr<State>> states;
  std::vector<std::unique_ptr<Slot>> slots;
  DevBuf d_sims, d_sc, d_local_stress, d_kpack, d_minptr, d_boxpair, d_pppm, d_copytab;
  std::vector<MdkCopy> h_copytab;
  DevBuf d_zerotab, d_rxstat;            // zero-fill descriptors; the solver statistics of a ReaxFF batch, gathered for one read-back
  std::vector<MdkZero> h_zerotab;
  std::vector<long long> h_rxstat;
  // x and v of every state an update advances, as they were before it: the retry after a list overflow restarts from
  // them, and a failed update (on this rank or on another) puts them back
  std::vector<std::unique_ptr<DevBuf>> bak_x, bak_v;
  std::map<std::array<int, 6>, hipfftHandle> pppm_plans;   // (nx, ny, nz, batch, stream, distance between grids) -> batched 3-d Z2Z plan
  std::vector<int> h_kpack;   // host copy, alive until the stream has consumed the upload
  int local_stress_count = 0;
  std::vector<SimDev> h_sims;
  std::vector<SimScalars> h_sc;
  std::vector<hipEvent_t> ev_pool;
  Profile prof;
  std::string err;
  double neigh_grow = 1.0;   // headroom factor of the cluster rows, x1.5 per overflow
  double jtab_grow = 1.0;    // headroom factor of the tile j tables, x1.25 per overflow (-> smaller cells)
  int overflow_bits = 0;     // what overflowed in the last run: 4 = a tile's j table, 8 = a cluster row
  double overflow_need_j = 1.0, overflow_need_row = 1.0;   // ... and the largest demand / capacity the run saw (the retry grows by at least that)
  long long unsettled_updates = 0;   // updates of a world > 1 without a communicator that the next call found unsettled and let stand
  // ReaxFF path (force_field "reax"): the force-field tables, settings of fix qeq/reax, list skin
  bool rx_ready = false, reax_active = false;
  int rx_stamp = 0;                    // bumped by every scema_md_reax_configure
  RxParams rx_host;
  std::vector<int> rx_type_map;        // LAMMPS type - 1 -> force-field type
  DevBuf d_rxparams, d_rxviews;
  std::vector<RxView> h_rxviews;
  double rx_skin = 0.75, rx_qeq_tol = 1e-6;   // (list skin in A, performance only.  1.0 until the wave-per-row list build of round 5 made a rebuild a fifth as dear:
                                              // 0.3: 1 177, 0.5: 1 191, 0.75: 1 193, 1.0: 1 177 evaluations/s on the 72-replica set, same box)
  int rx_qeq_maxiter = 200, rx_terms = 31;
  bool rx_sym = true;                  // the symmetric form of the charge solve where a batch allows it (SCEMA_REAX_QEQ_SYM=0: full rows)
  long long rx_qeq_iters = 0, rx_qeq_solves = 0, rx_qeq_slow = 0;
  int rx_qeq_launch_cold = 48;        // the same for the first solves of a run (empty history)
  int rx_qeq_launch = 32;             // conjugate-gradient iterations issued as batch launches per solve (follows what the last run needed)
  bool rx_qeq_launch_pinned = false;  // SCEMA_REAX_QEQ_LAUNCH fixes it (0: every solve runs in the single-workgroup loop)
  Comm comm;
  scema::OwnerDirectory dir;   // state key -> owning rank, identical on every rank (host/sim_plan.h)
  scema::SimPlan last_plan;
  PendingUpdate pending;
};

namespace scema_eng {

int fail(scema_md_engine *e, int code, const char *fmt, ...);

#define HIPCHK(call)                                                                                     \
  do {                                                                                                   \
    hipError_t _e = (call);                                                                              \
    if (_e != hipSuccess) return fail(e, SCEMA_MD_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(_e)); \
  } while (0)
#define NCCLCHK(call)                                                                                         \
  do {                                                                                                    \
    ncclResult_t _r = (call);                                                                             \
    if (_r != ncclSuccess) return fail(e, SCEMA_MD_ERR_DEVICE, "%s failed: %s", #call, ncclGetErrorString(_r)); \
  } while (0)

std::string topo_key(const char *matid, int replica);
std::string state_key(int qp, const char *matid, int replica);

template <class T>
int upload(scema_md_engine *e, DevBuf &b, const std::vector<T> &v) {
  HIPCHK(b.ensure(v.size() * sizeof(T)));
  if (!v.empty()) HIPCHK(hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return SCEMA_MD_OK;
}

// engine_topo.cpp
int build_topo(scema_md_engine *e, const scema_md_system *s, Topo &t);
// engine_kspace.cpp
void ewald_setup(const scema_md_params &p, const Topo &t, const double *box, EwaldSetup &out, bool g_only = false);
void ewald_tables(EwaldSetup &out);
void pppm_setup_host(const scema_md_params &p, const Topo &t, const double *box, double &g, int pg[3]);
double cached_coul_poly(scema_md_engine *, double g, double rc, double *poly, int *npoly, double *uscale);
void tilt_closest(double tilt[3], double xprd_new, double yprd_new, double xy, double xz, double yz, double xprd, double yprd);
int tilt_flip(const double tilt[3], double xprd, double yprd, double flipped[3], int nflip[3]);
bool deform_trajectory(const double *box0, const double *rates, double dt, int nsteps, double *box_end, std::vector<FlipEvent> &events,
                       std::vector<HostBox> &extremes);
double wall_s();
double round_trip(const char *fmt, double v);
// engine_run.cpp / engine_reax.cpp
int settle_pending(scema_md_engine *e, bool failed);
int ensure_slot(scema_md_engine *e, Slot &sl, int natoms, int maxneigh, int ncells, int nk, int capj);
int run_phase(scema_md_engine *e, std::vector<ActiveSim> &sims, const RunSpec &spec);
int run_phase_reax(scema_md_engine *e, std::vector<ActiveSim> &sims, const RunSpec &spec);
int prepare_slots(scema_md_engine *e, std::vector<ActiveSim> &sims);
int reupload_scalars(scema_md_engine *e, int ns);
// engine_state.cpp
State *find_state(scema_md_engine *e, int qp, const char *matid, int replica);
Topo *find_topo(scema_md_engine *e, const char *matid, int replica);
int make_state(scema_md_engine *e, Topo *t, const double *box, const double *x, const double *v, bool from_device, std::unique_ptr<State> &out);
int make_empty_state(scema_md_engine *e, Topo *t, std::unique_ptr<State> &out);
int resolve_state(scema_md_engine *e, const scema_mdsim &m, State **out, std::unique_ptr<State> *incoming = nullptr, bool *created = nullptr,
                  std::unique_ptr<State> *displaced = nullptr);
// engine_batch.cpp
int eval_chunk(scema_md_engine *e, std::vector<ActiveSim> &chunk, const EvalOpt &opt = EvalOpt(), size_t pool_off = 0);
// engine_comm.cpp
int prepare_incoming(scema_md_engine *e, const scema_mdsim *sims, const scema::SimPlan &plan, const std::vector<std::string> &src_keys,
                     std::map<int, std::unique_ptr<State>> &incoming);
int migrate_states(scema_md_engine *e, const scema::SimPlan &plan);
double plan_hash(const scema::SimPlan &P, const std::vector<double> &cost);
int check_gathered_trailers(scema_md_engine *e, const double *gathered, size_t stride, size_t off, int world, int rank, const char *when);
int handshake(scema_md_engine *e, int local_status, double hash);
int allgather_stresses(scema_md_engine *e, const std::vector<double> &local, scema_mdsim *sims, int n_sims);
// engine_debug.cpp
int debug_state(scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica, State **out, std::unique_ptr<State> &tmp);

}  // namespace scema_eng
