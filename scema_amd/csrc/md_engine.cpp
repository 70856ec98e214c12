// md_engine.cpp -- batch engine behind the C ABI of include/scema_md.h.
//
// Replaces STMDProblem<3>::lammps_straining (reference headers/stmd_problem.h:84-383): instead of
// two LAMMPS lifetimes and three restart files per quadrature-point replica, every replica state
// (x, v, box) stays resident in HBM keyed by (qp_id, matid, replica); a whole vector of MDSim
// requests is advanced in lockstep by the kernels of md_kernels.hip, one launch per stage for the
// whole batch, with no host synchronisation inside a run.
//
// Host-side arithmetic restated here, with the reference line it follows:
//   lbdim / strain correction ........ stmd_problem.h:210-225
//   nts rule ......................... stmd_problem.h:229-232
//   "%f" dts/tempt, "%.6e" rates ..... stmd_problem.h:164,235,241
//   state branch rule ................ stmd_problem.h:116-138,185-207
//   stress = -<P> * 1.01325e5 ........ stmd_problem.h:335-341
//   Hooke fallback ................... stmd_problem.h:386-392,479-483
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
#include <map>
#include <memory>
#include <sstream>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/scema_md.h"
#include "host/reax_ffield.h"
#include "host/sim_plan.h"
#include <hipfft/hipfft.h>

#include "md_kernels.h"
#include "md_equil.h"
#include "md_pppm.h"
#include "md_reax.h"
#include "md_types.h"

namespace {

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
static void box_derive(const double *b, HostBox &o) {
  for (int d = 0; d < 3; d++) o.lo[d] = b[d];
  o.h[0] = b[3] - b[0]; o.h[1] = b[4] - b[1]; o.h[2] = b[5] - b[2];
  o.h[3] = b[8]; o.h[4] = b[7]; o.h[5] = b[6];
  o.hinv[0] = 1.0 / o.h[0]; o.hinv[1] = 1.0 / o.h[1]; o.hinv[2] = 1.0 / o.h[2];
  o.hinv[3] = -o.h[3] / (o.h[1] * o.h[2]);
  o.hinv[4] = (o.h[3] * o.h[5] - o.h[1] * o.h[4]) / (o.h[0] * o.h[1] * o.h[2]);
  o.hinv[5] = -o.h[5] / (o.h[0] * o.h[1]);
  o.vol = o.h[0] * o.h[1] * o.h[2];
}
static void perp_widths(const HostBox &b, double w[3]) {
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

struct Topo {
  int natoms = 0, ntypes = 0;
  SysCopy original;
  std::vector<int> type;
  std::vector<double> q, mass_atom, lj;
  int nbonds = 0, nbonds_noshake = 0, nangles = 0, ndihedrals = 0, nimpropers = 0, nspecial = 0, nclus = 0, ncons = 0;
  double qsqsum = 0, qsum = 0, excl_cut = 0;
  double init_box[9];
  std::vector<double> init_x, init_v;
  // device copies
  DevBuf d_rtype;          // ReaxFF: force-field type per atom (element of the LAMMPS type), valid for rtype_stamp
  int rtype_stamp = -1;
  DevBuf d_type, d_q, d_mass, d_lj, d_bt_terms, d_bt_coef, d_ex_start, d_ex_list, d_clus_at, d_clus_n, d_clus_d, d_bt_desc, d_bt_atoms, d_bt_rank;
  int bt_ntile = 0, bt_maxloc = 1, bt_maxchunk = 1, bt_ncoef = 0, bt_cf_off[4] = {0, 0, 0, 0};
  double sp_w[6] = {0, 0, 0, 0, 0, 0};   // special_bonds weights: lj 1-2, 1-3, 1-4, coul 1-2, 1-3, 1-4
};

struct State {
  Topo *topo = nullptr;
  double box[9];
  DevBuf x, v;
  // Performance only (results do not depend on the skin): extra list skin for this state, chosen from how often its last
  // sampling run had to rebuild the list.  A freshly built crystal rebuilds every ~30 steps and is fastest with the
  // reference's 2.0 A; once thermalised it rebuilds every ~14 steps and 0.5 A more (one rebuild in ~22 steps, the far
  // band of the rows mostly skipped) is 6 % faster.
  double skin_extra = 0.0;
};

// work arrays of the ReaxFF path for one batch position (reax/rx_types.h RxView points into these)
struct RxSlot {
  int cap_pad = 0, cap_nb = 0, cap_bd = 0, cap_nbn = 0;
  DevBuf nbn_cnt, nbn, qpart;
  DevBuf nb_cnt, nb, hval, bd_cnt, bd, bd_rev, bd_bop, bd_c, bd_bo, bd_g, bd_cb, deltap, total_bo, cd_delta, hd, q, s, t, s_hist, t_hist, qwork, misc;
};

struct Slot {
  std::unique_ptr<RxSlot> rx;
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
  long long migrations = 0, allgathers = 0, handshakes = 0;
};

struct Profile {
  long long pair_launches = 0;
  double pair_ms = 0, pair_alg_bytes = 0;
  long long pair_sims = 0;   // simulations summed over the timed pair launches
  long long box_flips = 0;   // triclinic box flips applied (fix deform, flip yes)
  long long md_steps = 0, neigh_builds = 0, evals = 0;
  double unique_pairs_sum = 0;
  long long unique_pairs_n = 0;
  double skin_sum = 0;
  // ReaxFF: launches of k_rx_qeq_sweep
  long long rx_sweep_launches = 0;
  double rx_sweep_ms = 0, rx_sweep_entries = 0, rx_sweep_rows = 0;
};

}  // namespace

struct scema_md_engine {
  scema_md_params p;
  hipStream_t stream = nullptr;
  double skin_extra_fixed = 0.0;          // SCEMA_MD_SKIN_EXTRA: list skin = params.skin + this (performance only; may be negative)
  bool skin_adapt = false;                // SCEMA_MD_SKIN_ADAPT=1: per-state adaptation from the rebuild interval (round-1 behaviour)
  hipStream_t stream2 = nullptr;          // side stream: structure factors next to the bonded kernel
  hipStream_t stream3 = nullptr;          // second half batch of a large launch group (run_phase)
  hipEvent_t ev_up = nullptr;
  bool split_streams = true;              // SCEMA_MD_SPLIT=0 switches the two-half pipeline off
  int split_min = 32, split_max = 200;  // launch groups of this size range are split (larger ones gain nothing: measured 336 evals/s either way at 576)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  std::map<std::string, std::unique_ptr<Topo>> topos;
  std::map<std::string, std::unique_ptr<State>> states;
  std::vector<std::unique_ptr<Slot>> slots;
  DevBuf d_sims, d_sc, d_local_stress, d_kpack, d_minptr, d_boxpair, d_pppm, d_copytab;
  std::vector<MdkCopy> h_copytab;
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
  bool use_graphs = false;  // hipGraph replay of the MD step loop: opt-in (SCEMA_MD_GRAPH=1), measured slower on ROCm 7.2
  // ReaxFF path (force_field "reax"): the force-field tables, settings of fix qeq/reax, list skin
  bool rx_ready = false, reax_active = false;
  int rx_stamp = 0;                    // bumped by every scema_md_reax_configure
  RxParams rx_host;
  std::vector<int> rx_type_map;        // LAMMPS type - 1 -> force-field type
  DevBuf d_rxparams, d_rxviews;
  std::vector<RxView> h_rxviews;
  double rx_skin = 1.0, rx_qeq_tol = 1e-6;
  int rx_qeq_maxiter = 200, rx_terms = 31;
  long long rx_qeq_iters = 0, rx_qeq_solves = 0, rx_qeq_slow = 0;
  int rx_qeq_launch_cold = 48;        // the same for the first solves of a run (empty history)
  int rx_qeq_launch = 32;             // conjugate-gradient iterations issued as batch launches per solve (follows what the last run needed)
  bool rx_qeq_launch_pinned = false;  // SCEMA_REAX_QEQ_LAUNCH fixes it (0: every solve runs in the single-workgroup loop)
  Comm comm;
  scema::OwnerDirectory dir;   // state key -> owning rank, identical on every rank (host/sim_plan.h)
  scema::SimPlan last_plan;
};

namespace {

int fail(scema_md_engine *e, int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (e) e->err = buf;
  return code;
}

#define HIPCHK(call)                                                                                     \
  do {                                                                                                   \
    hipError_t _e = (call);                                                                              \
    if (_e != hipSuccess) return fail(e, SCEMA_MD_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(_e)); \
  } while (0)

std::string topo_key(const char *matid, int replica) { return std::string(matid ? matid : "") + "_" + std::to_string(replica); }
std::string state_key(int qp, const char *matid, int replica) { return std::to_string(qp) + "." + topo_key(matid, replica); }

template <class T>
int upload(scema_md_engine *e, DevBuf &b, const std::vector<T> &v) {
  HIPCHK(b.ensure(v.size() * sizeof(T)));
  if (!v.empty()) HIPCHK(hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return SCEMA_MD_OK;
}

// -------------------------------------------------------------------------------------------
// topology preprocessing
// -------------------------------------------------------------------------------------------
int build_topo(scema_md_engine *e, const scema_md_system *s, Topo &t) {
  const int n = s->natoms;
  if (n <= 0 || s->ntypes <= 0) return fail(e, SCEMA_MD_ERR_ARG, "natoms/ntypes out of range");
  if (n > MD_JMASK) return fail(e, SCEMA_MD_ERR_ARG, "too many atoms for the 27-bit neighbour index");
  t.natoms = n;
  // Lennard-Jones classes: atom types with identical eps/sigma rows are one class on the device (force-field
  // generators hand out an atom type per atom name; OPLS-AA has a dozen distinct Lennard-Jones sites).  Masses stay
  // per atom, charges are per atom anyway; the kernels index pair tables of ncls x ncls entries.
  const int nty = s->ntypes;
  std::vector<int> cls(nty, -1), rep;
  for (int u = 0; u < nty; u++) {
    for (size_t c = 0; c < rep.size() && cls[u] < 0; c++) {
      const int v = rep[c];
      bool same = true;
      for (int w = 0; w < nty && same; w++)
        same = s->eps[(size_t)u * nty + w] == s->eps[(size_t)v * nty + w] && s->sigma[(size_t)u * nty + w] == s->sigma[(size_t)v * nty + w] &&
               s->eps[(size_t)w * nty + u] == s->eps[(size_t)w * nty + v] && s->sigma[(size_t)w * nty + u] == s->sigma[(size_t)w * nty + v];
      if (same) cls[u] = (int)c;
    }
    if (cls[u] < 0) { cls[u] = (int)rep.size(); rep.push_back(u); }
  }
  const int ncls = (int)rep.size();
  if (ncls > MD_MAXTYPES) return fail(e, SCEMA_MD_ERR_ARG, "%d distinct Lennard-Jones types (of %d atom types): at most %d are supported", ncls, nty, MD_MAXTYPES);
  t.ntypes = ncls;
  t.type.resize(n);
  t.q.assign(s->charge, s->charge + n);
  t.mass_atom.resize(n);
  for (int i = 0; i < n; i++) {
    if (s->type[i] < 0 || s->type[i] >= nty) return fail(e, SCEMA_MD_ERR_ARG, "atom type out of range");
    t.type[i] = cls[s->type[i]];
    t.mass_atom[i] = s->mass[s->type[i]];
    t.qsqsum += t.q[i] * t.q[i];
    t.qsum += t.q[i];
  }
  const int nt2 = ncls * ncls;
  t.lj.resize(4 * nt2);
  for (int a = 0; a < ncls; a++)
    for (int b = 0; b < ncls; b++) {
      const size_t src = (size_t)rep[a] * nty + rep[b];
      const int k = a * ncls + b;
      const double s6 = std::pow(s->sigma[src], 6.0), s12 = s6 * s6;
      t.lj[k] = 48.0 * s->eps[src] * s12;
      t.lj[nt2 + k] = 24.0 * s->eps[src] * s6;
      t.lj[2 * nt2 + k] = 4.0 * s->eps[src] * s12;
      t.lj[3 * nt2 + k] = 4.0 * s->eps[src] * s6;
    }
  // ---- bond graph -> 1-2 / 1-3 / 1-4 partners (lowest level wins) ----
  std::vector<std::vector<int>> adj(n);
  double max_r0 = 0.0;
  for (int b = 0; b < s->nbonds; b++) {
    const int a = s->bond_atoms[2 * b], c = s->bond_atoms[2 * b + 1];
    if (a < 0 || a >= n || c < 0 || c >= n || a == c) return fail(e, SCEMA_MD_ERR_ARG, "bad bond %d", b);
    adj[a].push_back(c);
    adj[c].push_back(a);
    max_r0 = std::max(max_r0, s->bond_coeff[2 * s->bond_type[b] + 1]);
  }
  std::vector<int> sp_at;
  std::vector<double> sp_cf;
  std::vector<std::vector<int>> excl(n);
  std::vector<int> level(n, 0), frontier, next, touched;
  int max_excl_level = 0;
  for (int i = 0; i < n; i++) {
    level[i] = -1;
    touched.assign(1, i);
    frontier.assign(1, i);
    for (int lvl = 1; lvl <= 3; lvl++) {
      next.clear();
      for (int a : frontier)
        for (int c : adj[a])
          if (level[c] == 0) {
            level[c] = lvl;
            next.push_back(c);
            touched.push_back(c);
          }
      frontier.swap(next);
    }
    for (int c : touched) {
      if (c > i) {
        const int lvl = level[c];
        const double wl = s->special_lj[lvl - 1], wc = s->special_coul[lvl - 1];
        if (!(wl == 1.0 && wc == 1.0)) {
          sp_at.push_back(i);
          sp_at.push_back(c);
          sp_cf.push_back(wl);
          sp_cf.push_back(wc);
          excl[i].push_back(c);
          excl[c].push_back(i);
          max_excl_level = std::max(max_excl_level, lvl);
        }
      }
    }
    for (int c : touched) level[c] = 0;
  }
  t.nspecial = (int)sp_cf.size() / 2;
  std::vector<int> ex_start(n + 1, 0), ex_list;
  for (int i = 0; i < n; i++) {
    std::sort(excl[i].begin(), excl[i].end());
    ex_start[i + 1] = ex_start[i] + (int)excl[i].size();
    ex_list.insert(ex_list.end(), excl[i].begin(), excl[i].end());
  }
  // build-time exclusion gate: an excluded pair is at most max_excl_level bonds apart
  t.excl_cut = 1.5 * max_excl_level * max_r0;
  // ---- fix shake ... m <mass>: star clusters ----
  std::vector<char> shaken(s->nbonds, 0);
  std::vector<int> nsh(n, 0);
  if (e->p.shake_mass > 0.0)
    for (int b = 0; b < s->nbonds; b++) {
      const int a = s->bond_atoms[2 * b], c = s->bond_atoms[2 * b + 1];
      if (std::fabs(t.mass_atom[a] - e->p.shake_mass) <= 0.1 || std::fabs(t.mass_atom[c] - e->p.shake_mass) <= 0.1) {
        shaken[b] = 1;
        nsh[a]++;
        nsh[c]++;
      }
    }
  std::vector<int> cl_of(n, -1), clus_at, clus_n;
  std::vector<double> clus_d;
  for (int b = 0; b < s->nbonds; b++) {
    if (!shaken[b]) continue;
    const int a = s->bond_atoms[2 * b], c = s->bond_atoms[2 * b + 1];
    int cen, sat;
    if (nsh[a] > nsh[c] || (nsh[a] == nsh[c] && a < c)) { cen = a; sat = c; } else { cen = c; sat = a; }
    if (nsh[sat] != 1) return fail(e, SCEMA_MD_ERR_ARG, "SHAKE cluster is not star shaped (atom %d)", sat);
    int cl = cl_of[cen];
    if (cl < 0) {
      cl = (int)clus_n.size();
      cl_of[cen] = cl;
      clus_n.push_back(1);
      clus_at.insert(clus_at.end(), {cen, 0, 0, 0});
      clus_d.insert(clus_d.end(), {0.0, 0.0, 0.0});
    }
    const int k = clus_n[cl];
    if (k >= 4) return fail(e, SCEMA_MD_ERR_ARG, "SHAKE cluster of more than 4 atoms");
    clus_at[4 * cl + k] = sat;
    clus_d[3 * cl + (k - 1)] = s->bond_coeff[2 * s->bond_type[b] + 1];
    clus_n[cl] = k + 1;
    t.ncons++;
  }
  t.nclus = (int)clus_n.size();
  // ---- per-term coefficient expansion; unconstrained bonds first ----
  std::vector<int> bond_at, angle_at(s->angle_atoms, s->angle_atoms + 3 * (size_t)s->nangles),
      dih_at(s->dihedral_atoms, s->dihedral_atoms + 4 * (size_t)s->ndihedrals),
      imp_at(s->improper_atoms, s->improper_atoms + 4 * (size_t)s->nimpropers);
  std::vector<double> bond_cf, angle_cf, dih_cf, imp_cf;
  for (int pass = 0; pass < 2; pass++)
    for (int b = 0; b < s->nbonds; b++) {
      if ((int)shaken[b] != pass) continue;
      bond_at.push_back(s->bond_atoms[2 * b]);
      bond_at.push_back(s->bond_atoms[2 * b + 1]);
      bond_cf.push_back(s->bond_coeff[2 * s->bond_type[b]]);
      bond_cf.push_back(s->bond_coeff[2 * s->bond_type[b] + 1]);
      if (pass == 0) t.nbonds_noshake++;
    }
  t.nbonds = s->nbonds;
  t.nangles = s->nangles;
  t.ndihedrals = s->ndihedrals;
  t.nimpropers = s->nimpropers;
  for (int m = 0; m < s->nangles; m++) {
    angle_cf.push_back(s->angle_coeff[2 * s->angle_type[m]]);
    angle_cf.push_back(s->angle_coeff[2 * s->angle_type[m] + 1]);
  }
  for (int m = 0; m < s->ndihedrals; m++)
    for (int k = 0; k < 4; k++) dih_cf.push_back(s->dihedral_coeff[4 * s->dihedral_type[m] + k]);
  for (int m = 0; m < s->nimpropers; m++) {
    imp_cf.push_back(s->improper_coeff[2 * s->improper_type[m]]);
    imp_cf.push_back(s->improper_coeff[2 * s->improper_type[m] + 1]);
  }
  for (size_t k = 0; k < angle_at.size(); k++)
    if (angle_at[k] < 0 || angle_at[k] >= n) return fail(e, SCEMA_MD_ERR_ARG, "bad angle atom");
  for (size_t k = 0; k < dih_at.size(); k++)
    if (dih_at[k] < 0 || dih_at[k] >= n) return fail(e, SCEMA_MD_ERR_ARG, "bad dihedral atom");
  for (size_t k = 0; k < imp_at.size(); k++)
    if (imp_at[k] < 0 || imp_at[k] >= n) return fail(e, SCEMA_MD_ERR_ARG, "bad improper atom");
  // ---- bonded tiles (md_bonded.hip): no atomics, no zeroing ----
  // Atoms are ranked by a breadth-first walk of the bond graph, so that consecutive ranks are topological
  // neighbours whatever the numbering of the input.  A tile = BT_OWNERS consecutive ranks (its owners).  It
  // evaluates EVERY term that touches one of its owners and keeps only the forces on its owners, which it writes
  // with plain coalesced stores (fb is indexed by rank): a term whose atoms span two tiles is evaluated by both
  // (chain molecules cut every few hundred atoms: a few per cent of the terms; none for PE-10k, whose 96-atom
  // rings fill a tile two by two) and counted once for the virial and the energies -- by the tile that owns its
  // lowest-ranked atom (the others carry BT_NOCOUNT on their first atom index).  Positions of the tile's local
  // atoms (owners first, then the halo of up to three bonds) are staged and forces accumulated per tile in LDS.
  std::vector<int> rank(n, -1), by_rank;
  by_rank.reserve(n);
  {
    std::vector<int> queue;
    for (int root = 0; root < n; root++) {
      if (rank[root] >= 0) continue;
      rank[root] = (int)by_rank.size();
      by_rank.push_back(root);
      queue.assign(1, root);
      for (size_t h = 0; h < queue.size(); h++)
        for (int c : adj[queue[h]])
          if (rank[c] < 0) {
            rank[c] = (int)by_rank.size();
            by_rank.push_back(c);
            queue.push_back(c);
          }
    }
  }
  const int ntile = (n + BT_OWNERS - 1) / BT_OWNERS;
  struct TermRef { int kind, idx, count; };
  std::vector<std::vector<TermRef>> tile_terms(ntile);
  auto add_term = [&](const int *atoms, int cnt, int kind, int idx) {
    int rmin = rank[atoms[0]];
    for (int k = 1; k < cnt; k++) rmin = std::min(rmin, rank[atoms[k]]);
    int seen[4], ns_ = 0;
    for (int k = 0; k < cnt; k++) {
      const int tl = rank[atoms[k]] / BT_OWNERS;
      bool dup = false;
      for (int q = 0; q < ns_; q++) dup = dup || seen[q] == tl;
      if (dup) continue;
      seen[ns_++] = tl;
      tile_terms[tl].push_back({kind, idx, tl == rmin / BT_OWNERS ? 1 : 0});
    }
  };
  for (int m = 0; m < s->nbonds; m++) add_term(&bond_at[2 * m], 2, m < t.nbonds_noshake ? BT_BOND : BT_BOND_SHAKEN, m);
  for (int m = 0; m < s->nangles; m++) add_term(&angle_at[3 * m], 3, BT_ANGLE, m);
  for (int m = 0; m < s->ndihedrals; m++) add_term(&dih_at[4 * m], 4, BT_DIHEDRAL, m);
  for (int m = 0; m < s->nimpropers; m++) add_term(&imp_at[4 * m], 4, BT_IMPROPER, m);
  for (int m = 0; m < t.nspecial; m++) add_term(&sp_at[2 * m], 2, BT_SPECIAL, m);
  // Tile-ordered term stream: ONE 64-bit descriptor per term (BT_D_* in md_types.h: four 10-bit local atom indices, the
  // term's type, kind, special-bond level and the count flag); coefficients are looked up by type in small tables the
  // kernel stages in LDS.  Kinds follow each other, each padded to whole chunks of 64 descriptors, so a wave always
  // runs one formula; chunk c of a tile goes to wave c % 4 of its workgroup, which requests all its descriptors with
  // its first instructions: one memory latency per tile instead of one per kind and pass.
  std::vector<int> bond_ty;   // type of the reordered bonds
  for (int pass = 0; pass < 2; pass++)
    for (int b2 = 0; b2 < s->nbonds; b2++)
      if ((int)shaken[b2] == pass) bond_ty.push_back(s->bond_type[b2]);
  std::vector<int> sp_lvl(t.nspecial, 1);
  for (int m = 0; m < t.nspecial; m++)
    for (int lvl = 1; lvl <= 3; lvl++)
      if (sp_cf[2 * m] == s->special_lj[lvl - 1] && sp_cf[2 * m + 1] == s->special_coul[lvl - 1]) { sp_lvl[m] = lvl; break; }
  // coefficient tables: bonds (K, r0), angles (K, theta0), dihedrals (K1..K4), impropers (K, chi0)
  std::vector<double> coef;
  int cf_off[4];
  cf_off[0] = 0;
  coef.insert(coef.end(), s->bond_coeff, s->bond_coeff + 2 * (size_t)s->nbondtypes);
  cf_off[1] = (int)coef.size();
  coef.insert(coef.end(), s->angle_coeff, s->angle_coeff + 2 * (size_t)s->nangletypes);
  cf_off[2] = (int)coef.size();
  coef.insert(coef.end(), s->dihedral_coeff, s->dihedral_coeff + 4 * (size_t)s->ndihedraltypes);
  cf_off[3] = (int)coef.size();
  coef.insert(coef.end(), s->improper_coeff, s->improper_coeff + 2 * (size_t)s->nimpropertypes);
  if (coef.size() > BT_MAXCOEF)
    return fail(e, SCEMA_MD_ERR_ARG, "%zu bonded coefficients (bond/angle/dihedral/improper types): at most %d fit the LDS table of the bonded kernel", coef.size(), BT_MAXCOEF);
  if (s->nbondtypes > BT_D_TMASK + 1 || s->nangletypes > BT_D_TMASK + 1 || s->ndihedraltypes > BT_D_TMASK + 1 || s->nimpropertypes > BT_D_TMASK + 1)
    return fail(e, SCEMA_MD_ERR_ARG, "more than %d types of one bonded kind", BT_D_TMASK + 1);
  for (int k = 0; k < 4; k++) t.bt_cf_off[k] = cf_off[k];
  t.bt_ncoef = (int)coef.size();
  for (int k = 0; k < 3; k++) { t.sp_w[k] = s->special_lj[k]; t.sp_w[3 + k] = s->special_coul[k]; }
  std::vector<int> bt_desc((size_t)ntile * BT_DESC, 0), bt_atoms;
  std::vector<unsigned long long> bt_terms;
  const int natm[BT_NKIND] = {2, 2, 3, 4, 4, 2};
  std::vector<int> local_of(n, -1);
  t.bt_maxloc = 1;
  t.bt_maxchunk = 1;
  for (int tl = 0; tl < ntile; tl++) {
    int *desc = &bt_desc[(size_t)tl * BT_DESC];
    desc[0] = (int)bt_atoms.size();
    // local atom list: the owners in rank order (local index = rank - first rank of the tile), then the halo by rank
    std::vector<int> members, halo;
    const int r0 = tl * BT_OWNERS, r1 = std::min(n, r0 + BT_OWNERS);
    for (int r = r0; r < r1; r++) { members.push_back(by_rank[r]); local_of[by_rank[r]] = r - r0; }
    auto term_atoms = [&](const TermRef &tr) -> const int * {
      const int m = tr.idx;
      return (tr.kind <= BT_BOND_SHAKEN) ? &bond_at[2 * m] : (tr.kind == BT_ANGLE) ? &angle_at[3 * m] : (tr.kind == BT_DIHEDRAL) ? &dih_at[4 * m]
             : (tr.kind == BT_IMPROPER) ? &imp_at[4 * m] : &sp_at[2 * m];
    };
    for (const TermRef &tr : tile_terms[tl]) {
      const int *at = term_atoms(tr);
      for (int k = 0; k < natm[tr.kind]; k++)
        if (local_of[at[k]] < 0) { local_of[at[k]] = 0; halo.push_back(at[k]); }
    }
    std::sort(halo.begin(), halo.end(), [&](int a, int b) { return rank[a] < rank[b]; });
    for (size_t l = 0; l < halo.size(); l++) local_of[halo[l]] = (int)(members.size() + l);
    members.insert(members.end(), halo.begin(), halo.end());
    if ((int)members.size() > BT_D_LMASK + 1)
      return fail(e, SCEMA_MD_ERR_ARG, "a bonded tile touches %zu atoms (more than %d): topology too branched for the tile descriptors", members.size(), BT_D_LMASK + 1);
    desc[2] = (int)(bt_terms.size() / 64);   // first chunk of the tile
    for (int kind = 0; kind < BT_NKIND; kind++) {
      // Terms that follow each other in the input share atoms (the nine torsions around one bond): dealt to
      // consecutive lanes they would hit the same LDS accumulators in the same instruction.  A stride
      // permutation spreads them over the tile instead.
      std::vector<const TermRef *> of_kind;
      for (const TermRef &tr : tile_terms[tl])
        if (tr.kind == kind) of_kind.push_back(&tr);
      const int nk_ = (int)of_kind.size();
      int stride = 1;
      if (nk_ > 16)
        for (stride = 13; stride < nk_; stride += 2) {   // smallest odd stride >= 13 coprime with nk_
          int a_ = stride, b_ = nk_;
          while (b_) { const int t_ = a_ % b_; a_ = b_; b_ = t_; }
          if (a_ == 1) break;
        }
      if (stride >= nk_) stride = 1;
      for (int q = 0; q < nk_; q++) {
        const TermRef &tr = *of_kind[(int)(((long long)q * stride) % std::max(nk_, 1))];
        const int m = tr.idx;
        const int *at = term_atoms(tr);
        unsigned long long d = BT_D_VALID | ((unsigned long long)kind << BT_D_KIND_SHIFT);
        for (int k = 0; k < natm[kind]; k++) d |= (unsigned long long)local_of[at[k]] << (10 * k);
        const int ty = (kind <= BT_BOND_SHAKEN) ? bond_ty[m] : (kind == BT_ANGLE) ? s->angle_type[m] : (kind == BT_DIHEDRAL) ? s->dihedral_type[m]
                       : (kind == BT_IMPROPER) ? s->improper_type[m] : 0;
        d |= (unsigned long long)ty << BT_D_TYPE_SHIFT;
        if (kind == BT_SPECIAL) d |= (unsigned long long)sp_lvl[m] << BT_D_LVL_SHIFT;
        if (!tr.count) d |= BT_D_NOCOUNT;
        bt_terms.push_back(d);
      }
      while (bt_terms.size() % 64) bt_terms.push_back(0ull);   // whole chunks per kind (an invalid descriptor is all zero)
    }
    desc[3] = (int)(bt_terms.size() / 64) - desc[2];   // chunks of the tile
    desc[1] = (int)members.size();
    desc[14] = r1 - r0;   // owners
    t.bt_maxloc = std::max(t.bt_maxloc, (int)members.size());
    t.bt_maxchunk = std::max(t.bt_maxchunk, desc[3]);
    for (int atom : members) { bt_atoms.push_back(atom); local_of[atom] = -1; }
  }
  t.bt_ntile = ntile;
  t.original.take(*s);
  std::memcpy(t.init_box, s->box, sizeof t.init_box);
  t.init_x.assign(s->x, s->x + 3 * (size_t)n);
  t.init_v.assign(s->v, s->v + 3 * (size_t)n);
  int rc;
  if ((rc = upload(e, t.d_type, t.type))) return rc;
  if ((rc = upload(e, t.d_q, t.q))) return rc;
  if ((rc = upload(e, t.d_mass, t.mass_atom))) return rc;
  if ((rc = upload(e, t.d_lj, t.lj))) return rc;
  if ((rc = upload(e, t.d_bt_terms, bt_terms))) return rc;
  if ((rc = upload(e, t.d_bt_coef, coef))) return rc;
  if ((rc = upload(e, t.d_bt_desc, bt_desc))) return rc;
  if ((rc = upload(e, t.d_bt_atoms, bt_atoms))) return rc;
  if ((rc = upload(e, t.d_bt_rank, rank))) return rc;
  if ((rc = upload(e, t.d_ex_start, ex_start))) return rc;
  if ((rc = upload(e, t.d_ex_list, ex_list))) return rc;
  if ((rc = upload(e, t.d_clus_at, clus_at))) return rc;
  if ((rc = upload(e, t.d_clus_n, clus_n))) return rc;
  if ((rc = upload(e, t.d_clus_d, clus_d))) return rc;
  return SCEMA_MD_OK;
}

// -------------------------------------------------------------------------------------------
// Ewald run parameters from the current box: g_ewald rule of "kspace_style pppm <acc>"
// (in.set.lammps:36) and the k-vector set of the reciprocal sum it approximates
// -------------------------------------------------------------------------------------------
struct EwaldSetup {
  double g = 0.0;
  std::vector<int> kn;
  std::vector<int> krun;   // per k: length of the run of following k-vectors that continue its row (n3 + 1 each)
  std::vector<int> kgrp;   // groups of k-vectors (n1, +-n2, +-n3): n1, |n2|, |n3|, k index of (+,+), (-,+), (+,-), (-,-) or -1, pad
  int kmaxd[3] = {0, 0, 0};
};
void ewald_tables(EwaldSetup &out);
void ewald_setup(const scema_md_params &p, const Topo &t, const double *box, EwaldSetup &out, bool g_only = false) {
  out = EwaldSetup();
  if (t.qsqsum == 0.0) return;
  HostBox b;
  box_derive(box, b);
  const double accuracy = p.kspace_accuracy * MD_QQRD2E;
  const double q2 = t.qsqsum * MD_QQRD2E;
  const double rc = p.cut_coul;
  const double tt = accuracy * std::sqrt((double)t.natoms * rc * b.h[0] * b.h[1] * b.h[2]) / (2.0 * q2);
  out.g = (tt >= 1.0) ? (1.35 - 0.15 * std::log(accuracy)) / rc : std::sqrt(-std::log(tt)) / rc;
  if (g_only) return;   // PPPM starts from this g_ewald and has no use for the k list
  const double g = out.g;
  int kmax[3];
  double gsqmx = 0.0;
  for (int d = 0; d < 3; d++) {
    const double L = b.h[d];
    int km = 1;
    for (;;) {
      const double err = 2.0 * q2 * g / L * std::sqrt(1.0 / (MD_PI * km * t.natoms)) * std::exp(-MD_PI * MD_PI * km * km / (g * g * L * L));
      if (err <= accuracy) break;
      km++;
    }
    kmax[d] = km;
    const double u = 2.0 * MD_PI * km / L;
    gsqmx = std::max(gsqmx, u * u);
  }
  gsqmx *= 1.00001;
  const int r0 = kmax[0] + 2, r1 = kmax[1] + 2, r2 = kmax[2] + 2;
  for (int n1 = 0; n1 <= r0; n1++)
    for (int n2 = -r1; n2 <= r1; n2++)
      for (int n3 = -r2; n3 <= r2; n3++) {
        if (n1 == 0 && (n2 < 0 || (n2 == 0 && n3 <= 0))) continue;
        const double kx = 2.0 * MD_PI * (b.hinv[0] * n1);
        const double ky = 2.0 * MD_PI * (b.hinv[5] * n1 + b.hinv[1] * n2);
        const double kz = 2.0 * MD_PI * (b.hinv[4] * n1 + b.hinv[3] * n2 + b.hinv[2] * n3);
        if (kx * kx + ky * ky + kz * kz > gsqmx) continue;
        out.kn.push_back(n1);
        out.kn.push_back(n2);
        out.kn.push_back(n3);
        out.kmaxd[0] = std::max(out.kmaxd[0], std::abs(n1));
        out.kmaxd[1] = std::max(out.kmaxd[1], std::abs(n2));
        out.kmaxd[2] = std::max(out.kmaxd[2], std::abs(n3));
      }
  ewald_tables(out);
}

// row run lengths, (n1, +-n2, +-n3) groups and index ranges of a k-vector list.  The list is put in SNAKE order first:
// slabs of equal n1 ascending; the rows (n1, n2) of a slab ascending or descending in n2, alternating from slab to slab;
// the entries of a row ascending or descending in n3, alternating from row to row.  k_ewald_force walks the list with a
// phase cursor (one complex multiplication per k-vector inside a row): in snake order the cursor only ever moves a step
// or two between rows instead of rewinding n3 across the whole sphere (that rewind was half of the kernel's
// instructions).  krun[k] = +-(number of following k-vectors that continue the row), the sign is the row's direction.
// Also used after a box flip, when the list of the run is re-expressed in the new reciprocal basis.
void ewald_tables(EwaldSetup &out) {
  const int nk = (int)out.kn.size() / 3;
  {
    std::vector<int> idx(nk), kn2(out.kn.size());
    for (int k = 0; k < nk; k++) idx[k] = k;
    std::sort(idx.begin(), idx.end(), [&](int a, int b) {
      for (int d = 0; d < 3; d++)
        if (out.kn[3 * a + d] != out.kn[3 * b + d]) return out.kn[3 * a + d] < out.kn[3 * b + d];
      return false;
    });
    // lexicographic -> snake
    std::vector<int> snake;
    snake.reserve(nk);
    int slab = 0, rowno = 0;
    for (int s0 = 0; s0 < nk;) {
      int s1 = s0;
      while (s1 < nk && out.kn[3 * idx[s1]] == out.kn[3 * idx[s0]]) s1++;
      std::vector<std::pair<int, int>> rows;   // [begin, end) of the rows of this slab, in lexicographic order
      for (int r0 = s0; r0 < s1;) {
        int r1 = r0;
        while (r1 < s1 && out.kn[3 * idx[r1] + 1] == out.kn[3 * idx[r0] + 1]) r1++;
        rows.push_back({r0, r1});
        r0 = r1;
      }
      if (slab & 1) std::reverse(rows.begin(), rows.end());
      for (const auto &rw : rows) {
        if (rowno & 1) for (int k = rw.second - 1; k >= rw.first; k--) snake.push_back(idx[k]);
        else for (int k = rw.first; k < rw.second; k++) snake.push_back(idx[k]);
        rowno++;
      }
      slab++;
      s0 = s1;
    }
    for (int k = 0; k < nk; k++)
      for (int d = 0; d < 3; d++) kn2[3 * k + d] = out.kn[3 * snake[k] + d];
    out.kn.swap(kn2);
  }
  out.krun.assign(nk, 0);
  out.kgrp.clear();
  for (int d = 0; d < 3; d++) out.kmaxd[d] = 0;
  for (int k = 0; k < nk; k++)
    for (int d = 0; d < 3; d++) out.kmaxd[d] = std::max(out.kmaxd[d], std::abs(out.kn[3 * k + d]));
  for (int k = nk - 2; k >= 0; k--) {
    if (out.kn[3 * k] != out.kn[3 * k + 3] || out.kn[3 * k + 1] != out.kn[3 * k + 4]) continue;
    const int step = out.kn[3 * k + 5] - out.kn[3 * k + 2];
    if (step != 1 && step != -1) continue;
    // continue the run only in the same direction
    const int nxt = out.krun[k + 1];
    out.krun[k] = (nxt != 0 && (nxt > 0) == (step > 0)) ? nxt + step : step;
  }
  // k-vectors that differ only in the signs of n2, n3 share every phase-factor product of k_ewald_sfac
  std::map<long, int> gidx;
  for (int k = 0; k < nk; k++) {
    const int n1 = out.kn[3 * k], n2 = out.kn[3 * k + 1], n3 = out.kn[3 * k + 2];
    const long key = ((long)n1 << 40) | ((long)std::abs(n2) << 20) | (long)std::abs(n3);
    auto it = gidx.find(key);
    if (it == gidx.end()) {
      it = gidx.emplace(key, (int)out.kgrp.size() / 8).first;
      out.kgrp.insert(out.kgrp.end(), {n1, std::abs(n2), std::abs(n3), -1, -1, -1, -1, 0});
    }
    out.kgrp[8 * it->second + 3 + (n2 < 0 ? 1 : 0) + (n3 < 0 ? 2 : 0)] = k;
  }
}

// Real-space Ewald factor erfc(x) + 2x/sqrt(pi) exp(-x^2) = 1 - x H(u), u = x^2.  H is entire in u:
// H(u) = 2/sqrt(pi) sum_{n>=1} (-1)^(n+1) u^n/n! 2n/(2n+1).  Fit H on [0, (g rc)^2] by Chebyshev
// interpolation (degree grown until the tail is below 2e-16 relative) and hand the kernel monomial
// coefficients in t = 2u/umax - 1.  All in long double; the fit is checked against H on a fine grid.
static long double coul_H(long double u) {
  const long double c = 2.0L / sqrtl(acosl(-1.0L));
  if (u < 1.0L) {
    long double term = 1.0L, sum = 0.0L;
    for (int n = 1; n < 200; n++) {
      term *= u / n;  // u^n/n!
      const long double t = term * (2.0L * n) / (2.0L * n + 1.0L);
      sum += (n % 2 == 1) ? t : -t;
      if (t < 1e-24L * fabsl(sum)) break;
    }
    return c * sum;
  }
  const long double x = sqrtl(u);
  return (erfl(x) - c * x * expl(-u)) / x;
}

// One Chebyshev fit of H on [0, umax] with N coefficients, converted to monomials in t; returns the maximum
// error of x*H (the quantity the force uses), evaluated in double arithmetic exactly as the kernel does.
static double fit_coul_poly_n(long double umax, int N, double *poly) {
  const long double PI = acosl(-1.0L);
  std::vector<long double> c(N, 0.0L), fv(N);
  for (int j = 0; j < N; j++) fv[j] = coul_H(0.5L * umax * (cosl(PI * (j + 0.5L) / N) + 1.0L));
  for (int k = 0; k < N; k++) {
    long double s = 0.0L;
    for (int j = 0; j < N; j++) s += fv[j] * cosl(PI * k * (j + 0.5L) / N);
    c[k] = 2.0L * s / N;
  }
  // Chebyshev -> monomial in t
  std::vector<long double> a(N, 0.0L), Tkm1(N, 0.0L), Tk(N, 0.0L), Tn(N, 0.0L);
  Tkm1[0] = 1.0L;                      // T0
  a[0] += 0.5L * c[0];
  if (N > 1) {
    Tk[1] = 1.0L;                      // T1
    a[1] += c[1];
  }
  for (int k = 2; k < N; k++) {
    std::fill(Tn.begin(), Tn.end(), 0.0L);
    for (int m = 0; m < N - 1; m++) Tn[m + 1] += 2.0L * Tk[m];
    for (int m = 0; m < N; m++) Tn[m] -= Tkm1[m];
    for (int m = 0; m < N; m++) a[m] += c[k] * Tn[m];
    Tkm1 = Tk;
    Tk = Tn;
  }
  for (int m = 0; m < N; m++) poly[m] = (double)a[m];
  for (int m = N; m < MD_MAXPOLY; m++) poly[m] = 0.0;
  const double uscale = (double)(2.0L / umax);
  double maxerr = 0.0;
  for (int s = 0; s <= 400; s++) {
    const double u = (double)umax * s / 400.0 / 1.000001;
    const double t = u * uscale - 1.0;
    double p = poly[N - 1];
    for (int m = N - 2; m >= 0; m--) p = std::fma(p, t, poly[m]);
    const double x = std::sqrt(u);
    maxerr = std::max(maxerr, std::fabs(x * (p - (double)coul_H(u))));
  }
  return maxerr;
}

// Smallest number of coefficients whose fit error is below 2e-13 (absolute, on a factor of order one):
// three orders below the parity budget of the forces (1e-11 relative), seven below LAMMPS' own table
// (pair_modify table 12: ~1e-6).  Every coefficient is one FP64 FMA per coulomb pair in k_pair.
static double fit_coul_poly(double g, double rc, double *poly, int *npoly, double *uscale) {
  if (g <= 0.0) {
    poly[0] = 0.0;
    *npoly = 1;
    *uscale = 0.0;
    return 0.0;
  }
  const long double umax = (long double)(g * rc) * (g * rc) * 1.000001L;
  *uscale = (double)(2.0L / umax);
  double err = 0.0, target = 2e-13;
  // measurement knob: what the precision of this factor costs (LAMMPS' own table is good to ~1e-6); parity tests run at the default
  if (const char *tv = getenv("SCEMA_MD_POLY_TOL")) target = std::min(1e-3, std::max(1e-15, atof(tv)));
  for (int N = 6; N <= MD_MAXPOLY; N++) {   // k_pair<.., 16> and beyond spill registers: an odd count that suffices is worth having
    err = fit_coul_poly_n(umax, N, poly);
    *npoly = N;
    if (err < target) break;
  }
  return err;
}

// ---- PPPM set-up on the host (kspace_style 1): PPPM::set_grid_global and adjust_gewald of LAMMPS' pppm.cpp (17Nov16; ik
// differentiation, not staggered) with their loop structure [LAMMPS-ext: restated from the published source as remembered,
// LAMMPS is not in the reference tree]:
//   * per dimension the search starts at n = int(prd * g) + 1 with h = 1 / g and runs `while (err > accuracy) { err = E(h);
//     n++; h = prd / n; }` -- the increment follows the evaluation, so it stops ONE PAST the first admissible grid;
//   * the box is LAMMPS' triclinic box (in.init.lammps:27 `change_box all triclinic`; the engine's box always carries its
//     three tilts), for which the grid is rescaled: n = int(lamda2xT(n / prd)) + 1;
//   * each n is raised to a product of 2, 3, 5; the spacings are the reciprocals of x2lamdaT(n);
//   * g_ewald by Newton steps on (real-space error - k-space error) with a forward difference of 1e-6, stopped at the first
//     iterate with |f| < 1e-5.
// The oracle restates the same routine on its own (oracle/md_oracle.c pppm_setup). ----
static double pppm_ik_error(double h, double prd, double g, double q2, double natoms) {
  // estimate_ik_error with the arithmetic of pppm.cpp (pow, not repeated multiplication): g_ewald comes out of a Newton step with
  // a forward difference of 1e-6, which amplifies the last bits of this function a millionfold
  static const double ACONS5[5] = {1.0 / 23232.0, 7601.0 / 13628160.0, 143.0 / 69120.0, 517231.0 / 106536960.0, 106640677.0 / 11737571328.0};
  double sum = 0.0;
  for (int m = 0; m < 5; m++) sum += ACONS5[m] * std::pow(h * g, 2.0 * m);
  return q2 * std::pow(h * g, 5.0) * std::sqrt(g * prd * std::sqrt(2.0 * MD_PI) * sum / natoms) / (prd * prd);
}
static void pppm_setup_host(const scema_md_params &p, const Topo &t, const double *box, double &g, int pg[3]) {
  HostBox b;
  box_derive(box, b);   // b.h = xprd, yprd, zprd, yz, xz, xy
  const double accuracy = p.kspace_accuracy * MD_QQRD2E, q2 = t.qsqsum * MD_QQRD2E, rc = p.cut_coul, N = (double)t.natoms;
  auto factorable = [](int n) { while (n % 2 == 0) n /= 2; while (n % 3 == 0) n /= 3; while (n % 5 == 0) n /= 5; return n == 1; };
  int n[3];
  for (int d = 0; d < 3; d++) {
    const double prd = b.h[d];
    double h = 1.0 / g;
    n[d] = (int)(prd / h) + 1;
    double err = pppm_ik_error(h, prd, g, q2, N);
    while (err > accuracy && n[d] < 4096) {
      err = pppm_ik_error(h, prd, g, q2, N);
      n[d]++;
      h = prd / n[d];
    }
  }
  {
    const double t0 = n[0] / b.h[0], t1 = n[1] / b.h[1], t2 = n[2] / b.h[2];
    const double u0 = b.h[0] * t0, u1 = b.h[5] * t0 + b.h[1] * t1, u2 = b.h[4] * t0 + b.h[3] * t1 + b.h[2] * t2;
    // (n / prd) * prd is n or one ulp beside it: without a tilt contribution the truncation would be a coin flip on the last bit of
    // the box length.  Decided as exact arithmetic would (the oracle carries the same guard; DESIGN.md section 2, deviation 1).
    const double guard = 1.0e-9;
    n[0] = (int)(u0 + guard) + 1; n[1] = (int)(u1 + guard) + 1; n[2] = (int)(u2 + guard) + 1;
  }
  for (int d = 0; d < 3; d++) {
    n[d] = std::max(n[d], 2);
    while (!factorable(n[d])) n[d]++;
    pg[d] = n[d];
  }
  const double hs[3] = {1.0 / (b.hinv[0] * pg[0]), 1.0 / (b.hinv[5] * pg[0] + b.hinv[1] * pg[1]), 1.0 / (b.hinv[4] * pg[0] + b.hinv[3] * pg[1] + b.hinv[2] * pg[2])};
  auto f = [&](double gg) {
    const double df_r = 2.0 * q2 * std::exp(-gg * gg * rc * rc) / std::sqrt(N * rc * b.h[0] * b.h[1] * b.h[2]);
    double sq = 0.0;
    for (int d = 0; d < 3; d++) { const double e = pppm_ik_error(hs[d], b.h[d], gg, q2, N); sq += e * e; }
    return df_r - std::sqrt(sq) / std::sqrt(3.0);
  };
  for (int it = 0; it < 10000; it++) {
    const double step = 0.000001, f1 = f(g), f2 = f(g + step);
    g -= f1 / ((f2 - f1) / step);
    if (std::fabs(f(g)) < 0.00001) break;
  }
}

struct PolyFit { int n; double uscale, err; double c[MD_MAXPOLY]; };
static std::map<long, PolyFit> &poly_cache() { static std::map<long, PolyFit> m; return m; }
static double cached_coul_poly(scema_md_engine *, double g, double rc, double *poly, int *npoly, double *uscale) {
  if (g <= 0.0) return fit_coul_poly(g, rc, poly, npoly, uscale);
  const double x = g * rc;
  const long key = (long)std::ceil(x * 64.0);          // x rounded up to 1/64
  auto it = poly_cache().find(key);
  if (it == poly_cache().end()) {
    PolyFit f;
    f.err = fit_coul_poly(key / 64.0, 1.0, f.c, &f.n, &f.uscale);
    it = poly_cache().emplace(key, f).first;
  }
  std::memcpy(poly, it->second.c, sizeof(double) * MD_MAXPOLY);
  *npoly = it->second.n;
  *uscale = it->second.uscale;
  return it->second.err;
}

// fix deform's tilt rules (LAMMPS 17Nov16 fix_deform.cpp end_of_step; the oracle states them as omd_tilt_closest /
// omd_tilt_flip, k_post as the same arithmetic on the device).  tilt = xy, xz, yz.
void tilt_closest(double tilt[3], double xprd_new, double yprd_new, double xy, double xz, double yz, double xprd, double yprd) {
  const double denom[3] = {xprd_new, xprd_new, yprd_new};
  const double current[3] = {xy / xprd, xz / xprd, yz / yprd};
  for (int i = 0; i < 3; i++) {
    while (tilt[i] / denom[i] - current[i] > 0.0) tilt[i] -= denom[i];
    while (tilt[i] / denom[i] - current[i] < 0.0) tilt[i] += denom[i];
    if (std::fabs(tilt[i] / denom[i] - 1.0 - current[i]) < std::fabs(tilt[i] / denom[i] - current[i])) tilt[i] -= denom[i];
  }
}
int tilt_flip(const double tilt[3], double xprd, double yprd, double flipped[3], int nflip[3]) {
  const double xprdinv = 1.0 / xprd, yprdinv = 1.0 / yprd;
  flipped[0] = tilt[0]; flipped[1] = tilt[1]; flipped[2] = tilt[2];
  nflip[0] = nflip[1] = nflip[2] = 0;
  if (!(tilt[2] * yprdinv < -0.5 || tilt[2] * yprdinv > 0.5 || tilt[1] * xprdinv < -0.5 || tilt[1] * xprdinv > 0.5 ||
        tilt[0] * xprdinv < -0.5 || tilt[0] * xprdinv > 0.5))
    return 0;
  if (flipped[2] * yprdinv < -0.5) { flipped[2] += yprd; flipped[1] += flipped[0]; nflip[2] = 1; }
  else if (flipped[2] * yprdinv > 0.5) { flipped[2] -= yprd; flipped[1] -= flipped[0]; nflip[2] = -1; }
  if (flipped[1] * xprdinv < -0.5) { flipped[1] += xprd; nflip[1] = 1; }
  if (flipped[1] * xprdinv > 0.5) { flipped[1] -= xprd; nflip[1] = -1; }
  if (flipped[0] * xprdinv < -0.5) { flipped[0] += xprd; nflip[0] = 1; }
  if (flipped[0] * xprdinv > 0.5) { flipped[0] -= xprd; nflip[0] = -1; }
  return (nflip[0] || nflip[1] || nflip[2]) ? 1 : 0;
}

// The box trajectory of a fix-deform run is known in advance (rates, dt, number of steps): the host walks it with the
// arithmetic of k_post and finds the steps after which the triclinic box flips ("flip yes", the LAMMPS default used by
// in.strain.lammps:94-100).  A flip is then enqueued between two steps of the device-side run: new tilts, a forced list
// rebuild and the k-vector tables in the new reciprocal basis (run_phase).
struct FlipEvent {
  int step;          // the flip is detected at the end of this step and applied at the start of the next one
  double tilt[3];    // xy, xz, yz after the flip
  int nflip[3];      // lattice steps f_xy, f_xz, f_yz: a2' = a2 + f_xy a1, a3' = a3 + f_yz a2 + f_xz a1
};
// returns false if the run would need a yz flip: that changes xz by xy, which the linear tilt targets of the other
// components cannot follow -- LAMMPS refuses such a run ("Fix deform is changing yz too much with xy"; in.strain.lammps
// deforms all six components, so yz and xy are always both active).  xy and xz flip freely.
bool deform_trajectory(const double *box0, const double *rates, double dt, int nsteps, double *box_end, std::vector<FlipEvent> &events,
                       std::vector<HostBox> &extremes) {
  double cur[9];
  std::memcpy(cur, box0, sizeof cur);
  bool pending = false;
  FlipEvent pe{};
  for (int step = 1; step <= nsteps; step++) {
    if (pending) {   // applied at the start of this step
      cur[6] = pe.tilt[0]; cur[7] = pe.tilt[1]; cur[8] = pe.tilt[2];
      events.push_back(pe);
      pending = false;
    }
    const double t = step * dt;
    double nb[9];
    for (int d = 0; d < 3; d++) {
      const double L0 = box0[3 + d] - box0[d];
      nb[d] = box0[d] - 0.5 * L0 * rates[d] * t;
      nb[3 + d] = box0[3 + d] + 0.5 * L0 * rates[d] * t;
    }
    double tilt[3] = {box0[6] + rates[3] * (box0[4] - box0[1]) * t, box0[7] + rates[4] * (box0[5] - box0[2]) * t,
                      box0[8] + rates[5] * (box0[5] - box0[2]) * t};
    tilt_closest(tilt, nb[3] - nb[0], nb[4] - nb[1], cur[6], cur[7], cur[8], cur[3] - cur[0], cur[4] - cur[1]);
    nb[6] = tilt[0]; nb[7] = tilt[1]; nb[8] = tilt[2];
    std::memcpy(cur, nb, sizeof cur);
    pe.step = step;
    if (tilt_flip(tilt, nb[3] - nb[0], nb[4] - nb[1], pe.tilt, pe.nflip)) {
      if (pe.nflip[2] != 0) return false;
      pending = true;   // a flip that falls behind the last step of the run is never applied (the fix is gone by then)
      HostBox hb;
      box_derive(cur, hb);
      extremes.push_back(hb);
    }
  }
  std::memcpy(box_end, cur, sizeof cur);
  return true;
}

double wall_s() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

double round_trip(const char *fmt, double v) {
  char buf[512];
  snprintf(buf, sizeof buf, fmt, v);
  return strtod(buf, nullptr);
}

// -------------------------------------------------------------------------------------------
// one "run" of a batch
// -------------------------------------------------------------------------------------------
struct EwaldSetup;
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
};

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

#include "md_reax_engine.inc"

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
  // Two half batches on two streams (large batches only): every kernel but k_pair is latency bound and leaves most issue
  // slots idle, while k_pair saturates them and holds every wave slot of the chip; with two independent halves in flight
  // the small kernels of one half fill in as the pair workgroups of the other retire (the in-order streams fall half a
  // step out of phase by themselves).  The halves take the even and the odd ranks of the length order, so each is
  // itself sorted longest first.
  const int nhalf = (e->split_streams && e->stream3 != nullptr && ns >= e->split_min && ns < e->split_max) ? 2 : 1;
  if (nhalf == 2) {
    std::vector<int> o2;
    o2.reserve(ns);
    for (int r = 0; r < ns; r += 2) o2.push_back(order[r]);
    for (int r = 1; r < ns; r += 2) o2.push_back(order[r]);
    order.swap(o2);
  }
  const int hbeg[2] = {0, nhalf == 2 ? (ns + 1) / 2 : ns}, hcnt[2] = {nhalf == 2 ? (ns + 1) / 2 : ns, nhalf == 2 ? ns / 2 : 0};
  e->h_sims.assign(ns, SimDev());
  int maxbt = 1, maxloc = 1, maxcoef = 0;
  int maxrow = 64, maxcapj = 64, maxpoly = 1, maxatoms = 0, maxpad = 0, maxcells = 0, maxk = 0, mmax = 1, maxb = 0, maxa = 0, maxd = 0, maxi = 0, maxs = 0, maxclus = 0, maxsteps = 0;
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
      mn_out = (std::min(mn_out, cj_out) + 63) / 64 * 64;
      return cj_out <= MD_MAXJTAB && mdk_pair_lds_bytes(cj_out) <= 74 * 1024 && mdk_neigh_lds_bytes(cj_out, mn_out) <= 150 * 1024;
    };
    // first among cell edges between rlist/2 and rlist; if no such grid fits, among edges down to rlist/4 (so that a
    // slightly denser system degrades gradually instead of dropping to the uniform fallback below)
    static const int small_max = getenv("SCEMA_MD_SMALL_CELLS_MAX") ? atoi(getenv("SCEMA_MD_SMALL_CELLS_MAX")) : 8;   // replicas up to which the most-cells grid is taken
    const bool small_batch = ns <= small_max && !getenv("SCEMA_MD_BIG_CELLS");
    for (int pass = 0; pass < 2 && !fits; pass++) {
      int lo[3], hi[3];
      for (int d = 0; d < 3; d++) {
        const double w = std::min(w0[d], w1[d]);
        lo[d] = std::max(2, std::min(64, (int)std::floor(w / (rlist * 1.0001))));
        hi[d] = std::max(lo[d], std::min(64, (int)std::floor(w / ((pass == 0 ? 0.5 : 0.25) * rlist * 1.0001))));
      }
      double best = -1.0;
      for (int n0 = lo[0]; n0 <= hi[0]; n0++)
        for (int n1 = lo[1]; n1 <= hi[1]; n1++)
          for (int n2 = lo[2]; n2 <= hi[2]; n2++) {
            const int nc[3] = {n0, n1, n2};
            int mst[3], cj, mn;
            if (!size_grid(nc, mst, cj, mn)) continue;
            // batches that fill the chip take the largest cells (per-tile phases amortised over more rows); small ones the
            // most cells: a single replica on 120 tiles leaves half of the 512 workgroup slots empty and waits for one tile
            const double vol = small_batch ? (double)n0 * n1 * n2 : 1.0 / ((double)n0 * n1 * n2);
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
    }
    const auto tl3 = t_now();
    S.nk = (int)ew.kn.size() / 3;
    for (int d = 0; d < 3; d++) S.kmaxd[d] = std::max(ew.kmaxd[d], 0);
    S.g_ewald = ew.g;
    {
      // H depends on u only: fit once per (rounded-up) range and share it between simulations
      const double perr = cached_coul_poly(e, ew.g, P.cut_coul, S.coul_poly, &S.coul_npoly, &S.coul_uscale);
      if (perr > 1e-12 && !getenv("SCEMA_MD_POLY_TOL")) return fail(e, SCEMA_MD_ERR_ARG, "real-space Ewald polynomial fit error %.3e too large (g*rc = %.3f)", perr, ew.g * P.cut_coul);
      for (int m = 0; m < MD_MAXPOLY; m++) S.coul_poly_g[m] = S.coul_poly[m] * ew.g;
      static const int row_split = getenv("SCEMA_MD_ROW_SPLIT") ? atoi(getenv("SCEMA_MD_ROW_SPLIT")) : 1;
      S.sched_split = row_split;
    }
    {
      const double m = 0.1 * P.skin;   // margin of the row segments over the cutoffs (scan 0 .. 0.6 skin: flat optimum at 0.05-0.15)
      S.seg_a2 = (P.cut_coul + m) * (P.cut_coul + m);
      S.seg_b2 = (P.cut_lj + m) * (P.cut_lj + m);
      // skin pairs listed beyond cutmax + far_band sit at the back of the rows and are skipped until an atom has moved far_band/2
      double frac = 0.65;   // scan 0.25 .. 0.85 on PE-10k (rebuild every ~33 steps, the largest displacement passes 0.5 A after ~8): optimum 0.65-0.75
      if (const char *fv = getenv("SCEMA_MD_FAR_FRAC")) frac = atof(fv);
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
    maxsteps = std::max(maxsteps, A.nsteps);
  }
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
  bool pppm_clean[2] = {false, false};   // per half: the charge grids hold zeros (the buffer is laid out anew for every run)
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
      const bool same = !pppm_runs.empty() && pos != hbeg[1] && S.pg[0] == e->h_sims[pos - 1].pg[0] && S.pg[1] == e->h_sims[pos - 1].pg[1] && S.pg[2] == e->h_sims[pos - 1].pg[2];
      if (same) pppm_runs.back().second += 1;
      else pppm_runs.push_back({pos, 1});
    }
  }
  const bool pppm_in_lds = maxgrid > 0 && maxgrid <= mdk_pppm_solve_max() && (3 * (size_t)maxgrid + (size_t)maxdims) * 16 <= 150 * 1024 && !getenv("SCEMA_MD_PPPM_FFT");
  // Batched 3-d Z2Z plans over grids that lie maxgrid complex elements apart (the charge grids of neighbouring simulations, and
  // all their field grids: three per simulation, simulation-major).  A plan owns work space, so each stream has its own.
  auto pppm_plan = [&](const int pg[3], int batch, hipStream_t st, hipfftHandle &plan) -> int {
    if ((long long)maxgrid > 0x7fffffffLL) return fail(e, SCEMA_MD_ERR_ARG, "PPPM grid of %d points is too large", maxgrid);
    const std::array<int, 6> key = {pg[0], pg[1], pg[2], batch, st == e->stream ? 0 : st == e->stream2 ? 1 : 2, maxgrid};
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
    bool &clean = pppm_clean[(nhalf == 2 && pos0 == hbeg[1]) ? 1 : 0];
    mdk_pppm_spread(st, Dp, na, maxgrid, maxatoms, clean ? 1 : 0);
    clean = false;
    if (pppm_in_lds) {   // small grids: the whole solve in one launch, in LDS (md_pppm.hip k_pppm_solve); it leaves the charge grids zeroed
      if (new_box) mdk_pppm_gf(st, Dp, na, maxgrid);
      mdk_pppm_solve(st, Dp, na, maxgrid, maxdims);
      clean = true;
      mdk_pppm_force(st, Dp, na, maxgrid, maxatoms, add);
      return SCEMA_MD_OK;
    }
    auto transform = [&](bool fields, int dir) -> int {   // the charge grids forward, or the three field grids of every simulation back
      static const bool serial_fft = getenv("SCEMA_MD_PPPM_SERIAL") != nullptr;   // debugging: one transform per simulation and grid
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
  // (its k_pair does not fill the chip and the fork/join is pure latency).  So: batches of 4 to 255 replicas; a batch that
  // fills the chip many times over gains nothing, and inline its k_pair launches are timed and profiled undisturbed.
  const bool pppm_side = maxgrid > 0 && nhalf == 1 && e->stream2 != nullptr && ns >= 4 && ns < 256 && !getenv("SCEMA_MD_PPPM_INLINE");
  auto pppm_fork = [&](hipStream_t st, int pos0, int na, bool new_box) -> int {
    if (!pppm_side) return SCEMA_MD_OK;
    HIPCHK(hipEventRecord(e->ev_fork, st));
    HIPCHK(hipStreamWaitEvent(e->stream2, e->ev_fork, 0));
    const int rc = pppm_stage(e->stream2, pos0, na, new_box, 0);
    if (rc) return rc;
    HIPCHK(hipEventRecord(e->ev_join, e->stream2));
    return SCEMA_MD_OK;
  };
  HIPCHK(e->d_sims.ensure((size_t)ns * sizeof(SimDev)));
  HIPCHK(hipMemcpyAsync(e->d_sims.p, e->h_sims.data(), (size_t)ns * sizeof(SimDev), hipMemcpyHostToDevice, e->stream));
  const auto t_laid_out = std::chrono::steady_clock::now();
  const SimDev *D = e->d_sims.as<SimDev>();
  hipStream_t hs[2] = {e->stream, nhalf == 2 ? e->stream3 : e->stream};
  if (nhalf == 2) {   // the second stream starts behind the uploads
    HIPCHK(hipEventRecord(e->ev_up, e->stream));
    HIPCHK(hipStreamWaitEvent(e->stream3, e->ev_up, 0));
  }
  const int ev = (spec.sample || spec.ev_always || (spec.nh && spec.npt)) ? 1 : 0;   // the barostat needs the virial of every step
  const bool allow_side = nhalf == 1;
  // ---- setup (step 0) ----
  for (int h = 0; h < nhalf; h++) {
    hipStream_t st = hs[h];
    const SimDev *Dh = D + hbeg[h];
    const int nh = hcnt[h];
    mdk_phase_init(st, Dh, nh);
    mdk_neighbor(st, Dh, nh, maxatoms, maxpad, maxcells, maxrow, maxcapj);
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
  auto launch_step = [&](int h, int na, bool timed) -> int {
    hipStream_t st = hs[h];
    const SimDev *Dh = D + hbeg[h];
    if (spec.nh) { mdk_pre_nh(st, Dh, na); mdk_initial_integrate_nh(st, Dh, na, maxatoms); }
    else { mdk_pre(st, Dh, na); mdk_initial_integrate(st, Dh, na, maxatoms); }
    mdk_neighbor(st, Dh, na, maxatoms, maxpad, maxcells, maxrow, maxcapj);
    { const int rcp = pppm_fork(st, hbeg[h], na, spec.deform || (spec.nh && spec.npt)); if (rcp) return rcp; }
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
    HIPCHK(force_stage(e, st, allow_side, Dh, na, maxbt, maxloc, maxcoef, maxatoms, maxk, mmax, maxgrp, spec.ev_always, (ev && !spec.ev_always) ? 1 : 0, pppm_side));
    if (!pppm_side) { const int rcp = pppm_stage(st, hbeg[h], na, spec.deform || (spec.nh && spec.npt)); if (rcp) return rcp; }
    mdk_shake(st, Dh, na, maxclus, 1.0);
    mdk_final_integrate(st, Dh, na, maxatoms, 1);
    if (spec.nh) mdk_post_nh(st, Dh, na);
    else mdk_post(st, Dh, na);
    if (spec.deform) mdk_remap(st, Dh, na, maxatoms);
    return SCEMA_MD_OK;
  };
  auto active = [&](int h, int step) {   // active prefix of half h at this step (sorted by nsteps)
    int na = 0;
    while (na < hcnt[h] && e->h_sims[hbeg[h] + na].nsteps >= step) na++;
    return na;
  };
  // The step loop is launch-bound for small batches (about 20 kernels of a few microseconds each per step of a
  // single replica): the steps that share an active count can be captured once into a hipGraph and replayed.
  // Measured on ROCm 7.2 / MI355X (tools/graph_cmp.py, ms per update of 1 / 72 PE-10k replicas): plain launches
  // 25.8 / 236.7 on one stream, 28.3 / 232.2 with the side stream; graph replay 26.4 / 236.9 on one stream and
  // 55.2 / 251.3 with the side stream inside the graph -- no gain, so replay is opt-in (SCEMA_MD_GRAPH=1).  Not
  // with per-launch event timing (profile mode), which needs the individual launches.
  const bool use_graph = !prof && e->use_graphs && nhalf == 1 && maxgrid == 0;
  // box flips (fix deform, flip yes): step -> positions that flip after it
  std::map<int, std::vector<std::pair<int, int>>> flip_at;
  for (int pos = 0; pos < ns; pos++)
    for (size_t k = 0; k < flips[pos].size(); k++)
      if (flips[pos][k].step < e->h_sims[pos].nsteps) flip_at[flips[pos][k].step].push_back({pos, (int)k});
  std::vector<std::unique_ptr<DevBuf>> flip_bufs;          // k-vector tables in the new reciprocal basis, alive until the run has drained
  std::vector<std::unique_ptr<std::vector<int>>> flip_host;
  std::vector<std::unique_ptr<SimDev>> flip_desc;
  for (int step = 1; step <= maxsteps;) {
    const int na = active(0, step);
    if (na == 0) break;
    int run_len = e->h_sims[na - 1].nsteps - step + 1;   // steps until the active prefix shrinks (sorted by nsteps)
    const int nb = nhalf == 2 ? active(1, step) : 0;
    if (nb > 0) run_len = std::min(run_len, e->h_sims[hbeg[1] + nb - 1].nsteps - step + 1);
    {
      auto nxt = flip_at.lower_bound(step);
      if (nxt != flip_at.end()) run_len = std::min(run_len, nxt->first - step + 1);   // the launch group ends with the flipping step
    }
    bool replayed = false;
    if (use_graph && run_len >= 4) {
      hipStream_t st = hs[0];
      hipGraph_t graph = nullptr;
      hipGraphExec_t gexec = nullptr;
      bool ok = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess;
      if (ok) {
        const int rc_l = launch_step(0, na, false);
        ok = (hipStreamEndCapture(st, &graph) == hipSuccess) && rc_l == SCEMA_MD_OK && graph != nullptr;
      }
      if (ok) ok = hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0) == hipSuccess;
      if (ok) {
        for (int r = 0; r < run_len && ok; r++) ok = hipGraphLaunch(gexec, st) == hipSuccess;
        if (!ok) return fail(e, SCEMA_MD_ERR_DEVICE, "hipGraphLaunch failed");
        replayed = true;
      } else {
        (void)hipGetLastError();
        e->use_graphs = false;   // capture is not available here: plain launches from now on
      }
      if (gexec) (void)hipGraphExecDestroy(gexec);
      if (graph) (void)hipGraphDestroy(graph);
    }
    if (!replayed)
      for (int r = 0; r < run_len; r++) {
        int rc_l = launch_step(0, na, prof);
        if (rc_l) return rc_l;
        if (nb > 0 && (rc_l = launch_step(1, nb, prof))) return rc_l;
      }
    e->prof.md_steps += (long long)(na + nb) * run_len;
    step += run_len;
    // flips detected at the end of step - 1: between the two steps the box takes its flipped tilts, the list rebuild of
    // the next step is forced and the k-vector list is re-expressed in the new reciprocal basis (same vectors:
    // n2 += f_xy n1, n3 += f_yz n2 + f_xz n1), all stream-ordered behind the launches of step - 1
    auto fl = flip_at.find(step - 1);
    if (fl != flip_at.end())
      for (const auto &pk : fl->second) {
        const int pos = pk.first;
        const FlipEvent &fe = flips[pos][pk.second];
        const int h = (nhalf == 2 && pos >= hbeg[1]) ? 1 : 0;
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
  if (nhalf == 2) {
    HIPCHK(hipEventRecord(e->ev_up, e->stream3));
    HIPCHK(hipStreamWaitEvent(e->stream, e->ev_up, 0));
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
  }
  if (getenv("SCEMA_MD_TIMING") && ns > 0) {
    const SimScalars &c = e->h_sc[0];
    const SimDev &S0 = e->h_sims[0];
    fprintf(stderr, "[scema_md] sim 0: cells %dx%dx%d, j table max %d of %d, row max %d of %d, row entries/cluster %.1f, listed pairs/atom %.1f, builds %d\n",
            S0.nc[0], S0.nc[1], S0.nc[2], c.maxj_seen, S0.capj, c.maxneigh_seen, S0.maxneigh, (double)c.nrowent / (S0.npad / MD_CLUSTER),
            (double)c.nentries / S0.natoms, c.nbuilds);
    fprintf(stderr, "[scema_md] host: %.2f ms laying out %d simulations before the first launch of this run (k-space set-up on host threads %.2f, box range %.2f, cell grid %.2f, rest of the loop %.2f)\n",
            std::chrono::duration<double, std::milli>(t_laid_out - t_enter).count(), ns, t_kspace_ms, t_lay[0], t_lay[1], t_lay[3]);
    fprintf(stderr, "[scema_md] sim 0: far skin band walked on %d of %d steps; list skin %.2f A\n", c.nfar_steps, c.step, S0.skin);
#ifdef PAIR_TIMING
    fprintf(stderr, "[scema_md] k_pair wave clocks (sim 0, mean per wave): prologue %.0f, rows %.0f, barrier wait %.0f, flush %.0f (%llu waves)\n",
            (double)c.dbg[0] / c.dbg[4], (double)c.dbg[1] / c.dbg[4], (double)c.dbg[2] / c.dbg[4], (double)c.dbg[3] / c.dbg[4], c.dbg[4]);
    if (c.nbuilds > 0) {
      const double nw = (double)c.nbuilds * S0.ncells * MD_TILE_WAVES;
      fprintf(stderr, "[scema_md] k_neigh_build wave clocks (sim 0, mean per wave and build): table %.0f, rows %.0f, schedule %.0f\n",
              (double)c.dbg[5] / nw, (double)c.dbg[6] / nw, (double)c.dbg[7] / nw);
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
  if (fault & 16) return fail(e, SCEMA_MD_ERR_ARG, "a simulation became unstable (non-finite or runaway atom positions): overlapping atoms or parameters far from the replica's equilibrium");
  if (fault & 2) return fail(e, SCEMA_MD_ERR_ARG, "an excluded (special) pair stretched beyond the exclusion gate; topology or state is broken");
  e->overflow_bits = fault;
  if (fault & 1) return SCEMA_MD_ERR_OVERFLOW;
  if (fault & 64) return SCEMA_MD_ERR_OVERFLOW;   // the barostat took the box out of the range this segment was laid out for
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

State *find_state(scema_md_engine *e, int qp, const char *matid, int replica) {
  auto it = e->states.find(state_key(qp, matid, replica));
  return it == e->states.end() ? nullptr : it->second.get();
}
Topo *find_topo(scema_md_engine *e, const char *matid, int replica) {
  auto it = e->topos.find(topo_key(matid, replica));
  return it == e->topos.end() ? nullptr : it->second.get();
}

int make_state(scema_md_engine *e, Topo *t, const double *box, const double *x, const double *v, bool from_device,
               std::unique_ptr<State> &out) {
  // host-provided states are checked: positions index cells and tables on the device, so nothing non-finite goes up
  if (!from_device) {
    for (int k = 0; k < 9; k++)
      if (!std::isfinite(box[k])) return fail(e, SCEMA_MD_ERR_ARG, "non-finite box");
    if (!(box[3] > box[0]) || !(box[4] > box[1]) || !(box[5] > box[2])) return fail(e, SCEMA_MD_ERR_ARG, "box with non-positive extent");
    for (size_t k = 0; k < 3 * (size_t)t->natoms; k++)
      if (!std::isfinite(x[k]) || !std::isfinite(v[k]) || std::fabs(x[k]) >= 1.0e8)
        return fail(e, SCEMA_MD_ERR_ARG, "non-finite (or runaway) position or velocity of atom %zu", k / 3);
  }
  out.reset(new State());
  out->topo = t;
  std::memcpy(out->box, box, 9 * sizeof(double));
  const size_t bytes = 3 * (size_t)t->natoms * sizeof(double);
  HIPCHK(out->x.ensure(bytes));
  HIPCHK(out->v.ensure(bytes));
  // on the engine's stream (created non-blocking: it does not order against the null stream), so that every later
  // consumer -- backups, kernels -- sees the copy; host sources may be freed by the caller, so those are waited for
  const hipMemcpyKind kind = from_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  HIPCHK(hipMemcpyAsync(out->x.p, x, bytes, kind, e->stream));
  HIPCHK(hipMemcpyAsync(out->v.p, v, bytes, kind, e->stream));
  if (!from_device) HIPCHK(hipStreamSynchronize(e->stream));
  return SCEMA_MD_OK;
}

// buffers of a state whose content arrives from another rank
int make_empty_state(scema_md_engine *e, Topo *t, std::unique_ptr<State> &out) {
  out.reset(new State());
  out->topo = t;
  std::memset(out->box, 0, sizeof out->box);
  const size_t bytes = 3 * (size_t)t->natoms * sizeof(double);
  HIPCHK(out->x.ensure(bytes));
  HIPCHK(out->v.ensure(bytes));
  return SCEMA_MD_OK;
}

// state branch rule of stmd_problem.h:116-138,185-207
// `incoming`: the source state as it arrived from the rank that owned it (scema::PlanMove); it becomes the state of
// qp_id directly.  `created`: set when a new state object was stored under qp_id, with the state it displaced (if any),
// so that a failed update can put things back.
int resolve_state(scema_md_engine *e, const scema_mdsim &m, State **out, std::unique_ptr<State> *incoming = nullptr,
                  bool *created = nullptr, std::unique_ptr<State> *displaced = nullptr) {
  Topo *t = find_topo(e, m.matid, m.replica);
  if (!t) return fail(e, SCEMA_MD_ERR_NOSTATE, "replica %s_%d is not registered (init.%s_%d.bin missing)", m.matid, m.replica, m.matid, m.replica);
  if (created) *created = false;
  if (incoming && *incoming) {
    *out = incoming->get();
    auto &slot = e->states[state_key(m.qp_id, m.matid, m.replica)];
    if (displaced) *displaced = std::move(slot);
    slot = std::move(*incoming);
    if (created) *created = true;
    return SCEMA_MD_OK;
  }
  State *src = nullptr;
  if (m.qp_id != m.most_recent_qp_id) {
    src = find_state(e, m.most_recent_qp_id, m.matid, m.replica);
    if (m.most_recent_qp_id == SCEMA_MD_QP_NONE) {
      if (src) return fail(e, SCEMA_MD_ERR_NOSTATE, "state exists for the 'none' quadrature point id");
    } else if (!src)
      return fail(e, SCEMA_MD_ERR_NOSTATE, "no state last.%d.%s_%d to branch from", m.most_recent_qp_id, m.matid, m.replica);
  } else {
    src = find_state(e, m.qp_id, m.matid, m.replica);
  }
  State *dst = find_state(e, m.qp_id, m.matid, m.replica);
  if (src && src == dst) {
    *out = dst;
    return SCEMA_MD_OK;
  }
  std::unique_ptr<State> ns;
  int rc;
  if (src) {
    rc = make_state(e, t, src->box, src->x.as<double>(), src->v.as<double>(), true, ns);
    if (rc == SCEMA_MD_OK) ns->skin_extra = src->skin_extra;
  } else
    rc = make_state(e, t, t->init_box, t->init_x.data(), t->init_v.data(), false, ns);
  if (rc) return rc;
  *out = ns.get();
  auto &slot = e->states[state_key(m.qp_id, m.matid, m.replica)];
  if (displaced) *displaced = std::move(slot);
  slot = std::move(ns);
  if (created) *created = true;
  return SCEMA_MD_OK;
}

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

// which phases an evaluation runs and with which constraints (the hot path: strain with SHAKE, sample with SHAKE;
// the elastic-constant runs of init_material: both without SHAKE; its homogenisation run: sampling only)
struct EvalOpt {
  bool phase_a = true;
  int shake_a = 1, shake_b = 1;
};

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
    }
  else
    for (int i = 0; i < ns; i++) {
      std::memcpy(chunk[i].box0, chunk[i].st->box, sizeof chunk[i].box0);
      chunk[i].skin0 = chunk[i].st->skin_extra;
    }
  return SCEMA_MD_OK;
}

// full evaluation (phase A + phase B) of a chunk of simulations, with overflow retry
int eval_chunk(scema_md_engine *e, std::vector<ActiveSim> &chunk, const EvalOpt &opt = EvalOpt(), size_t pool_off = 0) {
  const int ns = (int)chunk.size();
  for (int attempt = 0; attempt < 6; attempt++) {
    int rc = prepare_slots(e, chunk);
    if (rc) return rc;
    // backup for a retry after neighbour overflow
    rc = backup_states(e, chunk, false, pool_off);
    if (rc) return rc;
    RunSpec A;
    A.deform = 1;
    A.use_shake = opt.shake_a;
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
      for (int i = 0; i < ns; i++) chunk[i].nsteps = chunk[i].nss;
      rc = run_phase(e, chunk, B);
      if (getenv("SCEMA_MD_TIMING")) fprintf(stderr, "[scema_md] chunk of %d: phase A %.1f ms, phase B %.1f ms (attempt %d)\n", ns, 1e3 * (t_a1 - t_a0), 1e3 * (wall_s() - t_a1), attempt);
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
    if (e->overflow_bits & 4) e->jtab_grow *= 1.25;
    if ((e->overflow_bits & 8) || !(e->overflow_bits & 4)) e->neigh_grow *= 1.5;
  }
  return fail(e, SCEMA_MD_ERR_OVERFLOW, "neighbour capacity exceeded after regrowth");
}

#define NCCLCHK(call)                                                                                         \
  do {                                                                                                    \
    ncclResult_t _r = (call);                                                                             \
    if (_r != ncclSuccess) return fail(e, SCEMA_MD_ERR_DEVICE, "%s failed: %s", #call, ncclGetErrorString(_r)); \
  } while (0)

// Replica states change GPU (scema::PlanMove): x, v and the box of the source state of simulation m.sim go from rank
// m.from to rank m.to, where they become the state that simulation continues from.  The source rank keeps its copy when
// another quadrature point branches from it (most_recent_qp_id != qp_id); a state that merely moved is dropped there after
// the update.  RCCL: one group of point-to-point sends/receives over xGMI on the engine's stream; host transport: the
// moves in plan order, blocking send/recv pairs (every rank walks the same list, so the pairs cannot cross).
constexpr int MIG_SIDE = 10;   // doubles that travel next to x and v of a migrating state: box[9], State::skin_extra
int migrate_states(scema_md_engine *e, const scema_mdsim *sims, const scema::SimPlan &plan, const std::vector<std::string> &src_keys,
                   std::map<int, std::unique_ptr<State>> &incoming) {
  Comm &c = e->comm;
  const int nm = (int)plan.moves.size();
  std::vector<double> hbox(MIG_SIDE * (size_t)nm, 0.0);   // box[9] + the state's list skin (State::skin_extra)
  std::vector<State *> src(nm, nullptr);
  for (int k = 0; k < nm; k++) {
    const scema::PlanMove &m = plan.moves[k];
    Topo *t = find_topo(e, sims[m.sim].matid, sims[m.sim].replica);
    if (!t) return fail(e, SCEMA_MD_ERR_NOSTATE, "replica %s_%d is not registered on rank %d", sims[m.sim].matid, sims[m.sim].replica, c.rank);
    if (c.rank == m.from) {
      auto it = e->states.find(src_keys[m.sim]);
      if (it == e->states.end())
        return fail(e, SCEMA_MD_ERR_NOSTATE, "rank %d is recorded as the owner of state %s but does not hold it", c.rank, src_keys[m.sim].c_str());
      src[k] = it->second.get();
      std::memcpy(&hbox[MIG_SIDE * (size_t)k], src[k]->box, 9 * sizeof(double));
      hbox[MIG_SIDE * (size_t)k + 9] = src[k]->skin_extra;
    }
    if (c.rank == m.to) {
      int rc = make_empty_state(e, t, incoming[m.sim]);
      if (rc) return rc;
    }
  }
  if (c.kind == 1) {
    HIPCHK(c.d_box.ensure(hbox.size() * sizeof(double)));
    HIPCHK(hipMemcpyAsync(c.d_box.p, hbox.data(), hbox.size() * sizeof(double), hipMemcpyHostToDevice, e->stream));
    NCCLCHK(ncclGroupStart());
    for (int k = 0; k < nm; k++) {
      const scema::PlanMove &m = plan.moves[k];
      double *dbox = c.d_box.as<double>() + MIG_SIDE * (size_t)k;
      if (c.rank == m.from) {
        const size_t cnt = 3 * (size_t)src[k]->topo->natoms;
        NCCLCHK(ncclSend(src[k]->x.p, cnt, ncclDouble, m.to, c.nccl, e->stream));
        NCCLCHK(ncclSend(src[k]->v.p, cnt, ncclDouble, m.to, c.nccl, e->stream));
        NCCLCHK(ncclSend(dbox, MIG_SIDE, ncclDouble, m.to, c.nccl, e->stream));
      }
      if (c.rank == m.to) {
        State *d = incoming[m.sim].get();
        const size_t cnt = 3 * (size_t)d->topo->natoms;
        NCCLCHK(ncclRecv(d->x.p, cnt, ncclDouble, m.from, c.nccl, e->stream));
        NCCLCHK(ncclRecv(d->v.p, cnt, ncclDouble, m.from, c.nccl, e->stream));
        NCCLCHK(ncclRecv(dbox, MIG_SIDE, ncclDouble, m.from, c.nccl, e->stream));
      }
    }
    NCCLCHK(ncclGroupEnd());
    HIPCHK(hipMemcpyAsync(hbox.data(), c.d_box.p, hbox.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int k = 0; k < nm; k++)
      if (c.rank == plan.moves[k].to) {
        std::memcpy(incoming[plan.moves[k].sim]->box, &hbox[MIG_SIDE * (size_t)k], 9 * sizeof(double));
        incoming[plan.moves[k].sim]->skin_extra = hbox[MIG_SIDE * (size_t)k + 9];
      }
  } else {
    if (!c.send || !c.recv) return fail(e, SCEMA_MD_ERR_ARG, "the host communicator has no send/recv callbacks: replica states cannot move between ranks");
    std::vector<double> buf;
    for (int k = 0; k < nm; k++) {
      const scema::PlanMove &m = plan.moves[k];
      if (c.rank != m.from && c.rank != m.to) continue;
      State *st = (c.rank == m.from) ? src[k] : incoming[m.sim].get();
      const size_t n3 = 3 * (size_t)st->topo->natoms;
      buf.resize(2 * n3 + MIG_SIDE);
      if (c.rank == m.from) {
        HIPCHK(hipMemcpyAsync(buf.data(), st->x.p, n3 * 8, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipMemcpyAsync(buf.data() + n3, st->v.p, n3 * 8, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        std::memcpy(buf.data() + 2 * n3, st->box, 9 * sizeof(double));
        buf[2 * n3 + 9] = st->skin_extra;
        if (c.send(c.ctx, buf.data(), (int64_t)(buf.size() * 8), m.to)) return fail(e, SCEMA_MD_ERR_DEVICE, "host send of a replica state to rank %d failed", m.to);
      } else {
        if (c.recv(c.ctx, buf.data(), (int64_t)(buf.size() * 8), m.from)) return fail(e, SCEMA_MD_ERR_DEVICE, "host receive of a replica state from rank %d failed", m.from);
        HIPCHK(hipMemcpyAsync(st->x.p, buf.data(), n3 * 8, hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(st->v.p, buf.data() + n3, n3 * 8, hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        std::memcpy(st->box, buf.data() + 2 * n3, 9 * sizeof(double));
        st->skin_extra = buf[2 * n3 + 9];
      }
    }
  }
  c.migrations += nm;
  return SCEMA_MD_OK;
}

// Result buffer of a rank: 6*cap stresses followed by SCEMA_MD_RESULT_TRAILER words -- [status of this rank's share
// (0 = fine, else the error code), hash of the plan this rank computed].  Every rank enters the collective whatever
// happened to its share, so that a rank-local failure (list overflow, a replica that blew up, a missing state) ends the
// update on ALL ranks together instead of leaving the others blocked in the collective.
static_assert(SCEMA_MD_RESULT_TRAILER == 2, "result trailer: status, plan hash");

// 52 bits of an FNV-1a hash: exactly representable in the double it travels in
double plan_hash(const scema::SimPlan &P, const std::vector<double> &cost) {
  unsigned long long h = 1469598103934665603ull;
  auto mix = [&](unsigned long long v) {
    for (int k = 0; k < 8; k++) { h ^= (v >> (8 * k)) & 0xffull; h *= 1099511628211ull; }
  };
  mix((unsigned long long)P.world); mix((unsigned long long)P.cap); mix(P.owner.size());
  for (size_t i = 0; i < P.owner.size(); i++) {
    mix((unsigned long long)P.owner[i]); mix((unsigned long long)P.pos[i]); mix((unsigned long long)(long long)P.home[i]);
    unsigned long long bits; std::memcpy(&bits, &cost[i], 8); mix(bits);
  }
  mix(P.moves.size());
  for (const scema::PlanMove &m : P.moves) { mix((unsigned long long)m.sim); mix((unsigned long long)m.from); mix((unsigned long long)m.to); }
  return (double)(h & ((1ull << 52) - 1));
}

// every rank contributes cnt doubles (host memory), out = world * cnt
int comm_allgather(scema_md_engine *e, const double *local, size_t cnt, std::vector<double> &out, DevBuf &d_send) {
  Comm &c = e->comm;
  out.assign(cnt * c.world, 0.0);
  if (c.kind == 1) {
    HIPCHK(d_send.ensure(cnt * sizeof(double)));
    HIPCHK(c.d_gather.ensure(cnt * c.world * sizeof(double)));
    HIPCHK(hipMemcpyAsync(d_send.p, local, cnt * sizeof(double), hipMemcpyHostToDevice, e->stream));
    NCCLCHK(ncclAllGather(d_send.p, c.d_gather.p, cnt, ncclDouble, c.nccl, e->stream));
    HIPCHK(hipMemcpyAsync(out.data(), c.d_gather.p, cnt * c.world * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
  } else {
    if (!c.ag || c.ag(c.ctx, local, out.data(), (int64_t)(cnt * sizeof(double))))
      return fail(e, SCEMA_MD_ERR_DEVICE, "host all-gather failed");
  }
  return SCEMA_MD_OK;
}

// what the ranks told each other: the first failing rank's code, or a plan mismatch
int check_gathered_trailers(scema_md_engine *e, const double *gathered, size_t stride, size_t off, int world, int rank, const char *when) {
  for (int r = 0; r < world; r++) {
    const int st = (int)gathered[r * stride + off];
    if (st != 0) {
      if (r == rank) return st;   // this rank's own message is already in e->err
      return fail(e, st, "rank %d failed %s (code %d): the update is abandoned on every rank", r, when, st);
    }
  }
  for (int r = 1; r < world; r++)
    if (gathered[r * stride + off + 1] != gathered[off + 1])
      return fail(e, SCEMA_MD_ERR_ARG, "ranks 0 and %d computed different plans for this update: replicas, states and request vectors must be the same on every rank "
                  "(scema_md_register_replica / set_state / drop_state / load_state_file / equilibrate are collective when a world > 1 is used)", r);
  return SCEMA_MD_OK;
}

// Agreement before anything moves: 2 doubles per rank (status of the local pre-checks, plan hash).  A plan that differs
// between ranks would pair sends with no receive, or give the all-gather different counts.
int handshake(scema_md_engine *e, int local_status, double hash) {
  Comm &c = e->comm;
  const double word[2] = {(double)local_status, hash};
  std::vector<double> all;
  int rc = comm_allgather(e, word, 2, all, c.d_word);
  if (rc) return rc;
  c.handshakes += 1;
  return check_gathered_trailers(e, all.data(), 2, 0, c.world, c.rank, "before the update started");
}

// ONE all-gather of 6*cap (+ trailer) doubles per rank, then every rank fills every sims[i].stress (all ranks hold all
// stresses, so the second share_scale_bridging_data broadcast of the caller, dealammps.cc:458, is not needed).
int allgather_stresses(scema_md_engine *e, const std::vector<double> &local, scema_mdsim *sims, int n_sims) {
  Comm &c = e->comm;
  const scema::SimPlan &plan = e->last_plan;
  const size_t cnt = local.size();
  int rc = comm_allgather(e, local.data(), cnt, c.h_gather, e->d_local_stress);
  if (rc) return rc;
  c.allgathers += 1;
  rc = check_gathered_trailers(e, c.h_gather.data(), cnt, cnt - SCEMA_MD_RESULT_TRAILER, c.world, c.rank, "during the update");
  if (rc) return rc;
  for (int i = 0; i < n_sims; i++) {
    const double *src = c.h_gather.data() + ((size_t)plan.owner[i] * cnt + 6 * (size_t)plan.pos[i]);
    for (int k = 0; k < 6; k++) sims[i].stress[k] = src[k];
    sims[i].stress_updated = 1;
  }
  return SCEMA_MD_OK;
}

}  // namespace

// ===========================================================================================
// C ABI
// ===========================================================================================
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

// ---- replica container file (our stand-in for the LAMMPS binary restart init.<mat>_<rep>.bin) ----
static const char REPL_MAGIC[8] = {'S', 'C', 'E', 'M', 'A', 'M', 'D', '1'};
static const char STATE_MAGIC[8] = {'S', 'C', 'E', 'M', 'A', 'S', 'T', '1'};

int scema_md_write_replica_file(const char *path, const scema_md_system *s) {
  FILE *fp = fopen(path, "wb");
  if (!fp) return SCEMA_MD_ERR_IO;
  int32_t hdr[10] = {s->natoms, s->ntypes, s->nbonds, s->nbondtypes, s->nangles, s->nangletypes, s->ndihedrals, s->ndihedraltypes, s->nimpropers, s->nimpropertypes};
  bool ok = fwrite(REPL_MAGIC, 1, 8, fp) == 8 && fwrite(hdr, 4, 10, fp) == 10;
  auto W = [&](const void *p, size_t sz, size_t n) { if (ok && n) ok = fwrite(p, sz, n, fp) == n; };
  W(s->special_lj, 8, 3); W(s->special_coul, 8, 3); W(s->box, 8, 9);
  W(s->type, 4, s->natoms); W(s->charge, 8, s->natoms); W(s->mass, 8, s->ntypes);
  W(s->eps, 8, (size_t)s->ntypes * s->ntypes); W(s->sigma, 8, (size_t)s->ntypes * s->ntypes);
  W(s->bond_atoms, 4, 2 * (size_t)s->nbonds); W(s->bond_type, 4, s->nbonds); W(s->bond_coeff, 8, 2 * (size_t)s->nbondtypes);
  W(s->angle_atoms, 4, 3 * (size_t)s->nangles); W(s->angle_type, 4, s->nangles); W(s->angle_coeff, 8, 2 * (size_t)s->nangletypes);
  W(s->dihedral_atoms, 4, 4 * (size_t)s->ndihedrals); W(s->dihedral_type, 4, s->ndihedrals); W(s->dihedral_coeff, 8, 4 * (size_t)s->ndihedraltypes);
  W(s->improper_atoms, 4, 4 * (size_t)s->nimpropers); W(s->improper_type, 4, s->nimpropers); W(s->improper_coeff, 8, 2 * (size_t)s->nimpropertypes);
  W(s->x, 8, 3 * (size_t)s->natoms); W(s->v, 8, 3 * (size_t)s->natoms);
  fclose(fp);
  return ok ? SCEMA_MD_OK : SCEMA_MD_ERR_IO;
}

int scema_md_load_replica_file(scema_md_engine *e, const char *matid, int32_t replica, const char *path) {
  if (!e || !path) return SCEMA_MD_ERR_ARG;
  FILE *fp = fopen(path, "rb");
  if (!fp) return fail(e, SCEMA_MD_ERR_IO, "cannot open %s", path);
  char magic[16] = {0};
  if (fread(magic, 1, 8, fp) != 8) { fclose(fp); return fail(e, SCEMA_MD_ERR_IO, "short file %s", path); }
  if (std::memcmp(magic, REPL_MAGIC, 8) != 0) {
    fclose(fp);
    if (std::memcmp(magic, "LammpS R", 8) == 0)
      return fail(e, SCEMA_MD_ERR_IO, "%s is a LAMMPS binary restart; convert it with write_data and scema_amd.lammps_data (SURVEY row f-1)", path);
    return fail(e, SCEMA_MD_ERR_IO, "%s: unknown replica file format", path);
  }
  int32_t h[10];
  bool ok = fread(h, 4, 10, fp) == 10;
  scema_md_system s;
  std::memset(&s, 0, sizeof s);
  std::vector<int32_t> type, ba, bt, aa, at, da, dt, ia, it;
  std::vector<double> q, mass, eps, sig, bc, ac, dc, ic, x, v;
  auto R = [&](void *p, size_t sz, size_t n) { if (ok && n) ok = fread(p, sz, n, fp) == n; };
  if (ok) {
    s.natoms = h[0]; s.ntypes = h[1]; s.nbonds = h[2]; s.nbondtypes = h[3]; s.nangles = h[4]; s.nangletypes = h[5];
    s.ndihedrals = h[6]; s.ndihedraltypes = h[7]; s.nimpropers = h[8]; s.nimpropertypes = h[9];
    for (int k = 0; k < 10; k++) if (h[k] < 0) ok = false;
  }
  if (ok) {
    R(s.special_lj, 8, 3); R(s.special_coul, 8, 3); R(s.box, 8, 9);
    type.resize(s.natoms); q.resize(s.natoms); mass.resize(s.ntypes); eps.resize((size_t)s.ntypes * s.ntypes); sig.resize(eps.size());
    ba.resize(2 * (size_t)s.nbonds); bt.resize(s.nbonds); bc.resize(2 * (size_t)s.nbondtypes);
    aa.resize(3 * (size_t)s.nangles); at.resize(s.nangles); ac.resize(2 * (size_t)s.nangletypes);
    da.resize(4 * (size_t)s.ndihedrals); dt.resize(s.ndihedrals); dc.resize(4 * (size_t)s.ndihedraltypes);
    ia.resize(4 * (size_t)s.nimpropers); it.resize(s.nimpropers); ic.resize(2 * (size_t)s.nimpropertypes);
    x.resize(3 * (size_t)s.natoms); v.resize(x.size());
    R(type.data(), 4, type.size()); R(q.data(), 8, q.size()); R(mass.data(), 8, mass.size()); R(eps.data(), 8, eps.size()); R(sig.data(), 8, sig.size());
    R(ba.data(), 4, ba.size()); R(bt.data(), 4, bt.size()); R(bc.data(), 8, bc.size());
    R(aa.data(), 4, aa.size()); R(at.data(), 4, at.size()); R(ac.data(), 8, ac.size());
    R(da.data(), 4, da.size()); R(dt.data(), 4, dt.size()); R(dc.data(), 8, dc.size());
    R(ia.data(), 4, ia.size()); R(it.data(), 4, it.size()); R(ic.data(), 8, ic.size());
    R(x.data(), 8, x.size()); R(v.data(), 8, v.size());
  }
  fclose(fp);
  if (!ok) return fail(e, SCEMA_MD_ERR_IO, "truncated or corrupt replica file %s", path);
  s.type = type.data(); s.charge = q.data(); s.mass = mass.data(); s.eps = eps.data(); s.sigma = sig.data();
  s.bond_atoms = ba.data(); s.bond_type = bt.data(); s.bond_coeff = bc.data();
  s.angle_atoms = aa.data(); s.angle_type = at.data(); s.angle_coeff = ac.data();
  s.dihedral_atoms = da.data(); s.dihedral_type = dt.data(); s.dihedral_coeff = dc.data();
  s.improper_atoms = ia.data(); s.improper_type = it.data(); s.improper_coeff = ic.data();
  s.x = x.data(); s.v = v.data();
  for (int b = 0; b < s.nbonds; b++) if (bt[b] < 0 || bt[b] >= s.nbondtypes) return fail(e, SCEMA_MD_ERR_IO, "bad bond type in %s", path);
  for (int b = 0; b < s.nangles; b++) if (at[b] < 0 || at[b] >= s.nangletypes) return fail(e, SCEMA_MD_ERR_IO, "bad angle type in %s", path);
  for (int b = 0; b < s.ndihedrals; b++) if (dt[b] < 0 || dt[b] >= s.ndihedraltypes) return fail(e, SCEMA_MD_ERR_IO, "bad dihedral type in %s", path);
  for (int b = 0; b < s.nimpropers; b++) if (it[b] < 0 || it[b] >= s.nimpropertypes) return fail(e, SCEMA_MD_ERR_IO, "bad improper type in %s", path);
  return scema_md_register_replica(e, matid, replica, &s);
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
  HIPCHK(hipSetDevice(e->p.device));
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
    if (rc_cfg) return rc_cfg;
  }
  // ---- who runs what (host/sim_plan.h): identical on every rank ----
  int pre_status = SCEMA_MD_OK;   // rank-local findings before anything runs; exchanged in the handshake
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
  const scema::SimPlan &plan = e->last_plan;
  const int per_rank = plan.cap;
  const double hash = plan_hash(plan, cost);
  e->local_stress_count = per_rank;
  const size_t nres = 6 * (size_t)std::max(per_rank, 1) + SCEMA_MD_RESULT_TRAILER;
  HIPCHK(e->d_local_stress.ensure(nres * sizeof(double)));
  std::vector<double> local(nres, 0.0);
  local[nres - 1] = hash;
  const bool collective = e->comm.kind && world > 1;
  // without a communicator the caller gathers this buffer: it must say what happened to this rank's share whenever a plan exists
  auto publish = [&](int st) {
    local[nres - 2] = (double)st;
    (void)hipMemcpyAsync(e->d_local_stress.p, local.data(), local.size() * sizeof(double), hipMemcpyHostToDevice, e->stream);
    (void)hipStreamSynchronize(e->stream);
    return st;
  };
  // a source state this rank is recorded to own but does not hold: found before any rank posts a receive for it
  for (const scema::PlanMove &m : plan.moves)
    if (m.from == rank && !pre_status && !e->states.count(src_keys[m.sim]))
      pre_status = fail(e, SCEMA_MD_ERR_NOSTATE, "rank %d is recorded as the owner of state %s but does not hold it", rank, src_keys[m.sim].c_str());
  if (collective) {
    const int rc = handshake(e, pre_status, hash);
    if (rc) return rc;
  } else if (pre_status)
    return publish(pre_status);
  // ---- states that have to change GPU first ----
  std::map<int, std::unique_ptr<State>> incoming;
  if (!hooke_mode && !plan.moves.empty()) {
    if (!e->comm.kind) {
      const scema::PlanMove &m = plan.moves[0];
      return publish(fail(e, SCEMA_MD_ERR_NOSTATE, "the state %s that quadrature point %d continues from lives on rank %d but the simulation is planned on rank %d: "
                          "attach a communicator (scema_md_comm_init_rccl / scema_md_comm_init_host) so that states can move between GPUs",
                          src_keys[m.sim].c_str(), sims[m.sim].qp_id, m.from, m.to));
    }
    int rc = migrate_states(e, sims, plan, src_keys, incoming);
    if (rc) return rc;   // a transport failure: nothing a status word could repair
  }
  // ---- this rank's share ----
  std::vector<ActiveSim> act;
  struct Created { std::string key; std::unique_ptr<State> displaced; };
  std::vector<Created> created;
  auto undo = [&]() {   // a failed update leaves the state store as it found it (the reference stops before write_restart)
    for (auto it = created.rbegin(); it != created.rend(); ++it) {
      if (it->displaced) e->states[it->key] = std::move(it->displaced);
      else e->states.erase(it->key);
    }
    created.clear();
  };
  int status = SCEMA_MD_OK;   // of this rank's share; with a communicator it travels in the trailer of the all-gather
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
    e->dir.commit(plan, dst_keys);
    for (int i = 0; i < n_sims; i++)
      if (plan.owner[i] != rank) e->states.erase(dst_keys[i]);
  }
  return SCEMA_MD_OK;
}

int scema_md_strain(scema_md_engine *e, scema_mdsim *sim, int32_t hooke_mode) { return scema_md_strain_batch(e, sim, 1, hooke_mode, 0, 1); }

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
  if (rc) return rc;
  for (int i = 0; i < n_sims; i++) {
    const double *src = gathered + ((size_t)plan.owner[i] * cnt + 6 * (size_t)plan.pos[i]);
    for (int k = 0; k < 6; k++) sims[i].stress[k] = src[k];
    sims[i].stress_updated = 1;
  }
  return SCEMA_MD_OK;
}

// ---- communicator (one process per GPU) ----
int scema_md_comm_unique_id(void *id) {
  if (!id) return SCEMA_MD_ERR_ARG;
  static_assert(SCEMA_MD_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
  ncclUniqueId u;
  if (ncclGetUniqueId(&u) != ncclSuccess) return SCEMA_MD_ERR_DEVICE;
  std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  return SCEMA_MD_OK;
}

int scema_md_comm_init_rccl(scema_md_engine *e, const void *id, int32_t rank, int32_t world) {
  if (!e || !id || world <= 0 || rank < 0 || rank >= world) return fail(e, SCEMA_MD_ERR_ARG, "bad arguments");
  if (e->comm.kind) return fail(e, SCEMA_MD_ERR_ARG, "a communicator is already attached");
  HIPCHK(hipSetDevice(e->p.device));
  ncclUniqueId u;
  std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
  NCCLCHK(ncclCommInitRank(&e->comm.nccl, world, u, rank));
  e->comm.kind = 1;
  e->comm.rank = rank;
  e->comm.world = world;
  return SCEMA_MD_OK;
}

int scema_md_comm_init_host(scema_md_engine *e, int32_t rank, int32_t world, scema_md_host_allgather_fn allgather, scema_md_host_send_fn send,
                            scema_md_host_recv_fn recv, void *ctx) {
  if (!e || !allgather || world <= 0 || rank < 0 || rank >= world) return fail(e, SCEMA_MD_ERR_ARG, "bad arguments");
  if (e->comm.kind) return fail(e, SCEMA_MD_ERR_ARG, "a communicator is already attached");
  e->comm.kind = 2;
  e->comm.rank = rank;
  e->comm.world = world;
  e->comm.ag = allgather;
  e->comm.send = send;
  e->comm.recv = recv;
  e->comm.ctx = ctx;
  return SCEMA_MD_OK;
}

void scema_md_comm_destroy(scema_md_engine *e) {
  if (!e || !e->comm.kind) return;
  (void)hipSetDevice(e->p.device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  if (e->comm.kind == 1 && e->comm.nccl) (void)ncclCommDestroy(e->comm.nccl);
  e->comm.nccl = nullptr;
  e->comm.kind = 0;
  e->comm.rank = 0;
  e->comm.world = 1;
  e->comm.ag = nullptr; e->comm.send = nullptr; e->comm.recv = nullptr; e->comm.ctx = nullptr;
}

int32_t scema_md_comm_world(const scema_md_engine *e) { return (e && e->comm.kind) ? e->comm.world : 1; }
int32_t scema_md_comm_rank(const scema_md_engine *e) { return (e && e->comm.kind) ? e->comm.rank : 0; }

int64_t scema_md_comm_handshakes(const scema_md_engine *e) { return e ? e->comm.handshakes : 0; }

int scema_md_comm_stats(const scema_md_engine *e, int64_t *allgathers, int64_t *migrations) {
  if (!e) return SCEMA_MD_ERR_ARG;
  if (allgathers) *allgathers = e->comm.allgathers;
  if (migrations) *migrations = e->comm.migrations;
  return SCEMA_MD_OK;
}

// the recorded owner of a state: rank, or -1 when no rank is recorded (the state, if it exists, is held locally)
int32_t scema_md_state_owner(const scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica) {
  return e ? e->dir.owner_of(state_key(qp_id, matid, replica)) : -1;
}

// ---- the planner alone: pure host arithmetic (host/sim_plan.h), no GPU needed ----
struct scema_plan_dir {
  scema::OwnerDirectory dir;
};
scema_plan_dir *scema_plan_dir_create(void) { return new scema_plan_dir(); }
void scema_plan_dir_destroy(scema_plan_dir *d) { delete d; }
int scema_plan_update(scema_plan_dir *d, const scema_mdsim *sims, int32_t n_sims, const double *cost, int32_t world, int32_t *owner, int32_t *pos,
                      int32_t *cap, int32_t *moves, int32_t *n_moves, int32_t commit) {
  if (!d || (!sims && n_sims > 0) || n_sims < 0 || world <= 0) return SCEMA_MD_ERR_ARG;
  std::vector<std::string> src(n_sims), dst(n_sims);
  std::vector<double> c(n_sims, 1.0);
  for (int i = 0; i < n_sims; i++) {
    dst[i] = state_key(sims[i].qp_id, sims[i].matid, sims[i].replica);
    if (sims[i].most_recent_qp_id == sims[i].qp_id) src[i] = dst[i];
    else if (sims[i].most_recent_qp_id != SCEMA_MD_QP_NONE) src[i] = state_key(sims[i].most_recent_qp_id, sims[i].matid, sims[i].replica);
    if (cost) c[i] = cost[i];
  }
  const scema::SimPlan P = d->dir.plan(src, dst, c, world);
  for (int i = 0; i < n_sims; i++) {
    if (owner) owner[i] = P.owner[i];
    if (pos) pos[i] = P.pos[i];
  }
  if (cap) *cap = P.cap;
  if (n_moves) *n_moves = (int32_t)P.moves.size();
  if (moves)
    for (size_t k = 0; k < P.moves.size(); k++) { moves[3 * k] = P.moves[k].sim; moves[3 * k + 1] = P.moves[k].from; moves[3 * k + 2] = P.moves[k].to; }
  if (commit) d->dir.commit(P, dst);
  return SCEMA_MD_OK;
}

// ---- state management ----
int scema_md_has_state(const scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica) {
  if (!e) return 0;
  return e->states.count(state_key(qp_id, matid, replica)) ? 1 : 0;
}

int scema_md_save_replica_file(scema_md_engine *e, const char *matid, int32_t replica, const char *path) {
  if (!e || !matid || !path) return SCEMA_MD_ERR_ARG;
  Topo *t = find_topo(e, matid, replica);
  if (!t) return fail(e, SCEMA_MD_ERR_NOSTATE, "replica %s_%d not registered", matid, (int)replica);
  scema_md_system s = t->original.sys;
  std::memcpy(s.box, t->init_box, sizeof s.box);
  s.x = t->init_x.data();
  s.v = t->init_v.data();
  const int rc = scema_md_write_replica_file(path, &s);
  return rc ? fail(e, rc, "cannot write %s", path) : SCEMA_MD_OK;
}

int32_t scema_md_replica_natoms(scema_md_engine *e, const char *matid, int32_t replica) {
  if (!e || !matid) return 0;
  Topo *t = find_topo(e, matid, replica);
  return t ? t->natoms : 0;
}

int scema_md_get_state(scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica, double box[9], double *x, double *v) {
  if (!e) return SCEMA_MD_ERR_ARG;
  HIPCHK(hipSetDevice(e->p.device));
  Topo *t = find_topo(e, matid, replica);
  if (!t) return fail(e, SCEMA_MD_ERR_NOSTATE, "replica %s_%d not registered", matid, replica);
  const size_t bytes = 3 * (size_t)t->natoms * 8;
  if (qp_id == SCEMA_MD_QP_NONE) {
    if (box) std::memcpy(box, t->init_box, 9 * 8);
    if (x) std::memcpy(x, t->init_x.data(), bytes);
    if (v) std::memcpy(v, t->init_v.data(), bytes);
    return SCEMA_MD_OK;
  }
  State *s = find_state(e, qp_id, matid, replica);
  if (!s) return fail(e, SCEMA_MD_ERR_NOSTATE, "no state for qp %d %s_%d", qp_id, matid, replica);
  if (box) std::memcpy(box, s->box, 9 * 8);
  if (x) HIPCHK(hipMemcpy(x, s->x.p, bytes, hipMemcpyDeviceToHost));
  if (v) HIPCHK(hipMemcpy(v, s->v.p, bytes, hipMemcpyDeviceToHost));
  return SCEMA_MD_OK;
}

int scema_md_set_state(scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica, const double box[9], const double *x, const double *v) {
  if (!e || !box || !x || !v) return SCEMA_MD_ERR_ARG;
  HIPCHK(hipSetDevice(e->p.device));
  Topo *t = find_topo(e, matid, replica);
  if (!t) return fail(e, SCEMA_MD_ERR_NOSTATE, "replica %s_%d not registered", matid, replica);
  std::unique_ptr<State> ns;
  int rc = make_state(e, t, box, x, v, false, ns);
  if (rc) return rc;
  e->states[state_key(qp_id, matid, replica)] = std::move(ns);
  // a state handed over by the host is taken as present wherever it was handed over (every rank reads the same lcts.*
  // files, stmd_sync.h:167-187): no rank is recorded as its only owner
  e->dir.erase(state_key(qp_id, matid, replica));
  return SCEMA_MD_OK;
}

int scema_md_drop_state(scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica) {
  if (!e) return SCEMA_MD_ERR_ARG;
  (void)hipSetDevice(e->p.device);
  e->states.erase(state_key(qp_id, matid, replica));
  e->dir.erase(state_key(qp_id, matid, replica));
  return SCEMA_MD_OK;
}

int scema_md_save_state_file(scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica, const char *path) {
  if (!e || !path) return SCEMA_MD_ERR_ARG;
  Topo *t = find_topo(e, matid, replica);
  if (!t) return fail(e, SCEMA_MD_ERR_NOSTATE, "replica %s_%d not registered", matid, replica);
  std::vector<double> x(3 * (size_t)t->natoms), v(x.size());
  double box[9];
  int rc = scema_md_get_state(e, qp_id, matid, replica, box, x.data(), v.data());
  if (rc) return rc;
  FILE *fp = fopen(path, "wb");
  if (!fp) return fail(e, SCEMA_MD_ERR_IO, "cannot write %s", path);
  int32_t n = t->natoms;
  bool ok = fwrite(STATE_MAGIC, 1, 8, fp) == 8 && fwrite(&n, 4, 1, fp) == 1 && fwrite(box, 8, 9, fp) == 9 &&
            fwrite(x.data(), 8, x.size(), fp) == x.size() && fwrite(v.data(), 8, v.size(), fp) == v.size();
  fclose(fp);
  return ok ? SCEMA_MD_OK : fail(e, SCEMA_MD_ERR_IO, "short write %s", path);
}

// ---- LAMMPS text dumps (dump custom ... id type xs ys zs vx vy vz ix iy iz): the state files of the reference's reax branch ----
static int load_state_dump(scema_md_engine *e, Topo *t, int32_t qp_id, const char *matid, int32_t replica, const char *path) {
  std::ifstream in(path);
  if (!in) return fail(e, SCEMA_MD_ERR_IO, "cannot open %s", path);
  std::string line;
  long long natoms = -1;
  double box[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  bool have_box = false, have_atoms = false;
  std::vector<double> x(3 * (size_t)t->natoms), v(x.size(), 0.0);
  std::vector<char> seen(t->natoms, 0);
  while (std::getline(in, line)) {
    if (line.rfind("ITEM: TIMESTEP", 0) == 0) {
      std::getline(in, line);
    } else if (line.rfind("ITEM: NUMBER OF ATOMS", 0) == 0) {
      std::getline(in, line);
      natoms = atoll(line.c_str());
      if (natoms != t->natoms) return fail(e, SCEMA_MD_ERR_IO, "%s holds %lld atoms, replica %s_%d has %d", path, natoms, matid, (int)replica, t->natoms);
    } else if (line.rfind("ITEM: BOX BOUNDS", 0) == 0) {
      const bool tri = line.find("xy xz yz") != std::string::npos;
      double b[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
      for (int d = 0; d < 3; d++) {
        std::getline(in, line);
        const int got = sscanf(line.c_str(), "%lf %lf %lf", &b[d][0], &b[d][1], &b[d][2]);
        if (got < (tri ? 3 : 2)) return fail(e, SCEMA_MD_ERR_IO, "%s: bad box bounds", path);
      }
      const double xy = tri ? b[0][2] : 0.0, xz = tri ? b[1][2] : 0.0, yz = tri ? b[2][2] : 0.0;
      // the bounds of a triclinic box are those of its bounding box
      box[0] = b[0][0] - std::min(std::min(0.0, xy), std::min(xz, xy + xz));
      box[3] = b[0][1] - std::max(std::max(0.0, xy), std::max(xz, xy + xz));
      box[1] = b[1][0] - std::min(0.0, yz);
      box[4] = b[1][1] - std::max(0.0, yz);
      box[2] = b[2][0];
      box[5] = b[2][1];
      box[6] = xy; box[7] = xz; box[8] = yz;
      have_box = true;
    } else if (line.rfind("ITEM: ATOMS", 0) == 0) {
      if (!have_box || natoms < 0) return fail(e, SCEMA_MD_ERR_IO, "%s: atoms before box or count", path);
      // columns by name
      std::vector<std::string> cols;
      {
        std::istringstream hs(line.substr(11));
        std::string c;
        while (hs >> c) cols.push_back(c);
      }
      auto col = [&](const char *name) { for (size_t k = 0; k < cols.size(); k++) if (cols[k] == name) return (int)k; return -1; };
      const int cid = col("id"), cxs = col("xs"), cys = col("ys"), czs = col("zs"), cx = col("x"), cy = col("y"), cz = col("z");
      const int cvx = col("vx"), cvy = col("vy"), cvz = col("vz"), cix = col("ix"), ciy = col("iy"), ciz = col("iz");
      const bool scaled = cxs >= 0 && cys >= 0 && czs >= 0;
      if (cid < 0 || (!scaled && (cx < 0 || cy < 0 || cz < 0))) return fail(e, SCEMA_MD_ERR_IO, "%s: the dump needs id and xs ys zs (or x y z)", path);
      const double hx = box[3] - box[0], hy = box[4] - box[1], hz = box[5] - box[2];
      std::vector<double> f(cols.size());
      for (long long r = 0; r < natoms; r++) {
        if (!std::getline(in, line)) return fail(e, SCEMA_MD_ERR_IO, "%s: %lld atom lines expected, %lld found", path, natoms, r);
        std::istringstream ls(line);
        for (size_t k = 0; k < cols.size(); k++)
          if (!(ls >> f[k])) return fail(e, SCEMA_MD_ERR_IO, "%s: short atom line %lld", path, r + 1);
        const long long a = (long long)f[cid] - 1;
        if (a < 0 || a >= t->natoms || seen[a]) return fail(e, SCEMA_MD_ERR_IO, "%s: atom ids are not a permutation of 1..%d", path, t->natoms);
        seen[a] = 1;
        const double i0 = cix >= 0 ? f[cix] : 0.0, i1 = ciy >= 0 ? f[ciy] : 0.0, i2 = ciz >= 0 ? f[ciz] : 0.0;
        if (scaled) {   // lamda coordinates + image counts -> unwrapped Cartesian (states are kept unwrapped)
          const double l0 = f[cxs] + i0, l1 = f[cys] + i1, l2 = f[czs] + i2;
          x[3 * a] = box[0] + hx * l0 + box[6] * l1 + box[7] * l2;
          x[3 * a + 1] = box[1] + hy * l1 + box[8] * l2;
          x[3 * a + 2] = box[2] + hz * l2;
        } else {
          x[3 * a] = f[cx] + hx * i0 + box[6] * i1 + box[7] * i2;
          x[3 * a + 1] = f[cy] + hy * i1 + box[8] * i2;
          x[3 * a + 2] = f[cz] + hz * i2;
        }
        if (cvx >= 0 && cvy >= 0 && cvz >= 0) { v[3 * a] = f[cvx]; v[3 * a + 1] = f[cvy]; v[3 * a + 2] = f[cvz]; }
      }
      have_atoms = true;
      break;   // one snapshot
    }
  }
  if (!have_atoms) return fail(e, SCEMA_MD_ERR_IO, "%s holds no ITEM: ATOMS section", path);
  return scema_md_set_state(e, qp_id, matid, replica, box, x.data(), v.data());
}

// precise != 0: 17 significant digits (a round trip through the file is exact); 0: LAMMPS' default dump format "%g" (what the
// reference's files hold: six significant digits)
int scema_md_save_state_dump(scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica, const char *path, int64_t ntimestep,
                             int32_t precise) {
  if (!e || !path) return SCEMA_MD_ERR_ARG;
  Topo *t = find_topo(e, matid, replica);
  if (!t) return fail(e, SCEMA_MD_ERR_NOSTATE, "replica %s_%d not registered", matid, (int)replica);
  std::vector<double> x(3 * (size_t)t->natoms), v(x.size());
  double box[9];
  int rc = scema_md_get_state(e, qp_id, matid, replica, box, x.data(), v.data());
  if (rc) return rc;
  FILE *fp = fopen(path, "w");
  if (!fp) return fail(e, SCEMA_MD_ERR_IO, "cannot write %s", path);
  const double xy = box[6], xz = box[7], yz = box[8];
  const double hx = box[3] - box[0], hy = box[4] - box[1], hz = box[5] - box[2];
  fprintf(fp, "ITEM: TIMESTEP\n%lld\nITEM: NUMBER OF ATOMS\n%d\n", (long long)ntimestep, t->natoms);
  fprintf(fp, "ITEM: BOX BOUNDS xy xz yz pp pp pp\n");
  fprintf(fp, "%-1.16e %-1.16e %-1.16e\n", box[0] + std::min(std::min(0.0, xy), std::min(xz, xy + xz)), box[3] + std::max(std::max(0.0, xy), std::max(xz, xy + xz)), xy);
  fprintf(fp, "%-1.16e %-1.16e %-1.16e\n", box[1] + std::min(0.0, yz), box[4] + std::max(0.0, yz), xz);
  fprintf(fp, "%-1.16e %-1.16e %-1.16e\n", box[2], box[5], yz);
  fprintf(fp, "ITEM: ATOMS id type xs ys zs vx vy vz ix iy iz\n");
  const char *fmt = precise ? "%d %d %.17g %.17g %.17g %.17g %.17g %.17g %d %d %d\n" : "%d %d %g %g %g %g %g %g %d %d %d\n";
  for (int i = 0; i < t->natoms; i++) {
    const double d2 = x[3 * i + 2] - box[2], l2 = d2 / hz;
    const double d1 = x[3 * i + 1] - box[1] - yz * l2, l1 = d1 / hy;
    const double d0 = x[3 * i] - box[0] - xy * l1 - xz * l2, l0 = d0 / hx;
    const double w0 = std::floor(l0), w1 = std::floor(l1), w2 = std::floor(l2);
    fprintf(fp, fmt, i + 1, t->original.type[i] + 1, l0 - w0, l1 - w1, l2 - w2, v[3 * i], v[3 * i + 1], v[3 * i + 2], (int)w0, (int)w1, (int)w2);
  }
  const bool ok = fclose(fp) == 0;
  return ok ? SCEMA_MD_OK : fail(e, SCEMA_MD_ERR_IO, "short write %s", path);
}

int scema_md_load_state_file(scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica, const char *path) {
  if (!e || !path) return SCEMA_MD_ERR_ARG;
  Topo *t = find_topo(e, matid, replica);
  if (!t) return fail(e, SCEMA_MD_ERR_NOSTATE, "replica %s_%d not registered", matid, replica);
  FILE *fp = fopen(path, "rb");
  if (!fp) return fail(e, SCEMA_MD_ERR_IO, "cannot open %s", path);
  char magic[16] = {0};
  int32_t n = 0;
  double box[9];
  std::vector<double> x(3 * (size_t)t->natoms), v(x.size());
  bool ok = fread(magic, 1, 8, fp) == 8;
  if (ok && std::memcmp(magic, "ITEM: TI", 8) == 0) {
    // a LAMMPS text dump, as the reax branch of the reference exchanges states (stmd_problem.h:190-194,261-264:
    // write_dump all custom <file> id type xs ys zs vx vy vz ix iy iz, read back by
    // rerun <file> dump x y z vx vy vz ix iy iz box yes scaled yes wrapped yes format native)
    fclose(fp);
    return load_state_dump(e, t, qp_id, matid, replica, path);
  }
  if (ok && std::memcmp(magic, "LammpS R", 8) == 0) {
    // a LAMMPS binary restart, as the reference writes last.<qp>.* / lcts.<qp>.* (stmd_problem.h:258,268): box and the
    // per-atom block; atoms are matched by tag (file order is whatever the writing processors had), positions are
    // unwrapped with the image flags (states are kept unwrapped here)
    fclose(fp);
    scema_lammps_restart_info info;
    if (scema_md_probe_lammps_restart(path, &info) != SCEMA_MD_OK) return fail(e, SCEMA_MD_ERR_IO, "%s: %s", path, info.error);
    if (info.natoms != t->natoms) return fail(e, SCEMA_MD_ERR_IO, "%s holds %lld atoms, replica %s_%d has %d", path, (long long)info.natoms, matid, replica, t->natoms);
    std::vector<int64_t> tag(t->natoms);
    std::vector<int32_t> image(3 * (size_t)t->natoms);
    std::vector<double> xf(x.size()), vf(x.size());
    if (scema_md_read_lammps_restart_atoms(path, t->natoms, tag.data(), nullptr, image.data(), xf.data(), vf.data()) != SCEMA_MD_OK)
      return fail(e, SCEMA_MD_ERR_IO, "%s: cannot read the per-atom block", path);
    std::memcpy(box, info.box, sizeof box);
    const double hx[3] = {box[3] - box[0], box[4] - box[1], box[5] - box[2]};
    std::vector<char> seen(t->natoms, 0);
    for (int i = 0; i < t->natoms; i++) {
      const int64_t a = tag[i] - 1;
      if (a < 0 || a >= t->natoms || seen[a]) return fail(e, SCEMA_MD_ERR_IO, "%s: atom tags are not a permutation of 1..%d", path, t->natoms);
      seen[a] = 1;
      const int *im = &image[3 * (size_t)i];
      x[3 * a] = xf[3 * (size_t)i] + hx[0] * im[0] + box[6] * im[1] + box[7] * im[2];
      x[3 * a + 1] = xf[3 * (size_t)i + 1] + hx[1] * im[1] + box[8] * im[2];
      x[3 * a + 2] = xf[3 * (size_t)i + 2] + hx[2] * im[2];
      for (int c = 0; c < 3; c++) v[3 * a + c] = vf[3 * (size_t)i + c];
    }
    return scema_md_set_state(e, qp_id, matid, replica, box, x.data(), v.data());
  }
  ok = ok && std::memcmp(magic, STATE_MAGIC, 8) == 0 && fread(&n, 4, 1, fp) == 1 && n == t->natoms && fread(box, 8, 9, fp) == 9 &&
       fread(x.data(), 8, x.size(), fp) == x.size() && fread(v.data(), 8, v.size(), fp) == v.size();
  fclose(fp);
  if (!ok) return fail(e, SCEMA_MD_ERR_IO, "%s is not a state file of %s_%d", path, matid, replica);
  return scema_md_set_state(e, qp_id, matid, replica, box, x.data(), v.data());
}

// The state of (qp, mat, rep) as a LAMMPS 17Nov16 binary restart: what stmd_problem.h:258 (last.<qp>.<mat>_<rep>.dump) and
// :268 (lcts.*) write, so that a LAMMPS-based SCEMa run can pick the simulation up (and the other way round, through
// scema_md_load_state_file).  Positions are written as stored (unwrapped, image flags 0): read_restart remaps them.
int scema_md_save_state_lammps(scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica, const char *path, double timestep,
                               int64_t ntimestep) {
  if (!e || !path) return SCEMA_MD_ERR_ARG;
  Topo *t = find_topo(e, matid, replica);
  if (!t) return fail(e, SCEMA_MD_ERR_NOSTATE, "replica %s_%d not registered", matid, replica);
  std::vector<double> x(3 * (size_t)t->natoms), v(x.size());
  scema_md_system s = t->original.sys;
  int rc = scema_md_get_state(e, qp_id, matid, replica, s.box, x.data(), v.data());
  if (rc) return rc;
  s.x = x.data();
  s.v = v.data();
  rc = scema_md_write_lammps_restart(path, &s, e->p.cut_lj, e->p.cut_coul, timestep, ntimestep);
  return rc ? fail(e, rc, "cannot write %s", path) : SCEMA_MD_OK;
}

// ---- parity / measurement hooks ----
// qp_id == SCEMA_MD_QP_NONE: a temporary copy of the registered init state (held by `tmp`)
static int debug_state(scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica, State **out, std::unique_ptr<State> &tmp) {
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

int scema_md_debug_compute(scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica, int32_t use_shake, double *f,
                           double *energies, double *virials, double *info) {
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

// ---- init_material: the equilibration schedule (lammps_scripts_opls/in.init.lammps:44-215) ----
namespace {
unsigned long long splitmix64(unsigned long long &st) {
  unsigned long long z = (st += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// velocity all create T seed rot yes dist gaussian: Gaussian, zero linear and angular momentum, rescaled to T on 3N-3 degrees
// of freedom.  LAMMPS' own random stream is not reproduced (any member of the ensemble serves; the schedule forgets it).
void velocity_create(const Topo &t, const std::vector<double> &x, double temperature, unsigned long long seed, std::vector<double> &v) {
  const int n = t.natoms;
  v.assign(3 * (size_t)n, 0.0);
  unsigned long long st = seed;
  for (int i = 0; i < n; i++)
    for (int k = 0; k < 3; k++) {
      const double u1 = ((double)(splitmix64(st) >> 11) + 0.5) / 9007199254740992.0, u2 = ((double)(splitmix64(st) >> 11) + 0.5) / 9007199254740992.0;
      v[3 * i + k] = std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * MD_PI * u2) / std::sqrt(t.mass_atom[i]);
    }
  double p[3] = {0, 0, 0}, mt = 0.0, cm[3] = {0, 0, 0};
  for (int i = 0; i < n; i++) {
    mt += t.mass_atom[i];
    for (int k = 0; k < 3; k++) p[k] += t.mass_atom[i] * v[3 * i + k];
  }
  for (int i = 0; i < n; i++)
    for (int k = 0; k < 3; k++) v[3 * i + k] -= p[k] / mt;
  for (int i = 0; i < n; i++)
    for (int k = 0; k < 3; k++) cm[k] += t.mass_atom[i] * x[3 * i + k] / mt;
  double L[3] = {0, 0, 0}, I[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int i = 0; i < n; i++) {
    const double m = t.mass_atom[i];
    const double r[3] = {x[3 * i] - cm[0], x[3 * i + 1] - cm[1], x[3 * i + 2] - cm[2]};
    const double *vv = &v[3 * i];
    L[0] += m * (r[1] * vv[2] - r[2] * vv[1]);
    L[1] += m * (r[2] * vv[0] - r[0] * vv[2]);
    L[2] += m * (r[0] * vv[1] - r[1] * vv[0]);
    const double r2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) I[a][b] += m * ((a == b ? r2 : 0.0) - r[a] * r[b]);
  }
  const double det = I[0][0] * (I[1][1] * I[2][2] - I[1][2] * I[2][1]) - I[0][1] * (I[1][0] * I[2][2] - I[1][2] * I[2][0]) +
                     I[0][2] * (I[1][0] * I[2][1] - I[1][1] * I[2][0]);
  double inv[3][3];
  inv[0][0] = (I[1][1] * I[2][2] - I[1][2] * I[2][1]) / det; inv[0][1] = (I[0][2] * I[2][1] - I[0][1] * I[2][2]) / det; inv[0][2] = (I[0][1] * I[1][2] - I[0][2] * I[1][1]) / det;
  inv[1][0] = (I[1][2] * I[2][0] - I[1][0] * I[2][2]) / det; inv[1][1] = (I[0][0] * I[2][2] - I[0][2] * I[2][0]) / det; inv[1][2] = (I[0][2] * I[1][0] - I[0][0] * I[1][2]) / det;
  inv[2][0] = (I[1][0] * I[2][1] - I[1][1] * I[2][0]) / det; inv[2][1] = (I[0][1] * I[2][0] - I[0][0] * I[2][1]) / det; inv[2][2] = (I[0][0] * I[1][1] - I[0][1] * I[1][0]) / det;
  double w[3];
  for (int a = 0; a < 3; a++) w[a] = inv[a][0] * L[0] + inv[a][1] * L[1] + inv[a][2] * L[2];
  double ke = 0.0;
  for (int i = 0; i < n; i++) {
    const double r[3] = {x[3 * i] - cm[0], x[3 * i + 1] - cm[1], x[3 * i + 2] - cm[2]};
    v[3 * i] -= w[1] * r[2] - w[2] * r[1];
    v[3 * i + 1] -= w[2] * r[0] - w[0] * r[2];
    v[3 * i + 2] -= w[0] * r[1] - w[1] * r[0];
    for (int k = 0; k < 3; k++) ke += t.mass_atom[i] * v[3 * i + k] * v[3 * i + k];
  }
  const double tcur = ke * MD_MVV2E / ((3.0 * n - 3.0) * MD_BOLTZ), sc = std::sqrt(temperature / tcur);
  for (double &q : v) q *= sc;
}

struct EquilCtx {
  scema_md_engine *e;
  State *s;
  DevBuf xb, vb;   // state at the start of the segment / of the minimisation, for a retry
};
int equil_backup(EquilCtx &c) {
  scema_md_engine *e = c.e;
  const size_t bytes = 3 * (size_t)c.s->topo->natoms * 8;
  HIPCHK(c.xb.ensure(bytes));
  HIPCHK(c.vb.ensure(bytes));
  HIPCHK(hipMemcpyAsync(c.xb.p, c.s->x.p, bytes, hipMemcpyDeviceToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(c.vb.p, c.s->v.p, bytes, hipMemcpyDeviceToDevice, e->stream));
  return SCEMA_MD_OK;
}
int equil_restore(EquilCtx &c) {
  scema_md_engine *e = c.e;
  const size_t bytes = 3 * (size_t)c.s->topo->natoms * 8;
  HIPCHK(hipMemcpyAsync(c.s->x.p, c.xb.p, bytes, hipMemcpyDeviceToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(c.s->v.p, c.vb.p, bytes, hipMemcpyDeviceToDevice, e->stream));
  return SCEMA_MD_OK;
}
void grow_lists(scema_md_engine *e) {
  if (e->overflow_bits & 4) e->jtab_grow *= 1.25;
  if ((e->overflow_bits & 8) || !(e->overflow_bits & 4)) e->neigh_grow *= 1.5;
}
// min_style sd ; minimize etol ftol maxiter maxeval.  info[4]: iterations, force evaluations, initial and final energy
int equil_minimize(EquilCtx &c, double etol, double ftol, int maxiter, int maxeval, int *stop, double *info) {
  scema_md_engine *e = c.e;
  int rc = equil_backup(c);
  if (rc) return rc;
  std::vector<ActiveSim> sims(1);
  sims[0].st = c.s;
  sims[0].nsteps = 0;
  sims[0].dt = 1.0;
  sims[0].temperature = 300.0;
  for (int attempt = 0; attempt < 8; attempt++) {
    if ((rc = prepare_slots(e, sims))) return rc;
    RunSpec R;
    R.nvt = 0; R.use_shake = 0; R.ev_always = 1; R.static_only = 1;
    R.minimize = 1; R.min_etol = etol; R.min_ftol = ftol; R.min_maxiter = maxiter; R.min_maxeval = maxeval;
    rc = run_phase(e, sims, R);
    if (rc != SCEMA_MD_ERR_OVERFLOW) break;
    grow_lists(e);
    if ((rc = equil_restore(c))) return rc;
    rc = SCEMA_MD_ERR_OVERFLOW;
  }
  if (rc) return rc;
  const SimScalars &sc = e->h_sc[0];
  if (stop) *stop = sc.min_stop;
  if (info) { info[0] = sc.min_iter; info[1] = sc.min_neval; info[2] = sc.min_einit; info[3] = sc.min_ecur; }
  return SCEMA_MD_OK;
}
// run N under fix nvt / fix npt ... iso with a ramp, in segments; lavg != NULL: box-length averages (two half-run windows)
int equil_run_nh(EquilCtx &c, int nsteps, double dt, double t_start, double t_stop, bool npt, double p_target, double p_period, double *lavg) {
  scema_md_engine *e = c.e;
  if (nsteps <= 0) return SCEMA_MD_OK;
  std::vector<ActiveSim> sims(1);
  sims[0].st = c.s;
  sims[0].dt = dt;
  sims[0].temperature = t_start;
  std::vector<EwaldSetup> ew_keep;
  SimScalars carry;
  std::memset(&carry, 0, sizeof carry);
  int done = 0, seg = npt ? 250 : nsteps;
  const double margin = 0.02;
  bool first = true;
  int failures = 0;
  while (done < nsteps) {
    const int len = std::min(seg, nsteps - done);
    int rc = equil_backup(c);
    if (rc) return rc;
    sims[0].nsteps = len;
    if (first) {
      if ((rc = prepare_slots(e, sims))) return rc;
    } else {
      e->h_sc.assign(1, carry);
      e->h_sc[0].keep_nh = 1;
      if ((rc = reupload_scalars(e, 1))) return rc;
    }
    RunSpec R;
    R.nvt = 1; R.use_shake = 0;
    R.nh = 1; R.npt = npt ? 1 : 0; R.keep = first ? 0 : 1; R.nh_total = nsteps; R.lavg_nav = lavg ? nsteps / 2 : 0;
    R.t_start = t_start; R.t_stop = t_stop; R.p_target = p_target; R.p_period = p_period; R.box_margin = margin;
    R.ew_keep = &ew_keep;
    rc = run_phase(e, sims, R);
    if (rc == SCEMA_MD_ERR_OVERFLOW) {
      if (++failures > 12) return fail(e, SCEMA_MD_ERR_ARG, "equilibration: a run segment kept failing (box leaving its range or lists overflowing)");
      if (e->overflow_bits & 1) grow_lists(e);
      else seg = std::max(10, seg / 2);   // the box left the range the segment was laid out for: shorter segments
      if ((rc = equil_restore(c))) return rc;
      if (!first) std::memcpy(c.s->box, carry.box, sizeof carry.box);
      continue;
    }
    if (rc) return rc;
    carry = e->h_sc[0];
    std::memcpy(c.s->box, carry.box, sizeof carry.box);
    done += len;
    first = false;
  }
  if (lavg)
    for (int d = 0; d < 3; d++) lavg[d] = carry.nlwin > 0 ? carry.lrun[d] / carry.nlwin : c.s->box[3 + d] - c.s->box[d];
  return SCEMA_MD_OK;
}
// change_box all x final 0 lx y final 0 ly z final 0 lz remap (tilts kept)
int equil_change_box(EquilCtx &c, const double len[3]) {
  scema_md_engine *e = c.e;
  double bb[18];
  std::memcpy(bb, c.s->box, 9 * sizeof(double));
  std::memcpy(bb + 9, c.s->box, 9 * sizeof(double));
  for (int d = 0; d < 3; d++) { bb[9 + d] = 0.0; bb[9 + 3 + d] = len[d]; }
  HIPCHK(e->d_boxpair.ensure(sizeof bb));
  HIPCHK(hipMemcpyAsync(e->d_boxpair.p, bb, sizeof bb, hipMemcpyHostToDevice, e->stream));
  mdk_change_box(e->stream, c.s->x.as<double>(), c.s->topo->natoms, e->d_boxpair.as<double>(), e->d_boxpair.as<double>() + 9);
  HIPCHK(hipStreamSynchronize(e->stream));
  std::memcpy(c.s->box, bb + 9, 9 * sizeof(double));
  return SCEMA_MD_OK;
}
}  // namespace

// test hooks: one minimisation / one thermostatted (barostatted) run on a stored state
int scema_md_debug_minimize(scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica, double etol, double ftol, int32_t maxiter,
                            int32_t maxeval, double *info) {
  if (!e) return SCEMA_MD_ERR_ARG;
  HIPCHK(hipSetDevice(e->p.device));
  if (qp_id == SCEMA_MD_QP_NONE) return fail(e, SCEMA_MD_ERR_ARG, "debug_minimize needs a stored state (scema_md_set_state first)");
  State *s = find_state(e, qp_id, matid, replica);
  if (!s) return fail(e, SCEMA_MD_ERR_NOSTATE, "no state for qp %d", (int)qp_id);
  EquilCtx c{e, s, DevBuf(), DevBuf()};
  int stop = -1;
  double inf[4] = {0, 0, 0, 0};
  const int rc = equil_minimize(c, etol, ftol, maxiter, maxeval, &stop, inf);
  if (rc) return rc;
  if (info) { info[0] = stop; for (int k = 0; k < 4; k++) info[1 + k] = inf[k]; }
  return SCEMA_MD_OK;
}
int scema_md_debug_run_nh(scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica, int32_t nsteps, double dt, double t_start,
                          double t_stop, int32_t npt, double p_target, double p_period, double *lavg) {
  if (!e || nsteps < 0) return SCEMA_MD_ERR_ARG;
  HIPCHK(hipSetDevice(e->p.device));
  if (qp_id == SCEMA_MD_QP_NONE) return fail(e, SCEMA_MD_ERR_ARG, "debug_run_nh needs a stored state (scema_md_set_state first)");
  State *s = find_state(e, qp_id, matid, replica);
  if (!s) return fail(e, SCEMA_MD_ERR_NOSTATE, "no state for qp %d", (int)qp_id);
  EquilCtx c{e, s, DevBuf(), DevBuf()};
  return equil_run_nh(c, nsteps, dt, t_start, t_stop, npt != 0, p_target, p_period, lavg);
}

int scema_md_equilibrate(scema_md_engine *e, const char *matid, int32_t replica, const scema_md_equilparams *p, double length[3], double *info) {
  if (!e || !p || !length) return SCEMA_MD_ERR_ARG;
  HIPCHK(hipSetDevice(e->p.device));
  Topo *t = find_topo(e, matid, replica);
  if (!t) return fail(e, SCEMA_MD_ERR_NOSTATE, "replica %s_%d is not registered", matid ? matid : "", (int)replica);
  if (p->nsteps_equil < 2 || p->timestep_length <= 0.0 || p->temperature <= 0.0)
    return fail(e, SCEMA_MD_ERR_ARG, "equilibrate: nsteps_equil >= 2, timestep_length > 0, temperature > 0 required");
  const int ns = p->nsteps_equil;
  const double dt = p->timestep_length, tempt = p->temperature;
  // in.init.lammps:48: velocity all create 200.0 ${sseed} rot yes dist gaussian (sseed = 1234, init_material_problem.h:167)
  std::vector<double> v0;
  velocity_create(*t, t->init_x, 200.0, p->seed ? (unsigned long long)p->seed : 1234ull, v0);
  std::unique_ptr<State> st;
  int rc = make_state(e, t, t->init_box, t->init_x.data(), v0.data(), false, st);
  if (rc) return rc;
  EquilCtx c{e, st.get(), DevBuf(), DevBuf()};
  int stop = -1;
  double minfo[4] = {0, 0, 0, 0}, lav[3];
  // :54-58 min_style sd ; minimize 1.0e-7 1.0e-11 ${nsi} 50000
  if ((rc = equil_minimize(c, 1.0e-7, 1.0e-11, ns, 50000, &stop, minfo))) return rc;
  // :105-215 the heat-up / cool-down schedule (fix shake is commented out in the script)
  if ((rc = equil_run_nh(c, ns, dt, 300.0, 300.0, false, 0.0, 1000.0, nullptr))) return rc;
  if ((rc = equil_run_nh(c, ns, dt, 300.0, 500.0, true, 1.0, 1000.0, nullptr))) return rc;
  if ((rc = equil_run_nh(c, 5 * ns, dt, 500.0, 500.0, true, 1.0, 1000.0, nullptr))) return rc;
  if ((rc = equil_run_nh(c, ns, dt, 500.0, tempt, true, 1.0, 1000.0, nullptr))) return rc;
  if ((rc = equil_run_nh(c, 2 * ns, dt, tempt, tempt, true, 1.0, 1000.0, lav))) return rc;
  if ((rc = equil_change_box(c, lav))) return rc;
  if ((rc = equil_run_nh(c, 20 * ns, dt, tempt, tempt, false, 0.0, 1000.0, nullptr))) return rc;
  if ((rc = equil_run_nh(c, 2 * ns, dt, tempt, tempt, true, 1.0, 1000.0, lav))) return rc;
  if ((rc = equil_change_box(c, lav))) return rc;
  if ((rc = equil_run_nh(c, ns, dt, tempt, tempt, false, 0.0, 1000.0, nullptr))) return rc;
  // the equilibrated state becomes the replica's initial state (what write_restart init.<mat>_<rep>.bin keeps, :208-210)
  std::memcpy(t->init_box, st->box, sizeof t->init_box);
  t->init_v.assign(3 * (size_t)t->natoms, 0.0);
  HIPCHK(hipMemcpy(t->init_x.data(), st->x.p, 3 * (size_t)t->natoms * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(t->init_v.data(), st->v.p, 3 * (size_t)t->natoms * 8, hipMemcpyDeviceToHost));
  for (int d = 0; d < 3; d++) length[d] = st->box[3 + d] - st->box[d];
  if (info) { info[0] = stop; for (int k = 0; k < 4; k++) info[1 + k] = minfo[k]; }
  return SCEMA_MD_OK;
}

// ---- init_material: what EQMDProblem::lammps_equilibration computes once the replica is equilibrated ----
int scema_md_init_material(scema_md_engine *e, const char *matid, int32_t replica, const scema_md_eqparams *p, double length[3],
                           double stress[6], double stiff[36]) {
  if (!e || !p || !length || !stress || !stiff) return SCEMA_MD_ERR_ARG;
  HIPCHK(hipSetDevice(e->p.device));
  Topo *t = find_topo(e, matid, replica);
  if (!t) return fail(e, SCEMA_MD_ERR_NOSTATE, "replica %s_%d is not registered", matid ? matid : "", (int)replica);
  if (p->nsteps_sample < 10 || p->strain_ampl <= 0.0 || p->strain_rate <= 0.0 || p->timestep_length <= 0.0)
    return fail(e, SCEMA_MD_ERR_ARG, "init_material: nsteps_sample >= 10, strain_ampl, strain_rate, timestep_length > 0 required");
  // box lengths after initiation (init_material_problem.h:196-208)
  for (int d = 0; d < 3; d++) length[d] = t->init_box[3 + d] - t->init_box[d];
  // ---- ELASTIC/in.homogenization.lammps: NVT + SHAKE sampling of the unstrained replica ----
  std::unique_ptr<State> equil;
  int rc = make_state(e, t, t->init_box, t->init_x.data(), t->init_v.data(), false, equil);
  if (rc) return rc;
  {
    std::vector<ActiveSim> one(1);
    one[0].st = equil.get();
    one[0].dt = p->timestep_length;
    one[0].temperature = p->temperature;
    one[0].nts = 0;
    one[0].nss = p->nsteps_sample;
    EvalOpt o;
    o.phase_a = false;
    if ((rc = eval_chunk(e, one, o))) return rc;
    // loc_rep_stress[k][l] = -pp{k+1}{l+1} * 1.01325e5 (init_material_problem.h:243-250); file order 00,01,02,11,12,22
    static const int RAW_OF_FILE[6] = {0, 3, 4, 1, 5, 2};
    for (int f = 0; f < 6; f++) stress[f] = -one[0].pavg[RAW_OF_FILE[f]] * 1.01325e5;
  }
  // ---- ELASTIC/in.modulus.lammps + bi-displace.mod.lammps: +-up in each of the six directions, from the state the
  // homogenisation run left ("restart.equil"); fix nvt only (fix shake is commented out there); fix deform ... delta
  // over nsstrain steps, then nssample steps of sampling ----
  const int nsstrain = (int)(std::ceil(p->strain_ampl / (p->timestep_length * p->strain_rate) / 10.0) * 10.0);   // :226
  const double T = nsstrain * p->timestep_length, up = p->strain_ampl;
  const double xy = equil->box[6], xz = equil->box[7], yz = equil->box[8];
  const double ly0 = equil->box[4] - equil->box[1], lz0 = equil->box[5] - equil->box[2];
  std::vector<std::unique_ptr<State>> st(12);
  std::vector<ActiveSim> runs(12);
  for (int dir = 0; dir < 6; dir++)
    for (int pn = 0; pn < 2; pn++) {
      const int i = 2 * dir + pn;
      const double sign = pn == 0 ? -1.0 : 1.0;   // "neg" first, then "pos"
      if ((rc = make_state(e, t, equil->box, equil->x.as<double>(), equil->v.as<double>(), true, st[i]))) return rc;
      ActiveSim &A = runs[i];
      A.st = st[i].get();
      A.dt = p->timestep_length;
      A.temperature = p->temperature;
      A.nts = nsstrain;
      A.nss = p->nsteps_sample;
      // engine rates (raw order xx,yy,zz,xy,xz,yz): L(t) = L0 (1 + r t); xy(t) = xy0 + r Ly0 t; xz, yz with Lz0
      double *r = A.rates;
      if (dir == 0) { r[0] = sign * up / T; r[3] = -sign * up * xy / (ly0 * T); r[4] = -sign * up * xz / (lz0 * T); }
      if (dir == 1) { r[1] = sign * up / T; r[5] = -sign * up * yz / (lz0 * T); }
      if (dir == 2) r[2] = sign * up / T;
      if (dir == 3) r[5] = sign * up / T;   // yz delta = sign up lz0
      if (dir == 4) r[4] = sign * up / T;   // xz delta = sign up lz0
      if (dir == 5) r[3] = sign * up / T;   // xy delta = sign up ly0
    }
  {
    EvalOpt o;
    o.shake_a = 0;
    o.shake_b = 0;
    if ((rc = eval_chunk(e, runs, o))) return rc;
  }
  // C_i,dir = 0.5 (C_i^neg + C_i^pos), d_i = -(p_i1 - p_i0)/(delta/len0) cfac with (pxx,pyy,pzz,pyz,pxz,pxy): the
  // unstrained p_i0 cancels in the average.  GPa: cfac = 1.01325e-4 (init.mod.lammps).
  static const int RAW_OF_VOIGT[6] = {0, 1, 2, 5, 4, 3};
  double C[6][6];
  for (int dir = 0; dir < 6; dir++)
    for (int i = 0; i < 6; i++) {
      const double pneg = runs[2 * dir].pavg[RAW_OF_VOIGT[i]], ppos = runs[2 * dir + 1].pavg[RAW_OF_VOIGT[i]];
      C[i][dir] = -(ppos - pneg) / (2.0 * up) * 1.01325e-4;
    }
  // C{ij}all: diagonal as computed, off-diagonal 0.5 (Cij + Cji); GPa -> Pa (init_material_problem.h:262-270)
  double call[6][6];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) call[i][j] = (i == j ? C[i][i] : 0.5 * (C[i][j] + C[j][i])) * 1.0e9;
  // 6x6 -> rank 4 exactly as init_material_problem.h:276-295 does it: index 3 -> (0,1), 4 -> (0,2), 5 -> (1,2)
  // (the script's Voigt order is 4 = yz, 5 = xz, 6 = xy; the mapping of the reference is reproduced, not corrected).
  // stiff: file order of read_write.h:149-171, (00,01,02,11,12,22) x (00,01,02,11,12,22).
  static const int HOSTVOIGT_OF_FILE[6] = {0, 3, 4, 1, 5, 2};
  for (int I = 0; I < 6; I++)
    for (int J = 0; J < 6; J++) stiff[I * 6 + J] = call[HOSTVOIGT_OF_FILE[I]][HOSTVOIGT_OF_FILE[J]];
  return SCEMA_MD_OK;
}

// ---- ReaxFF path ----
int scema_md_reax_configure(scema_md_engine *e, const char *ffield_path, const char *const *elements, int32_t n_elements, double qeq_tol, double skin) {
  if (!e || !ffield_path || !elements || n_elements <= 0) return fail(e, SCEMA_MD_ERR_ARG, "bad arguments");
  HIPCHK(hipSetDevice(e->p.device));
  std::vector<std::string> el(elements, elements + n_elements);
  std::string err;
  RxParams P;
  std::vector<int> map;
  if (!scema::read_reax_ffield(ffield_path, el, P, map, err)) return fail(e, SCEMA_MD_ERR_IO, "%s", err.c_str());
  if (const char *x = getenv("SCEMA_REAX_DROP_DSBO2")) P.lammps_dsbo2 = atoi(x) ? 1 : 0;
  e->rx_host = P;
  e->rx_type_map = map;
  if (qeq_tol > 0.0) e->rx_qeq_tol = qeq_tol;
  if (skin >= 0.0) e->rx_skin = skin;
  if (const char *x = getenv("SCEMA_REAX_SKIN")) e->rx_skin = atof(x);
  if (const char *x = getenv("SCEMA_REAX_QEQ_LAUNCH")) { e->rx_qeq_launch = e->rx_qeq_launch_cold = std::max(0, atoi(x)); e->rx_qeq_launch_pinned = true; }
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
  if (reset) e->prof = Profile();
  return SCEMA_MD_OK;
}

}  // extern "C"
