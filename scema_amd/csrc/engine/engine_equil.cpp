// engine_equil.cpp -- init_material: the equilibration schedule of in.init.lammps, the homogenisation run and the stiffness runs
#include "engine.h"

namespace scema_eng {

// ---- init_material: the equilibration schedule (lammps_scripts_opls/in.init.lammps:44-215) ----
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

}  // namespace scema_eng

extern "C" {

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
  if (e) (void)settle_pending(e, false);   // an update that waits for its verdict (no communicator) stands once the engine is used for something else
  if (!e || nsteps < 0) return SCEMA_MD_ERR_ARG;
  HIPCHK(hipSetDevice(e->p.device));
  if (qp_id == SCEMA_MD_QP_NONE) return fail(e, SCEMA_MD_ERR_ARG, "debug_run_nh needs a stored state (scema_md_set_state first)");
  State *s = find_state(e, qp_id, matid, replica);
  if (!s) return fail(e, SCEMA_MD_ERR_NOSTATE, "no state for qp %d", (int)qp_id);
  EquilCtx c{e, s, DevBuf(), DevBuf()};
  return equil_run_nh(c, nsteps, dt, t_start, t_stop, npt != 0, p_target, p_period, lavg);
}

int scema_md_equilibrate(scema_md_engine *e, const char *matid, int32_t replica, const scema_md_equilparams *p, double length[3], double *info) {
  if (e) (void)settle_pending(e, false);   // an update that waits for its verdict (no communicator) stands once the engine is used for something else
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
  if (e) (void)settle_pending(e, false);   // an update that waits for its verdict (no communicator) stands once the engine is used for something else
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

}  // extern "C"
