// engine_state.cpp -- the state store: branch rule of stmd_problem.h:116-138,185-207, replica and state files
#include "engine.h"

namespace scema_eng {

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
int resolve_state(scema_md_engine *e, const scema_mdsim &m, State **out, std::unique_ptr<State> *incoming, bool *created,
                  std::unique_ptr<State> *displaced) {
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

}  // namespace scema_eng

extern "C" {

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
  if (e) (void)settle_pending(e, false);   // an update that waits for its verdict (no communicator) stands once the engine is used for something else
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
  if (e) (void)settle_pending(e, false);   // an update that waits for its verdict (no communicator) stands once the engine is used for something else
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
  if (e) (void)settle_pending(e, false);   // an update that waits for its verdict (no communicator) stands once the engine is used for something else
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


}  // extern "C"
