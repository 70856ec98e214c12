// lammps_restart.cpp -- LAMMPS binary restart files in the layout of the version the reference pins
// ("17 Nov 2016", reference README.md:31-37): what `read_restart init.<mat>_<rep>.bin` (stmd_problem.h:204) and
// `write_restart last.<qp>...` (stmd_problem.h:258) exchange with LAMMPS.
//
// file := magic "LammpS RestartT\0" | int endian (1) | int versionnumeric
//         header records ... -1 | groups | type arrays (MASS) ... -1 | force fields ... -1 |
//         fix state lists | file layout ... -1 | per-proc atom blocks (PERPROC n, n doubles)
// record := int flag, then an int | bigint | double | string (int n, n chars) | vector (int n, n values)
// The record walk, the group list, the fix lists and the atomic per-atom block are pinned by the reference's own
// fixture examples/streched_polyhedron/nanoscale_input/init.sic_1.bin (tests/golden/); the atom_style full block
// and the coefficient blocks of lj/cut/coul/long, harmonic, opls follow that version's pack_restart /
// write_restart routines and are only pinned by a round trip through the writer below.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../../include/scema_md.h"

namespace {

enum {
  VERSION, SMALLINT, TAGINT, BIGINT, UNITS, NTIMESTEP, DIMENSION, NPROCS, PROCGRID, NEWTON_PAIR, NEWTON_BOND, XPERIODIC, YPERIODIC,
  ZPERIODIC, BOUNDARY, ATOM_STYLE, NATOMS, NTYPES, NBONDS, NBONDTYPES, BOND_PER_ATOM, NANGLES, NANGLETYPES, ANGLE_PER_ATOM, NDIHEDRALS,
  NDIHEDRALTYPES, DIHEDRAL_PER_ATOM, NIMPROPERS, NIMPROPERTYPES, IMPROPER_PER_ATOM, TRICLINIC, BOXLO, BOXHI, XY, XZ, YZ, SPECIAL_LJ,
  SPECIAL_COUL, MASS, PAIR, BOND, ANGLE, DIHEDRAL, IMPROPER, MULTIPROC, MPIIO, PROCSPERFILE, PERPROC, IMAGEINT, BOUNDMIN, TIMESTEP,
  ATOM_ID, ATOM_MAP_STYLE, ATOM_MAP_USER, ATOM_SORTFREQ, ATOM_SORTBIN, COMM_MODE, COMM_CUTOFF, COMM_VEL, NO_PAIR, NFLAGS
};
const char MAGIC[16] = {'L', 'a', 'm', 'm', 'p', 'S', ' ', 'R', 'e', 's', 't', 'a', 'r', 't', 'T', '\0'};

struct Restart {
  std::string version, units, atom_style, pair_style, bond_style, angle_style, dihedral_style, improper_style;
  int64_t natoms = 0, ntimestep = 0, nbonds = 0, nangles = 0, ndihedrals = 0, nimpropers = 0;
  int ntypes = 0, nbondtypes = 0, nangletypes = 0, ndihedraltypes = 0, nimpropertypes = 0;
  int triclinic = 0, nprocs = 0, imageint = 4, newton_bond = 1, no_pair = 0;
  double boxlo[3] = {0, 0, 0}, boxhi[3] = {0, 0, 0}, tilt[3] = {0, 0, 0}, timestep = 0;
  double special_lj[3] = {0, 0, 0}, special_coul[3] = {0, 0, 0};
  std::vector<std::string> groups;
  std::vector<double> mass;
  double cut_lj = 0, cut_coul = 0;
  int mix_flag = 0;
  std::vector<int> pair_set;                 // ntypes x ntypes, upper triangle
  std::vector<double> pair_eps, pair_sigma, pair_cut;
  std::vector<double> bond_c, angle_c, dihedral_c, improper_c;   // coefficient-major arrays as stored
  // atoms
  std::vector<int64_t> tag, image;
  std::vector<int32_t> type;
  std::vector<double> x, v, q;
  std::vector<int64_t> bonds, angles, dihedrals, impropers;      // (type, tags...) per entry
};

struct Reader {
  std::vector<unsigned char> b;
  size_t pos = 0;
  std::string err;
  bool ok = true;
  bool need(size_t n) {
    if (!ok) return false;
    if (pos + n > b.size()) { ok = false; err = "truncated restart file"; return false; }
    return true;
  }
  int32_t i32() { int32_t v = 0; if (need(4)) { std::memcpy(&v, &b[pos], 4); pos += 4; } return v; }
  int64_t i64() { int64_t v = 0; if (need(8)) { std::memcpy(&v, &b[pos], 8); pos += 8; } return v; }
  double f64() { double v = 0; if (need(8)) { std::memcpy(&v, &b[pos], 8); pos += 8; } return v; }
  std::string str() {
    const int32_t n = i32();
    if (n < 0 || !need((size_t)n)) { ok = false; if (err.empty()) err = "bad string length"; return ""; }
    std::string s((const char *)&b[pos], (size_t)n);
    pos += (size_t)n;
    while (!s.empty() && s.back() == '\0') s.pop_back();
    return s;
  }
  void dvec(std::vector<double> &out) {
    const int32_t n = i32();
    if (n < 0 || !need((size_t)n * 8)) { ok = false; if (err.empty()) err = "bad vector length"; return; }
    out.resize((size_t)n);
    if (n) std::memcpy(out.data(), &b[pos], (size_t)n * 8);
    pos += (size_t)n * 8;
  }
  void ivec(std::vector<int32_t> &out) {
    const int32_t n = i32();
    if (n < 0 || !need((size_t)n * 4)) { ok = false; if (err.empty()) err = "bad vector length"; return; }
    out.resize((size_t)n);
    if (n) std::memcpy(out.data(), &b[pos], (size_t)n * 4);
    pos += (size_t)n * 4;
  }
};

inline int64_t bits(double d) { int64_t v; std::memcpy(&v, &d, 8); return v; }   // LAMMPS' ubuf: integers travel as raw bits
inline double unbits(int64_t v) { double d; std::memcpy(&d, &v, 8); return d; }

// one "(flag, value)* -1" section; returns -1 at the end marker, or the flag of a record this walker does not decode
// (force-field styles: the caller reads the style block)
int walk(Reader &r, Restart &R) {
  std::vector<double> dv;
  std::vector<int32_t> iv;
  while (r.ok) {
    const int32_t flag = r.i32();
    if (!r.ok) return -2;
    if (flag < 0) return -1;
    switch (flag) {
      case VERSION: R.version = r.str(); break;
      case UNITS: R.units = r.str(); break;
      case ATOM_STYLE: {
        R.atom_style = r.str();
        const int32_t na = r.i32();   // arguments of the style (hybrid sub-styles, templates)
        for (int32_t k = 0; k < na && r.ok; k++) (void)r.str();
        break;
      }
      case SMALLINT: case TAGINT: case BIGINT:
        (void)r.i32(); break;
      case IMAGEINT: R.imageint = r.i32(); break;
      case NTIMESTEP: R.ntimestep = r.i64(); break;
      case NATOMS: R.natoms = r.i64(); break;
      case NBONDS: R.nbonds = r.i64(); break;
      case NANGLES: R.nangles = r.i64(); break;
      case NDIHEDRALS: R.ndihedrals = r.i64(); break;
      case NIMPROPERS: R.nimpropers = r.i64(); break;
      case NTYPES: R.ntypes = r.i32(); break;
      case NBONDTYPES: R.nbondtypes = r.i32(); break;
      case NANGLETYPES: R.nangletypes = r.i32(); break;
      case NDIHEDRALTYPES: R.ndihedraltypes = r.i32(); break;
      case NIMPROPERTYPES: R.nimpropertypes = r.i32(); break;
      case TRICLINIC: R.triclinic = r.i32(); break;
      case NPROCS: R.nprocs = r.i32(); break;
      case NEWTON_BOND: R.newton_bond = r.i32(); break;
      case DIMENSION: case NEWTON_PAIR: case XPERIODIC: case YPERIODIC: case ZPERIODIC: case BOND_PER_ATOM: case ANGLE_PER_ATOM:
      case DIHEDRAL_PER_ATOM: case IMPROPER_PER_ATOM: case ATOM_ID: case ATOM_MAP_STYLE: case ATOM_MAP_USER: case ATOM_SORTFREQ:
      case COMM_MODE: case COMM_VEL: case MULTIPROC: case MPIIO: case PROCSPERFILE:
        (void)r.i32(); break;
      case XY: R.tilt[0] = r.f64(); break;
      case XZ: R.tilt[1] = r.f64(); break;
      case YZ: R.tilt[2] = r.f64(); break;
      case TIMESTEP: R.timestep = r.f64(); break;
      case ATOM_SORTBIN: case COMM_CUTOFF: (void)r.f64(); break;
      case PROCGRID: case BOUNDARY: r.ivec(iv); break;
      case BOUNDMIN: r.dvec(dv); break;
      case BOXLO: r.dvec(dv); for (size_t k = 0; k < 3 && k < dv.size(); k++) R.boxlo[k] = dv[k]; break;
      case BOXHI: r.dvec(dv); for (size_t k = 0; k < 3 && k < dv.size(); k++) R.boxhi[k] = dv[k]; break;
      case SPECIAL_LJ: r.dvec(dv); for (size_t k = 0; k < 3 && k < dv.size(); k++) R.special_lj[k] = dv[k]; break;
      case SPECIAL_COUL: r.dvec(dv); for (size_t k = 0; k < 3 && k < dv.size(); k++) R.special_coul[k] = dv[k]; break;
      case MASS: r.dvec(R.mass); break;
      // 17Nov16 writes nothing for a pair style without restart info (the reference's init.sic_1.bin: pair sw, no record);
      // later versions write NO_PAIR followed by the style name as a string: consume it, or the record walk desynchronises
      case NO_PAIR: R.pair_style = r.str(); R.no_pair = 1; break;
      case PAIR: case BOND: case ANGLE: case DIHEDRAL: case IMPROPER: return flag;
      default:
        r.ok = false;
        r.err = "unknown record flag " + std::to_string(flag);
        return -2;
    }
  }
  return -2;
}

bool coeff_block(Reader &r, int ntypes_of_kind, int ncoef, std::vector<double> &out) {
  out.resize((size_t)ntypes_of_kind * ncoef);
  for (int c = 0; c < ncoef; c++)
    for (int t = 0; t < ntypes_of_kind; t++) out[(size_t)c * ntypes_of_kind + t] = r.f64();
  return r.ok;
}

int parse(const char *path, Restart &R, std::string &err, bool want_atoms) {
  if (!path) return SCEMA_MD_ERR_ARG;
  FILE *f = fopen(path, "rb");
  if (!f) { err = std::string("cannot open ") + path; return SCEMA_MD_ERR_IO; }
  Reader r;
  fseek(f, 0, SEEK_END);
  const long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  r.b.resize(sz > 0 ? (size_t)sz : 0);
  const size_t got = r.b.empty() ? 0 : fread(r.b.data(), 1, r.b.size(), f);
  fclose(f);
  if (got != r.b.size() || r.b.size() < 24 || std::memcmp(r.b.data(), MAGIC, 16) != 0) { err = "not a LAMMPS restart file"; return SCEMA_MD_ERR_IO; }
  r.pos = 16;
  const int32_t endian = r.i32();
  (void)r.i32();   // versionnumeric
  if (endian != 1) { err = "restart file written with the other endianness"; return SCEMA_MD_ERR_IO; }
  if (walk(r, R) != -1) { err = r.err.empty() ? "bad header" : r.err; return SCEMA_MD_ERR_IO; }
  const int32_t ng = r.i32();
  for (int32_t g = 0; g < ng && r.ok; g++) R.groups.push_back(r.str());
  if (walk(r, R) != -1) { err = r.err.empty() ? "bad type-array section" : r.err; return SCEMA_MD_ERR_IO; }
  for (;;) {   // force fields
    const int flag = walk(r, R);
    if (flag == -1) break;
    if (flag < 0) { err = r.err.empty() ? "bad force-field section" : r.err; return SCEMA_MD_ERR_IO; }
    const std::string style = r.str();
    if (flag == PAIR) {
      R.pair_style = style;
      if (style != "lj/cut/coul/long") { err = "pair style " + style + ": coefficient block layout not known to this reader"; return SCEMA_MD_ERR_ARG; }
      // settings: cut_lj_global, cut_coul, offset_flag, mix_flag, tail_flag, ncoultablebits, tabinner
      R.cut_lj = r.f64(); R.cut_coul = r.f64();
      (void)r.i32(); R.mix_flag = r.i32(); (void)r.i32(); (void)r.i32(); (void)r.f64();
      const int nt = R.ntypes;
      R.pair_set.assign((size_t)nt * nt, 0);
      R.pair_eps.assign((size_t)nt * nt, 0.0); R.pair_sigma.assign((size_t)nt * nt, 0.0); R.pair_cut.assign((size_t)nt * nt, 0.0);
      for (int i = 0; i < nt; i++)
        for (int j = i; j < nt; j++) {
          const int set = r.i32();
          R.pair_set[(size_t)i * nt + j] = set;
          if (set) { R.pair_eps[(size_t)i * nt + j] = r.f64(); R.pair_sigma[(size_t)i * nt + j] = r.f64(); R.pair_cut[(size_t)i * nt + j] = r.f64(); }
        }
    } else if (flag == BOND) {
      R.bond_style = style;
      if (style != "harmonic") { err = "bond style " + style + " not known to this reader"; return SCEMA_MD_ERR_ARG; }
      coeff_block(r, R.nbondtypes, 2, R.bond_c);
    } else if (flag == ANGLE) {
      R.angle_style = style;
      if (style != "harmonic") { err = "angle style " + style + " not known to this reader"; return SCEMA_MD_ERR_ARG; }
      coeff_block(r, R.nangletypes, 2, R.angle_c);
    } else if (flag == DIHEDRAL) {
      R.dihedral_style = style;
      if (style != "opls") { err = "dihedral style " + style + " not known to this reader"; return SCEMA_MD_ERR_ARG; }
      coeff_block(r, R.ndihedraltypes, 4, R.dihedral_c);
    } else {
      R.improper_style = style;
      if (style != "harmonic") { err = "improper style " + style + " not known to this reader"; return SCEMA_MD_ERR_ARG; }
      coeff_block(r, R.nimpropertypes, 2, R.improper_c);
    }
    if (!r.ok) { err = r.err; return SCEMA_MD_ERR_IO; }
  }
  // state of fixes: global (id, style, n bytes), then the list of per-atom ones (id, style, values per atom)
  const int32_t nglobal = r.i32();
  for (int32_t k = 0; k < nglobal && r.ok; k++) {
    (void)r.str(); (void)r.str();
    const int32_t n = r.i32();
    if (n < 0 || !r.need((size_t)n)) { r.ok = false; break; }
    r.pos += (size_t)n;
  }
  const int32_t nperatom = r.i32();
  for (int32_t k = 0; k < nperatom && r.ok; k++) { (void)r.str(); (void)r.str(); (void)r.i32(); }
  if (!r.ok || walk(r, R) != -1) { err = r.err.empty() ? "bad file-layout section" : r.err; return SCEMA_MD_ERR_IO; }
  if (!want_atoms) return SCEMA_MD_OK;
  const bool full = R.atom_style == "full";
  if (!full && R.atom_style != "atomic") { err = "atom style " + R.atom_style + " not known to this reader"; return SCEMA_MD_ERR_ARG; }
  std::vector<double> buf;
  while (r.ok && r.pos < r.b.size()) {
    const int32_t flag = r.i32();
    if (flag != PERPROC) { err = "unexpected record " + std::to_string(flag) + " in the atom section"; return SCEMA_MD_ERR_IO; }
    r.dvec(buf);
    if (!r.ok) break;
    size_t m = 0;
    while (m < buf.size()) {
      const size_t sz = (size_t)buf[m];
      const size_t fixed = full ? 14 : 11;
      if (sz < fixed || m + sz > buf.size()) { err = "bad per-atom record"; return SCEMA_MD_ERR_IO; }
      const double *a = &buf[m];
      for (int c = 0; c < 3; c++) R.x.push_back(a[1 + c]);
      const int64_t tag = bits(a[4]);
      R.tag.push_back(tag);
      R.type.push_back((int32_t)bits(a[5]));
      R.image.push_back(bits(a[7]));
      for (int c = 0; c < 3; c++) R.v.push_back(a[8 + c]);
      if (full) {
        R.q.push_back(a[11]);
        size_t p = 13;   // a[12] = molecule id
        auto topo = [&](int na, std::vector<int64_t> &out) -> bool {
          if (p >= sz) return false;
          const int64_t n = bits(a[p++]);
          if (n < 0 || p + (size_t)n * (1 + na) > sz) return false;
          for (int64_t k = 0; k < n; k++) {
            out.push_back(bits(a[p++]));
            if (na == 1) { out.push_back(tag); out.push_back(bits(a[p++])); }   // a bond is stored with one of its atoms
            else for (int t = 0; t < na; t++) out.push_back(bits(a[p++]));
          }
          return true;
        };
        if (!topo(1, R.bonds) || !topo(3, R.angles) || !topo(4, R.dihedrals) || !topo(4, R.impropers)) { err = "bad per-atom topology record"; return SCEMA_MD_ERR_IO; }
      }
      m += sz;
    }
  }
  if (!r.ok) { err = r.err; return SCEMA_MD_ERR_IO; }
  if ((int64_t)R.tag.size() != R.natoms) { err = "atom count differs from the header"; return SCEMA_MD_ERR_IO; }
  return SCEMA_MD_OK;
}

void decode_image(const Restart &R, int64_t im, int out[3]) {
  if (R.imageint == 8) { out[0] = (int)(im & 2097151) - 1048576; out[1] = (int)((im >> 21) & 2097151) - 1048576; out[2] = (int)(im >> 42) - 1048576; }
  else { out[0] = (int)(im & 1023) - 512; out[1] = (int)((im >> 10) & 1023) - 512; out[2] = (int)((im >> 20) & 1023) - 512; }
}

// atom_style full + the OPLS styles -> scema_md_system (0-based indices in ascending tag order, unwrapped coordinates)
template <class Sink>
int to_system(const Restart &R, std::string &err, Sink sink) {
  if (R.atom_style != "full") { err = "replica restart must be atom_style full (found " + R.atom_style + ")"; return SCEMA_MD_ERR_ARG; }
  if (R.pair_style != "lj/cut/coul/long") { err = "replica restart must carry pair_style lj/cut/coul/long"; return SCEMA_MD_ERR_ARG; }
  if (R.units != "real") { err = "replica restart must be in units real (found " + R.units + ")"; return SCEMA_MD_ERR_ARG; }
  const size_t n = (size_t)R.natoms;
  const int nt = R.ntypes;
  if ((int)R.mass.size() != nt) { err = "mass array size"; return SCEMA_MD_ERR_IO; }
  std::vector<size_t> order(n);
  for (size_t i = 0; i < n; i++) order[i] = i;
  std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return R.tag[a] < R.tag[b]; });
  std::map<int64_t, int32_t> index_of;
  for (size_t k = 0; k < n; k++) index_of[R.tag[order[k]]] = (int32_t)k;
  if (index_of.size() != n) { err = "duplicate atom tags"; return SCEMA_MD_ERR_IO; }
  const double L[3] = {R.boxhi[0] - R.boxlo[0], R.boxhi[1] - R.boxlo[1], R.boxhi[2] - R.boxlo[2]};
  std::vector<int32_t> type(n);
  std::vector<double> q(n), x(3 * n), v(3 * n);
  for (size_t k = 0; k < n; k++) {
    const size_t i = order[k];
    if (R.type[i] < 1 || R.type[i] > nt) { err = "atom type out of range"; return SCEMA_MD_ERR_IO; }
    type[k] = R.type[i] - 1;
    q[k] = R.q[i];
    int im[3];
    decode_image(R, R.image[i], im);
    x[3 * k] = R.x[3 * i] + im[0] * L[0] + im[1] * R.tilt[0] + im[2] * R.tilt[1];
    x[3 * k + 1] = R.x[3 * i + 1] + im[1] * L[1] + im[2] * R.tilt[2];
    x[3 * k + 2] = R.x[3 * i + 2] + im[2] * L[2];
    for (int c = 0; c < 3; c++) v[3 * k + c] = R.v[3 * i + c];
  }
  // pair coefficients: explicit pairs as stored, the others mixed the way pair lj/cut/coul/long would (mix_flag)
  std::vector<double> eps((size_t)nt * nt), sig((size_t)nt * nt);
  for (int i = 0; i < nt; i++)
    for (int j = i; j < nt; j++) {
      double e, s;
      const size_t ij = (size_t)i * nt + j;
      if (R.pair_set[ij]) {
        e = R.pair_eps[ij]; s = R.pair_sigma[ij];
        if (std::fabs(R.pair_cut[ij] - R.cut_lj) > 1e-12) { err = "per-pair LJ cutoffs are not supported"; return SCEMA_MD_ERR_ARG; }
      } else {
        const size_t ii = (size_t)i * nt + i, jj = (size_t)j * nt + j;
        if (!R.pair_set[ii] || !R.pair_set[jj]) { err = "pair coefficients missing"; return SCEMA_MD_ERR_IO; }
        const double ei = R.pair_eps[ii], ej = R.pair_eps[jj], si = R.pair_sigma[ii], sj = R.pair_sigma[jj];
        if (R.mix_flag == 0) { e = std::sqrt(ei * ej); s = std::sqrt(si * sj); }
        else if (R.mix_flag == 1) { e = std::sqrt(ei * ej); s = 0.5 * (si + sj); }
        else {
          const double s6 = 0.5 * (std::pow(si, 6.0) + std::pow(sj, 6.0));
          e = 2.0 * std::sqrt(ei * ej) * std::pow(si, 3.0) * std::pow(sj, 3.0) / (std::pow(si, 6.0) + std::pow(sj, 6.0));
          s = std::pow(s6, 1.0 / 6.0);
        }
      }
      eps[(size_t)i * nt + j] = eps[(size_t)j * nt + i] = e;
      sig[(size_t)i * nt + j] = sig[(size_t)j * nt + i] = s;
    }
  // topology: (type, tags) tuples; every term once (with newton_bond off LAMMPS stores a term with each of its atoms)
  auto terms = [&](const std::vector<int64_t> &raw, int na, int ntp, std::vector<int32_t> &at, std::vector<int32_t> &tp) -> bool {
    std::set<std::vector<int64_t>> seen;
    for (size_t p = 0; p + 1 + na <= raw.size(); p += 1 + na) {
      std::vector<int64_t> key(raw.begin() + p, raw.begin() + p + 1 + na);
      if (!R.newton_bond) {   // stored with each of its atoms: keep the first copy (a bond's copies list the atoms both ways)
        std::vector<int64_t> rev(key);
        std::reverse(rev.begin() + 1, rev.end());
        if (seen.count(key) || (na == 2 && seen.count(rev))) continue;
        seen.insert(key);
      }
      const int64_t t = key[0] < 0 ? -key[0] : key[0];
      if (t < 1 || t > ntp) return false;
      tp.push_back((int32_t)t - 1);
      for (int a = 0; a < na; a++) {
        auto it = index_of.find(key[1 + a]);
        if (it == index_of.end()) return false;
        at.push_back(it->second);
      }
    }
    return true;
  };
  std::vector<int32_t> bat, btp, aat, atp, dat, dtp, iat, itp;
  if (!terms(R.bonds, 2, R.nbondtypes, bat, btp) || !terms(R.angles, 3, R.nangletypes, aat, atp) || !terms(R.dihedrals, 4, R.ndihedraltypes, dat, dtp) ||
      !terms(R.impropers, 4, R.nimpropertypes, iat, itp)) { err = "topology entry refers to an unknown atom or type"; return SCEMA_MD_ERR_IO; }
  if ((int64_t)btp.size() != R.nbonds || (int64_t)atp.size() != R.nangles || (int64_t)dtp.size() != R.ndihedrals || (int64_t)itp.size() != R.nimpropers) {
    err = "topology counts differ from the header";
    return SCEMA_MD_ERR_IO;
  }
  // coefficient arrays: file keeps one array per coefficient; ours are per type.  LAMMPS keeps half the opls K's.
  auto per_type = [](const std::vector<double> &c, int ntp, int nc, double scale) {
    std::vector<double> out((size_t)ntp * nc);
    for (int t = 0; t < ntp; t++)
      for (int k = 0; k < nc; k++) out[(size_t)t * nc + k] = scale * c[(size_t)k * ntp + t];
    return out;
  };
  if (R.nbonds && R.bond_c.empty()) { err = "bond coefficients missing"; return SCEMA_MD_ERR_IO; }
  if (R.nangles && R.angle_c.empty()) { err = "angle coefficients missing"; return SCEMA_MD_ERR_IO; }
  if (R.ndihedrals && R.dihedral_c.empty()) { err = "dihedral coefficients missing"; return SCEMA_MD_ERR_IO; }
  if (R.nimpropers && R.improper_c.empty()) { err = "improper coefficients missing"; return SCEMA_MD_ERR_IO; }
  std::vector<double> bc = R.bond_c.empty() ? std::vector<double>() : per_type(R.bond_c, R.nbondtypes, 2, 1.0);
  std::vector<double> ac = R.angle_c.empty() ? std::vector<double>() : per_type(R.angle_c, R.nangletypes, 2, 1.0);
  std::vector<double> dc = R.dihedral_c.empty() ? std::vector<double>() : per_type(R.dihedral_c, R.ndihedraltypes, 4, 2.0);
  std::vector<double> ic = R.improper_c.empty() ? std::vector<double>() : per_type(R.improper_c, R.nimpropertypes, 2, 1.0);
  scema_md_system s;
  std::memset(&s, 0, sizeof s);
  s.natoms = (int32_t)n; s.ntypes = nt;
  s.type = type.data(); s.charge = q.data(); s.mass = R.mass.data(); s.eps = eps.data(); s.sigma = sig.data();
  s.nbonds = (int32_t)btp.size(); s.nbondtypes = R.nbondtypes; s.bond_atoms = bat.data(); s.bond_type = btp.data(); s.bond_coeff = bc.data();
  s.nangles = (int32_t)atp.size(); s.nangletypes = R.nangletypes; s.angle_atoms = aat.data(); s.angle_type = atp.data(); s.angle_coeff = ac.data();
  s.ndihedrals = (int32_t)dtp.size(); s.ndihedraltypes = R.ndihedraltypes; s.dihedral_atoms = dat.data(); s.dihedral_type = dtp.data(); s.dihedral_coeff = dc.data();
  s.nimpropers = (int32_t)itp.size(); s.nimpropertypes = R.nimpropertypes; s.improper_atoms = iat.data(); s.improper_type = itp.data(); s.improper_coeff = ic.data();
  for (int k = 0; k < 3; k++) { s.special_lj[k] = R.special_lj[k]; s.special_coul[k] = R.special_coul[k]; }
  for (int k = 0; k < 3; k++) { s.box[k] = R.boxlo[k]; s.box[3 + k] = R.boxhi[k]; s.box[6 + k] = R.tilt[k]; }
  s.x = x.data(); s.v = v.data();
  return sink(s);
}

// ---- writer ----
struct Writer {
  FILE *f;
  bool ok = true;
  void raw(const void *p, size_t n) { if (ok && n && fwrite(p, 1, n, f) != n) ok = false; }
  void i32(int32_t v) { raw(&v, 4); }
  void flag_int(int flag, int32_t v) { i32(flag); i32(v); }
  void flag_big(int flag, int64_t v) { i32(flag); raw(&v, 8); }
  void flag_dbl(int flag, double v) { i32(flag); raw(&v, 8); }
  void str(const char *s) { const int32_t n = (int32_t)strlen(s) + 1; i32(n); raw(s, (size_t)n); }
  void flag_str(int flag, const char *s) { i32(flag); str(s); }
  void flag_dvec(int flag, int32_t n, const double *v) { i32(flag); i32(n); raw(v, (size_t)n * 8); }
  void flag_ivec(int flag, int32_t n, const int32_t *v) { i32(flag); i32(n); raw(v, (size_t)n * 4); }
};

}  // namespace

extern "C" {

int scema_md_probe_lammps_restart(const char *path, scema_lammps_restart_info *info) {
  if (!info) return SCEMA_MD_ERR_ARG;
  Restart R;
  std::string err;
  std::memset(info, 0, sizeof *info);
  const int rc = parse(path, R, err, false);
  snprintf(info->error, sizeof info->error, "%s", err.c_str());
  if (rc) return rc;
  snprintf(info->version, sizeof info->version, "%s", R.version.c_str());
  snprintf(info->units, sizeof info->units, "%s", R.units.c_str());
  snprintf(info->atom_style, sizeof info->atom_style, "%s", R.atom_style.c_str());
  snprintf(info->pair_style, sizeof info->pair_style, "%s", R.pair_style.c_str());
  info->natoms = R.natoms; info->ntimestep = R.ntimestep;
  info->nbonds = R.nbonds; info->nangles = R.nangles; info->ndihedrals = R.ndihedrals; info->nimpropers = R.nimpropers;
  info->ntypes = R.ntypes; info->nbondtypes = R.nbondtypes; info->nangletypes = R.nangletypes; info->ndihedraltypes = R.ndihedraltypes;
  info->nimpropertypes = R.nimpropertypes; info->triclinic = R.triclinic; info->nprocs = R.nprocs;
  for (int k = 0; k < 3; k++) { info->box[k] = R.boxlo[k]; info->box[3 + k] = R.boxhi[k]; info->box[6 + k] = R.tilt[k]; }
  info->timestep = R.timestep;
  for (int k = 0; k < 3; k++) { info->special_lj[k] = R.special_lj[k]; info->special_coul[k] = R.special_coul[k]; }
  info->cut_lj = R.cut_lj; info->cut_coul = R.cut_coul;
  for (size_t k = 0; k < R.mass.size() && k < 16; k++) info->mass[k] = R.mass[k];
  return SCEMA_MD_OK;
}

int scema_md_read_lammps_restart_atoms(const char *path, int64_t capacity, int64_t *tag, int32_t *type, int32_t *image, double *x, double *v) {
  Restart R;
  std::string err;
  const int rc = parse(path, R, err, true);
  if (rc) return rc;
  if (R.natoms > capacity) return SCEMA_MD_ERR_ARG;
  for (int64_t i = 0; i < R.natoms; i++) {
    if (tag) tag[i] = R.tag[(size_t)i];
    if (type) type[i] = R.type[(size_t)i];
    if (image) { int im[3]; decode_image(R, R.image[(size_t)i], im); for (int c = 0; c < 3; c++) image[3 * i + c] = im[c]; }
    for (int c = 0; c < 3; c++) {
      if (x) x[3 * i + c] = R.x[3 * (size_t)i + c];
      if (v) v[3 * i + c] = R.v[3 * (size_t)i + c];
    }
  }
  return SCEMA_MD_OK;
}

int scema_md_load_lammps_restart(scema_md_engine *e, const char *matid, int32_t replica, const char *path) {
  if (!e || !matid) return SCEMA_MD_ERR_ARG;
  Restart R;
  std::string err;
  int rc = parse(path, R, err, true);
  if (!rc) rc = to_system(R, err, [&](const scema_md_system &s) { return scema_md_register_replica(e, matid, replica, &s); });
  if (rc && !err.empty()) fprintf(stderr, "[scema_md] %s: %s\n", path ? path : "(null)", err.c_str());
  return rc;
}

int scema_md_convert_lammps_restart(const char *restart_path, const char *replica_path) {
  if (!replica_path) return SCEMA_MD_ERR_ARG;
  Restart R;
  std::string err;
  int rc = parse(restart_path, R, err, true);
  if (!rc) rc = to_system(R, err, [&](const scema_md_system &s) { return scema_md_write_replica_file(replica_path, &s); });
  if (rc && !err.empty()) fprintf(stderr, "[scema_md] %s: %s\n", restart_path ? restart_path : "(null)", err.c_str());
  return rc;
}

int scema_md_write_lammps_restart(const char *path, const scema_md_system *s, double cut_lj, double cut_coul, double timestep, int64_t ntimestep) {
  if (!path || !s || s->natoms <= 0 || s->ntypes <= 0) return SCEMA_MD_ERR_ARG;
  FILE *f = fopen(path, "wb");
  if (!f) return SCEMA_MD_ERR_IO;
  Writer w{f};
  const int n = s->natoms, nt = s->ntypes;
  // per-atom topology the way read_data assigns it with newton_bond on: bond -> first atom, angle / dihedral / improper -> second atom
  std::vector<std::vector<int>> ob(n), oa(n), od(n), oi(n);
  for (int k = 0; k < s->nbonds; k++) ob[s->bond_atoms[2 * k]].push_back(k);
  for (int k = 0; k < s->nangles; k++) oa[s->angle_atoms[3 * k + 1]].push_back(k);
  for (int k = 0; k < s->ndihedrals; k++) od[s->dihedral_atoms[4 * k + 1]].push_back(k);
  for (int k = 0; k < s->nimpropers; k++) oi[s->improper_atoms[4 * k + 1]].push_back(k);
  size_t mb = 0, ma = 0, md = 0, mi = 0;
  for (int i = 0; i < n; i++) { mb = std::max(mb, ob[i].size()); ma = std::max(ma, oa[i].size()); md = std::max(md, od[i].size()); mi = std::max(mi, oi[i].size()); }
  w.raw(MAGIC, 16);
  w.i32(1);
  w.i32(0);
  w.flag_str(VERSION, "17 Nov 2016");
  w.flag_int(SMALLINT, 4); w.flag_int(IMAGEINT, 4); w.flag_int(TAGINT, 4); w.flag_int(BIGINT, 8);
  w.flag_str(UNITS, "real");
  w.flag_big(NTIMESTEP, ntimestep);
  w.flag_int(DIMENSION, 3);
  w.flag_int(NPROCS, 1);
  const int32_t grid[3] = {1, 1, 1}, zero6[6] = {0, 0, 0, 0, 0, 0};
  const double dzero6[6] = {0, 0, 0, 0, 0, 0};
  w.flag_ivec(PROCGRID, 3, grid);
  w.flag_int(NEWTON_PAIR, 1); w.flag_int(NEWTON_BOND, 1);
  w.flag_int(XPERIODIC, 1); w.flag_int(YPERIODIC, 1); w.flag_int(ZPERIODIC, 1);
  w.flag_ivec(BOUNDARY, 6, zero6);
  w.flag_dvec(BOUNDMIN, 6, dzero6);
  w.flag_str(ATOM_STYLE, "full");
  w.i32(0);
  w.flag_big(NATOMS, n);
  w.flag_int(NTYPES, nt);
  w.flag_big(NBONDS, s->nbonds); w.flag_int(NBONDTYPES, s->nbondtypes); w.flag_int(BOND_PER_ATOM, (int32_t)mb);
  w.flag_big(NANGLES, s->nangles); w.flag_int(NANGLETYPES, s->nangletypes); w.flag_int(ANGLE_PER_ATOM, (int32_t)ma);
  w.flag_big(NDIHEDRALS, s->ndihedrals); w.flag_int(NDIHEDRALTYPES, s->ndihedraltypes); w.flag_int(DIHEDRAL_PER_ATOM, (int32_t)md);
  w.flag_big(NIMPROPERS, s->nimpropers); w.flag_int(NIMPROPERTYPES, s->nimpropertypes); w.flag_int(IMPROPER_PER_ATOM, (int32_t)mi);
  w.flag_int(TRICLINIC, 1);
  w.flag_dvec(BOXLO, 3, s->box); w.flag_dvec(BOXHI, 3, s->box + 3);
  w.flag_dbl(XY, s->box[6]); w.flag_dbl(XZ, s->box[7]); w.flag_dbl(YZ, s->box[8]);
  w.flag_dvec(SPECIAL_LJ, 3, s->special_lj); w.flag_dvec(SPECIAL_COUL, 3, s->special_coul);
  w.flag_dbl(TIMESTEP, timestep);
  w.flag_int(ATOM_ID, 1); w.flag_int(ATOM_MAP_STYLE, 0); w.flag_int(ATOM_MAP_USER, 0); w.flag_int(ATOM_SORTFREQ, 1000);
  w.flag_dbl(ATOM_SORTBIN, 0.0);
  w.flag_int(COMM_MODE, 0); w.flag_dbl(COMM_CUTOFF, 0.0); w.flag_int(COMM_VEL, 0);
  w.i32(-1);
  w.i32(1); w.str("all");   // groups
  w.flag_dvec(MASS, nt, s->mass);
  w.i32(-1);
  w.flag_str(PAIR, "lj/cut/coul/long");
  {
    const int32_t offset_flag = 0, mix_flag = 0, tail_flag = 0, ncoultablebits = 12;
    const double tabinner = std::sqrt(2.0);
    w.raw(&cut_lj, 8); w.raw(&cut_coul, 8);
    w.i32(offset_flag); w.i32(mix_flag); w.i32(tail_flag); w.i32(ncoultablebits);
    w.raw(&tabinner, 8);
    for (int i = 0; i < nt; i++)
      for (int j = i; j < nt; j++) {
        w.i32(1);
        w.raw(&s->eps[(size_t)i * nt + j], 8); w.raw(&s->sigma[(size_t)i * nt + j], 8); w.raw(&cut_lj, 8);
      }
  }
  auto coeffs = [&](int flag, const char *style, int ntp, int nc, const double *c, double scale) {
    if (ntp <= 0) return;
    w.flag_str(flag, style);
    for (int k = 0; k < nc; k++)
      for (int t = 0; t < ntp; t++) { const double v = scale * c[(size_t)t * nc + k]; w.raw(&v, 8); }
  };
  coeffs(BOND, "harmonic", s->nbondtypes, 2, s->bond_coeff, 1.0);
  coeffs(ANGLE, "harmonic", s->nangletypes, 2, s->angle_coeff, 1.0);
  coeffs(DIHEDRAL, "opls", s->ndihedraltypes, 4, s->dihedral_coeff, 0.5);
  coeffs(IMPROPER, "harmonic", s->nimpropertypes, 2, s->improper_coeff, 1.0);
  w.i32(-1);
  w.i32(0); w.i32(0);   // no fix state
  w.flag_int(MULTIPROC, 0); w.flag_int(MPIIO, 0);
  w.i32(-1);
  // atoms: wrapped into the box, image flags carry the rest
  const double *B = s->box;
  const double L[3] = {B[3] - B[0], B[4] - B[1], B[5] - B[2]};
  std::vector<double> buf;
  for (int i = 0; i < n; i++) {
    double p[3] = {s->x[3 * i], s->x[3 * i + 1], s->x[3 * i + 2]};
    int im[3];
    im[2] = (int)std::floor((p[2] - B[2]) / L[2]);
    p[2] -= im[2] * L[2]; p[1] -= im[2] * B[8]; p[0] -= im[2] * B[7];
    im[1] = (int)std::floor((p[1] - B[1]) / L[1]);
    p[1] -= im[1] * L[1]; p[0] -= im[1] * B[6];
    im[0] = (int)std::floor((p[0] - B[0]) / L[0]);
    p[0] -= im[0] * L[0];
    for (int c = 0; c < 3; c++)
      if (im[c] < -512 || im[c] > 511) { fclose(f); return SCEMA_MD_ERR_ARG; }
    const int64_t image = ((int64_t)(im[2] + 512) << 20) | ((int64_t)(im[1] + 512) << 10) | (int64_t)(im[0] + 512);
    const size_t start = buf.size();
    buf.push_back(0.0);
    for (int c = 0; c < 3; c++) buf.push_back(p[c]);
    buf.push_back(unbits(i + 1)); buf.push_back(unbits(s->type[i] + 1)); buf.push_back(unbits(1)); buf.push_back(unbits(image));
    for (int c = 0; c < 3; c++) buf.push_back(s->v ? s->v[3 * i + c] : 0.0);
    buf.push_back(s->charge ? s->charge[i] : 0.0);
    buf.push_back(unbits(0));   // molecule id
    buf.push_back(unbits((int64_t)ob[i].size()));
    for (int k : ob[i]) { buf.push_back(unbits(s->bond_type[k] + 1)); buf.push_back(unbits(s->bond_atoms[2 * k + 1] + 1)); }
    buf.push_back(unbits((int64_t)oa[i].size()));
    for (int k : oa[i]) { buf.push_back(unbits(s->angle_type[k] + 1)); for (int a = 0; a < 3; a++) buf.push_back(unbits(s->angle_atoms[3 * k + a] + 1)); }
    buf.push_back(unbits((int64_t)od[i].size()));
    for (int k : od[i]) { buf.push_back(unbits(s->dihedral_type[k] + 1)); for (int a = 0; a < 4; a++) buf.push_back(unbits(s->dihedral_atoms[4 * k + a] + 1)); }
    buf.push_back(unbits((int64_t)oi[i].size()));
    for (int k : oi[i]) { buf.push_back(unbits(s->improper_type[k] + 1)); for (int a = 0; a < 4; a++) buf.push_back(unbits(s->improper_atoms[4 * k + a] + 1)); }
    buf[start] = (double)(buf.size() - start);
  }
  if (buf.size() > 0x7fffffffu) { fclose(f); return SCEMA_MD_ERR_ARG; }
  w.flag_dvec(PERPROC, (int32_t)buf.size(), buf.data());
  const bool ok = w.ok;
  if (fclose(f) != 0 || !ok) return SCEMA_MD_ERR_IO;
  return SCEMA_MD_OK;
}

}  // extern "C"
