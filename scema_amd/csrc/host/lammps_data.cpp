// lammps_data.cpp -- reader for LAMMPS text data files of atom_style full and atom_style charge (the output of
// `write_data`), so that a replica equilibrated with the reference's own in.init.lammps can be
// handed to the engine without LAMMPS' binary restart format.
//
// Understood sections: header counts and box (incl. "xy xz yz"), Masses, Pair Coeffs (eps sigma per
// type, mixed geometrically as pair lj/cut/coul/long does by default), PairIJ Coeffs (explicit
// pairs override), Bond/Angle/Dihedral/Improper Coeffs (harmonic K r0 | harmonic K theta0[deg] |
// opls K1..K4 | harmonic K chi0[deg]), Atoms (id mol type q x y z [ix iy iz]), Velocities,
// Bonds, Angles, Dihedrals, Impropers.  special_bonds is not stored in data files: the caller
// passes the weights (the reference uses lj/coul 0 0 1, in.init.lammps:31).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../../include/scema_md.h"

namespace {

std::string trim(const std::string &s) {
  size_t b = s.find_first_not_of(" \t\r\n"), e = s.find_last_not_of(" \t\r\n");
  return b == std::string::npos ? "" : s.substr(b, e - b + 1);
}
std::string strip_comment(const std::string &s) {
  size_t p = s.find('#');
  return trim(p == std::string::npos ? s : s.substr(0, p));
}
std::vector<std::string> split(const std::string &s) {
  std::vector<std::string> out;
  std::istringstream is(s);
  std::string w;
  while (is >> w) out.push_back(w);
  return out;
}

}  // namespace

// parse `path`; on success call sink(system) while the parsed arrays are alive
template <class Sink>
static int parse_lammps_data(const char *path, const double special_lj[3], const double special_coul[3], Sink sink) {
  if (!path) return SCEMA_MD_ERR_ARG;
  std::ifstream f(path);
  if (!f.is_open()) return SCEMA_MD_ERR_IO;
  std::vector<std::string> lines;
  for (std::string l; std::getline(f, l);) lines.push_back(l);
  if (lines.empty()) return SCEMA_MD_ERR_IO;
  long natoms = 0, nbonds = 0, nangles = 0, ndih = 0, nimp = 0;
  int ntypes = 0, nbt = 0, nat = 0, ndt = 0, nit = 0;
  double box[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  static const char *SECTIONS[] = {"Masses", "Pair Coeffs", "PairIJ Coeffs", "Bond Coeffs", "Angle Coeffs", "Dihedral Coeffs",
                                   "Improper Coeffs", "Atoms", "Velocities", "Bonds", "Angles", "Dihedrals", "Impropers",
                                   "BondBond Coeffs", "BondAngle Coeffs", "MiddleBondTorsion Coeffs", "EndBondTorsion Coeffs",
                                   "AngleTorsion Coeffs", "AngleAngleTorsion Coeffs", "BondBond13 Coeffs", "AngleAngle Coeffs"};
  auto section_of = [&](const std::string &l) -> std::string {
    for (const char *s : SECTIONS)
      if (l == s || l.rfind(std::string(s) + " ", 0) == 0) return s;
    return "";
  };
  size_t i = 1;  // first line is a title
  // ---- header ----
  for (; i < lines.size(); i++) {
    const std::string l = strip_comment(lines[i]);
    if (l.empty()) continue;
    if (!section_of(l).empty()) break;
    const std::vector<std::string> w = split(l);
    auto ends = [&](const char *a, const char *b = nullptr) {
      if (b) return w.size() >= 3 && w[w.size() - 2] == a && w.back() == b;
      return w.size() >= 2 && w.back() == a;
    };
    if (ends("atoms")) natoms = atol(w[0].c_str());
    else if (ends("bonds")) nbonds = atol(w[0].c_str());
    else if (ends("angles")) nangles = atol(w[0].c_str());
    else if (ends("dihedrals")) ndih = atol(w[0].c_str());
    else if (ends("impropers")) nimp = atol(w[0].c_str());
    else if (ends("atom", "types")) ntypes = atoi(w[0].c_str());
    else if (ends("bond", "types")) nbt = atoi(w[0].c_str());
    else if (ends("angle", "types")) nat = atoi(w[0].c_str());
    else if (ends("dihedral", "types")) ndt = atoi(w[0].c_str());
    else if (ends("improper", "types")) nit = atoi(w[0].c_str());
    else if (ends("xlo", "xhi")) { box[0] = atof(w[0].c_str()); box[3] = atof(w[1].c_str()); }
    else if (ends("ylo", "yhi")) { box[1] = atof(w[0].c_str()); box[4] = atof(w[1].c_str()); }
    else if (ends("zlo", "zhi")) { box[2] = atof(w[0].c_str()); box[5] = atof(w[1].c_str()); }
    else if (w.size() >= 6 && w[3] == "xy" && w[4] == "xz" && w[5] == "yz") { box[6] = atof(w[0].c_str()); box[7] = atof(w[1].c_str()); box[8] = atof(w[2].c_str()); }
  }
  if (natoms <= 0 || ntypes <= 0) return SCEMA_MD_ERR_IO;
  // Every counted item is a line of the file: a count beyond the number of lines (or a negative one) is a damaged header, and it must
  // be refused BEFORE it sizes an allocation (a header that says 999 999 999 atoms used to end in std::bad_alloc -> terminate; found
  // by tests/test_corrupt_files.py).  Atom ids index a 32-bit table downstream.
  {
    const long nl = (long)lines.size();
    const long counts[] = {natoms, nbonds, nangles, ndih, nimp, ntypes, nbt, nat, ndt, nit};
    for (long c : counts)
      if (c < 0 || c > nl) return SCEMA_MD_ERR_IO;
  }
  std::vector<double> mass(ntypes, 0.0), eps1(ntypes, 0.0), sig1(ntypes, 0.0), eps((size_t)ntypes * ntypes, -1.0), sig((size_t)ntypes * ntypes, 0.0);
  std::vector<double> bc(2 * (size_t)nbt), ac(2 * (size_t)nat), dc(4 * (size_t)ndt), ic(2 * (size_t)nit);
  std::vector<int32_t> type(natoms), bat(2 * (size_t)nbonds), btp(nbonds), aat(3 * (size_t)nangles), atp(nangles), dat(4 * (size_t)ndih), dtp(ndih),
      iat(4 * (size_t)nimp), itp(nimp);
  std::vector<double> q(natoms), x(3 * (size_t)natoms), v(3 * (size_t)natoms, 0.0);
  std::map<long, int> index_of;  // atom id -> 0-based index (ids need not be contiguous or ordered)
  std::vector<long> ids(natoms);
  const double DEG = std::acos(-1.0) / 180.0;

  const double L[3] = {box[3] - box[0], box[4] - box[1], box[5] - box[2]};
  // ---- sections ----
  struct Pending { std::vector<std::vector<std::string>> rows; };
  std::map<std::string, Pending> sec;
  std::string cur;
  // atom_style of the Atoms section: write_data names it in a comment ("Atoms # charge"); without one the column count decides
  // (charge: id type q x y z [ix iy iz]; full: id mol type q x y z [ix iy iz]).  charge = the reax scripts (in.set.lammps:17)
  int style_charge = -1;
  for (; i < lines.size(); i++) {
    const std::string l = strip_comment(lines[i]);
    if (l.empty()) continue;
    const std::string s = section_of(l);
    if (s == "Atoms") {
      if (lines[i].find("charge") != std::string::npos) style_charge = 1;
      else if (lines[i].find("full") != std::string::npos) style_charge = 0;
    }
    if (!s.empty()) { cur = s; continue; }
    if (!cur.empty()) sec[cur].rows.push_back(split(l));
  }
  auto need = [&](const char *name, size_t n, size_t minw) -> bool {
    auto it = sec.find(name);
    if (n == 0) return true;
    if (it == sec.end() || it->second.rows.size() != n) return false;
    for (auto &r : it->second.rows) if (r.size() < minw) return false;
    return true;
  };
  if (!need("Masses", ntypes, 2) || !need("Atoms", natoms, 6)) return SCEMA_MD_ERR_IO;
  if (style_charge < 0) {
    const size_t w = sec["Atoms"].rows[0].size();
    style_charge = (w == 6 || w == 9) ? 1 : 0;
  }
  const int c0 = style_charge ? 1 : 2;   // column of the atom type
  if (!need("Atoms", natoms, (size_t)c0 + 5)) return SCEMA_MD_ERR_IO;
  for (auto &r : sec["Masses"].rows) { int t = atoi(r[0].c_str()); if (t < 1 || t > ntypes) return SCEMA_MD_ERR_IO; mass[t - 1] = atof(r[1].c_str()); }
  if (sec.count("Pair Coeffs")) {
    for (auto &r : sec["Pair Coeffs"].rows) { if (r.size() < 3) return SCEMA_MD_ERR_IO; int t = atoi(r[0].c_str()); if (t < 1 || t > ntypes) return SCEMA_MD_ERR_IO; eps1[t - 1] = atof(r[1].c_str()); sig1[t - 1] = atof(r[2].c_str()); }
    for (int a = 0; a < ntypes; a++)
      for (int b = 0; b < ntypes; b++) { eps[(size_t)a * ntypes + b] = std::sqrt(eps1[a] * eps1[b]); sig[(size_t)a * ntypes + b] = std::sqrt(sig1[a] * sig1[b]); }
  }
  if (sec.count("PairIJ Coeffs"))
    for (auto &r : sec["PairIJ Coeffs"].rows) {
      if (r.size() < 4) return SCEMA_MD_ERR_IO;
      int a = atoi(r[0].c_str()) - 1, b = atoi(r[1].c_str()) - 1;
      if (a < 0 || b < 0 || a >= ntypes || b >= ntypes) return SCEMA_MD_ERR_IO;
      eps[(size_t)a * ntypes + b] = eps[(size_t)b * ntypes + a] = atof(r[2].c_str());
      sig[(size_t)a * ntypes + b] = sig[(size_t)b * ntypes + a] = atof(r[3].c_str());
    }
  if (style_charge && !sec.count("Pair Coeffs") && !sec.count("PairIJ Coeffs")) {
    // a ReaxFF replica: the force field comes from pair_coeff * * ffield.reax.2 ..., the data file carries none
    std::fill(eps.begin(), eps.end(), 0.0);
    std::fill(sig.begin(), sig.end(), 1.0);
  }
  for (double ev : eps) if (ev < 0.0) return SCEMA_MD_ERR_IO;  // no pair coefficients at all
  auto coeffs = [&](const char *name, int n, int ncoef, std::vector<double> &out, int deg_col) -> bool {
    if (n == 0) return true;
    if (!need(name, n, 1 + ncoef)) return false;
    for (auto &r : sec[name].rows) {
      int t = atoi(r[0].c_str());
      if (t < 1 || t > n) return false;
      for (int k = 0; k < ncoef; k++) out[(size_t)(t - 1) * ncoef + k] = atof(r[1 + k].c_str()) * (k == deg_col ? DEG : 1.0);
    }
    return true;
  };
  if (!coeffs("Bond Coeffs", nbt, 2, bc, -1) || !coeffs("Angle Coeffs", nat, 2, ac, 1) || !coeffs("Dihedral Coeffs", ndt, 4, dc, -1) ||
      !coeffs("Improper Coeffs", nit, 2, ic, 1))
    return SCEMA_MD_ERR_IO;
  {
    long k = 0;
    for (auto &r : sec["Atoms"].rows) {
      const long id = atol(r[0].c_str());
      ids[k] = id;
      index_of[id] = (int)k;
      const int t = atoi(r[c0].c_str());
      if (t < 1 || t > ntypes) return SCEMA_MD_ERR_IO;
      type[k] = t - 1;
      q[k] = atof(r[c0 + 1].c_str());
      double p[3] = {atof(r[c0 + 2].c_str()), atof(r[c0 + 3].c_str()), atof(r[c0 + 4].c_str())};
      if (r.size() >= (size_t)c0 + 8) {  // image flags: unwrap (the engine keeps unwrapped coordinates)
        const int ix = atoi(r[c0 + 5].c_str()), iy = atoi(r[c0 + 6].c_str()), iz = atoi(r[c0 + 7].c_str());
        p[0] += ix * L[0] + iy * box[6] + iz * box[7];
        p[1] += iy * L[1] + iz * box[8];
        p[2] += iz * L[2];
      }
      for (int c = 0; c < 3; c++) x[3 * k + c] = p[c];
      k++;
    }
    if ((long)index_of.size() != natoms) return SCEMA_MD_ERR_IO;
  }
  if (sec.count("Velocities"))
    for (auto &r : sec["Velocities"].rows) {
      if (r.size() < 4) return SCEMA_MD_ERR_IO;
      auto it = index_of.find(atol(r[0].c_str()));
      if (it == index_of.end()) return SCEMA_MD_ERR_IO;
      for (int c = 0; c < 3; c++) v[3 * (size_t)it->second + c] = atof(r[1 + c].c_str());
    }
  auto topo = [&](const char *name, long n, int na, int ntp, std::vector<int32_t> &at, std::vector<int32_t> &tp) -> bool {
    if (n == 0) return true;
    if (!need(name, n, 2 + na)) return false;
    long k = 0;
    for (auto &r : sec[name].rows) {
      const int t = atoi(r[1].c_str());
      if (t < 1 || t > ntp) return false;
      tp[k] = t - 1;
      for (int a = 0; a < na; a++) {
        auto it = index_of.find(atol(r[2 + a].c_str()));
        if (it == index_of.end()) return false;
        at[(size_t)na * k + a] = it->second;
      }
      k++;
    }
    return true;
  };
  if (!topo("Bonds", nbonds, 2, nbt, bat, btp) || !topo("Angles", nangles, 3, nat, aat, atp) || !topo("Dihedrals", ndih, 4, ndt, dat, dtp) ||
      !topo("Impropers", nimp, 4, nit, iat, itp))
    return SCEMA_MD_ERR_IO;
  scema_md_system s;
  std::memset(&s, 0, sizeof s);
  s.natoms = (int32_t)natoms; s.ntypes = ntypes;
  s.type = type.data(); s.charge = q.data(); s.mass = mass.data(); s.eps = eps.data(); s.sigma = sig.data();
  s.nbonds = (int32_t)nbonds; s.nbondtypes = nbt; s.bond_atoms = bat.data(); s.bond_type = btp.data(); s.bond_coeff = bc.data();
  s.nangles = (int32_t)nangles; s.nangletypes = nat; s.angle_atoms = aat.data(); s.angle_type = atp.data(); s.angle_coeff = ac.data();
  s.ndihedrals = (int32_t)ndih; s.ndihedraltypes = ndt; s.dihedral_atoms = dat.data(); s.dihedral_type = dtp.data(); s.dihedral_coeff = dc.data();
  s.nimpropers = (int32_t)nimp; s.nimpropertypes = nit; s.improper_atoms = iat.data(); s.improper_type = itp.data(); s.improper_coeff = ic.data();
  for (int k = 0; k < 3; k++) { s.special_lj[k] = special_lj ? special_lj[k] : (k == 2 ? 1.0 : 0.0); s.special_coul[k] = special_coul ? special_coul[k] : (k == 2 ? 1.0 : 0.0); }
  std::memcpy(s.box, box, sizeof box);
  s.x = x.data(); s.v = v.data();
  return sink(s);
}

extern "C" {

int scema_md_load_lammps_data(scema_md_engine *e, const char *matid, int32_t replica, const char *path, const double special_lj[3],
                              const double special_coul[3]) {
  if (!e || !matid) return SCEMA_MD_ERR_ARG;
  try {
    return parse_lammps_data(path, special_lj, special_coul, [&](const scema_md_system &s) { return scema_md_register_replica(e, matid, replica, &s); });
  } catch (const std::exception &) {   // (allocation failure on a damaged file: an error code, never an exception across the C ABI)
    return SCEMA_MD_ERR_IO;
  }
}

int scema_md_convert_lammps_data(const char *data_path, const char *replica_path, const double special_lj[3], const double special_coul[3]) {
  if (!replica_path) return SCEMA_MD_ERR_ARG;
  try {
    return parse_lammps_data(data_path, special_lj, special_coul, [&](const scema_md_system &s) { return scema_md_write_replica_file(replica_path, &s); });
  } catch (const std::exception &) {
    return SCEMA_MD_ERR_IO;
  }
}

}  // extern "C"
