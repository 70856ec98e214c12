// reax_ffield.cpp -- see reax_ffield.h.  File layout (ReaxFF user manual; what USER-REAXC's Read_Force_Field walks
// [LAMMPS-ext]): comment line; general parameters; atoms (4 lines each); bonds (2 lines each); off-diagonal terms; valence
// angles; torsions (0-j-k-0 = wildcard, a specific quadruple wins whatever the order); hydrogen bonds.
#include "reax_ffield.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace scema {
namespace {

struct Line {
  std::string word;             // leading non-numeric token, if any
  std::vector<double> v;        // the numbers up to the first non-numeric token after them
};

Line split(const std::string &s) {
  Line l;
  std::istringstream is(s);
  std::string tok;
  bool first = true;
  while (is >> tok) {
    char *end = nullptr;
    const double d = std::strtod(tok.c_str(), &end);
    const bool num = end != tok.c_str() && *end == 0;
    if (num) l.v.push_back(d);
    else if (first) l.word = tok;
    else break;
    first = false;
  }
  return l;
}

struct AtomRec { std::string name; double a[8], b[8], c[8], d[8]; };

}  // namespace

bool read_reax_ffield(const std::string &path, const std::vector<std::string> &elements, RxParams &P, std::vector<int> &type_map, std::string &err) {
  std::ifstream in(path);
  if (!in) { err = "cannot open force-field file " + path; return false; }
  std::vector<std::string> lines;
  for (std::string s; std::getline(in, s);) lines.push_back(s);
  size_t at = 1;   // line 0 is a comment
  auto next = [&](Line &l, size_t need) -> bool {
    if (at >= lines.size()) { err = path + ": unexpected end of file"; return false; }
    l = split(lines[at++]);
    if (l.v.size() < need) { err = path + ": line " + std::to_string(at) + " has too few numbers"; return false; }
    return true;
  };
  // a count read as a double: it must be a whole number of lines that the file still has (a damaged file may hold NaN, 1e300 or -7
  // there; (int) of those is undefined, and the atom count sizes an allocation)
  auto count = [&](double d, int &out) -> bool {
    if (!(d >= 0.0) || d > (double)lines.size()) { err = path + ": line " + std::to_string(at) + " holds a count that the file cannot contain"; return false; }
    out = (int)d;
    return true;
  };
  Line l;
  std::memset(&P, 0, sizeof(P));
  P.lammps_dsbo2 = 0;
  if (!next(l, 1)) return false;
  int ngp = 0;
  if (!count(l.v[0], ngp)) return false;
  for (int k = 0; k < ngp; k++) {
    if (!next(l, 1)) return false;
    if (k < RX_NGP) P.gp[k] = l.v[0];
  }
  if (!next(l, 1)) return false;
  int nat = 0;
  if (!count(l.v[0], nat)) return false;
  at += 3;   // the rest of the atom block's header
  std::vector<AtomRec> atoms(nat);
  for (int i = 0; i < nat; i++) {
    AtomRec &r = atoms[i];
    if (!next(l, 8)) return false;
    r.name = l.word;
    for (int k = 0; k < 8; k++) r.a[k] = l.v[k];
    if (!next(l, 8)) return false;
    for (int k = 0; k < 8; k++) r.b[k] = l.v[k];
    if (!next(l, 8)) return false;
    for (int k = 0; k < 8; k++) r.c[k] = l.v[k];
    if (!next(l, 8)) return false;
    for (int k = 0; k < 8; k++) r.d[k] = l.v[k];
  }
  // which of the file's types are kept, and as what
  std::vector<int> compact(nat, -1), kept;
  type_map.assign(elements.size(), -1);
  for (size_t k = 0; k < elements.size(); k++) {
    int f = -1;
    for (int i = 0; i < nat; i++)
      if (atoms[i].name == elements[k]) { f = i; break; }
    if (f < 0) { err = "element " + elements[k] + " is not in " + path; return false; }
    if (compact[f] < 0) {
      if ((int)kept.size() >= RX_MAXT) { err = "more than " + std::to_string(RX_MAXT) + " distinct elements"; return false; }
      compact[f] = (int)kept.size();
      kept.push_back(f);
    }
    type_map[k] = compact[f];
  }
  const int nt = P.nt = (int)kept.size();
  for (int c = 0; c < nt; c++) {
    const AtomRec &r = atoms[kept[c]];
    RxSbp &s = P.sbp[c];
    s.r_s = r.a[0]; s.valency = r.a[1]; s.mass = r.a[2]; s.r_vdw = r.a[3]; s.epsilon = r.a[4]; s.gamma = r.a[5]; s.r_pi = r.a[6]; s.valency_e = r.a[7];
    s.nlp_opt = 0.5 * (s.valency_e - s.valency);
    s.alpha = r.b[0]; s.gamma_w = r.b[1]; s.valency_boc = r.b[2]; s.p_ovun5 = r.b[3]; s.chi = r.b[5]; s.eta = 2.0 * r.b[6]; s.p_hbond = (r.b[7] >= 0.0 && r.b[7] < 100.0) ? (int)r.b[7] : 0;
    s.r_pi_pi = r.c[0]; s.p_lp2 = r.c[1]; s.b_o_131 = r.c[3]; s.b_o_132 = r.c[4]; s.b_o_133 = r.c[5];
    s.p_ovun2 = r.d[0]; s.p_val3 = r.d[1]; s.valency_val = r.d[3]; s.p_val5 = r.d[4];
    if (r.d[5] > 0.0 || r.d[6] > 0.0) { err = "inner-wall van der Waals parameters (rcore, ecore) are not supported"; return false; }
    if (s.mass < 21.0 && s.valency_val != s.valency_boc) s.valency_val = s.valency_boc;
  }
  if (P.gp[10] != 0.0 || P.gp[5] > 0.001) { err = "triple-bond stabilisation / C2 correction switched on in " + path + ": not supported"; return false; }
  auto T2 = [&](int a, int b) -> RxTbp & { return P.tbp[a * RX_MAXT + b]; };
  for (int a = 0; a < nt; a++)
    for (int b = 0; b < nt; b++) {
      RxTbp &t = T2(a, b);
      const RxSbp &x = P.sbp[a], &y = P.sbp[b];
      t.r_s = 0.5 * (x.r_s + y.r_s); t.r_p = 0.5 * (x.r_pi + y.r_pi); t.r_pp = 0.5 * (x.r_pi_pi + y.r_pi_pi);
      t.p_boc3 = std::sqrt(x.b_o_132 * y.b_o_132); t.p_boc4 = std::sqrt(x.b_o_131 * y.b_o_131); t.p_boc5 = std::sqrt(x.b_o_133 * y.b_o_133);
      t.D = std::sqrt(x.epsilon * y.epsilon); t.alpha = std::sqrt(x.alpha * y.alpha); t.r_vdW = 2.0 * std::sqrt(x.r_vdw * y.r_vdw);
      t.gamma_w = std::sqrt(x.gamma_w * y.gamma_w); t.gamma = std::pow(x.gamma * y.gamma, -1.5);
    }
  auto C = [&](double v) -> int { if (!(v >= 1.0) || v > (double)nat) return -1; return compact[(int)v - 1]; };
  // bonds
  if (!next(l, 1)) return false;
  int nbond = 0;
  if (!count(l.v[0], nbond)) return false;
  at += 1;
  for (int m = 0; m < nbond; m++) {
    Line l1, l2;
    if (!next(l1, 10) || !next(l2, 7)) return false;
    const int a = C(l1.v[0]), b = C(l1.v[1]);
    if (a < 0 || b < 0) continue;
    for (int side = 0; side < 2; side++) {
      RxTbp &t = side ? T2(b, a) : T2(a, b);
      t.De_s = l1.v[2]; t.De_p = l1.v[3]; t.De_pp = l1.v[4]; t.p_be1 = l1.v[5]; t.p_bo5 = l1.v[6]; t.v13cor = l1.v[7]; t.p_bo6 = l1.v[8]; t.p_ovun1 = l1.v[9];
      t.p_be2 = l2.v[0]; t.p_bo3 = l2.v[1]; t.p_bo4 = l2.v[2]; t.p_bo1 = l2.v[4]; t.p_bo2 = l2.v[5]; t.ovc = l2.v[6];
    }
  }
  // off-diagonal
  if (!next(l, 1)) return false;
  int noff = 0;
  if (!count(l.v[0], noff)) return false;
  for (int m = 0; m < noff; m++) {
    if (!next(l, 8)) return false;
    const int a = C(l.v[0]), b = C(l.v[1]);
    if (a < 0 || b < 0) continue;
    for (int side = 0; side < 2; side++) {
      RxTbp &t = side ? T2(b, a) : T2(a, b);
      if (l.v[2] > 0.0) t.D = l.v[2];
      if (l.v[3] > 0.0) t.r_vdW = 2.0 * l.v[3];
      if (l.v[4] > 0.0) t.alpha = l.v[4];
      if (l.v[5] > 0.0) t.r_s = l.v[5];
      if (l.v[6] > 0.0) t.r_p = l.v[6];
      if (l.v[7] > 0.0) t.r_pp = l.v[7];
    }
  }
  // valence angles
  if (!next(l, 1)) return false;
  int nang = 0;
  if (!count(l.v[0], nang)) return false;
  for (int m = 0; m < nang; m++) {
    if (!next(l, 10)) return false;
    const int a = C(l.v[0]), b = C(l.v[1]), c = C(l.v[2]);
    if (a < 0 || b < 0 || c < 0) continue;
    RxThbp &t1 = P.thbp[(a * RX_MAXT + b) * RX_MAXT + c], &t2 = P.thbp[(c * RX_MAXT + b) * RX_MAXT + a];
    if (t1.cnt >= RX_MAXANG) { err = "more than " + std::to_string(RX_MAXANG) + " parameter sets for one valence angle"; return false; }
    const RxThbPrm prm = {l.v[3], l.v[4], l.v[5], l.v[6], l.v[7], l.v[8], l.v[9]};
    const int n = t1.cnt;
    t1.prm[n] = prm; t1.cnt = n + 1;
    if (&t1 != &t2) { t2.prm[n] = prm; t2.cnt = n + 1; }
  }
  // torsions
  if (!next(l, 1)) return false;
  int ntor = 0;
  if (!count(l.v[0], ntor)) return false;
  std::vector<char> specific((size_t)RX_MAXT * RX_MAXT * RX_MAXT * RX_MAXT, 0);
  auto Q = [&](int a, int b, int c, int d) -> size_t { return ((size_t)(a * RX_MAXT + b) * RX_MAXT + c) * RX_MAXT + d; };
  for (int m = 0; m < ntor; m++) {
    if (!next(l, 9)) return false;
    const int fa = (l.v[0] >= 1.0 && l.v[0] <= (double)nat) ? (int)l.v[0] : (l.v[0] == 0.0 ? 0 : -1), fd = (l.v[3] >= 1.0 && l.v[3] <= (double)nat) ? (int)l.v[3] : (l.v[3] == 0.0 ? 0 : -1);
    const int b = C(l.v[1]), c = C(l.v[2]);
    if (b < 0 || c < 0) continue;
    auto set = [&](size_t q) { RxFbp &f = P.fbp[q]; f.cnt = 1; f.V1 = l.v[4]; f.V2 = l.v[5]; f.V3 = l.v[6]; f.p_tor1 = l.v[7]; f.p_cot1 = l.v[8]; };
    if (fa > 0 && fd > 0) {
      const int a = C(l.v[0]), d = C(l.v[3]);
      if (a < 0 || d < 0) continue;
      set(Q(a, b, c, d)); set(Q(d, c, b, a));
      specific[Q(a, b, c, d)] = specific[Q(d, c, b, a)] = 1;
    } else if (fa == 0 && fd == 0) {
      for (int a = 0; a < nt; a++)
        for (int d = 0; d < nt; d++) {
          if (!specific[Q(a, b, c, d)]) set(Q(a, b, c, d));
          if (!specific[Q(d, c, b, a)]) set(Q(d, c, b, a));
        }
    }
  }
  // hydrogen bonds
  if (!next(l, 1)) return false;
  int nhb = 0;
  if (!count(l.v[0], nhb)) return false;
  for (int m = 0; m < nhb; m++) {
    if (!next(l, 7)) return false;
    const int a = C(l.v[0]), b = C(l.v[1]), c = C(l.v[2]);
    if (a < 0 || b < 0 || c < 0) continue;
    RxHbp &h = P.hbp[(a * RX_MAXT + b) * RX_MAXT + c];
    h.r0_hb = l.v[3]; h.p_hb1 = l.v[4]; h.p_hb2 = l.v[5]; h.p_hb3 = l.v[6];
  }
  for (int a = 0; a < RX_MAXT * RX_MAXT; a++) {
    RxTbp &t = P.tbp[a];
    t.powgw = (t.gamma_w > 0.0) ? std::pow(1.0 / t.gamma_w, P.gp[28]) : 0.0;
    t.lr_s = (t.r_s > 0.0) ? std::log(t.r_s) : 0.0;
    t.lr_p = (t.r_p > 0.0) ? std::log(t.r_p) : 0.0;
    t.lr_pp = (t.r_pp > 0.0) ? std::log(t.r_pp) : 0.0;
    t.inv_rvdw = (t.r_vdW > 0.0) ? 1.0 / t.r_vdW : 0.0;
  }
  P.inv_pvdw1 = 1.0 / P.gp[28];
  P.bo_cut = 0.01 * P.gp[29];
  P.swa = P.gp[11];
  P.swb = P.gp[12];
  {  // 7th-order taper: 1 at swa, 0 at swb, three vanishing derivatives at both
    const double a = P.swa, b = P.swb, d7 = std::pow(b - a, 7.0);
    P.tap[7] = 20.0 / d7;
    P.tap[6] = -70.0 * (a + b) / d7;
    P.tap[5] = 84.0 * (a * a + 3.0 * a * b + b * b) / d7;
    P.tap[4] = -35.0 * (a * a * a + 9.0 * a * a * b + 9.0 * a * b * b + b * b * b) / d7;
    P.tap[3] = 140.0 * (a * a * a * b + 3.0 * a * a * b * b + a * b * b * b) / d7;
    P.tap[2] = -210.0 * (a * a * a * b * b + a * a * b * b * b) / d7;
    P.tap[1] = 140.0 * a * a * a * b * b * b / d7;
    P.tap[0] = (-35.0 * a * a * a * b * b * b * b + 21.0 * a * a * std::pow(b, 5.0) - 7.0 * a * std::pow(b, 6.0) + std::pow(b, 7.0)) / d7;
  }
  return true;
}

}  // namespace scema
