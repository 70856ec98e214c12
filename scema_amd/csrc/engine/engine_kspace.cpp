// engine_kspace.cpp -- what a LAMMPS `run` sets up on the host: g_ewald and the k-vector set, the PPPM grid, the polynomial of the real-space factor, fix deform's box path
#include "engine.h"
#include "../md_env.h"

namespace scema_eng {

// -------------------------------------------------------------------------------------------
// Ewald run parameters from the current box: g_ewald rule of "kspace_style pppm <acc>"
// (in.set.lammps:36) and the k-vector set of the reciprocal sum it approximates
// -------------------------------------------------------------------------------------------
void ewald_setup(const scema_md_params &p, const Topo &t, const double *box, EwaldSetup &out, bool g_only) {
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
  if (const char *tv = scema_env("SCEMA_MD_POLY_TOL")) target = std::min(1e-3, std::max(1e-15, atof(tv)));
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
void pppm_setup_host(const scema_md_params &p, const Topo &t, const double *box, double &g, int pg[3]) {
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
double cached_coul_poly(scema_md_engine *, double g, double rc, double *poly, int *npoly, double *uscale) {
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

}  // namespace scema_eng

// ---- the k-space set-up as a pure host function of the C ABI (no GPU, no engine): what a run would use for this box ----
// The parity tests compare the engine with the C oracle, whose PPPM set-up was written by the same hand as pppm_setup_host above;
// this entry lets a line-by-line Python restatement of PPPM::set_grid_global / adjust_gewald (tests/test_oracle_pppm.py) check the
// PRODUCT's answer directly (the chain product -> C oracle -> Python was circular in its 43 shared lines).
extern "C" int scema_md_kspace_setup(const scema_md_params *p, const double *box, double qsqsum, int32_t natoms, double *g_initial,
                                     double *g_ewald, int32_t *grid) {
  if (!p || !box || natoms <= 0 || !(qsqsum >= 0.0)) return SCEMA_MD_ERR_ARG;
  scema_eng::Topo t;
  t.natoms = natoms;
  t.qsqsum = qsqsum;
  scema_eng::EwaldSetup ew;
  scema_eng::ewald_setup(*p, t, box, ew, true);   // LAMMPS' initial estimate (the Ewald sum keeps it)
  if (g_initial) *g_initial = ew.g;
  double g = ew.g;
  int pg[3] = {0, 0, 0};
  if (p->kspace_style == 1 && qsqsum > 0.0) scema_eng::pppm_setup_host(*p, t, box, g, pg);
  if (g_ewald) *g_ewald = g;
  if (grid) { grid[0] = pg[0]; grid[1] = pg[1]; grid[2] = pg[2]; }
  return SCEMA_MD_OK;
}
