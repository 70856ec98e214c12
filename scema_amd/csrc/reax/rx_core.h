// rx_core.h -- the ReaxFF arithmetic, one function per (pass, atom): bond orders and their corrections, every energy term
// with its derivatives, and the back-propagation of dE/d(bond order) to forces.  The kernels of md_reax.hip call these
// with one lane per atom; tests/reax_host_driver.cpp compiles the same functions for the host (RX_HOST_TEST) so that every
// derivative can be checked against central differences of the oracle's energy without a GPU.
//
// Functional forms: USER-REAXC of LAMMPS 17Nov16 [LAMMPS-ext] (reaxc_bond_orders.cpp BOp/BO, reaxc_bonds.cpp,
// reaxc_multi_body.cpp Atom_Energy, reaxc_valence_angles.cpp, reaxc_torsion_angles.cpp, reaxc_hydrogen_bonds.cpp,
// reaxc_nonbonded.cpp), selected by lammps_scripts_reax/in.strain.lammps:10-12.  The derivative bookkeeping is this
// repository's own: reverse mode in two sweeps (energy terms gather dE/dBO and dE/dDelta, one pass takes them through the
// bond-order corrections to dE/dBO' and dE/dDelta', one pass turns those into forces), O(bonds) instead of USER-REAXC's
// O(bonds x neighbours) Add_dBond_to_Forces.
#pragma once
#include <math.h>
#include <string.h>

#include "rx_types.h"

#if defined(RX_HOST_TEST)
#define RX_FN static inline
#define RX_ATOMIC_ADD(p, v) (*(p) += (v))
#define RX_ATOMIC_OR(p, v) (*(p) |= (v))
#else
#define RX_FN __device__ __forceinline__
// (the builtins take pointers of any address space: global_atomic_* through the qualified pointers of RxView; relaxed, device scope, as atomicAdd / atomicOr)
#define RX_ATOMIC_ADD(p, v) ((void)__hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
#define RX_ATOMIC_OR(p, v) ((void)__hip_atomic_fetch_or((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
#endif

// Atom-indexed sums of the angle and torsion items (forces, dE/dDelta): with LACC they go to a workgroup's tables in LDS -- lf[3][npad], lcd[npad],
// added to the work set once per workgroup by the kernel -- instead of one device-wide atomic each (a torsion item has 26 of them, 20 of these two
// kinds; the pass was bound by them: 310 us per 72-replica launch against 106 us with the atomics compiled out, round 5).
#if defined(RX_HOST_TEST)
#define RX_LDS_ADD(p, v) (*(p) += (v))
#else
#define RX_LDS_ADD(p, v) ((void)__hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
#endif
#ifdef __cplusplus
#define RX_DEFAULT_NULL = nullptr
#else
#define RX_DEFAULT_NULL
#endif
#define RX_SQR(x) ((x) * (x))
#define RX_PI 3.14159265358979323846

typedef struct {
  double Delta, Delta_e, Delta_boc, Delta_val, vlpex, nlp, Delta_lp, dDelta_lp, Delta_lp_temp, dDelta_lp_temp;
} RxAtomD;

RX_FN void rx_shift(const RxView *V, int e, double s[3]) {
  const int code = (e >> 24) & 0x7F;
  const int sx = code % 5 - 2, sy = (code / 5) % 5 - 2, sz = code / 25 - 2;
  s[0] = sx * V->h[0] + sy * V->h[5] + sz * V->h[4];
  s[1] = sy * V->h[1] + sz * V->h[3];
  s[2] = sz * V->h[2];
}
// vector from atom i to the partner named by row entry e
// entry k of atom i's full neighbour row: the device keeps the rows row-major only (nbT), the host test driver entry-major (nb)
RX_FN int rx_nb_entry(const RxView *V, int i, int k) { return V->nbT ? V->nbT[(size_t)i * V->maxnb + k] : V->nb[(size_t)k * V->npad + i]; }
// one packed matrix entry (RxView::hpk)
RX_FN unsigned long long rx_hpack(double h, int col) {
  unsigned long long b;
  __builtin_memcpy(&b, &h, 8);
  return ((b + 0x8000ull) & ~0xFFFFull) | (unsigned long long)(col & 0xFFFF);
}
RX_FN double rx_hunpack(unsigned long long b, int *col) {
  *col = (int)(b & 0xFFFFull);
  b &= ~0xFFFFull;
  double h;
  __builtin_memcpy(&h, &b, 8);
  return h;
}
RX_FN int rx_partner(const RxView *V, int i, int e, double d[3]) {
  const int j = e & RX_JMASK;
  double s[3];
  rx_shift(V, e, s);
  d[0] = V->x[3 * j] - V->x[3 * i] + s[0];
  d[1] = V->x[3 * j + 1] - V->x[3 * i + 1] + s[1];
  d[2] = V->x[3 * j + 2] - V->x[3 * i + 2] + s[2];
  return j;
}
// the bond (i -> e) is handled by this end when the partner ranks higher (each bond once)
RX_FN int rx_owns(int i, int e) { const int j = e & RX_JMASK; return j > i || (j == i && ((e >> 24) & 0x7F) > RX_CODE0); }

RX_FN void rx_vt(double *v, const double *a, const double *f) {
  v[0] += a[0] * f[0]; v[1] += a[1] * f[1]; v[2] += a[2] * f[2]; v[3] += a[0] * f[1]; v[4] += a[0] * f[2]; v[5] += a[1] * f[2];
}
// x^(-1/3), x > 0.  On the device: the FP32 estimate exp2(-log2(x) / 3) (two hardware transcendentals) and two Newton steps
// y <- y (4 - x y^3) / 3 (quadratic: 1e-6 -> 2e-12 -> below the rounding of FP64) instead of the library's cbrt and a division.
RX_FN double rx_icbrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = (double)__builtin_amdgcn_exp2f(-0.33333334f * __builtin_amdgcn_logf((float)x));
  y = y * fma(-x * y, y * y, 4.0) * 0.33333333333333331;
  y = y * fma(-x * y, y * y, 4.0) * 0.33333333333333331;
  return y;
#else
  return 1.0 / cbrt(x);
#endif
}
// ---- FP64 elementary functions at the accuracy this path needs (a few ulp), for positive finite normal arguments only ----------------
// The library's log / pow / division / sqrt carry the IEEE corner cases and, for log and pow, double-double arithmetic: 85 and ~200
// instructions a call on the device, more than a third of the non-bonded pair and most of a bond-order entry.  Every argument here is a
// distance, a bond order above its threshold or a sum of positive terms, so: reciprocal and reciprocal square root from the hardware
// estimate (2^-26) and one third-order correction; log x by the classical reduction x = 2^k m, m in [sqrt(1/2), sqrt(2)),
// log m = 2 atanh(s), s = (m - 1)/(m + 1), written f - f^2/2 + s (f^2/2 + R(s^2)) with the series R(z) = sum 2 z^n/(2n+1) cut
// after n = 10 (z <= 0.0295: remainder 1e-18); x^p = exp(p log x).  The host build (tests/reax_host_driver.cpp) takes the library
// functions unless RX_DEVICE_MATH_ON_HOST asks for these algorithms with single-precision seeds in place of the hardware estimates, so
// that the formulas themselves are checked against the oracle without a GPU (tests/test_reax_host.py).
#if defined(__HIP_DEVICE_COMPILE__)
#define RX_DEVICE_MATH 1
#define RX_RCP_SEED(x) __builtin_amdgcn_rcp(x)
#define RX_RSQ_SEED(x) __builtin_amdgcn_rsq(x)
#elif defined(RX_DEVICE_MATH_ON_HOST)
#define RX_DEVICE_MATH 1
#define RX_RCP_SEED(x) ((double)(1.0f / (float)(x)))
#define RX_RSQ_SEED(x) ((double)(1.0f / sqrtf((float)(x))))
#endif
RX_FN double rx_rcp(double x) {
#ifdef RX_DEVICE_MATH
  const double y = RX_RCP_SEED(x);
  const double e = fma(-x, y, 1.0);
  return fma(y, fma(e, e, e), y);     // y (1 + e + e^2)
#else
  return 1.0 / x;
#endif
}
RX_FN double rx_rsqrt(double x) {
#ifdef RX_DEVICE_MATH
  const double y = RX_RSQ_SEED(x);
  const double e = fma(-x * y, y, 1.0);
  return fma(y, e * fma(0.375, e, 0.5), y);   // y (1 + e/2 + 3 e^2/8)
#else
  return 1.0 / sqrt(x);
#endif
}
RX_FN double rx_log(double x) {
#ifdef RX_DEVICE_MATH
#if defined(__HIP_DEVICE_COMPILE__)
  double m = __builtin_amdgcn_frexp_mant(x);
  int k = __builtin_amdgcn_frexp_exp(x);
#else
  int k;
  double m = frexp(x, &k);
#endif
  if (m < 0.70710678118654752) { m += m; k -= 1; }
  const double f = m - 1.0;
  const double s = f * rx_rcp(2.0 + f), z = s * s;
  double R = 2.0 / 21.0;
  R = fma(R, z, 2.0 / 19.0); R = fma(R, z, 2.0 / 17.0); R = fma(R, z, 2.0 / 15.0); R = fma(R, z, 2.0 / 13.0); R = fma(R, z, 2.0 / 11.0);
  R = fma(R, z, 2.0 / 9.0); R = fma(R, z, 2.0 / 7.0); R = fma(R, z, 2.0 / 5.0); R = fma(R, z, 2.0 / 3.0);
  R *= z;
  const double hfsq = 0.5 * f * f, dk = (double)k;
  // ln 2 in two parts: the first has 21 trailing zero bits, so its product with dk is exact
  return dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
#else
  return log(x);
#endif
}
RX_FN double rx_pow(double x, double p) {
#ifdef RX_DEVICE_MATH
  return exp(p * rx_log(x));
#else
  return pow(x, p);
#endif
}
// sqrt(x) for x >= 0 that is zero or a normal number
RX_FN double rx_sqrt(double x) {
#ifdef RX_DEVICE_MATH
  return (x > 0.0) ? x * rx_rsqrt(x) : 0.0;
#else
  return sqrt(x);
#endif
}
// sin(theta) of an angle in [0, pi] from its (clamped) cosine: sqrt((1 - c)(1 + c)) -- no acos, no sin
RX_FN double rx_sin_of_cos(double c) { return rx_sqrt((1.0 - c) * (1.0 + c)); }
template <bool LACC>
RX_FN void rx_add_f(const RxView *V, double *lf, int atom, int m, double v) {
  if (LACC) RX_LDS_ADD(&lf[(size_t)m * V->npad + atom], v);
  else RX_ATOMIC_ADD(&V->f[3 * atom + m], v);
}
template <bool LACC>
RX_FN void rx_add_cd(const RxView *V, double *lcd, int atom, double v) {
  if (LACC) RX_LDS_ADD(&lcd[atom], v);
  else RX_ATOMIC_ADD(&V->cd_delta[atom], v);
}
RX_FN double rx_taper(const RxParams *P, double r, double *dtap) {
  double t = P->tap[7], dt = 7.0 * P->tap[7];
  for (int m = 6; m >= 0; m--) t = t * r + P->tap[m];
  for (int m = 6; m >= 1; m--) dt = dt * r + m * P->tap[m];
  *dtap = dt;
  return t;
}

RX_FN void rx_atom_deltas(const RxParams *P, int type, double total_bo, RxAtomD *D) {
  const RxSbp *s = &P->sbp[type];
  const double p_lp1 = P->gp[15];
  D->Delta = total_bo - s->valency;
  D->Delta_e = total_bo - s->valency_e;
  D->Delta_boc = total_bo - s->valency_boc;
  D->Delta_val = total_bo - s->valency_val;
  const int half = (int)(D->Delta_e / 2.0);
  D->vlpex = D->Delta_e - 2.0 * half;
  const double explp1 = exp(-p_lp1 * RX_SQR(2.0 + D->vlpex));
  D->nlp = explp1 - half;
  D->Delta_lp = s->nlp_opt - D->nlp;
  D->dDelta_lp = 2.0 * p_lp1 * explp1 * (2.0 + D->vlpex);
  if (s->mass > 21.0) {
    D->Delta_lp_temp = s->nlp_opt - 0.5 * (s->valency_e - s->valency);
    D->dDelta_lp_temp = 0.0;
  } else {
    D->Delta_lp_temp = D->Delta_lp;
    D->dDelta_lp_temp = D->dDelta_lp;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// pass 1: uncorrected bond orders of atom i from its neighbour row (BOp of reaxc_bond_orders.cpp)
// ------------------------------------------------------------------------------------------------------------------
// uncorrected bond order of atom i with the partner of near-row entry e: BO' (total, cutoff NOT yet taken off), its pi parts, r and
// the coefficients of d in dBO'/dd; returns 0 when the pair is beyond the bond cutoff or BO' below the threshold
// (sbp, tbp: the type tables, e.g. staged in LDS; NULL: the ones of P)
// ... the arithmetic alone: types ti, tj, squared distance r2 (the kernel gathers the partner's record a chunk ahead); the type tables are
// the caller's (staged in LDS by k_rx_bonds, P's own for everyone else)
template <class SBP, class TBP>
RX_FN int rx_bond_prime_pair(const RxParams *P, const SBP *sbp, const TBP *tbp, int ti, int tj, double r2, double *bo, double *bp, double *bpp, double *r_out,
                             double *cs, double *cp, double *cpp) {
  if (r2 > RX_BOND_CUT * RX_BOND_CUT) return 0;
  const SBP *si = &sbp[ti], *sj = &sbp[tj];
  const TBP *t = &tbp[ti * RX_MAXT + tj];
  // (r / r_x)^p = exp(p (log r - log r_x)): ONE logarithm for the three bond orders of the entry (RxTbp::lr_*)
  const double rinv = rx_rsqrt(r2), r = r2 * rinv, r2inv = rinv * rinv, lr = 0.5 * rx_log(r2);
  double bs = 0;
  *bp = 0; *bpp = 0; *cs = 0; *cp = 0; *cpp = 0;
  if (si->r_s > 0.0 && sj->r_s > 0.0) {
    const double c12 = t->p_bo1 * exp(t->p_bo2 * (lr - t->lr_s));
    bs = (1.0 + P->bo_cut) * exp(c12);
    *cs = bs * t->p_bo2 * c12 * r2inv;
  }
  if (si->r_pi > 0.0 && sj->r_pi > 0.0) {
    const double c34 = t->p_bo3 * exp(t->p_bo4 * (lr - t->lr_p));
    *bp = exp(c34);
    *cp = *bp * t->p_bo4 * c34 * r2inv;
  }
  if (si->r_pi_pi > 0.0 && sj->r_pi_pi > 0.0) {
    const double c56 = t->p_bo5 * exp(t->p_bo6 * (lr - t->lr_pp));
    *bpp = exp(c56);
    *cpp = *bpp * t->p_bo6 * c56 * r2inv;
  }
  *bo = bs + *bp + *bpp;
  *r_out = r;
  return *bo >= P->bo_cut;
}
RX_FN void rx_bonds_prime(const RxParams *P, const RxView *V, int i) {
  const int np = V->npad, ti = V->rtype[i];
  const RxSbp *si = &P->sbp[ti];
  const size_t plane = (size_t)V->maxbd * np;
  int nb = 0;
  double sum = 0.0;
  // the near rows hold the entries of the neighbour rows that can come inside the bond cutoff before the next rebuild, in the same order
  const int *row = V->nbn ? V->nbn : V->nb;
  const int cnt = V->nbn ? V->nbn_cnt[i] : V->nb_cnt[i];
  for (int k = 0; k < cnt; k++) {
    const int e = row[(size_t)k * np + i];
    double d[3];
    const int j = rx_partner(V, i, e, d);
    double bo, bp, bpp, r, cs, cp, cpp;
    if (!rx_bond_prime_pair(P, P->sbp, P->tbp, ti, V->rtype[j], d[0] * d[0] + d[1] * d[1] + d[2] * d[2], &bo, &bp, &bpp, &r, &cs, &cp, &cpp)) continue;
    if (nb >= V->maxbd) { RX_ATOMIC_OR(V->overflow, 2); break; }
    const size_t o = (size_t)nb * np + i;
    V->bd[o] = e;
    V->bd_bop[o] = bo - P->bo_cut;
    V->bd_bop[plane + o] = bp;
    V->bd_bop[2 * plane + o] = bpp;
    V->bd_bop[3 * plane + o] = r;
    V->bd_c[o] = cs;
    V->bd_c[plane + o] = cp;
    V->bd_c[2 * plane + o] = cpp;
    sum += bo - P->bo_cut;
    nb++;
  }
  V->bd_cnt[i] = nb;
  V->deltap[i] = sum - si->valency;
}

// pass 1b: where does the partner keep this bond?
RX_FN void rx_bonds_rev(const RxView *V, int i) {
  const int np = V->npad, cnt = V->bd_cnt[i];
  for (int k = 0; k < cnt; k++) {
    const int e = V->bd[(size_t)k * np + i];
    const int j = e & RX_JMASK;
    const int want = i | ((124 - ((e >> 24) & 0x7F)) << 24);
    const int cj = V->bd_cnt[j];
    int rev = -1;
    for (int m = 0; m < cj; m++)
      if (V->bd[(size_t)m * np + j] == want) { rev = m; break; }
    V->bd_rev[(size_t)k * np + i] = rev;   // -1 only if the partner's row overflowed (flagged there)
  }
}

// the correction factors of one bond and their derivatives
typedef struct { double Y, X, Yi, Yj, Xi, Xj, Xb; } RxCorr;   // Y = f1, X = f4 f5; _i/_j: d/dDelta'_i, _j; Xb: dX/dBO'
RX_FN void rx_corr(const RxParams *P, int ti, int tj, double Di, double Dj, double B, RxCorr *c) {
  const RxTbp *t = &P->tbp[ti * RX_MAXT + tj];
  const RxSbp *si = &P->sbp[ti], *sj = &P->sbp[tj];
  c->Y = 1.0; c->X = 1.0; c->Yi = c->Yj = c->Xi = c->Xj = c->Xb = 0.0;
  if (t->ovc >= 0.001) {
    const double p1 = P->gp[0], p2 = P->gp[1];
    const double e1i = exp(-p1 * Di), e1j = exp(-p1 * Dj), e2i = exp(-p2 * Di), e2j = exp(-p2 * Dj);
    const double f2 = e1i + e1j, f3 = -1.0 / p2 * rx_log(0.5 * (e2i + e2j));
    const double vi = si->valency, vj = sj->valency;
    const double ui = rx_rcp(vi + f2 + f3), uj = rx_rcp(vj + f2 + f3);
    c->Y = 0.5 * ((vi + f2) * ui + (vj + f2) * uj);
    // d/dD of (v + f2)/(v + f2 + f3) = (f2' f3 - (v + f2) f3') / (v + f2 + f3)^2
    const double f2i = -p1 * e1i, f2j = -p1 * e1j, ie2 = rx_rcp(e2i + e2j), f3i = e2i * ie2, f3j = e2j * ie2;
    c->Yi = 0.5 * ((f2i * f3 - (vi + f2) * f3i) * ui * ui + (f2i * f3 - (vj + f2) * f3i) * uj * uj);
    c->Yj = 0.5 * ((f2j * f3 - (vi + f2) * f3j) * ui * ui + (f2j * f3 - (vj + f2) * f3j) * uj * uj);
  }
  if (t->v13cor >= 0.001) {
    const double Dbi = Di + si->valency - si->valency_boc, Dbj = Dj + sj->valency - sj->valency_boc;
    const double E4 = exp(-(t->p_boc4 * B * B - Dbi) * t->p_boc3 + t->p_boc5), E5 = exp(-(t->p_boc4 * B * B - Dbj) * t->p_boc3 + t->p_boc5);
    const double f4 = rx_rcp(1.0 + E4), f5 = rx_rcp(1.0 + E5);
    c->X = f4 * f5;
    const double f4i = -t->p_boc3 * E4 * f4 * f4, f5j = -t->p_boc3 * E5 * f5 * f5;
    c->Xi = f4i * f5;
    c->Xj = f4 * f5j;
    const double k = 2.0 * t->p_boc3 * t->p_boc4 * B;
    c->Xb = k * E4 * f4 * f4 * f5 + f4 * k * E5 * f5 * f5;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// pass 2: corrected bond orders (BO of reaxc_bond_orders.cpp); zeroes the gather arrays of this atom's row
// ------------------------------------------------------------------------------------------------------------------
RX_FN void rx_bonds_corrected(const RxParams *P, const RxView *V, int i) {
  const int np = V->npad, ti = V->rtype[i], cnt = V->bd_cnt[i];
  const size_t plane = (size_t)V->maxbd * np;
  const double Di = V->deltap[i];
  double sum = 0.0;
  for (int k = 0; k < cnt; k++) {
    const size_t o = (size_t)k * np + i;
    const int j = V->bd[o] & RX_JMASK;
    const double B = V->bd_bop[o], Bp = V->bd_bop[plane + o], Bpp = V->bd_bop[2 * plane + o];
    RxCorr c;
    rx_corr(P, ti, V->rtype[j], Di, V->deltap[j], B, &c);
    const double A0 = c.Y * c.X, A1 = A0 * c.Y;
    double bo = B * A0, bp = Bp * A1, bpp = Bpp * A1;
    if (bo < 1e-10) bo = 0.0;
    if (bp < 1e-10) bp = 0.0;
    if (bpp < 1e-10) bpp = 0.0;
    V->bd_bo[o] = bo; V->bd_bo[plane + o] = bp; V->bd_bo[2 * plane + o] = bpp;
    V->bd_g[o] = 0.0; V->bd_g[plane + o] = 0.0; V->bd_g[2 * plane + o] = 0.0;
    sum += bo;
  }
  V->total_bo[i] = sum;
  V->cd_delta[i] = 0.0;
  V->hd[i] = 0.0;
  V->f[3 * i] = 0.0; V->f[3 * i + 1] = 0.0; V->f[3 * i + 2] = 0.0;
}

// ------------------------------------------------------------------------------------------------------------------
// pass 3a: bond energies (Bonds) of the bonds this end owns + lone pair, over-, under-coordination (Atom_Energy) of atom i
// ------------------------------------------------------------------------------------------------------------------
RX_FN void rx_atom_terms(const RxParams *P, const RxView *V, int i, double *eng) {
  const int np = V->npad, ti = V->rtype[i], cnt = V->bd_cnt[i];
  const size_t plane = (size_t)V->maxbd * np;
  const RxSbp *s = &P->sbp[ti];
  RxAtomD A;
  rx_atom_deltas(P, ti, V->total_bo[i], &A);
  const double p_ovun3 = P->gp[32], p_ovun4 = P->gp[31], p_ovun6 = P->gp[6], p_ovun7 = P->gp[8], p_ovun8 = P->gp[9];
  // lone pair
  const double expvd2 = exp(-75.0 * A.Delta_lp), inv2 = 1.0 / (1.0 + expvd2);
  eng[RX_E_LP] += s->p_lp2 * A.Delta_lp * inv2;
  const double dElp = s->p_lp2 * inv2 + 75.0 * s->p_lp2 * A.Delta_lp * expvd2 * inv2 * inv2;
  double cdd = dElp * A.dDelta_lp;   // d(Delta_lp)/d(Delta) = dDelta_lp
  // sums over the bonds
  const double dfvl = (s->mass > 21.0) ? 0.0 : 1.0;
  double sum1 = 0.0, sum2 = 0.0;
  for (int k = 0; k < cnt; k++) {
    const size_t o = (size_t)k * np + i;
    const int e = V->bd[o], j = e & RX_JMASK, tj = V->rtype[j];
    const RxTbp *t = &P->tbp[ti * RX_MAXT + tj];
    const double bo = V->bd_bo[o], bpi = V->bd_bo[plane + o], bpi2 = V->bd_bo[2 * plane + o];
    if (rx_owns(i, e)) {   // bond energy, once per bond
      const double bs = bo - bpi - bpi2;
      const double pw = (bs > 0.0) ? rx_pow(bs, t->p_be2) : 0.0;
      const double ex = exp(t->p_be1 * (1.0 - pw));
      eng[RX_E_BOND] += -t->De_s * bs * ex - t->De_p * bpi - t->De_pp * bpi2;
      const double CEbo = -t->De_s * ex * (1.0 - t->p_be1 * t->p_be2 * pw);
      V->bd_g[o] += CEbo;                      // dE/dBO at fixed BO_pi, BO_pi2 (BO_s = BO - BO_pi - BO_pi2)
      V->bd_g[plane + o] += -CEbo - t->De_p;
      V->bd_g[2 * plane + o] += -CEbo - t->De_pp;
    }
    RxAtomD J;
    rx_atom_deltas(P, tj, V->total_bo[j], &J);
    sum1 += t->p_ovun1 * t->De_s * bo;
    sum2 += (J.Delta - dfvl * J.Delta_lp_temp) * (bpi + bpi2);
  }
  const double e1 = p_ovun3 * exp(p_ovun4 * sum2), inv1 = 1.0 / (1.0 + e1);
  const double Dlc = A.Delta - dfvl * A.Delta_lp_temp * inv1;
  const double e2 = exp(s->p_ovun2 * Dlc), inv_e2 = 1.0 / (1.0 + e2);
  const double DlpVi = 1.0 / (Dlc + s->valency + 1e-8);
  const double CEover1 = Dlc * DlpVi * inv_e2;
  eng[RX_E_OVER] += sum1 * CEover1;
  const double CEover2 = sum1 * DlpVi * inv_e2 * (1.0 - Dlc * (DlpVi + s->p_ovun2 * e2 * inv_e2));   // dE_over/dDlc
  const double e2n = 1.0 / e2, e6 = exp(p_ovun6 * Dlc), e8 = p_ovun7 * exp(p_ovun8 * sum2);
  const double inv_e2n = 1.0 / (1.0 + e2n), inv_e8 = 1.0 / (1.0 + e8);
  const double e_un = -s->p_ovun5 * (1.0 - e6) * inv_e2n * inv_e8;
  eng[RX_E_UNDER] += e_un;
  const double CEunder1 = inv_e2n * (s->p_ovun5 * p_ovun6 * e6 * inv_e8 + s->p_ovun2 * e_un * e2n);    // dE_under/dDlc
  const double CEunder2 = -e_un * p_ovun8 * e8 * inv_e8;                                                   // dE_under/dsum2 (direct)
  // dDlc/dDelta_i = 1 - dfvl dDelta_lp_temp inv1 ; dDlc/dsum2 = dfvl Delta_lp_temp p_ovun4 e1 inv1^2
  const double dD = 1.0 - dfvl * A.dDelta_lp_temp * inv1;
  const double dS2 = dfvl * A.Delta_lp_temp * p_ovun4 * e1 * inv1 * inv1;
  cdd += (CEover2 + CEunder1) * dD;
  const double c4 = (CEover2 + CEunder1) * dS2 + CEunder2;   // dE/dsum2
  for (int k = 0; k < cnt; k++) {
    const size_t o = (size_t)k * np + i;
    const int j = V->bd[o] & RX_JMASK, tj = V->rtype[j];
    const RxTbp *t = &P->tbp[ti * RX_MAXT + tj];
    const double bpi = V->bd_bo[plane + o], bpi2 = V->bd_bo[2 * plane + o];
    RxAtomD J;
    rx_atom_deltas(P, tj, V->total_bo[j], &J);
    V->bd_g[o] += CEover1 * t->p_ovun1 * t->De_s;
    const double w = c4 * (J.Delta - dfvl * J.Delta_lp_temp);
    V->bd_g[plane + o] += w;
    V->bd_g[2 * plane + o] += w;
    RX_ATOMIC_ADD(&V->cd_delta[j], c4 * (1.0 - dfvl * J.dDelta_lp_temp) * (bpi + bpi2));
  }
  RX_ATOMIC_ADD(&V->cd_delta[i], cdd);
}
// NOTE on bd_g: rx_atom_terms, rx_angle_terms and rx_hbond_terms write entries of the atom's own row without atomics;
// rx_torsion_terms reaches other atoms' rows (the far bond k-l), so all its updates are atomic.  Per-atom sums are atomic
// everywhere.  The kernels therefore run 3a, 3b, 3c, 3d as separate launches.

RX_FN double rx_cos_angle(const double *a, double ra, const double *b, double rb) {
  double c = (a[0] * b[0] + a[1] * b[1] + a[2] * b[2]) * rx_rcp(ra * rb);
  if (c > 1.0) c = 1.0;
  if (c < -1.0) c = -1.0;
  return c;
}
RX_FN double rx_angle(const double *a, double ra, const double *b, double rb, double *cosv) {
  const double c = rx_cos_angle(a, ra, b, rb);
  *cosv = c;
  return acos(c);
}
// d(cos theta)/da and /db for cos = a.b/(|a||b|)
RX_FN void rx_dcos(const double *a, double ra, const double *b, double rb, double c, double *da, double *db) {
  const double ia = rx_rcp(ra), ib = rx_rcp(rb), iab = ia * ib, ia2 = ia * ia, ib2 = ib * ib;
  for (int m = 0; m < 3; m++) {
    da[m] = b[m] * iab - c * a[m] * ia2;
    db[m] = a[m] * iab - c * b[m] * ib2;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// pass 3b: valence angle, penalty and three-body conjugation with central atom j (Valence_Angles)
// ------------------------------------------------------------------------------------------------------------------
// The valence-angle pass in three parts, so that the GPU can put a lane on every ANGLE (k_rx_angles lists the (first bond, second bond)
// items of a block of atoms) instead of walking a central atom's angles on one lane: what the angles of an atom share (rx_angle_pre),
// one angle i-j-k (rx_angle_item), and what the sums over an atom's angles feed back (rx_angle_post).  rx_angle_terms is the same
// work atom by atom (host driver, and the reference for the split).
typedef struct { double SBO2, CSBO2, dSBO1, dSBO_dDelta; } RxAnglePre;
typedef struct { double cdd, f[3], dE_dSBO; } RxAngleSum;   // over the angles of one central atom
RX_FN int rx_angle_pre(const RxParams *P, const RxView *V, int j, RxAnglePre *A) {
  const int np = V->npad, tj = V->rtype[j], cnt = V->bd_cnt[j];
  const size_t plane = (size_t)V->maxbd * np;
  if (cnt < 2) return 0;
  const double *gp = P->gp;
  const double p_val8 = gp[33], p_val9 = gp[16];
  RxAtomD J;
  rx_atom_deltas(P, tj, V->total_bo[j], &J);
  double SBOp = 0.0, prod = 1.0;
  for (int a = 0; a < cnt; a++) {
    const size_t o = (size_t)a * np + j;
    const double bo = V->bd_bo[o];
    SBOp += V->bd_bo[plane + o] + V->bd_bo[2 * plane + o];
    double t8 = bo * bo; t8 *= t8; t8 *= t8;
    prod *= exp(-t8);
  }
  double vlpadj;
  if (J.vlpex >= 0.0) {
    vlpadj = 0.0;
    A->dSBO_dDelta = P->lammps_dsbo2 ? 0.0 : (prod - 1.0);
  } else {
    vlpadj = J.nlp;
    A->dSBO_dDelta = (prod - 1.0) * (1.0 - p_val8 * J.dDelta_lp);
  }
  const double SBO = SBOp + (1.0 - prod) * (-J.Delta_boc - p_val8 * vlpadj);
  A->dSBO1 = -8.0 * prod * (J.Delta_boc + p_val8 * vlpadj);   // d(SBO)/d(BO_n) = dSBO1 BO_n^7
  if (SBO <= 0.0) { A->SBO2 = 0.0; A->CSBO2 = 0.0; }
  else if (SBO <= 1.0) { const double w = rx_pow(SBO, p_val9); A->SBO2 = w; A->CSBO2 = p_val9 * w * rx_rcp(SBO); }   // x^(p-1) = x^p / x
  else if (SBO < 2.0) { const double w = rx_pow(2.0 - SBO, p_val9); A->SBO2 = 2.0 - w; A->CSBO2 = p_val9 * w * rx_rcp(2.0 - SBO); }
  else { A->SBO2 = 2.0; A->CSBO2 = 0.0; }
  return 1;
}
// (ai < ak) is an angle of atom j when both bond orders pass the cutoffs of Valence_Angles
RX_FN int rx_angle_item_valid(const RxView *V, int j, int ai, int ak) {
  const int np = V->npad;
  if (!(ai < ak)) return 0;
  const double bo_i = V->bd_bo[(size_t)ai * np + j], bo_k = V->bd_bo[(size_t)ak * np + j];
  return bo_i - RX_THB_CUT > 0.0 && bo_k - RX_THB_CUT > 0.0 && bo_i > RX_THB_CUT && bo_k > RX_THB_CUT && bo_i * bo_k > RX_THB_CUTSQ;
}
// one angle i-j-k: energies, dE/dBO of its two bonds (atomic: the items of an atom run on different lanes), forces on i and k, the
// virial; what it adds to the central atom's sums comes back in S (added to, not set)
template <bool LACC = false>
RX_FN void rx_angle_item(const RxParams *P, const RxView *V, int j, int ai, int ak, double SBO2, double CSBO2, RxAngleSum *S, double *eng, double *vir,
                         double *lf = nullptr, double *lcd = nullptr) {
  const int np = V->npad, tj = V->rtype[j];
  const size_t plane = (size_t)V->maxbd * np;
  const RxSbp *sj = &P->sbp[tj];
  const double *gp = P->gp;
  const double p_val6 = gp[14], p_val10 = gp[17];
  const double p_pen2 = gp[19], p_pen3 = gp[20], p_pen4 = gp[21], p_coa2 = gp[2], p_coa3 = gp[38], p_coa4 = gp[30];
  RxAtomD J;
  rx_atom_deltas(P, tj, V->total_bo[j], &J);
  const double expval6 = exp(p_val6 * J.Delta_boc);
  double cdd_j = 0.0, dE_dSBO = 0.0;
  double *fj = S->f;
  {
    const size_t oi = (size_t)ai * np + j;
    const double bo_i = V->bd_bo[oi], BOA_ij = bo_i - RX_THB_CUT;
    double dji[3];
    const int i = rx_partner(V, j, V->bd[oi], dji);
    const double r_ij = V->bd_bop[3 * plane + oi];
    const int ti = V->rtype[i];
    {
      const size_t ok = (size_t)ak * np + j;
      const double bo_k = V->bd_bo[ok], BOA_jk = bo_k - RX_THB_CUT;
      double djk[3];
      const int k = rx_partner(V, j, V->bd[ok], djk);
      const double r_jk = V->bd_bop[3 * plane + ok];
      const int tk = V->rtype[k];
      const RxThbp *th = &P->thbp[(ti * RX_MAXT + tj) * RX_MAXT + tk];
      if (th->cnt == 0) return;
      double cos_t;
      const double theta = rx_angle(dji, r_ij, djk, r_jk, &cos_t);
      double sin_t = rx_sin_of_cos(cos_t);
      if (sin_t < 1.0e-5) sin_t = 1.0e-5;
      double dE_dtheta = 0.0, g_i = 0.0, g_k = 0.0;   // dE/dtheta, dE/dBO_ij, dE/dBO_jk
      for (int c = 0; c < th->cnt; c++) {
        const RxThbPrm *p = &th->prm[c];
        if (fabs(p->p_val1) <= 0.001) continue;
        // angle energy
        const double pw_i = rx_pow(BOA_ij, p->p_val4), pw_k = rx_pow(BOA_jk, p->p_val4);   // (x^(p-1) = x^p / x below)
        const double exp3ij = exp(-sj->p_val3 * pw_i), f7_ij = 1.0 - exp3ij, Cf7ij = sj->p_val3 * p->p_val4 * (pw_i * rx_rcp(BOA_ij)) * exp3ij;
        const double exp3jk = exp(-sj->p_val3 * pw_k), f7_jk = 1.0 - exp3jk, Cf7jk = sj->p_val3 * p->p_val4 * (pw_k * rx_rcp(BOA_jk)) * exp3jk;
        const double expval7 = exp(-p->p_val7 * J.Delta_boc);
        const double trm8 = 1.0 + expval6 + expval7;
        const double itrm8 = rx_rcp(trm8);
        const double f8_Dj = sj->p_val5 - (sj->p_val5 - 1.0) * (2.0 + expval6) * itrm8;
        const double Cf8j = ((1.0 - sj->p_val5) * (itrm8 * itrm8)) * (p_val6 * expval6 * trm8 - (2.0 + expval6) * (p_val6 * expval6 - p->p_val7 * expval7));
        const double theta_00 = p->theta_00 * RX_PI / 180.0;
        const double ex10 = exp(-p_val10 * (2.0 - SBO2));
        const double theta_0 = RX_PI - theta_00 * (1.0 - ex10);
        const double u = theta_0 - theta;
        const double expval2theta = exp(-p->p_val2 * u * u);
        const double hth = (p->p_val1 >= 0.0) ? p->p_val1 * (1.0 - expval2theta) : p->p_val1 * -expval2theta;
        const double dh_du = 2.0 * p->p_val1 * p->p_val2 * u * expval2theta;
        const double e_ang = f7_ij * f7_jk * f8_Dj * hth;
        eng[RX_E_ANGLE] += e_ang;
        g_i += Cf7ij * f7_jk * f8_Dj * hth;
        g_k += f7_ij * Cf7jk * f8_Dj * hth;
        cdd_j += f7_ij * f7_jk * Cf8j * hth;
        const double c4 = f7_ij * f7_jk * f8_Dj * dh_du;     // dE/du
        dE_dtheta -= c4;
        dE_dSBO += c4 * (theta_00 * p_val10 * ex10) * CSBO2;  // du/dSBO2 = dtheta_0/dSBO2
        // penalty
        const double exp_pen2ij = exp(-p_pen2 * RX_SQR(BOA_ij - 2.0)), exp_pen2jk = exp(-p_pen2 * RX_SQR(BOA_jk - 2.0));
        const double exp_pen3 = exp(-p_pen3 * J.Delta), exp_pen4 = exp(p_pen4 * J.Delta);
        const double trm_pen34 = 1.0 + exp_pen3 + exp_pen4;
        const double itrm_pen34 = rx_rcp(trm_pen34);
        const double f9_Dj = (2.0 + exp_pen3) * itrm_pen34;
        const double Cf9j = (-p_pen3 * exp_pen3 * trm_pen34 - (2.0 + exp_pen3) * (-p_pen3 * exp_pen3 + p_pen4 * exp_pen4)) * (itrm_pen34 * itrm_pen34);
        const double e_pen = p->p_pen1 * f9_Dj * exp_pen2ij * exp_pen2jk;
        eng[RX_E_PEN] += e_pen;
        cdd_j += p->p_pen1 * Cf9j * exp_pen2ij * exp_pen2jk;
        g_i += -2.0 * p_pen2 * (BOA_ij - 2.0) * e_pen;
        g_k += -2.0 * p_pen2 * (BOA_jk - 2.0) * e_pen;
        // three-body conjugation
        const double exp_coa2 = exp(p_coa2 * J.Delta_val);
        const double tbi = V->total_bo[i], tbk = V->total_bo[k];
        const double icoa2 = rx_rcp(1.0 + exp_coa2);
        const double e_coa = p->p_coa1 * icoa2 * exp(-p_coa3 * RX_SQR(tbi - BOA_ij)) * exp(-p_coa3 * RX_SQR(tbk - BOA_jk)) *
                             exp(-p_coa4 * RX_SQR(BOA_ij - 1.5)) * exp(-p_coa4 * RX_SQR(BOA_jk - 1.5));
        eng[RX_E_COA] += e_coa;
        cdd_j += -p_coa2 * exp_coa2 * icoa2 * e_coa;
        g_i += (2.0 * p_coa3 * (tbi - BOA_ij) - 2.0 * p_coa4 * (BOA_ij - 1.5)) * e_coa;
        g_k += (2.0 * p_coa3 * (tbk - BOA_jk) - 2.0 * p_coa4 * (BOA_jk - 1.5)) * e_coa;
        rx_add_cd<LACC>(V, lcd, i, -2.0 * p_coa3 * (tbi - BOA_ij) * e_coa);
        rx_add_cd<LACC>(V, lcd, k, -2.0 * p_coa3 * (tbk - BOA_jk) * e_coa);
      }
      RX_ATOMIC_ADD(&V->bd_g[oi], g_i);
      RX_ATOMIC_ADD(&V->bd_g[ok], g_k);
      // geometry: dE/dtheta -> forces on i, j, k
      const double ce = -dE_dtheta * rx_rcp(sin_t);   // dE/dcos
      double da[3], db[3], fi[3], fk[3];
      rx_dcos(dji, r_ij, djk, r_jk, cos_t, da, db);
      for (int m = 0; m < 3; m++) {
        fi[m] = -ce * da[m];
        fk[m] = -ce * db[m];
        fj[m] -= fi[m] + fk[m];
      }
      for (int m = 0; m < 3; m++) { rx_add_f<LACC>(V, lf, i, m, fi[m]); rx_add_f<LACC>(V, lf, k, m, fk[m]); }
      rx_vt(vir, dji, fi);
      rx_vt(vir, djk, fk);
    }
  }
  S->cdd += cdd_j;
  S->dE_dSBO += dE_dSBO;
}
// SBO feeds every bond of j and Delta_j; the force on j
template <bool LACC = false>
RX_FN void rx_angle_post(const RxView *V, int j, const RxAnglePre *A, const RxAngleSum *S, double *lf = nullptr, double *lcd = nullptr) {
  const int np = V->npad, cnt = V->bd_cnt[j];
  const size_t plane = (size_t)V->maxbd * np;
  double cdd_j = S->cdd;
  if (S->dE_dSBO != 0.0) {
    for (int a = 0; a < cnt; a++) {
      const size_t o = (size_t)a * np + j;
      const double bo = V->bd_bo[o];
      double b7 = bo * bo * bo; b7 = b7 * b7 * bo;
      RX_ATOMIC_ADD(&V->bd_g[o], S->dE_dSBO * A->dSBO1 * b7);
      RX_ATOMIC_ADD(&V->bd_g[plane + o], S->dE_dSBO);
      RX_ATOMIC_ADD(&V->bd_g[2 * plane + o], S->dE_dSBO);
    }
    cdd_j += S->dE_dSBO * A->dSBO_dDelta;
  }
  rx_add_cd<LACC>(V, lcd, j, cdd_j);
  for (int m = 0; m < 3; m++) rx_add_f<LACC>(V, lf, j, m, S->f[m]);
}
RX_FN void rx_angle_terms(const RxParams *P, const RxView *V, int j, double *eng, double *vir) {
  RxAnglePre A;
  if (!rx_angle_pre(P, V, j, &A)) return;
  RxAngleSum S = {0.0, {0.0, 0.0, 0.0}, 0.0};
  const int cnt = V->bd_cnt[j];
  for (int ai = 0; ai < cnt; ai++)
    for (int ak = ai + 1; ak < cnt; ak++)
      if (rx_angle_item_valid(V, j, ai, ak)) rx_angle_item(P, V, j, ai, ak, A.SBO2, A.CSBO2, &S, eng, vir);
  rx_angle_post(V, j, &A, &S);
}

// ------------------------------------------------------------------------------------------------------------------
// pass 3c: torsion and four-body conjugation over the bonds j-k that atom j owns (Torsion_Angles)
// ------------------------------------------------------------------------------------------------------------------
RX_FN void rx_cross(const double *a, const double *b, double *c) {
  c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
// Whether (j, ak, ai) is a work item of the torsion pass: ak a bond j-k that j owns with a partner row to come back through, ai another
// bond j-i of j, both above the bond-order cutoff
RX_FN int rx_torsion_item_valid(const RxView *V, int j, int ak, int ai) {
  const int np = V->npad;
  if (ai == ak) return 0;
  const size_t ojk = (size_t)ak * np + j, oij = (size_t)ai * np + j;
  if (!rx_owns(j, V->bd[ojk]) || !(V->bd_bo[ojk] > RX_THB_CUT) || V->bd_rev[ojk] < 0) return 0;
  return V->bd_bo[oij] > RX_THB_CUT;
}
// One work item: the torsions i-j-k-l over all bonds k-l of k, for the first leg j-i (ai) and the central bond j-k (ak) of atom j.
// Everything it adds to shared places is atomic (other items reach the same bonds and atoms); the sums over a central bond's items
// (dE/dBO_jk, dE/dBO_pi_jk, dE/dDelta, forces on j and k) therefore leave per item.  The GPU puts a lane on an item
// (k_rx_torsions compacts the items of a block of atoms first); rx_torsion_terms below is the same work atom by atom.
template <bool LACC = false>
RX_FN void rx_torsion_item(const RxParams *P, const RxView *V, int j, int ak, int ai, double *eng, double *vir, double *lf = nullptr, double *lcd = nullptr) {
  const int np = V->npad, tj = V->rtype[j];
  const size_t plane = (size_t)V->maxbd * np;
  const double p_tor2 = P->gp[23], p_tor3 = P->gp[24], p_tor4 = P->gp[25], p_cot2 = P->gp[27];
  RxAtomD J;
  rx_atom_deltas(P, tj, V->total_bo[j], &J);
  const size_t ojk = (size_t)ak * np + j;
  const int ejk = V->bd[ojk];
  const double bo_jk = V->bd_bo[ojk];
  double q[3];   // j -> k
  const int k = rx_partner(V, j, ejk, q);
  const int tk = V->rtype[k], cntk = V->bd_cnt[k], rev = V->bd_rev[ojk];
  const double r_jk = V->bd_bop[3 * plane + ojk], bpi_jk = V->bd_bo[plane + ojk];
  RxAtomD K;
  rx_atom_deltas(P, tk, V->total_bo[k], &K);
  const double BOA_jk = bo_jk - RX_THB_CUT;
  const double exp_tor2_jk = exp(-p_tor2 * BOA_jk), exp_cot2_jk = exp(-p_cot2 * RX_SQR(BOA_jk - 1.5));
  const double DjDk = J.Delta_boc + K.Delta_boc;
  const double exp_tor3 = exp(-p_tor3 * DjDk), exp_tor4 = exp(p_tor4 * DjDk), trm34 = 1.0 + exp_tor3 + exp_tor4;
  const double itrm34 = rx_rcp(trm34);
  const double f11 = (2.0 + exp_tor3) * itrm34;
  const double Cf11 = (-p_tor3 * exp_tor3 * trm34 - (2.0 + exp_tor3) * (-p_tor3 * exp_tor3 + p_tor4 * exp_tor4)) * (itrm34 * itrm34);
  const double mq[3] = {-q[0], -q[1], -q[2]};
  double g_jk = 0.0, gpi_jk = 0.0, cdd = 0.0, fj[3] = {0, 0, 0}, fk[3] = {0, 0, 0};
  {
      const size_t oij = (size_t)ai * np + j;
      const double bo_ij = V->bd_bo[oij];
      double p[3];   // j -> i
      const int eij = V->bd[oij];
      const int i = rx_partner(V, j, eij, p);
      const int ti = V->rtype[i];
      const double r_ij = V->bd_bop[3 * plane + oij], BOA_ij = bo_ij - RX_THB_CUT;
      const double cos_ijk = rx_cos_angle(p, r_ij, q, r_jk);
      double sin_ijk = rx_sin_of_cos(cos_ijk);
      if (sin_ijk >= 0 && sin_ijk <= RX_MIN_SINE) sin_ijk = RX_MIN_SINE;
      const double exp_tor2_ij = exp(-p_tor2 * BOA_ij), exp_cot2_ij = exp(-p_cot2 * RX_SQR(BOA_ij - 1.5));
      double g_ij = 0.0, fi[3] = {0, 0, 0};
      for (int al = 0; al < cntk; al++) {
        if (al == rev) continue;
        const size_t okl = (size_t)al * np + k;
        const int ekl = V->bd[okl];
        double s[3];   // k -> l
        const int l = rx_partner(V, k, ekl, s);
        // the same atom (same image) at both ends is no torsion: i seen from j equals l seen from k when p = q + s
        if (l == i && fabs(p[0] - q[0] - s[0]) + fabs(p[1] - q[1] - s[1]) + fabs(p[2] - q[2] - s[2]) < 1e-8) continue;
        const int tl = V->rtype[l];
        const RxFbp *fb = &P->fbp[((ti * RX_MAXT + tj) * RX_MAXT + tk) * RX_MAXT + tl];
        const double bo_kl = V->bd_bo[okl];
        if (!(fb->cnt && bo_kl > RX_THB_CUT && bo_ij * bo_jk * bo_kl > RX_THB_CUT)) continue;
        const double r_kl = V->bd_bop[3 * plane + okl], BOA_kl = bo_kl - RX_THB_CUT;
        const double cos_jkl = rx_cos_angle(mq, r_jk, s, r_kl);
        double sin_jkl = rx_sin_of_cos(cos_jkl);
        if (sin_jkl >= 0 && sin_jkl <= RX_MIN_SINE) sin_jkl = RX_MIN_SINE;
        // dihedral from the plane normals n1 = p x q, n2 = s x q
        double n1[3], n2[3];
        rx_cross(p, q, n1);
        rx_cross(s, q, n2);
        const double l1 = rx_sqrt(n1[0] * n1[0] + n1[1] * n1[1] + n1[2] * n1[2]), l2 = rx_sqrt(n2[0] * n2[0] + n2[1] * n2[1] + n2[2] * n2[2]);
        // (n2 = (k->l) x (j->k) = (k->j) x (k->l): parallel normals for the cis arrangement, cos(omega) = +1 there)
        double co = 1.0;
        const int ok_n = l1 > 0.0 && l2 > 0.0;
        const double i12 = ok_n ? rx_rcp(l1 * l2) : 0.0;
        if (ok_n) co = (n1[0] * n2[0] + n1[1] * n2[1] + n1[2] * n2[2]) * i12;
        if (co > 1.0) co = 1.0;
        if (co < -1.0) co = -1.0;
        const double cos2 = 2.0 * co * co - 1.0, cos3 = co * (4.0 * co * co - 3.0);
        const double exp_tor2_kl = exp(-p_tor2 * BOA_kl), exp_cot2_kl = exp(-p_cot2 * RX_SQR(BOA_kl - 1.5));
        const double fn10 = (1.0 - exp_tor2_ij) * (1.0 - exp_tor2_jk) * (1.0 - exp_tor2_kl);
        const double w = 2.0 - bpi_jk - f11;
        const double exp_tor1 = exp(fb->p_tor1 * w * w);
        const double CV = 0.5 * (fb->V1 * (1.0 + co) + fb->V2 * exp_tor1 * (1.0 - cos2) + fb->V3 * (1.0 + cos3));
        const double ss = sin_ijk * sin_jkl;
        const double e_tor = fn10 * ss * CV;
        eng[RX_E_TORS] += e_tor;
        const double fn12 = exp_cot2_ij * exp_cot2_jk * exp_cot2_kl;
        const double e_con = fb->p_cot1 * fn12 * (1.0 + (co * co - 1.0) * ss);
        eng[RX_E_CONJ] += e_con;
        // derivatives with respect to the bond orders and Deltas
        g_ij += p_tor2 * exp_tor2_ij * (1.0 - exp_tor2_jk) * (1.0 - exp_tor2_kl) * ss * CV - 2.0 * p_cot2 * (BOA_ij - 1.5) * e_con;
        g_jk += (1.0 - exp_tor2_ij) * p_tor2 * exp_tor2_jk * (1.0 - exp_tor2_kl) * ss * CV - 2.0 * p_cot2 * (BOA_jk - 1.5) * e_con;
        const double g_kl = (1.0 - exp_tor2_ij) * (1.0 - exp_tor2_jk) * p_tor2 * exp_tor2_kl * ss * CV - 2.0 * p_cot2 * (BOA_kl - 1.5) * e_con;
        RX_ATOMIC_ADD(&V->bd_g[okl], g_kl);
        const double dCV_dw = 0.5 * fb->V2 * (1.0 - cos2) * exp_tor1 * 2.0 * fb->p_tor1 * w;   // w = 2 - BO_pi(jk) - f11
        gpi_jk += -fn10 * ss * dCV_dw;
        cdd += -fn10 * ss * dCV_dw * Cf11;   // on Delta_j and on Delta_k alike
        // geometry
        const double dE_dsin_ijk = fn10 * sin_jkl * CV + fb->p_cot1 * fn12 * (co * co - 1.0) * sin_jkl;
        const double dE_dsin_jkl = fn10 * sin_ijk * CV + fb->p_cot1 * fn12 * (co * co - 1.0) * sin_ijk;
        const double dE_dco = fn10 * ss * 0.5 * (fb->V1 - 4.0 * fb->V2 * exp_tor1 * co + fb->V3 * (12.0 * co * co - 3.0)) + fb->p_cot1 * fn12 * 2.0 * co * ss;
        // sin(theta) = sqrt(1 - cos^2): dsin/dcos = -cos/sin
        const double ce_ijk = dE_dsin_ijk * (-cos_ijk * rx_rcp(sin_ijk)), ce_jkl = dE_dsin_jkl * (-cos_jkl * rx_rcp(sin_jkl));
        double dp[3] = {0, 0, 0}, dq[3] = {0, 0, 0}, ds[3] = {0, 0, 0};   // dE/dp, dE/dq, dE/ds
        double da[3], db[3];
        rx_dcos(p, r_ij, q, r_jk, cos_ijk, da, db);
        for (int m = 0; m < 3; m++) { dp[m] += ce_ijk * da[m]; dq[m] += ce_ijk * db[m]; }
        rx_dcos(mq, r_jk, s, r_kl, cos_jkl, da, db);
        for (int m = 0; m < 3; m++) { dq[m] -= ce_jkl * da[m]; ds[m] += ce_jkl * db[m]; }
        if (ok_n && co > -1.0 && co < 1.0) {
          // c = n1.n2/(|n1||n2|): g1 = dc/dn1, g2 = dc/dn2; n1 = p x q, n2 = s x q
          double g1[3], g2[3], t1[3], t2[3], t3[3], t4[3];
          const double c11 = co * (i12 * i12) * (l2 * l2), c22 = co * (i12 * i12) * (l1 * l1);   // co / l1^2, co / l2^2
          for (int m = 0; m < 3; m++) {
            g1[m] = n2[m] * i12 - c11 * n1[m];
            g2[m] = n1[m] * i12 - c22 * n2[m];
          }
          rx_cross(q, g1, t1);   // dc/dp
          rx_cross(g1, p, t2);   // dc/dq (through n1)
          rx_cross(q, g2, t3);   // dc/ds
          rx_cross(g2, s, t4);   // dc/dq (through n2)
          for (int m = 0; m < 3; m++) { dp[m] += dE_dco * t1[m]; dq[m] += dE_dco * (t2[m] + t4[m]); ds[m] += dE_dco * t3[m]; }
        }
        // p = x_i - x_j, q = x_k - x_j, s = x_l - x_k
        double fl[3];
        for (int m = 0; m < 3; m++) {
          fi[m] -= dp[m];
          fj[m] += dp[m] + dq[m];
          fk[m] += -dq[m] + ds[m];
          fl[m] = -ds[m];
        }
        for (int m = 0; m < 3; m++) rx_add_f<LACC>(V, lf, l, m, fl[m]);
        // virial about j: positions p (i), 0 (j), q (k), q + s (l)
        const double mdp[3] = {-dp[0], -dp[1], -dp[2]}, fkk[3] = {-dq[0] + ds[0], -dq[1] + ds[1], -dq[2] + ds[2]}, qs[3] = {q[0] + s[0], q[1] + s[1], q[2] + s[2]};
        rx_vt(vir, p, mdp);
        rx_vt(vir, q, fkk);
        rx_vt(vir, qs, fl);
      }
      RX_ATOMIC_ADD(&V->bd_g[oij], g_ij);   // other items reach this row as their far bond in the same pass
      for (int m = 0; m < 3; m++) rx_add_f<LACC>(V, lf, i, m, fi[m]);
  }
  if (g_jk != 0.0) RX_ATOMIC_ADD(&V->bd_g[ojk], g_jk);
  if (gpi_jk != 0.0) RX_ATOMIC_ADD(&V->bd_g[plane + ojk], gpi_jk);
  if (cdd != 0.0) { rx_add_cd<LACC>(V, lcd, j, cdd); rx_add_cd<LACC>(V, lcd, k, cdd); }
  for (int m = 0; m < 3; m++) { rx_add_f<LACC>(V, lf, j, m, fj[m]); rx_add_f<LACC>(V, lf, k, m, fk[m]); }
}
RX_FN void rx_torsion_terms(const RxParams *P, const RxView *V, int j, double *eng, double *vir) {
  const int cnt = V->bd_cnt[j];
  for (int ak = 0; ak < cnt; ak++)
    for (int ai = 0; ai < cnt; ai++)
      if (rx_torsion_item_valid(V, j, ak, ai)) rx_torsion_item(P, V, j, ak, ai, eng, vir);
}

// ------------------------------------------------------------------------------------------------------------------
// pass 3d: hydrogen bonds of hydrogen atom j: donors i among its bonds, acceptors k in its neighbour row (Hydrogen_Bonds)
// ------------------------------------------------------------------------------------------------------------------
RX_FN void rx_hbond_terms(const RxParams *P, const RxView *V, int j, double *eng, double *vir) {
  const int np = V->npad, tj = V->rtype[j];
  if (P->sbp[tj].p_hbond != 1) return;
  const size_t plane = (size_t)V->maxbd * np;
  const int cnt = V->bd_cnt[j], nn = V->nb_cnt[j];
  int ndon = 0;
  for (int a = 0; a < cnt; a++) {
    const size_t o = (size_t)a * np + j;
    if (P->sbp[V->rtype[V->bd[o] & RX_JMASK]].p_hbond == 2 && V->bd_bo[o] >= RX_HB_THRESHOLD) ndon++;
  }
  if (ndon == 0) return;
  double fj[3] = {0, 0, 0};
  for (int n = 0; n < nn; n++) {
    const int ek = rx_nb_entry(V, j, n);
    const int k = ek & RX_JMASK, tk = V->rtype[k];
    if (P->sbp[tk].p_hbond != 2) continue;
    double djk[3];
    rx_partner(V, j, ek, djk);
    const double r2 = djk[0] * djk[0] + djk[1] * djk[1] + djk[2] * djk[2];
    if (r2 > RX_HBOND_CUT * RX_HBOND_CUT) continue;
    const double r_jk = sqrt(r2);
    double fk[3] = {0, 0, 0};
    for (int a = 0; a < cnt; a++) {
      const size_t o = (size_t)a * np + j;
      const int ei = V->bd[o], i = ei & RX_JMASK, ti = V->rtype[i];
      const double bo_ij = V->bd_bo[o];
      if (P->sbp[ti].p_hbond != 2 || bo_ij < RX_HB_THRESHOLD) continue;
      if (ei == ek) continue;   // donor and acceptor are the same atom (same image)
      const RxHbp *h = &P->hbp[(ti * RX_MAXT + tj) * RX_MAXT + tk];
      if (h->r0_hb <= 0.0) continue;
      double dji[3];
      rx_partner(V, j, ei, dji);
      const double r_ij = V->bd_bop[3 * plane + o];
      const double cos_t = rx_cos_angle(dji, r_ij, djk, r_jk);
      const double s4 = 0.25 * RX_SQR(1.0 - cos_t);   // sin^4(theta/2)
      const double ex2 = exp(-h->p_hb2 * bo_ij), ex3 = exp(-h->p_hb3 * (h->r0_hb / r_jk + r_jk / h->r0_hb - 2.0));
      const double e_hb = h->p_hb1 * (1.0 - ex2) * ex3 * s4;
      eng[RX_E_HB] += e_hb;
      V->bd_g[o] += h->p_hb1 * h->p_hb2 * ex2 * ex3 * s4;
      const double dE_dcos = h->p_hb1 * (1.0 - ex2) * ex3 * (-0.5 * (1.0 - cos_t));
      const double dE_dr = e_hb * (-h->p_hb3) * (-h->r0_hb / (r_jk * r_jk) + 1.0 / h->r0_hb);
      double da[3], db[3], fi[3], fkk[3];
      rx_dcos(dji, r_ij, djk, r_jk, cos_t, da, db);
      for (int m = 0; m < 3; m++) {
        fi[m] = -dE_dcos * da[m];
        fkk[m] = -dE_dcos * db[m] - dE_dr * djk[m] / r_jk;
        fj[m] -= fi[m] + fkk[m];
        fk[m] += fkk[m];
      }
      for (int m = 0; m < 3; m++) RX_ATOMIC_ADD(&V->f[3 * i + m], fi[m]);
      rx_vt(vir, dji, fi);
      rx_vt(vir, djk, fkk);
    }
    if (fk[0] != 0.0 || fk[1] != 0.0 || fk[2] != 0.0)
      for (int m = 0; m < 3; m++) RX_ATOMIC_ADD(&V->f[3 * k + m], fk[m]);
  }
  for (int m = 0; m < 3; m++) RX_ATOMIC_ADD(&V->f[3 * j + m], fj[m]);
}

// ------------------------------------------------------------------------------------------------------------------
// pass 3e: tapered shielded van der Waals and Coulomb of atom i over its full neighbour row (vdW_Coulomb_Energy): the force on
// i only, half of each pair's energy and virial (the partner's lane does the same pair from its side) + polarisation energy
// ------------------------------------------------------------------------------------------------------------------
// entries k0, k0 + kstep, ... of the row: several waves may share a row; fi = force on i from these entries
// one non-bonded pair (tapered van der Waals with the shielded distance + shielded Coulomb) at distance r = sqrt(r2) <= swb:
// energies and s = (dE/dr) / r, so that dE/dd = s d
RX_FN void rx_nonbonded_pair(const RxParams *P, const RxTbp *t, double qq, double r2, double *evdw, double *ecoul, double *s_out) {
  const double p_vdW1 = P->gp[28], p_vdW1i = P->inv_pvdw1;
  const double rinv = rx_rsqrt(r2), r = r2 * rinv;
  double dTap;
  const double Tap = rx_taper(P, r, &dTap);
  // shielded distance fn13 = (r^p + gamma_w^-p)^(1/p); d(fn13)/dr = fn13 / (r^p + gamma_w^-p) * r^(p-1)
  const double powr = exp(0.5 * p_vdW1 * rx_log(r2));
  const double sum = powr + t->powgw;
  const double fn13 = exp(p_vdW1i * rx_log(sum));
  const double dfn13 = fn13 * rx_rcp(sum) * powr * rinv;
  const double ex2 = exp(0.5 * t->alpha * (1.0 - fn13 * t->inv_rvdw)), ex1 = ex2 * ex2;
  const double e_v = t->D * (ex1 - 2.0 * ex2);
  double dE = dTap * e_v - Tap * t->D * (t->alpha * t->inv_rvdw) * (ex1 - ex2) * dfn13;
  *evdw = Tap * e_v;
  const double r3g = r2 * r + t->gamma, c13i = rx_icbrt(r3g);
  *ecoul = Tap * qq * c13i;
  dE += qq * c13i * (dTap - Tap * r2 * (c13i * c13i * c13i));   // 1 / (r^3 + gamma) = c13i^3
  *s_out = dE * rinv;
}
// atom i's end of its pairs: the entries k0, k0 + kstep, ... of its list row (every pair is seen from both ends, half the energy each)
RX_FN void rx_nonbonded_part(const RxParams *P, const RxView *V, int i, int k0, int kstep, double *fi, double *eng, double *vir) {
  const int ti = V->rtype[i], cnt = V->nb_cnt[i];
  const double qi = RX_C_ELE * V->q[i];
  const double swb2 = P->swb * P->swb;
  const double xi0 = V->x[3 * i], xi1 = V->x[3 * i + 1], xi2 = V->x[3 * i + 2];
  double evdw = 0.0, ecoul = 0.0, w[6] = {0, 0, 0, 0, 0, 0};
  for (int k = k0; k < cnt; k += kstep) {
    const int e = rx_nb_entry(V, i, k);
    const int j = e & RX_JMASK;
    double sh[3];
    rx_shift(V, e, sh);
    const double d0 = V->x[3 * j] - xi0 + sh[0], d1 = V->x[3 * j + 1] - xi1 + sh[1], d2 = V->x[3 * j + 2] - xi2 + sh[2];
    const double r2 = d0 * d0 + d1 * d1 + d2 * d2;
    if (r2 > swb2) continue;
    double ev, ec, s;
    rx_nonbonded_pair(P, &P->tbp[ti * RX_MAXT + V->rtype[j]], qi * V->q[j], r2, &ev, &ec, &s);
    evdw += ev;
    ecoul += ec;
    fi[0] += s * d0; fi[1] += s * d1; fi[2] += s * d2;
    // pair virial d (x) f_j = -s d (x) d, half per end
    w[0] += s * d0 * d0; w[1] += s * d1 * d1; w[2] += s * d2 * d2;
    w[3] += s * d0 * d1; w[4] += s * d0 * d2; w[5] += s * d1 * d2;
  }
  eng[RX_E_VDW] += 0.5 * evdw;
  eng[RX_E_COUL] += 0.5 * ecoul;
  for (int m = 0; m < 6; m++) vir[m] -= 0.5 * w[m];
}
RX_FN void rx_nonbonded(const RxParams *P, const RxView *V, int i, double *eng, double *vir) {
  double fi[3] = {0, 0, 0};
  rx_nonbonded_part(P, V, i, 0, 1, fi, eng, vir);
  const int ti = V->rtype[i];
  const double qi = V->q[i];
  eng[RX_E_POL] += RX_KCALPMOL_TO_EV * (P->sbp[ti].chi * qi + 0.5 * P->sbp[ti].eta * qi * qi);
  for (int m = 0; m < 3; m++) RX_ATOMIC_ADD(&V->f[3 * i + m], fi[m]);
}

// ------------------------------------------------------------------------------------------------------------------
// pass 4a: dE/d(BO, BO_pi, BO_pi2) of every bond of atom i (both ends' gathers + the Delta sums) through the corrections:
// coefficient of d in the bond force (Delta' part aside) and this atom's share of dE/dDelta'_i
// ------------------------------------------------------------------------------------------------------------------
RX_FN void rx_back_corr(const RxParams *P, const RxView *V, int i) {
  const int np = V->npad, ti = V->rtype[i], cnt = V->bd_cnt[i];
  const size_t plane = (size_t)V->maxbd * np;
  const double Di = V->deltap[i], cdi = V->cd_delta[i];
  double hd = 0.0;
  for (int k = 0; k < cnt; k++) {
    const size_t o = (size_t)k * np + i;
    const int j = V->bd[o] & RX_JMASK, rev = V->bd_rev[o];
    if (rev < 0) { V->bd_cb[o] = 0.0; continue; }
    const size_t oj = (size_t)rev * np + j;
    double g = V->bd_g[o] + V->bd_g[oj] + cdi + V->cd_delta[j];
    double gp = V->bd_g[plane + o] + V->bd_g[plane + oj], gpp = V->bd_g[2 * plane + o] + V->bd_g[2 * plane + oj];
    const double B = V->bd_bop[o], Bp = V->bd_bop[plane + o], Bpp = V->bd_bop[2 * plane + o];
    RxCorr c;
    rx_corr(P, ti, V->rtype[j], Di, V->deltap[j], B, &c);
    // thresholded components carry no derivative
    if (B * c.Y * c.X < 1e-10) g = 0.0;
    if (Bp * c.Y * c.Y * c.X < 1e-10) gp = 0.0;
    if (Bpp * c.Y * c.Y * c.X < 1e-10) gpp = 0.0;
    const double Y2X = c.Y * c.Y * c.X;
    const double aB = g * (c.Y * c.X + B * c.Y * c.Xb) + (gp * Bp + gpp * Bpp) * c.Y * c.Y * c.Xb;
    const double aP = gp * Y2X, aPP = gpp * Y2X;
    const double cs = V->bd_c[o], cp = V->bd_c[plane + o], cpp = V->bd_c[2 * plane + o];
    V->bd_cb[o] = aB * (cs + cp + cpp) + aP * cp + aPP * cpp;
    hd += g * B * (c.Yi * c.X + c.Y * c.Xi) + (gp * Bp + gpp * Bpp) * (2.0 * c.Y * c.Yi * c.X + c.Y * c.Y * c.Xi);
  }
  V->hd[i] = hd;
}

// pass 4b: forces of atom i from its bonds' dependence on distance
RX_FN void rx_back_force(const RxParams *P, const RxView *V, int i, double *vir) {
  (void)P;
  const int np = V->npad, cnt = V->bd_cnt[i];
  const size_t plane = (size_t)V->maxbd * np;
  const double hi = V->hd[i];
  double fi[3] = {0, 0, 0};
  for (int k = 0; k < cnt; k++) {
    const size_t o = (size_t)k * np + i;
    double d[3];
    const int j = rx_partner(V, i, V->bd[o], d);
    const double ct = V->bd_c[o] + V->bd_c[plane + o] + V->bd_c[2 * plane + o];
    const double s = V->bd_cb[o] + (hi + V->hd[j]) * ct;   // dE/dd = s d
    fi[0] += s * d[0]; fi[1] += s * d[1]; fi[2] += s * d[2];
    vir[0] -= 0.5 * s * d[0] * d[0]; vir[1] -= 0.5 * s * d[1] * d[1]; vir[2] -= 0.5 * s * d[2] * d[2];
    vir[3] -= 0.5 * s * d[0] * d[1]; vir[4] -= 0.5 * s * d[0] * d[2]; vir[5] -= 0.5 * s * d[1] * d[2];
  }
  for (int m = 0; m < 3; m++) RX_ATOMIC_ADD(&V->f[3 * i + m], fi[m]);
}

// ------------------------------------------------------------------------------------------------------------------
// charge equilibration (fix qeq/reax): matrix entries of atom i's row, H_ij = Tap(r) 14.4 / (r^3 + gamma_ij)^(1/3)
// ------------------------------------------------------------------------------------------------------------------
// H_ij of list entry e of row i, or a negative number when the pair is outside the taper radius; *col = j
// (gamma_row: the gamma_ij of atom i's type against every type, e.g. staged in LDS; NULL: read from the parameter tables)
RX_FN double rx_qeq_entry(const RxParams *P, const RxView *V, int i, int e, int *col, const double *gamma_row RX_DEFAULT_NULL) {
  double d[3];
  const int j = rx_partner(V, i, e, d);
  *col = j;
  const double r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  if (r2 > P->swb * P->swb) return -1.0;
  const double r = r2 * rx_rsqrt(r2);
  double dTap;
  const double Tap = rx_taper(P, r, &dTap);
  const double gamma = gamma_row ? gamma_row[V->rtype[j]] : P->tbp[V->rtype[i] * RX_MAXT + V->rtype[j]].gamma;
  return Tap * RX_EV_TO_KCALPMOL * rx_icbrt(r2 * r + gamma);
}
// row i of the matrix, serially (host checks: tests/reax_host_driver.cpp; the kernels put a wave on the row): the entries inside the
// taper radius, in list order, at i * maxnb
RX_FN void rx_qeq_row(const RxParams *P, const RxView *V, int i) {
  const size_t base = (size_t)i * V->maxnb;
  int len = 0;
  for (int k = 0; k < V->nb_cnt[i]; k++) {
    int col;
    const double h = rx_qeq_entry(P, V, i, rx_nb_entry(V, i, k), &col);
    if (h < 0.0) continue;
    V->hval[base + len] = h;
    if (V->hcol16) V->hcol16[base + len] = (unsigned short)col;
    else V->hcol32[base + len] = col;
    len++;
  }
  V->hlen[i] = len;
}
RX_FN double rx_qeq_matvec_row(const RxParams *P, const RxView *V, int i, const double *x) {
  const size_t base = (size_t)i * V->maxnb;
  double y = P->sbp[V->rtype[i]].eta * x[i];
  for (int c = 0; c < V->hlen[i]; c++) y += V->hval[base + c] * x[V->hcol16 ? (int)V->hcol16[base + c] : V->hcol32[base + c]];
  return y;
}
