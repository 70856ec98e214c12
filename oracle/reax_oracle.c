/*
 * reax_oracle.c -- see reax_oracle.h.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (LAMMPS USER-REAXC is absent).
 *
 * Each block names the USER-REAXC routine whose arithmetic it restates [LAMMPS-ext: the package is not in the reference
 * tree; function names are given so that a later comparison with LAMMPS knows where to look]:
 *   read_ffield ............ reaxc_ffield.cpp  Read_Force_Field
 *   bond orders ............ reaxc_bond_orders.cpp  BOp, BO
 *   bonds .................. reaxc_bonds.cpp  Bonds
 *   lone pair / over / under reaxc_multi_body.cpp  Atom_Energy
 *   valence angles ......... reaxc_valence_angles.cpp  Valence_Angles
 *   torsions ............... reaxc_torsion_angles.cpp  Torsion_Angles
 *   hydrogen bonds ......... reaxc_hydrogen_bonds.cpp  Hydrogen_Bonds
 *   van der Waals / Coulomb  reaxc_nonbonded.cpp  vdW_Coulomb_Energy, Compute_Polarization_Energy
 *   taper .................. reaxc_init_md.cpp  Init_Taper
 *   charge equilibration ... fix_qeq_reax.cpp  init_matvec, compute_H, CG, calculate_Q
 */
#include "reax_oracle.h"

#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define RX_MAXT 8          /* atom types of a force-field file */
#define RX_MAXANG 4        /* parameter sets per angle triple */
#define C_ELE 332.06371
#define KCALPMOL_TO_EV 23.02
#define EV_TO_KCALPMOL 14.4
#define THB_CUT 0.001
#define THB_CUTSQ 0.00001
#define HB_THRESHOLD 1e-2
#define BOND_CUT 5.0
#define HBOND_CUT 7.5
#define MIN_SINE 1e-10
#define SQR(x) ((x) * (x))

typedef struct {
  char name[8];
  double r_s, valency, mass, r_vdw, epsilon, gamma, r_pi, valency_e, nlp_opt;
  double alpha, gamma_w, valency_boc, p_ovun5, chi, eta;
  int p_hbond;
  double r_pi_pi, p_lp2, b_o_131, b_o_132, b_o_133;
  double p_ovun2, p_val3, valency_val, p_val5, rcore2, ecore2, acore2;
} sbp_t;
typedef struct {
  double De_s, De_p, De_pp, p_be1, p_bo5, v13cor, p_bo6, p_ovun1, p_be2, p_bo3, p_bo4, p_bo1, p_bo2, ovc;
  double r_s, r_p, r_pp, p_boc3, p_boc4, p_boc5, D, alpha, r_vdW, gamma_w, gamma;
} tbp_t;
typedef struct { double theta_00, p_val1, p_val2, p_coa1, p_val7, p_pen1, p_val4; } thb_prm;
typedef struct { int cnt; thb_prm prm[RX_MAXANG]; } thbp_t;
typedef struct { int cnt, specific; double V1, V2, V3, p_tor1, p_cot1; } fbp_t;
typedef struct { double r0_hb, p_hb1, p_hb2, p_hb3; } hbp_t;

struct rxo_ff {
  int ngp, nt;
  double gp[64];
  sbp_t sbp[RX_MAXT];
  tbp_t tbp[RX_MAXT][RX_MAXT];
  thbp_t thbp[RX_MAXT][RX_MAXT][RX_MAXT];
  fbp_t fbp[RX_MAXT][RX_MAXT][RX_MAXT][RX_MAXT];
  hbp_t hbp[RX_MAXT][RX_MAXT][RX_MAXT];
  double bo_cut, swa, swb, tap[8];
};

/* ------------------------------------------------------------------ force-field file */
/* a count or a 1-based index read as a double: a damaged file may hold NaN, 1e300 or -7 there, and (int) of those is undefined */
static int as_int(double d) { return (d == d && d > -1.0e9 && d < 1.0e9) ? (int)d : -1000000; }
static int next_vals(FILE *fp, double *v, int maxv, char *first_word) {
  char line[1024];
  if (!fgets(line, sizeof line, fp)) return -1;
  int n = 0;
  char *tok = strtok(line, " \t\r\n");
  int first = 1;
  while (tok && n < maxv) {
    char *end;
    double d = strtod(tok, &end);
    if (end != tok && *end == 0) v[n++] = d;
    else if (first && first_word) { strncpy(first_word, tok, 7); first_word[7] = 0; }
    else if (tok[0] == '!' ) break;
    else if (!first) break;   /* trailing comment */
    first = 0;
    tok = strtok(NULL, " \t\r\n");
  }
  return n;
}

rxo_ff *rxo_read_ffield(const char *path) {
  FILE *fp = fopen(path, "r");
  if (!fp) { fprintf(stderr, "reax_oracle: cannot open %s\n", path); return NULL; }
  rxo_ff *ff = (rxo_ff *)calloc(1, sizeof(rxo_ff));
  char line[1024], w[8];
  double v[16];
  int ok = fgets(line, sizeof line, fp) != NULL;             /* header comment */
  ok = ok && next_vals(fp, v, 16, NULL) >= 1;
  ff->ngp = ok ? as_int(v[0]) : 0;
  for (int k = 0; ok && k < ff->ngp; k++) { ok = next_vals(fp, v, 16, NULL) >= 1; if (k < 64) ff->gp[k] = v[0]; }
  /* atoms: count line + three header lines, four lines per atom */
  ok = ok && next_vals(fp, v, 16, NULL) >= 1;
  ff->nt = ok ? as_int(v[0]) : 0;
  if (ff->nt > RX_MAXT || ff->nt < 0) { ok = 0; ff->nt = 0; }   /* (nt sizes the loops below: found by tests/test_corrupt_files.py under UBSan) */
  for (int k = 0; ok && k < 3; k++) ok = fgets(line, sizeof line, fp) != NULL;
  for (int i = 0; ok && i < ff->nt; i++) {
    sbp_t *s = &ff->sbp[i];
    w[0] = 0;
    ok = next_vals(fp, v, 16, w) >= 8;
    if (!ok) break;
    memcpy(s->name, w, 8);
    s->r_s = v[0]; s->valency = v[1]; s->mass = v[2]; s->r_vdw = v[3]; s->epsilon = v[4]; s->gamma = v[5]; s->r_pi = v[6]; s->valency_e = v[7];
    s->nlp_opt = 0.5 * (s->valency_e - s->valency);
    ok = next_vals(fp, v, 16, NULL) >= 8;
    if (!ok) break;
    s->alpha = v[0]; s->gamma_w = v[1]; s->valency_boc = v[2]; s->p_ovun5 = v[3]; s->chi = v[5]; s->eta = 2.0 * v[6]; s->p_hbond = as_int(v[7]);
    ok = next_vals(fp, v, 16, NULL) >= 8;
    if (!ok) break;
    s->r_pi_pi = v[0]; s->p_lp2 = v[1]; s->b_o_131 = v[3]; s->b_o_132 = v[4]; s->b_o_133 = v[5];
    ok = next_vals(fp, v, 16, NULL) >= 8;
    if (!ok) break;
    s->p_ovun2 = v[0]; s->p_val3 = v[1]; s->valency_val = v[3]; s->p_val5 = v[4]; s->rcore2 = v[5]; s->ecore2 = v[6]; s->acore2 = v[7];
    if (s->mass < 21.0 && s->valency_val != s->valency_boc) s->valency_val = s->valency_boc;   /* first-row fix-up of the reader */
  }
  /* combination rules (overridden below by the off-diagonal block) */
  for (int i = 0; i < ff->nt; i++)
    for (int j = 0; j < ff->nt; j++) {
      tbp_t *t = &ff->tbp[i][j];
      const sbp_t *a = &ff->sbp[i], *b = &ff->sbp[j];
      t->r_s = 0.5 * (a->r_s + b->r_s); t->r_p = 0.5 * (a->r_pi + b->r_pi); t->r_pp = 0.5 * (a->r_pi_pi + b->r_pi_pi);
      t->p_boc3 = sqrt(a->b_o_132 * b->b_o_132); t->p_boc4 = sqrt(a->b_o_131 * b->b_o_131); t->p_boc5 = sqrt(a->b_o_133 * b->b_o_133);
      t->D = sqrt(a->epsilon * b->epsilon); t->alpha = sqrt(a->alpha * b->alpha); t->r_vdW = 2.0 * sqrt(a->r_vdw * b->r_vdw);
      t->gamma_w = sqrt(a->gamma_w * b->gamma_w); t->gamma = pow(a->gamma * b->gamma, -1.5);
    }
  /* bonds: count line + one header line, two lines per bond */
  ok = ok && next_vals(fp, v, 16, NULL) >= 1;
  int nb = ok ? as_int(v[0]) : 0;
  ok = ok && fgets(line, sizeof line, fp) != NULL;
  for (int m = 0; ok && m < nb; m++) {
    ok = next_vals(fp, v, 16, NULL) >= 10;
    if (!ok) break;
    const int j = as_int(v[0]) - 1, k = as_int(v[1]) - 1;
    double u[16];
    ok = next_vals(fp, u, 16, NULL) >= 8;
    if (!ok) break;
    if (j < 0 || k < 0 || j >= ff->nt || k >= ff->nt) continue;
    for (int s = 0; s < 2; s++) {
      tbp_t *t = s ? &ff->tbp[k][j] : &ff->tbp[j][k];
      t->De_s = v[2]; t->De_p = v[3]; t->De_pp = v[4]; t->p_be1 = v[5]; t->p_bo5 = v[6]; t->v13cor = v[7]; t->p_bo6 = v[8]; t->p_ovun1 = v[9];
      t->p_be2 = u[0]; t->p_bo3 = u[1]; t->p_bo4 = u[2]; t->p_bo1 = u[4]; t->p_bo2 = u[5]; t->ovc = u[6];
    }
  }
  /* off-diagonal terms */
  ok = ok && next_vals(fp, v, 16, NULL) >= 1;
  int no = ok ? as_int(v[0]) : 0;
  for (int m = 0; ok && m < no; m++) {
    ok = next_vals(fp, v, 16, NULL) >= 8;
    if (!ok) break;
    const int j = as_int(v[0]) - 1, k = as_int(v[1]) - 1;
    if (j < 0 || k < 0 || j >= ff->nt || k >= ff->nt) continue;
    for (int s = 0; s < 2; s++) {
      tbp_t *t = s ? &ff->tbp[k][j] : &ff->tbp[j][k];
      if (v[2] > 0.0) t->D = v[2];
      if (v[3] > 0.0) t->r_vdW = 2.0 * v[3];
      if (v[4] > 0.0) t->alpha = v[4];
      if (v[5] > 0.0) t->r_s = v[5];
      if (v[6] > 0.0) t->r_p = v[6];
      if (v[7] > 0.0) t->r_pp = v[7];
    }
  }
  /* valence angles */
  ok = ok && next_vals(fp, v, 16, NULL) >= 1;
  int na = ok ? as_int(v[0]) : 0;
  for (int m = 0; ok && m < na; m++) {
    ok = next_vals(fp, v, 16, NULL) >= 10;
    if (!ok) break;
    const int j = as_int(v[0]) - 1, k = as_int(v[1]) - 1, l = as_int(v[2]) - 1;
    if (j < 0 || k < 0 || l < 0 || j >= ff->nt || k >= ff->nt || l >= ff->nt) continue;
    thbp_t *t1 = &ff->thbp[j][k][l], *t2 = &ff->thbp[l][k][j];
    if (t1->cnt >= RX_MAXANG) continue;
    thb_prm p = {v[3], v[4], v[5], v[6], v[7], v[8], v[9]};
    const int c = t1->cnt;
    t1->prm[c] = p; t1->cnt = c + 1;
    if (t2 != t1) { t2->prm[c] = p; t2->cnt = c + 1; }
  }
  /* torsions: specific quadruples win over the 0-j-k-0 wildcards, whatever their order in the file */
  ok = ok && next_vals(fp, v, 16, NULL) >= 1;
  int ntor = ok ? as_int(v[0]) : 0;
  for (int m = 0; ok && m < ntor; m++) {
    ok = next_vals(fp, v, 16, NULL) >= 9;
    if (!ok) break;
    const int j = as_int(v[0]) - 1, k = as_int(v[1]) - 1, l = as_int(v[2]) - 1, n = as_int(v[3]) - 1;
    if (k < 0 || l < 0 || k >= ff->nt || l >= ff->nt) continue;
    if (j >= 0 && n >= 0) {
      if (j >= ff->nt || n >= ff->nt) continue;
      for (int s = 0; s < 2; s++) {
        fbp_t *f = s ? &ff->fbp[n][l][k][j] : &ff->fbp[j][k][l][n];
        f->cnt = 1; f->specific = 1; f->V1 = v[4]; f->V2 = v[5]; f->V3 = v[6]; f->p_tor1 = v[7]; f->p_cot1 = v[8];
      }
    } else if (j < 0 && n < 0) {
      for (int p = 0; p < ff->nt; p++)
        for (int o = 0; o < ff->nt; o++)
          for (int s = 0; s < 2; s++) {
            fbp_t *f = s ? &ff->fbp[o][l][k][p] : &ff->fbp[p][k][l][o];
            if (f->specific) continue;
            f->cnt = 1; f->V1 = v[4]; f->V2 = v[5]; f->V3 = v[6]; f->p_tor1 = v[7]; f->p_cot1 = v[8];
          }
    }
  }
  /* hydrogen bonds (donor, hydrogen, acceptor) */
  ok = ok && next_vals(fp, v, 16, NULL) >= 1;
  int nh = ok ? as_int(v[0]) : 0;
  for (int m = 0; ok && m < nh; m++) {
    if (next_vals(fp, v, 16, NULL) < 7) break;
    const int j = as_int(v[0]) - 1, k = as_int(v[1]) - 1, l = as_int(v[2]) - 1;
    if (j < 0 || k < 0 || l < 0 || j >= ff->nt || k >= ff->nt || l >= ff->nt) continue;
    hbp_t *h = &ff->hbp[j][k][l];
    h->r0_hb = v[3]; h->p_hb1 = v[4]; h->p_hb2 = v[5]; h->p_hb3 = v[6];
  }
  fclose(fp);
  if (!ok) { fprintf(stderr, "reax_oracle: %s is not a complete ReaxFF force-field file\n", path); free(ff); return NULL; }
  ff->bo_cut = 0.01 * ff->gp[29];
  ff->swa = ff->gp[11];
  ff->swb = ff->gp[12];
  {  /* Init_Taper: 7th-order polynomial, 1 at swa, 0 with three vanishing derivatives at swb */
    const double a = ff->swa, b = ff->swb, d7 = pow(b - a, 7.0);
    const double a2 = a * a, a3 = a2 * a, b2 = b * b, b3 = b2 * b;
    ff->tap[7] = 20.0 / d7;
    ff->tap[6] = -70.0 * (a + b) / d7;
    ff->tap[5] = 84.0 * (a2 + 3.0 * a * b + b2) / d7;
    ff->tap[4] = -35.0 * (a3 + 9.0 * a2 * b + 9.0 * a * b2 + b3) / d7;
    ff->tap[3] = 140.0 * (a3 * b + 3.0 * a2 * b2 + a * b3) / d7;
    ff->tap[2] = -210.0 * (a3 * b2 + a2 * b3) / d7;
    ff->tap[1] = 140.0 * a3 * b3 / d7;
    ff->tap[0] = (-35.0 * a3 * b2 * b2 + 21.0 * a2 * b3 * b2 - 7.0 * a * b3 * b3 + b3 * b3 * b) / d7;
  }
  return ff;
}
void rxo_free_ffield(rxo_ff *ff) { free(ff); }
int rxo_ntypes(const rxo_ff *ff) { return ff->nt; }
const char *rxo_type_name(const rxo_ff *ff, int t) { return (t >= 0 && t < ff->nt) ? ff->sbp[t].name : ""; }
double rxo_type_mass(const rxo_ff *ff, int t) { return (t >= 0 && t < ff->nt) ? ff->sbp[t].mass : 0.0; }
double rxo_general(const rxo_ff *ff, int k) { return (k >= 0 && k < 64) ? ff->gp[k] : 0.0; }

/* ------------------------------------------------------------------ geometry */
typedef struct { int periodic; double lo[3], h[6], hinv[6]; } cell_t;
static void cell_make(const double *box, cell_t *c) {
  memset(c, 0, sizeof *c);
  if (!box) return;
  c->periodic = 1;
  for (int d = 0; d < 3; d++) c->lo[d] = box[d];
  c->h[0] = box[3] - box[0]; c->h[1] = box[4] - box[1]; c->h[2] = box[5] - box[2];
  c->h[3] = box[8]; c->h[4] = box[7]; c->h[5] = box[6];
  c->hinv[0] = 1.0 / c->h[0]; c->hinv[1] = 1.0 / c->h[1]; c->hinv[2] = 1.0 / c->h[2];
  c->hinv[3] = -c->h[3] / (c->h[1] * c->h[2]);
  c->hinv[4] = (c->h[3] * c->h[5] - c->h[1] * c->h[4]) / (c->h[0] * c->h[1] * c->h[2]);
  c->hinv[5] = -c->h[5] / (c->h[0] * c->h[1]);
}
static void minimg(const cell_t *c, double *d) {
  if (!c->periodic) return;
  double l0 = c->hinv[0] * d[0] + c->hinv[5] * d[1] + c->hinv[4] * d[2];
  double l1 = c->hinv[1] * d[1] + c->hinv[3] * d[2];
  double l2 = c->hinv[2] * d[2];
  l0 -= rint(l0); l1 -= rint(l1); l2 -= rint(l2);
  d[0] = c->h[0] * l0 + c->h[5] * l1 + c->h[4] * l2;
  d[1] = c->h[1] * l1 + c->h[3] * l2;
  d[2] = c->h[2] * l2;
}

/* bonds of the system: half list (i < j) + per-atom adjacency */
typedef struct { int i, j; double d[3], r; double BOp, BOp_s, BOp_pi, BOp_pi2; double BO, BO_s, BO_pi, BO_pi2; } bond_t;
typedef struct { int n, nb, cap; bond_t *b; int *adj_start, *adj; /* adj entry = bond index, sign: +(b+1) if atom is i, -(b+1) if j */
  double *total_bo, *Delta, *Delta_e, *Delta_boc, *Delta_val, *vlpex, *nlp, *Delta_lp, *dDelta_lp, *Delta_lp_temp, *dDelta_lp_temp; } work_t;

static void work_free(work_t *w) {
  free(w->b); free(w->adj_start); free(w->adj); free(w->total_bo); free(w->Delta); free(w->Delta_e); free(w->Delta_boc); free(w->Delta_val);
  free(w->vlpex); free(w->nlp); free(w->Delta_lp); free(w->dDelta_lp); free(w->Delta_lp_temp); free(w->dDelta_lp_temp);
}

/* BOp + BO of reaxc_bond_orders.cpp */
static void bond_orders(const rxo_ff *ff, int n, const int *type, const double *x, const cell_t *c, work_t *w) {
  memset(w, 0, sizeof *w);
  w->n = n;
  w->cap = 16 * n + 64;
  w->b = (bond_t *)calloc(w->cap, sizeof(bond_t));
  double *tb = w->total_bo = (double *)calloc(n, sizeof(double));
  for (int i = 0; i < n; i++)
    for (int j = i + 1; j < n; j++) {
      double d[3] = {x[3 * j] - x[3 * i], x[3 * j + 1] - x[3 * i + 1], x[3 * j + 2] - x[3 * i + 2]};
      minimg(c, d);
      const double r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
      if (r2 > BOND_CUT * BOND_CUT || r2 <= 0.0) continue;
      const double r = sqrt(r2);
      const sbp_t *si = &ff->sbp[type[i]], *sj = &ff->sbp[type[j]];
      const tbp_t *t = &ff->tbp[type[i]][type[j]];
      double BO_s = 0, BO_pi = 0, BO_pi2 = 0;
      if (si->r_s > 0.0 && sj->r_s > 0.0) BO_s = (1.0 + ff->bo_cut) * exp(t->p_bo1 * pow(r / t->r_s, t->p_bo2));
      if (si->r_pi > 0.0 && sj->r_pi > 0.0) BO_pi = exp(t->p_bo3 * pow(r / t->r_p, t->p_bo4));
      if (si->r_pi_pi > 0.0 && sj->r_pi_pi > 0.0) BO_pi2 = exp(t->p_bo5 * pow(r / t->r_pp, t->p_bo6));
      const double BO = BO_s + BO_pi + BO_pi2;
      if (BO < ff->bo_cut) continue;
      if (w->nb >= w->cap) { w->cap *= 2; w->b = (bond_t *)realloc(w->b, w->cap * sizeof(bond_t)); }
      bond_t *b = &w->b[w->nb++];
      memset(b, 0, sizeof *b);
      b->i = i; b->j = j; b->r = r;
      b->d[0] = d[0]; b->d[1] = d[1]; b->d[2] = d[2];
      b->BOp = BO - ff->bo_cut; b->BOp_s = BO_s - ff->bo_cut; b->BOp_pi = BO_pi; b->BOp_pi2 = BO_pi2;
      tb[i] += b->BOp;
      tb[j] += b->BOp;
    }
  /* adjacency */
  w->adj_start = (int *)calloc(n + 1, sizeof(int));
  for (int k = 0; k < w->nb; k++) { w->adj_start[w->b[k].i + 1]++; w->adj_start[w->b[k].j + 1]++; }
  for (int i = 0; i < n; i++) w->adj_start[i + 1] += w->adj_start[i];
  w->adj = (int *)calloc(2 * w->nb + 1, sizeof(int));
  int *fill = (int *)calloc(n, sizeof(int));
  for (int k = 0; k < w->nb; k++) {
    w->adj[w->adj_start[w->b[k].i] + fill[w->b[k].i]++] = k + 1;
    w->adj[w->adj_start[w->b[k].j] + fill[w->b[k].j]++] = -(k + 1);
  }
  free(fill);
  /* corrections */
  const double p_boc1 = ff->gp[0], p_boc2 = ff->gp[1];
  double *Deltap = (double *)calloc(n, sizeof(double)), *Deltap_boc = (double *)calloc(n, sizeof(double));
  for (int i = 0; i < n; i++) {
    Deltap[i] = tb[i] - ff->sbp[type[i]].valency;
    Deltap_boc[i] = tb[i] - ff->sbp[type[i]].valency_boc;
    tb[i] = 0.0;
  }
  for (int k = 0; k < w->nb; k++) {
    bond_t *b = &w->b[k];
    const int i = b->i, j = b->j;
    const tbp_t *t = &ff->tbp[type[i]][type[j]];
    if (t->ovc < 0.001 && t->v13cor < 0.001) {
      b->BO = b->BOp; b->BO_s = b->BOp_s; b->BO_pi = b->BOp_pi; b->BO_pi2 = b->BOp_pi2;
    } else {
      const double val_i = ff->sbp[type[i]].valency, val_j = ff->sbp[type[j]].valency;
      double f1 = 1.0, f4f5 = 1.0;
      if (t->ovc >= 0.001) {
        const double exp_p1i = exp(-p_boc1 * Deltap[i]), exp_p2i = exp(-p_boc2 * Deltap[i]);
        const double exp_p1j = exp(-p_boc1 * Deltap[j]), exp_p2j = exp(-p_boc2 * Deltap[j]);
        const double f2 = exp_p1i + exp_p1j;
        const double f3 = -1.0 / p_boc2 * log(0.5 * (exp_p2i + exp_p2j));
        f1 = 0.5 * ((val_i + f2) / (val_i + f2 + f3) + (val_j + f2) / (val_j + f2 + f3));
      }
      if (t->v13cor >= 0.001) {
        const double exp_f4 = exp(-(t->p_boc4 * SQR(b->BOp) - Deltap_boc[i]) * t->p_boc3 + t->p_boc5);
        const double exp_f5 = exp(-(t->p_boc4 * SQR(b->BOp) - Deltap_boc[j]) * t->p_boc3 + t->p_boc5);
        f4f5 = 1.0 / (1.0 + exp_f4) / (1.0 + exp_f5);
      }
      const double A0 = f1 * f4f5, A1 = A0 * f1;
      b->BO = b->BOp * A0;
      b->BO_pi = b->BOp_pi * A1;
      b->BO_pi2 = b->BOp_pi2 * A1;
      b->BO_s = b->BO - (b->BO_pi + b->BO_pi2);
    }
    if (b->BO < 1e-10) b->BO = 0.0;
    if (b->BO_s < 1e-10) b->BO_s = 0.0;
    if (b->BO_pi < 1e-10) b->BO_pi = 0.0;
    if (b->BO_pi2 < 1e-10) b->BO_pi2 = 0.0;
    tb[i] += b->BO;
    tb[j] += b->BO;
  }
  free(Deltap); free(Deltap_boc);
  w->Delta = (double *)calloc(n, sizeof(double)); w->Delta_e = (double *)calloc(n, sizeof(double));
  w->Delta_boc = (double *)calloc(n, sizeof(double)); w->Delta_val = (double *)calloc(n, sizeof(double));
  w->vlpex = (double *)calloc(n, sizeof(double)); w->nlp = (double *)calloc(n, sizeof(double));
  w->Delta_lp = (double *)calloc(n, sizeof(double)); w->dDelta_lp = (double *)calloc(n, sizeof(double));
  w->Delta_lp_temp = (double *)calloc(n, sizeof(double)); w->dDelta_lp_temp = (double *)calloc(n, sizeof(double));
  const double p_lp1 = ff->gp[15];
  for (int i = 0; i < n; i++) {
    const sbp_t *s = &ff->sbp[type[i]];
    w->Delta[i] = tb[i] - s->valency;
    w->Delta_e[i] = tb[i] - s->valency_e;
    w->Delta_boc[i] = tb[i] - s->valency_boc;
    w->Delta_val[i] = tb[i] - s->valency_val;
    w->vlpex[i] = w->Delta_e[i] - 2.0 * (int)(w->Delta_e[i] / 2.0);
    const double explp1 = exp(-p_lp1 * SQR(2.0 + w->vlpex[i]));
    w->nlp[i] = explp1 - (int)(w->Delta_e[i] / 2.0);
    w->Delta_lp[i] = s->nlp_opt - w->nlp[i];
    w->dDelta_lp[i] = 2.0 * p_lp1 * explp1 * (2.0 + w->vlpex[i]);
    if (s->mass > 21.0) {
      w->Delta_lp_temp[i] = s->nlp_opt - 0.5 * (s->valency_e - s->valency);
      w->dDelta_lp_temp[i] = 0.0;
    } else {
      w->Delta_lp_temp[i] = s->nlp_opt - w->nlp[i];
      w->dDelta_lp_temp[i] = w->dDelta_lp[i];
    }
  }
}

/* neighbour k of atom a through adjacency entry e: the other atom, the bond, and the vector a -> other */
static inline const bond_t *adj_bond(const work_t *w, int e, int *other, double d[3]) {
  const bond_t *b = &w->b[abs(e) - 1];
  if (e > 0) { *other = b->j; d[0] = b->d[0]; d[1] = b->d[1]; d[2] = b->d[2]; }
  else { *other = b->i; d[0] = -b->d[0]; d[1] = -b->d[1]; d[2] = -b->d[2]; }
  return b;
}
static double angle_of(const double *a, double ra, const double *b, double rb, double *cosv) {
  double c = (a[0] * b[0] + a[1] * b[1] + a[2] * b[2]) / (ra * rb);
  if (c > 1.0) c = 1.0;
  if (c < -1.0) c = -1.0;
  *cosv = c;
  return acos(c);
}

double rxo_energy(const rxo_ff *ff, int n, const int *type, const double *x, const double *box, const double *q, double parts[RXO_NPART]) {
  cell_t c;
  cell_make(box, &c);
  work_t w;
  bond_orders(ff, n, type, x, &c, &w);
  double e[RXO_NPART];
  memset(e, 0, sizeof e);
  const double *gp = ff->gp;
  /* ---- Bonds ---- */
  for (int k = 0; k < w.nb; k++) {
    const bond_t *b = &w.b[k];
    const tbp_t *t = &ff->tbp[type[b->i]][type[b->j]];
    const double pow_BOs_be2 = pow(b->BO_s, t->p_be2);
    const double exp_be12 = exp(t->p_be1 * (1.0 - pow_BOs_be2));
    e[RXO_BOND] += -t->De_s * b->BO_s * exp_be12 - t->De_p * b->BO_pi - t->De_pp * b->BO_pi2;
  }
  /* ---- Atom_Energy: lone pair, over- and under-coordination ---- */
  {
    const double p_ovun3 = gp[32], p_ovun4 = gp[31], p_ovun6 = gp[6], p_ovun7 = gp[8], p_ovun8 = gp[9];
    for (int i = 0; i < n; i++) {
      const sbp_t *s = &ff->sbp[type[i]];
      const double expvd2 = exp(-75.0 * w.Delta_lp[i]);
      e[RXO_LP] += s->p_lp2 * w.Delta_lp[i] / (1.0 + expvd2);
      const double dfvl = (s->mass > 21.0) ? 0.0 : 1.0;
      double sum_ovun1 = 0.0, sum_ovun2 = 0.0;
      for (int a = w.adj_start[i]; a < w.adj_start[i + 1]; a++) {
        int j;
        double d[3];
        const bond_t *b = adj_bond(&w, w.adj[a], &j, d);
        const tbp_t *t = &ff->tbp[type[i]][type[j]];
        sum_ovun1 += t->p_ovun1 * t->De_s * b->BO;
        sum_ovun2 += (w.Delta[j] - dfvl * w.Delta_lp_temp[j]) * (b->BO_pi + b->BO_pi2);
      }
      const double exp_ovun1 = p_ovun3 * exp(p_ovun4 * sum_ovun2);
      const double Delta_lpcorr = w.Delta[i] - (dfvl * w.Delta_lp_temp[i]) / (1.0 + exp_ovun1);
      const double exp_ovun2 = exp(s->p_ovun2 * Delta_lpcorr);
      const double inv_exp_ovun2 = 1.0 / (1.0 + exp_ovun2);
      const double DlpVi = 1.0 / (Delta_lpcorr + s->valency + 1e-8);
      e[RXO_OVER] += sum_ovun1 * Delta_lpcorr * DlpVi * inv_exp_ovun2;
      const double exp_ovun2n = 1.0 / exp_ovun2;
      const double exp_ovun6 = exp(p_ovun6 * Delta_lpcorr);
      const double exp_ovun8 = p_ovun7 * exp(p_ovun8 * sum_ovun2);
      e[RXO_UNDER] += -s->p_ovun5 * (1.0 - exp_ovun6) / (1.0 + exp_ovun2n) / (1.0 + exp_ovun8);
    }
  }
  /* ---- Valence_Angles: angle, penalty, three-body conjugation ---- */
  {
    const double p_val6 = gp[14], p_val8 = gp[33], p_val9 = gp[16], p_val10 = gp[17];
    const double p_pen2 = gp[19], p_pen3 = gp[20], p_pen4 = gp[21], p_coa2 = gp[2], p_coa3 = gp[38], p_coa4 = gp[30];
    for (int j = 0; j < n; j++) {
      const sbp_t *sj = &ff->sbp[type[j]];
      const int a0 = w.adj_start[j], a1 = w.adj_start[j + 1];
      double SBOp = 0.0, prod_SBO = 1.0;
      for (int a = a0; a < a1; a++) {
        const bond_t *b = &w.b[abs(w.adj[a]) - 1];
        SBOp += b->BO_pi + b->BO_pi2;
        double t8 = SQR(b->BO); t8 *= t8; t8 *= t8;
        prod_SBO *= exp(-t8);
      }
      const double vlpadj = (w.vlpex[j] >= 0.0) ? 0.0 : w.nlp[j];
      const double SBO = SBOp + (1.0 - prod_SBO) * (-w.Delta_boc[j] - p_val8 * vlpadj);
      double SBO2;
      if (SBO <= 0.0) SBO2 = 0.0;
      else if (SBO <= 1.0) SBO2 = pow(SBO, p_val9);
      else if (SBO < 2.0) SBO2 = 2.0 - pow(2.0 - SBO, p_val9);
      else SBO2 = 2.0;
      const double expval6 = exp(p_val6 * w.Delta_boc[j]);
      for (int ai = a0; ai < a1; ai++) {
        int i;
        double dji[3];
        const bond_t *bij = adj_bond(&w, w.adj[ai], &i, dji);
        const double BOA_ij = bij->BO - THB_CUT;
        if (!(BOA_ij > 0.0)) continue;
        for (int ak = ai + 1; ak < a1; ak++) {
          int k;
          double djk[3];
          const bond_t *bjk = adj_bond(&w, w.adj[ak], &k, djk);
          const double BOA_jk = bjk->BO - THB_CUT;
          if (!(BOA_jk > 0.0 && bij->BO > THB_CUT && bjk->BO > THB_CUT && bij->BO * bjk->BO > THB_CUTSQ)) continue;
          double cos_theta;
          const double theta = angle_of(dji, bij->r, djk, bjk->r, &cos_theta);
          const thbp_t *th = &ff->thbp[type[i]][type[j]][type[k]];
          for (int cnt = 0; cnt < th->cnt; cnt++) {
            const thb_prm *p = &th->prm[cnt];
            if (fabs(p->p_val1) <= 0.001) continue;
            const double exp3ij = exp(-sj->p_val3 * pow(BOA_ij, p->p_val4)), f7_ij = 1.0 - exp3ij;
            const double exp3jk = exp(-sj->p_val3 * pow(BOA_jk, p->p_val4)), f7_jk = 1.0 - exp3jk;
            const double expval7 = exp(-p->p_val7 * w.Delta_boc[j]);
            const double trm8 = 1.0 + expval6 + expval7;
            const double f8_Dj = sj->p_val5 - (sj->p_val5 - 1.0) * (2.0 + expval6) / trm8;
            const double theta_00 = p->theta_00 * M_PI / 180.0;
            const double theta_0 = M_PI - theta_00 * (1.0 - exp(-p_val10 * (2.0 - SBO2)));
            const double expval2theta = exp(-p->p_val2 * SQR(theta_0 - theta));
            const double expval12theta = (p->p_val1 >= 0.0) ? p->p_val1 * (1.0 - expval2theta) : p->p_val1 * -expval2theta;
            e[RXO_ANGLE] += f7_ij * f7_jk * f8_Dj * expval12theta;
            /* penalty */
            const double exp_pen2ij = exp(-p_pen2 * SQR(BOA_ij - 2.0)), exp_pen2jk = exp(-p_pen2 * SQR(BOA_jk - 2.0));
            const double exp_pen3 = exp(-p_pen3 * w.Delta[j]), exp_pen4 = exp(p_pen4 * w.Delta[j]);
            const double f9_Dj = (2.0 + exp_pen3) / (1.0 + exp_pen3 + exp_pen4);
            e[RXO_PEN] += p->p_pen1 * f9_Dj * exp_pen2ij * exp_pen2jk;
            /* three-body conjugation */
            const double exp_coa2 = exp(p_coa2 * w.Delta_val[j]);
            e[RXO_COA] += p->p_coa1 / (1.0 + exp_coa2) * exp(-p_coa3 * SQR(w.total_bo[i] - BOA_ij)) * exp(-p_coa3 * SQR(w.total_bo[k] - BOA_jk)) *
                          exp(-p_coa4 * SQR(BOA_ij - 1.5)) * exp(-p_coa4 * SQR(BOA_jk - 1.5));
          }
        }
      }
    }
  }
  /* ---- Torsion_Angles: torsion and four-body conjugation; every bond j-k once ---- */
  {
    const double p_tor2 = gp[23], p_tor3 = gp[24], p_tor4 = gp[25], p_cot2 = gp[27];
    for (int kb = 0; kb < w.nb; kb++) {
      const bond_t *bjk = &w.b[kb];
      if (!(bjk->BO > THB_CUT)) continue;
      const int j = bjk->i, k = bjk->j;
      const double BOA_jk = bjk->BO - THB_CUT;
      const double exp_tor2_jk = exp(-p_tor2 * BOA_jk), exp_cot2_jk = exp(-p_cot2 * SQR(BOA_jk - 1.5));
      const double DjDk = w.Delta_boc[j] + w.Delta_boc[k];
      const double exp_tor3 = exp(-p_tor3 * DjDk), exp_tor4 = exp(p_tor4 * DjDk);
      const double f11_DjDk = (2.0 + exp_tor3) / (1.0 + exp_tor3 + exp_tor4);
      const double djk[3] = {bjk->d[0], bjk->d[1], bjk->d[2]}, dkj[3] = {-bjk->d[0], -bjk->d[1], -bjk->d[2]};
      for (int ai = w.adj_start[j]; ai < w.adj_start[j + 1]; ai++) {
        int i;
        double dji[3];
        const bond_t *bij = adj_bond(&w, w.adj[ai], &i, dji);
        if (bij == bjk || !(bij->BO > THB_CUT)) continue;
        const double BOA_ij = bij->BO - THB_CUT;
        double cos_ijk;
        const double theta_ijk = angle_of(dji, bij->r, djk, bjk->r, &cos_ijk);
        double sin_ijk = sin(theta_ijk);
        if (sin_ijk >= 0 && sin_ijk <= MIN_SINE) sin_ijk = MIN_SINE;
        else if (sin_ijk <= 0 && sin_ijk >= -MIN_SINE) sin_ijk = -MIN_SINE;
        const double exp_tor2_ij = exp(-p_tor2 * BOA_ij), exp_cot2_ij = exp(-p_cot2 * SQR(BOA_ij - 1.5));
        for (int al = w.adj_start[k]; al < w.adj_start[k + 1]; al++) {
          int l;
          double dkl[3];
          const bond_t *bkl = adj_bond(&w, w.adj[al], &l, dkl);
          if (bkl == bjk || l == i) continue;
          const fbp_t *f = &ff->fbp[type[i]][type[j]][type[k]][type[l]];
          if (!(f->cnt && bkl->BO > THB_CUT && bij->BO * bjk->BO * bkl->BO > THB_CUT)) continue;
          const double BOA_kl = bkl->BO - THB_CUT;
          double cos_jkl;
          const double theta_jkl = angle_of(dkj, bjk->r, dkl, bkl->r, &cos_jkl);
          double sin_jkl = sin(theta_jkl);
          if (sin_jkl >= 0 && sin_jkl <= MIN_SINE) sin_jkl = MIN_SINE;
          else if (sin_jkl <= 0 && sin_jkl >= -MIN_SINE) sin_jkl = -MIN_SINE;
          /* dihedral i-j-k-l: normals of the planes (i,j,k) and (j,k,l) */
          double n1[3] = {dji[1] * djk[2] - dji[2] * djk[1], dji[2] * djk[0] - dji[0] * djk[2], dji[0] * djk[1] - dji[1] * djk[0]};
          double n2[3] = {dkj[1] * dkl[2] - dkj[2] * dkl[1], dkj[2] * dkl[0] - dkj[0] * dkl[2], dkj[0] * dkl[1] - dkj[1] * dkl[0]};
          const double nn = sqrt((n1[0] * n1[0] + n1[1] * n1[1] + n1[2] * n1[2]) * (n2[0] * n2[0] + n2[1] * n2[1] + n2[2] * n2[2]));
          double cos_omega = (nn > 0.0) ? (n1[0] * n2[0] + n1[1] * n2[1] + n1[2] * n2[2]) / nn : 1.0;
          /* n1 = (j->i) x (j->k), n2 = (k->j) x (k->l): parallel for the cis arrangement, cos(omega) = 1 there */
          if (cos_omega > 1.0) cos_omega = 1.0;
          if (cos_omega < -1.0) cos_omega = -1.0;
          const double cos2omega = 2.0 * SQR(cos_omega) - 1.0, cos3omega = cos_omega * (4.0 * SQR(cos_omega) - 3.0);
          const double exp_tor2_kl = exp(-p_tor2 * BOA_kl), exp_cot2_kl = exp(-p_cot2 * SQR(BOA_kl - 1.5));
          const double fn10 = (1.0 - exp_tor2_ij) * (1.0 - exp_tor2_jk) * (1.0 - exp_tor2_kl);
          const double exp_tor1 = exp(f->p_tor1 * SQR(2.0 - bjk->BO_pi - f11_DjDk));
          const double CV = 0.5 * (f->V1 * (1.0 + cos_omega) + f->V2 * exp_tor1 * (1.0 - cos2omega) + f->V3 * (1.0 + cos3omega));
          e[RXO_TORS] += fn10 * sin_ijk * sin_jkl * CV;
          const double fn12 = exp_cot2_ij * exp_cot2_jk * exp_cot2_kl;
          e[RXO_CONJ] += f->p_cot1 * fn12 * (1.0 + (SQR(cos_omega) - 1.0) * sin_ijk * sin_jkl);
        }
      }
    }
  }
  /* ---- Hydrogen_Bonds: donor i (p_hbond 2) - hydrogen j (p_hbond 1) ... acceptor k (p_hbond 2) ---- */
  for (int j = 0; j < n; j++) {
    if (ff->sbp[type[j]].p_hbond != 1) continue;
    for (int k = 0; k < n; k++) {
      if (k == j || ff->sbp[type[k]].p_hbond != 2) continue;
      double djk[3] = {x[3 * k] - x[3 * j], x[3 * k + 1] - x[3 * j + 1], x[3 * k + 2] - x[3 * j + 2]};
      minimg(&c, djk);
      const double r_jk = sqrt(djk[0] * djk[0] + djk[1] * djk[1] + djk[2] * djk[2]);
      if (r_jk > HBOND_CUT) continue;
      for (int a = w.adj_start[j]; a < w.adj_start[j + 1]; a++) {
        int i;
        double dji[3];
        const bond_t *bij = adj_bond(&w, w.adj[a], &i, dji);
        if (i == k || ff->sbp[type[i]].p_hbond != 2 || bij->BO < HB_THRESHOLD) continue;
        const hbp_t *h = &ff->hbp[type[i]][type[j]][type[k]];
        if (h->r0_hb <= 0.0) continue;
        double cos_theta;
        const double theta = angle_of(dji, bij->r, djk, r_jk, &cos_theta);
        const double s2 = sin(0.5 * theta), sin_xhz4 = SQR(SQR(s2));
        e[RXO_HB] += h->p_hb1 * (1.0 - exp(-h->p_hb2 * bij->BO)) * exp(-h->p_hb3 * (h->r0_hb / r_jk + r_jk / h->r0_hb - 2.0)) * sin_xhz4;
      }
    }
  }
  /* ---- vdW_Coulomb_Energy + polarisation ---- */
  {
    const double p_vdW1 = gp[28], p_vdW1i = 1.0 / p_vdW1;
    for (int i = 0; i < n; i++) {
      const sbp_t *si = &ff->sbp[type[i]];
      if (q) e[RXO_POL] += KCALPMOL_TO_EV * (si->chi * q[i] + 0.5 * si->eta * SQR(q[i]));
      for (int j = i + 1; j < n; j++) {
        double d[3] = {x[3 * j] - x[3 * i], x[3 * j + 1] - x[3 * i + 1], x[3 * j + 2] - x[3 * i + 2]};
        minimg(&c, d);
        const double r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        if (r2 > SQR(ff->swb)) continue;
        const double r = sqrt(r2);
        const tbp_t *t = &ff->tbp[type[i]][type[j]];
        double Tap = ff->tap[7];
        for (int m = 6; m >= 0; m--) Tap = Tap * r + ff->tap[m];
        /* shielded Morse-type van der Waals */
        const double fn13 = pow(pow(r, p_vdW1) + pow(1.0 / t->gamma_w, p_vdW1), p_vdW1i);
        const double exp1 = exp(t->alpha * (1.0 - fn13 / t->r_vdW)), exp2 = exp(0.5 * t->alpha * (1.0 - fn13 / t->r_vdW));
        e[RXO_VDW] += Tap * t->D * (exp1 - 2.0 * exp2);
        if (q) e[RXO_COUL] += Tap * C_ELE * q[i] * q[j] / cbrt(r2 * r + t->gamma);
      }
    }
  }
  work_free(&w);
  double tot = 0.0;
  for (int k = 0; k < RXO_NPART; k++) { tot += e[k]; if (parts) parts[k] = e[k]; }
  return tot;
}

int rxo_bond_orders(const rxo_ff *ff, int n, const int *type, const double *x, const double *box, int cap, int *ij, double *bo) {
  cell_t c;
  cell_make(box, &c);
  work_t w;
  bond_orders(ff, n, type, x, &c, &w);
  const int nb = w.nb;
  for (int k = 0; k < nb && k < cap; k++) {
    if (ij) { ij[2 * k] = w.b[k].i; ij[2 * k + 1] = w.b[k].j; }
    if (bo) { bo[3 * k] = w.b[k].BO; bo[3 * k + 1] = w.b[k].BO_pi; bo[3 * k + 2] = w.b[k].BO_pi2; }
  }
  work_free(&w);
  return nb;
}

/* ------------------------------------------------------------------ charge equilibration (fix qeq/reax) */
typedef struct { int n, nnz; int *row, *col; double *val, *dia; } hmat_t;
static void hmat_mul(const hmat_t *H, const double *x, double *y) {
  for (int i = 0; i < H->n; i++) y[i] = H->dia[i] * x[i];
  for (int k = 0; k < H->nnz; k++) { y[H->row[k]] += H->val[k] * x[H->col[k]]; y[H->col[k]] += H->val[k] * x[H->row[k]]; }
}
static int cg_solve(const hmat_t *H, const double *b, double *xs, double tol, int imax) {
  const int n = H->n;
  double *r = (double *)calloc(n, sizeof(double)), *d = (double *)calloc(n, sizeof(double)), *qv = (double *)calloc(n, sizeof(double)),
         *p = (double *)calloc(n, sizeof(double));
  hmat_mul(H, xs, qv);
  double b_norm = 0.0, sig_new = 0.0;
  for (int i = 0; i < n; i++) { r[i] = b[i] - qv[i]; d[i] = r[i] / H->dia[i]; b_norm += b[i] * b[i]; sig_new += r[i] * d[i]; }
  b_norm = sqrt(b_norm);
  int it;
  for (it = 1; it < imax && sqrt(sig_new) / b_norm > tol; it++) {
    hmat_mul(H, d, qv);
    double dq = 0.0;
    for (int i = 0; i < n; i++) dq += d[i] * qv[i];
    const double alpha = sig_new / dq;
    double sig_old = sig_new;
    sig_new = 0.0;
    for (int i = 0; i < n; i++) { xs[i] += alpha * d[i]; r[i] -= alpha * qv[i]; p[i] = r[i] / H->dia[i]; sig_new += r[i] * p[i]; }
    const double beta = sig_new / sig_old;
    for (int i = 0; i < n; i++) d[i] = p[i] + beta * d[i];
  }
  free(r); free(d); free(qv); free(p);
  return (it >= imax) ? -it : it;
}
int rxo_qeq(const rxo_ff *ff, int n, const int *type, const double *x, const double *box, double tol, int maxiter, double *q) {
  cell_t c;
  cell_make(box, &c);
  hmat_t H;
  H.n = n; H.nnz = 0;
  int cap = 64 * n + 64;
  H.row = (int *)malloc(cap * sizeof(int)); H.col = (int *)malloc(cap * sizeof(int)); H.val = (double *)malloc(cap * sizeof(double));
  H.dia = (double *)malloc(n * sizeof(double));
  for (int i = 0; i < n; i++) {
    H.dia[i] = ff->sbp[type[i]].eta;
    for (int j = i + 1; j < n; j++) {
      double d[3] = {x[3 * j] - x[3 * i], x[3 * j + 1] - x[3 * i + 1], x[3 * j + 2] - x[3 * i + 2]};
      minimg(&c, d);
      const double r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
      if (r2 > SQR(ff->swb)) continue;
      const double r = sqrt(r2);
      double Tap = ff->tap[7];
      for (int m = 6; m >= 0; m--) Tap = Tap * r + ff->tap[m];
      if (H.nnz >= cap) {
        cap *= 2;
        H.row = (int *)realloc(H.row, cap * sizeof(int)); H.col = (int *)realloc(H.col, cap * sizeof(int)); H.val = (double *)realloc(H.val, cap * sizeof(double));
      }
      H.row[H.nnz] = i; H.col[H.nnz] = j;
      H.val[H.nnz++] = Tap * EV_TO_KCALPMOL / cbrt(r2 * r + ff->tbp[type[i]][type[j]].gamma);
    }
  }
  double *bs = (double *)calloc(n, sizeof(double)), *bt = (double *)calloc(n, sizeof(double)), *s = (double *)calloc(n, sizeof(double)),
         *t = (double *)calloc(n, sizeof(double));
  for (int i = 0; i < n; i++) { bs[i] = -ff->sbp[type[i]].chi; bt[i] = -1.0; }
  const int it1 = cg_solve(&H, bs, s, tol, maxiter), it2 = cg_solve(&H, bt, t, tol, maxiter);
  double ss = 0.0, st = 0.0;
  for (int i = 0; i < n; i++) { ss += s[i]; st += t[i]; }
  const double u = ss / st;
  for (int i = 0; i < n; i++) q[i] = s[i] - u * t[i];
  free(bs); free(bt); free(s); free(t); free(H.row); free(H.col); free(H.val); free(H.dia);
  return (it1 < 0 || it2 < 0) ? -1 : it1 + it2;
}

/* ------------------------------------------------------------------ forces and virial by central differences */
void rxo_forces_fd(const rxo_ff *ff, int n, const int *type, const double *x, const double *box, const double *q, double h, double *f, double *virial) {
  double *xx = (double *)malloc(3 * (size_t)n * sizeof(double));
  memcpy(xx, x, 3 * (size_t)n * sizeof(double));
  if (f)
    for (int k = 0; k < 3 * n; k++) {
      xx[k] = x[k] + h;
      const double ep = rxo_energy(ff, n, type, xx, box, q, NULL);
      xx[k] = x[k] - h;
      const double em = rxo_energy(ff, n, type, xx, box, q, NULL);
      xx[k] = x[k];
      f[k] = -(ep - em) / (2.0 * h);
    }
  if (virial && box) {
    /* upper-triangular deformation gradient F = 1 + eps: keeps the box in LAMMPS' restricted triclinic form; for a
     * rotation-invariant energy dE/dF_ab (a <= b) is minus the symmetric virial W_ab */
    static const int A[6] = {0, 1, 2, 0, 0, 1}, B[6] = {0, 1, 2, 1, 2, 2};
    const double eps = 1e-6;
    for (int m = 0; m < 6; m++) {
      double ev[2];
      for (int s = 0; s < 2; s++) {
        const double e = s ? -eps : eps;
        double F[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        F[A[m]][B[m]] += e;
        const double av[3] = {box[3] - box[0], 0, 0}, bv[3] = {box[6], box[4] - box[1], 0}, cv[3] = {box[7], box[8], box[5] - box[2]};
        double a2[3], b2[3], c2[3], lo2[3];
        for (int r = 0; r < 3; r++) {
          a2[r] = F[r][0] * av[0] + F[r][1] * av[1] + F[r][2] * av[2];
          b2[r] = F[r][0] * bv[0] + F[r][1] * bv[1] + F[r][2] * bv[2];
          c2[r] = F[r][0] * cv[0] + F[r][1] * cv[1] + F[r][2] * cv[2];
          lo2[r] = F[r][0] * box[0] + F[r][1] * box[1] + F[r][2] * box[2];
        }
        double nb[9] = {lo2[0], lo2[1], lo2[2], lo2[0] + a2[0], lo2[1] + b2[1], lo2[2] + c2[2], b2[0], c2[0], c2[1]};
        for (int i = 0; i < n; i++)
          for (int r = 0; r < 3; r++) xx[3 * i + r] = F[r][0] * x[3 * i] + F[r][1] * x[3 * i + 1] + F[r][2] * x[3 * i + 2];
        ev[s] = rxo_energy(ff, n, type, xx, nb, q, NULL);
      }
      virial[m] = -(ev[0] - ev[1]) / (2.0 * eps);
    }
  }
  free(xx);
}

/* ------------------------------------------------------------------ tables for a second implementation
 * oracle/reax_torch.py restates the ENERGY of this file as one differentiable expression (reverse-mode differentiation then
 * gives forces and virial at the cost of an energy evaluation instead of 6N of them); it reads the parameters through this
 * export so that both work from the same reader.  Layouts (all double): gp[64]; sbp[nt][28] in the order of sbp_t with name
 * skipped and p_hbond as a double; tbp[nt][nt][25] in the order of tbp_t; thbp[nt][nt][nt][1 + 4*7] = cnt, then prm[4];
 * fbp[nt]^4[7] = cnt, specific, V1, V2, V3, p_tor1, p_cot1; hbp[nt]^3[4]; misc[11] = bo_cut, swa, swb, tap[0..7]. */
int rxo_export(const rxo_ff *ff, double *gp, double *sbp, double *tbp, double *thbp, double *fbp, double *hbp, double *misc) {
  const int nt = ff->nt;
  for (int k = 0; k < 64; k++) gp[k] = ff->gp[k];
  for (int i = 0; i < nt; i++) {
    const sbp_t *s = &ff->sbp[i];
    const double v[28] = {s->r_s, s->valency, s->mass, s->r_vdw, s->epsilon, s->gamma, s->r_pi, s->valency_e, s->nlp_opt, s->alpha, s->gamma_w,
                          s->valency_boc, s->p_ovun5, s->chi, s->eta, (double)s->p_hbond, s->r_pi_pi, s->p_lp2, s->b_o_131, s->b_o_132, s->b_o_133,
                          s->p_ovun2, s->p_val3, s->valency_val, s->p_val5, s->rcore2, s->ecore2, s->acore2};
    memcpy(sbp + 28 * i, v, sizeof v);
    for (int j = 0; j < nt; j++) {
      const tbp_t *t = &ff->tbp[i][j];
      const double u[25] = {t->De_s, t->De_p, t->De_pp, t->p_be1, t->p_bo5, t->v13cor, t->p_bo6, t->p_ovun1, t->p_be2, t->p_bo3, t->p_bo4, t->p_bo1, t->p_bo2,
                            t->ovc, t->r_s, t->r_p, t->r_pp, t->p_boc3, t->p_boc4, t->p_boc5, t->D, t->alpha, t->r_vdW, t->gamma_w, t->gamma};
      memcpy(tbp + 25 * (i * nt + j), u, sizeof u);
      for (int k = 0; k < nt; k++) {
        const thbp_t *th = &ff->thbp[i][j][k];
        double *o = thbp + 29 * ((i * nt + j) * nt + k);
        o[0] = th->cnt;
        for (int c = 0; c < RX_MAXANG; c++) {
          const thb_prm *p = &th->prm[c];
          const double w[7] = {p->theta_00, p->p_val1, p->p_val2, p->p_coa1, p->p_val7, p->p_pen1, p->p_val4};
          memcpy(o + 1 + 7 * c, w, sizeof w);
        }
        const hbp_t *h = &ff->hbp[i][j][k];
        double *oh = hbp + 4 * ((i * nt + j) * nt + k);
        oh[0] = h->r0_hb; oh[1] = h->p_hb1; oh[2] = h->p_hb2; oh[3] = h->p_hb3;
        for (int l = 0; l < nt; l++) {
          const fbp_t *f = &ff->fbp[i][j][k][l];
          double *of = fbp + 7 * (((i * nt + j) * nt + k) * nt + l);
          of[0] = f->cnt; of[1] = f->specific; of[2] = f->V1; of[3] = f->V2; of[4] = f->V3; of[5] = f->p_tor1; of[6] = f->p_cot1;
        }
      }
    }
  }
  misc[0] = ff->bo_cut; misc[1] = ff->swa; misc[2] = ff->swb;
  for (int k = 0; k < 8; k++) misc[3 + k] = ff->tap[k];
  return nt;
}
