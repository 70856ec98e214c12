/*
 * md_oracle.c -- CPU restatement (FP64) of the per-quadrature-point strained-MD stress
 * sampler of SCEMa (reference: headers/stmd_problem.h:84-383 driving the scripts
 * lammps_scripts/lammps_scripts_opls/{in.set,in.strain}.lammps and
 * ELASTIC/in.homogenization.lammps on LAMMPS 17Nov16).
 *
 * TEST INFRASTRUCTURE ONLY -- see md_oracle.h.  PARITY UNPINNED (LAMMPS is not available);
 * pinned by the known-answer/invariant tests in tests/test_oracle_*.py.
 *
 * What is restated, with the reference line that selects each piece:
 *   units real constants ............ in.set.lammps:13
 *   pair lj/cut/coul/long 12 9 ...... in.set.lammps:40, ELASTIC/potential.mod.lammps:5
 *   kspace 1e-4 (g_ewald rule) ...... in.set.lammps:36  (reciprocal part: plain Ewald sum --
 *                                      PPPM approximates exactly this sum)
 *   bond/angle harmonic, dihedral opls, improper harmonic ... in.set.lammps:44-57
 *   special_bonds lj/coul 0 0 1 ..... in.init.lammps:31 (carried by the restart file)
 *   neighbor 2.0 bin, every 1 delay 5 check yes ... in.set.lammps:27,32
 *   fix shake 0.001 20 1000 m 1.0 ... in.strain.lammps:71, in.homogenization.lammps:63
 *   fix nvt temp T T 100 ............ in.strain.lammps:80, in.homogenization.lammps:64
 *   fix deform 1 ... erate ... remap x ... in.strain.lammps:94-100
 *   fix ave/time 1 nav nav c_thermo_press[*] ave running ... in.homogenization.lammps:57-59
 *   host arithmetic (lbdim, nts, %.6e rates, atm->Pa) ... stmd_problem.h:210-244,335-341
 */
#include "md_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ---- units real (LAMMPS manual, "units real") ---- */
#define BOLTZ 0.0019872067
#define MVV2E (48.88821291 * 48.88821291)
#define FTM2V (1.0 / 48.88821291 / 48.88821291)
#define NKTV2P 68568.415
#define QQR2E 332.06371
#define QQRD2E 332.06371
#define EWALD_F 1.12837916709551257390 /* 2/sqrt(pi) */
#define MY_PI 3.14159265358979323846

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

void omd_default_params(omd_params *p) {
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
  p->kspace_pppm = 1;   /* kspace_style pppm, as in.set.lammps:36 asks; 0: the plain Ewald sum */
  p->pppm_mesh[0] = p->pppm_mesh[1] = p->pppm_mesh[2] = 0;
}

#define MAXCHAIN 8

struct omd_sim {
  int n, ntypes;
  int *type;
  double *q, *mass, *eps, *sigma;
  double *lj1, *lj2, *lj3, *lj4;
  int nbonds, nangles, ndihedrals, nimpropers;
  int *bond, *bond_type, *angle, *angle_type, *dihedral, *dihedral_type, *improper, *improper_type;
  double *bond_coeff, *angle_coeff, *dihedral_coeff, *improper_coeff;
  double special_lj[3], special_coul[3];
  omd_params p;
  /* state */
  double lo[3], hi[3], xy, xz, yz;
  double *x, *v, *f;
  /* special (1-2,1-3,1-4) pairs that are NOT plain pairs: excluded from the list */
  int nspecial;
  int *sp_i, *sp_j;
  double *sp_flj, *sp_fc;
  int *ex_start, *ex_list; /* per-atom exclusion lists (CSR) */
  /* shake */
  int nclus, ncons;
  int *clus_n;     /* atoms in cluster (2..4) */
  int *clus_atom;  /* 4 per cluster, [0] is the central atom */
  double *clus_d;  /* 3 per cluster */
  char *bond_shaken;
  int use_shake;
  /* ewald */
  double g_ewald;
  int nk, *kn;
  int pg[3];      /* PPPM grid (kspace_pppm) */
  double *pgf;    /* influence function of the box it was last computed for (pgf_key: box, g_ewald, grid): recomputed only when that changes */
  double pgf_key[13];
  int nflips; /* box flips applied so far (fix deform flip yes) */
  int kspace_frozen;
  double qsqsum, qsum;
  /* neighbour list (half, newton on) */
  int *wrapn; /* 3 per atom: x_wrapped = x - wrapn . h */
  double *xhold;
  double corners_hold[8][3];
  int npairs, pair_cap;
  int *pi, *pj;
  signed char *psh; /* 3 per pair */
  int ago;
  int nbuilds;
  /* nose-hoover chain */
  double eta[MAXCHAIN + 1], eta_dot[MAXCHAIN + 1], eta_dotdot[MAXCHAIN + 1], eta_mass[MAXCHAIN + 1];
  double t_current, tdof;
  /* last virial / energies */
  double vir[OMD_NPART * 6], eng[OMD_NPART];
  double timing[4];
  /* force field supplied from outside (omd_set_external_force): replaces every force routine of this file */
  omd_force_fn ext_fn;
  void *ext_ctx;
  int ext_calls;   /* force evaluations since the last omd_setup: 0 = the set-up evaluation of a run */
};

/* ------------------------------------------------------------------ box helpers */
typedef struct {
  double lo[3], h[6], hinv[6], vol;
} boxq;

static void box_derive(const omd_sim *s, boxq *b) {
  for (int d = 0; d < 3; d++) b->lo[d] = s->lo[d];
  b->h[0] = s->hi[0] - s->lo[0];
  b->h[1] = s->hi[1] - s->lo[1];
  b->h[2] = s->hi[2] - s->lo[2];
  b->h[3] = s->yz;
  b->h[4] = s->xz;
  b->h[5] = s->xy;
  b->hinv[0] = 1.0 / b->h[0];
  b->hinv[1] = 1.0 / b->h[1];
  b->hinv[2] = 1.0 / b->h[2];
  b->hinv[3] = -b->h[3] / (b->h[1] * b->h[2]);
  b->hinv[4] = (b->h[3] * b->h[5] - b->h[1] * b->h[4]) / (b->h[0] * b->h[1] * b->h[2]);
  b->hinv[5] = -b->h[5] / (b->h[0] * b->h[1]);
  b->vol = b->h[0] * b->h[1] * b->h[2];
}

/* general minimum image through fractional coordinates (bond-length scale vectors) */
static inline void minimg(const boxq *b, double d[3]) {
  double l0 = b->hinv[0] * d[0] + b->hinv[5] * d[1] + b->hinv[4] * d[2];
  double l1 = b->hinv[1] * d[1] + b->hinv[3] * d[2];
  double l2 = b->hinv[2] * d[2];
  l0 -= rint(l0);
  l1 -= rint(l1);
  l2 -= rint(l2);
  d[0] = b->h[0] * l0 + b->h[5] * l1 + b->h[4] * l2;
  d[1] = b->h[1] * l1 + b->h[3] * l2;
  d[2] = b->h[2] * l2;
}

static inline void shift_vec(const boxq *b, const signed char *sh, double out[3]) {
  out[0] = b->h[0] * sh[0] + b->h[5] * sh[1] + b->h[4] * sh[2];
  out[1] = b->h[1] * sh[1] + b->h[3] * sh[2];
  out[2] = b->h[2] * sh[2];
}

static inline void vtally(double *v, const double a[3], const double fb[3]) {
  v[0] += a[0] * fb[0];
  v[1] += a[1] * fb[1];
  v[2] += a[2] * fb[2];
  v[3] += a[0] * fb[1];
  v[4] += a[0] * fb[2];
  v[5] += a[1] * fb[2];
}

/* ------------------------------------------------------------------ creation */
static void *xcalloc(size_t n, size_t sz) {
  void *p = calloc(n ? n : 1, sz);
  if (!p) {
    fprintf(stderr, "md_oracle: out of memory\n");
    exit(1);
  }
  return p;
}
static void *dup_mem(const void *src, size_t bytes) {
  void *p = xcalloc(bytes ? bytes : 1, 1);
  if (bytes) memcpy(p, src, bytes);
  return p;
}

static void build_special(omd_sim *s) {
  int n = s->n;
  /* adjacency from all bonds */
  int *deg = (int *)xcalloc(n, sizeof(int));
  for (int b = 0; b < s->nbonds; b++) {
    deg[s->bond[2 * b]]++;
    deg[s->bond[2 * b + 1]]++;
  }
  int *start = (int *)xcalloc(n + 1, sizeof(int));
  for (int i = 0; i < n; i++) start[i + 1] = start[i] + deg[i];
  int *adj = (int *)xcalloc(start[n], sizeof(int));
  memset(deg, 0, n * sizeof(int));
  for (int b = 0; b < s->nbonds; b++) {
    int a = s->bond[2 * b], c = s->bond[2 * b + 1];
    adj[start[a] + deg[a]++] = c;
    adj[start[c] + deg[c]++] = a;
  }
  /* per atom: collect 1-2, 1-3, 1-4 partners, lowest level wins */
  int cap = 64 * n + 16;
  int *tmp_i = (int *)xcalloc(cap, sizeof(int));
  int *tmp_j = (int *)xcalloc(cap, sizeof(int));
  int *tmp_l = (int *)xcalloc(cap, sizeof(int));
  int cnt = 0;
  int *mark = (int *)xcalloc(n, sizeof(int)); /* level+1 stamp per root */
  int *lst = (int *)xcalloc(n, sizeof(int));
  for (int i = 0; i < n; i++) {
    int nl = 0;
    mark[i] = -1;
    lst[nl++] = i;
    int lvl_begin = 0, lvl_end = 1;
    for (int lvl = 1; lvl <= 3; lvl++) {
      for (int t = lvl_begin; t < lvl_end; t++) {
        int a = lst[t];
        for (int e = start[a]; e < start[a + 1]; e++) {
          int c = adj[e];
          if (mark[c] != 0) continue;
          mark[c] = lvl;
          lst[nl++] = c;
        }
      }
      lvl_begin = lvl_end;
      lvl_end = nl;
    }
    for (int t = 1; t < nl; t++) {
      int c = lst[t];
      if (c > i) {
        if (cnt >= cap) {
          fprintf(stderr, "md_oracle: special list overflow\n");
          exit(1);
        }
        tmp_i[cnt] = i;
        tmp_j[cnt] = c;
        tmp_l[cnt] = mark[c];
        cnt++;
      }
    }
    for (int t = 0; t < nl; t++) mark[lst[t]] = 0;
  }
  /* keep only pairs whose weights differ from (1,1) */
  s->sp_i = (int *)xcalloc(cnt, sizeof(int));
  s->sp_j = (int *)xcalloc(cnt, sizeof(int));
  s->sp_flj = (double *)xcalloc(cnt, sizeof(double));
  s->sp_fc = (double *)xcalloc(cnt, sizeof(double));
  s->nspecial = 0;
  int *exdeg = (int *)xcalloc(n + 1, sizeof(int));
  for (int k = 0; k < cnt; k++) {
    double flj = s->special_lj[tmp_l[k] - 1], fc = s->special_coul[tmp_l[k] - 1];
    if (flj == 1.0 && fc == 1.0) continue;
    int m = s->nspecial++;
    s->sp_i[m] = tmp_i[k];
    s->sp_j[m] = tmp_j[k];
    s->sp_flj[m] = flj;
    s->sp_fc[m] = fc;
    exdeg[tmp_i[k]]++;
    exdeg[tmp_j[k]]++;
  }
  s->ex_start = (int *)xcalloc(n + 1, sizeof(int));
  for (int i = 0; i < n; i++) s->ex_start[i + 1] = s->ex_start[i] + exdeg[i];
  s->ex_list = (int *)xcalloc(s->ex_start[n], sizeof(int));
  memset(exdeg, 0, (n + 1) * sizeof(int));
  for (int m = 0; m < s->nspecial; m++) {
    int a = s->sp_i[m], c = s->sp_j[m];
    s->ex_list[s->ex_start[a] + exdeg[a]++] = c;
    s->ex_list[s->ex_start[c] + exdeg[c]++] = a;
  }
  free(exdeg);
  free(deg);
  free(start);
  free(adj);
  free(tmp_i);
  free(tmp_j);
  free(tmp_l);
  free(mark);
  free(lst);
}

/* fix shake ... m <mass>: every bond that contains an atom whose type mass is within 0.1 of
 * <mass> is constrained; clusters = central atom + 1..3 satellites */
static void build_shake(omd_sim *s) {
  int n = s->n;
  s->bond_shaken = (char *)xcalloc(s->nbonds, 1);
  s->nclus = 0;
  s->ncons = 0;
  s->clus_n = (int *)xcalloc(n, sizeof(int));
  s->clus_atom = (int *)xcalloc(4 * (size_t)n, sizeof(int));
  s->clus_d = (double *)xcalloc(3 * (size_t)n, sizeof(double));
  if (s->p.shake_mass <= 0.0) return;
  int *nsh = (int *)xcalloc(n, sizeof(int)); /* shaken bonds per atom */
  for (int b = 0; b < s->nbonds; b++) {
    int a = s->bond[2 * b], c = s->bond[2 * b + 1];
    int ma = fabs(s->mass[s->type[a]] - s->p.shake_mass) <= 0.1;
    int mc = fabs(s->mass[s->type[c]] - s->p.shake_mass) <= 0.1;
    if (ma || mc) {
      s->bond_shaken[b] = 1;
      nsh[a]++;
      nsh[c]++;
    }
  }
  int *cl_of = (int *)xcalloc(n, sizeof(int));
  for (int i = 0; i < n; i++) cl_of[i] = -1;
  for (int b = 0; b < s->nbonds; b++) {
    if (!s->bond_shaken[b]) continue;
    int a = s->bond[2 * b], c = s->bond[2 * b + 1];
    /* central atom = the one with more shaken bonds; tie -> lower index */
    int cen, sat;
    if (nsh[a] > nsh[c] || (nsh[a] == nsh[c] && a < c)) {
      cen = a;
      sat = c;
    } else {
      cen = c;
      sat = a;
    }
    if (nsh[sat] != 1) {
      fprintf(stderr, "md_oracle: SHAKE cluster is not star shaped (atom %d)\n", sat);
      exit(1);
    }
    int cl = cl_of[cen];
    if (cl < 0) {
      cl = s->nclus++;
      cl_of[cen] = cl;
      s->clus_n[cl] = 1;
      s->clus_atom[4 * cl] = cen;
    }
    int k = s->clus_n[cl];
    if (k >= 4) {
      fprintf(stderr, "md_oracle: SHAKE cluster of more than 4 atoms\n");
      exit(1);
    }
    s->clus_atom[4 * cl + k] = sat;
    s->clus_d[3 * cl + (k - 1)] = s->bond_coeff[2 * s->bond_type[b] + 1];
    s->clus_n[cl] = k + 1;
    s->ncons++;
  }
  free(nsh);
  free(cl_of);
}

omd_sim *omd_create(int natoms, int ntypes, const int *type, const double *charge,
                    const double *mass_per_type, const double *eps, const double *sigma,
                    int nbonds, const int *bond_atoms, const int *bond_type, int nbondtypes,
                    const double *bond_coeff, int nangles, const int *angle_atoms,
                    const int *angle_type, int nangletypes, const double *angle_coeff,
                    int ndihedrals, const int *dihedral_atoms, const int *dihedral_type,
                    int ndihedraltypes, const double *dihedral_coeff, int nimpropers,
                    const int *improper_atoms, const int *improper_type, int nimpropertypes,
                    const double *improper_coeff, const double special_lj[3],
                    const double special_coul[3], const omd_params *params) {
  omd_sim *s = (omd_sim *)xcalloc(1, sizeof(omd_sim));
  s->n = natoms;
  s->ntypes = ntypes;
  s->type = (int *)dup_mem(type, natoms * sizeof(int));
  s->q = (double *)dup_mem(charge, natoms * sizeof(double));
  s->mass = (double *)dup_mem(mass_per_type, ntypes * sizeof(double));
  s->eps = (double *)dup_mem(eps, ntypes * ntypes * sizeof(double));
  s->sigma = (double *)dup_mem(sigma, ntypes * ntypes * sizeof(double));
  s->lj1 = (double *)xcalloc(ntypes * ntypes, sizeof(double));
  s->lj2 = (double *)xcalloc(ntypes * ntypes, sizeof(double));
  s->lj3 = (double *)xcalloc(ntypes * ntypes, sizeof(double));
  s->lj4 = (double *)xcalloc(ntypes * ntypes, sizeof(double));
  for (int k = 0; k < ntypes * ntypes; k++) {
    double s6 = pow(sigma[k], 6.0), s12 = s6 * s6;
    s->lj1[k] = 48.0 * eps[k] * s12;
    s->lj2[k] = 24.0 * eps[k] * s6;
    s->lj3[k] = 4.0 * eps[k] * s12;
    s->lj4[k] = 4.0 * eps[k] * s6;
  }
  s->nbonds = nbonds;
  s->bond = (int *)dup_mem(bond_atoms, 2 * (size_t)nbonds * sizeof(int));
  s->bond_type = (int *)dup_mem(bond_type, nbonds * sizeof(int));
  s->bond_coeff = (double *)dup_mem(bond_coeff, 2 * (size_t)nbondtypes * sizeof(double));
  s->nangles = nangles;
  s->angle = (int *)dup_mem(angle_atoms, 3 * (size_t)nangles * sizeof(int));
  s->angle_type = (int *)dup_mem(angle_type, nangles * sizeof(int));
  s->angle_coeff = (double *)dup_mem(angle_coeff, 2 * (size_t)nangletypes * sizeof(double));
  s->ndihedrals = ndihedrals;
  s->dihedral = (int *)dup_mem(dihedral_atoms, 4 * (size_t)ndihedrals * sizeof(int));
  s->dihedral_type = (int *)dup_mem(dihedral_type, ndihedrals * sizeof(int));
  s->dihedral_coeff = (double *)dup_mem(dihedral_coeff, 4 * (size_t)ndihedraltypes * sizeof(double));
  s->nimpropers = nimpropers;
  s->improper = (int *)dup_mem(improper_atoms, 4 * (size_t)nimpropers * sizeof(int));
  s->improper_type = (int *)dup_mem(improper_type, nimpropers * sizeof(int));
  s->improper_coeff = (double *)dup_mem(improper_coeff, 2 * (size_t)nimpropertypes * sizeof(double));
  for (int k = 0; k < 3; k++) {
    s->special_lj[k] = special_lj[k];
    s->special_coul[k] = special_coul[k];
  }
  if (params)
    s->p = *params;
  else
    omd_default_params(&s->p);
  if (s->p.t_chain > MAXCHAIN) s->p.t_chain = MAXCHAIN;
  s->x = (double *)xcalloc(3 * (size_t)natoms, sizeof(double));
  s->v = (double *)xcalloc(3 * (size_t)natoms, sizeof(double));
  s->f = (double *)xcalloc(3 * (size_t)natoms, sizeof(double));
  s->wrapn = (int *)xcalloc(3 * (size_t)natoms, sizeof(int));
  s->xhold = (double *)xcalloc(3 * (size_t)natoms, sizeof(double));
  s->qsqsum = 0.0;
  s->qsum = 0.0;
  for (int i = 0; i < natoms; i++) {
    s->qsqsum += charge[i] * charge[i];
    s->qsum += charge[i];
  }
  build_special(s);
  build_shake(s);
  s->tdof = 3.0 * natoms - 3.0;
  return s;
}

void omd_destroy(omd_sim *s) {
  if (!s) return;
  free(s->type); free(s->q); free(s->mass); free(s->eps); free(s->sigma);
  free(s->lj1); free(s->lj2); free(s->lj3); free(s->lj4);
  free(s->bond); free(s->bond_type); free(s->bond_coeff);
  free(s->angle); free(s->angle_type); free(s->angle_coeff);
  free(s->dihedral); free(s->dihedral_type); free(s->dihedral_coeff);
  free(s->improper); free(s->improper_type); free(s->improper_coeff);
  free(s->x); free(s->v); free(s->f); free(s->wrapn); free(s->xhold);
  free(s->sp_i); free(s->sp_j); free(s->sp_flj); free(s->sp_fc); free(s->ex_start); free(s->ex_list);
  free(s->clus_n); free(s->clus_atom); free(s->clus_d); free(s->bond_shaken);
  free(s->kn); free(s->pi); free(s->pj); free(s->psh); free(s->pgf);
  free(s);
}

void omd_set_state(omd_sim *s, const double box[9], const double *x, const double *v) {
  for (int d = 0; d < 3; d++) {
    s->lo[d] = box[d];
    s->hi[d] = box[3 + d];
  }
  s->xy = box[6];
  s->xz = box[7];
  s->yz = box[8];
  memcpy(s->x, x, 3 * (size_t)s->n * sizeof(double));
  memcpy(s->v, v, 3 * (size_t)s->n * sizeof(double));
}
void omd_get_state(const omd_sim *s, double box[9], double *x, double *v) {
  for (int d = 0; d < 3; d++) {
    box[d] = s->lo[d];
    box[3 + d] = s->hi[d];
  }
  box[6] = s->xy;
  box[7] = s->xz;
  box[8] = s->yz;
  if (x) memcpy(x, s->x, 3 * (size_t)s->n * sizeof(double));
  if (v) memcpy(v, s->v, 3 * (size_t)s->n * sizeof(double));
}
int omd_natoms(const omd_sim *s) { return s->n; }
int omd_nconstraints(const omd_sim *s) { return s->ncons; }
int omd_nclusters(const omd_sim *s) { return s->nclus; }
double omd_tdof(const omd_sim *s) { return s->tdof; }
double omd_g_ewald(const omd_sim *s) { return s->g_ewald; }
int omd_nkvec(const omd_sim *s) { return s->nk; }
void omd_pppm_grid(const omd_sim *s, int n[3]) { n[0] = s->pg[0]; n[1] = s->pg[1]; n[2] = s->pg[2]; }
int omd_npairs(const omd_sim *s) { return s->npairs; }
int omd_nflips(const omd_sim *s) { return s->nflips; }
void omd_last_timing(const omd_sim *s, double t[4]) {
  for (int k = 0; k < 4; k++) t[k] = s->timing[k];
}

/* ------------------------------------------------------------------ Ewald setup */
/* g_ewald rule of "kspace_style pppm <acc>" (LAMMPS manual kspace_style) */

/* ------------------------------------------------------------------ PPPM (kspace_style pppm 1e-4, in.set.lammps:36)
 * Restated from the published algorithm (Hockney & Eastwood; LAMMPS pppm.cpp 17Nov16 as remembered) [LAMMPS-ext, PARITY
 * UNPINNED]: order-5 charge assignment on a grid in lamda coordinates (so triclinic boxes need nothing special), optimal
 * influence function for ik differentiation with the alias sums taken directly (|m| <= 2 per dimension, numerator and
 * denominator alike; pppm.cpp has the denominator in closed form), energy and virial in reciprocal space, three inverse
 * transforms for the field, forces by the same weights.  Grid and g_ewald: set_grid_global / adjust_gewald with the loop
 * structure of pppm.cpp (see pppm_setup below).  Transforms are plain O(n^2) sums per line. */
#define PPPM_ORDER 5
#define PPPM_GRID_GUARD 1.0e-9
static const double ACONS5[5] = {1.0 / 23232.0, 7601.0 / 13628160.0, 143.0 / 69120.0, 517231.0 / 106536960.0, 106640677.0 / 11737571328.0};

static double pppm_ik_error(double h, double prd, double g, double q2, double natoms) {
  double sum = 0.0;
  for (int m = 0; m < PPPM_ORDER; m++) sum += ACONS5[m] * pow(h * g, 2.0 * m);
  return q2 * pow(h * g, (double)PPPM_ORDER) * sqrt(g * prd * sqrt(2.0 * MY_PI) * sum / natoms) / (prd * prd);
}
static int pppm_factorable(int n) {
  while (n % 2 == 0) n /= 2;
  while (n % 3 == 0) n /= 3;
  while (n % 5 == 0) n /= 5;
  return n == 1;
}
/* grid spacings h_x, h_y, h_z as set_grid_global leaves them: prd / n in an orthogonal box; in a triclinic one the
 * reciprocal of x2lamdaT(n) (the tilts enter h_y and h_z).  b->h = xprd, yprd, zprd, yz, xz, xy. */
static void pppm_spacings(const boxq *b, int triclinic, const int pg[3], double hh[3]) {
  if (!triclinic) {
    for (int d = 0; d < 3; d++) hh[d] = b->h[d] / pg[d];
    return;
  }
  const double v0 = pg[0], v1 = pg[1], v2 = pg[2];
  hh[0] = 1.0 / (b->hinv[0] * v0);
  hh[1] = 1.0 / (b->hinv[5] * v0 + b->hinv[1] * v1);
  hh[2] = 1.0 / (b->hinv[4] * v0 + b->hinv[3] * v1 + b->hinv[2] * v2);
}
/* newton_raphson_f: real-space error estimate minus compute_df_kspace (ik: RMS of the three per-dimension estimates) */
static double pppm_f(const omd_sim *s, const boxq *b, double g, double q2, const double hh[3]) {
  const double rc = s->p.cut_coul, N = (double)s->n;
  const double df_r = 2.0 * q2 * exp(-g * g * rc * rc) / sqrt(N * rc * b->h[0] * b->h[1] * b->h[2]);
  const double lprx = pppm_ik_error(hh[0], b->h[0], g, q2, N), lpry = pppm_ik_error(hh[1], b->h[1], g, q2, N),
               lprz = pppm_ik_error(hh[2], b->h[2], g, q2, N);
  return df_r - sqrt(lprx * lprx + lpry * lpry + lprz * lprz) / sqrt(3.0);
}
/* PPPM::set_grid_global + adjust_gewald of pppm.cpp (17Nov16, ik differentiation, no stagger) with their loop structure:
 *  - per dimension start at n = int(prd * g) + 1 with h = 1/g; `while (err > accuracy) { err = E(h); n++; h = prd/n; }`
 *    -- the increment follows the evaluation, so the loop leaves n ONE PAST the first grid whose estimate is admissible
 *    (or at the start value if E(1/g) already is);
 *  - a triclinic box (the reference's replicas always are one: in.init.lammps:27 `change_box all triclinic`) rescales:
 *    n = int(lamda2xT(n / prd)) + 1 -- with zero tilts that is n or n + 1 depending on how (n / prd) * prd rounds (guarded
 *    here, see below);
 *  - raise each n to a product of 2, 3, 5;
 *  - g_ewald: Newton steps with a forward-difference derivative of step 1e-6, stopping at the first iterate with
 *    |f| < SMALL = 1e-5 (not at convergence). */
static void pppm_setup(omd_sim *s, const boxq *b, double accuracy, double q2) {
  double g = s->g_ewald;
  const double N = (double)s->n;
  const int triclinic = 1;   /* the engine's box model (9 numbers with tilts) is LAMMPS' triclinic box, whatever the tilts are */
  int n[3];
  const int gridflag = s->p.pppm_mesh[0] > 0 && s->p.pppm_mesh[1] > 0 && s->p.pppm_mesh[2] > 0;
  for (int d = 0; d < 3 && gridflag; d++) n[d] = s->p.pppm_mesh[d];
  for (int d = 0; d < 3 && !gridflag; d++) {
    const double prd = b->h[d];
    double h = 1.0 / g;
    n[d] = (int)(prd / h) + 1;
    double err = pppm_ik_error(h, prd, g, q2, N);
    while (err > accuracy) {
      err = pppm_ik_error(h, prd, g, q2, N);
      n[d]++;
      h = prd / n[d];
    }
  }
  if (triclinic && !gridflag) {
    const double t0 = n[0] / b->h[0], t1 = n[1] / b->h[1], t2 = n[2] / b->h[2];
    const double u0 = b->h[0] * t0, u1 = b->h[5] * t0 + b->h[1] * t1, u2 = b->h[4] * t0 + b->h[3] * t1 + b->h[2] * t2;
    /* (n / prd) * prd is n or one ulp beside it, so without a tilt contribution the truncation is a coin flip on the last bit
     * of the box length (a 1e-12 change of the box moves a grid by one).  A guard of 1e-9 grid points decides such cases as
     * exact arithmetic would (n + 1, also LAMMPS' answer whenever the product rounds to n or above): the one place where this
     * restatement is deliberately not bit-faithful, because a checker cannot be a coin flip. */
    n[0] = (int)(u0 + PPPM_GRID_GUARD) + 1; n[1] = (int)(u1 + PPPM_GRID_GUARD) + 1; n[2] = (int)(u2 + PPPM_GRID_GUARD) + 1;
  }
  for (int d = 0; d < 3; d++) {
    while (!pppm_factorable(n[d])) n[d]++;
    s->pg[d] = n[d];
  }
  double hh[3];
  pppm_spacings(b, triclinic, s->pg, hh);
  /* adjust_gewald */
  for (int it = 0; it < 10000; it++) {
    const double step = 0.000001;
    const double f1 = pppm_f(s, b, g, q2, hh), f2 = pppm_f(s, b, g + step, q2, hh);
    const double dx = f1 / ((f2 - f1) / step);
    g -= dx;
    if (fabs(pppm_f(s, b, g, q2, hh)) < 0.00001) break;
  }
  s->g_ewald = g;
}

/* weights of the order-P cardinal B-spline at the P grid points around u (nearest point i = floor(u + 1/2) for odd P):
 * w[k] for the grid points i - 2 .. i + 2 */
static int pppm_weights(double u, double w[PPPM_ORDER]) {
  const int i = (int)floor(u + 0.5);
  const double dx = (double)i - u;   /* in (-1/2, 1/2] */
  for (int k = 0; k < PPPM_ORDER; k++) {
    /* M_P(t), t = distance of the point from the left end of the spline's support */
    const double t = -dx - (double)(k - 2) + 0.5 * PPPM_ORDER;
    double m[PPPM_ORDER + 1];
    /* M_1 on the unit intervals [j, j+1) that t - j can fall into */
    for (int j = 0; j < PPPM_ORDER; j++) m[j] = (t - j >= 0.0 && t - j < 1.0) ? 1.0 : 0.0;
    for (int n = 2; n <= PPPM_ORDER; n++)
      for (int j = 0; j + n <= PPPM_ORDER; j++) {
        const double tj = t - j;
        m[j] = (tj * m[j] + ((double)n - tj) * m[j + 1]) / (double)(n - 1);
      }
    w[k] = m[0];
  }
  return i;
}
static inline int pmod(int a, int n) { int r = a % n; return r < 0 ? r + n : r; }
/* in-place 3-d transform of a complex grid [nz][ny][nx] (x fastest), sign = -1 forward / +1 backward, unnormalised */
static void pppm_dft3(double *re, double *im, const int n[3], int sign) {
  const int nx = n[0], ny = n[1], nz = n[2];
  int nmax = nx > ny ? nx : ny;
  if (nz > nmax) nmax = nz;
  double *tr = (double *)xcalloc((size_t)nmax, sizeof(double)), *ti = (double *)xcalloc((size_t)nmax, sizeof(double));
  double *cw = (double *)xcalloc((size_t)nmax, sizeof(double)), *sw = (double *)xcalloc((size_t)nmax, sizeof(double));
  for (int dim = 0; dim < 3; dim++) {
    const int len = n[dim];
    const size_t stride = dim == 0 ? 1 : (dim == 1 ? (size_t)nx : (size_t)nx * ny);
    for (int k = 0; k < len; k++) { cw[k] = cos(2.0 * MY_PI * k / len); sw[k] = sign * sin(2.0 * MY_PI * k / len); }
    const int n1 = dim == 0 ? ny : nx, n2 = dim == 2 ? ny : nz;
    const size_t s1 = dim == 0 ? (size_t)nx : 1, s2 = dim == 2 ? (size_t)nx : (size_t)nx * ny;
    for (int a2 = 0; a2 < n2; a2++)
      for (int a1 = 0; a1 < n1; a1++) {
        const size_t base = (size_t)a1 * s1 + (size_t)a2 * s2;
        for (int k = 0; k < len; k++) { tr[k] = re[base + k * stride]; ti[k] = im[base + k * stride]; }
        for (int q = 0; q < len; q++) {
          double ar = 0.0, ai = 0.0;
          for (int k = 0; k < len; k++) {
            const int idx = (int)(((long long)q * k) % len);
            ar += tr[k] * cw[idx] - ti[k] * sw[idx];
            ai += tr[k] * sw[idx] + ti[k] * cw[idx];
          }
          re[base + q * stride] = ar;
          im[base + q * stride] = ai;
        }
      }
  }
  free(tr); free(ti); free(cw); free(sw);
}
static inline double sinc_pow(double x, int p) {
  if (x == 0.0) return 1.0;
  return pow(sin(x) / x, (double)p);
}
static void pppm_compute(omd_sim *s, const boxq *b, double *f, double *eng, double *vir) {
  if (s->g_ewald == 0.0 || s->pg[0] == 0) return;
  double t0 = now_s();
  const int n = s->n, nx = s->pg[0], ny = s->pg[1], nz = s->pg[2];
  const int ng[3] = {nx, ny, nz};
  const size_t NG = (size_t)nx * ny * nz;
  const double g = s->g_ewald;
  double *rr = (double *)xcalloc(NG, sizeof(double)), *ri = (double *)xcalloc(NG, sizeof(double));
  double *wts = (double *)xcalloc((size_t)n * 3 * PPPM_ORDER, sizeof(double));
  int *ibase = (int *)xcalloc((size_t)n * 3, sizeof(int));
  /* make_rho: charge density (charge per volume) on the grid */
  const double delvolinv = (double)NG / b->vol;
  for (int i = 0; i < n; i++) {
    const double d0 = s->x[3 * i] - b->lo[0], d1 = s->x[3 * i + 1] - b->lo[1], d2 = s->x[3 * i + 2] - b->lo[2];
    double l[3];
    l[0] = b->hinv[0] * d0 + b->hinv[5] * d1 + b->hinv[4] * d2;
    l[1] = b->hinv[1] * d1 + b->hinv[3] * d2;
    l[2] = b->hinv[2] * d2;
    for (int d = 0; d < 3; d++) {
      const double u = (l[d] - floor(l[d])) * ng[d];
      ibase[3 * i + d] = pppm_weights(u, wts + ((size_t)i * 3 + d) * PPPM_ORDER);
    }
    const double *wx = wts + ((size_t)i * 3) * PPPM_ORDER, *wy = wx + PPPM_ORDER, *wz = wy + PPPM_ORDER;
    const double z0 = delvolinv * s->q[i];
    for (int c = 0; c < PPPM_ORDER; c++) {
      const int gz = pmod(ibase[3 * i + 2] + c - 2, nz);
      for (int bb = 0; bb < PPPM_ORDER; bb++) {
        const int gy = pmod(ibase[3 * i + 1] + bb - 2, ny);
        const double zy = z0 * wz[c] * wy[bb];
        for (int a = 0; a < PPPM_ORDER; a++) {
          const int gx = pmod(ibase[3 * i] + a - 2, nx);
          rr[((size_t)gz * ny + gy) * nx + gx] += zy * wx[a];
        }
      }
    }
  }
  pppm_dft3(rr, ri, ng, -1);
  /* influence function (ik differentiation, alias sums |m| <= 2): a function of the box, g_ewald and the grid only */
  {
    double key[13];
    for (int k = 0; k < 3; k++) { key[k] = b->lo[k]; key[3 + k] = b->h[k]; key[6 + k] = b->h[3 + k]; key[10 + k] = (double)ng[k]; }
    key[9] = g;
    if (!s->pgf || memcmp(key, s->pgf_key, sizeof key) != 0) {
      free(s->pgf);
      s->pgf = (double *)xcalloc(NG, sizeof(double));
      memcpy(s->pgf_key, key, sizeof key);
      /* squared transform of the assignment function: the grid lives in lamda space, so it factorises over the lattice indices */
      double *w1 = (double *)xcalloc((size_t)5 * nx, sizeof(double)), *w2d = (double *)xcalloc((size_t)5 * ny, sizeof(double)), *w3 = (double *)xcalloc((size_t)5 * nz, sizeof(double));
      for (int m = 0; m < nx; m++) for (int a = -2; a <= 2; a++) w1[5 * m + a + 2] = sinc_pow(MY_PI * (m - nx * (2 * m / nx) + nx * a) / nx, 2 * PPPM_ORDER);
      for (int m = 0; m < ny; m++) for (int a = -2; a <= 2; a++) w2d[5 * m + a + 2] = sinc_pow(MY_PI * (m - ny * (2 * m / ny) + ny * a) / ny, 2 * PPPM_ORDER);
      for (int m = 0; m < nz; m++) for (int a = -2; a <= 2; a++) w3[5 * m + a + 2] = sinc_pow(MY_PI * (m - nz * (2 * m / nz) + nz * a) / nz, 2 * PPPM_ORDER);
      const double g2inv_ = 1.0 / (g * g);
      for (int m3 = 0; m3 < nz; m3++) {
        const int p3 = m3 - nz * (2 * m3 / nz);
        for (int m2 = 0; m2 < ny; m2++) {
          const int p2 = m2 - ny * (2 * m2 / ny);
          for (int m1 = 0; m1 < nx; m1++) {
            const int p1 = m1 - nx * (2 * m1 / nx);
            const size_t idx = ((size_t)m3 * ny + m2) * nx + m1;
            if (p1 == 0 && p2 == 0 && p3 == 0) continue;
            const double kx = 2.0 * MY_PI * (b->hinv[0] * p1);
            const double ky = 2.0 * MY_PI * (b->hinv[5] * p1 + b->hinv[1] * p2);
            const double kz = 2.0 * MY_PI * (b->hinv[4] * p1 + b->hinv[3] * p2 + b->hinv[2] * p3);
            const double sqk = kx * kx + ky * ky + kz * kz;
            double num = 0.0, den = 0.0;
            for (int a3 = -2; a3 <= 2; a3++)
              for (int a2 = -2; a2 <= 2; a2++)
                for (int a1 = -2; a1 <= 2; a1++) {
                  const int q1 = p1 + nx * a1, q2_ = p2 + ny * a2, q3 = p3 + nz * a3;
                  const double qx = 2.0 * MY_PI * (b->hinv[0] * q1);
                  const double qy = 2.0 * MY_PI * (b->hinv[5] * q1 + b->hinv[1] * q2_);
                  const double qz = 2.0 * MY_PI * (b->hinv[4] * q1 + b->hinv[3] * q2_ + b->hinv[2] * q3);
                  const double dot2 = qx * qx + qy * qy + qz * qz;
                  const double w2 = w1[5 * m1 + a1 + 2] * w2d[5 * m2 + a2 + 2] * w3[5 * m3 + a3 + 2];
                  den += w2;
                  num += (kx * qx + ky * qy + kz * qz) / dot2 * exp(-0.25 * dot2 * g2inv_) * w2;
                }
            s->pgf[idx] = 4.0 * MY_PI / sqk * num / (den * den);
          }
        }
      }
      free(w1); free(w2d); free(w3);
    }
  }
  /* poisson_ik: energy, virial, field spectra */
  double *ex = (double *)xcalloc(NG, sizeof(double)), *exi = (double *)xcalloc(NG, sizeof(double));
  double *ey = (double *)xcalloc(NG, sizeof(double)), *eyi = (double *)xcalloc(NG, sizeof(double));
  double *ez = (double *)xcalloc(NG, sizeof(double)), *ezi = (double *)xcalloc(NG, sizeof(double));
  const double scaleinv = 1.0 / (double)NG, g2inv = 1.0 / (g * g);
  double e = 0.0, v[6] = {0, 0, 0, 0, 0, 0};
  for (int m3 = 0; m3 < nz; m3++) {
    const int p3 = m3 - nz * (2 * m3 / nz);
    for (int m2 = 0; m2 < ny; m2++) {
      const int p2 = m2 - ny * (2 * m2 / ny);
      for (int m1 = 0; m1 < nx; m1++) {
        const int p1 = m1 - nx * (2 * m1 / nx);
        const size_t idx = ((size_t)m3 * ny + m2) * nx + m1;
        if (p1 == 0 && p2 == 0 && p3 == 0) continue;
        const double kx = 2.0 * MY_PI * (b->hinv[0] * p1);
        const double ky = 2.0 * MY_PI * (b->hinv[5] * p1 + b->hinv[1] * p2);
        const double kz = 2.0 * MY_PI * (b->hinv[4] * p1 + b->hinv[3] * p2 + b->hinv[2] * p3);
        const double sqk = kx * kx + ky * ky + kz * kz;
        const double gf = s->pgf[idx];
        const double ar = rr[idx] * scaleinv, ai = ri[idx] * scaleinv;
        const double eg = gf * (ar * ar + ai * ai);
        e += eg;
        const double vterm = -2.0 * (1.0 / sqk + 0.25 * g2inv);
        v[0] += eg * (1.0 + vterm * kx * kx); v[1] += eg * (1.0 + vterm * ky * ky); v[2] += eg * (1.0 + vterm * kz * kz);
        v[3] += eg * vterm * kx * ky; v[4] += eg * vterm * kx * kz; v[5] += eg * vterm * ky * kz;
        /* E(k) = -i k G rho(k):  (a + i b)(-i k) = b k - i a k */
        const double pr = gf * ar, pi = gf * ai;
        ex[idx] = kx * pi; exi[idx] = -kx * pr;
        ey[idx] = ky * pi; eyi[idx] = -ky * pr;
        ez[idx] = kz * pi; ezi[idx] = -kz * pr;
      }
    }
  }
  pppm_dft3(ex, exi, ng, +1);
  pppm_dft3(ey, eyi, ng, +1);
  pppm_dft3(ez, ezi, ng, +1);
  /* fieldforce_ik */
  for (int i = 0; i < n; i++) {
    const double *wx = wts + ((size_t)i * 3) * PPPM_ORDER, *wy = wx + PPPM_ORDER, *wz = wy + PPPM_ORDER;
    double fx = 0.0, fy = 0.0, fz = 0.0;
    for (int c = 0; c < PPPM_ORDER; c++) {
      const int gz = pmod(ibase[3 * i + 2] + c - 2, nz);
      for (int bb = 0; bb < PPPM_ORDER; bb++) {
        const int gy = pmod(ibase[3 * i + 1] + bb - 2, ny);
        const double zy = wz[c] * wy[bb];
        for (int a = 0; a < PPPM_ORDER; a++) {
          const int gx = pmod(ibase[3 * i] + a - 2, nx);
          const size_t idx = ((size_t)gz * ny + gy) * nx + gx;
          const double w = zy * wx[a];
          fx += w * ex[idx]; fy += w * ey[idx]; fz += w * ez[idx];
        }
      }
    }
    const double qf = QQRD2E * s->q[i];
    f[3 * i] += qf * fx; f[3 * i + 1] += qf * fy; f[3 * i + 2] += qf * fz;
  }
  e *= 0.5 * b->vol;
  for (int k = 0; k < 6; k++) v[k] *= 0.5 * b->vol;
  e -= g * s->qsqsum / sqrt(MY_PI) + 0.5 * MY_PI * s->qsum * s->qsum / (g * g * b->vol);
  eng[OMD_KSPACE] += QQRD2E * e;
  for (int k = 0; k < 6; k++) vir[OMD_KSPACE * 6 + k] += QQRD2E * v[k];
  free(rr); free(ri); free(wts); free(ibase); free(ex); free(exi); free(ey); free(eyi); free(ez); free(ezi);
  s->timing[1] += now_s() - t0;
}

static void ewald_setup(omd_sim *s) {
  boxq b;
  box_derive(s, &b);
  free(s->kn);
  s->kn = NULL;
  s->nk = 0;
  s->g_ewald = 0.0;
  if (s->qsqsum == 0.0) return;
  double accuracy = s->p.kspace_accuracy * QQR2E; /* two_charge_force = qqr2e in real units */
  double q2 = s->qsqsum * QQRD2E;
  double rc = s->p.cut_coul;
  double t = accuracy * sqrt((double)s->n * rc * b.h[0] * b.h[1] * b.h[2]) / (2.0 * q2);
  if (t >= 1.0)
    s->g_ewald = (1.35 - 0.15 * log(accuracy)) / rc;
  else
    s->g_ewald = sqrt(-log(t)) / rc;
  s->pg[0] = s->pg[1] = s->pg[2] = 0;
  if (s->p.kspace_pppm) {
    pppm_setup(s, &b, accuracy, q2);
    return;
  }
  double g = s->g_ewald;
  /* k-space truncation: the per-dimension RMS criterion of "kspace_style ewald" at the same
   * accuracy: err(km) = 2 q2 g / L sqrt(1/(pi km N)) exp(-pi^2 km^2 / (g^2 L^2)) */
  int kmax[3];
  double gsqmx = 0.0;
  for (int d = 0; d < 3; d++) {
    double L = b.h[d];
    int km = 1;
    for (;;) {
      double err = 2.0 * q2 * g / L * sqrt(1.0 / (MY_PI * km * s->n)) *
                   exp(-MY_PI * MY_PI * km * km / (g * g * L * L));
      if (err <= accuracy) break;
      km++;
    }
    kmax[d] = km;
    double u = 2.0 * MY_PI * km / L;
    if (u * u > gsqmx) gsqmx = u * u;
  }
  gsqmx *= 1.00001;
  /* enumerate the half space; triclinic k = 2 pi H^-T n */
  int cap = 0;
  for (int pass = 0; pass < 2; pass++) {
    int cnt = 0;
    /* allow a margin on the integer ranges so a tilted cell keeps every k inside the sphere */
    int r0 = kmax[0] + 2, r1 = kmax[1] + 2, r2 = kmax[2] + 2;
    for (int n1 = 0; n1 <= r0; n1++)
      for (int n2 = -r1; n2 <= r1; n2++)
        for (int n3 = -r2; n3 <= r2; n3++) {
          if (n1 == 0 && (n2 < 0 || (n2 == 0 && n3 <= 0))) continue;
          double kx = 2.0 * MY_PI * (b.hinv[0] * n1);
          double ky = 2.0 * MY_PI * (b.hinv[5] * n1 + b.hinv[1] * n2);
          double kz = 2.0 * MY_PI * (b.hinv[4] * n1 + b.hinv[3] * n2 + b.hinv[2] * n3);
          double sqk = kx * kx + ky * ky + kz * kz;
          if (sqk > gsqmx) continue;
          if (pass == 1) {
            s->kn[3 * cnt] = n1;
            s->kn[3 * cnt + 1] = n2;
            s->kn[3 * cnt + 2] = n3;
          }
          cnt++;
        }
    if (pass == 0) {
      cap = cnt;
      s->kn = (int *)xcalloc(3 * (size_t)cap, sizeof(int));
    } else
      s->nk = cnt;
  }
}

/* ------------------------------------------------------------------ neighbour list */
static void box_corners(const boxq *b, double c[8][3]) {
  int k = 0;
  for (int iz = 0; iz < 2; iz++)
    for (int iy = 0; iy < 2; iy++)
      for (int ix = 0; ix < 2; ix++) {
        c[k][0] = b->h[0] * ix + b->h[5] * iy + b->h[4] * iz + b->lo[0];
        c[k][1] = b->h[1] * iy + b->h[3] * iz + b->lo[1];
        c[k][2] = b->h[2] * iz + b->lo[2];
        k++;
      }
}

static int excluded(const omd_sim *s, int i, int j) {
  for (int e = s->ex_start[i]; e < s->ex_start[i + 1]; e++)
    if (s->ex_list[e] == j) return 1;
  return 0;
}

static void neigh_build(omd_sim *s) {
  double t0 = now_s();
  boxq b;
  box_derive(s, &b);
  int n = s->n;
  double rlist = (s->p.cut_lj > s->p.cut_coul ? s->p.cut_lj : s->p.cut_coul) + s->p.skin;
  double rl2 = rlist * rlist;
  /* perpendicular widths of the cell */
  double w[3];
  w[0] = 1.0 / sqrt(b.hinv[0] * b.hinv[0] + b.hinv[5] * b.hinv[5] + b.hinv[4] * b.hinv[4]);
  w[1] = 1.0 / sqrt(b.hinv[1] * b.hinv[1] + b.hinv[3] * b.hinv[3]);
  w[2] = 1.0 / fabs(b.hinv[2]);
  for (int d = 0; d < 3; d++)
    if (w[d] < 2.0 * rlist) {
      fprintf(stderr, "md_oracle: box width %g < 2*(cutoff+skin)=%g in dim %d\n", w[d], 2 * rlist, d);
      exit(1);
    }
  int nb[3];
  for (int d = 0; d < 3; d++) {
    nb[d] = (int)floor(w[d] / rlist);
    if (nb[d] < 1) nb[d] = 1;
  }
  int ncell = nb[0] * nb[1] * nb[2];
  double *lam = (double *)xcalloc(3 * (size_t)n, sizeof(double));
  int *cell = (int *)xcalloc(n, sizeof(int));
  int *cstart = (int *)xcalloc(ncell + 1, sizeof(int));
  int *corder = (int *)xcalloc(n, sizeof(int));
  for (int i = 0; i < n; i++) {
    double d0 = s->x[3 * i] - b.lo[0], d1 = s->x[3 * i + 1] - b.lo[1], d2 = s->x[3 * i + 2] - b.lo[2];
    double l[3];
    l[0] = b.hinv[0] * d0 + b.hinv[5] * d1 + b.hinv[4] * d2;
    l[1] = b.hinv[1] * d1 + b.hinv[3] * d2;
    l[2] = b.hinv[2] * d2;
    int c[3];
    for (int d = 0; d < 3; d++) {
      double fl = floor(l[d]);
      s->wrapn[3 * i + d] = (int)fl;
      l[d] -= fl;
      if (l[d] >= 1.0) l[d] = 0.0; /* guard rounding */
      lam[3 * i + d] = l[d];
      c[d] = (int)(l[d] * nb[d]);
      if (c[d] >= nb[d]) c[d] = nb[d] - 1;
    }
    cell[i] = (c[2] * nb[1] + c[1]) * nb[0] + c[0];
    cstart[cell[i] + 1]++;
  }
  for (int c = 0; c < ncell; c++) cstart[c + 1] += cstart[c];
  int *fill = (int *)xcalloc(ncell, sizeof(int));
  for (int i = 0; i < n; i++) corder[cstart[cell[i]] + fill[cell[i]]++] = i;
  free(fill);
  /* wrapped cartesian coordinates */
  double *xw = (double *)xcalloc(3 * (size_t)n, sizeof(double));
  for (int i = 0; i < n; i++) {
    const double *l = lam + 3 * i;
    xw[3 * i] = b.h[0] * l[0] + b.h[5] * l[1] + b.h[4] * l[2];
    xw[3 * i + 1] = b.h[1] * l[1] + b.h[3] * l[2];
    xw[3 * i + 2] = b.h[2] * l[2];
  }
  s->npairs = 0;
  for (int i = 0; i < n; i++) {
    int ci = cell[i];
    int c0 = ci % nb[0], c1 = (ci / nb[0]) % nb[1], c2 = ci / (nb[0] * nb[1]);
    for (int o2 = -1; o2 <= 1; o2++)
      for (int o1 = -1; o1 <= 1; o1++)
        for (int o0 = -1; o0 <= 1; o0++) {
          int a0 = c0 + o0, a1 = c1 + o1, a2 = c2 + o2;
          signed char sh[3] = {0, 0, 0};
          if (a0 < 0) { a0 += nb[0]; sh[0] = -1; } else if (a0 >= nb[0]) { a0 -= nb[0]; sh[0] = 1; }
          if (a1 < 0) { a1 += nb[1]; sh[1] = -1; } else if (a1 >= nb[1]) { a1 -= nb[1]; sh[1] = 1; }
          if (a2 < 0) { a2 += nb[2]; sh[2] = -1; } else if (a2 >= nb[2]) { a2 -= nb[2]; sh[2] = 1; }
          double sv[3];
          shift_vec(&b, sh, sv);
          int cj = (a2 * nb[1] + a1) * nb[0] + a0;
          for (int e = cstart[cj]; e < cstart[cj + 1]; e++) {
            int j = corder[e];
            if (j <= i) continue;
            /* image of j = xw_j + shift ; d = xw_i - (xw_j + shift) */
            double dx = xw[3 * i] - xw[3 * j] - sv[0];
            double dy = xw[3 * i + 1] - xw[3 * j + 1] - sv[1];
            double dz = xw[3 * i + 2] - xw[3 * j + 2] - sv[2];
            double r2 = dx * dx + dy * dy + dz * dz;
            if (r2 >= rl2) continue;
            if (excluded(s, i, j)) continue;
            if (s->npairs >= s->pair_cap) {
              s->pair_cap = s->pair_cap ? 2 * s->pair_cap : 1 << 20;
              s->pi = (int *)realloc(s->pi, s->pair_cap * sizeof(int));
              s->pj = (int *)realloc(s->pj, s->pair_cap * sizeof(int));
              s->psh = (signed char *)realloc(s->psh, 3 * (size_t)s->pair_cap);
            }
            int p = s->npairs++;
            s->pi[p] = i;
            s->pj[p] = j;
            s->psh[3 * p] = sh[0];
            s->psh[3 * p + 1] = sh[1];
            s->psh[3 * p + 2] = sh[2];
          }
        }
  }
  memcpy(s->xhold, s->x, 3 * (size_t)n * sizeof(double));
  box_corners(&b, s->corners_hold);
  s->ago = 0;
  s->nbuilds++;
  free(lam);
  free(cell);
  free(cstart);
  free(corder);
  free(xw);
  s->timing[2] += now_s() - t0;
}

/* rebuild trigger of "neigh_modify every 1 delay 5 check yes" with a deforming triclinic box:
 * an atom moved more than half of (skin - the two largest box-corner displacements) */
static int neigh_check(omd_sim *s) {
  boxq b;
  box_derive(s, &b);
  double c[8][3];
  box_corners(&b, c);
  double delta1 = 0.0, delta2 = 0.0;
  for (int k = 0; k < 8; k++) {
    double dx = c[k][0] - s->corners_hold[k][0], dy = c[k][1] - s->corners_hold[k][1],
           dz = c[k][2] - s->corners_hold[k][2];
    double d = sqrt(dx * dx + dy * dy + dz * dz);
    if (d > delta1)
      delta1 = d;
    else if (d > delta2)
      delta2 = d;
  }
  double delta = 0.5 * (s->p.skin - (delta1 + delta2));
  double deltasq = delta * delta;
  for (int i = 0; i < s->n; i++) {
    double dx = s->x[3 * i] - s->xhold[3 * i], dy = s->x[3 * i + 1] - s->xhold[3 * i + 1],
           dz = s->x[3 * i + 2] - s->xhold[3 * i + 2];
    if (dx * dx + dy * dy + dz * dz > deltasq) return 1;
  }
  return 0;
}

/* ------------------------------------------------------------------ force terms */
static void pair_compute(omd_sim *s, const boxq *b, double *f, double *eng, double *vir) {
  double t0 = now_s();
  int n = s->n, nt = s->ntypes;
  double *xw = (double *)xcalloc(3 * (size_t)n, sizeof(double));
  for (int i = 0; i < n; i++) {
    const int *w = s->wrapn + 3 * i;
    xw[3 * i] = s->x[3 * i] - (b->h[0] * w[0] + b->h[5] * w[1] + b->h[4] * w[2]);
    xw[3 * i + 1] = s->x[3 * i + 1] - (b->h[1] * w[1] + b->h[3] * w[2]);
    xw[3 * i + 2] = s->x[3 * i + 2] - (b->h[2] * w[2]);
  }
  double cl2 = s->p.cut_lj * s->p.cut_lj, cc2 = s->p.cut_coul * s->p.cut_coul;
  double g = s->g_ewald;
  double elj = 0, ecoul = 0;
  double vl[6] = {0}, vc[6] = {0};
  for (int p = 0; p < s->npairs; p++) {
    int i = s->pi[p], j = s->pj[p];
    double sv[3];
    shift_vec(b, s->psh + 3 * p, sv);
    double d[3] = {xw[3 * i] - xw[3 * j] - sv[0], xw[3 * i + 1] - xw[3 * j + 1] - sv[1],
                   xw[3 * i + 2] - xw[3 * j + 2] - sv[2]};
    double rsq = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    if (rsq >= cl2 && rsq >= cc2) continue;
    double r2inv = 1.0 / rsq;
    double flj = 0.0, fc = 0.0;
    if (rsq < cc2 && s->q[i] != 0.0 && s->q[j] != 0.0) {
      double r = sqrt(rsq);
      double grij = g * r;
      double expm2 = exp(-grij * grij);
      double erfcv = erfc(grij);
      double pref = QQRD2E * s->q[i] * s->q[j] / r;
      fc = pref * (erfcv + EWALD_F * grij * expm2) * r2inv;
      ecoul += pref * erfcv;
    }
    if (rsq < cl2) {
      int tt = s->type[i] * nt + s->type[j];
      double r6inv = r2inv * r2inv * r2inv;
      flj = r6inv * (s->lj1[tt] * r6inv - s->lj2[tt]) * r2inv;
      elj += r6inv * (s->lj3[tt] * r6inv - s->lj4[tt]);
    }
    double fl[3] = {d[0] * flj, d[1] * flj, d[2] * flj};
    double fq[3] = {d[0] * fc, d[1] * fc, d[2] * fc};
    for (int k = 0; k < 3; k++) {
      f[3 * i + k] += fl[k] + fq[k];
      f[3 * j + k] -= fl[k] + fq[k];
    }
    vtally(vl, d, fl);
    vtally(vc, d, fq);
  }
  /* special pairs (weights != 1): full real-space term with the weights applied, i.e. for
   * coul/long the k-space sum minus (1-f_coul) q_i q_j / r   */
  for (int m = 0; m < s->nspecial; m++) {
    int i = s->sp_i[m], j = s->sp_j[m];
    double d[3] = {s->x[3 * i] - s->x[3 * j], s->x[3 * i + 1] - s->x[3 * j + 1],
                   s->x[3 * i + 2] - s->x[3 * j + 2]};
    minimg(b, d);
    double rsq = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    double r2inv = 1.0 / rsq;
    double flj = 0.0, fc = 0.0;
    if (rsq < cc2 && g > 0.0) {
      double r = sqrt(rsq);
      double grij = g * r;
      double expm2 = exp(-grij * grij);
      double pref = QQRD2E * s->q[i] * s->q[j] / r;
      /* erfc(x) - (1 - f) = f - erf(x) */
      double e = s->sp_fc[m] - erf(grij);
      fc = pref * (e + EWALD_F * grij * expm2) * r2inv;
      ecoul += pref * e;
    }
    if (rsq < cl2 && s->sp_flj[m] != 0.0) {
      int tt = s->type[i] * nt + s->type[j];
      double r6inv = r2inv * r2inv * r2inv;
      flj = s->sp_flj[m] * r6inv * (s->lj1[tt] * r6inv - s->lj2[tt]) * r2inv;
      elj += s->sp_flj[m] * r6inv * (s->lj3[tt] * r6inv - s->lj4[tt]);
    }
    double fl[3] = {d[0] * flj, d[1] * flj, d[2] * flj};
    double fq[3] = {d[0] * fc, d[1] * fc, d[2] * fc};
    for (int k = 0; k < 3; k++) {
      f[3 * i + k] += fl[k] + fq[k];
      f[3 * j + k] -= fl[k] + fq[k];
    }
    vtally(vl, d, fl);
    vtally(vc, d, fq);
  }
  eng[OMD_LJ] += elj;
  eng[OMD_COUL] += ecoul;
  for (int k = 0; k < 6; k++) {
    vir[OMD_LJ * 6 + k] += vl[k];
    vir[OMD_COUL * 6 + k] += vc[k];
  }
  free(xw);
  s->timing[0] += now_s() - t0;
}

static void bond_compute(omd_sim *s, const boxq *b, double *f, double *eng, double *vir) {
  double e = 0, v[6] = {0};
  for (int m = 0; m < s->nbonds; m++) {
    if (s->use_shake && s->bond_shaken[m]) continue; /* fix shake turns these bonds off */
    int i1 = s->bond[2 * m], i2 = s->bond[2 * m + 1];
    double K = s->bond_coeff[2 * s->bond_type[m]], r0 = s->bond_coeff[2 * s->bond_type[m] + 1];
    double d[3] = {s->x[3 * i1] - s->x[3 * i2], s->x[3 * i1 + 1] - s->x[3 * i2 + 1],
                   s->x[3 * i1 + 2] - s->x[3 * i2 + 2]};
    minimg(b, d);
    double r = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    double dr = r - r0, rk = K * dr;
    double fb = (r > 0.0) ? -2.0 * rk / r : 0.0;
    e += rk * dr;
    double ff[3] = {d[0] * fb, d[1] * fb, d[2] * fb};
    for (int k = 0; k < 3; k++) {
      f[3 * i1 + k] += ff[k];
      f[3 * i2 + k] -= ff[k];
    }
    vtally(v, d, ff);
  }
  eng[OMD_BOND] += e;
  for (int k = 0; k < 6; k++) vir[OMD_BOND * 6 + k] += v[k];
}

static void angle_compute(omd_sim *s, const boxq *b, double *f, double *eng, double *vir) {
  double e = 0, v[6] = {0};
  for (int m = 0; m < s->nangles; m++) {
    int i1 = s->angle[3 * m], i2 = s->angle[3 * m + 1], i3 = s->angle[3 * m + 2];
    double K = s->angle_coeff[2 * s->angle_type[m]], th0 = s->angle_coeff[2 * s->angle_type[m] + 1];
    double d1[3], d2[3];
    for (int k = 0; k < 3; k++) {
      d1[k] = s->x[3 * i1 + k] - s->x[3 * i2 + k];
      d2[k] = s->x[3 * i3 + k] - s->x[3 * i2 + k];
    }
    minimg(b, d1);
    minimg(b, d2);
    double rsq1 = d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2];
    double rsq2 = d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2];
    double r1 = sqrt(rsq1), r2 = sqrt(rsq2);
    double c = (d1[0] * d2[0] + d1[1] * d2[1] + d1[2] * d2[2]) / (r1 * r2);
    if (c > 1.0) c = 1.0;
    if (c < -1.0) c = -1.0;
    double sn = sqrt(1.0 - c * c);
    if (sn < 0.001) sn = 0.001;
    double dth = acos(c) - th0;
    double tk = K * dth;
    e += tk * dth;
    /* dE/dc = -2 K dth / sin(theta) */
    double a = -2.0 * tk / sn;
    double a11 = a * c / rsq1, a12 = -a / (r1 * r2), a22 = a * c / rsq2;
    double f1[3], f3[3];
    for (int k = 0; k < 3; k++) {
      f1[k] = a11 * d1[k] + a12 * d2[k];
      f3[k] = a22 * d2[k] + a12 * d1[k];
      f[3 * i1 + k] += f1[k];
      f[3 * i2 + k] -= f1[k] + f3[k];
      f[3 * i3 + k] += f3[k];
    }
    vtally(v, d1, f1);
    vtally(v, d2, f3);
  }
  eng[OMD_ANGLE] += e;
  for (int k = 0; k < 6; k++) vir[OMD_ANGLE * 6 + k] += v[k];
}

static inline void cross(const double a[3], const double b[3], double c[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
static inline double dot3(const double a[3], const double b[3]) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

/* shared torsion geometry: cos(phi) of the IUPAC dihedral 1-2-3-4 and d cos(phi)/d r_k.
 * F=r1-r2, G=r2-r3, H=r4-r3, A=FxG, B=HxG, c=A.B/(|A||B|) */
static double torsion_cos(const boxq *b, const double *x, const int at[4], double F[3], double G[3],
                          double H[3], double dc[4][3]) {
  for (int k = 0; k < 3; k++) {
    F[k] = x[3 * at[0] + k] - x[3 * at[1] + k];
    G[k] = x[3 * at[1] + k] - x[3 * at[2] + k];
    H[k] = x[3 * at[3] + k] - x[3 * at[2] + k];
  }
  minimg(b, F);
  minimg(b, G);
  minimg(b, H);
  double A[3], B[3];
  cross(F, G, A);
  cross(H, G, B);
  double a2 = dot3(A, A), b2 = dot3(B, B);
  double ia = 1.0 / sqrt(a2), ib = 1.0 / sqrt(b2);
  double c = dot3(A, B) * ia * ib;
  if (c > 1.0) c = 1.0;
  if (c < -1.0) c = -1.0;
  double gA[3], gB[3];
  for (int k = 0; k < 3; k++) {
    gA[k] = B[k] * ia * ib - c * A[k] / a2;
    gB[k] = A[k] * ia * ib - c * B[k] / b2;
  }
  double GxgA[3], GxgB[3], gAxF[3], gBxH[3];
  cross(G, gA, GxgA);
  cross(G, gB, GxgB);
  cross(gA, F, gAxF);
  cross(gB, H, gBxH);
  for (int k = 0; k < 3; k++) {
    dc[0][k] = GxgA[k];
    dc[3][k] = GxgB[k];
    dc[1][k] = -GxgA[k] + gAxF[k] + gBxH[k];
    dc[2][k] = -(gAxF[k] + gBxH[k]) - GxgB[k];
  }
  return c;
}

static void torsion_apply(const int at[4], const double F[3], const double G[3], const double H[3],
                          double dc[4][3], double dEdc, double *f, double *v) {
  double ff[4][3];
  for (int a = 0; a < 4; a++)
    for (int k = 0; k < 3; k++) {
      ff[a][k] = -dEdc * dc[a][k];
      f[3 * at[a] + k] += ff[a][k];
    }
  /* virial relative to atom 3: r1-r3 = F+G, r2-r3 = G, r4-r3 = H */
  double FG[3] = {F[0] + G[0], F[1] + G[1], F[2] + G[2]};
  vtally(v, FG, ff[0]);
  vtally(v, G, ff[1]);
  vtally(v, H, ff[3]);
}

static void dihedral_compute(omd_sim *s, const boxq *b, double *f, double *eng, double *vir) {
  double e = 0, v[6] = {0};
  for (int m = 0; m < s->ndihedrals; m++) {
    const int *at = s->dihedral + 4 * m;
    const double *K = s->dihedral_coeff + 4 * s->dihedral_type[m];
    double F[3], G[3], H[3], dc[4][3];
    double c = torsion_cos(b, s->x, at, F, G, H, dc);
    /* E = K1/2 (1+cos p) + K2/2 (1-cos 2p) + K3/2 (1+cos 3p) + K4/2 (1-cos 4p), in cos p */
    double c2 = c * c;
    double cos2 = 2.0 * c2 - 1.0, cos3 = (4.0 * c2 - 3.0) * c, cos4 = 8.0 * c2 * c2 - 8.0 * c2 + 1.0;
    e += 0.5 * (K[0] * (1.0 + c) + K[1] * (1.0 - cos2) + K[2] * (1.0 + cos3) + K[3] * (1.0 - cos4));
    double dEdc = 0.5 * (K[0] - K[1] * 4.0 * c + K[2] * (12.0 * c2 - 3.0) - K[3] * (32.0 * c2 * c - 16.0 * c));
    torsion_apply(at, F, G, H, dc, dEdc, f, v);
  }
  eng[OMD_DIHEDRAL] += e;
  for (int k = 0; k < 6; k++) vir[OMD_DIHEDRAL * 6 + k] += v[k];
}

static void improper_compute(omd_sim *s, const boxq *b, double *f, double *eng, double *vir) {
  double e = 0, v[6] = {0};
  for (int m = 0; m < s->nimpropers; m++) {
    const int *at = s->improper + 4 * m;
    double K = s->improper_coeff[2 * s->improper_type[m]], chi0 = s->improper_coeff[2 * s->improper_type[m] + 1];
    double F[3], G[3], H[3], dc[4][3];
    double c = torsion_cos(b, s->x, at, F, G, H, dc);
    double sn = sqrt(1.0 - c * c);
    if (sn < 0.001) sn = 0.001;
    double dchi = acos(c) - chi0;
    e += K * dchi * dchi;
    double dEdc = -2.0 * K * dchi / sn;
    torsion_apply(at, F, G, H, dc, dEdc, f, v);
  }
  eng[OMD_IMPROPER] += e;
  for (int k = 0; k < 6; k++) vir[OMD_IMPROPER * 6 + k] += v[k];
}

static void ewald_compute(omd_sim *s, const boxq *b, double *f, double *eng, double *vir) {
  if (s->p.kspace_pppm) {
    pppm_compute(s, b, f, eng, vir);
    return;
  }
  if (s->nk == 0 || s->g_ewald == 0.0) return;
  double t0 = now_s();
  int n = s->n;
  double g = s->g_ewald, g2inv = 1.0 / (g * g);
  int km[3] = {0, 0, 0};
  for (int k = 0; k < s->nk; k++)
    for (int d = 0; d < 3; d++) {
      int a = abs(s->kn[3 * k + d]);
      if (a > km[d]) km[d] = a;
    }
  int stride = (km[0] > km[1] ? km[0] : km[1]);
  if (km[2] > stride) stride = km[2];
  stride += 1;
  /* per-atom cos/sin(m * 2 pi lamda_d), m = 0..km[d], by the angle-addition recurrence */
  size_t tab = (size_t)3 * stride * n;
  double *cs = (double *)xcalloc(tab, sizeof(double));
  double *sn = (double *)xcalloc(tab, sizeof(double));
#define CS(d, m, i) cs[((size_t)(d)*stride + (m)) * n + (i)]
#define SN(d, m, i) sn[((size_t)(d)*stride + (m)) * n + (i)]
  for (int i = 0; i < n; i++) {
    double d0 = s->x[3 * i] - b->lo[0], d1 = s->x[3 * i + 1] - b->lo[1], d2 = s->x[3 * i + 2] - b->lo[2];
    double l[3];
    l[0] = b->hinv[0] * d0 + b->hinv[5] * d1 + b->hinv[4] * d2;
    l[1] = b->hinv[1] * d1 + b->hinv[3] * d2;
    l[2] = b->hinv[2] * d2;
    for (int d = 0; d < 3; d++) {
      double th = 2.0 * MY_PI * (l[d] - floor(l[d]));
      double c1 = cos(th), s1 = sin(th);
      CS(d, 0, i) = 1.0;
      SN(d, 0, i) = 0.0;
      for (int m = 1; m <= km[d]; m++) {
        CS(d, m, i) = CS(d, m - 1, i) * c1 - SN(d, m - 1, i) * s1;
        SN(d, m, i) = SN(d, m - 1, i) * c1 + CS(d, m - 1, i) * s1;
      }
    }
  }
  double e = 0, v[6] = {0};
  double preu = 4.0 * MY_PI / b->vol;
  double *cr = (double *)xcalloc(n, sizeof(double));
  double *ci = (double *)xcalloc(n, sizeof(double));
  for (int k = 0; k < s->nk; k++) {
    int n1 = s->kn[3 * k], n2 = s->kn[3 * k + 1], n3 = s->kn[3 * k + 2];
    double kx = 2.0 * MY_PI * (b->hinv[0] * n1);
    double ky = 2.0 * MY_PI * (b->hinv[5] * n1 + b->hinv[1] * n2);
    double kz = 2.0 * MY_PI * (b->hinv[4] * n1 + b->hinv[3] * n2 + b->hinv[2] * n3);
    double sqk = kx * kx + ky * ky + kz * kz;
    double ug = preu * exp(-0.25 * sqk * g2inv) / sqk;
    int a1 = abs(n1), a2 = abs(n2), a3 = abs(n3);
    double s2 = n2 < 0 ? -1.0 : 1.0, s3 = n3 < 0 ? -1.0 : 1.0;
    double Sr = 0, Si = 0;
    for (int i = 0; i < n; i++) {
      /* exp(i(n1 t1 + n2 t2 + n3 t3)) */
      double c1 = CS(0, a1, i), s1 = SN(0, a1, i); /* n1 >= 0 */
      double c2 = CS(1, a2, i), sn2 = s2 * SN(1, a2, i);
      double c3 = CS(2, a3, i), sn3 = s3 * SN(2, a3, i);
      double c12 = c1 * c2 - s1 * sn2, s12 = s1 * c2 + c1 * sn2;
      double cc = c12 * c3 - s12 * sn3, ss = s12 * c3 + c12 * sn3;
      cr[i] = cc;
      ci[i] = ss;
      Sr += s->q[i] * cc;
      Si += s->q[i] * ss;
    }
    double uk = ug * (Sr * Sr + Si * Si);
    e += uk;
    double vterm = -2.0 * (1.0 / sqk + 0.25 * g2inv);
    v[0] += uk * (1.0 + vterm * kx * kx);
    v[1] += uk * (1.0 + vterm * ky * ky);
    v[2] += uk * (1.0 + vterm * kz * kz);
    v[3] += uk * vterm * kx * ky;
    v[4] += uk * vterm * kx * kz;
    v[5] += uk * vterm * ky * kz;
    /* F_i = -dE/dr_i = 2 ug q_i k (sin_i Sr - cos_i Si)  (x QQRD2E) */
    for (int i = 0; i < n; i++) {
      double pf = QQRD2E * 2.0 * ug * s->q[i] * (ci[i] * Sr - cr[i] * Si);
      f[3 * i] += pf * kx;
      f[3 * i + 1] += pf * ky;
      f[3 * i + 2] += pf * kz;
    }
  }
  /* self energy and neutralising background */
  e -= g * s->qsqsum / sqrt(MY_PI) + 0.5 * MY_PI * s->qsum * s->qsum / (g * g * b->vol);
  eng[OMD_KSPACE] += QQRD2E * e;
  for (int k = 0; k < 6; k++) vir[OMD_KSPACE * 6 + k] += QQRD2E * v[k];
  free(cs);
  free(sn);
  free(cr);
  free(ci);
#undef CS
#undef SN
  s->timing[1] += now_s() - t0;
}

static void force_compute(omd_sim *s) {
  boxq b;
  box_derive(s, &b);
  memset(s->f, 0, 3 * (size_t)s->n * sizeof(double));
  memset(s->eng, 0, sizeof(s->eng));
  memset(s->vir, 0, sizeof(s->vir));
  if (s->ext_fn) {
    const double box[9] = {s->lo[0], s->lo[1], s->lo[2], s->hi[0], s->hi[1], s->hi[2], s->xy, s->xz, s->yz};
    s->ext_fn(s->ext_ctx, s->ext_calls++, s->n, box, s->x, s->f, &s->vir[OMD_LJ * 6], &s->eng[OMD_LJ]);
    return;
  }
  pair_compute(s, &b, s->f, s->eng, s->vir);
  bond_compute(s, &b, s->f, s->eng, s->vir);
  angle_compute(s, &b, s->f, s->eng, s->vir);
  dihedral_compute(s, &b, s->f, s->eng, s->vir);
  improper_compute(s, &b, s->f, s->eng, s->vir);
  ewald_compute(s, &b, s->f, s->eng, s->vir);
}

void omd_freeze_kspace(omd_sim *s, int frozen) { s->kspace_frozen = frozen; }

void omd_set_external_force(omd_sim *s, omd_force_fn fn, void *ctx) {
  s->ext_fn = fn;
  s->ext_ctx = ctx;
  s->ext_calls = 0;
}

void omd_setup(omd_sim *s, int use_shake) {
  s->use_shake = use_shake && s->nclus > 0;
  s->tdof = 3.0 * s->n - 3.0 - (s->use_shake ? s->ncons : 0);
  s->ext_calls = 0;
  if (s->ext_fn) return;   /* no lists, no k-space of this file */
  if (!s->kspace_frozen) ewald_setup(s);
  neigh_build(s);
}

void omd_compute(omd_sim *s, double *f, double *energies, double *virials) {
  force_compute(s);
  if (f) memcpy(f, s->f, 3 * (size_t)s->n * sizeof(double));
  if (energies) memcpy(energies, s->eng, sizeof(s->eng));
  if (virials) memcpy(virials, s->vir, sizeof(s->vir));
}

double omd_temperature(const omd_sim *s, double ke[6]) {
  double t[6] = {0};
  for (int i = 0; i < s->n; i++) {
    double m = s->mass[s->type[i]];
    const double *v = s->v + 3 * i;
    t[0] += m * v[0] * v[0];
    t[1] += m * v[1] * v[1];
    t[2] += m * v[2] * v[2];
    t[3] += m * v[0] * v[1];
    t[4] += m * v[0] * v[2];
    t[5] += m * v[1] * v[2];
  }
  for (int k = 0; k < 6; k++) t[k] *= MVV2E;
  if (ke) memcpy(ke, t, sizeof(t));
  return (t[0] + t[1] + t[2]) / (s->tdof * BOLTZ);
}

/* ------------------------------------------------------------------ SHAKE */
/* Star cluster (central atom 0, nb satellites): find lambda_k with
 *   | s_0k + sum_j M_kj lambda_j r_0j |^2 = d_k^2 ,  M_kj = 1/m0 + delta_kj / m_k
 * where r = current separations, s = separations after the unconstrained update
 * x + dt v + dtfsq f/m.  nb==1 is solved in closed form, nb>1 by the fixed-point iteration on
 * the linearised system (tolerance on lambda, at most maxiter sweeps).  The constraint force
 * lambda_k/dtfsq * r_0k is added to f and tallied in the virial. */
static void shake_apply(omd_sim *s, const boxq *b, double dtv, double dtfsq, double *vir) {
  double v[6] = {0};
  for (int cl = 0; cl < s->nclus; cl++) {
    int na = s->clus_n[cl], nb = na - 1;
    const int *at = s->clus_atom + 4 * cl;
    const double *dist = s->clus_d + 3 * cl;
    double invm[4], xs[4][3];
    for (int a = 0; a < na; a++) {
      int i = at[a];
      invm[a] = 1.0 / s->mass[s->type[i]];
      for (int k = 0; k < 3; k++)
        xs[a][k] = s->x[3 * i + k] + dtv * s->v[3 * i + k] + dtfsq * invm[a] * s->f[3 * i + k];
    }
    double r[3][3], sv[3][3];
    for (int k = 0; k < nb; k++) {
      for (int c = 0; c < 3; c++) {
        r[k][c] = s->x[3 * at[0] + c] - s->x[3 * at[k + 1] + c];
        sv[k][c] = xs[0][c] - xs[k + 1][c];
      }
      minimg(b, r[k]);
      minimg(b, sv[k]);
    }
    double lam[3] = {0, 0, 0};
    if (nb == 1) {
      double m01 = invm[0] + invm[1];
      double r01sq = dot3(r[0], r[0]), s01sq = dot3(sv[0], sv[0]);
      double a = m01 * m01 * r01sq;
      double bb = 2.0 * m01 * dot3(sv[0], r[0]);
      double c = s01sq - dist[0] * dist[0];
      double determ = bb * bb - 4.0 * a * c;
      if (determ < 0.0) determ = 0.0;
      double l1 = (-bb + sqrt(determ)) / (2.0 * a), l2 = (-bb - sqrt(determ)) / (2.0 * a);
      lam[0] = (fabs(l1) <= fabs(l2)) ? l1 : l2;
    } else {
      double A[3][3] = {{0}}, Ainv[3][3] = {{0}}, M[3][3] = {{0}};
      for (int k = 0; k < nb; k++)
        for (int j = 0; j < nb; j++) {
          M[k][j] = invm[0] + (k == j ? invm[k + 1] : 0.0);
          A[k][j] = 2.0 * M[k][j] * dot3(sv[k], r[j]);
        }
      if (nb == 2) {
        double det = A[0][0] * A[1][1] - A[0][1] * A[1][0];
        Ainv[0][0] = A[1][1] / det;
        Ainv[0][1] = -A[0][1] / det;
        Ainv[1][0] = -A[1][0] / det;
        Ainv[1][1] = A[0][0] / det;
      } else {
        double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) -
                     A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                     A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
        double id = 1.0 / det;
        Ainv[0][0] = id * (A[1][1] * A[2][2] - A[1][2] * A[2][1]);
        Ainv[0][1] = -id * (A[0][1] * A[2][2] - A[0][2] * A[2][1]);
        Ainv[0][2] = id * (A[0][1] * A[1][2] - A[0][2] * A[1][1]);
        Ainv[1][0] = -id * (A[1][0] * A[2][2] - A[1][2] * A[2][0]);
        Ainv[1][1] = id * (A[0][0] * A[2][2] - A[0][2] * A[2][0]);
        Ainv[1][2] = -id * (A[0][0] * A[1][2] - A[0][2] * A[1][0]);
        Ainv[2][0] = id * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
        Ainv[2][1] = -id * (A[0][0] * A[2][1] - A[0][1] * A[2][0]);
        Ainv[2][2] = id * (A[0][0] * A[1][1] - A[0][1] * A[1][0]);
      }
      double ssq[3];
      for (int k = 0; k < nb; k++) ssq[k] = dot3(sv[k], sv[k]);
      int done = 0, iter = 0;
      while (!done && iter < s->p.shake_maxiter) {
        double rhs[3];
        for (int k = 0; k < nb; k++) {
          double w[3] = {0, 0, 0};
          for (int j = 0; j < nb; j++)
            for (int c = 0; c < 3; c++) w[c] += M[k][j] * lam[j] * r[j][c];
          rhs[k] = dist[k] * dist[k] - ssq[k] - dot3(w, w);
        }
        double ln[3];
        done = 1;
        for (int k = 0; k < nb; k++) {
          ln[k] = 0.0;
          for (int j = 0; j < nb; j++) ln[k] += Ainv[k][j] * rhs[j];
          if (fabs(ln[k] - lam[k]) > s->p.shake_tol) done = 0;
        }
        for (int k = 0; k < nb; k++) lam[k] = ln[k];
        for (int k = 0; k < nb; k++)
          if (isnan(lam[k])) done = 1;
        iter++;
      }
    }
    for (int k = 0; k < nb; k++) {
      double l = lam[k] / dtfsq;
      double ff[3] = {l * r[k][0], l * r[k][1], l * r[k][2]};
      for (int c = 0; c < 3; c++) {
        s->f[3 * at[0] + c] += ff[c];
        s->f[3 * at[k + 1] + c] -= ff[c];
      }
      vtally(v, r[k], ff);
    }
  }
  if (vir)
    for (int k = 0; k < 6; k++) vir[OMD_SHAKE * 6 + k] += v[k];
}

/* ------------------------------------------------------------------ Nose-Hoover chain */
/* fix nvt: Nose-Hoover chain half-step (Martyna-Tuckerman-Klein splitting, one sub-cycle,
 * no drag).  Scales the velocities, updates t_current. */
static void nhc_temp_integrate(omd_sim *s, double dt, double t_target) {
  int mt = s->p.t_chain;
  double t_freq = 1.0 / s->p.t_period;
  double dthalf = 0.5 * dt, dt4 = 0.25 * dt, dt8 = 0.125 * dt;
  double ke_target = s->tdof * BOLTZ * t_target;
  double kecurrent = s->tdof * BOLTZ * s->t_current;
  s->eta_mass[0] = s->tdof * BOLTZ * t_target / (t_freq * t_freq);
  for (int k = 1; k < mt; k++) s->eta_mass[k] = BOLTZ * t_target / (t_freq * t_freq);
  if (s->eta_mass[0] > 0.0)
    s->eta_dotdot[0] = (kecurrent - ke_target) / s->eta_mass[0];
  else
    s->eta_dotdot[0] = 0.0;
  double expfac;
  for (int k = mt - 1; k > 0; k--) {
    expfac = exp(-dt8 * s->eta_dot[k + 1]);
    s->eta_dot[k] *= expfac;
    s->eta_dot[k] += s->eta_dotdot[k] * dt4;
    s->eta_dot[k] *= expfac;
  }
  expfac = exp(-dt8 * s->eta_dot[1]);
  s->eta_dot[0] *= expfac;
  s->eta_dot[0] += s->eta_dotdot[0] * dt4;
  s->eta_dot[0] *= expfac;
  double factor = exp(-dthalf * s->eta_dot[0]);
  for (int i = 0; i < 3 * s->n; i++) s->v[i] *= factor;
  s->t_current *= factor * factor;
  kecurrent = s->tdof * BOLTZ * s->t_current;
  if (s->eta_mass[0] > 0.0)
    s->eta_dotdot[0] = (kecurrent - ke_target) / s->eta_mass[0];
  else
    s->eta_dotdot[0] = 0.0;
  for (int k = 0; k < mt; k++) s->eta[k] += dthalf * s->eta_dot[k];
  s->eta_dot[0] *= expfac;
  s->eta_dot[0] += s->eta_dotdot[0] * dt4;
  s->eta_dot[0] *= expfac;
  for (int k = 1; k < mt; k++) {
    expfac = exp(-dt8 * s->eta_dot[k + 1]);
    s->eta_dot[k] *= expfac;
    s->eta_dotdot[k] = (s->eta_mass[k - 1] * s->eta_dot[k - 1] * s->eta_dot[k - 1] - BOLTZ * t_target) / s->eta_mass[k];
    s->eta_dot[k] += s->eta_dotdot[k] * dt4;
    s->eta_dot[k] *= expfac;
  }
}

static double nh_energy(const omd_sim *s, double t_target) {
  int mt = s->p.t_chain;
  double e = s->tdof * BOLTZ * t_target * s->eta[0] + 0.5 * s->eta_mass[0] * s->eta_dot[0] * s->eta_dot[0];
  for (int k = 1; k < mt; k++)
    e += BOLTZ * t_target * s->eta[k] + 0.5 * s->eta_mass[k] * s->eta_dot[k] * s->eta_dot[k];
  return e;
}

/* ------------------------------------------------------------------ fix deform tilt rules */
/* tilt[] = raw targets xy, xz, yz; moved by whole box lengths (new x length for xy and xz, new y length for yz) to the
 * value closest to the current tilt RATIO (current tilt / current length), with LAMMPS' loop arithmetic */
void omd_tilt_closest(double tilt[3], double xprd_new, double yprd_new, double xy, double xz, double yz, double xprd, double yprd) {
  const double denom[3] = {xprd_new, xprd_new, yprd_new};
  const double current[3] = {xy / xprd, xz / xprd, yz / yprd};
  for (int i = 0; i < 3; i++) {
    while (tilt[i] / denom[i] - current[i] > 0.0) tilt[i] -= denom[i];
    while (tilt[i] / denom[i] - current[i] < 0.0) tilt[i] += denom[i];
    if (fabs(tilt[i] / denom[i] - 1.0 - current[i]) < fabs(tilt[i] / denom[i] - current[i])) tilt[i] -= denom[i];
  }
}
/* flip rule: returns 1 if a tilt (xy, xz, yz) exceeds half its box length; flipped[] = the tilts after the flip,
 * nflip[] = lattice steps f_xy, f_xz, f_yz (a2' = a2 + f_xy a1, a3' = a3 + f_yz a2 + f_xz a1) */
int omd_tilt_flip(const double tilt[3], double xprd, double yprd, double flipped[3], int nflip[3]) {
  const double xprdinv = 1.0 / xprd, yprdinv = 1.0 / yprd;
  flipped[0] = tilt[0]; flipped[1] = tilt[1]; flipped[2] = tilt[2];
  nflip[0] = nflip[1] = nflip[2] = 0;
  if (!(tilt[2] * yprdinv < -0.5 || tilt[2] * yprdinv > 0.5 || tilt[1] * xprdinv < -0.5 || tilt[1] * xprdinv > 0.5 ||
        tilt[0] * xprdinv < -0.5 || tilt[0] * xprdinv > 0.5))
    return 0;
  if (flipped[2] * yprdinv < -0.5) {
    flipped[2] += yprd;
    flipped[1] += flipped[0];
    nflip[2] = 1;
  } else if (flipped[2] * yprdinv > 0.5) {
    flipped[2] -= yprd;
    flipped[1] -= flipped[0];
    nflip[2] = -1;
  }
  if (flipped[1] * xprdinv < -0.5) { flipped[1] += xprd; nflip[1] = 1; }
  if (flipped[1] * xprdinv > 0.5) { flipped[1] -= xprd; nflip[1] = -1; }
  if (flipped[0] * xprdinv < -0.5) { flipped[0] += xprd; nflip[0] = 1; }
  if (flipped[0] * xprdinv > 0.5) { flipped[0] -= xprd; nflip[0] = -1; }
  return (nflip[0] || nflip[1] || nflip[2]) ? 1 : 0;
}

/* ------------------------------------------------------------------ run */
static void pressure_tensor(const omd_sim *s, double p[6]) {
  boxq b;
  box_derive(s, &b);
  double ke[6];
  omd_temperature(s, ke);
  for (int k = 0; k < 6; k++) {
    double w = 0.0;
    for (int part = 0; part < OMD_NPART; part++) w += s->vir[part * 6 + k];
    p[k] = (ke[k] + w) / b.vol * NKTV2P;
  }
}

int omd_run(omd_sim *s, int nsteps, double dt, double temperature, int nvt, int use_shake,
            const double *rates, double *press_avg, double *trace) {
  int n = s->n;
  double dtv = dt, dtf = 0.5 * dt * FTM2V;
  boxq b;
  /* ---- setup (step 0), as a fresh LAMMPS "run" ---- */
  omd_setup(s, use_shake);
  force_compute(s);
  box_derive(s, &b);
  if (s->use_shake) shake_apply(s, &b, dtv, 0.5 * dt * dt * FTM2V, s->vir);
  int mt = s->p.t_chain;
  if (nvt) {
    for (int k = 0; k <= MAXCHAIN; k++) s->eta[k] = s->eta_dot[k] = s->eta_dotdot[k] = 0.0;
    s->t_current = omd_temperature(s, NULL);
    double t_freq = 1.0 / s->p.t_period;
    s->eta_mass[0] = s->tdof * BOLTZ * temperature / (t_freq * t_freq);
    for (int k = 1; k < mt; k++) s->eta_mass[k] = BOLTZ * temperature / (t_freq * t_freq);
    for (int k = 1; k < mt; k++)
      s->eta_dotdot[k] = (s->eta_mass[k - 1] * s->eta_dot[k - 1] * s->eta_dot[k - 1] - BOLTZ * temperature) / s->eta_mass[k];
  }
  /* fix deform reference box */
  double lo0[3], hi0[3], xy0 = s->xy, xz0 = s->xz, yz0 = s->yz;
  for (int d = 0; d < 3; d++) {
    lo0[d] = s->lo[d];
    hi0[d] = s->hi[d];
  }
  int flip_pending = 0, flip_n[3] = {0, 0, 0};
  double flip_tilt[3] = {0, 0, 0};
  /* fix ave/time 1 nav nav ... ave running (in.homogenization.lammps:57-59) */
  int nav = 0, nwin = 0;
  double psum[6] = {0};
  if (press_avg) {
    nav = (nsteps > 10000) ? nsteps / 1000 : nsteps / 10;
    if (nav < 1) nav = 1;
    nwin = nsteps / nav;
  }
  for (int step = 1; step <= nsteps; step++) {
    /* initial_integrate */
    if (nvt) nhc_temp_integrate(s, dt, temperature);
    for (int i = 0; i < n; i++) {
      double dtfm = dtf / s->mass[s->type[i]];
      for (int k = 0; k < 3; k++) {
        s->v[3 * i + k] += dtfm * s->f[3 * i + k];
        s->x[3 * i + k] += dtv * s->v[3 * i + k];
      }
    }
    /* neighbour decision; a pending box flip is applied first and forces the rebuild (fix deform next_reneighbor) */
    s->ago++;
    if (flip_pending) {
      s->xy = flip_tilt[0];
      s->xz = flip_tilt[1];
      s->yz = flip_tilt[2];
      /* the same reciprocal vectors in the new basis: n2 += f_xy n1, n3 += f_yz n2 + f_xz n1 (a2' = a2 + f_xy a1,
       * a3' = a3 + f_yz a2 + f_xz a1); the k-vector SET of the run does not change */
      for (int k = 0; k < s->nk; k++) {
        int n1 = s->kn[3 * k], n2 = s->kn[3 * k + 1], n3 = s->kn[3 * k + 2];
        s->kn[3 * k + 1] = n2 + flip_n[0] * n1;
        s->kn[3 * k + 2] = n3 + flip_n[2] * n2 + flip_n[1] * n1;
      }
      flip_pending = 0;
      s->nflips++;
      if (!s->ext_fn) neigh_build(s);
    } else if (!s->ext_fn && s->ago >= s->p.neigh_delay && neigh_check(s))
      neigh_build(s);
    /* forces + SHAKE */
    force_compute(s);
    box_derive(s, &b);
    if (s->use_shake) shake_apply(s, &b, dtv, dt * dt * FTM2V, s->vir);
    /* final_integrate */
    for (int i = 0; i < n; i++) {
      double dtfm = dtf / s->mass[s->type[i]];
      for (int k = 0; k < 3; k++) s->v[3 * i + k] += dtfm * s->f[3 * i + k];
    }
    if (nvt) {
      s->t_current = omd_temperature(s, NULL);
      nhc_temp_integrate(s, dt, temperature);
    }
    /* end_of_step: pressure sample (the reference never combines it with fix deform in one run;
     * when a test does, the sample uses the box the forces were computed in) */
    if (press_avg && step <= nwin * nav) {
      double p[6];
      pressure_tensor(s, p);
      for (int k = 0; k < 6; k++) psum[k] += p[k];
    }
    /* end_of_step: fix deform (erate, remap x), default "flip yes" [LAMMPS 17Nov16 fix_deform.cpp end_of_step /
     * pre_exchange, restated from its documented behaviour: SURVEY.md A.5].  Box lengths and raw tilt targets are linear
     * in t from the box at the start of the run; a tilt target is then moved by whole box lengths to the value closest to
     * the current tilt (so that it continues from a flipped box); the box takes the targets and the atoms are remapped
     * affinely.  If a target then exceeds half a box length, the flipped tilts are recorded and applied at the start of
     * the next step's reneighbouring (forced), where only the box representation changes: the lattice is the same. */
    if (rates) {
      double t = step * dt;
      double nlo[3], nhi[3];
      for (int d = 0; d < 3; d++) {
        double L0 = hi0[d] - lo0[d];
        nlo[d] = lo0[d] - 0.5 * L0 * rates[d] * t;
        nhi[d] = hi0[d] + 0.5 * L0 * rates[d] * t;
      }
      double tilt[3]; /* xy, xz, yz */
      tilt[0] = xy0 + rates[3] * (hi0[1] - lo0[1]) * t;
      tilt[1] = xz0 + rates[4] * (hi0[2] - lo0[2]) * t;
      tilt[2] = yz0 + rates[5] * (hi0[2] - lo0[2]) * t;
      boxq bo;
      box_derive(s, &bo);
      omd_tilt_closest(tilt, nhi[0] - nlo[0], nhi[1] - nlo[1], s->xy, s->xz, s->yz, s->hi[0] - s->lo[0], s->hi[1] - s->lo[1]);
      for (int d = 0; d < 3; d++) {
        s->lo[d] = nlo[d];
        s->hi[d] = nhi[d];
      }
      s->xy = tilt[0];
      s->xz = tilt[1];
      s->yz = tilt[2];
      boxq bn;
      box_derive(s, &bn);
      for (int i = 0; i < n; i++) {
        double d0 = s->x[3 * i] - bo.lo[0], d1 = s->x[3 * i + 1] - bo.lo[1], d2 = s->x[3 * i + 2] - bo.lo[2];
        double l0 = bo.hinv[0] * d0 + bo.hinv[5] * d1 + bo.hinv[4] * d2;
        double l1 = bo.hinv[1] * d1 + bo.hinv[3] * d2;
        double l2 = bo.hinv[2] * d2;
        s->x[3 * i] = bn.h[0] * l0 + bn.h[5] * l1 + bn.h[4] * l2 + bn.lo[0];
        s->x[3 * i + 1] = bn.h[1] * l1 + bn.h[3] * l2 + bn.lo[1];
        s->x[3 * i + 2] = bn.h[2] * l2 + bn.lo[2];
      }
      flip_pending = omd_tilt_flip(tilt, nhi[0] - nlo[0], nhi[1] - nlo[1], flip_tilt, flip_n);
      /* A yz flip would change xz by xy, which the linear tilt targets of the other components cannot follow; LAMMPS
       * refuses such a run at init ("Fix deform is changing yz too much with xy": in.strain.lammps deforms all six
       * components, so yz and xy are always both active).  xy and xz flip freely. */
      if (flip_n[2] != 0) return -2;
    }
    if (trace) {
      double ke[6], p[6];
      double T = omd_temperature(s, ke);
      pressure_tensor(s, p);
      double pe = 0;
      for (int k = 0; k < OMD_NPART; k++) pe += s->eng[k];
      boxq bb;
      box_derive(s, &bb);
      double *tr = trace + 8 * (size_t)(step - 1);
      tr[0] = T;
      tr[1] = pe;
      tr[2] = 0.5 * (ke[0] + ke[1] + ke[2]);
      tr[3] = nvt ? nh_energy(s, temperature) : 0.0;
      tr[4] = bb.vol;
      tr[5] = p[0];
      tr[6] = p[1];
      tr[7] = p[2];
    }
  }
  if (press_avg)
    for (int k = 0; k < 6; k++) press_avg[k] = psum[k] / (double)(nwin * nav);
  return 0;
}

/* ------------------------------------------------------------------ host arithmetic of F8 */
double omd_round_rate(double rate) {
  char buf[64];
  snprintf(buf, sizeof buf, "%.6e", rate);
  return strtod(buf, NULL);
}
double omd_round_f(double v) {
  char buf[512];
  snprintf(buf, sizeof buf, "%f", v);
  return strtod(buf, NULL);
}

/* stmd_problem.h:229-232 ; strain in raw order xx,yy,zz,xy,xz,yz ; Frobenius norm of the
 * full symmetric tensor */
int omd_nts(const double e[6], double strain_rate, double dt) {
  double nrm = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2] + 2.0 * (e[3] * e[3] + e[4] * e[4] + e[5] * e[5]));
  double strain_time = nrm / strain_rate;
  int nts = (int)(ceil((strain_time / dt) / 10.0) * 10);
  if (nts < 10) nts = 10;
  return nts;
}

int omd_eval(omd_sim *s, const double strain_len[6], double timestep_length, double temperature,
             double strain_rate, int nsteps_sample, double stress_out[6]) {
  memset(s->timing, 0, sizeof(s->timing));
  double t_all = now_s();
  /* stmd_problem.h:213-225 : lbdim = lx,ly,lz ; diag /= own length, xy /= lz, yz /= lx, xz /= ly */
  double lb[3] = {s->hi[0] - s->lo[0], s->hi[1] - s->lo[1], s->hi[2] - s->lo[2]};
  double e[6];
  e[0] = strain_len[0] / lb[0];
  e[1] = strain_len[1] / lb[1];
  e[2] = strain_len[2] / lb[2];
  e[3] = strain_len[3] / lb[2]; /* [0][1] /= lbdim[2] */
  e[5] = strain_len[5] / lb[0]; /* [1][2] /= lbdim[0] */
  e[4] = strain_len[4] / lb[1]; /* [2][0] /= lbdim[1] */
  int nts = omd_nts(e, strain_rate, timestep_length);
  double dts = omd_round_f(timestep_length);
  double tempt = omd_round_f(temperature);
  double rates[6];
  for (int k = 0; k < 6; k++) rates[k] = omd_round_rate(e[k] / (nts * timestep_length));
  /* Phase A: in.strain.lammps */
  int rc = omd_run(s, nts, dts, tempt, 1, 1, rates, NULL, NULL);
  if (rc != 0) return rc;
  /* Phase B: ELASTIC/in.homogenization.lammps (fresh instance: thermostat state reset) */
  double pavg[6];
  rc = omd_run(s, nsteps_sample, dts, tempt, 1, 1, NULL, pavg, NULL);
  if (rc != 0) return rc;
  /* stmd_problem.h:335-341 */
  for (int k = 0; k < 6; k++) stress_out[k] = pavg[k] * (-1.0) * 1.01325e+05;
  double tot = now_s() - t_all;
  s->timing[3] = tot - s->timing[0] - s->timing[1] - s->timing[2];
  return nts;
}

/* ================================================================== init_material: equilibration schedule
 * What lammps_scripts_opls/in.init.lammps asks LAMMPS for, restated [LAMMPS-ext, from the documented behaviour of
 * 17Nov16's min_sd.cpp / min_linesearch.cpp / fix_nh.cpp; PARITY UNPINNED like the rest of this file]:
 *   :48      velocity all create 200.0 ${sseed} rot yes dist gaussian      -> omd_velocity_create (own generator, see there)
 *   :54-58   min_style sd ; minimize 1.0e-7 1.0e-11 ${nsi} 50000          -> omd_minimize
 *   :105-215 fix nvt / fix npt temp T0 T1 100.0 iso 1.0 1.0 1000 ; run N   -> omd_run_nh (no SHAKE: the script's fix shake is
 *            commented out), box-length averages over the second half of an NPT run (fix ave/time 1 nav nav ... ave running)
 *            followed by change_box all x final 0 <lx> ... remap               -> omd_equilibrate */

/* potential energy of the last force evaluation */
static double pe_total(const omd_sim *s) {
  double e = 0.0;
  for (int k = 0; k < OMD_NPART; k++) e += s->eng[k];
  return e;
}
/* energy_force() of a minimiser: reneighbour if needed (results do not depend on when), forces, energy */
static double min_energy_force(omd_sim *s) {
  s->ago++;
  if (neigh_check(s)) neigh_build(s);
  force_compute(s);
  return pe_total(s);
}

/* min_style sd with the default line search (quadratic, dmax 0.1), thermo_modify norm no (units real).
 * Stops: 0 energy tolerance, 1 force tolerance, 2 max iterations, 3 max force evaluations, 4 line search could not go on
 * (search direction not downhill / zero force / alpha backtracked to zero / zero quadratic).  info[4]: iterations, force
 * evaluations, initial and final potential energy. */
int omd_minimize(omd_sim *s, double etol, double ftol, int maxiter, int maxeval, double *info) {
  const double ALPHA_MAX = 1.0, ALPHA_REDUCE = 0.5, BACKTRACK_SLOPE = 0.4, QUADRATIC_TOL = 0.1, EMACH = 1.0e-8, EPS_QUAD = 1.0e-28,
               EPS_ENERGY = 1.0e-8, DMAX = 0.1;
  const int n3 = 3 * s->n;
  omd_setup(s, 0);
  force_compute(s);
  double ecurrent = pe_total(s);
  const double einitial = ecurrent;
  double *h = (double *)xcalloc((size_t)n3, sizeof(double)), *x0 = (double *)xcalloc((size_t)n3, sizeof(double));
  memcpy(h, s->f, (size_t)n3 * sizeof(double));
  int neval = 0, iter = 0, stop = 2;
  for (; iter < maxiter;) {
    iter++;
    const double eprevious = ecurrent, eoriginal = ecurrent;
    int fail = 0;
    {
      /* linemin_quadratic */
      double fdothall = 0.0, hmaxall = 0.0;
      for (int i = 0; i < n3; i++) {
        fdothall += s->f[i] * h[i];
        hmaxall = fmax(hmaxall, fabs(h[i]));
      }
      if (fdothall <= 0.0) fail = 1;
      else if (hmaxall == 0.0) fail = 1;
      else {
        const double alphamax = fmin(ALPHA_MAX, DMAX / hmaxall);
        memcpy(x0, s->x, (size_t)n3 * sizeof(double));
        double alpha = alphamax, fhprev = fdothall, engprev = eoriginal, alphaprev = 0.0;
        for (;;) {
          for (int i = 0; i < n3; i++) s->x[i] = x0[i] + alpha * h[i];
          ecurrent = min_energy_force(s);
          neval++;
          double fh = 0.0;
          for (int i = 0; i < n3; i++) fh += s->f[i] * h[i];
          const double delfh = fh - fhprev;
          if (fabs(fh) < EPS_QUAD || fabs(delfh) < EPS_QUAD) {
            memcpy(s->x, x0, (size_t)n3 * sizeof(double));
            ecurrent = min_energy_force(s);
            fail = 1;
            break;
          }
          const double relerr = fabs(1.0 - (0.5 * (alpha - alphaprev) * (fh + fhprev) + ecurrent) / engprev);
          const double alpha0 = alpha - (alpha - alphaprev) * fh / delfh;
          if (relerr <= QUADRATIC_TOL && alpha0 > 0.0 && alpha0 < alphamax) {
            for (int i = 0; i < n3; i++) s->x[i] = x0[i] + alpha0 * h[i];
            ecurrent = min_energy_force(s);
            neval++;
            if (ecurrent - eoriginal < EMACH) break;
          }
          const double de_ideal = -BACKTRACK_SLOPE * alpha * fdothall, de = ecurrent - eoriginal;
          if (de <= de_ideal) break;
          fhprev = fh;
          engprev = ecurrent;
          alphaprev = alpha;
          alpha *= ALPHA_REDUCE;
          if (alpha <= 0.0 || de_ideal >= -EMACH) {
            memcpy(s->x, x0, (size_t)n3 * sizeof(double));
            ecurrent = min_energy_force(s);
            fail = 1;
            break;
          }
        }
      }
    }
    if (fail) { stop = 4; break; }
    if (neval >= maxeval) { stop = 3; break; }
    if (fabs(ecurrent - eprevious) < etol * 0.5 * (fabs(ecurrent) + fabs(eprevious) + EPS_ENERGY)) { stop = 0; break; }
    double fdotf = 0.0;
    for (int i = 0; i < n3; i++) fdotf += s->f[i] * s->f[i];
    if (fdotf < ftol * ftol) { stop = 1; break; }
    memcpy(h, s->f, (size_t)n3 * sizeof(double));
  }
  if (info) {
    info[0] = iter;
    info[1] = neval;
    info[2] = einitial;
    info[3] = ecurrent;
  }
  free(h);
  free(x0);
  return stop;
}

/* velocity all create T seed rot yes dist gaussian: Gaussian velocities scaled by 1/sqrt(m), zero linear and angular momentum,
 * rescaled to exactly T with 3N-3 degrees of freedom.  LAMMPS draws from its own Park-Miller/Marsaglia streams in atom-id
 * order (loop all); that stream is not reproduced -- any seed gives an equally valid ensemble member and the 10^5..10^6 steps
 * that follow forget it -- so the generator here is a 64-bit SplitMix + Box-Muller, the same in the engine (host side). */
static unsigned long long splitmix64(unsigned long long *st) {
  unsigned long long z = (*st += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
void omd_velocity_create(omd_sim *s, double temperature, unsigned long long seed) {
  const int n = s->n;
  unsigned long long st = seed;
  for (int i = 0; i < n; i++) {
    const double m = s->mass[s->type[i]];
    for (int k = 0; k < 3; k += 1) {
      double u1 = ((double)(splitmix64(&st) >> 11) + 0.5) / 9007199254740992.0, u2 = ((double)(splitmix64(&st) >> 11) + 0.5) / 9007199254740992.0;
      s->v[3 * i + k] = sqrt(-2.0 * log(u1)) * cos(2.0 * MY_PI * u2) / sqrt(m);
    }
  }
  /* zero linear momentum */
  double p[3] = {0, 0, 0}, mt = 0.0;
  for (int i = 0; i < n; i++) {
    const double m = s->mass[s->type[i]];
    mt += m;
    for (int k = 0; k < 3; k++) p[k] += m * s->v[3 * i + k];
  }
  for (int i = 0; i < n; i++)
    for (int k = 0; k < 3; k++) s->v[3 * i + k] -= p[k] / mt;
  /* zero angular momentum about the centre of mass (unwrapped coordinates): v -= omega x r, I omega = L */
  double cm[3] = {0, 0, 0};
  for (int i = 0; i < n; i++) {
    const double m = s->mass[s->type[i]];
    for (int k = 0; k < 3; k++) cm[k] += m * s->x[3 * i + k] / mt;
  }
  double L[3] = {0, 0, 0}, I[3][3] = {{0}};
  for (int i = 0; i < n; i++) {
    const double m = s->mass[s->type[i]];
    const double r[3] = {s->x[3 * i] - cm[0], s->x[3 * i + 1] - cm[1], s->x[3 * i + 2] - cm[2]};
    const double *v = s->v + 3 * i;
    L[0] += m * (r[1] * v[2] - r[2] * v[1]);
    L[1] += m * (r[2] * v[0] - r[0] * v[2]);
    L[2] += m * (r[0] * v[1] - r[1] * v[0]);
    const double r2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) I[a][b] += m * ((a == b ? r2 : 0.0) - r[a] * r[b]);
  }
  {
    const double det = I[0][0] * (I[1][1] * I[2][2] - I[1][2] * I[2][1]) - I[0][1] * (I[1][0] * I[2][2] - I[1][2] * I[2][0]) +
                       I[0][2] * (I[1][0] * I[2][1] - I[1][1] * I[2][0]);
    double inv[3][3];
    inv[0][0] = (I[1][1] * I[2][2] - I[1][2] * I[2][1]) / det; inv[0][1] = (I[0][2] * I[2][1] - I[0][1] * I[2][2]) / det; inv[0][2] = (I[0][1] * I[1][2] - I[0][2] * I[1][1]) / det;
    inv[1][0] = (I[1][2] * I[2][0] - I[1][0] * I[2][2]) / det; inv[1][1] = (I[0][0] * I[2][2] - I[0][2] * I[2][0]) / det; inv[1][2] = (I[0][2] * I[1][0] - I[0][0] * I[1][2]) / det;
    inv[2][0] = (I[1][0] * I[2][1] - I[1][1] * I[2][0]) / det; inv[2][1] = (I[0][1] * I[2][0] - I[0][0] * I[2][1]) / det; inv[2][2] = (I[0][0] * I[1][1] - I[0][1] * I[1][0]) / det;
    double w[3];
    for (int a = 0; a < 3; a++) w[a] = inv[a][0] * L[0] + inv[a][1] * L[1] + inv[a][2] * L[2];
    for (int i = 0; i < n; i++) {
      const double r[3] = {s->x[3 * i] - cm[0], s->x[3 * i + 1] - cm[1], s->x[3 * i + 2] - cm[2]};
      s->v[3 * i] -= w[1] * r[2] - w[2] * r[1];
      s->v[3 * i + 1] -= w[2] * r[0] - w[0] * r[2];
      s->v[3 * i + 2] -= w[0] * r[1] - w[1] * r[0];
    }
  }
  s->tdof = 3.0 * n - 3.0;
  const double t = omd_temperature(s, NULL);
  const double sc = sqrt(temperature / t);
  for (int i = 0; i < 3 * n; i++) s->v[i] *= sc;
}

/* change_box all x final 0 lx y final 0 ly z final 0 lz remap (tilts kept): affine remap of the atoms into the new box */
void omd_change_box(omd_sim *s, const double len[3]) {
  boxq bo, bn;
  box_derive(s, &bo);
  for (int d = 0; d < 3; d++) {
    s->lo[d] = 0.0;
    s->hi[d] = len[d];
  }
  box_derive(s, &bn);
  for (int i = 0; i < s->n; i++) {
    double d0 = s->x[3 * i] - bo.lo[0], d1 = s->x[3 * i + 1] - bo.lo[1], d2 = s->x[3 * i + 2] - bo.lo[2];
    double l0 = bo.hinv[0] * d0 + bo.hinv[5] * d1 + bo.hinv[4] * d2, l1 = bo.hinv[1] * d1 + bo.hinv[3] * d2, l2 = bo.hinv[2] * d2;
    s->x[3 * i] = bn.h[0] * l0 + bn.h[5] * l1 + bn.h[4] * l2 + bn.lo[0];
    s->x[3 * i + 1] = bn.h[1] * l1 + bn.h[3] * l2 + bn.lo[1];
    s->x[3 * i + 2] = bn.h[2] * l2 + bn.lo[2];
  }
}

/* barostat of fix npt ... iso p p pperiod (pchain 3, mtk yes, drag 0, nreset 0; tilt factors scale with their box lengths) */
typedef struct {
  double omega_dot, omega_mass, etap[MAXCHAIN + 1], etap_dot[MAXCHAIN + 1], etap_dotdot[MAXCHAIN + 1], etap_mass[MAXCHAIN + 1];
  double p_freq, p_target, mtk_term2;
  int mp;
} baro;

static double pressure_scalar(const omd_sim *s, double t_current) {
  boxq b;
  box_derive(s, &b);
  double w = 0.0;
  for (int part = 0; part < OMD_NPART; part++) w += s->vir[part * 6] + s->vir[part * 6 + 1] + s->vir[part * 6 + 2];
  return (s->tdof * BOLTZ * t_current + w) / (3.0 * b.vol) * NKTV2P;
}
static void nhc_press_integrate(baro *B, double dt, double t_target) {
  const double kt = BOLTZ * t_target, dthalf = 0.5 * dt, dt4 = 0.25 * dt, dt8 = 0.125 * dt;
  const int mp = B->mp;
  /* iso: the three box dimensions carry the same omega_dot, each with its own mass term */
  double kecurrent = 3.0 * B->omega_mass * B->omega_dot * B->omega_dot;
  const double lkt_press = kt;
  B->etap_dotdot[0] = (kecurrent - lkt_press) / B->etap_mass[0];
  double expfac;
  for (int k = mp - 1; k > 0; k--) {
    expfac = exp(-dt8 * B->etap_dot[k + 1]);
    B->etap_dot[k] *= expfac;
    B->etap_dot[k] += B->etap_dotdot[k] * dt4;
    B->etap_dot[k] *= expfac;
  }
  expfac = exp(-dt8 * B->etap_dot[1]);
  B->etap_dot[0] *= expfac;
  B->etap_dot[0] += B->etap_dotdot[0] * dt4;
  B->etap_dot[0] *= expfac;
  for (int k = 0; k < mp; k++) B->etap[k] += dthalf * B->etap_dot[k];
  B->omega_dot *= exp(-dthalf * B->etap_dot[0]);
  kecurrent = 3.0 * B->omega_mass * B->omega_dot * B->omega_dot;
  B->etap_dotdot[0] = (kecurrent - lkt_press) / B->etap_mass[0];
  B->etap_dot[0] *= expfac;
  B->etap_dot[0] += B->etap_dotdot[0] * dt4;
  B->etap_dot[0] *= expfac;
  for (int k = 1; k < mp; k++) {
    expfac = exp(-dt8 * B->etap_dot[k + 1]);
    B->etap_dot[k] *= expfac;
    B->etap_dotdot[k] = (B->etap_mass[k - 1] * B->etap_dot[k - 1] * B->etap_dot[k - 1] - kt) / B->etap_mass[k];
    B->etap_dot[k] += B->etap_dotdot[k] * dt4;
    B->etap_dot[k] *= expfac;
  }
}
static void nh_omega_dot(const omd_sim *s, baro *B, double dt, double p_current, double t_current) {
  boxq b;
  box_derive(s, &b);
  const double mtk_term1 = s->tdof * BOLTZ * t_current / (3.0 * s->n);
  const double f_omega = (p_current - B->p_target) * b.vol / (B->omega_mass * NKTV2P) + mtk_term1 / B->omega_mass;
  B->omega_dot += f_omega * 0.5 * dt;
  B->mtk_term2 = 3.0 * B->omega_dot / (3.0 * s->n);
}
static void nh_v_press(omd_sim *s, const baro *B, double dt) {
  const double factor = exp(-0.25 * dt * (B->omega_dot + B->mtk_term2));
  for (int i = 0; i < 3 * s->n; i++) {
    s->v[i] *= factor;
    s->v[i] *= factor;
  }
}
/* half-step dilation about the box centre; xy scales with ly, xz and yz with lz (scalexy/scalexz/scaleyz yes) */
static void nh_remap(omd_sim *s, const baro *B, double dt) {
  boxq bo, bn;
  box_derive(s, &bo);
  const double expfac = exp(0.5 * dt * B->omega_dot);
  for (int d = 0; d < 3; d++) {
    const double c = 0.5 * (s->lo[d] + s->hi[d]);
    const double lo = s->lo[d], hi = s->hi[d];
    s->lo[d] = (lo - c) * expfac + c;
    s->hi[d] = (hi - c) * expfac + c;
  }
  s->xy *= expfac;
  s->xz *= expfac;
  s->yz *= expfac;
  box_derive(s, &bn);
  for (int i = 0; i < s->n; i++) {
    double d0 = s->x[3 * i] - bo.lo[0], d1 = s->x[3 * i + 1] - bo.lo[1], d2 = s->x[3 * i + 2] - bo.lo[2];
    double l0 = bo.hinv[0] * d0 + bo.hinv[5] * d1 + bo.hinv[4] * d2, l1 = bo.hinv[1] * d1 + bo.hinv[3] * d2, l2 = bo.hinv[2] * d2;
    s->x[3 * i] = bn.h[0] * l0 + bn.h[5] * l1 + bn.h[4] * l2 + bn.lo[0];
    s->x[3 * i + 1] = bn.h[1] * l1 + bn.h[3] * l2 + bn.lo[1];
    s->x[3 * i + 2] = bn.h[2] * l2 + bn.lo[2];
  }
}

/* One "run N" under fix nvt (npt = 0) or fix npt ... iso (npt = 1) with a temperature ramp t_start -> t_stop over the run, no
 * SHAKE, no deform.  lavg != NULL: box lengths averaged like fix ave/time 1 nav nav v_lx v_ly v_lz ave running with
 * nav = nsteps / 2 (in.init.lammps:150-157: one window, the second half of the run... as the fix defines it: the nav samples
 * that end at step nav, then those that end at 2 nav; ave running averages the two windows) -> lavg[3].
 * trace != NULL: per step 6 doubles T, pe, ke, conserved-quantity part of thermostat + barostat, volume, scalar pressure. */
int omd_run_nh(omd_sim *s, int nsteps, double dt, double t_start, double t_stop, int npt, double p_target, double p_period,
               double *lavg, double *trace) {
  const int n = s->n, mt = s->p.t_chain;
  const double dtv = dt, dtf = 0.5 * dt * FTM2V;
  omd_setup(s, 0);
  force_compute(s);
  for (int k = 0; k <= MAXCHAIN; k++) s->eta[k] = s->eta_dot[k] = s->eta_dotdot[k] = 0.0;
  s->t_current = omd_temperature(s, NULL);
  double t_target = t_start;
  {
    const double t_freq = 1.0 / s->p.t_period;
    s->eta_mass[0] = s->tdof * BOLTZ * t_target / (t_freq * t_freq);
    for (int k = 1; k < mt; k++) s->eta_mass[k] = BOLTZ * t_target / (t_freq * t_freq);
    for (int k = 1; k < mt; k++)
      s->eta_dotdot[k] = (s->eta_mass[k - 1] * s->eta_dot[k - 1] * s->eta_dot[k - 1] - BOLTZ * t_target) / s->eta_mass[k];
  }
  baro B;
  memset(&B, 0, sizeof B);
  double p_current = 0.0;
  if (npt) {
    B.mp = 3;
    B.p_freq = 1.0 / p_period;
    B.p_target = p_target;
    const double kt = BOLTZ * t_target;
    B.omega_mass = (n + 1) * kt / (B.p_freq * B.p_freq);   /* fixed for the run (omega_mass_flag 0) */
    for (int k = 0; k < B.mp; k++) B.etap_mass[k] = kt / (B.p_freq * B.p_freq);
    for (int k = 1; k < B.mp; k++) B.etap_dotdot[k] = (B.etap_mass[k - 1] * B.etap_dot[k - 1] * B.etap_dot[k - 1] - kt) / B.etap_mass[k];
    p_current = pressure_scalar(s, s->t_current);
  }
  const int nav = nsteps / 2;
  double lsum[3] = {0, 0, 0}, lrun[3] = {0, 0, 0};
  int nwin = 0;
  for (int step = 1; step <= nsteps; step++) {
    /* initial_integrate */
    if (npt) nhc_press_integrate(&B, dt, t_target);   /* thermostat target of the previous half step, as fix_nh orders it */
    t_target = t_start + (t_stop - t_start) * (double)step / (double)nsteps;
    nhc_temp_integrate(s, dt, t_target);
    if (npt) {
      p_current = pressure_scalar(s, s->t_current);   /* kinetic part after the thermostat half step, virial of the last forces */
      nh_omega_dot(s, &B, dt, p_current, s->t_current);
      nh_v_press(s, &B, dt);
    }
    for (int i = 0; i < n; i++) {
      const double dtfm = dtf / s->mass[s->type[i]];
      for (int k = 0; k < 3; k++) s->v[3 * i + k] += dtfm * s->f[3 * i + k];
    }
    if (npt) nh_remap(s, &B, dt);
    for (int i = 0; i < 3 * n; i++) s->x[i] += dtv * s->v[i];
    if (npt) nh_remap(s, &B, dt);
    s->ago++;
    if (s->ago >= s->p.neigh_delay && neigh_check(s)) neigh_build(s);
    force_compute(s);
    /* final_integrate */
    for (int i = 0; i < n; i++) {
      const double dtfm = dtf / s->mass[s->type[i]];
      for (int k = 0; k < 3; k++) s->v[3 * i + k] += dtfm * s->f[3 * i + k];
    }
    if (npt) nh_v_press(s, &B, dt);
    s->t_current = omd_temperature(s, NULL);
    if (npt) {
      p_current = pressure_scalar(s, s->t_current);
      nh_omega_dot(s, &B, dt, p_current, s->t_current);
    }
    nhc_temp_integrate(s, dt, t_target);
    if (npt) nhc_press_integrate(&B, dt, t_target);
    /* end_of_step: running average of the box lengths */
    if (lavg && nav > 0 && step <= 2 * nav) {
      for (int d = 0; d < 3; d++) lsum[d] += s->hi[d] - s->lo[d];
      if (step % nav == 0) {
        for (int d = 0; d < 3; d++) {
          lrun[d] += lsum[d] / nav;
          lsum[d] = 0.0;
        }
        nwin++;
      }
    }
    if (trace) {
      double *tr = trace + 6 * (size_t)(step - 1);
      double ke[6];
      const double T = omd_temperature(s, ke);
      boxq bb;
      box_derive(s, &bb);
      tr[0] = T;
      tr[1] = pe_total(s);
      tr[2] = 0.5 * (ke[0] + ke[1] + ke[2]);
      double ec = nh_energy(s, t_target);
      if (npt) {
        /* fix_nh::compute_scalar, iso: 0.5 W omega_dot^2 per box dimension + p_hydro V / nktv2p + barostat chain */
        ec += 1.5 * B.omega_mass * B.omega_dot * B.omega_dot;
        ec += B.p_target * bb.vol / NKTV2P;
        const double kt = BOLTZ * t_target;
        for (int k = 0; k < B.mp; k++) ec += kt * B.etap[k] + 0.5 * B.etap_mass[k] * B.etap_dot[k] * B.etap_dot[k];
      }
      tr[3] = ec;
      tr[4] = bb.vol;
      tr[5] = npt ? p_current : pressure_scalar(s, T);
    }
  }
  if (lavg)
    for (int d = 0; d < 3; d++) lavg[d] = nwin ? lrun[d] / nwin : s->hi[d] - s->lo[d];
  return 0;
}

/* in.init.lammps:44-215 on the registered configuration: velocities at 200 K, minimisation, the heat-up / cool-down
 * schedule in units of nsinit steps.  lengths[3] = box lengths at the end (what init.<mat>_<rep>.length stores). */
int omd_equilibrate(omd_sim *s, int nsinit, double dt, double tempt, unsigned long long seed, double lengths[3], double *min_info) {
  double lav[3];
  omd_velocity_create(s, 200.0, seed);
  omd_minimize(s, 1.0e-7, 1.0e-11, nsinit, 50000, min_info);
  omd_run_nh(s, nsinit, dt, 300.0, 300.0, 0, 0.0, 0.0, NULL, NULL);
  omd_run_nh(s, nsinit, dt, 300.0, 500.0, 1, 1.0, 1000.0, NULL, NULL);
  omd_run_nh(s, 5 * nsinit, dt, 500.0, 500.0, 1, 1.0, 1000.0, NULL, NULL);
  omd_run_nh(s, nsinit, dt, 500.0, tempt, 1, 1.0, 1000.0, NULL, NULL);
  omd_run_nh(s, 2 * nsinit, dt, tempt, tempt, 1, 1.0, 1000.0, lav, NULL);
  omd_change_box(s, lav);
  omd_run_nh(s, 20 * nsinit, dt, tempt, tempt, 0, 0.0, 0.0, NULL, NULL);
  omd_run_nh(s, 2 * nsinit, dt, tempt, tempt, 1, 1.0, 1000.0, lav, NULL);
  omd_change_box(s, lav);
  omd_run_nh(s, nsinit, dt, tempt, tempt, 0, 0.0, 0.0, NULL, NULL);
  for (int d = 0; d < 3; d++) lengths[d] = s->hi[d] - s->lo[d];
  return 0;
}
