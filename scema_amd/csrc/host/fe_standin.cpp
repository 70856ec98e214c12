// fe_standin.cpp -- minimal explicit-dynamics continuum stand-in behind include/scema_fe.h.
// Keeps the call contract of the hot path (FE_problem.h:1091-1161,1296-1373,1667-1700), not FEProblem's solver.
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "../../../include/scema_fe.h"

namespace {
const int RAW_OF[3][3] = {{0, 3, 4}, {3, 1, 5}, {4, 5, 2}};      // deal.II raw order xx,yy,zz,xy,xz,yz
const int FILE_OF[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};     // init.*.stiff file order 00,01,02,11,12,22

struct PointHistory {   // FE_problem.h PointHistory, the fields the contract needs
  double new_strain[6] = {0}, upd_strain[6] = {0}, newton_strain[6] = {0}, new_stress[6] = {0}, old_stress[6] = {0};
  bool to_be_updated_with_md = false;
  int32_t qpid = 0, id_to_get_results_from = (int32_t)0xFFFFFFFF, most_recent_id = (int32_t)0xFFFFFFFF;
};

void contract(const double *c, const double *eps, double *out) {   // sigma = C : eps (raw order in and out)
  for (int k = 0; k < 3; k++)
    for (int l = k; l < 3; l++) {
      double acc = 0.0;
      for (int m = 0; m < 3; m++)
        for (int n = 0; n < 3; n++) acc += c[FILE_OF[k][l] * 6 + FILE_OF[m][n]] * eps[RAW_OF[m][n]];
      out[RAW_OF[k][l]] = acc;
    }
}
double norm_sym(const double *t) {   // Frobenius norm of the full symmetric tensor (SymmetricTensor::norm)
  return std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + 2.0 * (t[3] * t[3] + t[4] * t[4] + t[5] * t[5]));
}
}  // namespace

struct scema_fe {
  scema_fe_config c;
  int nnx, nny, nnz, nnodes, ncells, nqp;
  double h[3];
  std::vector<double> u, v, mass, fint;
  std::vector<PointHistory> qph;
  double dN[8][8][3];   // shape-function gradients at the 8 Gauss points (identical for every cell of the regular mesh)
  double wq;            // quadrature weight x Jacobian
  int node(int i, int j, int k) const { return (k * nny + j) * nnx + i; }
  void cell_nodes(int cell, int *n) const {
    const int ci = cell % c.nx, cj = (cell / c.nx) % c.ny, ck = cell / (c.nx * c.ny);
    int m = 0;
    for (int dk = 0; dk < 2; dk++)
      for (int dj = 0; dj < 2; dj++)
        for (int di = 0; di < 2; di++) n[m++] = node(ci + di, cj + dj, ck + dk);
  }
};

extern "C" {

int scema_fe_create(const scema_fe_config *cfg, scema_fe **out) {
  if (!cfg || !out || cfg->nx < 1 || cfg->ny < 1 || cfg->nz < 1 || !(cfg->lx > 0) || !(cfg->ly > 0) || !(cfg->lz > 0) || !(cfg->density > 0) || !(cfg->dt > 0))
    return SCEMA_MD_ERR_ARG;
  scema_fe *f = new (std::nothrow) scema_fe();
  if (!f) return SCEMA_MD_ERR_ARG;
  f->c = *cfg;
  f->nnx = cfg->nx + 1; f->nny = cfg->ny + 1; f->nnz = cfg->nz + 1;
  f->nnodes = f->nnx * f->nny * f->nnz;
  f->ncells = cfg->nx * cfg->ny * cfg->nz;
  f->nqp = 8 * f->ncells;
  f->h[0] = cfg->lx / cfg->nx; f->h[1] = cfg->ly / cfg->ny; f->h[2] = cfg->lz / cfg->nz;
  f->u.assign(3 * (size_t)f->nnodes, 0.0);
  f->v.assign(3 * (size_t)f->nnodes, 0.0);
  f->fint.assign(3 * (size_t)f->nnodes, 0.0);
  f->mass.assign(f->nnodes, 0.0);
  f->qph.resize(f->nqp);
  for (int q = 0; q < f->nqp; q++) f->qph[q].qpid = q;   // cell*8 + q (FE_problem.h:485)
  const double g = 1.0 / std::sqrt(3.0);
  for (int q = 0; q < 8; q++) {
    const double xi[3] = {(q & 1) ? g : -g, (q & 2) ? g : -g, (q & 4) ? g : -g};
    for (int a = 0; a < 8; a++) {
      const double s[3] = {(a & 1) ? 1.0 : -1.0, (a & 2) ? 1.0 : -1.0, (a & 4) ? 1.0 : -1.0};
      for (int d = 0; d < 3; d++) {
        double val = 0.125 * s[d];
        for (int e = 0; e < 3; e++)
          if (e != d) val *= (1.0 + s[e] * xi[e]);
        f->dN[q][a][d] = val * 2.0 / f->h[d];   // d/dx = d/dxi * 2/h on the regular mesh
      }
    }
  }
  f->wq = f->h[0] * f->h[1] * f->h[2] / 8.0;
  const double mcell = cfg->density * f->h[0] * f->h[1] * f->h[2] / 8.0;   // lumped: an eighth of the cell's mass per corner
  int n[8];
  for (int cell = 0; cell < f->ncells; cell++) {
    f->cell_nodes(cell, n);
    for (int a = 0; a < 8; a++) f->mass[n[a]] += mcell;
  }
  *out = f;
  return SCEMA_MD_OK;
}

void scema_fe_destroy(scema_fe *f) { delete f; }
int32_t scema_fe_n_qp(const scema_fe *f) { return f ? f->nqp : 0; }
int32_t scema_fe_n_nodes(const scema_fe *f) { return f ? f->nnodes : 0; }

int scema_fe_solve(scema_fe *f, scema_qp *update_list, int32_t capacity, int32_t *n_update) {
  if (!f || !update_list || !n_update) return SCEMA_MD_ERR_ARG;
  const scema_fe_config &c = f->c;
  // internal forces from the current quadrature-point stresses: f_int = sum_q w B^T sigma
  std::fill(f->fint.begin(), f->fint.end(), 0.0);
  int n[8];
  for (int cell = 0; cell < f->ncells; cell++) {
    f->cell_nodes(cell, n);
    for (int q = 0; q < 8; q++) {
      const double *s = f->qph[cell * 8 + q].new_stress;
      for (int a = 0; a < 8; a++) {
        const double *g = f->dN[q][a];
        f->fint[3 * n[a] + 0] += f->wq * (s[0] * g[0] + s[3] * g[1] + s[4] * g[2]);
        f->fint[3 * n[a] + 1] += f->wq * (s[3] * g[0] + s[1] * g[1] + s[5] * g[2]);
        f->fint[3 * n[a] + 2] += f->wq * (s[4] * g[0] + s[5] * g[1] + s[2] * g[2]);
      }
    }
  }
  // explicit update of the velocities; loading: the bottom face is held, the top face moves in z at top_velocity
  std::vector<double> du(3 * (size_t)f->nnodes);
  for (int k = 0; k < f->nnz; k++)
    for (int j = 0; j < f->nny; j++)
      for (int i = 0; i < f->nnx; i++) {
        const int m = f->node(i, j, k);
        for (int d = 0; d < 3; d++) f->v[3 * m + d] -= c.dt * f->fint[3 * m + d] / f->mass[m];
        if (k == 0) f->v[3 * m] = f->v[3 * m + 1] = f->v[3 * m + 2] = 0.0;
        if (k == f->nnz - 1 && c.top_velocity != 0.0) f->v[3 * m + 2] = c.top_velocity;
        for (int d = 0; d < 3; d++) { du[3 * m + d] = c.dt * f->v[3 * m + d]; f->u[3 * m + d] += du[3 * m + d]; }
      }
  // update_strain_quadrature_point_history (FE_problem.h:1044-1106) + check_strain_quadrature_point_history (:1114-1161)
  *n_update = 0;
  for (int cell = 0; cell < f->ncells; cell++) {
    f->cell_nodes(cell, n);
    for (int q = 0; q < 8; q++) {
      PointHistory &p = f->qph[cell * 8 + q];
      double grad[3][3] = {{0}};
      for (int a = 0; a < 8; a++)
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++) grad[i][j] += du[3 * n[a] + i] * f->dN[q][a][j];
      std::memcpy(p.old_stress, p.new_stress, sizeof p.old_stress);
      for (int i = 0; i < 3; i++)
        for (int j = i; j < 3; j++) p.newton_strain[RAW_OF[i][j]] = 0.5 * (grad[i][j] + grad[j][i]);
      for (int k = 0; k < 6; k++) { p.new_strain[k] += p.newton_strain[k]; p.upd_strain[k] += p.newton_strain[k]; }
      p.most_recent_id = p.id_to_get_results_from;   // FE_problem.h:1099-1103 (no clustering: results always come from the point itself)
      p.id_to_get_results_from = p.qpid;
      p.to_be_updated_with_md = norm_sym(p.upd_strain) >= c.min_qp_strain || p.to_be_updated_with_md;
      if (p.to_be_updated_with_md) {   // write_md_updates_list (FE_problem.h:1296-1373); rotam = identity (:352-356)
        if (*n_update >= capacity) return SCEMA_MD_ERR_ARG;
        scema_qp &qp = update_list[(*n_update)++];
        qp.id = p.qpid;
        qp.most_recent_id = p.most_recent_id;
        qp.material = c.material;
        for (int k = 0; k < 6; k++) { qp.update_strain[k] = p.upd_strain[k]; qp.update_stress[k] = 0.0; }
      }
    }
  }
  return SCEMA_MD_OK;
}

int scema_fe_check(scema_fe *f, const scema_qp *update_list, int32_t n_update) {
  if (!f || (n_update > 0 && !update_list) || n_update < 0) return SCEMA_MD_ERR_ARG;
  // update_stress_quadrature_point_history (FE_problem.h:1631-1752)
  std::vector<int> where(f->nqp, -1);
  for (int k = 0; k < n_update; k++) {
    if (update_list[k].id < 0 || update_list[k].id >= f->nqp) return SCEMA_MD_ERR_ARG;
    where[update_list[k].id] = k;
  }
  for (PointHistory &p : f->qph) {
    if (p.to_be_updated_with_md) {
      const int k = where[p.id_to_get_results_from];
      if (k < 0) return SCEMA_MD_ERR_ARG;   // get_qp_with_id would fail
      const double *s = update_list[k].update_stress;
      for (int i = 0; i < 6; i++) p.new_stress[i] = f->c.hooke ? s[i] + p.old_stress[i] : s[i];
      for (int i = 0; i < 6; i++) p.upd_strain[i] = 0.0;
      p.to_be_updated_with_md = false;
    } else {
      double ds[6];
      contract(f->c.stiffness, p.newton_strain, ds);
      for (int i = 0; i < 6; i++) p.new_stress[i] += ds[i];
    }
  }
  return SCEMA_MD_OK;
}

int scema_fe_get(const scema_fe *f, double *displacement, double *qp_strain, double *qp_stress) {
  if (!f) return SCEMA_MD_ERR_ARG;
  if (displacement) std::memcpy(displacement, f->u.data(), f->u.size() * sizeof(double));
  for (int q = 0; q < f->nqp; q++) {
    if (qp_strain) std::memcpy(qp_strain + 6 * (size_t)q, f->qph[q].new_strain, 6 * sizeof(double));
    if (qp_stress) std::memcpy(qp_stress + 6 * (size_t)q, f->qph[q].new_stress, 6 * sizeof(double));
  }
  return SCEMA_MD_OK;
}

int scema_fe_set_velocity(scema_fe *f, const double *velocity) {
  if (!f || !velocity) return SCEMA_MD_ERR_ARG;
  std::memcpy(f->v.data(), velocity, f->v.size() * sizeof(double));
  return SCEMA_MD_OK;
}

double scema_fe_kinetic_energy(const scema_fe *f) {
  double e = 0.0;
  if (f)
    for (int m = 0; m < f->nnodes; m++) e += 0.5 * f->mass[m] * (f->v[3 * m] * f->v[3 * m] + f->v[3 * m + 1] * f->v[3 * m + 1] + f->v[3 * m + 2] * f->v[3 * m + 2]);
  return e;
}

}  // extern "C"
