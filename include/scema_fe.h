/*
 * scema_fe.h -- a minimal explicit-dynamics continuum stand-in for the FE side of SCEMa, so that the
 * STMDSync path can be driven through whole continuum steps without deal.II / PETSc: a cuboid of nx x ny x nz trilinear
 * hexahedra, 2x2x2 Gauss points each (quadrature point id = cell*8 + q, FE_problem.h:485), lumped mass, explicit time
 * stepping.  It is NOT a restatement of FEProblem's solver (out of scope); what it keeps is the CALL CONTRACT
 * of the hot path:
 *   - upd_strain accumulates the strain increments of a quadrature point since its last MD update; a point enters the
 *     update_list when |upd_strain| >= min_qp_strain (FE_problem.h:1144-1155, "model precision.md.min quadrature strain norm");
 *   - QP.most_recent_id = the id the point got its previous result from (UINT32_MAX before the first), QP.id = its own id
 *     (FE_problem.h:1091-1103,1344-1350);
 *   - after the MD round an updated point takes the returned stress as its ABSOLUTE stress (Hooke test mode: returned +
 *     old stress) and resets upd_strain; every other point continues linear-elastically, stress += C : d(strain)
 *     (FE_problem.h:1667-1700).
 * One continuum step = scema_fe_solve -> [scema_stmd_update on the list] -> scema_fe_check, the body of
 * HMMProblem::do_timestep (dealammps.cc:417-474).
 */
#ifndef SCEMA_FE_H
#define SCEMA_FE_H

#include <stdint.h>

#include "scema_stmd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int32_t nx, ny, nz;        /* cells (inputs_dogbone_cuboid.json: 3 x 3 x 8) */
  double lx, ly, lz;         /* edge lengths of the cuboid, m */
  double density;            /* kg/m^3 (macroscale_output/init.<mat>.density) */
  double stiffness[36];      /* Pa, file order of init.<mat>.stiff (read_write.h:149-171) */
  double dt;                 /* continuum time step, s */
  double top_velocity;       /* m/s: the face z = lz moves in z at this speed, the face z = 0 is held (dogbone loading) */
  double min_qp_strain;      /* 1e-10 in the reference's inputs */
  int32_t hooke;             /* "approximate md with hookes law": returned stresses are increments */
  int32_t material;          /* material index written into QP.material */
} scema_fe_config;

typedef struct scema_fe scema_fe;

int scema_fe_create(const scema_fe_config *cfg, scema_fe **out);
void scema_fe_destroy(scema_fe *f);
int32_t scema_fe_n_qp(const scema_fe *f);
int32_t scema_fe_n_nodes(const scema_fe *f);
/* FEProblem::solve: one explicit step of the mesh, strain update of every quadrature point, then the update_list
 * (capacity >= n_qp) of the points whose accumulated strain passed the threshold */
int scema_fe_solve(scema_fe *f, scema_qp *update_list, int32_t capacity, int32_t *n_update);
/* FEProblem::check + endstep: takes the update_list back with its update_stress filled */
int scema_fe_check(scema_fe *f, const scema_qp *update_list, int32_t n_update);
/* state for inspection: displacement[3*n_nodes], qp_strain[6*n_qp], qp_stress[6*n_qp] (raw order xx,yy,zz,xy,xz,yz); NULLs skipped */
int scema_fe_get(const scema_fe *f, double *displacement, double *qp_strain, double *qp_stress);
/* test hook: overwrite nodal velocities (3*n_nodes) */
int scema_fe_set_velocity(scema_fe *f, const double *velocity);
double scema_fe_kinetic_energy(const scema_fe *f);

#ifdef __cplusplus
}
#endif
#endif
