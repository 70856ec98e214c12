/*
 * reax_oracle.h -- CPU restatement (FP64, plain C) of the ReaxFF potential that SCEMa's reax branch selects
 * (reference lammps_scripts/lammps_scripts_reax/in.strain.lammps:10-12: `pair_style reax/c NULL safezone 50 mincap 100000`,
 * `pair_coeff * * ffield.reax.2 H C N O`, `fix qeq/reax 1 0.0 10.0 1e-6 reax/c`; SURVEY.md 8(f) row f-4, BASELINE config 5).
 *
 * TEST INFRASTRUCTURE ONLY: the checker of the HIP path for this force field (scema_amd/csrc/md_reax.hip, reax/rx_core.h;
 * force_field "reax" of scema_md_strain_batch).  Nothing in the product path may link, import or execute this code.
 * Dynamics (a whole strained evaluation with its expected stress): oracle/reax_md.py = the integrator, thermostat, fix deform
 * and pressure average of oracle/md_oracle.c around the forces of oracle/reax_torch.py, which differentiates the energy
 * expression of this file in reverse mode.
 *
 * PARITY UNPINNED: the arithmetic lives in LAMMPS 17Nov16, package USER-REAXC, which is neither vendored by the reference
 * nor installed here.  This file restates the published functional forms as that package implements them (van Duin et
 * al., J. Phys. Chem. A 105, 9396 (2001); Chenoweth et al., J. Phys. Chem. A 112, 1040 (2008); the force-field file format
 * of the ReaxFF user manual): bond orders with the over-coordination and 1-3 corrections, bond, lone-pair, over- and
 * under-coordination, valence-angle + penalty + three-body conjugation, torsion + four-body conjugation, hydrogen-bond,
 * tapered shielded van der Waals and Coulomb energies, and charge equilibration (two conjugate-gradient solves).  The
 * ENERGY is analytic; FORCES and the VIRIAL are central differences of it at fixed charges (as LAMMPS evaluates them: no
 * dq/dr terms), which is what a checker of hand-written force kernels needs.  Pinned by invariants only
 * (tests/test_oracle_reax.py).  Not restated: the terminal-triple-bond and C2 corrections (both switched off by the
 * reference's ffield.reax.2: parameters 11 and 6 are 0), the inner-wall van der Waals variants (rcore = 0 there).
 */
#ifndef REAX_ORACLE_H
#define REAX_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rxo_ff rxo_ff;

/* energy parts, in LAMMPS' `compute pair reax/c` spirit */
enum {
  RXO_BOND = 0, RXO_LP, RXO_OVER, RXO_UNDER, RXO_ANGLE, RXO_PEN, RXO_COA, RXO_TORS, RXO_CONJ, RXO_HB, RXO_VDW, RXO_COUL, RXO_POL,
  RXO_NPART
};

/* reads a ReaxFF force-field file (ffield.reax.*); NULL + message on stderr on failure */
rxo_ff *rxo_read_ffield(const char *path);
void rxo_free_ffield(rxo_ff *ff);
int rxo_ntypes(const rxo_ff *ff);
/* element symbol of force-field type t (0-based, file order), mass */
const char *rxo_type_name(const rxo_ff *ff, int t);
double rxo_type_mass(const rxo_ff *ff, int t);
double rxo_general(const rxo_ff *ff, int k);        /* general parameter k (0-based) */

/* type[i] = 0-based force-field type; box = xlo,ylo,zlo,xhi,yhi,zhi,xy,xz,yz, or NULL for an isolated cluster.
 * Periodic boxes must be at least 20 A wide in every direction (minimum image at the 10 A taper radius). */
double rxo_energy(const rxo_ff *ff, int n, const int *type, const double *x, const double *box, const double *q,
                  double parts[RXO_NPART]);
/* corrected bond orders of all pairs with BO' >= cutoff: returns the count, fills up to cap entries of (i, j, BO, BO_pi, BO_pi2) */
int rxo_bond_orders(const rxo_ff *ff, int n, const int *type, const double *x, const double *box, int cap, int *ij, double *bo);
/* charge equilibration (fix qeq/reax nevery cutlo cuthi tol): q out; returns CG iterations of the two solves summed, <0 on failure */
int rxo_qeq(const rxo_ff *ff, int n, const int *type, const double *x, const double *box, double tol, int maxiter, double *q);
/* central differences of the energy at fixed charges: f[3n] = -dE/dx, virial[6] = -dE/d(strain) (xx,yy,zz,xy,xz,yz; periodic only) */
void rxo_forces_fd(const rxo_ff *ff, int n, const int *type, const double *x, const double *box, const double *q, double h,
                   double *f, double *virial);

/* parameter tables as read (layouts in reax_oracle.c), for oracle/reax_torch.py; returns the number of types */
int rxo_export(const rxo_ff *ff, double *gp, double *sbp, double *tbp, double *thbp, double *fbp, double *hbp, double *misc);

#ifdef __cplusplus
}
#endif
#endif
