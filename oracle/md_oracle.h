/*
 * md_oracle.h -- CPU restatement (FP64, plain C) of the strained-MD stress sampler that
 * SCEMa runs behind STMDProblem<3>::strain (reference: headers/stmd_problem.h:84-383).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (scema_amd/, include/) may link,
 * import or execute this code; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in LAMMPS 17Nov16 (packages KSPACE,
 * MOLECULE, RIGID), an un-vendored dependency that is absent from /root/reference
 * (README.md:31-37, CMakeLists/archer.CMakeLists.txt:28-39) and the reference ships no
 * golden stresses for any OPLS system.  This file restates the published algorithms that the
 * reference's scripts select (lammps_scripts_opls/in.set.lammps:13-57, in.strain.lammps:68-124,
 * ELASTIC/in.homogenization.lammps:45-96) and is pinned by known-answer / invariant tests
 * (tests/test_oracle_*.py), not by reference outputs.
 */
#ifndef MD_ORACLE_H
#define MD_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* energy / virial parts */
enum {
  OMD_LJ = 0,      /* pair lj/cut                            */
  OMD_COUL = 1,    /* pair coul/long real space (+special corrections) */
  OMD_BOND = 2,
  OMD_ANGLE = 3,
  OMD_DIHEDRAL = 4,
  OMD_IMPROPER = 5,
  OMD_KSPACE = 6,  /* reciprocal Ewald sum (+self, for the energy) */
  OMD_SHAKE = 7,   /* constraint forces (virial only)        */
  OMD_NPART = 8
};

typedef struct {
  double cut_lj;        /* in.set.lammps:40  pair_style lj/cut/coul/long 12.0 9.0 */
  double cut_coul;
  double skin;          /* in.set.lammps:27  neighbor 2.0 bin */
  int    neigh_delay;   /* in.set.lammps:32  neigh_modify every 1 delay 5 check yes */
  double kspace_accuracy;   /* in.set.lammps:36 kspace_style pppm 0.0001 */
  double shake_tol;     /* in.strain.lammps:71 fix shake 0.001 20 1000 m 1.0 */
  int    shake_maxiter;
  double shake_mass;    /* <= 0: no SHAKE */
  double t_period;      /* in.strain.lammps:80 fix nvt temp T T 100.0 */
  int    t_chain;       /* Nose-Hoover chain length (fix nvt default 3) */
  int    kspace_pppm;   /* 1 (default): PPPM as `kspace_style pppm` asks for; 0: the plain Ewald sum it approximates
                         * (order 5, ik differentiation; grid and g_ewald by the rules of pppm.cpp as restated in md_oracle.c) */
  int    pppm_mesh[3];  /* all > 0: `kspace_modify mesh nx ny nz` (the grid is given, the search and the triclinic rescaling of
                         * set_grid_global are skipped; still raised to products of 2, 3, 5); default 0 0 0 = the rule decides */
} omd_params;

void omd_default_params(omd_params *p);

/* A force field from outside this file (the ReaxFF oracle, oracle/reax_md.py): fn fills f[3n] (kcal/mol/A), the virial
 * vir[6] = sum r (x) f (xx,yy,zz,xy,xz,yz; kcal/mol) and the potential energy for the positions x in the box given; `call` is 0
 * for the set-up evaluation of a run (omd_setup) and counts the evaluations after it.  With it set, omd_run / omd_eval are the
 * integrator, thermostat, fix deform (with flips) and pressure average of this file around those forces: no lists, no k-space,
 * no bonded terms of this file; SHAKE only if the system was created with constraints. */
typedef void (*omd_force_fn)(void *ctx, int call, int natoms, const double box[9], const double *x, double *f, double vir[6], double *energy);

typedef struct omd_sim omd_sim;

/* All index arrays are 0-based atom indices; type arrays are 0-based type indices.
 * angle/improper equilibrium values are in radians.  eps/sigma are full ntypes x ntypes
 * matrices (the restart file carries mixed coefficients). */
omd_sim *omd_create(int natoms, int ntypes, const int *type, const double *charge,
                    const double *mass_per_type, const double *eps, const double *sigma,
                    int nbonds, const int *bond_atoms, const int *bond_type, int nbondtypes,
                    const double *bond_coeff /* K,r0 */,
                    int nangles, const int *angle_atoms, const int *angle_type, int nangletypes,
                    const double *angle_coeff /* K,theta0 */,
                    int ndihedrals, const int *dihedral_atoms, const int *dihedral_type,
                    int ndihedraltypes, const double *dihedral_coeff /* K1..K4 */,
                    int nimpropers, const int *improper_atoms, const int *improper_type,
                    int nimpropertypes, const double *improper_coeff /* K,chi0 */,
                    const double special_lj[3], const double special_coul[3],
                    const omd_params *params);
void omd_destroy(omd_sim *s);

/* box = {xlo,ylo,zlo, xhi,yhi,zhi, xy,xz,yz}; x,v are [natoms*3] */
void omd_set_state(omd_sim *s, const double box[9], const double *x, const double *v);
void omd_get_state(const omd_sim *s, double box[9], double *x, double *v);

int omd_natoms(const omd_sim *s);
int omd_nconstraints(const omd_sim *s);
int omd_nclusters(const omd_sim *s);
double omd_tdof(const omd_sim *s);
double omd_g_ewald(const omd_sim *s);
int omd_nkvec(const omd_sim *s);
void omd_pppm_grid(const omd_sim *s, int n[3]);   /* PPPM grid of the last setup (0,0,0 with the Ewald sum) */
int omd_npairs(const omd_sim *s);   /* unique pairs currently in the neighbour list */
int omd_nflips(const omd_sim *s);   /* triclinic box flips applied so far (fix deform, default flip yes) */
/* fix deform's tilt rules (LAMMPS 17Nov16 fix_deform.cpp end_of_step), exposed for golden tests: tilt = xy, xz, yz */
void omd_tilt_closest(double tilt[3], double xprd_new, double yprd_new, double xy, double xz, double yz, double xprd, double yprd);
int omd_tilt_flip(const double tilt[3], double xprd, double yprd, double flipped[3], int nflip[3]);

/* (Re)initialise run-level quantities exactly as a fresh LAMMPS "run" does:
 * g_ewald and the k-vector set from the current box, neighbour list rebuild.
 * use_shake selects whether SHAKE'd bonds are removed from the bond list. */
void omd_setup(omd_sim *s, int use_shake);
void omd_set_external_force(omd_sim *s, omd_force_fn fn, void *ctx);
/* test hook: keep g_ewald and the k-vector set of the last setup (for d/d(strain) tests) */
void omd_freeze_kspace(omd_sim *s, int frozen);

/* Static evaluation at the current state (after omd_setup): forces [natoms*3],
 * energies[OMD_NPART], virials[OMD_NPART*6] in order xx,yy,zz,xy,xz,yz (kcal/mol).
 * SHAKE part is left zero. */
void omd_compute(omd_sim *s, double *f, double *energies, double *virials);

/* kinetic tensor (kcal/mol, order xx,yy,zz,xy,xz,yz) and temperature */
double omd_temperature(const omd_sim *s, double ke_tensor[6]);

/* One "run N" of velocity-Verlet with fix shake + fix nvt (+ fix deform if rates != NULL)
 * (+ fix ave/time of the pressure tensor if press_avg != NULL).  Restates the per-step
 * order of SURVEY.md A.2.  rates = engineering strain rates {xx,yy,zz,xy,xz,yz} in 1/fs.
 * nvt: 0 = NVE (no thermostat), 1 = Nose-Hoover chain.
 * If trace != NULL it receives per step 8 doubles: T, pe, ke, nh_energy, vol, pxx,pyy,pzz. */
int omd_run(omd_sim *s, int nsteps, double dt, double temperature, int nvt, int use_shake,
            const double *rates, double *press_avg /* 6, atm */, double *trace);

/* Full stress evaluation = STMDProblem::lammps_straining (stmd_problem.h:84-383) for OPLS:
 * strain_len = MDSim.strain in deal.II raw order xx,yy,zz,xy,xz,yz (Angstrom, i.e. strain x
 * init_length, stmd_sync.h:552-557).  stress_out in Pa, same order.  Returns nts. */
int omd_eval(omd_sim *s, const double strain_len[6], double timestep_length, double temperature,
             double strain_rate, int nsteps_sample, double stress_out[6]);

/* host arithmetic of stmd_problem.h:221-244, exposed for golden tests */
int omd_nts(const double true_strain[6], double strain_rate, double dt);
double omd_round_rate(double rate);      /* "%.6e" round trip, stmd_problem.h:241 */
double omd_round_f(double v);            /* "%f"   round trip, stmd_problem.h:164,235 */

/* ---- init_material's equilibration schedule (lammps_scripts_opls/in.init.lammps), see md_oracle.c ---- */
/* min_style sd + minimize etol ftol maxiter maxeval; returns the stop reason (0 etol, 1 ftol, 2 maxiter, 3 maxeval, 4 line search);
 * info[4] = iterations, force evaluations, initial and final potential energy */
int omd_minimize(omd_sim *s, double etol, double ftol, int maxiter, int maxeval, double *info);
/* velocity all create T seed rot yes dist gaussian (own random stream, see md_oracle.c) */
void omd_velocity_create(omd_sim *s, double temperature, unsigned long long seed);
/* change_box all x final 0 lx y final 0 ly z final 0 lz remap */
void omd_change_box(omd_sim *s, const double len[3]);
/* run N under fix nvt (npt 0) or fix npt temp t_start t_stop tperiod iso p p pperiod (npt 1), no SHAKE; lavg[3]: running
 * average of the box lengths over the two half-run windows; trace: 6 doubles per step (T, pe, ke, thermostat + barostat
 * part of the conserved quantity, volume, scalar pressure in atm) */
int omd_run_nh(omd_sim *s, int nsteps, double dt, double t_start, double t_stop, int npt, double p_target, double p_period,
               double *lavg, double *trace);
/* the whole schedule of in.init.lammps:44-215 in units of nsinit steps; lengths[3] = final box lengths */
int omd_equilibrate(omd_sim *s, int nsinit, double dt, double tempt, unsigned long long seed, double lengths[3], double *min_info);

/* timing of the last omd_eval: seconds spent in pair / kspace / neigh / other */
void omd_last_timing(const omd_sim *s, double t[4]);

#ifdef __cplusplus
}
#endif
#endif
