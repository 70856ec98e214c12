/*
 * scema_md.h -- C ABI of the MI355X-native microscale stress-evaluation engine that drops into
 * SCEMa's STMDProblem / STMDSync path.
 *
 * Plain C: pointers and sizes only, no torch / HIP / C++ types.  Each entry point names the
 * reference interface it replaces (paths relative to the SCEMa tree).
 *
 * Tensor conventions (reference): rank-2 symmetric tensors travel as 6 doubles in deal.II raw
 * order xx,yy,zz,xy,xz,yz (headers/scale_bridging_data.h:12-19, FE_problem.h:1344-1346); rank-4
 * stiffness as 36 doubles in init.*.stiff file order (headers/read_write.h:149-171).
 */
#ifndef SCEMA_MD_H
#define SCEMA_MD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCEMA_MD_OK 0
#define SCEMA_MD_ERR_ARG 1        /* bad argument / unknown force field (stmd_problem.h:462-467) */
#define SCEMA_MD_ERR_NOSTATE 2    /* required replica / state missing (stmd_problem.h:125-132 asserts) */
#define SCEMA_MD_ERR_DEVICE 3     /* HIP runtime failure */
#define SCEMA_MD_ERR_BOX 4        /* box smaller than 2*(cutoff+skin) */
#define SCEMA_MD_ERR_IO 5
#define SCEMA_MD_ERR_OVERFLOW 6   /* neighbour capacity exceeded even after regrowth */

#define SCEMA_MD_NPART 8          /* lj, coul, bond, angle, dihedral, improper, kspace, shake */
#define SCEMA_MD_QP_NONE ((int32_t)0xFFFFFFFF) /* most_recent_qp_id "none": uint32 max as int, stmd_problem.h:126 */

typedef struct scema_md_engine scema_md_engine;

/* Run-level settings = what the reference's LAMMPS scripts fix for the OPLS path
 * (lammps_scripts_opls/in.set.lammps:27-40, in.strain.lammps:71,80). */
typedef struct {
  double cut_lj;           /* 12.0  pair_style lj/cut/coul/long 12.0 9.0 */
  double cut_coul;         /*  9.0 */
  double skin;             /*  2.0  neighbor 2.0 bin */
  int32_t neigh_delay;     /*  5    neigh_modify every 1 delay 5 check yes */
  double kspace_accuracy;  /* 1e-4  kspace_style pppm 0.0001 (sets g_ewald; reciprocal sum: see kspace_style) */
  double shake_tol;        /* 1e-3  fix shake 0.001 20 1000 m 1.0 */
  int32_t shake_maxiter;   /* 20 */
  double shake_mass;       /* 1.0 ; <= 0 disables SHAKE */
  double t_period;         /* 100.0 fix nvt temp T T 100.0 */
  int32_t t_chain;         /* 3 */
  int32_t device;          /* HIP device ordinal */
  int32_t max_batch;       /* simulations advanced together per launch group; 0 = all that fit */
  int32_t profile;         /* !=0: HIP-event timing of every pair-kernel launch (scema_md_get_profile) */
  int32_t kspace_style;    /* 1 (default): PPPM as `kspace_style pppm 0.0001` asks for (in.set.lammps:36; order 5, ik differentiation,
                            * hipFFT; grid and g_ewald by the rules of pppm.cpp as restated in scema_amd/csrc/engine/engine_kspace.cpp / oracle/md_oracle.c;
                            * DESIGN.md 5c); 0: the reciprocal part as the plain Ewald sum PPPM approximates, at kspace_accuracy
                            * (LAMMPS' initial g_ewald estimate, k-vectors by the RMS criterion of kspace_style ewald) */
} scema_md_params;

void scema_md_default_params(scema_md_params *p);

/* Content of one equilibrated replica = what init.<mat>_<rep>.bin carries in the reference
 * (stmd_problem.h:99-100, stmd_sync.h:388).  0-based atom and type indices; angles in radians;
 * eps/sigma are ntypes x ntypes (already mixed).  box = xlo,ylo,zlo,xhi,yhi,zhi,xy,xz,yz. */
typedef struct {
  int32_t natoms, ntypes;
  const int32_t *type;
  const double *charge;
  const double *mass;      /* per type */
  const double *eps, *sigma;
  int32_t nbonds, nbondtypes;
  const int32_t *bond_atoms, *bond_type;
  const double *bond_coeff;      /* K, r0 */
  int32_t nangles, nangletypes;
  const int32_t *angle_atoms, *angle_type;
  const double *angle_coeff;     /* K, theta0 */
  int32_t ndihedrals, ndihedraltypes;
  const int32_t *dihedral_atoms, *dihedral_type;
  const double *dihedral_coeff;  /* K1..K4 (dihedral_style opls) */
  int32_t nimpropers, nimpropertypes;
  const int32_t *improper_atoms, *improper_type;
  const double *improper_coeff;  /* K, chi0 */
  double special_lj[3], special_coul[3]; /* in.init.lammps:31 -> 0 0 1 */
  double box[9];
  const double *x, *v;           /* [natoms*3] */
} scema_md_system;

/* Mirrors HMM::MDSim<3> (headers/md_sim.h:15-58). */
typedef struct {
  int32_t qp_id, most_recent_qp_id, replica /* 1-based */, material;
  const char *matid, *time_id, *output_folder, *restart_folder, *scripts_folder, *log_file, *force_field;
  double strain[6];      /* IN  raw order; Angstrom (strain x init_length, stmd_sync.h:552-557) */
  double stiffness[36];  /* IN  Hooke mode only, file order */
  double timestep_length, temperature, strain_rate;
  int32_t nsteps_sample;
  int32_t output_homog, checkpoint;
  double stress[6];      /* OUT Pa, raw order: -<P> * 101325 (stmd_problem.h:335-341) */
  int32_t stress_updated;/* OUT */
} scema_mdsim;

/* ---- engine lifetime ---- */
int scema_md_create(const scema_md_params *p, scema_md_engine **out);
void scema_md_destroy(scema_md_engine *e);
const char *scema_md_last_error(const scema_md_engine *e);

/* ---- replica registry: replaces "read_restart init.<mat>_<rep>.bin" (stmd_problem.h:204) ---- */
int scema_md_register_replica(scema_md_engine *e, const char *matid, int32_t replica, const scema_md_system *sys);
/* atoms of a registered replica, 0 if (matid, replica) is not registered */
int32_t scema_md_replica_natoms(scema_md_engine *e, const char *matid, int32_t replica);
/* reads a replica file written by scema_md_write_replica_file (our container for init.*.bin) */
int scema_md_load_replica_file(scema_md_engine *e, const char *matid, int32_t replica, const char *path);
int scema_md_write_replica_file(const char *path, const scema_md_system *sys);
/* a registered replica with its current initial state (e.g. after scema_md_equilibrate) as such a file: what
 * `write_restart init.<mat>_<rep>.bin` keeps in the reference (init_material_problem.h:208-210) */
int scema_md_save_replica_file(scema_md_engine *e, const char *matid, int32_t replica, const char *path);
/* LAMMPS text data file of atom_style full (`write_data` of a replica equilibrated with the reference's
 * in.init.lammps): register it directly, or convert it to the replica container.  special_bonds weights are
 * not stored in data files; NULL means the reference's "lj/coul 0 0 1" (in.init.lammps:31). */
int scema_md_load_lammps_data(scema_md_engine *e, const char *matid, int32_t replica, const char *path,
                              const double special_lj[3], const double special_coul[3]);
int scema_md_convert_lammps_data(const char *data_path, const char *replica_path, const double special_lj[3],
                                 const double special_coul[3]);

/* LAMMPS binary restart files in the layout of the version the reference pins (LAMMPS 17Nov16, reference README.md:31-37):
 * the `init.<mat>_<rep>.bin` that stmd_problem.h:204 reads and the `last.*` / `lcts.*` states that stmd_problem.h:258,268
 * write.  probe: header of any restart file.  read_atoms: tag, type, image flags (3 per atom), wrapped x, v in file order
 * (atom_style atomic or full; NULL outputs are skipped).  load / convert: atom_style full + pair lj/cut/coul/long +
 * bond/angle harmonic + dihedral opls + improper harmonic, units real -> replica (compare info.cut_lj / cut_coul with the
 * engine's parameters: the engine takes its cutoffs from scema_md_params).  write: one-process restart of a replica in
 * the same layout. */
typedef struct {
  char version[32], units[16], atom_style[32], pair_style[64];
  int64_t natoms, ntimestep, nbonds, nangles, ndihedrals, nimpropers;
  int32_t ntypes, nbondtypes, nangletypes, ndihedraltypes, nimpropertypes, triclinic, nprocs, reserved;
  double box[9];            /* xlo,ylo,zlo,xhi,yhi,zhi,xy,xz,yz */
  double timestep;
  double special_lj[3], special_coul[3];
  double cut_lj, cut_coul;  /* pair lj/cut/coul/long settings, 0 if the file has no such block */
  double mass[16];          /* first 16 types */
  char error[160];
} scema_lammps_restart_info;
int scema_md_probe_lammps_restart(const char *path, scema_lammps_restart_info *info);
int scema_md_read_lammps_restart_atoms(const char *path, int64_t capacity, int64_t *tag, int32_t *type, int32_t *image,
                                       double *x, double *v);
int scema_md_load_lammps_restart(scema_md_engine *e, const char *matid, int32_t replica, const char *path);
int scema_md_convert_lammps_restart(const char *restart_path, const char *replica_path);
int scema_md_write_lammps_restart(const char *path, const scema_md_system *sys, double cut_lj, double cut_coul,
                                  double timestep, int64_t ntimestep);

/* ---- the hot path ---- */
/* Replaces STMDProblem<3>::strain (stmd_problem.h:458-496) for the whole vector that
 * STMDSync::execute_inside_md_simulations iterates (stmd_sync.h:570-618).  Every rank passes the SAME vector.
 * Which rank runs simulation i is decided by the planner of scema_amd/csrc/host/sim_plan.h (identical on every rank):
 * a fresh balanced batch gives the reference's round robin i % world (stmd_sync.h:583); afterwards a simulation runs
 * where the state it continues from lives (a state is resident in ONE GPU's HBM, the update_list changes from step to
 * step, FE_problem.h:1330-1350), and ragged batches are levelled by MD steps (nts + nss).
 * State branch rule of stmd_problem.h:116-138,185-207: load from most_recent_qp_id if != qp_id, else qp_id, else the
 * registered init state; always store under qp_id.  hooke != 0: sigma = C:eps (stmd_problem.h:479-483).
 *   - no communicator attached (world may still be > 1): only this rank's share is evaluated, the others are left
 *     untouched (stress_updated = 0) and the caller gathers (scema_md_copy_local_stress + scema_md_scatter_gathered:
 *     the result buffer carries this rank's status word, so the caller enters its collective even after a failed call);
 *     a request whose source state lives on another rank is an error (SCEMA_MD_ERR_NOSTATE);
 *   - communicator attached (scema_md_comm_init_*): the call is collective -- the ranks first agree (a 16-byte
 *     handshake per rank: status of the local pre-checks + hash of the plan each rank computed; a mismatch is an error
 *     on every rank instead of a hang), states that have to change GPU travel, then ONE all-gather returns every stress
 *     to every rank (replaces STMDSync::share_stresses, stmd_sync.h:620-726): on return every sims[i].stress is set on
 *     every rank.  The all-gather carries a status word per rank: a share that failed on one rank (list overflow after
 *     regrowth, a replica that blew up, a missing state) makes the call fail on EVERY rank, none is left waiting.
 * Requests are validated on every rank before anything is planned, so a bad request is refused by all ranks alike.
 * A failed call leaves the state store and the owner directory as it found them -- ON EVERY RANK when a communicator is
 * attached (the ranks learn of each other's failure inside the call and all roll back): states that had already been
 * advanced are put back from their backups (the reference stops the whole run at this point, exit(1)).  Without a
 * communicator a rank whose own share succeeded learns of another rank's failure only in the caller's collective: its share
 * waits for that (scema_md_settle_update below) and is taken back there.
 * What the ranks must share for their plans to agree: the same replicas registered, the same request vectors, and every
 * call that edits the state store (set_state, drop_state, load_state_file, equilibrate) made on every rank. */
int scema_md_strain_batch(scema_md_engine *e, scema_mdsim *sims, int32_t n_sims, int32_t hooke,
                          int32_t rank, int32_t world);
/* literal per-simulation drop-in */
int scema_md_strain(scema_md_engine *e, scema_mdsim *sim, int32_t hooke);

/* ---- multi-GPU: one process per GPU, the engine owns the collective ----
 * RCCL over xGMI: rank 0 calls scema_md_comm_unique_id and the host program broadcasts the 128 bytes (MPI_Bcast in
 * SCEMa, any store elsewhere); every rank then calls scema_md_comm_init_rccl (ncclCommInitRank on the engine's
 * device).  Replaces the MPI traffic of stmd_sync.h:620-726 (per-simulation MPI_Isend/Recv of 6 doubles) by one
 * ncclAllGather per update, and the shared-file-system hand-over of last.<qp>.* states by ncclSend/ncclRecv. */
#define SCEMA_MD_COMM_ID_BYTES 128
int scema_md_comm_unique_id(void *id /* [SCEMA_MD_COMM_ID_BYTES] */);
int scema_md_comm_init_rccl(scema_md_engine *e, const void *id, int32_t rank, int32_t world);
/* Host transport instead (MPI of the host program; gloo in the CPU tests): buffers are host memory.
 * allgather: every rank contributes bytes_per_rank bytes, recv holds world * bytes_per_rank.  send/recv: blocking
 * point-to-point (may be NULL if states never have to move).  Return 0 on success. */
typedef int (*scema_md_host_allgather_fn)(void *ctx, const void *send, void *recv, int64_t bytes_per_rank);
typedef int (*scema_md_host_send_fn)(void *ctx, const void *buf, int64_t bytes, int32_t dst_rank);
typedef int (*scema_md_host_recv_fn)(void *ctx, void *buf, int64_t bytes, int32_t src_rank);
int scema_md_comm_init_host(scema_md_engine *e, int32_t rank, int32_t world, scema_md_host_allgather_fn allgather,
                            scema_md_host_send_fn send, scema_md_host_recv_fn recv, void *ctx);
void scema_md_comm_destroy(scema_md_engine *e);
int32_t scema_md_comm_world(const scema_md_engine *e);   /* 1 = no communicator */
int32_t scema_md_comm_rank(const scema_md_engine *e);
int scema_md_comm_stats(const scema_md_engine *e, int64_t *allgathers, int64_t *migrations);
int64_t scema_md_comm_handshakes(const scema_md_engine *e);   /* agreement handshakes so far (one per update when world > 1) */
/* rank recorded as the owner of a stored state, -1 = none recorded (held wherever it was set) */
int32_t scema_md_state_owner(const scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica);

/* Plan of the last scema_md_strain_batch: owner[i] = rank that ran simulation i, pos[i] = its slot in that rank's
 * result buffer, cap = slots per rank. */
int scema_md_last_plan(const scema_md_engine *e, int32_t n_sims, int32_t *owner, int32_t *pos, int32_t *cap);
/* Device-side result buffer of the last scema_md_strain_batch: 6*max(cap,1) doubles in HBM, simulation i of this rank
 * at offset 6*pos[i], followed by SCEMA_MD_RESULT_TRAILER words: the status of this rank's share (0, or the error code
 * scema_md_strain_batch returned) and the hash of the plan this rank computed.  The operand of the all-gather when the
 * caller runs the collective itself: scema_md_local_result_doubles() doubles per rank. */
#define SCEMA_MD_RESULT_TRAILER 2
void *scema_md_local_stress_device_ptr(scema_md_engine *e);
int32_t scema_md_local_stress_count(const scema_md_engine *e);   /* = cap */
int32_t scema_md_local_result_doubles(const scema_md_engine *e); /* = 6*max(cap,1) + SCEMA_MD_RESULT_TRAILER */
/* copy that buffer (all scema_md_local_result_doubles() of it) into caller memory (a device pointer, e.g. the send
 * buffer of the all-gather, or host) */
int scema_md_copy_local_stress(scema_md_engine *e, void *dst, int32_t dst_on_device);
/* after the caller's all-gather into gathered[world][scema_md_local_result_doubles()] (host memory): check every
 * rank's status word and plan hash (an error here is an error on every rank), then fill sims[i].stress /
 * stress_updated for every i by the plan of the last scema_md_strain_batch (rank-0 bookkeeping of stmd_sync.h:698-725) */
int scema_md_scatter_gathered(scema_md_engine *e, const double *gathered, scema_mdsim *sims, int32_t n_sims);
/* Without a communicator an update of a world of several ranks WAITS after scema_md_strain_batch (backups kept, owner directory
 * uncommitted) until the caller's collective has shown every rank's status word: scema_md_scatter_gathered settles it (an error
 * there takes this rank's share back too); a host that scatters the gathered buffer itself calls this with failed != 0 / 0.  The
 * next scema_md_strain_batch lets an unsettled update stand. */
int scema_md_settle_update(scema_md_engine *e, int32_t failed);
/* How many updates the NEXT call found unsettled and let stand (world > 1, no communicator attached, neither scema_md_scatter_gathered nor
 * scema_md_settle_update called in between): 0 in a host that follows the protocol; a warning is printed the first time. */
int64_t scema_md_unsettled_updates(const scema_md_engine *e);

/* The planner alone (scema_amd/csrc/host/sim_plan.h; pure host arithmetic, no GPU): owner/pos/cap as above, moves =
 * (simulation, from, to) triples of the states that would travel; cost NULL = equal cost; commit != 0 records the
 * owners for the next call, as a successful scema_md_strain_batch does. */
typedef struct scema_plan_dir scema_plan_dir;
scema_plan_dir *scema_plan_dir_create(void);
void scema_plan_dir_destroy(scema_plan_dir *d);
int scema_plan_update(scema_plan_dir *d, const scema_mdsim *sims, int32_t n_sims, const double *cost, int32_t world,
                      int32_t *owner, int32_t *pos, int32_t *cap, int32_t *moves /* 3 per move, room for n_sims */,
                      int32_t *n_moves, int32_t commit);

/* ---- persistent per-(qp,mat,replica) state: replaces last.<qp>.<mat>_<rep>.dump / lcts.* files
 * (stmd_problem.h:108-138,257-273; stmd_sync.h:167-187) ---- */
int scema_md_has_state(const scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica);
int scema_md_get_state(scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica,
                       double box[9], double *x, double *v);
int scema_md_set_state(scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica,
                       const double box[9], const double *x, const double *v);
int scema_md_drop_state(scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica);
int scema_md_save_state_file(scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica, const char *path);
/* load: the engine's own state container, or a LAMMPS 17Nov16 binary restart of the same replica as the reference writes
 * last.<qp>.* / lcts.<qp>.* (told apart by the magic string; atoms matched by tag, unwrapped with the image flags) */
int scema_md_load_state_file(scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica, const char *path);
/* the state as a LAMMPS 17Nov16 binary restart (what stmd_problem.h:258,268 write), for hand-over to a LAMMPS-based run */
int scema_md_save_state_lammps(scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica, const char *path,
                               double timestep, int64_t ntimestep);

/* the state as a LAMMPS text dump `id type xs ys zs vx vy vz ix iy iz` (what the reax branch of the reference writes as
 * last.<qp>.* / lcts.<qp>.*, stmd_problem.h:261-264,270-272, and re-reads with `rerun ... dump x y z vx vy vz ix iy iz box yes
 * scaled yes`, :190-194); read back by scema_md_load_state_file.  precise 0: LAMMPS' default "%g" columns (six digits, what the
 * reference's files hold); != 0: 17 digits (exact round trip). */
int scema_md_save_state_dump(scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica, const char *path,
                             int64_t ntimestep, int32_t precise);

/* ---- init_material: the quantities EQMDProblem::lammps_equilibration derives from an equilibrated
 * replica (init_material_problem.h:196-300): box lengths, initial stress (ELASTIC/in.homogenization.lammps: NVT + SHAKE
 * sampling) and the stiffness tensor by +-strain_ampl finite strains in the six directions (ELASTIC/in.modulus.lammps,
 * bi-displace.mod.lammps: fix deform ... delta over nsstrain steps, then nsteps_sample steps of sampling, fix nvt only).
 * The 13 runs are one batch on the GPU.  The equilibration schedule itself (in.init.lammps: minimise, heat, cool) is not
 * part of this call (scema_md_equilibrate below is): the registered replica is taken as the equilibrated state ("Reuse of
 * state data", :175-184).  With a ReaxFF force field selected (scema_md_reax_configure / scema_md_reax_activate) both calls run on
 * the ReaxFF force stage.
 * length[3]; stress[6] in Pa, file order 00,01,02,11,12,22; stiff[36] in Pa, file order of init.*.stiff. */
typedef struct {
  double timestep_length;  /* fs   "molecular dynamics parameters.timestep length" */
  double temperature;      /* K */
  int32_t nsteps_sample;   /* nssample0 = nssample */
  double strain_ampl;      /* "up": strain perturbation amplitude */
  double strain_rate;      /* 1/fs: nsstrain = ceil(up/(dt*rate)/10)*10 (init_material_problem.h:226) */
} scema_md_eqparams;
int scema_md_init_material(scema_md_engine *e, const char *matid, int32_t replica, const scema_md_eqparams *p,
                           double length[3], double stress[6], double stiff[36]);

/* ---- init_material, first part: the equilibration schedule EQMDProblem::lammps_equilibration runs when no state exists yet
 * (init_material_problem.h:167-174: in.init.lammps).  What lammps_scripts_opls/in.init.lammps:44-215 prescribes, on the GPU:
 * velocity create 200 K, min_style sd + minimize 1e-7 1e-11 nsinit 50000, then fix nvt 300 K (nsinit steps), fix npt iso 1 atm
 * 300->500 K (nsinit), 500 K (5 nsinit), 500->T (nsinit), T (2 nsinit, box lengths averaged, change_box to the averages), fix nvt
 * T (20 nsinit), fix npt T (2 nsinit, averaged, change_box), fix nvt T (nsinit); no SHAKE (commented out in the script).
 * The equilibrated state replaces the registered initial state of the replica (what write_restart init.<mat>_<rep>.bin
 * keeps, :208-210); length[3] = its box lengths (:196-208).  info (may be NULL) [5]: minimiser stop reason (0 energy
 * tolerance, 1 force tolerance, 2 iterations, 3 evaluations, 4 line search), iterations, force evaluations, initial and final
 * potential energy.  LAMMPS' random stream of `velocity create` is not reproduced (own generator, same seed semantics). */
typedef struct {
  int32_t nsteps_equil;    /* nsinit = "molecular dynamics parameters.number of equilibration steps" (init_material.cc:177) */
  double timestep_length;  /* fs */
  double temperature;      /* K, tempt */
  int64_t seed;            /* sseed; 0 = 1234 (init_material_problem.h:167) */
} scema_md_equilparams;
int scema_md_equilibrate(scema_md_engine *e, const char *matid, int32_t replica, const scema_md_equilparams *p, double length[3],
                         double *info);

/* ---- parity / measurement hooks ---- */
/* min_style sd + minimize on a stored state; info[5] as in scema_md_equilibrate */
int scema_md_debug_minimize(scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica, double etol, double ftol,
                            int32_t maxiter, int32_t maxeval, double *info);
/* one run under fix nvt (npt 0) or fix npt ... iso (npt 1) with the target ramped t_start -> t_stop, no SHAKE; lavg (may be
 * NULL) [3]: running average of the box lengths over the two half-run windows */
int scema_md_debug_run_nh(scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica, int32_t nsteps, double dt,
                          double t_start, double t_stop, int32_t npt, double p_target, double p_period, double *lavg);
/* Static evaluation at the stored state of (qp,mat,rep) (qp_id = SCEMA_MD_QP_NONE: the registered
 * init state): forces [natoms*3], energies[SCEMA_MD_NPART], virials[SCEMA_MD_NPART*6] (kcal/mol). */
int scema_md_debug_compute(scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica, int32_t use_shake,
                           double *f, double *energies, double *virials, double *info /* [8]: g_ewald,nk,npairs,tdof,... */);
/* One "run": nsteps of Verlet + SHAKE + NVT (+deform if rates) (+pressure average if press_avg). */
int scema_md_debug_run(scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica, int32_t nsteps,
                       double dt, double temperature, int32_t nvt, int32_t use_shake, const double *rates,
                       double *press_avg);

/* ---- ReaxFF replicas (md_force_field "reax": lammps_scripts/lammps_scripts_reax) ----
 * configure = what the reax scripts set up in LAMMPS before every run: `pair_style reax/c NULL safezone 50.0 mincap 100000`,
 * `pair_coeff * * <ffield> H C N O` (elements[k] = element of LAMMPS atom type k+1), `fix qeq/reax 1 0.0 10.0 <qeq_tol> reax/c`
 * (in.strain.lammps:10-12, ELASTIC/potential.mod.lammps:5-7); no SHAKE, no k-space, `neigh_modify every 1 delay 0 check no`.
 * qeq_tol <= 0: 1e-6.  skin < 0: default list skin (performance only: the pairs inside the 10 A taper radius do not depend
 * on it).  scema_md_strain_batch configures by itself from MDSim.scripts_folder + "/ffield.reax.2" and H C N O, as the
 * reference's script does, when a simulation asks for force_field "reax" and nothing was configured.  A replica for this
 * path is registered like any other (atom types, masses, box, x, v; no bonds; charges are equilibrated every step). */
int scema_md_reax_configure(scema_md_engine *e, const char *ffield_path, const char *const *elements, int32_t n_elements,
                            double qeq_tol, double skin);
/* which force field the parity hooks (scema_md_debug_run, scema_md_reax_debug_compute) use; strain_batch decides per call
 * from MDSim.force_field */
int scema_md_reax_activate(scema_md_engine *e, int32_t on);
/* parity switches (-1 keeps): exact_gradient 0 = drop the d(SBO)/d(Delta) term of the valence-angle energy for atoms with
 * vlpex >= 0, as USER-REAXC's Valence_Angles is remembered to do (unverifiable here, not energy conserving: DESIGN.md;
 * default 1 = the exact gradient); terms = bit mask of energy-term groups evaluated (31 = all) */
int scema_md_reax_set(scema_md_engine *e, int32_t exact_gradient, int32_t terms, int32_t qeq_maxiter);
/* How a ReaxFF batch is issued on the device -- results do not depend on it.  halves: the number of part batches, each on its own stream (default 2; batches of
   at least four replicas per part), 0 or 1 = one sequence of launches.  overlap: 1 = the bond-order chain of the force stage on a side stream next to the charge
   chain (default), 0 = one after the other.  -1 leaves a setting as it is.  A measurement aid: bench.py times the charge-equilibration sweep
   alone with both off.  (The reference has no counterpart: its LAMMPS ranks run one replica each, stmd_sync.h:583.) */
int scema_md_reax_concurrency(scema_md_engine *e, int32_t halves, int32_t overlap);
/* The same for the OPLS path: on = 1 (default) runs a launch group of 9 simulations and more as two to four part batches on streams of their own, 0 runs it whole
 * (one sequence of launches: what a kernel's time is with the chip to itself), -1 keeps the setting.  Results do not depend on it. */
int scema_md_batch_split(scema_md_engine *e, int32_t on);
/* out[3]: the current settings {batch split, ReaxFF part batches, ReaxFF overlap}, so that a caller that changes them for a measurement can
 * put back exactly what it found */
int scema_md_get_concurrency(const scema_md_engine *e, int32_t *out);
/* number of batched hipFFT plans the engine holds for PPPM meshes beyond the in-LDS solve (a plan owns a work area and is bound to a stream
 * when it runs: one per mesh size, batch count and stream of a part batch -- a test reads this), or a negative error code */
int scema_md_pppm_plan_count(const scema_md_engine *e);
/* static evaluation: f [natoms*3], eparts[13] (bond, lone pair, over, under, angle, penalty, 3-body conj., torsion, 4-body
 * conj., hydrogen bond, van der Waals, Coulomb, polarisation; kcal/mol), virial[6] (xx,yy,zz,xy,xz,yz), charges q[natoms],
 * info[6]: longest neighbour row, row capacity, bond-row capacity, CG iterations, image search (0 = minimum image), longest bond row */
int scema_md_reax_debug_compute(scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica, double *f, double *eparts,
                                double *virial, double *q, double *info);
/* out[7]: conjugate-gradient iterations, solves (two systems each), list skin, tolerance, solves finished by the single-workgroup
 * loop, iterations currently issued as batch launches per solve, evaluations repeated with the Jacobi preconditioner of fix qeq/reax
 * because the charge solve did not converge with the engine's own */
int scema_md_reax_stats(const scema_md_engine *e, double *out);

typedef struct {
  int64_t pair_launches;      /* timed pair-kernel launches */
  double pair_ms;             /* sum of their HIP-event durations */
  double pair_alg_bytes;      /* algorithmic bytes of those launches, SURVEY.md 8(d) formula */
  int64_t md_steps;           /* simulation-steps advanced (sum over simulations) */
  int64_t neigh_builds;
  double unique_pairs_per_sim;/* average unique pairs within cutoff+skin at the last build */
  int64_t evals;
  double list_skin_mean;      /* mean neighbour-list skin of those evaluations: params.skin + the adaptive extra (performance only) */
  int64_t pair_sims;          /* simulations summed over the timed pair launches (a large batch runs as two half batches) */
  int64_t box_flips;          /* triclinic box flips applied during straining runs (fix deform, default flip yes) */
  /* force_field "reax": the matrix sweep of the charge equilibration (k_rx_qeq_sweep), the HBM-bound kernel of that path */
  int64_t rx_sweep_launches;  /* timed launches */
  double rx_sweep_ms;         /* sum of their HIP-event durations */
  double rx_sweep_entries;    /* stored matrix entries those launches passed over (per replica: entries of its rows x sweeps it took
                               * part in); algorithmic bytes = (8 value + rx_sweep_col_bytes) per entry + 84 per row */
  double rx_sweep_rows;       /* rows likewise */
  int64_t rx_sweep_col_bytes; /* bytes of a stored column index beside the 8-byte value: 0 (replicas of up to 65 536 atoms: it rides in the value's word) or 4 */
  /* Launches on different streams may be in flight together (a batch runs as two half batches): the time during which AT LEAST ONE of the
   * timed launches ran -- the union of their HIP-event intervals on the device's common clock.  Equal to the sums above when nothing overlaps. */
  double pair_union_ms;
  double rx_sweep_union_ms;
  int64_t rx_sweep_symmetric; /* 1: the timed sweeps ran in the symmetric form (each pair of the matrix stored once, in its owner's row: rx_sweep_entries
                               * counts stored entries, half of the full rows'); 0: full rows */
} scema_md_profile;
int scema_md_get_profile(scema_md_engine *e, scema_md_profile *out, int32_t reset);

/* The library reads its environment through ONE table of declared switches (scema_amd/csrc/md_env.h, engine/engine_core.cpp:
 * performance A/B switches with measured defaults, test hooks, diagnostics; the reference has no counterpart -- its knobs are the
 * LAMMPS scripts).  This returns the declared switches that are set in the calling process's environment as "NAME=value" lines,
 * and their number: a reported run shows an empty list (bench.py: config.env_overrides). */
int scema_md_env_overrides(char *buf, int cap);

/* Calibration of the box a measurement ran on (no counterpart in the reference; measurement aid of bench.py): independent FP64 FMA
 * chains on every SIMD of the device, no memory traffic, ~0.1 s (a warm-up launch, then the best of three).  The boxes of a pool differ by several per cent under the pair
 * kernel's full-chip FP64 load; this figure (74-75 TFLOP/s on an MI355X at its sustained clock, 78.6 on the data sheet) lets
 * throughput values of different boxes be normalised.  Needs a HIP device; returns SCEMA_MD_ERR_DEVICE without one. */
int scema_md_box_fma_tflops(int32_t device, double *tflops);

/* The k-space set-up a run would use for this box, as a pure host function (no GPU, no engine): LAMMPS' initial g_ewald estimate
 * from the accuracy asked for (what `kspace_style ewald` keeps), and for kspace_style 1 the PPPM grid and the adjusted g_ewald by
 * PPPM::set_grid_global / adjust_gewald as restated in engine/engine_kspace.cpp (in.set.lammps:36 `kspace_style pppm 0.0001`).
 * grid = 0 0 0 for the Ewald sum or an uncharged system.  For checks against an independent restatement of those rules
 * (tests/test_oracle_pppm.py) and against LAMMPS' own log line "G vector ... grid = ..." (tools/export_lammps_case.py --run). */
int scema_md_kspace_setup(const scema_md_params *p, const double *box /* 9: lo[3] hi[3] xy xz yz */, double qsqsum /* sum q_i^2, e^2 */,
                          int32_t natoms, double *g_initial, double *g_ewald, int32_t *grid /* 3 */);

#ifdef __cplusplus
}
#endif
#endif
