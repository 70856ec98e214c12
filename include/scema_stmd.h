/*
 * scema_stmd.h -- C ABI of the host layer that mirrors SCEMa's MD batch scheduler
 * HMM::STMDSync<3> (reference headers/stmd_sync.h:53-156): init() loads the replica metadata,
 * update() turns the FE side's update_list of strains into stresses.  A SCEMa maintainer binds
 * these two calls where dealammps.cc:455 calls mmd_problem->update.
 */
#ifndef SCEMA_STMD_H
#define SCEMA_STMD_H

#include <stdint.h>

#include "scema_md.h"

#ifdef __cplusplus
extern "C" {
#endif

/* HMM::QP (reference headers/scale_bridging_data.h:12-19): 112 bytes, sent as raw bytes */
typedef struct {
  int32_t id;
  int32_t most_recent_id;
  int32_t material;
  double update_strain[6]; /* raw order xx,yy,zz,xy,xz,yz */
  double update_stress[6];
} scema_qp;

/* the arguments of STMDSync::init (reference stmd_sync.h:1023-1068) */
typedef struct {
  int32_t start_timestep;
  double md_timestep_length;       /* fs  */
  double md_temperature;           /* K   */
  int32_t md_nsteps_sample;
  double md_strain_rate;           /* 1/fs */
  const char *md_force_field;      /* "opls" | "reax" */
  const char *nanostatelocin;      /* nanoscale input   */
  const char *nanostatelocout;     /* nanoscale output  */
  const char *nanostatelocres;     /* nanoscale restart */
  const char *nanologloc;          /* nanoscale log ("none" allowed) */
  const char *macrostatelocout;    /* macroscale output */
  const char *md_scripts_directory;
  int32_t freq_checkpoint;
  int32_t freq_output_homog;
  int32_t n_materials;
  const char *const *mdtype;       /* material names */
  double cg_dir[3];                /* rotation common ground vector */
  int32_t nrepl;
  int32_t use_pjm_scheduler;       /* must be 0 (external scheduler branch is out of scope) */
  int32_t approx_md_with_hookes_law;
  int32_t verbose;                 /* print the reference's progress markers */
} scema_stmd_config;

/* all-gather supplied by the host program (MPI+RCCL in SCEMa, torch.distributed in bench.py):
 * every rank contributes count_per_rank doubles -- the engine's device buffer
 * scema_md_local_stress_device_ptr(), or `local_host` when there is no engine (Hooke test mode) --
 * and must fill gathered_host[world * count_per_rank].  Return 0 on success. */
typedef int (*scema_allgather_fn)(void *ctx, scema_md_engine *engine, const double *local_host, int32_t count_per_rank,
                                  double *gathered_host);

typedef struct scema_stmd scema_stmd;

/* engine may be NULL only when approx_md_with_hookes_law is set (the reference's fake backend) */
int scema_stmd_create(scema_md_engine *engine, int32_t rank, int32_t world, scema_allgather_fn allgather, void *ctx,
                      scema_stmd **out);
void scema_stmd_destroy(scema_stmd *s);
const char *scema_stmd_last_error(const scema_stmd *s);
/* on != 0: write last.<qp>.<mat>_<rep>.dump after every evaluation and the lcts.* checkpoints in LAMMPS' 17Nov16 binary
 * restart layout, as stmd_problem.h:258,268 do, so that a LAMMPS-based SCEMa run can take the simulations over (restart
 * files of either kind are read back by init).  Default off: states stay in HBM, checkpoints in the engine's container. */
int scema_stmd_set_lammps_state_files(scema_stmd *s, int32_t on);
/* STMDSync::init (stmd_sync.h:1023) */
int scema_stmd_init(scema_stmd *s, const scema_stmd_config *cfg);
/* STMDSync::update (stmd_sync.h:1070): update_list[i].update_strain in, .update_stress out (on every rank) */
int scema_stmd_update(scema_stmd *s, int32_t timestep, double present_time, int32_t newtonstep, scema_qp *update_list,
                      int32_t n_qp);
/* replica metadata loaded by init, for inspection: init_length[3], init_stress[6] raw, rotam[9], rho */
int scema_stmd_replica_data(const scema_stmd *s, int32_t material, int32_t replica0, double *init_length, double *init_stress,
                            double *rotam, double *rho);

/* STMDSync::set_md_procs (stmd_sync.h:189-278), the arithmetic alone: how the reference cuts `n_processes` MD ranks into
 * batches for `nmdruns` simulations -- ranks per batch = the largest admissible count (a factor or a multiple of
 * `cores_per_node`, between `min_cores` and n_processes) not above the fair share n_processes / nmdruns; number of
 * batches; colour of `this_process` (-1 = MPI_UNDEFINED: left over).  At this boundary a rank is a GPU and a simulation
 * never spans GPUs, so with fewer simulations than GPUs the surplus ranks of a batch idle; with nmdruns >= n_processes
 * the result is one rank per batch, n_processes batches (the case the engine's planner serves).  Returns
 * SCEMA_MD_ERR_ARG where the reference prints "md_batch_n_processes is not well set" and exits. */
int scema_stmd_set_md_procs(int32_t nmdruns, int32_t n_processes, int32_t this_process, int32_t min_cores, int32_t cores_per_node,
                            int32_t *md_batch_n_processes, int32_t *n_md_batches, int32_t *md_batch_pcolor);

/* EQMDProblem<3>::equil (init_material_problem.h:309-355) for a replica that is already equilibrated and registered
 * with the engine: computes box lengths, initial stress and stiffness on the GPU and writes
 * lengthof / stressof / stiffof = init.<mat>_<rep>.{length,stress,stiff}, the files scema_stmd_init reads.
 * Unknown force field -> SCEMA_MD_ERR_ARG (the reference prints and exits).  errbuf receives the message. */
int scema_eqmd_equil(scema_md_engine *engine, const char *cmat, const char *lengthof, const char *stressof, const char *stiffof,
                     int32_t rep, double mdts, double mdtem, int32_t mdnss, double mdss, double mdsa, const char *mdff, char *errbuf,
                     int32_t errlen);

/* EQMDProblem<3>::equil with the reference's whole argument list (init_material_problem.h:309-315): when systof
 * (init.<mat>_<rep>.bin) does not exist yet, the replica is read from <slocin>/<cmat>_<rep>.data (unless it is registered
 * already), equilibrated by the schedule of in.init.lammps on the GPU (scema_md_equilibrate; mdnse = nsinit) and written to
 * systof; then as scema_eqmd_equil.  qplogloc / scrloc (LAMMPS log and script folders) are accepted and unused. */
int scema_eqmd_equil_full(scema_md_engine *engine, const char *cmat, const char *slocin, const char *qplogloc, const char *scrloc,
                          const char *lengthof, const char *stressof, const char *stiffof, const char *systof, int32_t rep, double mdts,
                          double mdtem, int32_t mdnss, int32_t mdnse, double mdss, double mdsa, const char *mdff, char *errbuf,
                          int32_t errlen);

#ifdef __cplusplus
}
#endif
#endif
