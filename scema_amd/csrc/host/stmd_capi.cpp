// stmd_capi.cpp -- extern "C" face of the host layer (include/scema_stmd.h)
#include <cstring>
#include <new>
#include <vector>

#include "eqmd_problem.h"
#include "stmd_sync.h"

struct scema_stmd {
  scema::STMDSync sync;
  std::string err;
  scema_stmd(scema_md_engine *e, int r, int w, scema_allgather_fn ag, void *ctx) : sync(e, r, w, ag, ctx) {}
};

extern "C" {

int scema_stmd_create(scema_md_engine *engine, int32_t rank, int32_t world, scema_allgather_fn allgather, void *ctx, scema_stmd **out) {
  if (!out || world <= 0 || rank < 0 || rank >= world) return SCEMA_MD_ERR_ARG;
  *out = new (std::nothrow) scema_stmd(engine, rank, world, allgather, ctx);
  return *out ? SCEMA_MD_OK : SCEMA_MD_ERR_ARG;
}
void scema_stmd_destroy(scema_stmd *s) { delete s; }
const char *scema_stmd_last_error(const scema_stmd *s) { return s ? s->sync.last_error().c_str() : "null handle"; }

// reference stmd_sync.h:189-278
int scema_stmd_set_md_procs(int32_t nmdruns, int32_t n_processes, int32_t this_process, int32_t min_cores, int32_t cores_per_node,
                            int32_t *md_batch_n_processes, int32_t *n_md_batches, int32_t *md_batch_pcolor) {
  if (n_processes <= 0 || cores_per_node <= 0 || min_cores <= 0 || this_process < 0) return SCEMA_MD_ERR_ARG;
  const unsigned npbtch_min = (unsigned)min_cores, npnode = (unsigned)cores_per_node, P = (unsigned)n_processes;
  unsigned fair_npbtch;
  if (nmdruns > 0) {
    fair_npbtch = (unsigned)(n_processes / nmdruns);
    if (fair_npbtch == 0) fair_npbtch = 1;
  } else {
    fair_npbtch = P;
  }
  // admissible core counts: factors or multiples of the core count per node, from the minimum allocation to all
  std::vector<unsigned> list_possible_cores_per_job;
  for (unsigned ic = npbtch_min; ic <= P; ic++) {
    if (ic <= npnode) {
      if (npnode % ic == 0) list_possible_cores_per_job.push_back(ic);
    } else if (ic % npnode == 0) {
      list_possible_cores_per_job.push_back(ic);
    }
  }
  unsigned nb = 0;   // the reference's member is left as it was when nothing is admissible; here that is an error below
  for (unsigned ic = 0; ic < list_possible_cores_per_job.size(); ic++) {
    if (list_possible_cores_per_job[ic] > fair_npbtch) break;
    nb = list_possible_cores_per_job[ic];
  }
  if (nb < npbtch_min || nb > P || (npnode % nb != 0 && nb % npnode != 0)) return SCEMA_MD_ERR_ARG;
  int nbatches = (int)(P / nb);
  if (nbatches == 0) { nbatches = 1; nb = P; }
  int colour = -1;   // MPI_UNDEFINED
  if ((unsigned)this_process < nb * (unsigned)nbatches) colour = (int)((unsigned)this_process / nb);
  if (md_batch_n_processes) *md_batch_n_processes = (int32_t)nb;
  if (n_md_batches) *n_md_batches = nbatches;
  if (md_batch_pcolor) *md_batch_pcolor = colour;
  return SCEMA_MD_OK;
}

int scema_stmd_set_lammps_state_files(scema_stmd *s, int32_t on) {
  if (!s) return SCEMA_MD_ERR_ARG;
  s->sync.set_lammps_state_files(on != 0);
  return SCEMA_MD_OK;
}

int scema_stmd_init(scema_stmd *s, const scema_stmd_config *cfg) {
  if (!s || !cfg) return SCEMA_MD_ERR_ARG;
  return s->sync.init(*cfg);
}

int scema_stmd_update(scema_stmd *s, int32_t timestep, double present_time, int32_t newtonstep, scema_qp *update_list, int32_t n_qp) {
  if (!s || (n_qp > 0 && !update_list) || n_qp < 0) return SCEMA_MD_ERR_ARG;
  scema::ScaleBridgingData sbd;
  sbd.update_list.assign(update_list, update_list + n_qp);
  int rc = s->sync.update(timestep, present_time, newtonstep, sbd);
  if (rc) return rc;
  for (int i = 0; i < n_qp; i++) std::memcpy(update_list[i].update_stress, sbd.update_list[i].update_stress, 6 * sizeof(double));
  return SCEMA_MD_OK;
}

int scema_stmd_replica_data(const scema_stmd *s, int32_t material, int32_t replica0, double *init_length, double *init_stress, double *rotam,
                            double *rho) {
  if (!s) return SCEMA_MD_ERR_ARG;
  const auto &reps = s->sync.replicas();
  const size_t idx = (size_t)material * s->sync.nreplicas() + replica0;
  if (material < 0 || replica0 < 0 || idx >= reps.size()) return SCEMA_MD_ERR_ARG;
  const scema::ReplicaData &r = reps[idx];
  if (init_length) for (int i = 0; i < 3; i++) init_length[i] = r.init_length[i];
  if (init_stress) for (int i = 0; i < 6; i++) init_stress[i] = r.init_stress.raw[i];
  if (rotam) for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) rotam[3 * i + j] = r.rotam.m[i][j];
  if (rho) *rho = r.rho;
  return SCEMA_MD_OK;
}

int scema_eqmd_equil(scema_md_engine *engine, const char *cmat, const char *lengthof, const char *stressof, const char *stiffof,
                     int32_t rep, double mdts, double mdtem, int32_t mdnss, double mdss, double mdsa, const char *mdff, char *errbuf,
                     int32_t errlen) {
  if (!cmat || !lengthof || !stressof || !stiffof || !mdff) return SCEMA_MD_ERR_ARG;
  scema::EQMDProblem eq(engine);
  const int rc = eq.equil(cmat, lengthof, stressof, stiffof, rep, mdts, mdtem, mdnss, mdss, mdsa, mdff);
  if (rc && errbuf && errlen > 0) {
    std::strncpy(errbuf, eq.last_error().c_str(), (size_t)errlen - 1);
    errbuf[errlen - 1] = 0;
  }
  return rc;
}

int scema_eqmd_equil_full(scema_md_engine *engine, const char *cmat, const char *slocin, const char *qplogloc, const char *scrloc, const char *lengthof,
                          const char *stressof, const char *stiffof, const char *systof, int32_t rep, double mdts, double mdtem, int32_t mdnss,
                          int32_t mdnse, double mdss, double mdsa, const char *mdff, char *errbuf, int32_t errlen) {
  if (!cmat || !slocin || !lengthof || !stressof || !stiffof || !systof || !mdff) return SCEMA_MD_ERR_ARG;
  scema::EQMDProblem eq(engine);
  const int rc = eq.equil(cmat, slocin, qplogloc ? qplogloc : "", scrloc ? scrloc : "", lengthof, stressof, stiffof, systof, rep, mdts, mdtem, mdnss, mdnse,
                          mdss, mdsa, mdff);
  if (rc && errbuf && errlen > 0) {
    std::strncpy(errbuf, eq.last_error().c_str(), (size_t)errlen - 1);
    errbuf[errlen - 1] = 0;
  }
  return rc;
}

}  // extern "C"
