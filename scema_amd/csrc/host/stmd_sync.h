// stmd_sync.h -- host mirror of HMM::STMDSync<3> (reference headers/stmd_sync.h:53-1132).
// Same member functions, same arithmetic, same file names; MPI is replaced by (rank, world) + one
// all-gather callback, the per-batch LAMMPS loop by one engine call for the whole vector.
#pragma once
#include <dirent.h>

#include <algorithm>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "../../../include/scema_stmd.h"
#include "math_calc.h"
#include "md_sim.h"
#include "read_write.h"
#include "scale_bridging_data.h"
#include "stmd_problem.h"

namespace scema {

// reference stmd_sync.h:41-51
struct ReplicaData {
  double rho = 0;
  std::string mat;
  int repl = 0;
  int nflakes = 0;
  Tensor1 init_length;
  Tensor2 rotam;
  SymmetricTensor2 init_stress;
  SymmetricTensor4 init_stiff;
};

class STMDSync {
 public:
  STMDSync(scema_md_engine *engine, int rank, int world, scema_allgather_fn ag, void *ctx)
      : engine_(engine), rank_(rank), world_(world), allgather_(ag), ag_ctx_(ctx) {}

  const std::string &last_error() const { return err_; }
  // last.* after every evaluation and lcts.* checkpoints in LAMMPS' binary restart layout (stmd_problem.h:258,268)
  void set_lammps_state_files(bool on) { lammps_state_files = on; }
  const std::vector<ReplicaData> &replicas() const { return replica_data; }
  unsigned nreplicas() const { return nrepl; }

  // reference stmd_sync.h:1023-1068
  int init(const scema_stmd_config &c) {
    approx_md_with_hookes_law = c.approx_md_with_hookes_law != 0;
    start_timestep = c.start_timestep;
    md_timestep_length = c.md_timestep_length;
    md_temperature = c.md_temperature;
    md_nsteps_sample = c.md_nsteps_sample;
    md_strain_rate = c.md_strain_rate;
    md_force_field = c.md_force_field ? c.md_force_field : "";
    nanostatelocin = c.nanostatelocin ? c.nanostatelocin : "";
    nanostatelocout = c.nanostatelocout ? c.nanostatelocout : "";
    nanostatelocres = c.nanostatelocres ? c.nanostatelocres : "";
    nanologloc = c.nanologloc ? c.nanologloc : "none";
    macrostatelocout = c.macrostatelocout ? c.macrostatelocout : "";
    md_scripts_directory = c.md_scripts_directory ? c.md_scripts_directory : "";
    freq_checkpoint = c.freq_checkpoint;
    freq_output_homog = c.freq_output_homog;
    mdtype.clear();
    for (int i = 0; i < c.n_materials; i++) mdtype.push_back(c.mdtype[i]);
    for (int i = 0; i < 3; i++) cg_dir[i] = c.cg_dir[i];
    nrepl = (unsigned)c.nrepl;
    verbose = c.verbose != 0;
    if (c.use_pjm_scheduler) return fail(SCEMA_MD_ERR_ARG, "use pjm scheduler: external pilot-job branch is out of scope");
    if (!engine_ && !approx_md_with_hookes_law) return fail(SCEMA_MD_ERR_DEVICE, "no GPU engine and not in Hooke test mode");
    if (freq_checkpoint <= 0 || freq_output_homog <= 0) return fail(SCEMA_MD_ERR_ARG, "output frequencies must be positive");
    int rc;
    if ((rc = restart())) return rc;
    if ((rc = load_replica_generation_data())) return rc;
    if ((rc = load_replica_equilibration_data())) return rc;
    if (rank_ == 0) average_replica_data();
    return SCEMA_MD_OK;
  }

  // reference stmd_sync.h:1070-1132
  int update(int tstp, double ptime, int nstp, ScaleBridgingData &scale_bridging_data) {
    present_time = ptime;
    timestep = tstp;
    newtonstep = nstp;
    time_id = std::to_string(timestep) + "-" + std::to_string(newtonstep);
    output_homog = (timestep % freq_output_homog == 0);
    checkpoint_save = (timestep % freq_checkpoint == 0);
    std::vector<MDSim> md_simulations = prepare_md_simulations(scale_bridging_data);
    const int n_md = (int)md_simulations.size();
    if (verbose && rank_ == 0) {
      std::cout << "        Running " << n_md << " simulations:\n";
      for (int i = 0; i < n_md; i++) std::cout << md_simulations[i].qp_id << "-" << md_simulations[i].replica << " ";
      std::cout << std::endl;
    }
    if (n_md > 0) {
      // a rank whose share failed still enters the collective of share_stresses: its status word ends the update on
      // every rank (the reference's exit(1) would take the whole MPI job down; a blocked collective would not)
      const int rc_exec = execute_inside_md_simulations(md_simulations);
      int rc;
      if ((rc = share_stresses(md_simulations, rc_exec))) return rc;
      // every rank holds every stress after the all-gather: the serial branch of the reference
      // (stmd_sync.h:1119-1123) runs everywhere, so no broadcast of the update_list is needed afterwards
      if ((rc = store_md_simulations(md_simulations, scale_bridging_data))) return rc;
    }
    return SCEMA_MD_OK;
  }

 private:
  int fail(int code, const std::string &msg) {
    err_ = msg;
    std::cerr << msg << std::endl;
    return code;
  }

  // reference stmd_sync.h:167-187: nanoscale_input/restart/lcts.* become the current states
  int restart() {
    if (!engine_) return SCEMA_MD_OK;
    const std::string dir = nanostatelocin + "/restart";
    DIR *d = opendir(dir.c_str());
    if (!d) return SCEMA_MD_OK;  // nothing to restart from
    std::vector<std::string> names;
    while (dirent *e = readdir(d)) {
      const std::string n = e->d_name;
      if (n.rfind("lcts.", 0) == 0 && n.size() > 10 && n.substr(n.size() - 5) == ".dump") names.push_back(n);
    }
    closedir(d);
    pending_restart = names;  // loaded once the replicas are registered
    return SCEMA_MD_OK;
  }

  // reference stmd_sync.h:280-359
  int load_replica_generation_data() {
    for (size_t imd = 0; imd < mdtype.size(); imd++)
      for (unsigned irep = 0; irep < nrepl; irep++) {
        const std::string filename = nanostatelocin + "/" + mdtype[imd] + "_" + std::to_string(irep + 1) + ".json";
        if (!file_exists(filename)) return fail(SCEMA_MD_ERR_IO, "Missing data for replica #" + std::to_string(irep + 1) + " of material" + mdtype[imd] + ".");
      }
    replica_data.assign(nrepl * mdtype.size(), ReplicaData());
    for (size_t imd = 0; imd < mdtype.size(); imd++)
      for (unsigned irep = 0; irep < nrepl; irep++) {
        ReplicaData &r = replica_data[imd * nrepl + irep];
        r.mat = mdtype[imd];
        r.repl = irep + 1;
        const std::string filename = nanostatelocin + "/" + r.mat + "_" + std::to_string(r.repl) + ".json";
        FlatJson pt;
        if (!pt.parse_file(filename)) return fail(SCEMA_MD_ERR_IO, "Invalid JSON replica data input file (" + filename + ")");
        r.rho = std::stod(pt.get("relative_density", "0")) * 1000.;
        r.nflakes = std::stoi(pt.get("Nsheets", "0"));
        if (r.nflakes == 1) {
          Tensor1 nvrep;
          nvrep[0] = std::stod(pt.get("normal_vector.1.x", "0"));
          nvrep[1] = std::stod(pt.get("normal_vector.1.y", "0"));
          nvrep[2] = std::stod(pt.get("normal_vector.1.z", "0"));
          Tensor1 cg;
          for (int i = 0; i < 3; i++) cg[i] = cg_dir[i];
          r.rotam = compute_rotation_tensor(nvrep, cg);
        } else {
          r.rotam = Tensor2::identity();
        }
      }
    return SCEMA_MD_OK;
  }

  // reference stmd_sync.h:361-453
  int load_replica_equilibration_data() {
    for (size_t imd = 0; imd < mdtype.size(); imd++)
      for (unsigned irep = 0; irep < nrepl; irep++) {
        ReplicaData &r = replica_data[imd * nrepl + irep];
        const std::string base = nanostatelocin + "/init." + r.mat + "_" + std::to_string(r.repl);
        if (file_exists(base + ".length")) read_tensor((base + ".length").c_str(), r.init_length);
        else std::cerr << "Missing equilibrated initial length data for material " << r.mat << " replica #" << r.repl << std::endl;
        if (file_exists(base + ".stress")) read_tensor((base + ".stress").c_str(), r.init_stress);
        else std::cerr << "Missing equilibrated initial stress data for material " << r.mat << " replica #" << r.repl << std::endl;
        if (file_exists(base + ".stiff")) read_tensor((base + ".stiff").c_str(), r.init_stiff);
        else std::cerr << "Missing equilibrated initial stiffness data for material " << r.mat << " replica #" << r.repl << std::endl;
        const std::string bin = base + ".bin";
        if (file_exists(bin)) {
          if (rank_ == 0 && !nanostatelocout.empty()) {  // copy of the replica input system, as the reference does
            std::ifstream in(bin, std::ios::binary);
            std::ofstream out(nanostatelocout + "/init." + r.mat + "_" + std::to_string(r.repl) + ".bin", std::ios::binary);
            out << in.rdbuf();
          }
          if (engine_) {
            // init.<mat>_<rep>.bin is either our replica container or a LAMMPS binary restart as the reference's
            // init_material writes it (init_material_problem.h:209): told apart by the magic string
            char magic[16] = {0};
            { std::ifstream in(bin, std::ios::binary); in.read(magic, 15); }
            const bool lammps = std::string(magic) == "LammpS RestartT";
            int rc = lammps ? scema_md_load_lammps_restart(engine_, r.mat.c_str(), r.repl, bin.c_str())
                            : scema_md_load_replica_file(engine_, r.mat.c_str(), r.repl, bin.c_str());
            if (rc) return fail(rc, lammps ? "cannot use LAMMPS restart " + bin + " (see stderr)" : std::string(scema_md_last_error(engine_)));
          }
        } else {
          std::cerr << "Missing equilibrated initial system for material " << r.mat << " replica #" << r.repl << std::endl;
          if (!approx_md_with_hookes_law) return fail(SCEMA_MD_ERR_NOSTATE, "init." + r.mat + "_" + std::to_string(r.repl) + ".bin is required for MD");
        }
      }
    // restart states (lcts.<qp>.<mat>_<rep>.dump)
    for (const std::string &n : pending_restart) {
      const size_t p1 = n.find('.', 5);
      const size_t p2 = n.rfind('_');
      if (p1 == std::string::npos || p2 == std::string::npos || p2 < p1) continue;
      const int qp = std::atoi(n.substr(5, p1 - 5).c_str());
      const std::string mat = n.substr(p1 + 1, p2 - p1 - 1);
      const int rep = std::atoi(n.substr(p2 + 1).c_str());
      int rc = scema_md_load_state_file(engine_, qp, mat.c_str(), rep, (nanostatelocin + "/restart/" + n).c_str());
      if (rc) return fail(rc, std::string("Failed to load restart state ") + n + ": " + scema_md_last_error(engine_));
    }
    pending_restart.clear();
    return SCEMA_MD_OK;
  }

  // reference stmd_sync.h:455-489
  void average_replica_data() {
    if (macrostatelocout.empty()) return;
    for (size_t imd = 0; imd < mdtype.size(); imd++) {
      SymmetricTensor4 initial_stiffness_tensor;
      double initial_density = 0.;
      for (unsigned repl = 0; repl < nrepl; repl++) {
        initial_stiffness_tensor += rotate_tensor(replica_data[imd * nrepl + repl].init_stiff, replica_data[imd * nrepl + repl].rotam);
        initial_density += replica_data[imd * nrepl + repl].rho;
      }
      initial_stiffness_tensor /= nrepl;
      initial_density /= nrepl;
      write_tensor((macrostatelocout + "/init." + mdtype[imd] + ".stiff").c_str(), initial_stiffness_tensor);
      write_tensor((macrostatelocout + "/init." + mdtype[imd] + ".density").c_str(), initial_density);
    }
  }

  // reference stmd_sync.h:491-568 (set_md_procs :189-278 reduces to "one GPU per batch")
  std::vector<MDSim> prepare_md_simulations(const ScaleBridgingData &scale_bridging_data) {
    std::vector<MDSim> request_simulations;
    const std::vector<QP> &update_list = scale_bridging_data.update_list;
    const unsigned n_qp = (unsigned)update_list.size();
    for (unsigned qp = 0; qp < n_qp; ++qp)
      for (unsigned repl = 0; repl < nrepl; repl++) {
        MDSim md_sim;
        md_sim.qp_id = update_list[qp].id;
        md_sim.most_recent_qp_id = update_list[qp].most_recent_id;
        md_sim.replica = repl + 1;
        md_sim.material = update_list[qp].material;
        const int replica_data_index = md_sim.material * nrepl + repl;
        md_sim.matid = replica_data[replica_data_index].mat;
        md_sim.time_id = time_id;
        md_sim.force_field = md_force_field;
        md_sim.timestep_length = md_timestep_length;
        md_sim.temperature = md_temperature;
        md_sim.nsteps_sample = md_nsteps_sample;
        md_sim.strain_rate = md_strain_rate;
        md_sim.output_folder = nanostatelocout;
        md_sim.restart_folder = nanostatelocres;
        md_sim.scripts_folder = md_scripts_directory;
        md_sim.output_homog = false;  // reference stmd_sync.h:531
        md_sim.checkpoint = checkpoint_save;
        if (!approx_md_with_hookes_law) md_sim.define_file_names(nanologloc);
        SymmetricTensor2 cg_loc_rep_strain(update_list[qp].update_strain);
        // common ground -> replica orientation
        md_sim.strain = rotate_tensor(cg_loc_rep_strain, replica_data[replica_data_index].rotam.transposed());
        // strain -> length variation (turned back into a strain with the current box, stmd_problem.h:222-225)
        if (!approx_md_with_hookes_law)
          for (int j = 0; j < 3; j++) {
            md_sim.strain(j, j) *= replica_data[replica_data_index].init_length[j];
            md_sim.strain(j, (j + 1) % 3) *= replica_data[replica_data_index].init_length[(j + 2) % 3];
          }
        md_sim.stiffness = replica_data[replica_data_index].init_stiff;
        request_simulations.push_back(md_sim);
      }
    return request_simulations;
  }

  // reference stmd_sync.h:570-618: round robin i % n_md_batches; here one GPU per batch, dealt by the engine's planner
  int execute_inside_md_simulations(std::vector<MDSim> &md_simulations) {
    STMDProblem stmd_problem(engine_, rank_, world_, verbose, lammps_state_files);
    int rc = stmd_problem.strain_batch(md_simulations, approx_md_with_hookes_law);
    if (rc) return fail(rc, stmd_problem.last_error());
    return SCEMA_MD_OK;
  }

  // reference stmd_sync.h:620-726: 6 doubles per simulation to everybody, ONE collective.  With a communicator attached
  // to the engine (scema_md_comm_init_rccl / _host) that all-gather already ran inside scema_md_strain_batch (MD and Hooke
  // mode alike) and every stress -- or every rank's failure -- is here; otherwise it runs through the host program's
  // callback, laid out by the engine's plan (MD) or by the round robin i % world (stateless Hooke mode).  Whether the
  // callback runs is decided by what every rank knows (world, mode, communicator), never by this rank's own results;
  // each rank's record ends with its status word and plan hash (SCEMA_MD_RESULT_TRAILER).
  int share_stresses(std::vector<MDSim> &md_simulations, int rc_exec) {
    const int n = (int)md_simulations.size();
    const bool engine_gathered = engine_ && scema_md_comm_world(engine_) > 1;
    if (world_ > 1 && !engine_gathered) {
      if (!allgather_) return fail(SCEMA_MD_ERR_ARG, "world > 1 needs an all-gather: attach a communicator to the engine or pass a callback");
      const bool md = engine_ && !approx_md_with_hookes_law;
      std::vector<int> owner(n), pos(n);
      int per_rank = (n + world_ - 1) / world_;
      for (int i = 0; i < n; i++) { owner[i] = i % world_; pos[i] = i / world_; }
      if (md) {
        // A rank whose call ended before planning (a replica it has not registered, a device error) has no plan and so does not know
        // the record size of the stress collective; the others may have one.  Whether everybody has a plan is therefore agreed on
        // first, in a collective of ONE word per rank that every rank enters whatever happened to it (such a rank used to
        // return here and leave the others in the all-gather below).
        const bool have_plan = scema_md_last_plan(engine_, n, owner.data(), pos.data(), &per_rank) == SCEMA_MD_OK;
        double mine = have_plan ? 0.0 : (double)(rc_exec ? rc_exec : SCEMA_MD_ERR_ARG);
        std::vector<double> all(world_, 0.0);
        if (allgather_(ag_ctx_, nullptr, &mine, 1, all.data())) return fail(SCEMA_MD_ERR_DEVICE, "all-gather of the plan status failed");
        for (int r = 0; r < world_; r++)
          if (all[r] != 0.0) {
            if (r == rank_ && rc_exec) return rc_exec;   // err_ is set
            (void)scema_md_settle_update(engine_, 1);   // this rank may have run its share: it did not happen
            return fail((int)all[r], "rank " + std::to_string(r) + " could not plan this update (code " + std::to_string((int)all[r]) + "): the update is abandoned on every rank");
          }
      }
      const size_t cnt = 6 * (size_t)std::max(per_rank, 1) + SCEMA_MD_RESULT_TRAILER;
      std::vector<double> local(cnt, 0.0), gathered(cnt * world_, 0.0);
      for (int i = 0; i < n; i++)
        if (owner[i] == rank_ && md_simulations[i].stress_updated)
          for (int k = 0; k < 6; k++) local[6 * (size_t)pos[i] + k] = md_simulations[i].stress.raw[k];
      local[cnt - 2] = (double)rc_exec;   // the MD path sends the engine's device buffer, which carries the same word
      int rc = allgather_(ag_ctx_, md ? engine_ : nullptr, local.data(), (int)cnt, gathered.data());
      if (rc) { if (md) (void)scema_md_settle_update(engine_, 1); return fail(SCEMA_MD_ERR_DEVICE, "all-gather of the stresses failed"); }
      if (rc_exec) return rc_exec;   // err_ is set (and the engine has taken this rank's share back already)
      // the engine's share of this rank is waiting for the verdict of the collective (scema_md_settle_update): it is taken back
      // when any rank failed or the plans differ, and stands otherwise
      for (int r = 0; r < world_; r++)
        if (gathered[r * cnt + cnt - 2] != 0.0) {
          if (md) (void)scema_md_settle_update(engine_, 1);
          return fail((int)gathered[r * cnt + cnt - 2], "rank " + std::to_string(r) + " failed during the update (code " +
                      std::to_string((int)gathered[r * cnt + cnt - 2]) + "): the update is abandoned on every rank");
        }
      for (int r = 1; r < world_; r++)
        if (gathered[r * cnt + cnt - 1] != gathered[cnt - 1]) {
          if (md) (void)scema_md_settle_update(engine_, 1);
          return fail(SCEMA_MD_ERR_ARG, "ranks 0 and " + std::to_string(r) + " computed different plans for this update");
        }
      if (md) (void)scema_md_settle_update(engine_, 0);
      for (int i = 0; i < n; i++) {
        const double *src = gathered.data() + ((size_t)owner[i] * cnt + (size_t)pos[i] * 6);
        for (int k = 0; k < 6; k++) md_simulations[i].stress.raw[k] = src[k];
        md_simulations[i].stress_updated = true;
      }
    }
    if (rc_exec) return rc_exec;   // single rank, or the engine's own collective already told every rank
    // reference stmd_sync.h:712-725
    for (int i = 0; i < n; i++)
      if (!md_simulations[i].stress_updated)
        return fail(SCEMA_MD_ERR_ARG, "Stress not set or not communicated to rank (" + std::to_string(rank_) + ") . " + std::to_string(md_simulations[i].qp_id));
    return SCEMA_MD_OK;
  }

  // reference stmd_sync.h:857-876
  int get_sim_id(const std::vector<MDSim> &md_simulations, int qp_id, int rep) {
    // qp-major, repl-minor construction order makes the common case O(1); fall back to the scan
    for (size_t i = 0; i < md_simulations.size(); i++)
      if (md_simulations[i].qp_id == qp_id && md_simulations[i].replica == rep) return (int)i;
    return -1;
  }

  // reference stmd_sync.h:878-922
  int store_md_simulations(const std::vector<MDSim> &md_simulations, ScaleBridgingData &scale_bridging_data) {
    const unsigned n_qp = (unsigned)scale_bridging_data.update_list.size();
    for (unsigned qp = 0; qp < n_qp; ++qp) {
      const int qp_id = scale_bridging_data.update_list[qp].id;
      SymmetricTensor2 cg_loc_stress;
      for (unsigned rep = 0; rep < nrepl; ++rep) {
        int md_sim_id = (int)(qp * nrepl + rep);
        if (md_sim_id >= (int)md_simulations.size() || md_simulations[md_sim_id].qp_id != qp_id || md_simulations[md_sim_id].replica != (int)rep + 1)
          md_sim_id = get_sim_id(md_simulations, qp_id, rep + 1);
        if (md_sim_id < 0) return fail(SCEMA_MD_ERR_ARG, "Error: MDSim not found for qp " + std::to_string(qp_id) + " replica " + std::to_string(rep + 1));
        const MDSim &md_simulation = md_simulations[md_sim_id];
        const unsigned replica_data_index = md_simulation.material * nrepl + rep;
        SymmetricTensor2 loc_rep_stress = md_simulation.stress;
        if (!approx_md_with_hookes_law) loc_rep_stress -= replica_data[replica_data_index].init_stress;
        cg_loc_stress += rotate_tensor(loc_rep_stress, replica_data[replica_data_index].rotam);
      }
      cg_loc_stress /= nrepl;
      for (int i = 0; i < 6; ++i) scale_bridging_data.update_list[qp].update_stress[i] = cg_loc_stress.access_raw_entry(i);
    }
    return SCEMA_MD_OK;
  }

  scema_md_engine *engine_;
  int rank_, world_;
  scema_allgather_fn allgather_;
  void *ag_ctx_;
  std::string err_;

  int start_timestep = 0, timestep = 0, newtonstep = 0;
  double present_time = 0;
  std::string time_id;
  std::vector<std::string> mdtype;
  unsigned nrepl = 0;
  std::vector<ReplicaData> replica_data;
  double cg_dir[3] = {1, 0, 0};
  double md_timestep_length = 0, md_temperature = 0, md_strain_rate = 0;
  int md_nsteps_sample = 0;
  std::string md_force_field;
  int freq_checkpoint = 1, freq_output_homog = 1;
  bool output_homog = false, checkpoint_save = false;
  std::string macrostatelocout, nanostatelocin, nanostatelocout, nanostatelocres, nanologloc, md_scripts_directory;
  bool approx_md_with_hookes_law = false, verbose = false, lammps_state_files = false;
  std::vector<std::string> pending_restart;
};

}  // namespace scema
