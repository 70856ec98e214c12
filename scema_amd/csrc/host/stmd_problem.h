// stmd_problem.h -- host mirror of HMM::STMDProblem<3> (reference headers/stmd_problem.h:40-496).
// strain() keeps the reference's signature and error behaviour; lammps_straining() is replaced by
// the GPU engine behind include/scema_md.h, and strain_batch() hands the engine the whole vector
// that STMDSync::execute_inside_md_simulations iterates.
#pragma once
#include <fstream>
#include <iomanip>
#include <iostream>
#include <vector>

#include "../../../include/scema_md.h"
#include "md_sim.h"

namespace scema {

class STMDProblem {
 public:
  // mdcomm/pcolor of the reference become (engine, rank, world): one engine = one GPU = one batch
  // lammps_states: write lcts.* in LAMMPS' own restart layout and, like the reference (stmd_problem.h:258), a
  // last.<qp>.<mat>_<rep>.dump after every evaluation, so that a LAMMPS-based run can take the simulations over
  STMDProblem(scema_md_engine *engine, int rank, int world, bool verbose, bool lammps_states = false)
      : engine_(engine), rank_(rank), world_(world), verbose_(verbose), lammps_states_(lammps_states) {}

  // reference stmd_problem.h:458-496
  int strain(MDSim &md_sim, bool approx_md_with_hookes_law) {
    std::vector<MDSim *> one(1, &md_sim);
    return run(one, approx_md_with_hookes_law, 0, 1);
  }

  // all simulations of one update(); the engine's planner decides which of them run on this rank
  int strain_batch(std::vector<MDSim> &sims, bool approx_md_with_hookes_law) {
    std::vector<MDSim *> p;
    for (auto &s : sims) p.push_back(&s);
    return run(p, approx_md_with_hookes_law, rank_, world_);
  }

  // reference stmd_problem.h:386-392
  static SymmetricTensor2 stress_from_hookes_law(const SymmetricTensor2 &strain, const SymmetricTensor4 &stiffness) {
    return contract(stiffness, strain);
  }

  const std::string &last_error() const { return err_; }

 private:
  int run(std::vector<MDSim *> &sims, bool hooke, int rank, int world) {
    const int n = (int)sims.size();
    for (int i = 0; i < n; i++) {
      MDSim &m = *sims[i];
      // reference stmd_problem.h:462-467 (exit(1) there; an error code here, the C shim may exit)
      if (m.force_field != "opls" && m.force_field != "reax") {
        std::cerr << "Error: Force field is " << m.force_field << " but only 'opls' and 'reax' are implemented... " << std::endl;
        err_ = "unknown force field " + m.force_field;
        return SCEMA_MD_ERR_ARG;
      }
    }
    // which simulations this rank runs: i % world for the stateless Hooke mode (the reference's round robin,
    // stmd_sync.h:583); for MD the engine's planner decides (states are resident on ONE GPU, host/sim_plan.h)
    std::vector<int> owner(n);
    for (int i = 0; i < n; i++) owner[i] = i % world;
    std::vector<int> mine;
    if (hooke && engine_ && scema_md_comm_world(engine_) > 1) {
      // the reference's fake backend with a communicator attached to the engine: dealt i % world and gathered by the
      // engine's own collective, like an MD update
      std::vector<scema_mdsim> c(n);
      for (int i = 0; i < n; i++) c[i] = sims[i]->to_c();
      const int rc = scema_md_strain_batch(engine_, c.data(), n, 1, rank, world);
      if (rc != SCEMA_MD_OK) {
        err_ = scema_md_last_error(engine_);
        return rc;
      }
      for (int i = 0; i < n; i++) {
        if (owner[i] == rank) mine.push_back(i);
        for (int k = 0; k < 6; k++) sims[i]->stress.raw[k] = c[i].stress[k];
        sims[i]->stress_updated = c[i].stress_updated != 0;
      }
    } else if (hooke) {
      // "approximate md with hookes law": the reference's own fake backend (stmd_problem.h:479-483)
      for (int i = 0; i < n; i++)
        if (owner[i] == rank) mine.push_back(i);
      if (verbose_)
        for (int i : mine) std::cout << " \t" << sims[i]->qp_id << "-" << sims[i]->replica << "-start" << std::endl << std::flush;
      for (int i : mine) {
        sims[i]->stress = stress_from_hookes_law(sims[i]->strain, sims[i]->stiffness);
        sims[i]->stress_updated = true;
      }
    } else {
      if (!engine_) {
        err_ = "no GPU engine: the MD path has no CPU fallback (only the Hooke test mode runs without a GPU)";
        return SCEMA_MD_ERR_DEVICE;
      }
      std::vector<scema_mdsim> c(n);
      for (int i = 0; i < n; i++) c[i] = sims[i]->to_c();
      int rc = scema_md_strain_batch(engine_, c.data(), n, 0, rank, world);
      if (rc != SCEMA_MD_OK) {
        err_ = scema_md_last_error(engine_);
        return rc;
      }
      if (world > 1) (void)scema_md_last_plan(engine_, n, owner.data(), nullptr, nullptr);
      for (int i = 0; i < n; i++)
        if (owner[i] == rank) mine.push_back(i);
      if (verbose_)
        for (int i : mine) std::cout << " \t" << sims[i]->qp_id << "-" << sims[i]->replica << "-start" << std::endl << std::flush;
      // with a communicator attached to the engine every stress is already here (one all-gather inside the call)
      for (int i = 0; i < n; i++)
        if (c[i].stress_updated) {
          for (int k = 0; k < 6; k++) sims[i]->stress.raw[k] = c[i].stress[k];
          sims[i]->stress_updated = true;
        }
      for (int i : mine) {
        const std::string tail = std::to_string(sims[i]->qp_id) + "." + sims[i]->matid + "_" + std::to_string(sims[i]->replica) + ".dump";
        // reference stmd_problem.h:258: last.<qp>.<mat>_<rep>.dump after the straining run (here: the state after the
        // evaluation; kept in HBM anyway, written only on request)
        // the reax branch exchanges states as text dumps (stmd_problem.h:261-264), the opls branch as binary restarts (:258)
        const bool reax = sims[i]->force_field == "reax";
        if (lammps_states_ && !sims[i]->output_folder.empty()) {
          const std::string last = sims[i]->output_folder + "/last." + tail;
          rc = reax ? scema_md_save_state_dump(engine_, sims[i]->qp_id, sims[i]->matid.c_str(), sims[i]->replica, last.c_str(), 0, 1)
                    : scema_md_save_state_lammps(engine_, sims[i]->qp_id, sims[i]->matid.c_str(), sims[i]->replica, last.c_str(), sims[i]->timestep_length, 0);
          if (rc != SCEMA_MD_OK) {
            err_ = scema_md_last_error(engine_);
            return rc;
          }
        }
        // reference stmd_problem.h:266-273: lcts.<qp>.<mat>_<rep>.dump every "checkpoint frequency" steps
        if (sims[i]->checkpoint && !sims[i]->restart_folder.empty()) {
          const std::string path = sims[i]->restart_folder + "/lcts." + tail;
          rc = !lammps_states_ ? scema_md_save_state_file(engine_, sims[i]->qp_id, sims[i]->matid.c_str(), sims[i]->replica, path.c_str())
               : reax        ? scema_md_save_state_dump(engine_, sims[i]->qp_id, sims[i]->matid.c_str(), sims[i]->replica, path.c_str(), 0, 1)
                             : scema_md_save_state_lammps(engine_, sims[i]->qp_id, sims[i]->matid.c_str(), sims[i]->replica, path.c_str(), sims[i]->timestep_length, 0);
          if (rc != SCEMA_MD_OK) {
            err_ = scema_md_last_error(engine_);
            return rc;
          }
        }
      }
    }
    for (int i : mine) {
      write_local_data(*sims[i]);
      if (verbose_) std::cout << " \t" << sims[i]->qp_id << "-" << sims[i]->replica << std::endl << std::flush;
    }
    return SCEMA_MD_OK;
  }

  // reference stmd_problem.h:394-456 : one CSV row per evaluation, columns in k<=l order 00,01,02,11,12,22
  void write_local_data(const MDSim &m) {
    if (m.output_folder.empty()) return;
    const std::string filename = m.output_folder + "/mddata_qpid" + std::to_string(m.qp_id) + "_repl" + std::to_string(m.replica) + ".csv";
    std::ofstream ofile(filename, std::ios_base::app);
    if (!ofile.is_open()) return;
    if ((long)ofile.tellp() == 0) {
      ofile << "qp_id,material_id,time_id,temperature,strain_rate,force_field,replica_id";
      for (int k = 0; k < 3; k++)
        for (int l = k; l < 3; l++) ofile << ",strain_" << k << l;
      for (int k = 0; k < 3; k++)
        for (int l = k; l < 3; l++) ofile << ",stress_" << k << l;
      ofile << std::endl;
    }
    ofile << m.qp_id << "," << m.matid << "," << m.time_id << "," << m.temperature << "," << m.strain_rate << "," << m.force_field << "," << m.replica;
    for (int k = 0; k < 3; k++)
      for (int l = k; l < 3; l++) ofile << "," << std::setprecision(16) << m.strain(k, l);
    for (int k = 0; k < 3; k++)
      for (int l = k; l < 3; l++) ofile << "," << std::setprecision(16) << m.stress(k, l);
    ofile << std::endl;
  }

  scema_md_engine *engine_;
  int rank_, world_;
  bool verbose_, lammps_states_;
  std::string err_;
};

}  // namespace scema
