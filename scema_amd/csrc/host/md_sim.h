// md_sim.h -- per-simulation request/response record (reference headers/md_sim.h:15-58)
#pragma once
#include <string>

#include "../../../include/scema_md.h"
#include "tensors.h"

namespace scema {

struct MDSim {
  int qp_id = 0;
  int most_recent_qp_id = 0;
  int replica = 0;  // 1-based
  int material = 0;
  std::string matid;
  std::string time_id;
  SymmetricTensor2 strain;     // input
  SymmetricTensor4 stiffness;  // input
  SymmetricTensor2 stress;     // output
  bool stress_updated = false;
  std::string output_folder, restart_folder, scripts_folder, log_file;
  double timestep_length = 0, temperature = 0;
  int nsteps_sample = 0;
  double strain_rate = 0;
  std::string force_field;
  bool output_homog = false;
  bool checkpoint = false;

  // reference md_sim.h:50-56
  void define_file_names(const std::string &nanologloc) {
    if (nanologloc != "none") log_file = nanologloc + "/" + time_id + "." + std::to_string(qp_id) + "." + matid + "_" + std::to_string(replica);
    else log_file = "none";
  }

  // view for the C ABI (pointers stay valid while *this lives)
  scema_mdsim to_c() const {
    scema_mdsim m;
    m.qp_id = qp_id; m.most_recent_qp_id = most_recent_qp_id; m.replica = replica; m.material = material;
    m.matid = matid.c_str(); m.time_id = time_id.c_str(); m.output_folder = output_folder.c_str();
    m.restart_folder = restart_folder.c_str(); m.scripts_folder = scripts_folder.c_str(); m.log_file = log_file.c_str();
    m.force_field = force_field.c_str();
    for (int i = 0; i < 6; i++) { m.strain[i] = strain.raw[i]; m.stress[i] = stress.raw[i]; }
    for (int i = 0; i < 36; i++) m.stiffness[i] = stiffness.c[i];
    m.timestep_length = timestep_length; m.temperature = temperature; m.strain_rate = strain_rate;
    m.nsteps_sample = nsteps_sample; m.output_homog = output_homog; m.checkpoint = checkpoint;
    m.stress_updated = stress_updated;
    return m;
  }
};

}  // namespace scema
