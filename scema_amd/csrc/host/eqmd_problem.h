// eqmd_problem.h -- host mirror of HMM::EQMDProblem<3> (reference headers/init_material_problem.h:30-355).
// equil() with the reference's full argument list first brings the replica to its equilibrated state -- "Compute state
// data" (no init.<mat>_<rep>.bin yet: read <slocin>/<mat>_<rep>.data, run the schedule of in.init.lammps on the GPU, write
// the state file) or "Reuse of state data" (:167-184) -- then calls the engine's init_material (box lengths, initial stress,
// stiffness by +-strain_ampl finite strains; 13 MD runs in one GPU batch) and writes the three files STMDSync::init reads
// back (stmd_sync.h:382-445), with the reference's writers (read_write.h:180-244 there, read_write.h here).
#pragma once
#include <string>

#include "../../../include/scema_md.h"
#include "read_write.h"
#include "tensors.h"

namespace scema {

class EQMDProblem {
 public:
  explicit EQMDProblem(scema_md_engine *engine) : engine_(engine) {}
  const std::string &last_error() const { return err_; }

  // EQMDProblem::equil with every argument of the reference (init_material_problem.h:309-315): slocin = folder of
  // <cmat>_<rep>.data (LAMMPS write_data, atom_style full), systof = init.<cmat>_<rep>.bin, mdnse = nsinit of in.init.lammps;
  // qplogloc and scrloc (LAMMPS log and script folders) have no counterpart here and are accepted for signature parity.
  int equil(const std::string &cmat, const std::string &slocin, const std::string &qplogloc, const std::string &scrloc, const std::string &lengthof,
            const std::string &stressof, const std::string &stiffof, const std::string &systof, int rep, double mdts, double mdtem, int mdnss, int mdnse,
            double mdss, double mdsa, const std::string &mdff) {
    (void)qplogloc;
    if (mdff != "opls" && mdff != "reax") {
      err_ = "Error: Force field is " + mdff + " but only 'opls' and 'reax' are implemented... ";
      return SCEMA_MD_ERR_ARG;
    }
    if (!engine_) {
      err_ = "init_material needs an engine (no CPU fallback)";
      return SCEMA_MD_ERR_DEVICE;
    }
    const bool registered = scema_md_replica_natoms(engine_, cmat.c_str(), rep) > 0;
    const bool reax = mdff == "reax";
    if (reax) {
      // init_material_problem.h:119-121,157-160: pair_coeff * * <scriptsloc>/ffield.reax.2 H C N O + fix qeq/reax ... 1e-6
      static const char *const elements[4] = {"H", "C", "N", "O"};
      const std::string ff = scrloc + "/ffield.reax.2";
      const int rc = scema_md_reax_configure(engine_, ff.c_str(), elements, 4, 1.0e-6, -1.0);   // also selects the force field for what follows
      if (rc) { err_ = scema_md_last_error(engine_); return rc; }
    }
    struct Off { scema_md_engine *e; bool on; ~Off() { if (on) scema_md_reax_activate(e, 0); } } off{engine_, reax};
    if (!file_exists(systof)) {
      // "Compute state data...": in.init.lammps on the data file
      if (!registered) {
        const std::string data = slocin + "/" + cmat + "_" + std::to_string(rep) + ".data";
        static const double sp_lj[3] = {0.0, 0.0, 1.0}, sp_coul[3] = {0.0, 0.0, 1.0};   // special_bonds lj/coul 0.0 0.0 1.0 (in.init.lammps:33)
        const int rc = scema_md_load_lammps_data(engine_, cmat.c_str(), rep, data.c_str(), sp_lj, sp_coul);
        if (rc) { err_ = scema_md_last_error(engine_); return rc; }
      }
      scema_md_equilparams q;
      q.nsteps_equil = mdnse;
      q.timestep_length = mdts;
      q.temperature = mdtem;
      q.seed = 1234;   // init_material_problem.h:167
      double len[3], info[5];
      int rc = scema_md_equilibrate(engine_, cmat.c_str(), rep, &q, len, info);
      if (rc) { err_ = scema_md_last_error(engine_); return rc; }
      // "Saving state data..." (write_restart ${systemoutputfile}, :208-210)
      rc = scema_md_save_replica_file(engine_, cmat.c_str(), rep, systof.c_str());
      if (rc) { err_ = scema_md_last_error(engine_); return rc; }
    } else if (!registered) {
      err_ = "Reuse of state data: replica " + cmat + "_" + std::to_string(rep) + " must be registered with the engine (STMDSync::init does that from " + systof + ")";
      return SCEMA_MD_ERR_NOSTATE;
    }
    return equil(cmat, lengthof, stressof, stiffof, rep, mdts, mdtem, mdnss, mdss, mdsa, mdff);
  }

  // the part of EQMDProblem::equil (init_material_problem.h:309-315) that runs once the state exists:
  // cmat, lengthof/stressof/stiffof, rep (1-based), mdts, mdtem, mdnss, mdss (strain rate), mdsa (strain amplitude), mdff
  int equil(const std::string &cmat, const std::string &lengthof, const std::string &stressof, const std::string &stiffof, int rep,
            double mdts, double mdtem, int mdnss, double mdss, double mdsa, const std::string &mdff) {
    if (mdff != "opls" && mdff != "reax") {   // init_material_problem.h:336-341 (print + exit(1) there)
      err_ = "Error: Force field is " + mdff + " but only 'opls' and 'reax' are implemented... ";
      return SCEMA_MD_ERR_ARG;
    }
    // "reax": the caller has selected the force field (scema_md_reax_configure / the full equil above does it from scrloc)
    if (!engine_) {
      err_ = "init_material needs an engine (no CPU fallback)";
      return SCEMA_MD_ERR_DEVICE;
    }
    scema_md_eqparams p;
    p.timestep_length = mdts;
    p.temperature = mdtem;
    p.nsteps_sample = mdnss;
    p.strain_ampl = mdsa;
    p.strain_rate = mdss;
    double length[3], stress[6], stiff[36];
    const int rc = scema_md_init_material(engine_, cmat.c_str(), rep, &p, length, stress, stiff);
    if (rc) {
      err_ = scema_md_last_error(engine_);
      return rc;
    }
    Tensor1 len;
    for (int d = 0; d < 3; d++) len[d] = length[d];
    SymmetricTensor2 sig;   // file order 00,01,02,11,12,22 -> raw xx,yy,zz,xy,xz,yz
    sig.raw[0] = stress[0]; sig.raw[3] = stress[1]; sig.raw[4] = stress[2]; sig.raw[1] = stress[3]; sig.raw[5] = stress[4]; sig.raw[2] = stress[5];
    SymmetricTensor4 c4;
    for (int i = 0; i < 36; i++) c4.c[i] = stiff[i];
    if (!write_tensor(lengthof.c_str(), len) || !write_tensor(stressof.c_str(), sig) || !write_tensor(stiffof.c_str(), c4)) {
      err_ = "cannot write the init.* files";
      return SCEMA_MD_ERR_IO;
    }
    return SCEMA_MD_OK;
  }

 private:
  scema_md_engine *engine_;
  std::string err_;
};

}  // namespace scema
