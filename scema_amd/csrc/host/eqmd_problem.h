// eqmd_problem.h -- host mirror of HMM::EQMDProblem<3> (reference headers/init_material_problem.h:30-355) for an
// already equilibrated replica: equil() calls the engine's init_material (box lengths, initial stress, stiffness by
// +-strain_ampl finite strains; 13 MD runs in one GPU batch) and writes the three files STMDSync::init reads back
// (stmd_sync.h:382-445), with the reference's writers (read_write.h:180-244 there, read_write.h here).
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

  // arguments of EQMDProblem::equil (init_material_problem.h:309-315) that matter once the state exists:
  // cmat, lengthof/stressof/stiffof, rep (1-based), mdts, mdtem, mdnss, mdss (strain rate), mdsa (strain amplitude), mdff
  int equil(const std::string &cmat, const std::string &lengthof, const std::string &stressof, const std::string &stiffof, int rep,
            double mdts, double mdtem, int mdnss, double mdss, double mdsa, const std::string &mdff) {
    if (mdff != "opls" && mdff != "reax") {   // init_material_problem.h:336-341 (print + exit(1) there)
      err_ = "Error: Force field is " + mdff + " but only 'opls' and 'reax' are implemented... ";
      return SCEMA_MD_ERR_ARG;
    }
    if (mdff == "reax") {
      err_ = "force field 'reax' is not built yet";
      return SCEMA_MD_ERR_ARG;
    }
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
