// scema_harness.cpp -- a C++ caller of the C ABI, shaped like the reference's call site
// HMMProblem::do_timestep (dealammps.cc:417-474): build an update_list of QP records, call
// update(), read the stresses back.  Links against scema_amd/libscema_md.so only.
//
//   g++ -std=c++17 -Iinclude examples/scema_harness.cpp -Lscema_amd -lscema_md -Wl,-rpath,$PWD/scema_amd -o harness
//   ./harness <nanoscale_input> <nanoscale_output> <macroscale_output> <material> <nrepl> <hooke 0|1> <n_qp>
//
// With hooke=1 this runs anywhere (the reference's "approximate md with hookes law" test mode); with
// hooke=0 it needs an MI355X and init.<mat>_<r>.bin replica containers in <nanoscale_input>.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "scema_stmd.h"

int main(int argc, char **argv) {
  if (argc < 8) {
    std::fprintf(stderr, "usage: %s nano_in nano_out macro_out material nrepl hooke n_qp\n", argv[0]);
    return 2;
  }
  const char *mat = argv[4];
  const int nrepl = std::atoi(argv[5]), hooke = std::atoi(argv[6]), n_qp = std::atoi(argv[7]);

  scema_md_engine *engine = nullptr;
  if (!hooke) {
    scema_md_params p;
    scema_md_default_params(&p);   // the in.set.lammps settings
    if (scema_md_create(&p, &engine) != SCEMA_MD_OK) {
      std::fprintf(stderr, "no GPU: the MD path has no CPU fallback\n");
      return 1;
    }
  }
  scema_stmd *sync = nullptr;
  if (scema_stmd_create(engine, /*rank=*/0, /*world=*/1, nullptr, nullptr, &sync) != SCEMA_MD_OK) return 1;

  const char *materials[1] = {mat};
  scema_stmd_config cfg{};
  cfg.start_timestep = 1;
  cfg.md_timestep_length = 2.0;     // inputs_dogbone_cuboid.json:50-53
  cfg.md_temperature = 300.0;
  cfg.md_nsteps_sample = 100;
  cfg.md_strain_rate = 1.0e-4;
  cfg.md_force_field = "opls";
  cfg.nanostatelocin = argv[1];
  cfg.nanostatelocout = argv[2];
  cfg.nanostatelocres = argv[2];
  cfg.nanologloc = "none";
  cfg.macrostatelocout = argv[3];
  cfg.md_scripts_directory = "";
  cfg.freq_checkpoint = 100;
  cfg.freq_output_homog = 1000;
  cfg.n_materials = 1;
  cfg.mdtype = materials;
  cfg.cg_dir[0] = 1.0; cfg.cg_dir[1] = 0.0; cfg.cg_dir[2] = 0.0;
  cfg.nrepl = nrepl;
  cfg.use_pjm_scheduler = 0;
  cfg.approx_md_with_hookes_law = hooke;
  cfg.verbose = 0;
  if (scema_stmd_init(sync, &cfg) != SCEMA_MD_OK) {
    std::fprintf(stderr, "init failed: %s\n", scema_stmd_last_error(sync));
    return 1;
  }

  // what FEProblem::write_md_updates_list produces (FE_problem.h:1296-1373): ids, material, strain
  std::vector<scema_qp> update_list(n_qp);
  for (int q = 0; q < n_qp; q++) {
    update_list[q].id = q;
    update_list[q].most_recent_id = SCEMA_MD_QP_NONE;
    update_list[q].material = 0;
    const double ezz = 1.0e-3 + 1.0e-4 * q;
    const double e[6] = {-0.3 * ezz, -0.3 * ezz, ezz, 1.0e-5 * q, 0.0, -2.0e-5};
    for (int k = 0; k < 6; k++) { update_list[q].update_strain[k] = e[k]; update_list[q].update_stress[k] = 0.0; }
  }
  if (scema_stmd_update(sync, /*timestep=*/1, /*time=*/5.0e-7, /*newtonstep=*/1, update_list.data(), n_qp) != SCEMA_MD_OK) {
    std::fprintf(stderr, "update failed: %s\n", scema_stmd_last_error(sync));
    return 1;
  }
  for (int q = 0; q < n_qp; q++) {
    std::printf("qp %d stress", update_list[q].id);
    for (int k = 0; k < 6; k++) std::printf(" %.16g", update_list[q].update_stress[k]);
    std::printf("\n");
  }
  scema_stmd_destroy(sync);
  if (engine) scema_md_destroy(engine);
  return 0;
}
