// scema_hmm_harness.cpp -- the reference's time loop (HMMProblem::run / do_timestep, dealammps.cc:417-537) around the C ABI:
// a continuum stand-in (include/scema_fe.h) produces the update_list of strained quadrature points, scema_stmd_update turns
// it into stresses, the stand-in takes them back.  Shapes of BASELINE.json configs 1 and 3: a 3x3x8 cuboid (576 quadrature
// points), 1 or 10 continuum steps.
//
//   g++ -std=c++17 -Iinclude examples/scema_hmm_harness.cpp -Lscema_amd -lscema_md -Wl,-rpath,$PWD/scema_amd -o hmm
//   ./hmm <nanoscale_input> <nanoscale_output> <macroscale_output> <material> <hooke 0|1> <nx> <ny> <nz> <steps>
//
// hooke=1 runs anywhere (the reference's "approximate md with hookes law"); hooke=0 needs an MI355X and the replica files.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "scema_fe.h"

int main(int argc, char **argv) {
  if (argc < 10) {
    std::fprintf(stderr, "usage: %s nano_in nano_out macro_out material hooke nx ny nz steps\n", argv[0]);
    return 2;
  }
  const char *mat = argv[4];
  const int hooke = std::atoi(argv[5]), nx = std::atoi(argv[6]), ny = std::atoi(argv[7]), nz = std::atoi(argv[8]), steps = std::atoi(argv[9]);
  scema_md_engine *engine = nullptr;
  if (!hooke) {
    scema_md_params p;
    scema_md_default_params(&p);
    if (scema_md_create(&p, &engine) != SCEMA_MD_OK) { std::fprintf(stderr, "no GPU: the MD path has no CPU fallback\n"); return 1; }
  }
  scema_stmd *sync = nullptr;
  if (scema_stmd_create(engine, 0, 1, nullptr, nullptr, &sync) != SCEMA_MD_OK) return 1;
  const char *materials[1] = {mat};
  scema_stmd_config cfg{};
  cfg.start_timestep = 1;
  cfg.md_timestep_length = 2.0; cfg.md_temperature = 300.0; cfg.md_nsteps_sample = 100; cfg.md_strain_rate = 1.0e-4;   // inputs_dogbone_cuboid.json:50-53
  cfg.md_force_field = "opls";
  cfg.nanostatelocin = argv[1]; cfg.nanostatelocout = argv[2]; cfg.nanostatelocres = argv[2]; cfg.nanologloc = "none";
  cfg.macrostatelocout = argv[3]; cfg.md_scripts_directory = "";
  cfg.freq_checkpoint = 100; cfg.freq_output_homog = 1000;
  cfg.n_materials = 1; cfg.mdtype = materials;
  cfg.cg_dir[0] = 1.0;
  cfg.nrepl = 1;
  cfg.approx_md_with_hookes_law = hooke;
  if (scema_stmd_init(sync, &cfg) != SCEMA_MD_OK) { std::fprintf(stderr, "init failed: %s\n", scema_stmd_last_error(sync)); return 1; }

  // the continuum reads what STMDSync::init averaged over the replicas (FE_problem.h:411,428): init.<mat>.stiff / .density
  scema_fe_config fc{};
  fc.nx = nx; fc.ny = ny; fc.nz = nz;
  fc.lx = 0.01 * nx; fc.ly = 0.01 * ny; fc.lz = 0.01 * nz;
  {
    const std::string base = std::string(argv[3]) + "/init." + mat;
    FILE *fs = std::fopen((base + ".stiff").c_str(), "r"), *fd = std::fopen((base + ".density").c_str(), "r");
    if (!fs || !fd) { std::fprintf(stderr, "cannot read %s.{stiff,density}\n", base.c_str()); return 1; }
    for (int k = 0; k < 36; k++) if (std::fscanf(fs, "%lf", &fc.stiffness[k]) != 1) return 1;
    if (std::fscanf(fd, "%lf", &fc.density) != 1) return 1;
    std::fclose(fs); std::fclose(fd);
  }
  fc.dt = 5.0e-7;                // "continuum time.timestep length"
  fc.top_velocity = 0.002 * fc.lz / fc.dt * 0.1;   // a tenth of the nominal 0.002 strain per step over the whole bar, applied at the loaded face
  fc.min_qp_strain = 1.0e-10;
  fc.hooke = hooke;
  scema_fe *fe = nullptr;
  if (scema_fe_create(&fc, &fe) != SCEMA_MD_OK) return 1;
  std::vector<scema_qp> update_list(scema_fe_n_qp(fe));
  std::vector<double> stress(6 * (size_t)scema_fe_n_qp(fe));
  for (int step = 1; step <= steps; step++) {
    int32_t n = 0;
    if (scema_fe_solve(fe, update_list.data(), (int32_t)update_list.size(), &n) != SCEMA_MD_OK) return 1;
    if (n > 0 && scema_stmd_update(sync, step, step * fc.dt, 1, update_list.data(), n) != SCEMA_MD_OK) {
      std::fprintf(stderr, "update failed: %s\n", scema_stmd_last_error(sync));
      return 1;
    }
    if (scema_fe_check(fe, update_list.data(), n) != SCEMA_MD_OK) return 1;
    scema_fe_get(fe, nullptr, nullptr, stress.data());
    double smax = 0.0;
    for (double s : stress) smax = std::fmax(smax, std::fabs(s));
    std::printf("step %d md_updates %d max_stress_Pa %.9e\n", step, n, smax);
  }
  scema_fe_destroy(fe);
  scema_stmd_destroy(sync);
  if (engine) scema_md_destroy(engine);
  return 0;
}
