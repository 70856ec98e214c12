// md_pppm.h -- host-callable launch wrappers of the PPPM kernels (md_pppm.hip); the transforms between them are hipFFT calls
// issued by the engine (engine/engine_run.cpp pppm_stage)
#pragma once
#include <hip/hip_runtime.h>
struct SimDev;
size_t mdk_pppm_lds_limit();
// charges -> grid 0 (complex, imaginary part 0); maxgrid = largest nx*ny*nz of the batch
// zeroed != 0: the charge grids are known to hold zeros (k_pppm_solve leaves them so)
void mdk_pppm_spread(hipStream_t st, const SimDev *d, int ns, int maxgrid, int maxatoms, int zeroed, int maxgridp = 0);   // maxgridp: largest grid with 5 more points per x row (0: no padded LDS copy)
// small grids (maxgrid <= mdk_pppm_solve_max()): forward transform, energy / virial / field spectra and the three inverse transforms in one
// launch, in LDS (replaces the transforms of the engine and mdk_pppm_poisson); maxdims = largest nx + ny + nz of the batch
int mdk_pppm_solve_max();
void mdk_pppm_solve(hipStream_t st, const SimDev *d, int ns, int maxgrid, int maxdims);
// influence function of the current box into SimDev::pgf
void mdk_pppm_gf(hipStream_t st, const SimDev *d, int ns, int maxgrid);
// after the forward transform of grid 0: energy, virial, field spectra into grids 1..3
void mdk_pppm_poisson(hipStream_t st, const SimDev *d, int ns, int maxgrid);
// after the inverse transforms of the field grids: forces added to SimDev::f (add != 0) or stored there (the chain runs ahead of
// the kernel that assembles the force of the step, which then adds them: mdk_ewald_force fkeep)
void mdk_pppm_force(hipStream_t st, const SimDev *d, int ns, int maxgrid, int maxatoms, int add, int real_fields = 0);   // real_fields: after mdk_pppm_solve
