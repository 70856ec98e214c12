// md_equil.h -- host-callable launch wrappers of the equilibration-schedule kernels (md_equil.hip)
#pragma once
#include <hip/hip_runtime.h>
struct SimDev;
// fix nvt / fix npt ... iso with a temperature ramp: the per-step bookkeeping around the production force kernels
void mdk_setup_post_nh(hipStream_t st, const SimDev *d, int ns);
void mdk_pre_nh(hipStream_t st, const SimDev *d, int ns);
void mdk_initial_integrate_nh(hipStream_t st, const SimDev *d, int ns, int maxatoms);
void mdk_post_nh(hipStream_t st, const SimDev *d, int ns);
// min_style sd: accumulators, trial point x = x0 + alpha h (x0s, hs: per replica device arrays [3 natoms]), scalar products,
// the line-search decision of every replica
void mdk_min_pre(hipStream_t st, const SimDev *d, int ns);
void mdk_min_move(hipStream_t st, const SimDev *d, int ns, int maxatoms, double *const *x0s, double *const *hs);
void mdk_min_reduce(hipStream_t st, const SimDev *d, int ns, int maxatoms, double *const *hs);
void mdk_min_decide(hipStream_t st, const SimDev *d, int ns);
// change_box ... remap of one replica (box_old, box_new: 9 doubles each, on the device)
void mdk_change_box(hipStream_t st, double *x, int natoms, const double *box_old, const double *box_new);
