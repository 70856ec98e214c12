// md_kernels.h -- host-callable launch wrappers of the gfx950 kernels (md_kernels.hip, md_pair.hip)
#pragma once
#include <hip/hip_runtime.h>
struct SimDev;
void mdk_phase_init(hipStream_t st, const SimDev *d, int ns);
void mdk_keep_validate(hipStream_t st, const SimDev *d, int ns, int maxatoms);
void mdk_setup_post(hipStream_t st, const SimDev *d, int ns);
void mdk_pre(hipStream_t st, const SimDev *d, int ns);
void mdk_initial_integrate(hipStream_t st, const SimDev *d, int ns, int maxatoms, bool pack = false);   // pack: also the slot records of k_pair (then no k_pack)
void mdk_neighbor(hipStream_t st, const SimDev *d, int ns, int maxatoms, int maxpad, int maxcells, int maxrow, int capj, bool pack = true, bool together = false);   // together: as soon as one replica of the launch rebuilds its rows, all do (k_cell_build)
void mdk_neigh_build(hipStream_t st, const SimDev *d, int ns, int maxcells, int maxrow, int capj);
// dynamic LDS of the tile kernels for a j-table capacity (the engine sizes the cell grid so that these fit)
size_t mdk_pair_lds_bytes(int capj);
size_t mdk_neigh_lds_bytes(int capj, int maxrow);
// vir: accumulate the pair virial (needed when the pressure is sampled); eng: also energies (parity hook)
void mdk_pair(hipStream_t st, const SimDev *d, int ns, int maxcells, int capj, int vir, int eng, int npoly, int cle = 0);   // cle: cut_coul <= cut_lj for the whole batch
// bonded terms + special pairs, one workgroup per bonded tile; parts != 0: per-part virial/energy (parity hook)
void mdk_bonded(hipStream_t st, const SimDev *d, int ns, int maxtiles, int maxloc, int maxcoef, int parts);
// reciprocal Ewald sum in two parts, so that the first (structure factors; needs only positions) can run on a
// second stream next to the bonded kernel.  pairvir != 0: k_ewald_force also folds the production pair virial
// (slot-ordered forces x positions + the per-wave image-shift partials of k_pair) into the virial of the step
void mdk_ewald_recip(hipStream_t st, const SimDev *d, int ns, int maxk, int mmax, int maxgrp);
void mdk_ewald_force(hipStream_t st, const SimDev *d, int ns, int maxatoms, int pairvir, int fkeep = 0);   // fkeep: add to the forces a PPPM chain left in f
void mdk_shake(hipStream_t st, const SimDev *d, int ns, int maxclus, double dtfsq_scale);
// assembly of f + fix shake + second half-kick in one pass (steps without a per-atom reciprocal sum); fkeep: PPPM forces wait in f
void mdk_finish(hipStream_t st, const SimDev *d, int ns, int maxunits, int pairvir, int fkeep);
void mdk_final_integrate(hipStream_t st, const SimDev *d, int ns, int maxatoms, int kick);
void mdk_post(hipStream_t st, const SimDev *d, int ns, int next_pre = 0);
void mdk_remap(hipStream_t st, const SimDev *d, int ns, int maxatoms);
// triclinic box flip of ONE simulation between two steps (fix deform, flip yes): new tilts, forced list rebuild
void mdk_flip(hipStream_t st, const SimDev *sim, double xy, double xz, double yz);
void mdk_phase_end(hipStream_t st, const SimDev *d, int ns, int maxatoms);
// ncopies device-to-device copies of doubles in one launch; the table lives in device memory, maxn = the longest copy
struct MdkCopy { const double *src; double *dst; long long n; };
void mdk_copy_many(hipStream_t st, const MdkCopy *tab, int ncopies, long long maxn);
// many zero fills in one launch (n 32-bit words each): a hipMemsetAsync per replica costs a launch gap each
struct MdkZero { int *p; long long n; };
void mdk_zero_many(hipStream_t st, const MdkZero *tab, int nfills, long long maxn);
