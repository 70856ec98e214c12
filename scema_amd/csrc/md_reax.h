// md_reax.h -- host-callable launch wrappers of the ReaxFF kernels (md_reax.hip)
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "reax/rx_types.h"
struct SimDev;
// start of a run: flags and counters; the charge-equilibration history is zeroed for the replicas that do not keep theirs (RxView::warm)
void mdk_reax_phase_init(hipStream_t st, const RxView *v, int ns, int maxpad);
// how a solve of the charge equilibration is issued as launches over the batch
struct RxQeqPlan {
  int launch = 0;  // conjugate-gradient iterations issued as launches over the batch (replicas that need more finish inside k_rx_qeq_finish)
  int setup = 0;   // the solve of a run's step 0 (a replica that kept its history starts from the newest solution)
  int precond = 0; // the batch's views have pm_on set: build the sparse approximate inverse before the solve
  int sym = 0;     // the symmetric form of the solve (every replica of the batch: minimum image, 16-bit columns, two vectors of the replica in LDS)
};
// the whole force stage of one step for the first ns replicas: (neighbour rows if the rebuild flag of the step is set,)
// charge equilibration, bond orders, energy terms, forces into SimDev::f, virial and energies into SimScalars.
// terms: bit 0 bond/lone pair/over/under, 1 angles, 2 torsions, 3 hydrogen bonds, 4 non-bonded (31 = all; parity hook)
// side: when given, the bond-order chain (bond orders, corrections, bonded terms, back-propagation) runs on a second stream next to the charge chain
// (matrix rows, conjugate gradients, non-bonded pairs): neither reads what the other writes until the forces are summed (see md_reax.hip)
struct RxSide { hipStream_t st2; hipEvent_t fork, mid, join; };
void mdk_reax_forces(hipStream_t st, const SimDev *d, RxView *v, const RxParams *P, int ns, int maxatoms, double rlist, double qeq_tol, int qeq_maxiter, const RxQeqPlan &plan, int terms,
                     bool col16, std::vector<hipEvent_t> *sweep_events = nullptr, size_t *sweep_events_used = nullptr, const RxSide *side = nullptr);
// sweep_events: when given, a HIP-event pair is recorded around every launch of k_rx_qeq_sweep (pool grown on demand)
// the solver statistics of the batch in one array (8 words per replica: RxView::qstat[0..5], RxView::sweep_acc[0..1]) for ONE read-back
void mdk_reax_collect_stats(hipStream_t st, const RxView *v, int ns, long long *out);
