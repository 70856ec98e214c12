// rx_types.h -- data layout of the ReaxFF force field path (BASELINE config 5): the parameter
// tables of one force-field file and the per-replica work arrays in HBM.
//
// What it replaces in the reference: `pair_style reax/c NULL safezone 50.0 mincap 100000`, `pair_coeff * * ffield.reax.2 H C N O`,
// `fix qeq/reax 1 0.0 10.0 1e-6 reax/c` (lammps_scripts/lammps_scripts_reax/in.strain.lammps:10-12,
// ELASTIC/potential.mod.lammps:5-7), i.e. LAMMPS' USER-REAXC package [LAMMPS-ext].
//
// Rows: every per-atom list (neighbours, bonds) is stored entry-major, element (k, i) at [k * npad + i], so that a wave
// whose lanes are consecutive atoms reads consecutive addresses at every step k of its row walk.
#pragma once
#include <stdint.h>

#define RX_MAXT 6            /* force-field types kept (elements named by pair_coeff) */
#define RX_MAXANG 4          /* parameter sets per valence-angle triple */
#define RX_NGP 40
#define RX_KS 8               /* waves of a workgroup of the charge-equilibration kernels: 64 rows per workgroup, 8 per wave */
#define RX_QEQ_COLD_SOLVES 4   /* solves of a run that starts from an empty history which count as cold (the extrapolation uses four past solutions) */
#ifndef RX_SWR   /* (at most 64: one wave does the row-local part of the sweep, a row per lane) */
#define RX_SWR 64             /* rows per workgroup of the matrix sweep (8 per wave): the workgroup stages the gathered vector in LDS once for all of them (32: 400 against 410 evaluations/s, profiles/r04_e_*) */
#endif
#define RX_PM_MAX 8            /* entries of a preconditioner row: the atom and up to 7 neighbours within RX_PM_RADIUS (the nearest in list order) */
#define RX_PM_RADIUS 2.0       /* Angstrom: the bonded neighbours; gated offline (tools/qeq_precond_gate.py, profiles/r05_qeq_precond_gate.txt):
                                  1.7 .. 2.5 A save 58 .. 62 % of the iterations of the Jacobi preconditioner on PE-1620, 3.0 A only 17 % */
#define RX_JMASK 0x00FFFFFF  /* row entry: [23:0] atom, [30:24] image code (sx+2) + 5 (sy+2) + 25 (sz+2) */
#define RX_CODE0 62          /* code of the zero shift */

#define RX_C_ELE 332.06371
#define RX_KCALPMOL_TO_EV 23.02
#define RX_EV_TO_KCALPMOL 14.4
#define RX_THB_CUT 0.001
#define RX_THB_CUTSQ 0.00001
#define RX_HB_THRESHOLD 1e-2
#define RX_BOND_CUT 5.0
#define RX_HBOND_CUT 7.5
#define RX_MIN_SINE 1e-10

// energy parts (the order of the oracle's, so that tests compare part by part)
enum { RX_E_BOND = 0, RX_E_LP, RX_E_OVER, RX_E_UNDER, RX_E_ANGLE, RX_E_PEN, RX_E_COA, RX_E_TORS, RX_E_CONJ, RX_E_HB, RX_E_VDW, RX_E_COUL, RX_E_POL, RX_NPART };

typedef struct {
  double r_s, valency, mass, r_vdw, epsilon, gamma, r_pi, valency_e, nlp_opt;
  double alpha, gamma_w, valency_boc, p_ovun5, chi, eta;
  double r_pi_pi, p_lp2, b_o_131, b_o_132, b_o_133;
  double p_ovun2, p_val3, valency_val, p_val5;
  int p_hbond, pad_;
} RxSbp;
typedef struct {
  double De_s, De_p, De_pp, p_be1, p_bo5, v13cor, p_bo6, p_ovun1, p_be2, p_bo3, p_bo4, p_bo1, p_bo2, ovc;
  double r_s, r_p, r_pp, p_boc3, p_boc4, p_boc5, D, alpha, r_vdW, gamma_w, gamma;
  double powgw;           // (1/gamma_w)^p_vdW1: the shielding constant of the van der Waals term (derived by the reader)
  double lr_s, lr_p, lr_pp;   // log r_s, log r_p, log r_pp: (r / r_x)^p = exp(p (log r - log r_x)), one logarithm per bond-order entry (derived by the reader)
  double inv_rvdw;        // 1 / r_vdW (derived by the reader)
} RxTbp;
typedef struct { double theta_00, p_val1, p_val2, p_coa1, p_val7, p_pen1, p_val4; } RxThbPrm;
typedef struct { int cnt, pad_; RxThbPrm prm[RX_MAXANG]; } RxThbp;
typedef struct { int cnt, pad_; double V1, V2, V3, p_tor1, p_cot1; } RxFbp;
typedef struct { double r0_hb, p_hb1, p_hb2, p_hb3; } RxHbp;

typedef struct {
  int nt;                 // types kept
  int lammps_dsbo2;       // 1: the valence-angle term drops d(SBO)/d(Delta) when vlpex >= 0 (a reading of USER-REAXC's Valence_Angles that
                          // could not be checked and breaks energy conservation; off by default)
  double gp[RX_NGP];
  double bo_cut, swa, swb;
  double inv_pvdw1;       // 1 / gp[28] (derived by the reader)
  double tap[8];
  RxSbp sbp[RX_MAXT];
  RxTbp tbp[RX_MAXT * RX_MAXT];
  // per type pair: the distance beyond which the uncorrected bond order is below bo_cut for certain (it falls with the distance: found by
  // bisection when the force field is loaded, never above RX_BOND_CUT), and (that or RX_PM_RADIUS, whichever is larger, + list skin)^2 --
  // the radius of the NEAR rows, which hold the candidates of the bond-order pass and of the preconditioner's pattern
  double rbond[RX_MAXT * RX_MAXT], rnear2[RX_MAXT * RX_MAXT];
  RxThbp thbp[RX_MAXT * RX_MAXT * RX_MAXT];
  RxFbp fbp[RX_MAXT * RX_MAXT * RX_MAXT * RX_MAXT];
  RxHbp hbp[RX_MAXT * RX_MAXT * RX_MAXT];
} RxParams;

// Every pointer of the work set is a pointer to GLOBAL memory, and says so in device code: the kernels read this structure from memory, and a
// pointer read from memory is a generic one to the compiler -- FLAT loads and stores, each waited for with both memory counters at zero
// (md_device.h).  With the qualifier every access through the view is a global_load / global_store / global_atomic.
#if defined(__HIP_DEVICE_COMPILE__)
#define RX_G __attribute__((address_space(1)))
#else
#define RX_G
#endif
// one replica's ReaxFF work set
typedef struct {
  int n, npad;            // atoms, row stride (n rounded up to 64)
  int maxnb, maxbd;       // row capacities (entries)
  double h[6], lo[3];     // box: lx, ly, lz, yz, xz, xy (LAMMPS h order) and origin
  const int RX_G *rtype;       // [n] force-field type of every atom
  const double RX_G *x;        // [n][3] positions (unwrapped)
  double RX_G *q;              // [n] charges (charge equilibration)
  // neighbour rows inside the list radius (full: j appears in i's row and i in j's)
  int RX_G *nb_cnt;            // [n]
  int RX_G *nb;                // [maxnb][npad] entry-major rows: the host test driver only (NULL on the device, which keeps nbT; rx_nb_entry)
  // near rows: the entries of nb inside the bond cutoff + skin, same order (the bond-order pass walks these; NULL: walk nb)
  int RX_G *nbn_cnt;           // [n]
  int RX_G *nbn;               // [maxnbn][npad]
  int RX_G *nbnT;              // [npad][maxnbn] the near rows once more, row-major (the bond-order pass puts a wave on a row)
  int maxnbn, pad0_;
  double rnear2;          // the largest near-row radius of the force field, squared (RxParams::rnear2 holds the one of each type pair)
  // bond rows: pairs with BO' >= cutoff (full)
  int RX_G *bd_cnt;            // [n]
  int RX_G *bd;                // [maxbd][npad] atom | image code
  int RX_G *bd_rev;            // [maxbd][npad] slot of this atom in the partner's row
  double RX_G *bd_bop;         // [4][maxbd][npad] uncorrected: BO' (total, cutoff taken off), BO'_pi, BO'_pi2, r
  double RX_G *bd_c;           // [3][maxbd][npad] dBO'_s/dd = c_s d, dBO'_pi/dd = c_pi d, dBO'_pi2/dd = c_pi2 d
  double RX_G *bd_bo;          // [3][maxbd][npad] corrected: BO, BO_pi, BO_pi2
  double RX_G *bd_g;           // [3][maxbd][npad] dE/d(BO, BO_pi, BO_pi2) gathered by the energy terms (each end into its own or the partner's row)
  double RX_G *bd_cb;          // [maxbd][npad] after the back-propagation: coefficient of d in the bond force, Delta' part aside
  double RX_G *deltap;         // [n] Delta'_i = sum BO' - valency
  double RX_G *total_bo;       // [n] sum of corrected bond orders
  double RX_G *cd_delta;       // [n] dE/d(Delta_i)
  double RX_G *hd;             // [n] dE/d(Delta'_i)
  double RX_G *f;              // [n][3]
  // charge equilibration
  // the matrix of the charge equilibration, rebuilt every step, ROW-MAJOR (row i at i * maxnb): only the row entries inside the taper
  // radius (about 70 % of a row of the list), in list order; the kernels that walk it put the lanes of a wave over the entries of one
  // row (contiguous loads) and reduce across the wave
  // Replicas of up to 65 536 atoms keep an entry in ONE 64-bit word (hpk): the column in bits 0-15, the value's upper 48 bits (rounded to
  // nearest: relative error <= 2^-37 = 7e-12, five orders below the 1e-6 the solve stops at) in bits 16-63 -- 8 instead of 10 bytes per entry
  // for the HBM-bound matrix sweep, one load instead of two.  Larger replicas (and the host checks): hval + hcol32 (hpk NULL).
  unsigned long long RX_G *hpk; // [npad][maxnb]
  double RX_G *hval;           // [npad][maxnb] H_ij
  unsigned short RX_G *hcol16; // [npad][maxnb] column (atom index) of the entry; replicas of up to 65 536 atoms (else NULL and hcol32)
  int RX_G *hcol32;
  int RX_G *hlen;              // [npad] entries of the row
  int RX_G *hown;              // [npad][maxnb] the list entries (atom | image code) of the row's pairs inside the taper radius that this
                          // end owns (rx_owns: each pair once), compacted; the non-bonded pass walks these
  int RX_G *hownlen;           // [npad]
  int RX_G *nbT;               // [npad][maxnb] the list rows, row-major (a wave per row writes and reads them)
  int RX_G *nb_own0;           // [npad] minimum-image rows are sorted by partner: the entries from here on are the pairs this end owns (rx_owns)
  double RX_G *s, *t;          // [npad] the two solutions
  double RX_G *s_hist, *t_hist;  // [4][npad] and [3][npad]: previous solutions, newest first (initial guesses are extrapolated from them)
  double RX_G *qwork;          // [10][npad]: five arrays of (s-system, t-system) pairs per atom: residual r, search direction d,
                          // matrix-vector product q = H d, preconditioned residual z = M r (the vector the matrix sweeps gather), and a
                          // second residual array (the update of iteration `it` reads r of parity it & 1 and writes the other one: the
                          // preconditioner gathers the NEW residuals of an atom's bonded neighbours, which other workgroups own)
  // preconditioner of the conjugate gradients (round 5): a sparse approximate inverse on the bonded pattern -- row i of M = row i of
  // the inverse of H restricted to i and its neighbours within RX_PM_RADIUS, symmetrised; pm_on = 0: the Jacobi one of fix qeq/reax
  int pm_on, pm_pad_;
  int RX_G *pm_len;            // [npad] entries of the row (the atom itself first)
  int RX_G *pm_col;            // [RX_PM_MAX][npad]
  double RX_G *pm_raw;         // [RX_PM_MAX][npad] rows of the local inverses
  double RX_G *pm_val;         // [RX_PM_MAX][npad] symmetrised: the preconditioner
  double RX_G *qpart;          // per-block partial sums of the solver's scalar products (layout: md_reax.hip)
  int RX_G *qstat;             // [6] since the start of the run: iterations, solves, most iterations in one solve (cold solves aside), solves
                          // finished by the single-workgroup loop, (scratch), most iterations in one of the run's first (cold) solves
  int warm;               // the history arrays hold the solutions of the run this one continues (kept, not zeroed, at the start)
  double RX_G *eparts;         // [RX_NPART] energy parts of the step
  int mimg[3];            // neighbour search: 0,0,0 = minimum image (box at least two list radii wide), else images up to mimg[d] boxes away
  int RX_G *overflow;          // bit 1: neighbour row full, bit 2: bond row full, bit 4: charge equilibration did not converge
  long long RX_G *sweep_acc;   // [2] since the start of the run: matrix entries and rows that launches of k_rx_qeq_sweep passed over for this
                          // replica (stored entries of its rows x sweeps it took part in): the kernel's algorithmic traffic (bench.py)
} RxView;
#ifdef __cplusplus
static_assert(sizeof(double RX_G *) == sizeof(double *), "the qualified pointers of RxView have the size of plain ones: host and device passes see one layout");
#endif
