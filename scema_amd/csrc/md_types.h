// md_types.h -- device-visible data layout of the batched MD micro-solver (gfx950).
//
// One SimDev per simulation that is being advanced in the current launch group; every kernel's grid
// covers all simulations of the group, so one launch advances the whole batch of quadrature-point
// replicas one stage (DESIGN.md "Data layout in HBM").
#pragma once
#include <stdint.h>

#define MD_MAXCHAIN 8
#define MD_NPART 8
#define MD_MAXTYPES 16
#define MD_MAXPOLY 48
#define MD_TILE_WAVES 8       /* waves of a k_pair tile workgroup (md_pair.hip) */
#define MD_CLUSTER 4          /* atoms per i-cluster (consecutive slots inside one cell) */
#define MD_JMASK 0x007FFFFF   /* tile j-table entry: [22:0] slot of j, [27:23] image code (md_pair.hip) */
#define MD_MAXJTAB 8191       /* row entries carry a 13-bit index into the tile's j table */


// units real
#define MD_BOLTZ 0.0019872067
#define MD_MVV2E (48.88821291 * 48.88821291)
#define MD_FTM2V (1.0 / 48.88821291 / 48.88821291)
#define MD_NKTV2P 68568.415
#define MD_QQRD2E 332.06371
#define MD_EWALD_F 1.12837916709551257390
#define MD_PI 3.14159265358979323846

// bonded tiles (md_bonded.hip): kinds of terms; descriptor of one tile = BT_DESC ints:
// [0] offset into bt_atoms, [1] local atoms (owners first), [2] first chunk of 64 term descriptors, [3] chunks, [14] owners
enum { BT_BOND = 0, BT_BOND_SHAKEN = 1, BT_ANGLE = 2, BT_DIHEDRAL = 3, BT_IMPROPER = 4, BT_SPECIAL = 5, BT_NKIND = 6 };
#define BT_DESC 16           /* [2] first chunk of term descriptors, [3] chunks, [14] owners of the tile */
/* 64-bit term descriptor: local atom indices (10 bits each) | type | special-bond level | no-count flag | kind | valid */
#define BT_D_LMASK 0x3FF
#define BT_D_TYPE_SHIFT 40
#define BT_D_TMASK 0xFFF
#define BT_D_LVL_SHIFT 52
#define BT_D_NOCOUNT (1ull << 54) /* evaluated for this tile's owner forces only; virial/energy counted by the tile that owns the lowest-ranked atom */
#define BT_D_KIND_SHIFT 55
#define BT_D_VALID (1ull << 58)
#define BT_MAXCOEF 768       /* doubles of the bonded coefficient tables staged in LDS */
#define BT_OWNERS 192        /* owner atoms (consecutive breadth-first ranks of the bond graph) per tile */

enum { P_LJ = 0, P_COUL = 1, P_BOND = 2, P_ANGLE = 3, P_DIHEDRAL = 4, P_IMPROPER = 5, P_KSPACE = 6, P_SHAKE = 7 };

// mutable per-simulation scalars, resident in HBM (one cache line group per simulation)
struct SimScalars {
  double box[9];       // current xlo,ylo,zlo,xhi,yhi,zhi,xy,xz,yz
  double box_prev[9];  // box before the last fix-deform update (remap source)
  double box0[9];      // fix-deform reference box (start of the run)
  double vir[MD_NPART * 6];
  double eng[MD_NPART];
  double ke[6];
  double t_current;
  double eta[MD_MAXCHAIN + 1], eta_dot[MD_MAXCHAIN + 1], eta_dotdot[MD_MAXCHAIN + 1], eta_mass[MD_MAXCHAIN + 1];
  double vscale;       // deferred Nose-Hoover velocity scale, applied by the next kick
  double psum[6];      // running sum of the sampled pressure tensor (atm)
  double deltasq;      // neighbour trigger: (skin - corner motion)^2 / 4
  double far_dsq;      // squared atom displacement from which the far skin band (segment C2) can reach the cutoff
  double corners_hold[24];
  unsigned long long nentries;  // (i,j) pairs listed at the last build (= entries of a full per-atom list)
  unsigned long long nentries_ref;  // the same count inside the reference's list radius (cutoff + params.skin)
  unsigned long long nrowent;   // row entries actually stored (one per (cluster, j))
  int nsamples;
  int step;
  int ago;
  int check;           // ago >= delay
  int rebuild;
  int force_rebuild;   // set by a box flip: the next step rebuilds whatever the displacements
  int overflow;
  int maxneigh_seen;
  int nfar_steps;      // steps of this phase on which the far skin band was walked (SCEMA_MD_TIMING)
  int need_far;        // this step some atom moved >= sqrt(far_dsq): k_pair walks segment C2 too
  int maxj_seen;       // largest tile j table at the last builds
  int nbuilds;
  // ---- init_material's equilibration schedule only (md_equil.hip): fix npt ... iso, temperature ramps, min_style sd ----
  int keep_nh;         // set by the host before the upload: this run continues the previous one (thermostat/barostat state, step count kept)
  int nh_step;         // steps done of the whole fix (a run may be issued in segments: the cell grid follows the box)
  double t_target_now; // thermostat target of the current step (ramp t_start -> t_stop over nh_total steps)
  double omega_dot, omega_mass, mtk_term2;   // barostat (iso: the three box dimensions share one rate)
  double etap[MD_MAXCHAIN + 1], etap_dot[MD_MAXCHAIN + 1], etap_dotdot[MD_MAXCHAIN + 1], etap_mass[MD_MAXCHAIN + 1];
  double dil;          // exp(dt/2 omega_dot) of the current step: each of the two half-step remaps dilates by it
  double cen[3];       // centre of the dilation (box centre)
  double len0[3];      // box lengths at the start of the segment (the cell grid holds for +-box_margin around them)
  double lsum[3], lrun[3];   // fix ave/time of the box lengths: window sum, sum of window means
  int nlwin;
  int pppm_ticket;     // k_pppm_solve as two workgroups per replica: how many of them have read the charge grid (the last one zeroes it)
  // minimiser state (one line search at a time, decided on the device between two force evaluations)
  int min_phase, min_stop, min_iter, min_neval, min_newdir, pad2_;
  double min_alpha_now, min_alpha_next;   // where x sits on the current search line before / after the move of this evaluation (min_incremental)
  double min_alpha, min_alphamax, min_fdothall, min_eorig, min_eprev, min_fhprev, min_engprev, min_alphaprev, min_fh_trial, min_ecur, min_einit;
  double min_dots[4];  // f.h, f.f, max |f| of the last evaluation (+ spare)
#if defined(PAIR_TIMING) || defined(PAIR_COUNT)
  unsigned long long dbg[20];   // [12..17]: k_neigh_build per row: set-up, chunk loop, row end (cycles); chunks through the exclusion walk, own-cell chunks, chunks
  unsigned long long dbg2[8];   // k_pppm_solve phase clocks
#endif
};

// Every pointer of SimDev points to GLOBAL memory and says so in device code (the kernels read this structure from memory; a pointer read from
// memory is a generic one to the compiler: FLAT loads and stores, each waited for with both memory counters at zero -- md_device.h).
#if defined(__HIP_DEVICE_COMPILE__)
#define MD_G __attribute__((address_space(1)))
#else
#define MD_G
#endif
struct SimDev {
  // sizes
  int natoms, npad, ntypes;
  int nbonds, nangles, ndihedrals, nimpropers, nspecial, nclus;
  int maxneigh;               // capacity (entries) of one i-cluster row
  int capj;                   // capacity (entries) of one tile's (= cell's) j table
  int nc[3], ncells, mst[3];  // cell grid and stencil half-widths
  int nk, kmaxd[3];
  int nsteps;                 // steps of this run for this simulation
  int nav, nwin;              // fix ave/time windows (0 = no sampling)
  int nvt, use_shake, deform;
  int t_chain, neigh_delay, shake_maxiter;
  // equilibration schedule only (md_equil.hip)
  int npt;                    // 1: fix npt ... iso, 0: fix nvt; both with the ramp below
  int ramp;                   // 1: thermostat target from t_start/t_stop (else t_target)
  int nh_total;               // steps of the whole fix (ramp denominator)
  int lavg_nav;               // > 0: average the box lengths over windows of this many steps
  double t_start, t_stop, p_target, p_freq, box_margin;
  double min_etol, min_ftol, min_dmax;
  int min_maxiter, min_maxeval;
  int min_incremental;        // 1: trial points by increments x += (alpha - alpha_now) h (a force stage that wraps x at rebuilds: ReaxFF)
  // scalars
  double dt, t_target, t_freq, tdof, g_ewald, qsqsum, qsum;
  double cut_lj2, cut_coul2, rlist2, skin, excl_cut2, shake_tol;
  double rates[6];
  // real-space Ewald factor erfc(x) + 2x/sqrt(pi) exp(-x^2) = 1 - x H(u), u = x^2 = g^2 r^2:
  // H as a polynomial in t = u*coul_uscale - 1 on [-1,1] (fitted per run from g_ewald, cut_coul)
  int coul_npoly;
  double coul_uscale;
  double coul_poly[MD_MAXPOLY];
  double coul_poly_g[MD_MAXPOLY];   // the same coefficients times g_ewald (k_pair)
  int nfree;
  int keep_list;            // this run continues one that has just ended on the same slot: its cell grid and neighbour rows stand (k_phase_init)
  double rlist_ref2;      // (cutoff + the reference's skin)^2: pairs inside it are what the roofline accounting prices
  double seg_a2, seg_b2;  // row segments by build-time distance: (cut_coul+m)^2, (cut_lj+m)^2
  double far_band;        // width (A) of the near skin band C1
  double seg_c2;          // skin band split: (cutmax + skin/2)^2, beyond it segment C2 (skipped while nothing moved far enough)
  // topology (shared by all simulations of one (material, replica))
  const int MD_G *type;
  const double MD_G *q, *mass;       // per atom
  const double MD_G *lj;             // 4 * ntypes^2 : lj1,lj2,lj3,lj4
  // bonded terms: one 64-bit descriptor per term in tile order (BT_D_*), coefficient tables by type
  const unsigned long long MD_G *bt_terms;
  const double MD_G *bt_coef;          // bonds (K,r0) | angles (K,theta0) | dihedrals (K1..K4) | impropers (K,chi0)
  int bt_ncoef, bt_cf_off[4];
  int nbonds_noshake;
  double sp_w[6];                 // special_bonds weights: lj 1-2,1-3,1-4, coul 1-2,1-3,1-4
  const int MD_G *ex_start, *ex_list;
  const int MD_G *bt_desc, *bt_atoms;   // tile descriptors, local atom lists
  const int MD_G *bt_rank;              // atom -> breadth-first rank in the bond graph (index into fb)
  int bt_ntile;
  const int MD_G *clus_at, *clus_n; const double MD_G *clus_d;
  const int MD_G *free_at;              // the atoms outside the SHAKE clusters (nfree of them)
  // state
  double MD_G *x, *v, *f;
  double MD_G *fs;       // pair forces in slot order, [3][npad] (zeroed by k_pack, accumulated by k_pair, folded into f by k_ewald_force)
  double MD_G *fb;       // bonded forces in breadth-first-rank order, [natoms][3] (every entry written by the tile that owns it, k_bonded)
  double MD_G *virb;     // per bonded tile: 6 partial sums of the lumped bonded virial (no atomics; folded by k_ewald_force)
  int MD_G *slot_of;     // atom -> slot
  int MD_G *tile_nj;     // per cell: entries of its j table
  int MD_G *tile_order;  // per cell, at its cluster range: the cell's clusters grouped by the wave of k_pair that takes them
  double MD_G *virp;     // per cell and wave of k_pair: 6 partial sums of the pair virial's image-shift part (no atomics)
  int MD_G *tile_wstart; // per cell: 9 group boundaries into tile_order (k_pair's schedule, fixed at build time)
  int MD_G *tile_jtab;   // per cell: capj entries (image code | slot), own cell first
  // pair structures
  double4 MD_G *xq;      // slot records as two arrays of 16-byte halves: (x,y)[npad] then (z,q)[npad] (wrapped positions, charge)
  int MD_G *stype;       // slot-ordered type
  int MD_G *perm;        // slot -> atom
  int MD_G *slot_tmp;    // unsorted cell fill
  int MD_G *wrapn;       // atom -> integer wrap (3)
  double MD_G *xhold;
  int MD_G *cell_of, *ckey, *cell_count, *cell_start, *cell_fill;
  int MD_G *numneigh, *neigh;  // per cluster: {entries in segments A+B+C1 (front), entries in segment C2 (back)}; rows of maxneigh entries
  // ewald
  const int MD_G *kn;    // 3 ints per k
  const int MD_G *krun;  // per k: +-(number of following k-vectors that continue its row: same n1, n2; n3 + 1 or n3 - 1 each), sign = direction
  const int MD_G *kgrp;  // 8 ints per group of k-vectors (n1, +-n2, +-n3): n1, |n2|, |n3|, k index of (+,+), (-,+), (+,-), (-,-) or -1
  int ngrp;
  double MD_G *sfac;     // 2 per k
  double MD_G *kvec;     // 4 per k : kx,ky,kz,ug
  // PPPM (md_pppm.hip; pg[0] == 0: the Ewald sum above is used)
  int pg[3], pad_pg_;
  double MD_G *pgrid;    // complex grid [nz][ny][nx] (x fastest) of this simulation: charge density / its transform
  double MD_G *pfield;   // the three complex field grids of this simulation, pgstride complex elements apart (the batch keeps the charge
                    // grids of all simulations together, and all field grids: one batched, contiguous transform per direction)
  long long pgstride;
  double MD_G *pgf;      // influence function [nz][ny][nx]
  SimScalars MD_G *sc;
};
static_assert(sizeof(double MD_G *) == sizeof(double *), "the qualified pointers of SimDev have the size of plain ones: host and device passes see one layout");
