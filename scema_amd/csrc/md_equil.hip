// md_equil.hip -- kernels of init_material's equilibration schedule: what
// lammps_scripts_opls/in.init.lammps:44-215 asks LAMMPS for besides the force field itself [LAMMPS-ext: restated from the
// documented behaviour of 17Nov16's fix_nh.cpp, min_sd.cpp, min_linesearch.cpp; parity unpinned like the rest of the MD path]:
//   fix npt temp T0 T1 100.0 iso 1.0 1.0 1000   Nose-Hoover chains on particles and on the (isotropic) box rate, MTK terms,
//                                               tilt factors scaled with their box lengths, temperature ramp
//   fix nvt temp T0 T1 100.0                    the same integrator without the barostat (k_pre/k_post of md_kernels.hip hold a
//                                               constant target)
//   fix ave/time 1 nav nav v_lx v_ly v_lz ave running
//   min_style sd ; minimize etol ftol maxiter maxeval   steepest descent with the quadratic line search (dmax 0.1)
// Forces come from the production kernels (md_pair.hip, md_bonded.hip, md_kernels.hip); only the per-step bookkeeping differs.
// One launch covers every replica of the batch; the line search of every replica is decided on the device between two force
// evaluations (k_min_decide), so replicas of one batch minimise independently without host round trips.
// oracle/md_oracle.c (omd_run_nh, omd_minimize) is the CPU restatement these are tested against.
#include <hip/hip_runtime.h>

#include "md_device.h"
#include "md_equil.h"
#include "md_types.h"

// ------------------------------------------------------------------------------------------
// fix nvt / fix npt
// ------------------------------------------------------------------------------------------
__device__ inline double pressure_scalar(const SimDev &S, const SimScalars &sc) {
  BoxD b;
  box_derive(sc.box, b);
  double w = 0.0;
  for (int p = 0; p < MD_NPART; p++) w += sc.vir[p * 6] + sc.vir[p * 6 + 1] + sc.vir[p * 6 + 2];
  return (S.tdof * MD_BOLTZ * sc.t_current + w) / (3.0 * b.vol) * MD_NKTV2P;
}
// nhc_press_integrate (one sub-cycle, no drag); iso: the three box dimensions carry the same omega_dot, each with its mass term
__device__ inline void nhc_press_half(const SimDev &S, SimScalars &sc, double t_target) {
  const int mp = 3;
  const double dt = S.dt, dthalf = 0.5 * dt, dt4 = 0.25 * dt, dt8 = 0.125 * dt, kt = MD_BOLTZ * t_target;
  double kecurrent = 3.0 * sc.omega_mass * sc.omega_dot * sc.omega_dot;
  sc.etap_dotdot[0] = (kecurrent - kt) / sc.etap_mass[0];
  double expfac;
  for (int k = mp - 1; k > 0; k--) {
    expfac = exp(-dt8 * sc.etap_dot[k + 1]);
    sc.etap_dot[k] *= expfac;
    sc.etap_dot[k] += sc.etap_dotdot[k] * dt4;
    sc.etap_dot[k] *= expfac;
  }
  expfac = exp(-dt8 * sc.etap_dot[1]);
  sc.etap_dot[0] *= expfac;
  sc.etap_dot[0] += sc.etap_dotdot[0] * dt4;
  sc.etap_dot[0] *= expfac;
  for (int k = 0; k < mp; k++) sc.etap[k] += dthalf * sc.etap_dot[k];
  sc.omega_dot *= exp(-dthalf * sc.etap_dot[0]);
  kecurrent = 3.0 * sc.omega_mass * sc.omega_dot * sc.omega_dot;
  sc.etap_dotdot[0] = (kecurrent - kt) / sc.etap_mass[0];
  sc.etap_dot[0] *= expfac;
  sc.etap_dot[0] += sc.etap_dotdot[0] * dt4;
  sc.etap_dot[0] *= expfac;
  for (int k = 1; k < mp; k++) {
    expfac = exp(-dt8 * sc.etap_dot[k + 1]);
    sc.etap_dot[k] *= expfac;
    sc.etap_dotdot[k] = (sc.etap_mass[k - 1] * sc.etap_dot[k - 1] * sc.etap_dot[k - 1] - kt) / sc.etap_mass[k];
    sc.etap_dot[k] += sc.etap_dotdot[k] * dt4;
    sc.etap_dot[k] *= expfac;
  }
}
__device__ inline void nh_omega_dot(const SimDev &S, SimScalars &sc) {
  BoxD b;
  box_derive(sc.box, b);
  const double p_current = pressure_scalar(S, sc);
  const double mtk_term1 = S.tdof * MD_BOLTZ * sc.t_current / (3.0 * S.natoms);
  const double f_omega = (p_current - S.p_target) * b.vol / (sc.omega_mass * MD_NKTV2P) + mtk_term1 / sc.omega_mass;
  sc.omega_dot += f_omega * 0.5 * S.dt;
  sc.mtk_term2 = 3.0 * sc.omega_dot / (3.0 * S.natoms);
}

// after the step-0 force evaluation of a run: fix setup (t_current, masses; a continued run keeps its state)
__global__ void k_setup_post_nh(const SimDev *sims) {
  const SimDev &S = sims[blockIdx.x];
  SimScalars &sc = *S.sc;
  if (threadIdx.x != 0) return;
  for (int d = 0; d < 3; d++) sc.len0[d] = sc.box[3 + d] - sc.box[d];
  if (sc.keep_nh) return;
  sc.t_current = (sc.ke[0] + sc.ke[1] + sc.ke[2]) / (S.tdof * MD_BOLTZ);
  sc.nh_step = 0;
  sc.t_target_now = S.t_start;
  const double tf2 = S.t_freq * S.t_freq, kt = MD_BOLTZ * S.t_start;
  sc.eta_mass[0] = S.tdof * kt / tf2;
  for (int k = 1; k < S.t_chain; k++) sc.eta_mass[k] = kt / tf2;
  for (int k = 1; k < S.t_chain; k++) sc.eta_dotdot[k] = (sc.eta_mass[k - 1] * sc.eta_dot[k - 1] * sc.eta_dot[k - 1] - kt) / sc.eta_mass[k];
  sc.omega_dot = 0.0; sc.mtk_term2 = 0.0; sc.dil = 1.0;
  for (int k = 0; k <= MD_MAXCHAIN; k++) sc.etap[k] = sc.etap_dot[k] = sc.etap_dotdot[k] = sc.etap_mass[k] = 0.0;
  for (int d = 0; d < 3; d++) { sc.lsum[d] = 0.0; sc.lrun[d] = 0.0; }
  sc.nlwin = 0;
  if (S.npt) {
    const double pf2 = S.p_freq * S.p_freq;
    sc.omega_mass = (S.natoms + 1) * kt / pf2;   // fixed for the run
    for (int k = 0; k < 3; k++) sc.etap_mass[k] = kt / pf2;
    for (int k = 1; k < 3; k++) sc.etap_dotdot[k] = (sc.etap_mass[k - 1] * sc.etap_dot[k - 1] * sc.etap_dot[k - 1] - kt) / sc.etap_mass[k];
  }
}

// beginning of a step: k_pre of md_kernels.hip + barostat chain, ramped target, barostat kick of the velocities, box dilation
__global__ void k_pre_nh(const SimDev *sims) {
  const SimDev &S = sims[blockIdx.x];
  SimScalars &sc = *S.sc;
  if (threadIdx.x == 0) {
    sc.step += 1;
    sc.nh_step += 1;
    sc.ago += 1;
    sc.rebuild = sc.force_rebuild;
    sc.force_rebuild = 0;
    sc.check = (sc.ago >= S.neigh_delay) ? 1 : 0;
    if (S.npt) nhc_press_half(S, sc, sc.t_target_now);   // with the target of the previous step, as fix_nh orders it
    sc.t_target_now = S.t_start + (S.t_stop - S.t_start) * ((double)sc.nh_step / (double)S.nh_total);
    double vs = nhc_half(S, sc);
    sc.dil = 1.0;
    if (S.npt) {
      nh_omega_dot(S, sc);   // kinetic part after the thermostat half step, virial of the last force evaluation
      const double fp = exp(-0.25 * S.dt * (sc.omega_dot + sc.mtk_term2));
      vs *= fp;
      vs *= fp;
      // the two half-step remaps: box and atoms dilate about the box centre, tilt factors with their lengths
      const double E = exp(0.5 * S.dt * sc.omega_dot);
      sc.dil = E;
      for (int d = 0; d < 3; d++) {
        const double c = 0.5 * (sc.box[d] + sc.box[3 + d]);
        sc.cen[d] = c;
        double lo = sc.box[d], hi = sc.box[3 + d];
        lo = (lo - c) * E + c; hi = (hi - c) * E + c;
        lo = (lo - c) * E + c; hi = (hi - c) * E + c;
        sc.box[d] = lo; sc.box[3 + d] = hi;
      }
      for (int k = 6; k < 9; k++) { sc.box[k] *= E; sc.box[k] *= E; }
    }
    sc.vscale *= vs;
    double c[24];
    box_corners(sc.box, c);
    double d1 = 0.0, d2 = 0.0;
    for (int k = 0; k < 8; k++) {
      const double dx = c[3 * k] - sc.corners_hold[3 * k], dy = c[3 * k + 1] - sc.corners_hold[3 * k + 1], dz = c[3 * k + 2] - sc.corners_hold[3 * k + 2];
      const double d = sqrt(dx * dx + dy * dy + dz * dz);
      if (d > d1) { d2 = d1; d1 = d; }
      else if (d > d2) d2 = d;
    }
    const double delta = 0.5 * (S.skin - (d1 + d2));
    sc.deltasq = (delta > 0.0) ? delta * delta : -1.0;
    const double far = 0.5 * (S.far_band - (d1 + d2));
    sc.far_dsq = (far > 0.0) ? far * far * (1.0 - 1.0e-9) : -1.0;
    sc.need_far = 0;
    for (int k = 0; k < 6; k++) sc.ke[k] = 0.0;
    for (int k = 0; k < MD_NPART * 6; k++) sc.vir[k] = 0.0;
    for (int k = 0; k < MD_NPART; k++) sc.eng[k] = 0.0;
  }
  for (int k = threadIdx.x; k < 2 * S.nk; k += blockDim.x) S.sfac[k] = 0.0;
  for (int k = threadIdx.x; k < S.ncells; k += blockDim.x) S.cell_count[k] = 0;
}

// v = v*vscale + dt/2 f/m ; x: half-step remap, drift, half-step remap = c + E^2 (x - c) + E dt v
__global__ __launch_bounds__(TPB) void k_initial_integrate_nh(const SimDev *sims) {
  const SimDev &S = sims[blockIdx.y];
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= S.natoms) return;
  SimScalars &sc = *S.sc;
  const double vs = sc.vscale, E = sc.dil;
  const double dtfm = 0.5 * S.dt * MD_FTM2V / S.mass[i];
  double dsq = 0.0, vn[3], xn[3];
  bool sane = true;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    vn[k] = S.v[3 * i + k] * vs + dtfm * S.f[3 * i + k];
    const double x1 = S.npt ? sc.cen[k] + (S.x[3 * i + k] - sc.cen[k]) * E : S.x[3 * i + k];
    const double x2 = x1 + S.dt * vn[k];
    xn[k] = S.npt ? sc.cen[k] + (x2 - sc.cen[k]) * E : x2;
    sane = sane && fabs(xn[k]) < 1.0e8;
  }
  if (!sane) {
    atomicOr(&sc.overflow, 16);
#pragma unroll
    for (int k = 0; k < 3; k++) { vn[k] = 0.0; xn[k] = S.x[3 * i + k]; }
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    S.v[3 * i + k] = vn[k];
    S.x[3 * i + k] = xn[k];
    const double d = xn[k] - S.xhold[3 * i + k];
    dsq += d * d;
  }
  if (sc.check && dsq > sc.deltasq) sc.rebuild = 1;
  if (dsq >= sc.far_dsq) sc.need_far = 1;
}

// end of a step: barostat kick of the velocities, temperature, omega_dot, both chains, box-length averages, grid guard
__global__ void k_post_nh(const SimDev *sims) {
  const SimDev &S = sims[blockIdx.x];
  SimScalars &sc = *S.sc;
  if (threadIdx.x != 0) return;
  sc.nfar_steps += sc.need_far;
  double f2 = 1.0;
  if (S.npt) {
    const double fp = exp(-0.25 * S.dt * (sc.omega_dot + sc.mtk_term2));
    f2 *= fp;
    f2 *= fp;
    for (int k = 0; k < 6; k++) sc.ke[k] *= f2 * f2;
  }
  sc.t_current = (sc.ke[0] + sc.ke[1] + sc.ke[2]) / (S.tdof * MD_BOLTZ);
  if (S.npt) nh_omega_dot(S, sc);
  const double ft = nhc_half(S, sc);
  for (int k = 0; k < 6; k++) sc.ke[k] *= ft * ft;
  sc.vscale = f2 * ft;
  if (S.npt) nhc_press_half(S, sc, sc.t_target_now);
  if (S.lavg_nav > 0 && sc.nh_step <= 2 * S.lavg_nav) {
    for (int d = 0; d < 3; d++) sc.lsum[d] += sc.box[3 + d] - sc.box[d];
    if (sc.nh_step % S.lavg_nav == 0) {
      for (int d = 0; d < 3; d++) { sc.lrun[d] += sc.lsum[d] / S.lavg_nav; sc.lsum[d] = 0.0; }
      sc.nlwin += 1;
    }
  }
  // the cell grid and the k-space tables of this segment hold for box lengths within +-box_margin of those at its start
  if (S.npt)
    for (int d = 0; d < 3; d++) {
      const double r = (sc.box[3 + d] - sc.box[d]) / sc.len0[d];
      if (r > 1.0 + S.box_margin || r < 1.0 - S.box_margin) atomicOr(&sc.overflow, 64);
    }
}

// ------------------------------------------------------------------------------------------
// min_style sd
// ------------------------------------------------------------------------------------------
enum { MIN_INIT = 0, MIN_TRIAL = 1, MIN_QUAD = 2, MIN_RESET = 3, MIN_DONE = 4 };
#define MIN_ALPHA_MAX 1.0
#define MIN_ALPHA_REDUCE 0.5
#define MIN_BACKTRACK_SLOPE 0.4
#define MIN_QUADRATIC_TOL 0.1
#define MIN_EMACH 1.0e-8
#define MIN_EPS_QUAD 1.0e-28
#define MIN_EPS_ENERGY 1.0e-8

// before a force evaluation: zero the accumulators (k_pre without a thermostat)
__global__ void k_min_pre(const SimDev *sims) {
  const SimDev &S = sims[blockIdx.x];
  SimScalars &sc = *S.sc;
  if (threadIdx.x == 0) {
    sc.step += 1;
    sc.ago += 1;
    sc.rebuild = 0;
    sc.check = 1;
    const double delta = 0.5 * S.skin;
    sc.deltasq = delta * delta;
    const double far = 0.5 * S.far_band;
    sc.far_dsq = far * far * (1.0 - 1.0e-9);
    sc.need_far = 0;
    for (int k = 0; k < 6; k++) sc.ke[k] = 0.0;
    for (int k = 0; k < MD_NPART * 6; k++) sc.vir[k] = 0.0;
    for (int k = 0; k < MD_NPART; k++) sc.eng[k] = 0.0;
    for (int k = 0; k < 4; k++) sc.min_dots[k] = 0.0;
    // where the previous move left x on its search line (the move of this evaluation starts from there)
    sc.min_alpha_now = sc.min_alpha_next;
    sc.min_alpha_next = (sc.min_phase == MIN_RESET) ? 0.0 : sc.min_alpha;
  }
  for (int k = threadIdx.x; k < 2 * S.nk; k += blockDim.x) S.sfac[k] = 0.0;
  for (int k = threadIdx.x; k < S.ncells; k += blockDim.x) S.cell_count[k] = 0;
}
// trial point of the line search: (new direction: x0 = x, h = f) x = x0 + alpha h
__global__ __launch_bounds__(TPB) void k_min_move(const SimDev *sims, double *const *x0s, double *const *hs) {
  const SimDev &S = sims[blockIdx.y];
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= S.natoms) return;
  SimScalars &sc = *S.sc;
  if (sc.min_phase == MIN_DONE) return;
  double *x0 = x0s[blockIdx.y], *h = hs[blockIdx.y];
  const double alpha = (sc.min_phase == MIN_RESET) ? 0.0 : sc.min_alpha;
  const double anow = sc.min_newdir ? 0.0 : sc.min_alpha_now;   // read by every thread before k_min_decide moves it on
  double dsq = 0.0;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (sc.min_newdir) { x0[3 * i + k] = S.x[3 * i + k]; h[3 * i + k] = S.f[3 * i + k]; }
    const double xn = S.min_incremental ? S.x[3 * i + k] + (alpha - anow) * h[3 * i + k] : x0[3 * i + k] + alpha * h[3 * i + k];
    S.x[3 * i + k] = xn;
    const double d = xn - S.xhold[3 * i + k];
    dsq += d * d;
  }
  if (dsq > sc.deltasq) sc.rebuild = 1;
  if (dsq >= sc.far_dsq) sc.need_far = 1;
}
// after a force evaluation: f.h, f.f, max |f|
__global__ __launch_bounds__(TPB) void k_min_reduce(const SimDev *sims, double *const *hs) {
  const SimDev &S = sims[blockIdx.y];
  SimScalars &sc = *S.sc;
  __shared__ double s_red[3][TPB / 64];
  const int i = blockIdx.x * TPB + threadIdx.x;
  double fh = 0.0, ff = 0.0, fm = 0.0;
  if (i < S.natoms) {
    const double *h = hs[blockIdx.y];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const double f = S.f[3 * i + k];
      fh += f * h[3 * i + k];
      ff += f * f;
      fm = fmax(fm, fabs(f));
    }
  }
  fh = wave_sum(fh); ff = wave_sum(ff);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) fm = fmax(fm, __shfl_down(fm, o, 64));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { s_red[0][wave] = fh; s_red[1][wave] = ff; s_red[2][wave] = fm; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0, c = 0.0;
    for (int w = 0; w < TPB / 64; w++) { a += s_red[0][w]; b += s_red[1][w]; c = fmax(c, s_red[2][w]); }
    atomicAdd(&sc.min_dots[0], a);
    atomicAdd(&sc.min_dots[1], b);
    // max of non-negative doubles = max of their bit patterns
    atomicMax((unsigned long long *)&sc.min_dots[2], (unsigned long long)__double_as_longlong(c));
  }
}
// the line search and the iteration of min_sd.cpp as a state machine: one call per force evaluation and replica
__global__ void k_min_decide(const SimDev *sims) {
  const SimDev &S = sims[blockIdx.x];
  SimScalars &sc = *S.sc;
  if (threadIdx.x != 0 || sc.min_phase == MIN_DONE) return;
  double ecurrent = 0.0;
  for (int k = 0; k < MD_NPART; k++) ecurrent += sc.eng[k];
  const double fh = sc.min_dots[0], ff = sc.min_dots[1], fmaxv = sc.min_dots[2];
  sc.min_ecur = ecurrent;
  sc.min_newdir = 0;
  bool accept = false, backtrack = false;
  switch (sc.min_phase) {
    case MIN_INIT:
      sc.min_einit = ecurrent;
      sc.min_eprev = ecurrent;
      sc.min_iter = 0;
      sc.min_neval = 0;
      accept = true;   // starts the first line search along h = f
      break;
    case MIN_TRIAL: {
      sc.min_neval += 1;
      const double delfh = fh - sc.min_fhprev;
      if (fabs(fh) < MIN_EPS_QUAD || fabs(delfh) < MIN_EPS_QUAD) { sc.min_phase = MIN_RESET; sc.min_stop = 4; return; }
      const double relerr = fabs(1.0 - (0.5 * (sc.min_alpha - sc.min_alphaprev) * (fh + sc.min_fhprev) + ecurrent) / sc.min_engprev);
      const double alpha0 = sc.min_alpha - (sc.min_alpha - sc.min_alphaprev) * fh / delfh;
      sc.min_fh_trial = fh;
      if (relerr <= MIN_QUADRATIC_TOL && alpha0 > 0.0 && alpha0 < sc.min_alphamax) {
        // evaluate at the projected minimum; the backtracking state (alpha) is kept for the case it is not good enough
        sc.min_phase = MIN_QUAD;
        sc.min_alphaprev = sc.min_alpha;   // parked: restored below
        sc.min_alpha = alpha0;
        return;
      }
      backtrack = true;
      break;
    }
    case MIN_QUAD: {
      sc.min_neval += 1;
      const double alpha_trial = sc.min_alphaprev;   // the alpha of the trial that led here
      sc.min_alpha = alpha_trial;
      if (ecurrent - sc.min_eorig < MIN_EMACH) accept = true;
      else backtrack = true;
      break;
    }
    case MIN_RESET:
      sc.min_phase = MIN_DONE;   // energy and forces at the starting point of the failed line search are in place
      return;
    default: return;
  }
  if (backtrack) {
    const double de_ideal = -MIN_BACKTRACK_SLOPE * sc.min_alpha * sc.min_fdothall, de = ecurrent - sc.min_eorig;
    if (de <= de_ideal) accept = true;
    else {
      sc.min_fhprev = sc.min_fh_trial;
      sc.min_engprev = ecurrent;
      sc.min_alphaprev = sc.min_alpha;
      sc.min_alpha *= MIN_ALPHA_REDUCE;
      if (sc.min_alpha <= 0.0 || de_ideal >= -MIN_EMACH) { sc.min_phase = MIN_RESET; sc.min_stop = 4; return; }
      sc.min_phase = MIN_TRIAL;
      return;
    }
  }
  if (accept) {
    if (sc.min_phase != MIN_INIT) {
      // end of an iteration of min_sd.cpp
      if (sc.min_neval >= S.min_maxeval) { sc.min_phase = MIN_DONE; sc.min_stop = 3; return; }
      if (fabs(ecurrent - sc.min_eprev) < S.min_etol * 0.5 * (fabs(ecurrent) + fabs(sc.min_eprev) + MIN_EPS_ENERGY)) { sc.min_phase = MIN_DONE; sc.min_stop = 0; return; }
      if (ff < S.min_ftol * S.min_ftol) { sc.min_phase = MIN_DONE; sc.min_stop = 1; return; }
      if (sc.min_iter >= S.min_maxiter) { sc.min_phase = MIN_DONE; sc.min_stop = 2; return; }
    } else if (S.min_maxiter <= 0) { sc.min_phase = MIN_DONE; sc.min_stop = 2; return; }
    // next line search along h = f from here
    sc.min_iter += 1;
    sc.min_eprev = ecurrent;
    sc.min_eorig = ecurrent;
    sc.min_fdothall = ff;
    if (ff <= 0.0 || fmaxv == 0.0) { sc.min_phase = MIN_DONE; sc.min_stop = 4; return; }
    sc.min_alphamax = fmin(MIN_ALPHA_MAX, S.min_dmax / fmaxv);
    sc.min_alpha = sc.min_alphamax;
    sc.min_fhprev = ff;
    sc.min_engprev = ecurrent;
    sc.min_alphaprev = 0.0;
    sc.min_newdir = 1;
    sc.min_phase = MIN_TRIAL;
  }
}

// change_box all x final 0 lx y final 0 ly z final 0 lz remap: one replica, new lengths from the host
__global__ __launch_bounds__(TPB) void k_change_box(double *x, int natoms, const double *box_old, const double *box_new) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= natoms) return;
  BoxD bo, bn;
  box_derive(box_old, bo);
  box_derive(box_new, bn);
  const double d0 = x[3 * i] - bo.lo[0], d1 = x[3 * i + 1] - bo.lo[1], d2 = x[3 * i + 2] - bo.lo[2];
  const double l0 = bo.hinv[0] * d0 + bo.hinv[5] * d1 + bo.hinv[4] * d2, l1 = bo.hinv[1] * d1 + bo.hinv[3] * d2, l2 = bo.hinv[2] * d2;
  x[3 * i] = bn.h[0] * l0 + bn.h[5] * l1 + bn.h[4] * l2 + bn.lo[0];
  x[3 * i + 1] = bn.h[1] * l1 + bn.h[3] * l2 + bn.lo[1];
  x[3 * i + 2] = bn.h[2] * l2 + bn.lo[2];
}

static inline dim3 grid2(int nx, int ns) { return dim3((unsigned)nx, (unsigned)ns, 1); }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

void mdk_setup_post_nh(hipStream_t st, const SimDev *d, int ns) { hipLaunchKernelGGL(k_setup_post_nh, dim3(ns), dim3(64), 0, st, d); }
void mdk_pre_nh(hipStream_t st, const SimDev *d, int ns) { hipLaunchKernelGGL(k_pre_nh, dim3(ns), dim3(64), 0, st, d); }
void mdk_initial_integrate_nh(hipStream_t st, const SimDev *d, int ns, int maxatoms) {
  hipLaunchKernelGGL(k_initial_integrate_nh, grid2(cdiv(maxatoms, TPB), ns), dim3(TPB), 0, st, d);
}
void mdk_post_nh(hipStream_t st, const SimDev *d, int ns) { hipLaunchKernelGGL(k_post_nh, dim3(ns), dim3(64), 0, st, d); }
void mdk_min_pre(hipStream_t st, const SimDev *d, int ns) { hipLaunchKernelGGL(k_min_pre, dim3(ns), dim3(64), 0, st, d); }
void mdk_min_move(hipStream_t st, const SimDev *d, int ns, int maxatoms, double *const *x0s, double *const *hs) {
  hipLaunchKernelGGL(k_min_move, grid2(cdiv(maxatoms, TPB), ns), dim3(TPB), 0, st, d, x0s, hs);
}
void mdk_min_reduce(hipStream_t st, const SimDev *d, int ns, int maxatoms, double *const *hs) {
  hipLaunchKernelGGL(k_min_reduce, grid2(cdiv(maxatoms, TPB), ns), dim3(TPB), 0, st, d, hs);
}
void mdk_min_decide(hipStream_t st, const SimDev *d, int ns) { hipLaunchKernelGGL(k_min_decide, dim3(ns), dim3(64), 0, st, d); }
void mdk_change_box(hipStream_t st, double *x, int natoms, const double *box_old, const double *box_new) {
  hipLaunchKernelGGL(k_change_box, dim3(cdiv(natoms, TPB)), dim3(TPB), 0, st, x, natoms, box_old, box_new);
}
