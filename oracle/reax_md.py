"""Whole strained evaluations with the ReaxFF oracle: what STMDProblem::lammps_straining asks LAMMPS for when
md_force_field = "reax" (reference headers/stmd_problem.h:84-383 with lammps_scripts/lammps_scripts_reax/in.set.lammps,
in.strain.lammps and ELASTIC/in.homogenization.lammps), restated on the CPU.

TEST INFRASTRUCTURE ONLY, PARITY UNPINNED (no LAMMPS, no USER-REAXC here).  Nothing new is invented for the dynamics: the
integrator, the Nose-Hoover chain of `fix nvt`, `fix deform ... remap x` with its box flips, the `fix ave/time` pressure
average and the host arithmetic (length scaling, nts rule, "%.6e" rates, stress = -<P> * 101325) are the ones of
oracle/md_oracle.c (omd_run / omd_eval), which the OPLS path is checked against; this module only plugs another force field
into them (omd_set_external_force):

  forces, virial   oracle/reax_torch.py: reverse-mode derivative of the energy expression of oracle/reax_oracle.c at fixed
                   charges (as pair reax/c does: no dq/dr terms)
  charges          `fix qeq/reax 1 0.0 10.0 1e-6 reax/c` (in.strain.lammps:12): every step, two Jacobi-preconditioned CG solves
                   from the extrapolated previous solutions (init_matvec: cubic for s, quadratic for t), histories empty at the
                   start of each run as a newly created fix has them
  no SHAKE, no k-space (the reax scripts have neither); tdof = 3N - 3.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import pyoracle as po
from . import reax_torch as rt

FORCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                       C.POINTER(C.c_double), C.POINTER(C.c_double))


class ReaxMD:
    """One replica: atoms of LAMMPS types 1..ntypes mapped to the elements `elements` (pair_coeff * * ffield H C N O)."""

    def __init__(self, ffield_path: str, elements, lammps_type, mass, box, x, v, qeq_tol: float = 1e-6, qeq_maxiter: int = 200):
        self.R = rt.ReaxEnergy(ffield_path)
        lt = np.asarray(lammps_type, dtype=np.int64)
        self.rtype = self.R.ff.types([elements[t] for t in lt])
        n = len(lt)
        self.n = n
        ntypes = int(lt.max()) + 1
        z_i = np.zeros(0, np.int32)
        z2 = np.zeros((0, 2))
        sysd = dict(natoms=n, ntypes=ntypes, type=lt.astype(np.int32), charge=np.zeros(n), mass=np.asarray(mass, float)[:ntypes],
                    eps=np.zeros((ntypes, ntypes)), sigma=np.ones((ntypes, ntypes)),
                    bonds=np.zeros((0, 2), np.int32), bond_type=z_i, bond_coeff=z2,
                    angles=np.zeros((0, 3), np.int32), angle_type=z_i, angle_coeff=z2,
                    dihedrals=np.zeros((0, 4), np.int32), dihedral_type=z_i, dihedral_coeff=np.zeros((0, 4)),
                    impropers=np.zeros((0, 4), np.int32), improper_type=z_i, improper_coeff=z2,
                    special_lj=np.zeros(3), special_coul=np.zeros(3), box=np.asarray(box, float), x=np.asarray(x, float), v=np.asarray(v, float))
        self.o = po.Oracle(sysd, po.default_params(shake_mass=0.0))
        self.tol, self.imax = qeq_tol, qeq_maxiter
        self.q = np.zeros(n)
        self.qeq_iters = 0
        self.qeq_solves = 0
        self.last = {}
        self._cb = FORCE_FN(self._force)
        L = po.lib()
        L.omd_set_external_force.argtypes = [C.c_void_p, FORCE_FN, C.c_void_p]
        L.omd_set_external_force.restype = None
        L.omd_set_external_force(self.o.h, self._cb, None)

    # ---- fix qeq/reax: pre_force of every step ----
    def _charges(self, call, x, box, pairs):
        n = self.n
        if call == 0:   # a new run creates the fix anew: empty histories
            self.s_hist = np.zeros((5, n))
            self.t_hist = np.zeros((5, n))
        H, dia = self.R.h_matrix(self.rtype, x, box, pairs)
        sh, th = self.s_hist, self.t_hist
        t0 = th[2] + 3.0 * (th[0] - th[1])
        s0 = 4.0 * (sh[0] + sh[2]) - (6.0 * sh[1] + sh[3])
        chi = self.R.p.sbp["chi"][self.rtype]
        s, it1 = self.R.cg(H, dia, -chi, s0, self.tol, self.imax)
        t, it2 = self.R.cg(H, dia, -np.ones(n), t0, self.tol, self.imax)
        if it1 >= self.imax or it2 >= self.imax:
            raise RuntimeError("charge equilibration did not converge")
        self.qeq_iters += it1 + it2
        self.qeq_solves += 1
        u = s.sum() / t.sum()
        self.q = s - u * t
        self.s_hist = np.vstack([s[None], sh[:4]])
        self.t_hist = np.vstack([t[None], th[:4]])

    def _force(self, ctx, call, n, box_p, x_p, f_p, vir_p, e_p):
        box = np.ctypeslib.as_array(box_p, shape=(9,)).copy()
        x = np.ctypeslib.as_array(x_p, shape=(n, 3)).copy()
        pairs = rt.pair_list(x, box, self.R.p.swb)
        self._charges(call, x, box, pairs)
        f, w, e, parts = self.R.forces(self.rtype, x, box, self.q, virial=True, pairs=pairs)
        np.ctypeslib.as_array(f_p, shape=(n, 3))[:] = f
        np.ctypeslib.as_array(vir_p, shape=(6,))[:] = w
        e_p[0] = e
        self.last = dict(energy=e, parts=parts, virial=w)

    # ---- what the tests drive ----
    def get_state(self):
        return self.o.get_state()

    def set_state(self, box, x, v):
        self.o.set_state(box, x, v)

    def compute(self):
        """static evaluation of the current state: forces, energy, virial, charges"""
        self.o.setup(False)
        f, e, w = self.o.compute()
        return f, self.last["energy"], self.last["virial"].copy(), self.q.copy()

    def run(self, nsteps, dt, temperature, nvt=True, rates=None, sample=False, trace=False):
        return self.o.run(nsteps, dt, temperature, nvt=nvt, use_shake=False, rates=rates, sample=sample, trace=trace)

    def eval(self, strain_len, dt, temperature, strain_rate, nss):
        """one STMDProblem::strain: stress [Pa] in the raw order xx, yy, zz, xy, xz, yz, and the straining steps"""
        return self.o.eval(strain_len, dt, temperature, strain_rate, nss)
