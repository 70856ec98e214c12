"""init_material for ReaxFF replicas (md_force_field "reax": the reax copies of in.init.lammps and ELASTIC/*.lammps differ from the
opls ones only in the force-field lines, lammps_scripts_reax/in.init.lammps:2-4): the minimiser, the Nose-Hoover NPT runs and
the schedule of md_equil.hip driven by the ReaxFF force stage (md_reax.hip).  The reax oracle has no dynamics, so what is
shown is (i) the minimiser against a restatement of min_sd.cpp that takes its energies and forces from the oracle, on a molecule,
(ii) what the integrator guarantees, (iii) the chain of calls end to end.  PARITY UNPINNED at the oracle (tests/test_gpu_reax.py)."""
import os

import numpy as np
import pytest

from oracle import pyreax as pr
from scema_amd import capi
from test_oracle_reax import FFIELD
from test_reax_host import _pe_cell

pytestmark = pytest.mark.gpu

MVV2E = 48.88821291 ** 2
BOLTZ = 0.0019872067


@pytest.fixture(scope="module")
def ff():
    f = pr.ForceField(FFIELD)
    yield f
    f.close()


def _sd_reference(fn, x, etol, ftol, maxiter, maxeval=50000):
    """min_style sd with the quadratic line search (min_linesearch.cpp as restated in oracle/md_oracle.c::omd_minimize),
    energies and forces from `fn`"""
    e, f = fn(x)
    h = f.copy()
    einit, neval, it = e, 0, 0
    while it < maxiter:
        it += 1
        eprev = eorig = e
        fdoth, hmax = float((f * h).sum()), float(np.abs(h).max())
        if fdoth <= 0.0 or hmax == 0.0:
            return x, e, it, neval, 4, einit
        amax = min(1.0, 0.1 / hmax)
        x0, alpha, fhprev, engprev, aprev, fail = x.copy(), amax, fdoth, eorig, 0.0, False
        while True:
            x = x0 + alpha * h
            e, f = fn(x); neval += 1
            fh = float((f * h).sum())
            delfh = fh - fhprev
            if abs(fh) < 1e-28 or abs(delfh) < 1e-28:
                x = x0; e, f = fn(x); fail = True; break
            relerr = abs(1.0 - (0.5 * (alpha - aprev) * (fh + fhprev) + e) / engprev)
            a0 = alpha - (alpha - aprev) * fh / delfh
            if relerr <= 0.1 and 0.0 < a0 < amax:
                x = x0 + a0 * h
                e, f = fn(x); neval += 1
                if e - eorig < 1e-8:
                    break
            de_ideal, de = -0.4 * alpha * fdoth, e - eorig
            if de <= de_ideal:
                break
            fhprev, engprev, aprev = fh, e, alpha
            alpha *= 0.5
            if alpha <= 0.0 or de_ideal >= -1e-8:
                x = x0; e, f = fn(x); fail = True; break
        if fail:
            return x, e, it, neval, 4, einit
        if neval >= maxeval:
            return x, e, it, neval, 3, einit
        if abs(e - eprev) < etol * 0.5 * (abs(e) + abs(eprev) + 1e-8):
            return x, e, it, neval, 0, einit
        if float((f * f).sum()) < ftol * ftol:
            return x, e, it, neval, 1, einit
        h = f.copy()
    return x, e, it, neval, 2, einit


def test_minimiser_on_a_molecule_follows_the_oracle_driven_reference(ff):
    from test_oracle_reax import _glycine_like
    from test_reax_host import BIGBOX, _sym
    t, x = _glycine_like(ff)
    sym = _sym(ff, t)
    e = capi.Engine()
    e.reax_configure(FFIELD, qeq_tol=1e-10)
    e.register_replica("m", 1, capi.reax_system(sym, x, BIGBOX))

    def fn(xx):                                   # energy and forces of the oracle (charges re-equilibrated at every point)
        q, _ = ff.qeq(t, xx, box=BIGBOX, tol=1e-11, maxiter=500)
        en, _ = ff.energy(t, xx, box=BIGBOX, q=q)
        return en, ff.forces(t, xx, box=BIGBOX, q=q, h=1e-5)

    for maxiter in (1, 3):
        e.set_state(0, "m", 1, BIGBOX, x, np.zeros_like(x))
        r = e.minimize("m", 1, 0, etol=0.0, ftol=0.0, maxiter=maxiter)
        xr, er, itr, nev, stop, einit = _sd_reference(fn, x.copy(), 0.0, 0.0, maxiter)
        assert (r["iterations"], r["evaluations"], r["stop"]) == (itr, nev, stop)
        assert abs(r["e_initial"] - einit) < 1e-7 * abs(einit) and abs(r["e_final"] - er) < 1e-6 * abs(er)
        assert np.abs(e.get_state(0, "m", 1)[1] - xr).max() < 1e-5      # central-difference forces drive the reference
    e.close()


def test_minimiser_on_a_condensed_cell_goes_downhill_and_is_consistent(ff):
    sym, x, box = _pe_cell(ff, amp=0.12)
    e = capi.Engine()
    e.reax_configure(FFIELD, qeq_tol=1e-8)
    e.register_replica("m", 1, capi.reax_system(sym, x, box))
    e0 = sum(e.reax_compute("m", 1)["e"].values())
    e.set_state(0, "m", 1, box, x, np.zeros_like(x))
    es = []
    for k in (2, 6, 20):
        e.set_state(0, "m", 1, box, x, np.zeros_like(x))
        r = e.minimize("m", 1, 0, etol=0.0, ftol=0.0, maxiter=k)
        assert r["iterations"] == k and r["stop"] == 2 and abs(r["e_initial"] - e0) < 1e-8 * abs(e0)
        es.append(r["e_final"])
    assert es[0] < e0 and es[1] < es[0] and es[2] < es[1]
    # the energy the minimiser reports is the energy of the state it leaves
    b, xm, _ = e.get_state(0, "m", 1)
    e.register_replica("c", 1, capi.reax_system(sym, xm, b))
    assert abs(sum(e.reax_compute("c", 1)["e"].values()) - es[2]) < 1e-7 * abs(es[2])
    e.close()


def test_npt_run_and_the_schedule_with_the_reax_force_stage(ff, tmp_path):
    sym, x, box = _pe_cell(ff, amp=0.05)
    m = np.array([12.011 if s == "C" else 1.008 for s in sym])
    rng = np.random.default_rng(2)
    v = rng.standard_normal(x.shape) * np.sqrt(BOLTZ * 200.0 / (m[:, None] * MVV2E))
    v -= (m[:, None] * v).sum(0) / m.sum()
    e = capi.Engine()
    e.reax_configure(FFIELD, qeq_tol=1e-6)
    e.register_replica("m", 1, capi.reax_system(sym, x, box, v=v))
    # a barostat that cannot move (period 1e7 fs) is the thermostat alone
    e.set_state(0, "m", 1, box, x, v)
    e.run_nh("m", 1, 0, 20, 0.25, 200.0, 200.0, npt=True, p_target=1.0, p_period=1e7)
    b1, x1, _ = e.get_state(0, "m", 1)
    e.set_state(1, "m", 1, box, x, v)
    e.run_nh("m", 1, 1, 20, 0.25, 200.0, 200.0, npt=False)
    b2, x2, _ = e.get_state(1, "m", 1)
    assert np.abs(b1 - box).max() < 1e-6 and np.abs(x1 - x2).max() < 1e-6
    # a live barostat moves the box the way the pressure asks: this jittered crystal is under compression (P >> 1 atm)
    e.set_state(2, "m", 1, box, x, v)
    lav = e.run_nh("m", 1, 2, 60, 0.25, 200.0, 300.0, npt=True, p_target=1.0, p_period=100.0, average_lengths=True)
    b3, x3, v3 = e.get_state(2, "m", 1)
    l0, l3 = box[3:6] - box[:3], b3[3:6] - b3[:3]
    assert np.all(l3 > l0 * 1.0005) and np.allclose(l3 / l0, (l3 / l0)[0], rtol=1e-12)     # isotropic expansion
    assert np.all(lav > l0) and np.all(lav < l3 * 1.0000001)
    assert np.isfinite(x3).all() and np.isfinite(v3).all()
    # EQMDProblem::equil with mdff "reax": scrloc holds ffield.reax.2; schedule, state file, the three tensors
    from scema_amd import stmd
    folder = str(tmp_path / "nano_in"); os.makedirs(folder)
    base = stmd.eqmd_equil_full(e, "m", str(tmp_path), folder, 1, mdts=0.25, mdtem=250.0, mdnss=10, mdnse=2, mdss=2e-3, mdsa=0.005, mdff="reax",
                                scrloc=os.path.dirname(FFIELD))
    length = np.loadtxt(base + ".length"); stress = np.loadtxt(base + ".stress"); stiff = np.loadtxt(base + ".stiff")
    assert os.path.exists(base + ".bin") and np.all(length > 15.0) and np.isfinite(stress).all() and np.isfinite(stiff).all()
    assert np.abs(stiff).max() > 1e8                                   # a solid: GPa-scale stiffness entries
    bq = e.get_state(capi.QP_NONE, "m", 1)[0]
    assert np.allclose(bq[3:6] - bq[:3], length, rtol=1e-12)
    e.close()
