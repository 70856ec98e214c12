"""Pins for the oracle's restatement of init_material's equilibration schedule (lammps_scripts_opls/in.init.lammps:44-215;
SURVEY.md 8(f) row f-2): min_style sd with the quadratic line search, fix npt ... iso (Nose-Hoover chains on particles and
box, MTK terms), temperature ramps, box-length averaging, change_box remap, velocity create.  PARITY UNPINNED (LAMMPS is not
available): what is checked is what the algorithms guarantee."""
from copy import deepcopy

import numpy as np
import pytest

from oracle import pyoracle as po

BOLTZ = 0.0019872067
KW = dict(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=1e-5, shake_mass=0.0)


def _no_lj(d):
    d = deepcopy(d)
    d["eps"] = d["eps"] * 0.0     # the 5 A test cutoff truncates an unshifted LJ: energy jumps unrelated to the integrator
    return d


def test_velocity_create_temperature_and_momenta(small_pe):
    o = po.Oracle(small_pe, po.default_params(**KW))
    o.velocity_create(200.0, seed=1234)
    t, _ = o.temperature()
    box, x, v = o.get_state()
    m = small_pe["mass"][small_pe["type"]]
    assert abs(t - 200.0) < 1e-9
    assert np.abs((m[:, None] * v).sum(0)).max() < 1e-10
    cm = (m[:, None] * x).sum(0) / m.sum()
    L = (m[:, None] * np.cross(x - cm, v)).sum(0)
    assert np.abs(L).max() < 1e-8
    o.velocity_create(200.0, seed=1234)
    assert np.array_equal(o.get_state()[2], v)                      # same seed, same velocities
    o.velocity_create(200.0, seed=99)
    assert np.abs(o.get_state()[2] - v).max() > 1e-4


def test_steepest_descent_goes_downhill_and_stops_on_a_criterion(small_pe):
    from scema_amd.systems import build_pe
    d = build_pe(2, 3, 5, jitter=0.08, seed=3)
    o = po.Oracle(d, po.default_params(**KW))
    f0, e0, _ = (o.setup(False), o.compute())[1]
    r = o.minimize(etol=1e-7, ftol=1e-11, maxiter=400)
    assert r["e_initial"] == pytest.approx(e0.sum(), rel=1e-12)
    assert r["e_final"] < r["e_initial"] - 100.0
    assert r["stop"] in (0, 1, 4) and r["evaluations"] >= r["iterations"]
    f1, e1, _ = o.compute()
    assert e1.sum() == pytest.approx(r["e_final"], rel=1e-10)
    assert np.abs(f1).max() < 0.2 * np.abs(f0).max()
    # every accepted step decreases the energy: a budget of k iterations never ends higher than one of k-1
    es = []
    for k in (1, 2, 5, 10):
        o2 = po.Oracle(d, po.default_params(**KW))
        es.append(o2.minimize(etol=0.0, ftol=0.0, maxiter=k)["e_final"])
    assert all(b < a for a, b in zip(es, es[1:])) and es[0] < r["e_initial"]
    # iteration and evaluation budgets are honoured
    o3 = po.Oracle(d, po.default_params(**KW))
    r3 = o3.minimize(etol=0.0, ftol=0.0, maxiter=3)
    assert r3["stop"] == 2 and r3["iterations"] == 3
    o4 = po.Oracle(d, po.default_params(**KW))
    r4 = o4.minimize(etol=0.0, ftol=0.0, maxiter=1000, maxeval=7)
    assert r4["stop"] == 3 and 7 <= r4["evaluations"] <= 9


def _calm():
    from scema_amd.systems import build_pe
    return build_pe(2, 3, 5, jitter=0.02, seed=3)     # near its minimum: the runs below stay in a gentle regime


def test_npt_conserved_quantity_and_second_order():
    o = po.Oracle(_no_lj(_calm()), po.default_params(**KW))
    o.velocity_create(200.0)
    box, x, v = o.get_state()
    res = []
    for dt in (0.25, 0.125):
        o.set_state(box, x, v)
        _, tr = o.run_nh(int(50 / dt), dt, 200.0, 200.0, npt=True, p_target=1.0, p_period=50.0, trace=True)
        h = tr[:, 1] + tr[:, 2] + tr[:, 3]
        res.append((np.abs(h - h[0]).max(), tr))
    (f1, tr1), (f2, _) = res
    assert f1 < 2e-3 * tr1[:, 2].mean()                 # thermostat + barostat energy included: conserved
    assert f2 < 0.6 * f1                                # and better with a smaller step
    assert tr1[:, 3].max() - tr1[:, 3].min() > 20 * f1  # while the reservoirs really exchange energy
    assert abs(tr1[-1, 4] / tr1[0, 4] - 1.0) > 5e-3     # and the box moves


def test_barostat_drives_the_pressure_towards_its_target():
    o = po.Oracle(_no_lj(_calm()), po.default_params(**KW))
    o.velocity_create(200.0)
    _, tr = o.run_nh(1500, 0.5, 200.0, 200.0, npt=True, p_target=1.0, p_period=100.0, trace=True)
    p_early, p_late = np.abs(tr[:50, 5]).mean(), np.abs(tr[-500:, 5].mean())
    assert p_early > 2000.0 and p_late < 0.25 * p_early


def test_temperature_ramp_and_nvt_is_the_same_fix_without_the_barostat(small_pe):
    o = po.Oracle(_calm(), po.default_params(**KW))
    o.minimize(maxiter=50)
    o.velocity_create(100.0)
    box, x, v = o.get_state()
    _, tr = o.run_nh(4000, 0.5, 50.0, 300.0, npt=False, trace=True)     # 2 ps: the chain (period 100 fs) follows the target
    assert abs(tr[-400:, 0].mean() - 287.5) < 20.0 and tr[:400, 0].mean() < 80.0
    # constant target: omd_run_nh(npt = 0) is omd_run with nvt, no SHAKE
    o.set_state(box, x, v)
    o.run_nh(30, 0.5, 200.0, 200.0, npt=False)
    xa = o.get_state()[1]
    o.set_state(box, x, v)
    o.run(30, 0.5, 200.0, nvt=True, use_shake=False)
    assert np.abs(o.get_state()[1] - xa).max() < 1e-12


def test_box_length_average_and_change_box(small_pe):
    o = po.Oracle(small_pe, po.default_params(**KW))
    o.velocity_create(200.0)
    n = 40
    lav, tr = o.run_nh(n, 0.5, 200.0, 200.0, npt=True, p_target=1.0, p_period=100.0, average_lengths=True, trace=True)
    box = o.get_state()[0]
    vol_end = np.prod(box[3:6] - box[:3])
    assert vol_end == pytest.approx(tr[-1, 4], rel=1e-12)
    # isotropic: all three lengths scale together, so the averaged lengths keep the initial aspect ratios
    l0 = small_pe["box"][3:6] - small_pe["box"][:3]
    assert np.allclose(lav / l0, (lav / l0)[0], rtol=1e-12)
    # two windows of n/2 steps each: the mean of cbrt(volume) over all steps, scaled
    s = np.cbrt(tr[:, 4] / np.prod(l0))
    assert (lav / l0)[0] == pytest.approx(s.mean(), rel=1e-10)
    _, x, _ = o.get_state()
    def lamda(b, xx):
        h = np.array([[b[3] - b[0], b[6], b[7]], [0.0, b[4] - b[1], b[8]], [0.0, 0.0, b[5] - b[2]]])
        return np.linalg.solve(h, (xx - b[:3]).T).T
    frac = lamda(box, x)
    o.change_box(lav)
    b2, x2, _ = o.get_state()
    assert np.allclose(b2[:3], 0.0) and np.allclose(b2[3:6], lav) and np.allclose(b2[6:], box[6:])   # tilts kept
    assert np.abs(lamda(b2, x2) - frac).max() < 1e-12
