"""Pins for the oracle's integrator pieces (SURVEY.md §8(c) pins 5-6): NVE energy drift,
Nose-Hoover conserved quantity, SHAKE residual, fix-deform box trajectory, kinetic tensor,
running mean == direct mean."""
import numpy as np

from oracle import pyoracle as po

BOLTZ = 0.0019872067


def params(**kw):
    base = dict(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=1e-5)
    base.update(kw)
    return po.default_params(**base)


def relaxed(small_pe):
    from copy import deepcopy
    return deepcopy(small_pe)


def no_lj(d):
    """The 5 A test cutoff truncates an unshifted LJ (as lj/cut does), which adds energy jumps that
    have nothing to do with the integrator; the strict conservation checks switch LJ off."""
    d = relaxed(d)
    d["eps"] = d["eps"] * 0.0
    return d


def test_nve_energy_conservation(small_pe):
    o = po.Oracle(no_lj(small_pe), params(shake_mass=0.0))
    _, tr = o.run(200, 0.25, 300.0, nvt=False, use_shake=False, trace=True)
    etot = tr[:, 1] + tr[:, 2]
    # velocity Verlet: bounded O((w dt)^2) fluctuation of the stiff C-H modes, no drift
    assert np.abs(etot - etot[0]).max() < 5e-3 * tr[:, 2].mean()
    # second order: halving dt cuts the fluctuation ~4x
    o2 = po.Oracle(no_lj(small_pe), params(shake_mass=0.0))
    _, tr2 = o2.run(400, 0.125, 300.0, nvt=False, use_shake=False, trace=True)
    e2 = tr2[:, 1] + tr2[:, 2]
    assert np.abs(e2 - e2[0]).max() < 0.35 * np.abs(etot - etot[0]).max()
    # with the truncated LJ the jumps stay small compared with the kinetic energy
    o3 = po.Oracle(relaxed(small_pe), params(shake_mass=0.0))
    _, tr3 = o3.run(100, 0.25, 300.0, nvt=False, use_shake=False, trace=True)
    e3 = tr3[:, 1] + tr3[:, 2]
    assert np.abs(e3 - e3[0]).max() < 1e-2 * tr3[:, 2].mean()


def test_nose_hoover_conserved_quantity(small_pe):
    o = po.Oracle(no_lj(small_pe), params(shake_mass=0.0))
    _, tr = o.run(200, 0.25, 300.0, nvt=True, use_shake=False, trace=True)
    h = tr[:, 1] + tr[:, 2] + tr[:, 3]
    assert np.abs(h - h[0]).max() < 5e-3 * tr[:, 2].mean()
    o2 = po.Oracle(no_lj(small_pe), params(shake_mass=0.0))
    _, tr2 = o2.run(400, 0.125, 300.0, nvt=True, use_shake=False, trace=True)
    h2 = tr2[:, 1] + tr2[:, 2] + tr2[:, 3]
    assert np.abs(h2 - h2[0]).max() < 0.35 * np.abs(h - h[0]).max()
    # thermostat really acts: NH energy moves
    assert np.abs(tr[:, 3]).max() > 1e-3


def test_shake_residual_and_dof(small_pe):
    d = relaxed(small_pe)
    o = po.Oracle(d, params())
    assert o.nclusters == 120 and o.nconstraints == 240
    o.setup(True)
    assert o.tdof == 3 * 360 - 3 - 240
    o.run(30, 1.0, 300.0, nvt=True, use_shake=True)
    box, x, v = o.get_state()
    ch = [(b[0], b[1]) for b, t in zip(d["bonds"], d["bond_type"]) if t == 1]
    L = np.array([box[3] - box[0], box[4] - box[1], box[5] - box[2]])
    res = []
    for a, b in ch:
        dx = x[a] - x[b]
        # small tilt: fractional minimum image
        H = np.array([[L[0], box[6], box[7]], [0, L[1], box[8]], [0, 0, L[2]]])
        s = np.linalg.solve(H, dx); s -= np.rint(s); dx = H @ s
        res.append(np.linalg.norm(dx) / 1.09 - 1.0)
    # "fix shake 0.001 20": constraints hold to about the tolerance
    assert np.abs(res).max() < 2e-3


def test_deform_box_trajectory_and_remap(small_pe):
    d = relaxed(small_pe)
    o = po.Oracle(d, params())
    box0, x0, v0 = o.get_state()
    rates = np.array([1e-5, -2e-5, 3e-5, 1.5e-5, -0.5e-5, 2.5e-5])
    n, dt = 20, 2.0
    o.run(n, dt, 300.0, nvt=True, use_shake=True, rates=rates)
    box, x, v = o.get_state()
    t = n * dt
    L0 = box0[3:6] - box0[:3]
    for k in range(3):
        assert abs(box[k] - (box0[k] - 0.5 * L0[k] * rates[k] * t)) < 1e-12
        assert abs(box[3 + k] - (box0[3 + k] + 0.5 * L0[k] * rates[k] * t)) < 1e-12
    assert abs(box[6] - (box0[6] + rates[3] * L0[1] * t)) < 1e-12   # xy uses Ly0
    assert abs(box[7] - (box0[7] + rates[4] * L0[2] * t)) < 1e-12   # xz uses Lz0
    assert abs(box[8] - (box0[8] + rates[5] * L0[2] * t)) < 1e-12   # yz uses Lz0


def test_kinetic_tensor_and_temperature(small_pe):
    d = relaxed(small_pe)
    o = po.Oracle(d, params(shake_mass=0.0))
    o.setup(False)
    T, ke = o.temperature()
    m = d["mass"][d["type"]]
    v = d["v"]
    mvv2e = 48.88821291 ** 2
    assert abs(ke[0] - (m * v[:, 0] ** 2).sum() * mvv2e) < 1e-9
    assert abs(ke[3] - (m * v[:, 0] * v[:, 1]).sum() * mvv2e) < 1e-9
    assert abs(T - 300.0) < 1e-9   # generator scales to exactly T with dof 3N-3


def test_running_average_is_direct_mean(small_pe):
    o1 = po.Oracle(relaxed(small_pe), params())
    pavg, tr = o1.run(20, 1.0, 300.0, nvt=True, use_shake=True, sample=True, trace=True)
    # fix ave/time 1 2 2 ... ave running over 20 steps == mean over steps 1..20
    assert np.allclose(pavg[:3], tr[:, 5:8].mean(0), rtol=1e-12)
    # nss = 25 -> nav = 2, 12 windows -> steps 1..24 only
    o2 = po.Oracle(relaxed(small_pe), params())
    pavg2, tr2 = o2.run(25, 1.0, 300.0, nvt=True, use_shake=True, sample=True, trace=True)
    assert np.allclose(pavg2[:3], tr2[:24, 5:8].mean(0), rtol=1e-12)


def test_eval_is_deterministic_and_reports_nts(small_pe):
    lens = small_pe["box"][3:6] - small_pe["box"][:3]
    strain = np.array([-0.3 * 1.2e-3 * lens[0], -0.3 * 1.2e-3 * lens[1], 1.2e-3 * lens[2], 5e-5 * lens[2], -3e-5 * lens[1], 2e-5 * lens[0]])
    out = []
    for _ in range(2):
        o = po.Oracle(relaxed(small_pe), params())
        s, nts = o.eval(strain, 2.0, 300.0, 1e-4, 20)
        out.append(s)
        assert nts == 10
    assert np.array_equal(out[0], out[1])
    assert np.all(np.isfinite(out[0]))


def test_init_material_restatement_properties(small_pe):
    """CPU restatement of EQMDProblem::lammps_equilibration: lengths are the box, the stiffness is the
    symmetrised 6x6 of the script mapped by the reference's index rule, and the procedure leaves the oracle's state alone."""
    from oracle import pyoracle as po
    kw = dict(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=1e-5)
    o = po.Oracle(small_pe, po.default_params(**kw))
    b0, x0, v0 = o.get_state()
    length, stress, stiff = po.init_material(o, 2.0, 300.0, 10, 0.005, 2.5e-4)
    assert np.allclose(length, small_pe["box"][3:6] - small_pe["box"][:3])
    assert np.all(np.isfinite(stress)) and np.all(np.isfinite(stiff))
    assert np.allclose(stiff, stiff.T, rtol=0, atol=1e-12 * np.abs(stiff).max())
    # a stiff crystal: the normal moduli (file indices 0, 3, 5 = 0000, 1111, 2222) are positive and of GPa order
    assert all(1e8 < stiff[i, i] < 1e12 for i in (0, 3, 5))
    b1, x1, v1 = o.get_state()
    assert np.array_equal(np.asarray(b0), np.asarray(b1)) and np.array_equal(x0, x1) and np.array_equal(v0, v1)


def test_shake_clusters_of_two_and_four():
    """fix shake ... m 1.0 on a molecular system: ethane gives star clusters of 4 (C + 3 H), the heavy-H
    diatomics clusters of 2 (closed form); every constrained bond stays at r0 within the tolerance, the constraint
    count enters the degrees of freedom."""
    from oracle import pyoracle as po
    from scema_amd.systems import build_ethane_oh
    d = build_ethane_oh()
    kw = dict(cut_lj=5.5, cut_coul=5.0, skin=1.0, kspace_accuracy=1e-5)
    o = po.Oracle(d, po.default_params(**kw))
    o.setup(use_shake=True)
    nmol = d["natoms"] // 10
    assert o.nclusters == 3 * nmol and o.nconstraints == 7 * nmol
    assert o.tdof == 3 * d["natoms"] - 3 - 7 * nmol
    o.run(30, 1.0, 200.0, nvt=True, use_shake=True)
    _, x, _ = o.get_state()
    x = np.asarray(x).reshape(-1, 3)
    typ = d["type"]
    worst = 0.0
    for (a, b), t in zip(d["bonds"], d["bond_type"]):
        if typ[a] == 1 or typ[b] == 1:
            r = np.linalg.norm(x[a] - x[b])          # molecules are whole (positions are unwrapped)
            worst = max(worst, abs(r - d["bond_coeff"][t, 1]) / d["bond_coeff"][t, 1])
    assert worst < 2e-3          # fix shake 0.001: relative bond-length tolerance per iteration sweep
