"""Pins for the CPU oracle's static physics (SURVEY.md §8(c) pins 2-4): energy-force and
energy-virial consistency for every term, Madelung constant, LJ lattice sum vs brute force,
closed forms.  These are what stands in for reference golden vectors (parity unpinned)."""
import numpy as np
import pytest

from oracle import pyoracle as po

QQR2E = 332.06371


def small_params(**kw):
    base = dict(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=1e-6)
    base.update(kw)
    return po.default_params(**base)


def test_energy_force_consistency_per_term(small_pe):
    # with the Ewald sum: PPPM's ik differentiation gives forces that are not the exact gradient of its energy (1e-5 here)
    o = po.Oracle(small_pe, small_params(kspace_pppm=0))
    o.setup(use_shake=False)
    f, e, w = o.compute()
    box, x, v = o.get_state()
    # central finite differences of every energy part; analytic force is the total, so compare sums
    h = 1e-5
    rng = np.random.default_rng(0)
    for i in rng.choice(o.n, 6, replace=False):
        for k in range(3):
            xp = x.copy(); xp[i, k] += h
            o.set_state(box, xp, v); _, ep, _ = o.compute()
            xm = x.copy(); xm[i, k] -= h
            o.set_state(box, xm, v); _, em, _ = o.compute()
            fd = -(ep.sum() - em.sum()) / (2 * h)
            assert abs(fd - f[i, k]) <= 1e-7 * max(1.0, abs(f[i, k])), (i, k, fd, f[i, k])
    # total force vanishes (Newton 3 in every term)
    assert np.abs(f.sum(0)).max() < 1e-9


def _deform(box, x, eta):
    """x' = (I+eta) x with eta upper triangular keeps LAMMPS' restricted triclinic form."""
    F = np.eye(3) + eta
    lo, hi = box[:3], box[3:6]
    a = np.array([hi[0] - lo[0], 0, 0]); b = np.array([box[6], hi[1] - lo[1], 0]); c = np.array([box[7], box[8], hi[2] - lo[2]])
    a2, b2, c2, lo2 = F @ a, F @ b, F @ c, F @ lo
    nb = np.array([lo2[0], lo2[1], lo2[2], lo2[0] + a2[0], lo2[1] + b2[1], lo2[2] + c2[2], b2[0], c2[0], c2[1]])
    return nb, x @ F.T


@pytest.mark.parametrize("comp", range(6))
def test_energy_virial_consistency(small_pe, comp):
    """dE/d(eta_ab) = -W_ab for every part, all six components including the tilts (pin 3):
    this is the test that pins the configurational stress."""
    o = po.Oracle(small_pe, small_params())
    o.setup(use_shake=False)
    o.freeze_kspace(True)
    f, e, w = o.compute()
    box, x, v = o.get_state()
    ab = [(0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2)][comp]
    h = 2e-7
    es = []
    for sgn in (+1, -1):
        eta = np.zeros((3, 3)); eta[ab] = sgn * h
        nb, nx = _deform(box, x, eta)
        o.set_state(nb, nx, v)
        o.setup(use_shake=False)
        _, ee, _ = o.compute()
        es.append(ee)
    dE = (es[0] - es[1]) / (2 * h)
    for part in range(po.NPART - 1):
        scale = max(1.0, np.abs(w[part]).max())
        assert abs(dE[part] + w[part, comp]) < 2e-5 * scale, (po.PARTS[part], comp, dE[part], w[part, comp])


def test_madelung_nacl():
    """NaCl Madelung constant 1.747565 through real-space + Ewald sum (pin 4)."""
    a = 5.64
    ncell = 4
    basis = [(0, 0, 0, 1), (.5, .5, 0, 1), (.5, 0, .5, 1), (0, .5, .5, 1),
             (.5, 0, 0, -1), (0, .5, 0, -1), (0, 0, .5, -1), (.5, .5, .5, -1)]
    x, q = [], []
    for i in range(ncell):
        for j in range(ncell):
            for k in range(ncell):
                for bx, by, bz, qq in basis:
                    x.append([(i + bx) * a, (j + by) * a, (k + bz) * a]); q.append(qq)
    x = np.array(x); q = np.array(q, float)
    n = len(q)
    L = ncell * a
    d = dict(natoms=n, ntypes=1, type=np.zeros(n, np.int32), charge=q, mass=np.array([1.0e5]),
             eps=np.zeros((1, 1)), sigma=np.ones((1, 1)),
             bonds=np.zeros((0, 2), np.int32), bond_type=np.zeros(0, np.int32), bond_coeff=np.zeros((0, 2)),
             angles=np.zeros((0, 3), np.int32), angle_type=np.zeros(0, np.int32), angle_coeff=np.zeros((0, 2)),
             dihedrals=np.zeros((0, 4), np.int32), dihedral_type=np.zeros(0, np.int32), dihedral_coeff=np.zeros((0, 4)),
             impropers=np.zeros((0, 4), np.int32), improper_type=np.zeros(0, np.int32), improper_coeff=np.zeros((0, 2)),
             special_lj=np.ones(3), special_coul=np.ones(3),
             box=np.array([0, 0, 0, L, L, L, 0, 0, 0.0]), x=x, v=np.zeros_like(x))
    o = po.Oracle(d, po.default_params(cut_lj=9.0, cut_coul=9.0, skin=1.0, kspace_accuracy=1e-8, shake_mass=0.0))
    o.setup(use_shake=False)
    f, e, w = o.compute()
    etot = e[1] + e[6]
    madelung = -etot / (n / 2) * (a / 2) / QQR2E
    assert abs(madelung - 1.747565) < 2e-5, madelung
    assert np.abs(f).max() < 1e-4   # perfect lattice
    # virial theorem for a pure 1/r system: trace W = E
    assert abs((w[1, :3].sum() + w[6, :3].sum()) - etot) < 1e-4 * abs(etot)


def test_lj_fcc_lattice_vs_bruteforce():
    """LJ fcc crystal: energy and virial of the cell-list path equal a direct O(N^2 x images) sum."""
    a = 5.3
    nc = 4
    basis = [(0, 0, 0), (.5, .5, 0), (.5, 0, .5), (0, .5, .5)]
    x = np.array([[(i + b[0]) * a, (j + b[1]) * a, (k + b[2]) * a] for i in range(nc) for j in range(nc)
                  for k in range(nc) for b in basis])
    rng = np.random.default_rng(3)
    x += rng.normal(0, 0.05, x.shape)
    n = len(x); L = nc * a
    eps, sig, rc = 0.238, 3.405, 8.5
    d = dict(natoms=n, ntypes=1, type=np.zeros(n, np.int32), charge=np.zeros(n), mass=np.array([39.95]),
             eps=np.array([[eps]]), sigma=np.array([[sig]]),
             bonds=np.zeros((0, 2), np.int32), bond_type=np.zeros(0, np.int32), bond_coeff=np.zeros((0, 2)),
             angles=np.zeros((0, 3), np.int32), angle_type=np.zeros(0, np.int32), angle_coeff=np.zeros((0, 2)),
             dihedrals=np.zeros((0, 4), np.int32), dihedral_type=np.zeros(0, np.int32), dihedral_coeff=np.zeros((0, 4)),
             impropers=np.zeros((0, 4), np.int32), improper_type=np.zeros(0, np.int32), improper_coeff=np.zeros((0, 2)),
             special_lj=np.ones(3), special_coul=np.ones(3),
             box=np.array([0, 0, 0, L, L, L, 0, 0, 0.0]), x=x, v=np.zeros_like(x))
    o = po.Oracle(d, po.default_params(cut_lj=rc, cut_coul=rc, skin=2.0, shake_mass=0.0))
    o.setup(use_shake=False)
    f, e, w = o.compute()
    # brute force with explicit images
    E = 0.0; W = np.zeros(6); F = np.zeros_like(x)
    shifts = np.array([[i, j, k] for i in (-1, 0, 1) for j in (-1, 0, 1) for k in (-1, 0, 1)]) * L
    for s in shifts:
        dx = x[:, None, :] - (x[None, :, :] + s)
        r2 = (dx ** 2).sum(-1)
        m = (r2 < rc * rc) & (r2 > 1e-12)
        r2i = np.where(m, 1.0 / np.where(m, r2, 1.0), 0.0)
        r6 = r2i ** 3 * sig ** 6
        E += 0.5 * (4 * eps * (r6 * r6 - r6) * m).sum()
        fp = (48 * eps * r6 * r6 - 24 * eps * r6) * r2i
        F += (dx * fp[:, :, None]).sum(1)
        for c, (p, q) in enumerate([(0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2)]):
            W[c] += 0.5 * (dx[:, :, p] * dx[:, :, q] * fp).sum()
    assert abs(e[0] - E) < 1e-9 * abs(E)
    assert np.allclose(f, F, rtol=1e-9, atol=1e-9)
    assert np.allclose(w[0], W, rtol=1e-9, atol=1e-8)


def _mini(x, q, types, eps, sigma, L=60.0, **topo):
    n = len(x)
    z = lambda *s: np.zeros(s, np.int32)
    d = dict(natoms=n, ntypes=len(eps), type=np.array(types, np.int32), charge=np.array(q, float),
             mass=np.full(len(eps), 12.0), eps=np.array(eps, float), sigma=np.array(sigma, float),
             bonds=z(0, 2), bond_type=z(0), bond_coeff=np.zeros((0, 2)),
             angles=z(0, 3), angle_type=z(0), angle_coeff=np.zeros((0, 2)),
             dihedrals=z(0, 4), dihedral_type=z(0), dihedral_coeff=np.zeros((0, 4)),
             impropers=z(0, 4), improper_type=z(0), improper_coeff=np.zeros((0, 2)),
             special_lj=np.array([0, 0, 1.0]), special_coul=np.array([0, 0, 1.0]),
             box=np.array([0, 0, 0, L, L, L, 0, 0, 0.0]), x=np.array(x, float) + L / 2, v=np.zeros((n, 3)))
    d.update(topo)
    return d


def test_two_body_lj_closed_form():
    r = 4.2
    d = _mini([[0, 0, 0], [r, 0, 0]], [0, 0], [0, 0], [[0.1]], [[3.4]])
    o = po.Oracle(d, po.default_params(shake_mass=0.0))
    o.setup(False)
    f, e, w = o.compute()
    s6 = (3.4 / r) ** 6
    assert abs(e[0] - 4 * 0.1 * (s6 * s6 - s6)) < 1e-14
    fr = 24 * 0.1 * (2 * s6 * s6 - s6) / r
    assert abs(f[1, 0] - fr) < 1e-13 and abs(f[0, 0] + fr) < 1e-13
    assert abs(w[0, 0] - fr * r) < 1e-12
    # beyond the 12 A cutoff: nothing
    d2 = _mini([[0, 0, 0], [12.5, 0, 0]], [0, 0], [0, 0], [[0.1]], [[3.4]])
    o2 = po.Oracle(d2, po.default_params(shake_mass=0.0)); o2.setup(False)
    assert o2.compute()[1][0] == 0.0


def test_bond_angle_closed_forms():
    th = np.deg2rad(100.0)
    x = [[1.1, 0, 0], [0, 0, 0], [1.2 * np.cos(th), 1.2 * np.sin(th), 0]]
    d = _mini(x, [0, 0, 0], [0, 0, 0], [[0.0]], [[1.0]],
              bonds=np.array([[0, 1], [1, 2]], np.int32), bond_type=np.array([0, 0], np.int32),
              bond_coeff=np.array([[300.0, 1.0]]),
              angles=np.array([[0, 1, 2]], np.int32), angle_type=np.array([0], np.int32),
              angle_coeff=np.array([[50.0, np.deg2rad(109.5)]]))
    o = po.Oracle(d, po.default_params(shake_mass=0.0)); o.setup(False)
    f, e, w = o.compute()
    assert abs(e[2] - 300.0 * (0.1 ** 2 + 0.2 ** 2)) < 1e-12          # E = K (r-r0)^2
    assert abs(e[3] - 50.0 * (th - np.deg2rad(109.5)) ** 2) < 1e-12  # E = K (theta-theta0)^2
    # bond force on atom 0 along +x : -2K(r-r0) ; angle force on atom 0 is perpendicular to its bond
    fb0 = -2 * 300.0 * 0.1
    fa = f[0] - np.array([fb0, 0, 0])
    assert abs(fa[0]) < 1e-10
    # |F_angle on 0| = |dE/dtheta| / r1
    assert abs(abs(fa[1]) - abs(2 * 50.0 * (th - np.deg2rad(109.5))) / 1.1) < 1e-10


@pytest.mark.parametrize("phi_deg", [0.0, 60.0, 90.0, 180.0, 137.0])
def test_opls_dihedral_values(phi_deg):
    """E = K1/2(1+cos p) + K2/2(1-cos 2p) + K3/2(1+cos 3p) + K4/2(1-cos 4p); trans = 180 deg."""
    phi = np.deg2rad(phi_deg)
    x = [[1.0, 1.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 0.0], [np.cos(phi), 0.0, np.sin(phi)]]
    K = [1.3, -0.05, 0.2, 0.7]
    d = _mini(x, [0] * 4, [0] * 4, [[0.0]], [[1.0]],
              dihedrals=np.array([[0, 1, 2, 3]], np.int32), dihedral_type=np.array([0], np.int32),
              dihedral_coeff=np.array([K]))
    o = po.Oracle(d, po.default_params(shake_mass=0.0)); o.setup(False)
    f, e, w = o.compute()
    expect = 0.5 * (K[0] * (1 + np.cos(phi)) + K[1] * (1 - np.cos(2 * phi)) + K[2] * (1 + np.cos(3 * phi)) + K[3] * (1 - np.cos(4 * phi)))
    assert abs(e[4] - expect) < 1e-12
    # torque balance and zero net force
    assert np.abs(f.sum(0)).max() < 1e-12
    xx = np.array(x)
    assert np.abs(np.cross(xx, f).sum(0)).max() < 1e-11


def test_special_pairs_cancel_kspace():
    """1-2 pair with weight 0: real-space correction + k-space == no Coulomb interaction between
    the two bonded atoms, i.e. the pair behaves as two isolated charges in the periodic background
    (SURVEY.md A.3).  Checked by comparing against the same two charges not bonded, minus the
    bare Coulomb term."""
    r = 1.1
    x = [[0, 0, 0], [r, 0, 0]]
    common = dict()
    bonded = _mini(x, [0.4, -0.4], [0, 0], [[0.0]], [[1.0]], L=40.0,
                   bonds=np.array([[0, 1]], np.int32), bond_type=np.array([0], np.int32),
                   bond_coeff=np.array([[0.0, 1.0]]))
    free = _mini(x, [0.4, -0.4], [0, 0], [[0.0]], [[1.0]], L=40.0)
    p = po.default_params(shake_mass=0.0, kspace_accuracy=1e-7)
    ob = po.Oracle(bonded, p); ob.setup(False)
    of = po.Oracle(free, p); of.setup(False)
    fb, eb, wb = ob.compute()
    ff, ef, wf = of.compute()
    bare = QQR2E * 0.4 * (-0.4) / r
    assert abs((ef[1] + ef[6]) - (eb[1] + eb[6]) - bare) < 1e-9
    assert abs((ff[0, 0] - fb[0, 0]) + bare / r) < 1e-9   # attraction: F_x on atom 0 = -bare/r > 0


def _improper_only(x, K=12.5, chi0_deg=20.0):
    """Four atoms in a 60 A box that interact through ONE harmonic improper only (no LJ, no charge, no bonds)."""
    z = lambda *sh: np.zeros(sh, np.int32)
    return dict(natoms=4, ntypes=1, type=z(4), charge=np.zeros(4), mass=np.array([12.0]), eps=np.zeros((1, 1)), sigma=np.ones((1, 1)),
                bonds=z(0, 2), bond_type=z(0), bond_coeff=np.zeros((0, 2)), angles=z(0, 3), angle_type=z(0), angle_coeff=np.zeros((0, 2)),
                dihedrals=z(0, 4), dihedral_type=z(0), dihedral_coeff=np.zeros((0, 4)),
                impropers=np.array([[0, 1, 2, 3]], np.int32), improper_type=z(1), improper_coeff=np.array([[K, np.deg2rad(chi0_deg)]]),
                special_lj=np.ones(3), special_coul=np.ones(3),
                box=np.array([0, 0, 0, 60, 60, 60, 0, 0, 0.0]), x=np.asarray(x, float), v=np.zeros((4, 3)))


def test_improper_harmonic_closed_form_and_consistency():
    """improper_style harmonic (in.set.lammps:56): E = K (chi - chi0)^2 with chi the angle between the planes
    (1,2,3) and (2,3,4); forces = -dE/dx by central differences; virial = sum r (x) f; zero net force and torque."""
    K, chi0 = 12.5, 20.0
    for chi in (35.0, 80.0, 140.0):
        c, s_ = np.cos(np.deg2rad(chi)), np.sin(np.deg2rad(chi))
        # atoms 2,3 on the z axis; atom 1 in the xz plane, atom 4 rotated by chi about z
        x = np.array([[30 + 1.1, 30, 30 - 0.4], [30, 30, 30], [30, 30, 31.5], [30 + 0.9 * c, 30 + 0.9 * s_, 31.9]])
        o = po.Oracle(_improper_only(x, K, chi0), po.default_params(shake_mass=0.0))
        o.setup(use_shake=False)
        f, e, w = o.compute()
        assert abs(e[5] - K * np.deg2rad(chi - chi0) ** 2) < 1e-10
        assert np.abs(e[[0, 1, 2, 3, 4, 6]]).max() == 0.0
        assert np.abs(f.sum(0)).max() < 1e-10 and np.abs(np.cross(x - x.mean(0), f).sum(0)).max() < 1e-9
        h = 1e-5
        for i in range(4):
            for k in range(3):
                ep = []
                for sgn in (1, -1):
                    xx = x.copy(); xx[i, k] += sgn * h
                    o2 = po.Oracle(_improper_only(xx, K, chi0), po.default_params(shake_mass=0.0)); o2.setup(use_shake=False)
                    ep.append(o2.compute()[1][5])
                assert abs(-(ep[0] - ep[1]) / (2 * h) - f[i, k]) < 1e-6 * max(1.0, np.abs(f).max())
        # virial of an isolated term = sum_i r_i (x) f_i
        wref = np.array([(x[:, a] * f[:, b]).sum() for a, b in ((0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2))])
        assert np.abs(w[5] - wref).max() < 1e-9 * max(1.0, np.abs(wref).max())


def lj_bench_system(nc=5):
    """The initial state of LAMMPS' own Lennard-Jones benchmark (bench/in.lj: fcc lattice at reduced density 0.8442,
    pair lj/cut 2.5, no shift, no tail correction) in units where eps = sigma = 1."""
    rho = 0.8442
    a = (4.0 / rho) ** (1.0 / 3.0)
    basis = [(0, 0, 0), (.5, .5, 0), (.5, 0, .5), (0, .5, .5)]
    x = np.array([[(i + b[0]) * a, (j + b[1]) * a, (k + b[2]) * a] for i in range(nc) for j in range(nc)
                  for k in range(nc) for b in basis])
    n = len(x); L = nc * a
    d = dict(natoms=n, ntypes=1, type=np.zeros(n, np.int32), charge=np.zeros(n), mass=np.array([1.0]),
             eps=np.array([[1.0]]), sigma=np.array([[1.0]]),
             bonds=np.zeros((0, 2), np.int32), bond_type=np.zeros(0, np.int32), bond_coeff=np.zeros((0, 2)),
             angles=np.zeros((0, 3), np.int32), angle_type=np.zeros(0, np.int32), angle_coeff=np.zeros((0, 2)),
             dihedrals=np.zeros((0, 4), np.int32), dihedral_type=np.zeros(0, np.int32), dihedral_coeff=np.zeros((0, 4)),
             impropers=np.zeros((0, 4), np.int32), improper_type=np.zeros(0, np.int32), improper_coeff=np.zeros((0, 2)),
             special_lj=np.ones(3), special_coul=np.ones(3),
             box=np.array([0, 0, 0, L, L, L, 0, 0, 0.0]), x=x, v=np.zeros_like(x))
    return d, rho


# Step-0 thermo lines (Step Temp E_pair E_mol TotEng Press) of two logs LAMMPS ships for this lattice, identical in every
# release of that era:
#   bench/log.*.lj.*    (32 000 atoms, T = 1.44):  0 1.44 -6.7733681 0 -4.6134356 ...
#   examples/melt/log.* ( 4 000 atoms, T = 3.0 ):  0 3    -6.7733681 0 -2.2744931 -3.7033504
# TotEng - E_pair = 3/2 T (N-1)/N and Press = rho T (N-1)/N + W/(3V) reproduce both lines to the last printed digit.
LAMMPS_LJ_EPAIR = -6.7733681
LAMMPS_LJ_BENCH_TOTENG = -4.6134356
LAMMPS_LJ_MELT_TOTENG, LAMMPS_LJ_MELT_PRESS = -2.2744931, -3.7033504


def lj_bench_check(e_lj, w_lj, n, rho):
    """E_pair, TotEng and Press the way LAMMPS prints them (thermo_style one, lj units); per-atom lattice sums do not
    depend on the number of cells, the kinetic parts use the logs' atom counts."""
    vol = n / rho
    epair = e_lj / n
    assert abs(epair - LAMMPS_LJ_EPAIR) < 6e-8                                                    # all printed digits
    assert abs(epair + 1.5 * 1.44 * (1 - 1 / 32000) - LAMMPS_LJ_BENCH_TOTENG) < 6e-8
    assert abs(epair + 1.5 * 3.0 * (1 - 1 / 4000) - LAMMPS_LJ_MELT_TOTENG) < 6e-8
    p = rho * 3.0 * (1 - 1 / 4000) + (w_lj[0] + w_lj[1] + w_lj[2]) / (3.0 * vol)
    assert abs(p - LAMMPS_LJ_MELT_PRESS) < 6e-8                                                   # all printed digits


def test_lammps_lj_benchmark_step0_known_answer():
    """Numbers LAMMPS itself publishes: the oracle's pair energy and virial pressure of the Lennard-Jones benchmark's
    initial lattice agree with the step-0 thermo lines of LAMMPS' bench and melt logs to every printed digit."""
    d, rho = lj_bench_system(5)
    o = po.Oracle(d, po.default_params(cut_lj=2.5, cut_coul=2.5, skin=0.3, shake_mass=0.0))
    o.setup(use_shake=False)
    f, e, w = o.compute()
    assert np.abs(f).max() < 1e-9          # perfect lattice
    lj_bench_check(e[0], w[0], d["natoms"], rho)
