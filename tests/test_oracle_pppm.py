"""Pins for the oracle's PPPM (`kspace_style pppm 1e-4`, lammps_scripts_opls/in.set.lammps:36): order-5
charge assignment in lamda coordinates, optimal influence function for ik differentiation, grid and g_ewald by the rules of
pppm.cpp as restated in oracle/md_oracle.c.  PARITY UNPINNED (no LAMMPS); what is checked: the mesh sum converges to the plain
Ewald sum as the accuracy is tightened, at the rate asked for; its forces carry no net force; the estimate that picks the grid
is honest; a triclinic cell is treated like its orthogonal image."""
import numpy as np
import pytest

from oracle import pyoracle as po

KW = dict(cut_lj=5.0, cut_coul=4.0, skin=1.0)


def _run(d, acc, pppm, mesh=None):
    p = po.default_params(kspace_accuracy=acc, kspace_pppm=1 if pppm else 0, **KW)
    if mesh is not None:
        p.pppm_mesh[:] = list(mesh)
    o = po.Oracle(d, p)
    o.setup(False)
    f, e, w = o.compute()
    return o, f, e, w


def test_pppm_converges_to_the_ewald_sum(small_pe):
    _, fr, er, wr = _run(small_pe, 1e-10, False)           # tight Ewald reference
    frms = np.sqrt((fr ** 2).sum(1).mean())
    errs = []
    for acc in (1e-4, 1e-5, 1e-6):
        o, f, e, w = _run(small_pe, acc, True)
        err = np.sqrt(((f - fr) ** 2).sum(1).mean())
        errs.append(err)
        nx, ny, nz = o.pppm_grid
        assert min(nx, ny, nz) >= 6 and all(_factorable(n) for n in (nx, ny, nz))
        # accuracy is an absolute RMS force error relative to the force between two unit charges 1 A apart (332 kcal/mol/A)
        assert err < 8.0 * acc * 332.06371, (acc, err)
        assert np.abs(f.sum(0)).max() < 1e-9 * frms * len(f)         # no net force
        assert abs((e[1] + e[6]) - (er[1] + er[6])) < 3e4 * acc * max(1.0, abs(er[1] + er[6]))
    assert errs[1] < 0.3 * errs[0] and errs[2] < 0.3 * errs[1]
    # g_ewald was adjusted upward from the initial estimate the Ewald path keeps
    oe, *_ = _run(small_pe, 1e-4, False)
    op, *_ = _run(small_pe, 1e-4, True)
    assert oe.g_ewald < op.g_ewald < 1.2 * oe.g_ewald and oe.pppm_grid == (0, 0, 0)


def _factorable(n):
    for p in (2, 3, 5):
        while n % p == 0:
            n //= p
    return n == 1


def test_pppm_in_a_triclinic_cell_equals_its_lattice_equivalent(small_pe):
    """xy = +lx/2 and xy = -lx/2 (atoms unchanged) are two representations of one lattice (what a box flip switches between): the
    mesh sum, done in lamda coordinates, must not care beyond its own discretisation error (the two grids cut space differently).
    Both on the grid the rule gives the first one (`kspace_modify mesh`): set_grid_global's triclinic rescaling depends on the
    SIGN of the tilt (int(xy nx / xprd + ny) + 1), so left to itself the second representation would get a much coarser grid."""
    from copy import deepcopy
    d1, d2 = deepcopy(small_pe), deepcopy(small_pe)
    lx = d1["box"][3] - d1["box"][0]
    d1["box"][6] = 0.5 * lx
    d2["box"][6] = -0.5 * lx
    o1, f1, e1, _ = _run(d1, 1e-6, True)
    o2, f2, e2, _ = _run(d2, 1e-6, True, mesh=o1.pppm_grid)
    assert o2.pppm_grid == o1.pppm_grid
    frms = np.sqrt((f1 ** 2).sum(1).mean())
    assert np.sqrt(((f1 - f2) ** 2).sum(1).mean()) < 2e-5 * frms
    assert abs((e1[1] + e1[6]) - (e2[1] + e2[6])) < 2e-3


def test_pppm_short_trajectory_conserves_energy_like_the_ewald_run(small_pe):
    from copy import deepcopy
    d = deepcopy(small_pe)
    d["eps"] = d["eps"] * 0.0
    res = []
    for pppm in (False, True):
        o = po.Oracle(d, po.default_params(kspace_accuracy=1e-5, kspace_pppm=1 if pppm else 0, shake_mass=0.0, **KW))
        _, tr = o.run(60, 0.25, 300.0, nvt=False, use_shake=False, trace=True)
        et = tr[:, 1] + tr[:, 2]
        res.append(np.abs(et - et[0]).max() / tr[:, 2].mean())
    assert res[1] < 5e-3 and res[1] < 5.0 * res[0] + 1e-3          # ik differentiation is not exactly conservative, but close


def _estimate_ik_error(h, prd, g, q2, natoms):
    acons = [1.0 / 23232.0, 7601.0 / 13628160.0, 143.0 / 69120.0, 517231.0 / 106536960.0, 106640677.0 / 11737571328.0]
    s = sum(a * (h * g) ** (2.0 * m) for m, a in enumerate(acons))
    return q2 * (h * g) ** 5.0 * np.sqrt(g * prd * np.sqrt(2.0 * np.pi) * s / natoms) / (prd * prd)


def _set_grid_global(box, g, acc, q2, natoms):
    """PPPM::set_grid_global of pppm.cpp (17Nov16, ik, triclinic) restated a second time, line by line, in Python: the search
    starts at int(prd g) + 1 with h = 1 / g, the increment FOLLOWS the evaluation (one past the first admissible grid), the
    triclinic rescaling int(lamda2xT(n / prd)) + 1, then products of 2, 3, 5."""
    lo, hi, (xy, xz, yz) = box[:3], box[3:6], box[6:9]
    prd = hi - lo
    n = []
    for d in range(3):
        h = 1.0 / g
        nd = int(prd[d] / h) + 1
        err = _estimate_ik_error(h, prd[d], g, q2, natoms)
        while err > acc:
            err = _estimate_ik_error(h, prd[d], g, q2, natoms)
            nd += 1
            h = prd[d] / nd
        n.append(nd)
    t = [n[0] / prd[0], n[1] / prd[1], n[2] / prd[2]]
    u = [prd[0] * t[0], xy * t[0] + prd[1] * t[1], xz * t[0] + yz * t[1] + prd[2] * t[2]]
    n = [int(v + 1e-9) + 1 for v in u]          # guard: (n / prd) * prd must not decide the grid by its last bit (md_oracle.c)
    return tuple(next(m for m in range(v, 10 * v + 8) if _factorable(m)) for v in n)


@pytest.mark.parametrize("tilt", [(0.0, 0.0, 0.0), (0.7, -0.4, 0.5), (-0.7, 0.4, -0.5), (3.0, 0.0, 0.0)])
@pytest.mark.parametrize("acc", [1e-4, 1e-5])
def test_grid_follows_set_grid_global(small_pe, tilt, acc):
    """The oracle's C restatement against the Python one above, for boxes whose tilts push the triclinic rescaling both ways,
    and one hand-checked property: the result is never below the first admissible grid + 1 per dimension when the search ran."""
    from copy import deepcopy
    d = deepcopy(small_pe)
    d["box"][6:9] = tilt
    o = po.Oracle(d, po.default_params(kspace_accuracy=acc, kspace_pppm=0, **KW))
    o.setup(False)
    g0 = o.g_ewald                                     # the initial estimate (the Ewald path keeps it)
    q2 = float((np.asarray(d["charge"]) ** 2).sum()) * 332.06371
    exp = _set_grid_global(np.asarray(d["box"], float), g0, acc * 332.06371, q2, d["natoms"])
    op, *_ = _run(d, acc, True)
    assert op.pppm_grid == exp, (op.pppm_grid, exp)
    # a box that differs in the 13th digit gets the same grid (without the guard the x grid, which no tilt touches, flips)
    d2 = deepcopy(d)
    d2["box"] = np.asarray(d["box"], float) * (1.0 + 3e-13)
    op2, *_ = _run(d2, acc, True)
    assert op2.pppm_grid == op.pppm_grid
    # adjust_gewald stops at the first Newton iterate with |f| < 1e-5: the residual is small but not converged to round-off
    assert op.g_ewald != g0


# ---- the PRODUCT's set-up (scema_md_kspace_setup: engine/engine_kspace.cpp, a pure host function of the C ABI) against the Python
# restatement, with no C oracle in between (product and C oracle share a hand; this chain does not pass through it) ----
def _initial_g(acc, rc, prd, q2, natoms):
    """KSpace g_ewald estimate of pppm.cpp init(): accuracy*sqrt(N rc V)/(2 q2) -> sqrt(-log)/rc, or the fallback"""
    t = acc * np.sqrt(natoms * rc * prd[0] * prd[1] * prd[2]) / (2.0 * q2)
    return (1.35 - 0.15 * np.log(acc)) / rc if t >= 1.0 else np.sqrt(-np.log(t)) / rc


def _adjust_gewald(box, grid, g, acc, rc, q2, natoms):
    """PPPM::adjust_gewald / newton_raphson_f / compute_qopt-free ik branch of pppm.cpp (17Nov16), line by line: Newton steps on
    f(g) = real-space error - k-space error with a forward difference of 1e-6, at most 10 000, stopped at the first iterate with
    |f| < 1e-5; the grid spacings of a triclinic box are the reciprocals of x2lamdaT(n)."""
    lo, hi, (xy, xz, yz) = box[:3], box[3:6], box[6:9]
    prd = hi - lo
    # h_inv of domain.cpp (triclinic): [1/xprd, 1/yprd, 1/zprd, -yz/(yprd zprd), (yz xy - yprd xz)/(xprd yprd zprd), -xy/(xprd yprd)]
    hinv = [1.0 / prd[0], 1.0 / prd[1], 1.0 / prd[2], -yz / (prd[1] * prd[2]), (yz * xy - prd[1] * xz) / (prd[0] * prd[1] * prd[2]), -xy / (prd[0] * prd[1])]
    # x2lamdaT(v) = (h_inv[0] v0, h_inv[5] v0 + h_inv[1] v1, h_inv[4] v0 + h_inv[3] v1 + h_inv[2] v2)
    t = [hinv[0] * grid[0], hinv[5] * grid[0] + hinv[1] * grid[1], hinv[4] * grid[0] + hinv[3] * grid[1] + hinv[2] * grid[2]]
    hs = [1.0 / t[0], 1.0 / t[1], 1.0 / t[2]]

    def f(gg):
        df_r = 2.0 * q2 * np.exp(-gg * gg * rc * rc) / np.sqrt(natoms * rc * prd[0] * prd[1] * prd[2])
        e = [_estimate_ik_error(hs[d], prd[d], gg, q2, natoms) for d in range(3)]
        return df_r - np.sqrt(e[0] ** 2 + e[1] ** 2 + e[2] ** 2) / np.sqrt(3.0)

    for _ in range(10000):
        dx = 0.000001
        f1, f2 = f(g), f(g + dx)
        g -= f1 / ((f2 - f1) / dx)
        if abs(f(g)) < 0.00001:
            break
    return g


@pytest.mark.parametrize("tilt", [(0.0, 0.0, 0.0), (0.7, -0.4, 0.5), (-0.7, 0.4, -0.5), (3.0, 0.0, 0.0)])
@pytest.mark.parametrize("acc", [1e-4, 1e-5])
def test_the_products_pppm_setup_follows_the_python_restatement(small_pe, tilt, acc):
    from copy import deepcopy
    from scema_amd import capi
    d = deepcopy(small_pe)
    d["box"][6:9] = tilt
    box = np.asarray(d["box"], float)
    qsq = float((np.asarray(d["charge"]) ** 2).sum())
    q2 = qsq * 332.06371
    P = capi.default_params(kspace_accuracy=acc, **KW)
    g0, g1, grid = capi.kspace_setup(P, box, qsq, d["natoms"])
    a = acc * 332.06371
    g0_py = _initial_g(a, KW["cut_coul"], box[3:6] - box[:3], q2, d["natoms"])
    assert abs(g0 - g0_py) < 1e-14 * g0_py
    grid_py = _set_grid_global(box, g0_py, a, q2, d["natoms"])
    assert grid == grid_py, (grid, grid_py)
    g1_py = _adjust_gewald(box, grid_py, g0_py, a, KW["cut_coul"], q2, d["natoms"])
    # (a Newton step with a 1e-6 forward difference multiplies the last bits of f by a million: 1e-9 is what two correct
    # implementations in different arithmetic libraries agree to; a wrong rule is off in the second digit)
    assert abs(g1 - g1_py) < 1e-9 * g1_py and g1 != g0
    # the 13th-digit box gets the same grid from the product too
    g0b, g1b, gridb = capi.kspace_setup(P, box * (1.0 + 3e-13), qsq, d["natoms"])
    assert gridb == grid
    # Ewald sum: the initial estimate stands, no grid; an uncharged system: nothing
    assert capi.kspace_setup(capi.default_params(kspace_accuracy=acc, kspace_style=0, **KW), box, qsq, d["natoms"])[1:] == (g0, (0, 0, 0))
    assert capi.kspace_setup(P, box, 0.0, d["natoms"]) == (0.0, 0.0, (0, 0, 0))


def test_the_products_pppm_setup_at_the_reference_size():
    """PE-10k at the reference's settings (12/9 A, 1e-4): the grid DESIGN.md quotes (12 x 12 x 10..12) is what both restatements give"""
    from scema_amd import capi
    from scema_amd.systems import build_pe
    d = build_pe(6, 9, 16)
    box = np.asarray(d["box"], float)
    qsq = float((np.asarray(d["charge"]) ** 2).sum())
    g0, g1, grid = capi.kspace_setup(capi.default_params(), box, qsq, d["natoms"])
    a, q2 = 1e-4 * 332.06371, qsq * 332.06371
    g0_py = _initial_g(a, 9.0, box[3:6] - box[:3], q2, d["natoms"])
    grid_py = _set_grid_global(box, g0_py, a, q2, d["natoms"])
    assert grid == grid_py and grid[0] == 12 and grid[1] == 12 and grid[2] in (10, 12)
    assert abs(g1 - _adjust_gewald(box, grid_py, g0_py, a, 9.0, q2, d["natoms"])) < 1e-9 * g1 and 0.20 < g1 < 0.23
