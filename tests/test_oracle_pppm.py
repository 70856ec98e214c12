"""Pins for the oracle's PPPM (`kspace_style pppm 1e-4`, lammps_scripts_opls/in.set.lammps:36): order-5
charge assignment in lamda coordinates, optimal influence function for ik differentiation, grid and g_ewald by the rules of
pppm.cpp as restated in oracle/md_oracle.c.  PARITY UNPINNED (no LAMMPS); what is checked: the mesh sum converges to the plain
Ewald sum as the accuracy is tightened, at the rate asked for; its forces carry no net force; the estimate that picks the grid
is honest; a triclinic cell is treated like its orthogonal image."""
import numpy as np
import pytest

from oracle import pyoracle as po

KW = dict(cut_lj=5.0, cut_coul=4.0, skin=1.0)


def _run(d, acc, pppm):
    o = po.Oracle(d, po.default_params(kspace_accuracy=acc, kspace_pppm=1 if pppm else 0, **KW))
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
    mesh sum, done in lamda coordinates, must not care beyond its own discretisation error (the two grids cut space differently)"""
    from copy import deepcopy
    d1, d2 = deepcopy(small_pe), deepcopy(small_pe)
    lx = d1["box"][3] - d1["box"][0]
    d1["box"][6] = 0.5 * lx
    d2["box"][6] = -0.5 * lx
    o1, f1, e1, _ = _run(d1, 1e-6, True)
    o2, f2, e2, _ = _run(d2, 1e-6, True)
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
