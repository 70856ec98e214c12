"""fix deform's triclinic box flip (LAMMPS default `flip yes`, used by in.strain.lammps:94-100 through `fix deform ... erate
... remap x`) in the CPU oracle: the rule itself on hand-derived vectors, and the physics it must not change -- a flip only
re-expresses the same lattice.  SURVEY.md A.5; VERDICT r01 item 6."""
import numpy as np
import pytest


def test_flip_rule_golden_vectors():
    from oracle import pyoracle as po
    # xy beyond +Lx/2: one box length down
    f, t, n = po.tilt_flip([5.2, 0.0, 0.0], 10.0, 8.0)
    assert f and np.allclose(t, [-4.8, 0.0, 0.0]) and list(n) == [-1, 0, 0]
    # yz beyond -Ly/2: yz += Ly and xz += xy (a3' = a3 + a2)
    f, t, n = po.tilt_flip([1.5, 0.25, -4.1], 10.0, 8.0)
    assert f and np.allclose(t, [1.5, 1.75, 3.9]) and list(n) == [0, 0, 1]
    # yz flip that pushes xz over the edge as well: both flip
    f, t, n = po.tilt_flip([4.0, 3.0, -4.5], 10.0, 8.0)
    assert f and np.allclose(t, [4.0, -3.0, 3.5]) and list(n) == [0, -1, 1]
    # inside +-1/2: nothing
    f, t, n = po.tilt_flip([4.9, -4.9, 3.9], 10.0, 8.0)
    assert not f and np.allclose(t, [4.9, -4.9, 3.9]) and list(n) == [0, 0, 0]


def test_tilt_target_continues_from_a_flipped_box():
    from oracle import pyoracle as po
    # raw target 13.0 (start + rate t) while the box sits at xy/Lx = -0.45 after a flip: the closest image is -7.0
    t = po.tilt_closest([13.0, 0.0, 0.0], 10.0, 8.0, -4.5, 0.0, 0.0, 10.0, 8.0)
    assert np.allclose(t, [-7.0, 0.0, 0.0])
    # small tilts are left alone (up to the rounding of the add/subtract pair LAMMPS' loop performs)
    t = po.tilt_closest([0.3001, -0.2, 0.1], 10.0, 8.0, 0.3, -0.2001, 0.1, 10.0, 8.0)
    assert np.allclose(t, [0.3001, -0.2, 0.1], rtol=0, atol=1e-14)


def _affine(box_old, box_new, x):
    """positions carried affinely from one triclinic box to another (what `change_box ... remap` does)"""
    def hmat(b):
        return np.array([[b[3] - b[0], b[6], b[7]], [0.0, b[4] - b[1], b[8]], [0.0, 0.0, b[5] - b[2]]])
    lam = np.linalg.solve(hmat(box_old), (np.asarray(x, float) - np.asarray(box_old[:3], float)).T)
    return (hmat(box_new) @ lam).T + np.asarray(box_new[:3], float)


def _oracle(d, **kw):
    from oracle import pyoracle as po
    base = dict(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=1e-5)
    base.update(kw)
    return po.Oracle(d, po.default_params(**base))


def test_the_two_representations_of_one_lattice_give_the_same_forces(small_pe):
    """xy = +Lx/2 and xy = -Lx/2 span the same lattice: energies, forces and virials of every part agree.  With the Ewald sum,
    which sees the lattice only; the PPPM grid follows the box edges, so its 1e-5 discretisation error differs between the two."""
    d1 = dict(small_pe); d2 = dict(small_pe)
    lx = small_pe["box"][3] - small_pe["box"][0]
    b1 = np.array(small_pe["box"], float); b2 = b1.copy()
    b1[6] = 0.5 * lx; b2[6] = -0.5 * lx
    d1["box"] = b1; d2["box"] = b2
    o1, o2 = _oracle(d1, kspace_pppm=0), _oracle(d2, kspace_pppm=0)
    o1.setup(True); o2.setup(True)
    f1, e1, w1 = o1.compute(); f2, e2, w2 = o2.compute()
    assert o1.npairs == o2.npairs
    assert np.abs(f1 - f2).max() < 1e-9 * np.abs(f1).max()
    assert np.abs(e1 - e2).max() < 1e-9 * np.abs(e1).max()
    assert np.abs(w1 - w2).max() < 1e-9 * np.abs(w1).max()


def test_a_shear_run_flips_and_the_physics_does_not_notice(small_pe):
    from oracle import pyoracle as po
    d = dict(small_pe)
    box = np.array(small_pe["box"], float)
    lx, ly = box[3] - box[0], box[4] - box[1]
    box[6] = 0.485 * lx
    d["box"] = box
    d["x"] = _affine(small_pe["box"], box, small_pe["x"])
    rate_xy = 0.004 * lx / ly          # xy grows by 0.004 Lx per fs: crosses Lx/2 after ~4 steps of 1 fs
    rates = np.array([0, 0, 0, rate_xy, 0, 0], float)
    # with the Ewald sum, whose k-vector list is what this test follows through the flip.  (With PPPM a NEW run on the flipped
    # box would also get another grid: set_grid_global's triclinic rescaling int(xy nx / xprd + ny) + 1 depends on the sign
    # of the tilt, tests/test_oracle_pppm.py.)
    o = _oracle(d, kspace_pppm=0)
    nsteps = 16
    _, tr = o.run(nsteps, 1.0, 300.0, nvt=False, use_shake=False, rates=rates, trace=True)
    assert o.nflips == 1
    bx, x, v = o.get_state()
    assert -0.5 * lx < bx[6] < -0.4 * lx                       # continued from the flipped tilt, not from the raw target
    e = tr[:, 1] + tr[:, 2]                                    # pe + ke: the work of the shear enters smoothly
    de = np.abs(np.diff(e))
    assert de.max() < 4.0 * np.median(de) + 1e-9
    # the k-vector list carried through the flip (integers re-expressed in the new reciprocal basis) describes the same
    # vectors as a freshly generated list for the flipped box, up to the few that crossed the truncation sphere while the
    # box sheared by 6 %: a wrong basis change would scramble S(k) and show at O(1)
    o.freeze_kspace(True)
    o.setup(False)
    _, e_carried, _ = o.compute()
    o.freeze_kspace(False)
    o.setup(False)
    _, e_fresh, _ = o.compute()
    ks = po.PARTS.index("kspace")
    assert abs(e_carried[ks] - e_fresh[ks]) < 1e-4 * abs(e_fresh[ks])
