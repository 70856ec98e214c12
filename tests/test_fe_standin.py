"""The continuum stand-in (include/scema_fe.h) and, through it, whole continuum steps of the hot
path: solve -> STMDSync::update -> check is the body of HMMProblem::do_timestep (dealammps.cc:417-474).  On CPU the MD is
the reference's own fake backend ("approximate md with hookes law"); BASELINE configs 1 and 3 give the shapes: a 3x3x8 mesh
= 576 quadrature points, 1 and 10 continuum steps."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def iso_stiffness(lam=60e9, mu=30e9):
    """isotropic C in init.*.stiff file order (00,01,02,11,12,22) x (00,01,02,11,12,22)"""
    pairs = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
    c = np.zeros((6, 6))
    d = np.eye(3)
    for a, (i, j) in enumerate(pairs):
        for b, (k, l) in enumerate(pairs):
            c[a, b] = lam * d[i, j] * d[k, l] + mu * (d[i, k] * d[j, l] + d[i, l] * d[j, k])
    return c


def hooke_raw(c, eps_raw):
    from oracle import pyoracle as po
    return po.hooke(c.ravel(), eps_raw)


def test_symbols_and_uniform_strain_field():
    from scema_amd import capi, fe
    L = capi.lib()
    for s in fe.SYMBOLS:
        assert hasattr(L, s), s
    nx, ny, nz, lx, ly, lz = 3, 3, 8, 0.03, 0.03, 0.08
    f = fe.FE(nx, ny, nz, lx, ly, lz, 1000.0, iso_stiffness(), dt=1e-7, min_qp_strain=1e-10)
    assert f.n_qp == 576 and f.n_nodes == 4 * 4 * 9                      # inputs_dogbone_cuboid.json 3x3x8, qpid = cell*8+q
    x = f.node_coords(nx, ny, nz, lx, ly, lz)
    gam = 2.0e3                                                           # 1/s
    v = np.zeros_like(x); v[:, 2] = gam * x[:, 2]; v[:, 0] = -0.3 * gam * x[:, 0]
    f.set_velocity(v)
    ul = f.solve()
    _, e, _ = f.get()
    up = slice(9 * 8, None)                                                # above the held bottom layer of cells
    assert np.allclose(e[:, 2], gam * 1e-7, rtol=1e-12) and np.allclose(e[up, 0], -0.3 * gam * 1e-7, rtol=1e-12)
    assert np.abs(e[up][:, [1, 3, 4, 5]]).max() < 1e-18
    # every point passed the threshold: 576 requests, first call: most_recent = none, id = cell*8+q
    assert [u[0] for u in ul] == list(range(576)) and all(u[1] == capi.QP_NONE for u in ul)
    assert np.allclose([u[3] for u in ul], e)


def test_below_threshold_the_point_continues_linear_elastically():
    from scema_amd import fe
    c = iso_stiffness()
    f = fe.FE(1, 1, 2, 0.01, 0.01, 0.02, 1000.0, c, dt=1e-7, min_qp_strain=1.0)    # threshold out of reach
    x = f.node_coords(1, 1, 2, 0.01, 0.01, 0.02)
    v = np.zeros_like(x); v[:, 2] = 50.0 * x[:, 2]
    f.set_velocity(v)
    assert f.solve() == []
    f.check(np.zeros((0, 6)))
    _, e, s = f.get()
    for q in range(f.n_qp):
        assert np.allclose(s[q], hooke_raw(c, e[q]), rtol=1e-12, atol=1e-6)   # FE_problem.h:1700 new_stress += stiff * newton_strain


def _setup_sync(tmp_path, c):
    from scema_amd import stmd
    nin = str(tmp_path / "nanoscale_input")
    stmd.write_nanoscale_input(nin, "g0", 1, init_length=[44.4, 44.37, 40.54], init_stress_raw=np.zeros(6), stiff_file_order=c)
    s = stmd.STMDSync(None)
    s.init(nanostatelocin=nin, mdtype=("g0",), nrepl=1, approx_md_with_hookes_law=True, macrostatelocout=str(tmp_path))
    return s


@pytest.mark.parametrize("nsteps", [1, 10])   # BASELINE config 1 (1 timestep) and config 3 (10 continuum steps)
def test_continuum_steps_over_the_cuboid_mesh_with_the_hooke_backend(tmp_path, nsteps):
    """3x3x8 cells, 576 quadrature points: every step hands the update_list to STMDSync::update and takes the stresses back.
    In the Hooke test mode the returned stress is an increment (FE_problem.h:1687-1692): after any number of steps every
    point must carry sigma = C : eps_total, and the ids must follow the bookkeeping of FE_problem.h:1091-1103."""
    from scema_amd import capi, fe
    c = iso_stiffness()
    sync = _setup_sync(tmp_path, c)
    f = fe.FE(3, 3, 8, 0.03, 0.03, 0.08, 1000.0, c, dt=2e-8, top_velocity=5.0, min_qp_strain=1e-10, hooke=True)
    n_updates = []
    for step in range(1, nsteps + 1):
        ul = f.solve()
        if step == 1:
            assert all(u[1] == capi.QP_NONE for u in ul)
        else:
            assert all(u[1] == u[0] for u in ul)                     # later calls continue from the point's own state
        stress = sync.update(step, step * 2e-8, 1, ul) if ul else np.zeros((0, 6))
        f.check(stress)
        n_updates.append(len(ul))
    _, e, s = f.get()
    assert n_updates[0] >= 9 * 8 and n_updates[-1] >= n_updates[0]    # the loaded layer first, the stress wave reaches more cells later
    # (a point whose accumulated strain sits between 0 and the 1e-10 threshold for a step is continued linear-elastically
    # AND keeps that strain in upd_strain, so the Hooke test mode counts it twice once the point passes the threshold --
    # the reference's own arithmetic, FE_problem.h:1687-1700, reproduced; it is worth 1e-10 of strain, hence the atol)
    smax = np.abs(s).max()
    assert smax > 1e6
    for q in range(f.n_qp):
        assert np.allclose(s[q], hooke_raw(c, e[q]), rtol=1e-9, atol=1e-4 * smax)
    sync.close()


def test_free_vibration_conserves_energy():
    """no MD, no loading: kinetic + strain energy of the explicit scheme stays within O(dt^2) of its start value"""
    from scema_amd import fe
    c = iso_stiffness()
    nx, ny, nz, lx, ly, lz = 2, 2, 4, 0.02, 0.02, 0.04
    dt = 2e-8                                                         # cell 0.01 m, wave speed ~1.1e4 m/s: CFL ~ 9e-7 s
    f = fe.FE(nx, ny, nz, lx, ly, lz, 1000.0, c, dt=dt, min_qp_strain=1.0)
    x = f.node_coords(nx, ny, nz, lx, ly, lz)
    v = np.zeros_like(x); v[:, 2] = 3.0 * np.sin(np.pi * x[:, 2] / (2 * lz))   # bottom face at rest, as the stand-in holds it
    f.set_velocity(v)
    wq = (lx / nx) * (ly / ny) * (lz / nz) / 8.0
    def total():
        _, e, s = f.get()
        w = np.array([1, 1, 1, 2, 2, 2.0])
        return f.kinetic_energy() + 0.5 * wq * float((s * e * w).sum())
    e0 = total()
    es = []
    for _ in range(300):
        assert f.solve() == []
        f.check(np.zeros((0, 6)))
        es.append(total())
    assert np.abs(np.array(es) - e0).max() < 1e-2 * e0        # symplectic Euler: the energy oscillates at O(dt omega), it does not drift


@pytest.mark.gpu
def test_two_continuum_steps_with_md_on_the_examples_mesh(tmp_path, small_pe):
    """The reference's runnable example is a 1x1x2 mesh = 16 quadrature points, 2 steps (examples/streched_polyhedron): the
    same shape with the GPU engine behind STMDSync and small PE replicas; every returned stress against the oracle's
    evaluation of the same request sequence."""
    from scema_amd import capi, fe, stmd
    from oracle import pyoracle as po
    kw = dict(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=1e-5)
    lens = small_pe["box"][3:6] - small_pe["box"][:3]
    nin = str(tmp_path / "nanoscale_input")
    s0 = np.array([1.0e6, -2.0e6, 0.5e6, 1.0e5, 0.0, -3.0e5])
    c = iso_stiffness()
    stmd.write_nanoscale_input(nin, "pe", 1, init_length=lens, init_stress_raw=s0, stiff_file_order=c, sysd=small_pe)
    eng = capi.Engine(capi.default_params(**kw))
    sync = stmd.STMDSync(eng)
    sync.init(nanostatelocin=nin, mdtype=("pe",), nrepl=1, md_nsteps_sample=10, macrostatelocout=str(tmp_path), nanostatelocout=str(tmp_path))
    f = fe.FE(1, 1, 2, 0.01, 0.01, 0.02, 1000.0, c, dt=1e-7, top_velocity=150.0, min_qp_strain=1e-10)
    oracles = {}
    for step in (1, 2):
        ul = f.solve()
        assert len(ul) > 0
        got = sync.update(step, step * 1e-7, 1, ul)
        for k, (qid, recent, mat, eps) in enumerate(ul):
            o = oracles.setdefault(qid, po.Oracle(small_pe, po.default_params(**kw)))
            # the id bookkeeping runs every step for every point (FE_problem.h:1091-1103): from the second step on a point names
            # itself, whether or not it has run MD before (then there is no last.<qp> state and the replica starts from init)
            assert recent == (capi.QP_NONE if step == 1 else qid)
            sig, _ = o.eval(po.prepare_strain(eps, np.eye(3), lens, hooke=False), 2.0, 300.0, 1e-4, 10)
            exp = po.store(sig[None], s0[None], np.eye(3)[None], False)
            assert np.abs(got[k] - exp).max() < 1e-6 * np.abs(sig).max()
        f.check(got)
    _, _, s = f.get()
    assert np.isfinite(s).all()
    sync.close(); eng.close()
