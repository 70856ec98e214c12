"""ReaxFF path on the GPU (md_reax.hip) against the oracle (oracle/reax_oracle.c): SURVEY.md 8(f) row f-4, BASELINE config 5.

PARITY UNPINNED at the oracle level (LAMMPS USER-REAXC is not available; see oracle/reax_oracle.h): what is shown here is
that the kernels reproduce the oracle's energies term by term, its equilibrated charges, the central differences of its
energy (forces) and its strain derivative (virial), on isolated molecules, condensed cells in both neighbour-search modes,
and on configurations reached by the engine's own dynamics (thermostat, fix deform, list rebuilds, wrapping)."""
import os

import numpy as np
import pytest

from oracle import pyreax as pr
from scema_amd import capi
from test_oracle_reax import FFIELD, _glycine_like
from test_reax_host import BIGBOX, _check_dirs, _mixture, _pe_cell, _sym

pytestmark = pytest.mark.gpu

FTM2V = 1.0 / 48.88821291 / 48.88821291
NKTV2P = 68568.415
MVV2E = 48.88821291 * 48.88821291


@pytest.fixture(scope="module")
def ff():
    f = pr.ForceField(FFIELD)
    yield f
    f.close()


@pytest.fixture()
def eng():
    e = capi.Engine()
    e.reax_configure(FFIELD, qeq_tol=1e-10)
    e.reax_set(exact_gradient=1)
    yield e
    e.close()


def _oracle_virial(ff, t, x, box, q):
    w = np.zeros(6)
    pr.lib().rxo_forces_fd(ff.h, len(t), pr._p(np.ascontiguousarray(t, dtype=np.int32)), pr._p(np.ascontiguousarray(x)), pr._p(np.ascontiguousarray(box)),
                           pr._p(np.ascontiguousarray(q)), 1e-5, None, pr._p(w))
    return w


def _compare_static(ff, eng, sym, x, box, name="m", full_fd=False, seed=1):
    t = ff.types(sym)
    eng.register_replica(name, 1, capi.reax_system(sym, x, box))
    r = eng.reax_compute(name, 1)
    q, _ = ff.qeq(t, x, box=box, tol=1e-10, maxiter=500)
    assert np.abs(r["q"] - q).max() < 1e-7 and abs(r["q"].sum()) < 1e-8
    # energies with the engine's own charges (their difference from the oracle's is at the solver tolerance)
    _, po = ff.energy(t, x, box=box, q=r["q"])
    for k in pr.PARTS:
        assert abs(r["e"][k] - po[k]) < 1e-9 * max(1.0, abs(po[k])), (k, r["e"][k], po[k])
    f = r["f"]
    assert np.abs(f.sum(0)).max() < 1e-6 * max(1.0, np.abs(f).max())
    if full_fd:
        fo = ff.forces(t, x, box=box, q=r["q"], h=1e-5)
        assert np.abs(f - fo).max() < 2e-6 * max(1.0, np.abs(fo).max())
    else:
        good, worst = _check_dirs(ff, t, x, box, r["q"], f, seed=seed)
        assert good >= 5, worst
    w = _oracle_virial(ff, t, x, box, r["q"])
    assert np.abs(r["w"] - w).max() < 2e-6 * np.abs(w).max(), (r["w"], w)
    return r, po


def test_isolated_molecule_all_terms(ff, eng):
    t, x = _glycine_like(ff)
    r, po = _compare_static(ff, eng, _sym(ff, t), x + 0.0, BIGBOX, full_fd=True)
    assert r["image_search"] == 0
    assert all(abs(po[k]) > 1e-6 for k in ("bond", "lp", "over", "under", "angle", "tors", "conj", "hb", "vdw", "coul", "pol"))


def test_atoms_without_any_neighbour_inside_the_taper_radius(ff, eng):
    """gas-phase case: a row of the charge matrix with no entry at all, walked in the same wave pass as a full one --
    the masked lanes must not read a column index out of the unwritten row.  Two lone hydrogens 40 A from a molecule, one of them
    between the molecule's atoms in the file so that its row is paired with a non-empty one."""
    t, x = _glycine_like(ff)
    sym = _sym(ff, t)
    sym = sym[:1] + ["H"] + sym[1:] + ["H"]
    x = np.vstack([x[:1], [[24.0, 23.0, 25.0]], x[1:], [[-24.0, 22.0, -23.0]]])
    box = np.array([-30.0, -30.0, -30.0, 30.0, 30.0, 30.0, 0.0, 0.0, 0.0])
    # the scratch memory the rows live in is not zeroed by the allocator: poison it first with a denser system of the same name
    from scema_amd.systems import build_pe
    d = build_pe(2, 3, 5)
    eng.register_replica("poison", 1, capi.reax_system(["C" if d["mass"][k] > 5 else "H" for k in d["type"]], d["x"], d["box"]))
    eng.reax_compute("poison", 1)
    r, po = _compare_static(ff, eng, sym, x, box, name="lone", full_fd=True)
    assert np.all(np.isfinite(r["q"])) and np.all(np.isfinite(r["f"]))
    # a lone atom keeps the charge the constraint leaves it and feels nothing
    assert np.abs(r["f"][1]).max() < 1e-9 and np.abs(r["f"][-1]).max() < 1e-9


def test_each_term_group_alone(ff, eng):
    """the parity switch `terms` isolates a launch: its forces are the central differences of just those energy parts"""
    t, x = _glycine_like(ff)
    sym = _sym(ff, t)
    eng.register_replica("m", 1, capi.reax_system(sym, x, BIGBOX))
    q = eng.reax_compute("m", 1)["q"]
    groups = {1: ("bond", "lp", "over", "under"), 2: ("angle", "pen", "coa"), 4: ("tors", "conj"), 8: ("hb",), 16: ("vdw", "coul", "pol")}
    h = 1e-5
    for mask, parts in groups.items():
        eng.reax_set(terms=mask)
        f = eng.reax_compute("m", 1)["f"]
        fd = np.zeros_like(x)
        for i in range(len(x)):
            for c in range(3):
                xp, xm = x.copy(), x.copy()
                xp[i, c] += h
                xm[i, c] -= h
                pp, pm = ff.energy(t, xp, box=BIGBOX, q=q)[1], ff.energy(t, xm, box=BIGBOX, q=q)[1]
                fd[i, c] = -sum(pp[k] - pm[k] for k in parts) / (2 * h)
        assert np.abs(f - fd).max() < 2e-6 * max(1.0, np.abs(fd).max()), mask
    eng.reax_set(terms=31)


def test_condensed_cell_minimum_image(ff, eng):
    from scema_amd.systems import build_pe
    d = build_pe(3, 5, 9)                                  # 1620 atoms, 22.2 x 24.65 x 22.8 A: every width >= 2 (10 + skin)
    x = d["x"] + 0.08 * np.random.default_rng(7).standard_normal(d["x"].shape)
    sym = ["C" if d["mass"][k] > 5 else "H" for k in d["type"]]
    eng.reax_configure(FFIELD, qeq_tol=1e-10, skin=1.0)
    eng.reax_set(exact_gradient=1)
    r, po = _compare_static(ff, eng, sym, x, d["box"])
    assert r["image_search"] == 0 and r["maxneigh_seen"] <= r["maxnb"]
    assert abs(po["tors"]) > 100 and abs(po["over"]) > 100


def test_condensed_triclinic_cell_with_images(ff, eng):
    sym, x, box = _pe_cell(ff, tilt=(3.0, -2.0, 1.5), amp=0.12)      # 20.27 A thick: several images inside the list radius
    r, _ = _compare_static(ff, eng, sym, x, box, seed=2)
    assert r["image_search"] > 0


def test_mixture_with_hydrogen_bonds(ff, eng):
    sym, x, box = _mixture()
    r, po = _compare_static(ff, eng, sym, x, box, seed=3)
    assert po["hb"] < -1.0


def test_lammps_gradient_switch(ff, eng):
    """the switch that leaves out one d(SBO)/d(Delta) term of the valence-angle energy (off by default); energies are untouched"""
    sym, x, box = _pe_cell(ff, amp=0.1)
    eng.register_replica("m", 1, capi.reax_system(sym, x, box))
    a = eng.reax_compute("m", 1)
    eng.reax_set(exact_gradient=0)
    b = eng.reax_compute("m", 1)
    assert a["e"] == b["e"] or all(abs(a["e"][k] - b["e"][k]) < 1e-9 * max(1.0, abs(a["e"][k])) for k in a["e"])
    dev = np.abs(a["f"] - b["f"])
    assert 0.0 < dev.max() < 0.1 * np.abs(a["f"]).max() and np.median(dev) < 0.01 * np.median(np.abs(a["f"]))


def _kinetic(mass_atom, v):
    return 0.5 * MVV2E * (mass_atom[:, None] * v * v).sum()


def test_charges_do_not_depend_on_how_the_solver_is_launched(ff, monkeypatch):
    """the conjugate-gradient iterations run as launches over the batch, as many as the host issues; the rest, if any, in one
    workgroup per replica.  Same recurrences either way: no launches at all, too few, and plenty give the same charges."""
    sym, x, box = _pe_cell(ff, amp=0.1)
    res = []
    for launches in ("0", "5", "200"):
        monkeypatch.setenv("SCEMA_REAX_QEQ_LAUNCH", launches)
        e = capi.Engine()
        e.reax_configure(FFIELD, qeq_tol=1e-10)
        e.register_replica("m", 1, capi.reax_system(sym, x, box))
        r = e.reax_compute("m", 1)
        st = e.reax_stats()
        res.append((r["q"].copy(), r["qeq_iters"], st["qeq_slow_solves"]))
        e.close()
    (q0, it0, slow0), (q1, it1, slow1), (q2, it2, slow2) = res
    assert slow0 == 1 and slow1 == 1 and slow2 == 0
    # the scalar products are summed in another order in the two forms: a solve may stop one iteration apart at the tolerance
    assert max(it0, it1, it2) - min(it0, it1, it2) <= 2 and it0 > 5
    assert np.abs(q1 - q0).max() < 1e-9 and np.abs(q2 - q0).max() < 1e-9
    assert np.abs(q0).max() > 0.05


def test_the_preconditioner_changes_the_iteration_count_not_the_charges(ff, monkeypatch):
    """fix qeq/reax's tolerance fixes the charges; the preconditioner only decides how many iterations it takes to get there.  The bonded-pattern
    approximate inverse (default) and the reference's Jacobi preconditioner stop on the same measure, sqrt(r.D^-1 r) / |b|: same charges to the
    tolerance's order, in well under half the iterations (gated offline first: profiles/r05_qeq_precond_gate.txt)."""
    from scema_amd.systems import build_pe
    d = build_pe(3, 5, 9, jitter=0.08, seed=5)      # PE-1620, the replica of BASELINE config 5: wide enough for one image per neighbour,
    sym = ["C" if d["mass"][t] > 5 else "H" for t in d["type"]]   # which the bonded pattern needs (smaller boxes keep the Jacobi preconditioner)
    x, box = d["x"], d["box"]
    res = {}
    for name, env in (("sai", None), ("jacobi", "0")):
        if env is None:
            monkeypatch.delenv("SCEMA_REAX_QEQ_PRECOND", raising=False)
        else:
            monkeypatch.setenv("SCEMA_REAX_QEQ_PRECOND", env)
        for tol in (1e-6, 1e-10):
            e = capi.Engine()
            e.reax_configure(FFIELD, qeq_tol=tol)
            e.register_replica("m", 1, capi.reax_system(sym, x, box))
            r = e.reax_compute("m", 1)
            res[(name, tol)] = (r["q"].copy(), r["qeq_iters"], r["f"].copy())
            e.close()
    q_ref = res[("jacobi", 1e-10)][0]
    assert np.abs(res[("sai", 1e-10)][0] - q_ref).max() < 2e-8                      # the same linear systems (measured: 3.5e-9)
    for name in ("sai", "jacobi"):
        assert np.abs(res[(name, 1e-6)][0] - q_ref).max() < 5e-5, name              # the reference's tolerance: the same order of error either way
    assert res[("sai", 1e-6)][1] < 0.6 * res[("jacobi", 1e-6)][1] and res[("sai", 1e-10)][1] < 0.6 * res[("jacobi", 1e-10)][1]
    f_ref = res[("jacobi", 1e-10)][2]
    assert np.abs(res[("sai", 1e-6)][2] - f_ref).max() < 1e-4 * np.abs(f_ref).max()


def test_verlet_step_and_energy_conservation(ff, eng):
    sym, x, box = _pe_cell(ff, amp=0.02)
    n = len(sym)
    rng = np.random.default_rng(0)
    masses = dict(H=1.008, C=12.011, N=14.007, O=15.999)
    m = np.array([masses[s] for s in sym])
    v = rng.standard_normal((n, 3)) * np.sqrt(0.0019872067 * 300.0 / (m[:, None] * MVV2E))
    v -= (m[:, None] * v).sum(0) / m.sum()
    eng.register_replica("m", 1, capi.reax_system(sym, x, box, v=v))
    eng.set_state(0, "m", 1, box, x, v)
    r0 = eng.reax_compute("m", 1, qp=0)
    ke = _kinetic(m, v)
    e0 = sum(r0["e"].values()) + ke
    # one velocity-Verlet step by hand from the oracle-checked forces
    dt = 0.1
    eng.debug_run("m", 1, 1, dt, 300.0, qp=0, nvt=False, use_shake=False)
    _, x1, v1 = eng.get_state(0, "m", 1)
    vh = v + 0.5 * dt * FTM2V * r0["f"] / m[:, None]
    xe = x + dt * vh
    # the engine wraps into the box when it builds its rows: compare modulo lattice vectors
    d = x1 - xe
    lens = box[3:6] - box[:3]
    d -= np.round(d / lens) * lens
    assert np.abs(d).max() < 1e-10
    r1 = eng.reax_compute("m", 1, qp=0)
    assert np.abs(v1 - (vh + 0.5 * dt * FTM2V * r1["f"] / m[:, None])).max() < 1e-10
    # energy conservation over 60 fs (measured: -0.05 kcal/mol of 1285 kinetic at this step, -0.58 at 0.2 fs, +0.07 at 0.05 fs;
    # ReaxFF's energy has small jumps at its bond-order cutoffs, so a hot disordered mixture wanders by +-7 whatever the step)
    eng.debug_run("m", 1, 599, dt, 300.0, qp=0, nvt=False, use_shake=False)
    _, x2, v2 = eng.get_state(0, "m", 1)
    r2 = eng.reax_compute("m", 1, qp=0)
    e2 = sum(r2["e"].values()) + _kinetic(m, v2)
    assert abs(e2 - e0) < 1e-3 * ke, (e0, e2, ke)
    assert abs(_kinetic(m, v2) - ke) > 0.05 * ke          # energy did flow between kinetic and potential
    # and the configuration reached by the dynamics is still one the oracle agrees on
    t = ff.types(sym)
    _, po = ff.energy(t, x2, box=box, q=r2["q"])
    for k in pr.PARTS:
        assert abs(r2["e"][k] - po[k]) < 1e-9 * max(1.0, abs(po[k])), k
    good, worst = _check_dirs(ff, t, x2, box, r2["q"], r2["f"], seed=5)
    assert good >= 5, worst
    # with the d(SBO)/d(Delta) term left out the same run gains about 100 kcal/mol: why that switch is off by default
    eng.set_state(1, "m", 1, box, x, v)
    eng.reax_set(exact_gradient=0)
    eng.debug_run("m", 1, 600, dt, 300.0, qp=1, nvt=False, use_shake=False)
    _, _, v3 = eng.get_state(1, "m", 1)
    r3 = eng.reax_compute("m", 1, qp=1)
    assert abs(sum(r3["e"].values()) + _kinetic(m, v3) - e0) > 0.02 * ke
    eng.reax_set(exact_gradient=1)


def test_sampled_pressure_against_the_oracle(ff, eng):
    """compute pressure + fix ave/time of the sampling run: with a time step so short that nothing moves, the average is the
    static pressure tensor, (sum m v v + W)/V with W the oracle's strain derivative"""
    sym, x, box = _mixture(seed=9)
    n = len(sym)
    rng = np.random.default_rng(1)
    masses = dict(H=1.008, C=12.011, N=14.007, O=15.999)
    m = np.array([masses[s] for s in sym])
    v = rng.standard_normal((n, 3)) * np.sqrt(0.0019872067 * 300.0 / (m[:, None] * MVV2E))
    v -= (m[:, None] * v).sum(0) / m.sum()
    eng.register_replica("m", 1, capi.reax_system(sym, x, box, v=v))
    eng.set_state(0, "m", 1, box, x, v)
    pavg = eng.debug_run("m", 1, 10, 1e-6, 300.0, qp=0, nvt=False, use_shake=False, sample=True)
    t = ff.types(sym)
    q, _ = ff.qeq(t, x, box=box, tol=1e-10, maxiter=500)
    w = _oracle_virial(ff, t, x, box, q)
    ke = MVV2E * np.array([(m * v[:, a] * v[:, b]).sum() for a, b in ((0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2))])
    vol = np.prod(box[3:6] - box[:3])
    ref = (ke + w) / vol * NKTV2P
    assert np.abs(pavg - ref).max() < 1e-5 * np.abs(ref).max(), (pavg, ref)


def test_strain_batch_with_reax_force_field(ff, tmp_path):
    """the hot path with md_force_field "reax": configured from scripts_folder/ffield.reax.2 and H C N O as the reference's
    script does; strained, thermostatted, sampled; the stress is -<P> 101325 of the sampling run; the state it leaves
    behind is one the oracle agrees on; a second update continues from it"""
    import shutil
    scripts = tmp_path / "lammps_scripts_reax"
    scripts.mkdir()
    shutil.copy(FFIELD, scripts / "ffield.reax.2")
    sym, x, box = _mixture(seed=10)
    n = len(sym)
    masses = dict(H=1.008, C=12.011, N=14.007, O=15.999)
    m = np.array([masses[s] for s in sym])
    v = np.random.default_rng(2).standard_normal((n, 3)) * np.sqrt(0.0019872067 * 300.0 / (m[:, None] * MVV2E))
    v -= (m[:, None] * v).sum(0) / m.sum()
    e = capi.Engine()
    e.register_replica("g0", 1, capi.reax_system(sym, x, box, v=v))
    lens = box[3:6] - box[:3]
    strains = [np.array([0.004, -0.001, 0.0, 0.002, 0.0, -0.001]) * lens[[0, 1, 2, 2, 1, 0]], np.array([-0.002, 0.003, 0.001, 0.0, 0.002, 0.0]) * lens[[0, 1, 2, 2, 1, 0]]]
    def sims(recent):
        return [capi.make_sim(k, "g0", 1, strains[k], nss=20, dt=0.25, temperature=300.0, strain_rate=1e-3, most_recent=recent[k], force_field="reax",
                              scripts_folder=str(scripts)) for k in range(2)]
    out = e.strain_batch(sims([capi.QP_NONE, capi.QP_NONE]))
    s1 = np.array([list(o.stress) for o in out])
    assert all(o.stress_updated for o in out) and np.isfinite(s1).all() and np.abs(s1).max() > 1e5
    assert np.abs(s1[0] - s1[1]).max() > 1e3            # different strains, different stresses
    st = e.reax_stats()
    assert st["qeq_solves"] > 0 and st["qeq_tol"] == 1e-6 and 2 <= st["qeq_iters"] / st["qeq_solves"] < 60
    # the states it left: boxes strained as asked, configurations the oracle agrees on
    e.reax_set(exact_gradient=1)
    t = ff.types(sym)
    for k in range(2):
        b, xs, vs = e.get_state(k, "g0", 1)
        eps = strains[k] / lens[[0, 1, 2, 2, 1, 0]]
        assert np.allclose((b[3:6] - b[:3]) / lens - 1.0, eps[:3], atol=2e-5)
        r = e.reax_compute("g0", 1, qp=k)
        _, po = ff.energy(t, xs, box=b, q=r["q"])
        for name in pr.PARTS:
            assert abs(r["e"][name] - po[name]) < 1e-9 * max(1.0, abs(po[name])), name
        good, worst = _check_dirs(ff, t, xs, b, r["q"], r["f"], seed=6 + k)
        assert good >= 5, worst
    # determinism of the whole update up to the order of atomic sums, and continuation from the stored states
    e2 = capi.Engine()
    e2.register_replica("g0", 1, capi.reax_system(sym, x, box, v=v))
    s1b = np.array([list(o.stress) for o in e2.strain_batch(sims([capi.QP_NONE, capi.QP_NONE]))])
    assert np.abs(s1b - s1).max() < 1e-6 * np.abs(s1).max()
    out2 = e.strain_batch(sims([0, 1]))
    s2 = np.array([list(o.stress) for o in out2])
    assert np.isfinite(s2).all() and np.abs(s2 - s1).max() > 1e3
    # a batch may not mix force fields
    bad = sims([0, 1])
    bad[1] = capi.make_sim(1, "g0", 1, strains[1], nss=20, dt=0.25, temperature=300.0, strain_rate=1e-3, most_recent=1, force_field="opls")
    with pytest.raises(capi.EngineError, match="mixes force fields"):
        e.strain_batch(bad)
    e.close()
    e2.close()


def test_fallback_kernels_of_large_replicas_equal_the_default_ones():
    """k_rx_nonbonded_once (every pair once, partner forces in an LDS table: the default for replicas of up to 6 000 atoms) against
    k_rx_nonbonded (both ends of every pair, no table: larger replicas; SCEMA_MD_RX_NB_ONCE=0 forces it, read once per process ->
    child processes) on the triclinic cell with several images inside the list radius: same forces, energies and virial"""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import json, sys, numpy as np\n"
            "sys.path.insert(0, 'tests')\n"
            "from oracle import pyreax as pr\n"
            "from scema_amd import capi\n"
            "from test_oracle_reax import FFIELD\n"
            "from test_reax_host import _pe_cell\n"
            "ff = pr.ForceField(FFIELD)\n"
            "sym, x, box = _pe_cell(ff, tilt=(3.0, -2.0, 1.5), amp=0.12)\n"
            "e = capi.Engine()\n"
            "e.reax_configure(FFIELD, qeq_tol=1e-10)\n"
            "e.register_replica('m', 1, capi.reax_system(sym, x, box))\n"
            "r = e.reax_compute('m', 1)\n"
            "print(json.dumps(dict({'f': np.asarray(r['f']).ravel().tolist(), 'w': np.asarray(r['w']).ravel().tolist()}, **{k: float(v) for k, v in r['e'].items()})))\n")
    out = {}
    for name, env in (("once", {}), ("both_ends", {"SCEMA_MD_RX_NB_ONCE": "0"}), ("col32", {"SCEMA_MD_RX_COL32": "1"}), ("near_full", {"SCEMA_MD_RX_NEAR_FULL": "1"}), ("item_atomics", {"SCEMA_MD_RX_ITEM_LDS": "0"}), ("full_rows", {"SCEMA_REAX_QEQ_SYM": "0"}),
                      ("items_in_place", {"SCEMA_MD_RX_ITEMCAP": "0"}), ("items_mixed", {"SCEMA_MD_RX_ITEMCAP": "100"})):
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr[-2000:]
        out[name] = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    a = out["once"]
    fa = np.array(a["f"])
    # (the third run stores the columns of the charge-equilibration matrix as 32-bit indices beside full FP64 values, as replicas beyond 65 536
    # atoms do: the default packs a value's upper 48 bits and a 16-bit column into one word -- this comparison bounds what that rounding does)
    # (the last two give the angle and torsion kernels an item list of 0 / 100 entries per 256 atoms, so that their work items are
    # done where they are found -- the path of a system denser than any tested -- entirely / for the items beyond the list)
    # (near_full: the near rows with every pair inside 5 A + skin, as until round 5; the default keeps a type pair's candidates within the reach of
    # its bond order -- the bond rows, and everything after them, must not notice)
    # (item_atomics: the angle and torsion items add their forces and dE/dDelta with device-wide atomics, as replicas too large for the LDS tables do)
    # (full_rows: the matrix of the charge equilibration with every pair in both rows and the two-launch iteration that goes with it; the default
    # stores a pair once and makes both of its products in one sweep)
    for name in ("both_ends", "col32", "near_full", "item_atomics", "full_rows", "items_in_place", "items_mixed"):
        b = out[name]
        assert np.abs(fa - np.array(b["f"])).max() < 1e-10 * np.abs(fa).max(), name
        assert np.abs(np.array(a["w"]) - np.array(b["w"])).max() < 1e-10 * np.abs(np.array(a["w"])).max(), name
        for k in ("vdw", "coul", "pol", "angle", "pen", "coa", "tors", "conj"):
            assert abs(a[k] - b[k]) < 1e-10 * max(1.0, abs(a[k])), (name, k)


def test_symmetric_solve_and_lds_tables_on_minimum_image_replicas():
    """The forms that need one image per neighbour, on the replicas that have it -- PE-1620 (BASELINE config 5) and PE-2880, whose tables (two vectors of
    the replica for the symmetric sweep: 92 kB; forces + dE/dDelta of the angle / torsion items: 92 kB; partner forces of the non-bonded pass: 69 kB) are
    past the 64 kB a launch gets without the opt-in: the symmetric form of the charge solve (each pair stored once, one workgroup per replica between
    two sweeps) against full rows, and the item sums through LDS against device-wide atomics.  Same systems solved to 1e-10: same charges and forces.
    (The switches are read once per process: child processes.)"""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import json, sys, numpy as np\n"
            "from scema_amd import capi\n"
            "from scema_amd.systems import build_pe\n"
            "cells = tuple(int(c) for c in sys.argv[1:4])\n"
            "d = build_pe(*cells, jitter=0.08, seed=5)\n"
            "sym = ['C' if d['mass'][t] > 5 else 'H' for t in d['type']]\n"
            "e = capi.Engine()\n"
            "e.reax_configure(sys.argv[4], qeq_tol=1e-10)\n"
            "e.register_replica('m', 1, capi.reax_system(sym, d['x'], d['box']))\n"
            "r = e.reax_compute('m', 1)\n"
            "print(json.dumps({'f': np.asarray(r['f']).ravel().tolist(), 'q': np.asarray(r['q']).ravel().tolist(), 'w': np.asarray(r['w']).ravel().tolist(), 'iters': int(r['qeq_iters']), 'n': len(sym)}))\n")
    for cells in ((3, 5, 9), (4, 6, 10)):
        out = {}
        for name, env in (("default", {}), ("full_rows", {"SCEMA_REAX_QEQ_SYM": "0"}), ("item_atomics", {"SCEMA_MD_RX_ITEM_LDS": "0"})):
            p = subprocess.run([sys.executable, "-c", code] + [str(c) for c in cells] + [FFIELD], capture_output=True, text=True, timeout=600, cwd=root,
                               env=dict(os.environ, **env))
            assert p.returncode == 0, p.stderr[-2000:]
            out[name] = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
        a = out["default"]
        assert a["n"] == 12 * cells[0] * cells[1] * cells[2]
        fa, qa, wa = np.array(a["f"]), np.array(a["q"]), np.array(a["w"])
        for name in ("full_rows", "item_atomics"):
            b = out[name]
            assert np.abs(qa - np.array(b["q"])).max() < 2e-8, (cells, name)                       # (two solves of one system to 1e-10)
            assert np.abs(fa - np.array(b["f"])).max() < 1e-7 * np.abs(fa).max(), (cells, name)
            assert np.abs(wa - np.array(b["w"])).max() < 1e-7 * np.abs(wa).max(), (cells, name)
        assert abs(a["iters"] - out["full_rows"]["iters"]) <= 2, (a["iters"], out["full_rows"]["iters"])   # the same recurrences


def test_undersized_neighbour_rows_overflow_and_regrow(monkeypatch):
    """SCEMA_MD_NEIGH_GROW0 starts the engine with row capacities at 0.4 of their estimate: the wave-per-row list build must flag full rows (full and
    near rows, bond rows) instead of writing past them, and the engine regrows and repeats -- same forces as the run that never overflowed"""
    from scema_amd.systems import build_pe
    d = build_pe(3, 5, 9, jitter=0.08, seed=5)
    sym = ["C" if d["mass"][t] > 5 else "H" for t in d["type"]]
    res = []
    for grow0 in (None, "0.4"):
        if grow0 is None:
            monkeypatch.delenv("SCEMA_MD_NEIGH_GROW0", raising=False)
        else:
            monkeypatch.setenv("SCEMA_MD_NEIGH_GROW0", grow0)
        e = capi.Engine()
        e.reax_configure(FFIELD, qeq_tol=1e-10)
        e.register_replica("m", 1, capi.reax_system(sym, d["x"], d["box"]))
        r = e.reax_compute("m", 1)
        res.append((np.asarray(r["f"]).copy(), r["maxnb"], r["maxneigh_seen"]))
        e.close()
    f0, f1 = res[0][0], res[1][0]
    assert np.abs(f0 - f1).max() < 1e-8 * np.abs(f0).max()
    assert res[1][2] == res[0][2] and res[1][1] >= res[1][2]   # the same longest row, inside the regrown capacity
