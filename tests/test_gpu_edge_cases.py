"""GPU edge cases of the hot path: heterogeneous batches, uncharged systems (no k-space), tiny systems,
sampling-window rules, and the error paths the reference handles with exit(1)/assert."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
KW = dict(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=1e-5)


def relerr(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return np.abs(a - b).max() / max(1e-300, np.abs(b).max())


def _lens(d):
    return d["box"][3:6] - d["box"][:3]


def test_heterogeneous_batch_two_materials(small_pe):
    """One update() with replicas of different size and topology: each equals its own single run and the oracle."""
    from scema_amd import capi
    from scema_amd.systems import build_pe
    from oracle import pyoracle as po
    big = build_pe(3, 4, 6, jitter=0.04, seed=21)            # 864 atoms, different box
    eng = capi.Engine(capi.default_params(**KW))
    eng.register_replica("a", 1, small_pe)
    eng.register_replica("b", 1, big)
    sa = np.array([-3e-4, -3e-4, 1.0e-3, 0, 0, 0]) * np.array([*_lens(small_pe), _lens(small_pe)[2], _lens(small_pe)[1], _lens(small_pe)[0]])
    sb = np.array([4e-4, -2e-4, -1.5e-3, 0, 0, 0]) * np.array([*_lens(big), _lens(big)[2], _lens(big)[1], _lens(big)[0]])
    sims = [capi.make_sim(0, "a", 1, sa, nss=10, most_recent=capi.QP_NONE, material=0),
            capi.make_sim(1, "b", 1, sb, nss=10, most_recent=capi.QP_NONE, material=1),
            capi.make_sim(2, "a", 1, 2 * sa, nss=10, most_recent=capi.QP_NONE, material=0)]
    out = eng.strain_batch(sims)
    got = [np.array(o.stress[:]) for o in out]
    for d, st, g in ((small_pe, sa, got[0]), (big, sb, got[1]), (small_pe, 2 * sa, got[2])):
        exp, _ = po.Oracle(d, po.default_params(**KW)).eval(st, 2.0, 300.0, 1e-4, 10)
        assert relerr(g, exp) < 1e-6
    eng.close()


def _fcc(n_cell=4, a=5.3, charges=False):
    basis = [(0, 0, 0), (.5, .5, 0), (.5, 0, .5), (0, .5, .5)]
    x = np.array([[(i + b[0]) * a, (j + b[1]) * a, (k + b[2]) * a] for i in range(n_cell) for j in range(n_cell) for k in range(n_cell) for b in basis])
    rng = np.random.default_rng(3)
    x += rng.normal(0, 0.05, x.shape)
    n = len(x); L = n_cell * a
    z = lambda *s: np.zeros(s, np.int32)
    m = 39.95
    v = rng.normal(0, 1, (n, 3)) * np.sqrt(0.0019872067 * 80.0 / (m * 48.88821291 ** 2)); v -= v.mean(0)
    return dict(natoms=n, ntypes=1, type=z(n), charge=np.zeros(n), mass=np.array([m]), eps=np.array([[0.238]]), sigma=np.array([[3.405]]),
                bonds=z(0, 2), bond_type=z(0), bond_coeff=np.zeros((0, 2)), angles=z(0, 3), angle_type=z(0), angle_coeff=np.zeros((0, 2)),
                dihedrals=z(0, 4), dihedral_type=z(0), dihedral_coeff=np.zeros((0, 4)), impropers=z(0, 4), improper_type=z(0),
                improper_coeff=np.zeros((0, 2)), special_lj=np.ones(3), special_coul=np.ones(3),
                box=np.array([0, 0, 0, L, L, L, 0, 0, 0.0]), x=x, v=v)


def test_uncharged_atomic_system_no_kspace_no_shake():
    """LJ argon: q = 0 everywhere (g_ewald = 0, no k-vectors), no bonds, no SHAKE clusters."""
    from scema_amd import capi
    from oracle import pyoracle as po
    d = _fcc()
    kw = dict(cut_lj=8.5, cut_coul=8.5, skin=2.0)
    eng = capi.Engine(capi.default_params(**kw))
    eng.register_replica("ar", 1, d)
    f, e, w, info = eng.debug_compute("ar", 1)
    o = po.Oracle(d, po.default_params(**kw)); o.setup(False)
    fo, eo, wo = o.compute()
    assert info["nk"] == 0 and info["nclus"] == 0 and info["npairs"] == o.npairs
    assert relerr(f, fo) < 1e-11 and abs(e[0] - eo[0]) < 1e-10 * abs(eo[0]) and relerr(w[0], wo[0]) < 1e-10
    L = d["box"][3]
    st = np.array([2e-4 * L, 2e-4 * L, -5e-4 * L, 1e-4 * L, 0, 0])
    got = np.array(eng.strain_batch([capi.make_sim(0, "ar", 1, st, nss=20, temperature=80.0, most_recent=capi.QP_NONE)])[0].stress[:])
    exp, _ = o.eval(st, 2.0, 80.0, 1e-4, 20)
    assert relerr(got, exp) < 1e-6
    eng.close()


def test_lone_atom_has_an_empty_row_and_feels_nothing():
    """A cluster WITHOUT a listed neighbour (an atom 43 A from the others, alone in its cell): its row is empty, and k_pair must read
    nothing of the memory the row sits in -- which, in a reused slot, holds a crystal's entries (round 4: the prefetch reads whole chunks
    without lane masks; an empty row's one turn of the stream is guarded on the scalar unit)."""
    from scema_amd import capi
    from scema_amd.systems import build_pe
    from oracle import pyoracle as po
    z = lambda *s: np.zeros(s, np.int32)
    d = dict(natoms=3, ntypes=1, type=z(3), charge=np.array([0.3, -0.3, 0.0]), mass=np.array([12.0]), eps=np.array([[0.1]]), sigma=np.array([[3.4]]),
             bonds=z(0, 2), bond_type=z(0), bond_coeff=np.zeros((0, 2)), angles=z(0, 3), angle_type=z(0), angle_coeff=np.zeros((0, 2)),
             dihedrals=z(0, 4), dihedral_type=z(0), dihedral_coeff=np.zeros((0, 4)), impropers=z(0, 4), improper_type=z(0),
             improper_coeff=np.zeros((0, 2)), special_lj=np.ones(3), special_coul=np.ones(3),
             box=np.array([0, 0, 0, 60, 60, 60, 0, 0, 0.0]), x=np.array([[30.0, 30, 30], [34.2, 30, 30], [5.0, 5.0, 5.0]]), v=np.zeros((3, 3)))
    eng = capi.Engine(capi.default_params(shake_mass=0.0, kspace_style=0))
    eng.register_replica("dense", 1, build_pe(4, 6, 12))          # fills the engine's work arrays with a crystal's rows first
    eng.debug_compute("dense", 1)
    eng.register_replica("lone", 1, d)
    f, e, w, info = eng.debug_compute("lone", 1)
    o = po.Oracle(d, po.default_params(shake_mass=0.0, kspace_pppm=0)); o.setup(False)
    fo, eo, wo = o.compute()
    assert np.abs(f - fo).max() < 1e-11 * np.abs(fo).max() and info["npairs"] == o.npairs
    assert np.abs(f[2]).max() < 1e-12 * np.abs(f[0]).max()                # the lone, uncharged atom: no pair force, no reciprocal force
    assert np.abs(f[0] + f[1]).max() < 1e-9 * np.abs(f[0]).max()
    eng.close()


def test_two_atoms_closed_form():
    """Smallest possible system: two LJ+coulomb atoms in a 60 A box; force equals the analytic pair force."""
    from scema_amd import capi
    r = 4.2
    z = lambda *s: np.zeros(s, np.int32)
    d = dict(natoms=2, ntypes=1, type=z(2), charge=np.array([0.3, -0.3]), mass=np.array([12.0]), eps=np.array([[0.1]]), sigma=np.array([[3.4]]),
             bonds=z(0, 2), bond_type=z(0), bond_coeff=np.zeros((0, 2)), angles=z(0, 3), angle_type=z(0), angle_coeff=np.zeros((0, 2)),
             dihedrals=z(0, 4), dihedral_type=z(0), dihedral_coeff=np.zeros((0, 4)), impropers=z(0, 4), improper_type=z(0),
             improper_coeff=np.zeros((0, 2)), special_lj=np.ones(3), special_coul=np.ones(3),
             box=np.array([0, 0, 0, 60, 60, 60, 0, 0, 0.0]), x=np.array([[30.0, 30, 30], [30 + r, 30, 30]]), v=np.zeros((2, 3)))
    eng = capi.Engine(capi.default_params(shake_mass=0.0))
    # the work arrays of the engine are reused: a dense replica first, so that the rows of the two-atom system (one of its two
    # clusters lists the pair, the other lists NOTHING) sit in memory that holds a crystal's entries -- an empty row must read none of it
    from scema_amd.systems import build_pe
    dense = build_pe(4, 6, 12)        # 3 456 atoms, 29.6 A wide: the default cutoffs fit
    eng.register_replica("dense", 1, dense)
    eng.debug_compute("dense", 1)
    eng.register_replica("two", 1, d)
    f, e, w, info = eng.debug_compute("two", 1)
    from oracle import pyoracle as po
    o = po.Oracle(d, po.default_params(shake_mass=0.0)); o.setup(False)
    fo, eo, wo = o.compute()
    assert np.abs(f - fo).max() < 1e-11 * np.abs(fo).max()
    s6 = (3.4 / r) ** 6
    assert abs(e[0] - 4 * 0.1 * (s6 * s6 - s6)) < 1e-13
    eng.close()


def test_sampling_window_rule(small_pe):
    """nss = 25: nav = 2, 12 windows -> the average runs over steps 1..24 (in.homogenization.lammps:57-59)."""
    from scema_amd import capi
    from oracle import pyoracle as po
    eng = capi.Engine(capi.default_params(**KW))
    eng.register_replica("pe", 1, small_pe)
    eng.set_state(1, "pe", 1, small_pe["box"], small_pe["x"], small_pe["v"])
    pavg = eng.debug_run("pe", 1, 25, 1.0, 300.0, qp=1, sample=True)
    o = po.Oracle(small_pe, po.default_params(**KW))
    pavg_o, _ = o.run(25, 1.0, 300.0, sample=True)
    assert relerr(pavg, pavg_o) < 1e-8
    eng.close()


def test_neighbour_overflow_regrow(small_pe, monkeypatch):
    """Undersized j tables / rows (test hook): the evaluation is restored from its backup, capacities grow x1.5 per
    attempt, and the result equals the normally sized run."""
    from scema_amd import capi
    lens = _lens(small_pe)
    st = np.array([-3e-4 * lens[0], -3e-4 * lens[1], 1e-3 * lens[2], 0, 0, 0])
    res = []
    for grow0 in (None, "0.6", "0.25"):   # (0.25: rows so short that the truncated run blows up before its end -- the overflow still comes first)
        if grow0 is None:
            monkeypatch.delenv("SCEMA_MD_NEIGH_GROW0", raising=False)
        else:
            monkeypatch.setenv("SCEMA_MD_NEIGH_GROW0", grow0)
        eng = capi.Engine(capi.default_params(**KW))
        eng.register_replica("pe", 1, small_pe)
        res.append(np.array(eng.strain_batch([capi.make_sim(0, "pe", 1, st, nss=10, most_recent=capi.QP_NONE)])[0].stress[:]))
        eng.close()
    assert relerr(res[1], res[0]) < 1e-9 and relerr(res[2], res[0]) < 1e-9


def test_smaller_cells_when_the_j_table_would_not_fit():
    """Long cutoff on the 3456-atom crystal: the tile j table of rlist/2 cells exceeds the LDS budget, the engine
    switches to rlist/3 cells (stencil +-3); forces, energies and virials still match the oracle."""
    from scema_amd import capi
    from scema_amd.systems import build_pe
    from oracle import pyoracle as po
    d = build_pe(4, 6, 12, jitter=0.03, seed=5)
    kw = dict(cut_lj=12.7, cut_coul=9.0, skin=2.0)
    eng = capi.Engine(capi.default_params(**kw))
    eng.register_replica("g0", 1, d)
    f, en, w, info = eng.debug_compute("g0", 1, use_shake=True)
    o = po.Oracle(d, po.default_params(**kw)); o.setup(True)
    fo, eo, wo = o.compute()
    assert info["npairs"] == o.npairs
    assert relerr(f, fo) < 1e-11
    assert np.abs(en[:7] - eo[:7]).max() < 1e-9 * np.abs(eo).max()
    assert np.abs(w[:7] - wo[:7]).max() < 1e-9 * np.abs(wo).max()
    eng.close()


def test_error_paths(small_pe):
    from scema_amd import capi
    eng = capi.Engine()                                           # reference cutoffs: 2*(12+2) = 28 A > 14.8 A box
    eng.register_replica("pe", 1, small_pe)
    lens = _lens(small_pe)
    st = np.array([0, 0, 1e-3 * lens[2], 0, 0, 0])
    with pytest.raises(capi.EngineError, match="box width"):
        eng.strain_batch([capi.make_sim(0, "pe", 1, st, nss=10, most_recent=capi.QP_NONE)])
    eng.close()
    eng = capi.Engine(capi.default_params(**KW))
    eng.register_replica("pe", 1, small_pe)
    # a shear that tilts the box far past L/2 is no error: the box flips on the way (fix deform flip yes)
    out = eng.strain_batch([capi.make_sim(0, "pe", 1, np.array([0, 0, 0, 0.9 * lens[0] * lens[2] / lens[1], 0, 0]), nss=10,
                                          most_recent=capi.QP_NONE, strain_rate=1e-2)])
    assert out[0].stress_updated == 1 and eng.profile()["box_flips"] >= 1
    eng.drop_state(0, "pe", 1)
    with pytest.raises(capi.EngineError, match="not registered"):
        eng.strain_batch([capi.make_sim(0, "nomat", 1, st, nss=10, most_recent=capi.QP_NONE)])
    xb = np.array(small_pe["x"], float).copy(); xb[3, 1] = np.inf
    with pytest.raises(capi.EngineError, match="non-finite"):       # a broken state never reaches the device
        eng.set_state(77, "pe", 1, small_pe["box"], xb, small_pe["v"])
    with pytest.raises(capi.EngineError, match="non-finite strain"):
        eng.strain_batch([capi.make_sim(0, "pe", 1, np.array([np.nan, 0, 0, 0, 0, 0]), nss=10, most_recent=capi.QP_NONE)])
    with pytest.raises(capi.EngineError, match="positive and finite"):
        eng.strain_batch([capi.make_sim(0, "pe", 1, st, nss=10, most_recent=capi.QP_NONE, strain_rate=0.0)])
    with pytest.raises(capi.EngineError, match="straining steps requested"):
        eng.strain_batch([capi.make_sim(0, "pe", 1, st, nss=10, most_recent=capi.QP_NONE, strain_rate=1e-14)])
    with pytest.raises(capi.EngineError, match="reax"):           # config 5 is a 'next' row: explicit, not silent
        eng.strain_batch([capi.make_sim(0, "pe", 1, st, nss=10, most_recent=capi.QP_NONE, force_field="reax")])
    eng.close()


def _alias_types(d, copies):
    """The same system with every atom type split into `copies` aliases (identical Lennard-Jones rows and masses), the
    way force-field generators hand out one atom type per atom name."""
    nt = int(d["ntypes"])
    rng = np.random.default_rng(5)
    out = dict(d)
    out["ntypes"] = nt * copies
    out["type"] = np.asarray(d["type"]) * copies + rng.integers(0, copies, size=len(d["type"]))
    out["mass"] = np.repeat(np.asarray(d["mass"], float), copies)
    out["eps"] = np.repeat(np.repeat(np.asarray(d["eps"], float).reshape(nt, nt), copies, 0), copies, 1).ravel()
    out["sigma"] = np.repeat(np.repeat(np.asarray(d["sigma"], float).reshape(nt, nt), copies, 0), copies, 1).ravel()
    return out


def test_many_atom_types_few_lennard_jones_classes(small_pe):
    """24 atom types that are aliases of 2 Lennard-Jones sites: the engine works on LJ classes, so the type count of the
    data file is not limited to the 16 classes the pair tables hold; 17 genuinely different sites are refused."""
    from scema_amd import capi
    from oracle import pyoracle as po
    many = _alias_types(small_pe, 12)
    assert many["ntypes"] == 24
    eng = capi.Engine(capi.default_params(**KW))
    eng.register_replica("pe", 1, small_pe)
    eng.register_replica("pe24", 1, many)
    f0, e0, w0, _ = eng.debug_compute("pe", 1)
    f1, e1, w1, _ = eng.debug_compute("pe24", 1)
    assert np.array_equal(f0, f1) or relerr(f1, f0) < 1e-13
    o = po.Oracle(many, po.default_params(**KW))
    o.setup(use_shake=False)
    assert relerr(f1, o.compute()[0]) < 1e-11      # the oracle works on the 24 types as given
    lens = _lens(small_pe)
    st = np.array([-3e-4, -3e-4, 1.0e-3, 0, 0, 0]) * np.array([*lens, lens[2], lens[1], lens[0]])
    out = eng.strain_batch([capi.make_sim(0, "pe24", 1, st, nss=10, most_recent=capi.QP_NONE)])
    exp, _ = o.eval(st, 2.0, 300.0, 1e-4, 10)
    assert relerr(out[0].stress[:], exp) < 1e-6
    # 17 distinct sites
    bad = _alias_types(small_pe, 9)     # 18 types ...
    sg = np.asarray(bad["sigma"], float).reshape(18, 18).copy()
    for k in range(18):                 # ... each with its own sigma
        sg[k, :] *= 1.0 + 1e-3 * k
        sg[:, k] *= 1.0 + 1e-3 * k
    bad["sigma"] = sg.ravel()
    with pytest.raises(capi.EngineError, match="Lennard-Jones"):
        eng.register_replica("bad", 1, bad)
    eng.close()


def test_lammps_lj_benchmark_step0_known_answer_on_the_gpu():
    """The step-0 thermo line of LAMMPS' own Lennard-Jones benchmark logs (E_pair, TotEng, Press), from k_pair."""
    from scema_amd import capi
    from test_oracle_physics import lj_bench_system, lj_bench_check
    d, rho = lj_bench_system(6)          # 864 atoms; per-atom values do not depend on the size
    eng = capi.Engine(capi.default_params(cut_lj=2.5, cut_coul=2.5, skin=0.3, shake_mass=0.0))
    eng.register_replica("lj", 1, d)
    f, e, w, info = eng.debug_compute("lj", 1)
    assert np.abs(f).max() < 1e-9
    lj_bench_check(e[0], w[0], d["natoms"], rho)
    eng.close()


def test_unstable_replica_is_reported_not_returned(small_pe):
    """Two atoms on top of each other: forces are non-finite from the first step.  The engine never stores a non-finite
    position (positions index cells and tables), flags the replica and returns an error instead of a NaN stress --
    where LAMMPS would stop with lost atoms."""
    from scema_amd import capi
    bad = dict(small_pe)
    x = np.array(small_pe["x"], float).copy()
    x[200] = x[17]                      # different chains: not an excluded pair
    bad["x"] = x
    eng = capi.Engine(capi.default_params(**KW))
    eng.register_replica("bad", 1, bad)
    eng.register_replica("pe", 1, small_pe)
    lens = _lens(small_pe)
    st = np.array([-3e-4, -3e-4, 1.0e-3, 0, 0, 0]) * np.array([*lens, lens[2], lens[1], lens[0]])
    with pytest.raises(capi.EngineError, match="unstable"):
        eng.strain_batch([capi.make_sim(0, "bad", 1, st, nss=10, most_recent=capi.QP_NONE)])
    # the engine is still usable, and a healthy replica next to a broken one is not what gets blamed
    out = eng.strain_batch([capi.make_sim(1, "pe", 1, st, nss=10, most_recent=capi.QP_NONE)])
    assert np.all(np.isfinite(out[0].stress[:])) and out[0].stress_updated == 1
    eng.close()


def test_list_skin_is_a_performance_knob_only(small_pe, monkeypatch):
    """SCEMA_MD_SKIN_EXTRA adds to (or takes from) the neighbour skin, for performance only.  With a wider and a thinner skin
    than the reference's, forces and a full evaluation still equal the oracle's, which keeps the reference's skin."""
    from scema_amd import capi
    from oracle import pyoracle as po
    lens = _lens(small_pe)
    st = np.array([-3e-4, -3e-4, 1.0e-3, 5e-5, 0, -4e-5]) * np.array([*lens, lens[2], lens[1], lens[0]])
    exp, _ = po.Oracle(small_pe, po.default_params(**KW)).eval(st, 2.0, 300.0, 1e-4, 20)
    res = {}
    for extra in ("0", "0.4", "-0.3"):
        monkeypatch.setenv("SCEMA_MD_SKIN_EXTRA", extra)
        eng = capi.Engine(capi.default_params(**KW))
        eng.register_replica("pe", 1, small_pe)
        out = eng.strain_batch([capi.make_sim(0, "pe", 1, st, nss=20, most_recent=capi.QP_NONE)])
        res[extra] = np.array(out[0].stress[:])
        f, e, w, info = eng.debug_compute("pe", 1)
        res["f" + extra] = f
        eng.close()
    monkeypatch.delenv("SCEMA_MD_SKIN_EXTRA")
    assert relerr(res["0"], exp) < 1e-6 and relerr(res["0.4"], exp) < 1e-6 and relerr(res["-0.3"], exp) < 1e-6
    assert relerr(res["f0.4"], res["f0"]) < 1e-12 and relerr(res["f-0.3"], res["f0"]) < 1e-12


def test_list_skin_adapts_to_the_rebuild_frequency(small_pe, monkeypatch):
    """Opt-in (SCEMA_MD_SKIN_ADAPT=1): a state whose sampling run rebuilt its list more often than every 19 steps is evaluated
    with 25 % more skin next time (and the profile says so); the stresses keep matching the oracle, which never changes its
    skin."""
    from scema_amd import capi
    from oracle import pyoracle as po
    monkeypatch.setenv("SCEMA_MD_SKIN_ADAPT", "1")
    kw = dict(KW, skin=0.6)                       # a thin skin: this small hot system rebuilds every few steps
    eng = capi.Engine(capi.default_params(profile=1, **kw))
    eng.register_replica("pe", 1, small_pe)
    o = po.Oracle(small_pe, po.default_params(**kw))
    lens = _lens(small_pe)
    st = np.array([-2e-4, -2e-4, 6e-4, 0, 0, 0]) * np.array([*lens, lens[2], lens[1], lens[0]])
    skins = []
    for k in range(2):
        out = eng.strain_batch([capi.make_sim(3, "pe", 1, st, nss=60, most_recent=(capi.QP_NONE if k == 0 else 3))])
        exp, _ = o.eval(st, 2.0, 300.0, 1e-4, 60)
        assert relerr(out[0].stress[:], exp) < 2e-6
        skins.append(eng.profile(reset=True)["list_skin_mean"])
    assert skins[0] == pytest.approx(0.6) and skins[1] == pytest.approx(0.75)
    eng.close()


def _affine(box_old, box_new, x):
    """positions carried affinely from one triclinic box to another (what `change_box ... remap` does)"""
    def hmat(b):
        return np.array([[b[3] - b[0], b[6], b[7]], [0.0, b[4] - b[1], b[8]], [0.0, 0.0, b[5] - b[2]]])
    lam = np.linalg.solve(hmat(box_old), (np.asarray(x, float) - np.asarray(box_old[:3], float)).T)
    return (hmat(box_new) @ lam).T + np.asarray(box_new[:3], float)


def test_triclinic_box_flip_matches_the_oracle(small_pe):
    """fix deform `flip yes` (the LAMMPS default behind in.strain.lammps:94-100): a shear run that carries xy across +Lx/2 and
    xz across -Lx/2.  The box flips by one lattice vector between two steps, the list is rebuilt, the k-vector list is
    re-expressed in the new reciprocal basis -- and the trajectory stays on the oracle's."""
    from scema_amd import capi
    from oracle import pyoracle as po
    kw = dict(cut_lj=4.5, cut_coul=4.0, skin=0.9, kspace_accuracy=1e-5)   # the doubly sheared box is 11.8 A wide across x
    d = dict(small_pe)
    box = np.array(small_pe["box"], float)
    lx, ly, lz = box[3] - box[0], box[4] - box[1], box[5] - box[2]
    box[6] = 0.485 * lx
    box[7] = -0.47 * lx
    box[8] = 0.0
    d["box"] = box
    d["x"] = _affine(small_pe["box"], box, small_pe["x"])      # chains are bonded through the periodic boundary: shear the atoms with the box
    rates = np.array([1e-5, -1e-5, 2e-5, 0.004 * lx / ly, -0.003 * lx / lz, 1e-5], float)   # xy up, xz down: both cross
    e = capi.Engine(capi.default_params(**kw))
    e.register_replica("pe", 1, d)
    e.set_state(3, "pe", 1, d["box"], d["x"], d["v"])
    nsteps = 24
    e.debug_run("pe", 1, nsteps, 1.0, 300.0, qp=3, nvt=True, use_shake=True, rates=rates)
    bx, x, v = e.get_state(3, "pe", 1)
    o = po.Oracle(d, po.default_params(**kw))
    o.run(nsteps, 1.0, 300.0, nvt=True, use_shake=True, rates=rates)
    bo, xo, vo = o.get_state()
    assert o.nflips == 2 and e.profile()["box_flips"] == o.nflips
    assert np.abs(bx - bo).max() < 1e-11
    assert -0.5 * lx < bx[6] < -0.4 * lx and 0.4 * lx < bx[7] < 0.5 * lx
    assert np.abs(x - xo).max() < 1e-8 and np.abs(v - vo).max() < 1e-8 * np.abs(vo).max() + 1e-12
    # yz cannot flip while xy is deformed too (it would change xz by xy): LAMMPS refuses the run, and so do we
    d2 = dict(d); b2 = box.copy(); b2[6] = 0.0; b2[7] = 0.0; b2[8] = 0.49 * ly
    d2["box"] = b2; d2["x"] = _affine(small_pe["box"], b2, small_pe["x"])
    e.register_replica("pe", 2, d2)
    e.set_state(4, "pe", 2, d2["box"], d2["x"], d2["v"])
    with pytest.raises(capi.EngineError, match="yz too much"):
        e.debug_run("pe", 2, 12, 1.0, 300.0, qp=4, rates=np.array([0, 0, 0, 0, 0, 0.004 * ly / lz], float))
    o2 = po.Oracle(d2, po.default_params(**kw))
    with pytest.raises(RuntimeError):
        o2.run(12, 1.0, 300.0, rates=np.array([0, 0, 0, 0, 0, 0.004 * ly / lz], float))
    e.close()


def test_persistent_state_shears_across_the_flip_over_several_updates(small_pe):
    """shear tilt accumulates over continuum steps on the persistent state; where LAMMPS flips, so must we.
    Four update() calls of pure xy shear (each 4 % of Ly), stresses against the oracle."""
    from scema_amd import capi
    from oracle import pyoracle as po
    kw = dict(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=1e-5)
    d = dict(small_pe)
    box = np.array(small_pe["box"], float)
    lx, ly, lz = box[3] - box[0], box[4] - box[1], box[5] - box[2]
    box[6] = 0.42 * lx
    d["box"] = box
    d["x"] = _affine(small_pe["box"], box, small_pe["x"])
    e = capi.Engine(capi.default_params(**kw))
    e.register_replica("pe", 1, d)
    o = po.Oracle(d, po.default_params(**kw))
    strain = np.array([0.0, 0.0, 0.0, 0.04 * lz, 0.0, 0.0])     # MDSim.strain[xy] is divided by lz (stmd_problem.h:222-225)
    flips = 0
    for k in range(4):
        sim = capi.make_sim(1, "pe", 1, strain, nss=10, strain_rate=1e-3, most_recent=(capi.QP_NONE if k == 0 else 1))
        got = np.array(e.strain_batch([sim])[0].stress[:])
        exp, nts = o.eval(strain, 2.0, 300.0, 1e-3, 10)
        assert np.abs(got - exp).max() < 1e-6 * np.abs(exp).max(), k
    assert o.nflips >= 1 and e.profile()["box_flips"] == o.nflips
    bx, _, _ = e.get_state(1, "pe", 1)
    assert abs(bx[6]) <= 0.5 * (bx[3] - bx[0]) * 1.02
    e.close()


@pytest.mark.parametrize("acc,min_groups", [(3e-6, 257), (1e-6, 513), (3e-10, 1025)])
def test_large_k_sets_stay_on_the_table_path(small_pe, acc, min_groups):
    """replicas of a few 10^4 atoms at the reference's accuracy have more k-vector groups than a workgroup has threads (256):
    threads then own 2 or 4 groups each (beyond 1024 groups the rest takes the plain path).  Same reciprocal energy, virial and forces as the oracle's plain sum."""
    from scema_amd import capi
    from oracle import pyoracle as po
    kw = dict(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=acc)
    e = capi.Engine(capi.default_params(kspace_style=0, **kw))            # the Ewald sum (the default is PPPM)
    e.register_replica("pe", 1, small_pe)
    f, en, w, info = e.debug_compute("pe", 1, use_shake=False)
    o = po.Oracle(small_pe, po.default_params(kspace_pppm=0, **kw))
    o.setup(False)
    fo, eo, wo = o.compute()
    assert info["nk"] == o.nkvec and info["nk"] > 3.2 * min_groups        # about 4 k-vectors per group
    assert abs(en[6] - eo[6]) < 1e-10 * abs(eo[6]) and np.abs(w[6] - wo[6]).max() < 1e-10 * np.abs(wo[6]).max()
    assert np.abs(f - fo).max() < 1e-10 * np.abs(fo).max()
    e.close()


@pytest.mark.parametrize("acc,library", [(1e-4, False), (1e-4, True), (1e-6, False)])
def test_pppm_matches_the_oracle_pppm(small_pe, acc, library, monkeypatch):
    """kspace_style 1 (`kspace_style pppm`, in.set.lammps:36): charge assignment in LDS, transforms (small grids: the fused solve in
    LDS, k_pppm_solve; larger ones, or SCEMA_MD_PPPM_FFT=1: batched hipFFT + k_pppm_poisson), influence function, field
    interpolation -- against the oracle's PPPM (plain sums per line instead of FFTs): same grid rule, same adjusted g_ewald,
    reciprocal energy, virial and forces to round-off; and, like the oracle's, close to the Ewald sum at the accuracy asked for."""
    from scema_amd import capi
    from oracle import pyoracle as po
    if library:
        monkeypatch.setenv("SCEMA_MD_PPPM_FFT", "1")
    kw = dict(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=acc)
    e = capi.Engine(capi.default_params(kspace_style=1, **kw))
    e.register_replica("pe", 1, small_pe)
    f, en, w, info = e.debug_compute("pe", 1, use_shake=False)
    o = po.Oracle(small_pe, po.default_params(kspace_pppm=1, **kw))
    o.setup(False)
    fo, eo, wo = o.compute()
    assert info["nk"] == 0 and abs(info["g_ewald"] - o.g_ewald) < 1e-12   # Newton step with a 1e-6 forward difference: last bits x 1e6
    assert abs(en[6] - eo[6]) < 1e-10 * abs(eo[6]) and np.abs(w[6] - wo[6]).max() < 1e-10 * np.abs(wo[6]).max()
    assert abs(en[1] - eo[1]) < 1e-10 * abs(eo[1])                      # real-space part with the adjusted g_ewald
    assert np.abs(f - fo).max() < 1e-10 * np.abs(fo).max()
    oe = po.Oracle(small_pe, po.default_params(kspace_pppm=0, **kw)); oe.setup(False)
    fe = oe.compute()[0]
    assert np.sqrt(((f - fe) ** 2).sum(1).mean()) < 12.0 * acc * 332.06371
    e.close()


def test_pppm_trajectory_with_deform_and_full_evaluation(small_pe):
    """the mesh path through a whole evaluation: straining run (influence function of every step's box), sampling run, stress"""
    from scema_amd import capi
    from oracle import pyoracle as po
    kw = dict(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=1e-5)
    e = capi.Engine(capi.default_params(kspace_style=1, **kw))
    e.register_replica("pe", 1, small_pe)
    lens = small_pe["box"][3:6] - small_pe["box"][:3]
    strain = np.array([-4e-4 * lens[0], -4e-4 * lens[1], 1.2e-3 * lens[2], 2e-4 * lens[2], 0.0, -1e-4 * lens[0]])
    out = e.strain_batch([capi.make_sim(q, "pe", 1, strain * (1 + 0.2 * q), nss=20, most_recent=capi.QP_NONE) for q in range(3)])
    for q in range(3):
        o = po.Oracle(small_pe, po.default_params(kspace_pppm=1, **kw))
        exp, _ = o.eval(strain * (1 + 0.2 * q), 2.0, 300.0, 1e-4, 20)
        got = np.array(out[q].stress[:])
        assert np.abs(got - exp).max() < 1e-7 * np.abs(exp).max(), (q, got, exp)
    e.close()


def test_pppm_grid_too_large_for_the_lds_and_mixed_grids(small_pe):
    """a grid beyond the LDS (tight accuracy) takes the global-atomics spreading path; replicas with different boxes (hence different
    grids) interleaved in one batch are regrouped by grid and take one batched transform per group (the smaller grid lies in a
    buffer laid out for the larger one)"""
    from copy import deepcopy
    from scema_amd import capi
    from oracle import pyoracle as po
    kw = dict(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=3e-8)
    e = capi.Engine(capi.default_params(kspace_style=1, **kw))
    e.register_replica("pe", 1, small_pe)
    f, en, w, info = e.debug_compute("pe", 1, use_shake=False)
    o = po.Oracle(small_pe, po.default_params(kspace_pppm=1, **kw))
    o.setup(False)
    fo, eo, wo = o.compute()
    nx, ny, nz = o.pppm_grid
    assert nx * ny * nz * 8 > 144 * 1024                                     # larger than the LDS budget of k_pppm_spread
    assert abs(en[6] - eo[6]) < 1e-9 * abs(eo[6]) and np.abs(f - fo).max() < 1e-9 * np.abs(fo).max()
    e.close()
    # mixed grids: the second replica type lives in a box stretched by 50 % along z (18 instead of 15 grid planes)
    kw = dict(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=1e-5)
    d2 = deepcopy(small_pe)
    d2["box"] = d2["box"].copy(); d2["x"] = d2["x"].copy()
    lz = d2["box"][5] - d2["box"][2]
    d2["x"][:, 2] = d2["box"][2] + (d2["x"][:, 2] - d2["box"][2]) * 1.5
    d2["box"][5] = d2["box"][2] + 1.5 * lz
    e = capi.Engine(capi.default_params(kspace_style=1, **kw))
    e.register_replica("a", 1, small_pe)
    e.register_replica("b", 1, d2)
    st = np.array([1e-3, -5e-4, 2e-3, 0.0, 1e-3, 0.0])
    mats = ["a", "b", "a", "b", "b"]                                          # interleaved: a a | b b b after the regrouping
    scale = [1.0, 1.0, 0.7, -0.6, 1.3]
    out = e.strain_batch([capi.make_sim(q, m, 1, st * scale[q], nss=10, most_recent=capi.QP_NONE) for q, m in enumerate(mats)])
    grids = {}
    for q, m in enumerate(mats):
        oq = po.Oracle(small_pe if m == "a" else d2, po.default_params(kspace_pppm=1, **kw))
        exp, _ = oq.eval(st * scale[q], 2.0, 300.0, 1e-4, 10)
        grids[m] = oq.pppm_grid
        assert np.abs(np.array(out[q].stress[:]) - exp).max() < 1e-7 * np.abs(exp).max(), q
    assert grids["a"] != grids["b"]
    e.close()


def test_rarely_taken_list_build_paths_give_the_same_lists():
    """k_neigh_build: a quarter of a cell's clusters whose reach exceeds its list capacity walks the whole j table instead (never seen
    at the default capacity).  Forced here by a tiny capacity (SCEMA_MD_QCAP16, read once per process -> a child process): same
    pairs, forces and energies as the default run -- pad slots of a cell must not list each other on that path."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import json, numpy as np\n"
            "from scema_amd import capi\n"
            "from scema_amd.systems import build_pe\n"
            "d = build_pe(4, 6, 12, jitter=0.03, seed=11)\n"
            "d['box'][6:9] = [0.4, -0.3, 0.2]\n"
            "e = capi.Engine()\n"
            "e.register_replica('g0', 1, d)\n"
            "f, en, w, info = e.debug_compute('g0', 1, use_shake=True)\n"
            "print(json.dumps({'npairs': float(info['npairs']), 'f': np.asarray(f).ravel().tolist(), 'en': np.asarray(en)[:7].tolist(), 'w': np.asarray(w)[:7].ravel().tolist()}))\n")
    out = {}
    # (the third run takes the three-kernel cell binning of replicas too large for the one-launch k_cell_build)
    for name, env in (("default", {}), ("whole_table", {"SCEMA_MD_QCAP16": "2"}), ("three_kernel_binning", {"SCEMA_MD_CELL_BUILD": "0"})):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-2000:]
        out[name] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    a = out["default"]
    for name in ("whole_table", "three_kernel_binning"):
        b = out[name]
        assert a["npairs"] == b["npairs"], name
        assert np.all(np.isfinite(b["en"])) and np.all(np.isfinite(b["w"])), name
        assert np.abs(np.array(a["f"]) - np.array(b["f"])).max() < 1e-11 * np.abs(np.array(a["f"])).max(), name
        assert np.abs(np.array(a["en"]) - np.array(b["en"])).max() < 1e-11 * np.abs(np.array(a["en"])).max(), name


def _child(code, env):
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=root, env=dict(os.environ, **env))
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-2500:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


def test_fp32_list_builds_list_a_superset_and_change_no_force():
    """k_neigh_build tests its candidates in FP32 on tile-relative coordinates against a radius widened by the error bound of that arithmetic
    (every build but the first of a run; SCEMA_MD_NEIGH_EXACT=0: the first too, =1: none -- read once per process, hence child processes).  The
    rows are then a superset of the exact ones by a band of ~1e-5 of the pairs, k_pair tests every r^2 in FP64: same forces, same trajectory."""
    code = ("import json, numpy as np\n"
            "from scema_amd import capi\n"
            "from scema_amd.systems import build_pe\n"
            "d = build_pe(4, 6, 12, jitter=0.05, seed=11, shake_project=True)\n"
            "e = capi.Engine()\n"
            "e.register_replica('g0', 1, d)\n"
            "f, en, w, info = e.debug_compute('g0', 1, use_shake=True)\n"
            "e.set_state(5, 'g0', 1, d['box'], d['x'], d['v'])\n"
            "e.debug_run('g0', 1, 40, 2.0, 300.0, qp=5, nvt=True, use_shake=True, rates=[2e-5, -1e-5, 1e-5, 3e-6, -2e-6, 1e-6])\n"
            "x = e.get_state(5, 'g0', 1)[1]\n"
            "print(json.dumps({'f': np.asarray(f).ravel().tolist(), 'npairs': float(info['npairs']), 'x': np.asarray(x).ravel().tolist(), 'e': np.asarray(en)[:7].tolist()}))\n")
    exact = _child(code, {"SCEMA_MD_NEIGH_EXACT": "1"})
    dflt = _child(code, {})
    fp32 = _child(code, {"SCEMA_MD_NEIGH_EXACT": "0"})
    fe = np.array(exact["f"])
    assert dflt["npairs"] == exact["npairs"]                                    # the first build of a run is the exact one
    assert exact["npairs"] <= fp32["npairs"] <= exact["npairs"] * (1 + 1e-4)    # the eps band: a few pairs in a hundred thousand
    for other in (dflt, fp32):
        assert np.abs(np.array(other["f"]) - fe).max() < 1e-11 * np.abs(fe).max()
        assert np.abs(np.array(other["e"]) - np.array(exact["e"])).max() < 1e-10 * np.abs(exact["e"]).max()
        # 40 steps of NVT + SHAKE + deform with list rebuilds: the same trajectory to the noise of the FP64 atomics
        assert np.abs(np.array(other["x"]) - np.array(exact["x"])).max() < 1e-8


def test_kept_neighbour_rows_equal_rebuilt_ones(small_pe):
    """Neighbour rows survive from the straining run to the sampling run and from one update to the next where the device finds every atom
    within the list's displacement bound of its reference position (k_keep_validate); SCEMA_MD_KEEP_LIST=0 rebuilds at every run start.  An
    update sequence in which one state is REPLACED between two updates (its rows no longer fit: the build must be made) and another continues:
    same stresses either way."""
    code = ("import json, numpy as np\n"
            "from scema_amd import capi\n"
            "from scema_amd.systems import build_pe\n"
            "d = build_pe(2, 3, 5, jitter=0.05, seed=7); d['box'][6:9] = [0.7, -0.4, 0.5]\n"
            "kw = dict(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=1e-5)\n"
            "e = capi.Engine(capi.default_params(**kw))\n"
            "e.register_replica('pe', 1, d)\n"
            "L = d['box'][3:6] - d['box'][:3]\n"
            "st = np.array([-3e-4 * L[0], -3e-4 * L[1], 1e-3 * L[2], 2e-5 * L[2], 0, 0])\n"
            "out = []\n"
            "a = e.strain_batch([capi.make_sim(q, 'pe', 1, st * (1 + 0.2 * q), nss=20, most_recent=capi.QP_NONE) for q in (0, 1)])\n"
            "out += [list(o.stress) for o in a]\n"
            "a = e.strain_batch([capi.make_sim(q, 'pe', 1, -st, nss=20) for q in (0, 1)])\n"
            "out += [list(o.stress) for o in a]\n"
            "box, x, v = e.get_state(1, 'pe', 1)\n"
            "rng = np.random.default_rng(3)\n"
            "e.set_state(0, 'pe', 1, box, x + rng.normal(0, 0.02, x.shape), v)      # qp 0 becomes (a perturbed copy of) qp 1's state\n"
            "a = e.strain_batch([capi.make_sim(q, 'pe', 1, st, nss=20) for q in (0, 1)])\n"
            "out += [list(o.stress) for o in a]\n"
            "p = e.profile()\n"
            "print(json.dumps({'s': out, 'builds': p['neigh_builds'], 'steps': p['md_steps']}))\n")
    keep = _child(code, {})
    nokeep = _child(code, {"SCEMA_MD_KEEP_LIST": "0"})
    a, b = np.array(keep["s"]), np.array(nokeep["s"])
    assert np.abs(a - b).max() < 1e-9 * np.abs(b).max()
    assert keep["steps"] == nokeep["steps"]
    # 6 evaluations x 2 runs rebuild at their start without the kept rows; with them only the very first run of each state and the run
    # that follows the replaced state do
    assert keep["builds"] <= nokeep["builds"] - 6, (keep["builds"], nokeep["builds"])


def test_round6_launch_variants_give_the_same_stresses(small_pe):
    """Round 6 changed WHEN and WHERE things run for small batches, never what is computed: the replicas of a launch rebuild their neighbour
    rows together as soon as one asks for it (SCEMA_MD_REBUILD_TOGETHER=0: each on its own trigger), the bonded kernel follows the PPPM chain on
    the side stream for batches of 8 and more (SCEMA_MD_BONDED_SIDE=0), the in-LDS PPPM solve of batches under 8 is two workgroups per replica
    (SCEMA_MD_PPPM_SOLVE_TWO=0), every batch that runs whole takes the grid with the most cells (SCEMA_MD_SMALL_BATCH_MAX=0: the largest
    cells).  An 8-replica update sequence (the largest batch that runs whole) with different strains per replica (so that their lists age differently) and a 2-replica one: the
    same stresses with every switch, and more list builds with the common trigger than with the replicas' own."""
    code = ("import json, os, numpy as np\n"
            "from scema_amd import capi\n"
            "from scema_amd.systems import build_pe\n"
            "d = build_pe(2, 3, 5, jitter=0.05, seed=7); d['box'][6:9] = [0.7, -0.4, 0.5]\n"
            "kw = dict(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=1e-5)\n"
            "e = capi.Engine(capi.default_params(**kw))\n"
            "e.register_replica('pe', 1, d)\n"
            "L = d['box'][3:6] - d['box'][:3]\n"
            "st = np.array([-3e-4 * L[0], -3e-4 * L[1], 1e-3 * L[2], 2e-5 * L[2], 0, 0])\n"
            "out = []\n"
            "for n in (8, 2):\n"
            "    a = e.strain_batch([capi.make_sim(100 * n + q, 'pe', 1, st * (1 + 0.3 * q), nss=40, most_recent=capi.QP_NONE) for q in range(n)])\n"
            "    out += [list(o.stress) for o in a]\n"
            "    a = e.strain_batch([capi.make_sim(100 * n + q, 'pe', 1, -st * (1 + 0.1 * q), nss=40) for q in range(n)])\n"
            "    out += [list(o.stress) for o in a]\n"
            "p = e.profile()\n"
            "print(json.dumps({'s': out, 'builds': p['neigh_builds'], 'steps': p['md_steps']}))\n")
    ref = _child(code, {})
    a = np.array(ref["s"])
    for env in ({"SCEMA_MD_REBUILD_TOGETHER": "0"}, {"SCEMA_MD_BONDED_SIDE": "0"}, {"SCEMA_MD_PPPM_SOLVE_TWO": "0"}, {"SCEMA_MD_SMALL_BATCH_MAX": "0"},
                {"SCEMA_MD_REBUILD_TOGETHER": "0", "SCEMA_MD_BONDED_SIDE": "0", "SCEMA_MD_PPPM_SOLVE_TWO": "0", "SCEMA_MD_SMALL_BATCH_MAX": "0"}):
        other = _child(code, env)
        b = np.array(other["s"])
        assert np.abs(a - b).max() < 1e-9 * np.abs(b).max(), (env, np.abs(a - b).max() / np.abs(b).max())
        assert other["steps"] == ref["steps"]
        if env.get("SCEMA_MD_REBUILD_TOGETHER") == "0" and len(env) == 1:
            assert ref["builds"] >= other["builds"], (ref["builds"], other["builds"])   # a common trigger builds at least as often


def _parts_code(n=13, sheared=(2, 7)):
    """the child process of the part-batch tests: n replicas of ragged length, two of them sheared across a box flip; two updates"""
    code = ("import json, os, numpy as np\n"
            "from scema_amd import capi\n"
            "from scema_amd.systems import build_pe\n"
            "d = build_pe(2, 3, 5, jitter=0.05, seed=7); d['box'][6:9] = [0.7, -0.4, 0.5]\n"
            "kw = dict(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=1e-5, neigh_delay=0, kspace_style=int(os.environ.get('TEST_KSPACE', '1')))\n"
            "e = capi.Engine(capi.default_params(**kw))\n"
            "e.register_replica('pe', 1, d)\n"
            "L = d['box'][3:6] - d['box'][:3]\n"
            "st = np.array([-3e-4 * L[0], -3e-4 * L[1], 1e-3 * L[2], 2e-5 * L[2], 0, 0])\n"
            "out = []\n"
            "n = 13\n"
            "big = np.array([0, 0, 0, 0.9 * L[0] * L[2] / L[1], 0, 0])\n"
            "a = e.strain_batch([capi.make_sim(q, 'pe', 1, st * (1 + 0.3 * q) + (big if q in (2, 7) else 0), nss=40 - q, most_recent=capi.QP_NONE,\n"
            "                                  strain_rate=1e-2 if q in (2, 7) else 1e-4) for q in range(n)])\n"
            "out += [list(o.stress) for o in a]\n"
            "a = e.strain_batch([capi.make_sim(q, 'pe', 1, -st * (1 + 0.1 * q), nss=28 + q) for q in range(n)])\n"
            "out += [list(o.stress) for o in a]\n"
            "p = e.profile()\n"
            "print(json.dumps({'s': out, 'steps': p['md_steps'], 'flips': p['box_flips']}))\n")
    return code.replace("n = 13", f"n = {n}").replace("(2, 7)", repr(tuple(sheared)))


def test_part_batches_give_the_same_stresses_as_the_whole_batch():
    """Batches of 10 replicas and more run as part batches on streams of their own (engine_run.cpp: three parts for 9 replicas, four for 10-63, two
    halves from 64 on); the parts are independent, so the count changes WHEN things run, never what is computed.  A batch of
    13 replicas of ragged length (nss 40 down to 28) with different strains -- two of them sheared until their boxes flip, which is
    host work between two steps of their part --, then its reverse from the states it left: the same stresses
    whole (SCEMA_MD_SPLIT=0), as the table's four parts, as two and as three parts, and the same number of MD steps.  (neigh_modify delay 0: with the reference's `delay 5` and this test's skin of 1 A a list is
    used past its validity wherever an atom covers half the skin within five steps of a build -- LAMMPS' "dangerous builds", certain in a box
    sheared by 0.3 A per step -- and the result then depends on the step a list was built at, here as in LAMMPS.)"""
    code = _parts_code()
    ref = _child(code, {"SCEMA_MD_SPLIT": "0"})
    assert ref["flips"] >= 2
    a = np.array(ref["s"])
    assert a.shape == (26, 6) and np.isfinite(a).all()
    for env in ({}, {"SCEMA_MD_PARTS": "2"}, {"SCEMA_MD_PARTS": "3"}, {"SCEMA_MD_ONE_STREAM": "1"}):   # (ONE_STREAM: an engine without its side streams)
        other = _child(code, env)
        b = np.array(other["s"])
        assert np.abs(a - b).max() < 1e-9 * np.abs(b).max(), (env, np.abs(a - b).max() / np.abs(b).max())
        assert other["steps"] == ref["steps"]
        assert other["flips"] == ref["flips"]
    # the plain Ewald sum: a flip re-expresses the k-vector tables of its replica, uploaded on the stream of that replica's part
    ref = _child(code, {"SCEMA_MD_SPLIT": "0", "TEST_KSPACE": "0"})
    other = _child(code, {"TEST_KSPACE": "0"})
    a, b = np.array(ref["s"]), np.array(other["s"])
    assert np.abs(a - b).max() < 1e-9 * np.abs(b).max() and other["flips"] == ref["flips"] >= 2


def test_rows_kept_from_the_straining_run_pass_the_lists_own_test_first():
    """The sampling run of an evaluation keeps the neighbour rows of its straining run (k_phase_init) -- where the list's displacement test says
    so for the positions and the box it starts from: the straining run ends with one more remap of fix deform behind its last force evaluation
    (3e-3 A per step at the reference's strain rate, 0.3 A at 1e-2 per fs, against half a skin of 0.5 A).  Until round 6 the rows stood
    unchecked and a fast shear lost pairs in the set-up evaluation of the sampling run (1e-7 of the stress; found by the part-batch test
    above).  One replica sheared at 1e-2 per fs across a box flip, then strained back: the same stresses as with a list build at the start of
    every run (SCEMA_MD_KEEP_LIST=0), to rounding."""
    code = ("import json, numpy as np\n"
            "from scema_amd import capi\n"
            "from scema_amd.systems import build_pe\n"
            "d = build_pe(2, 3, 5, jitter=0.05, seed=7); d['box'][6:9] = [0.7, -0.4, 0.5]\n"
            "e = capi.Engine(capi.default_params(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=1e-5, neigh_delay=0))\n"
            "e.register_replica('pe', 1, d)\n"
            "L = d['box'][3:6] - d['box'][:3]\n"
            "st = np.array([-5e-4 * L[0], -5e-4 * L[1], 1.6e-3 * L[2], 0.9 * L[0] * L[2] / L[1], 0, 0])\n"
            "out = [list(e.strain_batch([capi.make_sim(0, 'pe', 1, st, nss=38, most_recent=capi.QP_NONE, strain_rate=1e-2)])[0].stress)]\n"
            "out += [list(e.strain_batch([capi.make_sim(0, 'pe', 1, -0.02 * st, nss=30, strain_rate=1e-3)])[0].stress)]\n"
            "p = e.profile()\n"
            "print(json.dumps({'s': out, 'builds': p['neigh_builds'], 'flips': p['box_flips']}))\n")
    kept, built = _child(code, {}), _child(code, {"SCEMA_MD_KEEP_LIST": "0"})
    a, b = np.array(kept["s"]), np.array(built["s"])
    assert kept["flips"] >= 1 and kept["flips"] == built["flips"]
    assert np.abs(a - b).max() < 1e-11 * np.abs(b).max(), np.abs(a - b).max() / np.abs(b).max()
    assert kept["builds"] <= built["builds"]


def test_part_batches_of_two_materials_give_the_same_stresses_as_the_whole_batch():
    """A launch group of replicas of two materials of different size (1 080 and 1 728 atoms, different boxes, grids and PPPM meshes) and ragged
    length, one of them sheared across a box flip, over two updates (the second continues the states of the first with the strains
    reversed): whole, as the table's parts and as two halves -- the same stresses to rounding.  (delay 0 as in the test above.)"""
    code = ("import json, numpy as np\n"
            "from scema_amd import capi\n"
            "from scema_amd.systems import build_pe\n"
            "a = build_pe(2, 3, 5, jitter=0.05, seed=7); a['box'][6:9] = [0.7, -0.4, 0.5]\n"
            "b = build_pe(2, 4, 6, jitter=0.04, seed=11)\n"
            "e = capi.Engine(capi.default_params(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=1e-5, neigh_delay=0))\n"
            "e.register_replica('a', 1, a); e.register_replica('b', 1, b)\n"
            "def st(d, q):\n"
            "    L = d['box'][3:6] - d['box'][:3]\n"
            "    return np.array([-3e-4 * L[0], 2e-4 * L[1], 1e-3 * L[2], 2e-5 * L[2], -1e-5 * L[2], 0]) * (1 + 0.2 * q)\n"
            "La = a['box'][3:6] - a['box'][:3]\n"
            "big = np.array([0, 0, 0, 0.9 * La[0] * La[2] / La[1], 0, 0])\n"
            "n = 15\n"
            "mat = lambda q: ('a', a, 0) if q % 3 else ('b', b, 1)\n"
            "out = []\n"
            "for sign, rec in ((1, capi.QP_NONE), (-1, None)):\n"
            "    sims = []\n"
            "    for q in range(n):\n"
            "        name, d, m = mat(q)\n"
            "        s = sign * st(d, q) + (big if (q == 4 and sign == 1) else 0)\n"
            "        sims.append(capi.make_sim(q, name, 1, s, nss=22 + (5 * q) % 13, most_recent=rec, material=m, strain_rate=1e-2 if (q == 4 and sign == 1) else 1e-4))\n"
            "    out += [list(o.stress) for o in e.strain_batch(sims)]\n"
            "p = e.profile()\n"
            "print(json.dumps({'s': out, 'steps': p['md_steps'], 'flips': p['box_flips']}))\n")
    ref = _child(code, {"SCEMA_MD_SPLIT": "0"})
    a = np.array(ref["s"])
    assert a.shape == (30, 6) and np.isfinite(a).all() and ref["flips"] >= 1
    for env in ({}, {"SCEMA_MD_PARTS": "2"}, {"SCEMA_MD_PARTS": "3"}):
        other = _child(code, env)
        b = np.array(other["s"])
        assert np.abs(a - b).max() < 1e-9 * np.abs(b).max(), (env, np.abs(a - b).max() / np.abs(b).max())
        assert other["steps"] == ref["steps"] and other["flips"] == ref["flips"]


def test_part_batches_transform_their_pppm_meshes_through_plans_of_their_own():
    """Meshes beyond the in-LDS solve go through hipFFT, whose plans own a work area and are bound to a stream when they run: one per mesh
    size, batch count AND stream.  Round 6's part batches first ran with plans keyed by "main stream / side stream / any other": the second
    and the fourth part of a batch shared theirs, and two replicas with the same large mesh, one in each of those parts, came out wrong by
    1e-4 in one run of ten (with six parts: nine of ten -- how it was found).  A race is no test; the plans are counted instead: twelve
    replicas of one box, every mesh through hipFFT (SCEMA_MD_PPPM_FFT=1), as four parts of three hold 4 streams x (charge meshes, field
    meshes) = 8 plans, the batch whole 2 -- and the stresses agree."""
    code = ("import json, os, numpy as np\n"
            "from scema_amd import capi\n"
            "from scema_amd.systems import build_pe\n"
            "d = build_pe(2, 3, 5, jitter=0.05, seed=7)\n"
            "e = capi.Engine(capi.default_params(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=1e-5, neigh_delay=0))\n"
            "e.register_replica('pe', 1, d)\n"
            "L = d['box'][3:6] - d['box'][:3]\n"
            "st = np.array([-3e-4 * L[0], -3e-4 * L[1], 1e-3 * L[2], 2e-5 * L[2], 0, 0])\n"
            "out = [list(o.stress) for o in e.strain_batch([capi.make_sim(q, 'pe', 1, st, nss=14, most_recent=capi.QP_NONE) for q in range(12)])]\n"
            "print(json.dumps({'s': out, 'plans': e.pppm_plan_count()}))\n")
    whole = _child(code, {"SCEMA_MD_SPLIT": "0", "SCEMA_MD_PPPM_FFT": "1"})
    parts = _child(code, {"SCEMA_MD_PARTS": "4", "SCEMA_MD_PPPM_FFT": "1"})
    assert whole["plans"] == 2 and parts["plans"] == 8, (whole["plans"], parts["plans"])
    a, b = np.array(whole["s"]), np.array(parts["s"])
    assert np.abs(a - b).max() < 1e-9 * np.abs(b).max()


@pytest.mark.parametrize("n", [11, 23, 37])
def test_a_replica_does_not_know_its_batch(n):
    """What a replica returns does not depend on the company it is evaluated in: n replicas of ragged length and random strains in one
    update() -- part batches, common list rebuilds, kept rows in the second update, which continues the states of the first with other
    strains -- against every replica evaluated alone in an engine of its own, two updates each.  (delay 0: see the part-batch test.)"""
    from scema_amd import capi
    from scema_amd.systems import build_pe
    d = build_pe(2, 3, 5, jitter=0.05, seed=7)
    d["box"][6:9] = [0.4, -0.3, 0.2]
    L = d["box"][3:6] - d["box"][:3]
    rng = np.random.default_rng(100 + n)
    amp = np.array([1e-3 * L[0], 1e-3 * L[1], 1e-3 * L[2], 3e-4 * L[2], 3e-4 * L[2], 3e-4 * L[1]])
    strains = [rng.uniform(-1, 1, (n, 6)) * amp for _ in range(2)]
    nss = [rng.integers(12, 30, n) for _ in range(2)]
    kw = dict(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=1e-4, neigh_delay=0)

    def run(qs):
        e = capi.Engine(capi.default_params(**kw))
        e.register_replica("pe", 1, d)
        out = []
        for u in range(2):
            res = e.strain_batch([capi.make_sim(int(q), "pe", 1, strains[u][q], nss=int(nss[u][q]), most_recent=capi.QP_NONE if u == 0 else None) for q in qs])
            out.append(np.array([list(o.stress) for o in res]))
        e.close()
        return out

    together = run(range(n))
    scale = max(np.abs(t).max() for t in together)
    for q in rng.choice(n, size=min(n, 7), replace=False):
        alone = run([int(q)])
        for u in range(2):
            assert np.abs(together[u][q] - alone[u][0]).max() < 1e-9 * scale, (int(q), u, np.abs(together[u][q] - alone[u][0]).max() / scale)
