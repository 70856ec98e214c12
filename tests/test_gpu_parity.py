"""GPU parity tests: the HIP engine (through the C ABI) against the CPU oracle on identical seeded
inputs.  FP64 throughout; tolerances are stated at each assert (summation order differs, nothing else)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng_small():
    from scema_amd import capi
    p = capi.default_params(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=1e-5)
    e = capi.Engine(p)
    yield e
    e.close()


def oracle_small(d, **kw):
    from oracle import pyoracle as po
    base = dict(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=1e-5)
    base.update(kw)
    return po.Oracle(d, po.default_params(**base))


def relerr(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return np.abs(a - b).max() / max(1e-300, np.abs(b).max())


@pytest.mark.parametrize("use_shake", [False, True])
def test_static_forces_energies_virials(small_pe, eng_small, use_shake):
    from scema_amd import capi
    eng_small.register_replica("pe", 1, small_pe)
    f, e, w, info = eng_small.debug_compute("pe", 1, use_shake=use_shake)
    o = oracle_small(small_pe)
    o.setup(use_shake=use_shake)
    fo, eo, wo = o.compute()
    assert info["nk"] == o.nkvec and abs(info["g_ewald"] - o.g_ewald) < 1e-14
    assert info["npairs"] == o.npairs          # same unique pairs within cutoff+skin
    assert info["tdof"] == o.tdof
    assert relerr(f, fo) < 1e-11
    for part in range(7):
        scale = max(1.0, abs(eo[part]))
        assert abs(e[part] - eo[part]) < 1e-10 * scale, capi.PARTS[part]
        assert np.abs(w[part] - wo[part]).max() < 1e-10 * max(1.0, np.abs(wo[part]).max()), capi.PARTS[part]


def test_static_full_cutoffs_midsize():
    """3456-atom PE crystal at the reference's real cutoffs (12 / 9 / skin 2, kspace 1e-4)."""
    from scema_amd import capi
    from scema_amd.systems import build_pe
    from oracle import pyoracle as po
    d = build_pe(4, 6, 12, jitter=0.03, seed=11)
    d["box"][6:9] = [0.4, -0.3, 0.2]
    e = capi.Engine()
    e.register_replica("g0", 1, d)
    f, en, w, info = e.debug_compute("g0", 1, use_shake=True)
    o = po.Oracle(d)
    o.setup(True)
    fo, eo, wo = o.compute()
    assert info["npairs"] == o.npairs and info["nk"] == o.nkvec
    assert relerr(f, fo) < 1e-11
    assert np.abs(en[:7] - eo[:7]).max() < 1e-9 * np.abs(eo).max()
    assert np.abs(w[:7] - wo[:7]).max() < 1e-9 * np.abs(wo).max()
    e.close()


@pytest.mark.parametrize("mode", ["nve", "nvt_shake", "nvt_shake_deform"])
def test_short_run_trajectory(small_pe, eng_small, mode):
    """20 steps of the full per-step order (A.2) land on the same state as the oracle."""
    eng_small.register_replica("pe", 2, small_pe)
    nvt = mode != "nve"
    shake = mode != "nve"
    rates = np.array([1e-5, -2e-5, 3e-5, 1.5e-5, -0.5e-5, 2.5e-5]) if mode.endswith("deform") else None
    nsteps, dt = 20, 1.0
    eng_small.set_state(5, "pe", 2, small_pe["box"], small_pe["x"], small_pe["v"])
    pavg = eng_small.debug_run("pe", 2, nsteps, dt, 300.0, qp=5, nvt=nvt, use_shake=shake, rates=rates, sample=True)
    box, x, v = eng_small.get_state(5, "pe", 2)
    o = oracle_small(small_pe)
    pavg_o, _ = o.run(nsteps, dt, 300.0, nvt=nvt, use_shake=shake, rates=rates, sample=True)
    bo, xo, vo = o.get_state()
    assert np.abs(box - bo).max() < 1e-12
    assert np.abs(x - xo).max() < 1e-9          # Angstrom
    assert relerr(v, vo) < 1e-8
    assert relerr(pavg, pavg_o) < 1e-8
    eng_small.drop_state(5, "pe", 2)


def test_full_evaluation_matches_oracle(small_pe, eng_small):
    """scema_md_strain_batch == omd_eval (F8): strain in Angstrom -> stress in Pa, within 1e-6 relative
    (the north-star budget is 1e-4)."""
    from scema_amd import capi
    eng_small.register_replica("pe", 3, small_pe)
    lens = small_pe["box"][3:6] - small_pe["box"][:3]
    strain = np.array([-0.3 * 1.2e-3 * lens[0], -0.3 * 1.2e-3 * lens[1], 1.2e-3 * lens[2], 5e-5 * lens[2], -3e-5 * lens[1], 2e-5 * lens[0]])
    sims = [capi.make_sim(7, "pe", 3, strain, nss=20, most_recent=capi.QP_NONE)]
    out = eng_small.strain_batch(sims)
    got = np.array(out[0].stress[:])
    assert out[0].stress_updated == 1
    o = oracle_small(small_pe)
    exp, nts = o.eval(strain, 2.0, 300.0, 1e-4, 20)
    assert nts == 10
    assert relerr(got, exp) < 1e-6
    # state persisted under qp 7 and matches the oracle's end state
    box, x, v = eng_small.get_state(7, "pe", 3)
    bo, xo, vo = o.get_state()
    assert np.abs(box - bo).max() < 1e-12 and np.abs(x - xo).max() < 1e-8
    # second call continues from the stored state (history dependence), again matching the oracle
    sims2 = [capi.make_sim(7, "pe", 3, 0.5 * strain, nss=20)]
    got2 = np.array(eng_small.strain_batch(sims2)[0].stress[:])
    exp2, _ = o.eval(0.5 * strain, 2.0, 300.0, 1e-4, 20)
    assert relerr(got2, exp2) < 1e-6


def test_batch_is_independent_and_ordered(small_pe, eng_small):
    """A batch of sims with different strains / nts equals the same sims run one by one."""
    from scema_amd import capi
    eng_small.register_replica("pe", 4, small_pe)
    lens = small_pe["box"][3:6] - small_pe["box"][:3]
    rng = np.random.default_rng(5)
    strains = []
    for i in range(5):
        ezz = rng.uniform(1e-3, 6e-3)    # nts 10..30: ragged batch
        strains.append(np.array([-0.3 * ezz * lens[0], -0.3 * ezz * lens[1], ezz * lens[2], 0, 0, 0]))
    sims = [capi.make_sim(100 + i, "pe", 4, s, nss=10, most_recent=capi.QP_NONE) for i, s in enumerate(strains)]
    out = eng_small.strain_batch(sims)
    batch = np.array([o.stress[:] for o in out])
    for i, s in enumerate(strains):
        eng_small.drop_state(100 + i, "pe", 4)
    single = []
    for i, s in enumerate(strains):
        o1 = eng_small.strain_batch([capi.make_sim(100 + i, "pe", 4, s, nss=10, most_recent=capi.QP_NONE)])
        single.append(o1[0].stress[:])
    assert relerr(batch, np.array(single)) < 1e-9


def test_lammps_data_file_registers_the_same_system(small_pe, eng_small, tmp_path):
    """A `write_data`-style text file read by scema_md_load_lammps_data gives the same forces as the arrays."""
    from scema_amd.systems import write_lammps_data
    p = str(tmp_path / "pe.data")
    write_lammps_data(p, small_pe)
    eng_small.register_replica("pe", 8, small_pe)
    f0, e0, w0, _ = eng_small.debug_compute("pe", 8)
    eng_small.load_lammps_data("pe", 9, p, natoms=small_pe["natoms"])
    f1, e1, w1, _ = eng_small.debug_compute("pe", 9)
    assert relerr(f1, f0) < 1e-12 and np.abs(e1 - e0).max() < 1e-9 * np.abs(e0).max()


def test_lammps_restart_file_registers_the_same_system(small_pe, eng_small, tmp_path):
    """A binary restart in the LAMMPS 17Nov16 layout (what stmd_problem.h:204 reads) registers the same replica."""
    from scema_amd import capi
    p = str(tmp_path / "init.pe_1.bin")
    capi.write_lammps_restart(p, small_pe, eng_small.params.cut_lj, eng_small.params.cut_coul)
    info = capi.probe_lammps_restart(p)
    assert (info.cut_lj, info.cut_coul) == (eng_small.params.cut_lj, eng_small.params.cut_coul)
    eng_small.register_replica("pe", 18, small_pe)
    f0, e0, w0, _ = eng_small.debug_compute("pe", 18)
    eng_small.load_lammps_restart("pe", 19, p, natoms=small_pe["natoms"])
    f1, e1, w1, _ = eng_small.debug_compute("pe", 19)
    assert relerr(f1, f0) < 1e-12 and np.abs(e1 - e0).max() < 1e-9 * np.abs(e0).max()


def test_empty_batch_and_hooke_mode(small_pe, eng_small):
    from scema_amd import capi
    import ctypes as C
    assert capi.lib().scema_md_strain_batch(eng_small.h, None, C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(1)) == 0
    Cst = np.arange(36, dtype=float).reshape(6, 6); Cst = (Cst + Cst.T) * 1e8
    eps = np.array([1e-3, -2e-4, 5e-4, 3e-4, -1e-4, 2e-4])
    out = eng_small.strain_batch([capi.make_sim(0, "nomat", 1, eps, stiffness=Cst)], hooke=True)   # no replica needed
    from oracle import pyoracle as po
    assert np.allclose(out[0].stress[:], po.hooke(Cst.ravel(), eps), rtol=1e-14)
    assert out[0].stress_updated == 1


def test_ragged_world_sharding_leaves_other_ranks_untouched(small_pe, eng_small):
    """rank 1 of 3 evaluates simulations 1 and 4 only; the others keep stress_updated = 0."""
    from scema_amd import capi
    eng_small.register_replica("pe", 10, small_pe)
    lens = small_pe["box"][3:6] - small_pe["box"][:3]
    st = np.array([-4e-4 * lens[0], -4e-4 * lens[1], 1.2e-3 * lens[2], 0, 0, 0])
    sims = [capi.make_sim(200 + i, "pe", 10, st * (1 + 0.1 * i), nss=10, most_recent=capi.QP_NONE) for i in range(5)]
    out = eng_small.strain_batch(sims, rank=1, world=3)
    assert [o.stress_updated for o in out] == [0, 1, 0, 0, 1]
    ptr, cnt = eng_small.local_stress_ptr()
    assert cnt == 2 and ptr
    # result buffer: 6*cap stresses, then this rank's status word and the hash of the plan it computed
    assert eng_small.local_result_doubles() == 14
    host = np.zeros(14)
    eng_small.copy_local_stress(host.ctypes.data, False)
    assert np.allclose(host[:6], out[1].stress[:]) and np.allclose(host[6:12], out[4].stress[:])
    assert host[12] == 0.0 and host[13] > 0.0 and host[13] == np.floor(host[13])


def test_impropers_match_oracle(small_pe, eng_small):
    """improper_style harmonic on the GPU (k_bonded, every term once): a PE crystal decorated with one improper per
    carbon (C, H, H, next C) -- forces, energies and virials per part against the oracle."""
    from scema_amd import capi
    d = dict(small_pe)
    typ = np.asarray(d["type"])
    bonds = np.asarray(d["bonds"])
    nb = {i: [] for i in range(d["natoms"])}
    for a, b in bonds:
        nb[int(a)].append(int(b)); nb[int(b)].append(int(a))
    imps = []
    for i in range(d["natoms"]):
        if typ[i] != 0:
            continue
        hs = [j for j in nb[i] if typ[j] == 1]
        cs = [j for j in nb[i] if typ[j] == 0]
        if len(hs) >= 2 and cs:
            imps.append([i, hs[0], hs[1], cs[0]])
    assert len(imps) > 50
    d["impropers"] = np.array(imps, np.int32)
    d["improper_type"] = (np.arange(len(imps)) % 2).astype(np.int32)
    d["improper_coeff"] = np.array([[4.5, np.deg2rad(35.0)], [2.0, np.deg2rad(120.0)]])
    eng_small.register_replica("pe_imp", 1, d)
    f, e, w, info = eng_small.debug_compute("pe_imp", 1, use_shake=False)
    o = oracle_small(d)
    o.setup(use_shake=False)
    fo, eo, wo = o.compute()
    assert abs(eo[5]) > 1.0                      # the improper part is really there
    assert relerr(f, fo) < 1e-11
    for part in range(7):
        assert abs(e[part] - eo[part]) < 1e-10 * max(1.0, abs(eo[part])), capi.PARTS[part]
        assert np.abs(w[part] - wo[part]).max() < 1e-10 * max(1.0, np.abs(wo[part]).max()), capi.PARTS[part]
    # and a short trajectory through the production path (lumped virial, forces via LDS tiles)
    lens = d["box"][3:6] - d["box"][:3]
    st = np.array([-3e-4 * lens[0], -3e-4 * lens[1], 1e-3 * lens[2], 0, 0, 0])
    got = np.array(eng_small.strain_batch([capi.make_sim(7, "pe_imp", 1, st, nss=10, most_recent=capi.QP_NONE)])[0].stress[:])
    exp, _ = oracle_small(d).eval(st, 2.0, 300.0, 1e-4, 10)
    assert relerr(got, exp) < 1e-6


def test_shake_clusters_of_two_and_four_trajectory():
    """k_shake on clusters of 2 (closed form) and 4 atoms (fixed-point iteration), three atom types, tilted box:
    20 steps of NVT + SHAKE + deform land on the oracle's state."""
    from scema_amd import capi
    from scema_amd.systems import build_ethane_oh
    from oracle import pyoracle as po
    d = build_ethane_oh()
    kw = dict(cut_lj=5.5, cut_coul=5.0, skin=1.0, kspace_accuracy=1e-5)
    eng = capi.Engine(capi.default_params(**kw))
    eng.register_replica("eth", 1, d)
    f, e, w, info = eng.debug_compute("eth", 1, use_shake=True)
    o = po.Oracle(d, po.default_params(**kw))
    o.setup(use_shake=True)
    fo, eo, wo = o.compute()
    assert info["nclus"] == o.nclusters and info["npairs"] == o.npairs
    assert relerr(f, fo) < 1e-11
    for part in range(8):
        assert np.abs(w[part] - wo[part]).max() < 1e-10 * max(1.0, np.abs(wo[part]).max()), capi.PARTS[part]
    rates = np.array([1e-5, -2e-5, 3e-5, 1.5e-5, -0.5e-5, 2.5e-5])
    eng.set_state(3, "eth", 1, d["box"], d["x"], d["v"])
    pavg = eng.debug_run("eth", 1, 20, 1.0, 200.0, qp=3, nvt=True, use_shake=True, rates=rates, sample=True)
    box, x, v = eng.get_state(3, "eth", 1)
    o2 = po.Oracle(d, po.default_params(**kw))
    pavg_o, _ = o2.run(20, 1.0, 200.0, nvt=True, use_shake=True, rates=rates, sample=True)
    bo, xo, vo = o2.get_state()
    assert np.abs(box - bo).max() < 1e-12
    assert np.abs(x - xo).max() < 1e-9
    assert relerr(v, vo) < 1e-8
    assert relerr(pavg, pavg_o) < 1e-8
    eng.close()


@pytest.mark.parametrize("use_shake", [False, True])
def test_the_ewald_sum_instead_of_pppm(small_pe, use_shake):
    """kspace_style 0: the reciprocal part as the plain Ewald sum at the same accuracy (the default is PPPM, as in.set.lammps:36
    asks): static parts and a full evaluation against the oracle's Ewald sum, tolerances as for the default above."""
    from scema_amd import capi
    e = capi.Engine(capi.default_params(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=1e-5, kspace_style=0))
    e.register_replica("pe", 1, small_pe)
    f, en, w, info = e.debug_compute("pe", 1, use_shake=use_shake)
    o = oracle_small(small_pe, kspace_pppm=0)
    o.setup(use_shake)
    fo, eo, wo = o.compute()
    assert info["nk"] == o.nkvec > 0
    assert relerr(f, fo) < 1e-10 and relerr(en, eo) < 1e-10 and relerr(w, wo) < 1e-9
    if use_shake:
        lens = small_pe["box"][3:6] - small_pe["box"][:3]
        strain = np.array([-0.3 * 1.2e-3 * lens[0], -0.3 * 1.2e-3 * lens[1], 1.2e-3 * lens[2], 5e-5 * lens[2], -3e-5 * lens[1], 2e-5 * lens[0]])
        got = np.array(e.strain_batch([capi.make_sim(7, "pe", 1, strain, nss=20, most_recent=capi.QP_NONE)])[0].stress[:])
        o2 = oracle_small(small_pe, kspace_pppm=0)
        exp, nts = o2.eval(strain, 2.0, 300.0, 1e-4, 20)
        assert nts == 10 and relerr(got, exp) < 1e-6
    e.close()


def test_coulomb_cutoff_beyond_the_lj_cutoff(small_pe):
    """pair_style lj/cut/coul/long with the coulomb cutoff the LARGER of the two: the pair kernel's general form (the reference's
    12 / 9 takes the form specialised for cut_coul <= cut_lj).  Static parts and a full evaluation against the oracle."""
    from scema_amd import capi
    kw = dict(cut_lj=4.0, cut_coul=5.0, skin=1.0, kspace_accuracy=1e-5)
    e = capi.Engine(capi.default_params(**kw))
    e.register_replica("pe", 1, small_pe)
    f, en, w, info = e.debug_compute("pe", 1, use_shake=True)
    o = oracle_small(small_pe, **kw)
    o.setup(True)
    fo, eo, wo = o.compute()
    assert info["npairs"] == o.npairs
    assert relerr(f, fo) < 1e-10 and relerr(en, eo) < 1e-10 and relerr(w, wo) < 1e-9
    lens = small_pe["box"][3:6] - small_pe["box"][:3]
    strain = np.array([-0.3 * 1.2e-3 * lens[0], -0.3 * 1.2e-3 * lens[1], 1.2e-3 * lens[2], 5e-5 * lens[2], -3e-5 * lens[1], 2e-5 * lens[0]])
    got = np.array(e.strain_batch([capi.make_sim(7, "pe", 1, strain, nss=20, most_recent=capi.QP_NONE)])[0].stress[:])
    o2 = oracle_small(small_pe, **kw)
    exp, nts = o2.eval(strain, 2.0, 300.0, 1e-4, 20)
    assert nts == 10 and relerr(got, exp) < 1e-6
    e.close()


def test_launch_groups_smaller_than_the_batch(small_pe):
    """max_batch = 2 over 5 simulations: three launch groups (2 + 2 + 1) give what one group gives, for two consecutive
    updates on persistent states -- the path the dogbone_file3D mesh takes (4 864 points x replicas exceed one launch
    group of 1 024, inputs_dogbone_file3D.json:36).  A failure in the LAST group puts the first groups' states back."""
    from scema_amd import capi
    KW = dict(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=1e-5)
    lens = small_pe["box"][3:6] - small_pe["box"][:3]
    def strains(u):
        return [np.array([-3e-4 * lens[0], -3e-4 * lens[1], (1.0e-3 + 2e-4 * k) * lens[2] * (1 if u == 0 else -1), 3e-5 * k * lens[2], 0.0, -2e-5 * k * lens[0]])
                for k in range(5)]
    res = {}
    for mb in (0, 2):
        eng = capi.Engine(capi.default_params(max_batch=mb, **KW))
        eng.register_replica("pe", 1, small_pe)
        out = []
        for u in range(2):
            sims = [capi.make_sim(300 + k, "pe", 1, s, nss=10, most_recent=capi.QP_NONE if u == 0 else None) for k, s in enumerate(strains(u))]
            arr = eng.strain_batch(sims)
            assert all(a.stress_updated for a in arr)
            out.append(np.array([list(a.stress) for a in arr]))
        if mb == 2:
            # third update: the fifth simulation (alone in the last group) blows up -> the whole update fails and the four
            # states advanced by the first two groups are put back: repeating update 2's continuation gives the same numbers
            before = eng.get_state(300, "pe", 1)
            sims = [capi.make_sim(300 + k, "pe", 1, s, nss=10) for k, s in enumerate(strains(0))]
            sims[4].timestep_length = 60.0
            with pytest.raises(capi.EngineError):
                eng.strain_batch(sims)
            after = eng.get_state(300, "pe", 1)
            assert np.array_equal(before[1], after[1]) and np.array_equal(before[2], after[2]) and np.array_equal(before[0], after[0])
        res[mb] = out
        eng.close()
    for u in range(2):
        err = np.abs(res[0][u] - res[2][u]).max() / np.abs(res[0][u]).max()
        assert err < 1e-9, (u, err)
