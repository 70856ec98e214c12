"""init_material's equilibration schedule on the GPU (md_equil.hip + the production force kernels) against the oracle's
restatement (oracle/md_oracle.c: omd_minimize, omd_run_nh, omd_equilibrate) on identical inputs, through the C ABI.
SURVEY.md 8(f) row f-2; reference: lammps_scripts_opls/in.init.lammps:44-215, init_material_problem.h:167-210.
FP64 throughout; what differs between the two is summation order only, tolerances at each assert."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KW = dict(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=1e-5)


@pytest.fixture()
def eng():
    from scema_amd import capi
    e = capi.Engine(capi.default_params(**KW))
    yield e
    e.close()


def _oracle(d):
    from oracle import pyoracle as po
    return po.Oracle(d, po.default_params(shake_mass=0.0, **KW))


def _jittered():
    from scema_amd.systems import build_pe
    d = build_pe(2, 3, 5, jitter=0.08, seed=3)
    d["box"][6:9] = [0.5, -0.3, 0.4]
    return d


def test_steepest_descent_follows_the_oracle_step_for_step(eng):
    d = _jittered()
    eng.register_replica("pe", 1, d)
    for maxiter in (1, 4, 25):
        eng.set_state(0, "pe", 1, d["box"], d["x"], d["v"])
        r = eng.minimize("pe", 1, 0, etol=0.0, ftol=0.0, maxiter=maxiter)
        o = _oracle(d)
        ro = o.minimize(etol=0.0, ftol=0.0, maxiter=maxiter)
        assert (r["stop"], r["iterations"], r["evaluations"]) == (ro["stop"], ro["iterations"], ro["evaluations"])
        assert abs(r["e_initial"] - ro["e_initial"]) < 1e-9 * abs(ro["e_initial"])
        assert abs(r["e_final"] - ro["e_final"]) < 1e-9 * max(1.0, abs(ro["e_final"]))
        xo = o.get_state()[1]
        assert np.abs(eng.get_state(0, "pe", 1)[1] - xo).max() < 1e-9       # the same line-search decisions, the same path
    # the reference's criteria (minimize 1.0e-7 1.0e-11): the same stop reason at the same point
    eng.set_state(0, "pe", 1, d["box"], d["x"], d["v"])
    r = eng.minimize("pe", 1, 0, etol=1e-7, ftol=1e-11, maxiter=300)
    o = _oracle(d)
    ro = o.minimize(etol=1e-7, ftol=1e-11, maxiter=300)
    assert r["stop"] == ro["stop"] and abs(r["iterations"] - ro["iterations"]) <= 1
    assert abs(r["e_final"] - ro["e_final"]) < 1e-6 * abs(ro["e_final"]) and r["e_final"] < r["e_initial"] - 100.0


def test_two_replicas_minimise_independently_in_one_batch(eng):
    """the line search of every replica is decided on the device: a batch gives what each replica gives alone"""
    from scema_amd import capi
    d = _jittered()
    eng.register_replica("pe", 1, d)
    eng.set_state(0, "pe", 1, d["box"], d["x"], d["v"])
    r1 = eng.minimize("pe", 1, 0, etol=0.0, ftol=0.0, maxiter=6)
    x1 = eng.get_state(0, "pe", 1)[1]
    o = _oracle(d)
    o.minimize(etol=0.0, ftol=0.0, maxiter=6)
    assert np.abs(x1 - o.get_state()[1]).max() < 1e-9 and r1["iterations"] == 6


@pytest.mark.parametrize("npt", [False, True])
def test_nose_hoover_run_with_ramp_matches_the_oracle(eng, small_pe, npt):
    eng.register_replica("pe", 1, small_pe)
    o = _oracle(small_pe)
    o.velocity_create(150.0, seed=5)
    box, x, v = o.get_state()
    eng.set_state(0, "pe", 1, box, x, v)
    n, dt = 60, 0.5
    lav = eng.run_nh("pe", 1, 0, n, dt, 150.0, 260.0, npt=npt, p_target=1.0, p_period=100.0, average_lengths=True)
    lavo, _ = o.run_nh(n, dt, 150.0, 260.0, npt=npt, p_target=1.0, p_period=100.0, average_lengths=True)
    bo, xo, vo = o.get_state()
    bg, xg, vg = eng.get_state(0, "pe", 1)
    assert np.abs(bg - bo).max() < 1e-10 and np.abs(lav - lavo).max() < 1e-10
    assert np.abs(xg - xo).max() < 1e-9 and np.abs(vg - vo).max() < 1e-10
    if npt:
        assert abs((bg[3] - bg[0]) / (box[3] - box[0]) - 1.0) > 1e-4       # the box really moved
        assert np.allclose(bg[6:] / box[6:], (bg[3] - bg[0]) / (box[3] - box[0]), rtol=1e-12)   # tilts scale with the lengths


def test_a_long_barostatted_run_is_issued_in_segments_without_a_seam(eng, small_pe):
    """the cell grid of a segment holds for +-2 % in the box lengths; a run longer than a segment continues thermostat,
    barostat, ramp and averages across the seam (the k-space setup is the one of the run's start)"""
    eng.register_replica("pe", 1, small_pe)
    o = _oracle(small_pe)
    o.velocity_create(200.0, seed=9)
    box, x, v = o.get_state()
    eng.set_state(0, "pe", 1, box, x, v)
    n, dt = 620, 0.25                        # segments of 250 steps: 250 + 250 + 120
    lav = eng.run_nh("pe", 1, 0, n, dt, 200.0, 230.0, npt=True, p_target=1.0, p_period=200.0, average_lengths=True)
    lavo, _ = o.run_nh(n, dt, 200.0, 230.0, npt=True, p_target=1.0, p_period=200.0, average_lengths=True)
    bo, xo, vo = o.get_state()
    bg, xg, vg = eng.get_state(0, "pe", 1)
    assert np.abs(bg - bo).max() < 1e-8 and np.abs(lav - lavo).max() < 1e-8
    assert np.abs(xg - xo).max() < 1e-6      # 620 steps of a chaotic system: round-off has grown, nothing else


def test_whole_schedule_matches_the_oracle(eng):
    """in.init.lammps end to end with nsinit = 6 (6 + 6 + 30 + 6 + 12 + 120 + 12 + 6 steps after the minimisation)"""
    from scema_amd.systems import build_pe
    d = build_pe(2, 3, 5, jitter=0.03, seed=4)
    d["box"][6:9] = [0.3, 0.2, -0.4]
    eng.register_replica("pe", 1, d)
    length, info = eng.equilibrate("pe", 1, 6, 0.5, 250.0, seed=1234)
    o = _oracle(d)
    lo, io = o.equilibrate(6, 0.5, 250.0, seed=1234)
    assert info["iterations"] == int(io[0]) and info["evaluations"] == int(io[1])
    assert abs(info["e_final"] - io[3]) < 1e-8 * abs(io[3])
    assert np.abs(length - lo).max() < 1e-8
    bo, xo, vo = o.get_state()
    from scema_amd import capi
    bg, xg, vg = eng.get_state(capi.QP_NONE, "pe", 1)
    assert np.abs(bg - bo).max() < 1e-8 and np.abs(xg - xo).max() < 1e-6 and np.abs(vg - vo).max() < 1e-7
    # the equilibrated state is now the replica's initial state: a stress evaluation starts from it
    lens = bg[3:6] - bg[:3]
    out = eng.strain_batch([capi.make_sim(0, "pe", 1, np.array([1e-3 * lens[0], 0, 0, 0, 0, 0]), nss=10, dt=0.5, most_recent=capi.QP_NONE)])
    assert np.isfinite(np.array(out[0].stress[:])).all()


def test_eqmd_equil_from_a_data_file_to_the_files_stmd_init_reads(tmp_path):
    """EQMDProblem::equil as init_material.cc calls it: <slocin>/<mat>_<rep>.data -> in.init.lammps schedule ("Compute state
    data") -> init.<mat>_<rep>.bin + .length/.stress/.stiff; a second call finds the state ("Reuse of state data") and does not
    equilibrate again; STMDSync::init reads everything back."""
    import json
    import os
    from scema_amd import capi, stmd
    from scema_amd.systems import build_pe, write_lammps_data
    d = build_pe(2, 3, 5, jitter=0.03, seed=4)
    slocin, folder = str(tmp_path / "data"), str(tmp_path / "nano_in")
    os.makedirs(slocin); os.makedirs(folder)
    write_lammps_data(os.path.join(slocin, "pe_1.data"), d)
    e = capi.Engine(capi.default_params(**KW))
    base = stmd.eqmd_equil_full(e, "pe", slocin, folder, 1, mdts=0.5, mdtem=250.0, mdnss=10, mdnse=4, mdss=2.5e-4)
    assert os.path.exists(base + ".bin")
    length = np.loadtxt(base + ".length")
    # the same schedule on the oracle, from the same file contents
    o = _oracle(d)
    lo, _ = o.equilibrate(4, 0.5, 250.0, seed=1234)
    assert np.abs(length - lo).max() < 1e-6
    b1 = open(base + ".bin", "rb").read()
    stmd.eqmd_equil_full(e, "pe", slocin, folder, 1, mdts=0.5, mdtem=250.0, mdnss=10, mdnse=4, mdss=2.5e-4)   # reuse
    assert open(base + ".bin", "rb").read() == b1
    json.dump({"relative_density": 0.95, "Nsheets": 0, "normal_vector": {}}, open(os.path.join(folder, "pe_1.json"), "w"))
    out = str(tmp_path / "out")
    for sub in ("nano_out", "nano_res", "macro_out"):
        os.makedirs(os.path.join(out, sub))
    e2 = capi.Engine(capi.default_params(**KW))      # a fresh engine: topology and state both come from init.pe_1.bin
    sync = stmd.STMDSync(e2)
    sync.init(md_nsteps_sample=10, nanostatelocin=folder, nanostatelocout=os.path.join(out, "nano_out"),
              nanostatelocres=os.path.join(out, "nano_res"), macrostatelocout=os.path.join(out, "macro_out"), mdtype=["pe"], nrepl=1)
    assert np.allclose(sync.replica_data(0, 0)["init_length"], length, rtol=1e-12)
    bg = e2.get_state(capi.QP_NONE, "pe", 1)[0]
    assert np.allclose(bg[3:6] - bg[:3], length, rtol=1e-12)      # the equilibrated box travelled with the file
    e.close(); e2.close()


def test_barostatted_run_with_the_mesh_solver(eng, small_pe):
    """fix npt with kspace_style pppm: the influence function follows the box every step, the grid and g_ewald of the run's start
    carry across segment seams"""
    from scema_amd import capi
    from oracle import pyoracle as po
    e = capi.Engine(capi.default_params(kspace_style=1, **KW))
    e.register_replica("pe", 1, small_pe)
    o = po.Oracle(small_pe, po.default_params(shake_mass=0.0, kspace_pppm=1, **KW))
    o.velocity_create(150.0, seed=5)
    box, x, v = o.get_state()
    e.set_state(0, "pe", 1, box, x, v)
    e.run_nh("pe", 1, 0, 300, 0.5, 150.0, 200.0, npt=True, p_target=1.0, p_period=100.0)     # 250 + 50: one seam
    o.run_nh(300, 0.5, 150.0, 200.0, npt=True, p_target=1.0, p_period=100.0)
    bo, xo, _ = o.get_state()
    bg, xg, _ = e.get_state(0, "pe", 1)
    assert np.abs(bg - bo).max() < 1e-8 and np.abs(xg - xo).max() < 1e-6
    e.close()
