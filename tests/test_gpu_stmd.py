"""GPU: the whole drop-in path -- STMDSync.init/update (host mirror) -> scema_md_strain_batch ->
HIP kernels -- against the oracle's F8 evaluation plus the oracle's L3 arithmetic."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KW = dict(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=1e-5)


def _dirs(tmp_path):
    d = {k: str(tmp_path / k) for k in ("nanoscale_input", "nanoscale_output", "nanoscale_restart", "macroscale_output")}
    for v in d.values():
        os.makedirs(v, exist_ok=True)
    return d


def test_update_md_matches_oracle_and_restarts(small_pe, tmp_path):
    from scema_amd import capi, stmd
    from oracle import pyoracle as po
    dirs = _dirs(tmp_path)
    lens = small_pe["box"][3:6] - small_pe["box"][:3]
    s0 = np.array([2.0e6, -1.0e6, 3.0e6, 4.0e5, -2.0e5, 1.0e5])
    stmd.write_nanoscale_input(dirs["nanoscale_input"], "pe", 1, init_length=lens, init_stress_raw=s0,
                               stiff_file_order=np.zeros(36), nsheets=0, sysd=small_pe)
    eng = capi.Engine(capi.default_params(**KW))
    sync = stmd.STMDSync(eng)
    common = dict(nanostatelocin=dirs["nanoscale_input"], nanostatelocout=dirs["nanoscale_output"],
                  nanostatelocres=dirs["nanoscale_restart"], macrostatelocout=dirs["macroscale_output"],
                  mdtype=("pe",), nrepl=1, md_nsteps_sample=20, freq_checkpoint=1)
    sync.init(**common)
    eps = np.array([[-4e-4, -4e-4, 1.2e-3, 5e-5, -3e-5, 2e-5], [3e-4, -2e-4, -9e-4, 0.0, 1e-5, 0.0]])
    qps = [(11, capi.QP_NONE, 0, eps[0]), (12, capi.QP_NONE, 0, eps[1])]
    got = sync.update(1, 0.0, 1, qps)
    oracles = []
    for k in range(2):
        o = po.Oracle(small_pe, po.default_params(**KW))
        strain_len = po.prepare_strain(eps[k], np.eye(3), lens, hooke=False)
        sig, _ = o.eval(strain_len, 2.0, 300.0, 1e-4, 20)
        exp = po.store(sig[None], s0[None], np.eye(3)[None], False)
        assert np.abs(got[k] - exp).max() < 1e-6 * np.abs(sig).max()
        oracles.append(o)
    # checkpoint files lcts.<qp>.<mat>_<rep>.dump (stmd_problem.h:108-110,266-273)
    for q in (11, 12):
        assert os.path.exists(os.path.join(dirs["nanoscale_restart"], f"lcts.{q}.pe_1.dump"))
    # CSV log of the Angstrom-valued strain and the Pa stress (quirk 1 of SURVEY appendix C)
    assert os.path.exists(os.path.join(dirs["nanoscale_output"], "mddata_qpid11_repl1.csv"))
    # continue on the same engine ...
    got2 = sync.update(2, 1e-6, 1, [(11, 11, 0, 0.5 * eps[0]), (12, 12, 0, 0.5 * eps[1])])
    # ... and after a restart from the checkpoint (STMDSync::restart, stmd_sync.h:167-187)
    os.makedirs(os.path.join(dirs["nanoscale_input"], "restart"), exist_ok=True)
    eng2 = capi.Engine(capi.default_params(**KW))
    # checkpoint written at step 1 is the state BEFORE step 2
    import shutil
    sync.close(); eng.close()
    # re-create the step-1 checkpoints: they were overwritten at step 2, so replay from scratch
    eng3 = capi.Engine(capi.default_params(**KW))
    s3 = stmd.STMDSync(eng3)
    d3 = _dirs(tmp_path / "replay")
    stmd.write_nanoscale_input(d3["nanoscale_input"], "pe", 1, init_length=lens, init_stress_raw=s0,
                               stiff_file_order=np.zeros(36), nsheets=0, sysd=small_pe)
    c3 = dict(common, nanostatelocin=d3["nanoscale_input"], nanostatelocout=d3["nanoscale_output"],
              nanostatelocres=d3["nanoscale_restart"], macrostatelocout=d3["macroscale_output"])
    s3.init(**c3)
    s3.update(1, 0.0, 1, qps)
    os.makedirs(os.path.join(d3["nanoscale_input"], "restart"), exist_ok=True)
    for q in (11, 12):
        shutil.copy(os.path.join(d3["nanoscale_restart"], f"lcts.{q}.pe_1.dump"), os.path.join(d3["nanoscale_input"], "restart"))
    s4 = stmd.STMDSync(eng2)
    s4.init(**c3)
    got4 = s4.update(2, 1e-6, 1, [(11, 11, 0, 0.5 * eps[0]), (12, 12, 0, 0.5 * eps[1])])
    assert np.abs(got4 - got2).max() < 1e-9 * np.abs(got2).max()
    # and the oracle agrees on the second step too
    for k in range(2):
        strain_len = po.prepare_strain(0.5 * eps[k], np.eye(3), lens, hooke=False)
        sig, _ = oracles[k].eval(strain_len, 2.0, 300.0, 1e-4, 20)
        exp = po.store(sig[None], s0[None], np.eye(3)[None], False)
        assert np.abs(got2[k] - exp).max() < 1e-6 * np.abs(sig).max()


def test_branching_from_most_recent_qp(small_pe):
    """most_recent_id != id: load the other quadrature point's state, store under the own id
    (stmd_problem.h:116-138)."""
    from scema_amd import capi
    eng = capi.Engine(capi.default_params(**KW))
    eng.register_replica("pe", 1, small_pe)
    lens = small_pe["box"][3:6] - small_pe["box"][:3]
    st = np.array([-4e-4 * lens[0], -4e-4 * lens[1], 1.2e-3 * lens[2], 0, 0, 0])
    eng.strain_batch([capi.make_sim(1, "pe", 1, st, nss=10, most_recent=capi.QP_NONE)])
    a = eng.strain_batch([capi.make_sim(2, "pe", 1, st, nss=10, most_recent=1)])[0].stress[:]
    b = eng.strain_batch([capi.make_sim(1, "pe", 1, st, nss=10, most_recent=1)])[0].stress[:]
    assert np.allclose(a, b, rtol=1e-12)           # qp 2 branched from qp 1's state: same evaluation
    assert eng.has_state(2, "pe", 1)
    with pytest.raises(capi.EngineError):          # branching from a state that does not exist (assert in the reference)
        eng.strain_batch([capi.make_sim(3, "pe", 1, st, nss=10, most_recent=77)])
    with pytest.raises(capi.EngineError):          # unknown force field (stmd_problem.h:462-467)
        eng.strain_batch([capi.make_sim(3, "pe", 1, st, nss=10, most_recent=capi.QP_NONE, force_field="sw")])
    eng.close()


def test_init_material_matches_oracle(small_pe):
    """box lengths, initial stress and the stiffness tensor of an equilibrated replica, 13 MD runs in one
    GPU batch, against the CPU restatement of the same procedure (init_material_problem.h:196-300)."""
    from scema_amd import capi
    from oracle import pyoracle as po
    kw = dict(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=1e-5)
    eng = capi.Engine(capi.default_params(**kw))
    eng.register_replica("pe", 1, small_pe)
    args = dict(dt=2.0, temperature=300.0, nss=10, strain_ampl=0.005, strain_rate=2.5e-4)   # nsstrain = 10
    length, stress, stiff = eng.init_material("pe", 1, **args)
    o = po.Oracle(small_pe, po.default_params(**kw))
    lo, so, co = po.init_material(o, args["dt"], args["temperature"], args["nss"], args["strain_ampl"], args["strain_rate"])
    assert np.allclose(length, lo, rtol=1e-14)
    assert np.abs(stress - so).max() < 1e-6 * np.abs(so).max()
    assert np.abs(stiff - co).max() < 1e-6 * np.abs(co).max()
    assert np.allclose(stiff, stiff.T, rtol=0, atol=1e-9 * np.abs(stiff).max())    # C{ij}all is symmetrised
    eng.close()


def test_eqmd_equil_feeds_stmd_init(small_pe, tmp_path):
    """EQMDProblem::equil writes init.<mat>_<rep>.{length,stress,stiff}; STMDSync::init reads them back (the reference's
    init_material -> dealammps hand-over through nanoscale_input), and an update() runs on top."""
    from scema_amd import capi, stmd
    eng = capi.Engine(capi.default_params(**KW))
    eng.register_replica("pe", 1, small_pe)
    folder = str(tmp_path / "nano_in")
    os.makedirs(folder)
    length, stress, stiff = eng.init_material("pe", 1, nss=10, strain_rate=2.5e-4)
    stmd.eqmd_equil(eng, "pe", folder, 1, mdnss=10, mdss=2.5e-4)
    got_len = np.loadtxt(os.path.join(folder, "init.pe_1.length"))
    got_sig = np.loadtxt(os.path.join(folder, "init.pe_1.stress"))
    got_c = np.loadtxt(os.path.join(folder, "init.pe_1.stiff")).reshape(6, 6)
    # the two calls are separate MD batches with atomically accumulated forces: equal to summation order
    assert np.allclose(got_len, length, rtol=1e-15)
    assert np.abs(got_sig - stress).max() < 1e-9 * np.abs(stress).max()
    assert np.abs(got_c - stiff).max() < 1e-8 * np.abs(stiff).max()
    with pytest.raises(capi.EngineError, match="Force field"):
        stmd.eqmd_equil(eng, "pe", folder, 1, mdff="charmm")
    # hand-over: json + bin next to the three files, then the reference's init/update sequence
    import json
    json.dump({"relative_density": 0.95, "Nsheets": 0, "normal_vector": {}}, open(os.path.join(folder, "pe_1.json"), "w"))
    stmd.write_replica_file(os.path.join(folder, "init.pe_1.bin"), small_pe)
    out = str(tmp_path / "out")
    for d in ("nano_out", "nano_res", "macro_out"):
        os.makedirs(os.path.join(out, d))
    sync = stmd.STMDSync(eng)
    sync.init(md_nsteps_sample=10, nanostatelocin=folder, nanostatelocout=os.path.join(out, "nano_out"),
              nanostatelocres=os.path.join(out, "nano_res"), macrostatelocout=os.path.join(out, "macro_out"), mdtype=["pe"], nrepl=1)
    rd = sync.replica_data(0, 0)
    assert np.allclose(rd["init_length"], length, rtol=1e-15)
    assert np.abs(np.array(rd["init_stress"])[[0, 3, 4, 1, 5, 2]] - got_sig).max() < 1e-12 * np.abs(got_sig).max()
    eng.close()


def test_init_bin_may_be_a_lammps_restart(small_pe, tmp_path):
    """nanoscale_input/init.<mat>_<rep>.bin in LAMMPS' own binary restart layout (the file the reference's init_material
    leaves there, init_material_problem.h:209) is told apart from the replica container by its magic string and gives
    the same update() result."""
    from scema_amd import capi, stmd
    lens = small_pe["box"][3:6] - small_pe["box"][:3]
    s0 = np.zeros(6)
    eps = np.array([[-4e-4, -4e-4, 1.2e-3, 5e-5, -3e-5, 2e-5]])
    res = []
    for kind in ("container", "lammps"):
        dirs = _dirs(tmp_path / kind)
        stmd.write_nanoscale_input(dirs["nanoscale_input"], "pe", 1, init_length=lens, init_stress_raw=s0,
                                   stiff_file_order=np.zeros(36), nsheets=0, sysd=small_pe)
        if kind == "lammps":
            capi.write_lammps_restart(os.path.join(dirs["nanoscale_input"], "init.pe_1.bin"), small_pe, KW["cut_lj"], KW["cut_coul"])
        eng = capi.Engine(capi.default_params(**KW))
        sync = stmd.STMDSync(eng)
        sync.init(nanostatelocin=dirs["nanoscale_input"], nanostatelocout=dirs["nanoscale_output"],
                  nanostatelocres=dirs["nanoscale_restart"], macrostatelocout=dirs["macroscale_output"],
                  mdtype=("pe",), nrepl=1, md_nsteps_sample=20, freq_checkpoint=1)
        res.append(np.array(sync.update(1, 0.0, 1, [(5, capi.QP_NONE, 0, eps[0])])))
        sync.close(); eng.close()
    assert np.abs(res[0] - res[1]).max() < 1e-8 * np.abs(res[0]).max()


def test_states_travel_as_lammps_restart_files(small_pe, tmp_path):
    """With scema_stmd_set_lammps_state_files the run writes last.<qp>.* after every evaluation and its lcts.* checkpoints
    in LAMMPS' 17Nov16 binary restart layout (stmd_problem.h:258,268), and STMDSync::restart (stmd_sync.h:167-187) reads such
    files back: the continued run equals the uninterrupted one.  The files are also read by the second, plain-Python
    reader (oracle/lammps_restart.py)."""
    import shutil
    from scema_amd import capi, stmd
    from oracle import lammps_restart as lr
    dirs = _dirs(tmp_path)
    lens = small_pe["box"][3:6] - small_pe["box"][:3]
    s0 = np.zeros(6)
    stmd.write_nanoscale_input(dirs["nanoscale_input"], "pe", 1, init_length=lens, init_stress_raw=s0,
                               stiff_file_order=np.zeros(36), nsheets=0, sysd=small_pe)
    common = dict(nanostatelocin=dirs["nanoscale_input"], nanostatelocout=dirs["nanoscale_output"],
                  nanostatelocres=dirs["nanoscale_restart"], macrostatelocout=dirs["macroscale_output"],
                  mdtype=("pe",), nrepl=1, md_nsteps_sample=20, freq_checkpoint=1)
    eps = np.array([-4e-4, -4e-4, 1.2e-3, 5e-5, -3e-5, 2e-5])
    eng = capi.Engine(capi.default_params(**KW))
    sync = stmd.STMDSync(eng)
    sync.set_lammps_state_files(True)
    sync.init(**common)
    sync.update(1, 0.0, 1, [(11, capi.QP_NONE, 0, eps)])
    lcts = os.path.join(dirs["nanoscale_restart"], "lcts.11.pe_1.dump")
    last = os.path.join(dirs["nanoscale_output"], "last.11.pe_1.dump")
    for f in (lcts, last):
        info = capi.probe_lammps_restart(f)
        assert info.version == b"17 Nov 2016" and info.atom_style == b"full" and info.natoms == small_pe["natoms"]
        assert info.pair_style == b"lj/cut/coul/long"
    py = lr.read_restart(lcts)
    eng._natoms[("pe", 1)] = small_pe["natoms"]      # the replica was registered by STMDSync::init, not through this wrapper
    box1, x1, v1 = eng.get_state(11, "pe", 1)
    assert np.allclose([py["BOXLO"][0], py["BOXHI"][0], py["XY"]], [box1[0], box1[3], box1[6]], rtol=0, atol=1e-12)
    assert len(py["atoms"]) == small_pe["natoms"] and py["PAIR"] == "lj/cut/coul/long"
    # keep the step-1 files aside (step 2 overwrites them), continue in memory, then restart a fresh engine from the file
    os.makedirs(os.path.join(dirs["nanoscale_input"], "restart"), exist_ok=True)
    shutil.copy(lcts, os.path.join(dirs["nanoscale_input"], "restart"))
    got2 = sync.update(2, 1e-6, 1, [(11, 11, 0, 0.5 * eps)])
    sync.close(); eng.close()
    eng2 = capi.Engine(capi.default_params(**KW))
    sync2 = stmd.STMDSync(eng2)
    sync2.init(**common)
    got2r = sync2.update(2, 1e-6, 1, [(11, 11, 0, 0.5 * eps)])
    assert np.abs(got2r - got2).max() < 1e-9 * np.abs(got2).max()
    sync2.close(); eng2.close()
