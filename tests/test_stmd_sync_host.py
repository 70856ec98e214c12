"""Host layer (STMDSync mirror) on CPU: the reference's own fake backend "approximate md with hookes
law" (stmd_problem.h:479-483, docs/configuration.md:16) drives the whole L3 plumbing --
prepare -> execute -> all-gather -> store -- and is compared with the oracle's restatement."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _setup(tmp_path, nrepl=2):
    import __graft_entry__ as g
    g.build()
    from scema_amd import stmd
    gold = json.load(open(os.path.join(GOLD, "init_sic_1_stiff.json")))
    C1 = np.array(gold["stiff_file_order"])
    rng = np.random.default_rng(1)
    reps = []
    nin = str(tmp_path / "nanoscale_input")
    for r in range(nrepl):
        Cr = C1 * (1.0 + 0.1 * r)
        normal = np.array([0.2 + 0.3 * r, 1.0, -0.4]); normal /= np.linalg.norm(normal)
        s0 = rng.normal(0, 1e6, 6)
        L0 = np.array([40.0 + r, 41.0, 42.0])
        stmd.write_nanoscale_input(nin, "g0", r + 1, init_length=L0, init_stress_raw=s0, stiff_file_order=Cr,
                                   relative_density=0.9 + 0.01 * r, nsheets=1, normal=normal)
        reps.append(dict(C=Cr, normal=normal, s0=s0, L0=L0))
    for d in ("nanoscale_output", "nanoscale_restart", "macroscale_output"):
        os.makedirs(tmp_path / d, exist_ok=True)
    return nin, reps


def _expected(reps, strains, cg=(1.0, 0.0, 0.0)):
    from oracle import pyoracle as po
    out = []
    for eps in strains:
        sig, s0s, Rs = [], [], []
        for rp in reps:
            R = po.rotation_tensor(rp["normal"], cg)
            e_rep = po.prepare_strain(eps, R, rp["L0"], hooke=True)
            sig.append(po.hooke(rp["C"], e_rep)); s0s.append(rp["s0"]); Rs.append(R)
        out.append(po.store(np.array(sig), np.array(s0s), np.array(Rs), True))
    return np.array(out)


def test_hooke_update_matches_oracle(tmp_path):
    from scema_amd import stmd
    nin, reps = _setup(tmp_path)
    s = stmd.STMDSync(None)
    s.init(nanostatelocin=nin, nanostatelocout=str(tmp_path / "nanoscale_output"), nanostatelocres=str(tmp_path / "nanoscale_restart"),
           macrostatelocout=str(tmp_path / "macroscale_output"), nrepl=2, approx_md_with_hookes_law=True)
    rng = np.random.default_rng(4)
    strains = rng.normal(0, 1e-3, (5, 6))
    got = s.update(3, 1.5e-6, 1, [(10 + i, 10 + i, 0, strains[i]) for i in range(5)])
    exp = _expected(reps, strains)
    assert np.allclose(got, exp, rtol=1e-13, atol=1e-6)
    # replica metadata as loaded (file order -> raw order, rho = relative_density*1000)
    rd = s.replica_data(0, 1)
    assert np.allclose(rd["init_length"], reps[1]["L0"]) and np.allclose(rd["init_stress"], reps[1]["s0"])
    assert abs(rd["rho"] - 910.0) < 1e-9
    # average_replica_data wrote macroscale_output/init.g0.{stiff,density} (stmd_sync.h:455-489)
    from oracle import pyoracle as po
    avgC = sum(po.rotate_sym4(rp["C"], po.rotation_tensor(rp["normal"], (1, 0, 0))) for rp in reps) / 2
    wrote = np.array([float(l) for l in open(tmp_path / "macroscale_output" / "init.g0.stiff")])
    assert np.allclose(wrote, avgC, rtol=1e-12)
    assert abs(float(open(tmp_path / "macroscale_output" / "init.g0.density").read()) - 905.0) < 1e-9
    # per-evaluation CSV rows (stmd_problem.h:394-456): header once, one row per (qp, replica)
    csv = open(tmp_path / "nanoscale_output" / "mddata_qpid10_repl1.csv").read().strip().split("\n")
    assert csv[0].startswith("qp_id,material_id,time_id,temperature,strain_rate,force_field,replica_id,strain_00,strain_01")
    assert len(csv) == 2 and csv[1].startswith("10,g0,3-1,300,0.0001,opls,1,")


def test_unknown_force_field_is_an_error(tmp_path):
    from scema_amd import stmd, capi
    nin, _ = _setup(tmp_path, nrepl=1)
    s = stmd.STMDSync(None)
    s.init(nanostatelocin=nin, nrepl=1, approx_md_with_hookes_law=True, md_force_field="sw")
    with pytest.raises(capi.EngineError):
        s.update(1, 0.0, 1, [(0, 0, 0, np.zeros(6))])


def test_md_mode_without_gpu_fails_loudly(tmp_path):
    from scema_amd import stmd, capi
    nin, _ = _setup(tmp_path, nrepl=1)
    s = stmd.STMDSync(None)
    with pytest.raises(capi.EngineError):
        s.init(nanostatelocin=nin, nrepl=1, approx_md_with_hookes_law=False)


def test_missing_replica_json_is_an_error(tmp_path):
    from scema_amd import stmd, capi
    nin, _ = _setup(tmp_path, nrepl=1)
    s = stmd.STMDSync(None)
    with pytest.raises(capi.EngineError):
        s.init(nanostatelocin=nin, nrepl=2, approx_md_with_hookes_law=True)   # g0_2.json absent (stmd_sync.h:288-292)


WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from scema_amd import stmd
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
nin = sys.argv[2]
s = stmd.STMDSync(None, rank, world, stmd.torch_allgather(None, rank, world))
nrepl = int(sys.argv[5]) if len(sys.argv) > 5 else 2
s.init(nanostatelocin=nin, nrepl=nrepl, approx_md_with_hookes_law=True)
strains = np.load(sys.argv[3])
got = s.update(1, 0.0, 1, [(i, i, 0, strains[i]) for i in range(len(strains))])
np.save(sys.argv[4] + f".{rank}.npy", got)
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 8])
def test_ranks_share_stresses_with_one_allgather(tmp_path, world):
    """world_size 2 and 8 (one node of MI355X, BASELINE config 4) over gloo: simulations are dealt i % world
    (stmd_sync.h:583), one all-gather returns 6 doubles per simulation (replaces share_stresses,
    stmd_sync.h:620-726); every rank ends with the same update_stress as the single-rank run."""
    nin, reps = _setup(tmp_path)
    rng = np.random.default_rng(9)
    strains = rng.normal(0, 1e-3, (7 if world == 2 else 21, 6))      # not a multiple of the world size: ragged shards
    np.save(tmp_path / "strains.npy", strains)
    (tmp_path / "worker.py").write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(29517 + world), str(tmp_path / "worker.py"), ROOT, nin, str(tmp_path / "strains.npy"), str(tmp_path / "out")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    exp = _expected(reps, strains)
    for rank in range(world):
        got = np.load(str(tmp_path / "out") + f".{rank}.npy")
        assert np.allclose(got, exp, rtol=1e-13, atol=1e-6)


def test_one_simulation_on_two_ranks_does_not_deadlock(tmp_path):
    """an update whose every simulation lands on ONE rank (a one-point update_list with one replica)
    must still be collective -- whether the all-gather runs is decided by (world, mode), never by a rank's own results."""
    nin, reps = _setup(tmp_path, nrepl=1)
    rng = np.random.default_rng(11)
    strains = rng.normal(0, 1e-3, (1, 6))
    np.save(tmp_path / "strains.npy", strains)
    (tmp_path / "worker.py").write_text(WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", str(tmp_path / "worker.py"), ROOT, nin, str(tmp_path / "strains.npy"), str(tmp_path / "out"), "1"]
    r = subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    exp = _expected(reps, strains)
    for rank in range(2):
        assert np.allclose(np.load(str(tmp_path / "out") + f".{rank}.npy"), exp, rtol=1e-13, atol=1e-6)
