"""GPU: the N > 1 path of the engine itself -- two ranks (sharing the one GPU of the test box, the engine's host transport
over gloo) run an update sequence in which the update_list shrinks, reorders and branches (FE_problem.h:1330-1350,
stmd_problem.h:116-120) and a ragged batch forces a replica state to change rank.  Every stress must equal the
single-rank run's: a state that stays behind, or is restarted from init because its owner changed, would show here."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=1e-5)


def sequence(lens):
    """[(qp ids, most_recent ids, strains)] per update"""
    from scema_amd import capi
    def st(ezz, k):
        return np.array([-0.3 * ezz * lens[0], -0.3 * ezz * lens[1], ezz * lens[2], 2e-5 * k * lens[2], -1e-5 * k * lens[1], 0.0])
    u1 = ([0, 1, 2, 3, 4, 5], [capi.QP_NONE] * 6, [st(1.2e-3 + 1e-4 * k, k) for k in range(6)])
    # shrunk + reordered; qp 4 gets a long straining run (nts 40); qp 7 branches from qp 2's state
    u2 = ([4, 1, 5, 7], [4, 1, 5, 2], [st(7.0e-3, 1), st(1.0e-3, 2), st(-0.8e-3, 3), st(1.1e-3, 4)])
    u3 = ([7, 0, 1, 2, 3, 4, 5], [7, 0, 1, 2, 3, 4, 5], [st(0.9e-3 + 1e-4 * k, k) for k in range(7)])
    return [u1, u2, u3]


def run_sequence(eng, lens, rank=0, world=1):
    from scema_amd import capi
    out = []
    for qps, recent, strains in sequence(lens):
        sims = [capi.make_sim(q, "pe", 1, s, nss=10, most_recent=r) for q, r, s in zip(qps, recent, strains)]
        arr = eng.strain_batch(sims, rank=rank, world=world)
        assert all(a.stress_updated for a in arr)
        out.append(np.array([list(a.stress) for a in arr]))
    return out


WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch.distributed as dist
from scema_amd import capi, comm
from scema_amd.systems import build_pe
from test_gpu_multirank import KW, run_sequence
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
d = build_pe(2, 3, 5, jitter=0.05, seed=7)
d["box"][6:9] = [0.7, -0.4, 0.5]
eng = capi.Engine(capi.default_params(**KW))
comm.attach_gloo(eng, rank, world)
eng.register_replica("pe", 1, d)
lens = d["box"][3:6] - d["box"][:3]
out = run_sequence(eng, lens, rank, world)
st = eng.comm_stats()
owners = [eng.state_owner(q, "pe", 1) for q in range(8)]
held = [int(eng.has_state(q, "pe", 1)) for q in range(8)]
np.savez(sys.argv[2] + f".{rank}.npz", u1=out[0], u2=out[1], u3=out[2], stats=np.array([st["allgathers"], st["migrations"]]),
         owners=np.array(owners), held=np.array(held))
dist.barrier(); eng.close(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_follow_their_states_and_move_them_when_needed(tmp_path, small_pe, world):
    """two and four ranks sharing the test box's one GPU over the host transport (the box allows six processes on its card)"""
    from scema_amd import capi
    (tmp_path / "worker.py").write_text(WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(29547 + world), str(tmp_path / "worker.py"), ROOT, str(tmp_path / "out")]
    r = subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    eng = capi.Engine(capi.default_params(**KW))
    eng.register_replica("pe", 1, small_pe)
    lens = small_pe["box"][3:6] - small_pe["box"][:3]
    ref = run_sequence(eng, lens)
    eng.close()
    got = [np.load(str(tmp_path / "out") + f".{k}.npz") for k in range(world)]
    for k in range(world):
        for u, name in enumerate(("u1", "u2", "u3")):
            err = np.abs(got[k][name] - ref[u]).max() / np.abs(ref[u]).max()
            assert err < 1e-8, (k, name, err)          # FP64 atomics only; a lost state would be O(1)
        assert got[k]["stats"][0] == 3                  # ONE collective per update
        if world == 2:
            assert got[k]["stats"][1] >= 1              # the ragged second update moved a state
        assert list(got[k]["owners"]) == list(got[0]["owners"])   # the directory is the same everywhere
    # every state is held by exactly the rank recorded as its owner (qp 6 never existed)
    for q in (0, 1, 2, 3, 4, 5, 7):
        o = int(got[0]["owners"][q])
        assert o in range(world) and [int(got[k]["held"][q]) for k in range(world)] == [1 if k == o else 0 for k in range(world)]
    assert all(got[k]["held"][6] == 0 for k in range(world))


def test_without_a_communicator_a_remote_source_state_is_an_error(small_pe):
    """rank 0 of 2, nothing attached: its own share runs; a request that continues from a state rank 1 would hold says so."""
    from scema_amd import capi
    eng = capi.Engine(capi.default_params(**KW))
    eng.register_replica("pe", 1, small_pe)
    lens = small_pe["box"][3:6] - small_pe["box"][:3]
    st = np.array([-4e-4 * lens[0], -4e-4 * lens[1], 1.2e-3 * lens[2], 0, 0, 0])
    sims = [capi.make_sim(q, "pe", 1, st, nss=10, most_recent=capi.QP_NONE) for q in range(4)]
    out = eng.strain_batch(sims, rank=0, world=2)
    assert [o.stress_updated for o in out] == [1, 0, 1, 0]
    # qp 1 lives on rank 1 (recorded, not held here); a ragged batch that would pull it over cannot be served
    big = st * 6.0
    sims2 = [capi.make_sim(1, "pe", 1, st), capi.make_sim(3, "pe", 1, st), capi.make_sim(0, "pe", 1, st), capi.make_sim(9, "pe", 1, big, most_recent=3)]
    with pytest.raises(capi.EngineError, match="communicator"):
        eng.strain_batch(sims2, rank=0, world=2)
    # the failed call left the store as it was: qp 0 continues normally
    out3 = eng.strain_batch([capi.make_sim(0, "pe", 1, st), capi.make_sim(1, "pe", 1, st)], rank=0, world=2)
    assert [o.stress_updated for o in out3] == [1, 0]
    # this caller never settled its updates (no scatter_gathered, no settle_update): each next call let the one before stand -- and counted it
    assert eng.unsettled_updates() >= 1
    eng.close()


def test_callback_transport_takes_this_ranks_share_back_when_another_rank_failed(small_pe):
    """rank 0 of 2 without a communicator runs its share successfully; the caller's collective then shows that rank 1
    failed.  scema_md_scatter_gathered reports it AND this rank's share did not happen: the states it created are gone, the state it
    advanced stands where it stood, and the owner directory is uncommitted -- a retry of the same update then gives the same stresses."""
    import ctypes as C
    from scema_amd import capi
    eng = capi.Engine(capi.default_params(**KW))
    eng.register_replica("pe", 1, small_pe)
    lens = small_pe["box"][3:6] - small_pe["box"][:3]
    st = np.array([-4e-4 * lens[0], -4e-4 * lens[1], 1.2e-3 * lens[2], 0, 0, 0])

    def gathered_with(status_of_rank1):
        n = eng.local_result_doubles()
        mine = np.zeros(n)
        eng.copy_local_stress(mine.ctypes.data, False)
        g = np.stack([mine, mine.copy()])
        g[1, -2] = status_of_rank1
        return g

    # update 1: fresh batch; rank 1 "succeeds": the update stands
    sims = [capi.make_sim(q, "pe", 1, st, nss=10, most_recent=capi.QP_NONE) for q in range(4)]
    out = eng.strain_batch(sims, rank=0, world=2)
    eng.scatter_gathered(gathered_with(0.0), out)
    assert eng.has_state(0, "pe", 1) and eng.has_state(2, "pe", 1)
    x_before = eng.get_state(0, "pe", 1)[1].copy()
    first = np.array(list(out[0].stress))
    # update 2: rank 0's share runs, rank 1 "fails" (status 7): this rank's share goes back
    sims2 = [capi.make_sim(q, "pe", 1, st, nss=10, most_recent=q) for q in range(4)] + [capi.make_sim(8, "pe", 1, st, nss=10, most_recent=capi.QP_NONE)]
    out2 = eng.strain_batch(sims2, rank=0, world=2)
    mine_new = eng.last_plan(5)[0][4] == 0                                          # does the planner give the new point to this rank?
    assert out2[0].stress_updated and eng.has_state(8, "pe", 1) == mine_new         # (done, and waiting for the verdict)
    second = np.array(list(out2[0].stress))
    with pytest.raises(capi.EngineError, match="rank 1"):
        eng.scatter_gathered(gathered_with(7.0), out2)
    assert not eng.has_state(8, "pe", 1)                                            # the state the update created is gone
    assert np.array_equal(eng.get_state(0, "pe", 1)[1], x_before)                   # the state it advanced is where it was
    # the retry is the same update again: same stresses (to the FP64 atomics' summation noise), and it stands this time
    out3 = eng.strain_batch(sims2, rank=0, world=2)
    eng.scatter_gathered(gathered_with(0.0), out3)
    third = np.array(list(out3[0].stress))
    assert np.abs(third - second).max() < 1e-8 * np.abs(second).max() and np.abs(second - first).max() > 1e-6 * np.abs(first).max()
    assert eng.has_state(8, "pe", 1) == mine_new and not np.array_equal(eng.get_state(0, "pe", 1)[1], x_before)
    assert eng.unsettled_updates() == 0      # every update of this caller was settled by scatter_gathered
    eng.close()


_RCCL_ONE = r"""
import sys
import numpy as np
import torch                                           # first, as in bench.py: torch ships its own copies of the HIP runtime and of RCCL
assert torch.cuda.is_available()
torch.zeros(4, device="cuda").sum().item()             # torch has initialised the device and its streams
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from scema_amd import capi
from scema_amd.systems import build_pe
from test_gpu_multirank import KW
d = build_pe(2, 3, 5, jitter=0.05, seed=7); d["box"][6:9] = [0.7, -0.4, 0.5]
lens = d["box"][3:6] - d["box"][:3]
st = np.array([-4e-4 * lens[0], -4e-4 * lens[1], 1.2e-3 * lens[2], 0, 0, 0])
res = []
for with_comm in (False, True):
    eng = capi.Engine(capi.default_params(**KW))
    if with_comm:
        uid = eng.comm_unique_id()
        assert len(uid) == capi.COMM_ID_BYTES
        eng.comm_init_rccl(uid, 0, 1)
    eng.register_replica("pe", 1, d)
    out = eng.strain_batch([capi.make_sim(q, "pe", 1, st * (1 + 0.1 * q), nss=10, most_recent=capi.QP_NONE) for q in range(3)])
    res.append(np.array([list(o.stress) for o in out]))
    if with_comm:
        assert eng.comm_stats()["allgathers"] == 1
        eng.comm_destroy()
    eng.close()
assert np.abs(res[0] - res[1]).max() < 1e-9 * np.abs(res[0]).max()
print("RCCL-ONE-RANK-OK", float(np.abs(res[0]).max()))
"""


def test_rccl_calls_run_on_one_rank(tmp_path):
    """ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy inside the library, on the one GPU of the test
    box (world = 1): the stresses come back through the collective and equal the run without a communicator.  In a process
    of its own with torch loaded FIRST, as bench.py does it: torch ships its own copies of the HIP runtime and of RCCL, and
    the library has to work with whichever the loader already holds."""
    import subprocess
    import sys
    script = tmp_path / "rccl_one.py"
    script.write_text(_RCCL_ONE)
    r = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL-ONE-RANK-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


WORKER_EDGE = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch.distributed as dist
from scema_amd import capi, comm
from scema_amd.systems import build_pe
from test_gpu_multirank import KW
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
d = build_pe(2, 3, 5, jitter=0.05, seed=7)
d["box"][6:9] = [0.7, -0.4, 0.5]
eng = capi.Engine(capi.default_params(**KW))
comm.attach_gloo(eng, rank, world)
eng.register_replica("pe", 1, d)
lens = d["box"][3:6] - d["box"][:3]
st = np.array([-4e-4 * lens[0], -4e-4 * lens[1], 1.2e-3 * lens[2], 0, 0, 0])
log = {}
# (1) two fresh points, one per rank; then an update_list that holds only rank 0's point: rank 1 has nothing to run and
#     must still take part in the collective (ADVICE r2 high: no rank decides on its own whether to enter)
a = eng.strain_batch([capi.make_sim(q, "pe", 1, st, nss=10, most_recent=capi.QP_NONE) for q in (0, 1)], rank=rank, world=world)
b = eng.strain_batch([capi.make_sim(0, "pe", 1, st, nss=10)], rank=rank, world=world)
log["one_owner"] = np.array(list(b[0].stress)); log["one_owner_plan"] = eng.last_plan(1)[0]
# (2) a share that fails on ONE rank (qp 1 lives on rank 1; a 60 fs time step blows the replica up there): every rank gets
#     the error, nobody is left in the collective, the directory and the stored states stay usable
sims = [capi.make_sim(0, "pe", 1, st, nss=10), capi.make_sim(1, "pe", 1, st, nss=10)]
sims[1].timestep_length = 60.0
try:
    eng.strain_batch(sims, rank=rank, world=world)
    log["fail_msg"] = "no error"
except capi.EngineError as exc:
    log["fail_msg"] = str(exc)
log["owners_after_failure"] = np.array([eng.state_owner(q, "pe", 1) for q in (0, 1)])
c = eng.strain_batch([capi.make_sim(q, "pe", 1, st, nss=10) for q in (0, 1)], rank=rank, world=world)
log["after_failure"] = np.array([list(x.stress) for x in c])
# (3) a state-store edit made on ONE rank only changes that rank's plan: the handshake says so on every rank
if rank == 1:
    eng.drop_state(1, "pe", 1)
try:
    eng.strain_batch([capi.make_sim(q, "pe", 1, st, nss=10) for q in (1, 0, 2)], rank=rank, world=world)
    log["mismatch_msg"] = "no error"
except capi.EngineError as exc:
    log["mismatch_msg"] = str(exc)
st_ = eng.comm_stats()
log["stats"] = np.array([st_["allgathers"], st_["handshakes"]])
np.savez(sys.argv[2] + f".{rank}.npz", **log)
dist.barrier(); eng.close(); dist.destroy_process_group()
'''


def test_collective_edge_cases_one_owner_one_failure_one_mismatch(tmp_path, small_pe):
    """Two ranks over the host transport: an update that only one rank has work for, a share that fails on one rank, and
    plans that differ between the ranks -- none of them may leave a rank blocked in a collective."""
    from scema_amd import capi
    (tmp_path / "worker.py").write_text(WORKER_EDGE)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29571", str(tmp_path / "worker.py"), ROOT, str(tmp_path / "out")]
    r = subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    got = [np.load(str(tmp_path / "out") + f".{k}.npz") for k in range(2)]
    # single-rank run of the same sequence (the failed update leaves no trace)
    eng = capi.Engine(capi.default_params(**KW))
    eng.register_replica("pe", 1, small_pe)
    lens = small_pe["box"][3:6] - small_pe["box"][:3]
    st = np.array([-4e-4 * lens[0], -4e-4 * lens[1], 1.2e-3 * lens[2], 0, 0, 0])
    eng.strain_batch([capi.make_sim(q, "pe", 1, st, nss=10, most_recent=capi.QP_NONE) for q in (0, 1)])
    b = eng.strain_batch([capi.make_sim(0, "pe", 1, st, nss=10)])
    c = eng.strain_batch([capi.make_sim(q, "pe", 1, st, nss=10) for q in (0, 1)])
    ref_b = np.array(list(b[0].stress)); ref_c = np.array([list(x.stress) for x in c])
    eng.close()
    for k in range(2):
        assert int(got[k]["one_owner_plan"][0]) == 0
        assert np.abs(got[k]["one_owner"] - ref_b).max() < 1e-8 * np.abs(ref_b).max()
        msg = str(got[k]["fail_msg"])
        assert msg != "no error" and ("non-finite" in msg or "unstable" in msg or "rank 1 failed" in msg), msg
        assert list(got[k]["owners_after_failure"]) == [0, 1]
        assert np.abs(got[k]["after_failure"] - ref_c).max() < 1e-8 * np.abs(ref_c).max()
        assert "different plans" in str(got[k]["mismatch_msg"]), str(got[k]["mismatch_msg"])
        assert got[k]["stats"][0] == 4 and got[k]["stats"][1] == 5       # stress all-gathers: 4 updates got that far; handshakes: all 5
    assert "rank 1 failed" in str(got[0]["fail_msg"])


WORKER_ALLOC = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch.distributed as dist
from scema_amd import capi, comm
from scema_amd.systems import build_pe
from test_gpu_multirank import KW, sequence
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
d = build_pe(2, 3, 5, jitter=0.05, seed=7)
d["box"][6:9] = [0.7, -0.4, 0.5]
eng = capi.Engine(capi.default_params(**KW))
comm.attach_gloo(eng, rank, world)
eng.register_replica("pe", 1, d)
lens = d["box"][3:6] - d["box"][:3]
u1, u2, _ = sequence(lens)
def run(u):
    qps, recent, strains = u
    arr = eng.strain_batch([capi.make_sim(q, "pe", 1, s, nss=10, most_recent=r) for q, r, s in zip(qps, recent, strains)], rank=rank, world=world)
    return np.array([list(a.stress) for a in arr])
log = {"u1": run(u1)}
# the ragged second update moves a state; the rank that receives it cannot allocate it (injected): EVERY rank must get the error,
# before anything was sent, and nobody may be left inside the exchange or the collective
os.environ["SCEMA_MD_TEST_FAIL_INCOMING"] = "-1"
try:
    run(u2)
    log["msg"] = "no error"
except capi.EngineError as exc:
    log["msg"] = str(exc); log["code"] = exc.code if hasattr(exc, "code") else -1
del os.environ["SCEMA_MD_TEST_FAIL_INCOMING"]
st = eng.comm_stats()
log["stats_after_failure"] = np.array([st["allgathers"], st["handshakes"], st["migrations"]])
log["owners_after_failure"] = np.array([eng.state_owner(q, "pe", 1) for q in range(6)])
# the same update again: it runs, moves the state, and equals the single-rank run
log["u2"] = run(u2)
st = eng.comm_stats()
log["stats"] = np.array([st["allgathers"], st["handshakes"], st["migrations"]])
np.savez(sys.argv[2] + f".{rank}.npz", **log)
dist.barrier(); eng.close(); dist.destroy_process_group()
'''


def test_a_rank_that_cannot_take_a_migrating_state_fails_the_update_on_every_rank(tmp_path, small_pe):
    """the states that migrate in are allocated BEFORE the handshake, so the receiving rank's allocation failure travels in
    its status word: both ranks end the call with the same error, nothing was sent, the directory is unchanged and the retry works."""
    from scema_amd import capi
    (tmp_path / "worker.py").write_text(WORKER_ALLOC)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29583", str(tmp_path / "worker.py"), ROOT, str(tmp_path / "out")]
    r = subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    got = [np.load(str(tmp_path / "out") + f".{k}.npz") for k in range(2)]
    eng = capi.Engine(capi.default_params(**KW))
    eng.register_replica("pe", 1, small_pe)
    lens = small_pe["box"][3:6] - small_pe["box"][:3]
    ref = run_sequence(eng, lens)
    eng.close()
    msgs = [str(g["msg"]) for g in got]
    assert all(m != "no error" for m in msgs), msgs
    # one rank reports its own injected failure, the other hears of it through the handshake
    assert sum("injected" in m for m in msgs) == 1 and sum("before the update started" in m for m in msgs) == 1, msgs
    for k in range(2):
        assert list(got[k]["stats_after_failure"]) == [1, 2, 0]          # the failed update: a handshake, no exchange, no stress collective
        assert list(got[k]["owners_after_failure"]) == [0, 1, 0, 1, 0, 1]  # the directory of the first update stands
        assert got[k]["stats"][2] >= 1                                     # the retry moved the state
        for u, name in enumerate(("u1", "u2")):
            assert np.abs(got[k][name] - ref[u]).max() < 1e-8 * np.abs(ref[u]).max(), (k, name)


WORKER_MIG = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch.distributed as dist
from scema_amd import capi, comm
from scema_amd.systems import build_pe
from test_gpu_multirank import KW, sequence
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
what = sys.argv[3]
dist.init_process_group("gloo")
d = build_pe(2, 3, 5, jitter=0.05, seed=7)
d["box"][6:9] = [0.7, -0.4, 0.5]
eng = capi.Engine(capi.default_params(**KW))
comm.attach_gloo(eng, rank, world)
eng.register_replica("pe", 1, d)
lens = d["box"][3:6] - d["box"][:3]
u1, u2, _ = sequence(lens)
def run(u):
    qps, recent, strains = u
    arr = eng.strain_batch([capi.make_sim(q, "pe", 1, s, nss=10, most_recent=r) for q, r, s in zip(qps, recent, strains)], rank=rank, world=world)
    return np.array([list(a.stress) for a in arr])
log = {"u1": run(u1)}
# the ragged second update moves a state; AFTER the handshake a device copy of the exchange fails on the rank that sends (injected): the
# exchange must still be completed (the peer is inside its receive), and both ranks must learn of the error in the stress all-gather
os.environ["SCEMA_MD_TEST_FAIL_MIGRATE"] = what
try:
    run(u2)
    log["msg"] = "no error"
except capi.EngineError as exc:
    log["msg"] = str(exc)
del os.environ["SCEMA_MD_TEST_FAIL_MIGRATE"]
st = eng.comm_stats()
log["stats_after_failure"] = np.array([st["allgathers"], st["handshakes"], st["migrations"]])
log["owners_after_failure"] = np.array([eng.state_owner(q, "pe", 1) for q in range(6)])
log["u2"] = run(u2)
st = eng.comm_stats()
log["stats"] = np.array([st["allgathers"], st["handshakes"], st["migrations"]])
np.savez(sys.argv[2] + f".{rank}.npz", **log)
dist.barrier(); eng.close(); dist.destroy_process_group()
'''


def test_an_error_inside_the_exchange_of_states_keeps_to_the_protocol_on_the_host_transport(tmp_path, small_pe):
    """VERDICT r5 item 3, host transport, world 2: a device copy fails on ONE rank after the ranks agreed to exchange.  That rank still posts
    every send and receive of its moves, enters the stress all-gather with the error as its status word, and both ranks end the update
    with an error; nothing is committed and the retry moves the state and equals the single-rank run."""
    from scema_amd import capi
    (tmp_path / "worker.py").write_text(WORKER_MIG)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29589", str(tmp_path / "worker.py"), ROOT, str(tmp_path / "out"), "hostcopy"]
    r = subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    got = [np.load(str(tmp_path / "out") + f".{k}.npz") for k in range(2)]
    eng = capi.Engine(capi.default_params(**KW))
    eng.register_replica("pe", 1, small_pe)
    lens = small_pe["box"][3:6] - small_pe["box"][:3]
    ref = run_sequence(eng, lens)
    eng.close()
    msgs = [str(g["msg"]) for g in got]
    assert all(m != "no error" for m in msgs), msgs
    # at least one rank reports its own failed copy; a rank that took no part in a move hears of it in the stress all-gather
    assert any("while replica states migrate" in m for m in msgs), msgs
    assert all("while replica states migrate" in m or "during the update" in m for m in msgs), msgs
    for k in range(2):
        assert list(got[k]["stats_after_failure"][:2]) == [2, 2]          # the failed update: handshake AND stress collective, on both ranks
        assert list(got[k]["owners_after_failure"]) == [0, 1, 0, 1, 0, 1]  # the directory of the first update stands
        assert got[k]["stats"][2] >= 1                                     # the retry moved the state
        for u, name in enumerate(("u1", "u2")):
            assert np.abs(got[k][name] - ref[u]).max() < 1e-8 * np.abs(ref[u]).max(), (k, name)


_RCCL_SELF_MOVE = r"""
import os, sys
import numpy as np
import torch
assert torch.cuda.is_available()
torch.zeros(4, device="cuda").sum().item()
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from scema_amd import capi
from scema_amd.systems import build_pe
from test_gpu_multirank import KW
d = build_pe(2, 3, 5, jitter=0.05, seed=7); d["box"][6:9] = [0.7, -0.4, 0.5]
lens = d["box"][3:6] - d["box"][:3]
st = np.array([-4e-4 * lens[0], -4e-4 * lens[1], 1.2e-3 * lens[2], 0, 0, 0])
def first(eng):
    return eng.strain_batch([capi.make_sim(q, "pe", 1, st * (1 + 0.1 * q), nss=10, most_recent=capi.QP_NONE) for q in range(3)])
def second(eng):   # qp 0 and 1 continue from their own states, qp 5 branches from qp 2's
    out = eng.strain_batch([capi.make_sim(0, "pe", 1, st), capi.make_sim(1, "pe", 1, 0.5 * st), capi.make_sim(5, "pe", 1, st, most_recent=2)])
    return np.array([list(o.stress) for o in out])
# reference: no communicator
eng = capi.Engine(capi.default_params(**KW)); eng.register_replica("pe", 1, d); first(eng); ref = second(eng); ref3 = second(eng); eng.close()
# RCCL with one rank; every continued state travels through ncclSend / ncclRecv to this very rank
os.environ["SCEMA_MD_TEST_SELF_MOVE"] = "1"
eng = capi.Engine(capi.default_params(**KW))
eng.comm_init_rccl(eng.comm_unique_id(), 0, 1)
eng.register_replica("pe", 1, d)
first(eng)
assert eng.comm_stats()["migrations"] == 0          # fresh states: nothing to move
# every injected failure ends the update with an error and leaves no trace: the update after it equals the reference
for what, where in (("dbox", "out of device memory for the boxes"), ("upload", "upload of the boxes"), ("enqueue", "ncclSend"), ("group", "ncclGroupEnd")):
    os.environ["SCEMA_MD_TEST_FAIL_MIGRATE"] = what
    n_ag = eng.comm_stats()["allgathers"]
    try:
        second(eng)
        raise SystemExit(f"{what}: no error")
    except capi.EngineError as exc:
        assert where in str(exc), (what, str(exc))
    assert eng.comm_stats()["allgathers"] == n_ag + 1, what    # the rank entered the stress collective with its status word
    del os.environ["SCEMA_MD_TEST_FAIL_MIGRATE"]
got = second(eng)
assert eng.comm_stats()["migrations"] == 3, eng.comm_stats()
assert np.abs(got - ref).max() < 1e-9 * np.abs(ref).max(), np.abs(got - ref).max()
got3 = second(eng)
assert np.abs(got3 - ref3).max() < 1e-9 * np.abs(ref3).max()
eng.comm_destroy(); eng.close()
print("RCCL-SELF-MOVE-OK", float(np.abs(got - ref).max() / np.abs(ref).max()))
"""


def test_rccl_point_to_point_calls_and_their_error_paths_on_one_rank(tmp_path):
    """VERDICT r5 item 3, RCCL: with one rank no state ever changes GPU, so migrate_states' ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd
    had never run on the one-GPU test box.  SCEMA_MD_TEST_SELF_MOVE sends every continued state from this rank to itself through that very
    group: the stresses equal the run without a communicator, and each injected failure -- two before the exchange (device buffer, its
    upload), two inside it (a point-to-point call, the end of the group) -- ends the update with its error AFTER the rank has entered the
    stress all-gather, and leaves no trace."""
    script = tmp_path / "rccl_self.py"
    script.write_text(_RCCL_SELF_MOVE)
    r = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL-SELF-MOVE-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
