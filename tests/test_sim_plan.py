"""The planner that deals the simulations of one update() to the ranks (scema_amd/csrc/host/sim_plan.h) through the C ABI
(scema_plan_*): pure host arithmetic, no GPU.  Replaces the reference's i % n_md_batches (stmd_sync.h:583), which relies
on a shared file system for the last.<qp>.* states (stmd_problem.h:117-138); here a state lives on ONE GPU."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sims_for(qps, recent=None, rep=1):
    from scema_amd import capi
    out = []
    for k, q in enumerate(qps):
        mr = capi.QP_NONE if recent is None else recent[k]
        out.append(capi.make_sim(q, "g0", rep, np.zeros(6), most_recent=mr))
    return out


def test_fresh_balanced_batch_is_the_reference_round_robin():
    from scema_amd import capi
    d = capi.PlanDir()
    for world in (1, 2, 3, 8):
        owner, pos, cap, moves = d.update(sims_for(range(13)), world, commit=False)
        assert list(owner) == [i % world for i in range(13)]          # stmd_sync.h:583
        assert list(pos) == [i // world for i in range(13)]
        assert cap == (13 + world - 1) // world and len(moves) == 0


def test_states_stay_where_they_are_when_the_update_list_shrinks_or_reorders():
    """FE_problem.h:1330-1350 lists only the quadrature points that need MD: the same (qp, replica) must land on the rank
    that holds its state, whatever its position in the vector."""
    from scema_amd import capi
    d = capi.PlanDir()
    qps = list(range(12))
    owner0, _, _, _ = d.update(sims_for(qps), 4)
    home = dict(zip(qps, owner0))
    sub = [9, 2, 7, 4, 11, 0, 5]
    owner1, pos1, cap1, moves1 = d.update(sims_for(sub, recent=sub), 4)
    assert [int(o) for o in owner1] == [home[q] for q in sub] and len(moves1) == 0
    for r in range(4):   # result slots of a rank are dense, in vector order
        assert sorted(pos1[owner1 == r]) == list(range(int((owner1 == r).sum())))
    assert cap1 == max(int((owner1 == r).sum()) for r in range(4))


def test_branching_runs_where_the_source_state_lives():
    """most_recent_qp_id != qp_id (clustering, stmd_problem.h:116-120): the new quadrature point continues from another
    one's state, so it is planned on that state's rank and then owned there."""
    from scema_amd import capi
    d = capi.PlanDir()
    owner0, _, _, _ = d.update(sims_for([0, 1, 2, 3]), 2)
    owner1, _, _, moves = d.update(sims_for([10, 11], recent=[1, 2]), 2)
    assert list(owner1) == [owner0[1], owner0[2]] and len(moves) == 0
    owner2, _, _, _ = d.update(sims_for([10, 11], recent=[10, 11]), 2)
    assert list(owner2) == list(owner1)
    # "none" restarts from the registered init state (available everywhere): dealt like a fresh simulation
    owner3, _, _, _ = d.update(sims_for([0], recent=None), 2, commit=False)
    assert owner3[0] in (0, 1)
    # two branches from sources on the same rank: one of them moves (its state is copied over), the load is level
    owner4, _, _, moves4 = d.update(sims_for([20, 21], recent=[1, 3]), 2, commit=False)
    assert sorted(owner4) == [0, 1] and len(moves4) == 1 and moves4[0][1] == 1 and moves4[0][2] == 0


def test_ragged_costs_are_levelled_and_moves_are_reported():
    from scema_amd import capi
    rng = np.random.default_rng(3)
    d = capi.PlanDir()
    n, world = 64, 8
    qps = list(range(n))
    d.update(sims_for(qps), world)                                   # balanced first update: i % 8
    cost = 100.0 + 10.0 * rng.integers(1, 11, n)                     # nts 10..100 + nss 100
    owner, pos, cap, moves = d.update(sims_for(qps, recent=qps), world, cost=cost)
    load = np.array([cost[owner == r].sum() for r in range(world)])
    sticky = np.array([cost[np.arange(n) % world == r].sum() for r in range(world)])
    assert load.max() <= sticky.max()
    assert load.max() - load.min() <= cost.max()                     # no rank leads by more than one simulation
    for s, f, t in moves:
        assert f == s % world and t == owner[s] and f != t
    moved = {int(m[0]) for m in moves}
    assert all((owner[i] == i % world) != (i in moved) for i in range(n))
    # equal costs again: only what the new costs make necessary moves (ranks end within one simulation of each other),
    # and a third update with the same costs moves nothing
    owner2, _, _, moves2 = d.update(sims_for(qps, recent=qps), world)
    cnt = np.bincount(owner2, minlength=world)
    assert cnt.max() - cnt.min() <= 1 and len(moves2) == int((owner2 != owner).sum())
    owner3, _, _, moves3 = d.update(sims_for(qps, recent=qps), world)
    assert len(moves3) == 0 and list(owner3) == list(owner2)


WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from scema_amd import capi
sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from test_sim_plan import sims_for
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
d = capi.PlanDir()
rng = np.random.default_rng(11)          # same request sequence on every rank, as in the reference (every rank holds md_sims)
trace = []
qps = list(range(10))
for step in range(5):
    sub = sorted(rng.choice(qps, size=int(rng.integers(4, 10)), replace=False).tolist(), key=lambda q: (q * 7) % 10)
    cost = 100.0 + 10.0 * rng.integers(1, 11, len(sub))
    recent = sub if step else None
    owner, pos, cap, moves = d.update(sims_for(sub, recent=recent), world, cost=cost)
    trace += list(owner) + list(pos) + [cap, len(moves)] + [int(v) for v in moves.ravel()]
t = torch.tensor(trace, dtype=torch.int64)
parts = [torch.empty_like(t) for _ in range(world)]
dist.all_gather(parts, t)
assert all(torch.equal(p, t) for p in parts), "ranks computed different plans"
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
'''


def test_every_rank_computes_the_same_plan(tmp_path):
    """world_size 2 over gloo: no communication is needed to agree on the plan -- checked by comparing."""
    (tmp_path / "worker.py").write_text(WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", str(tmp_path / "worker.py"), ROOT]
    r = subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
