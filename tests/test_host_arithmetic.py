"""Golden vectors for the host-side (L3) arithmetic, SURVEY.md §8(c) pin 1: hand-derivable values
for rotate_tensor / compute_rotation_tensor (math_calc.h:23-71), the length scaling
(stmd_sync.h:552-557), the nts rule and %.6e rounding (stmd_problem.h:229-244), Hooke sigma=C:eps
on the shipped init.sic_1.stiff values (stmd_problem.h:386-392), replica averaging with
init-stress subtraction (stmd_sync.h:878-922) and the raw-order vs file-order permutation."""
import json
import os

import numpy as np

from oracle import pyoracle as po

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_nts_rule():
    # uniaxial 0.0018, rate 1e-4, dt 2 -> 9 steps -> 10
    assert po.nts([0.0018, 0, 0, 0, 0, 0], 1e-4, 2.0) == 10
    # 0.009 at 2e-4 -> 22.5 steps -> 30
    assert po.nts([0.009, 0, 0, 0, 0, 0], 2e-4, 2.0) == 30
    # floor of 10
    assert po.nts([1e-9, 0, 0, 0, 0, 0], 1e-4, 2.0) == 10
    # Frobenius norm counts shears twice: |e| = sqrt(2)*0.002 -> 14.14 steps -> 20
    assert po.nts([0, 0, 0, 0.002, 0, 0], 1e-4, 2.0) == 20
    # 0.016 -> 80 steps exactly is a ceil hazard; 0.0161 -> 80.5 -> 90
    assert po.nts([0, 0, 0.0161, 0, 0, 0], 1e-4, 2.0) == 90


def test_rate_rounding():
    assert po.round_rate(1.23456789e-5) == 1.234568e-05
    assert po.round_rate(-6.000000049e-5) == -6.0e-05
    assert po.round_rate(0.0) == 0.0


def test_rotation_tensor_golden():
    # rotating x onto y: K = [[0,-1,0],[1,0,0],[0,0,0]], a.b=0 -> R = I + K + K^2
    R = po.rotation_tensor([1, 0, 0], [0, 1, 0])
    assert np.allclose(R, [[0, -1, 0], [1, 0, 0], [0, 0, 1]], atol=1e-15)
    # a == b -> identity
    assert np.allclose(po.rotation_tensor([0, 0, 1], [0, 0, 1]), np.eye(3), atol=1e-15)
    # general: R a = b and R orthogonal
    a = np.array([1.0, 2.0, -0.5]); a /= np.linalg.norm(a)
    b = np.array([-0.3, 0.4, 1.2]); b /= np.linalg.norm(b)
    R = po.rotation_tensor(a, b)
    assert np.allclose(R @ a, b, atol=1e-14)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-14)


def test_rotate_tensor_golden():
    R = po.rotation_tensor([1, 0, 0], [0, 1, 0])
    # raw order xx,yy,zz,xy,xz,yz ; 90 deg about z: xx<->yy, xy->-xy, xz->yz, yz->-xz
    out = po.rotate_sym2([1, 2, 3, 4, 5, 6], R)
    assert np.allclose(out, [2, 1, 3, -4, -6, 5], atol=1e-14)


def test_prepare_strain_length_scaling():
    eps = np.array([1e-3, 2e-3, 3e-3, 4e-4, 5e-4, 6e-4])
    L0 = np.array([10.0, 20.0, 30.0])
    out = po.prepare_strain(eps, np.eye(3), L0, hooke=False)
    # diag x own length ; xy x Lz ; yz x Lx ; xz x Ly   (stmd_sync.h:553-556)
    assert np.allclose(out, [1e-2, 4e-2, 9e-2, 4e-4 * 30, 5e-4 * 20, 6e-4 * 10], rtol=1e-15)
    assert np.allclose(po.prepare_strain(eps, np.eye(3), L0, hooke=True), eps)
    # rotation uses transpose(rotam): common ground -> replica frame
    R = po.rotation_tensor([0, 1, 0], [1, 0, 0])
    out = po.prepare_strain(eps, R, [1, 1, 1], hooke=False)
    assert np.allclose(out, po.rotate_sym2(eps, R.T))


def test_hooke_on_shipped_stiffness():
    """sigma = C : eps with the reference's own 36-entry tensor
    (examples/streched_polyhedron/nanoscale_input/init.sic_1.stiff, copied as a data fixture)."""
    C = np.array(json.load(open(os.path.join(GOLD, "init_sic_1_stiff.json")))["stiff_file_order"])
    assert C.shape == (36,)
    eps = np.array([1e-3, -2e-4, 5e-4, 3e-4, -1e-4, 2e-4])
    out = po.hooke(C, eps)
    # independent numpy restatement: full 3x3x3x3 contraction
    fidx = {(0, 0): 0, (0, 1): 1, (0, 2): 2, (1, 1): 3, (1, 2): 4, (2, 2): 5}
    ridx = {(0, 0): 0, (1, 1): 1, (2, 2): 2, (0, 1): 3, (0, 2): 4, (1, 2): 5}
    f = lambda a, b: fidx[(min(a, b), max(a, b))]
    r = lambda a, b: ridx[(min(a, b), max(a, b))]
    exp = np.zeros(6)
    for k in range(3):
        for l in range(k, 3):
            exp[r(k, l)] = sum(C[f(k, l) * 6 + f(m, n)] * eps[r(m, n)] for m in range(3) for n in range(3))
    assert np.allclose(out, exp, rtol=1e-14)
    # first entry by hand: C0000 e00 + 2 C0001 e01 + 2 C0002 e02 + C0011 e11 + 2 C0012 e12 + C0022 e22
    hand = C[0] * eps[0] + 2 * C[1] * eps[3] + 2 * C[2] * eps[4] + C[3] * eps[1] + 2 * C[4] * eps[5] + C[5] * eps[2]
    assert abs(out[0] - hand) < 1e-9 * abs(hand)


def test_store_replica_average():
    s = np.array([[10.0, 20, 30, 1, 2, 3], [14.0, 24, 34, 3, 4, 5]])
    s0 = np.array([[1.0, 1, 1, 0, 0, 0], [2.0, 2, 2, 1, 1, 1]])
    R = np.stack([np.eye(3), np.eye(3)])
    assert np.allclose(po.store(s, s0, R, False), [(9 + 12) / 2, (19 + 22) / 2, (29 + 32) / 2, 1.5, 2.5, 3.5])
    assert np.allclose(po.store(s, s0, R, True), s.mean(0))
    R2 = np.stack([np.eye(3), po.rotation_tensor([1, 0, 0], [0, 1, 0])])
    exp = 0.5 * ((s[0] - s0[0]) + po.rotate_sym2(s[1] - s0[1], R2[1]))
    assert np.allclose(po.store(s, s0, R2, False), exp)


def test_file_order_vs_raw_order(tmp_path):
    """init.*.stress files are written 00,01,02,11,12,22 (read_write.h:217-220) while the wire order is
    xx,yy,zz,xy,xz,yz.  The shipped init.sic_1.stress values pin the mapping."""
    g = json.load(open(os.path.join(GOLD, "init_sic_1_stiff.json")))
    p = tmp_path / "init.sic_1.stress"
    p.write_text("\n".join(repr(v) for v in g["stress_file_order"]) + "\n")
    raw = po.read_sym2(str(p))
    f = g["stress_file_order"]
    assert np.allclose(raw, [f[0], f[3], f[5], f[1], f[2], f[4]], rtol=0, atol=0)


def test_set_md_procs_golden_vectors():
    """STMDSync::set_md_procs (stmd_sync.h:189-278): hand-derived cases.  Admissible ranks per batch are the factors and
    the multiples of the cores per node; the largest one not above the fair share P / nmdruns is taken."""
    import pytest
    from scema_amd import capi, stmd
    # 576 simulations on 8 ranks (GPUs): fair share 0 -> 1, one rank per batch, 8 batches, colour = rank
    assert [stmd.set_md_procs(576, 8, r, 1, 8) for r in (0, 3, 7)] == [(1, 8, 0), (1, 8, 3), (1, 8, 7)]
    # 2 simulations on 8 ranks, 8 cores per node: fair share 4 -> batches of 4 ranks, colours 0 0 0 0 1 1 1 1
    assert [stmd.set_md_procs(2, 8, r, 1, 8)[2] for r in range(8)] == [0, 0, 0, 0, 1, 1, 1, 1]
    assert stmd.set_md_procs(2, 8, 0, 1, 8)[:2] == (4, 2)
    # 3 simulations on 48 ranks, 24 per node: fair share 16; admissible <= 16: 1,2,3,4,6,8,12 -> 12; 4 batches
    assert stmd.set_md_procs(3, 48, 47, 1, 24) == (12, 4, 3)
    # 1 simulation on 60 ranks, 24 per node: fair share 60; admissible: factors of 24 and 24, 48 -> 48; 1 batch, 12 ranks left over
    assert stmd.set_md_procs(1, 60, 10, 1, 24) == (48, 1, 0)
    assert stmd.set_md_procs(1, 60, 55, 1, 24) == (48, 1, -1)
    # no update this step (nmdruns = 0): everything in one batch
    assert stmd.set_md_procs(0, 8, 5, 1, 8) == (8, 1, 0)
    # minimum allocation above the fair share: nothing admissible -> the reference exits
    with pytest.raises(capi.EngineError):
        stmd.set_md_procs(100, 8, 0, 4, 8)
