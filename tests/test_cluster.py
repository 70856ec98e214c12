"""Strain-history clustering: spline fit of the strain histories, all-pairs L2 distances, greedy
cover of the similarity graph.  The goldens under tests/golden/cluster_golden.json come from the reference itself (its
spline.h compiled in place, its coarsegrain_dependency_network.py imported; tests/golden/make_cluster_golden.py)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    return json.load(open(os.path.join(HERE, "golden", "cluster_golden.json")))


def _edge_lines(case, order):
    lists = {int(k): v for k, v in case["lists"].items()}
    return [(i, int(other)) for i in order for other, _ in lists[i]]


def test_oracle_spline_matches_the_reference_spline(gold):
    from oracle import cluster_oracle as co
    for g in gold["splines"]:
        got = co.splinify_component(g["y"], g["npts"])
        assert np.array_equal(got, np.array(g["values"])), (len(g["y"]), g["npts"])     # same arithmetic, same bits


def test_oracle_cover_matches_the_reference_script(gold):
    from oracle import cluster_oracle as co
    for case in gold["covers"]:
        for o in case["orders"]:
            assert co.cover(_edge_lines(case, o["file_order"]), case["num_gps"]) == o["mapping"]


def test_oracle_similarity_lists_are_symmetric_and_thresholded(gold):
    from oracle import cluster_oracle as co
    case = gold["covers"][1]
    hist = np.array(case["hist"])
    sp = np.array([co.splinify(h, case["npts"]) for h in hist])
    lists = co.similar_lists(case["ids"], sp, case["threshold"])
    assert {int(k): [(int(a), b) for a, b in v] for k, v in case["lists"].items()} == {k: v for k, v in lists.items()}
    for i, l in lists.items():
        for j, d in l:
            assert d < case["threshold"] and (i, d) in lists[j]


# ---- the product's host side (C ABI, no GPU needed) ----
def _built():
    import __graft_entry__ as g
    g.build()


def test_host_splinify_matches_the_reference_spline(gold):
    _built()
    from scema_amd import cluster
    for g in gold["splines"]:
        y = np.array(g["y"])
        hist = np.zeros((1, len(y), 6))
        for k in range(6):
            hist[0, :, k] = y * (k + 1)          # six components, each a multiple of the golden curve
        sp = cluster.splinify(hist, g["npts"]).reshape(g["npts"], 6)
        assert np.array_equal(sp[:, 0], np.array(g["values"]))
        for k in range(1, 6):                    # the fit is linear in the data (to rounding)
            assert np.allclose(sp[:, k], (k + 1) * np.array(g["values"]), rtol=1e-12, atol=1e-18)
    with pytest.raises(Exception):
        cluster.splinify(np.zeros((1, 2, 6)), 5)   # fewer than 3 steps (strain2spline.h:145-148)


def test_host_cover_matches_the_reference_script(gold):
    _built()
    from scema_amd import cluster
    for case in gold["covers"]:
        for o in case["orders"]:
            got = cluster.cover(_edge_lines(case, o["file_order"]), case["num_gps"])
            assert got.tolist() == o["mapping"]
    assert cluster.cover([], 5).tolist() == [0, 1, 2, 3, 4]      # nothing similar: every point runs its own MD


def test_host_similar_lists_from_a_distance_matrix(gold):
    _built()
    from oracle import cluster_oracle as co
    from scema_amd import cluster
    case = gold["covers"][2]
    hist = np.array(case["hist"])
    sp = cluster.splinify(hist, case["npts"])
    n = len(sp)
    dm = np.array([[co.l2_norm(sp[a], sp[b]) if a != b else 0.0 for b in range(n)] for a in range(n)])
    start, other, dist = cluster.similar(dm, case["threshold"])
    ids = case["ids"]
    lists = {int(k): v for k, v in case["lists"].items()}
    for a in range(n):
        got = [(ids[o], d) for o, d in zip(other[start[a]:start[a + 1]], dist[start[a]:start[a + 1]])]
        assert got == [(int(o), d) for o, d in lists[ids[a]]]


@pytest.mark.gpu
def test_gpu_all_pairs_distances_are_bit_identical(gold):
    from oracle import cluster_oracle as co
    from scema_amd import cluster
    rng = np.random.default_rng(4)
    for n, d in ((1, 6), (5, 12), (33, 60), (70, 66), (130, 600)):
        sp = rng.normal(0, 1e-3, (n, d))
        dm = cluster.compare(sp)
        assert np.array_equal(dm, dm.T) and np.all(np.diag(dm) == 0.0)
        for a, b in [(0, n - 1), (n // 2, n // 3), (n - 1, n // 2)] + [tuple(rng.integers(0, n, 2)) for _ in range(20)]:
            assert dm[a, b] == co.l2_norm(sp[a], sp[b]), (n, d, a, b)


@pytest.mark.gpu
def test_gpu_clustering_step_matches_the_reference_mapping(gold):
    """splines (host) -> distances (GPU) -> lists -> cover == what the reference script returns when it reads the
    similarity files in ascending id order."""
    from scema_amd import cluster
    for case in gold["covers"]:
        got = cluster.cluster(case["ids"], np.array(case["hist"]), case["npts"], case["threshold"], case["num_gps"])
        assert got.tolist() == case["orders"][0]["mapping"]


@pytest.mark.gpu
def test_gpu_full_size_distance_properties():
    """4 864 histories (every quadrature point of the dogbone mesh), 10 spline points: symmetry, zero diagonal, triangle
    inequality on sampled triples, and scale linearity |c a - c b| = c |a - b| for a power-of-two c."""
    from scema_amd import cluster
    rng = np.random.default_rng(11)
    n, d = 4864, 60
    sp = rng.normal(0, 1e-3, (n, d))
    dm = cluster.compare(sp)
    assert np.array_equal(dm, dm.T) and np.all(np.diag(dm) == 0.0)
    i, j, k = rng.integers(0, n, (3, 2000))
    assert np.all(dm[i, k] <= dm[i, j] + dm[j, k] + 1e-15)
    assert np.array_equal(cluster.compare(4.0 * sp[:512]), 4.0 * dm[:512, :512])
    ref = np.sqrt(((sp[i] - sp[j]) ** 2).sum(1))
    assert np.allclose(dm[i, j], ref, rtol=1e-14)


@pytest.mark.gpu
def test_gpu_edge_list_equals_the_thresholded_matrix(gold):
    from scema_amd import cluster
    rng = np.random.default_rng(8)
    sp = rng.normal(0, 1e-3, (300, 60))
    dm = cluster.compare(sp)
    thr = float(np.quantile(dm[np.triu_indices(300, 1)], 0.2))      # 20 % of the pairs: more than the first capacity guess
    pairs, dist = cluster.edges(sp, thr)
    a, b = np.nonzero(np.triu(dm < thr, 1))
    assert np.array_equal(pairs, np.stack([a, b], 1)) and np.array_equal(dist, dm[a, b])
    p0, d0 = cluster.edges(sp, 0.0)
    assert len(p0) == 0 and len(d0) == 0


@pytest.mark.gpu
def test_gpu_identical_histories_are_an_edge_not_a_crash():
    """Two identical histories have distance 0; the reference script would divide by it (ZeroDivisionError on the unused
    edge weight).  Here they are simply similar: one of them takes its results from the other."""
    from scema_amd import cluster
    rng = np.random.default_rng(2)
    base = np.cumsum(rng.normal(0, 1e-3, (3, 6, 6)), 1)
    hist = np.stack([base[0], base[1], base[0], base[2]])          # histories 0 and 2 coincide
    m = cluster.cluster([4, 7, 9, 12], hist, 8, 1e-6, 16).tolist()
    assert m[4] == m[9] and m[4] in (4, 9)
    assert m[7] == 7 and m[12] == 12 and all(m[i] == i for i in range(16) if i not in (4, 9))
