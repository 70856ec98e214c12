"""ctypes binding of include/scema_cluster.h (strain-history clustering): spline fit and cover
on the host, all-pairs distances on the GPU (no CPU fallback for the distances)."""
import ctypes as C

import numpy as np

from . import capi

SYMBOLS = ["scema_hist_splinify", "scema_hist_compare", "scema_hist_compare_device", "scema_hist_edges", "scema_hist_similar", "scema_hist_cover",
           "scema_hist_cluster"]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _chk(rc, what):
    if rc != 0:
        raise capi.EngineError(f"{what} failed (rc={rc})")


def splinify(hist, npts: int) -> np.ndarray:
    """hist[n][steps][6] -> spline[n][npts*6] (Strain6D::splinify, strain2spline.h:140-180)."""
    h = np.ascontiguousarray(hist, np.float64)
    n, steps, six = h.shape
    assert six == 6
    out = np.zeros((n, npts * 6))
    _chk(capi.lib().scema_hist_splinify(_p(h), C.c_int32(n), C.c_int32(steps), C.c_int32(npts), _p(out)), "scema_hist_splinify")
    return out


def compare(splines, device: int = 0) -> np.ndarray:
    """All-pairs L2 distances on the GPU (compare_L2_norm, strain2spline.h:469-487)."""
    s = np.ascontiguousarray(splines, np.float64)
    n, d = s.shape
    out = np.zeros((n, n))
    _chk(capi.lib().scema_hist_compare(_p(s), C.c_int32(n), C.c_int32(d), _p(out), C.c_int32(device)), "scema_hist_compare")
    return out


def edges(splines, threshold: float, device: int = 0):
    """Similar pairs (a < b, diff < threshold) straight from the GPU, sorted: (pairs[m][2], dist[m])."""
    s = np.ascontiguousarray(splines, np.float64)
    n, d = s.shape
    cap = max(1024, 16 * n)
    cnt = C.c_int64(0)
    for _ in range(2):
        pairs = np.zeros((cap, 2), np.int32); dist = np.zeros(cap)
        rc = capi.lib().scema_hist_edges(_p(s), C.c_int32(n), C.c_int32(d), C.c_double(threshold), C.c_int32(device), C.c_int64(cap),
                                         _p(pairs), _p(dist), C.byref(cnt))
        if rc != 6:      # SCEMA_MD_ERR_OVERFLOW
            break
        cap = cnt.value
    _chk(rc, "scema_hist_edges")
    m = cnt.value
    order = np.lexsort((pairs[:m, 1], pairs[:m, 0]))
    return pairs[:m][order], dist[:m][order]


def similar(diff, threshold: float):
    """-> (start[n+1], other[m], dist[m]): what most_similar_histories_to_file writes, history by history."""
    dm = np.ascontiguousarray(diff, np.float64)
    n = dm.shape[0]
    start = np.zeros(n + 1, np.int64)
    capi.lib().scema_hist_similar(_p(dm), C.c_int32(n), C.c_double(threshold), C.c_int64(0), _p(start), None, None)
    m = int(start[n])
    other = np.zeros(max(m, 1), np.int32); dist = np.zeros(max(m, 1))
    _chk(capi.lib().scema_hist_similar(_p(dm), C.c_int32(n), C.c_double(threshold), C.c_int64(m), _p(start), _p(other), _p(dist)),
         "scema_hist_similar")
    return start, other[:m], dist[:m]


def cover(edges, num_gps: int) -> np.ndarray:
    """edges[m][2] in the order the reference script reads them -> mapping[num_gps] (coarsegrain_dependency_network.py)."""
    e = np.ascontiguousarray(np.asarray(edges, np.int32).reshape(-1, 2))
    mapping = np.zeros(num_gps, np.int32)
    _chk(capi.lib().scema_hist_cover(_p(e) if len(e) else None, C.c_int64(len(e)), C.c_int32(num_gps), _p(mapping)), "scema_hist_cover")
    return mapping


def cluster(ids, hist, npts: int, threshold: float, num_gps: int, device: int = 0) -> np.ndarray:
    """The whole step (FEProblem::spline_building + spline_comparison, FE_problem.h:1196-1270): mapping[num_gps]."""
    i = np.ascontiguousarray(ids, np.int32)
    h = np.ascontiguousarray(hist, np.float64)
    n, steps, _ = h.shape
    mapping = np.zeros(num_gps, np.int32)
    _chk(capi.lib().scema_hist_cluster(_p(i), _p(h), C.c_int32(n), C.c_int32(steps), C.c_int32(npts), C.c_double(threshold),
                                       C.c_int32(num_gps), C.c_int32(device), _p(mapping)), "scema_hist_cluster")
    return mapping
