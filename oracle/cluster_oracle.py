"""Test infrastructure only (never imported by the product path): CPU restatement of the reference's strain-history
clustering step, pinned against the reference itself:

* `splinify`      <- MatHistPredict::Strain6D::splinify, headers/strain2spline.h:140-180, which fits tk::spline
                     (headers/spline.h:293-394: natural cubic spline, zero curvature at both ends, band-matrix LU) to
                     each of the 6 strain components over t = n/(steps-1) and samples it at t = n/(npts-1).
                     Pinned against the reference's spline.h compiled in place (oracle/_ref/libspline_ref.so).
* `l2_norm`       <- compare_L2_norm, strain2spline.h:469-487.
* `similar_lists` <- compare_histories_with_all_ranks, single-rank branch strain2spline.h:601-612, with
                     choose_most_similar_history :265-291 (diff < threshold) and the file body of
                     most_similar_histories_to_file :301-314.
* `cover`         <- clustering/coarsegrain_dependency_network.py:24-95 (greedy removal of the highest-degree node and
                     its neighbours).  Pinned against that script itself, imported from /root/reference
                     (tests/golden/make_cluster_golden.py).
"""
import numpy as np


def _spline_coeffs(x, y):
    """Natural cubic spline through (x, y): coefficients (a, b, c) of f = a h^3 + b h^2 + c h + y_i, in the arithmetic
    order of spline.h:293-366 (rows scaled by the inverse diagonal, then Gaussian elimination on the tridiagonal band)."""
    n = len(x)
    lo = [0.0] * n; di = [0.0] * n; up = [0.0] * n; rhs = [0.0] * n
    for i in range(1, n - 1):
        lo[i] = 1.0 / 3.0 * (x[i] - x[i - 1])
        di[i] = 2.0 / 3.0 * (x[i + 1] - x[i - 1])
        up[i] = 1.0 / 3.0 * (x[i + 1] - x[i])
        rhs[i] = (y[i + 1] - y[i]) / (x[i + 1] - x[i]) - (y[i] - y[i - 1]) / (x[i] - x[i - 1])
    di[0] = 2.0; up[0] = 0.0; rhs[0] = 0.0                # zero second derivative at both ends
    di[n - 1] = 2.0; lo[n - 1] = 0.0; rhs[n - 1] = 0.0
    inv = [0.0] * n
    for i in range(n):                                     # row scaling
        inv[i] = 1.0 / di[i]
        lo[i] *= inv[i]; up[i] *= inv[i]; di[i] = 1.0
    for k in range(n - 1):                                 # elimination
        f = -lo[k + 1] / di[k]
        lo[k + 1] = -f
        di[k + 1] = di[k + 1] + f * up[k]
    z = [0.0] * n
    for i in range(n):                                     # forward substitution
        s = 0.0
        if i > 0:
            s += lo[i] * z[i - 1]
        z[i] = rhs[i] * inv[i] - s
    b = [0.0] * n
    for i in range(n - 1, -1, -1):                         # back substitution
        s = 0.0
        if i < n - 1:
            s += up[i] * b[i + 1]
        b[i] = (z[i] - s) / di[i]
    a = [0.0] * n; c = [0.0] * n
    for i in range(n - 1):
        a[i] = 1.0 / 3.0 * (b[i + 1] - b[i]) / (x[i + 1] - x[i])
        c[i] = (y[i + 1] - y[i]) / (x[i + 1] - x[i]) - 1.0 / 3.0 * (2.0 * b[i] + b[i + 1]) * (x[i + 1] - x[i])
    h = x[n - 1] - x[n - 2]
    a[n - 1] = 0.0
    c[n - 1] = 3.0 * a[n - 2] * h * h + 2.0 * b[n - 2] * h + c[n - 2]
    return a, b, c


def _spline_eval(x, y, a, b, c, t):
    n = len(x)
    idx = max(int(np.searchsorted(x, t, side="left")) - 1, 0)       # std::lower_bound
    h = t - x[idx]
    if t < x[0]:
        return (b[0] * h + c[0]) * h + y[0]
    if t > x[n - 1]:
        return (b[n - 1] * h + c[n - 1]) * h + y[n - 1]
    return ((a[idx] * h + b[idx]) * h + c[idx]) * h + y[idx]


def splinify_component(y, npts):
    steps = len(y)
    if steps < 3:
        raise ValueError("need at least 3 strain steps")          # strain2spline.h:145-148
    x = [n / (steps - 1) for n in range(steps)]
    yy = [float(v) for v in y]
    a, b, c = _spline_coeffs(x, yy)
    return np.array([_spline_eval(x, yy, a, b, c, n / (npts - 1)) for n in range(npts)])


def splinify(hist, npts):
    """hist[steps][6] (xx,yy,zz,xy,xz,yz per step) -> spline[npts*6], point-major like strain2spline.h:170-178."""
    hist = np.asarray(hist, float)
    comps = [splinify_component(hist[:, k], npts) for k in range(6)]
    return np.stack(comps, axis=1).ravel()


def l2_norm(a, b):
    s = 0.0
    for u, v in zip(np.asarray(a, float).tolist(), np.asarray(b, float).tolist()):
        d = u - v
        s += d * d
    return float(np.sqrt(s))


def similar_lists(ids, splines, threshold):
    """-> {id: [(other_id, diff), ...]} in the order the reference pushes them (pairs a < b of the history vector)."""
    n = len(ids)
    out = {int(i): [] for i in ids}
    for a in range(n):
        for b in range(a + 1, n):
            d = l2_norm(splines[a], splines[b])
            if d < threshold:
                out[int(ids[a])].append((int(ids[b]), d))
                out[int(ids[b])].append((int(ids[a]), d))
    return out


def cover(edge_lines, num_gps):
    """edge_lines: the (cell1, cell2) pairs in the order the script reads them (file by file, line by line).  Returns
    mapping[num_gps]: the greedy cover of coarsegrain_dependency_network.py:45-86.  Degree ties go to the node that
    entered the graph LAST (the script sorts the degree dictionary, which is in insertion order, with a stable sort and
    takes the final element)."""
    order = []                 # node insertion order
    adj = {}
    for c1, c2 in edge_lines:
        for c in (c1, c2):
            if c not in adj:
                adj[c] = set(); order.append(c)
        if c1 != c2:
            adj[c1].add(c2); adj[c2].add(c1)
        else:
            adj[c1].add(c1)    # networkx self-loop: counts twice in the degree, listed once among the neighbours
    mapping = list(range(num_gps))
    alive = [c for c in order]
    while alive:
        best = None; bestdeg = -1
        for c in alive:
            deg = len(adj[c]) + (1 if c in adj[c] else 0)
            if deg >= bestdeg:
                best, bestdeg = c, deg
        mapping[best] = best
        gone = {best} | set(adj[best])
        for nb in adj[best]:
            mapping[nb] = best
        mapping[best] = best
        for g in gone:
            for nb in adj.get(g, ()):
                if nb not in gone:
                    adj[nb].discard(g)
        for g in gone:
            adj.pop(g, None)
        alive = [c for c in alive if c not in gone]
    return mapping
