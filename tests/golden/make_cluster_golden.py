"""Generates tests/golden/cluster_golden.json from the REFERENCE itself (run in the build container, where /root/reference
exists; the JSON is what travels):

* spline values  : the reference's headers/spline.h compiled in place (oracle/_ref/libspline_ref.so, oracle/ref_build.mk)
* cover mappings : clustering/coarsegrain_dependency_network.py imported from /root/reference and run on similarity files
                   written in the format of Strain6D::most_similar_histories_to_file (strain2spline.h:301-314).  The
                   script reads the files in glob order, which depends on the file system; the generator pins it to the
                   order recorded in the golden ("file_order") by patching glob.glob -- nothing else is changed.

usage: python tests/golden/make_cluster_golden.py
"""
import ctypes as C
import glob
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def ref_spline(y, npts):
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libspline_ref.so"))
    y = np.ascontiguousarray(y, np.float64)
    out = np.zeros(npts)
    rc = lib.ref_splinify_component(y.ctypes.data_as(C.c_void_p), C.c_int(len(y)), C.c_int(npts), out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return out


def ref_cover(lists, file_order, num_gps):
    """lists: {id: [(other, diff), ...]} -> mapping, by the reference script."""
    sys.path.insert(0, "/root/reference/clustering")
    import coarsegrain_dependency_network as ref
    with tempfile.TemporaryDirectory() as d:
        names = []
        for i in file_order:
            fn = os.path.join(d, "last.%d.similar_hist" % i)
            with open(fn, "w") as f:
                for other, diff in lists[i]:
                    f.write("%d %d %r\n" % (i, other, diff))
            names.append(fn)
        orig = glob.glob
        ref.glob.glob = lambda pattern: list(names)
        try:
            out = os.path.join(d, "mapping.csv")
            ref.coarsegrain_dependency_network(d, out, num_gps)
        finally:
            ref.glob.glob = orig
        return [int(l.split()[1]) for l in open(out)]


def main():
    rng = np.random.default_rng(20261003)
    gold = {"splines": [], "covers": []}
    for steps, npts in ((3, 5), (4, 10), (7, 10), (12, 25), (30, 7), (501, 40)):
        y = np.cumsum(rng.normal(0, 1e-3, steps)) + 1e-4 * np.arange(steps)
        gold["splines"].append({"y": y.tolist(), "npts": npts, "values": ref_spline(y, npts).tolist()})
    from oracle import cluster_oracle as co
    for n, num_gps, steps, npts, thr in ((6, 8, 5, 6, 4e-3), (24, 30, 8, 10, 3e-3), (60, 64, 6, 10, 2.5e-3), (40, 40, 10, 12, 1e9),
                                         (12, 12, 4, 5, 0.0)):
        ids = sorted(rng.choice(num_gps, size=n, replace=False).tolist())
        centres = rng.normal(0, 2e-3, (max(2, n // 6), 6))
        hist = np.array([np.cumsum(np.tile(centres[rng.integers(len(centres))] / steps, (steps, 1)) + rng.normal(0, 1.5e-4, (steps, 6)), 0)
                         for _ in range(n)])
        splines = np.array([co.splinify(h, npts) for h in hist])
        lists = co.similar_lists(ids, splines, thr)
        case = {"ids": ids, "num_gps": num_gps, "threshold": thr, "npts": npts, "hist": hist.tolist(),
                "lists": {str(k): v for k, v in lists.items()}, "orders": []}
        for order_name in ("ascending", "descending", "shuffled"):
            order = list(ids) if order_name == "ascending" else list(reversed(ids)) if order_name == "descending" else rng.permutation(ids).tolist()
            case["orders"].append({"file_order": order, "mapping": ref_cover(lists, order, num_gps)})
        gold["covers"].append(case)
    with open(os.path.join(HERE, "cluster_golden.json"), "w") as f:
        json.dump(gold, f)
    print("splines:", len(gold["splines"]), "covers:", len(gold["covers"]))


if __name__ == "__main__":
    main()
