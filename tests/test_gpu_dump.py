"""States as LAMMPS text dumps: how the reax branch of the reference hands states from one LAMMPS lifetime to the next
(stmd_problem.h:190-194 `rerun <file> dump x y z vx vy vz ix iy iz box yes scaled yes wrapped yes format native`,
:261-264 `write_dump all custom <file> id type xs ys zs vx vy vz ix iy iz`)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KW = dict(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=1e-5)


def test_state_round_trip_through_a_text_dump(small_pe, tmp_path):
    from scema_amd import capi
    e = capi.Engine(capi.default_params(**KW))
    e.register_replica("pe", 1, small_pe)
    rng = np.random.default_rng(3)
    # atoms several boxes away from the home cell: image flags matter
    shift = rng.integers(-2, 3, size=(len(small_pe["x"]), 3)).astype(float)
    b = small_pe["box"]
    h = np.array([[b[3] - b[0], b[6], b[7]], [0.0, b[4] - b[1], b[8]], [0.0, 0.0, b[5] - b[2]]])
    x = small_pe["x"] + shift @ h.T
    v = rng.standard_normal(x.shape) * 1e-3
    e.set_state(5, "pe", 1, b, x, v)
    path = str(tmp_path / "last.5.pe_1.dump")
    e.save_state_dump(5, "pe", 1, path, ntimestep=110)
    lines = open(path).read().splitlines()
    assert lines[0] == "ITEM: TIMESTEP" and lines[1] == "110" and lines[3] == str(len(x))
    assert lines[4] == "ITEM: BOX BOUNDS xy xz yz pp pp pp" and lines[8] == "ITEM: ATOMS id type xs ys zs vx vy vz ix iy iz"
    # LAMMPS' bounding-box convention: xlo_bound = xlo + min(0, xy, xz, xy + xz) ...
    xlo_b, xhi_b, xy = map(float, lines[5].split())
    assert xy == b[6] and abs(xlo_b - (b[0] + min(0.0, b[6], b[7], b[6] + b[7]))) < 1e-14 and abs(xhi_b - (b[3] + max(0.0, b[6], b[7], b[6] + b[7]))) < 1e-14
    cols = np.array([l.split() for l in lines[9:]], dtype=float)
    assert np.all((cols[:, 2:5] >= 0.0) & (cols[:, 2:5] < 1.0))          # scaled, wrapped
    assert np.array_equal(cols[:, 0], np.arange(1, len(x) + 1)) and np.array_equal(cols[:, 1], small_pe["type"] + 1)
    e.load_state_file(6, "pe", 1, path)
    b2, x2, v2 = e.get_state(6, "pe", 1)
    assert np.abs(b2 - b).max() < 1e-13 and np.abs(x2 - x).max() < 1e-12 and np.array_equal(v2, v)
    # LAMMPS' default column format (%g): what a file written by the reference holds
    e.save_state_dump(5, "pe", 1, path, precise=False)
    e.load_state_file(7, "pe", 1, path)
    x3 = e.get_state(7, "pe", 1)[1]
    assert 1e-9 < np.abs(x3 - x).max() < 1e-3
    # a hand-written dump in another column order and atom order, unscaled coordinates
    perm = rng.permutation(len(x))
    with open(path, "w") as fp:
        fp.write("ITEM: TIMESTEP\n0\nITEM: NUMBER OF ATOMS\n%d\nITEM: BOX BOUNDS xy xz yz pp pp pp\n" % len(x))
        fp.write("%.17g %.17g %.17g\n%.17g %.17g %.17g\n%.17g %.17g %.17g\n" % (xlo_b, xhi_b, b[6], b[1] + min(0.0, b[8]), b[4] + max(0.0, b[8]), b[7], b[2], b[5], b[8]))
        fp.write("ITEM: ATOMS vx vy vz id x y z\n")
        for i in perm:
            fp.write("%.17g %.17g %.17g %d %.17g %.17g %.17g\n" % (v[i, 0], v[i, 1], v[i, 2], i + 1, x[i, 0], x[i, 1], x[i, 2]))
    e.load_state_file(8, "pe", 1, path)
    b4, x4, v4 = e.get_state(8, "pe", 1)
    assert np.abs(x4 - x).max() < 1e-12 and np.array_equal(v4, v)
    with open(path, "w") as fp:
        fp.write("ITEM: TIMESTEP\n0\nITEM: NUMBER OF ATOMS\n3\n")
    with pytest.raises(capi.EngineError, match="atoms"):
        e.load_state_file(9, "pe", 1, path)
    e.close()
