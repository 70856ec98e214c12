"""On-disk formats either side of the path: the replica container that stands in for
init.<mat>_<rep>.bin, and LAMMPS `write_data` text files of atom_style full."""
import ctypes as C
import os

import numpy as np
import pytest


def _built():
    import __graft_entry__ as g
    g.build()


KEYS = ["type", "charge", "mass", "eps", "sigma", "bonds", "bond_type", "bond_coeff", "angles", "angle_type", "angle_coeff",
        "dihedrals", "dihedral_type", "dihedral_coeff", "impropers", "improper_type", "improper_coeff", "special_lj",
        "special_coul", "box", "x", "v"]


def test_replica_container_round_trip(small_pe, tmp_path):
    _built()
    from scema_amd import stmd
    from scema_amd.systems import read_replica_file
    p = str(tmp_path / "init.pe_1.bin")
    stmd.write_replica_file(p, small_pe)
    back = read_replica_file(p)
    for k in KEYS:
        assert np.array_equal(np.asarray(back[k]), np.asarray(small_pe[k]).astype(np.asarray(back[k]).dtype)), k


def test_lammps_data_file_converts_to_the_same_replica(small_pe, tmp_path):
    """write_data-style text -> scema_md_convert_lammps_data -> container: identical topology and state
    (angles go through degrees, everything else is exact through repr())."""
    _built()
    from scema_amd import capi
    from scema_amd.systems import write_lammps_data, read_replica_file
    d = dict(small_pe)
    data = str(tmp_path / "pe.data"); out = str(tmp_path / "pe.bin")
    write_lammps_data(data, d)
    rc = capi.lib().scema_md_convert_lammps_data(data.encode(), out.encode(), None, None)
    assert rc == 0
    back = read_replica_file(out)
    for k in KEYS:
        a, b = np.asarray(back[k], float), np.asarray(d[k], float)
        if k in ("angle_coeff", "improper_coeff"):
            assert np.allclose(a, b, rtol=1e-15, atol=1e-15), k
        else:
            assert np.array_equal(a, b), k
    # image flags are unwrapped on read; ids need not be ordered
    lines = open(data).read().split("\n")
    ia = lines.index("Atoms # full") + 2
    first = lines[ia].split()
    lx = float(d["box"][3] - d["box"][0])
    first[4] = repr(float(first[4]) - lx); first[7] = "1"       # same atom, wrapped one box to the left
    lines[ia] = " ".join(first)
    lines[ia], lines[ia + 1] = lines[ia + 1], lines[ia]          # shuffled order
    open(data, "w").write("\n".join(lines))
    assert capi.lib().scema_md_convert_lammps_data(data.encode(), out.encode(), None, None) == 0
    back2 = read_replica_file(out)
    # atom order follows the file; compare as sets through the shuffled pair
    assert np.allclose(back2["x"][1], d["x"][0], atol=1e-12) and np.allclose(back2["x"][0], d["x"][1], atol=1e-12)
    assert sorted(map(tuple, np.sort(back2["bonds"], axis=1).tolist()))[:3] is not None


def test_atom_style_charge_data_file(tmp_path):
    """the reax scripts use atom_style charge (lammps_scripts_reax/in.set.lammps:17): `id type q x y z [ix iy iz]`, no bonded
    sections and no pair coefficients (the force field comes from ffield.reax.2)"""
    from scema_amd import capi
    from scema_amd.systems import read_replica_file
    rng = np.random.default_rng(4)
    n, box = 7, np.array([0.0, -1.0, 2.0, 9.0, 8.5, 11.0, 0.6, -0.4, 0.3])
    typ = rng.integers(1, 5, n)
    q = rng.normal(0, 0.2, n)
    x = rng.uniform(1.0, 7.0, (n, 3))
    v = rng.normal(0, 1e-3, (n, 3))
    img = rng.integers(-1, 2, (n, 3))
    order = rng.permutation(n)
    for named in (True, False):
        p = tmp_path / f"g0_1_{int(named)}.data"
        with open(p, "w") as fp:
            fp.write("LAMMPS data file via write_data\n\n%d atoms\n4 atom types\n\n" % n)
            fp.write("%.17g %.17g xlo xhi\n%.17g %.17g ylo yhi\n%.17g %.17g zlo zhi\n%.17g %.17g %.17g xy xz yz\n\n" % (box[0], box[3], box[1], box[4], box[2], box[5], box[6], box[7], box[8]))
            fp.write("Masses\n\n1 1.008\n2 12.011\n3 14.007\n4 15.999\n\n")
            fp.write("Atoms # charge\n\n" if named else "Atoms\n\n")
            for i in order:
                fp.write("%d %d %.17g %.17g %.17g %.17g %d %d %d\n" % (i + 1, typ[i], q[i], x[i, 0], x[i, 1], x[i, 2], img[i, 0], img[i, 1], img[i, 2]))
            fp.write("\nVelocities\n\n")
            for i in order:
                fp.write("%d %.17g %.17g %.17g\n" % (i + 1, v[i, 0], v[i, 1], v[i, 2]))
        out = str(tmp_path / "o.bin")
        assert capi.lib().scema_md_convert_lammps_data(str(p).encode(), out.encode(), None, None) == 0
        d = read_replica_file(out)
        h = np.array([[box[3] - box[0], box[6], box[7]], [0.0, box[4] - box[1], box[8]], [0.0, 0.0, box[5] - box[2]]])
        xu = x[order] + img[order] @ h.T                       # unwrapped with the image flags; file order is kept
        assert d["natoms"] == n and d["ntypes"] == 4 and len(d["bonds"]) == 0
        assert np.array_equal(d["type"], typ[order] - 1) and np.allclose(d["charge"], q[order], rtol=1e-15)
        assert np.abs(d["x"] - xu).max() < 1e-13 and np.allclose(d["v"], v[order], rtol=1e-15) and np.allclose(d["box"], box)
        assert np.all(d["eps"] == 0.0)


def test_bad_files_are_rejected(tmp_path):
    _built()
    from scema_amd import capi
    p = tmp_path / "junk.data"
    p.write_text("title\n\n3 atoms\n1 atom types\n\n0 1 xlo xhi\n0 1 ylo yhi\n0 1 zlo zhi\n\nMasses\n\n1 12.0\n\nAtoms\n\n1 1 1 0.0 0 0 0\n")
    assert capi.lib().scema_md_convert_lammps_data(str(p).encode(), str(tmp_path / "o.bin").encode(), None, None) != 0   # 3 atoms declared, 1 given
    assert capi.lib().scema_md_convert_lammps_data(b"/nonexistent", str(tmp_path / "o.bin").encode(), None, None) != 0


# ---- LAMMPS binary restart (17Nov16 layout) ----
SIC = os.path.join(os.path.dirname(__file__), "golden", "lammps_17Nov16_init.sic_1.bin")


def test_restart_reader_on_the_reference_fixture():
    """The reference ships one real LAMMPS 17Nov16 restart (examples/streched_polyhedron/nanoscale_input/init.sic_1.bin,
    written by init_material_problem.h:209; copied as data to tests/golden/): header walk, group list, fix lists and the
    atomic per-atom block of the C++ reader against it, and against the plain-Python reader under oracle/."""
    _built()
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
    import lammps_restart as lr
    from scema_amd import capi
    info = capi.probe_lammps_restart(SIC)
    assert info.version == b"17 Nov 2016" and info.units == b"metal" and info.atom_style == b"atomic" and info.pair_style == b""
    assert (info.natoms, info.ntypes, info.ntimestep, info.nprocs, info.triclinic) == (192, 1, 4, 1, 1)
    assert (info.nbonds, info.nangles, info.ndihedrals, info.nimpropers) == (0, 0, 0, 0)
    assert info.timestep == 0.001 and info.mass[0] == 28.06
    # 2 x 3 x 4 diamond-cubic cells of 5.43 A (8 atoms each), slightly dilated by the preceding NPT run
    L = np.array(info.box[3:6]) - np.array(info.box[0:3])
    assert np.allclose(L / np.array([2, 3, 4]), 5.4309, atol=2e-3) and list(info.box[6:9]) == [0.0, 0.0, 0.0]
    ref = lr.read_restart(SIC)
    assert ref["NATOMS"] == 192 and ref["groups"] == ["all"] and ref["fix_global"] == [] and ref["MASS"] == [28.06]
    assert list(info.box[0:3]) == ref["BOXLO"] and list(info.box[3:6]) == ref["BOXHI"]
    at = capi.read_lammps_restart_atoms(SIC, 192)
    assert sorted(at["tag"].tolist()) == list(range(1, 193)) and set(at["type"].tolist()) == {1}
    rim = np.array([lr.as_int(a[7]) for a in ref["atoms"]])
    assert np.array_equal(at["image"], np.stack([(rim & 1023) - 512, ((rim >> 10) & 1023) - 512, (rim >> 20) - 512], 1))
    assert np.abs(at["image"]).max() == 1      # a few atoms crossed a face during the 4 steps
    rx = np.array([a[1:4] for a in ref["atoms"]]); rv = np.array([a[8:11] for a in ref["atoms"]])
    rt = np.array([lr.as_int(a[4]) for a in ref["atoms"]])
    assert np.array_equal(at["x"], rx) and np.array_equal(at["v"], rv) and np.array_equal(at["tag"], rt)
    lo, hi = np.array(info.box[0:3]), np.array(info.box[3:6])
    assert np.all(at["x"] >= lo - 1e-9) and np.all(at["x"] <= hi + 1e-9)
    # nearest-neighbour distance of the diamond lattice: a sqrt(3)/4
    d = at["x"][None, :, :] - at["x"][:, None, :]
    d -= np.round(d / L) * L
    r = np.sqrt((d ** 2).sum(-1)) + np.eye(192) * 1e9
    assert np.allclose(r.min(1), 5.4309 * np.sqrt(3) / 4, atol=5e-3)
    # not a replica of the OPLS path: refused with a reason, not misread
    assert capi.lib().scema_md_convert_lammps_restart(SIC.encode(), b"/tmp/never_written.bin") != 0


def test_restart_writer_round_trip(small_pe, tmp_path):
    """replica -> restart file in the 17Nov16 layout (atom_style full, lj/cut/coul/long, harmonic / opls blocks) ->
    both readers -> the same replica.  Atoms outside the box are wrapped on write and unwrapped through their image
    flags on read."""
    _built()
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
    import lammps_restart as lr
    from scema_amd import capi
    from scema_amd.systems import read_replica_file
    d = dict(small_pe)
    d["x"] = np.array(d["x"], float).copy()
    box = np.asarray(d["box"], float)
    d["x"][0] += [box[3] - box[0], 0.0, 0.0]           # one box to the right
    d["x"][1] -= [0.0, 2 * (box[4] - box[1]), 0.0]     # two boxes down
    rst = str(tmp_path / "last.7.pe_1.bin"); out = str(tmp_path / "pe.bin")
    capi.write_lammps_restart(rst, d, 12.0, 9.0, timestep=2.0, ntimestep=110)
    info = capi.probe_lammps_restart(rst)
    assert info.atom_style == b"full" and info.units == b"real" and info.pair_style == b"lj/cut/coul/long"
    assert (info.natoms, info.ntimestep, info.cut_lj, info.cut_coul) == (d["natoms"], 110, 12.0, 9.0)
    assert list(info.special_lj) == list(d["special_lj"]) and list(info.box) == list(box)
    assert capi.lib().scema_md_convert_lammps_restart(rst.encode(), out.encode()) == 0
    back = read_replica_file(out)
    for k in KEYS:
        a, b = np.asarray(back[k], float), np.asarray(d[k], float)
        if k == "x":
            assert np.allclose(a, b, rtol=0, atol=1e-12), k      # wrap + unwrap
        elif k in ("bonds", "angles", "dihedrals", "impropers", "bond_type", "angle_type", "dihedral_type", "improper_type"):
            continue                                             # term order follows the owning atoms: compared as sets below
        else:
            assert np.array_equal(a, b), k
    for at_k, tp_k in (("bonds", "bond_type"), ("angles", "angle_type"), ("dihedrals", "dihedral_type"), ("impropers", "improper_type")):
        assert len(back[tp_k]) == len(d[tp_k])
        if len(d[tp_k]) == 0:
            continue
        s1 = sorted(tuple(r) + (t,) for r, t in zip(np.asarray(back[at_k]).reshape(len(back[tp_k]), -1).tolist(), list(back[tp_k])))
        s2 = sorted(tuple(r) + (t,) for r, t in zip(np.asarray(d[at_k]).reshape(len(d[tp_k]), -1).tolist(), list(np.asarray(d[tp_k]))))
        assert s1 == s2, at_k
    # the plain-Python reader sees the same file the same way
    ref = lr.read_restart(rst)
    assert ref["ATOM_STYLE"] == "full" and ref["PAIR"] == "lj/cut/coul/long" and ref["pair_settings"]["cut_coul"] == 9.0
    assert ref["NBONDS"] == len(d["bond_type"]) and ref["BOND"] == "harmonic" and ref["DIHEDRAL"] == "opls"
    assert np.allclose(np.array(ref["dihedral_coeff"]).T * 2.0, np.asarray(d["dihedral_coeff"], float))   # LAMMPS keeps K/2
    a0 = ref["atoms"][0]
    assert lr.as_int(a0[4]) == 1 and lr.as_int(a0[7]) == (512 << 20) | (512 << 10) | 513 and a0[11] == d["charge"][0]
    # truncated file
    raw = open(rst, "rb").read()
    bad = str(tmp_path / "cut.bin"); open(bad, "wb").write(raw[: len(raw) // 2])
    assert capi.lib().scema_md_convert_lammps_restart(bad.encode(), out.encode()) != 0
