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


def test_bad_files_are_rejected(tmp_path):
    _built()
    from scema_amd import capi
    p = tmp_path / "junk.data"
    p.write_text("title\n\n3 atoms\n1 atom types\n\n0 1 xlo xhi\n0 1 ylo yhi\n0 1 zlo zhi\n\nMasses\n\n1 12.0\n\nAtoms\n\n1 1 1 0.0 0 0 0\n")
    assert capi.lib().scema_md_convert_lammps_data(str(p).encode(), str(tmp_path / "o.bin").encode(), None, None) != 0   # 3 atoms declared, 1 given
    assert capi.lib().scema_md_convert_lammps_data(b"/nonexistent", str(tmp_path / "o.bin").encode(), None, None) != 0
