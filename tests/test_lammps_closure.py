"""The LAMMPS closure kit (tools/export_lammps_case.py): the exported case drives the reference's own three scripts with the
variables stmd_problem.h:159-244,309-325 sets.  Without a LAMMPS executable (none exists on the images of this project) the
verdict is "invariant-verified" and the test checks the exported inputs; with one it runs them and demands the north
star's 1e-4."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_exported_case_issues_the_references_command_sequence(small_pe, tmp_path):
    import export_lammps_case as x
    from oracle import pyoracle as po
    lens = small_pe["box"][3:6] - small_pe["box"][:3]
    strain = np.array([-0.3 * 1.2e-3 * lens[0], -0.3 * 1.2e-3 * lens[1], 1.2e-3 * lens[2], 5e-5 * lens[2], -3e-5 * lens[1], 2e-5 * lens[0]])
    case = x.export(str(tmp_path), small_pe, strain, "/ref/lammps_scripts_opls", nss=20)
    true = strain / np.array([lens[0], lens[1], lens[2], lens[2], lens[1], lens[0]])
    assert case["nts"] == po.nts(true, 1e-4, 2.0) == 10                              # stmd_problem.h:229-232
    assert case["rates"] == [po.round_rate(v / (10 * 2.0)) for v in true]             # "%.6e", stmd_problem.h:241
    a = open(tmp_path / "phase_a.lammps").read().splitlines()
    want = ["variable mdt string g0", "variable tempt equal 300.000000", "include /ref/lammps_scripts_opls/in.set.lammps",
            "read_restart init.g0_1.bin", "variable dts equal 2.000000", "variable nts equal 10",
            "variable ceeps_00 equal %.6e" % case["rates"][0], "variable ceeps_01 equal %.6e" % case["rates"][3],
            "variable ceeps_12 equal %.6e" % case["rates"][5], "include /ref/lammps_scripts_opls/in.strain.lammps",
            "write_restart last.0.g0_1.dump"]
    pos = [a.index(w) for w in want]
    assert pos == sorted(pos)                                                         # same order as the reference issues them
    b = open(tmp_path / "phase_b.lammps").read()
    for w in ("read_restart last.0.g0_1.dump", "reset_timestep 0", "variable locbe string /ref/lammps_scripts_opls/ELASTIC",
              "variable nssample0 equal 20", "include /ref/lammps_scripts_opls/ELASTIC/in.homogenization.lammps", "${pp23}"):
        assert w in b
    assert "Atoms # full" in open(tmp_path / "replica.data").read()


def test_exported_reax_case_follows_the_reax_branch(tmp_path):
    """md_force_field "reax" (stmd_problem.h:190-194,261-264,297-302): atom_style charge data, the state as a text dump that comes
    back through read_restart init.bin + rerun, the reax script folder"""
    import export_lammps_case as x
    from scema_amd import capi
    from scema_amd.systems import synthetic_strains
    r = x.reax_replica((2, 3, 5))
    lens = r["box"][3:6] - r["box"][:3]
    case = x.export_reax(str(tmp_path), r, synthetic_strains(1, lens, seed=2026)[0], "/ref/lammps_scripts_reax")
    assert case["force_field"] == "reax" and case["dt"] == 0.25 and case["nss"] == 20 and case["nts"] % 10 == 0
    a = open(tmp_path / "phase_a.lammps").read().splitlines()
    want = ["include /ref/lammps_scripts_reax/in.set.lammps", "read_restart init.g0_1.bin", "variable dts equal 0.250000",
            "include /ref/lammps_scripts_reax/in.strain.lammps", "write_dump all custom last.0.g0_1.dump id type xs ys zs vx vy vz ix iy iz"]
    pos = [a.index(w) for w in want]
    assert pos == sorted(pos)
    b = open(tmp_path / "phase_b.lammps").read().splitlines()
    want = ["read_restart init.g0_1.bin", "rerun last.0.g0_1.dump dump x y z vx vy vz ix iy iz box yes scaled yes wrapped yes format native",
            "reset_timestep 0", "include /ref/lammps_scripts_reax/ELASTIC/in.homogenization.lammps"]
    pos = [b.index(w) for w in want]
    assert pos == sorted(pos)
    # the data file is one the engine's own reader takes (atom_style charge, types H C N O = 1..4)
    out = str(tmp_path / "o.bin")
    assert capi.lib().scema_md_convert_lammps_data(str(tmp_path / "replica.data").encode(), out.encode(), None, None) == 0
    from scema_amd.systems import read_replica_file
    d = read_replica_file(out)
    assert d["natoms"] == r["natoms"] and d["ntypes"] == 4 and sorted(set(d["type"].tolist())) == [0, 1]


@pytest.mark.gpu
def test_closure_verdict_on_this_host(small_pe, tmp_path):
    """runs LAMMPS where there is one ("LAMMPS-verified", 1e-4 demanded); says "invariant-verified" where there is none"""
    import export_lammps_case as x
    from scema_amd.systems import build_pe10k, synthetic_strains
    d = build_pe10k()
    lens = d["box"][3:6] - d["box"][:3]
    res = x.verify(str(tmp_path), d, synthetic_strains(1, lens, seed=2026)[0], os.environ.get("SCEMA_SCRIPTS", "/root/reference/lammps_scripts/lammps_scripts_opls"))
    print(res["verdict"])
    assert res["gpu"] is not None
    if res["lammps"] is None:
        assert res["verdict"].startswith("invariant-verified")
    else:
        assert res["rel_err_vs_lammps"]["gpu"] < 1e-4, res["verdict"]


@pytest.mark.gpu
def test_reax_closure_verdict_on_this_host(tmp_path):
    """the reax case: both variants of the valence-angle gradient are evaluated; with a LAMMPS (USER-REAXC) the closer one is named and
    1e-4 is demanded of it, without one the verdict says so"""
    import export_lammps_case as x
    from scema_amd.systems import synthetic_strains
    r = x.reax_replica((3, 5, 9))
    lens = r["box"][3:6] - r["box"][:3]
    res = x.verify_reax(str(tmp_path), r, synthetic_strains(1, lens, seed=2026)[0], os.environ.get("SCEMA_SCRIPTS_REAX", "/root/reference/lammps_scripts/lammps_scripts_reax"))
    print(res["verdict"])
    assert res["gpu_exact"] is not None and res["gpu_drop_dsbo2"] is not None
    dev = np.abs(np.array(res["gpu_exact"]) - np.array(res["gpu_drop_dsbo2"])).max() / np.abs(np.array(res["gpu_exact"])).max()
    assert dev > 1e-6          # the two variants are different answers: only LAMMPS can say which is the reference's
    if res["lammps"] is None:
        assert res["verdict"].startswith("invariant-verified")
    else:
        assert min(res["rel_err_vs_lammps"][k] for k in ("gpu_exact", "gpu_drop_dsbo2")) < 1e-4, res["verdict"]


def test_static_case_inputs_and_the_term_by_term_comparison(small_pe, tmp_path):
    """SURVEY 7 (iii) / VERDICT r4: the static inputs issue the reference's settings by path, `run 0`, LAMMPS' per-style energies, both
    pressure forms and a force dump -- once as the scripts stand and once with `pair_modify table 0` AFTER the restart is read (the file
    carries the table bits) -- and the comparison names the first term that differs.  LAMMPS' output files are imitated here."""
    import export_lammps_case as x
    lens = small_pe["box"][3:6] - small_pe["box"][:3]
    case = x.export(str(tmp_path), small_pe, np.array([1e-3 * lens[0], 0, 0, 0, 0, 0]), "/ref/lammps_scripts_opls", nss=20)
    assert case["static_inputs"] == ["static.lammps", "static_table0.lammps"]
    for tag in ("static", "static_table0"):
        a = open(tmp_path / (tag + ".lammps")).read().splitlines()
        want = ["include /ref/lammps_scripts_opls/in.set.lammps", "read_restart init.g0_1.bin", "compute vir all pressure NULL virial",
                "thermo_style custom step ebond eangle edihed eimp evdwl ecoul elong pe pxx pyy pzz pxy pxz pyz c_vir[1] c_vir[2] c_vir[3] c_vir[4] c_vir[5] c_vir[6]",
                "run 0", f"write_dump all custom {tag}.forces id fx fy fz modify sort id format float %.15g"]
        pos = [a.index(w) for w in want]
        assert pos == sorted(pos)
        has_t0 = [k for k, l in enumerate(a) if l.startswith("pair_modify table 0")]
        assert (len(has_t0) == 1 and a.index("read_restart init.g0_1.bin") < has_t0[0] < a.index("run 0")) if tag == "static_table0" else not has_t0
        assert not any(l.startswith("fix ") for l in a)             # a force-field evaluation: no SHAKE, no thermostat
    # the comparison: imitate LAMMPS' two output files from a set of numbers, perturb one term, and ask which differs first
    rng = np.random.default_rng(5)
    n = small_pe["natoms"]
    mine = dict(terms=dict(zip(x.STATIC_TERMS, [120.0, 340.0, 55.0, 0.0, -900.0, 4000.0, -3800.0, -185.0])),
                press=dict(zip(x.STATIC_PRESS, rng.normal(0, 3000, 6))), vir=dict(zip([f"c_vir[{k}]" for k in range(1, 7)], rng.normal(0, 3000, 6))),
                forces=rng.normal(0, 20, (n, 3)))
    def write(tag, terms, forces):
        vals = [terms[t] for t in x.STATIC_TERMS] + [mine["press"][p] for p in x.STATIC_PRESS] + [mine["vir"][f"c_vir[{k}]"] for k in range(1, 7)]
        open(tmp_path / (tag + ".out"), "w").write("SCEMA_STATIC " + " ".join("%.15g" % v for v in vals) + "\n")
        with open(tmp_path / (tag + ".forces"), "w") as f:
            f.write(f"ITEM: TIMESTEP\n0\nITEM: NUMBER OF ATOMS\n{n}\nITEM: BOX BOUNDS xy xz yz pp pp pp\n0 1 0\n0 1 0\n0 1 0\nITEM: ATOMS id fx fy fz\n")
            for i in range(n):
                f.write(f"{i + 1} " + " ".join("%.15g" % v for v in forces[i]) + "\n")
    write("same", mine["terms"], mine["forces"])
    vals, forces = x.read_static(str(tmp_path), "same")
    rows, first = x.compare_static(vals, forces, mine, 1e-7)
    assert first is None and len(rows) == 8 + 6 + 6 + 1 and max(r[3] for r in rows) < 1e-13
    t2 = dict(mine["terms"], ecoul=mine["terms"]["ecoul"] * (1 + 3e-6), pe=mine["terms"]["pe"] + 3e-6 * mine["terms"]["ecoul"])   # an erfc table's size of error
    write("tab", t2, mine["forces"] * (1 + 1e-6))
    vals, forces = x.read_static(str(tmp_path), "tab")
    assert x.compare_static(vals, forces, mine, 1e-7)[1] == "ecoul"      # ... is found in the coulomb term first, not in the sum
    assert x.compare_static(vals, forces, mine, 1e-4)[1] is None         # and is inside the north star's tolerance
    write("frc", mine["terms"], mine["forces"] + 1e-3)
    vals, forces = x.read_static(str(tmp_path), "frc")
    assert x.compare_static(vals, forces, mine, 1e-7)[1] == "forces"


def test_static_reax_input(tmp_path):
    import export_lammps_case as x
    from scema_amd.systems import synthetic_strains
    r = x.reax_replica((2, 3, 5))
    lens = r["box"][3:6] - r["box"][:3]
    case = x.export_reax(str(tmp_path), r, synthetic_strains(1, lens, seed=2026)[0], "/ref/lammps_scripts_reax")
    assert case["static_inputs"] == ["static.lammps"]
    a = open(tmp_path / "static.lammps").read().splitlines()
    want = ["include /ref/lammps_scripts_reax/in.set.lammps", "read_restart init.g0_1.bin", "pair_coeff * * /ref/lammps_scripts_reax/ffield.reax.2 H C N O",
            "fix qeq all qeq/reax 1 0.0 10.0 1e-6 reax/c", "compute reax all pair reax/c", "run 0",
            "write_dump all custom static.forces id q fx fy fz modify sort id format float %.15g"]
    pos = [a.index(w) for w in want]
    assert pos == sorted(pos)
    assert " ".join(f"c_reax[{k}]" for k in range(1, 15)) in open(tmp_path / "static.lammps").read()
    assert len(x.REAX_TERMS) == 14


@pytest.mark.gpu
def test_static_terms_of_the_engine_for_the_closure_kit(small_pe):
    """what --run puts next to LAMMPS' `run 0` line: the engine's per-style energies, pressure with and without the kinetic part and
    forces, in LAMMPS' names and units -- checked against the oracle's static evaluation so that the mapping of parts to names is pinned"""
    import export_lammps_case as x
    from oracle import pyoracle as po
    from scema_amd.systems import build_pe
    small_pe = build_pe(4, 6, 12, jitter=0.05, seed=11, shake_project=True)   # the smallest box the reference's 12 + 2 A list radius fits twice
    mine = x.static_ours(small_pe)
    o = po.Oracle(small_pe)
    o.setup(use_shake=False)
    f, en, w = o.compute()
    names = dict(evdwl=0, ecoul=1, ebond=2, eangle=3, edihed=4, eimp=5, elong=6)
    for n, k in names.items():
        assert abs(mine["terms"][n] - en[k]) <= 1e-9 * max(1.0, abs(en[k])), n
    assert abs(mine["terms"]["pe"] - sum(en[:7])) < 1e-8 * abs(sum(en[:7]))
    assert np.abs(mine["forces"] - f).max() < 1e-9 * np.abs(f).max()
    vol = float(np.prod(small_pe["box"][3:6] - small_pe["box"][:3]))
    assert np.abs(np.array(list(mine["vir"].values())) - np.asarray(w).sum(0) / vol * 68568.415).max() < 1e-6 * np.abs(list(mine["vir"].values())).max()
    assert all(np.isfinite(v) for v in mine["press"].values())
