#!/usr/bin/env python3
"""LAMMPS closure kit: export one stress evaluation of the synthetic replica as a case that the REFERENCE's own scripts run.

No LAMMPS exists on the build or GPU images of this project, so the oracle of the MD path is pinned by invariants only
(DESIGN.md §2).  This tool turns "invariant-verified" into "LAMMPS-verified" on any host that has a LAMMPS executable
with the KSPACE, MOLECULE and RIGID packages (the reference pins 17Nov16, README.md:31-37):

  <out>/replica.data           the replica (atom_style full, `write_data` layout)
  <out>/make_init.lammps       read_data -> write_restart init.<mat>_<rep>.bin   (what init_material leaves behind)
  <out>/phase_a.lammps         the commands STMDProblem::lammps_straining issues before and after in.strain.lammps
                               (stmd_problem.h:159-258): variables mdt / locs / tempt, include in.set.lammps, read_restart,
                               dts / nts / ceeps_kl ("%.6e"), include in.strain.lammps, write_restart last.*
  <out>/phase_b.lammps         the second LAMMPS lifetime (stmd_problem.h:284-341): in.set, read_restart last.*,
                               reset_timestep 0, locbe / nssample0, include ELASTIC/in.homogenization.lammps, print pp11..pp23
  <out>/case.json              strain (Angstrom-valued MDSim.strain), nts, rates, and -- when they can be computed here --
                               the stresses of the CPU oracle and of the GPU engine for the same request

`--run` executes the three inputs with the LAMMPS found (`--lmp`, $SCEMA_LAMMPS, or lmp / lmp_serial / lmp_mpi on PATH),
converts sigma = -pp * 101325 Pa (stmd_problem.h:335-341) and prints the comparison.  The reference scripts are included
by path (`--scripts`, default $SCEMA_SCRIPTS or /root/reference/lammps_scripts/lammps_scripts_opls): nothing of them is
copied.  The reference asks for `kspace_style pppm 1e-4`, and so do the oracle and the engine by default (DESIGN.md §2,
deviation 1): besides the stresses, `--run` reports the PPPM grid and G vector LAMMPS prints in phase_a.log next to the ones the
restated rules of pppm.cpp give here -- the one part of the PPPM path that only LAMMPS itself can pin.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def find_lammps(explicit=None):
    for c in (explicit, os.environ.get("SCEMA_LAMMPS"), "lmp", "lmp_serial", "lmp_mpi", "lammps"):
        if c and shutil.which(c):
            return shutil.which(c)
    return None


def request(d, strain_len, dt=2.0, rate=1e-4):
    """host arithmetic of stmd_problem.h:213-244 for a box of the replica's lengths"""
    lb = d["box"][3:6] - d["box"][:3]
    e = np.array([strain_len[0] / lb[0], strain_len[1] / lb[1], strain_len[2] / lb[2], strain_len[3] / lb[2], strain_len[4] / lb[1],
                  strain_len[5] / lb[0]])
    nrm = np.sqrt((e[:3] ** 2).sum() + 2.0 * (e[3:] ** 2).sum())
    nts = max(int(np.ceil(nrm / rate / dt / 10.0) * 10), 10)
    rates = [float("%.6e" % (v / (nts * dt))) for v in e]
    return nts, rates


def export(out, d, strain_len, scripts, mat="g0", rep=1, qp=0, dt=2.0, temperature=300.0, rate=1e-4, nss=100):
    from scema_amd.systems import write_lammps_data
    os.makedirs(out, exist_ok=True)
    write_lammps_data(os.path.join(out, "replica.data"), d)
    nts, rates = request(d, strain_len, dt, rate)
    state = f"{mat}_{rep}"
    with open(os.path.join(out, "make_init.lammps"), "w") as f:
        f.write(f"""# replica -> init.{state}.bin, the file STMDProblem reads first (stmd_problem.h:99-100,204)
variable locs string {scripts}
include {scripts}/in.set.lammps
special_bonds lj/coul 0.0 0.0 1.0   # in.init.lammps:31, carried by the restart file from here on
read_data replica.data
write_restart init.{state}.bin
""")
    name = {0: "00", 1: "11", 2: "22", 3: "01", 4: "02", 5: "12"}
    with open(os.path.join(out, "phase_a.lammps"), "w") as f:
        f.write(f"""# LAMMPS lifetime 1 of STMDProblem::lammps_straining (stmd_problem.h:156-275)
variable mdt string {mat}
variable locs string {scripts}
variable tempt equal {temperature:f}
include {scripts}/in.set.lammps
read_restart init.{state}.bin
print 'initially computed'
variable ll1 equal lx
variable ll2 equal ly
variable ll3 equal lz
variable dts equal {dt:f}
variable nts equal {nts}
""")
        for k in range(6):
            f.write(f"variable ceeps_{name[k]} equal {rates[k]:.6e}\n")
        f.write(f"include {scripts}/in.strain.lammps\nwrite_restart last.{qp}.{state}.dump\n")
    with open(os.path.join(out, "phase_b.lammps"), "w") as f:
        f.write(f"""# LAMMPS lifetime 2 (stmd_problem.h:284-369)
variable mdt string {mat}
variable locs string {scripts}
variable tempt equal {temperature:f}
include {scripts}/in.set.lammps
read_restart last.{qp}.{state}.dump
reset_timestep 0
variable dts equal {dt:f}
variable locbe string {scripts}/ELASTIC
variable nssample0 equal {nss}
variable nssample  equal {nss}
include {scripts}/ELASTIC/in.homogenization.lammps
print "SCEMA_PP ${{pp11}} ${{pp22}} ${{pp33}} ${{pp12}} ${{pp13}} ${{pp23}}" file pp.out
""")
    case = dict(strain_len=[float(v) for v in strain_len], nts=nts, rates=rates, dt=dt, temperature=temperature, strain_rate=rate, nss=nss,
                natoms=int(d["natoms"]), scripts=scripts)
    json.dump(case, open(os.path.join(out, "case.json"), "w"), indent=1)
    return case


def run_lammps(lmp, out, log=True):
    """-> stress[6] in Pa, raw order xx,yy,zz,xy,xz,yz (stmd_problem.h:335-341)"""
    for inp in ("make_init.lammps", "phase_a.lammps", "phase_b.lammps"):
        r = subprocess.run([lmp, "-in", inp, "-log", inp.replace(".lammps", ".log") if log else "none", "-screen", "none"], cwd=out)
        if r.returncode != 0:
            raise RuntimeError(f"{lmp} -in {inp} failed (rc={r.returncode}); see {out}")
    pp = [float(v) for v in open(os.path.join(out, "pp.out")).read().split()[1:7]]
    return -np.array(pp) * 1.01325e5


def pppm_from_log(path):
    """the first `G vector (1/distance) = g` / `grid = nx ny nz` pair of a LAMMPS log (PPPM initialisation) or None"""
    import re
    try:
        txt = open(path).read()
    except OSError:
        return None
    g = re.search(r"G vector \(1/distance\)\s*=\s*([0-9.eE+-]+)", txt)
    n = re.search(r"grid\s*=\s*(\d+)\s+(\d+)\s+(\d+)", txt)
    if not g or not n:
        return None
    return dict(g_ewald=float(g.group(1)), grid=[int(n.group(k)) for k in (1, 2, 3)])


def pppm_ours(d):
    """grid and g_ewald of the restated set-up rules for the replica's initial box (oracle/md_oracle.c pppm_setup) or None"""
    try:
        from oracle import pyoracle as po
        o = po.Oracle(d)
        o.setup()
        return dict(g_ewald=float(o.g_ewald), grid=[int(v) for v in o.pppm_grid])
    except Exception as exc:
        print("oracle not available:", exc)
        return None


def ours(d, case, want_gpu=True):
    """(oracle stress or None, GPU engine stress or None) for the same request"""
    o = g = None
    try:
        from oracle import pyoracle as po
        o, _ = po.Oracle(d).eval(case["strain_len"], case["dt"], case["temperature"], case["strain_rate"], case["nss"])
    except Exception as exc:   # the oracle is test infrastructure: absent in a product checkout
        print("oracle not available:", exc)
    if want_gpu:
        try:
            import torch
            if torch.cuda.is_available():
                from scema_amd import capi
                e = capi.Engine()
                e.register_replica("g0", 1, d)
                sim = capi.make_sim(0, "g0", 1, case["strain_len"], nss=case["nss"], dt=case["dt"], temperature=case["temperature"],
                                    strain_rate=case["strain_rate"], most_recent=capi.QP_NONE)
                g = np.array(e.strain_batch([sim])[0].stress[:])
                e.close()
        except Exception as exc:
            print("GPU engine not available:", exc)
    return o, g


def verify(out, d, strain_len, scripts, lmp=None, **kw):
    """export, run LAMMPS if there is one, compare; returns a dict with the verdict string"""
    case = export(out, d, strain_len, scripts, **kw)
    lmp = find_lammps(lmp)
    o, g = ours(d, case)
    res = dict(case=case, oracle=None if o is None else [float(v) for v in o], gpu=None if g is None else [float(v) for v in g], lammps=None)
    if lmp is None:
        res["verdict"] = "invariant-verified (no LAMMPS executable on this host; run tools/export_lammps_case.py --run where one exists)"
    elif not os.path.isdir(scripts):
        res["verdict"] = f"invariant-verified (LAMMPS found at {lmp}, but the reference scripts are not at {scripts})"
    else:
        s = run_lammps(lmp, out)
        res["lammps"] = [float(v) for v in s]
        errs = {k: float(np.abs(np.array(v) - s).max() / np.abs(s).max()) for k, v in (("oracle", res["oracle"]), ("gpu", res["gpu"])) if v is not None}
        res["rel_err_vs_lammps"] = errs
        res["pppm"] = dict(lammps=pppm_from_log(os.path.join(out, "phase_a.log")), ours=pppm_ours(d))
        res["verdict"] = "LAMMPS-verified: " + ", ".join(f"{k} max rel err {v:.3e}" for k, v in errs.items()) + f" ({lmp}); PPPM set-up {res['pppm']}"
    json.dump(res, open(os.path.join(out, "verdict.json"), "w"), indent=1)
    return res


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--out", default="lammps_case")
    ap.add_argument("--cells", type=int, nargs=3, default=[6, 9, 16])
    ap.add_argument("--strain-index", type=int, default=0, help="which of the synthetic strains of bench.py (seed 2026)")
    ap.add_argument("--scripts", default=os.environ.get("SCEMA_SCRIPTS", "/root/reference/lammps_scripts/lammps_scripts_opls"))
    ap.add_argument("--lmp", default=None)
    ap.add_argument("--nss", type=int, default=100)
    ap.add_argument("--run", action="store_true", help="run LAMMPS (if found) and compare")
    a = ap.parse_args()
    from scema_amd.systems import build_pe, synthetic_strains
    d = build_pe(*a.cells, shake_project=True)
    lens = d["box"][3:6] - d["box"][:3]
    strain = synthetic_strains(max(a.strain_index + 1, 1), lens, seed=2026)[a.strain_index]
    if a.run:
        res = verify(a.out, d, strain, a.scripts, lmp=a.lmp, nss=a.nss)
        print(res["verdict"])
    else:
        case = export(a.out, d, strain, a.scripts, nss=a.nss)
        print(f"wrote {a.out}/ (nts {case['nts']}, rates {case['rates']}); run:  cd {a.out} && for f in make_init phase_a phase_b; do lmp -in $f.lammps; done")


if __name__ == "__main__":
    main()
