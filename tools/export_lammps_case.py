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

`--force-field reax` exports BASELINE config 5's case instead (lammps_scripts_reax: `atom_style charge` data file with types H C N O,
`pair_style reax/c` + `fix qeq/reax 1 0.0 10.0 1e-6`, no SHAKE, no k-space; dt 0.25 fs): the state travels between the two LAMMPS
lifetimes as the reference's text dump (`write_dump all custom last.* id type xs ys zs vx vy vz ix iy iz`, read back by
`read_restart init.bin` + `rerun ... dump x y z vx vy vz ix iy iz box yes scaled yes wrapped yes format native`,
stmd_problem.h:190-194,261-264,297-302).  With `--run` the engine's stress is computed twice -- with the exact gradient (the
default) and with the valence-angle term USER-REAXC is believed to drop (SCEMA_REAX_DROP_DSBO2=1) -- and both are
compared with LAMMPS, which settles which of the two the reference computes; needs a LAMMPS with USER-REAXC.

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


def reax_replica(cells=(3, 5, 9)):
    """the ReaxFF replica of bench.py --force-field reax: the polyethylene cell with LAMMPS types 1..4 = H C N O as `pair_coeff * *
    ffield.reax.2 H C N O` names them, thermal velocities (seed 3), zero charges (fix qeq/reax sets them)"""
    from scema_amd.systems import build_pe
    d = build_pe(*cells)
    sym = ["C" if d["mass"][t] > 5 else "H" for t in d["type"]]
    m = np.array([12.011 if c == "C" else 1.008 for c in sym])
    v = np.random.default_rng(3).standard_normal((len(sym), 3)) * np.sqrt(0.0019872067 * 300.0 / (m[:, None] * 48.88821291 ** 2))
    v -= (m[:, None] * v).sum(0) / m.sum()
    return dict(natoms=len(sym), sym=sym, lmp_type=np.array([2 if c == "C" else 1 for c in sym]), x=d["x"], v=v, box=d["box"])


def write_charge_data(path, r):
    """`write_data` layout of atom_style charge: id type q x y z (the reax scripts, lammps_scripts_reax/in.set.lammps:17)"""
    b = [float(v) for v in r["box"]]
    with open(path, "w") as f:
        f.write(f"LAMMPS data file: scema_amd synthetic ReaxFF replica\n\n{r['natoms']} atoms\n4 atom types\n\n")
        f.write(f"{b[0]!r} {b[3]!r} xlo xhi\n{b[1]!r} {b[4]!r} ylo yhi\n{b[2]!r} {b[5]!r} zlo zhi\n{b[6]!r} {b[7]!r} {b[8]!r} xy xz yz\n\n")
        f.write("Masses\n\n1 1.008\n2 12.011\n3 14.007\n4 15.999\n\nAtoms # charge\n\n")
        for i in range(r["natoms"]):
            f.write(f"{i + 1} {int(r['lmp_type'][i])} 0.0 {float(r['x'][i, 0])!r} {float(r['x'][i, 1])!r} {float(r['x'][i, 2])!r}\n")
        f.write("\nVelocities\n\n")
        for i in range(r["natoms"]):
            f.write(f"{i + 1} {float(r['v'][i, 0])!r} {float(r['v'][i, 1])!r} {float(r['v'][i, 2])!r}\n")


def export_reax(out, r, strain_len, scripts, mat="g0", rep=1, qp=0, dt=0.25, temperature=300.0, rate=1e-3, nss=20):
    """the reax branch of STMDProblem::lammps_straining (stmd_problem.h:92-94,190-194,261-264,297-302) as three LAMMPS inputs"""
    os.makedirs(out, exist_ok=True)
    write_charge_data(os.path.join(out, "replica.data"), r)
    nts, rates = request(r, strain_len, dt, rate)
    state = f"{mat}_{rep}"
    rerun = "dump x y z vx vy vz ix iy iz box yes scaled yes wrapped yes format native"
    with open(os.path.join(out, "make_init.lammps"), "w") as f:
        f.write(f"""# replica -> init.{state}.bin (what init_material leaves behind; reax: in.init.lammps of lammps_scripts_reax)
variable locs string {scripts}
include {scripts}/in.set.lammps
read_data replica.data
pair_style reax/c NULL safezone 50.0 mincap 100000
pair_coeff * * {scripts}/ffield.reax.2 H C N O
write_restart init.{state}.bin
""")
    name = {0: "00", 1: "11", 2: "22", 3: "01", 4: "02", 5: "12"}
    with open(os.path.join(out, "phase_a.lammps"), "w") as f:
        f.write(f"""# LAMMPS lifetime 1 of STMDProblem::lammps_straining, md_force_field "reax" (stmd_problem.h:156-275)
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
        f.write(f"include {scripts}/in.strain.lammps\nwrite_dump all custom last.{qp}.{state}.dump id type xs ys zs vx vy vz ix iy iz\n")
    with open(os.path.join(out, "phase_b.lammps"), "w") as f:
        f.write(f"""# LAMMPS lifetime 2 (stmd_problem.h:284-369): the state comes back through rerun on top of init.bin
variable mdt string {mat}
variable locs string {scripts}
variable tempt equal {temperature:f}
include {scripts}/in.set.lammps
read_restart init.{state}.bin
rerun last.{qp}.{state}.dump {rerun}
reset_timestep 0
variable dts equal {dt:f}
variable locbe string {scripts}/ELASTIC
variable nssample0 equal {nss}
variable nssample  equal {nss}
include {scripts}/ELASTIC/in.homogenization.lammps
print "SCEMA_PP ${{pp11}} ${{pp22}} ${{pp33}} ${{pp12}} ${{pp13}} ${{pp23}}" file pp.out
""")
    case = dict(force_field="reax", strain_len=[float(v) for v in strain_len], nts=nts, rates=rates, dt=dt, temperature=temperature, strain_rate=rate,
                nss=nss, natoms=int(r["natoms"]), scripts=scripts,
                note="lammps_scripts_reax/ELASTIC/in.homogenization.lammps:61-62 is shipped with a line broken inside c_thermo_press[6] (SURVEY Appendix C 9): "
                     "join the two lines in a copy of the scripts before running")
    json.dump(case, open(os.path.join(out, "case.json"), "w"), indent=1)
    return case


def ours_reax(r, case, scripts):
    """GPU engine stresses for the same request: exact gradient (default) and the variant that drops dSBO2; oracle if importable"""
    res = dict(oracle=None, gpu_exact=None, gpu_drop_dsbo2=None)
    ff = os.path.join(scripts, "ffield.reax.2")
    if not os.path.exists(ff):
        ff = os.path.join(ROOT, "examples", "ffield.reax.2")
    try:
        import torch
        if torch.cuda.is_available():
            from scema_amd import capi
            for key, exact in (("gpu_exact", 1), ("gpu_drop_dsbo2", 0)):
                e = capi.Engine()
                e.reax_configure(ff, qeq_tol=1e-6)
                e.reax_set(exact_gradient=exact)
                e.register_replica("g0", 1, capi.reax_system(r["sym"], r["x"], r["box"], v=r["v"]))
                sim = capi.make_sim(0, "g0", 1, case["strain_len"], nss=case["nss"], dt=case["dt"], temperature=case["temperature"],
                                    strain_rate=case["strain_rate"], most_recent=capi.QP_NONE, force_field="reax")
                res[key] = [float(v) for v in e.strain_batch([sim])[0].stress[:]]
                e.close()
    except Exception as exc:
        print("GPU engine not available:", exc)
    try:
        from oracle import reax_md
        lt = np.array([1 if c == "C" else 0 for c in r["sym"]])
        M = reax_md.ReaxMD(ff, ["H", "C", "N", "O"], lt, [1.008, 12.011, 14.007, 15.999], r["box"], r["x"], r["v"])
        s, _ = M.eval(np.array(case["strain_len"]), case["dt"], case["temperature"], case["strain_rate"], case["nss"])
        res["oracle"] = [float(v) for v in s]
    except Exception as exc:
        print("oracle not available:", exc)
    return res


def verify_reax(out, r, strain_len, scripts, lmp=None, **kw):
    case = export_reax(out, r, strain_len, scripts, **kw)
    lmp = find_lammps(lmp)
    res = dict(case=case, lammps=None, **ours_reax(r, case, scripts))
    if lmp is None:
        res["verdict"] = "invariant-verified (no LAMMPS executable on this host; run tools/export_lammps_case.py --force-field reax --run where one with USER-REAXC exists)"
    elif not os.path.isdir(scripts):
        res["verdict"] = f"invariant-verified (LAMMPS found at {lmp}, but the reference scripts are not at {scripts})"
    else:
        s = run_lammps(lmp, out)
        res["lammps"] = [float(v) for v in s]
        errs = {k: float(np.abs(np.array(res[k]) - s).max() / np.abs(s).max()) for k in ("oracle", "gpu_exact", "gpu_drop_dsbo2") if res[k] is not None}
        res["rel_err_vs_lammps"] = errs
        closer = min((k for k in ("gpu_exact", "gpu_drop_dsbo2") if k in errs), key=lambda k: errs[k], default=None)
        res["verdict"] = ("LAMMPS-verified: " + ", ".join(f"{k} max rel err {v:.3e}" for k, v in errs.items()) + f" ({lmp})" +
                          (f"; the reference's valence-angle gradient is the '{closer}' variant" if closer else ""))
    json.dump(res, open(os.path.join(out, "verdict.json"), "w"), indent=1)
    return res


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
    ap.add_argument("--force-field", default="opls", choices=["opls", "reax"])
    ap.add_argument("--cells", type=int, nargs=3, default=None, help="PE supercell (default 6 9 16 for opls, 3 5 9 for reax)")
    ap.add_argument("--strain-index", type=int, default=0, help="which of the synthetic strains of bench.py (seed 2026)")
    ap.add_argument("--scripts", default=None, help="the reference's script folder (default $SCEMA_SCRIPTS or /root/reference/lammps_scripts/lammps_scripts_<force field>)")
    ap.add_argument("--lmp", default=None)
    ap.add_argument("--nss", type=int, default=None, help="sampling steps (default 100 for opls, 20 for reax)")
    ap.add_argument("--run", action="store_true", help="run LAMMPS (if found) and compare")
    a = ap.parse_args()
    from scema_amd.systems import build_pe, synthetic_strains
    reax = a.force_field == "reax"
    a.scripts = a.scripts or os.environ.get("SCEMA_SCRIPTS") or f"/root/reference/lammps_scripts/lammps_scripts_{a.force_field}"
    a.cells = a.cells or ([3, 5, 9] if reax else [6, 9, 16])
    a.nss = a.nss or (20 if reax else 100)
    if reax:
        r = reax_replica(tuple(a.cells))
        lens = r["box"][3:6] - r["box"][:3]
        strain = synthetic_strains(max(a.strain_index + 1, 1), lens, seed=2026)[a.strain_index]
        if a.run:
            print(verify_reax(a.out, r, strain, a.scripts, lmp=a.lmp, nss=a.nss)["verdict"])
        else:
            case = export_reax(a.out, r, strain, a.scripts, nss=a.nss)
            print(f"wrote {a.out}/ (reax; nts {case['nts']}, rates {case['rates']}); run:  cd {a.out} && for f in make_init phase_a phase_b; do lmp -in $f.lammps; done")
        return
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
