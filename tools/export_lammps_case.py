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
  <out>/static.lammps          the STATIC case: in.set.lammps, read_restart init.*, `run 0` with LAMMPS' per-style energies
                               (pe evdwl ecoul elong ebond eangle edihed eimp), the six pressure components with and without the
                               kinetic part, and a force dump -- once as the scripts stand (pair_modify table 12, the 17Nov16 default)
  <out>/static_table0.lammps   ... and once with `pair_modify table 0` (analytic erfc, what the engine computes).  `--run` compares both
                               term by term with scema_md_debug_compute and names the FIRST term that differs: a miss of the
                               time-averaged stress then says whether the force field, the PPPM set-up, the erfc table or the
                               dynamics (SHAKE, thermostat, fix deform) is where the two part ways.

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
    static = export_static_reax(out, scripts, mat, rep, temperature)
    case = dict(force_field="reax", strain_len=[float(v) for v in strain_len], nts=nts, rates=rates, dt=dt, temperature=temperature, strain_rate=rate,
                nss=nss, natoms=int(r["natoms"]), scripts=scripts, static_inputs=static,
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
        res["static"] = run_static_reax(lmp, out, r, scripts)
        firsts = {k: v.get("first_term_that_differs") for k, v in (res["static"] or {}).items() if isinstance(v, dict)}
        res["verdict"] = ("LAMMPS-verified: " + ", ".join(f"{k} max rel err {v:.3e}" for k, v in errs.items()) + f" ({lmp})" +
                          (f"; the reference's valence-angle gradient is the '{closer}' variant" if closer else "") +
                          f"; static case, first term that differs: {firsts}")
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
    static = export_static(out, scripts, mat, rep, temperature)
    case = dict(strain_len=[float(v) for v in strain_len], nts=nts, rates=rates, dt=dt, temperature=temperature, strain_rate=rate, nss=nss,
                natoms=int(d["natoms"]), scripts=scripts, static_inputs=static)
    json.dump(case, open(os.path.join(out, "case.json"), "w"), indent=1)
    return case


# ---- the static case: one force evaluation, term by term (lammps_scripts_opls/in.set.lammps:36,42) ----
STATIC_TERMS = ["ebond", "eangle", "edihed", "eimp", "evdwl", "ecoul", "elong", "pe"]
STATIC_PRESS = ["pxx", "pyy", "pzz", "pxy", "pxz", "pyz"]


def export_static(out, scripts, mat="g0", rep=1, temperature=300.0):
    """static.lammps / static_table0.lammps: the reference's settings (in.set.lammps by path), the init restart, `run 0`"""
    state = f"{mat}_{rep}"
    cols = " ".join(STATIC_TERMS) + " " + " ".join(STATIC_PRESS) + " " + " ".join(f"c_vir[{k}]" for k in range(1, 7))
    prn = " ".join(f"$({t})" for t in STATIC_TERMS) + " " + " ".join(f"$({t})" for t in STATIC_PRESS) + " " + " ".join(f"$(c_vir[{k}])" for k in range(1, 7))
    for tag, extra in (("static", ""), ("static_table0", "pair_modify table 0   # analytic erfc instead of the 4096-entry table (after read_restart: the file carries the table bits)\n")):
        with open(os.path.join(out, tag + ".lammps"), "w") as f:
            f.write(f"""# one force evaluation of init.{state}.bin under the reference's settings, LAMMPS' own terms printed (no fix: no SHAKE force, no thermostat)
variable mdt string {mat}
variable locs string {scripts}
variable tempt equal {temperature:f}
include {scripts}/in.set.lammps
read_restart init.{state}.bin
{extra}compute vir all pressure NULL virial
thermo_style custom step {cols}
thermo_modify format float %.15g
run 0
print "SCEMA_STATIC {prn}" file {tag}.out
write_dump all custom {tag}.forces id fx fy fz modify sort id format float %.15g
""")
    return ["static.lammps", "static_table0.lammps"]


def export_static_reax(out, scripts, mat="g0", rep=1, temperature=300.0):
    """the ReaxFF static case: per-term energies of `compute pair reax/c` (c_reax[1..14]: eb ea elp emol ev epen ecoa ehb et eco ew ep efi eqeq),
    the charges fix qeq/reax settles on (lammps_scripts_reax/in.strain.lammps:12), virial pressure, forces"""
    state = f"{mat}_{rep}"
    cols = "pe " + " ".join(f"c_reax[{k}]" for k in range(1, 15)) + " " + " ".join(f"c_vir[{k}]" for k in range(1, 7))
    prn = "$(pe) " + " ".join(f"$(c_reax[{k}])" for k in range(1, 15)) + " " + " ".join(f"$(c_vir[{k}])" for k in range(1, 7))
    with open(os.path.join(out, "static.lammps"), "w") as f:
        f.write(f"""# one force evaluation of init.{state}.bin with pair reax/c + fix qeq/reax as the reference's scripts set them
variable mdt string {mat}
variable locs string {scripts}
variable tempt equal {temperature:f}
include {scripts}/in.set.lammps
read_restart init.{state}.bin
pair_style reax/c NULL safezone 50.0 mincap 100000
pair_coeff * * {scripts}/ffield.reax.2 H C N O
fix qeq all qeq/reax 1 0.0 10.0 1e-6 reax/c
compute reax all pair reax/c
compute vir all pressure NULL virial
thermo_style custom step {cols}
thermo_modify format float %.15g
run 0
print "SCEMA_STATIC {prn}" file static.out
write_dump all custom static.forces id q fx fy fz modify sort id format float %.15g
""")
    return ["static.lammps"]


def read_static(out, tag, ncol_dump=3):
    """-> (numbers of the SCEMA_STATIC line, per-atom array of the dump sorted by id)"""
    vals = [float(v) for v in open(os.path.join(out, tag + ".out")).read().split()[1:]]
    rows, on = [], False
    for line in open(os.path.join(out, tag + ".forces")):
        if line.startswith("ITEM: ATOMS"):
            on = True
        elif line.startswith("ITEM:"):
            on = False
        elif on:
            rows.append([float(v) for v in line.split()[1:1 + ncol_dump]])
    return vals, np.array(rows)


def static_ours(d):
    """the same terms from the engine's parity hook (scema_md_debug_compute on the registered init state): kcal/mol, atm, kcal/mol/A"""
    import torch
    if not torch.cuda.is_available():
        return None
    from scema_amd import capi
    e = capi.Engine()
    e.register_replica("g0", 1, d)
    f, en, w, info = e.debug_compute("g0", 1)
    e.close()
    b = np.asarray(d["box"], float)
    vol = float(np.prod(b[3:6] - b[:3]))
    m = np.asarray(d["mass"])[np.asarray(d["type"])]
    v = np.asarray(d["v"])
    mvv2e, nktv2p = 48.88821291 ** 2, 68568.415
    ke = np.array([(m * v[:, a] * v[:, c]).sum() * mvv2e for a, c in ((0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2))])
    wsum = w.sum(0)   # every part: pair, bonded, k-space (no SHAKE in a static evaluation)
    terms = dict(ebond=en[2], eangle=en[3], edihed=en[4], eimp=en[5], evdwl=en[0], ecoul=en[1], elong=en[6], pe=float(en[:7].sum()))
    press = dict(zip(STATIC_PRESS, (ke + wsum) / vol * nktv2p))
    vir = dict(zip([f"c_vir[{k}]" for k in range(1, 7)], wsum / vol * nktv2p))
    return dict(terms=terms, press=press, vir=vir, forces=f, g_ewald=info["g_ewald"])


def compare_static(lmp_vals, lmp_forces, mine, tol):
    """term by term, in the order a difference would propagate (bonded terms, pair terms, k-space, sums, virial, forces): -> rows and
    the name of the first term whose relative difference exceeds tol (None: all agree)"""
    names = STATIC_TERMS + STATIC_PRESS + [f"c_vir[{k}]" for k in range(1, 7)]
    L = dict(zip(names, lmp_vals))
    ours = dict(mine["terms"]); ours.update(mine["press"]); ours.update(mine["vir"])
    rows, first = [], None
    scale_e = max(abs(L["pe"]), 1e-12)
    scale_p = max(max(abs(L[k]) for k in STATIC_PRESS), 1e-12)
    for n in names:
        ref = scale_e if n in STATIC_TERMS else scale_p   # relative to the total: a term that is zero in both is not "different"
        rel = abs(L[n] - ours[n]) / max(abs(L[n]), 1e-3 * ref)
        rows.append((n, L[n], float(ours[n]), float(rel)))
        if first is None and rel > tol:
            first = n
    ff = np.abs(lmp_forces - mine["forces"]).max() / np.abs(lmp_forces).max()
    rows.append(("forces (max over atoms / max |f|)", float(np.abs(lmp_forces).max()), float(np.abs(mine["forces"]).max()), float(ff)))
    if first is None and ff > tol:
        first = "forces"
    return rows, first


def run_static(lmp, out, d, scripts):
    """run both static inputs and compare: as the scripts stand the erfc table limits the agreement to ~1e-6, with table 0 the
    force field must agree to ~1e-9 and what is left is the PPPM set-up"""
    mine = static_ours(d)
    if mine is None:
        return None
    res = {}
    for tag, tol in (("static", 1e-5), ("static_table0", 1e-7)):
        r = subprocess.run([lmp, "-in", tag + ".lammps", "-log", tag + ".log", "-screen", "none"], cwd=out)
        if r.returncode != 0:
            res[tag] = dict(error=f"{lmp} -in {tag}.lammps failed (rc={r.returncode})")
            continue
        vals, forces = read_static(out, tag)
        rows, first = compare_static(vals, forces, mine, tol)
        res[tag] = dict(tolerance=tol, first_term_that_differs=first, rows=rows, pppm_lammps=pppm_from_log(os.path.join(out, tag + ".log")))
    return res


REAX_TERMS = ["eb", "ea", "elp", "emol", "ev", "epen", "ecoa", "ehb", "et", "eco", "ew", "ep", "efi", "eqeq"]   # compute pair reax/c, c_reax[1..14]


def static_ours_reax(r, scripts, exact):
    """the engine's static ReaxFF evaluation in LAMMPS' term order (scema_md_reax_debug_compute: bond, lone pair, over, under, angle,
    penalty, 3-body conj., torsion, 4-body conj., hydrogen bond, van der Waals, Coulomb, polarisation)"""
    import torch
    if not torch.cuda.is_available():
        return None
    from scema_amd import capi
    ff = os.path.join(scripts, "ffield.reax.2")
    if not os.path.exists(ff):
        ff = os.path.join(ROOT, "examples", "ffield.reax.2")
    e = capi.Engine()
    e.reax_configure(ff, qeq_tol=1e-6)
    e.reax_set(exact_gradient=exact)
    e.register_replica("g0", 1, capi.reax_system(r["sym"], r["x"], r["box"], v=r["v"]))
    c = e.reax_compute("g0", 1)
    e.close()
    ep = list(c["e"].values())   # in the order of the C ABI's eparts[13]
    b = np.asarray(r["box"], float)
    vol = float(np.prod(b[3:6] - b[:3]))
    t = dict(eb=ep[0], elp=ep[1], ea=ep[2] + ep[3], emol=0.0, ev=ep[4], epen=ep[5], ecoa=ep[6], et=ep[7], eco=ep[8], ehb=ep[9], ew=ep[10], ep=ep[11],
             efi=0.0, eqeq=ep[12])
    return dict(terms=t, pe=float(np.sum(ep)), vir=np.asarray(c["w"]) / vol * 68568.415, q=np.asarray(c["q"]), forces=np.asarray(c["f"]))


def run_static_reax(lmp, out, r, scripts, tol=1e-6):
    """the ReaxFF static case against both gradient variants of the engine: energies and charges do not depend on the variant, forces and
    virial do -- the first place where 'exact' and 'drop dSBO2' can be told apart without any dynamics"""
    rr = subprocess.run([lmp, "-in", "static.lammps", "-log", "static.log", "-screen", "none"], cwd=out)
    if rr.returncode != 0:
        return dict(error=f"{lmp} -in static.lammps failed (rc={rr.returncode})")
    vals, per_atom = read_static(out, "static", ncol_dump=4)
    L = dict(zip(["pe"] + REAX_TERMS + [f"c_vir[{k}]" for k in range(1, 7)], vals))
    res = {}
    for key, exact in (("gpu_exact", 1), ("gpu_drop_dsbo2", 0)):
        mine = static_ours_reax(r, scripts, exact)
        if mine is None:
            return None
        rows, first = [], None
        scale = max(abs(L["pe"]), 1e-12)
        for n in REAX_TERMS + ["pe"]:
            o = mine["pe"] if n == "pe" else mine["terms"][n]
            rel = abs(L[n] - o) / max(abs(L[n]), 1e-3 * scale)
            rows.append((n, L[n], float(o), float(rel)))
            if first is None and rel > tol:
                first = n
        dq = float(np.abs(per_atom[:, 0] - mine["q"]).max())
        rows.append(("charges (max |dq|, e)", float(np.abs(per_atom[:, 0]).max()), float(np.abs(mine["q"]).max()), dq))
        if first is None and dq > 1e-5:   # the solver's tolerance is 1e-6 on the residual norm
            first = "charges"
        pv = np.array([L[f"c_vir[{k}]"] for k in range(1, 7)])
        rv = float(np.abs(pv - mine["vir"]).max() / np.abs(pv).max())
        rows.append(("virial pressure (max rel of 6)", float(np.abs(pv).max()), float(np.abs(mine["vir"]).max()), rv))
        ff = float(np.abs(per_atom[:, 1:4] - mine["forces"]).max() / np.abs(per_atom[:, 1:4]).max())
        rows.append(("forces (max over atoms / max |f|)", float(np.abs(per_atom[:, 1:4]).max()), float(np.abs(mine["forces"]).max()), ff))
        if first is None and max(rv, ff) > 1e-5:
            first = "forces / virial"
        res[key] = dict(first_term_that_differs=first, rows=rows)
    return res


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


def pppm_product(d):
    """the same from the PRODUCT (scema_md_kspace_setup: a host function of the C ABI, no GPU needed) or None"""
    try:
        from scema_amd import capi
        q = np.asarray(d["charge"], float)
        g0, g1, grid = capi.kspace_setup(capi.default_params(), np.asarray(d["box"], float), float((q ** 2).sum()), int(d["natoms"]))
        return dict(g_ewald=g1, grid=list(grid), g_initial=g0)
    except Exception as exc:
        print("library not available:", exc)
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
        res["pppm"] = dict(lammps=pppm_from_log(os.path.join(out, "phase_a.log")), ours=pppm_ours(d), product=pppm_product(d))
        res["static"] = run_static(lmp, out, d, scripts)
        firsts = {k: v.get("first_term_that_differs") for k, v in (res["static"] or {}).items()}
        res["verdict"] = ("LAMMPS-verified: " + ", ".join(f"{k} max rel err {v:.3e}" for k, v in errs.items()) + f" ({lmp}); PPPM set-up {res['pppm']}; "
                          f"static case, first term that differs (as the scripts stand / pair_modify table 0): {firsts.get('static')} / {firsts.get('static_table0')}")
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
