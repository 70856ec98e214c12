#!/usr/bin/env python3
"""Physics-level sanity check of the whole init_material chain on the GPU: the synthetic all-atom OPLS polyethylene crystal
(PE-10k, chains along z) is equilibrated with the schedule of in.init.lammps, then ELASTIC/in.modulus.lammps gives its
stiffness tensor.  Crystalline polyethylene is extremely anisotropic: literature values (experiment and all-atom force fields)
put the chain-axis stiffness C33 at 250-340 GPa and the transverse C11, C22 at 7-15 GPa.  One JSON line.
usage (GPU box): python3 tools/elastic_demo.py [--nsinit 300] [--nss 2000]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nsinit", type=int, default=300)
    ap.add_argument("--nss", type=int, default=2000)
    ap.add_argument("--dt", type=float, default=1.0)
    a = ap.parse_args()
    from scema_amd import capi
    from scema_amd.systems import build_pe
    d = build_pe(6, 9, 16, jitter=0.05, seed=11)
    e = capi.Engine()
    e.register_replica("pe", 1, d)
    t0 = time.time()
    length, info = e.equilibrate("pe", 1, a.nsinit, a.dt, 300.0)
    t1 = time.time()
    length2, stress, stiff = e.init_material("pe", 1, dt=a.dt, temperature=300.0, nss=a.nss, strain_ampl=0.005, strain_rate=1e-4)
    t2 = time.time()
    # file order of init.*.stiff rows/cols: 00,01,02,11,12,22 -> Voigt diagonal C11 = [0,0], C22 = [3,3], C33 = [5,5]
    gpa = stiff / 1e9
    print(json.dumps(dict(what="PE-10k: in.init.lammps schedule, then ELASTIC/in.modulus.lammps (13 runs in one batch)", nsinit=a.nsinit, nss=a.nss, dt_fs=a.dt,
                          equilibrate_s=t1 - t0, init_material_s=t2 - t1, box_lengths=list(length2),
                          initial_stress_MPa=list(stress / 1e6),
                          C11_GPa=gpa[0, 0], C22_GPa=gpa[3, 3], C33_GPa=gpa[5, 5], C12_GPa=gpa[0, 3], C13_GPa=gpa[0, 5], C23_GPa=gpa[3, 5],
                          shear_GPa=[gpa[1, 1], gpa[2, 2], gpa[4, 4]])))


if __name__ == "__main__":
    main()
