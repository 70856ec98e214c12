#!/usr/bin/env python3
"""init_material's equilibration schedule (in.init.lammps) on the synthetic PE-10k replica at the reference's cutoffs: wall time,
minimiser statistics, box lengths, temperature and pressure of the equilibrated state.  One JSON line.
usage (GPU box): python3 tools/equil_demo.py [--nsinit 100] [--dt 1.0] [--cells 6 9 16]"""
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
    ap.add_argument("--nsinit", type=int, default=100)
    ap.add_argument("--dt", type=float, default=1.0)
    ap.add_argument("--temperature", type=float, default=300.0)
    ap.add_argument("--cells", type=int, nargs=3, default=[6, 9, 16])
    a = ap.parse_args()
    from scema_amd import capi
    from scema_amd.systems import build_pe
    d = build_pe(*a.cells, jitter=0.05, seed=11)
    e = capi.Engine()
    e.register_replica("pe", 1, d)
    l0 = d["box"][3:6] - d["box"][:3]
    t0 = time.time()
    length, info = e.equilibrate("pe", 1, a.nsinit, a.dt, a.temperature)
    wall = time.time() - t0
    box, x, v = e.get_state(capi.QP_NONE, "pe", 1)
    m = d["mass"][d["type"]]
    temp = (m[:, None] * v * v).sum() * 48.88821291 ** 2 / ((3 * len(m) - 3) * 0.0019872067)
    e.set_state(0, "pe", 1, box, x, v)
    pavg = e.debug_run("pe", 1, 200, a.dt, a.temperature, qp=0, nvt=True, use_shake=False, sample=True)
    steps = a.nsinit * 33
    print(json.dumps(dict(what="in.init.lammps schedule on the GPU", atoms=len(m), nsinit=a.nsinit, dt_fs=a.dt, md_steps=steps, wall_s=wall,
                          ms_per_md_step=1e3 * wall / steps, minimiser=info, box_lengths_start=list(l0), box_lengths_end=list(length),
                          temperature_end_K=temp, pressure_after_atm=float(np.mean(pavg[:3])))))


if __name__ == "__main__":
    main()
