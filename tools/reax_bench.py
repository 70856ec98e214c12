#!/usr/bin/env python3
"""ReaxFF replica set (BASELINE config 5): stress evaluations per second of a batch of synthetic polyethylene replicas run with
md_force_field "reax" (lammps_scripts_reax), one JSON line.  A parity-test configuration, not the headline bench (bench.py);
this tool exists so that the kernels of md_reax.hip can be profiled:  rocprofv3 --kernel-trace --stats -- python3 tools/reax_bench.py"""
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
    ap.add_argument("--sims", type=int, default=72)
    ap.add_argument("--cells", type=int, nargs=3, default=[3, 5, 9])
    ap.add_argument("--updates", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nss", type=int, default=20)
    ap.add_argument("--dt", type=float, default=0.25)
    ap.add_argument("--rate", type=float, default=1e-3)
    ap.add_argument("--equil-steps", type=int, default=200)
    ap.add_argument("--ffield", default=os.path.join(ROOT, "tests", "golden", "ffield.reax.2"))
    a = ap.parse_args()
    from scema_amd import capi
    from scema_amd.systems import build_pe, synthetic_strains
    d = build_pe(*a.cells)
    sym = ["C" if d["mass"][k] > 5 else "H" for k in d["type"]]
    n = len(sym)
    m = np.array([12.011 if s == "C" else 1.008 for s in sym])
    mvv2e = 48.88821291 ** 2
    v = np.random.default_rng(3).standard_normal((n, 3)) * np.sqrt(0.0019872067 * 300.0 / (m[:, None] * mvv2e))
    v -= (m[:, None] * v).sum(0) / m.sum()
    e = capi.Engine()
    e.reax_configure(a.ffield, qeq_tol=1e-6)
    e.register_replica("g0", 1, capi.reax_system(sym, d["x"], d["box"], v=v))
    # thermalise one replica outside the timed region and register the result
    if a.equil_steps > 0:
        e.set_state(0, "g0", 1, d["box"], d["x"], v)
        e.debug_run("g0", 1, a.equil_steps, a.dt, 300.0, qp=0, nvt=True, use_shake=False)
        b, x, v = e.get_state(0, "g0", 1)
        e.drop_state(0, "g0", 1)
        e.register_replica("g0", 1, capi.reax_system(sym, x, b, v=v))
    lens = d["box"][3:6] - d["box"][:3]
    strains = synthetic_strains(a.sims, lens, seed=2026)
    first = True
    def update(u):
        nonlocal first
        sgn = 1.0 if u % 2 == 0 else -1.0
        sims = [capi.make_sim(k, "g0", 1, sgn * strains[k], nss=a.nss, dt=a.dt, temperature=300.0, strain_rate=a.rate,
                              most_recent=capi.QP_NONE if first else k, force_field="reax") for k in range(a.sims)]
        out = e.strain_batch(sims)
        first = False
        return np.array([list(o.stress) for o in out])
    for u in range(a.warmup):
        update(u)
    st0, p0 = e.reax_stats(), e.profile()
    t0 = time.time()
    for u in range(a.updates):
        s = update(a.warmup + u)
    dt_wall = time.time() - t0
    st1, p1 = e.reax_stats(), e.profile()
    steps = p1["md_steps"] - p0["md_steps"]
    line = dict(metric="stress evaluations per second (ReaxFF replicas)", value=a.sims * a.updates / dt_wall, unit="evals/s", n_gpus=1, updates=a.updates,
                warmup=a.warmup, ms_per_update=1e3 * dt_wall / a.updates, dtype="f64", data="synthetic",
                config=dict(workload="%d polyethylene replicas of %d atoms, md_force_field reax (ffield.reax.2, H C N O), dt %.2f fs, nss %d" % (a.sims, n, a.dt, a.nss),
                            sim_steps=int(steps), ms_per_sim_step=1e3 * dt_wall / max(steps, 1),
                            qeq_iterations_per_solve=(st1["qeq_iters"] - st0["qeq_iters"]) / max(st1["qeq_solves"] - st0["qeq_solves"], 1),
                            list_skin=st1["skin"], neigh_builds=int(p1["neigh_builds"] - p0["neigh_builds"]),
                            stress_checksum=float(np.abs(s).sum())))
    print(json.dumps(line))


if __name__ == "__main__":
    main()
