#!/usr/bin/env python3
"""Generates tests/golden/oracle_config3_pe10k.json from a strain history recorded by tests/test_gpu_config3.py
(SCEMA_RECORD_CONFIG3=<path> on the GPU box): for each pinned quadrature point the CPU oracle (oracle/md_oracle.c) evaluates the
same request sequence -- prepare_md_simulations' length scaling (stmd_sync.h:553-556), STMDProblem::strain, the replica average
with the init-stress subtraction (stmd_sync.h:903-905) -- from the registered PE-10k replica, each evaluation continuing from
the state of the one before (stmd_problem.h:116-138).  ~13 s per evaluation, one process per point.
    python tests/golden/make_golden_config3.py gpurun_out/config3_history.json"""
import json
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def _point(job):
    from oracle import pyoracle as po
    from scema_amd.systems import build_pe10k
    q, evs, s0, lens = job
    o = po.Oracle(build_pe10k())
    out = []
    for ev in evs:
        eps = np.array(ev["update_strain"])
        sig, nts = o.eval(po.prepare_strain(eps, np.eye(3), np.array(lens), hooke=False), 2.0, 300.0, 1e-4, 100)
        exp = po.store(sig[None], np.array(s0)[None], np.eye(3)[None], False)
        out.append(dict(step=ev["step"], update_strain=ev["update_strain"], nts=int(nts), oracle_stress_before_init_subtraction=[float(v) for v in sig],
                        oracle_update_stress=[float(v) for v in np.ravel(exp)]))
    return q, out


def main():
    rec = json.load(open(sys.argv[1]))
    jobs = [(q, evs, rec["init_stress"], rec["lens"]) for q, evs in rec["points"].items()]
    with mp.Pool(len(jobs)) as pool:
        pts = dict(pool.map(_point, jobs))
    out = dict(generator="tests/golden/make_golden_config3.py", recorded_by="tests/test_gpu_config3.py (SCEMA_RECORD_CONFIG3)", fe=rec["fe"], nsteps=rec["nsteps"],
               fixture="scema_amd.systems.build_pe10k(); init_stress = the engine's evaluation of a 1e-9 strain (stored here)",
               init_stress=rec["init_stress"], lens=rec["lens"], points=pts)
    path = os.path.join(ROOT, "tests", "golden", "oracle_config3_pe10k.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
