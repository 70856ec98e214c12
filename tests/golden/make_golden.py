#!/usr/bin/env python3
"""Generates tests/golden/oracle_eval_small_pe.json: full stress evaluations of the CPU oracle
(oracle/md_oracle.c) on the seeded 360-atom PE fixture of tests/conftest.py, used by the GPU tests as a
committed known answer (the oracle itself is pinned by tests/test_oracle_*.py).  Run from the repo root:
    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from scema_amd.systems import build_pe  # noqa: E402

KW = dict(cut_lj=5.0, cut_coul=4.0, skin=1.0, kspace_accuracy=1e-5)


def main():
    d = build_pe(2, 3, 5, jitter=0.05, seed=7)
    d["box"][6:9] = [0.7, -0.4, 0.5]
    lens = d["box"][3:6] - d["box"][:3]
    cases = []
    for k, (ezz, sh) in enumerate([(1.2e-3, (5e-5, -3e-5, 2e-5)), (-9e-4, (0.0, 1e-5, 0.0)), (6.5e-3, (2e-4, 0.0, -1e-4))]):
        strain = [-0.3 * ezz * lens[0], -0.3 * ezz * lens[1], ezz * lens[2], sh[0] * lens[2], sh[1] * lens[1], sh[2] * lens[0]]
        o = po.Oracle(d, po.default_params(**KW))
        s1, nts1 = o.eval(strain, 2.0, 300.0, 1e-4, 20)
        s2, nts2 = o.eval([0.5 * v for v in strain], 2.0, 300.0, 1e-4, 20)   # second call continues from the stored state
        cases.append(dict(strain_len=[float(v) for v in strain], nss=20, dt=2.0, temperature=300.0, strain_rate=1e-4,
                          nts=[int(nts1), int(nts2)], stress_first=[float(v) for v in s1], stress_second=[float(v) for v in s2]))
    out = dict(generator="tests/golden/make_golden.py", fixture="build_pe(2,3,5,jitter=0.05,seed=7), tilts 0.7,-0.4,0.5",
               params=KW, cases=cases)
    with open(os.path.join(ROOT, "tests", "golden", "oracle_eval_small_pe.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out)[:300])


if __name__ == "__main__":
    main()
