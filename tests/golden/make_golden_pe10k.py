#!/usr/bin/env python3
"""Generates tests/golden/oracle_eval_pe10k.json: full-size known answers of the CPU oracle (oracle/md_oracle.c) for
BASELINE.json configs 2 and 3, at the reference's own settings (lj/cut/coul/long 12/9, skin 2, kspace 1e-4, dt 2 fs,
300 K, strain rate 1e-4 /fs, 100 sampling steps: input_configurations/inputs_dogbone_cuboid.json:50-53).

  config 2  single PE-10k replica (10 368 atoms), three strains (tension + small shears, shear dominated,
            compression with nts = 20), each followed by a second evaluation that continues from the stored state;
  config 3  ten consecutive updates of two quadrature points (q = 0 and q = 37) of the 72-replica batch bench.py and
            tests/test_gpu_fullsize.py run: strains = synthetic_strains(72, lens, seed = 2026 + update)[q].

The oracle takes ~13 s per evaluation on one core; the 26 evaluations run in a process pool.  Run from the repo root:
    python tests/golden/make_golden_pe10k.py
"""
import json
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

NSS, DT, TEMP, RATE = 100, 2.0, 300.0, 1e-4
QPS = (0, 37)
NUPD = 10


def _system():
    from scema_amd.systems import build_pe10k
    return build_pe10k()


def _chain(strains):
    """consecutive evaluations of ONE replica state (history dependence, stmd_problem.h:116-138)"""
    from oracle import pyoracle as po
    o = po.Oracle(_system())
    out = []
    for s in strains:
        st, nts = o.eval(s, DT, TEMP, RATE, NSS)
        out.append(dict(strain_len=[float(v) for v in s], nts=int(nts), stress=[float(v) for v in st]))
    return out


def main():
    from scema_amd.systems import synthetic_strains
    d = _system()
    lens = d["box"][3:6] - d["box"][:3]
    lx, ly, lz = lens
    s0 = synthetic_strains(72, lens, seed=2026)[0]
    shear = np.array([1.0e-4 * lx, -1.0e-4 * ly, 2.0e-4 * lz, 1.5e-3 * lz, -4.0e-4 * ly, 6.0e-4 * lx])
    comp = np.array([0.9e-3 * lx, 0.9e-3 * ly, -3.0e-3 * lz, -5.0e-5 * lz, 8.0e-5 * ly, 2.0e-5 * lx])
    chains = [[s0, 0.5 * s0], [shear, 0.5 * shear], [comp, 0.5 * comp]]
    names = ["tension", "shear", "compression_nts20"]
    for q in QPS:
        chains.append([synthetic_strains(72, lens, seed=2026 + k)[q] for k in range(NUPD)])
    with mp.Pool(min(len(chains), os.cpu_count() or 1)) as pool:
        res = pool.map(_chain, chains)
    out = dict(generator="tests/golden/make_golden_pe10k.py",
               fixture="scema_amd.systems.build_pe10k() (6x9x16 PE cells, 10 368 atoms, seed 1234, 300 K, SHAKE-projected)",
               params=dict(cut_lj=12.0, cut_coul=9.0, skin=2.0, kspace_accuracy=1e-4, nss=NSS, dt=DT, temperature=TEMP, strain_rate=RATE),
               config2=[dict(name=n, evals=r) for n, r in zip(names, res[:3])],
               config3=dict(n_sims=72, seed0=2026, updates=NUPD, qps={str(q): r for q, r in zip(QPS, res[3:])}))
    with open(os.path.join(ROOT, "tests", "golden", "oracle_eval_pe10k.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out)[:400])


if __name__ == "__main__":
    main()
