#!/usr/bin/env python3
"""Offline gate for the ReaxFF charge equilibration ("Next round" 3): would a better preconditioner than the Jacobi one of
`fix qeq/reax` (lammps_scripts_reax/in.strain.lammps:12: tolerance 1e-6) save conjugate-gradient iterations on the replica set of
BASELINE config 5?  The tolerance fixes the answer, not the solver: every variant stops on the REFERENCE's measure,
sqrt(r . D^-1 r) / |b| <= 1e-6 (FixQEqReax::CG), whatever it preconditions with.

Runs on the CPU with the oracle (test infrastructure): a thermalised PE-1620 replica, one evaluation of 10 + 20 steps, both systems
(H s = -chi, H t = -1) warm-started from the extrapolated previous solutions as the engine does (history kept across the two runs).
Per step the matrix is taken from the oracle and each variant solves from the same starting vectors:

  jacobi     M^-1 = D^-1                                          (today: the reference's)
  sai R      row i of M^-1 = row i of (H[P_i, P_i])^-1, P_i = {j : r_ij <= R} + {i}, symmetrised: a sparse approximate inverse on the
             near pattern (R = 3.5 A: ~23 entries per row against the matrix's ~545; R = 2.0 A: the bonded neighbours, ~4), rebuilt
             EVERY step (the best case for it) and, as `sai R stale`, built once at the first step of the evaluation and kept
  block      exact inverse of the diagonal blocks of the bonded clusters (a carbon and its hydrogens)

Decision rule (set before the numbers were seen): build it only if a variant saves >= 35 % of the iterations after warm start
(its own product costs ~4 % of a sweep, and it adds a kernel per iteration to a loop that is launch-gap bound).
usage: python tools/qeq_precond_gate.py [--equil 60] [--cells 3 5 9]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pcg(H, apply_minv, dia, b, x0, tol=1e-6, imax=200):
    """preconditioned CG; stops on the reference's measure sqrt(r . D^-1 r) / |b| (FixQEqReax::CG), whatever M is"""
    x = x0.copy()
    r = b - H @ x
    z = apply_minv(r)
    d = z.copy()
    bn = np.sqrt(b @ b)
    rz = r @ z
    it = 1
    while it < imax and np.sqrt(r @ (r / dia)) / bn > tol:
        q = H @ d
        alpha = rz / (d @ q)
        x += alpha * d
        r -= alpha * q
        z = apply_minv(r)
        rz_old, rz = rz, r @ z
        d = z + (rz / rz_old) * d
        it += 1
    return x, it


def sai_local(H, x, box, R):
    """sparse approximate inverse: row i = row i of the inverse of H restricted to the atoms within R of i; symmetrised"""
    import scipy.sparse as sp
    from oracle import reax_torch as rt
    n = H.shape[0]
    pi, pj, _ = rt.pair_list(x, box, R)
    nbr = [[i] for i in range(n)]
    for a, b in zip(pi, pj):
        nbr[a].append(b); nbr[b].append(a)
    Hc = H.tocsr()
    rows, cols, vals = [], [], []
    for i in range(n):
        P = np.array(nbr[i])
        A = Hc[P][:, P].toarray()
        e = np.zeros(len(P)); e[0] = 1.0
        m = np.linalg.solve(A, e)
        rows += [i] * len(P); cols += P.tolist(); vals += m.tolist()
    M = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
    M = 0.5 * (M + M.T)
    return M, float(M.nnz) / n


def block_bonded(H, x, box, sym):
    """exact inverse of the blocks {C + its hydrogens within 1.3 A}"""
    import scipy.sparse as sp
    from oracle import reax_torch as rt
    n = H.shape[0]
    pi, pj, _ = rt.pair_list(x, box, 1.3)
    owner = np.arange(n)
    for a, b in zip(pi, pj):
        if sym[a] == "C" and sym[b] == "H": owner[b] = a
        if sym[b] == "C" and sym[a] == "H": owner[a] = b
    Hc = H.tocsr()
    rows, cols, vals = [], [], []
    for c in np.unique(owner):
        P = np.nonzero(owner == c)[0]
        Ainv = np.linalg.inv(Hc[P][:, P].toarray())
        for a, ia in enumerate(P):
            for b, ib in enumerate(P):
                rows.append(ia); cols.append(ib); vals.append(Ainv[a, b])
    return sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, nargs=3, default=[3, 5, 9])
    ap.add_argument("--equil", type=int, default=60, help="NVT steps of 0.25 fs before the evaluation (the bench equilibrates 200 on the GPU)")
    ap.add_argument("--nss", type=int, default=20)
    ap.add_argument("--radii", type=float, nargs="+", default=[2.0, 3.5], help="pattern radii of the sparse approximate inverse, Angstrom")
    a = ap.parse_args()
    import torch
    torch.set_num_threads(max(1, (os.cpu_count() or 2) - 1))
    from oracle import reax_md
    from scema_amd.systems import build_pe, synthetic_strains
    d = build_pe(*a.cells)
    sym = ["C" if d["mass"][t] > 5 else "H" for t in d["type"]]
    lt = np.array([1 if c == "C" else 0 for c in sym])
    m = np.array([12.011 if c == "C" else 1.008 for c in sym])
    v = np.random.default_rng(3).standard_normal((len(sym), 3)) * np.sqrt(0.0019872067 * 300.0 / (m[:, None] * 48.88821291 ** 2))
    v -= (m[:, None] * v).sum(0) / m.sum()
    M = reax_md.ReaxMD(os.path.join(ROOT, "examples", "ffield.reax.2"), ["H", "C", "N", "O"], lt, [1.008, 12.011, 14.007, 15.999], d["box"], d["x"], v)
    radii = a.radii
    names = ["jacobi"] + [f"sai {R}" for R in radii] + [f"sai {R} stale" for R in radii] + ["block"]
    stats = {k: {"s": [], "t": []} for k in names}
    state = {"on": False, "stale": {}, "nnz": {}, "keep_hist": False}

    def charges(call, x, box, pairs):
        n = M.n
        if call == 0 and not state["keep_hist"]:
            M.s_hist = np.zeros((5, n)); M.t_hist = np.zeros((5, n))
        H, dia = M.R.h_matrix(M.rtype, x, box, pairs)
        sh, th = M.s_hist, M.t_hist
        t0 = th[2] + 3.0 * (th[0] - th[1])
        s0 = 4.0 * (sh[0] + sh[2]) - (6.0 * sh[1] + sh[3])
        chi = M.R.p.sbp["chi"][M.rtype]
        s, it1 = pcg(H, lambda r: r / dia, dia, -chi, s0)
        t, it2 = pcg(H, lambda r: r / dia, dia, -np.ones(n), t0)
        if state["on"]:
            stats["jacobi"]["s"].append(it1); stats["jacobi"]["t"].append(it2)
            xs = np.asarray(x, float).reshape(-1, 3)
            variants = []
            for R in radii:
                Mi, nz = sai_local(H, xs, box, R)
                state["nnz"][f"sai {R}"] = nz
                state["stale"].setdefault(R, Mi)
                variants += [(f"sai {R}", Mi), (f"sai {R} stale", state["stale"][R])]
            variants.append(("block", block_bonded(H, xs, box, sym)))
            for name, Mi in variants:
                stats[name]["s"].append(pcg(H, lambda r: Mi @ r, dia, -chi, s0)[1])
                stats[name]["t"].append(pcg(H, lambda r: Mi @ r, dia, -np.ones(n), t0)[1])
        M.qeq_iters += it1 + it2; M.qeq_solves += 1
        u = s.sum() / t.sum()
        M.q = s - u * t
        M.s_hist = np.vstack([s[None], sh[:4]]); M.t_hist = np.vstack([t[None], th[:4]])

    M._charges = charges
    t0 = time.time()
    if a.equil > 0:
        M.run(a.equil, 0.25, 300.0, nvt=True)
    print(f"equilibrated {a.equil} steps in {time.time() - t0:.0f} s; Jacobi iterations per solve so far {M.qeq_iters / max(M.qeq_solves, 1) / 2:.1f}", flush=True)
    lens = d["box"][3:6] - d["box"][:3]
    strain = synthetic_strains(1, lens, seed=2026)[0]
    state["on"] = True; state["keep_hist"] = True   # the engine keeps the solver history from the straining run to the sampling run
    M.eval(strain, 0.25, 300.0, 1e-3, a.nss)
    print(f"evaluation ({len(stats['jacobi']['s'])} force evaluations) done after {time.time() - t0:.0f} s")
    base = np.mean(stats["jacobi"]["s"]) + np.mean(stats["jacobi"]["t"])
    print(f"{'preconditioner':16s} {'entries/row':>11s} {'iters s':>8s} {'iters t':>8s} {'both':>6s} {'saved':>7s}")
    for k in names:
        s_, t_ = np.mean(stats[k]["s"]), np.mean(stats[k]["t"])
        nz = state["nnz"].get(k.replace(" stale", ""), 1.0 if k == "jacobi" else 3.0)
        print(f"{k:16s} {nz:11.1f} {s_:8.2f} {t_:8.2f} {s_ + t_:6.2f} {100 * (1 - (s_ + t_) / base):6.1f}%")


if __name__ == "__main__":
    main()
