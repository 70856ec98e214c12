#!/usr/bin/env python3
"""Offline gate for k_pair's row order ("Next round" 1): no GPU.

Emulates what k_neigh_build stores for ONE PE-10k replica (cells, k-d ordered 4-atom clusters, the tile ownership rule, the
union rows with their 4-bit masks and the A / B / C1 / C2 segments) on a thermalised configuration produced by the CPU oracle,
then prices k_pair's row loop with the static instruction counts of DESIGN 5.3 (chunk preamble 30, distance block 11, LJ block 20,
coulomb block 19: a block runs for a 64-entry chunk when ANY lane needs it) under different orders of the entries inside a row:

  (i)   today's order (table order inside the segments),
  (ii)  candidate orders keyed by the four-atom in-range pattern, taken at build time,
  (iii) the same orders priced 5 / 10 / 15 steps after the build (the atoms have moved, the order has not).

Prints the block counts per replica-step next to the counting build's (37 700 chunks, 3.84 distance blocks per chunk,
74.5 % reach the LJ block, 36.9 % the coulomb block) so that the emulation can be judged, and the instruction totals per order.

Usage: python tools/pair_order_gate.py [--therm 300] [--cache /tmp/gate_traj.npz]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CUT_COUL, CUT_LJ, SKIN = 9.0, 12.0, 2.0
RLIST = CUT_LJ + SKIN
SEG_A, SEG_B, SEG_C = CUT_COUL + 0.2, CUT_LJ + 0.2, CUT_LJ + 0.65 * SKIN
W_CHUNK, W_DIST, W_LJ, W_COUL = 30, 11, 20, 19


def trajectory(therm, lifetimes, cache):
    if cache and os.path.exists(cache):
        z = np.load(cache)
        return z["box"], [z[f"x{k}"] for k in lifetimes], z["excl_i"], z["excl_j"]
    from oracle import pyoracle as po
    from scema_amd.systems import build_pe10k
    d = build_pe10k()
    o = po.Oracle(d, po.default_params())
    t0 = time.time()
    o.run(therm, 2.0, 300.0, nvt=True, use_shake=True)
    snaps, done = [], 0
    for k in lifetimes:
        if k > done:
            o.run(k - done, 2.0, 300.0, nvt=True, use_shake=True)
            done = k
        box, x, _ = o.get_state()
        snaps.append(np.array(x).reshape(-1, 3).copy())
    print(f"# oracle: {therm}+{done} steps in {time.time() - t0:.0f} s", file=sys.stderr)
    # 1-2 and 1-3 exclusions from the bond list (special_bonds 0 0 1)
    n = int(d["natoms"])
    b = np.asarray(d["bond_atoms"] if "bond_atoms" in d else d["bonds"]).reshape(-1, 2)
    adj = [[] for _ in range(n)]
    for i, j in b:
        adj[i].append(j); adj[j].append(i)
    ei, ej = [], []
    for i in range(n):
        s = set(adj[i])
        for j in adj[i]:
            s.update(adj[j])
        s.discard(i)
        for j in s:
            ei.append(i); ej.append(j)
    ei, ej = np.array(ei), np.array(ej)
    if cache:
        np.savez(cache, box=np.array(box), excl_i=ei, excl_j=ej, **{f"x{k}": s for k, s in zip(lifetimes, snaps)})
    return np.array(box), snaps, ei, ej


def kd_order(idx, x):
    """recursive median splits along the longest extent, left part a multiple of 4 atoms (k_cell_sort)"""
    if len(idx) <= 4:
        return list(idx)
    p = x[idx]
    ext = p.max(0) - p.min(0)
    dim = int(ext.argmax())
    o = idx[np.argsort(p[:, dim], kind="stable")]
    nl = ((len(o) // 2 + 3) // 4) * 4
    if nl >= len(o):
        nl -= 4
    return kd_order(o[:nl], x) + kd_order(o[nl:], x)


def greedy_order(idx, x):
    """alternative cluster formation (--clusters greedy): the remaining atom farthest from the others' centroid and its three nearest"""
    out, rem = [], set(int(i) for i in idx)
    while len(rem) > 4:
        arr = np.array(sorted(rem)); P = x[arr]
        s = arr[((P - P.mean(0)) ** 2).sum(1).argmax()]
        grp = arr[np.argsort(((P - x[s]) ** 2).sum(1), kind="stable")[:4]]
        out += [int(g) for g in grp]
        rem -= set(int(g) for g in grp)
    return out + sorted(rem)


ORDER_IN_CELL = {"kd": kd_order, "greedy": greedy_order}
CLUSTERS = {"how": "kd"}


def build_rows(box, x0, ei, ej):
    """-> per cluster: atoms[4] (-1 pad), and row = (jatom, shift[3], mask) arrays in table order, plus segment ids"""
    lo, hi = box[:3], box[3:6]
    L = hi - lo
    assert np.abs(box[6:]).max() < 1e-9, "gate assumes the orthogonal start box"
    n = len(x0)
    fl = np.floor((x0 - lo) / L)
    xw = x0 - fl * L
    # cell grid: largest cells with edges in [rlist/2, rlist] that the engine picks for PE-10k: 4 x 6 x 5
    nc = np.array([4, 6, 5])
    edge = L / nc
    mst = np.ceil(RLIST / edge).astype(int)
    c = np.minimum(((xw - lo) / edge).astype(int), nc - 1)
    cell = (c[:, 2] * nc[1] + c[:, 1]) * nc[0] + c[:, 0]
    ncell = int(nc.prod())
    slot_atoms, cell_start = [], [0]
    for cc in range(ncell):
        idx = np.nonzero(cell == cc)[0]
        o = ORDER_IN_CELL[CLUSTERS["how"]](idx, xw)
        o += [-1] * ((-len(o)) % 4)
        slot_atoms += o
        cell_start.append(len(slot_atoms))
    slot_atoms = np.array(slot_atoms)
    cell_start = np.array(cell_start)
    excl = set(zip(ei.tolist(), ej.tolist()))
    clusters = []
    for cc in range(ncell):
        c0, c1, c2 = cc % nc[0], (cc // nc[0]) % nc[1], cc // (nc[0] * nc[1])
        cs, ce = cell_start[cc], cell_start[cc + 1]
        # candidate table: own cell first (slots, pads included), then the half stencil in the kernel's loop order
        tj, tshift, town = [np.arange(cs, ce)], [np.zeros((ce - cs, 3))], [np.ones(ce - cs, bool)]
        for o2 in range(0, mst[2] + 1):
            a2, s2 = (c2 + o2) % nc[2], (c2 + o2) // nc[2]
            if s2 > 1:
                continue
            for o1 in range(0 if o2 == 0 else -mst[1], mst[1] + 1):
                a1, s1 = (c1 + o1) % nc[1], (c1 + o1) // nc[1]
                if s1 < -1 or s1 > 1:
                    continue
                for o0 in range(1 if (o2 == 0 and o1 == 0) else -mst[0], mst[0] + 1):
                    a0, s0 = (c0 + o0) % nc[0], (c0 + o0) // nc[0]
                    if s0 < -1 or s0 > 1:
                        continue
                    cj = (a2 * nc[1] + a1) * nc[0] + a0
                    js = np.arange(cell_start[cj], cell_start[cj + 1])
                    tj.append(js)
                    tshift.append(np.tile(np.array([s0, s1, s2]) * L, (len(js), 1)))
                    town.append(np.zeros(len(js), bool))
        tj = np.concatenate(tj); tshift = np.concatenate(tshift); town = np.concatenate(town)
        ja = slot_atoms[tj]
        real = ja >= 0
        pj = np.where(real[:, None], xw[np.maximum(ja, 0)] + tshift, 1e9)
        for cl in range(cs // 4, ce // 4):
            at = slot_atoms[4 * cl:4 * cl + 4]
            if at[0] < 0:
                continue
            pi = np.where((at >= 0)[:, None], xw[np.maximum(at, 0)], -1e9)
            d2 = ((pi[:, None, :] - pj[None, :, :]) ** 2).sum(-1)       # [4, ntab]
            m = d2 < RLIST * RLIST
            # own cell: each pair once by slot order
            for a in range(4):
                m[a] &= ~(town & (tj <= 4 * cl + a))
                if at[a] >= 0:
                    near = np.nonzero(m[a] & (d2[a] < 9.0))[0]          # exclusion gate (bonded partners are < 3 A away)
                    for l in near:
                        if (int(at[a]), int(ja[l])) in excl:
                            m[a, l] = False
            keep = m.any(0)
            l = np.nonzero(keep)[0]
            clusters.append(dict(atoms=at, j=ja[l], shift=tshift[l], mask=m[:, l], rmin0=np.sqrt(d2[:, l].min(0)), d0=np.sqrt(d2[:, l])))
    return clusters, L


def features(cl, xt, L, lo):
    """[12, n] booleans at positions xt: listed_a, lj_a, coul_a"""
    at = cl["atoms"]
    # unwrapped dynamics: the list keeps its image shifts; positions relative to the build-time wrap
    pi = np.where((at >= 0)[:, None], xt[np.maximum(at, 0)], -1e9)
    pj = xt[cl["j"]] + cl["shift"]
    d2 = ((pi[:, None, :] - pj[None, :, :]) ** 2).sum(-1)
    listed = cl["mask"]
    lj = listed & (d2 < CUT_LJ ** 2)
    co = listed & (d2 < CUT_COUL ** 2)
    return listed, lj, co


def price(order_fn, clusters, xt_w, L, lo, need_far):
    tot = dict(chunks=0, dist=0, lj=0, coul=0, lanes_dist=0, lanes_lj=0, lanes_coul=0, entries=0)
    for cl in clusters:
        listed, lj, co = features(cl, xt_w, L, lo)
        perm = order_fn(cl, need_far)
        listed, lj, co = listed[:, perm], lj[:, perm], co[:, perm]
        n = len(perm)
        tot["entries"] += n
        nch = (n + 63) // 64
        pad = nch * 64 - n
        def blocks(f):
            f = np.pad(f, ((0, 0), (0, pad))).reshape(4, nch, 64)
            return int(f.any(2).sum()), int(f.sum())
        b, ln = blocks(listed); tot["dist"] += b; tot["lanes_dist"] += ln
        b, ln = blocks(lj); tot["lj"] += b; tot["lanes_lj"] += ln
        b, ln = blocks(co); tot["coul"] += b; tot["lanes_coul"] += ln
        tot["chunks"] += nch
    tot["insts"] = W_CHUNK * tot["chunks"] + W_DIST * tot["dist"] + W_LJ * tot["lj"] + W_COUL * tot["coul"]
    return tot


def seg_of(cl):
    r = cl["rmin0"]
    return np.where(r < SEG_A, 0, np.where(r < SEG_B, 1, np.where(r < SEG_C, 2, 3)))


def order_today(cl, need_far):
    s = seg_of(cl)
    idx = np.arange(len(s))
    front = np.concatenate([idx[s == 0], idx[s == 1], idx[s == 2]])
    return np.concatenate([front, idx[s == 3][::-1]]) if need_far else front


def make_pattern_order(margin, variant):
    """orders inside [A|B|C1] (C2 stays the far band that need_far gates) by build-time pattern"""
    def fn(cl, need_far):
        s = seg_of(cl)
        d0, m = cl["d0"], cl["mask"]
        co = m & (d0 < CUT_COUL + margin)
        lj = m & (d0 < CUT_LJ + margin)
        w = np.array([1, 2, 4, 8])[:, None]
        pc, pl, pm = (co * w).sum(0), (lj * w).sum(0), (m * w).sum(0)
        nco, nlj = co.sum(0), lj.sum(0)
        gray = np.array([0, 1, 3, 2, 7, 6, 4, 5, 15, 14, 12, 13, 8, 9, 11, 10])
        inv_gray = np.argsort(gray)
        idx = np.arange(len(s))
        front = idx[s < 3]
        if variant == "count":          # by how many atoms are inside each cutoff, then by which
            key = np.lexsort((inv_gray[pl[front]], inv_gray[pc[front]], -nlj[front], -nco[front]))
        elif variant == "coulpat":      # coulomb pattern (Gray order), then LJ pattern
            key = np.lexsort((inv_gray[pm[front]], inv_gray[pl[front]], inv_gray[pc[front]] + 16 * (nco[front] == 0)))
        elif variant == "ljfirst":      # LJ pattern first (the dearer block), then coulomb pattern
            key = np.lexsort((inv_gray[pm[front]], inv_gray[pc[front]], inv_gray[pl[front]]))
        elif variant == "lex12":        # the 12 bits as one number: coul bits high, then LJ, then listed
            key = np.argsort(-(pc[front].astype(np.int64) * 256 + pl[front] * 16 + pm[front]), kind="stable")
        elif variant == "rmin":         # plain sort by the nearest distance (what finer segments converge to)
            key = np.argsort(cl["rmin0"][front], kind="stable")
        elif variant == "rmean":
            dm = np.where(m, d0, np.nan)
            key = np.argsort(np.nanmean(dm[:, front], 0), kind="stable")
        else:
            raise ValueError(variant)
        front = front[key]
        return np.concatenate([front, idx[s == 3][::-1]]) if need_far else front
    return fn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--therm", type=int, default=300)
    ap.add_argument("--cache", default="/tmp/gate_traj.npz")
    ap.add_argument("--clusters", default="kd", choices=["kd", "greedy"], help="how a cell's atoms are grouped into 4-atom clusters (kd: as k_cell_sort does)")
    args = ap.parse_args()
    CLUSTERS["how"] = args.clusters
    life = [0, 5, 10, 15]
    box, snaps, ei, ej = trajectory(args.therm, life, args.cache)
    t0 = time.time()
    clusters, L = build_rows(box, snaps[0], ei, ej)
    lo = box[:3]
    nent = sum(len(c["j"]) for c in clusters)
    npair = sum(int(c["mask"].sum()) for c in clusters)
    print(f"# rows: {len(clusters)} clusters, {nent / len(clusters):.0f} entries per row, {npair / 10368:.0f} listed pairs per atom, mask density {npair / nent / 4:.2f}  ({time.time() - t0:.0f} s)")
    # the unwrapped coordinates of later snapshots stay consistent with the build-time wrap: apply the same per-atom wrap
    fl = np.floor((snaps[0] - lo) / L)
    orders = [("today", order_today)]
    for v in ("rmin", "rmean", "count", "coulpat", "ljfirst", "lex12"):
        for mg in ((0.0, 0.3) if v not in ("rmin", "rmean") else (0.0,)):
            orders.append((f"{v}+{mg}", make_pattern_order(mg, v)))
    # need_far: DESIGN 5.3 says the far band is walked on 61 % of the steps; price both and mix
    print("order           age  chunks   dist/ch  lj%    coul%   lanes d/lj/c     insts(M)  vs today")
    base = {}
    for name, fn in orders:
        for k, xs in zip(life, snaps):
            xw = xs - fl * L
            mix = {}
            for nf in (False, True):
                mix[nf] = price(fn, clusters, xw, L, lo, nf)
            t = {kk: 0.39 * mix[False][kk] + 0.61 * mix[True][kk] for kk in mix[True]}
            if name == "today":
                base[k] = t["insts"]
            print(f"{name:14s} {k:4d} {t['chunks']:8.0f} {t['dist'] / t['chunks']:8.2f} {100 * t['lj'] / t['dist']:6.1f} {100 * t['coul'] / t['dist']:6.1f}   "
                  f"{t['lanes_dist'] / t['dist']:.1f}/{t['lanes_lj'] / t['lj']:.1f}/{t['lanes_coul'] / t['coul']:.1f}   {t['insts'] / 1e6:8.3f}  {100 * (t['insts'] / base[k] - 1):+6.1f} %", flush=True)


if __name__ == "__main__":
    main()
