"""ReaxFF energy of oracle/reax_oracle.c as ONE differentiable FP64 expression (torch, CPU): forces and the virial come from
reverse-mode differentiation of that expression, at the cost of an energy evaluation instead of the 6N + 12 evaluations that
rxo_forces_fd needs -- which is what lets the oracle run whole strained evaluations (oracle/reax_md.py).

TEST INFRASTRUCTURE ONLY, PARITY UNPINNED (LAMMPS USER-REAXC is not in the reference tree; see reax_oracle.h).  Pinned here:
the energy equals rxo_energy part by part (<= 1e-10), forces and virial equal its central differences
(tests/test_oracle_reax_dyn.py).  The product package never imports this module; the product's own derivatives
(scema_amd/csrc/reax/rx_core.h) are hand-derived, so this is an independent check of them.

Every block below follows the block of the same name in reax_oracle.c (which names the USER-REAXC routine it restates).
Discrete decisions (which pairs are bonds, which bonds make angles, the integer part of Delta_e / 2, piecewise branches) are
taken on detached values, as the C code takes them on plain doubles; unselected branches get safe arguments so that no
0 * inf reaches the gradient.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import pyreax

THB_CUT = 0.001
THB_CUTSQ = 0.00001
HB_THRESHOLD = 1e-2
BOND_CUT = 5.0
HBOND_CUT = 7.5
MIN_SINE = 1e-10
C_ELE = 332.06371
KCALPMOL_TO_EV = 23.02
EV_TO_KCALPMOL = 14.4
PARTS = pyreax.PARTS
F64 = torch.float64

SBP = ["r_s", "valency", "mass", "r_vdw", "epsilon", "gamma", "r_pi", "valency_e", "nlp_opt", "alpha", "gamma_w", "valency_boc", "p_ovun5",
       "chi", "eta", "p_hbond", "r_pi_pi", "p_lp2", "b_o_131", "b_o_132", "b_o_133", "p_ovun2", "p_val3", "valency_val", "p_val5", "rcore2",
       "ecore2", "acore2"]
TBP = ["De_s", "De_p", "De_pp", "p_be1", "p_bo5", "v13cor", "p_bo6", "p_ovun1", "p_be2", "p_bo3", "p_bo4", "p_bo1", "p_bo2", "ovc", "r_s", "r_p",
       "r_pp", "p_boc3", "p_boc4", "p_boc5", "D", "alpha", "r_vdW", "gamma_w", "gamma"]
THB = ["theta_00", "p_val1", "p_val2", "p_coa1", "p_val7", "p_pen1", "p_val4"]


class Params:
    """The tables of one ffield file, as oracle/reax_oracle.c read them (rxo_export)."""

    def __init__(self, ff: "pyreax.ForceField"):
        L = pyreax.lib()
        nt = ff.ntypes
        gp = np.zeros(64); sbp = np.zeros((nt, 28)); tbp = np.zeros((nt, nt, 25)); thbp = np.zeros((nt, nt, nt, 29))
        fbp = np.zeros((nt, nt, nt, nt, 7)); hbp = np.zeros((nt, nt, nt, 4)); misc = np.zeros(11)
        L.rxo_export.argtypes = [C.c_void_p] * 8
        assert L.rxo_export(ff.h, *[a.ctypes.data_as(C.c_void_p) for a in (gp, sbp, tbp, thbp, fbp, hbp, misc)]) == nt
        self.nt = nt
        self.gp = gp
        self.sbp = {k: sbp[:, i].copy() for i, k in enumerate(SBP)}
        self.tbp = {k: tbp[:, :, i].copy() for i, k in enumerate(TBP)}
        self.th_cnt = thbp[..., 0].astype(int)
        self.th = {k: thbp[..., 1:].reshape(nt, nt, nt, 4, 7)[..., i].copy() for i, k in enumerate(THB)}
        self.fb_cnt = fbp[..., 0].astype(int)
        self.fb = {k: fbp[..., 2 + i].copy() for i, k in enumerate(["V1", "V2", "V3", "p_tor1", "p_cot1"])}
        self.hb = {k: hbp[..., i].copy() for i, k in enumerate(["r0_hb", "p_hb1", "p_hb2", "p_hb3"])}
        self.bo_cut, self.swa, self.swb = misc[0], misc[1], misc[2]
        self.tap = misc[3:11].copy()


def _t(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=F64)


def _cell(box):
    """h = xprd, yprd, zprd, yz, xz, xy and the 3x3 matrix whose ROWS are the cell vectors a, b, c"""
    lo = np.array(box[:3], float)
    lx, ly, lz = box[3] - box[0], box[4] - box[1], box[5] - box[2]
    xy, xz, yz = box[6], box[7], box[8]
    return lo, np.array([[lx, 0.0, 0.0], [xy, ly, 0.0], [xz, yz, lz]])


def pair_list(x: np.ndarray, box, rcut: float):
    """all pairs i < j whose minimum-image distance is <= rcut: (i, j, image shift as a Cartesian vector).  Like minimg() of
    reax_oracle.c this needs a box at least 2 rcut wide (one image per pair)."""
    n = len(x)
    if box is None:
        hm = None
    else:
        _, hm = _cell(box)
        hinv = np.linalg.inv(hm)
    ii, jj, sh = [], [], []
    blk = max(1, int(4.0e6 // max(n, 1)))
    for a in range(0, n, blk):
        d = x[None, :, :] - x[a:a + blk, None, :]            # [b, n, 3] = x_j - x_i
        if hm is not None:
            lam = d @ hinv
            s = -np.rint(lam) @ hm
            d = d + s
        else:
            s = np.zeros_like(d)
        r2 = (d * d).sum(-1)
        bi, bj = np.nonzero((r2 <= rcut * rcut) & (r2 > 0.0))
        keep = bj > bi + a
        ii.append(bi[keep] + a); jj.append(bj[keep]); sh.append(s[bi[keep], bj[keep]])
    return np.concatenate(ii), np.concatenate(jj), np.concatenate(sh)


class ReaxEnergy:
    def __init__(self, ffield_path: str):
        self.ff = pyreax.ForceField(ffield_path)
        self.p = Params(self.ff)

    # ------------------------------------------------------------------ the energy expression
    def energy(self, types, x, box=None, q=None, strain=None, pairs=None):
        """types: int[n] force-field types; x: torch [n, 3]; box: 9 numbers or None; q: torch [n] or None;
        strain: torch [3, 3] (deformation gradient F = 1 + strain applied to every pair vector) or None.
        Returns (total, dict of parts), all torch scalars."""
        P = self.p
        gp = P.gp
        ty = np.asarray(types, dtype=np.int64)
        n = len(ty)
        xd = x.detach().numpy()
        if pairs is None:
            pairs = pair_list(xd, box, P.swb)
        pi, pj, psh = pairs
        pit, pjt = torch.as_tensor(pi), torch.as_tensor(pj)
        dvec = x[pjt] - x[pit] + _t(psh)
        if strain is not None:
            dvec = dvec + dvec @ strain.T                     # d' = (1 + eps) d
        r2 = (dvec * dvec).sum(1)
        r = torch.sqrt(r2)
        rd = r.detach().numpy()
        zero = torch.zeros((), dtype=F64)
        e = {k: zero for k in PARTS}

        def tb(name, a, b):
            return _t(P.tbp[name][a, b])

        # ---- bond orders (BOp, BO) ----
        bsel = np.nonzero(rd <= BOND_CUT)[0]
        bi, bj = pi[bsel], pj[bsel]
        ti, tj = ty[bi], ty[bj]
        rb = r[torch.as_tensor(bsel)]
        sb = P.sbp

        def bop(par1, par2, r0name, sname):
            ok = (sb[sname][ti] > 0.0) & (sb[sname][tj] > 0.0)
            r0 = np.where(ok, P.tbp[r0name][ti, tj], 1.0)
            v = torch.exp(tb(par1, ti, tj) * torch.pow(rb / _t(r0), tb(par2, ti, tj)))
            return torch.where(torch.as_tensor(ok), v, torch.zeros_like(v))

        BOs_raw = (1.0 + P.bo_cut) * bop("p_bo1", "p_bo2", "r_s", "r_s")
        BOpi_raw = bop("p_bo3", "p_bo4", "r_p", "r_pi")
        BOpp_raw = bop("p_bo5", "p_bo6", "r_pp", "r_pi_pi")
        BOsum = BOs_raw + BOpi_raw + BOpp_raw
        keep = np.nonzero(BOsum.detach().numpy() >= P.bo_cut)[0]
        kt = torch.as_tensor(keep)
        bi, bj, ti, tj = bi[keep], bj[keep], ti[keep], tj[keep]
        bvec = dvec[torch.as_tensor(bsel[keep])]               # i -> j
        rb = rb[kt]
        BOp = BOsum[kt] - P.bo_cut
        BOp_s = BOs_raw[kt] - P.bo_cut
        BOp_pi = BOpi_raw[kt]
        BOp_pi2 = BOpp_raw[kt]
        nb = len(bi)
        bit, bjt = torch.as_tensor(bi), torch.as_tensor(bj)

        def scatter2(v):
            return torch.zeros(n, dtype=F64).index_add(0, bit, v).index_add(0, bjt, v)

        tbo_p = scatter2(BOp)
        Deltap = tbo_p - _t(sb["valency"][ty])
        Deltap_boc = tbo_p - _t(sb["valency_boc"][ty])
        p_boc1, p_boc2 = gp[0], gp[1]
        ovc_on = P.tbp["ovc"][ti, tj] >= 0.001
        v13_on = P.tbp["v13cor"][ti, tj] >= 0.001
        val_i, val_j = _t(sb["valency"][ti]), _t(sb["valency"][tj])
        Dpi, Dpj = Deltap[bit], Deltap[bjt]
        f2 = torch.exp(-p_boc1 * Dpi) + torch.exp(-p_boc1 * Dpj)
        f3 = -1.0 / p_boc2 * torch.log(0.5 * (torch.exp(-p_boc2 * Dpi) + torch.exp(-p_boc2 * Dpj)))
        f1c = 0.5 * ((val_i + f2) / (val_i + f2 + f3) + (val_j + f2) / (val_j + f2 + f3))
        f1 = torch.where(torch.as_tensor(ovc_on), f1c, torch.ones_like(f1c))
        pb3, pb4, pb5 = tb("p_boc3", ti, tj), tb("p_boc4", ti, tj), tb("p_boc5", ti, tj)
        exp_f4 = torch.exp(-(pb4 * BOp * BOp - Deltap_boc[bit]) * pb3 + pb5)
        exp_f5 = torch.exp(-(pb4 * BOp * BOp - Deltap_boc[bjt]) * pb3 + pb5)
        f45c = 1.0 / (1.0 + exp_f4) / (1.0 + exp_f5)
        f45 = torch.where(torch.as_tensor(v13_on), f45c, torch.ones_like(f45c))
        A0 = f1 * f45
        A1 = A0 * f1
        corr = torch.as_tensor(ovc_on | v13_on)
        BO = torch.where(corr, BOp * A0, BOp)
        BO_pi = torch.where(corr, BOp_pi * A1, BOp_pi)
        BO_pi2 = torch.where(corr, BOp_pi2 * A1, BOp_pi2)
        BO_s = torch.where(corr, BO - (BO_pi + BO_pi2), BOp_s)

        def floor0(v):
            return torch.where(v < 1e-10, torch.zeros_like(v), v)

        BO, BO_s, BO_pi, BO_pi2 = floor0(BO), floor0(BO_s), floor0(BO_pi), floor0(BO_pi2)
        total_bo = scatter2(BO)
        Delta = total_bo - _t(sb["valency"][ty])
        Delta_e = total_bo - _t(sb["valency_e"][ty])
        Delta_boc = total_bo - _t(sb["valency_boc"][ty])
        Delta_val = total_bo - _t(sb["valency_val"][ty])
        half = torch.trunc(Delta_e.detach() / 2.0)             # (int)(Delta_e / 2): towards zero
        vlpex = Delta_e - 2.0 * half
        p_lp1 = gp[15]
        explp1 = torch.exp(-p_lp1 * (2.0 + vlpex) ** 2)
        nlp = explp1 - half
        nlp_opt = _t(sb["nlp_opt"][ty])
        Delta_lp = nlp_opt - nlp
        heavy = sb["mass"][ty] > 21.0
        Delta_lp_temp = torch.where(torch.as_tensor(heavy), nlp_opt - 0.5 * _t(sb["valency_e"][ty] - sb["valency"][ty]), nlp_opt - nlp)

        # ---- Bonds ----
        p_be2 = tb("p_be2", ti, tj)
        BOs_safe = torch.where(BO_s > 0.0, BO_s, torch.ones_like(BO_s))
        exp_be12 = torch.exp(tb("p_be1", ti, tj) * (1.0 - torch.pow(BOs_safe, p_be2)))
        exp_be12 = torch.where(BO_s > 0.0, exp_be12, torch.exp(tb("p_be1", ti, tj)))
        e["bond"] = (-tb("De_s", ti, tj) * BO_s * exp_be12 - tb("De_p", ti, tj) * BO_pi - tb("De_pp", ti, tj) * BO_pi2).sum()

        # ---- Atom_Energy ----
        p_ovun3, p_ovun4, p_ovun6, p_ovun7, p_ovun8 = gp[32], gp[31], gp[6], gp[8], gp[9]
        expvd2 = torch.exp(-75.0 * Delta_lp)
        e["lp"] = (_t(sb["p_lp2"][ty]) * Delta_lp / (1.0 + expvd2)).sum()
        dfvl = _t(np.where(heavy, 0.0, 1.0))
        w1 = tb("p_ovun1", ti, tj) * tb("De_s", ti, tj) * BO
        sum_ovun1 = scatter2(w1)
        bpp = BO_pi + BO_pi2
        sum_ovun2 = torch.zeros(n, dtype=F64).index_add(0, bit, (Delta[bjt] - dfvl[bit] * Delta_lp_temp[bjt]) * bpp) \
                                             .index_add(0, bjt, (Delta[bit] - dfvl[bjt] * Delta_lp_temp[bit]) * bpp)
        exp_ovun1 = p_ovun3 * torch.exp(p_ovun4 * sum_ovun2)
        Delta_lpcorr = Delta - (dfvl * Delta_lp_temp) / (1.0 + exp_ovun1)
        exp_ovun2 = torch.exp(_t(sb["p_ovun2"][ty]) * Delta_lpcorr)
        DlpVi = 1.0 / (Delta_lpcorr + _t(sb["valency"][ty]) + 1e-8)
        e["over"] = (sum_ovun1 * Delta_lpcorr * DlpVi / (1.0 + exp_ovun2)).sum()
        exp_ovun6 = torch.exp(p_ovun6 * Delta_lpcorr)
        exp_ovun8 = p_ovun7 * torch.exp(p_ovun8 * sum_ovun2)
        e["under"] = (-_t(sb["p_ovun5"][ty]) * (1.0 - exp_ovun6) / (1.0 + 1.0 / exp_ovun2) / (1.0 + exp_ovun8)).sum()

        # ---- adjacency (detached): per atom the bonds it takes part in, with the direction atom -> other ----
        BOd = BO.detach().numpy()
        adj = [[] for _ in range(n)]                           # (bond index, sign, other atom)
        for k in range(nb):
            adj[bi[k]].append((k, 1.0, bj[k]))
            adj[bj[k]].append((k, -1.0, bi[k]))

        # ---- Valence_Angles ----
        p_val6, p_val8, p_val9, p_val10 = gp[14], gp[33], gp[16], gp[17]
        p_pen2, p_pen3, p_pen4, p_coa2, p_coa3, p_coa4 = gp[19], gp[20], gp[21], gp[2], gp[38], gp[30]
        SBOp = scatter2(bpp)
        bo8 = BO ** 8
        prod_SBO = torch.exp(-scatter2(bo8))
        vlpadj = torch.where(vlpex.detach() >= 0.0, torch.zeros_like(nlp), nlp)
        SBO = SBOp + (1.0 - prod_SBO) * (-Delta_boc - p_val8 * vlpadj)
        sd = SBO.detach()
        lo_arg = torch.where((sd > 0.0) & (sd <= 1.0), SBO, torch.full_like(SBO, 0.5))
        hi_arg = torch.where((sd > 1.0) & (sd < 2.0), 2.0 - SBO, torch.full_like(SBO, 0.5))
        SBO2 = torch.where(sd <= 0.0, torch.zeros_like(SBO),
                           torch.where(sd <= 1.0, torch.pow(lo_arg, p_val9),
                                       torch.where(sd < 2.0, 2.0 - torch.pow(hi_arg, p_val9), torch.full_like(SBO, 2.0))))
        a_c, a_b1, a_s1, a_b2, a_s2, a_i, a_k, a_set = [], [], [], [], [], [], [], []
        for j in range(n):
            lst = adj[j]
            for u in range(len(lst)):
                k1, s1, i = lst[u]
                if not (BOd[k1] - THB_CUT > 0.0):
                    continue
                for v in range(u + 1, len(lst)):
                    k2, s2, k = lst[v]
                    if not (BOd[k2] - THB_CUT > 0.0 and BOd[k1] > THB_CUT and BOd[k2] > THB_CUT and BOd[k1] * BOd[k2] > THB_CUTSQ):
                        continue
                    for c in range(P.th_cnt[ty[i], ty[j], ty[k]]):
                        if abs(P.th["p_val1"][ty[i], ty[j], ty[k], c]) <= 0.001:
                            continue
                        a_c.append(j); a_b1.append(k1); a_s1.append(s1); a_b2.append(k2); a_s2.append(s2); a_i.append(i); a_k.append(k); a_set.append(c)
        if a_c:
            a_c, a_i, a_k, a_set = (np.array(v, dtype=np.int64) for v in (a_c, a_i, a_k, a_set))
            b1, b2 = torch.as_tensor(np.array(a_b1)), torch.as_tensor(np.array(a_b2))
            dji = bvec[b1] * _t(a_s1)[:, None]
            djk = bvec[b2] * _t(a_s2)[:, None]
            cos_t = ((dji * djk).sum(1) / (rb[b1] * rb[b2])).clamp(-1.0, 1.0)
            theta = torch.acos(cos_t)
            jt = torch.as_tensor(a_c)
            tyi, tyj, tyk = ty[a_i], ty[a_c], ty[a_k]

            def th(name):
                return _t(P.th[name][tyi, tyj, tyk, a_set])

            BOA_ij, BOA_jk = BO[b1] - THB_CUT, BO[b2] - THB_CUT
            pv3 = _t(sb["p_val3"][tyj])
            f7_ij = 1.0 - torch.exp(-pv3 * torch.pow(BOA_ij, th("p_val4")))
            f7_jk = 1.0 - torch.exp(-pv3 * torch.pow(BOA_jk, th("p_val4")))
            expval6 = torch.exp(p_val6 * Delta_boc[jt])
            expval7 = torch.exp(-th("p_val7") * Delta_boc[jt])
            trm8 = 1.0 + expval6 + expval7
            pv5 = _t(sb["p_val5"][tyj])
            f8_Dj = pv5 - (pv5 - 1.0) * (2.0 + expval6) / trm8
            theta_00 = th("theta_00") * math.pi / 180.0
            theta_0 = math.pi - theta_00 * (1.0 - torch.exp(-p_val10 * (2.0 - SBO2[jt])))
            expval2theta = torch.exp(-th("p_val2") * (theta_0 - theta) ** 2)
            pv1 = th("p_val1")
            expval12theta = torch.where(pv1 >= 0.0, pv1 * (1.0 - expval2theta), pv1 * -expval2theta)
            e["angle"] = (f7_ij * f7_jk * f8_Dj * expval12theta).sum()
            exp_pen2ij = torch.exp(-p_pen2 * (BOA_ij - 2.0) ** 2)
            exp_pen2jk = torch.exp(-p_pen2 * (BOA_jk - 2.0) ** 2)
            exp_pen3, exp_pen4 = torch.exp(-p_pen3 * Delta[jt]), torch.exp(p_pen4 * Delta[jt])
            f9_Dj = (2.0 + exp_pen3) / (1.0 + exp_pen3 + exp_pen4)
            e["pen"] = (th("p_pen1") * f9_Dj * exp_pen2ij * exp_pen2jk).sum()
            exp_coa2 = torch.exp(p_coa2 * Delta_val[jt])
            it_, kt_ = torch.as_tensor(a_i), torch.as_tensor(a_k)
            e["coa"] = (th("p_coa1") / (1.0 + exp_coa2) * torch.exp(-p_coa3 * (total_bo[it_] - BOA_ij) ** 2) * torch.exp(-p_coa3 * (total_bo[kt_] - BOA_jk) ** 2) *
                        torch.exp(-p_coa4 * (BOA_ij - 1.5) ** 2) * torch.exp(-p_coa4 * (BOA_jk - 1.5) ** 2)).sum()

        # ---- Torsion_Angles ----
        p_tor2, p_tor3, p_tor4, p_cot2 = gp[23], gp[24], gp[25], gp[27]
        t_jk, t_ij, t_sij, t_kl, t_skl, t_i, t_j, t_k, t_l = [], [], [], [], [], [], [], [], []
        for kb in range(nb):
            if not (BOd[kb] > THB_CUT):
                continue
            j, k = bi[kb], bj[kb]
            for (k1, s1, i) in adj[j]:
                if k1 == kb or not (BOd[k1] > THB_CUT):
                    continue
                for (k2, s2, l) in adj[k]:
                    if k2 == kb or l == i:
                        continue
                    if not (P.fb_cnt[ty[i], ty[j], ty[k], ty[l]] and BOd[k2] > THB_CUT and BOd[k1] * BOd[kb] * BOd[k2] > THB_CUT):
                        continue
                    t_jk.append(kb); t_ij.append(k1); t_sij.append(s1); t_kl.append(k2); t_skl.append(s2)
                    t_i.append(i); t_j.append(j); t_k.append(k); t_l.append(l)
        if t_jk:
            bjk, bij, bkl = (torch.as_tensor(np.array(v)) for v in (t_jk, t_ij, t_kl))
            t_i, t_j, t_k, t_l = (np.array(v, dtype=np.int64) for v in (t_i, t_j, t_k, t_l))
            djk = bvec[bjk]
            dkj = -djk
            dji = bvec[bij] * _t(t_sij)[:, None]
            dkl = bvec[bkl] * _t(t_skl)[:, None]

            def sine_of(a, ra, b, rbb):
                c = ((a * b).sum(1) / (ra * rbb)).clamp(-1.0, 1.0)
                s = torch.sin(torch.acos(c))
                sdv = s.detach()
                s = torch.where((sdv >= 0) & (sdv <= MIN_SINE), torch.full_like(s, MIN_SINE), s)
                return torch.where((sdv <= 0) & (sdv >= -MIN_SINE), torch.full_like(s, -MIN_SINE), s)

            sin_ijk = sine_of(dji, rb[bij], djk, rb[bjk])
            sin_jkl = sine_of(dkj, rb[bjk], dkl, rb[bkl])
            n1 = torch.cross(dji, djk, dim=1)
            n2 = torch.cross(dkj, dkl, dim=1)
            nn = torch.sqrt((n1 * n1).sum(1) * (n2 * n2).sum(1))
            nn_safe = torch.where(nn > 0.0, nn, torch.ones_like(nn))
            cos_om = torch.where(nn > 0.0, (n1 * n2).sum(1) / nn_safe, torch.ones_like(nn)).clamp(-1.0, 1.0)
            cos2 = 2.0 * cos_om ** 2 - 1.0
            cos3 = cos_om * (4.0 * cos_om ** 2 - 3.0)
            BOA_ij, BOA_jk, BOA_kl = BO[bij] - THB_CUT, BO[bjk] - THB_CUT, BO[bkl] - THB_CUT
            fn10 = (1.0 - torch.exp(-p_tor2 * BOA_ij)) * (1.0 - torch.exp(-p_tor2 * BOA_jk)) * (1.0 - torch.exp(-p_tor2 * BOA_kl))
            DjDk = Delta_boc[torch.as_tensor(t_j)] + Delta_boc[torch.as_tensor(t_k)]
            exp_tor3, exp_tor4 = torch.exp(-p_tor3 * DjDk), torch.exp(p_tor4 * DjDk)
            f11 = (2.0 + exp_tor3) / (1.0 + exp_tor3 + exp_tor4)

            def fb(name):
                return _t(P.fb[name][ty[t_i], ty[t_j], ty[t_k], ty[t_l]])

            exp_tor1 = torch.exp(fb("p_tor1") * (2.0 - BO_pi[bjk] - f11) ** 2)
            CV = 0.5 * (fb("V1") * (1.0 + cos_om) + fb("V2") * exp_tor1 * (1.0 - cos2) + fb("V3") * (1.0 + cos3))
            e["tors"] = (fn10 * sin_ijk * sin_jkl * CV).sum()
            fn12 = torch.exp(-p_cot2 * (BOA_ij - 1.5) ** 2) * torch.exp(-p_cot2 * (BOA_jk - 1.5) ** 2) * torch.exp(-p_cot2 * (BOA_kl - 1.5) ** 2)
            e["conj"] = (fb("p_cot1") * fn12 * (1.0 + (cos_om ** 2 - 1.0) * sin_ijk * sin_jkl)).sum()

        # ---- Hydrogen_Bonds: donor i (p_hbond 2) - hydrogen j (p_hbond 1) ... acceptor k (p_hbond 2) ----
        hbt = sb["p_hbond"][ty].astype(int)
        h_pair, h_sgn, h_bond, h_sb, h_i, h_j, h_k = [], [], [], [], [], [], []
        cand = np.nonzero((rd <= HBOND_CUT) & (((hbt[pi] == 1) & (hbt[pj] == 2)) | ((hbt[pi] == 2) & (hbt[pj] == 1))))[0]
        for m in cand:
            if hbt[pi[m]] == 1:
                j, k, sg = pi[m], pj[m], 1.0
            else:
                j, k, sg = pj[m], pi[m], -1.0
            for (k1, s1, i) in adj[j]:
                if i == k or hbt[i] != 2 or BOd[k1] < HB_THRESHOLD or P.hb["r0_hb"][ty[i], ty[j], ty[k]] <= 0.0:
                    continue
                h_pair.append(m); h_sgn.append(sg); h_bond.append(k1); h_sb.append(s1); h_i.append(i); h_j.append(j); h_k.append(k)
        if h_pair:
            hp, hbd = torch.as_tensor(np.array(h_pair)), torch.as_tensor(np.array(h_bond))
            h_i, h_j, h_k = (np.array(v, dtype=np.int64) for v in (h_i, h_j, h_k))
            djk = dvec[hp] * _t(h_sgn)[:, None]
            r_jk = r[hp]
            dji = bvec[hbd] * _t(h_sb)[:, None]
            cos_t = ((dji * djk).sum(1) / (rb[hbd] * r_jk)).clamp(-1.0, 1.0)
            theta = torch.acos(cos_t)
            sin4 = torch.sin(0.5 * theta) ** 4

            def hb(name):
                return _t(P.hb[name][ty[h_i], ty[h_j], ty[h_k]])

            e["hb"] = (hb("p_hb1") * (1.0 - torch.exp(-hb("p_hb2") * BO[hbd])) * torch.exp(-hb("p_hb3") * (hb("r0_hb") / r_jk + r_jk / hb("r0_hb") - 2.0)) * sin4).sum()

        # ---- vdW_Coulomb_Energy + polarisation ----
        p_vdW1 = gp[28]
        tpi, tpj = ty[pi], ty[pj]
        Tap = torch.full_like(r, P.tap[7])
        for m in range(6, -1, -1):
            Tap = Tap * r + P.tap[m]
        fn13 = torch.pow(torch.pow(r, p_vdW1) + torch.pow(1.0 / tb("gamma_w", tpi, tpj), p_vdW1), 1.0 / p_vdW1)
        al, rv = tb("alpha", tpi, tpj), tb("r_vdW", tpi, tpj)
        exp1 = torch.exp(al * (1.0 - fn13 / rv))
        exp2 = torch.exp(0.5 * al * (1.0 - fn13 / rv))
        e["vdw"] = (Tap * tb("D", tpi, tpj) * (exp1 - 2.0 * exp2)).sum()
        if q is not None:
            e["coul"] = (Tap * C_ELE * q[pit] * q[pjt] / torch.pow(r2 * r + tb("gamma", tpi, tpj), 1.0 / 3.0)).sum()
            e["pol"] = (KCALPMOL_TO_EV * (_t(sb["chi"][ty]) * q + 0.5 * _t(sb["eta"][ty]) * q * q)).sum()
        tot = sum(e.values())
        return tot, e

    # ------------------------------------------------------------------ forces and virial by reverse-mode differentiation
    def forces(self, types, x, box=None, q=None, virial=False, pairs=None):
        """f = -dE/dx at fixed charges (as LAMMPS evaluates them: no dq/dr terms); W_ab = -dE/d(eps_ab) in the order
        xx, yy, zz, xy, xz, yz; also returns the energy parts."""
        xt = torch.tensor(np.asarray(x, float).reshape(-1, 3), dtype=F64, requires_grad=True)
        eps = torch.zeros(3, 3, dtype=F64, requires_grad=True) if virial else None
        qt = None if q is None else _t(q)
        tot, parts = self.energy(types, xt, box, qt, strain=eps, pairs=pairs)
        grads = torch.autograd.grad(tot, [xt] + ([eps] if virial else []))
        f = -grads[0].numpy()
        parts = {k: float(v.detach()) for k, v in parts.items()}
        tot = tot.detach()
        if not virial:
            return f, float(tot), parts
        g = grads[1].numpy()
        w = -np.array([g[0, 0], g[1, 1], g[2, 2], 0.5 * (g[0, 1] + g[1, 0]), 0.5 * (g[0, 2] + g[2, 0]), 0.5 * (g[1, 2] + g[2, 1])])
        return f, w, float(tot), parts

    # ------------------------------------------------------------------ charge equilibration (fix qeq/reax), one solve
    def h_matrix(self, types, x, box, pairs=None):
        """H of fix_qeq_reax.cpp compute_H as a symmetric scipy CSR matrix (diagonal eta included)"""
        import scipy.sparse as sp
        P = self.p
        ty = np.asarray(types, dtype=np.int64)
        n = len(ty)
        xd = np.asarray(x, float).reshape(-1, 3)
        if pairs is None:
            pairs = pair_list(xd, box, P.swb)
        pi, pj, psh = pairs
        d = xd[pj] - xd[pi] + psh
        r2 = (d * d).sum(1)
        r = np.sqrt(r2)
        Tap = np.full_like(r, P.tap[7])
        for m in range(6, -1, -1):
            Tap = Tap * r + P.tap[m]
        val = Tap * EV_TO_KCALPMOL / np.cbrt(r2 * r + P.tbp["gamma"][ty[pi], ty[pj]])
        dia = P.sbp["eta"][ty]
        H = sp.coo_matrix((np.concatenate([val, val, dia]), (np.concatenate([pi, pj, np.arange(n)]), np.concatenate([pj, pi, np.arange(n)]))), shape=(n, n)).tocsr()
        return H, dia

    @staticmethod
    def cg(H, dia, b, x0, tol, imax):
        """FixQEqReax::CG: Jacobi-preconditioned conjugate gradients, stop at sqrt(r.p) / |b| <= tol"""
        x = x0.copy()
        r = b - H @ x
        d = r / dia
        b_norm = np.sqrt(b @ b)
        sig_new = r @ d
        it = 1
        while it < imax and np.sqrt(sig_new) / b_norm > tol:
            qv = H @ d
            alpha = sig_new / (d @ qv)
            x += alpha * d
            r -= alpha * qv
            p = r / dia
            sig_old = sig_new
            sig_new = r @ p
            d = p + (sig_new / sig_old) * d
            it += 1
        return x, it
