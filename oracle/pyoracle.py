"""ctypes binding of the CPU oracle (oracle/md_oracle.c, oracle/host_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (scema_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SAN = os.environ.get("SCEMA_SANITIZE") == "1"   # tools/run_asan.sh: the AddressSanitizer + UBSan build
_LIB = os.path.join(_HERE, "_build_asan" if _SAN else "_build", "libmd_oracle.so")

NPART = 8
PARTS = ["lj", "coul", "bond", "angle", "dihedral", "improper", "kspace", "shake"]


class OmdParams(C.Structure):
    _fields_ = [("cut_lj", C.c_double), ("cut_coul", C.c_double), ("skin", C.c_double),
                ("neigh_delay", C.c_int), ("kspace_accuracy", C.c_double), ("shake_tol", C.c_double),
                ("shake_maxiter", C.c_int), ("shake_mass", C.c_double), ("t_period", C.c_double),
                ("t_chain", C.c_int), ("kspace_pppm", C.c_int), ("pppm_mesh", C.c_int * 3)]


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("md_oracle.c", "md_oracle.h", "host_oracle.c", "host_oracle.h")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["asan"] if _SAN else []))
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        L = _lib
        L.omd_create.restype = C.c_void_p
        L.omd_tdof.restype = C.c_double
        L.omd_g_ewald.restype = C.c_double
        L.omd_temperature.restype = C.c_double
        L.omd_round_rate.restype = C.c_double
        L.omd_round_rate.argtypes = [C.c_double]
        L.omd_round_f.restype = C.c_double
        L.omd_round_f.argtypes = [C.c_double]
        L.omd_nts.argtypes = [C.c_void_p, C.c_double, C.c_double]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def default_params(**kw) -> OmdParams:
    p = OmdParams()
    lib().omd_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class Oracle:
    """One MD system + state."""

    def __init__(self, sysd: dict, params: OmdParams | None = None):
        L = lib()
        self.p = params if params is not None else default_params()
        d = sysd
        self.n = int(d["natoms"])
        arr = lambda k, t: np.ascontiguousarray(d[k], dtype=t)
        self._keep = [arr("type", np.int32), arr("charge", np.float64), arr("mass", np.float64),
                      arr("eps", np.float64), arr("sigma", np.float64),
                      arr("bonds", np.int32), arr("bond_type", np.int32), arr("bond_coeff", np.float64),
                      arr("angles", np.int32), arr("angle_type", np.int32), arr("angle_coeff", np.float64),
                      arr("dihedrals", np.int32), arr("dihedral_type", np.int32), arr("dihedral_coeff", np.float64),
                      arr("impropers", np.int32), arr("improper_type", np.int32), arr("improper_coeff", np.float64),
                      arr("special_lj", np.float64), arr("special_coul", np.float64)]
        k = self._keep
        self.h = C.c_void_p(L.omd_create(
            C.c_int(self.n), C.c_int(int(d["ntypes"])), _p(k[0]), _p(k[1]), _p(k[2]), _p(k[3]), _p(k[4]),
            C.c_int(len(k[6])), _p(k[5]), _p(k[6]), C.c_int(len(k[7])), _p(k[7]),
            C.c_int(len(k[9])), _p(k[8]), _p(k[9]), C.c_int(len(k[10])), _p(k[10]),
            C.c_int(len(k[12])), _p(k[11]), _p(k[12]), C.c_int(len(k[13])), _p(k[13]),
            C.c_int(len(k[15])), _p(k[14]), _p(k[15]), C.c_int(len(k[16])), _p(k[16]),
            _p(k[17]), _p(k[18]), C.byref(self.p)))
        self.set_state(d["box"], d["x"], d["v"])

    def __del__(self):
        try:
            lib().omd_destroy(self.h)
        except Exception:
            pass

    def set_state(self, box, x, v):
        box = np.ascontiguousarray(box, dtype=np.float64)
        x = np.ascontiguousarray(x, dtype=np.float64)
        v = np.ascontiguousarray(v, dtype=np.float64)
        lib().omd_set_state(self.h, _p(box), _p(x), _p(v))

    def get_state(self):
        box = np.zeros(9)
        x = np.zeros((self.n, 3))
        v = np.zeros((self.n, 3))
        lib().omd_get_state(self.h, _p(box), _p(x), _p(v))
        return box, x, v

    def setup(self, use_shake: bool = True):
        lib().omd_setup(self.h, C.c_int(1 if use_shake else 0))

    def freeze_kspace(self, frozen=True):
        lib().omd_freeze_kspace(self.h, C.c_int(1 if frozen else 0))

    def compute(self):
        f = np.zeros((self.n, 3))
        e = np.zeros(NPART)
        w = np.zeros((NPART, 6))
        lib().omd_compute(self.h, _p(f), _p(e), _p(w))
        return f, e, w

    def temperature(self):
        ke = np.zeros(6)
        t = lib().omd_temperature(self.h, _p(ke))
        return t, ke

    def run(self, nsteps, dt, temperature, nvt=True, use_shake=True, rates=None, sample=False, trace=False):
        pavg = np.zeros(6) if sample else None
        tr = np.zeros((nsteps, 8)) if trace else None
        r = None if rates is None else np.ascontiguousarray(rates, dtype=np.float64)
        rc = lib().omd_run(self.h, C.c_int(nsteps), C.c_double(dt), C.c_double(temperature),
                           C.c_int(1 if nvt else 0), C.c_int(1 if use_shake else 0),
                           _p(r) if r is not None else None, _p(pavg) if sample else None,
                           _p(tr) if trace else None)
        if rc != 0:
            raise RuntimeError(f"omd_run failed rc={rc}")
        return pavg, tr

    def eval(self, strain_len, dt, temperature, strain_rate, nss):
        s = np.ascontiguousarray(strain_len, dtype=np.float64)
        out = np.zeros(6)
        nts = lib().omd_eval(self.h, _p(s), C.c_double(dt), C.c_double(temperature), C.c_double(strain_rate),
                             C.c_int(nss), _p(out))
        if nts < 0:
            raise RuntimeError(f"omd_eval failed rc={nts}")
        return out, nts

    # ---- init_material's equilibration schedule (in.init.lammps), md_oracle.c ----
    def minimize(self, etol=1e-7, ftol=1e-11, maxiter=1000, maxeval=50000):
        info = np.zeros(4)
        stop = lib().omd_minimize(self.h, C.c_double(etol), C.c_double(ftol), C.c_int(maxiter), C.c_int(maxeval), _p(info))
        return dict(stop=int(stop), iterations=int(info[0]), evaluations=int(info[1]), e_initial=info[2], e_final=info[3])

    def velocity_create(self, temperature, seed=1234):
        lib().omd_velocity_create(self.h, C.c_double(temperature), C.c_uint64(seed))

    def change_box(self, lengths):
        l = np.ascontiguousarray(lengths, dtype=np.float64)
        lib().omd_change_box(self.h, _p(l))

    def run_nh(self, nsteps, dt, t_start, t_stop=None, npt=False, p_target=1.0, p_period=1000.0, average_lengths=False, trace=False):
        lav = np.zeros(3) if average_lengths else None
        tr = np.zeros((nsteps, 6)) if trace else None
        rc = lib().omd_run_nh(self.h, C.c_int(nsteps), C.c_double(dt), C.c_double(t_start), C.c_double(t_start if t_stop is None else t_stop),
                              C.c_int(1 if npt else 0), C.c_double(p_target), C.c_double(p_period),
                              _p(lav) if average_lengths else None, _p(tr) if trace else None)
        if rc != 0:
            raise RuntimeError(f"omd_run_nh failed rc={rc}")
        return lav, tr

    def equilibrate(self, nsinit, dt, tempt, seed=1234):
        lengths, info = np.zeros(3), np.zeros(4)
        lib().omd_equilibrate(self.h, C.c_int(nsinit), C.c_double(dt), C.c_double(tempt), C.c_uint64(seed), _p(lengths), _p(info))
        return lengths, info

    def timing(self):
        t = np.zeros(4)
        lib().omd_last_timing(self.h, _p(t))
        return dict(pair=t[0], kspace=t[1], neigh=t[2], other=t[3])

    @property
    def tdof(self):
        return lib().omd_tdof(self.h)

    @property
    def g_ewald(self):
        return lib().omd_g_ewald(self.h)

    @property
    def pppm_grid(self):
        n = (C.c_int * 3)()
        lib().omd_pppm_grid(self.h, n)
        return tuple(n)

    @property
    def nkvec(self):
        return lib().omd_nkvec(self.h)

    @property
    def npairs(self):
        return lib().omd_npairs(self.h)

    @property
    def nflips(self):
        return lib().omd_nflips(self.h)

    @property
    def nconstraints(self):
        return lib().omd_nconstraints(self.h)

    @property
    def nclusters(self):
        return lib().omd_nclusters(self.h)


# ---- host (L3) arithmetic ----
def init_material(o: "Oracle", dt, temperature, nss, strain_ampl, strain_rate):
    """CPU restatement of what the reference derives from an equilibrated replica (test infrastructure):
    EQMDProblem::lammps_equilibration, init_material_problem.h:196-300, with ELASTIC/in.homogenization.lammps
    (NVT + SHAKE sampling -> initial stress) and ELASTIC/in.modulus.lammps + bi-displace.mod.lammps (+-up in six
    directions from the post-sampling state, fix nvt only, fix deform delta over nsstrain steps, then sampling).
    Returns (length[3], stress[6] Pa in file order 00,01,02,11,12,22, stiff[6,6] Pa in file order)."""
    box0, x0, v0 = o.get_state()
    length = np.array(box0[3:6]) - np.array(box0[:3])
    pp, _ = o.run(nss, dt, temperature, nvt=True, use_shake=True, sample=True)           # pxx,pyy,pzz,pxy,pxz,pyz (atm)
    stress = -np.array([pp[0], pp[3], pp[4], pp[1], pp[5], pp[2]]) * 1.01325e5               # init_material_problem.h:243-250
    box, x, v = o.get_state()                                                                # "restart.equil"
    nsstrain = int(np.ceil(strain_ampl / (dt * strain_rate) / 10.0) * 10)                    # :226
    T = nsstrain * dt
    ly0, lz0 = box[4] - box[1], box[5] - box[2]
    xy, xz, yz = box[6], box[7], box[8]
    p1 = np.zeros((6, 2, 6))
    for d in range(6):
        for k, sign in enumerate((-1.0, 1.0)):
            r = np.zeros(6)                                                                  # raw order xx,yy,zz,xy,xz,yz
            if d == 0:
                r[0] = sign * strain_ampl / T; r[3] = -sign * strain_ampl * xy / (ly0 * T); r[4] = -sign * strain_ampl * xz / (lz0 * T)
            elif d == 1:
                r[1] = sign * strain_ampl / T; r[5] = -sign * strain_ampl * yz / (lz0 * T)
            elif d == 2:
                r[2] = sign * strain_ampl / T
            elif d == 3:
                r[5] = sign * strain_ampl / T
            elif d == 4:
                r[4] = sign * strain_ampl / T
            else:
                r[3] = sign * strain_ampl / T
            o.set_state(box, x, v)
            o.run(nsstrain, dt, temperature, nvt=True, use_shake=False, rates=r)
            p1[d, k], _ = o.run(nss, dt, temperature, nvt=True, use_shake=False, sample=True)
    raw_of_voigt = [0, 1, 2, 5, 4, 3]                                                        # d1..d6 = xx,yy,zz,yz,xz,xy
    Cm = np.zeros((6, 6))
    for d in range(6):
        for i in range(6):
            Cm[i, d] = -(p1[d, 1, raw_of_voigt[i]] - p1[d, 0, raw_of_voigt[i]]) / (2.0 * strain_ampl) * 1.01325e-4
    call = 0.5 * (Cm + Cm.T) * 1.0e9                                                         # C{ij}all, GPa -> Pa
    hv = [0, 3, 4, 1, 5, 2]                                    # file index -> the reference's 6x6 index (:276-295)
    stiff = np.array([[call[hv[i], hv[j]] for j in range(6)] for i in range(6)])
    o.set_state(box0, x0, v0)
    return length, stress, stiff


def tilt_closest(tilt, xprd_new, yprd_new, xy, xz, yz, xprd, yprd):
    """fix deform: raw tilt targets (xy, xz, yz) moved by whole box lengths to the value closest to the current tilt ratio"""
    t = np.ascontiguousarray(tilt, dtype=np.float64).copy()
    lib().omd_tilt_closest(_p(t), C.c_double(xprd_new), C.c_double(yprd_new), C.c_double(xy), C.c_double(xz), C.c_double(yz),
                           C.c_double(xprd), C.c_double(yprd))
    return t


def tilt_flip(tilt, xprd, yprd):
    """fix deform flip rule: (flipped?, tilts after the flip, lattice steps f_xy, f_xz, f_yz)"""
    t = np.ascontiguousarray(tilt, dtype=np.float64)
    out = np.zeros(3)
    nf = np.zeros(3, dtype=np.int32)
    rc = lib().omd_tilt_flip(_p(t), C.c_double(xprd), C.c_double(yprd), _p(out), _p(nf))
    return bool(rc), out, nf


def nts(true_strain, rate, dt) -> int:
    e = np.ascontiguousarray(true_strain, dtype=np.float64)
    return lib().omd_nts(_p(e), C.c_double(rate), C.c_double(dt))


def round_rate(x: float) -> float:
    return lib().omd_round_rate(x)


def rotation_tensor(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
    R = np.zeros(9)
    lib().ho_rotation_tensor(_p(a), _p(b), _p(R))
    return R.reshape(3, 3)


def rotate_sym2(t_raw, R):
    t = np.ascontiguousarray(t_raw, dtype=np.float64); R = np.ascontiguousarray(R, dtype=np.float64)
    out = np.zeros(6)
    lib().ho_rotate_sym2(_p(t), _p(R), _p(out))
    return out


def rotate_sym4(c_file, R):
    c = np.ascontiguousarray(c_file, dtype=np.float64); R = np.ascontiguousarray(R, dtype=np.float64)
    out = np.zeros(36)
    lib().ho_rotate_sym4(_p(c), _p(R), _p(out))
    return out


def prepare_strain(eps_raw, rotam, init_length, hooke):
    e = np.ascontiguousarray(eps_raw, dtype=np.float64); R = np.ascontiguousarray(rotam, dtype=np.float64)
    L0 = np.ascontiguousarray(init_length, dtype=np.float64)
    out = np.zeros(6)
    lib().ho_prepare_strain(_p(e), _p(R), _p(L0), C.c_int(1 if hooke else 0), _p(out))
    return out


def hooke(c_file, eps_raw):
    c = np.ascontiguousarray(c_file, dtype=np.float64); e = np.ascontiguousarray(eps_raw, dtype=np.float64)
    out = np.zeros(6)
    lib().ho_hooke(_p(c), _p(e), _p(out))
    return out


def store(stress_raw, init_stress_raw, rotam, hooke_mode):
    s = np.ascontiguousarray(stress_raw, dtype=np.float64).reshape(-1, 6)
    i0 = np.ascontiguousarray(init_stress_raw, dtype=np.float64).reshape(-1, 6)
    R = np.ascontiguousarray(rotam, dtype=np.float64).reshape(-1, 9)
    out = np.zeros(6)
    lib().ho_store(C.c_int(s.shape[0]), _p(s), _p(i0), _p(R), C.c_int(1 if hooke_mode else 0), _p(out))
    return out


def read_sym2(path):
    out = np.zeros(6)
    if lib().ho_read_sym2(path.encode(), _p(out)) != 0:
        raise FileNotFoundError(path)
    return out


def read_sym4(path):
    out = np.zeros(36)
    if lib().ho_read_sym4(path.encode(), _p(out)) != 0:
        raise FileNotFoundError(path)
    return out
