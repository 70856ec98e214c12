"""ctypes binding of oracle/reax_oracle.c (ReaxFF restatement, SURVEY.md 8(f) row f-4, first step).

TEST INFRASTRUCTURE ONLY, PARITY UNPINNED (see reax_oracle.h).  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SAN = os.environ.get("SCEMA_SANITIZE") == "1"   # tools/run_asan.sh: the AddressSanitizer + UBSan build
_LIB = os.path.join(_HERE, "_build_asan" if _SAN else "_build", "libreax_oracle.so")
PARTS = ["bond", "lp", "over", "under", "angle", "pen", "coa", "tors", "conj", "hb", "vdw", "coul", "pol"]


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("reax_oracle.c", "reax_oracle.h")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["asan"] if _SAN else []))
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = _lib = C.CDLL(build())
        L.rxo_read_ffield.restype = C.c_void_p
        L.rxo_read_ffield.argtypes = [C.c_char_p]
        L.rxo_free_ffield.argtypes = [C.c_void_p]
        L.rxo_ntypes.argtypes = [C.c_void_p]
        L.rxo_type_name.restype = C.c_char_p
        L.rxo_type_name.argtypes = [C.c_void_p, C.c_int]
        L.rxo_type_mass.restype = C.c_double
        L.rxo_type_mass.argtypes = [C.c_void_p, C.c_int]
        L.rxo_general.restype = C.c_double
        L.rxo_general.argtypes = [C.c_void_p, C.c_int]
        L.rxo_energy.restype = C.c_double
        L.rxo_energy.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
        L.rxo_bond_orders.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.rxo_qeq.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p]
        L.rxo_forces_fd.restype = None
        L.rxo_forces_fd.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class ForceField:
    def __init__(self, path: str):
        self.h = lib().rxo_read_ffield(path.encode())
        if not self.h:
            raise IOError(f"cannot read ReaxFF force field {path}")
        self.ntypes = lib().rxo_ntypes(self.h)
        self.names = [lib().rxo_type_name(self.h, t).decode() for t in range(self.ntypes)]
        self.masses = [lib().rxo_type_mass(self.h, t) for t in range(self.ntypes)]

    def general(self, k: int) -> float:
        return lib().rxo_general(self.h, k)

    def types(self, symbols) -> np.ndarray:
        """element symbols -> 0-based force-field types (what `pair_coeff * * ffield H C N O` does per LAMMPS type)"""
        return np.array([self.names.index(s) for s in symbols], dtype=np.int32)

    @staticmethod
    def _args(type_, x, box, q):
        t = np.ascontiguousarray(type_, dtype=np.int32)
        xx = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, 3)
        b = None if box is None else np.ascontiguousarray(box, dtype=np.float64)
        qq = None if q is None else np.ascontiguousarray(q, dtype=np.float64)
        assert len(t) == len(xx) and (qq is None or len(qq) == len(t)) and (b is None or len(b) == 9)
        return t, xx, b, qq

    def energy(self, type_, x, box=None, q=None):
        t, xx, b, qq = self._args(type_, x, box, q)
        parts = np.zeros(len(PARTS))
        e = lib().rxo_energy(self.h, len(t), _p(t), _p(xx), _p(b), _p(qq), _p(parts))
        return e, dict(zip(PARTS, parts))

    def bond_orders(self, type_, x, box=None):
        t, xx, b, _ = self._args(type_, x, box, None)
        n = lib().rxo_bond_orders(self.h, len(t), _p(t), _p(xx), _p(b), 0, None, None)
        ij = np.zeros((n, 2), dtype=np.int32)
        bo = np.zeros((n, 3))
        lib().rxo_bond_orders(self.h, len(t), _p(t), _p(xx), _p(b), n, _p(ij), _p(bo))
        return ij, bo

    def qeq(self, type_, x, box=None, tol=1e-6, maxiter=200):
        t, xx, b, _ = self._args(type_, x, box, None)
        q = np.zeros(len(t))
        it = lib().rxo_qeq(self.h, len(t), _p(t), _p(xx), _p(b), tol, maxiter, _p(q))
        if it < 0:
            raise RuntimeError("charge equilibration did not converge")
        return q, it

    def forces(self, type_, x, box=None, q=None, h=1e-5, virial=False):
        t, xx, b, qq = self._args(type_, x, box, q)
        f = np.zeros_like(xx)
        w = np.zeros(6) if (virial and b is not None) else None
        lib().rxo_forces_fd(self.h, len(t), _p(t), _p(xx), _p(b), _p(qq), h, _p(f), _p(w))
        return (f, w) if virial else f

    def close(self):
        if self.h:
            lib().rxo_free_ffield(self.h)
            self.h = None
