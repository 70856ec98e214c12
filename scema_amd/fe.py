"""ctypes binding of include/scema_fe.h: the minimal explicit-dynamics continuum stand-in that drives
STMDSync through whole continuum steps -- solve() -> STMDSync.update() -> check(), the body of HMMProblem::do_timestep
(reference dealammps.cc:417-474)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .stmd import QP

SYMBOLS = ["scema_fe_create", "scema_fe_destroy", "scema_fe_n_qp", "scema_fe_n_nodes", "scema_fe_solve", "scema_fe_check", "scema_fe_get",
           "scema_fe_set_velocity", "scema_fe_kinetic_energy"]


class Config(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32), ("lx", C.c_double), ("ly", C.c_double), ("lz", C.c_double),
                ("density", C.c_double), ("stiffness", C.c_double * 36), ("dt", C.c_double), ("top_velocity", C.c_double),
                ("min_qp_strain", C.c_double), ("hooke", C.c_int32), ("material", C.c_int32)]


class FE:
    def __init__(self, nx, ny, nz, lx, ly, lz, density, stiffness, dt, top_velocity=0.0, min_qp_strain=1e-10, hooke=False, material=0):
        L = capi.lib()
        L.scema_fe_destroy.argtypes = [C.c_void_p]
        L.scema_fe_destroy.restype = None
        L.scema_fe_n_qp.argtypes = [C.c_void_p]
        L.scema_fe_n_nodes.argtypes = [C.c_void_p]
        L.scema_fe_kinetic_energy.argtypes = [C.c_void_p]
        L.scema_fe_kinetic_energy.restype = C.c_double
        c = Config(nx, ny, nz, lx, ly, lz, density)
        c.stiffness[:] = list(np.asarray(stiffness, float).ravel())
        c.dt = dt; c.top_velocity = top_velocity; c.min_qp_strain = min_qp_strain
        c.hooke = 1 if hooke else 0
        c.material = material
        self.h = C.c_void_p()
        if L.scema_fe_create(C.byref(c), C.byref(self.h)) != 0:
            raise capi.EngineError("scema_fe_create: bad configuration")
        self.n_qp = L.scema_fe_n_qp(self.h)
        self.n_nodes = L.scema_fe_n_nodes(self.h)
        self._list = (QP * self.n_qp)()

    def close(self):
        if self.h:
            capi.lib().scema_fe_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def solve(self):
        """-> update_list as [(id, most_recent_id, material, strain6)], the argument of STMDSync.update"""
        n = C.c_int32(0)
        if capi.lib().scema_fe_solve(self.h, self._list, C.c_int32(self.n_qp), C.byref(n)) != 0:
            raise capi.EngineError("scema_fe_solve failed")
        self._n = n.value
        return [(q.id, q.most_recent_id, q.material, np.array(q.update_strain[:])) for q in self._list[:n.value]]

    def check(self, update_stress):
        """update_stress: (n_update, 6) array in the order solve() returned the list"""
        s = np.asarray(update_stress, float).reshape(-1, 6)
        assert len(s) == self._n
        for k in range(self._n):
            self._list[k].update_stress[:] = list(s[k])
        if capi.lib().scema_fe_check(self.h, self._list, C.c_int32(self._n)) != 0:
            raise capi.EngineError("scema_fe_check failed")

    def get(self):
        u = np.zeros((self.n_nodes, 3)); e = np.zeros((self.n_qp, 6)); s = np.zeros((self.n_qp, 6))
        capi.lib().scema_fe_get(self.h, capi._p(u), capi._p(e), capi._p(s))
        return u, e, s

    def set_velocity(self, v):
        v = np.ascontiguousarray(v, np.float64)
        assert v.shape == (self.n_nodes, 3)
        capi.lib().scema_fe_set_velocity(self.h, capi._p(v))

    def kinetic_energy(self) -> float:
        return capi.lib().scema_fe_kinetic_energy(self.h)

    def node_coords(self, nx, ny, nz, lx, ly, lz):
        g = np.mgrid[0:nz + 1, 0:ny + 1, 0:nx + 1].reshape(3, -1).T[:, ::-1].astype(float)   # node = (k*nny + j)*nnx + i
        return g * np.array([lx / nx, ly / ny, lz / nz])
