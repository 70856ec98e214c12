"""ctypes binding of the C ABI declared in include/scema_md.h (libscema_md.so).

This is the only way Python reaches the engine: plain pointers and sizes, no torch types.  The
library is built in-tree by `scema_amd/csrc/Makefile` (`__graft_entry__.build()`); there is no CPU
fallback -- a missing library or a missing GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# tools/run_asan.sh (CPU tests only): the build whose host layer is compiled with AddressSanitizer + UBSan
# SCEMA_MD_LIB=<file name in this directory>: an A/B build of the library for same-box kernel comparisons (csrc/Makefile: LIBNAME);
# bench.py lists the variable under config.env_overrides
LIB_PATH = os.path.join(_HERE, os.environ.get("SCEMA_MD_LIB") or ("libscema_md_asan.so" if os.environ.get("SCEMA_SANITIZE") == "1" else "libscema_md.so"))

NPART = 8
PARTS = ["lj", "coul", "bond", "angle", "dihedral", "improper", "kspace", "shake"]
QP_NONE = -1  # uint32 max as int32 (stmd_problem.h:126)

# every symbol include/scema_md.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "scema_md_default_params", "scema_md_create", "scema_md_destroy", "scema_md_last_error",
    "scema_md_register_replica", "scema_md_load_replica_file", "scema_md_write_replica_file",
    "scema_md_load_lammps_data", "scema_md_convert_lammps_data",
    "scema_md_probe_lammps_restart", "scema_md_read_lammps_restart_atoms", "scema_md_load_lammps_restart",
    "scema_md_convert_lammps_restart", "scema_md_write_lammps_restart",
    "scema_md_strain_batch", "scema_md_strain", "scema_md_local_stress_device_ptr",
    "scema_md_local_stress_count", "scema_md_local_result_doubles", "scema_md_copy_local_stress", "scema_md_scatter_gathered", "scema_md_settle_update", "scema_md_has_state", "scema_md_get_state",
    "scema_md_set_state", "scema_md_drop_state", "scema_md_save_state_file", "scema_md_load_state_file", "scema_md_save_state_lammps",
    "scema_md_init_material", "scema_md_debug_compute", "scema_md_debug_run", "scema_md_get_profile", "scema_md_env_overrides",
    "scema_md_comm_unique_id", "scema_md_comm_init_rccl", "scema_md_comm_init_host", "scema_md_comm_destroy",
    "scema_md_comm_world", "scema_md_comm_rank", "scema_md_comm_stats", "scema_md_comm_handshakes", "scema_md_state_owner", "scema_md_last_plan",
    "scema_plan_dir_create", "scema_plan_dir_destroy", "scema_plan_update", "scema_md_kspace_setup",
    "scema_md_save_state_dump", "scema_md_replica_natoms", "scema_md_save_replica_file", "scema_md_equilibrate", "scema_md_debug_minimize", "scema_md_debug_run_nh",
    "scema_md_reax_configure", "scema_md_reax_activate", "scema_md_reax_set", "scema_md_reax_concurrency", "scema_md_batch_split", "scema_md_get_concurrency", "scema_md_pppm_plan_count", "scema_md_unsettled_updates", "scema_md_reax_debug_compute", "scema_md_reax_stats", "scema_md_box_fma_tflops",
]
COMM_ID_BYTES = 128
HOST_ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)
HOST_SEND_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32)
HOST_RECV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32)


class EquilParams(C.Structure):
    _fields_ = [("nsteps_equil", C.c_int32), ("timestep_length", C.c_double), ("temperature", C.c_double), ("seed", C.c_int64)]


class Params(C.Structure):
    _fields_ = [("cut_lj", C.c_double), ("cut_coul", C.c_double), ("skin", C.c_double), ("neigh_delay", C.c_int32),
                ("kspace_accuracy", C.c_double), ("shake_tol", C.c_double), ("shake_maxiter", C.c_int32),
                ("shake_mass", C.c_double), ("t_period", C.c_double), ("t_chain", C.c_int32), ("device", C.c_int32),
                ("max_batch", C.c_int32), ("profile", C.c_int32), ("kspace_style", C.c_int32)]


_P = C.c_void_p


class System(C.Structure):
    _fields_ = [("natoms", C.c_int32), ("ntypes", C.c_int32), ("type", _P), ("charge", _P), ("mass", _P),
                ("eps", _P), ("sigma", _P),
                ("nbonds", C.c_int32), ("nbondtypes", C.c_int32), ("bond_atoms", _P), ("bond_type", _P), ("bond_coeff", _P),
                ("nangles", C.c_int32), ("nangletypes", C.c_int32), ("angle_atoms", _P), ("angle_type", _P), ("angle_coeff", _P),
                ("ndihedrals", C.c_int32), ("ndihedraltypes", C.c_int32), ("dihedral_atoms", _P), ("dihedral_type", _P),
                ("dihedral_coeff", _P),
                ("nimpropers", C.c_int32), ("nimpropertypes", C.c_int32), ("improper_atoms", _P), ("improper_type", _P),
                ("improper_coeff", _P),
                ("special_lj", C.c_double * 3), ("special_coul", C.c_double * 3), ("box", C.c_double * 9), ("x", _P), ("v", _P)]


class MDSim(C.Structure):
    """Mirror of HMM::MDSim<3> (headers/md_sim.h:15-58)."""
    _fields_ = [("qp_id", C.c_int32), ("most_recent_qp_id", C.c_int32), ("replica", C.c_int32), ("material", C.c_int32),
                ("matid", C.c_char_p), ("time_id", C.c_char_p), ("output_folder", C.c_char_p),
                ("restart_folder", C.c_char_p), ("scripts_folder", C.c_char_p), ("log_file", C.c_char_p),
                ("force_field", C.c_char_p),
                ("strain", C.c_double * 6), ("stiffness", C.c_double * 36),
                ("timestep_length", C.c_double), ("temperature", C.c_double), ("strain_rate", C.c_double),
                ("nsteps_sample", C.c_int32), ("output_homog", C.c_int32), ("checkpoint", C.c_int32),
                ("stress", C.c_double * 6), ("stress_updated", C.c_int32)]


class EqParams(C.Structure):
    """scema_md_eqparams (include/scema_md.h)."""
    _fields_ = [("timestep_length", C.c_double), ("temperature", C.c_double), ("nsteps_sample", C.c_int32),
                ("strain_ampl", C.c_double), ("strain_rate", C.c_double)]


class Profile(C.Structure):
    _fields_ = [("pair_launches", C.c_int64), ("pair_ms", C.c_double), ("pair_alg_bytes", C.c_double),
                ("md_steps", C.c_int64), ("neigh_builds", C.c_int64), ("unique_pairs_per_sim", C.c_double),
                ("evals", C.c_int64), ("list_skin_mean", C.c_double), ("pair_sims", C.c_int64), ("box_flips", C.c_int64),
                ("rx_sweep_launches", C.c_int64), ("rx_sweep_ms", C.c_double), ("rx_sweep_entries", C.c_double), ("rx_sweep_rows", C.c_double), ("rx_sweep_col_bytes", C.c_int64),
                ("pair_union_ms", C.c_double), ("rx_sweep_union_ms", C.c_double), ("rx_sweep_symmetric", C.c_int64)]


_lib = None


def lib():
    """Load libscema_md.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(make -C scema_amd/csrc); the engine has no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.scema_md_last_error.restype = C.c_char_p
        L.scema_md_last_error.argtypes = [_P]
        L.scema_md_local_stress_device_ptr.restype = C.c_void_p
        L.scema_md_local_stress_device_ptr.argtypes = [_P]
        L.scema_md_local_stress_count.argtypes = [_P]
        L.scema_md_destroy.argtypes = [_P]
        L.scema_md_destroy.restype = None
        L.scema_md_comm_destroy.argtypes = [_P]
        L.scema_md_comm_destroy.restype = None
        L.scema_md_comm_world.argtypes = [_P]
        L.scema_md_comm_rank.argtypes = [_P]
        L.scema_plan_dir_create.restype = C.c_void_p
        L.scema_plan_dir_destroy.argtypes = [_P]
        L.scema_plan_dir_destroy.restype = None
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def default_params(**kw) -> Params:
    p = Params()
    lib().scema_md_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def make_system(d: dict):
    """Build a `System` struct from the dict produced by scema_amd.systems.build_pe; returns
    (struct, keepalive list)."""
    a = lambda k, t: np.ascontiguousarray(d[k], dtype=t)
    keep = dict(type=a("type", np.int32), charge=a("charge", np.float64), mass=a("mass", np.float64),
                eps=a("eps", np.float64), sigma=a("sigma", np.float64),
                bonds=a("bonds", np.int32), bond_type=a("bond_type", np.int32), bond_coeff=a("bond_coeff", np.float64),
                angles=a("angles", np.int32), angle_type=a("angle_type", np.int32), angle_coeff=a("angle_coeff", np.float64),
                dihedrals=a("dihedrals", np.int32), dihedral_type=a("dihedral_type", np.int32),
                dihedral_coeff=a("dihedral_coeff", np.float64),
                impropers=a("impropers", np.int32), improper_type=a("improper_type", np.int32),
                improper_coeff=a("improper_coeff", np.float64), x=a("x", np.float64), v=a("v", np.float64))
    s = System()
    s.natoms = int(d["natoms"]); s.ntypes = int(d["ntypes"])
    s.type = _p(keep["type"]); s.charge = _p(keep["charge"]); s.mass = _p(keep["mass"])
    s.eps = _p(keep["eps"]); s.sigma = _p(keep["sigma"])
    s.nbonds = len(keep["bond_type"]); s.nbondtypes = len(keep["bond_coeff"])
    s.bond_atoms = _p(keep["bonds"]); s.bond_type = _p(keep["bond_type"]); s.bond_coeff = _p(keep["bond_coeff"])
    s.nangles = len(keep["angle_type"]); s.nangletypes = len(keep["angle_coeff"])
    s.angle_atoms = _p(keep["angles"]); s.angle_type = _p(keep["angle_type"]); s.angle_coeff = _p(keep["angle_coeff"])
    s.ndihedrals = len(keep["dihedral_type"]); s.ndihedraltypes = len(keep["dihedral_coeff"])
    s.dihedral_atoms = _p(keep["dihedrals"]); s.dihedral_type = _p(keep["dihedral_type"])
    s.dihedral_coeff = _p(keep["dihedral_coeff"])
    s.nimpropers = len(keep["improper_type"]); s.nimpropertypes = len(keep["improper_coeff"])
    s.improper_atoms = _p(keep["impropers"]); s.improper_type = _p(keep["improper_type"])
    s.improper_coeff = _p(keep["improper_coeff"])
    s.special_lj[:] = list(np.asarray(d["special_lj"], float))
    s.special_coul[:] = list(np.asarray(d["special_coul"], float))
    s.box[:] = list(np.asarray(d["box"], float))
    s.x = _p(keep["x"]); s.v = _p(keep["v"])
    return s, keep


class RestartInfo(C.Structure):
    """scema_lammps_restart_info (include/scema_md.h)."""
    _fields_ = [("version", C.c_char * 32), ("units", C.c_char * 16), ("atom_style", C.c_char * 32), ("pair_style", C.c_char * 64),
                ("natoms", C.c_int64), ("ntimestep", C.c_int64), ("nbonds", C.c_int64), ("nangles", C.c_int64),
                ("ndihedrals", C.c_int64), ("nimpropers", C.c_int64),
                ("ntypes", C.c_int32), ("nbondtypes", C.c_int32), ("nangletypes", C.c_int32), ("ndihedraltypes", C.c_int32),
                ("nimpropertypes", C.c_int32), ("triclinic", C.c_int32), ("nprocs", C.c_int32), ("reserved", C.c_int32),
                ("box", C.c_double * 9), ("timestep", C.c_double), ("special_lj", C.c_double * 3), ("special_coul", C.c_double * 3),
                ("cut_lj", C.c_double), ("cut_coul", C.c_double), ("mass", C.c_double * 16), ("error", C.c_char * 160)]


def probe_lammps_restart(path: str) -> RestartInfo:
    info = RestartInfo()
    rc = lib().scema_md_probe_lammps_restart(path.encode(), C.byref(info))
    if rc != 0:
        raise IOError(f"{path}: rc={rc}: {info.error.decode()}")
    return info


def read_lammps_restart_atoms(path: str, natoms: int) -> dict:
    tag = np.zeros(natoms, np.int64); typ = np.zeros(natoms, np.int32); img = np.zeros((natoms, 3), np.int32)
    x = np.zeros((natoms, 3)); v = np.zeros((natoms, 3))
    rc = lib().scema_md_read_lammps_restart_atoms(path.encode(), C.c_int64(natoms), _p(tag), _p(typ), _p(img), _p(x), _p(v))
    if rc != 0:
        raise IOError(f"{path}: rc={rc}")
    return dict(tag=tag, type=typ, image=img, x=x, v=v)


def write_lammps_restart(path: str, sysd: dict, cut_lj: float, cut_coul: float, timestep: float = 2.0, ntimestep: int = 0):
    s, keep = make_system(sysd)
    rc = lib().scema_md_write_lammps_restart(path.encode(), C.byref(s), C.c_double(cut_lj), C.c_double(cut_coul),
                                             C.c_double(timestep), C.c_int64(ntimestep))
    if rc != 0:
        raise IOError(f"cannot write {path} (rc={rc})")


REAX_PARTS = ["bond", "lp", "over", "under", "angle", "pen", "coa", "tors", "conj", "hb", "vdw", "coul", "pol"]


def reax_system(symbols, x, box, v=None, elements=("H", "C", "N", "O"), masses=(1.008, 12.011, 14.007, 15.999)) -> dict:
    """System dict of a ReaxFF replica (atom_style charge, lammps_scripts_reax/in.set.lammps:17): LAMMPS type k+1 = elements[k],
    no bonded topology; charges are equilibrated by the engine every step."""
    n = len(symbols)
    nt = len(elements)
    z2 = np.zeros((0, 2))
    return dict(natoms=n, ntypes=nt, type=np.array([elements.index(s) for s in symbols], np.int32), charge=np.zeros(n), mass=np.array(masses, float),
                eps=np.zeros((nt, nt)), sigma=np.ones((nt, nt)), bonds=np.zeros((0, 2), np.int32), bond_type=np.zeros(0, np.int32), bond_coeff=z2,
                angles=np.zeros((0, 3), np.int32), angle_type=np.zeros(0, np.int32), angle_coeff=z2, dihedrals=np.zeros((0, 4), np.int32),
                dihedral_type=np.zeros(0, np.int32), dihedral_coeff=np.zeros((0, 4)), impropers=np.zeros((0, 4), np.int32),
                improper_type=np.zeros(0, np.int32), improper_coeff=z2, special_lj=np.zeros(3), special_coul=np.zeros(3),
                box=np.asarray(box, float), x=np.asarray(x, float), v=np.zeros((n, 3)) if v is None else np.asarray(v, float))


class EngineError(RuntimeError):
    pass


class Engine:
    """Thin owner of a `scema_md_engine*`."""

    def __init__(self, params: Params | None = None, **kw):
        L = lib()
        self.params = params if params is not None else default_params(**kw)
        self.h = C.c_void_p()
        rc = L.scema_md_create(C.byref(self.params), C.byref(self.h))
        if rc != 0:
            raise EngineError(f"scema_md_create failed (rc={rc}): no usable HIP device; the engine has no CPU fallback")
        self._natoms = {}

    def close(self):
        if self.h:
            lib().scema_md_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise EngineError(f"rc={rc}: {lib().scema_md_last_error(self.h).decode()}")

    def register_replica(self, matid: str, replica: int, sysd: dict):
        s, keep = make_system(sysd)
        self._chk(lib().scema_md_register_replica(self.h, matid.encode(), C.c_int32(replica), C.byref(s)))
        self._natoms[(matid, replica)] = int(sysd["natoms"])

    def load_replica_file(self, matid: str, replica: int, path: str, natoms: int | None = None):
        self._chk(lib().scema_md_load_replica_file(self.h, matid.encode(), C.c_int32(replica), path.encode()))
        if natoms is not None:
            self._natoms[(matid, replica)] = natoms

    def load_lammps_data(self, matid: str, replica: int, path: str, natoms: int, special_lj=None, special_coul=None):
        slj = None if special_lj is None else np.ascontiguousarray(special_lj, np.float64)
        sc = None if special_coul is None else np.ascontiguousarray(special_coul, np.float64)
        self._chk(lib().scema_md_load_lammps_data(self.h, matid.encode(), C.c_int32(replica), path.encode(), _p(slj), _p(sc)))
        self._natoms[(matid, replica)] = natoms

    def load_lammps_restart(self, matid: str, replica: int, path: str, natoms: int):
        self._chk(lib().scema_md_load_lammps_restart(self.h, matid.encode(), C.c_int32(replica), path.encode()))
        self._natoms[(matid, replica)] = natoms

    def strain_batch(self, sims, hooke: bool = False, rank: int = 0, world: int = 1):
        arr = (MDSim * len(sims))(*sims) if not isinstance(sims, C.Array) else sims
        self._chk(lib().scema_md_strain_batch(self.h, arr, C.c_int32(len(arr)), C.c_int32(1 if hooke else 0),
                                              C.c_int32(rank), C.c_int32(world)))
        return arr

    def local_stress_ptr(self):
        return lib().scema_md_local_stress_device_ptr(self.h), lib().scema_md_local_stress_count(self.h)

    def local_result_doubles(self) -> int:
        """doubles in the result buffer of the last strain_batch: 6*max(cap,1) stresses + status word + plan hash"""
        return int(lib().scema_md_local_result_doubles(self.h))

    def copy_local_stress(self, dst_ptr: int, on_device: bool):
        self._chk(lib().scema_md_copy_local_stress(self.h, C.c_void_p(dst_ptr), C.c_int32(1 if on_device else 0)))

    def scatter_gathered(self, gathered: np.ndarray, arr):
        g = np.ascontiguousarray(gathered, np.float64)
        self._chk(lib().scema_md_scatter_gathered(self.h, _p(g), arr, C.c_int32(len(arr))))

    def last_plan(self, n: int):
        """(owner[n], pos[n], cap) of the last strain_batch (host/sim_plan.h)."""
        owner = np.zeros(n, np.int32); pos = np.zeros(n, np.int32); cap = C.c_int32(0)
        self._chk(lib().scema_md_last_plan(self.h, C.c_int32(n), _p(owner), _p(pos), C.byref(cap)))
        return owner, pos, cap.value

    # ---- communicator: one process per GPU ----
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(COMM_ID_BYTES)
        rc = lib().scema_md_comm_unique_id(buf)
        if rc != 0:
            raise EngineError(f"scema_md_comm_unique_id failed (rc={rc})")
        return buf.raw

    def comm_init_rccl(self, uid: bytes, rank: int, world: int):
        assert len(uid) == COMM_ID_BYTES
        self._chk(lib().scema_md_comm_init_rccl(self.h, C.c_char_p(uid), C.c_int32(rank), C.c_int32(world)))

    def comm_init_host(self, rank: int, world: int, allgather, send=None, recv=None):
        """Host transport: allgather(send: bytes) -> bytes of all ranks; send(data: bytes, dst); recv(nbytes, src) -> bytes."""
        def _ag(ctx, sp, rp, nbytes):
            try:
                out = allgather(C.string_at(sp, nbytes))
                C.memmove(rp, out, nbytes * world)
                return 0
            except Exception:
                return 1

        def _send(ctx, p, nbytes, dst):
            try:
                send(C.string_at(p, nbytes), dst)
                return 0
            except Exception:
                return 1

        def _recv(ctx, p, nbytes, src):
            try:
                C.memmove(p, recv(nbytes, src), nbytes)
                return 0
            except Exception:
                return 1

        self._cb = (HOST_ALLGATHER_FN(_ag), HOST_SEND_FN(_send) if send else C.cast(None, HOST_SEND_FN),
                    HOST_RECV_FN(_recv) if recv else C.cast(None, HOST_RECV_FN))
        self._chk(lib().scema_md_comm_init_host(self.h, C.c_int32(rank), C.c_int32(world), self._cb[0], self._cb[1], self._cb[2], None))

    def comm_destroy(self):
        lib().scema_md_comm_destroy(self.h)

    def comm_stats(self) -> dict:
        a = C.c_int64(0); m = C.c_int64(0)
        self._chk(lib().scema_md_comm_stats(self.h, C.byref(a), C.byref(m)))
        lib().scema_md_comm_handshakes.restype = C.c_int64
        lib().scema_md_comm_handshakes.argtypes = [_P]
        return dict(allgathers=a.value, migrations=m.value, handshakes=int(lib().scema_md_comm_handshakes(self.h)))

    def state_owner(self, qp, matid, replica) -> int:
        return int(lib().scema_md_state_owner(self.h, C.c_int32(qp), matid.encode(), C.c_int32(replica)))

    def has_state(self, qp, matid, replica) -> bool:
        return bool(lib().scema_md_has_state(self.h, C.c_int32(qp), matid.encode(), C.c_int32(replica)))

    def natoms(self, matid, replica) -> int:
        """atoms of a registered replica (also one the library registered itself, e.g. from a file)"""
        if (matid, replica) not in self._natoms:
            n = int(lib().scema_md_replica_natoms(self.h, matid.encode(), C.c_int32(replica)))
            if n <= 0:
                raise EngineError(f"replica {matid}_{replica} is not registered")
            self._natoms[(matid, replica)] = n
        return self._natoms[(matid, replica)]

    def get_state(self, qp, matid, replica):
        n = self.natoms(matid, replica)
        box = np.zeros(9); x = np.zeros((n, 3)); v = np.zeros((n, 3))
        self._chk(lib().scema_md_get_state(self.h, C.c_int32(qp), matid.encode(), C.c_int32(replica), _p(box), _p(x), _p(v)))
        return box, x, v

    def set_state(self, qp, matid, replica, box, x, v):
        box = np.ascontiguousarray(box, np.float64); x = np.ascontiguousarray(x, np.float64); v = np.ascontiguousarray(v, np.float64)
        self._chk(lib().scema_md_set_state(self.h, C.c_int32(qp), matid.encode(), C.c_int32(replica), _p(box), _p(x), _p(v)))

    def drop_state(self, qp, matid, replica):
        self._chk(lib().scema_md_drop_state(self.h, C.c_int32(qp), matid.encode(), C.c_int32(replica)))

    def save_state_file(self, qp, matid, replica, path):
        self._chk(lib().scema_md_save_state_file(self.h, C.c_int32(qp), matid.encode(), C.c_int32(replica), path.encode()))

    def save_state_lammps(self, qp, matid, replica, path, timestep=2.0, ntimestep=0):
        self._chk(lib().scema_md_save_state_lammps(self.h, C.c_int32(qp), matid.encode(), C.c_int32(replica), path.encode(),
                                                   C.c_double(timestep), C.c_int64(ntimestep)))

    def save_state_dump(self, qp, matid, replica, path, ntimestep=0, precise=True):
        self._chk(lib().scema_md_save_state_dump(self.h, C.c_int32(qp), matid.encode(), C.c_int32(replica), path.encode(), C.c_int64(ntimestep),
                                                 C.c_int32(1 if precise else 0)))

    def load_state_file(self, qp, matid, replica, path):
        self._chk(lib().scema_md_load_state_file(self.h, C.c_int32(qp), matid.encode(), C.c_int32(replica), path.encode()))

    def debug_compute(self, matid, replica, qp=QP_NONE, use_shake=False):
        n = self._natoms[(matid, replica)]
        f = np.zeros((n, 3)); e = np.zeros(NPART); w = np.zeros((NPART, 6)); info = np.zeros(8)
        self._chk(lib().scema_md_debug_compute(self.h, C.c_int32(qp), matid.encode(), C.c_int32(replica),
                                               C.c_int32(1 if use_shake else 0), _p(f), _p(e), _p(w), _p(info)))
        return f, e, w, dict(g_ewald=info[0], nk=int(info[1]), npairs=info[2], tdof=info[3], t_current=info[4],
                             maxneigh_seen=int(info[5]), maxneigh=int(info[6]), nclus=int(info[7]))

    def debug_run(self, matid, replica, nsteps, dt, temperature, qp=QP_NONE, nvt=True, use_shake=True, rates=None, sample=False):
        r = None if rates is None else np.ascontiguousarray(rates, np.float64)
        pavg = np.zeros(6) if sample else None
        self._chk(lib().scema_md_debug_run(self.h, C.c_int32(qp), matid.encode(), C.c_int32(replica), C.c_int32(nsteps),
                                           C.c_double(dt), C.c_double(temperature), C.c_int32(1 if nvt else 0),
                                           C.c_int32(1 if use_shake else 0), _p(r), _p(pavg)))
        return pavg

    def init_material(self, matid, replica, dt=2.0, temperature=300.0, nss=100, strain_ampl=0.005, strain_rate=1e-4):
        """EQMDProblem::lammps_equilibration for an equilibrated replica (init_material_problem.h:196-300): returns
        (length[3], stress[6] Pa file order 00,01,02,11,12,22, stiff[6,6] Pa file order)."""
        p = EqParams(dt, temperature, nss, strain_ampl, strain_rate)
        length, stress, stiff = np.zeros(3), np.zeros(6), np.zeros(36)
        self._chk(lib().scema_md_init_material(self.h, matid.encode(), C.c_int32(replica), C.byref(p), _p(length), _p(stress), _p(stiff)))
        return length, stress, stiff.reshape(6, 6)

    # ---- ReaxFF replicas (force_field "reax") ----
    def reax_configure(self, ffield: str, elements=("H", "C", "N", "O"), qeq_tol: float = 1e-6, skin: float = -1.0):
        """pair_coeff * * <ffield> H C N O + fix qeq/reax 1 0.0 10.0 <qeq_tol> (lammps_scripts_reax/in.strain.lammps:10-12)"""
        arr = (C.c_char_p * len(elements))(*[e.encode() for e in elements])
        self._chk(lib().scema_md_reax_configure(self.h, ffield.encode(), arr, C.c_int32(len(elements)), C.c_double(qeq_tol), C.c_double(skin)))

    def reax_activate(self, on: bool = True):
        self._chk(lib().scema_md_reax_activate(self.h, C.c_int32(1 if on else 0)))

    def reax_set(self, exact_gradient: int = -1, terms: int = -1, qeq_maxiter: int = -1):
        self._chk(lib().scema_md_reax_set(self.h, C.c_int32(exact_gradient), C.c_int32(terms), C.c_int32(qeq_maxiter)))

    def reax_concurrency(self, halves: int = -1, overlap: int = -1):
        """how a ReaxFF batch is issued (two half batches on two streams; bond-order chain next to the charge chain): 1 / 0, -1 leaves it"""
        self._chk(lib().scema_md_reax_concurrency(self.h, C.c_int32(halves), C.c_int32(overlap)))

    def unsettled_updates(self) -> int:
        lib().scema_md_unsettled_updates.restype = C.c_int64
        return int(lib().scema_md_unsettled_updates(self.h))

    def batch_split(self, on: int = -1):
        self._chk(lib().scema_md_batch_split(self.h, C.c_int32(on)))

    def pppm_plan_count(self) -> int:
        """batched hipFFT plans the engine holds (one per mesh size, batch count and stream)"""
        lib().scema_md_pppm_plan_count.restype = C.c_int
        return int(lib().scema_md_pppm_plan_count(self.h))

    def concurrency(self) -> dict:
        """the current issue settings: {"split": 0/1, "reax_halves": n, "reax_overlap": 0/1}"""
        out = (C.c_int32 * 3)()
        self._chk(lib().scema_md_get_concurrency(self.h, out))
        return dict(split=int(out[0]), reax_halves=int(out[1]), reax_overlap=int(out[2]))

    def reax_compute(self, matid, replica, qp=QP_NONE):
        n = self._natoms[(matid, replica)]
        f = np.zeros((n, 3)); e = np.zeros(len(REAX_PARTS)); w = np.zeros(6); q = np.zeros(n); info = np.zeros(6)
        self._chk(lib().scema_md_reax_debug_compute(self.h, C.c_int32(qp), matid.encode(), C.c_int32(replica), _p(f), _p(e), _p(w), _p(q), _p(info)))
        return dict(f=f, e=dict(zip(REAX_PARTS, e)), w=w, q=q, maxneigh_seen=int(info[0]), maxnb=int(info[1]), maxbd=int(info[2]),
                    qeq_iters=int(info[3]), image_search=int(info[4]), maxbonds_seen=int(info[5]))

    # ---- init_material's equilibration schedule (in.init.lammps; md_equil.hip) ----
    def minimize(self, matid, replica, qp, etol=1e-7, ftol=1e-11, maxiter=1000, maxeval=50000) -> dict:
        info = np.zeros(5)
        self._chk(lib().scema_md_debug_minimize(self.h, C.c_int32(qp), matid.encode(), C.c_int32(replica), C.c_double(etol), C.c_double(ftol),
                                                C.c_int32(maxiter), C.c_int32(maxeval), _p(info)))
        return dict(stop=int(info[0]), iterations=int(info[1]), evaluations=int(info[2]), e_initial=info[3], e_final=info[4])

    def run_nh(self, matid, replica, qp, nsteps, dt, t_start, t_stop=None, npt=False, p_target=1.0, p_period=1000.0, average_lengths=False):
        lav = np.zeros(3) if average_lengths else None
        self._chk(lib().scema_md_debug_run_nh(self.h, C.c_int32(qp), matid.encode(), C.c_int32(replica), C.c_int32(nsteps), C.c_double(dt),
                                              C.c_double(t_start), C.c_double(t_start if t_stop is None else t_stop), C.c_int32(1 if npt else 0),
                                              C.c_double(p_target), C.c_double(p_period), _p(lav) if average_lengths else None))
        return lav

    def equilibrate(self, matid, replica, nsteps_equil, dt, temperature, seed=1234):
        p = EquilParams(nsteps_equil, dt, temperature, seed)
        length, info = np.zeros(3), np.zeros(5)
        self._chk(lib().scema_md_equilibrate(self.h, matid.encode(), C.c_int32(replica), C.byref(p), _p(length), _p(info)))
        return length, dict(stop=int(info[0]), iterations=int(info[1]), evaluations=int(info[2]), e_initial=info[3], e_final=info[4])

    def reax_stats(self) -> dict:
        out = np.zeros(7)
        self._chk(lib().scema_md_reax_stats(self.h, _p(out)))
        return dict(qeq_iters=int(out[0]), qeq_solves=int(out[1]), skin=out[2], qeq_tol=out[3], qeq_slow_solves=int(out[4]),
                    qeq_launched_iters=int(out[5]), precond_fallbacks=int(out[6]))

    def profile(self, reset=False) -> dict:
        p = Profile()
        self._chk(lib().scema_md_get_profile(self.h, C.byref(p), C.c_int32(1 if reset else 0)))
        return {k: getattr(p, k) for k, _ in Profile._fields_}


def env_overrides() -> list:
    """the library's declared environment switches that are set in this process ("NAME=value"); empty in a clean run"""
    buf = C.create_string_buffer(4096)
    lib().scema_md_env_overrides.restype = C.c_int
    n = lib().scema_md_env_overrides(buf, C.c_int(len(buf)))
    return [l for l in buf.value.decode().split("\n") if l] if n else []


def box_fp64_tflops(device: int = 0) -> float:
    """FP64 FMA ceiling of the device as measured now (scema_md_box_fma_tflops): the calibration bench.py prints beside its rates"""
    out = C.c_double(0.0)
    lib().scema_md_box_fma_tflops.restype = C.c_int
    rc = lib().scema_md_box_fma_tflops(C.c_int32(device), C.byref(out))
    if rc:
        raise EngineError(f"scema_md_box_fma_tflops failed with code {rc} (no HIP device?)")
    return float(out.value)


def kspace_setup(params, box, qsqsum: float, natoms: int):
    """(g_initial, g_ewald, grid) a run would use for this box: scema_md_kspace_setup, a pure host function (no GPU)"""
    L = lib()
    L.scema_md_kspace_setup.restype = C.c_int
    b = np.ascontiguousarray(box, np.float64)
    g0, g1 = C.c_double(0.0), C.c_double(0.0)
    grid = (C.c_int32 * 3)()
    rc = L.scema_md_kspace_setup(C.byref(params), _p(b), C.c_double(qsqsum), C.c_int32(natoms), C.byref(g0), C.byref(g1), grid)
    if rc != 0:
        raise EngineError(f"scema_md_kspace_setup rc={rc}")
    return g0.value, g1.value, tuple(grid)


def make_sim(qp_id: int, matid: str, replica: int, strain_len, *, most_recent: int | None = None, material: int = 0,
             dt=2.0, temperature=300.0, strain_rate=1e-4, nss=100, force_field="opls", stiffness=None,
             time_id="0-0", output_folder="", restart_folder="", scripts_folder="", log_file="none",
             checkpoint=False) -> MDSim:
    m = MDSim()
    m.qp_id = qp_id
    m.most_recent_qp_id = qp_id if most_recent is None else most_recent
    m.replica = replica
    m.material = material
    m.matid = matid.encode(); m.time_id = time_id.encode(); m.output_folder = output_folder.encode()
    m.restart_folder = restart_folder.encode(); m.scripts_folder = scripts_folder.encode()
    m.log_file = log_file.encode(); m.force_field = force_field.encode()
    m.strain[:] = list(np.asarray(strain_len, float))
    if stiffness is not None:
        m.stiffness[:] = list(np.asarray(stiffness, float).ravel())
    m.timestep_length = dt; m.temperature = temperature; m.strain_rate = strain_rate
    m.nsteps_sample = nss
    m.output_homog = 0
    m.checkpoint = 1 if checkpoint else 0
    return m


class PlanDir:
    """The planner alone (scema_plan_* of include/scema_md.h; pure host arithmetic, runs without a GPU)."""

    def __init__(self):
        self.h = C.c_void_p(lib().scema_plan_dir_create())

    def __del__(self):
        try:
            if self.h:
                lib().scema_plan_dir_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def update(self, sims, world: int, cost=None, commit: bool = True):
        arr = (MDSim * len(sims))(*sims) if not isinstance(sims, C.Array) else sims
        n = len(arr)
        owner = np.zeros(n, np.int32); pos = np.zeros(n, np.int32); cap = C.c_int32(0)
        moves = np.zeros((max(n, 1), 3), np.int32); nm = C.c_int32(0)
        c = None if cost is None else np.ascontiguousarray(cost, np.float64)
        rc = lib().scema_plan_update(self.h, arr, C.c_int32(n), _p(c), C.c_int32(world), _p(owner), _p(pos), C.byref(cap), _p(moves),
                                     C.byref(nm), C.c_int32(1 if commit else 0))
        if rc != 0:
            raise EngineError(f"scema_plan_update rc={rc}")
        return owner, pos, cap.value, moves[:nm.value].copy()
