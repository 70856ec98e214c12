"""ctypes binding of include/scema_stmd.h: the host layer that mirrors SCEMa's STMDSync
(reference headers/stmd_sync.h).  `STMDSync.init/update` keep the reference's names and argument
meaning; the multi-rank all-gather is supplied by the caller (torch.distributed in this repo)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

SYMBOLS = ["scema_stmd_create", "scema_stmd_destroy", "scema_stmd_last_error", "scema_stmd_init", "scema_stmd_set_lammps_state_files", "scema_stmd_set_md_procs",
           "scema_stmd_update", "scema_stmd_replica_data", "scema_eqmd_equil", "scema_eqmd_equil_full"]


class QP(C.Structure):
    """HMM::QP, 112 bytes (reference headers/scale_bridging_data.h:12-19)."""
    _fields_ = [("id", C.c_int32), ("most_recent_id", C.c_int32), ("material", C.c_int32),
                ("update_strain", C.c_double * 6), ("update_stress", C.c_double * 6)]


assert C.sizeof(QP) == 112


class Config(C.Structure):
    _fields_ = [("start_timestep", C.c_int32), ("md_timestep_length", C.c_double), ("md_temperature", C.c_double),
                ("md_nsteps_sample", C.c_int32), ("md_strain_rate", C.c_double), ("md_force_field", C.c_char_p),
                ("nanostatelocin", C.c_char_p), ("nanostatelocout", C.c_char_p), ("nanostatelocres", C.c_char_p),
                ("nanologloc", C.c_char_p), ("macrostatelocout", C.c_char_p), ("md_scripts_directory", C.c_char_p),
                ("freq_checkpoint", C.c_int32), ("freq_output_homog", C.c_int32), ("n_materials", C.c_int32),
                ("mdtype", C.POINTER(C.c_char_p)), ("cg_dir", C.c_double * 3), ("nrepl", C.c_int32),
                ("use_pjm_scheduler", C.c_int32), ("approx_md_with_hookes_law", C.c_int32), ("verbose", C.c_int32)]


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_double))


def torch_allgather(engine: "capi.Engine | None", rank: int, world: int):
    """All-gather through torch.distributed: device buffers over RCCL when there is an engine
    (backend nccl), host tensors otherwise (gloo; the Hooke test mode)."""
    import torch
    import torch.distributed as dist

    def fn(ctx, eng_ptr, local_host, count, gathered_host):
        try:
            out = np.ctypeslib.as_array(gathered_host, shape=(count * world,))
            if engine is not None and eng_ptr:
                send = torch.empty(count, dtype=torch.float64, device="cuda")
                recv = torch.empty(count * world, dtype=torch.float64, device="cuda")
                engine.copy_local_stress(send.data_ptr(), True)
                dist.all_gather_into_tensor(recv, send)
                out[:] = recv.cpu().numpy()
            else:
                loc = torch.from_numpy(np.ctypeslib.as_array(local_host, shape=(count,)).copy())
                parts = [torch.empty_like(loc) for _ in range(world)]
                dist.all_gather(parts, loc)
                out[:] = torch.cat(parts).numpy()
            return 0
        except Exception as exc:  # pragma: no cover
            print("allgather failed:", exc, flush=True)
            return 1

    return ALLGATHER_FN(fn)


class STMDSync:
    """Mirror of HMM::STMDSync<3> (reference headers/stmd_sync.h:53-156)."""

    def __init__(self, engine: "capi.Engine | None", rank: int = 0, world: int = 1, allgather=None):
        L = capi.lib()
        L.scema_stmd_last_error.restype = C.c_char_p
        L.scema_stmd_last_error.argtypes = [C.c_void_p]
        L.scema_stmd_destroy.argtypes = [C.c_void_p]
        L.scema_stmd_destroy.restype = None
        self._cb = allgather  # keep alive
        self.h = C.c_void_p()
        eh = engine.h if engine is not None else None
        rc = L.scema_stmd_create(eh, C.c_int32(rank), C.c_int32(world), allgather if allgather is not None else None,
                                 None, C.byref(self.h))
        if rc != 0:
            raise capi.EngineError(f"scema_stmd_create rc={rc}")
        self.engine = engine

    def set_lammps_state_files(self, on: bool = True):
        self._chk(capi.lib().scema_stmd_set_lammps_state_files(self.h, C.c_int32(1 if on else 0)))

    def close(self):
        if self.h:
            capi.lib().scema_stmd_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise capi.EngineError(f"rc={rc}: {capi.lib().scema_stmd_last_error(self.h).decode()}")

    def init(self, *, start_timestep=1, md_timestep_length=2.0, md_temperature=300.0, md_nsteps_sample=100,
             md_strain_rate=1e-4, md_force_field="opls", nanostatelocin="", nanostatelocout="", nanostatelocres="",
             nanologloc="none", macrostatelocout="", md_scripts_directory="", freq_checkpoint=100, freq_output_homog=1000,
             mdtype=("g0",), cg_dir=(1.0, 0.0, 0.0), nrepl=1, use_pjm_scheduler=False, approx_md_with_hookes_law=False,
             verbose=False):
        c = Config()
        c.start_timestep = start_timestep
        c.md_timestep_length = md_timestep_length; c.md_temperature = md_temperature
        c.md_nsteps_sample = md_nsteps_sample; c.md_strain_rate = md_strain_rate
        c.md_force_field = md_force_field.encode()
        c.nanostatelocin = nanostatelocin.encode(); c.nanostatelocout = nanostatelocout.encode()
        c.nanostatelocres = nanostatelocres.encode(); c.nanologloc = nanologloc.encode()
        c.macrostatelocout = macrostatelocout.encode(); c.md_scripts_directory = md_scripts_directory.encode()
        c.freq_checkpoint = freq_checkpoint; c.freq_output_homog = freq_output_homog
        names = (C.c_char_p * len(mdtype))(*[m.encode() for m in mdtype])
        c.n_materials = len(mdtype); c.mdtype = names
        c.cg_dir[:] = list(cg_dir)
        c.nrepl = nrepl
        c.use_pjm_scheduler = 1 if use_pjm_scheduler else 0
        c.approx_md_with_hookes_law = 1 if approx_md_with_hookes_law else 0
        c.verbose = 1 if verbose else 0
        self._cfg = (c, names)
        self._chk(capi.lib().scema_stmd_init(self.h, C.byref(c)))

    def update(self, timestep: int, present_time: float, newtonstep: int, qps):
        """qps: list of (id, most_recent_id, material, strain6).  Returns an (n,6) array of update_stress."""
        arr = (QP * len(qps))()
        for k, (qid, recent, mat, strain) in enumerate(qps):
            arr[k].id = qid; arr[k].most_recent_id = recent; arr[k].material = mat
            arr[k].update_strain[:] = list(np.asarray(strain, float))
        self._chk(capi.lib().scema_stmd_update(self.h, C.c_int32(timestep), C.c_double(present_time), C.c_int32(newtonstep),
                                               arr, C.c_int32(len(arr))))
        return np.array([list(a.update_stress) for a in arr])

    def replica_data(self, material: int, replica0: int):
        L0 = np.zeros(3); s0 = np.zeros(6); R = np.zeros(9); rho = C.c_double()
        self._chk(capi.lib().scema_stmd_replica_data(self.h, C.c_int32(material), C.c_int32(replica0), capi._p(L0), capi._p(s0),
                                                     capi._p(R), C.byref(rho)))
        return dict(init_length=L0, init_stress=s0, rotam=R.reshape(3, 3), rho=rho.value)


def set_md_procs(nmdruns: int, n_processes: int, this_process: int, min_cores: int, cores_per_node: int):
    """STMDSync::set_md_procs (stmd_sync.h:189-278): (ranks per batch, number of batches, colour of this rank; -1 = left over)"""
    nb = C.c_int32(); nbt = C.c_int32(); col = C.c_int32()
    rc = capi.lib().scema_stmd_set_md_procs(C.c_int32(nmdruns), C.c_int32(n_processes), C.c_int32(this_process), C.c_int32(min_cores),
                                            C.c_int32(cores_per_node), C.byref(nb), C.byref(nbt), C.byref(col))
    if rc != 0:
        raise capi.EngineError("md_batch_n_processes is not well set")
    return nb.value, nbt.value, col.value


def eqmd_equil(engine: "capi.Engine", cmat: str, folder: str, rep: int, *, mdts=2.0, mdtem=300.0, mdnss=100, mdss=1e-4, mdsa=0.005,
               mdff="opls"):
    """EQMDProblem::equil for an equilibrated, registered replica: writes init.<cmat>_<rep>.{length,stress,stiff}."""
    import os
    base = os.path.join(folder, f"init.{cmat}_{rep}")
    err = C.create_string_buffer(512)
    rc = capi.lib().scema_eqmd_equil(engine.h, cmat.encode(), (base + ".length").encode(), (base + ".stress").encode(),
                                     (base + ".stiff").encode(), C.c_int32(rep), C.c_double(mdts), C.c_double(mdtem), C.c_int32(mdnss),
                                     C.c_double(mdss), C.c_double(mdsa), mdff.encode(), err, C.c_int32(512))
    if rc != 0:
        raise capi.EngineError(f"eqmd_equil rc={rc}: {err.value.decode()}")
    return base


def eqmd_equil_full(engine: "capi.Engine", cmat: str, slocin: str, folder: str, rep: int, *, mdts=2.0, mdtem=300.0, mdnss=100, mdnse=1000,
                    mdss=1e-4, mdsa=0.005, mdff="opls", qplogloc="", scrloc=""):
    """EQMDProblem::equil with the reference's argument list: if <folder>/init.<cmat>_<rep>.bin is missing, the replica is read
    from <slocin>/<cmat>_<rep>.data and equilibrated (in.init.lammps on the GPU), then the three init.* files are written."""
    import os
    base = os.path.join(folder, f"init.{cmat}_{rep}")
    err = C.create_string_buffer(512)
    rc = capi.lib().scema_eqmd_equil_full(engine.h, cmat.encode(), slocin.encode(), qplogloc.encode(), scrloc.encode(), (base + ".length").encode(),
                                          (base + ".stress").encode(), (base + ".stiff").encode(), (base + ".bin").encode(), C.c_int32(rep),
                                          C.c_double(mdts), C.c_double(mdtem), C.c_int32(mdnss), C.c_int32(mdnse), C.c_double(mdss), C.c_double(mdsa),
                                          mdff.encode(), err, C.c_int32(512))
    if rc != 0:
        raise capi.EngineError(f"eqmd_equil rc={rc}: {err.value.decode()}")
    return base


def write_replica_file(path: str, sysd: dict):
    s, keep = capi.make_system(sysd)
    rc = capi.lib().scema_md_write_replica_file(path.encode(), C.byref(s))
    if rc != 0:
        raise IOError(f"cannot write {path} (rc={rc})")


def write_nanoscale_input(folder: str, matid: str, replica: int, *, init_length, init_stress_raw, stiff_file_order,
                          relative_density=0.95, nsheets=0, normal=None, sysd: dict | None = None):
    """Lay down the files STMDSync::init reads (reference stmd_sync.h:280-453):
    <mat>_<r>.json, init.<mat>_<r>.{length,stress,stiff} (+ .bin when a system is given)."""
    import json
    import os
    os.makedirs(folder, exist_ok=True)
    js = {"relative_density": relative_density, "Nsheets": nsheets, "normal_vector": {}}
    if normal is not None:
        js["normal_vector"] = {"1": {"x": float(normal[0]), "y": float(normal[1]), "z": float(normal[2])}}
    with open(os.path.join(folder, f"{matid}_{replica}.json"), "w") as f:
        json.dump(js, f)
    base = os.path.join(folder, f"init.{matid}_{replica}")
    with open(base + ".length", "w") as f:
        for v in init_length:
            f.write(repr(float(v)) + "\n")
    r = np.asarray(init_stress_raw, float)
    with open(base + ".stress", "w") as f:   # file order 00,01,02,11,12,22 from raw xx,yy,zz,xy,xz,yz
        for v in (r[0], r[3], r[4], r[1], r[5], r[2]):
            f.write(repr(float(v)) + "\n")
    with open(base + ".stiff", "w") as f:
        for v in np.asarray(stiff_file_order, float).ravel():
            f.write(repr(float(v)) + "\n")
    if sysd is not None:
        write_replica_file(base + ".bin", sysd)
