#!/usr/bin/env python3
"""bench.py -- stress evaluations per second of the MI355X-native strained-MD path.

One "step" = one STMDSync::update() worth of work: a batch of `--sims` quadrature-point replicas
(default 576 x PE-10k), each strained for nts=10 MD steps and sampled for nss=100 MD
steps, starting from the state the previous step left in HBM.  The replica is equilibrated once, outside
the timed region (2 000 NVT+SHAKE steps from the synthetic crystal), so that the first and the last
update of a run do the same work.  With N GPUs (one process per GPU) the batch is sharded by the engine's
planner (scema_amd/csrc/host/sim_plan.h; a fresh balanced batch: simulation i -> rank i % N,
stmd_sync.h:583) and the stresses return through ONE ncclAllGather (RCCL over xGMI) inside
scema_md_strain_batch, so the total work per step is fixed ("strong" scaling).

`python bench.py --gpus N` without RANK in the environment starts the N ranks itself (before anything
touches a GPU) and fails if the node has fewer than N devices.

`--force-field reax` runs BASELINE config 5 instead: 72 replicas of a 1 620-atom polyethylene cell with md_force_field "reax"
(lammps_scripts_reax: ffield.reax.2, QEq to 1e-6 every step, dt 0.25 fs, 10 + 20 MD steps per evaluation), same JSON contract,
with the roofline block of its HBM-bound kernel (the matrix sweep of the charge equilibration) and its own CPU baseline.

Prints ONE JSON line on rank 0 (contract in the task statement).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

EQ_QP = 1 << 30   # scratch quadrature-point id of the equilibration run


def _cpu_eval(k, cells, strain, nss, pppm=1):
    """one oracle evaluation in a worker process (test infrastructure timed as the CPU baseline)"""
    from oracle import pyoracle as po
    from scema_amd.systems import build_pe
    d = build_pe(*cells, shake_project=True)
    o = po.Oracle(d, po.default_params(kspace_pppm=pppm))
    t0 = time.time()
    _, nts = o.eval(strain, 2.0, 300.0, 1e-4, nss)
    t1 = time.time()
    return k, t0, t1, int(nts), o.timing()


def _lammps_baseline(lmp, scripts, cells, strains, nss, ncore):
    """The reference's own CPU path: its three LAMMPS scripts on the exported replica (tools/export_lammps_case.py), one
    serial LAMMPS per core (stmd_sync.h:189-278 with n_sims >= ranks); only the two lifetimes of an evaluation are timed."""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import export_lammps_case as x
    from scema_amd.systems import build_pe
    d = build_pe(*cells, shake_project=True)
    base = tempfile.mkdtemp(prefix="scema_lmp_")
    dirs = []
    for k in range(ncore):
        out = os.path.join(base, str(k))
        x.export(out, d, strains[k], scripts, nss=nss)
        subprocess.check_call([lmp, "-in", "make_init.lammps", "-log", "none", "-screen", "none"], cwd=out)
        dirs.append(out)
    t0 = time.time()
    procs = [subprocess.Popen(f"{lmp} -in phase_a.lammps -log none -screen none && {lmp} -in phase_b.lammps -log none -screen none", shell=True, cwd=o)
             for o in dirs]
    rcs = [p.wait() for p in procs]
    dt = time.time() - t0
    if any(rcs):
        raise RuntimeError(f"LAMMPS failed in {base}")
    return ncore / dt, dt


CORES_RULE = {"rule": None}


def _host_cores(per_process_gb, cap=None):
    """processes the CPU baseline may start: every host core THIS JOB MAY USE -- the affinity mask and the cgroup CPU quota, not
    os.cpu_count() (a GPU box reports 256 cores and gives a one-GPU job 16 of them: 256 workers then take 12 times as long) --
    unless memory (or a cap) says fewer.  Which rule decided is recorded (CORES_RULE) and printed with the baseline."""
    total = os.cpu_count() or 1
    n, rule = total, "os.cpu_count"
    try:
        a = len(os.sched_getaffinity(0))
        if a < n:
            n, rule = a, "affinity mask"
    except Exception:
        pass
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: None if t.split()[0] == "max" else float(t.split()[0]) / float(t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: None if int(t) <= 0 else int(t) / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()))):
        try:
            q = parse(open(path).read())
            if q and max(1, int(q + 0.5)) < n:
                n, rule = max(1, int(q + 0.5)), "cgroup cpu quota"
        except Exception:
            pass
    # Neither the affinity mask nor a cgroup quota constrains this job: a one-GPU job of this pool still gets only 16 of the host's
    # cores and nothing in /sys says so (measured: 256 workers on a "256-core" box deliver 0.68 evaluations/s, 32 deliver 1.1, i.e.
    # both are oversubscribed), so 16 is the pool default for an unconstrained job; a constraint found above is used as it is
    if rule == "os.cpu_count" and n > 16:
        n, rule = 16, "pool default (no affinity or cgroup constraint found; a one-GPU job's share of the host)"
    elif n > 16:   # a mask or quota wider than that share: 32 workers were measured oversubscribed here; SCEMA_CPU_BASELINE_CORES overrides
        n, rule = 16, rule + " of more than 16, capped at the pool's 16-core share"
    if os.environ.get("SCEMA_CPU_BASELINE_CORES"):
        n, rule = max(1, int(os.environ["SCEMA_CPU_BASELINE_CORES"])), "SCEMA_CPU_BASELINE_CORES"
    try:
        import psutil
        m = max(1, int(psutil.virtual_memory().available / 2**30 / per_process_gb * 0.6))
        if m < n:
            n, rule = m, "available memory"
    except Exception:
        pass
    if cap and cap < n:
        n, rule = cap, f"cap of {cap} workers"
    CORES_RULE["rule"] = rule
    return max(1, n)


def _cpu_eval_reax(k, cells, strain, nss, dt, rate):
    """one evaluation of the ReaxFF oracle (oracle/reax_md.py) in a worker process, one thread"""
    import torch
    torch.set_num_threads(1)
    from oracle import reax_md
    from scema_amd.systems import build_pe
    d = build_pe(*cells)
    sym = ["C" if d["mass"][t] > 5 else "H" for t in d["type"]]
    lt = np.array([1 if c == "C" else 0 for c in sym])
    m = np.array([12.011 if c == "C" else 1.008 for c in sym])
    v = np.random.default_rng(3).standard_normal((len(sym), 3)) * np.sqrt(0.0019872067 * 300.0 / (m[:, None] * 48.88821291 ** 2))
    v -= (m[:, None] * v).sum(0) / m.sum()
    M = reax_md.ReaxMD(os.path.join(ROOT, "examples", "ffield.reax.2"), ["H", "C", "N", "O"], lt, [1.008, 12.011, 14.007, 15.999], d["box"], d["x"], v)
    t0 = time.time()
    _, nts = M.eval(strain, dt, 300.0, rate, nss)
    return k, t0, time.time(), int(nts), M.qeq_iters / max(M.qeq_solves, 1)


def cpu_baseline_reax(cells, strains, nss, dt, rate, cap=64):
    """The ReaxFF oracle timed on the host cores, one single-threaded process per core (capped at 64: every process holds a
    reverse-mode graph of ~0.6 GB): CPU restatement (oracle/reax_md.py: torch FP64 autograd forces + CG), NOT LAMMPS."""
    import concurrent.futures as cf
    import multiprocessing as mp
    ncore = min(_host_cores(1.0, cap=cap), len(strains))
    # torch's autograd engine asks for the GPU count on its first backward(), which opens /dev/kfd: the workers are CPU-only and
    # must not count as users of the box's GPU (oracle/nogpu_shim.c hides the device nodes from them)
    shim = os.path.join(ROOT, "oracle", "_build", "libnogpu_shim.so")
    old = os.environ.get("LD_PRELOAD")
    if os.path.exists(shim):
        os.environ["LD_PRELOAD"] = shim + ((":" + old) if old else "")
    try:
        with cf.ProcessPoolExecutor(max_workers=ncore, mp_context=mp.get_context("spawn")) as ex:
            res = list(ex.map(_cpu_eval_reax, range(ncore), [cells] * ncore, [strains[k] for k in range(ncore)], [nss] * ncore, [dt] * ncore, [rate] * ncore))
    finally:
        if old is None:
            os.environ.pop("LD_PRELOAD", None)
        else:
            os.environ["LD_PRELOAD"] = old
    dtw = max(r[2] for r in res) - min(r[1] for r in res)
    natoms = 12 * cells[0] * cells[1] * cells[2]
    return {"value": ncore / dtw, "unit": "evals/s", "cores": ncore, "kind": "port", "lammps_on_this_host": None,
            "sample": f"{ncore} PE-{natoms} ReaxFF evaluations ({res[0][3]}+{nss} MD steps each, {res[0][4]:.1f} CG iterations per solve), one single-threaded process "
                      f"per core of this job's CPU share ({ncore} cores; the host reports {os.cpu_count()}): {dtw:.1f} s wall, {np.mean([r[2] - r[1] for r in res]):.1f} s mean per evaluation; "
                      "CPU restatement (oracle/reax_md.py), not LAMMPS USER-REAXC"}


def cpu_baseline(cells, strains, nss, pppm=1):
    """The CPU path timed on the host cores the way the reference runs it: one serial MD engine per core, one replica each
    (stmd_sync.h:189-278 with n_sims >= ranks).  If a LAMMPS executable and the reference's scripts ($SCEMA_SCRIPTS) are on
    this host, that IS the reference path ("kind": "reference"); otherwise the oracle (CPU restatement, NOT LAMMPS), one
    PROCESS per core, started before this program touches the GPU; wall = first evaluation start to last evaluation end."""
    import concurrent.futures as cf
    import multiprocessing as mp
    import shutil
    ncore = max(1, min(len(strains), _host_cores(0.4)))   # every core this job may use (see _host_cores), memory permitting
    lmp = next((shutil.which(n) for n in (os.environ.get("SCEMA_LAMMPS") or "lmp", "lmp_serial", "lmp_mpi", "lammps") if shutil.which(n)), None)
    scripts = os.environ.get("SCEMA_SCRIPTS", "")
    natoms = 12 * cells[0] * cells[1] * cells[2]
    if lmp and os.path.isdir(scripts):
        try:
            rate, dt = _lammps_baseline(lmp, scripts, cells, strains, nss, ncore)
            return {"value": rate, "unit": "evals/s", "cores": ncore, "kind": "reference", "lammps_on_this_host": lmp,
                    "sample": f"{ncore} PE-{natoms} evaluations through the reference's in.set / in.strain / ELASTIC/in.homogenization scripts, one serial "
                              f"LAMMPS per host core on {ncore} of {os.cpu_count()} cores: {dt:.1f} s wall"}
        except Exception as exc:
            print("bench.py: LAMMPS baseline failed, falling back to the CPU restatement:", exc, file=sys.stderr)
    with cf.ProcessPoolExecutor(max_workers=ncore, mp_context=mp.get_context("spawn")) as ex:
        res = list(ex.map(_cpu_eval, range(ncore), [cells] * ncore, [strains[k] for k in range(ncore)], [nss] * ncore, [pppm] * ncore))
    t0 = min(r[1] for r in res)
    t1 = max(r[2] for r in res)
    dt = t1 - t0
    tm = res[0][4]
    alone = np.mean([r[2] - r[1] for r in res])
    return {"value": ncore / dt, "unit": "evals/s", "cores": ncore, "kind": "port",
            "lammps_on_this_host": lmp,   # a LAMMPS found here (with $SCEMA_SCRIPTS) would be the real baseline; none is installed on these images
            "sample": f"{ncore} PE-{natoms} evaluations ({res[0][3]}+{nss} MD steps each), one process per core of this job's CPU share ({ncore} cores; the host "
                      f"reports {os.cpu_count()}): {dt:.1f} s wall, {alone:.1f} s mean per evaluation (replica 0: pair {tm['pair']:.1f} s, "
                      f"kspace {tm['kspace']:.1f} s, neigh {tm['neigh']:.1f} s); CPU restatement (oracle/md_oracle.c), not LAMMPS"}


def kernel_source_hashes():
    """git blob hashes (sha1 of "blob <size>\\0" + content: what `git hash-object` prints) of the sources k_pair is compiled from"""
    import hashlib
    out = {}
    for f in ("md_pair.hip", "md_pair_dev.h", "md_device.h", "md_types.h"):
        data = open(os.path.join(ROOT, "scema_amd", "csrc", f), "rb").read()
        out[f] = hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()
    return out


def count_gpus_without_hip():
    """GPUs of this node from the KFD topology in sysfs (nodes with SIMDs; CPUs have simd_count 0): the parent that forks the rank
    launcher must not have opened /dev/kfd, and torch.cuda.device_count() falls back to hipGetDeviceCount where amdsmi is absent."""
    import glob
    n = 0
    for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            props = dict(line.split()[:2] for line in open(path) if len(line.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
        except OSError:
            pass
    vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")
    if vis:
        n = min(n, len([v for v in vis.split(",") if v.strip() != ""]))
    return n


def spawn_ranks(args):
    """--gpus N from a plain shell: start N ranks (torch.distributed.run) BEFORE any GPU call in this process."""
    have = args.gpus if args.dry_run_ranks else count_gpus_without_hip()
    if have == 0:   # no KFD topology in this container's sysfs: ask torch (amdsmi where present; otherwise this does open the device)
        import torch
        have = torch.cuda.device_count()
    if have < args.gpus and not args.share_gpus and not args.dry_run_ranks:
        print(f"bench.py: --gpus {args.gpus} but this node has {have} GPU(s)", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))


def _workload(args):
    """what a leg needs before the GPU is touched: the replica, its box lengths, time step and strain rate"""
    from scema_amd.systems import build_pe
    reax = args.force_field == "reax"
    DT = 0.25 if reax else 2.0                      # fs; ReaxFF needs the short step (bond orders change within femtoseconds)
    d = build_pe(*args.cells, shake_project=not reax)   # seed 1234, 300 K, SHAKE-projected velocities
    lens = d["box"][3:6] - d["box"][:3]
    rate = 1e-3 if reax else (2e-4 if args.strain_set == "file3d" else 1e-4)
    return d, lens, DT, rate


def _reax_leg_args(args):
    """BASELINE config 5 as a short second leg of the default run: 72 x PE-1620 ReaxFF replicas, 2 warm-up + 4 timed updates"""
    import copy
    r = copy.copy(args)
    r.force_field, r.sims, r.cells, r.nss, r.equil_steps = "reax", 72, [3, 5, 9], 20, 200
    r.steps, r.warmup, r.monotonic_updates, r.monotonic, r.strain_set, r.equil_cache, r.share8_updates = 4, 2, 0, False, "balanced", None, 0
    return r


def env_overrides():
    """every SCEMA_* variable of this process's environment that the engine or this program reads: a reported run has none"""
    from scema_amd import capi
    eng_side = capi.env_overrides()
    mine = [f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("SCEMA_BENCH_") or k in ("SCEMA_CPU_BASELINE_CORES", "SCEMA_LAMMPS", "SCEMA_SCRIPTS", "SCEMA_MD_LIB", "SCEMA_SANITIZE")]
    return eng_side + mine


class Requests:
    """The request vector of an update (one MDSim per quadrature point, what prepare_md_simulations fills in C++ in the reference,
    stmd_sync.h:491-568), built once; every update only rewrites the strains and most_recent ids in place.  Host code only: the
    dry run of the control plane (--dry-run-ranks) builds exactly the requests the engine would get."""

    def __init__(self, args, n, lens, rate, DT):
        self.args, self.n, self.lens, self.rate, self.DT = args, n, lens, rate, DT
        self.mono = {"on": args.monotonic}
        self.arr = None
        self.nts_mean = 10.0

    def build(self, istep):
        import ctypes
        from scema_amd import capi
        from scema_amd.systems import synthetic_strains
        args, n, lens, rate, DT = self.args, self.n, self.lens, self.rate, self.DT
        strains = synthetic_strains(n, lens, seed=2026 + istep, scale=(5.0 if args.strain_set == "file3d" else 1.0),
                                    mode=("imbalanced" if args.strain_set == "imbalanced" else "balanced"))
        # The SURVEY 8(d) strains are all tensile: applied update after update to persistent states they would pull the
        # replica 3.5 % out of its equilibrium within the 25 updates of a driver run (5 GPa of tension, lists rebuilt 40 %
        # more often: a different workload at the end than at the start).  Odd updates therefore take the draw with the
        # opposite sign (a load/unload cycle): same magnitudes, same nts, and every update sees a replica within one
        # strain increment of the equilibrated state.
        if istep % 2 == 1 and not self.mono["on"]:
            strains = -strains
        if self.arr is None:
            sims = [capi.make_sim(q, "g0", 1, strains[q], nss=args.nss, most_recent=capi.QP_NONE, strain_rate=rate, dt=DT,
                                  force_field=args.force_field) for q in range(n)]
            arr = (capi.MDSim * n)(*sims)
            raw = np.frombuffer(arr, dtype=np.uint8).reshape(n, ctypes.sizeof(capi.MDSim))
            o_s, o_m = capi.MDSim.strain.offset, capi.MDSim.most_recent_qp_id.offset
            self.arr, self.keep = arr, sims
            self.strain, self.recent = raw[:, o_s:o_s + 48].view(np.float64), raw[:, o_m:o_m + 4].view(np.int32)
        self.strain[:, :] = strains
        # straining steps per replica (reference stmd_problem.h:222-232): nts = max(ceil(|eps|_F / rate / dt / 10) * 10, 10)
        true = np.asarray(strains, float) / np.array([lens[0], lens[1], lens[2], lens[2], lens[1], lens[0]])
        fro = np.sqrt((true[:, :3] ** 2).sum(1) + 2.0 * (true[:, 3:] ** 2).sum(1))
        self.nts = np.maximum(np.ceil(fro / rate / DT / 10.0) * 10.0, 10.0)
        self.nts_mean = float(self.nts.mean())
        self.recent[:, 0] = capi.QP_NONE if istep == 0 else np.arange(n, dtype=np.int32)
        return self.arr


def dry_run_ranks(args, rank, world, dist):
    """--dry-run-ranks N: the control plane that `--gpus N` takes, with no engine and no GPU -- ranks spawned the same way, gloo
    rendezvous, the 128-byte communicator id from rank 0 to all, the planner (scema_plan_update, the very code the engine plans with)
    over the same request vectors update after update, max-over-ranks of a clock, per-rank records gathered, one JSON line from rank 0.
    What it cannot show: RCCL itself and the MD.  `value` is null: nothing was measured."""
    import hashlib
    import torch
    from scema_amd import capi
    d, lens, DT, rate = _workload(args)
    n = args.sims
    uid = [os.urandom(capi.COMM_ID_BYTES) if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    assert len(uid[0]) == capi.COMM_ID_BYTES
    digests = [None] * world
    dist.all_gather_object(digests, hashlib.sha256(uid[0]).hexdigest())
    assert len(set(digests)) == 1, "the communicator id differs between ranks"
    req = Requests(args, n, lens, rate, DT)
    plan = capi.PlanDir()
    mine_total, moves_total, owners_digest = 0, 0, []
    t0 = time.perf_counter()
    for istep in range(args.warmup + args.steps):
        arr = req.build(istep)
        owner, pos, cap, moves = plan.update(arr, world, cost=req.nts + args.nss, commit=True)
        assert cap == int(np.bincount(owner, minlength=world).max())
        # within a rank the result positions are dense and unique: the all-gather's compaction map is well defined
        for r in range(world):
            pr = np.sort(pos[owner == r])
            assert (pr == np.arange(len(pr))).all(), (istep, r)
        if istep == 0 and args.strain_set != "imbalanced":
            assert (owner == np.arange(n) % world).all(), "a fresh balanced batch is dealt i % N (stmd_sync.h:583)"
        moves_total += len(moves)
        mine_total = int((owner == rank).sum())   # (as run_leg reports it: the last update's share)
        owners_digest.append(hashlib.sha256(owner.tobytes() + pos.tobytes()).hexdigest())
    elapsed = time.perf_counter() - t0
    plans = [None] * world
    dist.all_gather_object(plans, owners_digest)
    assert all(p == plans[0] for p in plans), "ranks computed different plans"
    t = torch.tensor([elapsed], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    mine = {"rank": rank, "sims": mine_total, "elapsed_s": elapsed, "pair_ms": 0.0, "md_steps": 0}
    allr = [None] * world
    dist.all_gather_object(allr, mine)
    out = None
    if rank == 0:
        out = {"metric": "stress_evals_per_sec", "value": None, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "dry_run": True,
               "config": {"workload": f"DRY RUN of the control plane of --gpus {world}: {n} requests per update(), planner only, no engine, no GPU",
                          "n_sims": n, "sims_on_rank0": int(allr[0]["sims"]), "state_migrations": moves_total,
                          "per_rank": allr, "plan_digest": owners_digest[-1][:16], "max_rank_elapsed_s": float(t.item())}}
    dist.barrier()
    return out


_RCCL_FAILED = []
_ABANDONED = []   # engines whose RCCL attach never came back: kept alive, and the process leaves through os._exit


def attach_comm(eng, new_engine, args, rank, world, device, dist):
    """The data path of --gpus N: RCCL inside the engine, proved before anything depends on it by ONE empty collective update (the
    agreement handshake and the stress all-gather with no request) under a watchdog.  If any rank fails or does not come back in time,
    EVERY rank moves -- together, agreed over the gloo control plane -- to the engine's host transport on a fresh engine (same protocol,
    the all-gather of a few kB on the host) and the JSON line says so in config.collective: a scaling run then still produces numbers.
    Returns (engine, description, comm_stats after the probe)."""
    import threading
    import torch
    from scema_amd import comm
    timeout_s = float(args.comm_timeout)
    why = _RCCL_FAILED[0] if _RCCL_FAILED else None   # (a later leg of the same run does not wait for the same failure again: every rank saw it)
    if args.dist_backend == "nccl" and why is None:
        uid = [None]
        if rank == 0:
            try:
                uid[0] = eng.comm_unique_id()
            except Exception as exc:
                uid[0] = None
                why = f"rank 0: {exc!r}"
        dist.broadcast_object_list(uid, src=0)   # (on the main thread: the watchdog below must never leave a gloo collective half entered)
        err, done = [None], threading.Event()
        if uid[0] is not None:
            def work():
                try:
                    torch.cuda.set_device(device)
                    eng.comm_init_rccl(uid[0], rank, world)
                    eng.strain_batch([], rank=rank, world=world)
                except BaseException as exc:
                    err[0] = repr(exc)
                done.set()
            th = threading.Thread(target=work, daemon=True)
            th.start()
            if not done.wait(timeout_s):
                err[0] = f"no answer from RCCL within {timeout_s:.0f} s"
                _ABANDONED.append(eng)
            if err[0]:
                why = f"rank {rank}: {err[0]}"
        flag = torch.tensor([0 if why else 1], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            return eng, "ncclAllGather inside scema_md_strain_batch", eng.comm_stats()
        whys = [None] * world
        dist.all_gather_object(whys, why)
        why = "; ".join(w for w in whys if w) or "unknown"
        _RCCL_FAILED.append(why)
        if rank == 0:
            print(f"[bench] RCCL attach failed ({why}): every rank moves to the host transport over gloo", file=sys.stderr, flush=True)
        if eng not in _ABANDONED:
            try:
                eng.comm_destroy()
            except Exception:
                pass
        eng = new_engine()
    comm.attach_gloo(eng, rank, world)
    eng.strain_batch([], rank=rank, world=world)
    return eng, "host transport (gloo)" + (f" -- RCCL attach failed: {why}" if why else ""), eng.comm_stats()


def run_leg(args, rank, world, device, cpu, torch, dist):
    """one workload through the engine: equilibrate (untimed), warm up, time `steps` updates; rank 0 returns the JSON record"""
    from scema_amd.systems import synthetic_strains
    reax = args.force_field == "reax"
    d, lens, DT, rate = _workload(args)
    n = args.sims
    from scema_amd import capi
    # ablation knobs for kernel experiments only (never set in a reported run)
    extra = {k[12:].lower(): (int(float(v)) if float(v).is_integer() else float(v)) for k, v in os.environ.items() if k.startswith("SCEMA_BENCH_")}
    extra["kspace_style"] = 1 if args.kspace == "pppm" else 0
    eng = capi.Engine(capi.default_params(**dict(dict(device=device, profile=1), **extra)))
    collective, comm0 = None, {"allgathers": 0, "handshakes": 0, "migrations": 0}
    if world > 1:
        eng, collective, comm0 = attach_comm(eng, lambda: capi.Engine(capi.default_params(**dict(dict(device=device, profile=1), **extra))), args, rank, world, device, dist)

    # ---- equilibrated replica (outside the timed region): rank 0 runs it, every rank registers the same state ----
    if reax:
        # atom_style charge, types H C N O as pair_coeff names them; ffield.reax.2 is the reference's own parameter file
        # (lammps_scripts_reax/ffield.reax.2; tests/golden holds it as a data fixture, the reference tree is not on the GPU box)
        sym = ["C" if d["mass"][t] > 5 else "H" for t in d["type"]]
        m = np.array([12.011 if c == "C" else 1.008 for c in sym])
        v0 = np.random.default_rng(3).standard_normal((len(sym), 3)) * np.sqrt(0.0019872067 * 300.0 / (m[:, None] * 48.88821291 ** 2))
        v0 -= (m[:, None] * v0).sum(0) / m.sum()
        d = capi.reax_system(sym, d["x"], d["box"], v=v0)
        eng.reax_configure(args.ffield, qeq_tol=1e-6)
    eng.register_replica("g0", 1, d)
    if args.equil_steps > 0:
        state = [None]
        if rank == 0 and args.equil_cache and os.path.exists(args.equil_cache):
            z = np.load(args.equil_cache)
            state[0] = (z["box"], z["x"], z["v"])
        elif rank == 0:
            eng.set_state(EQ_QP, "g0", 1, d["box"], d["x"], d["v"])
            eng.debug_run("g0", 1, args.equil_steps, DT, 300.0, qp=EQ_QP, nvt=True, use_shake=not reax)   # reax_configure selected the ReaxFF stage
            state[0] = eng.get_state(EQ_QP, "g0", 1)
            if args.equil_cache:
                np.savez(args.equil_cache, box=state[0][0], x=state[0][1], v=state[0][2])
        if world > 1:
            dist.broadcast_object_list(state, src=0)
        box, x, v = state[0]
        d = dict(d, box=box, x=x, v=v)
        eng.register_replica("g0", 1, d)   # drops the scratch state with the old registration
        lens = d["box"][3:6] - d["box"][:3]
    checksum = 0.0

    req = Requests(args, n, lens, rate, DT)
    mono = req.mono
    requests = req.build

    def update(istep):
        nonlocal checksum
        # collective when a communicator is attached: states that change GPU travel first, ONE all-gather returns every stress
        arr = eng.strain_batch(requests(istep), rank=rank, world=world)
        assert all(a.stress_updated for a in arr)
        checksum = float(sum(a.stress[2] for a in arr))
        return arr

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(args.warmup):
        update(w)
    eng.profile(reset=True)
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        update(args.warmup + k)
    fence()
    elapsed = time.perf_counter() - t0
    own_elapsed = elapsed
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = eng.profile()
    comm = {k: v - comm0.get(k, 0) for k, v in eng.comm_stats().items()}   # (without attach_comm's probe; collectives and checksum of warm-up + timed region: what follows is measurement aid)
    checksum_timed = checksum
    # The roofline kernel WITH THE CHIP TO ITSELF: two more updates after the timed region with the batch issued as one sequence of launches on
    # one stream (the timed region runs it as two half batches whose launches overlap each other and the other kernels of both halves: a
    # per-launch time there is not a chip-exclusive time).  The engine's settings are put back exactly as they were found.
    found = eng.concurrency()
    if reax:
        eng.reax_concurrency(0, 0)
    else:
        eng.batch_split(0)
    eng.profile(reset=True)
    for k in range(2):
        update(args.warmup + args.steps + k)
    fence()
    prof_alone = eng.profile()
    eng.reax_concurrency(found["reax_halves"], found["reax_overlap"])
    eng.batch_split(found["split"])
    assert eng.concurrency() == found
    owner, _, cap = eng.last_plan(n)
    nts_mean = req.nts_mean
    rstat = eng.reax_stats() if reax else None

    # The SURVEY 8(d) strain set AS WRITTEN (every draw tensile, update after update) next to the load/unload cycle of the
    # headline: a few more updates from the state the timed loop left, timed the same way, never part of `value`.
    mono_rate = None
    if args.monotonic_updates > 0 and not args.monotonic and args.strain_set == "balanced":
        mono["on"] = True
        fence()
        tm0 = time.perf_counter()
        for k in range(args.monotonic_updates):
            update(args.warmup + args.steps + 2 + k)
        fence()
        tm = time.perf_counter() - tm0
        if world > 1:
            t = torch.tensor([tm], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            tm = float(t.item())
        mono_rate = n * args.monotonic_updates / tm

    # The 8-GPU operating point on THIS GPU (SCALE is skipped on pools without an 8-GPU node): the share rank 0 of 8
    # gets of the same requests -- quadrature points i % 8 == 0 (stmd_sync.h:583), 72 of 576 -- continued from the states the loop left,
    # one warm-up and three timed updates; never part of `value`.  projected_8gpu_x = 8 x that rate / value (the stress all-gather of 27 kB
    # and the 16-byte handshake are microseconds against 170 ms of MD).
    share8 = None
    if world == 1 and n >= 64 and n % 8 == 0 and args.share8_updates > 0 and not args.monotonic:
        import ctypes
        mono["on"] = False
        idx = np.arange(0, n, 8)
        base = args.warmup + args.steps + 2 + (args.monotonic_updates if mono_rate else 0)
        def share_update(k):
            full = requests(base + k)
            sub = (capi.MDSim * len(idx))()
            for j, i in enumerate(idx):
                ctypes.memmove(ctypes.byref(sub[j]), ctypes.byref(full[int(i)]), ctypes.sizeof(capi.MDSim))
            out8 = eng.strain_batch(sub, rank=0, world=1)
            assert all(a.stress_updated for a in out8)
        share_update(0)
        fence()
        ts0 = time.perf_counter()
        for k in range(args.share8_updates):
            share_update(1 + k)
        fence()
        ts = time.perf_counter() - ts0
        share8 = {"sims": int(len(idx)), "updates": args.share8_updates, "evals_per_s": len(idx) * args.share8_updates / ts, "ms_per_update": 1e3 * ts / args.share8_updates}
    # calibration of the box (the boxes of a pool differ by 4-6 % under this FP64 load): its FP64 FMA ceiling, measured now (0.2 s)
    box_tflops = None
    if rank == 0:
        try:
            box_tflops = capi.box_fp64_tflops(device)
        except Exception as exc:   # (an A/B library of an older tree has no such entry)
            print("bench.py: no box calibration:", exc, file=sys.stderr)

    # N > 1: what every rank did in the timed loop (its own clock), so that a scaling line explains itself
    per_rank_stats = None
    if world > 1:
        mine = {"rank": rank, "sims": int((owner == rank).sum()), "elapsed_s": own_elapsed, "pair_ms": prof["pair_ms"], "md_steps": prof["md_steps"]}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        per_rank_stats = [dict(r, evals_per_s=r["sims"] * args.steps / r["elapsed_s"] if r["elapsed_s"] > 0 else 0.0) for r in allr]
    out = None
    if rank == 0:
        natoms = int(d["natoms"])
        per_rank = int(cap)
        # PMC figures cannot be collected inside a timed run: tools/pmc_pair.sh measures them per replica-step on the same
        # kernel (separate rocprofv3 --pmc passes; gfx950 x2 correction on FETCH_SIZE) and commits profiles/pair_pmc.json
        pmc, pmc_stale = None, None
        ppath = os.path.join(ROOT, "profiles", "pair_pmc.json")
        if os.path.exists(ppath) and natoms == 10368:
            pmc = json.load(open(ppath))
            # the counters belong to ONE build of the kernel: the file records the git blob hashes of its sources, and a tree whose
            # sources differ gets no instruction-issue figure (null + this note) instead of an old count under a new time
            have = kernel_source_hashes()
            if pmc.get("kernel_sources") != have:
                pmc_stale = {"recorded": pmc.get("kernel_sources"), "tree": have}
                print("bench.py: profiles/pair_pmc.json was taken on other sources of k_pair (re-run tools/pmc_pair.sh): frac_valu_issue and traffic are null",
                      file=sys.stderr)
                pmc = None
        value = n * args.steps / elapsed
        # ---- the roofline record of k_pair (schema 2, round 5; every figure reproduces from a table under profiles/) ----
        # The kernel is bound by vector-instruction ISSUE, not by HBM: `bound`, `achieved`, `peak`, `frac` describe THAT
        # bound, measured with the batch whole (the chip to itself); the byte-based figures of SURVEY 8(d) stand beside them as frac_hbm*.
        #   bytes: N*(4*nbar + 56) + 48 per replica and launch; `stored` = every pair once = half the full per-atom list inside
        #   cutoff + the reference's skin (counted by the exact list build of each run)
        def bytes_of(pf):
            full = pf["pair_alg_bytes"]
            fixed = pf["pair_sims"] * (56.0 * natoms + 48.0)
            return 0.5 * (full - fixed) + fixed, full
        stored, full = bytes_of(prof)
        pair_s = prof["pair_ms"] * 1e-3
        launches = max(prof["pair_launches"], 1)
        union_s = prof.get("pair_union_ms", 0.0) * 1e-3 or pair_s     # time with at least one timed launch running (= the sum when nothing overlaps)
        in_flight = pair_s / union_s if union_s > 0 else 1.0
        w_stored, w_full = bytes_of(prof_alone)
        w_s = prof_alone["pair_ms"] * 1e-3
        w_launches = max(prof_alone["pair_launches"], 1)
        w_avg = w_s / w_launches
        w_sims = prof_alone["pair_sims"] / w_launches
        nsimd, clk, cyc = 256 * 4, 2.4e9, 4.0
        peak_issue = nsimd * clk / cyc / 1e9                         # G wave-instructions/s the chip can issue at 4 cycles each
        insts_per_sim_step = pmc["valu_insts_per_sim_step"] if pmc else None
        pairs_per_sim_step = 0.5 * (w_full - prof_alone["pair_sims"] * (56.0 * natoms + 48.0)) / 4.0 / max(prof_alone["pair_sims"], 1)   # listed pairs (inside cutoff + skin)
        frac_issue = (insts_per_sim_step * w_sims / w_avg / 1e9 / peak_issue) if insts_per_sim_step and w_avg > 0 else None
        # schema 3 (round 6): the top-level achieved / peak / frac are the SURVEY 8(d) byte figures against 8 TB/s -- the yardstick
        # north_star names --; the bound that actually binds the kernel (vector-instruction issue) stands beside them as frac_valu_issue
        roof = {"schema": 3, "bound": "hbm", "unit": "GB/s", "peak": 8000.0,
                "achieved": w_stored / w_s / 1e9 if w_s > 0 else None,
                "frac": w_stored / w_s / 1e9 / 8000.0 if w_s > 0 else None,
                "frac_of": "hbm: SURVEY 8(d) bytes of the list the kernel stores (each pair once) / chip-exclusive launch time / 8 TB/s",
                "binding_bound": "fp64_valu_issue",
                "frac_valu_issue": frac_issue,
                "valu_issue_achieved_ginstr_s": (insts_per_sim_step * w_sims / w_avg / 1e9) if insts_per_sim_step and w_avg > 0 else None,
                "valu_issue_peak_ginstr_s": peak_issue,
                "frac_hbm": w_stored / w_s / 1e9 / 8000.0 if w_s > 0 else None,
                "frac_hbm_full_list": w_full / w_s / 1e9 / 8000.0 if w_s > 0 else None,
                "whole_avg_launch_ms": 1e3 * w_avg, "whole_launches": prof_alone["pair_launches"], "whole_sims_per_launch": w_sims,
                "whole_alg_bytes_per_launch": w_stored / w_launches, "whole_hbm_gbps": w_stored / w_s / 1e9 if w_s > 0 else None,
                # 55 flop per pair inside the LJ cutoff (4.876 M of the ~7.7 M listed pairs of a PE-10k replica-step) over the FP64 vector peak
                "useful_flop_frac": (55.0 * 4.876e6 / 7.725e6 * pairs_per_sim_step * w_sims / w_avg / 78.6e12) if w_avg > 0 else None,
                "traffic": None, "traffic_source": None, "pmc_stale": pmc_stale,
                "timed_avg_launch_ms": 1e3 * pair_s / launches, "timed_launches": prof["pair_launches"], "timed_sims_per_launch": prof["pair_sims"] / launches,
                "avg_launch_ms": 1e3 * pair_s / launches, "launches": prof["pair_launches"], "sims_per_launch": prof["pair_sims"] / launches,   # (names of schema 1: the timed region)
                "timed_launches_in_flight": in_flight,
                "frac_hbm_timed_union": stored / union_s / 1e9 / 8000.0 if union_s > 0 else None,
                "frac_hbm_timed_per_launch": stored / pair_s / 1e9 / 8000.0 if pair_s > 0 else None,
                "rank0_pair_share_of_wall": union_s / elapsed,
                "kernel": "k_pair (lj/cut/coul/long force+virial; every pair once, 4-atom cluster rows, LDS reaction-force tiles)",
                "accounting": "achieved = SURVEY 8(d) algorithmic bytes of a launch -- sum over its replicas of N (4 nbar_stored + 56) + 48, nbar_stored = listed pairs "
                              "per atom with each pair stored once, counted by the exact list build of every run -- / whole_avg_launch_ms; peak 8 TB/s; frac = "
                              "achieved / peak (= frac_hbm).  The kernel is NOT bound by HBM but by vector-instruction issue (DESIGN.md 4, 5.3): frac_valu_issue = "
                              "SQ_INSTS_VALU of a launch (profiles/pair_pmc.json, per replica-step x replicas per launch; null when that file was taken on other "
                              "sources of the kernel) / whole_avg_launch_ms / (1 024 SIMDs x 2.4 GHz / 4 cycles per wave instruction).  whole_* = two updates after "
                              "the timed region with the batch as ONE sequence of launches (scema_md_batch_split(0)): HIP-event time per launch with the chip to "
                              "itself -- what a rocprofv3 kernel table of a SCEMA_MD_SPLIT=0 run shows.  timed_* = the timed region, where the batch runs as two "
                              "half batches on two streams: launches overlap, so a per-launch event interval is not a chip-exclusive time; frac_hbm_timed_union = "
                              "bytes / time with at least one pair launch running"}
        if pmc:
            roof["traffic"] = pmc["hbm_bytes_per_sim_step_corrected"] * w_sims
            roof["traffic_source"] = pmc.get("source", "profiles/pair_pmc.json")
            roof["valu_insts_per_launch"] = insts_per_sim_step * w_sims
        workload = (f"{n} x PE-{natoms} OPLS replicas per update(), {nts_mean:.0f}+{args.nss} MD steps each "
                    "(dt 2 fs, 300 K, lj/cut/coul/long 12/9 + " + ("PPPM" if args.kspace == "pppm" else "Ewald") + " 1e-4 + SHAKE + NVT), persistent per-QP state, "
                    f"replica equilibrated for {args.equil_steps} steps before the timed region")
        if reax:
            # k_rx_qeq_sweep, the HBM-bound kernel of this path: per launch it reads, for every replica that still
            # iterates, each stored matrix entry once (8 B value + a 2-byte column index; 4 bytes for replicas beyond 65 536 atoms; only the
            # entries inside the taper radius are stored) and per row 84 B (row length 4, own preconditioned residual
            # pair 16, search direction and product pairs read + written 64); the gathered pairs of the columns are cache traffic by design
            # and not counted.  Entries and rows are counted on the device per sweep taken part in.
            def sweep_of(pf):
                t = pf["rx_sweep_ms"] * 1e-3
                by = (8.0 + pf["rx_sweep_col_bytes"]) * pf["rx_sweep_entries"] + 84.0 * pf["rx_sweep_rows"]
                return t, by, max(pf["rx_sweep_launches"], 1)
            sw_s, sw_bytes, sw_n = sweep_of(prof)
            sw_union = prof.get("rx_sweep_union_ms", 0.0) * 1e-3 or sw_s
            a_s, a_bytes, a_n = sweep_of(prof_alone)
            pmc = None
            ppath = os.path.join(ROOT, "profiles", "reax_pmc.json")
            if os.path.exists(ppath):
                pmc = json.load(open(ppath))
            # schema 2 (round 5): `achieved` / `frac` = the kernel with the chip to itself (two updates after the timed region with the batch as ONE
            # sequence of launches on one stream: what a rocprofv3 table of such a run shows); the timed region's figures beside them
            roof = {"schema": 2, "bound": "hbm", "unit": "GB/s", "peak": 8000.0,
                    "achieved": a_bytes / a_s / 1e9 if a_s > 0 else None, "frac": a_bytes / a_s / 1e9 / 8000.0 if a_s > 0 else None,
                    "whole_avg_launch_ms": 1e3 * a_s / a_n, "whole_launches": prof_alone["rx_sweep_launches"], "whole_alg_bytes_per_launch": a_bytes / a_n,
                    # (counters taken with 72 replicas per launch: per replica x the replicas of a whole-batch launch here, as the OPLS block scales its own)
                    "traffic": (pmc["hbm_bytes_per_sweep_corrected"] * per_rank / 72.0) if pmc and "hbm_bytes_per_sweep_corrected" in pmc else None,
                    "traffic_source": ((pmc or {}).get("command", "") + "; 72 replicas per launch, scaled by replicas per launch") if pmc else None,
                    # (the symmetric sweep is not a pure stream any more: its vector-instruction count beside its bytes, same counters, same scaling)
                    "frac_valu_issue": (pmc["SQ_INSTS_VALU"] * per_rank / 72.0 / (a_s / a_n) / 614.4e9) if pmc and "SQ_INSTS_VALU" in pmc and a_s > 0 else None,
                    "timed_avg_launch_ms": 1e3 * sw_s / sw_n, "timed_launches": prof["rx_sweep_launches"], "timed_launches_in_flight": sw_s / sw_union if sw_union > 0 else 1.0,
                    "frac_timed_union": sw_bytes / sw_union / 1e9 / 8000.0 if sw_union > 0 else None,
                    "frac_timed_per_launch": sw_bytes / sw_s / 1e9 / 8000.0 if sw_s > 0 else None,
                    "avg_launch_ms": 1e3 * sw_s / sw_n, "launches": prof["rx_sweep_launches"],   # (names of schema 1: the timed region)
                    "alone": {"frac": a_bytes / a_s / 1e9 / 8000.0 if a_s > 0 else None},       # (schema 1 kept this nested; = frac now)
                    "kernel": ("k_rx_qeq_sweep_sym (charge equilibration: y = H z for both conjugate-gradient systems; each pair of the symmetric matrix stored ONCE, "
                               "both of its products in one pass -- the transposed one through LDS atomics: half the bytes of the full rows, and no longer "
                               "a pure stream)") if prof.get("rx_sweep_symmetric") else
                              "k_rx_qeq_sweep (charge equilibration: y = H z for both conjugate-gradient systems, one pass over the stored matrix rows)",
                    "symmetric": bool(prof.get("rx_sweep_symmetric")),
                    # the stored entry: the upper 48 bits of the FP64 value (rounded to nearest, <= 2^-37 relative = 7e-12, five orders under the
                    # solver's 1e-6) + a 16-bit column in one 64-bit word -- narrower storage than the reference's FP64 matrix, said here
                    "matrix_entry_bits": 48 if int(prof["rx_sweep_col_bytes"]) == 0 else 64,
                    "frac_clock": "HIP events around every sweep launch on its stream (includes the launch-to-launch gap the event pair sees: 13-30 % above "
                                  "a rocprofv3 kernel duration at these 40-50 us launches; the conservative reading)",
                    "accounting": f"bytes = ({8 + int(prof['rx_sweep_col_bytes'])} B x stored matrix entries + 84 B x rows, summed over the replicas and sweeps that took part, counted on the "
                                  "device) / HIP-event time of the kernel's launches (launches that find every replica converged cost time and move nothing); traffic = counter "
                                  "bytes of ONE sweep over the whole batch (profiles/reax_pmc.json).  whole_* / frac: the batch as one sequence of launches on one stream "
                                  "(scema_md_reax_concurrency(0, 0)), the chip to itself.  timed_*: the timed region, where the batch runs as two half batches on two "
                                  "streams, each with its bond-order chain on a side stream -- a launch covers HALF the replicas, launches overlap, and "
                                  "frac_timed_union = bytes / time with at least one sweep running, next to the other kernels of both halves",
                    "alg_bytes_per_full_sweep": sw_bytes / max(prof["rx_sweep_rows"] / natoms, 1.0) * per_rank,   # one sweep over every replica of this rank
                    "stored_entries_per_row": prof["rx_sweep_entries"] / max(prof["rx_sweep_rows"], 1.0),
                    "rank0_sweep_share_of_wall": sw_s / elapsed,
                    "qeq_iterations_per_solve": rstat["qeq_iters"] / max(rstat["qeq_solves"], 1)}
            workload = (f"{n} x PE-{natoms} ReaxFF replicas per update() (md_force_field reax: ffield.reax.2 with H C N O, fix qeq/reax to 1e-6 every step), "
                        f"{nts_mean:.0f}+{args.nss} MD steps each (dt {DT} fs, 300 K, NVT, fix deform), persistent per-QP state, replica equilibrated for "
                        f"{args.equil_steps} steps before the timed region")
        out = {
            "metric": "stress_evals_per_sec", "value": value, "unit": "evals/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(args.steps, 1),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "replica_steps_per_s": value * (nts_mean + args.nss),
            "box_fp64_tflops": box_tflops,
            "share8_evals_per_s": share8["evals_per_s"] if share8 else None,
            "projected_8gpu_x": (8.0 * share8["evals_per_s"] / value) if share8 and value > 0 else None,
            "rccl_fallback": bool(collective and "RCCL attach failed" in collective),
            "config": {"workload": workload, "force_field": args.force_field,
                       "strain_set": args.strain_set + (" (monotonic)" if args.monotonic else " (load/unload: odd updates take the draw with the opposite sign)"),
                       "strain_set_monotonic_evals_per_s": mono_rate, "strain_set_monotonic_updates": args.monotonic_updates if mono_rate else 0,
                       "n_sims": n, "atoms_per_replica": natoms, "md_steps_per_eval": nts_mean + args.nss,
                       "sharding": "engine planner (host/sim_plan.h): fresh batch i % N, then sticky to the GPU that holds the state, levelled by MD steps",
                       "sims_on_rank0": int((owner == 0).sum()), "collective": collective,
                       "allgathers": comm["allgathers"], "handshakes": comm["handshakes"], "state_migrations": comm["migrations"],
                       "stress_zz_checksum_Pa": checksum_timed,
                       "share_of_8": (dict(share8, note="quadrature points i % 8 == 0 of the same requests (rank 0's share on 8 GPUs, stmd_sync.h:583), continued from the "
                                                        "states of the timed loop on this one GPU; outside `value`") if share8 else None),
                       "list_skin_A": prof.get("list_skin_mean", 0.0),
                       "steps_per_list_rebuild": prof["md_steps"] / max(prof["neigh_builds"], 1)},
            "roofline": roof,
        }
        if cpu is not None:
            out["cpu_baseline"] = dict(cpu, cores_rule=cpu.get("cores_rule") or CORES_RULE["rule"])
        if world > 1:
            out["config"]["per_rank"] = per_rank_stats
        out["config"]["env_overrides"] = env_overrides()
    if world > 1:
        dist.barrier()
    eng.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--sims", type=int, default=576, help="quadrature-point replicas per update()")
    ap.add_argument("--nss", type=int, default=100)
    ap.add_argument("--cells", type=int, nargs=3, default=[6, 9, 16], help="PE supercell (6 9 16 = PE-10k)")
    ap.add_argument("--strain-set", default="balanced", choices=["balanced", "file3d", "imbalanced"],
                    help="balanced: nts=10 for every replica (default, SURVEY 8d); file3d: x5 strains at rate 2e-4 (nts=30); "
                         "imbalanced: eps_zz log-uniform in [1e-3,2e-2] (nts 10..100)")
    ap.add_argument("--equil-steps", type=int, default=2000, help="NVT+SHAKE steps that equilibrate the synthetic crystal before anything is timed")
    ap.add_argument("--equil-cache", default=None, help="npz file: load the equilibrated state from it if it exists, else write it (profiling runs: "
                    "keeps the 2 000 single-replica steps out of a PMC pass)")
    ap.add_argument("--monotonic", action="store_true", help="apply the tensile strain draws update after update (no unloading on odd updates)")
    ap.add_argument("--kspace", default="pppm", choices=["pppm", "ewald"], help="reciprocal part: PPPM (order 5, ik, hipFFT) as the reference's "
                    "`kspace_style pppm 0.0001` asks for (default, the reported configuration) or the plain Ewald sum at the same accuracy")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-field", default="opls", choices=["opls", "reax"],
                    help="opls: the headline workload (576 x PE-10k); reax: BASELINE config 5, 72 x PE-1620 ReaxFF replicas (--sims / --cells / --nss default to that)")
    ap.add_argument("--monotonic-updates", type=int, default=4, help="after the timed loop, this many all-tensile updates (the SURVEY 8(d) set as written) are "
                    "timed as well and reported as config.strain_set_monotonic_evals_per_s (0: skip; never part of `value`)")
    ap.add_argument("--share8-updates", type=int, default=3, help="after the timed loop (one GPU, batches of 64 and more): this many updates of the share rank 0 of 8 "
                    "would run (requests i %% 8 == 0) -> share8_evals_per_s, projected_8gpu_x (0: skip; never part of `value`)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI inside the engine (default); gloo = the engine's host transport over gloo (tests)")
    ap.add_argument("--dry-run-ranks", type=int, default=0, help="N: run the control plane of --gpus N (rank spawn, rendezvous, id exchange, planner, "
                    "per-rank JSON assembly) with no engine and no GPU; prints a line with \"dry_run\": true and no value")
    ap.add_argument("--comm-timeout", type=float, default=300.0,
                    help="seconds the RCCL attach + its one-update probe may take before every rank moves to the host transport")
    ap.add_argument("--share-gpus", action="store_true", help="tests: let several ranks share a GPU (needs --dist-backend gloo)")
    ap.add_argument("--reax-leg", default="auto", choices=["auto", "on", "off"],
                    help="after the OPLS loop, outside `value`: a short ReaxFF replica-set leg (BASELINE config 5: 72 x PE-1620, 2 warm-up + 4 timed updates) "
                         "reported as config.reax in the same JSON line.  auto: for the default workload (576 x PE-10k) on one GPU")
    ap.add_argument("--ffield", default=os.path.join(ROOT, "examples", "ffield.reax.2"),
                    help="ReaxFF parameter file (the reference's lammps_scripts_reax/ffield.reax.2; the tree keeps a copy as data under examples/)")
    args = ap.parse_args()
    if args.force_field == "reax":   # the replica set of BASELINE config 5, unless the command line says otherwise
        given = " ".join(sys.argv[1:])
        if "--sims" not in given: args.sims = 72
        if "--cells" not in given: args.cells = [3, 5, 9]
        if "--nss" not in given: args.nss = 20
        if "--equil-steps" not in given: args.equil_steps = 200

    if args.dry_run_ranks:
        args.gpus = args.dry_run_ranks
    if "RANK" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")

    if args.dry_run_ranks:
        import torch.distributed as dist
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
        else:
            raise SystemExit("bench.py: --dry-run-ranks needs N > 1")
        out = dry_run_ranks(args, rank, world, dist)
        if rank == 0:
            print(json.dumps(out), flush=True)
        dist.destroy_process_group()
        return

    d, lens, DT, rate = _workload(args)
    from scema_amd.systems import synthetic_strains
    reax = args.force_field == "reax"
    want_reax_leg = (not reax) and world == 1 and (args.reax_leg == "on" or (args.reax_leg == "auto" and args.sims == 576 and list(args.cells) == [6, 9, 16]))
    rargs = _reax_leg_args(args) if want_reax_leg else None

    # CPU baselines first: their worker processes start while nothing in this process has touched the GPU
    cpu = cpu_rx = None
    if world == 1 and not args.no_cpu_baseline:
        ncpu = _host_cores(0.4)
        if reax:
            cpu = cpu_baseline_reax(tuple(args.cells), synthetic_strains(max(ncpu, 8), lens, seed=2026), args.nss, DT, rate)
        else:
            cpu = cpu_baseline(tuple(args.cells), synthetic_strains(max(ncpu, 32), lens, seed=2026), args.nss, 1 if args.kspace == "pppm" else 0)
        cpu["cores_rule"] = CORES_RULE["rule"]
        if rargs is not None:   # a few workers only: every one holds a reverse-mode graph, and the leg must stay short
            _, rl, rdt, rrate = _workload(rargs)
            cpu_rx = cpu_baseline_reax(tuple(rargs.cells), synthetic_strains(8, rl, seed=2026), rargs.nss, rdt, rrate, cap=4)
            cpu_rx["cores_rule"] = CORES_RULE["rule"]

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    ndev = torch.cuda.device_count()
    if world > ndev and not args.share_gpus:
        raise SystemExit(f"bench.py: {world} ranks but {ndev} GPU(s)")
    device = local_rank % ndev
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")   # control plane only (id exchange, barrier, max of the timings); the data path is RCCL

    out = run_leg(args, rank, world, device, cpu, torch, dist)
    if rargs is not None:
        try:
            r = run_leg(rargs, rank, world, device, cpu_rx, torch, dist)
        except Exception as exc:   # the headline line must not depend on the second leg
            r = None
            if rank == 0:
                out["config"]["reax"] = {"error": repr(exc)}
        if rank == 0 and r is not None:
            c = r["config"]
            out["config"]["reax"] = {
                "workload": c["workload"], "evals_per_s": r["value"], "ms_per_update": r["ms_per_step"], "steps": r["steps"], "warmup": r["warmup"],
                "md_steps_per_eval": c["md_steps_per_eval"], "replica_steps_per_s": r["value"] * c["md_steps_per_eval"],
                "atoms_per_replica": c["atoms_per_replica"], "n_sims": c["n_sims"], "stress_zz_checksum_Pa": c["stress_zz_checksum_Pa"],
                "roofline": r["roofline"], "cpu_baseline": r.get("cpu_baseline"),
                "note": "BASELINE config 5 as a second leg of this run, after the OPLS loop and outside `value`; an evaluation here is "
                        f"{c['md_steps_per_eval']:.0f} MD steps of 0.25 fs on 1 620 atoms (the OPLS line's: 110 steps of 2 fs on 10 368 atoms): compare replica_steps_per_s, not evals_per_s"}
    if rank == 0:
        rx = out["config"].get("reax") or {}
        if "evals_per_s" in rx:   # the second leg once more in flat keys (a parser that keeps one level of nesting keeps these)
            out["reax_evals_per_s"] = rx["evals_per_s"]
            out["reax_replica_steps_per_s"] = rx["replica_steps_per_s"]
            out["reax_frac"] = rx["roofline"]["frac"]
            out["reax_frac_timed_union"] = rx["roofline"]["frac_timed_union"]
            out["reax_whole_avg_launch_ms"] = rx["roofline"]["whole_avg_launch_ms"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if _ABANDONED:   # a thread is still inside RCCL: no interpreter shutdown through it
        sys.stdout.flush(); sys.stderr.flush()
        # the line above says "rccl_fallback": true; a caller that ASKED for RCCL on the command line also gets a distinct exit code,
        # the default invocation (the driver's) keeps 0: its number is valid, measured over the host transport and labelled so
        asked = any(a == "--dist-backend" or a.startswith("--dist-backend=") for a in sys.argv[1:]) and args.dist_backend == "nccl"
        os._exit(3 if asked else 0)


if __name__ == "__main__":
    main()
