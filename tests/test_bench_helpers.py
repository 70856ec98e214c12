"""bench.py's host-side helpers (no GPU): the CPU-baseline core count, the GPU count from sysfs, and the shim that keeps the
CPU-only torch workers of the ReaxFF baseline away from the GPU's device nodes."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_cores_is_the_jobs_share_not_the_hosts_count(monkeypatch):
    import bench
    n = bench._host_cores(0.4)
    assert 1 <= n <= 16                                   # a one-GPU job of this pool gets 16 cores whatever os.cpu_count() says
    assert bench._host_cores(0.4, cap=2) <= 2
    monkeypatch.setenv("SCEMA_CPU_BASELINE_CORES", "3")
    assert bench._host_cores(1000.0) in (1, 3)            # the memory bound still applies after the override


def test_gpu_count_comes_from_sysfs_and_respects_visibility(monkeypatch):
    import bench
    n = bench.count_gpus_without_hip()
    assert n >= 0                                         # no /sys/class/kfd here: 0; on a GPU box the nodes with SIMDs
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
    assert bench.count_gpus_without_hip() <= 1


def test_the_shim_hides_the_gpu_device_nodes_and_nothing_else(tmp_path):
    import __graft_entry__ as g
    g.build()
    shim = os.path.join(ROOT, "oracle", "_build", "libnogpu_shim.so")
    assert os.path.exists(shim)
    code = ("import os, errno\n"
            "for p in ('/dev/kfd', '/dev/dri/renderD128'):\n"
            "    try:\n"
            "        os.close(os.open(p, os.O_RDWR)); print('OPENED', p)\n"
            "    except OSError as e:\n"
            "        print('blocked' if e.errno == errno.ENOENT else 'other', p)\n"
            "open(r'%s', 'w').write('x'); print(open('/dev/null').read() == '', open(r'%s').read())\n" % (tmp_path / "f", tmp_path / "f"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LD_PRELOAD=shim), capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("blocked") == 2 and "OPENED" not in r.stdout and "True x" in r.stdout


def _dry(extra, port):
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run-ranks", "8", "--steps", "2", "--warmup", "1"] + extra
    r = subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                  # ONE line, from rank 0
    return json.loads(lines[0])


def test_the_control_plane_of_gpus_8_runs_without_a_gpu():
    """everything `bench.py --gpus 8` does around the engine -- the spawn of 8 ranks, the gloo rendezvous, the id
    exchange, the planner over the request vectors of every update, the max over ranks and the per-rank JSON assembly -- at world 8,
    which no GPU test of this pool can reach (process guard: 6).  Balanced set: 72 simulations per rank, no state moves."""
    out = _dry([], 0)
    assert out["dry_run"] is True and out["value"] is None and out["n_gpus"] == 8
    pr = out["config"]["per_rank"]
    assert [r["rank"] for r in pr] == list(range(8)) and all(r["sims"] == 72 for r in pr)
    assert out["config"]["state_migrations"] == 0 and out["config"]["sims_on_rank0"] == 72


def test_the_ragged_strain_set_is_levelled_over_8_ranks_in_the_dry_run():
    """the imbalanced set (nts 10..100): the planner levels MD steps, so simulations per rank differ and states move between updates"""
    out = _dry(["--strain-set", "imbalanced"], 1)
    sims = [r["sims"] for r in out["config"]["per_rank"]]
    assert sum(sims) == 576 and max(sims) - min(sims) >= 1 and min(sims) > 40
    assert out["config"]["state_migrations"] >= 1


def test_pair_pmc_counters_belong_to_the_kernel_sources_of_this_tree():
    """VERDICT r5 / ADVICE r5: roofline.frac_valu_issue prices this run's launch time with SQ_INSTS_VALU of profiles/pair_pmc.json, which was
    counted on ONE build of k_pair.  The file records the git blob hashes of the kernel's sources; bench.py drops the figure when the tree's
    differ.  Here: the hash function is git's, and the committed counters are those of the committed sources (a change of md_pair.hip without a
    fresh tools/pmc_pair.sh run fails this test instead of silently leaving frac_valu_issue null in the driver's line)."""
    import hashlib
    import json
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    have = bench.kernel_source_hashes()
    assert set(have) == {"md_pair.hip", "md_pair_dev.h", "md_device.h", "md_types.h"}
    data = open(os.path.join(ROOT, "scema_amd", "csrc", "md_pair.hip"), "rb").read()
    assert have["md_pair.hip"] == hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()
    try:   # (where git is at hand: the same number git prints)
        out = subprocess.run(["git", "hash-object", os.path.join(ROOT, "scema_amd", "csrc", "md_pair.hip")], capture_output=True, text=True, timeout=30)
        if out.returncode == 0 and out.stdout.strip():
            assert out.stdout.strip() == have["md_pair.hip"]
    except (OSError, subprocess.TimeoutExpired):
        pass
    pmc = json.load(open(os.path.join(ROOT, "profiles", "pair_pmc.json")))
    assert pmc["kernel_sources"] == have, "profiles/pair_pmc.json was counted on other sources of k_pair: run tools/pmc_pair.sh on the GPU box and commit its pair_pmc_<tag>.json"
    assert pmc["valu_insts_per_sim_step"] > 1e6 and pmc["hbm_bytes_per_sim_step_corrected"] > 1e6
