"""GPU: bench.py's multi-rank path.  `python bench.py --gpus 2` must start its two ranks itself; on the one-GPU test box
the ranks share the device and the engine's collective runs over its host transport (gloo), since RCCL wants one device
per rank -- the RCCL variant runs where two devices exist and says so when it cannot."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--sims", "6", "--steps", "2", "--warmup", "1", "--nss", "10", "--cells", "4", "6", "12", "--equil-steps", "40", "--no-cpu-baseline"]


def _run(cmd, ok=True):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    if not ok:
        return r
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_two_rank_bench_spawns_its_ranks_and_matches_single_rank():
    one = _run([sys.executable, "bench.py", "--gpus", "1"] + COMMON)
    two = _run([sys.executable, "bench.py", "--gpus", "2", "--dist-backend", "gloo", "--share-gpus"] + COMMON)
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    c1, c2 = one["config"]["stress_zz_checksum_Pa"], two["config"]["stress_zz_checksum_Pa"]
    assert abs(c1 - c2) <= 1e-8 * abs(c1), (c1, c2)      # FP64 atomics: summation order differs from run to run
    assert two["config"]["allgathers"] == 3 and two["config"]["sims_on_rank0"] == 3     # ONE collective per update (1 warm-up + 2 timed)
    for k in ("metric", "value", "unit", "ms_per_step", "roofline", "scaling", "dtype"):
        assert k in two


def test_more_gpus_than_devices_is_an_error_not_a_smaller_run():
    import torch
    n = torch.cuda.device_count() + 1
    r = _run([sys.executable, "bench.py", "--gpus", str(n)] + COMMON, ok=False)
    assert r.returncode != 0 and "n_gpus" not in r.stdout


def test_two_rank_bench_over_rccl():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("RCCL PATH NOT EXERCISED: this box has one GPU (ncclAllGather inside scema_md_strain_batch needs one device per rank)")
    one = _run([sys.executable, "bench.py", "--gpus", "1"] + COMMON)
    two = _run([sys.executable, "bench.py", "--gpus", "2"] + COMMON)
    assert two["n_gpus"] == 2 and two["config"]["collective"].startswith("ncclAllGather")
    c1, c2 = one["config"]["stress_zz_checksum_Pa"], two["config"]["stress_zz_checksum_Pa"]
    assert abs(c1 - c2) <= 1e-8 * abs(c1), (c1, c2)
