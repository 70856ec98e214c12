"""GPU: bench.py's multi-rank path (round-robin sharding + one all-gather) with two ranks sharing the
one GPU of the test box (gloo for the collective, since RCCL wants one device per rank): the gathered
checksum must equal the single-rank run's."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_two_rank_bench_matches_single_rank():
    common = ["--sims", "6", "--steps", "1", "--warmup", "0", "--nss", "10", "--cells", "4", "6", "12", "--no-cpu-baseline"]
    one = _run([sys.executable, "bench.py", "--gpus", "1"] + common)
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29533", "bench.py", "--gpus", "2", "--dist-backend", "gloo"] + common)
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    c1, c2 = one["config"]["stress_zz_checksum_Pa"], two["config"]["stress_zz_checksum_Pa"]
    assert abs(c1 - c2) <= 1e-9 * abs(c1)
    for k in ("metric", "value", "unit", "ms_per_step", "roofline", "scaling", "dtype"):
        assert k in two
