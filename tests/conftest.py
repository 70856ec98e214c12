import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` tests must not silently pass (or skip) on a box without a GPU: the run is refused there
    def no_gpu():
        import torch
        return not torch.cuda.is_available()

    gpu_items = [it for it in items if it.get_closest_marker("gpu") is not None]
    if gpu_items and "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or "") and no_gpu():
        raise pytest.UsageError("tests marked `gpu` were selected but this box has no GPU: the engine has no CPU fallback")


@pytest.fixture(scope="session", autouse=True)
def _library_is_built():
    """The C-ABI library and the oracle are compiled once per session (a no-op when they are up to date): a fresh
    checkout must not fail GPU tests with 'library not built', and nothing ever falls back to a CPU path instead."""
    import __graft_entry__ as g
    g.build()


@pytest.fixture(scope="session")
def small_pe():
    """360-atom PE crystal with jitter + tilt, used with reduced cutoffs (box >= 2*(rc+skin))."""
    from scema_amd.systems import build_pe
    d = build_pe(2, 3, 5, jitter=0.05, seed=7)
    d["box"][6:9] = [0.7, -0.4, 0.5]
    return d
