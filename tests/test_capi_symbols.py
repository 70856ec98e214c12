"""CPU-side checks of the C-ABI library: it loads and exports every symbol include/scema_md.h declares;
without a GPU engine creation fails loudly (no fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _built():
    import __graft_entry__ as g
    g.build()


def test_library_exports_every_declared_symbol():
    _built()
    from scema_amd import capi
    from scema_amd import stmd
    hdr = open(os.path.join(ROOT, "include", "scema_md.h")).read()
    declared = set(re.findall(r"\b(scema_(?:md|plan)_[a-z_]+)\s*\(", hdr))
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    hdr2 = open(os.path.join(ROOT, "include", "scema_stmd.h")).read()
    declared2 = set(re.findall(r"\b(scema_(?:stmd|eqmd)_[a-z_]+)\s*\(", hdr2))
    assert declared2 == set(stmd.SYMBOLS), declared2 ^ set(stmd.SYMBOLS)
    from scema_amd import cluster
    hdr3 = open(os.path.join(ROOT, "include", "scema_cluster.h")).read()
    declared3 = set(re.findall(r"\b(scema_hist_[a-z_]+)\s*\(", hdr3))
    assert declared3 == set(cluster.SYMBOLS), declared3 ^ set(cluster.SYMBOLS)
    from scema_amd import fe
    hdr4 = open(os.path.join(ROOT, "include", "scema_fe.h")).read()
    declared4 = set(re.findall(r"\b(scema_fe_[a-z_]+)\s*\(", hdr4))
    assert declared4 == set(fe.SYMBOLS), declared4 ^ set(fe.SYMBOLS)
    L = capi.lib()
    for s in declared | declared2 | declared3 | declared4:
        assert hasattr(L, s), s


def test_struct_layouts_match_header_sizes():
    from scema_amd import capi
    # QP wire record is 112 bytes (scale_bridging_data.h:12-19); MDSim mirror must keep C alignment
    assert ctypes.sizeof(capi.MDSim) % 8 == 0
    assert capi.MDSim.strain.offset % 8 == 0 and capi.MDSim.stress.offset % 8 == 0


def test_no_gpu_means_loud_failure():
    _built()
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from scema_amd import capi
    with pytest.raises(capi.EngineError):
        capi.Engine()


def test_every_environment_switch_the_sources_read_is_declared():
    """md_env.h: an undeclared name passed to scema_env() aborts the process the first time that line runs -- which may be on the
    GPU box only.  Every literal name in the sources must be in the table of engine_core.cpp, and every table entry must be read."""
    import glob
    src = os.path.join(ROOT, "scema_amd", "csrc")
    table = set(re.findall(r'^\s*\{"(SCEMA_[A-Z0-9_]+)",', open(os.path.join(src, "engine", "engine_core.cpp")).read(), flags=re.M))
    used = set()
    for f in glob.glob(os.path.join(src, "**", "*"), recursive=True):
        if f.endswith((".cpp", ".hip", ".h")) and "_obj" not in f:
            used |= set(re.findall(r'scema_env\("(SCEMA_[A-Z0-9_]+)"\)', open(f).read()))
    assert used <= table, used - table
    assert table <= used, table - used
