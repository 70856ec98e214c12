"""The C ABI from the reference's own language: examples/scema_harness.cpp (a stand-in for the call site
dealammps.cc:455) is compiled with g++ against include/ and libscema_md.so and run in the reference's
Hooke test mode; its stresses equal the oracle's L3 arithmetic."""
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_caller_links_and_runs(tmp_path):
    import __graft_entry__ as g
    g.build()
    from scema_amd import stmd
    from oracle import pyoracle as po
    exe = str(tmp_path / "harness")
    subprocess.check_call(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "scema_harness.cpp"),
                           "-L" + os.path.join(ROOT, "scema_amd"), "-lscema_md", "-Wl,-rpath," + os.path.join(ROOT, "scema_amd"), "-o", exe])
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "init_sic_1_stiff.json")))
    C = np.array(gold["stiff_file_order"])
    nin = str(tmp_path / "nin"); nout = str(tmp_path / "nout"); mout = str(tmp_path / "mout")
    os.makedirs(nout); os.makedirs(mout)
    stmd.write_nanoscale_input(nin, "g0", 1, init_length=[40.0, 41.0, 42.0], init_stress_raw=np.zeros(6), stiff_file_order=C, nsheets=0)
    out = subprocess.run([exe, nin, nout, mout, "g0", "1", "1", "3"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = [l.split() for l in out.stdout.strip().split("\n") if l.startswith("qp")]
    assert len(rows) == 3
    for q, r in enumerate(rows):
        ezz = 1.0e-3 + 1.0e-4 * q
        eps = np.array([-0.3 * ezz, -0.3 * ezz, ezz, 1.0e-5 * q, 0.0, -2.0e-5])
        exp = po.hooke(C, eps)
        got = np.array([float(v) for v in r[3:9]])
        assert np.allclose(got, exp, rtol=1e-13, atol=1e-6)


def test_cpp_cluster_caller_links_and_runs(tmp_path):
    """include/scema_cluster.h from C++: spline fit, similarity lists and greedy cover on the host."""
    import __graft_entry__ as g
    g.build()
    exe = str(tmp_path / "cluster_harness")
    subprocess.check_call(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "scema_cluster_harness.cpp"),
                           "-L" + os.path.join(ROOT, "scema_amd"), "-lscema_md", "-Wl,-rpath," + os.path.join(ROOT, "scema_amd"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    m = {int(l.split()[1]): int(l.split()[3]) for l in out.stdout.strip().split("\n")}
    # quadrature points 0 and 2 share a history: the one that entered the graph last (2) runs the MD for both
    assert m == {0: 2, 1: 1, 2: 2, 3: 3, 4: 4, 5: 5}


def test_cpp_time_loop_over_the_cuboid_mesh(tmp_path):
    """examples/scema_hmm_harness.cpp: the reference's do_timestep loop (dealammps.cc:417-474) in C++ on the C ABI -- continuum
    stand-in + STMDSync in the Hooke test mode, 3x3x8 cells = 576 quadrature points, 10 continuum steps (BASELINE config 3)."""
    import __graft_entry__ as g
    g.build()
    from scema_amd import stmd
    exe = str(tmp_path / "hmm")
    subprocess.check_call(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "scema_hmm_harness.cpp"),
                           "-L" + os.path.join(ROOT, "scema_amd"), "-lscema_md", "-Wl,-rpath," + os.path.join(ROOT, "scema_amd"), "-o", exe])
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "init_sic_1_stiff.json")))
    C = np.array(gold["stiff_file_order"])
    nin = str(tmp_path / "nin"); nout = str(tmp_path / "nout"); mout = str(tmp_path / "mout")
    os.makedirs(nout); os.makedirs(mout)
    stmd.write_nanoscale_input(nin, "g0", 1, init_length=[40.0, 41.0, 42.0], init_stress_raw=np.zeros(6), stiff_file_order=C, nsheets=0)
    out = subprocess.run([exe, nin, nout, mout, "g0", "1", "3", "3", "8", "10"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = [l.split() for l in out.stdout.strip().split("\n") if l.startswith("step")]
    assert [int(r[1]) for r in rows] == list(range(1, 11))
    nupd = [int(r[3]) for r in rows]
    assert nupd[0] >= 72 and nupd[-1] <= 576 and all(b >= a for a, b in zip(nupd, nupd[1:]))   # the loaded layer first, then the wave spreads
    smax = [float(r[5]) for r in rows]
    assert smax[0] > 0 and all(np.isfinite(smax))
