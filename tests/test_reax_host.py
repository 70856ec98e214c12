"""The ReaxFF arithmetic of the product (scema_amd/csrc/reax/rx_core.h, the functions the HIP kernels run one lane per atom),
compiled for the host by tests/reax_host_driver.cpp and held against the oracle (oracle/reax_oracle.c): energies part by
part, forces against central differences of the oracle's energy, virial against its strain derivative, charges against
its equilibration.  No GPU needed: this is where every derivative of the reverse-mode force evaluation is pinned; the
`-m gpu` tests then compare the kernels' output with the same oracle.

ReaxFF's energy has small jumps where a bond order crosses a threshold (bond-order cutoff for the pi parts, the
valence-angle and torsion cutoffs); a central difference that straddles one is off by jump/2h, so the directional checks
use h = 1e-5 and let one direction in six miss.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import pyreax as pr
from test_oracle_reax import FFIELD, _ethane, _ethylene, _glycine_like, _water

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ELEMENTS = ["H", "C", "N", "O"]          # pair_coeff * * ffield.reax.2 H C N O  (lammps_scripts_reax/in.strain.lammps:11)
ALL = 31


@pytest.fixture(scope="module")
def ff():
    f = pr.ForceField(FFIELD)
    yield f
    f.close()


# "library": the host build as the product's host tools see it (libm).  "device_math": the same functions with the device's own log / pow /
# reciprocal / reciprocal-square-root algorithms (rx_core.h, RX_DEVICE_MATH_ON_HOST: single-precision seeds stand in for the hardware
# estimates), so that what the kernels compute is held against the oracle here as well, term by term and derivative by derivative.
@pytest.fixture(scope="module", params=["library", "device_math"])
def drv(request):
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    san = os.environ.get("SCEMA_SANITIZE") == "1"       # tools/run_asan.sh: the driver and the parameter reader under ASan + UBSan
    dm = request.param == "device_math"
    so = os.path.join(out, "libreax_host" + ("_dm" if dm else "") + ("_asan" if san else "") + ".so")
    srcs = [os.path.join(ROOT, "tests", "reax_host_driver.cpp"), os.path.join(ROOT, "scema_amd", "csrc", "host", "reax_ffield.cpp")]
    deps = srcs + [os.path.join(ROOT, "scema_amd", "csrc", "reax", f) for f in ("rx_core.h", "rx_types.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in deps):
        flags = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"] if san else ["-O2"]
        subprocess.check_call(["g++"] + flags + (["-DRX_DEVICE_MATH_ON_HOST"] if dm else []) + ["-fPIC", "-shared", "-std=c++17", "-o", so] + srcs)
    L = C.CDLL(so)
    L.rxh_create.restype = C.c_void_p
    L.rxh_create.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.c_int]
    L.rxh_destroy.argtypes = [C.c_void_p]
    L.rxh_compute.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 3 + [C.c_double, C.c_void_p, C.c_double, C.c_int] + [C.c_void_p] * 5
    arr = (C.c_char_p * len(ELEMENTS))(*[e.encode() for e in ELEMENTS])

    class Driver:
        def __init__(self, quirk):
            self.h = L.rxh_create(FFIELD.encode(), arr, len(ELEMENTS), quirk)
            assert self.h

        def __call__(self, sym, x, box, q=None, terms=ALL, rlist=10.0, tol=1e-10):
            n = len(sym)
            t = np.array([ELEMENTS.index(s) for s in sym], dtype=np.int32)
            x = np.ascontiguousarray(x, float)
            box = np.ascontiguousarray(box, float)
            f, e, w, qo, cnt = np.zeros((n, 3)), np.zeros(len(pr.PARTS)), np.zeros(6), np.zeros(n), np.zeros(3, dtype=np.int32)
            qq = None if q is None else np.ascontiguousarray(q, float)
            p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
            rc = L.rxh_compute(self.h, n, p(t), p(x), p(box), rlist, p(qq), tol, terms, p(f), p(e), p(w), p(qo), p(cnt))
            assert rc == 0, rc
            return dict(f=f, e=dict(zip(pr.PARTS, e)), w=w, q=qo, counts=cnt)

    exact, lammps = Driver(0), Driver(1)
    yield exact, lammps
    L.rxh_destroy(exact.h)
    L.rxh_destroy(lammps.h)


BIGBOX = np.array([-15.0, -15.0, -15.0, 15.0, 15.0, 15.0, 4.0, -3.0, 2.0])


def _sym(ff, t):
    return [ff.names[k] for k in t]


def _pe_cell(ff, tilt=(0.0, 0.0, 0.0), amp=0.1, seed=5):
    from scema_amd.systems import build_pe
    d = build_pe(3, 5, 8)                        # 1440 atoms, 22.2 x 24.65 x 20.27 A: the smallest cell the oracle's minimum image takes
    box = d["box"].copy()
    box[6:9] = tilt
    x = d["x"].copy()
    ly, lz = box[4] - box[1], box[5] - box[2]
    x[:, 0] += tilt[0] * d["x"][:, 1] / ly + tilt[1] * d["x"][:, 2] / lz
    x[:, 1] += tilt[2] * d["x"][:, 2] / lz
    x += amp * np.random.default_rng(seed).standard_normal(x.shape)
    sym = ["C" if d["mass"][t] > 5 else "H" for t in d["type"]]
    return sym, x, box


def _mixture(seed=11, L=21.0):
    """water, ammonia, methane and formaldehyde on a jittered lattice: every element, hydrogen bonds, lone pairs"""
    rng = np.random.default_rng(seed)
    a = 1.09 / np.sqrt(3)
    mols = {
        "water": (["O", "H", "H"], _water(np.zeros(3))),
        "ammonia": (["N", "H", "H", "H"], np.array([[0, 0, 0.1], [0.94, 0, -0.27], [-0.47, 0.81, -0.27], [-0.47, -0.81, -0.27]])),
        "methane": (["C", "H", "H", "H", "H"], np.array([[0, 0, 0], [a, a, a], [-a, -a, a], [-a, a, -a], [a, -a, -a]])),
        "formaldehyde": (["C", "O", "H", "H"], np.array([[0, 0, 0], [1.21, 0, 0], [-0.59, 0.94, 0], [-0.59, -0.94, 0]])),
    }
    names = list(mols)
    sym, xs = [], []
    nside = 6
    for i in range(nside):
        for j in range(nside):
            for k in range(nside):
                s, x = mols[names[(i + 2 * j + 3 * k) % 4]]
                Q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
                xs.append(x @ Q.T + (np.array([i, j, k]) + 0.5) * L / nside + 0.15 * rng.standard_normal(3))
                sym += s
    return sym, np.vstack(xs), np.array([0, 0, 0, L, L, L, 0.0, 0.0, 0.0])


def _check_dirs(ff, t, x, box, q, f, ndir=6, h=1e-5, seed=1):
    rng = np.random.default_rng(seed)
    good = 0
    worst = 0.0
    for _ in range(ndir):
        d = rng.standard_normal(x.shape)
        d /= np.linalg.norm(d)
        de = (ff.energy(t, x + h * d, box=box, q=q)[0] - ff.energy(t, x - h * d, box=box, q=q)[0]) / (2 * h)
        an = -(f * d).sum()
        err = abs(de - an) / (1.0 + abs(an))
        worst = max(worst, err)
        good += err < 2e-5
    return good, worst


def test_parser_and_energies_match_the_oracle(ff, drv):
    exact, lammps = drv
    for t, x in (_glycine_like(ff), _ethane(ff, 0.4), _ethylene(ff, 0.5)):
        q, _ = ff.qeq(t, x, box=BIGBOX, tol=1e-10, maxiter=500)
        r = exact(_sym(ff, t), x, BIGBOX)
        assert np.abs(r["q"] - q).max() < 1e-8
        r = exact(_sym(ff, t), x, BIGBOX, q=q)
        _, po = ff.energy(t, x, box=BIGBOX, q=q)
        for k in pr.PARTS:
            assert abs(r["e"][k] - po[k]) < 1e-9 * max(1.0, abs(po[k])), k
        assert lammps(_sym(ff, t), x, BIGBOX, q=q)["e"] == r["e"]      # the switch touches derivatives only


@pytest.mark.parametrize("terms,parts", [(1, ("bond", "lp", "over", "under")), (2, ("angle", "pen", "coa")), (4, ("tors", "conj")), (8, ("hb",)),
                                         (16, ("vdw", "coul", "pol")), (ALL, tuple(pr.PARTS))])
def test_forces_of_each_pass_against_central_differences(ff, drv, terms, parts):
    exact, _ = drv
    t, x = _glycine_like(ff)
    sym = _sym(ff, t)
    q, _ = ff.qeq(t, x, box=BIGBOX, tol=1e-10, maxiter=500)
    r = exact(sym, x, BIGBOX, q=q, terms=terms)
    h = 1e-5
    fd = np.zeros_like(x)
    for i in range(len(x)):
        for c in range(3):
            xp, xm = x.copy(), x.copy()
            xp[i, c] += h
            xm[i, c] -= h
            pp, pm = ff.energy(t, xp, box=BIGBOX, q=q)[1], ff.energy(t, xm, box=BIGBOX, q=q)[1]
            fd[i, c] = -sum(pp[k] - pm[k] for k in parts) / (2 * h)
    assert np.abs(fd).max() > 0.5
    assert np.abs(r["f"] - fd).max() < 2e-6 * max(1.0, np.abs(fd).max())
    # virial of this pass = sum r (x) f for an isolated molecule
    w = np.einsum("ia,ib->ab", x, r["f"])
    ref = np.array([w[0, 0], w[1, 1], w[2, 2], w[0, 1], w[0, 2], w[1, 2]])
    assert np.abs(r["w"] - ref).max() < 1e-8 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("tilt", [(0.0, 0.0, 0.0), (3.0, -2.0, 1.5)])
def test_condensed_polyethylene_cell(ff, drv, tilt):
    exact, lammps = drv
    sym, x, box = _pe_cell(ff, tilt=tilt, amp=0.08 if tilt[0] == 0 else 0.15)
    t = ff.types(sym)
    q, _ = ff.qeq(t, x, box=box, tol=1e-10, maxiter=500)
    r = exact(sym, x, box)
    assert np.abs(r["q"] - q).max() < 1e-8 and abs(r["q"].sum()) < 1e-9
    r = exact(sym, x, box, q=q)
    _, po = ff.energy(t, x, box=box, q=q)
    for k in pr.PARTS:
        assert abs(r["e"][k] - po[k]) < 1e-9 * max(1.0, abs(po[k])), k
    assert abs(po["tors"]) > 100 and abs(po["over"]) > 100 and abs(po["angle"]) > 100
    good, worst = _check_dirs(ff, t, x, box, q, r["f"])
    assert good >= 5, worst
    assert np.abs(r["f"].sum(0)).max() < 1e-7 * np.abs(r["f"]).max()
    # virial against the strain derivative of the oracle's energy
    w = np.zeros(6)
    pr.lib().rxo_forces_fd(ff.h, len(t), pr._p(np.ascontiguousarray(t, dtype=np.int32)), pr._p(np.ascontiguousarray(x)), pr._p(box), pr._p(q), 1e-5, None, pr._p(w))
    assert np.abs(r["w"] - w).max() < 1e-6 * np.abs(w).max()
    # the switch that leaves out d(SBO)/d(Delta) for atoms with vlpex >= 0 (a reading of USER-REAXC that cannot be checked
    # here; off by default): it changes the angle pass only
    rl = lammps(sym, x, box, q=q)
    dev = np.abs(rl["f"] - r["f"]).max()
    assert 0.0 < dev < 0.1 * np.abs(r["f"]).max()
    assert np.median(np.abs(rl["f"] - r["f"])) < 0.01 * np.median(np.abs(r["f"]))
    only_angles = lammps(sym, x, box, q=q, terms=ALL & ~2)
    assert np.abs(only_angles["f"] - exact(sym, x, box, q=q, terms=ALL & ~2)["f"]).max() == 0.0


def test_condensed_mixture_with_hydrogen_bonds(ff, drv):
    exact, _ = drv
    sym, x, box = _mixture()
    t = ff.types(sym)
    q, _ = ff.qeq(t, x, box=box, tol=1e-10, maxiter=500)
    r = exact(sym, x, box)
    assert np.abs(r["q"] - q).max() < 1e-8
    r = exact(sym, x, box, q=q)
    _, po = ff.energy(t, x, box=box, q=q)
    for k in pr.PARTS:
        assert abs(r["e"][k] - po[k]) < 1e-9 * max(1.0, abs(po[k])), k
    assert po["hb"] < -1.0 and abs(po["lp"]) > 0.01
    good, worst = _check_dirs(ff, t, x, box, q, r["f"], seed=3)
    assert good >= 5, worst


def test_list_radius_beyond_the_cutoff_changes_nothing(ff, drv):
    exact, _ = drv
    sym, x, box = _mixture(seed=4)
    a = exact(sym, x, box, rlist=10.0)
    b = exact(sym, x, box, rlist=10.5)      # the skin of the neighbour rows
    assert np.abs(a["q"] - b["q"]).max() < 1e-9
    assert np.abs(a["f"] - b["f"]).max() < 1e-7 * np.abs(a["f"]).max()
    assert all(abs(a["e"][k] - b["e"][k]) < 1e-8 * max(1.0, abs(a["e"][k])) for k in pr.PARTS)


def test_small_box_uses_images(ff, drv):
    """a box thinner than the cutoff: the rows hold several images of the same atom; the energy of a 2x2x2 supercell is 8 times it"""
    exact, _ = drv
    rng = np.random.default_rng(2)
    a = 1.09 / np.sqrt(3)
    mol = np.array([[0, 0, 0], [a, a, a], [-a, -a, a], [-a, a, -a], [a, -a, -a]])
    L = 8.0
    x = np.vstack([mol + np.array([2.0, 2.0, 2.0]), mol @ np.linalg.qr(rng.standard_normal((3, 3)))[0].T + np.array([6.0, 5.5, 6.2])])
    sym = ["C", "H", "H", "H", "H"] * 2
    box = np.array([0, 0, 0, L, L, L, 1.0, 0.5, -0.7])
    q = 0.1 * np.array([-4, 1, 1, 1, 1] * 2, float)
    small = exact(sym, x, box, q=q)
    av, bv, cv = np.array([L, 0, 0]), np.array([1.0, L, 0]), np.array([0.5, -0.7, L])
    xs = np.vstack([x + i * av + j * bv + k * cv for i in range(2) for j in range(2) for k in range(2)])
    box2 = np.array([0, 0, 0, 2 * L, 2 * L, 2 * L, 2.0, 1.0, -1.4])
    big = exact(sym * 8, xs, box2, q=np.tile(q, 8))
    for k in pr.PARTS:
        assert abs(8 * small["e"][k] - big["e"][k]) < 1e-9 * max(1.0, abs(big["e"][k])), k
    assert np.abs(big["f"][:10] - small["f"]).max() < 1e-9 * np.abs(small["f"]).max()
    assert np.abs(8 * small["w"] - big["w"]).max() < 1e-9 * np.abs(big["w"]).max()
