"""energy drift of the ReaxFF path in NVE as a function of the time step (diagnostic)"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from scema_amd import capi
from test_reax_host import _mixture, _pe_cell
from test_oracle_reax import FFIELD
MVV2E = 48.88821291 ** 2
masses = dict(H=1.008, C=12.011, N=14.007, O=15.999)
which = sys.argv[1] if len(sys.argv) > 1 else "mix"
if which == "mix":
    sym, x, box = _mixture(seed=8)
else:
    sym, x, box = _pe_cell(None, amp=0.02)
n = len(sym)
m = np.array([masses[s] for s in sym])
v = np.random.default_rng(0).standard_normal((n, 3)) * np.sqrt(0.0019872067 * 300.0 / (m[:, None] * MVV2E))
v -= (m[:, None] * v).sum(0) / m.sum()
for exact in (1, 0):
    for dt, nst in ((0.2, 300), (0.1, 600), (0.05, 1200)):
        e = capi.Engine()
        e.reax_configure(FFIELD, qeq_tol=1e-10, skin=1.0)
        e.reax_set(exact_gradient=exact)
        e.register_replica("m", 1, capi.reax_system(sym, x, box, v=v))
        e.set_state(0, "m", 1, box, x, v)
        def etot():
            r = e.reax_compute("m", 1, qp=0)
            _, _, vv = e.get_state(0, "m", 1)
            return sum(r["e"].values()) + 0.5 * MVV2E * (m[:, None] * vv * vv).sum(), 0.5 * MVV2E * (m[:, None] * vv * vv).sum()
        e0, k0 = etot()
        tr = []
        for seg in range(4):
            e.debug_run("m", 1, nst // 4, dt, 300.0, qp=0, nvt=False, use_shake=False)
            tr.append(etot()[0] - e0)
        print(which, "exact" if exact else "lammps", "dt", dt, "KE0 %.1f" % k0, "drift", ["%.4f" % t for t in tr], e.reax_stats(), flush=True)
        e.close()
