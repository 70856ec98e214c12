"""usage (GPU box): [SCEMA_MD_GRAPH=1] [SCEMA_MD_ONE_STREAM=1] python tools/graph_cmp.py <n replicas>
ms per update() of n PE-10k replicas without per-launch profiling: hipGraph replay vs plain launches."""
import sys, time, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from scema_amd import capi
from scema_amd.systems import build_pe, synthetic_strains
n = int(sys.argv[1])
d = build_pe(6, 9, 16, shake_project=True)
eng = capi.Engine(capi.default_params(profile=0))
eng.register_replica("g0", 1, d)
lens = d["box"][3:6] - d["box"][:3]
def upd(it):
    st = synthetic_strains(n, lens, seed=2026 + it)
    sims = [capi.make_sim(q, "g0", 1, st[q], nss=100, most_recent=(capi.QP_NONE if it == 0 else q)) for q in range(n)]
    return eng.strain_batch(sims)
upd(0)
t0 = time.perf_counter(); arr = upd(1); arr = upd(2); t1 = time.perf_counter()
mode = "graph replay" if os.environ.get("SCEMA_MD_GRAPH") else "plain launches"
streams = "one stream" if os.environ.get("SCEMA_MD_ONE_STREAM") else "side stream"
print(f"n={n} {mode}, {streams}: {(t1 - t0) / 2 * 1e3:.2f} ms per update, checksum {sum(a.stress[2] for a in arr):.6e}")
