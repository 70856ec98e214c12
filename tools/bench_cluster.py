"""Times k_hist_compare (all-pairs spline distances) on device-resident data and prices it against the
HBM roofline: algorithmic bytes = n*d*8 read + n*n*8 written.  usage: python tools/bench_cluster.py [n] [d]"""
import ctypes as C
import json
import sys

import torch

sys.path.insert(0, ".")
from scema_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4864
d = int(sys.argv[2]) if len(sys.argv) > 2 else 60
sp = torch.randn(n, d, dtype=torch.float64, device="cuda") * 1e-3
out = torch.empty(n, n, dtype=torch.float64, device="cuda")
L = capi.lib()
st = torch.cuda.current_stream().cuda_stream
call = lambda: L.scema_hist_compare_device(C.c_void_p(sp.data_ptr()), C.c_int32(n), C.c_int32(d), C.c_void_p(out.data_ptr()), C.c_void_p(st))
for _ in range(3):
    assert call() == 0
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
a.record()
for _ in range(reps):
    call()
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / reps
bytes_alg = n * d * 8 + n * n * 8
ref = torch.cdist(sp, sp)
print(json.dumps({"kernel": "k_hist_compare", "n": n, "d": d, "ms": ms, "alg_bytes": bytes_alg, "GBps": bytes_alg / ms / 1e6,
                  "frac_of_8TBps": bytes_alg / ms / 1e6 / 8000.0, "pair_rate_G_per_s": n * n / 2 / ms / 1e6,
                  "max_abs_dev_vs_torch_cdist": float((out - ref).abs().max())}))
