"""Practical HBM ceiling of the box: device-to-device copy and a read-only reduction over 4 GiB (torch; not part of the product).
Usage (GPU box): python tools/hbm_ceiling.py"""
import torch, time
n = 1 << 29   # doubles: 4 GiB
a = torch.ones(n, dtype=torch.float64, device="cuda"); b = torch.empty_like(a)
def t(f, reps=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
dt = t(lambda: b.copy_(a)); print(f"copy 4 GiB -> 4 GiB: {2 * n * 8 / dt / 1e12:.2f} TB/s (read + write)")
dt = t(lambda: a.sum()); print(f"sum over 4 GiB: {n * 8 / dt / 1e12:.2f} TB/s (read)")
dt = t(lambda: b.fill_(1.0)); print(f"fill 4 GiB: {n * 8 / dt / 1e12:.2f} TB/s (write)")
