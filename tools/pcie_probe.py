"""Pinned host <-> device copy rate of this box (context for the host-staged transport numbers)."""
import time
import torch
n = 256 << 20
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
for name, fn in (("H2D", lambda: d.copy_(h, non_blocking=True)), ("D2H", lambda: h.copy_(d, non_blocking=True))):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    print(f"{name} pinned 256 MiB: {5 * n / (time.perf_counter() - t) / 1e9:.1f} GB/s")
