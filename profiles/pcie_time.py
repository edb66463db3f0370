"""PCIe-inclusive figure for DESIGN.md: H2D + D2H of a C3 volume (pinned host memory) beside the on-device time."""
import time, torch
n = 512 * 2048 * 2048
host = torch.empty(n, dtype=torch.float32).pin_memory()
dev = torch.empty(n, dtype=torch.float32, device="cuda")
for name, fn in [("H2D", lambda: dev.copy_(host, non_blocking=True)), ("D2H", lambda: host.copy_(dev, non_blocking=True))]:
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name} 8.59 GB pinned: {dt*1e3:.0f} ms = {n*4/dt/1e9:.1f} GB/s")
