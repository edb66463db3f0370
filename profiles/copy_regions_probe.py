"""Is the same-region penalty of the strided passes a property of plain streaming too?  K buffers of 8.6 GB allocated one after the
other; torch's device-to-device copy timed between every ordered pair (ms; 17.2 GB moved per copy).
    python profiles/copy_regions_probe.py [K]"""
import sys

import numpy as np
import torch

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
n = 512 * 2048 * 2048
bufs = [torch.zeros(n, dtype=torch.float32, device=dev) for _ in range(K)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t = np.zeros((K, K))
for i in range(K):
    for j in range(K):
        if i == j:
            continue
        bufs[j].copy_(bufs[i])
        e0.record()
        for _ in range(3):
            bufs[j].copy_(bufs[i])
        e1.record()
        torch.cuda.synchronize(dev)
        t[i, j] = e0.elapsed_time(e1) / 3
print(f"device-to-device copy of {4 * n / 1e9:.1f} GB from buffer i (rows) to buffer j (columns), ms:")
for i in range(K):
    print("  " + "  ".join("  -  " if i == j else f"{t[i, j]:5.3f}" for j in range(K)))
off = t[~np.eye(K, dtype=bool)]
print(f"fastest pair {off.min():.3f} ms = {2 * 4 * n / off.min() / 1e9:.2f} TB/s, slowest {off.max():.3f} ms = {2 * 4 * n / off.max() / 1e9:.2f} TB/s")
