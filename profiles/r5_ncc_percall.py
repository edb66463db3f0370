"""Wall time of each of the first calls of compute_displacements on the C5 grid in a fresh process (is there a transient?)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench_ncc  # noqa: E402
from ipp_amd import crossmips  # noqa: E402

dev = torch.device("cuda", 0)
tiles, jit, step = bench_ncc.make_grid(dev)
torch.cuda.synchronize(dev)
ts = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 14):
    t0 = time.perf_counter()
    crossmips.compute_displacements(tiles, bench_ncc.OVERLAP, bench_ncc.OVERLAP, *bench_ncc.DISPL)
    torch.cuda.synchronize(dev)
    ts.append((time.perf_counter() - t0) * 1e3)
print("ms per call:", " ".join(f"{t:.2f}" for t in ts), flush=True)
