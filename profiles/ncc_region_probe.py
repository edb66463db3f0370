"""Does the region of device memory the TILES lie in decide the speed of the NCC batch?  The 8 x 8 grid of bench_ncc.py built after
SKIP_GB gigabytes have been allocated (and are kept) in front of it: pairs/s of crossmips.compute_displacements.
    python profiles/ncc_region_probe.py [skip_gb ...]     (one process per value is fairer: run it once per value)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench_ncc
from ipp_amd import crossmips

dev = torch.device("cuda", 0)
skip = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
hold = torch.empty(int(skip * (1 << 30)), dtype=torch.uint8, device=dev) if skip > 0 else None
tiles, jit, step = bench_ncc.make_grid(dev)
args = (bench_ncc.OVERLAP, bench_ncc.OVERLAP, *bench_ncc.DISPL)
res = crossmips.compute_displacements(tiles, *args)
torch.cuda.synchronize(dev)
t0 = time.perf_counter()
for _ in range(10):
    crossmips.compute_displacements(tiles, *args)
torch.cuda.synchronize(dev)
dt = (time.perf_counter() - t0) / 10
print(f"{skip:5.0f} GB held in front of the tiles: {dt * 1e3:.2f} ms per {len(res)} pairs = {len(res) / dt:.0f} pairs/s", flush=True)
