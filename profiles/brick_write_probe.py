"""One decwrap-sized brick (486 x 486 x 909 float32 = 859 MB of noise in a pinned buffer) through brickio.save_lz4 with a pool of nine
threads, as decwrap's writers do it: seconds per brick, alone on the host.    python profiles/brick_write_probe.py"""
import os
import shutil
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ipp_amd import brickio

d = "/tmp/brick_probe"
shutil.rmtree(d, ignore_errors=True)
os.makedirs(d)
dev = torch.device("cuda", 0)
core = torch.rand((909, 486, 486), device=dev)
host = torch.empty(core.numel(), dtype=torch.float32, pin_memory=True)
view = host.view(core.shape)
pool = ThreadPoolExecutor(9)
for rep in range(6):
    t0 = time.perf_counter()
    view.copy_(core, non_blocking=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    brickio.save_lz4(os.path.join(d, f"b{rep}.lz4.tmp"), view.numpy(), chunk_size=32 << 20, pool=pool)
    t2 = time.perf_counter()
    os.replace(os.path.join(d, f"b{rep}.lz4.tmp"), os.path.join(d, f"b{rep}.lz4"))
    print(f"brick {rep}: D2H {1e3 * (t1 - t0):.0f} ms, save_lz4 {1e3 * (t2 - t1):.0f} ms = {core.numel() * 4 / (t2 - t1) / 1e9:.1f} GB/s, "
          f"file {os.path.getsize(os.path.join(d, f'b{rep}.lz4')) / (core.numel() * 4):.4f} of the samples", flush=True)
back = brickio.load_lz4(os.path.join(d, "b0.lz4"), pool=pool)
print("reads back:", bool(np.array_equal(back, view.numpy())))
shutil.rmtree(d, ignore_errors=True)
