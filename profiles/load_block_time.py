"""mi_load_block on a decwrap-sized block: 512 x 512 x 959 incl. pads from a 486 x 486 x 909 uint16 box at a volume corner (mirrored
pads on three sides) and from an interior box (no mirroring); run under rocprofv3 --kernel-trace --stats for the kernel's time.
    python profiles/load_block_time.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ipp_amd import lsdeconv as L

dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
vol = rng.integers(0, 65535, size=(1100, 700, 700), dtype=np.uint16)
pad = (13, 13, 25)
for name, p1, p2 in (("corner box", (1, 1, 1), (486, 486, 909)), ("interior box", (101, 101, 101), (586, 586, 1009))):
    staging = {}
    t = L.load_block_device(vol, p1, p2, pad, dev, staging)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        t = L.load_block_device(vol, p1, p2, pad, dev, staging)
    torch.cuda.synchronize()
    print(f"{name}: block {tuple(t.shape)}, load_block_device {((time.perf_counter() - t0) / 3) * 1e3:.1f} ms per call (box copy + upload + conversion)", flush=True)
    want = L.load_block(vol, p1, p2, pad)
    print(f"   bit-identical to the host load_block: {bool(np.array_equal(t.cpu().numpy(), want))}", flush=True)
