"""The TIFF writer with deflate on the device (mi_tiff_write_series_device) against the host writer (libdeflate level 1 on the host's
cores): 256 slices of 2048 x 2048 uint16 resident on the device / in host memory, noise-like and smooth content.
    python profiles/tiff_device_speed.py"""
import os
import shutil
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ipp_amd import brickio

dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
d = "/tmp/tiff_dev_speed"
nz = 256
noise = (rng.random((nz, 2048, 2048)) * 3000 + 200).astype(np.uint16)
smooth = (np.cumsum(rng.standard_normal((nz, 2048, 2048)).astype(np.float32), axis=2) * 5 + 3000).clip(0, 65535).astype(np.uint16)
for name, v in (("noise-like", noise), ("smooth", smooth)):
    t = torch.from_numpy(v).to(dev)
    for rep in range(2):
        shutil.rmtree(d, ignore_errors=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        brickio.save_tiff_series_device(d, t)
        dt = time.perf_counter() - t0
        sz = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d))
        ok = bool(np.array_equal(brickio.load_tiff_series(d, 0, 8), v[:8]))
        print(f"{name:10s} device deflate: {v.nbytes / dt / 1e9:5.2f} GB/s ({dt:.2f} s), files {sz / v.nbytes:.3f} of the samples, first slices identical {ok}", flush=True)
    shutil.rmtree(d, ignore_errors=True)
    t0 = time.perf_counter()
    h = t.cpu().numpy()
    t1 = time.perf_counter()
    brickio.save_tiff_series(d, h)
    dt = time.perf_counter() - t0
    sz = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d))
    print(f"{name:10s} host deflate:   {v.nbytes / dt / 1e9:5.2f} GB/s ({dt:.2f} s incl. {t1 - t0:.2f} s of D2H into pageable memory), files {sz / v.nbytes:.3f} of the samples", flush=True)
    del t
shutil.rmtree(d, ignore_errors=True)
