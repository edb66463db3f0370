"""The library's TIFF writer / reader alone on this host (no GPU): 128 slices of 2048 x 2048 uint16 -- noise-like and smooth content,
deflate and raw -- write and read rates.    python profiles/tiff_io_speed.py"""
import os
import shutil
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from ipp_amd import brickio, capi

rng = np.random.default_rng(0)
d = "/tmp/tiff_io_speed"
noise = (rng.random((128, 2048, 2048)) * 3000 + 200).astype(np.uint16)
smooth = (np.cumsum(rng.standard_normal((128, 2048, 2048)).astype(np.float32), axis=2) * 5 + 3000).clip(0, 65535).astype(np.uint16)
print(f"codec {capi.lib().mi_tiff_codec().decode()}, {len(os.sched_getaffinity(0))} cores", flush=True)
for name, v in (("noise-like", noise), ("smooth", smooth)):
    for comp in ("tiff_adobe_deflate", None):
        shutil.rmtree(d, ignore_errors=True)
        t = time.perf_counter()
        brickio.save_tiff_series(d, v, compression=comp)
        dt = time.perf_counter() - t
        sz = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d))
        t = time.perf_counter()
        a = brickio.load_tiff_series(d)
        dr = time.perf_counter() - t
        print(f"{name:10s} {'deflate' if comp else 'raw':8s}: write {v.nbytes / dt / 1e9:5.2f} GB/s (files {sz / v.nbytes:.2f} of the samples), read {v.nbytes / dr / 1e9:5.2f} GB/s, "
              f"identical {bool(np.array_equal(a, v))}", flush=True)
shutil.rmtree(d, ignore_errors=True)
