"""decwrap.py end to end on a FOLDER OF TIFF SLICES (the reference's own input and output form: LsDeconv.m:585-588, 1120-1145) --
16-bit slices in, deflate-compressed 16-bit slices out -- with the library's TIFF reader / writer (default) or Pillow
(MI_TIFF_PILLOW=1).    python profiles/decwrap_tiff_probe.py [nz ny nx]"""
import os
import shutil
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from ipp_amd import brickio  # noqa: E402

shape = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (512, 2048, 2048)
root = "/tmp/decwrap_tiff"
shutil.rmtree(root, ignore_errors=True)
src = os.path.join(root, "stack")
rng = np.random.default_rng(1)
t0 = time.perf_counter()
for z0 in range(0, shape[0], 64):                   # sparse beads on a noisy background, 64 slices at a time
    n = min(64, shape[0] - z0)
    sl = rng.integers(600, 700, size=(n,) + shape[1:], dtype=np.uint16)
    idx = rng.integers(0, sl.size, size=sl.size // 2000)
    sl.reshape(-1)[idx] = rng.integers(5000, 60000, size=idx.size, dtype=np.uint16)
    brickio.save_tiff_series(src, sl, first_index=z0 + 1)
t_gen = time.perf_counter() - t0
in_bytes = sum(os.path.getsize(os.path.join(src, f)) for f in os.listdir(src))
from ipp_amd import decwrap  # noqa: E402
os.environ["MI_DECWRAP_NPY"] = "0"
t0 = time.perf_counter()
rc = decwrap.main(["-i", src, "-dxy", "0.422", "-dz", "1.0", "-ex", "488", "-em", "525", "--use-fft", "-it", "6", "--block-size-max", "300000000",
                   "--gpu-indices", "1", "--gpu-workers-per-gpu", "5"])
dt = time.perf_counter() - t0
out = os.path.join(src, "deconvolved")
n_out = len([f for f in os.listdir(out) if f.endswith(".tif")])
out_bytes = sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out) if f.endswith(".tif"))
tm = getattr(decwrap.main, "last_timing", {})
nvox = float(np.prod(shape))
route = "Pillow" if os.environ.get("MI_TIFF_PILLOW") else "the library's reader / writer"
print(f"TIFF folder in, TIFF folder out through {route}: {shape[2]} x {shape[1]} x {shape[0]} uint16 = {nvox * 2 / 1e9:.1f} GB in {shape[0]} slices "
      f"({in_bytes / 1e9:.1f} GB of deflate TIFFs, written in {t_gen:.1f} s); rc {rc}, {dt:.1f} s wall = {nvox / dt / 1e6:.0f} Mvoxel/s end to end "
      f"(blocks phase {tm.get('blocks_wall_s', 0.0):.1f} s, assembly + slices out {tm.get('assembly_wall_s', 0.0):.1f} s); {n_out} slices out, {out_bytes / 1e9:.1f} GB", flush=True)
a = brickio.load_tiff_series(out, 0, 2)
print(f"   first output slices: dtype {a.dtype}, shape {a.shape[1:]}, max {int(a.max())}", flush=True)
shutil.rmtree(root, ignore_errors=True)
