"""ms per fused RL iteration (circular grid, native FFT pipeline) and per pass for an arbitrary (Z, Y, X) shape:
    python profiles/shape_time.py 1024 576 4096 [kz ky kx]
e.g. the rank-local array of BASELINE C4 on 8 GPUs (4096x4096x1024 volume, 63x63x127 PSF: 512 + 2x32 rows -> 576)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ipp_amd import capi, decon

shape = tuple(int(v) for v in sys.argv[1:4])
kshape = tuple(int(v) for v in sys.argv[4:7]) if len(sys.argv) >= 7 else (31, 15, 15)
dev = torch.device("cuda", 0)
z, y, x = np.meshgrid(*[np.arange(k) - (k - 1) / 2 for k in kshape], indexing="ij")
psf = np.exp(-(z / 8.0) ** 2 - (y / 3.0) ** 2 - (x / 3.0) ** 2).astype(np.float32)
psf /= psf.sum()
ctx = decon.RLContext(shape, psf, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_FFT, device=dev)
bl = torch.rand(shape, device=dev) + 0.1
ctx.iterate(bl, None, 2)
torch.cuda.synchronize()
t0 = time.perf_counter()
ctx.iterate(bl, None, 5)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 5 * 1e3
nvox = float(np.prod(shape))
print(f"shape {shape}: {ms:.2f} ms/iteration = {nvox / ms / 1e6:.1f} Gvoxel*it/s, device bytes {ctx.device_bytes / 2**30:.1f} GiB", flush=True)
for name in ("y_forward", "z_conv", "y_inverse", "x_fused", "x_fused_update"):
    t = ctx.time_pass(name, bl, reps=5)
    b = {"y_forward": 8, "y_inverse": 8, "z_conv": 12, "x_fused": 12, "x_fused_update": 16}[name]
    print(f"   {name:15s} {t:7.3f} ms  {b * nvox / t / 1e9:6.2f} TB/s", flush=True)
