"""A/B timing of z-pass variants inside one process (same box, same clocks): env switches read at launch time."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ipp_amd import capi, decon

shape = (512, 2048, 2048)
dev = torch.device("cuda", 0)
z, y, x = np.meshgrid(*[np.arange(k) - (k - 1) / 2 for k in (31, 15, 15)], indexing="ij")
psf = np.exp(-(z / 8.0) ** 2 - (y / 3.0) ** 2 - (x / 3.0) ** 2).astype(np.float32)
psf /= psf.sum()
bl = torch.rand(shape, device=dev) + 0.1
ctx = decon.RLContext(shape, psf, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_FFT, device=dev)
ctx.iterate(bl, None, 1)
switches = sys.argv[1:] or ["MI_FFT_ZEARLY"]
for rep in range(3):
    for sw in [None] + switches:
        for k in switches:
            os.environ.pop(k, None)
        if sw:
            os.environ[sw] = "1"
        print(f"{sw or 'default':16s}: z {ctx.time_pass('z_conv', bl, reps=10):6.3f} ms   x {ctx.time_pass('x_fused', bl, reps=5):6.3f}", flush=True)
