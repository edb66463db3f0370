"""Kernel breakdown helper: a few fused RL iterations of the zero-boundary flavour on the native padded grid (run under rocprofv3).
usage: python3 profiles/padded_native_probe.py [c2|c3]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ipp_amd import capi, decon

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
shape, kshape = {"c2": ((256, 1024, 1024), (31, 15, 15)), "c3": ((512, 2048, 2048), (61, 31, 31))}[wl]
dev = torch.device("cuda", 0)
z, y, x = np.meshgrid(*[np.arange(k) - (k - 1) / 2 for k in kshape], indexing="ij")
psf = np.exp(-(z / 8.0) ** 2 - (y / 3.0) ** 2 - (x / 3.0) ** 2).astype(np.float32)
psf /= psf.sum()
psf_inv = np.ascontiguousarray(psf[::-1, ::-1, ::-1])
os.environ.setdefault("MI_NO_SEPARABLE", "1")
bl = torch.rand(shape, device=dev) + 0.1
ctx = decon.RLContext(shape, psf, psf_inv, boundary=capi.BOUNDARY_ZERO, engine=capi.ENGINE_FFT, device=dev)
ctx.iterate(bl, None, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
ctx.iterate(bl, None, 4)
torch.cuda.synchronize()
print(f"{wl} zero-boundary, native padded grid: {(time.perf_counter() - t0) / 4 * 1e3:.2f} ms/iteration", flush=True)
