"""Time per RL iteration of the zero-boundary ('spatial', decon.m:25-120) flavour on the three engines: direct,
FFT through the hand-written pipeline on a padded 2^a*{1,3,9} volume, FFT through rocFFT on a padded 7-smooth volume.
usage: python profiles/padded_time.py [c2|c3]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ipp_amd import capi, decon

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
shape, kshape = {"c2": ((256, 1024, 1024), (31, 15, 15)), "c3": ((512, 2048, 2048), (61, 31, 31))}[wl]
dev = torch.device("cuda", 0)
z, y, x = np.meshgrid(*[np.arange(k) - (k - 1) / 2 for k in kshape], indexing="ij")
psf = np.exp(-(z / 8.0) ** 2 - (y / 3.0) ** 2 - (x / 3.0) ** 2).astype(np.float32)
psf /= psf.sum()
psf_inv = np.ascontiguousarray(psf[::-1, ::-1, ::-1])
bl0 = torch.rand(shape, device=dev) + 0.1


def run(engine, label, iters=4):
    t0 = time.perf_counter()
    ctx = decon.RLContext(shape, psf, psf_inv, boundary=capi.BOUNDARY_ZERO, engine=engine, device=dev)
    torch.cuda.synchronize()
    print(f"{wl} zero-boundary {label}: context created in {time.perf_counter() - t0:.2f} s", flush=True)
    bl = bl0.clone()
    ratio = None if (engine == capi.ENGINE_FFT and "native" in label) else torch.empty_like(bl0)
    ctx.iterate(bl, ratio, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.iterate(bl, ratio, iters)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    print(f"{wl} zero-boundary {label}: {ms:.1f} ms/iteration, device bytes {ctx.device_bytes / 2**30:.2f} GiB", flush=True)
    out = bl.clone()
    del ctx
    return out


a = run(capi.ENGINE_FFT, "fft native padded")
os.environ["MI_FFT_ROCFFT"] = "1"
b = run(capi.ENGINE_FFT, "fft rocFFT padded")
del os.environ["MI_FFT_ROCFFT"]
print("native vs rocFFT rel diff", float((a - b).abs().max() / b.abs().max()))
if wl == "c2":
    c = run(capi.ENGINE_DIRECT, "direct")
    print("native vs direct rel diff", float((a - c).abs().max() / c.abs().max()))
