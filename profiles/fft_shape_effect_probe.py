"""What the FFT shape of a deconFFT block does to its result: a remainder block of decwrap (130 x 130 x 280 incl. pads, beads on a noisy
background as in decwrap_scale_probe.py, PSF 9 x 9 x 19, 6 iterations, regularisation every 3rd) on the 7-smooth grid (rocFFT route),
on the grid of the hand-written pipeline with the default placement of the PSF and with the 7-smooth grid's (mi_rl_options.psf_grid), and on the CPU
oracle for each: core values and the 99.99th percentile that becomes
the block's clip bound (LsDeconv.m:1300-1307).
    python profiles/fft_shape_effect_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ipp_amd import decon as D, lsdeconv as L
from oracle import rl_oracle as R

rng = np.random.default_rng(1)
shape = (280, 130, 130)  # z y x
vol = rng.integers(600, 700, size=shape).astype(np.float32)
idx = rng.integers(0, vol.size, size=vol.size // 2000)
vol.reshape(-1)[idx] = rng.integers(5000, 60000, size=idx.size).astype(np.float32)
psf = R.gaussian_psf((19, 9, 9), (2.5, 1.2, 1.2))
pad = (25, 13, 13)
bl_xyz = shape[::-1]
smooth, native = L.next_fast_len(bl_xyz), L.native_fft_shape(bl_xyz)
print("block", bl_xyz, "7-smooth", smooth, "native", native)
dev = torch.device("cuda", 0)
t = torch.from_numpy(vol).to(dev)
D.gauss3d_gpu(t, [0.5, 0.5, 2.5], [13, 13, 25])      # decwrap's default pre-filter (LsDeconv.m:917-919)
vol = t.cpu().numpy()
res = {}
for name, fs, grid in (("7-smooth", smooth, None), ("native", native, None), ("native+grid", native, smooth)):
    t = torch.from_numpy(vol.copy()).to(dev)
    D.decon(t, psf, 6, 0.0, 0.0, 3, 1, True, fs, False, psf_grid=grid)
    g = t.cpu().numpy()
    o = R.decon_fft(vol, psf, tuple(fs[::-1]), 6, 0.0, 0.0, 3, psf_grid_zyx=None if grid is None else tuple(grid[::-1]))
    core = (slice(pad[0], -pad[0]), slice(pad[1], -pad[1]), slice(pad[2], -pad[2]))
    res[name] = (g, o)
    print(f"{name:9s} grid {fs}: device vs oracle max |diff| {np.abs(g - o).max():.3e} (max value {o.max():.4g}); "
          f"99.99th percentile whole block: device {np.percentile(g, 99.99):.5g}, oracle {np.percentile(o, 99.99):.5g}; "
          f"core only: {np.percentile(g[core], 99.99):.5g}; max of core {g[core].max():.5g}, max of the pads {max(g[:pad[0]].max(), g[-pad[0]:].max(), g[:, :pad[1]].max(), g[:, -pad[1]:].max(), g[:, :, :pad[2]].max(), g[:, :, -pad[2]:].max()):.5g}")
core = (slice(pad[0], -pad[0]), slice(pad[1], -pad[1]), slice(pad[2], -pad[2]))
a = res["7-smooth"][0]
for other in ("native", "native+grid"):
    b = res[other][0]
    print(f"7-smooth vs {other}, core voxels: max |diff| {np.abs(a[core] - b[core]).max():.4g} = {np.abs(a[core] - b[core]).max() / a[core].max():.3e} of the core's max, "
          f"relative L2 {np.linalg.norm(a[core] - b[core]) / np.linalg.norm(a[core]):.3e}; whole block: max |diff| {np.abs(a - b).max():.4g}")
