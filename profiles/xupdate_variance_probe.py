"""Pass times of the native pipeline over repeated allocations inside one process (one box, one clock state): every trial
re-creates the context (its two spectrum arrays come back from the driver or the pool at other addresses / in the other order)
while the volume stays where it is.

Finding of round 2 (profiles/r02_xupdate_variance.txt): the passes are bimodal per ALLOCATION -- update launch of the fused x
pass 5.9-6.2 or 7.0-7.3 ms, z pass 4.6 or 4.9 ms, y passes 3.2 or 3.35 ms -- and an allocation keeps its mode for its lifetime.
Offsets of the arrays from 2-MiB boundaries (0 .. 1 MiB in steps of 4 KiB .. 256 KiB, both arrays, 36 combinations) and a
rotated tile order do not select the mode: it is a property of the physical pages behind the allocation."""
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
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    ctx = decon.RLContext(shape, psf, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_FFT, device=dev)
    ctx.iterate(bl, None, 1)
    tu = ctx.time_pass("x_fused_update", bl, reps=5)
    tr = ctx.time_pass("x_fused", bl, reps=5)
    tz = ctx.time_pass("z_conv", bl, reps=5)
    ty = ctx.time_pass("y_forward", bl, reps=5) + ctx.time_pass("y_inverse", bl, reps=5)
    print(f"trial {trial}: update {tu:.3f} ms, ratio {tr:.3f} ms, z {tz:.3f} ms, y fwd + inv {ty:.3f}: iteration {tu + tr + 2 * (tz + ty):.2f}",
          flush=True)
    del ctx
