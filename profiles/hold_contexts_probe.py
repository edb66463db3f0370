"""Do spectrum arrays that are allocated while earlier ones are still held land differently?  One process, C3: contexts created one
after the other and ALL kept alive (each takes its 24 GB from memory no earlier context occupies); the y and x pass times of each.
    python profiles/hold_contexts_probe.py [n]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ipp_amd import capi, decon

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
shape, kshape = (512, 2048, 2048), (61, 31, 31)
dev = torch.device("cuda", 0)
z, y, x = np.meshgrid(*[np.arange(k) - (k - 1) / 2 for k in kshape], indexing="ij")
psf = np.exp(-(z / 8.0) ** 2 - (y / 3.0) ** 2 - (x / 3.0) ** 2).astype(np.float32)
psf /= psf.sum()
bl = torch.rand(shape, device=dev) + 0.1
bl2 = torch.rand(shape, device=dev) + 0.1   # a second volume: does the update launch follow the spectrum arrays or the volume?
held = []
for i in range(n):
    ctx = decon.RLContext(shape, psf, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_FFT, device=dev)
    ctx.iterate(bl, None, 1)
    t = {name: ctx.time_pass(name, bl, reps=5) for name in ("y_forward", "y_inverse", "z_conv", "x_fused", "x_fused_update")}
    t["x_fused_update(other volume)"] = ctx.time_pass("x_fused_update", bl2, reps=5)
    free, total = torch.cuda.mem_get_info(dev)
    print(f"context {i + 1} (held: {len(held)} earlier, {(total - free) / 2**30:.0f} GiB in use): " + "  ".join(f"{k} {v:.3f}" for k, v in t.items()), flush=True)
    held.append(ctx)
