"""Row padding of the spectrum arrays (PROBE_VAR = MI_FFT_ZPAD: paired z side, MI_FFT_XPAD: x side; float4 per row): pass times
over several allocations per value."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ipp_amd import capi, decon

shape = tuple(int(v) for v in os.environ.get("PROBE_SHAPE", "512 2048 2048").split())
pads = [v for v in sys.argv[1:]] or ["0", "4", "8", "12", "16", "24", "32"]  # "x,z" sets both
dev = torch.device("cuda", 0)
z, y, x = np.meshgrid(*[np.arange(k) - (k - 1) / 2 for k in (31, 15, 15)], indexing="ij")
psf = np.exp(-(z / 8.0) ** 2 - (y / 3.0) ** 2 - (x / 3.0) ** 2).astype(np.float32)
psf /= psf.sum()
bl = torch.rand(shape, device=dev) + 0.1
res = {p: [] for p in pads}
for rep in range(3):
    for pad in pads:
        if pad.count(",") == 2:
            os.environ["MI_FFT_XPAD"], os.environ["MI_FFT_ZPAD"], os.environ["MI_FFT_STGAP"] = pad.split(",")
        elif "," in pad:
            os.environ["MI_FFT_XPAD"], os.environ["MI_FFT_ZPAD"] = pad.split(",")
        else:
            os.environ[os.environ.get("PROBE_VAR", "MI_FFT_ZPAD")] = pad
        ctx = decon.RLContext(shape, psf, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_FFT, device=dev)
        ctx.iterate(bl, None, 1)
        res[pad].append(tuple(ctx.time_pass(n, bl, reps=5) for n in ("z_conv", "y_forward", "y_inverse", "x_fused", "x_fused_update")))
        del ctx
for pad in pads:
    print(f"pad {pad:>14s}: " + "   ".join("z %.3f yf %.3f yi %.3f x %.3f/%.3f = %.2f" % (r + (r[3] + r[4] + 2 * (r[0] + r[1] + r[2]),)) for r in res[pad]), flush=True)
