"""Phase knock-out timing of the z pass (library built with EXTRA=-DMI_ZDBG; results of the pass are wrong then).
bits of MI_FFT_ZDBG: 1 no spectrum loads, 2 no stores, 4 no transforms, 8 no point-wise step, 16 no OTF loads"""
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
for dbg in [int(v) for v in sys.argv[1:]] or [0, 1, 2, 3, 16, 19, 4, 8, 12, 31, 28]:
    os.environ["MI_FFT_ZDBG"] = str(dbg)
    ctx = decon.RLContext(shape, psf, None, boundary=capi.BOUNDARY_CIRCULAR, engine=capi.ENGINE_FFT, device=dev)
    ts = {n: ctx.time_pass(n, bl, reps=5) for n in ("z_conv", "x_fused", "y_forward")}
    print(f"dbg {dbg:2d}: z {ts['z_conv']:6.3f} ms   (x {ts['x_fused']:.3f}, y {ts['y_forward']:.3f})", flush=True)
    del ctx
