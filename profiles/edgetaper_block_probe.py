"""edgetaper_3d routes on a decwrap-sized block (512 x 512 x 959 incl. pads, LsMakePSF PSF 9 x 9 x 19) and on a C3 block: the route the
cost model picks (no override) against each forced one.    python profiles/edgetaper_block_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ipp_amd import decon  # noqa: E402

dev = torch.device("cuda", 0)
for shape, kshape in (((959, 512, 512), (19, 9, 9)), ((256, 1024, 1024), (31, 15, 15)), ((512, 2048, 2048), (61, 31, 31))):
    psf = torch.from_numpy(bench.make_psf(kshape)).to(dev)
    for eng in (None, "slabs", "fft", "direct"):
        if eng is None:
            os.environ.pop("MI_EDGETAPER_ENGINE", None)
        else:
            os.environ["MI_EDGETAPER_ENGINE"] = eng
        if eng == "direct" and shape[0] * shape[1] * shape[2] > 3e8 and kshape[0] > 30:
            continue                                     # (seconds)
        bl = bench.make_volume(shape, dev)
        decon.edgetaper_3d(bl, psf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            decon.edgetaper_3d(bl, psf)
        torch.cuda.synchronize()
        print(f"{shape[2]} x {shape[1]} x {shape[0]}, PSF {kshape[2]} x {kshape[1]} x {kshape[0]}: {eng or 'cost model':>10}: "
              f"{(time.perf_counter() - t0) / 3 * 1e3:7.2f} ms", flush=True)
        del bl
        torch.cuda.empty_cache()
