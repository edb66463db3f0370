"""Compute-only cost of ONE rank of the C3 slab decomposition (no halo traffic), measured on a single GPU:
per-iteration time of forward_ratio + adjoint_update on the rank-local array for N = 1, 2, 4, 8 slabs.
    python profiles/slab_rank_cost.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ipp_amd import capi, decon, slab  # noqa: E402

dev = torch.device("cuda", 0)
vshape, kshape = bench.WORKLOADS["c3"]
psf = bench.make_psf(kshape)
for world in (1, 2, 4, 8):
    n_loc = vshape[1] // world
    sy = slab.psf_shift(vshape[1], kshape[1], "fft")
    h = max(sy, kshape[1] - 1 - sy)
    rows = capi.lib().mi_fft_good_size(n_loc + 2 * h, 1) if world > 1 else vshape[1]
    shape = (vshape[0], rows, vshape[2])
    shifts = (slab.psf_shift(vshape[2], kshape[2], "fft"), sy, slab.psf_shift(vshape[0], kshape[0], "fft"))
    ctx = decon.RLContext(shape, psf, None, boundary=(2, 2, 2), engine=capi.ENGINE_FFT, device=dev, shift_xyz=shifts)
    bl = torch.rand(shape, device=dev) + 0.1
    ratio = torch.empty_like(bl)
    for _ in range(2):
        ctx.forward_ratio(bl, ratio); ctx.adjoint_update(ratio, bl)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.forward_ratio(bl, ratio); ctx.adjoint_update(ratio, bl)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    halo_mb = 2 * h * vshape[0] * vshape[2] * 4 / 1e6
    print(f"N={world}: local rows {rows} (slab {n_loc} + 2x{h} halo), {ms:.2f} ms/iteration compute, "
          f"halo {halo_mb:.0f} MB per neighbour pair per exchange, 2 exchanges/iteration", flush=True)
    del ctx, bl, ratio
    torch.cuda.empty_cache()
