"""Compute-only cost of ONE rank of the C3 slab decomposition (no halo traffic), measured on a single GPU:
per-iteration time on the rank-local array for N = 1, 2, 4, 8 slabs, with the two protocols of slab.py: the fused steps
(sharded_ratio + sharded_update, halos as x-transformed rows) and forward_ratio + adjoint_update (halos as real rows).
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

    def timed(step, n=5):
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def unfused():
        ctx.forward_ratio(bl, ratio); ctx.adjoint_update(ratio, bl)

    def fused():
        ctx.sharded_ratio(bl); ctx.sharded_update(bl, True)

    def halos():  # pack 2 + unpack 2 halo blocks of the x-transformed buffer, twice per iteration
        for _ in range(2):
            a, b = ctx.spectrum_pack(n_loc, h), ctx.spectrum_pack(h, h)
            ctx.spectrum_unpack(a, 0, h); ctx.spectrum_unpack(b, h + n_loc, h)

    ms_u = timed(unfused)
    ctx.sharded_begin(bl)
    ms_f = timed(fused)
    ms_h = timed(halos) if world > 1 else 0.0
    halo_mb = h * vshape[0] * vshape[2] * 4 / 1e6
    print(f"N={world}: local rows {rows} (slab {n_loc} + 2x{h} halo): fused steps {ms_f:.2f} ms/iteration "
          f"(+ {ms_h:.2f} ms pack/unpack of spectrum halos), unfused {ms_u:.2f} ms; {halo_mb:.0f} MB per direction per neighbour "
          f"per exchange, 2 exchanges/iteration", flush=True)
    del ctx, bl, ratio
    torch.cuda.empty_cache()
