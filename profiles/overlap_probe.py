"""One-GPU contention timeline for the split x pass of the sharded RL step.

A rank of the C3 slab decomposition at N = 8 (512 x 288 x 2048 local array, 31-row PSF -> 15 halo rows) runs "part 2" of its
ratio step (every x tile that holds no edge row) while a stand-in for a collective's kernels -- `busy` work-groups of 256 threads
that hold their compute units for `us` microseconds on a second stream, issued first -- is resident.  Variants: static tile
stride (the single-GPU launch geometry), `free` compute units left unlaunched, tiles handed out by a device counter.

    python profiles/overlap_probe.py            # table
    rocprofv3 --kernel-trace ... -- python3 profiles/overlap_probe.py trace   # one launch per variant, for the timeline
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ipp_amd import capi, decon, slab  # noqa: E402

dev = torch.device("cuda", 0)
vshape, kshape = bench.WORKLOADS["c3"]
psf = bench.make_psf(kshape)
world = int(os.environ.get("PROBE_WORLD", "8"))
n_loc = vshape[1] // world
sy = slab.psf_shift(vshape[1], kshape[1], "fft")
h = max(sy, kshape[1] - 1 - sy)
rows = capi.lib().mi_fft_good_size(n_loc + 2 * h, 1)
shape = (vshape[0], rows, vshape[2])
shifts = (slab.psf_shift(vshape[2], kshape[2], "fft"), sy, slab.psf_shift(vshape[0], kshape[0], "fft"))
ctx = decon.RLContext(shape, psf, None, boundary=(2, 2, 2), engine=capi.ENGINE_FFT, device=dev, shift_xyz=shifts)
bl = torch.rand(shape, device=dev) + 0.1
edges = (h, 2 * h, n_loc, n_loc + h)
ctx.sharded_begin(bl)
ctx.sharded_ratio(bl)
trace = len(sys.argv) > 1 and sys.argv[1] == "trace"
reps = 1 if trace else 10
print(f"rank-local array {shape[2]} x {shape[1]} x {shape[0]} (N = {world}), edge rows {edges}; part 2 of the ratio step; "
      f"{reps} repetition(s) per line")
print(f"{'stand-in':>22} {'tiles':>8} {'free CUs':>8} {'x launch ms':>12} {'both done ms':>13}")
for busy, us in ((0, 0.0), (8, 300.0), (8, 1500.0), (32, 300.0), (32, 1500.0)):
    for dyn in (False, True):
        for free in (0, 8, 32):
            if busy == 0 and free not in (0, 32):
                continue
            ctx.set_overlap(free, dyn)
            x_ms, all_ms = ctx.overlap_probe(bl, edges, busy, us, reps)
            what = "none" if busy == 0 else f"{busy} WGs x {us:.0f} us"
            print(f"{what:>22} {'dynamic' if dyn else 'static':>8} {free:>8} {x_ms:>12.3f} {all_ms:>13.3f}", flush=True)
