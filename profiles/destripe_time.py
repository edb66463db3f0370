"""Time of mi_destripe_z (filter_subband_3d_z) on block-sized volumes; prints ms and the effective GB/s over the 2 x volume
bytes the filter must move at least (read + write of the block)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ipp_amd import decon

dev = torch.device("cuda", 0)
for (z, y, x) in [(512, 512, 1024), (512, 2048, 2048)]:
    t = torch.rand((z, y, x), device=dev) + 0.5
    decon.filter_subband_3d_z(t, 2.0)          # warm-up: code objects, pool
    torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        decon.filter_subband_3d_z(t, 2.0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    gb = 2 * t.numel() * 4 / 1e9
    print(f"{x}x{y}x{z}: {ms:.1f} ms per block, {gb / ms * 1e3:.0f} GB/s of block read+write, {t.numel() / ms / 1e6:.2f} Gvoxel/s", flush=True)
    del t
    torch.cuda.empty_cache()
