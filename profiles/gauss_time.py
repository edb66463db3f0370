"""Times gauss3d_gpu on a C3-sized volume (the regularisation step sigma=0.5 and decwrap's default pre-filter)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ipp_amd import decon
dev = torch.device("cuda", 0)
x = torch.rand((512, 2048, 2048), device=dev)
for name, sig, ks in [("reg step sigma=0.5 (5 taps)", 0.5, None), ("pre-filter sigma=(0.5,0.5,2.5) size=(13,13,25)", [0.5, 0.5, 2.5], [13, 13, 25])]:
    decon.gauss3d_gpu(x, sig, ks)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        decon.gauss3d_gpu(x, sig, ks)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
    print(f"{name}: {ms:.2f} ms  ({x.numel() * 8 / ms / 1e6:.0f} GB/s on the 8 B/voxel ideal)")
