"""Kernel breakdown helper: the face-slab edge taper of a C3 block, three calls (run under rocprofv3)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from ipp_amd import decon
dev = torch.device("cuda", 0)
shape, kshape = bench.WORKLOADS["c3"]
psf = torch.from_numpy(bench.make_psf(kshape)).to(dev)
os.environ["MI_EDGETAPER_ENGINE"] = "slabs"
bl = bench.make_volume(shape, dev)
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    decon.edgetaper_3d(bl, psf)
    torch.cuda.synchronize(); print("slabs edgetaper C3: %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
