import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from ipp_amd import decon
dev = torch.device("cuda", 0)
shape, kshape = bench.WORKLOADS["c3"]
psf = torch.from_numpy(bench.make_psf(kshape)).to(dev)
decon.edgetaper_3d(bench.make_volume((64, 64, 64), dev), psf[:9, :9, :9].contiguous())  # loads the code objects
for eng in ("slabs", "fft", "slabs", "fft", "direct"):
    os.environ["MI_EDGETAPER_ENGINE"] = eng
    bl = bench.make_volume(shape, dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    decon.edgetaper_3d(bl, psf)
    torch.cuda.synchronize(); print(eng, "edgetaper C3: %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    del bl; torch.cuda.empty_cache()
